"""Host-side facts the training loop needs: how many CPU cores this process is actually granted.

torch sizes its intra-op pool from os.cpu_count().  On a container that exposes 256 logical CPUs but grants a 16-core
cgroup share (the 1-GPU boxes this engine is measured on) that means 256 spinning OpenMP workers on 16 cores: every
parallel CPU op (the 177 MB copy of a uint8 batch into pinned memory, even a 128 K-element target copy) costs
milliseconds and the thread that enqueues the GPU kernels is descheduled -- measured: a trainer.fit() step of 14-19 ms
against 3.7 ms once the pool matches the share (tools/probe/fit_time2.py)."""
import os

import torch


def effective_cpus() -> int:
    """min(os.cpu_count(), scheduler affinity, cgroup CPU quota)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:                                                   # cgroup v2
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
        return n
    except (OSError, ValueError, IndexError):
        pass
    try:                                                   # cgroup v1
        quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if quota > 0:
            n = min(n, max(1, int(quota / period + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def limit_host_threads() -> int:
    """Shrink torch's intra-op pool to the granted cores (never grows it).  Returns the thread count in force."""
    want = effective_cpus()
    if torch.get_num_threads() > want:
        torch.set_num_threads(want)
    return torch.get_num_threads()
