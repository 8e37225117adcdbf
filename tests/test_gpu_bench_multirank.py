"""Rehearsal of the driver's multi-GPU bench on ONE GPU: `python bench.py --gpus 2` through its own self-launch (a
`torch.distributed.run` child started before anything touches the GPU), both ranks on cuda:0
(FDET_SINGLE_DEVICE=1) with gloo carrying the collectives (FDET_DIST_BACKEND=gloo) -- so that the first real SCALE run
cannot fail on plumbing: rank environment, process group, parameter broadcast, the bucketed all-reduce inside the
backward pass, max-over-ranks timing, one JSON line from rank 0.  The whole-job loss after the last step must equal the
one-process run on the concatenated batch."""
import json
import os
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


def _bench(args, extra_env=None):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args + ["--no-cpu-baseline", "--no-inference",
                       "--no-configs", "--no-feed", "--no-p16"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                 # exactly ONE JSON line, from rank 0
    return json.loads(lines[0])


@pytest.mark.timeout(1200)
def test_bench_two_ranks_on_one_gpu_equals_one_process_on_the_concatenated_batch():
    two = _bench(["--gpus", "2", "--batch", "4", "--steps", "2", "--warmup", "1"],
                 {"FDET_DIST_BACKEND": "gloo", "FDET_SINGLE_DEVICE": "1"})
    assert two["n_gpus"] == 2 and two["n_ranks_seen"] == 2
    assert two["config"]["global_batch"] == 8 and two["config"]["parallelism"] == "dp2"
    assert two["scaling"] == "weak" and two["steps"] == 2 and two["warmup"] == 1
    assert "allreduce_ms_exposed" in two and two["allreduce_ms_exposed"] >= 0.0 and two["allreduce_backend"] == "gloo"
    assert "roofline" in two and two["value"] > 0
    one = _bench(["--gpus", "1", "--batch", "8", "--concat-ranks", "2", "--steps", "2", "--warmup", "1"])
    assert one["n_ranks_seen"] == 1 and "allreduce_ms_exposed" not in one
    # the loss of the THIRD optimisation step (1 warm-up + 2 timed): both jobs drew the same images, the same dropout
    # masks (global-image counters) and made the same two updates (up to Adam's sign-like first steps on ~0 gradients)
    assert abs(two["final_loss"] - one["final_loss"]) <= 1e-4 * abs(one["final_loss"]), (two["final_loss"], one["final_loss"])
