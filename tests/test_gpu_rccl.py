"""RCCL on the one GPU of the test box: the data-parallel product path (ModelMeta.fused_train_step /
ModelMetaSSD.fused_train_step) inside a one-rank `nccl` process group with the gradient exchange forced on must leave
the parameters bit-identical to a run without any process group.  The gloo tests (tests/test_dataparallel_gloo.py,
tests/test_gpu_dataparallel.py) prove the N-rank arithmetic; this one proves that the same calls work against RCCL:
init with device_id, broadcast + checksum, async all-reduce on slice views launched from `after_block`, stream-ordered
wait before Adam.  Each run is a fresh child process (a process group is process-global state)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(kind, dp, out):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(HERE, "rccl_worker.py"), kind, "1" if dp else "0", str(out)],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return torch.load(str(out), weights_only=True)


@pytest.mark.parametrize("kind", ["yolo", "ssd"])
def test_one_rank_rccl_group_equals_no_group(kind, tmp_path):
    ref = _run(kind, False, tmp_path / "ref.pt")
    got = _run(kind, True, tmp_path / "dp.pt")
    assert got["info"]["backend"] == "nccl" and got["info"]["world"] == 1
    assert got["info"]["reducer_enabled"] and not ref["info"]["reducer_enabled"]
    assert got["info"]["losses"] == ref["info"]["losses"]
    assert torch.equal(got["flat"], ref["flat"])
