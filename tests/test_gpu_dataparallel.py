"""SURVEY.md 8(e) oracle on the PRODUCT path: two rank processes (fresh children, both on cuda:0, gloo for the
collectives) run ModelMeta.fused_train_step -- the bucketed SUM all-reduce launched from the backward pass's
`after_block` hook, the start-of-training parameter broadcast, the per-rank dropout streams -- and must end on the
parameters a single process reaches on the concatenated batch."""
import os
import socket
import subprocess
import sys

import pytest
import torch

from conftest import REPO

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(REPO, "tests"))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _run_ranks(tmp_path, F_, b_local, steps, live, world=2, kind="yolo"):
    out = str(tmp_path / "dp_rank0.pt")
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(REPO, "tests", "dp_worker.py"), out, str(F_),
                                       str(b_local), str(steps), str(int(live)), kind], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = [p.communicate(timeout=600)[0] for p in procs]
    for p, lg in zip(procs, logs):
        assert p.returncode == 0, lg[-3000:]
    return torch.load(out, weights_only=True)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("live_dropout", [False, True], ids=["injected_masks", "per_rank_dropout_streams"])
def test_two_ranks_fused_train_step_equals_single_process(tmp_path, live_dropout):
    import dp_worker
    F_, b_local, steps, world = 64, 2, 2, 2
    got = _run_ranks(tmp_path, F_, b_local, steps, live_dropout, world)
    # single process, concatenated batch, rank 0's initial weights
    params, grad, losses, enabled = dp_worker.run_rank(0, 1, F_, world * b_local, steps, live_dropout, param_seed=0)
    assert not enabled
    for a, b in zip(got["loss_sum"], losses):
        assert abs(a - b) <= 1e-5 * abs(b), (got["loss_sum"], losses)
    # STEP 1 (SURVEY.md 8e's oracle): identical parameters and masks on both sides, the forward is batch-invariant bit for
    # bit, so the all-reduced gradient differs from the single-process one by fp32 summation order only (2 + 2 images and
    # an all-reduce against 4; the weight-gradient slabs cut the batch at other places)
    g1_ref, g1_got = dp_worker.run_rank.grad_step1, got["grad_step1"]
    g1max = float(g1_ref.abs().max())
    assert float((g1_got - g1_ref).abs().max()) <= 2e-5 * g1max, float((g1_got - g1_ref).abs().max()) / g1max
    g_ref, g_got = grad, got["grad"]
    gmax = float(g_ref.abs().max())
    # step 2's all-reduced flat gradient, at the north star's 1e-4 only: step 2 starts from parameters that already differ where Adam's first update
    # flipped on a ~0 gradient (below)
    assert float((g_got - g_ref).abs().max()) <= 1e-4 * gmax
    lr = 1e-4
    off = 0
    for k, p_ref in params.items():
        p_got = got["params"][k]
        n = p_ref.numel()
        g1, g2 = g1_ref[off:off + n].view_as(p_ref), g_ref[off:off + n].view_as(p_ref)   # flat order = state_dict order
        off += n
        d = (p_got - p_ref).abs()
        # Adam's first updates are ~lr*sign(g): an entry whose gradient is ~0 may flip the sign of its update (at most
        # 2 lr per step apart) ...
        assert float(d.max()) <= 2.1 * lr * steps, k
        tight = d <= 1e-6 * max(1.0, float(p_ref.abs().max()))
        # ... and EVERY entry whose gradient is clear of zero in both steps agrees to fp32 rounding (1e-6 relative,
        # SURVEY.md 8e); how many entries sit near zero is a property of the batch, not of the arithmetic
        # (step 2 moves a parameter along m2 ~ 0.9 g1 + g2: that combination must be clear of zero as well)
        m2 = 0.9 * g1 + g2
        clear = (g1.abs() > 1e-3 * float(g1.abs().max())) & (g2.abs() > 1e-3 * float(g2.abs().max())) & \
                (m2.abs() > 1e-2 * float(m2.abs().max()))
        assert int(clear.sum()) >= 0.1 * n or n < 64, (k, int(clear.sum()), n)      # the check is not vacuous
        assert bool(tight[clear].all()), (k, int((~tight[clear]).sum()), int(clear.sum()))
        assert float(tight.float().mean()) >= 0.99, (k, float(tight.float().mean()))


@pytest.mark.timeout(900)
@pytest.mark.parametrize("live_dropout", [False, True], ids=["injected_masks", "per_rank_dropout_streams"])
def test_two_ranks_ssd_fused_train_step_equals_single_process(tmp_path, live_dropout):
    """SSD: `ssd_loss` divides by the positive count of the WHOLE batch (losses/SSDLoss.py:86).  Two ranks (one of them
    holding an image without faces) exchange the three fp64 batch sums before the backward pass and the flat gradient
    after it; loss, gradient and parameters equal the single-process step on the concatenated batch."""
    import dp_worker
    b_local, steps, world = 2, 2, 2
    got = _run_ranks(tmp_path, 16, b_local, steps, live_dropout, world, kind="ssd")
    params, grad, losses, enabled = dp_worker.run_rank_ssd(0, 1, world * b_local, steps, live_dropout, param_seed=0)
    assert not enabled
    for a, b in zip(got["loss_sum"], losses):
        assert abs(a - b) <= 1e-5 * abs(b), (got["loss_sum"], losses)
    gmax = float(grad.abs().max())
    # the SECOND step's flat gradient: it is taken at parameters that may already differ in a few entries by Adam's
    # sign-like first update (2 lr), and 2 + 2 images sum in another order than 4: within the north star's 1e-4
    assert float((got["grad"] - grad).abs().max()) <= 1e-4 * gmax
    lr = 1e-4
    for k, p_ref in params.items():
        d = (got["params"][k] - p_ref).abs()
        assert float(d.max()) <= 2.1 * lr * steps, k
        frac_tight = float((d <= 1e-6 * max(1.0, float(p_ref.abs().max()))).float().mean())
        assert frac_tight >= 0.995, (k, frac_tight)
