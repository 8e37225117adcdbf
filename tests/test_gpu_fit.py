"""`trainer.fit` -- the counterpart of train_model.py:27-61 (Trainer.fit + to_torchscript) -- on a fixed synthetic set:
  (1) the first 10 optimisation steps follow the CPU oracle's (`oracle.train_step`: forward + batch-SUM YoloLoss + autograd +
      Adam on the same uint8 frames, targets and dropout masks): 1e-4 relative in the loss at step 0, 3e-3 after Adam updates;
  (2) over a 60-step run with live dropout the loss of the training set, evaluated without dropout after every epoch by the
      validation hook, goes down epoch after epoch (the per-step training loss carries the Dropout2d noise of 4-image batches);
  (3) the validation hook runs, MultiStepLR is stepped once per epoch, the TorchScript export loads and reproduces the
      eager boxes."""
import pytest
import torch

import oracle as O

pytestmark = pytest.mark.gpu


def _batches(n, B, size, S, seed):
    g = torch.Generator().manual_seed(seed)
    out = []
    for b in range(n):
        x = torch.randint(0, 256, (B, 3, size, size), dtype=torch.uint8, generator=g)
        boxes = O.synthetic_boxes(B, size, seed=seed * 100 + b)
        y = torch.stack([O.encode_targets(bb, (size, size), S) for bb in boxes])
        out.append((x, y, boxes))
    return out


def test_fit_follows_oracle_then_loss_goes_down(tmp_path):
    import fdet_amd
    from fdet_amd.models import ModelMeta
    from fdet_amd.models.PoolResnet import PoolResnet
    from fdet_amd.trainer import fit
    F_, size, S, B, lr0, lr = 64, 480, 10, 4, 1e-4, 3e-4     # the reference's learning rate for the oracle comparison
    spec = O.poolresnet_spec(F_, (3, size, size), S)
    P0 = O.init_params(spec, seed=7)
    masks = O.make_dropout_masks(spec, B, seed=8)
    train = _batches(10, B, size, S, seed=1)

    def build(lr_):
        model = PoolResnet(F_, (3, size, size), S).cuda()
        model.load_state_dict({k: v.clone() for k, v in P0.items()})
        model.set_dropout_masks(masks)
        return ModelMeta(model=model, lr=lr_, log_path=tmp_path / "out.log")

    # (1) ten steps against the oracle
    mm = build(lr0)
    got = []
    fit(mm, train, None, epochs=1, on_step=lambda i, tr, o: got.append(float(o["loss"])))
    P = {k: v.clone() for k, v in P0.items()}
    state = {"exp_avg": {k: torch.zeros_like(v) for k, v in P.items()}, "exp_avg_sq": {k: torch.zeros_like(v) for k, v in P.items()}}
    torch.set_num_threads(16)
    for t, (x, y, _) in enumerate(train):
        loss, _, _ = O.train_step(spec, P, state, t + 1, x.float() / 255.0, y, masks, lr=lr0)
        # Step 0 is pure forward + loss parity (1e-4, the north star's tolerance).  From step 1 on the parameters carry
        # Adam updates, and Adam's first steps are +-lr for EVERY entry whatever its gradient's magnitude (m / sqrt(v) =
        # sign(g) while the moments hold one sample): entries whose gradient is ~0 move the other way at any rounding
        # difference -- a different summation order in a weight-gradient kernel is enough -- so two correct
        # implementations drift apart by ~1e-4 per step in the loss.  Per-entry gradient parity is tests/test_gpu_model.py's job.
        tol = 1e-4 if t == 0 else 3e-3
        assert abs(got[t] - float(loss)) <= tol * abs(float(loss)), (t, got[t], float(loss))
    # (2) + (3): 6 epochs x 10 steps from scratch with live dropout, validation every epoch, export at the end
    mm = build(lr)
    mm.model.set_dropout_masks(None)
    path = tmp_path / "model.pt"
    hist = fit(mm, train, train, epochs=6, torchscript_path=str(path))     # validation set = training set, eval mode
    means = [float(m["loss"]) for m in hist["val"]]
    # live dropout makes the trajectory stochastic: an epoch may stall (never rise by more than 1 %), the run must descend
    assert all(b < 1.01 * a for a, b in zip(means, means[1:])), means
    assert means[-1] < 0.93 * means[0], means
    assert len(hist["train"]) == 6 and all(torch.isfinite(m["loss"]) for m in hist["train"])
    assert mm.opt.param_groups[0]["lr"] == lr                       # six scheduler steps, milestone at 40
    assert "training, loss" in (tmp_path / "out.log").read_text()
    scripted = torch.jit.load(str(path))
    mm.model.eval()
    u8 = train[0][0][:2].cuda()
    with torch.no_grad():
        eager = mm.model(u8, predict=torch.tensor(1))
        again = scripted(u8, predict=torch.tensor(1))
    assert torch.equal(eager.cpu(), again.cpu())
