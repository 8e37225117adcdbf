"""BASELINE.json config 3 at ITS OWN size: Resnet backbone, filters 64, 3x640x640, S=20 (maps 320 -> 160 -> 80 -> 40 -> 20:
the column-segmented bf16x3 kernels, the scalar-fed 3x3/s2 stem, four pooled blocks through the separate tail
kernels).  (a) two images against the oracle: forward, loss (1e-4) and every parameter gradient (L2, see
tests/test_gpu_model.py::test_gradient_entries_differ_only_through_pool_routing for why not per entry);
(b) the per-GPU batch of the 8-GPU configuration (32 images): a batch is the concatenation of its halves."""
import pytest
import torch

import oracle as O

pytestmark = pytest.mark.gpu

F_, SIZE, S, NB = 64, 640, 20, 10


@pytest.fixture(scope="module")
def setup():
    import fdet_amd  # noqa: F401
    from fdet_amd.models.Resnet import Resnet
    spec = O.resnet_spec(F_, (3, SIZE, SIZE), S, NB)
    P = O.init_params(spec, seed=9)
    model = Resnet(filters=F_, input_shape=(3, SIZE, SIZE), num_of_patches=S, num_of_residual_blocks=NB)
    model.load_state_dict({k: v.clone() for k, v in P.items()})
    return spec, P, model.cuda().train()


@pytest.mark.timeout(900)
def test_resnet64_640_two_images_vs_oracle(setup):
    from fdet_amd import hotpath as hp
    spec, P, model = setup
    B = 2
    torch.set_num_threads(16)
    x = torch.rand(B, 3, SIZE, SIZE, generator=torch.Generator().manual_seed(1))
    boxes = O.synthetic_boxes(B, SIZE, seed=2)
    y = torch.stack([O.encode_targets(b, (SIZE, SIZE), S) for b in boxes])
    masks = O.make_dropout_masks(spec, B, seed=3)
    leaves = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    y_ref = O.model_forward(spec, leaves, x, masks)
    loss_ref = O.batch_loss(y_ref, y)
    G_ref = dict(zip(leaves, torch.autograd.grad(loss_ref, list(leaves.values()))))
    eng = model.engine
    assert eng.x3
    names, params = model.named_stack_params()
    Pd = {n: p.data for n, p in zip(names, params)}
    y_hat, saved = eng.forward(x.cuda(), Pd, {k: v.cuda() for k, v in masks.items()}, save=True)
    lpi, lsum, dy = hp.yolo_loss_fwd_bwd(y_hat, y.cuda(), want_grad=True)
    assert tuple(y_hat.shape) == (B, 5, S, S)
    assert torch.allclose(y_hat.cpu(), y_ref.detach(), atol=1e-4)
    assert abs(float(lsum) - float(loss_ref)) <= 1e-4 * max(1.0, float(loss_ref))
    G = {n: torch.empty_like(p) for n, p in Pd.items()}
    eng.backward(saved, dy, Pd, G)
    for n in names:
        got, ref = G[n].cpu().double(), G_ref[n].double()
        rel_l2 = float((got - ref).norm() / ref.norm().clamp_min(1e-30))
        assert rel_l2 <= 5e-3, (n, rel_l2)
    # eval mode as well (no dropout): the demo path's conv stack at this size
    model.eval()
    with torch.no_grad():
        ye = model(x.cuda()).cpu()
        ye_ref = O.model_forward(spec, P, x, None)
    model.train()
    assert torch.allclose(ye, ye_ref, atol=1e-4)


@pytest.mark.timeout(900)
def test_resnet64_640_batch32_is_concatenation_of_its_halves(setup):
    from fdet_amd import hotpath as hp
    spec, P, model = setup
    B = 32                                                   # per-GPU batch of the 8-GPU configuration
    eng = model.engine
    names, params = model.named_stack_params()
    Pd = {n: p.data for n, p in zip(names, params)}
    x = torch.rand(B, 3, SIZE, SIZE, generator=torch.Generator().manual_seed(5)).cuda()
    y = hp.encode_targets(O.synthetic_boxes(B, SIZE, seed=6), (SIZE, SIZE), S)
    md = {k: v.cuda() for k, v in O.make_dropout_masks(spec, B, seed=7).items()}

    def run(sl):
        m_ = {k: v[sl].contiguous() for k, v in md.items()}
        yh, saved = eng.forward(x[sl].contiguous(), Pd, m_, save=True)
        lpi, lsum, dy = hp.yolo_loss_fwd_bwd(yh, y[sl].contiguous(), want_grad=True)
        G = {n: torch.empty_like(p) for n, p in Pd.items()}
        eng.backward(saved, dy, Pd, G)
        del saved
        return yh, lpi, float(lsum), {n: v.double().cpu() for n, v in G.items()}

    y_all, lpi_all, ls_all, G_all = run(slice(0, B))
    y_a, lpi_a, ls_a, G_a = run(slice(0, B // 2))
    y_b, lpi_b, ls_b, G_b = run(slice(B // 2, B))
    assert bool(torch.isfinite(y_all).all())
    assert torch.equal(y_all, torch.cat([y_a, y_b]))
    assert torch.equal(lpi_all, torch.cat([lpi_a, lpi_b]))
    assert abs(ls_all - ls_a - ls_b) <= 1e-4 * max(1.0, abs(ls_all))
    for n in names:
        tot = G_a[n] + G_b[n]
        scale = max(1e-6, float(tot.abs().max()))
        assert float((G_all[n] - tot).abs().max()) <= 1e-4 * scale, n
