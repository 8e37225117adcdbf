"""Parity at BASELINE.json's FULL size (PoolResnet-medium, 3x480x480, 256 images) through
size-independent properties: an image's output does not depend on its neighbours in the batch
(subset vs oracle), a batch is the concatenation of its halves (bit-exact forward, gradients add
up), and the per-image detection math is checked against the oracle for every image."""
import pytest
import torch

import oracle as O

pytestmark = pytest.mark.gpu

B, F_, SIZE, S = 256, 64, 480, 10


@pytest.fixture(scope="module")
def setup():
    import fdet_amd
    from fdet_amd.models.PoolResnet import PoolResnet
    spec = O.poolresnet_spec(F_, (3, SIZE, SIZE), S)
    P = O.init_params(spec, seed=7)
    model = PoolResnet(filters=F_, input_shape=(3, SIZE, SIZE), num_of_patches=S)
    model.load_state_dict({k: v.clone() for k, v in P.items()})
    model = model.cuda()
    g = torch.Generator().manual_seed(21)
    x = torch.rand(B, 3, SIZE, SIZE, generator=g)
    masks = O.make_dropout_masks(spec, B, seed=5)
    return spec, P, model, x, masks


def test_subset_of_full_batch_matches_oracle(setup):
    spec, P, model, x, masks = setup
    sub = [0, 17, 128, 255]
    model.eval()
    with torch.no_grad():
        y = model(x.cuda()).cpu()
    y_ref = O.model_forward(spec, P, x[sub], None)
    assert torch.allclose(y[sub], y_ref, atol=1e-4)
    model.train()
    model.set_dropout_masks(masks)
    with torch.no_grad():
        yt = model(x.cuda()).cpu()
    yt_ref = O.model_forward(spec, P, x[sub], {k: v[sub] for k, v in masks.items()})
    assert torch.allclose(yt[sub], yt_ref, atol=1e-4)
    model.set_dropout_masks(None)


def test_batch_is_concatenation_of_its_halves(setup):
    """Forward bit-exact; parameter gradients of the whole batch = sum over the halves (fp32
    summation order differs in the slab reductions: 1e-4 of each tensor's scale)."""
    spec, P, model, x, masks = setup
    from fdet_amd import hotpath as hp
    eng = model.engine
    names, params = model.named_stack_params()
    Pd = {n: p.data for n, p in zip(names, params)}
    boxes = O.synthetic_boxes(B, SIZE, seed=3)
    y = hp.encode_targets(boxes, (SIZE, SIZE), S)
    xd = x.cuda()
    md = {k: v.cuda() for k, v in masks.items()}

    def run(sl):
        m_ = {k: v[sl].contiguous() for k, v in md.items()}
        yh, saved = eng.forward(xd[sl].contiguous(), Pd, m_, save=True)
        lpi, lsum, dy = hp.yolo_loss_fwd_bwd(yh, y[sl].contiguous(), want_grad=True)
        G = {n: torch.empty_like(p) for n, p in Pd.items()}
        eng.backward(saved, dy, Pd, G)
        return yh, lpi, lsum, G

    y_all, lpi_all, lsum_all, G_all = run(slice(0, B))
    y_a, lpi_a, lsum_a, G_a = run(slice(0, B // 2))
    y_b, lpi_b, lsum_b, G_b = run(slice(B // 2, B))
    assert torch.equal(y_all, torch.cat([y_a, y_b]))
    assert torch.equal(lpi_all, torch.cat([lpi_a, lpi_b]))
    assert abs(float(lsum_all) - float(lsum_a) - float(lsum_b)) <= 1e-4 * max(1.0, abs(float(lsum_all)))
    assert abs(float(lsum_all) - float(lpi_all.double().sum())) <= 1e-4 * max(1.0, abs(float(lsum_all)))
    for n in names:
        tot = G_a[n].double() + G_b[n].double()
        scale = max(1e-6, float(tot.abs().max()))
        assert float((G_all[n].double() - tot).abs().max()) <= 1e-4 * scale, n


def test_detection_math_on_every_image_of_a_full_batch():
    """encode -> (perturb) -> decode + NMS for 256 images in one launch each, against the oracle per image:
    grid indices, rounded boxes and keep order bit-exact; loss per image within 1e-4."""
    from fdet_amd import hotpath as hp
    boxes = O.synthetic_boxes(B, SIZE, seed=11)
    enc = hp.encode_targets(boxes, (SIZE, SIZE), S).cpu()
    for i in range(B):
        assert torch.equal(enc[i], O.encode_targets(boxes[i], (SIZE, SIZE), S))
    g = torch.Generator().manual_seed(4)
    maps = torch.rand(B, 5, S, S, generator=g)
    rows, counts = hp.reduce_bounding_boxes(maps.cuda(), 0.7, 0.3, float(SIZE), float(SIZE))
    rows, counts = rows.cpu(), counts.cpu()
    red = O.ReduceBoundingBoxes(0.7, 0.3, (3, SIZE, SIZE), S)
    for i in range(B):
        ref = red(maps[i])
        assert int(counts[i]) == ref.shape[0]
        assert torch.equal(rows[i, : ref.shape[0]], ref)
    lpi, lsum, _ = hp.yolo_loss_fwd_bwd(maps.cuda(), enc.cuda(), want_grad=False)
    ref = torch.stack([O.yolo_loss(maps[i], enc[i]) for i in range(B)])
    assert torch.allclose(lpi.cpu(), ref, rtol=1e-4, atol=1e-4)
    # round trip (dataset.py:125-139): decoding an encoded map returns the boxes of distinct cells
    r2, c2 = hp.reduce_bounding_boxes(enc.cuda(), 0.5, 0.99, float(SIZE), float(SIZE))
    for i in range(0, B, 16):
        got = r2[i, : int(c2[i])].cpu()
        want = O.ReduceBoundingBoxes(0.5, 0.99, (3, SIZE, SIZE), S)(enc[i])
        assert torch.equal(got, want)
