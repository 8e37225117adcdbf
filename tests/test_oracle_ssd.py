"""SSD detection-math oracle (oracle/ssd_oracle.py) against fixtures produced by the imported
reference (tools/make_goldens_ssd.py): multi-scale encode bit-exact, hard-negative mask exact,
loss and autograd gradients within 1e-5, decode with priors bit-exact, reducer output exact."""
import torch

import oracle as O
from oracle import ssd_oracle as S

SIZE = 480


def test_ssd_encode_bit_exact(golden):
    g = golden("g9_ssd")
    for k in range(g["enc"].shape[0]):
        b = g[f"enc_boxes_{k}"]
        b = b if b.numel() else torch.tensor([])
        assert torch.equal(S.ssd_encode(b, (SIZE, SIZE)), g["enc"][k]), k


def test_ssd_loss_mask_and_grads(golden):
    g = golden("g9_ssd")
    pred, y = g["loss_pred"], g["loss_y"]
    mask = S.hard_negative_mining(-torch.log(pred[:, :, 0]), y[:, :, 0], 10)
    assert torch.equal(mask.to(torch.uint8), g["mask"])
    loss, gc, gl = S.ssd_loss_and_grads(pred[:, :, 0], pred[:, :, 1:], y[:, :, 0], y[:, :, 1:], 10)
    assert abs(float(loss) - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    assert torch.allclose(gc, g["loss_gc"], rtol=1e-5, atol=1e-9)
    assert torch.allclose(gl, g["loss_gl"], rtol=1e-5, atol=1e-9)


def test_ssd_decode_and_reduce(golden):
    g = golden("g9_ssd")
    for n in range(g["dec_in"].shape[0]):
        x = g["dec_in"][n]
        # the scaled map (reference scale_batch_bbx_xywh) is what the oracle thresholds and rounds
        sc = g["dec_scaled"][n]
        scores, bbx = S.ssd_decode_pre_nms(x, 0.5, (3, SIZE, SIZE))
        i = torch.where(sc[:, 0] > 0.5)[0]
        assert torch.equal(scores, sc[i, 0])
        ref = sc[i].clone(); ref[:, 3] += ref[:, 1]; ref[:, 4] += ref[:, 2]
        assert torch.equal(bbx, torch.round(ref[:, 1:]))
        out = S.reduce_ssd_bounding_boxes(x, 0.5, 0.5, (3, SIZE, SIZE))
        k = int(g["dec_counts"][n])
        assert out.shape[0] == k
        assert torch.equal(out, g["dec_out"][n, :k])


def test_ssd_model_oracle_matches_reference_class(golden):
    """oracle/ssd_model_oracle.py against the reference's SSD class (filters 16, 2 images, eval mode):
    forward output, ssd_loss and the gradient of every parameter (sum / abs-sum, small tensors in full).
    The parameters are regenerated from the seed (the generator asserted that this reproduces the
    reference's default init bit for bit)."""
    from oracle import ssd_model_oracle as SM
    g = golden("g10_ssd_model")
    fil, seed = int(g["m_filters"]), int(g["m_seed"])
    P = SM.init_params(fil, seed)
    x = torch.rand(2, 3, SIZE, SIZE, generator=torch.Generator().manual_seed(int(g["m_x_seed"])))
    loss, y, G = SM.loss_and_grads(fil, P, x, g["m_target"])
    assert torch.allclose(y, g["m_y"], rtol=1e-5, atol=1e-6)
    assert abs(float(loss) - float(g["m_loss"])) <= 1e-5 * abs(float(g["m_loss"]))
    for n, gr in G.items():
        ga = float(g["m_gabs/" + n])
        assert abs(float(gr.double().abs().sum()) - ga) <= 1e-4 * max(ga, 1e-6), n
        assert abs(float(gr.double().sum()) - float(g["m_gsum/" + n])) <= 1e-4 * max(ga, 1e-6), n
        if ("m_grad/" + n) in g:
            assert torch.allclose(gr, g["m_grad/" + n], rtol=1e-4, atol=1e-7 * max(1.0, ga)), n
