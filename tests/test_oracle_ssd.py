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
