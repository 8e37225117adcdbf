#!/usr/bin/env python3
"""tests/golden/g9_ssd.npz: SSD detection math fixtures produced by RUNNING THE REFERENCE on the CPU
(build container only):   python tools/make_goldens_ssd.py

Imported from /root/reference and executed as-is: losses/SSDLoss.py (ssd_loss, hard_negative_mining),
datasets/WIDERFace/dataset_ssd.py::WIDERFaceDatasetSSD.convert_bbx_to_feature_map (by path),
datasets/utils.py::ReduceSSDBoundingBoxes (torchvision stubbed as in tools/make_goldens.py, so the
NMS inside it is this repo's restatement: only the pre-NMS decode is reference-pinned)."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_goldens as MG  # noqa: E402

R = MG.import_reference()
REF = MG.REF
ssdloss = MG._load_by_path("ref_ssdloss", os.path.join(REF, "losses", "SSDLoss.py"))
dssd = MG._load_by_path("ref_dataset_ssd", os.path.join(REF, "datasets", "WIDERFace", "dataset_ssd.py"))
utils = R["utils"]
PS = (60, 30, 15, 7)
P = sum(p * p for p in PS)
size = 480
ds = dssd.WIDERFaceDatasetSSD(None, None, (size, size), targets=[], transform=None)
g = torch.Generator().manual_seed(31)


def boxes(n):
    rows = []
    for _ in range(n):
        w = int(torch.randint(4, 200, (1,), generator=g)); h = int(torch.randint(4, 200, (1,), generator=g))
        x = int(torch.randint(0, size - w, (1,), generator=g)); y = int(torch.randint(0, size - h, (1,), generator=g))
        rows.append([1.0, x, y, w, h])
    return torch.tensor(rows, dtype=torch.float32) if rows else torch.tensor([])


out = {}
# ---- encode: several box lists incl. empty, boundary, collisions
lists = [boxes(0), boxes(1), boxes(3), boxes(12), torch.tensor([[1., 479, 0, 1, 1], [1., 0, 479, 5, 1], [1., 100, 100, 50, 50], [1., 101, 101, 60, 40]])]
enc = []
for k, b in enumerate(lists):
    fms = [ds.convert_bbx_to_feature_map(b, (size, size), ps).permute(1, 2, 0).reshape(-1, 5) for ps in PS]
    enc.append(torch.cat(fms, 0))
    out[f"enc_boxes_{k}"] = b if b.numel() else torch.zeros(0, 5)
out["enc"] = torch.stack(enc)
# ---- loss + grads (autograd of the reference's ssd_loss), B images
B = 4
y = torch.stack([enc[2], enc[3], enc[4], enc[1]])                       # targets (B,P,5)
pred = torch.rand(B, P, 5, generator=g) * 0.98 + 0.01
pred[0, :50, 0] = 1e-9                                                  # below the BCE clamp
pred[1, 100:120, 0] = 1.0 - 1e-9
c = pred[:, :, 0].clone().requires_grad_(True); l = pred[:, :, 1:].clone().requires_grad_(True)
loss = ssdloss.ssd_loss(c, l, y[:, :, 0], y[:, :, 1:], 10)
gc, gl = torch.autograd.grad(loss, [c, l])
out.update(loss_pred=pred, loss_y=y, loss=loss.detach(), loss_gc=gc, loss_gl=gl)
with torch.no_grad():
    mask = ssdloss.hard_negative_mining(-torch.log(pred[:, :, 0]), y[:, :, 0], 10)
out["mask"] = mask.to(torch.uint8)
# ---- decode (pre-NMS pieces are reference code) and the full reducer
red = utils.ReduceSSDBoundingBoxes(0.5, 0.5, (3, size, size), PS, with_priors=True)
dec_in = torch.stack([enc[2], enc[3], torch.rand(P, 5, generator=g) * torch.tensor([0.52, 1, 1, 0.3, 0.3])])
scaled = [red.scale_batch_bbx_xywh(x.clone()) for x in dec_in]
out["dec_in"] = dec_in
out["dec_scaled"] = torch.stack(scaled)
full = [red(x.clone()) for x in dec_in]
out["dec_counts"] = torch.tensor([f.shape[0] for f in full])
K = max(f.shape[0] for f in full)
pad = torch.zeros(len(full), K, 5)
for i, f in enumerate(full):
    pad[i, : f.shape[0]] = f
out["dec_out"] = pad
MG.save("g9_ssd", **out)

# ----------------------------------------------------------------------------- SSD model (models/SSD.py)
import types  # noqa: E402
import oracle as O  # noqa: E402
from oracle import ssd_model_oracle as SM  # noqa: E402
ptf = types.ModuleType("ptflops"); ptf.get_model_complexity_info = lambda *a, **k: ("", "")
sys.modules["ptflops"] = ptf
bsm = MG._load_by_path("models.BaseSSDModel", os.path.join(REF, "models", "BaseSSDModel.py"))
sys.modules["models"].BaseSSDModel = bsm
ssdmod = MG._load_by_path("models.SSD", os.path.join(REF, "models", "SSD.py"))
FIL, SEED = 16, 5
torch.manual_seed(SEED)
ref_model = ssdmod.SSD(filters=FIL, input_shape=(3, size, size)).eval()
sd = {k: v for k, v in ref_model.state_dict().items()}
Pm = SM.init_params(FIL, SEED)
assert set(Pm) == set(sd), (sorted(set(Pm) ^ set(sd)))
for k in sd:
    assert torch.equal(Pm[k], sd[k]), k                                # the seed reproduces the reference's init
gx = torch.Generator().manual_seed(9)
xm = torch.rand(2, 3, size, size, generator=gx)
with torch.no_grad():
    ym = ref_model(xm)
tgt = torch.stack([enc[2], enc[3]])
for p_ in ref_model.parameters():
    p_.requires_grad_(True)
yg = ref_model(xm)
lm = ssdloss.ssd_loss(yg[:, :, 0], yg[:, :, 1:], tgt[:, :, 0], tgt[:, :, 1:], 10)
lm.backward()
gout = {"m_x_seed": torch.tensor(9), "m_seed": torch.tensor(SEED), "m_filters": torch.tensor(FIL), "m_y": ym, "m_target": tgt,
        "m_loss": lm.detach()}
for n_, p_ in ref_model.named_parameters():
    gout["m_gsum/" + n_] = p_.grad.double().sum().float()
    gout["m_gabs/" + n_] = p_.grad.double().abs().sum().float()
    if p_.grad.numel() <= 4096:
        gout["m_grad/" + n_] = p_.grad.clone()
MG.save("g10_ssd_model", **gout)
