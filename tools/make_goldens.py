#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE'S OWN PYTHON on the CPU.

Run in the build container only (needs /root/reference, which never travels to the GPU
box):      python tools/make_goldens.py

What is imported from the reference (executed as-is, nothing copied into this repo):
  losses/YoloLoss.py::yolo_loss                    (torch only)
  datasets/WIDERFace/dataset.py::WIDERFaceDataset.convert_bbx_to_feature_map (by path)
  datasets/utils.py::ReduceBoundingBoxes           (needs torchvision -> stub below)
  models/BaseModel.py, models/PoolResnet.py, models/Resnet.py (stubs below)

Third-party modules absent from this image are replaced by minimal stand-ins in
sys.modules (torchvision, albumentations, torchinfo, pytorch_lightning).  The stand-in
`torchvision.ops.nms` / `box_iou` are THIS repo's oracle restatement, therefore fixtures
that pass through NMS / box_iou do NOT pin those two functions (parity unpinned for
them); every other number in the fixtures is produced by reference code.

The trained-weight fixture is made WITHOUT executing anything from the archive: the
TorchScript zip's raw tensor storages (`<name>/data/N`) are read as bytes
(torch.load(weights_only=True) refuses TorchScript archives; torch.jit.load is not used).
"""
import importlib.util
import io
import os
import sys
import types
import zipfile

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)
import oracle as O  # noqa: E402  (only for the nms/box_iou stand-ins and mask helper)


# ----------------------------------------------------------------------------- stubs
def _install_stubs():
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    tvtt = types.ModuleType("torchvision.transforms.transforms")
    tvo = types.ModuleType("torchvision.ops")

    class Resize(torch.nn.Module):          # tensor path of torchvision 0.11.2 Resize
        def __init__(self, size):
            super().__init__()
            self.size = tuple(size)

        def forward(self, x):
            return (O.preprocess_u8(x, self.size) * 255.0).round().to(torch.uint8) \
                if x.dtype == torch.uint8 else torch.nn.functional.interpolate(
                    x if x.dim() == 4 else x[None], size=self.size, mode="bilinear", align_corners=False)

    class ToPILImage:
        def __call__(self, x):
            raise RuntimeError("not used")

    tvtt.Resize = Resize
    tvtt.ToPILImage = ToPILImage
    tvt.transforms = tvtt
    tvt.Resize = Resize
    tvt.ToPILImage = ToPILImage
    tvo.nms = O.nms
    tvo.box_iou = O.box_iou
    tv.transforms = tvt
    tv.ops = tvo
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt,
                        "torchvision.transforms.transforms": tvtt, "torchvision.ops": tvo})
    A = types.ModuleType("albumentations")
    Ap = types.ModuleType("albumentations.pytorch")
    Apt = types.ModuleType("albumentations.pytorch.transforms")
    Apt.ToTensorV2 = object
    Ap.transforms = Apt
    A.pytorch = Ap
    sys.modules.update({"albumentations": A, "albumentations.pytorch": Ap,
                        "albumentations.pytorch.transforms": Apt})
    ti = types.ModuleType("torchinfo")
    ti.summary = lambda *a, **k: ""
    sys.modules["torchinfo"] = ti
    pl = types.ModuleType("pytorch_lightning")
    pl.LightningModule = torch.nn.Module
    pl.Trainer = object
    sys.modules["pytorch_lightning"] = pl


def _load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def import_reference():
    _install_stubs()
    # the reference's top-level packages are `datasets`, `models`, `losses`; HuggingFace
    # `datasets` is installed too, so the reference root must come first on sys.path.
    for m in [m for m in sys.modules if m == "datasets" or m.startswith("datasets.")]:
        del sys.modules[m]
    sys.path.insert(0, REF)
    # `datasets/__init__.py` and `datasets/WIDERFace/__init__.py` pull gdown/Lightning data
    # modules; register bare packages instead and load the needed files by path.
    pkg = types.ModuleType("datasets"); pkg.__path__ = [os.path.join(REF, "datasets")]
    sys.modules["datasets"] = pkg
    utils = _load_by_path("datasets.utils", os.path.join(REF, "datasets", "utils.py"))
    pkg.utils = utils
    dataset = _load_by_path("ref_dataset", os.path.join(REF, "datasets", "WIDERFace", "dataset.py"))
    losses = _load_by_path("ref_yololoss", os.path.join(REF, "losses", "YoloLoss.py"))
    mpkg = types.ModuleType("models"); mpkg.__path__ = [os.path.join(REF, "models")]
    sys.modules["models"] = mpkg
    bm = _load_by_path("models.BaseModel", os.path.join(REF, "models", "BaseModel.py"))
    mpkg.BaseModel = bm.BaseModel
    pr = _load_by_path("models.PoolResnet", os.path.join(REF, "models", "PoolResnet.py"))
    rn = _load_by_path("models.Resnet", os.path.join(REF, "models", "Resnet.py"))
    return dict(utils=utils, dataset=dataset, yolo_loss=losses.yolo_loss,
                PoolResnet=pr.PoolResnet, Resnet=rn.Resnet)


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    conv = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        conv[k] = np.asarray(v)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **conv)
    print("wrote", name, {k: conv[k].shape for k in list(conv)[:6]}, "...")


# ----------------------------------------------------------------------------- G1 loss
def g1_loss(R):
    torch.manual_seed(0)
    # KAT from SURVEY.md 8c
    pred = torch.rand(5, 10, 10); gt = torch.zeros(5, 10, 10)
    gt[:, 3, 4] = torch.tensor([1, .2, .7, .1, .15])
    kat = R["yolo_loss"](pred, gt)
    assert abs(float(kat) - 5.158348560333252) < 1e-6, float(kat)
    for S in (10, 15):
        g = torch.Generator().manual_seed(100 + S)
        B = 6
        pred = torch.rand(B, 5, S, S, generator=g) * 0.98 + 0.01
        gt = torch.zeros(B, 5, S, S)
        for n in range(B):
            if n == 1:
                continue                                   # zero-object image
            for _ in range(1 + n % 3):
                i = int(torch.randint(0, S, (1,), generator=g)); j = int(torch.randint(0, S, (1,), generator=g))
                gt[n, :, i, j] = torch.cat([torch.ones(1), torch.rand(4, generator=g)])
        gt[2, 3, :, :] *= 1e-6                             # object cell with tiny w
        pred[3, :, 0, 0] = float("nan")                    # NaN cell -> 0.1 (Q6)
        pred[3, 2, 1, 1] = float("nan")
        pred[4, 3, 2, 2] = 0.0                             # sqrt singularity, obj==0 there unless hit (Q8)
        gt[5, :, 4, 4] = torch.tensor([1, .3, .4, .2, .25]); pred[5, 3, 4, 4] = 0.0   # obj cell with p3==0 -> inf grad
        losses, grads = [], []
        for n in range(B):
            p = pred[n].clone().requires_grad_(True)
            l = R["yolo_loss"](p, gt[n])
            (gp,) = torch.autograd.grad(l, p)
            losses.append(l.detach()); grads.append(gp)
        save(f"g1_loss_S{S}", pred=pred, gt=gt, loss=torch.stack(losses), grad=torch.stack(grads),
             kat_loss=kat.detach())


# ----------------------------------------------------------------------------- G2 encode
def g2_encode(R):
    DS = R["dataset"].WIDERFaceDataset
    cases = []
    g = torch.Generator().manual_seed(7)

    def rnd_boxes(n, size):
        rows = []
        for _ in range(n):
            w = int(torch.randint(1, size // 2, (1,), generator=g)); h = int(torch.randint(1, size // 2, (1,), generator=g))
            x = int(torch.randint(0, size - w + 1, (1,), generator=g)); y = int(torch.randint(0, size - h + 1, (1,), generator=g))
            rows.append([1.0, x, y, w, h])
        return torch.tensor(rows, dtype=torch.float32).reshape(-1, 5)

    cases.append((480, 10, torch.tensor([[1, 100, 200, 50, 60], [1, 479, 10, 30, 30], [1, 480, 480, 5, 5]], dtype=torch.float32)))  # SURVEY KAT
    cases.append((480, 10, torch.zeros(0, 5)))                                                    # empty
    cases.append((480, 10, torch.tensor([[1, 50, 60, 20, 20], [1, 55, 70, 30, 10]], dtype=torch.float32)))  # collision: last wins
    cases.append((480, 10, torch.tensor([[1, -3, 500, 20, 20]], dtype=torch.float32)))             # out of range: clamp, unclamped offsets
    cases.append((480, 15, rnd_boxes(2, 480)))
    cases.append((480, 10, rnd_boxes(50, 480)))
    cases.append((640, 20, rnd_boxes(7, 640)))
    cases.append((480, 10, rnd_boxes(1, 480)))
    maxn = max(c[2].shape[0] for c in cases)
    sizes, Ss, ns, boxes, maps = [], [], [], [], []
    for size, S, b in cases:
        ds = DS(data_dir=None, num_of_patches=S, input_shape=(size, size), targets=[])
        fm = ds.convert_bbx_to_feature_map(b, (size, size)) if b.shape[0] else torch.zeros(5, S, S)
        if b.shape[0] == 0:
            fm = ds.convert_bbx_to_feature_map(b, (size, size))
        pad = torch.zeros(maxn, 5); pad[: b.shape[0]] = b
        full = torch.zeros(5, 20, 20); full[:, :S, :S] = fm
        sizes.append(size); Ss.append(S); ns.append(b.shape[0]); boxes.append(pad); maps.append(full)
    save("g2_encode", size=np.array(sizes), S=np.array(Ss), n=np.array(ns),
         boxes=torch.stack(boxes), maps=torch.stack(maps))
    # SURVEY KAT check
    fm = maps[0]
    assert torch.allclose(fm[:, 2, 4], torch.tensor([1, .0833333, .1666667, .1041667, .125]), atol=1e-6)


# ----------------------------------------------------------------------------- G3/G4 decode
def g3_decode(R):
    RB = R["utils"].ReduceBoundingBoxes
    DS = R["dataset"].WIDERFaceDataset
    g = torch.Generator().manual_seed(11)
    recs = []
    for (size, S, pt, iou) in [(480, 10, 0.7, 0.01), (480, 10, 0.5, 0.5), (480, 15, 0.5, 0.5), (640, 20, 0.3, 0.3)]:
        for trial in range(4):
            x = torch.rand(5, S, S, generator=g)
            if trial == 1:
                x[0] = x[0] * 0.2                                  # few / no boxes
            if trial == 2:
                x[0] = 0.0                                         # empty result
            if trial == 3:
                x[0] = (x[0] > 0.5).float() * 0.9                  # ties in score
                x[0, 0, 0] = pt                                    # == threshold: strict > drops it
            rb = RB(pt, iou, (3, size, size), S)
            sx = rb.scale_batch_bbx_xywh(x.clone())
            bb, exist = rb.remove_low_probabilty_bbx(sx)
            K = 0
            pre = torch.zeros(S * S, 5)
            if int(exist) == 1:
                bb = rb.convert_batch_to_xyxy(bb.clone())
                K = bb.shape[0]
                pre[:K, 0] = bb[:, 0]; pre[:K, 1:] = torch.round(bb[:, 1:])
            out = rb(x.clone())                                    # full forward (nms = stand-in)
            outp = torch.zeros(S * S, 5); outp[: out.shape[0]] = out
            xin = torch.zeros(5, 20, 20); xin[:, :S, :S] = x
            prep = torch.zeros(400, 5); prep[: S * S] = pre
            outpp = torch.zeros(400, 5); outpp[: S * S] = outp
            recs.append((size, S, pt, iou, xin, K, prep, out.shape[0], outpp))
    save("g3_decode", size=np.array([r[0] for r in recs]), S=np.array([r[1] for r in recs]),
         pt=np.array([r[2] for r in recs]), iou=np.array([r[3] for r in recs]),
         x=torch.stack([r[4] for r in recs]), K=np.array([r[5] for r in recs]),
         pre=torch.stack([r[6] for r in recs]), Kout=np.array([r[7] for r in recs]),
         out=torch.stack([r[8] for r in recs]))
    # G8 property of the commented-out reference check (dataset.py:125-139):
    # decode(encode(b)) == b for integer boxes in distinct cells.
    ok = 0
    for S, size in [(10, 480), (15, 480), (20, 640)]:
        ds = DS(data_dir=None, num_of_patches=S, input_shape=(size, size), targets=[])
        ps = size // S
        for t in range(20):
            cells = torch.randperm(S * S, generator=g)[:5]
            rows = []
            for c in cells:
                i, j = int(c) // S, int(c) % S
                xx = i * ps + int(torch.randint(0, ps, (1,), generator=g)); yy = j * ps + int(torch.randint(0, ps, (1,), generator=g))
                rows.append([1.0, xx, yy, int(torch.randint(1, 100, (1,), generator=g)), int(torch.randint(1, 100, (1,), generator=g))])
            b = torch.tensor(rows, dtype=torch.float32)
            fm = ds.convert_bbx_to_feature_map(b, (size, size))
            rb = RB(0.5, 1.1, (3, size, size), S)                   # iou thr > 1: NMS keeps all
            dec = rb(fm)
            a = b[torch.argsort(b[:, 1] * 10000 + b[:, 2])]
            d = dec[torch.argsort(dec[:, 1] * 10000 + dec[:, 2])]
            assert torch.equal(a, d), (a, d)
            ok += 1
    print("encode->decode round trip exact on", ok, "trials (reference code)")


# ----------------------------------------------------------------------------- G5/G7 conv stack
def _ref_masks(model, x):
    """Run the reference model in train mode and recover every Dropout2d's per-(n,c)
    scale from its input/output (forward hooks; the reference is not modified)."""
    recs = {}
    hooks = []

    def mk(name):
        def hook(mod, inp, out):
            i = inp[0].detach(); o = out.detach()
            N, C = i.shape[:2]
            fi = i.reshape(N, C, -1); fo = o.reshape(N, C, -1)
            idx = fi.abs().argmax(dim=2, keepdim=True)
            num = torch.gather(fo, 2, idx).squeeze(2); den = torch.gather(fi, 2, idx).squeeze(2)
            m = torch.where(den != 0, num / den, torch.zeros_like(num))
            p = mod.p
            m = torch.where(m.abs() > 0.5, torch.full_like(m, 1.0 / (1.0 - p)), torch.zeros_like(m))
            recs[name] = m
        return hook

    for k, blk in enumerate(model.residual_blocks):
        hooks.append(blk.dropout2d.register_forward_hook(mk(f"residual_blocks.{k}")))
    hooks.append(model.dropout2d.register_forward_hook(mk("head")))
    y = model(x)
    for h in hooks:
        h.remove()
    return y, recs


def g5_convstack(R):
    for kind, Cls, kw, size, S in [("poolresnet", R["PoolResnet"], {}, 480, 10),
                                   ("resnet", R["Resnet"], {}, 240, 15)]:
        for filters in (8,):
            torch.manual_seed(0)
            nb = 10 if kind == "poolresnet" else 6
            if kind == "resnet":
                # 240 -> stem 120 -> 60 -> 30 -> 15  (3 pooled blocks) then flat
                pass
            model = Cls(filters=filters, input_shape=(3, size, size), num_of_patches=S,
                        num_of_residual_blocks=nb, **kw)
            B = 2
            g = torch.Generator().manual_seed(5)
            x_u8 = torch.randint(0, 256, (B, 3, size, size), generator=g, dtype=torch.uint8)
            x = x_u8.float() / 255.0                       # stored as u8 to keep the fixture small
            model.eval()
            with torch.no_grad():
                y_eval = model(x)
            model.train()
            torch.manual_seed(123)
            y_train, masks = _ref_masks(model, x)
            # targets + loss + grads through the reference loss
            boxes = O.synthetic_boxes(B, size, seed=9)
            boxes[0] = torch.tensor([[1, 100, 120, 60, 80]], dtype=torch.float32) * torch.tensor([1, size / 480, size / 480, size / 480, size / 480])
            boxes[0] = torch.round(boxes[0])
            ds = R["dataset"].WIDERFaceDataset(data_dir=None, num_of_patches=S, input_shape=(size, size), targets=[])
            y = torch.stack([ds.convert_bbx_to_feature_map(b, (size, size)) for b in boxes])
            loss = 0
            for n in range(B):
                loss = loss + R["yolo_loss"](y_train[n], y[n])
            names = [n for n, _ in model.named_parameters()]
            grads = torch.autograd.grad(loss, [p for _, p in model.named_parameters()])
            arrays = dict(x_u8=x_u8, y=y, y_eval=y_eval, y_train=y_train, loss=loss.detach())
            for n, p in model.named_parameters():
                arrays["param/" + n] = p.detach().clone()
            for n, gr in zip(names, grads):
                arrays["grad/" + n] = gr
            for n, m in masks.items():
                arrays["mask/" + n] = m
            # G7: one Adam step (torch.optim._multi_tensor.Adam formula == optim.Adam foreach)
            opt = torch.optim.Adam(model.parameters(), lr=1e-4, foreach=True)
            for p, gr in zip(model.parameters(), grads):
                p.grad = gr.clone()
            opt.step()
            for n, p in model.named_parameters():
                arrays["param_after/" + n] = p.detach()
            save(f"g5_{kind}_F{filters}", **arrays)


# ----------------------------------------------------------------------------- G6 trained weights
def read_archive_tensors(path):
    """Raw-byte read of a TorchScript zip's tensor storages; nothing is unpickled or
    executed.  Layout (verified by sizes): data/0 conv1.weight, data/1 conv1.bias, then per
    block conv1.w, conv1.b, conv2.w, conv2.b, then out.weight, out.bias."""
    z = zipfile.ZipFile(path)
    names = [n for n in z.namelist() if "/data/" in n and n.rsplit("/", 1)[1].isdigit()]
    names.sort(key=lambda n: int(n.rsplit("/", 1)[1]))
    raws = [np.frombuffer(z.read(n), dtype="<f4").copy() for n in names]
    F_ = raws[1].size
    k_stem = int(round((raws[0].size / (F_ * 3)) ** 0.5))
    nblocks = (len(raws) - 4) // 4
    k_head = int(round((raws[-2].size / (5 * F_)) ** 0.5))
    P = {"conv1.weight": raws[0].reshape(F_, 3, k_stem, k_stem), "conv1.bias": raws[1]}
    for k in range(nblocks):
        P[f"residual_blocks.{k}.conv1.weight"] = raws[2 + 4 * k].reshape(F_, F_, 3, 3)
        P[f"residual_blocks.{k}.conv1.bias"] = raws[3 + 4 * k]
        P[f"residual_blocks.{k}.conv2.weight"] = raws[4 + 4 * k].reshape(F_, F_, 3, 3)
        P[f"residual_blocks.{k}.conv2.bias"] = raws[5 + 4 * k]
    P["out.weight"] = raws[-2].reshape(5, F_, k_head, k_head)
    P["out.bias"] = raws[-1]
    assert P["out.bias"].size == 5
    return P, F_, nblocks


def g6_trained(R):
    from PIL import Image
    path = os.path.join(REF, "saved_models/official/PoolResnet/small_model_10x10_480.pth")
    P, F_, nb = read_archive_tensors(path)
    model = R["PoolResnet"](filters=F_, input_shape=(3, 480, 480), num_of_patches=10,
                            num_of_residual_blocks=nb, probability_threshold=0.7, iou_threshold=0.01)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
    model.eval()
    imgs, outs, dets, nd = [], [], [], []
    for name in ("13.jpg", "1.jpg", "17.jpg"):
        im = Image.open(os.path.join(REF, "imgs/test_imgs", name)).convert("RGB").resize((480, 480), Image.BILINEAR)
        u8 = torch.from_numpy(np.asarray(im).copy()).permute(2, 0, 1).contiguous()
        with torch.no_grad():
            y = model(torch.stack([u8, u8]).float() / 255.0)            # conv stack (reference)
            det = model(torch.stack([u8, u8]), predict=torch.tensor(1))  # demo path, image 0 (Q12)
        imgs.append(u8); outs.append(y[0])
        d = torch.zeros(100, 5); d[: det.shape[0]] = det
        dets.append(d); nd.append(det.shape[0])
        print(name, "dets", det.shape[0], det[:3].tolist())
    arrays = dict(images=torch.stack(imgs), y=torch.stack(outs), dets=torch.stack(dets), ndets=np.array(nd))
    for k, v in P.items():
        arrays["param/" + k] = v
    save("g6_trained_small", **arrays)
    # medium archive: cross-check of the SURVEY probe (13.jpg -> 2 boxes, top 0.979)
    pathm = os.path.join(REF, "saved_models/official/PoolResnet/medium_model_10x10_480.pth")
    Pm, Fm, nbm = read_archive_tensors(pathm)
    mm = R["PoolResnet"](filters=Fm, input_shape=(3, 480, 480), num_of_patches=10, num_of_residual_blocks=nbm,
                         probability_threshold=0.7, iou_threshold=0.01)
    mm.load_state_dict({k: torch.from_numpy(v) for k, v in Pm.items()}); mm.eval()
    with torch.no_grad():
        det = mm(torch.stack([imgs[0], imgs[0]]), predict=torch.tensor(1))
    print("medium 13.jpg:", det.tolist())


# ----------------------------------------------------------------------------- G9 metrics
def g9_metrics(R):
    """ModelMeta.step's metric block (:170-218) cannot be imported without Lightning's
    Trainer state (self.log, self.current_epoch); its arithmetic is restated in
    oracle.step_metrics.  No fixture: PARITY UNPINNED beyond the decode it calls."""


if __name__ == "__main__":
    torch.set_num_threads(8)
    R = import_reference()
    g1_loss(R)
    g2_encode(R)
    g3_decode(R)
    g5_convstack(R)
    g6_trained(R)
    print("done")
