"""Time the 15x15 block chain (8 blocks, N=256) forward (training flavour: a, c, out kept) and backward."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import importlib
pkg = importlib.import_module("pytorch-face-detection-from-scratch_amd")
hp = importlib.import_module("pytorch-face-detection-from-scratch_amd.hotpath")
N, C, H, W, nb = 256, 64, 15, 15, 8
g = torch.Generator().manual_seed(1)
x = torch.randn(N, C, H, W, generator=g).cuda()
nf, nbk = hp.packed_sizes(C, C)
wf1, wb1, wf2, wb2, b1, b2 = [], [], [], [], [], []
for k in range(nb):
    for lf, lb, lbias in ((wf1, wb1, b1), (wf2, wb2, b2)):
        w = (torch.randn(C, C, 3, 3, generator=g) * 0.05).cuda()
        f_ = torch.empty(nf, device="cuda"); b_ = torch.empty(nbk, device="cuda")
        hp.pack_conv3x3_weights(w, f_, b_, x3=True)
        lf.append(f_); lb.append(b_); lbias.append((torch.randn(C, generator=g) * 0.1).cuda())
sc = [((torch.rand(N, C, generator=g) > 0.25).float() / 0.75).cuda() for _ in range(nb)]
mk = lambda: [torch.empty(N, C, H, W, device="cuda") for _ in range(nb)]
a_d, c_d, o_d, z1, z2 = mk(), mk(), mk(), mk(), mk()
dx = torch.empty(N, C, H, W, device="cuda")
def fwd(): hp.block_chain_fwd(x, wf1, b1, wf2, b2, sc, a_d, c_d, o_d)
def fwd_eval(): hp.block_chain_fwd(x, wf1, b1, wf2, b2, sc, None, None, [None] * (nb - 1) + [o_d[-1]])
def bwd(): hp.block_chain_bwd(x, wb1, wb2, sc, a_d, c_d, z1, z2, dx)
ps = importlib.import_module("pytorch-face-detection-from-scratch_amd.ps")
mkps = lambda n_: [ps.PsTensor(N, C, H, W, "cuda") for _ in range(n_)]
a_p, c_p, o_p, z1p, z2p = mkps(nb), mkps(nb), mkps(nb - 1), mkps(nb), mkps(nb)
x_ps = ps.PsTensor.from_f32(x)
def fwd_ps(): ps.block_chain_fwd_ps(x_ps, wf1, b1, wf2, b2, sc, a_p, c_p, o_p, o_d[-1])
def bwd_ps(): ps.block_chain_bwd_ps(x, wb1, wb2, sc, a_p, c_p, z1p, z2p, dx)
nbw = ps.conv3x3_wgrad_ps_ws_bytes(2 * nb, N, C, H, W)
ws = torch.empty(nbw // 4, device="cuda")
dW = [torch.empty(C, C, 3, 3, device="cuda") for _ in range(2 * nb)]; db = [torch.empty(C, device="cuda") for _ in range(2 * nb)]
xs = []; zs = []
for k in range(nb):
    xs += [x_ps if k == 0 else o_p[k - 1], a_p[k]]; zs += [z1p[k], z2p[k]]
def wgrad_ps(): ps.conv3x3_wgrad_ps_batched(xs, zs, dW, db, ws)
out = {}
for name, f in (("fwd", fwd), ("fwd_eval", fwd_eval), ("bwd", bwd), ("fwd_ps", fwd_ps), ("bwd_ps", bwd_ps), ("wgrad_ps16", wgrad_ps)):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    out[name + "_ms"] = round(e0.elapsed_time(e1) / 50, 4)
print(json.dumps(out))
