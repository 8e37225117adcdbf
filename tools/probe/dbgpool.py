import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
import fdet_amd
from fdet_amd import hotpath as hp, ps
N,C,H,W = 1,64,60,60
g = torch.Generator().manual_seed(0)
x = torch.randn(N,C,H,W,generator=g); skip = torch.zeros(N,C,H,W)
w = torch.randn(C,C,3,3,generator=g)*0.1; b = torch.zeros(C)
nf, nb = hp.packed_sizes(C,C)
wf = torch.empty(nf,device="cuda"); wb = torch.empty(nb,device="cuda")
hp.pack_conv3x3_weights(w.cuda(), wf, wb, x3=True)
xp = ps.PsTensor.from_f32(x.cuda()); sp = ps.PsTensor.from_f32(skip.cuda())
pool_f = torch.full((N,C,H//2,W//2), float("nan"), device="cuda")
pp = ps.PsTensor(N,C,H//2,W//2,"cuda"); ps.conv3x3_ps_fwd_pool(xp, wf, b.cuda(), sp, None, pp, pool_f, None)
c = F.leaky_relu(F.conv2d(xp.to_f32().cpu(), w, b, padding=1), 0.2)
ref = F.max_pool2d(c, 2)
got = pool_f.cpu(); print("ps vs f32 out", (pp.to_f32().cpu()-got).abs().max().item())
err = (got-ref).abs()
print("max err", err.max().item(), "nan", torch.isnan(got).sum().item())
bad = err > 1e-3
print("bad frac", bad.float().mean().item())
print("bad by channel", bad.float().mean(dim=(0,2,3))[:16])
print("bad by row", bad.float().mean(dim=(0,1,3)))
print("bad by col", bad.float().mean(dim=(0,1,2)))
# does got match some other pooling of c? e.g. shifted
for dy in (0,1):
  for dx_ in (0,1):
    cc = F.pad(c,(0,2,0,2))[:,:,dy:dy+H,dx_:dx_+W]
    print(dy,dx_, (F.max_pool2d(cc,2)-got).abs().max().item())
