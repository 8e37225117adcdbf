import torch, torch.nn.functional as F, sys
sys.path.insert(0,'.')
import fdet_amd
from fdet_amd import hotpath as hp
g = torch.Generator().manual_seed(1)
N, C, H, W = 3, 16, 12, 20
c = torch.randn(N, C, H, W, generator=g); x = torch.randn(N, C, H, W, generator=g)
scale = (torch.rand(N, C, generator=g) > 0.25).float() / 0.75
out = torch.empty(N, C, H, W, device="cuda")
hp.block_tail_fwd(c.cuda(), x.cuda(), scale.cuda(), out, 1)
ref = c*scale[:, :, None, None] + x
d=(out.cpu()-ref).abs()
print('max diff', d.max().item(), 'n mismatch', (d>0).sum().item(), 'of', d.numel())
ref2 = torch.addcmul(x, c, scale[:, :, None, None].expand_as(c))
print('vs fused addcmul: mismatches', ((out.cpu()-ref2)!=0).sum().item())
i=(d>0).nonzero()[0].tolist(); print(i, c[tuple(i)].item(), scale[i[0],i[1]].item(), x[tuple(i)].item(), out.cpu()[tuple(i)].item(), ref[tuple(i)].item())
