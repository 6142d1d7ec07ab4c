import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tc_gan_amd.stimuli import stimulus_batch
rs = np.random.RandomState(3)
B, NB, N = 37, 8, 101
bw = torch.as_tensor(rs.rand(B, NB) * 0.9 + 0.05, device='cuda', dtype=torch.float32)
con = torch.as_tensor(rs.rand(B, NB) * 30, device='cuda', dtype=torch.float32)
zin = torch.as_tensor(rs.choice(2, (B, 2 * N)) * 2.0 - 1, device='cuda', dtype=torch.float32)
v = torch.as_tensor(rs.rand(2 * N) * 0.7, device='cuda', dtype=torch.float32)
amp = 1 + v[None, :] * zin
want = stimulus_batch(bw, con, 0.1, N, amp=amp)
got = stimulus_batch(bw, con, 0.1, N, zin=zin, v=v)
base = stimulus_batch(bw, con, 0.1, N)
d = (got - want).abs()
print('max diff', float(d.max()), 'n diff', int((d > 0).sum()), 'of', d.numel(), 'rel', float((d / want.abs().clamp(min=1e-30)).max()))
ones = torch.ones_like(amp)
print('amp=1 vs none equal', bool(torch.equal(stimulus_batch(bw, con, 0.1, N, amp=ones), base)))
print('hetero v=0 vs none equal', bool(torch.equal(stimulus_batch(bw, con, 0.1, N, zin=zin, v=torch.zeros(1, device='cuda')), base)))
# is it the product order?  want = ((amp*con)*s1)*s2
for zname, z in (('pm1', zin), ('unif', torch.as_tensor(rs.rand(B, 2 * N) * 2 - 1, device='cuda', dtype=torch.float32))):
    for vv in (v, torch.as_tensor([0.31, 0.057], device='cuda', dtype=torch.float32), torch.as_tensor([0.173], device='cuda', dtype=torch.float32)):
        vs = vv if vv.numel() == 2 * N else (vv.expand(2) if vv.numel() == 1 else vv).repeat_interleave(N)
        a = 1 + vs[None, :] * z
        a2 = torch.add(torch.mul(vs[None, :], z), 1.0)
        w = stimulus_batch(bw, con, 0.1, N, amp=a)
        g = stimulus_batch(bw, con, 0.1, N, zin=z, v=vv)
        d = (g - w).abs()
        print(zname, vv.numel(), 'n diff', int((d > 0).sum()), 'max rel', float((d / w.abs().clamp(min=1e-30)).max()), 'amp forms equal', bool(torch.equal(a, a2)),
              'fma form equal', bool(torch.equal(a, torch.addcmul(torch.ones_like(z), vs[None, :].expand_as(z), z))))
