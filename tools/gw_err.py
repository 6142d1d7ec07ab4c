import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tc_gan_amd import genops
for (B, NB, T, M) in [(2, 4, 7, 20), (2, 8, 30, 100), (3, 8, 50, 200), (2, 8, 1200, 200)]:
    rs = np.random.RandomState(B * 1000 + T)
    d = (rs.randn(B, NB, T, M) * np.exp(rs.uniform(-12, 0, (B, NB, T, M)))).astype('float32')
    x = (rs.rand(B, NB, T, M) * 100 * np.exp(rs.uniform(-6, 0, (B, NB, T, M)))).astype('float32')
    K = NB * T
    want = np.einsum('bki,bkj->bij', d.reshape(B, K, M).astype('float64'), x.reshape(B, K, M).astype('float64'))
    scale = np.einsum('bki,bkj->bij', np.abs(d.reshape(B, K, M)).astype('float64'), np.abs(x.reshape(B, K, M)).astype('float64'))
    dt, xt = torch.as_tensor(d).cuda(), torch.as_tensor(x).cuda()
    res = {}
    for k in (1, 2):
        got = genops.weight_grad(dt, xt, kernel=k).cpu().numpy()
        res[k] = got
        e = np.abs(got - want) / scale
        print((B, NB, T, M), 'kernel', k, 'max err/scale %.3e  rms %.3e' % (e.max(), np.sqrt((e ** 2).mean())))
    bm = torch.bmm(dt.reshape(B, K, M).transpose(1, 2), xt.reshape(B, K, M)).cpu().numpy()
    e = np.abs(bm - want) / scale
    print((B, NB, T, M), 'rocBLAS  max err/scale %.3e  rms %.3e' % (e.max(), np.sqrt((e ** 2).mean())))
