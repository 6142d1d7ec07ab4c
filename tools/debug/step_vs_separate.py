import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tc_gan_amd.critic import Critic, Updater
rs = np.random.RandomState(7)
n, nx = 96, 8
kw = dict(seed=3, precision='bf16', normalization='none', nonlinearity='rectify')
a, b = Critic(nx, [512, 512, 512], **kw), Critic(nx, [512, 512, 512], **kw)
ua, ub = Updater(0.01, 'rmsprop', reg_l2_decay=1e-3), Updater(0.01, 'rmsprop', reg_l2_decay=1e-3)
for it in range(3):
    xg = torch.as_tensor(rs.rand(n, nx) * 5, device='cuda', dtype=torch.float32)
    xd = torch.as_tensor(rs.rand(n, nx) * 5, device='cuda', dtype=torch.float32)
    cond = torch.as_tensor(np.stack([np.full(n, 20.), rs.rand(n) * 2 - 1, rs.randint(0, 2, n)], axis=1), device='cuda', dtype=torch.float32)
    eps = torch.as_tensor(rs.rand(n, 1), device='cuda', dtype=torch.float32)
    pens = torch.as_tensor(rs.rand(2), device='cuda', dtype=torch.float64)
    xp_a = a.interpolate(eps, xd, xg)
    stats = a.loss_grad(xg, cond, xd, cond, xp_a, cond, 10.0).cpu().numpy().copy()
    ga = a.grads.clone()
    ua(a.params, a.grads)
    acc_a = a.accuracy_device(xg, cond, xd, cond).cpu().numpy()
    xp_b, tail = b.step(ub, xg, xd, cond, eps, 10.0, pens64=pens)
    print(it, 'separate stats', stats, 'acc', acc_a, '|g|', float(ga.norm()))
    print(it, 'step     stats', b.stats.cpu().numpy(), 'tail', tail.cpu().numpy()[:5], '|g|', float(b.grads.norm()),
          'xp equal', bool(torch.equal(xp_a, xp_b)), 'params equal', bool(torch.equal(a.params, b.params)), 'grads equal', bool(torch.equal(ga, b.grads)))
