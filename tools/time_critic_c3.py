"""Critic update (loss + gradient + optimizer step) at the C3 shape: 8 stimuli + 3 conditions -> 3 x 512 ReLU -> 1, 1024 rows
per input, bf16 GEMM operands (bench.py's C3 lines) and fp32.  A/B two builds of the library with SSN_LIBDIR."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tc_gan_amd.critic import Critic, Updater
for precision in ('bf16', 'fp32'):
    batch = 1024
    c = Critic(8, [512] * 3, normalization='none', precision=precision)
    upd = Updater(learning_rate=1e-3, update_name='adam-wgan')
    rs = np.random.RandomState(0)
    xg, xd = (torch.as_tensor(rs.rand(batch, 8) * 5, device='cuda', dtype=torch.float32) for _ in range(2))
    xp = 0.5 * (xg + xd)
    cond = torch.as_tensor(np.stack([np.full(batch, 20.), rs.rand(batch), np.zeros(batch)], 1), device='cuda', dtype=torch.float32)
    def step():
        st = c.loss_grad(xg, cond, xd, cond, xp, cond, 10.0)
        upd(c.params, c.grads)
        return st
    first = step().cpu().numpy().copy()
    for _ in range(5): step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): step()
    e1.record(); torch.cuda.synchronize()
    print('%s: %7.1f us per critic update   first-step stats %s' % (precision, e0.elapsed_time(e1) / 50 * 1e3, np.array2string(first, precision=7)))
