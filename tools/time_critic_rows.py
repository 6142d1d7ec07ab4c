"""GPU time of one critic loss + gradient at the C3 shape (3 x 1024 rows, 11-512-512-512-1, bf16 operands), launches
back to back: HIP events around 50 calls (the queue stays ahead of the device, so this is device time, not launch time).
SSN_CRITIC_ROWS=0 -> layer-by-layer chain; SSN_LIBDIR -> another build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tc_gan_amd.critic import Critic
batch = 1024
c = Critic(8, [512] * 3, normalization='none', precision='bf16')
rs = np.random.RandomState(0)
xg, xd = (torch.as_tensor(rs.rand(batch, 8) * 5, device='cuda', dtype=torch.float32) for _ in range(2))
xp = 0.5 * (xg + xd)
cond = torch.as_tensor(np.stack([np.full(batch, 20.), rs.rand(batch), np.zeros(batch)], 1), device='cuda', dtype=torch.float32)
big = torch.empty(1 << 28, device='cuda')
def run(n):
    big.zero_()                      # ~0.4 ms of device work: the host queues the n calls behind it
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): c.loss_grad(xg, cond, xd, cond, xp, cond, 10.0)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
run(5)
print('ROWS=%s LIBDIR=%s: %.1f us per loss_grad (10 calls) %.1f (4 calls)  stats %s' % (
    os.environ.get('SSN_CRITIC_ROWS', '1'), os.path.basename(os.environ.get('SSN_LIBDIR', 'main')), run(10), run(4),
    np.array2string(c.stats.cpu().numpy(), precision=7)))
