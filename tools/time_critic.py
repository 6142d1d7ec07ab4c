"""Critic update (loss + gradient + optimizer step) at the paper shape (4 x 128 critic with LayerNorm on layers 2-4, 128 rows
per input) and a small plain critic: fused row-block path against the layer-by-layer path (SSN_CRITIC_FUSED=0)."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    import numpy as np, torch
    from tc_gan_amd.critic import Critic, Updater
    for name, layers, norm, batch in [('paper 4x128 LN', [128] * 4, ['none', 'layer', 'layer', 'layer'], 128),
                                      ('plain 3x64', [64] * 3, 'none', 1024)]:
        c = Critic(8, layers, normalization=norm, precision='fp32')
        upd = Updater(learning_rate=1e-3, update_name='rmsprop')
        rs = np.random.RandomState(0)
        xg, xd = (torch.as_tensor(rs.rand(batch, 8) * 5, device='cuda', dtype=torch.float32) for _ in range(2))
        xp = 0.5 * (xg + xd)
        cond = torch.as_tensor(np.stack([np.full(batch, 20.), rs.rand(batch), np.zeros(batch)], 1), device='cuda', dtype=torch.float32)
        def step():
            c.loss_grad(xg, cond, xd, cond, xp, cond, 10.0)
            upd(c.params, c.grads)
        for _ in range(5): step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): step()
        e1.record(); torch.cuda.synchronize()
        print('  %-16s %7.1f us per critic update' % (name, e0.elapsed_time(e1) / 50 * 1e3))
    sys.exit(0)
for fused in ('1', '0'):
    print('SSN_CRITIC_FUSED=%s' % fused)
    sys.stdout.flush()
    subprocess.run([sys.executable, os.path.abspath(__file__), 'child'], env=dict(os.environ, SSN_CRITIC_FUSED=fused), check=True)
