import os, sys, torch
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/tc_gan_amd') else os.getcwd())
from tc_gan_amd import genops
B, NB, M, T, skip = 1024, 8, 200, 1200, 1000
KERNEL, GEN = 8, {}
if len(sys.argv) > 1 and sys.argv[1] == 'paper':       # the paper's run: 128 draws, 2N = 202, 240 / 200 steps, tau_E = 2, the library's own choice
    B, NB, M, T, skip, KERNEL, GEN = 128, 8, 202, 240, 200, 0, dict(tau_E=2.0)
g = torch.Generator(device='cuda'); g.manual_seed(1)
W = (torch.rand((B, M, M), device='cuda', generator=g) - 0.6) * 0.02
ext = torch.rand((B, NB, M), device='cuda', generator=g) * 20
gp = genops.make_gen_params(seqlen=T, skip_steps=skip, kernel=KERNEL, **GEN)
out = genops.gen_forward(W, ext, gp, save=True)
gta = torch.rand((B, NB, M), device='cuda', generator=g)
traj, df = out['traj'], out['df']
def bwd(): return genops.gen_backward(W, traj, df, gta, 1.0, 0.01, gp)
bwd(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3): bwd()
e1.record(); torch.cuda.synchronize()
print('adjoint (in place, repeated) %.3f ms' % (e0.elapsed_time(e1) / 3))
if hasattr(genops.libssnode, 'ssn_debug_duo_stamps'):
    import ctypes
    buf = (ctypes.c_ulonglong * 16)()
    genops.libssnode.ssn_debug_duo_stamps(buf)
    n = max(int(buf[8]), 1)
    for name, o in (('wave 0', 0), ('wave 3', 4)):
        print('  draw 0 %s: chain %.0f  barrier %.0f  serial %.0f  barrier %.0f cycles per step' % ((name,) + tuple(buf[o + i] / n for i in range(4))))
xmax = genops.rate_bound(gp)
out = genops.gen_forward(W, ext, gp, save=True)      # (the loop above ran the in-place sweep over its own output)
traj, df = out['traj'], out['df']
dmax = torch.empty((B,), device='cuda')
def two():
    d, dm = genops.gen_backward(W, traj, df.clone(), gta, 1.0 / out['n_dyn'], 0.01 / out['n_rate'], gp, want_dmax=True)
    return genops.weight_grad(d, traj, dmax=dm, xmax=xmax)
def fused(): return genops.gen_backward_fused(W, traj, df, gta, 1.0 / out['n_dyn'], 0.01 / out['n_rate'], gp, xmax)
def clone(): return df.clone()
for name, fn in (('df.clone()', clone), ('two launches (adjoint + dL/dW, incl. the clone)', two), ('fused adjoint + dL/dW', fused)):
    fn(); torch.cuda.synchronize()
    e0.record()
    for _ in range(10): r = fn()
    e1.record(); torch.cuda.synchronize()
    print('%s %.3f ms' % (name, e0.elapsed_time(e1) / 10))
a = two(); b = fused()[0]
print('max |fused - two| / max |two| per draw: %.2e' % float(((a - b).abs().reshape(B, -1).max(dim=1).values / a.abs().reshape(B, -1).max(dim=1).values).max()))
if hasattr(genops.libssnode, 'ssn_debug_fuse_stamps'):
    import ctypes
    buf = (ctypes.c_ulonglong * 32)()
    genops.libssnode.ssn_debug_fuse_stamps(buf)
    for w in range(4):
        n = max(int(buf[8 * w + 4]), 1)
        print('  wave %d: update+serial %.0f  barrier %.0f  chain %.0f  barrier %.0f cycles per step' % ((w,) + tuple(buf[8 * w + i] / n for i in range(4))))
