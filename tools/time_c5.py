"""C5 forward (FF_lalazar get_FF_output, box 40, 16384 samples) timing for profiling runs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tc_gan_amd import ff_model
nsam, box, nhid = 16384, 40, 1
G = box ** 3
gen = torch.Generator(device='cuda'); gen.manual_seed(99)
wid = torch.rand((nsam, G), device='cuda', generator=gen)
con = (torch.rand((nsam, nhid, G), device='cuda', generator=gen) < 0.01).float()
strn = torch.rand((nsam, nhid, G), device='cuda', generator=gen)
ths = torch.rand((nsam, nhid), device='cuda', generator=gen) * 2 - 1
stim = ff_model.default_stimuli()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
out = ff_model.ff_forward(ff_model.START_PARAMS, wid, con, strn, ths, stim, box)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    out = ff_model.ff_forward(ff_model.START_PARAMS, wid, con, strn, ths, stim, box)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
print('ff_forward %.3f ms  %.2f TB/s algorithmic (12 B per sample-point)' % (ms, nsam * G * 12 / ms * 1e-9))
idx, val = ff_model.sparsify(con, strn)
outs = ff_model.ff_forward_sparse(ff_model.START_PARAMS, wid, idx, val, ths, stim, box)
torch.cuda.synchronize()
e0.record()
for _ in range(n):
    outs = ff_model.ff_forward_sparse(ff_model.START_PARAMS, wid, idx, val, ths, stim, box)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
nb = nsam * (G * 4 + idx.shape[2] * 8)
print('ff_forward_sparse %.3f ms  %.2f TB/s algorithmic (4 B per sample-point + 8 B per connection, %d slots per unit)   max |sparse - dense| %.3e (max %.3e)'
      % (ms, nb / ms * 1e-9, idx.shape[2], float((outs - out).abs().max()), float(out.abs().max())))
