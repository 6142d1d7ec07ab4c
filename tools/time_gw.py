"""Time the dL/dW GEMM (csrc/ssn_gw.hip) at the BASELINE config 3 shape against torch.bmm (rocBLAS)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tc_gan_amd import genops
B, NB, T, M = 1024, 8, 1200, 200
ONLY = sys.argv[1] if len(sys.argv) > 1 else ''
d = torch.randn((B, NB, T, M), device='cuda') * 1e-3
x = torch.rand((B, NB, T, M), device='cuda') * 50


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


flops = 2.0 * B * M * M * NB * T
bytes_ = 2.0 * B * NB * T * M * 4
dmax = d.abs().reshape(B, -1).max(dim=1).values
for name, fn in [('split-bf16 MFMA', lambda: genops.weight_grad(d, x, kernel=2)),
                 ('fp16 two-part MFMA', lambda: genops.weight_grad(d, x, kernel=3, dmax=dmax, xmax=1000.0)),
                 ('rocBLAS bmm', lambda: torch.bmm(d.reshape(B, NB * T, M).transpose(1, 2), x.reshape(B, NB * T, M)))]:
    if ONLY and ONLY not in name:
        continue
    ms = timeit(fn, 2 if ONLY else 5)
    print('%-18s %.2f ms  %.1f TFLOP/s  %.2f TB/s of operands' % (name, ms, flops / ms * 1e-9, bytes_ / ms * 1e-9))
if ONLY:
    sys.exit(0)
a16 = genops.weight_grad(d, x, kernel=3, dmax=dmax, xmax=1000.0)
a = genops.weight_grad(d, x, kernel=2)
r = torch.bmm(d.reshape(B, NB * T, M).transpose(1, 2).double(), x.reshape(B, NB * T, M).double())
print('max rel err vs fp64: fp16 two-part %.2e, split %.2e, rocBLAS fp32 %.2e' % (
    float(((a16.double() - r).abs() / r.abs().max()).max()),
    float(((a.double() - r).abs() / r.abs().max()).max()),
    float(((torch.bmm(d.reshape(B, NB * T, M).transpose(1, 2), x.reshape(B, NB * T, M)).double() - r).abs() / r.abs().max()).max())))
