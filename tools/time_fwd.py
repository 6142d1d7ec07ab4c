#!/usr/bin/env python3
"""HIP-event timing of the plain generator forward at the C3 shape (1024 draws x 8 stimuli x 1200 steps, 2N = 200) for
a list of kernel codes.  usage: tools/time_fwd.py [kernel ...] [--save] [--draws B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tc_gan_amd import genops  # noqa: E402


def timed(fn, n=5):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    args = sys.argv[1:]
    save = '--save' in args
    B = int(args[args.index('--draws') + 1]) if '--draws' in args else 1024
    kernels = [int(a) for a in args if a.isdigit() and (args.index(a) == 0 or args[args.index(a) - 1] != '--draws')] or [4, 8]
    NB, M, T, skip = 8, 200, 1200, 1000
    g = torch.Generator(device='cuda'); g.manual_seed(1)
    W = (torch.rand((B, M, M), device='cuda', generator=g) - 0.6) * 0.02
    ext = torch.rand((B, NB, M), device='cuda', generator=g) * 20
    ref = None
    for k in kernels:
        gp = genops.make_gen_params(seqlen=T, skip_steps=skip, kernel=k)
        ms = timed(lambda: genops.gen_forward(W, ext, gp, save=save))
        ta = genops.gen_forward(W, ext, gp)['time_avg']
        if ref is None:
            ref = ta
        if k == 8 and '--fine' in args and hasattr(genops.libssnode, 'ssn_debug_duo_stamps_fine'):   # -DSSN_DUO_STAMP=2
            import ctypes
            buf = (ctypes.c_ulonglong * 64)()
            torch.cuda.synchronize()
            genops.libssnode.ssn_debug_duo_stamps_fine(buf)
            for w in range(8):
                n = max(int(buf[8 * w + 7]), 1)
                seg = [buf[8 * w + i] / n for i in range(6)]
                print('  draw %d wave %d: chain %5.0f early %5.0f barrier %5.0f | serial %5.0f publish %5.0f barrier %5.0f | step %6.0f ticks (%d steps)'
                      % (w // 4, w % 4, *seg, buf[8 * w + 6] / n, n))
            print('  (kernel %.3f ms for %d steps: compare with the ticks per step to calibrate the counter)' % (ms, T))
        elif k == 8 and hasattr(genops.libssnode, 'ssn_debug_duo_stamps'):      # diagnostic build (-DSSN_DUO_STAMP=1)
            import ctypes
            buf = (ctypes.c_ulonglong * 16)()
            torch.cuda.synchronize()
            genops.libssnode.ssn_debug_duo_stamps(buf)
            n = max(int(buf[8]), 1)
            for d in (0, 1):
                print('  draw %d wave 0: chain %.0f  barrier %.0f  serial %.0f  barrier %.0f cycles per step (s_memtime ticks); wave 3: chain %.0f serial %.0f barriers %.0f'
                      % ((d,) + tuple(buf[4 * d + i] / n for i in range(4)) + tuple(buf[9 + 3 * d + i] / n for i in range(3))))
        print('kernel %d: forward%s %.3f ms   max |time_avg - first kernel| %.3e (max %.3e)' % (
            k, ' + stores' if save else '', ms, float((ta - ref).abs().max()), float(ref.abs().max())), flush=True)


if __name__ == '__main__':
    main()
