#!/usr/bin/env python3
"""HIP-event timing of the fp32 fixed-point solver kernels at the C2 shape with 8 stimuli per draw (2N = 200, 4096 draws,
2000 steps, atol = 0) for a list of variants.  usage: tools/time_solver.py [variant ...] [--draws B]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tc_gan_amd import ssnode  # noqa: E402


def main():
    args = sys.argv[1:]
    B = int(args[args.index('--draws') + 1]) if '--draws' in args else 4096
    variants = [int(a) for i, a in enumerate(args) if a.isdigit() and (i == 0 or args[i - 1] != '--draws')] or [6, 8]
    NB, M, T = 8, 200, 2000
    g = torch.Generator(device='cuda'); g.manual_seed(1)
    W = (torch.rand((B, M, M), device='cuda', generator=g) - 0.6) * 0.02
    ext = torch.rand((NB, M), device='cuda', generator=g) * 20
    ref = None
    for v in variants:
        def run():
            return ssnode.fixed_points_batch(W, ext, 0.01, 2.2, max_iter=T, atol=0.0, dtype='float32', variant=v,
                                             return_torch=True)
        run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            res = run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        if ref is None:
            ref = res.x
        print('variant %d: %.2f ms  %.3e SSN-steps/s   max |x - first variant| %.3e (max %.3e)' % (
            v, ms, M * B * NB * T / (ms * 1e-3), float((res.x - ref).abs().max()), float(ref.abs().max())), flush=True)
        from tc_gan_amd import clib
        if v == 8 and hasattr(clib.libssnode, 'ssn_debug_duo_stamps_fine'):          # diagnostic build (-DSSN_DUO_STAMP=1)
            import ctypes
            buf = (ctypes.c_ulonglong * 64)()
            clib.libssnode.ssn_debug_duo_stamps_fine(buf)
            for w in range(8):
                n = max(int(buf[8 * w + 5]), 1)
                print('  workgroup 0, draw %d, wave %d, cycles per step: chain + early %.0f, verdict %.0f, barrier %.0f, serial %.0f, '
                      'barrier %.0f  (%d steps)' % ((w // 4, w % 4) + tuple(buf[8 * w + i] / n for i in range(5)) + (n,)), flush=True)

if __name__ == '__main__':
    main()
