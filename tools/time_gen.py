#!/usr/bin/env python3
"""HIP-event timing of the generator kernels at the C3 shape (1024 draws x 8 stimuli x 1200 steps, 2N = 200):
forward without / with trajectory stores, adjoint sweep, dL/dW batched GEMM.  usage: tools/time_gen.py [kernel]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tc_gan_amd import genops  # noqa: E402


def timed(fn, n=3):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    kernel = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    B, NB, M, T, skip = 1024, 8, 200, 1200, 1000
    g = torch.Generator(device='cuda'); g.manual_seed(1)
    W = (torch.rand((B, M, M), device='cuda', generator=g) - 0.6) * 0.02
    ext = torch.rand((B, NB, M), device='cuda', generator=g) * 20
    gp = genops.make_gen_params(seqlen=T, skip_steps=skip, kernel=kernel)
    print('forward          %.2f ms' % timed(lambda: genops.gen_forward(W, ext, gp)))
    out = {}

    def fwd_save():
        out.update(genops.gen_forward(W, ext, gp, save=True))
    print('forward + stores %.2f ms' % timed(fwd_save))
    gta = torch.rand((B, NB, M), device='cuda', generator=g)
    traj, df = out['traj'], out['df']
    keep = df.clone()

    def bwd():
        df.copy_(keep)
        return genops.gen_backward(W, traj, df, gta, 1.0, 0.01, gp)
    t_copy = timed(lambda: df.copy_(keep))
    print('adjoint sweep    %.2f ms' % (timed(bwd) - t_copy))
    if hasattr(genops.libssnode, 'ssn_debug_split_stamps'):          # diagnostic build (-DSSN_SPLIT_STAMP=1)
        import ctypes
        buf = (ctypes.c_ulonglong * 8)()
        torch.cuda.synchronize()
        genops.libssnode.ssn_debug_split_stamps(buf)
        n = max(int(buf[2]), 1)
        print('  adjoint serial wave 0: serial parts %.0f + barrier waits %.0f cycles per step (two phases)' % (buf[0] / n, buf[1] / n))
    delta = bwd()
    print('dL/dW bmm        %.2f ms' % timed(lambda: genops.weight_grad(delta, traj)))


if __name__ == '__main__':
    main()
