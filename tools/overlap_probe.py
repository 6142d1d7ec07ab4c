#!/usr/bin/env python3
"""How much of a critic update hides behind a generator forward when the two run on different streams?  (C3 shape: the
forward fills every CU with one 512-thread workgroup of 256-VGPR waves, so a second kernel only gets the slots a finished
workgroup leaves.)  Prints: forward alone, critic update alone, both one after the other, both on two streams."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from tc_gan_amd import genops  # noqa: E402


def main():
    torch.cuda.set_device(0)
    gan, (N, models, NB, T, skip), bandwidths = bench.make_c3_gan(1, 0, paper='--paper' in sys.argv, disc_precision='bf16', gen_kernel='auto')
    it = gan.learning()
    for _ in range(8):
        next(it)
    bw = np.tile(np.asarray(bandwidths, dtype='float32')[None], (models, 1))
    noise = gan.gen.gen_noise(None, bw)
    ext, z, W = gan.gen._device_inputs(bw, np.full_like(bw, 20.0), noise['model_zs'], noise.get('model_zs_in'))
    gp = gan.gen.gen_params(200.0)
    nx = gan.disc.nx
    rs = np.random.RandomState(0)
    batch = models
    xg = torch.as_tensor(rs.rand(batch, nx) * 5, device='cuda', dtype=torch.float32)
    xd = torch.as_tensor(rs.rand(batch, nx) * 5, device='cuda', dtype=torch.float32)
    xp = 0.5 * (xg + xd)
    cd = torch.as_tensor(np.stack([np.full(batch, 20.), rs.rand(batch) * 2 - 1, rs.randint(0, 2, batch)], axis=1), device='cuda', dtype=torch.float32)

    def fwd():
        return genops.gen_forward(W, ext, gp)

    def upd():
        gan.disc.loss_grad(xg, cd, xd, cd, xp, cd, 10.0)
        gan.disc_updater(gan.disc.params, gan.disc.grads)
        return gan.disc.accuracy_device(xg, cd, xd, cd)

    side = torch.cuda.Stream()

    def timed(fn, n=10):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    def both_seq():
        fwd(); upd()

    def both_par():
        main_s = torch.cuda.current_stream()
        side.wait_stream(main_s)
        with torch.cuda.stream(side):
            fwd()
        upd()
        main_s.wait_stream(side)

    print('forward alone %.3f ms, critic update alone %.3f ms, one after the other %.3f ms, two streams %.3f ms' % (
        timed(fwd), timed(upd), timed(both_seq), timed(both_par)))


if __name__ == '__main__':
    main()
