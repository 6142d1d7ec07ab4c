#!/usr/bin/env python3
"""Wall time per iteration of the GAN loop alone (bench.make_c3_gan; no comparison loops): for A/B runs of the host-side switches
(TCGAN_PREQUEUE, TCGAN_MT_TAIL, TCGAN_MT_FUSE_W).  usage: tools/time_gan_loop.py paper|c3 [iterations] [z_mode] [repeats]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    shape = sys.argv[1] if len(sys.argv) > 1 else 'paper'
    n = int(sys.argv[2]) if len(sys.argv) > 2 else (100 if shape == 'paper' else 12)
    z_mode = sys.argv[3] if len(sys.argv) > 3 else 'refstream'
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    gan, _, _ = bench.make_c3_gan(paper=shape == 'paper', z_mode=z_mode)
    it = gan.learning()

    def one_iter():
        while True:
            info = next(it)
            if not info.is_discriminator:
                return info

    for _ in range(3):
        one_iter()
    out = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            one_iter()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / n * 1e3)
    print('%s %s: %s ms per iteration' % (shape, z_mode, ' '.join('%.3f' % t for t in out)), flush=True)


if __name__ == '__main__':
    main()
