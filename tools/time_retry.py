#!/usr/bin/env python3
"""Per-iteration wall time of the C3 loop at 2048 models in one process (the size at which this seed's dynamics explode at the
third generator step): what a refused step costs.  usage: tools/time_retry.py [iterations] [models]   (TCGAN_SUBSET_RETRY=0/1, BENCH_GEN_LR=0.01 for the step size of rounds 1-4)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    models = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    gan, _, _ = bench.make_c3_gan(models=models, z_mode='refstream')
    it = gan.learning()
    for k in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        while True:
            info = next(it)
            if not info.is_discriminator:
                break
        torch.cuda.synchronize()
        print('iteration %d: %.1f ms, refused draws %d, gen_loss %.4g' % (k, (time.perf_counter() - t0) * 1e3, gan.gen.poisoned_draws(),
                                                                        info.gen_loss), flush=True)


if __name__ == '__main__':
    main()
