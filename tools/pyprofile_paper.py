#!/usr/bin/env python3
"""Host-side profile (cProfile) of the GAN loop at the paper's shape on the reference's noise stream: where the Python time of an
iteration goes (the loop is latency-bound there: DESIGN 3.8b).  usage: tools/pyprofile_paper.py [iterations] [z_mode]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    z_mode = sys.argv[2] if len(sys.argv) > 2 else 'refstream'
    gan, _, _ = bench.make_c3_gan(paper=True, z_mode=z_mode)
    it = gan.learning()

    def one_iter():
        while True:
            info = next(it)
            if not info.is_discriminator:
                return info

    for _ in range(5):
        one_iter()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        one_iter()
    torch.cuda.synchronize()
    print('%s: %.3f ms per iteration un-profiled' % (z_mode, (time.perf_counter() - t0) / n * 1e3))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(n):
        one_iter()
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats('tottime').print_stats(45)
    st.sort_stats('cumulative').print_stats(60)


if __name__ == '__main__':
    main()
