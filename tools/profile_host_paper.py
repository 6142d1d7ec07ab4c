"""Where the HOST's time goes in the GAN loop at the paper's shape (the loop is host-bound there: DESIGN 3.8b).
cProfile over N iterations of bench.py's c3paper GAN, top functions by own time and by cumulative time."""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
gan, shape, bandwidths = bench.make_c3_gan(1, 0, paper=True, disc_precision='bf16', gen_kernel='auto')
it = gan.learning()
def one():
    while True:
        info = next(it)
        if not info.is_discriminator:
            return info
for _ in range(10): one()
torch.cuda.synchronize()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
pr = cProfile.Profile()
import time
t0 = time.perf_counter()
pr.enable()
for _ in range(N): one()
pr.disable()
torch.cuda.synchronize()
print('%.3f ms per iteration under cProfile' % ((time.perf_counter() - t0) / N * 1e3))
for key in ('tottime', 'cumulative'):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).strip_dirs().sort_stats(key).print_stats(28)
    print('\n'.join(l[:150] for l in s.getvalue().split('\n')[4:44]))
