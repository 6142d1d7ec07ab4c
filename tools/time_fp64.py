"""fp64 fixed-point solver at the reference's default size (N = 102 -> 2N = 204, 8 bandwidths): resident tile kernel
(library default) against the streaming kernel (variant 0)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tc_gan_amd.ssnode import fixed_points_batch
from tc_gan_amd import weight_gen, stimuli
N, B, NB, T = 102, 256, 8, 2000
J = np.array([[.0957, .0638], [.1197, .0479]]); D = np.array([[.7660, .5106], [.9575, .3830]]); S = np.array([[.6667, .2], [1.333, .2]]) / 8
rs = np.random.RandomState(0)
W = weight_gen.generate_weight_batch(N, J + D / 4, D / 2, S, rs.rand(B, 2 * N, 2 * N), dtype='float64')
exts = stimuli.input([0, 0.0625, 0.125, 0.1875, 0.25, 0.5, 0.75, 1], np.linspace(-.5, .5, N), 0.25 / 8, [20.])
out = {}
for name, variant in [('resident tile', None), ('streaming', 0)]:
    fixed_points_batch(W, exts, 0.01, 2.2, max_iter=50, atol=0.0, dtype='float64', variant=variant, return_torch=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = fixed_points_batch(W, exts, 0.01, 2.2, max_iter=T, atol=0.0, dtype='float64', variant=variant, return_torch=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    out[name] = dt
    print('%-14s %.1f ms  %.3e SSN-steps/s' % (name, dt * 1e3, 2.0 * N * B * NB * T / dt))
print('speed-up %.2fx' % (out['streaming'] / out['resident tile']))
