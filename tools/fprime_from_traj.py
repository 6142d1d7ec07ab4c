"""VERDICT r3 item 1, second half: can the adjoint sweep do without the f'(u) stream (7.9 GB written by the saving forward,
7.9 GB read back) by recovering f(u_t) from two consecutive trajectory rows,  f(u_t) = (r_t - (1 - eps) r_{t-1}) / eps,  and
f' from f (power-law branch: f' = n k^(1/n) f^(1 - 1/n))?  This script prices the ACCURACY side on the CPU: an fp32 Euler
trajectory of the C3 generator (2N = 200, 8 stimuli, 1200 steps; numpy, the arithmetic of networks/ssn.py:566-576), f' both ways,
errors relative to the largest f' of the step.  No GPU, no reference import.  Result: see the printout."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ssn_numpy as on          # weights, stimuli (test infrastructure; this is a measurement tool, not product)

N, NB, T = 100, 8, 1200
k, n = np.float32(0.01), np.float32(2.2)
rs = np.random.RandomState(0)
jds = on.new_JDS()
W = on.generate_weight(N, jds['J'], jds['D'], jds['S'], rs.rand(2 * N, 2 * N)).astype('float32')
bw = np.resize(np.asarray(on.DEFAULT_PARAMS['bandwidths']), NB)
ext = on.stimulus_input(bw, np.linspace(-.5, .5, N), on.DEFAULT_PARAMS['smoothness'], [20.]).astype('float32')   # (NB, 2N)
eps = np.concatenate([np.full(N, 0.01, 'float32'), np.full(N, 0.1, 'float32')])           # dt / tau: tau_E = 10, tau_I = 1, dt = 0.1
r = np.zeros((NB, 2 * N), 'float32')
worst = dict(all=0.0, silent=0.0, active=0.0)
spurious = []
for t in range(T):
    u = r @ W.T + ext
    up = np.maximum(u, np.float32(0))
    f = k * up ** n                                        # (rates stay far below the saturating branch here)
    fp_true = np.where(u > 0, n * k * up ** (n - 1), 0).astype('float32')
    r1 = ((1 - eps) * r + eps * f).astype('float32')
    f_rec = ((r1 - (1 - eps) * r) / eps).astype('float32')                      # what the adjoint would recompute
    f_rec = np.maximum(f_rec, 0)
    fp_rec = (n * k ** (1 / n) * f_rec ** (1 - 1 / n)).astype('float32')
    scale = fp_true.max()
    err = np.abs(fp_rec - fp_true) / scale
    worst['all'] = max(worst['all'], err.max())
    if (u <= 0).any():
        worst['silent'] = max(worst['silent'], err[u <= 0].max())
        spurious.append(float(fp_rec[u <= 0].max() / scale))
    worst['active'] = max(worst['active'], err[u > 0].max())
    r = r1
print('max |f\'_recovered - f\'| / max f\' over %d steps: all rows %.2e, rows with u > 0 %.2e, silent rows (true f\' = 0) %.2e'
      % (T, worst['all'], worst['active'], worst['silent']))
print('largest spurious f\' of a silent row relative to the largest f\' of its step: %.2e (median over steps %.2e)'
      % (max(spurious), float(np.median(spurious))))
