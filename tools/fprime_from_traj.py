"""VERDICT r3 item 1, second half: can the adjoint sweep do without the f'(u) stream (7.9 GB written by the saving forward,
7.9 GB read back) by recovering f(u_t) from two consecutive trajectory rows,  f(u_t) = (r_t - (1 - eps) r_{t-1}) / eps,  and
f' from f (power-law branch: f' = n k^(1/n) f^(1 - 1/n))?  This script prices the ACCURACY side on the CPU: an fp32 Euler
trajectory of the C3 generator (2N = 200, 8 stimuli, 1200 steps; numpy, the arithmetic of networks/ssn.py:566-576), f' both ways,
errors relative to the largest f' of the step.  No GPU, no reference or oracle import: a synthetic network of the usual shape.  Result: see the printout."""
import numpy as np

N, NB, T = 100, 8, 1200
k, n = np.float32(0.01), np.float32(2.2)
rs = np.random.RandomState(0)
# a synthetic SSN of the usual shape (ring of N sites, Gaussian connection probability, E columns positive, I columns negative;
# the magnitudes of the reference's default J, D, S) and eight centred stimuli of growing width at contrast 20
x = np.linspace(-.5, .5, N)
d2 = (x[:, None] - x[None, :]) ** 2
J = np.array([[.0957, .0638], [.1197, .0479]]); D = np.array([[.7660, .5106], [.9575, .3830]]); S = np.array([[.6667, .2], [1.333, .2]]) / 8
z = rs.rand(2 * N, 2 * N)
W = np.zeros((2 * N, 2 * N))
for a in range(2):
    for b in range(2):
        blk = (z[a * N:(a + 1) * N, b * N:(b + 1) * N] < D[a, b] * np.exp(-d2 / (2 * S[a, b] ** 2))) * J[a, b]
        W[a * N:(a + 1) * N, b * N:(b + 1) * N] = blk * (1 if b == 0 else -1)
W = W.astype('float32')
bw = np.array([0.0625, 0.125, 0.1875, 0.25, 0.5, 0.75, 1.0, 0.03])[:NB]
ext1 = 20.0 / ((1 + np.exp((np.abs(x)[None, :] - bw[:, None] / 2) * 32)))                 # (NB, N): smoothed boxes
ext = np.concatenate([ext1, ext1], axis=1).astype('float32')
eps = np.concatenate([np.full(N, 0.01, 'float32'), np.full(N, 0.1, 'float32')])           # dt / tau: tau_E = 10, tau_I = 1, dt = 0.1
r = np.zeros((NB, 2 * N), 'float32')
worst = dict(all=0.0, silent=0.0, active=0.0)
spurious, own, own_e = [], [], []
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
    act = fp_true > 1e-3 * scale
    if act.any():
        own.append(float((np.abs(fp_rec - fp_true)[act] / fp_true[act]).max()))
        own_e.append(float((np.abs(fp_rec - fp_true)[:, :N][act[:, :N]] / fp_true[:, :N][act[:, :N]]).max()) if act[:, :N].any() else 0.0)
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
print('per value, relative to its OWN f\' (rows above 1e-3 of the step\'s largest): max over the sweep %.2e, median of the per-step maxima %.2e; '
      'excitatory rows (eps = 0.01) alone: %.2e / %.2e;  stored fp32 f\': 6e-8.  Largest rate %.1f' %
      (max(own), float(np.median(own)), max(own_e), float(np.median(own_e)), float(r.max())))
