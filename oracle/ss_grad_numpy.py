"""TEST INFRASTRUCTURE ONLY.  numpy restatement of the reference's fixed-point implicit gradient
(tc_gan/gradient_expressions/SS_grad.py:17-99) and of the W-derivative tensors (make_w_batch.py:36-121),
straight from the formulas (dense inverse per (draw, stimulus)).  Parity unpinned against the reference (Theano
graphs, no fixtures; its own test only prints the values, tests/test_dynamics.py:249-275): pinned here by finite
differences of the pinned fixed-point solver (tests/test_ss_grad_gpu.py)."""
import numpy as np

SIGN = np.array([[1, -1], [1, -1]], dtype='float64')        # make_w_batch.py:5


def _wnn(S, N):
    x = np.linspace(-0.5, 0.5, N)
    xx = (x[:, None] - x[None, :]).reshape(1, N, 1, N)       # x_i - x_j
    s = np.asarray(S, dtype='float64').reshape(2, 1, 2, 1)
    return np.exp(-xx ** 2 / (2 * s ** 2)), xx, s            # [2, N, 2, N]


def make_W(Z, J, D, S, N):
    """make_w_batch.py:8-34."""
    wnn, _, _ = _wnn(S, N)
    j = (SIGN * J).reshape(2, 1, 2, 1)
    d = (SIGN * D).reshape(2, 1, 2, 1)
    z = np.asarray(Z, dtype='float64').reshape(-1, 2, N, 2, N)
    return (wnn * (j + d * z)).reshape(-1, 2 * N, 2 * N)


def make_dW(which, Z, J, D, S, N):
    """make_w_batch.py:36-121 with identity d theta'/d theta -> [nz (1 for J), 2N, 2N, 2, 2]."""
    wnn, xx, s = _wnn(S, N)
    z = np.asarray(Z, dtype='float64').reshape(-1, 2, N, 2, N)
    nz = 1 if which == 'J' else z.shape[0]
    out = np.zeros((nz, 2, N, 2, N, 2, 2))
    j = (SIGN * J).reshape(2, 1, 2, 1)
    d = (SIGN * D).reshape(2, 1, 2, 1)
    if which == 'J':
        base = (wnn * SIGN.reshape(2, 1, 2, 1))[None]
    elif which == 'D':
        base = wnn * SIGN.reshape(2, 1, 2, 1) * z
    else:
        base = wnn * (2 * xx ** 2 / (2 * s ** 3)) * (j + d * z)
    for p in range(2):
        for q in range(2):
            out[:, p, :, q, :, p, q] = base[:, p, :, q, :]
    return out.reshape(nz, 2 * N, 2 * N, 2, 2)


def phi(V, io_type, k, n, r0, r1):
    """SS_grad.py:76-99."""
    v0 = (r0 / k) ** (1.0 / n)
    Vc = np.maximum(V, 0.0)
    with np.errstate(divide='ignore', invalid='ignore'):
        low = np.where(Vc > 0, n * k * Vc ** (n - 1.0), 0.0)
    if io_type == 'asym_power':
        return low
    if io_type == 'asym_linear':
        return n * k * np.clip(V, 0.0, v0) ** (n - 1.0) * (V > 0)
    arg = (n * r0 / v0) * (Vc - v0) / (r1 - r0)
    high = (n * r0 / v0) * np.cosh(arg) ** -2.0
    return np.where(Vc <= v0, low, high)


def WRgrad_batch(R, W, DW, I, n, k, io_type='asym_tanh', r0=200.0, r1=1000.0):
    """SS_grad.py:17-74: R [nz, nb, M], W [nz, M, M], DW [nz or 1, M, M, 2, 2], I [nb, M] or [nz, nb, M]."""
    R, W, DW, I = (np.asarray(a, dtype='float64') for a in (R, W, DW, I))
    nz, nb, M = R.shape
    V = np.einsum('zij,zbj->zbi', W, R) + (I if I.ndim == 3 else I[None])
    ph = phi(V, io_type, k, n, r0, r1)
    out = np.zeros((nz, nb, M, 2, 2))
    for z in range(nz):
        dw = DW[z if DW.shape[0] == nz else 0]
        for b in range(nb):
            A = np.eye(M) - ph[z, b][:, None] * W[z]
            B = ph[z, b][:, None, None] * np.einsum('ijpq,j->ipq', dw, R[z, b])
            out[z, b] = np.linalg.solve(A, B.reshape(M, 4)).reshape(M, 2, 2)
    return out
