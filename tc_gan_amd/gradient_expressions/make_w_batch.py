"""W and its parameter derivatives from the latent noise z, on the GPU -- numeric mirror of
``tc_gan/gradient_expressions/make_w_batch.py`` (the reference builds Theano expressions; these functions
evaluate them).  X is the site grid; only the reference's own grid ``linspace(-.5, .5, N)`` is built into
the kernels (make_w_batch.py is always called with it: run/gan.py:376, tests/test_dynamics.py:180)."""
import ctypes

import numpy as np
import torch

from .. import clib
from ..clib import libssnode
from ..weight_gen import generate_weight_batch

sign = np.array([[1, -1], [1, -1]], dtype='int16')      # make_w_batch.py:5


def _check_grid(N, X):
    if X is not None and not np.allclose(np.asarray(X, dtype='float64'), np.linspace(-0.5, 0.5, N)):
        raise NotImplementedError('only X = linspace(-.5, .5, N) is built into the kernels')


def make_W_with_x(Z, J, D, S, N, X=None):
    """make_w_batch.py:8-34: W (nz, 2N, 2N) as a CUDA tensor (dtype of Z; float32 for numpy input)."""
    _check_grid(N, X)
    dtype = 'float64' if (torch.is_tensor(Z) and Z.dtype == torch.float64) else \
        ('float64' if (not torch.is_tensor(Z) and np.asarray(Z).dtype == np.float64) else 'float32')
    return generate_weight_batch(N, J, D, S, Z, dtype=dtype)


def _make_dW(which, Z, J, D, S, N, X, dtheta):
    _check_grid(N, X)
    clib.require_gpu()
    if dtheta is not None and not np.allclose(np.asarray(dtheta).reshape(4, 4), np.eye(4)):
        raise NotImplementedError("only the identity Jacobian d theta' / d theta (theta' = theta) is built")
    z = Z if torch.is_tensor(Z) else torch.as_tensor(np.ascontiguousarray(Z))
    td = torch.float64 if z.dtype == torch.float64 else torch.float32
    z = z.to('cuda', td).contiguous()
    nz = 1 if which == 0 else int(z.shape[0])          # dW/dJ does not depend on z (make_w_batch.py:36-63)
    M = 2 * N
    out = torch.empty((nz, M, M, 2, 2), device='cuda', dtype=td)
    ct, fn = (ctypes.c_double, libssnode.ssn_build_dw_f64) if td == torch.float64 else (ctypes.c_float, libssnode.ssn_build_dw_f32)
    arrs = [(ct * 4)(*np.asarray(a, dtype='double').reshape(4)) for a in (J, D, S)]
    clib.check(fn(z.data_ptr(), arrs[0], arrs[1], arrs[2], which, out.data_ptr(), nz, int(N),
                  clib.stream_ptr()), 'ssn_build_dw')
    return out


def make_WJ_with_x(Z, J, D, S, N, X=None, dj=None):
    """make_w_batch.py:36-63: dW/dJ, shape (1, 2N, 2N, 2, 2) (independent of z)."""
    return _make_dW(0, Z, J, D, S, N, X, dj)


def make_WD_with_x(Z, J, D, S, N, X=None, dd=None):
    """make_w_batch.py:65-93: dW/dD, shape (nz, 2N, 2N, 2, 2)."""
    return _make_dW(1, Z, J, D, S, N, X, dd)


def make_WS_with_x(Z, J, D, S, N, X=None, ds=None):
    """make_w_batch.py:95-121: dW/dS, shape (nz, 2N, 2N, 2, 2)."""
    return _make_dW(2, Z, J, D, S, N, X, ds)
