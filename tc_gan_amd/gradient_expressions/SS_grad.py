"""Implicit gradient of the SSN fixed point w.r.t. the connectivity parameters, on the GPU -- numeric mirror of
``tc_gan/gradient_expressions/SS_grad.py``: at a fixed point r = f(W r + I),

    dr/dtheta = (1 - Phi W)^-1 Phi (dW/dtheta r),     Phi = diag f'(W r + I).

``ssn_ss_grad_system_*`` (csrc/ssn_ssgrad.hip) builds the batched systems and ``ssn_lu_solve_*`` (same file) solves
them: Gaussian elimination with partial pivoting, one workgroup per (draw, stimulus) system, in place."""
import ctypes

import numpy as np
import torch

from .. import clib
from ..clib import libssnode
from ..ssnode import DEFAULT_PARAMS


def WRgrad_batch(R, W, DW, I, n, k, nz, nb, N, CGAN=False, io_type=DEFAULT_PARAMS['io_type'],
                 r0=DEFAULT_PARAMS['rate_soft_bound'], r1=DEFAULT_PARAMS['rate_hard_bound']):
    """SS_grad.py:17-74.  R (nz, nb, 2N) fixed points, W (nz, 2N, 2N), DW (nz or 1, 2N, 2N, 2, 2),
    I (nb, 2N) -- or (nz, nb, 2N) with ``CGAN=True`` -- -> dr/dtheta (nz, nb, 2N, 2, 2), a CUDA tensor."""
    clib.require_gpu()
    M = 2 * N

    def dev(x, td=None):
        t = x if torch.is_tensor(x) else torch.as_tensor(np.ascontiguousarray(x))
        return t.to('cuda', td or (torch.float64 if t.dtype == torch.float64 else torch.float32)).contiguous()
    Wd = dev(W)
    td = Wd.dtype
    Rd, DWd, Id = dev(R, td), dev(DW, td), dev(I, td)
    assert Wd.shape == (nz, M, M) and Rd.shape == (nz, nb, M)
    assert DWd.shape[1:] == (M, M, 2, 2) and DWd.shape[0] in (1, nz)
    assert Id.shape == ((nz, nb, M) if CGAN else (nb, M))
    A = torch.empty((nz, nb, M, M), device='cuda', dtype=td)
    rhs = torch.empty((nz, nb, M, 4), device='cuda', dtype=td)
    p = clib.SolverParams(io_type=clib.IO_CODES[io_type], max_iter=0, k=float(k), n=float(n), tau_E=1.0, tau_I=1.0,
                          dt=1.0, atol=0.0, rate_soft_bound=float(r0), rate_hard_bound=float(r1))
    fn = libssnode.ssn_ss_grad_system_f64 if td == torch.float64 else libssnode.ssn_ss_grad_system_f32
    clib.check(fn(Rd.data_ptr(), Wd.data_ptr(), DWd.data_ptr(), int(DWd.shape[0] == nz),
                  Id.data_ptr(), int(bool(CGAN)), int(nz), int(nb), int(M), ctypes.byref(p), A.data_ptr(), rhs.data_ptr(),
                  clib.stream_ptr()), 'ssn_ss_grad_system')
    if nz * nb == 0:
        return rhs.reshape(nz, nb, M, 2, 2)
    info = torch.zeros(nz * nb, device='cuda', dtype=torch.int32)
    fn = libssnode.ssn_lu_solve_f64 if td == torch.float64 else libssnode.ssn_lu_solve_f32
    clib.check(fn(A.data_ptr(), rhs.data_ptr(), info.data_ptr(), int(nz * nb), int(M), 4,
                  clib.stream_ptr()), 'ssn_lu_solve')
    if int(info.count_nonzero()) != 0:          # (the reference's theano `solve` raises on a singular matrix too)
        raise np.linalg.LinAlgError('1 - Phi W is singular for {} of the {} (draw, stimulus) systems'
                                    .format(int(info.count_nonzero()), nz * nb))
    return rhs.reshape(nz, nb, M, 2, 2)         # solved in place
