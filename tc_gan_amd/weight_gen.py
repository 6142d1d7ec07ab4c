"""Connectivity from noise -- mirror of ``tc_gan/weight_gen.py`` (lines 6-35),
evaluated by the HIP kernel ``ssn_build_w_*``
(== gradient_expressions/make_w_batch.py:8-34 ``make_W_with_x``)."""
import ctypes

import numpy

from . import clib
from .clib import libssnode


def generate_weight_batch(N, J, delta, sigma, z, dtype='float32'):
    """Device form: z (B, 2N, 2N) array or CUDA tensor -> CUDA tensor W (B, 2N, 2N)."""
    import torch
    clib.require_gpu()
    td = {'float32': torch.float32, 'float64': torch.float64}[str(numpy.dtype(dtype))]
    if isinstance(z, torch.Tensor):
        dz = z.to('cuda', td).contiguous()
    else:
        dz = torch.as_tensor(numpy.ascontiguousarray(z)).to('cuda', td).contiguous()
    assert dz.dim() == 3 and dz.shape[1] == dz.shape[2] == 2 * N
    W = torch.empty_like(dz)
    if td == torch.float32:
        ct, fn = ctypes.c_float, libssnode.ssn_build_w_f32
    else:
        ct, fn = ctypes.c_double, libssnode.ssn_build_w_f64
    arrs = [(ct * 4)(*numpy.asarray(a, dtype='double').reshape(4)) for a in (J, delta, sigma)]
    clib.check(fn(dz.data_ptr(), arrs[0], arrs[1], arrs[2], W.data_ptr(), int(dz.shape[0]), int(N),
                  clib.stream_ptr()), 'ssn_build_w')
    return W


def generate_weight(N, J, delta, sigma, z):
    """
    Generate 2N-by-2N connectivity matrix (weight_gen.py:13-26); numpy in, numpy out (fp64).
    """
    z = numpy.asarray(z, dtype='double')
    return generate_weight_batch(N, J, delta, sigma, z[None], dtype='float64')[0].cpu().numpy()


def generate_parameter(N, J, delta, sigma, seed=None):
    """
    Generate 2N-by-2N connectivity matrix and "latent" variable z (weight_gen.py:29-35).
    """
    rs = numpy.random.RandomState(seed)
    z = rs.uniform(size=(2 * N, 2 * N))
    return generate_weight(N, J, delta, sigma, z), z
