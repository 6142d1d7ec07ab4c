"""Stimulus profiles -- mirror of ``tc_gan/stimuli.py`` (lines 3-10), device evaluated.

``input(bv, x, l, c, o)`` returns, contrast-major then offset then bandwidth,
rows ``con * band(x - off, b, l)`` duplicated for the E and I populations.  The
arithmetic runs on the GPU: the HIP kernel ``ssn_stimulus_f64`` when `x` is the
canonical ``linspace(-.5, .5, N)`` grid without offsets (the only form the hot
path uses, networks/ssn.py:167-193), elementwise torch-on-CUDA otherwise.
"""
import ctypes

import numpy as np

from . import clib
from .clib import libssnode
from .utils import to_device


def stimulus_batch(bandwidths, contrasts, smoothness, num_sites, dtype='float32', amp=None, zin=None, v=None):
    """Device form: `bandwidths`, `contrasts` of shape (B, NB) -> CUDA tensor (B, NB, 2*num_sites)
    (networks/ssn.py:177-188).  `amp` (B, 2*num_sites): per-draw input amplification of the
    heterogeneous-input SSN (ssn.py:679-686) -- or `zin` (B, 2*num_sites) and `v` (1, 2 or 2*num_sites values), float32 device
    tensors: amp = 1 + v * zin formed in the stimulus launch itself (`ssn_stimulus_hetero_f32`; the bits of the torch expression)."""
    import torch
    clib.require_gpu()
    td = {'float32': torch.float32, 'float64': torch.float64}[str(np.dtype(dtype))]
    bw = to_device(bandwidths, td).contiguous()
    con = to_device(contrasts, td).contiguous()
    assert bw.shape == con.shape and bw.dim() == 2
    B, NB = bw.shape
    ext = torch.empty((B, NB, 2 * num_sites), device='cuda', dtype=td)
    if zin is not None:
        assert amp is None and td == torch.float32 and zin.is_cuda and v.is_cuda and zin.dtype == v.dtype == torch.float32
        zin, v = zin.contiguous(), v.contiguous().reshape(-1)
        assert zin.shape == (B, 2 * num_sites) and v.numel() in (1, 2, 2 * num_sites)
        clib.check(libssnode.ssn_stimulus_hetero_f32(bw.data_ptr(), con.data_ptr(), ctypes.c_float(smoothness), zin.data_ptr(),
                                                     v.data_ptr(), int(v.numel()), ext.data_ptr(), int(B), int(NB), int(num_sites),
                                                     clib.stream_ptr()), 'ssn_stimulus_hetero_f32')
        return ext
    fn, ct = ((libssnode.ssn_stimulus_amp_f32, ctypes.c_float) if td == torch.float32
              else (libssnode.ssn_stimulus_amp_f64, ctypes.c_double))
    if amp is not None:
        amp = to_device(amp, td).contiguous()
        assert amp.shape == (B, 2 * num_sites)
    clib.check(fn(bw.data_ptr(), con.data_ptr(), ct(smoothness), amp.data_ptr() if amp is not None else None,
                  ext.data_ptr(), int(B), int(NB), int(num_sites),
                  clib.stream_ptr()), 'ssn_stimulus')
    return ext


def sigm(x, l=.1):
    import torch
    return 1. / (1 + torch.exp(-x / l))


def band(x, b, l=.1):
    return sigm(x + (b / 2), l) * sigm((b / 2) - x, l)


def input(bv, x, l=.1, c=[20.], o=[0.]):
    import torch
    clib.require_gpu()
    x = np.asarray(x, dtype='double')
    bv = [float(b) for b in bv]
    c = [float(v) for v in c]
    o = [float(v) for v in o]
    N = len(x)
    canonical = (N > 1 and list(o) == [0.0] and
                 np.array_equal(x, np.linspace(-0.5, 0.5, N)))
    if canonical:
        bw = np.tile(np.asarray(bv), len(c))[None, :]
        con = np.repeat(np.asarray(c), len(bv))[None, :]
        return stimulus_batch(bw, con, l, N, dtype='float64')[0].cpu().numpy()
    xd = torch.as_tensor(x, device='cuda')
    rows = []
    for con in c:
        for off in o:
            for b in bv:
                prof = band(xd - off, b, l)
                rows.append(con * torch.cat([prof, prof]))
    return torch.stack(rows).cpu().numpy()
