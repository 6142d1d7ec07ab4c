"""Thin Python wrappers over the fixed-time generator kernels (include/ssnode_mi355x.h section 3).

Everything here takes and returns torch CUDA tensors; the arithmetic is in
``csrc/ssn_gen.hip`` / ``ssn_mfma.hip`` (forward + BPTT adjoint) and ``csrc/ssn_gw.hip``
(dL/dW = delta^T . traj, a hand-written batched GEMM on the bf16 matrix cores with exactly split fp32 operands).
"""
import ctypes

import numpy as np
import torch

from . import clib
from .clib import libssnode

_DT = {torch.float32: ('f32', ctypes.c_float), torch.float64: ('f64', ctypes.c_double)}


def _stream():
    return clib.stream_ptr()


def make_gen_params(io_type='asym_tanh', k=0.01, n=2.2, tau_E=10., tau_I=1., dt=0.1, seqlen=1200,
                    skip_steps=1000, rate_soft_bound=200., rate_hard_bound=1000.,
                    rate_penalty_threshold=200., kernel=0):
    """Defaults: networks/wgan.py:39-63 (tau_E=10, tau_I=1, dt=0.1, seqlen=1200, skip_steps=1000).
    kernel: 0 library default (MFMA kernels for fp32 with NB >= 4 and enough draws, VALU tile kernels otherwise),
    1 tile, 2 fp32 MFMA (two 4-stimulus groups per workgroup), 3 fp32 MFMA (one group per workgroup), 4 / 5 forward on
    the fp16-split MFMA kernel (asym_tanh only; the default where it applies unless SSN_FWD_SPLIT=0) and adjoint sweep
    on its fp16-split form (any I/O function)."""
    return clib.GenParams(io_type=clib.IO_CODES[io_type], seqlen=int(seqlen), skip_steps=int(skip_steps),
                          kernel=int(kernel), k=float(k), n=float(n), tau_E=float(tau_E), tau_I=float(tau_I),
                          dt=float(dt), rate_soft_bound=float(rate_soft_bound),
                          rate_hard_bound=float(rate_hard_bound),
                          rate_penalty_threshold=float(rate_penalty_threshold))


_PEN_SCRATCH = {}
# Measurement hook (bench.py): a list to which every plain (save=False) `gen_forward` call appends the pair of events
# recorded around its forward launch -- the kernel's duration INSIDE a running loop, on its launch stream; None = off.
FORWARD_EVENTS = None


def _penalty_scratch(device):
    """Per (device, stream) scratch of ssn_penalty_means_* (partials + ticket; zeroed once, the kernel keeps the ticket zero)."""
    key = (device.index, clib.stream_ptr().value)
    ws = _PEN_SCRATCH.get(key)
    if ws is None:
        ws = _PEN_SCRATCH[key] = torch.zeros(2 * 256 + 1, device=device, dtype=torch.float64)
    return ws


def forward_variant(B, NB, M, gp, save=False):
    """The fp32 forward kernel `gen_forward` runs for this shape (``ssn_gen_forward_variant``: 1 VALU, 2 / 3 fp32 MFMA,
    4 / 5 fp16-split MFMA, -1 refused)."""
    return int(libssnode.ssn_gen_forward_variant(int(B), int(NB), int(M), int(gp.seqlen), int(bool(save)), ctypes.byref(gp)))


def gen_forward(W, ext, gp, save=False, probe=None):
    """W (B, M, M), ext (B, NB, M) CUDA tensors -> dict(time_avg, dynamics_penalty, rate_penalty[, traj, df]).

    dynamics_penalty / rate_penalty are the means of networks/ssn.py:626,632 (0-dim tensors).
    ``probe`` = (ids, probes), int64 CUDA tensors of one length: the conditional prober's gather
    ``tuning_curve[k] = time_avg[ids[k], :, probes[k]]`` (cwgan.py:91-98) rides in the launch that forms the penalty means
    and comes back as ``out['tuning_curve']``."""
    clib.require_gpu()
    assert W.is_cuda and ext.is_cuda and W.dtype == ext.dtype and W.dtype in _DT
    W = W.contiguous(); ext = ext.contiguous()
    B, NB, M = ext.shape
    assert W.shape == (B, M, M)
    T, skip = gp.seqlen, gp.skip_steps
    suffix, _ = _DT[W.dtype]
    ta3 = torch.empty((3,) + tuple(ext.shape), device=ext.device, dtype=ext.dtype)      # one allocation: time_avg, dyn_row, rate_row
    ta, dyn, rate = ta3[0], ta3[1], ta3[2]
    traj = df = None
    if save:
        traj = torch.empty((B, NB, T, M), device=W.device, dtype=W.dtype)
        df = torch.empty_like(traj)
    ev = None
    if FORWARD_EVENTS is not None and not save:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    rc = getattr(libssnode, 'ssn_gen_forward_' + suffix)(
        W.data_ptr(), ext.data_ptr(), ta.data_ptr(), dyn.data_ptr(), rate.data_ptr(),
        traj.data_ptr() if save else None, df.data_ptr() if save else None,
        B, NB, M, ctypes.byref(gp), _stream())
    clib.check(rc, 'ssn_gen_forward_' + suffix)
    if ev is not None:
        ev[1].record()
        FORWARD_EVENTS.append(ev)
    n_dyn = B * (T - skip - 1) * NB * M
    n_rate = B * (T - skip) * NB * M
    # both penalty means in one launch (fp64 sums in a fixed order), instead of two reductions and two scalings
    pens = torch.empty(2, device=W.device, dtype=torch.float64)
    tc = None
    if probe is not None:
        ids, pr = probe
        assert ids.dtype == pr.dtype == torch.int64 and ids.is_cuda and pr.is_cuda and ids.numel() == pr.numel()
        tc = torch.empty((ids.numel(), NB), device=W.device, dtype=W.dtype)
        rc = getattr(libssnode, 'ssn_penalty_means_probe_' + suffix)(
            dyn.data_ptr(), rate.data_ptr(), dyn.numel(), (1.0 / n_dyn) if n_dyn > 0 else float('nan'), 1.0 / n_rate,
            _penalty_scratch(W.device).data_ptr(), pens.data_ptr(), ta.data_ptr(), ids.contiguous().data_ptr(),
            pr.contiguous().data_ptr(), tc.data_ptr(), int(ids.numel()), int(NB), int(M), _stream())
        clib.check(rc, 'ssn_penalty_means_probe_' + suffix)
    else:
        rc = getattr(libssnode, 'ssn_penalty_means_' + suffix)(
            dyn.data_ptr(), rate.data_ptr(), dyn.numel(), (1.0 / n_dyn) if n_dyn > 0 else float('nan'), 1.0 / n_rate,
            _penalty_scratch(W.device).data_ptr(), pens.data_ptr(), _stream())
        clib.check(rc, 'ssn_penalty_means_' + suffix)
    out = dict(time_avg=ta, dynamics_penalty=pens[0], rate_penalty=pens[1], penalties=pens, n_dyn=n_dyn, n_rate=n_rate)
    if tc is not None:
        out['tuning_curve'] = tc
    if save:
        out.update(traj=traj, df=df)
    return out


def gen_backward(W, traj, df, g_time_avg, c_dyn, c_rate, gp, want_g_ext=False, want_dmax=False):
    """Adjoint sweep; `df` is overwritten by the shifted delta and returned (with dL/d ext when asked).
    c_dyn / c_rate multiply SUM(dyn_row) / SUM(rate_row) in the loss.
    ``want_dmax`` (fp32): the last element of the returned tuple is max |delta| per draw, a (B,) tensor, when the sweep
    that ran tracks it (``ssn_gen_backward_max_f32``: the fp16-split kernels 4 / 5 / 6 / 8), else None -- the bound `weight_grad` needs
    for its fp16 form."""
    clib.require_gpu()
    B, NB, T, M = traj.shape
    suffix, _ = _DT[W.dtype]
    g_time_avg = g_time_avg.to(W.dtype).contiguous()
    g_ext = torch.empty((B, NB, M), device=W.device, dtype=W.dtype) if want_g_ext else None
    dmax = None
    if want_dmax and suffix == 'f32':
        dmax = torch.empty((B,), device=W.device, dtype=torch.float32)
        tracked = ctypes.c_int(0)
        rc = libssnode.ssn_gen_backward_max_f32(
            W.data_ptr(), traj.data_ptr(), df.data_ptr(), g_time_avg.data_ptr(),
            g_ext.data_ptr() if want_g_ext else None, dmax.data_ptr(), ctypes.byref(tracked), float(c_dyn), float(c_rate),
            B, NB, M, ctypes.byref(gp), _stream())
        clib.check(rc, 'ssn_gen_backward_max_f32')
        if not tracked.value:
            dmax = None
    else:
        rc = getattr(libssnode, 'ssn_gen_backward_ext_' + suffix)(
            W.data_ptr(), traj.data_ptr(), df.data_ptr(), g_time_avg.data_ptr(),
            g_ext.data_ptr() if want_g_ext else None, float(c_dyn), float(c_rate),
            B, NB, M, ctypes.byref(gp), _stream())
        clib.check(rc, 'ssn_gen_backward_ext_' + suffix)
    out = (df,) + ((g_ext,) if want_g_ext else ()) + ((dmax,) if want_dmax else ())
    return out if len(out) > 1 else df


def gen_backward_fused_supported(B, NB, M, gp, xmax):
    """Whether `gen_backward_fused` takes this shape (``ssn_gen_backward_fused_supported``)."""
    return xmax is not None and bool(libssnode.ssn_gen_backward_fused_supported(int(B), int(NB), int(M), ctypes.byref(gp), float(xmax)))


def gen_backward_fused(W, traj, df, g_time_avg, c_dyn, c_rate, gp, xmax, want_g_ext=False):
    """Adjoint sweep and dL/dW in one launch (``ssn_gen_backward_fused_f32``): returns (gW (B, M, M), g_ext or None,
    dmax (B,)); `df` is only read.  What `gen_backward(..., want_dmax=True)` + `weight_grad(..., dmax, xmax)` compute."""
    clib.require_gpu()
    B, NB, T, M = traj.shape
    assert W.dtype == torch.float32 and df.shape == traj.shape
    g_time_avg = g_time_avg.to(W.dtype).contiguous()
    g_ext = torch.empty((B, NB, M), device=W.device, dtype=W.dtype) if want_g_ext else None
    gW = torch.empty((B, M, M), device=W.device, dtype=W.dtype)
    dmax = torch.empty((B,), device=W.device, dtype=torch.float32)
    rc = libssnode.ssn_gen_backward_fused_f32(
        W.data_ptr(), traj.data_ptr(), df.data_ptr(), g_time_avg.data_ptr(), g_ext.data_ptr() if want_g_ext else None,
        gW.data_ptr(), dmax.data_ptr(), float(xmax), float(c_dyn), float(c_rate), B, NB, M, ctypes.byref(gp), _stream())
    clib.check(rc, 'ssn_gen_backward_fused_f32')
    return gW, g_ext, dmax


def rate_bound(gp):
    """A bound on every rate of the fixed-time generator started at 0, or None: with the saturating I/O function and
    dt <= tau each Euler step is a convex combination of the state and f(u) <= rate_hard_bound (networks/ssn.py:566-576)."""
    if gp.io_type == clib.IO_CODES['asym_tanh'] and 0 < gp.dt <= min(gp.tau_E, gp.tau_I) and 0 < gp.rate_hard_bound < float('inf'):
        return float(gp.rate_hard_bound)
    return None


def weight_grad(delta, traj, kernel=0, dmax=None, xmax=None):
    """dL/dW[b] = delta[b]^T . traj[b] over K = NB*T (``ssn_weight_grad_*``; kernel: 0 automatic, 1 plain FMAs,
    2 split-bf16 MFMA: three bf16 parts per operand, six partial products; 3 the fp16 form of ``ssn_weight_grad_scaled_f32``:
    two fp16 parts per operand by round to nearest, three partial products -- it needs ``dmax`` (B,) >= max |delta[b]| on the
    device and a scalar ``xmax`` >= max |traj|, and with kernel 0 it is taken whenever both are given)."""
    clib.require_gpu()
    B, NB, T, M = traj.shape
    suffix, _ = _DT[traj.dtype]
    delta = delta.contiguous(); traj = traj.contiguous()
    assert delta.shape == traj.shape and delta.dtype == traj.dtype
    gW = torch.empty((B, M, M), device=traj.device, dtype=traj.dtype)
    scaled_ok = suffix == 'f32' and dmax is not None and xmax is not None and 32 < M <= 224 and NB * T * M < (1 << 29)
    if kernel == 3 and not scaled_ok:
        raise ValueError('weight_grad kernel 3 needs fp32, 32 < M <= 224, dmax and xmax')
    if scaled_ok and kernel in (0, 3):
        assert dmax.shape == (B,) and dmax.dtype == torch.float32 and dmax.is_cuda
        rc = libssnode.ssn_weight_grad_scaled_f32(delta.data_ptr(), traj.data_ptr(), gW.data_ptr(), B, NB * T, M,
                                                  dmax.contiguous().data_ptr(), float(xmax), _stream())
        clib.check(rc, 'ssn_weight_grad_scaled_f32')
        return gW
    rc = getattr(libssnode, 'ssn_weight_grad_' + suffix)(delta.data_ptr(), traj.data_ptr(), gW.data_ptr(), B, NB * T, M,
                                                         int(kernel), _stream())
    clib.check(rc, 'ssn_weight_grad_' + suffix)
    return gW


def jds_grad_parts(gW, z, J, D, S):
    """Chain rule through make_W_with_x per draw: the (B, 4, 3) float64 CUDA tensor of ``ssn_jds_grad_*``
    ([:, pq, 0 / 1 / 2] = this draw's share of dL/dJ_pq, dL/dD_pq, dL/dS_pq); `jds_grad` or `ssn_gen_grads_f32` add the draws."""
    clib.require_gpu()
    B, M, _ = gW.shape
    suffix, ct = _DT[gW.dtype]
    arrs = [(ct * 4)(*np.asarray(a, dtype='double').reshape(4)) for a in (J, D, S)]
    out = torch.empty((B, 4, 3), device=gW.device, dtype=torch.float64)
    rc = getattr(libssnode, 'ssn_jds_grad_' + suffix)(gW.contiguous().data_ptr(), z.contiguous().data_ptr(),
                                                      arrs[0], arrs[1], arrs[2], out.data_ptr(), B, M // 2, _stream())
    clib.check(rc, 'ssn_jds_grad_' + suffix)
    return out


_GRADS_WS = {}


def gen_grads(parts, dmean, pens64, dynamics_cost, rate_cost, nv=0, g_ext=None, ext_base=None, zin=None):
    """``ssn_gen_grads_f32``: the flat gradient vector [dL/dV (nv), dL/dJ (4), dL/dD (4), dL/dS (4)] followed by the step's loss
    -- (nv + 13,) float32 on the device, one launch, fp64 sums in a fixed order."""
    clib.require_gpu()
    B = parts.shape[0]
    assert parts.shape == (B, 4, 3) and parts.dtype == torch.float64 and parts.is_contiguous()
    key = (parts.device.index, clib.stream_ptr().value)
    ws = _GRADS_WS.get(key)
    if ws is None:
        ws = _GRADS_WS[key] = torch.zeros(int(libssnode.ssn_gen_grads_ws_doubles()), device=parts.device, dtype=torch.float64)
    out = torch.empty(nv + 13, device=parts.device, dtype=torch.float32)
    a = clib.GenGrads(jds_part=parts.data_ptr(), B=int(B), nv=int(nv), dmean=dmean.data_ptr(),
                      pens64=pens64.data_ptr() if pens64 is not None else None, dynamics_cost=float(dynamics_cost),
                      rate_cost=float(rate_cost), ws=ws.data_ptr(), out=out.data_ptr())
    if nv:
        assert g_ext.dtype == ext_base.dtype == zin.dtype == torch.float32 and g_ext.shape == ext_base.shape
        g_ext, ext_base, zin = g_ext.contiguous(), ext_base.contiguous(), zin.contiguous()
        a.g_ext, a.ext_base, a.zin = g_ext.data_ptr(), ext_base.data_ptr(), zin.data_ptr()
        a.NB, a.M = int(g_ext.shape[1]), int(g_ext.shape[2])
    clib.check(libssnode.ssn_gen_grads_f32(ctypes.byref(a), _stream()), 'ssn_gen_grads_f32')
    return out


def jds_grad(gW, z, J, D, S, as_tensor=False):
    """Chain rule through make_W_with_x: returns (gJ, gD, gS) as float64 (2, 2) arrays -- numpy by default, CUDA
    tensors with ``as_tensor=True`` (no host synchronisation)."""
    clib.require_gpu()
    B, M, _ = gW.shape
    N = M // 2
    suffix, ct = _DT[gW.dtype]
    arrs = [(ct * 4)(*np.asarray(a, dtype='double').reshape(4)) for a in (J, D, S)]
    out = torch.empty((B, 4, 3), device=gW.device, dtype=torch.float64)
    rc = getattr(libssnode, 'ssn_jds_grad_' + suffix)(gW.contiguous().data_ptr(), z.contiguous().data_ptr(),
                                                      arrs[0], arrs[1], arrs[2], out.data_ptr(), B, N, _stream())
    clib.check(rc, 'ssn_jds_grad_' + suffix)
    tot = out.sum(dim=0)                         # (4, 3), fixed summation order
    if as_tensor:
        return tot[:, 0].reshape(2, 2), tot[:, 1].reshape(2, 2), tot[:, 2].reshape(2, 2)
    tot = tot.cpu().numpy()
    return tot[:, 0].reshape(2, 2), tot[:, 1].reshape(2, 2), tot[:, 2].reshape(2, 2)
