"""Feed-forward tuning-curve generator (the reference's ``FF_lalazar_model.py`` generator, BASELINE
config 5) on the GPU: `get_FF_output` (FF_functions/lalazar_func.py:16-45) and its parameter gradient.

Parameters follow the model script: the trainable quantities are logs
(``RF_low, RF_del, THR_del, Js, As`` enter as ``exp(.)``, ``THR`` linearly; FF_lalazar_model.py:95-103, 175).
"""
import ctypes

import numpy as np
import torch

from . import clib
from .clib import libssnode

PARAM_NAMES = ('RF_low', 'RF_del', 'THR', 'THR_del', 'Js')          # FF_lalazar_model.py:103 (PARAM)
START_PARAMS = dict(RF_low=np.log(2.), RF_del=np.log(.3), Js=np.log(1000), THR=0., THR_del=np.log(1.),
                    As=np.log(1.))                                    # FF_lalazar_model.py:72


def default_stimuli():
    """FF_lalazar_model.py:139."""
    return np.array([[x, y, z] for x in [-1, 0, 1] for y in [-1, 0, 1] for z in [-1, 0, 1]], dtype='float32')


def generate_samples(rng, nsam, box_width, nhid=1, device=None):
    """FF_lalazar_model.py:154-167: receptive-field widths, sparse 0/1 connectivity (box^3/100 draws with
    replacement per hidden unit), strengths, threshold samples -- drawn on the host like the script does."""
    G = box_width ** 3
    nff = int(G / 100)
    wid = rng.rand(nsam, G)
    con = np.zeros((nsam, nhid, G))
    for s in range(nsam):
        for h in range(nhid):
            con[s, h, rng.choice(G, nff)] = 1
    strn = rng.rand(nsam, nhid, G)
    ths = rng.uniform(-1, 1, [nsam, nhid])
    return con, strn, wid, ths


def _f32(t):
    return torch.as_tensor(t).to('cuda', torch.float32).contiguous()


def _params(p, nsam, nhid, ni, box):
    return clib.FFParams(nsam=nsam, nhid=nhid, ni=ni, box=box,
                         RF_l=float(np.exp(p['RF_low'])), RF_d=float(np.exp(p['RF_del'])), TH=float(p['THR']),
                         TH_d=float(np.exp(p['THR_del'])), J=float(np.exp(p['Js'])), a=float(np.exp(p['As'])))


def ff_forward(params, RF_w, FF_con, FF_str, TH_sam, stim, box_width, keep=False):
    """Hidden activations (nsam, ni, nhid) as a CUDA tensor; `keep` also returns what the backward needs."""
    clib.require_gpu()
    RF_w, FF_con, FF_str, TH_sam, stim = (_f32(t) for t in (RF_w, FF_con, FF_str, TH_sam, stim))
    nsam, nhid = TH_sam.shape
    ni = stim.shape[0]
    RF_w = RF_w.reshape(nsam, -1)
    assert RF_w.shape[1] == box_width ** 3 and FF_con.shape == (nsam, nhid, box_width ** 3) == FF_str.shape
    out = torch.empty((nsam, ni, nhid), device='cuda', dtype=torch.float32)
    q = torch.empty_like(out) if keep else None
    den = torch.empty_like(out) if keep else None
    fp = _params(params, nsam, nhid, ni, box_width)
    st = clib.stream_ptr()
    clib.check(libssnode.ssn_ff_forward_f32(RF_w.data_ptr(), FF_con.data_ptr(), FF_str.data_ptr(), TH_sam.data_ptr(),
                                            stim.data_ptr(), out.data_ptr(), q.data_ptr() if keep else None,
                                            den.data_ptr() if keep else None, ctypes.byref(fp), st), 'ssn_ff_forward_f32')
    if keep:
        return out, dict(q=q, den=den, RF_w=RF_w, FF_con=FF_con, FF_str=FF_str, TH_sam=TH_sam, stim=stim, fp=fp)
    return out


def sparsify(FF_con, FF_str):
    """The connection lists of `ff_forward_sparse` from the dense arrays the model script draws (FF_lalazar_model.py:154-167:
    FF_con has box^3 / 100 ones per unit, set by index, so an index drawn twice counts once): (conn_idx int32, conn_str float32),
    both (nsam, nhid, ncon) CUDA tensors, ncon = the largest connection count of a unit, shorter lists padded with index -1."""
    con = torch.as_tensor(FF_con).to('cuda')
    strn = _f32(FF_str)
    nsam, nhid, G = con.shape
    mask = con != 0
    counts = mask.sum(dim=2)
    ncon = int(counts.max()) if counts.numel() else 0
    # stable sort puts the connected grid points first, in index order
    order = torch.sort((~mask).to(torch.uint8), dim=2, stable=True).indices[:, :, :ncon]
    valid = torch.arange(ncon, device='cuda')[None, None, :] < counts[:, :, None]
    idx = torch.where(valid, order, torch.full_like(order, -1)).to(torch.int32).contiguous()
    val = torch.where(valid, torch.gather(strn, 2, order), torch.zeros((), device='cuda')).contiguous()
    return idx, val


def ff_forward_sparse(params, RF_w, conn_idx, conn_str, TH_sam, stim, box_width, keep=False):
    """`ff_forward` from connection lists (`sparsify`; `ssn_ff_forward_sparse_f32`): the same hidden activations without the
    two dense per-point streams FF_con, FF_str."""
    clib.require_gpu()
    RF_w, TH_sam, stim = (_f32(t) for t in (RF_w, TH_sam, stim))
    conn_idx = torch.as_tensor(conn_idx).to('cuda', torch.int32).contiguous()
    conn_str = _f32(conn_str)
    nsam, nhid = TH_sam.shape
    ni = stim.shape[0]
    RF_w = RF_w.reshape(nsam, -1)
    ncon = conn_idx.shape[2] if conn_idx.dim() == 3 else 0
    assert RF_w.shape[1] == box_width ** 3 and tuple(conn_idx.shape) == (nsam, nhid, ncon) == tuple(conn_str.shape)
    out = torch.empty((nsam, ni, nhid), device='cuda', dtype=torch.float32)
    q = torch.empty_like(out) if keep else None
    den = torch.empty_like(out) if keep else None
    fp = _params(params, nsam, nhid, ni, box_width)
    clib.check(libssnode.ssn_ff_forward_sparse_f32(RF_w.data_ptr(), conn_idx.data_ptr(), conn_str.data_ptr(), int(ncon),
                                                   TH_sam.data_ptr(), stim.data_ptr(), out.data_ptr(),
                                                   q.data_ptr() if keep else None, den.data_ptr() if keep else None,
                                                   ctypes.byref(fp), clib.stream_ptr()), 'ssn_ff_forward_sparse_f32')
    if keep:
        return out, dict(q=q, den=den, RF_w=RF_w, conn_idx=conn_idx, conn_str=conn_str, TH_sam=TH_sam, stim=stim, fp=fp)
    return out


def ff_backward(params, saved, out, g_out):
    """Gradient of  sum(g_out * out)  w.r.t. the five trainable (log-space) parameters, as a dict."""
    gq = (g_out.to(torch.float32) * (out > 0)).contiguous()
    nsam, ni, nhid = out.shape
    dsig = torch.empty((nsam, nhid, 2), device='cuda', dtype=torch.float32)
    st = clib.stream_ptr()
    if 'conn_idx' in saved:            # forward was `ff_forward_sparse`: the same second pass from the connection lists
        clib.check(libssnode.ssn_ff_backward_sparse_f32(saved['RF_w'].data_ptr(), saved['conn_idx'].data_ptr(),
                                                        saved['conn_str'].data_ptr(), int(saved['conn_idx'].shape[2]),
                                                        saved['stim'].data_ptr(), saved['q'].data_ptr(), saved['den'].data_ptr(),
                                                        gq.data_ptr(), dsig.data_ptr(), ctypes.byref(saved['fp']), st),
                   'ssn_ff_backward_sparse_f32')
    else:
        clib.check(libssnode.ssn_ff_backward_f32(saved['RF_w'].data_ptr(), saved['FF_con'].data_ptr(),
                                                 saved['FF_str'].data_ptr(), saved['stim'].data_ptr(),
                                                 saved['q'].data_ptr(), saved['den'].data_ptr(), gq.data_ptr(),
                                                 dsig.data_ptr(), ctypes.byref(saved['fp']), st), 'ssn_ff_backward_f32')
    fp = saved['fp']
    d = dsig.sum(dim=(0, 1), dtype=torch.float64)
    gq64 = gq.to(torch.float64)
    ts = saved['TH_sam'].to(torch.float64)
    pw = torch.sign(ts) * ts.abs() ** fp.a                         # (nsam, nhid)
    g = dict(
        RF_low=float(d[0]) * fp.RF_l,                               # d/d log RF_l
        RF_del=float(d[1]) * fp.RF_d,
        Js=float((gq64 * saved['q'].to(torch.float64)).sum()),      # drive is linear in J = exp(Js)
        THR=float(-gq64.sum()),
        THR_del=float(-(gq64 * pw[:, None, :]).sum()) * fp.TH_d,
    )
    return g
