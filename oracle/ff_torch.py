"""ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

fp64 torch restatement of the feed-forward tuning-curve generator of the reference's
``FF_functions/lalazar_func.py:16-45`` (``get_FF_output``) as it is called from
``FF_lalazar_model.py:175`` (parameters passed through exp: RF_l = exp(RF_low), ...).

Parity status: PARITY UNPINNED against the reference (a Theano graph; the reference holds no test or
fixture for it).  Pinned against itself only: GPU kernels are compared with this restatement, and its
gradients with finite differences (tests/test_ff_gpu.py, tests/test_oracle_ff.py)."""
import numpy as np
import torch

DT = torch.float64


def grid_positions(box_width):
    """FF_lalazar_model.py:133-137: box^3 points of linspace(-3, 3, box) in (x, y, z) order, z fastest."""
    a = torch.linspace(-3, 3, box_width, dtype=DT)
    X, Y, Z = torch.meshgrid(a, a, a, indexing='ij')
    return torch.stack([X, Y, Z], dim=-1).reshape(-1, 3)


def default_stimuli():
    """FF_lalazar_model.py:139: the 27 points of {-1, 0, 1}^3."""
    return torch.tensor([[x, y, z] for x in (-1, 0, 1) for y in (-1, 0, 1) for z in (-1, 0, 1)], dtype=DT)


def ff_output(RF_l, RF_d, TH, TH_d, J, a, RF_w, FF_con, FF_str, TH_sam, pos, stim):
    """lalazar_func.py:16-45.  RF_w (nsam, G); FF_con, FF_str (nsam, nhid, G); TH_sam (nsam, nhid);
    pos (G, 3); stim (ni, 3) -> hidden activations (nsam, ni, nhid)."""
    dist_sq = ((pos[None, :, :] - stim[:, None, :]) ** 2).sum(dim=2)[None]          # (1, ni, G)
    widths = RF_w[:, None, :]                                                        # (nsam, 1, G)
    expo = dist_sq / (2 * (widths * RF_d + RF_l) ** 2)
    act = torch.exp(-expo)
    act = act / act.sum(dim=2, keepdim=True)                                         # (nsam, ni, G)
    weights = J * FF_con * FF_str                                                    # (nsam, nhid, G)
    drive = torch.einsum('sig,shg->sih', act, weights)
    thr = TH + torch.sign(TH_sam) * TH_sam.abs() ** a * TH_d                         # (nsam, nhid)
    return torch.relu(drive - thr[:, None, :])


def ff_critic_loss(w, xd, xg, xp, lam=1.0, plam=1.0):
    """FF_lalazar_model.py:398-437 with ``SD.make_net(INSHAPE, "WGAN")`` (no hidden layers: one bias-free linear
    unit): mean D(xg) - mean D(xd) + lam * mean((||dD(log(1 + xp))/dxp|| - 1)^2) + plam * sum(w^2);
    the gradient norm through autograd, as the script's T.jacobian does."""
    xp = xp.clone().requires_grad_(True)
    d = torch.log(1 + xp) @ w
    g, = torch.autograd.grad(d.sum(), xp, create_graph=True)
    pen = ((torch.sqrt((g ** 2).sum(dim=1)) - 1) ** 2).mean()
    wdist = (xg @ w).mean() - (xd @ w).mean()
    return wdist + lam * pen + plam * (w ** 2).sum(), wdist
