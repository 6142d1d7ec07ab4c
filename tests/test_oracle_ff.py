"""CPU: the FF oracle's gradients agree with finite differences (it has no reference fixture to pin)."""
import numpy as np
import torch

from oracle import ff_torch as of


def test_ff_oracle_finite_differences():
    rs = np.random.RandomState(0)
    box, nsam, nhid = 5, 2, 1
    G = box ** 3
    inputs = dict(RF_w=torch.as_tensor(rs.rand(nsam, G)), FF_con=torch.as_tensor((rs.rand(nsam, nhid, G) < .3) * 1.0),
                  FF_str=torch.as_tensor(rs.rand(nsam, nhid, G)), TH_sam=torch.as_tensor(rs.uniform(-1, 1, (nsam, nhid))),
                  pos=of.grid_positions(box), stim=of.default_stimuli())
    base = dict(RF_l=2.0, RF_d=0.3, TH=0.05, TH_d=1.0, J=3.0, a=1.0)
    Gw = torch.as_tensor(rs.randn(nsam, 27, nhid))

    def loss(**kw):
        return float((Gw * of.ff_output(**{k: torch.tensor(v, dtype=torch.float64) for k, v in kw.items()}, **inputs)).sum())

    pt = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in base.items()}
    grads = torch.autograd.grad((Gw * of.ff_output(**pt, **inputs)).sum(), list(pt.values()), allow_unused=True)
    for (name, g) in zip(pt, grads):
        if name == 'a':
            continue
        h = 1e-6
        up, dn = dict(base), dict(base)
        up[name] += h; dn[name] -= h
        np.testing.assert_allclose(float(g), (loss(**up) - loss(**dn)) / (2 * h), rtol=1e-5, atol=1e-8)
