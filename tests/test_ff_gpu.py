"""GPU parity of the feed-forward generator (BASELINE config 5 path) against oracle/ff_torch.py:
outputs and the gradient w.r.t. the five trainable log-parameters."""
import numpy as np
import pytest
import torch

from oracle import ff_torch as of

pytestmark = pytest.mark.gpu


def _oracle_params(p):
    return dict(RF_l=torch.exp(p['RF_low']), RF_d=torch.exp(p['RF_del']), TH=p['THR'], TH_d=torch.exp(p['THR_del']),
                J=torch.exp(p['Js']), a=torch.exp(p['As']))


@pytest.mark.parametrize('lattice', [True, False])
@pytest.mark.parametrize('box,nsam,nhid', [(6, 3, 1), (10, 4, 2), (13, 2, 1)])
def test_ff_forward_and_gradient_vs_oracle(box, nsam, nhid, lattice):
    from tc_gan_amd import ff_model
    rs = np.random.RandomState(box)
    con, strn, wid, ths = ff_model.generate_samples(rs, nsam, box, nhid)
    # denser connectivity than box^3/100 so that small test grids produce non-zero drive
    con = (rs.rand(*con.shape) < 0.2).astype(float)
    stim = ff_model.default_stimuli()
    if not lattice:      # generic kernel: perturbed positions, fewer stimuli
        stim = (stim + rs.randn(*stim.shape).astype('float32') * 0.1)[:20]
    params = dict(ff_model.START_PARAMS, Js=np.log(3.0), THR=0.1)
    out, saved = ff_model.ff_forward(params, wid, con, strn, ths, stim, box, keep=True)
    pt = {k: torch.tensor(float(v), dtype=torch.float64, requires_grad=True) for k, v in params.items()}
    want = of.ff_output(pos=of.grid_positions(box), stim=torch.as_tensor(stim, dtype=torch.float64),
                        RF_w=torch.as_tensor(wid), FF_con=torch.as_tensor(con), FF_str=torch.as_tensor(strn),
                        TH_sam=torch.as_tensor(ths), **_oracle_params(pt))
    assert float((want > 0).double().mean()) > 0.2
    np.testing.assert_allclose(out.cpu().numpy(), want.detach().numpy(), rtol=2e-4, atol=2e-5)
    G = torch.as_tensor(rs.randn(*want.shape))
    grads = torch.autograd.grad((G * want).sum(), [pt[k] for k in ff_model.PARAM_NAMES])
    got = ff_model.ff_backward(params, saved, out, G.to('cuda'))
    for name, w in zip(ff_model.PARAM_NAMES, grads):
        np.testing.assert_allclose(got[name], float(w), rtol=2e-3, atol=1e-4)


def test_ff_sparse_connectivity_shape_of_the_model_script():
    """box^3/100 connections per unit drawn with replacement (FF_lalazar_model.py:154-167)."""
    from tc_gan_amd import ff_model
    rs = np.random.RandomState(1)
    con, strn, wid, ths = ff_model.generate_samples(rs, 3, 10, 1)
    assert con.shape == (3, 1, 1000) and con.sum(axis=2).max() <= 10
    out = ff_model.ff_forward(ff_model.START_PARAMS, wid, con, strn, ths, ff_model.default_stimuli(), 10)
    assert out.shape == (3, 27, 1) and bool(torch.isfinite(out).all())
