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


@pytest.mark.parametrize('lattice', [True, False])
@pytest.mark.parametrize('box,nsam,nhid', [(8, 3, 1), (12, 4, 2), (13, 2, 1), (40, 2, 1)])
def test_ff_sparse_forward_vs_dense_and_oracle(box, nsam, nhid, lattice):
    """`ssn_ff_forward_sparse_f32` (connection lists instead of the dense FF_con / FF_str streams): the same hidden activations,
    pre-threshold drive and denominators as the dense entry point and as oracle/ff_torch.py -- lattice fast path (box % 4 == 0,
    the model script's 27 stimuli) and generic path (any box, any stimuli), several hidden units sharing one denominator pass,
    lists of different lengths (padding slots), an index that the script's `np.random.choice` drew twice (counts once)."""
    from tc_gan_amd import ff_model
    rs = np.random.RandomState(box + nhid)
    con, strn, wid, ths = ff_model.generate_samples(rs, nsam, box, nhid)
    if box < 40:            # denser than box^3 / 100 so that small grids produce a drive; unit (0, 0) stays as the script drew it
        dense = (rs.rand(*con.shape) < 0.2).astype(float)
        dense[0, 0] = con[0, 0]
        con = dense
    stim = ff_model.default_stimuli()
    if not lattice:
        stim = (stim + rs.randn(*stim.shape).astype('float32') * 0.1)[:20]
    params = dict(ff_model.START_PARAMS, Js=np.log(3.0 if box < 40 else 1000.0), THR=0.1)
    out, saved = ff_model.ff_forward(params, wid, con, strn, ths, stim, box, keep=True)
    idx, val = ff_model.sparsify(con, strn)
    counts = (con != 0).sum(axis=2)
    assert idx.shape[2] == counts.max() and int((idx >= 0).sum()) == int(counts.sum())
    assert counts.min() < counts.max() or nsam * nhid == 1           # padding slots exist
    got, keep = ff_model.ff_forward_sparse(params, wid, idx, val, ths, stim, box, keep=True)
    np.testing.assert_allclose(keep['den'].cpu().numpy(), saved['den'].cpu().numpy(), rtol=2e-5)
    np.testing.assert_allclose(keep['q'].cpu().numpy(), saved['q'].cpu().numpy(), rtol=5e-5, atol=1e-6)
    np.testing.assert_allclose(got.cpu().numpy(), out.cpu().numpy(), rtol=5e-5, atol=1e-6)
    pt = {k: torch.tensor(float(v), dtype=torch.float64) for k, v in params.items()}
    want = of.ff_output(pos=of.grid_positions(box), stim=torch.as_tensor(stim, dtype=torch.float64),
                        RF_w=torch.as_tensor(wid), FF_con=torch.as_tensor(con), FF_str=torch.as_tensor(strn),
                        TH_sam=torch.as_tensor(ths), **_oracle_params(pt))
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=2e-4, atol=2e-5)
    # the gradient of the five trainable parameters from the lists: the dense second pass, and fp64 autograd
    G = torch.as_tensor(rs.randn(*want.shape))
    g_dense = ff_model.ff_backward(params, saved, out, G.to('cuda'))
    g_sparse = ff_model.ff_backward(params, keep, got, G.to('cuda'))
    pg = {k: torch.tensor(float(v), dtype=torch.float64, requires_grad=True) for k, v in params.items()}
    w2 = of.ff_output(pos=of.grid_positions(box), stim=torch.as_tensor(stim, dtype=torch.float64),
                      RF_w=torch.as_tensor(wid), FF_con=torch.as_tensor(con), FF_str=torch.as_tensor(strn),
                      TH_sam=torch.as_tensor(ths), **_oracle_params(pg))
    grads = torch.autograd.grad((G * w2).sum(), [pg[k] for k in ff_model.PARAM_NAMES])
    for name, w in zip(ff_model.PARAM_NAMES, grads):
        np.testing.assert_allclose(g_sparse[name], g_dense[name], rtol=5e-4, atol=1e-5)
        np.testing.assert_allclose(g_sparse[name], float(w), rtol=2e-3, atol=1e-4)
    # an empty list is a unit without input: drive 0
    none = ff_model.ff_forward_sparse(params, wid, torch.full_like(idx, -1), val, ths, stim, box)
    thr = params['THR'] + np.sign(ths) * np.abs(ths) ** np.exp(params['As']) * np.exp(params['THR_del'])
    np.testing.assert_allclose(none.cpu().numpy(), np.broadcast_to(np.maximum(-thr, 0)[:, None, :], none.shape), rtol=1e-6, atol=1e-7)


def test_ff_sparse_connectivity_shape_of_the_model_script():
    """box^3/100 connections per unit drawn with replacement (FF_lalazar_model.py:154-167)."""
    from tc_gan_amd import ff_model
    rs = np.random.RandomState(1)
    con, strn, wid, ths = ff_model.generate_samples(rs, 3, 10, 1)
    assert con.shape == (3, 1, 1000) and con.sum(axis=2).max() <= 10
    out = ff_model.ff_forward(ff_model.START_PARAMS, wid, con, strn, ths, ff_model.default_stimuli(), 10)
    assert out.shape == (3, 27, 1) and bool(torch.isfinite(out).all())


def test_ff_wgan_steps_vs_autograd(tmp_path):
    """One critic step and one generator step of the FF WGAN (FF_lalazar_model.py:223-330, 398-454): closed-form
    critic gradient vs autograd on the restated loss, Adam(.5, .9) update, generator gradient through the kernels."""
    from oracle import gan_torch as og
    from tc_gan_amd import ff_model, ff_wgan
    box, nsam = 8, 6
    curves = np.random.RandomState(3).rand(40, 27) * 30
    gan = ff_wgan.FFWGAN(box, curves, nsam=nsam, seed=7)
    w0 = gan.w.cpu().numpy().astype('float64')
    # -- critic loss / gradient on fixed minibatches
    rs = np.random.RandomState(2)
    xd, xg = rs.rand(nsam, 27) * 20, rs.rand(nsam, 27) * 20
    ee = rs.rand(nsam, 1)
    xp = ee * xd + (1 - ee) * xg
    tw = of.torch.tensor(w0, dtype=of.DT, requires_grad=True)
    loss_o, wdist_o = of.ff_critic_loss(tw, *(of.torch.tensor(a, dtype=of.DT) for a in (xd, xg, xp)))
    g_o, = of.torch.autograd.grad(loss_o, tw)
    wdist, loss, grad = gan.critic_loss_grad(*(torch.as_tensor(a, device='cuda', dtype=torch.float32) for a in (xd, xg, xp)))
    np.testing.assert_allclose(wdist, float(wdist_o), rtol=1e-5)
    np.testing.assert_allclose(loss, float(loss_o), rtol=1e-5)
    np.testing.assert_allclose(grad.cpu().numpy(), g_o.numpy(), rtol=2e-5, atol=1e-6)
    # -- one critic step = Adam(beta1 .5, beta2 .9) on that kind of gradient, RNG order of the script
    rng = np.random.RandomState(7)
    rng.choice(np.arange(1), 1)
    a = np.sqrt(6.0 / 28)
    np.testing.assert_allclose(w0, rng.uniform(-a, a, 27), rtol=1e-6)
    ss = ff_model.generate_samples(rng, nsam, box, 1)
    ff_model.generate_samples(rng, nsam, box, 1)
    idx = rng.choice(np.arange(len(curves)), nsam)
    eps = rng.rand(nsam, 1)
    p = {k: of.torch.tensor(float(v), dtype=of.DT) for k, v in ff_model.START_PARAMS.items()}
    con, strn, wid, ths = (of.torch.tensor(a, dtype=of.DT) for a in ss)
    out = of.ff_output(p['RF_low'].exp(), p['RF_del'].exp(), p['THR'], p['THR_del'].exp(), p['Js'].exp(), p['As'].exp(),
                        wid, con, strn, ths, of.grid_positions(box), of.default_stimuli())[:, :, 0]
    xd_t = of.torch.tensor(curves[idx], dtype=of.DT)
    eps_t = of.torch.tensor(eps, dtype=of.DT)
    tw = of.torch.tensor(w0, dtype=of.DT, requires_grad=True)
    loss_o, wdist_o = of.ff_critic_loss(tw, xd_t, out, eps_t * xd_t + (1 - eps_t) * out)
    g_o, = of.torch.autograd.grad(loss_o, tw)
    w1 = og.adam_step(w0, g_o.numpy(), {}, 0.01, beta1=0.5, beta2=0.9)
    got = gan.critic_step()
    np.testing.assert_allclose(got, float(wdist_o), rtol=2e-3, atol=1e-4)
    np.testing.assert_allclose(gan.w.cpu().numpy(), w1, rtol=2e-3, atol=2e-5)
    # -- generator step: loss value and the direction of the Adam step (first step moves by lr * sign(grad))
    before = dict(gan.params)
    gloss, _ = gan.generator_step()
    assert np.isfinite(gloss)
    moved = [abs(gan.params[n] - before[n]) for n in ff_model.PARAM_NAMES]
    assert all(m <= 0.01001 for m in moved) and max(moved) > 0.009
    # -- the loop writes the script's files
    gan.train(2, outdir=str(tmp_path), n_critic=1, n_critic_first=2)
    tag = 'wgan_FF_8_8'
    assert open(tmp_path / 'FF_logs' / ('FF_log_' + tag + '.csv')).readline() == 'RF\tRFd\tJ\tth\tth_d\n'
    assert len(open(tmp_path / 'FF_logs' / ('FF_losslog_' + tag + '.csv')).readlines()) == 1 + (2 + 1) + (1 + 1)
    assert len(open(tmp_path / ('tuning_curves' + tag + '.csv')).readlines()) == 2 * nsam
    assert (tmp_path / 'disc_params' / ('D_par_0_' + tag + '.npy')).exists()


def test_full_size_c5_properties():
    """BASELINE config 5 at FULL size (box 40 = 64000 grid points, 27 stimuli, 16384 samples; 12.6 GB of inputs): every
    output finite, a sample's curve independent of its position in the batch and of the batch around it (bitwise), and
    spot rows against the fp64 restatement (oracle/ff_torch.py) at the small-size tolerance."""
    from tc_gan_amd import ff_model
    nsam, box, nhid = 16384, 40, 1
    G = box ** 3
    gen = torch.Generator(device='cuda'); gen.manual_seed(5)
    wid = torch.rand((nsam, G), device='cuda', generator=gen)
    con = (torch.rand((nsam, nhid, G), device='cuda', generator=gen) < 0.01).float()
    strn = torch.rand((nsam, nhid, G), device='cuda', generator=gen)
    ths = torch.rand((nsam, nhid), device='cuda', generator=gen) * 2 - 1
    stim = ff_model.default_stimuli()
    params = dict(ff_model.START_PARAMS, Js=np.log(40.0), THR=0.05)       # enough drive for non-zero curves at 1 % density
    out = ff_model.ff_forward(params, wid, con, strn, ths, stim, box)
    assert out.shape == (nsam, 27, nhid) and bool(torch.isfinite(out).all())
    assert float((out > 0).float().mean()) > 0.05
    # batch independence: a slice of the batch, and one sample on its own, reproduce their rows bit for bit
    rows = [0, 777, 8191, 16383]
    sub = ff_model.ff_forward(params, wid[5000:5003], con[5000:5003], strn[5000:5003], ths[5000:5003], stim, box)
    assert torch.equal(sub, out[5000:5003])
    for r in rows:
        one = ff_model.ff_forward(params, wid[r:r + 1], con[r:r + 1], strn[r:r + 1], ths[r:r + 1], stim, box)
        assert torch.equal(one[0], out[r])
    # spot rows vs the fp64 restatement
    pt = {k: torch.tensor(float(v), dtype=torch.float64) for k, v in params.items()}
    idx = torch.as_tensor(rows, device='cuda')
    want = of.ff_output(pos=of.grid_positions(box), stim=torch.as_tensor(stim, dtype=torch.float64),
                        RF_w=wid[idx].cpu().double(), FF_con=con[idx].cpu().double(), FF_str=strn[idx].cpu().double(),
                        TH_sam=ths[idx].cpu().double(), **_oracle_params(pt))
    np.testing.assert_allclose(out[idx].cpu().numpy(), want.numpy(), rtol=2e-4, atol=2e-5)
