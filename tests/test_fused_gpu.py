"""GPU parity of the one-launch BPTT backward (`ssn_gen_backward_fused_f32`: adjoint sweep + dL/dW on chip) against fp64
autograd of oracle/gan_torch.py and against the two-launch path it replaces (reference semantics: the `theano.grad` of
networks/wgan.py:236-242 through the scan of networks/ssn.py:354-385, 566-576)."""
import numpy as np
import pytest
import torch

from oracle import ssn_numpy as on
from test_generator_gpu import GEN, P, _horizon_oracle, _problem

pytestmark = pytest.mark.gpu


def _setup(N, B, NB, T, skip, seed, theta=2.0, gen=GEN, kernel=0):
    from tc_gan_amd import genops, stimuli, weight_gen
    jds, z, bws, con = _problem(N, B, NB, seed, T, skip, theta)
    zt = torch.as_tensor(z).to('cuda', torch.float32)
    W = weight_gen.generate_weight_batch(N, jds['J'], jds['D'], jds['S'], zt, dtype='float32')
    ext = stimuli.stimulus_batch(bws, con, P['smoothness'], N, dtype='float32')
    gp = genops.make_gen_params(seqlen=T, skip_steps=skip, rate_penalty_threshold=theta, kernel=kernel, **gen)
    return jds, zt, W, ext, gp


@pytest.mark.parametrize('shape', ['c3', 'paper'])
def test_fused_backward_vs_oracle_at_production_horizons(shape):
    """The cases of `test_bptt_gradients_vs_oracle_at_production_horizons` (C3: 2N = 200, 8 stimuli, 1200 / 1000 steps; the
    paper's run: 2N = 202, tau_E = 2, 240 / 200, deg-heteroin input) through the fused launch: dL/dW per draw within 1e-4 of
    its largest element, dL/d(J, D, S[, V]) within 2e-3, against fp64 autograd."""
    from tc_gan_amd import genops, stimuli, weight_gen
    o = _horizon_oracle(shape)
    N, B, NB = o['N'], o['B'], o['NB']
    jds = o['jds']
    zt = torch.as_tensor(o['z']).to('cuda', torch.float32)
    W = weight_gen.generate_weight_batch(N, jds['J'], jds['D'], jds['S'], zt, dtype='float32')
    base = stimuli.stimulus_batch(o['bws'], o['con'], P['smoothness'], N, dtype='float32')
    zin = torch.as_tensor(o['zin']).to('cuda', torch.float32)
    ext = base if o['v'] is None else ((1 + o['v'] * zin)[:, None, :] * base).contiguous()
    gp = genops.make_gen_params(seqlen=o['T'], skip_steps=o['skip'], rate_penalty_threshold=o['theta'], kernel=8, **o['gen'])
    out = genops.gen_forward(W, ext, gp, save=True)
    xmax = genops.rate_bound(gp)
    assert genops.gen_backward_fused_supported(B, NB, 2 * N, gp, xmax)
    df0 = out['df'].clone()
    Gd = torch.as_tensor(o['G']).to('cuda', torch.float32)
    gW, g_ext, dmax = genops.gen_backward_fused(W, out['traj'], out['df'], Gd, o['costs'][0] / out['n_dyn'],
                                                o['costs'][1] / out['n_rate'], gp, xmax, want_g_ext=True)
    assert torch.equal(df0, out['df'])                       # f'(u) is only read
    got = gW.cpu().numpy().astype('float64')
    assert np.isfinite(got).all() and np.isfinite(dmax.cpu().numpy()).all()
    err = np.abs(got - o['gW']).reshape(B, -1).max(axis=1) / np.abs(o['gW']).reshape(B, -1).max(axis=1)
    assert err.max() < 1e-4, err
    gJ, gD, gS = genops.jds_grad(gW, zt, jds['J'], jds['D'], jds['S'])
    for g, w in ((gJ, o['gJ']), (gD, o['gD']), (gS, o['gS'])):
        np.testing.assert_allclose(g, w, rtol=2e-3, atol=2e-3 * np.abs(w).max())
    if o['v'] is not None:
        gV = float((g_ext.double() * base.double() * zin.double()[:, None, :]).sum())
        np.testing.assert_allclose(gV, o['gV'], rtol=2e-3)
    print('fused backward, horizon %s: max |dL/dW - fp64| / max |dL/dW| = %.2e' % (shape, err.max()))


@pytest.mark.parametrize('N,B,NB,T,skip', [(100, 5, 8, 120, 80), (101, 3, 8, 60, 40), (104, 2, 5, 50, 30), (75, 3, 8, 64, 50),
                                           (64, 2, 3, 40, 20), (50, 4, 8, 90, 60), (20, 2, 1, 30, 10), (100, 2, 8, 7, 0),
                                           (100, 1, 8, 3, 2), (100, 1, 4, 1, 0), (90, 3, 7, 2, 1)])
def test_fused_backward_matches_the_two_launch_path(N, B, NB, T, skip):
    """Every size class of the fused kernel (2N <= 104 / 152 / 208; partial tiles; 1 ... 8 stimuli; sweeps shorter than the
    three-step load queue) against the fp32 sweep + bf16 x 3 product: dL/dW within 3e-6 sqrt(NB T) of the draw's largest
    element (the tolerance of the fp16 form of the product), dL/d ext within 1e-5, and max |delta| per draw a bound on the
    deltas of the fp32 sweep (to the 1e-6 the two sweeps differ by).  Gradient amplitudes per draw span 1e-6 ... 1e4, one
    draw has zero gradient."""
    from tc_gan_amd import genops
    jds, zt, W, ext, gp = _setup(N, B, NB, T, skip, 5 * N + T)
    gp2 = genops.make_gen_params(seqlen=T, skip_steps=skip, rate_penalty_threshold=2.0, kernel=2 if NB >= 4 and 2 * N <= 208 else 0,
                                 **GEN)
    out = genops.gen_forward(W, ext, gp2, save=True)
    rs = np.random.RandomState(N + B)
    amp = np.resize(np.array([1.0, 1e-6, 1e4, 0.0, 1.0]), B)
    gta = torch.as_tensor(rs.randn(B, NB, 2 * N) * amp[:, None, None], device='cuda', dtype=torch.float32)
    c_dyn, c_rate = 1e-3, 2e-3
    xmax = float(out['traj'].max()) + 1.0
    assert genops.gen_backward_fused_supported(B, NB, 2 * N, gp, xmax)
    gW, g_ext, dmax = genops.gen_backward_fused(W, out['traj'], out['df'], gta, c_dyn, c_rate, gp, xmax, want_g_ext=True)
    gW0, _, _ = genops.gen_backward_fused(W, out['traj'], out['df'], gta, c_dyn, c_rate, gp, xmax)
    assert torch.equal(gW, gW0)                              # with and without dL/d ext: the same bits
    d, ge = genops.gen_backward(W, out['traj'], out['df'].clone(), gta, c_dyn, c_rate, gp2, want_g_ext=True)
    ref = genops.weight_grad(d, out['traj'], kernel=2 if 2 * N <= 224 and 2 * N > 32 else 1).cpu().numpy()
    got = gW.cpu().numpy()
    for i in range(B):
        scale = max(np.abs(ref[i]).max(), 1e-300)
        np.testing.assert_allclose(got[i], ref[i], rtol=0, atol=3e-6 * scale * np.sqrt(NB * T), err_msg='draw %d' % i)
    gs = np.abs(ge.cpu().numpy()).reshape(B, -1).max(axis=1)
    np.testing.assert_allclose(g_ext.cpu().numpy(), ge.cpu().numpy(), rtol=0, atol=1e-5 * max(gs.max(), 1e-300))
    for i in range(B):
        np.testing.assert_allclose(g_ext[i].cpu().numpy(), ge[i].cpu().numpy(), rtol=0, atol=1e-5 * gs[i] + 1e-37)
    true = d.abs().reshape(B, -1).max(dim=1).values.cpu().numpy()
    dm = dmax.cpu().numpy()
    # (the two-launch sweep does not store delta_1, the fused one counts it: no upper bound on the shortest sweeps)
    assert (dm >= true * (1 - 1e-4)).all() and (T < 10 or (dm <= np.maximum(true * 4, 1e-30)).all()), (dm, true)


def test_fused_backward_poisons_a_draw_that_outgrows_its_lagged_scale():
    """As the two-draw sweep (`test_split_adjoint_marks_a_draw_that_outgrows_its_lagged_scale`): f' of one draw x 1e5 at one
    step -> that draw's dL/dW and `dmax` are NaN, the other draw is untouched."""
    from tc_gan_amd import genops
    N, B, NB, T, skip = 100, 2, 8, 60, 40
    jds, zt, W, ext, gp = _setup(N, B, NB, T, skip, 3, kernel=8)
    out = genops.gen_forward(W, ext, gp, save=True)
    G = torch.ones((B, NB, 2 * N), device='cuda', dtype=torch.float32)
    xmax = genops.rate_bound(gp)
    clean, _, dm0 = genops.gen_backward_fused(W, out['traj'], out['df'], G, 1e-3, 1e-3, gp, xmax)
    out['df'][1, :, 30, :] *= 1e5
    gW, _, dmax = genops.gen_backward_fused(W, out['traj'], out['df'], G, 1e-3, 1e-3, gp, xmax)
    assert torch.isfinite(clean).all() and torch.isfinite(dm0).all()
    assert torch.equal(gW[0], clean[0]) and float(dmax[0]) == float(dm0[0])
    assert bool(torch.isnan(dmax[1])) and bool(torch.isnan(gW[1]).any())


def test_fused_backward_rate_above_the_bound_is_loud():
    """xmax below the largest rate overflows the fp16 parts of x: inf / NaN in dL/dW, never a silently wrong value."""
    from tc_gan_amd import genops
    N, B, NB, T, skip = 100, 1, 8, 40, 20
    jds, zt, W, ext, gp = _setup(N, B, NB, T, skip, 9, kernel=8)
    out = genops.gen_forward(W, ext, gp, save=True)
    G = torch.ones((B, NB, 2 * N), device='cuda', dtype=torch.float32)
    top = float(out['traj'].max())
    assert top > 1.0
    gW, _, _ = genops.gen_backward_fused(W, out['traj'], out['df'], G, 0.0, 0.0, gp, top / 64.0)
    assert not bool(torch.isfinite(gW).all())


def test_fused_backward_refusals():
    from tc_gan_amd import clib, genops
    gp = genops.make_gen_params(seqlen=20, skip_steps=10, **GEN)
    assert not genops.gen_backward_fused_supported(2, 9, 200, gp, 100.0)          # more than 8 stimuli
    assert not genops.gen_backward_fused_supported(2, 8, 210, gp, 100.0)          # 2N > 208
    assert not genops.gen_backward_fused_supported(2, 8, 200, gp, float('inf'))
    assert not genops.gen_backward_fused_supported(2, 8, 200, gp, None)
    assert genops.gen_backward_fused_supported(2, 8, 200, gp, 100.0)
    t = torch.zeros((2, 9, 20, 200), device='cuda')
    with pytest.raises(clib.SSNLibraryError):
        genops.gen_backward_fused(torch.zeros((2, 200, 200), device='cuda'), t, t, torch.zeros((2, 9, 200), device='cuda'),
                                  0.0, 0.0, gp, 100.0)


@pytest.mark.parametrize('ssn_type,V0', [('default', 0), ('deg-heteroin', 0.3)])
def test_gan_loop_with_the_fused_backward_follows_the_two_launch_loop(ssn_type, V0):
    """`--gen-kernel duo-fused` in the training loop (networks/cwgan.py): same seeds, same forward kernel, the generator
    parameters after a few iterations agree with the 'duo' run to the 1e-5 the two backward forms differ by, the info
    fields of the first iteration are identical (nothing before the first generator update depends on the backward), and
    the choice is recorded in the checkpoint state."""
    from test_cwgan_gpu import TEST_PARAMS, _fake_data
    from tc_gan_amd.networks.cwgan import make_gan

    def run(kernel, iters=4):
        cfg = dict(TEST_PARAMS, num_sites=50, num_models=6, bandwidths=[0.0625, 0.125, 0.25, 0.5, 0.75, 1.0, 0.3, 0.4],
                   contrasts=[20.], z_device_seed=3, gen_kernel=kernel, ssn_type=ssn_type)
        if ssn_type != 'default':
            cfg['V0'] = V0
        gan, _ = make_gan(cfg)
        gan.set_dataset(_fake_data(gan, 9, np.random.RandomState(4)))
        it = gan.learning()
        infos = []
        while len(infos) < iters:
            info = next(it)
            if info.is_discriminator:
                continue
            infos.append(info.gen_loss)
        return gan, infos

    a, ia = run('duo')
    b, ib = run('duo-fused')
    assert b.gen.gen_kernel == 'duo-fused' and b.gen.fused_backward and not a.gen.fused_backward
    assert ia[0] == ib[0]
    for name in ('J', 'D', 'S'):
        np.testing.assert_allclose(getattr(b.gen, name), getattr(a.gen, name), rtol=2e-4, atol=1e-6)
        assert not np.array_equal(getattr(a.gen, name), np.asarray(TEST_PARAMS[name + '0']))
    if ssn_type != 'default':
        np.testing.assert_allclose(b.gen.V, a.gen.V, rtol=2e-4, atol=1e-6)
    assert b.gen.poisoned_draws() == 0
