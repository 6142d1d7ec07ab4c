"""GPU tests of the conditional BPTT-WGAN assembly (tc_gan_amd/networks/cwgan.py): schedule and info
fields (as networks/tests/test_cwgan.py:38-53), and one full critic + generator update against the fp64
oracle with identical host RNG streams."""
import numpy as np
import pytest
import torch

from oracle import gan_torch as og
from oracle import sampler_numpy as osn
from oracle import ssn_numpy as on

pytestmark = pytest.mark.gpu

JDS = on.new_JDS()
TEST_PARAMS = dict(            # small version of networks/tests/test_wgan.py TEST_PARAMS
    num_sites=10, seqlen=40, skip_steps=30, num_models=4, probes_per_model=2, norm_probes=[0, 0.5],
    include_inhibitory_neurons=True, bandwidths=[0.0625, 0.125, 0.25, 0.75], contrasts=[5., 20.],
    J0=JDS['J'], D0=JDS['D'], S0=JDS['S'], critic_iters_init=3, critic_iters=2, lipschitz_cost=10.0,
    gen=dict(learning_rate=0.01, update_name='sgd', dynamics_cost=1.0, rate_cost=0.01,
             rate_penalty_threshold=5.0, J_min=1e-3, J_max=10, D_min=1e-3, D_max=10, S_min=1e-3, S_max=10),
    disc=dict(learning_rate=0.01, update_name='sgd', layers=[16, 16], normalization='none',
              nonlinearity='rectify', precision='fp32'),
)


def _oracle_minibatch(rng, data, gan, num_models, probes_per_model):
    """One minibatch as the reference draws it, from oracle/sampler_numpy.py (loop-form restatement of
    cwgan.py:322-391, checked against the product's sampler stream for stream in tests/test_oracle_sampler.py)."""
    grid = osn.gridify(data, num_contrasts=len(gan.contrasts), num_bandwidths=len(gan.bandwidths), num_cell_types=2,
                       num_probes=len(gan.norm_probes))
    return osn.select_minibatch(rng, grid, [0, 1], gan.norm_probes, gan.contrasts, gan.bandwidths, gan.e_ratio,
                                num_models, probes_per_model)


def _fake_data(gan, truth_size, rs):
    ncols = len(gan.bandwidths) * len(gan.contrasts) * len(gan.norm_probes) * 2
    return rs.rand(truth_size, ncols) * 10


def test_schedule_and_info_fields():
    from tc_gan_amd.networks.cwgan import make_gan
    gan, rest = make_gan(dict(TEST_PARAMS, truth_size=7))
    assert rest == {'truth_size': 7}
    gan.set_dataset(_fake_data(gan, 7, np.random.RandomState(3)))
    it = gan.learning()
    kinds = []
    for _ in range(3 + 1 + 2 + 1):
        info = next(it)
        kinds.append(info.is_discriminator)
        if info.is_discriminator:
            for field in ('disc_loss', 'accuracy', 'gen_time', 'disc_time', 'xd', 'xg', 'cd', 'cg',
                          'dynamics_penalty', 'rate_penalty', 'gen_step', 'disc_step'):
                assert hasattr(info, field), field
            assert info.xg.shape == (8, 4) and info.cd.shape == (8, 3)
            assert np.isfinite(info.disc_loss)
        else:
            for field in ('gen_loss', 'gen_forward_time', 'gen_train_time', 'gen_time', 'disc_time', 'gen_step'):
                assert hasattr(info, field), field
            assert np.isfinite(info.gen_loss)
    assert kinds == [True] * 3 + [False] + [True] * 2 + [False]
    names = gan.gen.get_flat_param_names()
    assert names[:4] == ('J_EE', 'J_EI', 'J_IE', 'J_II') and len(names) == 12
    assert len(gan.get_gen_param()) == 3
    assert all(np.all(p > 0) for p in gan.get_gen_param())


@pytest.mark.parametrize('norm', ['none', 'layer', ['none', 'layer']])
def test_one_critic_and_generator_update_vs_oracle(norm):
    """Same seeds -> same minibatch, eps, zs (host RandomState order of cwgan.py:471-523); compare the
    critic loss, the post-update critic parameters, the generator loss and the post-update (J, D, S)."""
    from tc_gan_amd.networks.cwgan import make_gan
    cfg = dict(TEST_PARAMS, critic_iters_init=1, critic_iters=1)
    cfg['disc'] = dict(cfg['disc'], normalization=norm)
    ckw = dict(normalization=norm)
    gan, _ = make_gan(cfg)
    data = _fake_data(gan, 9, np.random.RandomState(4))
    gan.set_dataset(data)
    p0 = [og.t64(p) for p in gan.disc.get_param_values()]
    it = gan.learning()
    dinfo = next(it)
    ginfo = next(it)

    # ---- oracle replay with an identical RandomState --------------------------------------------------
    rng = np.random.RandomState(0)
    # the minibatch comes from the loop-form restatement of cwgan.py:322-391 (oracle/sampler_numpy.py), NOT from
    # the product's sampler: the replay shares the seed with the run and nothing else
    batch = _oracle_minibatch(rng, data, gan, 4, 2)
    eps = og.t64(rng.rand(len(batch['tuning_curves']), 1))
    N = 10
    kw = osn.gen_kwargs(batch)
    zs = og.t64(rng.rand(4, 2 * N, 2 * N))
    gen_common = dict(num_sites=N, smoothness=on.DEFAULT_PARAMS['smoothness'], io_type='asym_tanh', k=0.01, n=2.2,
                      tau_E=10., tau_I=1., dt=0.1, seqlen=40, skip_steps=30, rate_penalty_threshold=5.0,
                      dynamics_cost=1.0, rate_cost=0.01)
    J, D, S = (og.t64(JDS[k]) for k in 'JDS')
    _, aux = og.generator_loss(J, D, S, zs, kw['stimulator_bandwidths'], kw['stimulator_contrasts'],
                               kw['prober_model_ids'], kw['prober_norm_probes'], kw['prober_cell_types'], p0,
                               critic_kw=ckw, **gen_common)
    xg = aux['tuning_curve'].detach()
    xd = og.t64(batch['tuning_curves'])
    cd = og.t64(batch['conditions'])
    np.testing.assert_allclose(dinfo.xg.cpu().numpy(), xg.numpy(), rtol=2e-4, atol=1e-5)
    np.testing.assert_allclose(dinfo.xd.cpu().numpy(), xd.numpy(), rtol=1e-6)
    xp = eps * xd + (1 - eps) * xg
    ps = [p.clone().requires_grad_(True) for p in p0]
    dloss = og.critic_loss(ps, xg, xd, xp, cd, cd, cd, 10.0, **ckw)
    dgr = torch.autograd.grad(dloss, ps)
    np.testing.assert_allclose(dinfo.disc_loss, float(dloss), rtol=1e-3, atol=1e-4)
    p1 = [p - 0.01 * g for p, g in zip(p0, dgr)]                 # sgd
    got1 = gan.disc.get_param_values()
    # (the generator step below does not change the critic)
    for a, b in zip(got1, p1):
        np.testing.assert_allclose(a, b.numpy(), rtol=2e-3, atol=2e-5)
    acc = og.critic_forward(p1, xg, cd, **ckw).mean() - og.critic_forward(p1, xd, cd, **ckw).mean()
    np.testing.assert_allclose(dinfo.accuracy, float(acc), rtol=1e-3, atol=1e-4)

    # generator step: new zs, same batch conditions, updated critic
    zs2 = og.t64(rng.rand(4, 2 * N, 2 * N))
    Jg, Dg, Sg = (og.t64(JDS[k]).clone().requires_grad_(True) for k in 'JDS')
    gloss, _ = og.generator_loss(Jg, Dg, Sg, zs2, kw['stimulator_bandwidths'], kw['stimulator_contrasts'],
                                 kw['prober_model_ids'], kw['prober_norm_probes'], kw['prober_cell_types'], p1,
                                 critic_kw=ckw, **gen_common)
    gJ, gD, gS = torch.autograd.grad(gloss, [Jg, Dg, Sg])
    np.testing.assert_allclose(ginfo.gen_loss, float(gloss), rtol=1e-3, atol=1e-4)
    for name, g in (('J', gJ), ('D', gD), ('S', gS)):
        want = np.clip(JDS[name] - 0.01 * g.numpy(), 1e-3, 10)
        got = getattr(gan.gen, name)
        np.testing.assert_allclose(got - JDS[name], want - JDS[name], rtol=5e-3, atol=5e-3 * np.abs(want - JDS[name]).max())


def test_device_noise_mode_runs_and_rate_bound_skips_critic():
    from tc_gan_amd.networks.cwgan import make_gan
    cfg = dict(TEST_PARAMS, z_device_seed=123)
    cfg['disc'] = dict(cfg['disc'], rate_penalty_bound=1e-9)     # any positive rate penalty skips the update
    gan, _ = make_gan(cfg)
    gan.set_dataset(_fake_data(gan, 5, np.random.RandomState(1)))
    before = gan.disc.get_flat().copy()
    info = next(gan.learning())
    assert info.is_discriminator and np.isnan(info.disc_loss) and np.isnan(info.accuracy)
    np.testing.assert_array_equal(gan.disc.get_flat(), before)
    assert gan.disc_updater.step == 0              # the speculative update was rolled back, optimizer state included
    st = gan.disc_updater._state
    assert st is None or float(st[0].abs().max()) == 0.0


def test_rate_bound_skips_exactly_the_steps_over_the_bound():
    """cwgan.py:493-498 decided on the device (`ssn_critic_step_gated_run`): with a bound between the smallest and the largest
    rate penalty of a run, exactly the steps over the bound are reported skipped (NaN loss and accuracy), they leave the
    critic and its Adam state as they found them -- the step count too, so the bias correction of the next update is the one
    of an uninterrupted sequence -- and the other steps are the steps of a run that was never offered those batches."""
    from tc_gan_amd.networks.cwgan import make_gan

    def run(bound, steps=8):
        cfg = dict(TEST_PARAMS, critic_iters_init=steps, z_device_seed=5)
        cfg['gen'] = dict(cfg['gen'], rate_penalty_threshold=0.5)
        cfg['disc'] = dict(cfg['disc'], update_name='adam-wgan', rate_penalty_bound=bound)
        gan, _ = make_gan(cfg)
        gan.set_dataset(_fake_data(gan, 9, np.random.RandomState(2)))
        it = gan.learning()
        rows = []
        for _ in range(steps):
            before = gan.disc.get_flat().copy()
            info = next(it)
            rows.append((info.rate_penalty, info.disc_loss, info.accuracy, before, gan.disc.get_flat().copy(), gan.disc_updater.step))
        return rows

    free = run(-1.0)
    pens = np.array([r[0] for r in free])
    assert pens.min() < pens.max()
    bound = float(np.sort(pens)[len(pens) // 2 - 1] + np.sort(pens)[len(pens) // 2]) / 2
    gated = run(bound)
    np.testing.assert_array_equal([r[0] for r in gated], pens)            # the forwards do not depend on the critic
    applied = 0
    for (pen, loss, acc, before, after, count) in gated:
        if np.float32(pen) > np.float32(bound):
            assert np.isnan(loss) and np.isnan(acc)
            np.testing.assert_array_equal(after, before)
        else:
            applied += 1
            assert np.isfinite(loss) and np.isfinite(acc) and not np.array_equal(after, before)
        assert count == applied
    assert 0 < applied < len(gated)
    # until the first skipped step the two runs are the same run
    first = next(i for i, r in enumerate(gated) if np.isnan(r[1]))
    for a, b in zip(free[:first], gated[:first]):
        assert a[1] == b[1] and a[2] == b[2]
        np.testing.assert_array_equal(a[4], b[4])


@pytest.mark.parametrize('ssn_type,V0', [('heteroin', [0.3, 0.1]), ('deg-heteroin', 0.4)])
def test_heteroin_generator_update_vs_oracle(ssn_type, V0):
    """Heterogeneous-input SSNs (networks/ssn.py:645-772): parameter order [V, J, D, S], noise order
    zs then zs_in, and the V gradient through dL/d ext, against the oracle with the same host RNG stream."""
    from tc_gan_amd.networks.cwgan import make_gan
    cfg = dict(TEST_PARAMS, critic_iters_init=1, critic_iters=1, ssn_type=ssn_type, V0=V0)
    cfg['gen'] = dict(cfg['gen'], V_min=0, V_max=1)
    gan, _ = make_gan(cfg)
    names = gan.gen.get_flat_param_names()
    assert names[:2] == ('V_E', 'V_I') if ssn_type == 'heteroin' else names[0] == 'V'
    assert names[-4:] == ('S_EE', 'S_EI', 'S_IE', 'S_II')
    data = _fake_data(gan, 9, np.random.RandomState(4))
    gan.set_dataset(data)
    p0 = [og.t64(p) for p in gan.disc.get_param_values()]
    it = gan.learning()
    dinfo = next(it)
    p1 = [og.t64(p) for p in gan.disc.get_param_values()]
    ginfo = next(it)
    # oracle replay
    rng = np.random.RandomState(0)
    batch = _oracle_minibatch(rng, data, gan, 4, 2)    # oracle/sampler_numpy.py, not the product's sampler
    rng.rand(len(batch['tuning_curves']), 1)           # eps
    N = 10
    rng.rand(4, 2 * N, 2 * N); rng.choice(2, (4, 2 * N))      # critic step noise: zs, zs_in
    zs = og.t64(rng.rand(4, 2 * N, 2 * N))
    zs_in = rng.choice(2, (4, 2 * N)) * 2 - 1
    kw = osn.gen_kwargs(batch)
    Jg, Dg, Sg = (og.t64(JDS[k]).clone().requires_grad_(True) for k in 'JDS')
    Vg = og.t64(V0).clone().requires_grad_(True)
    gloss, _ = og.generator_loss(Jg, Dg, Sg, zs, kw['stimulator_bandwidths'], kw['stimulator_contrasts'],
                                 kw['prober_model_ids'], kw['prober_norm_probes'], kw['prober_cell_types'], p1,
                                 num_sites=N, smoothness=on.DEFAULT_PARAMS['smoothness'], io_type='asym_tanh', k=0.01,
                                 n=2.2, tau_E=10., tau_I=1., dt=0.1, seqlen=40, skip_steps=30,
                                 rate_penalty_threshold=5.0, dynamics_cost=1.0, rate_cost=0.01, V=Vg, zs_in=zs_in)
    gJ, gD, gS, gV = torch.autograd.grad(gloss, [Jg, Dg, Sg, Vg])
    np.testing.assert_allclose(ginfo.gen_loss, float(gloss), rtol=1e-3, atol=1e-4)
    want_V = np.clip(np.asarray(V0, dtype=float) - 0.01 * gV.numpy(), 0, 1)
    np.testing.assert_allclose(np.asarray(gan.gen.V) - np.asarray(V0), want_V - np.asarray(V0), rtol=1e-2,
                               atol=1e-2 * np.abs(want_V - np.asarray(V0)).max() + 1e-9)
    want_J = np.clip(JDS['J'] - 0.01 * gJ.numpy(), 1e-3, 10)
    np.testing.assert_allclose(gan.gen.J - JDS['J'], want_J - JDS['J'], rtol=1e-2, atol=1e-2 * np.abs(want_J - JDS['J']).max())


def test_checkpoint_resume_continues_the_same_run(tmp_path):
    """Four generator steps in one go == two steps, checkpoint, a NEW GAN object restored from it, two more
    (parameters, optimizer states and the shared host RandomState all travel)."""
    from tc_gan_amd.networks.cwgan import make_gan
    cfg = dict(TEST_PARAMS, critic_iters_init=3, critic_iters=2)
    cfg['gen'] = dict(cfg['gen'], update_name='adam-wgan')
    cfg['disc'] = dict(cfg['disc'], update_name='rmsprop', normalization=['none', 'layer'])

    def run(gan, start, n_gen_steps):
        it = gan.learning(start)
        done, losses = 0, []
        while done < n_gen_steps:
            info = next(it)
            if not info.is_discriminator:
                done += 1
                losses.append(info.gen_loss)
        return losses

    def fresh():
        gan, _ = make_gan(cfg)
        gan.set_dataset(_fake_data(gan, 9, np.random.RandomState(4)))
        return gan

    full = fresh()
    losses_full = run(full, 0, 4)
    first = fresh()
    run(first, 0, 2)
    path = str(tmp_path / 'checkpoint.pkl')
    first.save_checkpoint(path, gen_step=1)
    second = fresh()
    start = second.load_checkpoint(path)
    assert start == 2
    losses_tail = run(second, start, 2)
    np.testing.assert_allclose(losses_tail, losses_full[2:], rtol=1e-4, atol=1e-6)
    for a, b in zip(second.get_gen_param(), full.get_gen_param()):
        np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(second.disc.get_flat(), full.disc.get_flat(), rtol=1e-4, atol=1e-6)
    assert second.disc_updater.step == full.disc_updater.step == 3 + 3 * 2
    # format guards: the file says which format it is and what it was written with; a device-noise state of the round-1
    # format (a torch.Generator state array) and an unknown version are refused with a KnownError, not an IndexError
    import pickle
    from tc_gan_amd.execution import KnownError
    ck = pickle.load(open(path, 'rb'))
    assert ck['version'] == 2 and ck['state']['gen_kernel'] == 'auto' and ck['state']['world'] == 1
    noisy, _ = make_gan(dict(cfg, z_device_seed=3))
    noisy.set_dataset(_fake_data(noisy, 9, np.random.RandomState(4)))
    old = dict(ck, state=dict(ck['state'], zgen=np.zeros(16, dtype='uint8')))
    pickle.dump(old, open(str(tmp_path / 'old.pkl'), 'wb'))
    with pytest.raises(KnownError, match='earlier format'):
        noisy.load_checkpoint(str(tmp_path / 'old.pkl'))
    pickle.dump(dict(ck, version=99), open(str(tmp_path / 'v99.pkl'), 'wb'))
    with pytest.raises(KnownError, match='format version'):
        noisy.load_checkpoint(str(tmp_path / 'v99.pkl'))


@pytest.mark.parametrize('truth_size,probes_per_model,norm_probes,inhibitory', [
    (1, 1, [0], False), (1, 1, [0], True), (8, 2, [0, 0.5], False), (8, 2, [0], True), (8, 2, [0, 0.5], True)])
def test_conditional_tuning_curves_vs_fixed_point_dataset(truth_size, probes_per_model, norm_probes, inhibitory):
    """networks/tests/test_conditional_tuning_curve.py:141-204: the dataset comes from the fixed-point solver
    (`ssnode.sample_tuning_curves`, fp64, atol 1e-10); feeding the z of the sampled draws to the fp32 fixed-time
    generator (seqlen 4000, skip = seqlen - 1) through the `RandomChoiceSampler` minibatch must reproduce the
    dataset's tuning curves and the solver's fixed points (reference tolerance 5e-4)."""
    from tc_gan_amd import ssnode
    from tc_gan_amd.networks.cwgan import RandomChoiceSampler, make_gan
    from tc_gan_amd.gradient_expressions.utils import sample_sites_from_stim_space
    N, seqlen = 20, 4000
    bandwidths, contrasts = [0.0625, 0.25, 0.75], [5.]     # one contrast, as in the reference test (20 saturates an I
    # cell at N = 20, whose fixed point the fixed-time run approaches too slowly)
    cfg = dict(TEST_PARAMS, num_sites=N, seqlen=seqlen, skip_steps=seqlen - 1, num_models=truth_size,
               probes_per_model=probes_per_model, norm_probes=norm_probes, include_inhibitory_neurons=inhibitory,
               bandwidths=bandwidths, contrasts=contrasts, include_time_avg=True)
    gan, _ = make_gan(cfg)
    data, (zs, rates, fpinfo) = ssnode.sample_tuning_curves(
        sample_sites=sample_sites_from_stim_space(norm_probes, N), NZ=truth_size, seed=1, bandwidths=bandwidths,
        contrast=contrasts, N=N, track_offset_identity=True, include_inhibitory_neurons=inhibitory,
        J=JDS['J'], D=JDS['D'], S=JDS['S'], io_type='asym_tanh', atol=1e-10, dt=5e-4, max_iter=400000)
    data = np.array(data.T)
    zs, rates = np.asarray(zs), np.asarray(rates)
    assert fpinfo.rejections == 0 and zs.shape == (truth_size, 2 * N, 2 * N)
    kw = dict(bandwidths=bandwidths, contrasts=contrasts, norm_probes=norm_probes, e_ratio=gan.e_ratio,
              include_inhibitory_neurons=inhibitory)
    sampler = RandomChoiceSampler.from_grid_data(data, seed=np.random.RandomState(3), **kw)
    # the same draws on an "index dataset" tell which truth sample every minibatch row came from
    index_data = np.broadcast_to(np.arange(truth_size, dtype='float64')[:, None], data.shape)
    id_sampler = RandomChoiceSampler.from_grid_data(index_data, seed=np.random.RandomState(3), **kw)
    batch = sampler.select_minibatch(truth_size, probes_per_model)
    ids = id_sampler.select_minibatch(truth_size, probes_per_model).tc_md[:, :, 0].astype(int)
    used = sorted(set(ids.flat))
    padded = np.zeros(truth_size, dtype=int)          # num_models >= number of distinct samples ([1] in the reference test)
    padded[:len(used)] = used
    gen_kwargs = batch.gen_kwargs
    gen_kwargs['prober_model_ids'] = np.array([used.index(i) for i in ids.flatten()], dtype='uint16')
    # every model has to see the contrast of ITS rows: rows of a model share one contrast, but the squeezed models do not
    row_contrast = batch.conditions[:, 0]
    con = np.full(truth_size, contrasts[0])
    con[gen_kwargs['prober_model_ids']] = row_contrast
    gen_kwargs['stimulator_contrasts'] = np.broadcast_to(con[:, None], (truth_size, len(bandwidths))).astype('float32')
    out = gan.gen.forward(model_zs=zs[padded], model_rate_penalty_threshold=200., **gen_kwargs)
    xg = out.prober_tuning_curve.cpu().numpy()
    np.testing.assert_allclose(xg, batch.tuning_curves, rtol=5e-4, atol=5e-4)
    # time_avg (models, stimuli = this model's contrast x bandwidths, neurons) vs the solver's fixed points
    ta = out.model_time_avg.cpu().numpy()
    fp = rates.reshape(truth_size, len(contrasts), len(bandwidths), 2 * N)
    for m in set(gen_kwargs['prober_model_ids']):
        ci = contrasts.index(float(con[m]))
        np.testing.assert_allclose(ta[m], fp[padded[m], ci], rtol=5e-4, atol=5e-4)


@pytest.mark.parametrize('tail,retry,ssn_type', [('fused', 'subset', 'default'), ('fused', 'all', 'default'),
                                                 ('fused', 'subset', 'heteroin'), ('per-parameter', 'subset', 'default')])
def test_poisoned_generator_gradient_is_not_hidden_by_the_parameter_bounds(tail, retry, ssn_type, caplog, monkeypatch):
    """a draw whose fp16 adjoint outgrows its scale makes the summed gradient NaN; the optimizer's clip used to turn
    the NaN update into the lower bound (fmaxf(NaN, lo) = lo), so the parameters stayed finite, the loss stayed finite and
    nothing reported.  Now: the clip keeps NaN as Theano's does (wgan.py:244-251: a switch on comparisons) -- the
    parameter-by-parameter tail returns non-finite parameters, the drivers' NaN guards see them, the loop says how many draws
    were poisoned -- and the default tail withholds the update (device-side gate: parameters and Adam state untouched), makes
    the step again on the fp32 kernels (forward, adjoint, dL/dW: what `gen_kernel mfma-fp32` runs every step, and what the
    reference's fp32 arithmetic does with such a draw) and applies THAT gradient -- for the refused draws only ('subset': the
    other draws keep the gradient pieces they have) or for all of them (TCGAN_SUBSET_RETRY=0, or more than half refused)."""
    import logging
    from tc_gan_amd.networks import cwgan
    monkeypatch.setattr(cwgan, '_SUBSET_RETRY', retry == 'subset')
    from argparse import Namespace
    from tc_gan_amd.networks.cwgan import make_gan
    from tc_gan_amd.utils import StopWatch

    def step(kernel, poison):
        cfg = dict(TEST_PARAMS, num_sites=60, num_models=4, probes_per_model=1, seqlen=60, skip_steps=40,
                   bandwidths=[0.0625, 0.125, 0.25, 0.5, 0.75, 1.0, 0.3, 0.4], contrasts=[20.], gen_kernel=kernel, critic_iters=0,
                   ssn_type=ssn_type, **(dict(V=[0.3, 0.1]) if ssn_type != 'default' else {}))
        cfg['gen'] = dict(cfg['gen'], update_name='adam-wgan')
        gan, _ = make_gan(cfg)
        if tail == 'per-parameter':
            gan.gen_updaters['J'].learning_rate *= 2          # updaters out of step with each other: the parameter-by-parameter tail
        gan.set_dataset(_fake_data(gan, 9, np.random.RandomState(4)))
        gan.gen_forward_watch, gan.gen_train_watch, gan.disc_train_watch = StopWatch(), StopWatch(), StopWatch()
        batch = gan.next_minibatch()
        prepared = gan._prepare_gen(batch)
        assert (gan._gen_tail_fused() is not None) == (tail == 'fused')
        if poison:
            gan.gen._saved['fwd']['df'][1, :, 50, :] *= 1e5        # one draw's f' at one step: its adjoint grows by 1e5 within a step
        info = gan.train_generator(Namespace(gen_step=0), batch, prepared)
        return gan, info, np.concatenate([np.ravel(getattr(gan.gen, name)) for name in gan._pnames])

    with caplog.at_level(logging.WARNING):
        gan, info, values = step('duo', poison=True)
    assert gan.gen.poisoned_draws() == 1
    if tail == 'per-parameter':
        assert not np.isfinite(values).all()                  # ... and not the lower bound 1e-3
        assert any('adjoint' in rec.getMessage() and '1 of 4 draws' in rec.getMessage() for rec in caplog.records)
        return
    assert any('recomputed on the fp32 kernels' in rec.getMessage() and '1 draws' in rec.getMessage() for rec in caplog.records)
    # the step that was applied is the fp32 kernels' step on the unpoisoned problem (the second pass recomputes f' from the
    # forward), Adam state and step count those of ONE update
    _, info32, want = step('mfma-fp32', poison=False)
    assert np.isfinite(values).all() and np.isfinite(info.gen_loss)
    np.testing.assert_allclose(values, want, rtol=1e-5, atol=1e-7)
    assert all(gan.gen_updaters[name].step == 1 for name in gan._pnames)


def _small_run(records, ssn_type='default', poke=None, **over):
    from tc_gan_amd.networks.cwgan import make_gan
    JDS = on.new_JDS()
    cfg = dict(num_sites=10, seqlen=40, skip_steps=30, num_models=6, probes_per_model=2, norm_probes=[0, 0.5],
               include_inhibitory_neurons=True, bandwidths=[0.0625, 0.125, 0.25, 0.75], contrasts=[5., 20.],
               J0=JDS['J'], D0=JDS['D'], S0=JDS['S'], critic_iters_init=3, critic_iters=2, lipschitz_cost=10.0, ssn_type=ssn_type,
               gen=dict(learning_rate=0.01, update_name='adam-wgan', dynamics_cost=1.0, rate_cost=0.01, rate_penalty_threshold=5.0),
               disc=dict(learning_rate=0.01, update_name='adam-wgan', layers=[16, 16], normalization='none',
                         nonlinearity='rectify', precision='fp32'))
    cfg.update(over)
    gan, _ = make_gan(cfg)
    ncols = len(gan.bandwidths) * len(gan.contrasts) * len(gan.norm_probes) * 2
    gan.set_dataset(np.random.RandomState(3).rand(9, ncols) * 10)
    it = gan.learning()
    out, taken = [], 0
    for k in range(records):
        info = next(it)
        out.append(info.disc_loss if info.is_discriminator else info.gen_loss)
        if not info.is_discriminator:
            taken += gan._next_ctx is not None
            if poke is not None and info.gen_step == poke:
                gan.gen.J = np.asarray(gan.gen.J) * 1.01          # (somebody changes the generator between two iterations)
    st = gan.rng.get_state()
    return np.array(out), st[1].copy(), int(st[2]), gan.get_gen_param(), gan.disc.get_flat().copy(), taken


@pytest.mark.parametrize('over', [dict(), dict(ssn_type='deg-heteroin'), dict(ssn_type='heteroin', V=[0.3, 0.1]), dict(z_host_draw=True)])
def test_first_critic_forward_queued_behind_the_generator_update_gives_the_same_run(over, monkeypatch):
    """`_prequeue_next_disc`: the next iteration's first critic forward is queued behind the optimizer launch, W (and the input
    variability) formed from the device-resident parameters (`ssn_build_w_devparams_f32`), before the host has read the new
    values.  Same records, parameters and RandomState as the loop that reads first (TCGAN_PREQUEUE=0), bit for bit -- also
    when somebody changes a parameter between two iterations (the prepared step is dropped and made again)."""
    from tc_gan_amd.networks import cwgan
    for poke in (None, 2):
        monkeypatch.setattr(cwgan, '_PREQUEUE', True)
        a = _small_run(20, poke=poke, **over)
        monkeypatch.setattr(cwgan, '_PREQUEUE', False)
        b = _small_run(20, poke=poke, **over)
        assert a[5] >= 4 and b[5] == 0              # (the prepared step was there after every generator step / never)
        assert np.array_equal(a[0], b[0]) and np.isfinite(a[0]).all()
        assert np.array_equal(a[1], b[1]) and a[2] == b[2]
        for x, y in zip(a[3], b[3]):
            assert np.array_equal(x, y)
        assert np.array_equal(a[4], b[4])
