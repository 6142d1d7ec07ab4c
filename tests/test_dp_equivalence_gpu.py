"""Data-parallel equivalence on the GPU: two ranks (gloo, sharing the one card of the test box) training
with host-side noise must follow the single-process run -- same minibatches, same z rows per model, one
averaged gradient per update."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _config():
    sys.path.insert(0, ROOT)
    from oracle import ssn_numpy as on          # parameters only
    jds = on.new_JDS()
    return dict(
        num_sites=10, seqlen=40, skip_steps=30, num_models=4, probes_per_model=2, norm_probes=[0, 0.5],
        include_inhibitory_neurons=True, bandwidths=[0.0625, 0.125, 0.25, 0.75], contrasts=[5., 20.],
        J0=jds['J'], D0=jds['D'], S0=jds['S'], critic_iters_init=2, critic_iters=2, lipschitz_cost=10.0,
        gen=dict(learning_rate=0.01, update_name='adam-wgan', dynamics_cost=1.0, rate_cost=0.01,
                 rate_penalty_threshold=5.0, J_min=1e-3, J_max=10, D_min=1e-3, D_max=10, S_min=1e-3, S_max=10),
        disc=dict(learning_rate=0.01, update_name='adam-wgan', layers=[16, 16], normalization='none',
                  nonlinearity='rectify', precision='fp32'))


def _train(n_gen_steps=2, z_device_seed=None, **over):
    from tc_gan_amd.networks.cwgan import make_gan
    gan, _ = make_gan(dict(_config(), z_device_seed=z_device_seed, **over))
    data = np.random.RandomState(4).rand(9, 4 * 2 * 2 * 2) * 10
    gan.set_dataset(data)
    it = gan.learning()
    losses, accs, order, norms = [], [], [], []
    done = 0
    while done < n_gen_steps:
        info = next(it)
        if info.is_discriminator:
            losses.append(info.disc_loss)
            accs.append(info.accuracy)
            order.append((info.gen_step, info.disc_step))
            # what DiscParamStatsRecorder.record writes for this record (disc_param_stats): the norms of the critic as THIS step
            # left it, from the one-shot cache the loop fills -- and, beside them, the norms read back from the parameters now
            norms.append(gan.disc.param_nnorms())
        else:
            losses.append(info.gen_loss)
            order.append((info.gen_step, -1))
            done += 1
    # [collectives issued, records in the reference's order?] then the accuracies of the critic steps
    extra = np.array([gan.reducer.calls, float(order == sorted(order, key=lambda t: (t[0], t[1] < 0, t[1])))] + accs)
    _train.norms = np.asarray(norms, dtype='float64')
    return np.concatenate([np.ravel(p) for p in gan.get_gen_param()]), gan.disc.get_flat(), np.array(losses), extra


def _train_unconditional(n_gen_steps=2, gen_kernel='auto'):
    """The unconditional GAN (networks/wgan.py): fixed probes, no condition columns; device noise, batch of 4 draws sharded over
    the ranks; `gen_kernel='duo-fused'`: the one-launch backward under data parallelism."""
    from tc_gan_amd.networks.wgan import make_gan
    cfg = _config()
    gan, _ = make_gan(dict(
        J0=cfg['J0'], D0=cfg['D0'], S0=cfg['S0'], gen=cfg['gen'], disc=cfg['disc'], critic_iters_init=2, critic_iters=2,
        include_inhibitory_neurons=True, lipschitz_cost=10.0, num_sites=10, seqlen=40, skip_steps=30, batchsize=4,
        bandwidths=[0.0625, 0.125, 0.25, 0.75], contrasts=[20.], z_device_seed=7, gen_kernel=gen_kernel))
    ncols = len(gan.bandwidths) * len(gan.contrasts) * len(gan.sample_sites) * 2
    gan.set_dataset(np.random.RandomState(6).rand(9, ncols) * 10)
    it = gan.learning()
    losses, done = [], 0
    while done < n_gen_steps:
        info = next(it)
        losses.append(info.disc_loss if info.is_discriminator else info.gen_loss)
        done += not info.is_discriminator
    return (np.concatenate([np.ravel(p) for p in gan.get_gen_param()]), gan.disc.get_flat(), np.array(losses),
            np.array([gan.reducer.calls]))


def _train_moments(n_steps=3):
    """Moment matching: the loss depends on the moments of the GLOBAL minibatch (one all-reduce of the per-channel
    sums before the loss, one of the parameter gradients after the adjoint sweep)."""
    from tc_gan_amd.networks.moment_matching import make_moment_matcher
    cfg = _config()
    mm, _ = make_moment_matcher(dict(
        num_sites=10, seqlen=40, skip_steps=30, batchsize=6, sample_sites=[0, 0.5], include_inhibitory_neurons=True,
        bandwidths=[0.0625, 0.25, 0.75], contrasts=[5., 20.], J0=cfg['J0'], D0=cfg['D0'], S0=cfg['S0'], lam=0.1,
        moment_weights_regularization=1e-3, moment_weight_type='ew_mean', learning_rate=0.01,
        update_name='adam-wgan', dynamics_cost=1.0, rate_cost=0.01, rate_penalty_threshold=5.0))
    mm.set_dataset(np.random.RandomState(5).rand(9, mm.num_mom_conds) * 8)
    it = mm.learning()
    infos = [next(it) for _ in range(n_steps)]
    return (np.concatenate([np.ravel(p) for p in mm.get_gen_param()]),
            np.concatenate([np.ravel(i.gen_moments) for i in infos]), np.array([i.loss for i in infos]))


def _find_fixed_points():
    """Rejection sampling of fixed points (truth data): some draws fail at the rate bound, so the accepted set
    depends on the global submission order."""
    from oracle import ssn_numpy as on          # weights / stimuli only
    from tc_gan_amd.ssnode import find_fixed_points
    N = 24
    jds = on.new_JDS()
    rs = np.random.RandomState(3)

    def gen():
        while True:
            z = rs.rand(2 * N, 2 * N)
            # every third draw is scaled up so that its dynamics blow past the rate bound (rejected, code 2)
            scale = 40.0 if rs.rand() < 0.35 else 1.0
            yield z, on.generate_weight(N, jds['J'] * scale, jds['D'], jds['S'], z)
    exts = on.stimulus_input([0.25, 1.0], np.linspace(-.5, .5, N), 1 / 32., [20.])
    zs, xs, info = find_fixed_points(9, gen(), exts, k=0.01, n=2.2, io_type='asym_power', rate_stop_at=2000.,
                                     max_iter=3000, atol=1e-6, dtype='float64')
    return zs.ravel(), xs.ravel(), np.array([info.rejections, info.unused] + [info.counter.get(c, 0) for c in (1, 2)])


def _worker(rank, world, port, out, what='gan'):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    res = {'gan': _train, 'gan_devnoise': lambda: _train(z_device_seed=31), 'moments': _train_moments,
           'gan_heteroin': lambda: _train(n_gen_steps=3, ssn_type='deg-heteroin', V=0.3),
           'find': _find_fixed_points, 'wgan': _train_unconditional,
           'wgan_fused': lambda: _train_unconditional(gen_kernel='duo-fused')}[what]()
    if what == 'gan':
        res = res + (_train.norms,)
    out.put((rank,) + res)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_follow_the_single_process_run():
    sys.path.insert(0, ROOT)
    jds1, critic1, losses1, extra1 = _train()
    assert extra1[0] == 0 and extra1[1] == 1
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = 29700 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([out.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(2):
        np.testing.assert_allclose(res[r][3], losses1, rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(res[r][1], jds1, rtol=2e-4, atol=1e-6)
        np.testing.assert_allclose(res[r][2], critic1, rtol=5e-3, atol=5e-5)
    np.testing.assert_array_equal(res[0][2], res[1][2])          # replicas stay bit-identical
    np.testing.assert_array_equal(res[0][1], res[1][1])
    # ONE collective per update: 2 + 2 critic steps and 2 generator steps = 6 all-reduces.  The accuracy of the
    # updated critic (cwgan.py:505-507) has no collective of its own: it rides in the next one, and every record still
    # reaches the driver in the reference's order with the JOB-wide value (= the single process's, which sees all samples)
    for r in range(2):
        assert res[r][4][0] == 6 and res[r][4][1] == 1
        np.testing.assert_allclose(res[r][4][2:], extra1[2:], atol=1e-6)
    np.testing.assert_array_equal(res[0][4], res[1][4])
    # disc_param_stats: every critic record carries the norms of the critic as ITS step left it (the records of a
    # data-parallel run are handed over one collective late and used to pick up the next step's norms)
    norms1 = _train.norms
    assert norms1.shape[0] == 4 and np.abs(np.diff(norms1, axis=0)).max() > 1e-4 * np.abs(norms1).max()   # (steps differ: a shift would show)
    for r in range(2):
        np.testing.assert_allclose(res[r][5], norms1, rtol=1e-3)


def test_two_ranks_with_device_noise_follow_the_single_process_run():
    """`z_device_seed` (performance mode): the ranks fill disjoint rows of ONE Philox stream, so the 2-rank job trains on
    the same weight draws as the single process -- not on two copies of half of them."""
    sys.path.insert(0, ROOT)
    jds1, critic1, losses1, _ = _train(z_device_seed=31)
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = 26700 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out, 'gan_devnoise')) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([out.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(2):
        np.testing.assert_allclose(res[r][3], losses1, rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(res[r][1], jds1, rtol=2e-4, atol=1e-6)
        np.testing.assert_allclose(res[r][2], critic1, rtol=5e-3, atol=5e-5)
    np.testing.assert_array_equal(res[0][2], res[1][2])


def test_two_ranks_heterogeneous_input_on_the_reference_stream_follow_the_single_process_run():
    """deg-heteroin on the default noise (the reference's RandomState continued on the device): each rank generates its rows
    of zs AND its rows of zs_in behind it (two windows far apart in the stream: a launch each, `ssn_build_w_mt19937_tail_begin_f32`
    / `ssn_mt19937_random_sample_tail_begin_f32`), the first critic forward of an iteration is queued behind the optimizer
    launch with V read on the device -- and the job follows the single process."""
    sys.path.insert(0, ROOT)
    jds1, critic1, losses1, _ = _train(n_gen_steps=3, ssn_type='deg-heteroin', V=0.3)
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = 25700 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out, 'gan_heteroin')) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([out.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(2):
        np.testing.assert_allclose(res[r][3], losses1, rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(res[r][1], jds1, rtol=2e-4, atol=1e-6)
        np.testing.assert_allclose(res[r][2], critic1, rtol=5e-3, atol=5e-5)
    np.testing.assert_array_equal(res[0][2], res[1][2])
    np.testing.assert_array_equal(res[0][1], res[1][1])


@pytest.mark.parametrize('what', ['wgan', 'wgan_fused'])
def test_two_ranks_unconditional_gan_follows_the_single_process_run(what):
    """networks/wgan.py under data parallelism (the loop, the reducer and the sharded Philox rows are the conditional GAN's):
    same losses and parameters as one process, replicas bit-identical, six collectives for 4 critic + 2 generator updates --
    with the two-launch backward and with the one-launch one (`duo-fused`)."""
    sys.path.insert(0, ROOT)
    kernel = 'duo-fused' if what == 'wgan_fused' else 'auto'
    jds1, critic1, losses1, calls1 = _train_unconditional(gen_kernel=kernel)
    assert calls1[0] == 0
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = 25700 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out, what)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([out.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(2):
        np.testing.assert_allclose(res[r][3], losses1, rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(res[r][1], jds1, rtol=2e-4, atol=1e-6)
        np.testing.assert_allclose(res[r][2], critic1, rtol=5e-3, atol=5e-5)
        assert res[r][4][0] == 6
    np.testing.assert_array_equal(res[0][2], res[1][2])
    np.testing.assert_array_equal(res[0][1], res[1][1])


def test_two_ranks_moment_matching_follows_the_single_process_run():
    sys.path.insert(0, ROOT)
    jds1, moments1, losses1 = _train_moments()
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = 28700 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out, 'moments')) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([out.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(2):
        np.testing.assert_allclose(res[r][3], losses1, rtol=2e-4, atol=1e-7)
        np.testing.assert_allclose(res[r][2], moments1, rtol=2e-4, atol=1e-7)
        np.testing.assert_allclose(res[r][1], jds1, rtol=2e-4, atol=1e-6)
    np.testing.assert_array_equal(res[0][1], res[1][1])


def test_two_ranks_find_the_same_fixed_point_sample():
    sys.path.insert(0, ROOT)
    zs1, xs1, counts1 = _find_fixed_points()
    assert counts1[0] > 0                      # the case has rejections
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = 27700 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out, 'find')) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([out.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(2):
        np.testing.assert_array_equal(res[r][1], zs1)            # the same draws accepted, in the same order
        np.testing.assert_allclose(res[r][2], xs1, rtol=1e-12)
        np.testing.assert_array_equal(res[r][3], counts1)
