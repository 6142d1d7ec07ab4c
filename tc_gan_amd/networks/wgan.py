"""Unconditional BPTT Wasserstein GAN on the GPU -- host-side mirror of ``tc_gan/networks/wgan.py``.

* `DEFAULT_PARAMS`, `grid_stimulator_inputs`, `probes_from_stim_space` (wgan.py:39-63, 293-296, 447-451; kept in
  `networks._common`, re-exported here under the reference's names);
* `UnConditionalDiscriminator` (wgan.py:66-97): the critic of `tc_gan_amd.critic` without condition columns;
* `BPTTWassersteinGAN` (wgan.py:299-444): fixed prober, fixed stimulus grid, `random_minibatches` of the truth table;
  ``learning()`` yields the same `Namespace` records in the same order and consumes the shared RandomState in the
  reference's order (minibatch shuffle, eps, zs[, zs_in]; generator step: zs[, zs_in]);
* `make_gan(config) -> (gan, rest)` (wgan.py:454-509).

The update loop itself (queueing, the one-call critic step, data parallelism with one collective per update, checkpoints)
is the conditional GAN's (`networks.cwgan.ConditionalBPTTWassersteinGAN`): the two differ in what a minibatch is, and in
whether the critic sees condition columns -- nothing else, so only that is written here.
"""
import numpy as np

from ..critic import Critic, Updater
from ..utils import random_minibatches
from ._common import DEFAULT_PARAMS, grid_stimulator_inputs, probes_from_stim_space  # noqa: F401  (re-exported)
from .cwgan import ConditionalBPTTWassersteinGAN, GradientAllReducer, _v_bounds
from .ssn import TuningCurveGenerator


def UnConditionalDiscriminator(shape, layers, normalization='none', nonlinearity='rectify', loss_type='WD',
                               net_options=None, precision='fp32', seed=0):
    """wgan.py:66-97: critic on the `shape[1]` columns of a tuning-curve batch, no conditions.  (`net_options`: the layer
    options of simple_discriminator.py, i.e. `use_scale`; `loss_type` must be the Wasserstein one.)"""
    if loss_type != 'WD':
        raise ValueError('the BPTT GANs are Wasserstein GANs (loss_type WD), got {!r}'.format(loss_type))
    return Critic(nx=int(shape[1]), layers=layers, normalization=normalization, nonlinearity=nonlinearity,
                  precision=precision, seed=seed, conditional=False, net_options=net_options)


class UnconditionalMinibatch(object):
    """One step's data: `batchsize` rows of the truth table (wgan.py:385-390) beside the stimulus grid every model of the
    step sees (`grid_stimulator_inputs`, wgan.py:331-332).  Same surface as `cwgan.ConditionalMinibatch`."""

    conditions = None

    def __init__(self, tuning_curves, stimulator_contrasts, stimulator_bandwidths):
        self.tuning_curves = tuning_curves
        self.stimulator_contrasts = stimulator_contrasts
        self.stimulator_bandwidths = stimulator_bandwidths
        assert len(tuning_curves) == len(stimulator_contrasts) == len(stimulator_bandwidths)

    num_models = batchsize = property(lambda self: len(self.tuning_curves))

    @property
    def gen_kwargs(self):
        return dict(stimulator_bandwidths=self.stimulator_bandwidths, stimulator_contrasts=self.stimulator_contrasts)

    def shard(self, rank, world):
        per = self.batchsize // world
        sl = slice(rank * per, (rank + 1) * per)
        return UnconditionalMinibatch(self.tuning_curves[sl], self.stimulator_contrasts[sl], self.stimulator_bandwidths[sl])


class BPTTWassersteinGAN(ConditionalBPTTWassersteinGAN):
    """wgan.py:299-444."""

    def __init__(self, gen, disc, gen_updaters, disc_updater, bandwidths, contrasts, include_inhibitory_neurons,
                 rate_penalty_threshold, batchsize, critic_iters_init, critic_iters, lipschitz_cost,
                 disc_rate_penalty_bound, dynamics_cost, rate_cost, param_bounds, seed=0):
        self.gen, self.disc = gen, disc
        self.gen_updaters, self.disc_updater = gen_updaters, disc_updater
        self.bandwidths, self.contrasts = np.asarray(bandwidths), np.asarray(contrasts)
        self.include_inhibitory_neurons = include_inhibitory_neurons
        self.rate_penalty_threshold = rate_penalty_threshold
        self._batchsize = int(batchsize)                       # of the whole job; gen.batchsize is this rank's share
        self.critic_iters_init, self.critic_iters = critic_iters_init, critic_iters
        self.lipschitz_cost = lipschitz_cost
        self.disc_rate_penalty_bound = disc_rate_penalty_bound
        self.dynamics_cost, self.rate_cost = dynamics_cost, rate_cost
        self.param_bounds = param_bounds
        self._init_loop_state(seed)
        assert self._batchsize % self.reducer.world == 0, 'batchsize must be divisible by the number of ranks'
        self.stimulator_contrasts, self.stimulator_bandwidths = grid_stimulator_inputs(
            self.contrasts, self.bandwidths, self._batchsize)           # wgan.py:331-332

    batchsize = property(lambda self: self._batchsize)
    num_models = batchsize
    probes_per_model = 1

    @property
    def sample_sites(self):
        """wgan.py:344-350: the probed sites (E half of the probe list when inhibitory neurons are probed too)."""
        probes = [int(p) for p in self.gen.probes]
        return probes[:len(probes) // 2] if self.include_inhibitory_neurons else probes

    def set_dataset(self, data, **kwargs):
        kwargs.setdefault('seed', self.rng)       # the shuffles SHARE the GAN's RandomState (wgan.py:364-366)
        rows = random_minibatches(self._batchsize, np.asarray(data), **kwargs)
        self.dataset = (UnconditionalMinibatch(xd, self.stimulator_contrasts, self.stimulator_bandwidths) for xd in rows)


def make_gan(config):
    """make_gan(config: dict) -> (GAN, dict): build the GAN and return the unconsumed part of `config` (wgan.py:454-509).
    ``config['gen']`` / ``config['disc']`` hold the trainer options with the reference's names, as for the conditional
    GAN (`cwgan.make_gan`)."""
    kwargs = dict(DEFAULT_PARAMS, **config)
    gen_cfg = dict(DEFAULT_PARAMS['gen'], **config.get('gen', {}))
    disc_cfg = dict(DEFAULT_PARAMS['disc'], **config.get('disc', {}))
    kwargs.pop('gen', None)
    kwargs.pop('disc', None)
    take = kwargs.pop

    bandwidths, contrasts = take('bandwidths'), take('contrasts')
    num_sites = take('num_sites')
    include_inhibitory_neurons = take('include_inhibitory_neurons')
    probes = probes_from_stim_space(take('sample_sites'), num_sites, include_inhibitory_neurons)
    ssn_type = take('ssn_type', 'default')
    ssn_impl = take('ssn_impl', 'default')
    if ssn_impl not in ('default', 'mapclone'):
        raise ValueError('Unknown ssn_impl: {}'.format(ssn_impl))
    if 'V0' in kwargs:                              # wgan.py:471-472
        kwargs['V'] = kwargs.pop('V0')
    V = kwargs.pop('V', 0)
    dist_in = kwargs.pop('dist_in', 'bernoulli')
    for key in ('V_min', 'V_max'):
        kwargs.pop(key, None)
    reducer = GradientAllReducer()
    batchsize = take('batchsize')
    gen_kernel = gen_cfg.pop('kernel', None) or take('gen_kernel', 'auto')
    kwargs.pop('gen_kernel', None)
    gen = TuningCurveGenerator(
        num_sites=num_sites, num_tcdom=len(bandwidths) * len(contrasts), smoothness=take('smoothness'),
        J=take('J0'), D=take('D0'), S=take('S0'), k=take('k'), n=take('n'),
        tau_E=take('tau_E'), tau_I=take('tau_I'), dt=take('dt'), io_type=take('io_type'),
        seqlen=take('seqlen'), skip_steps=take('skip_steps'), batchsize=batchsize // reducer.world, probes=probes,
        include_rate_penalty=take('include_rate_penalty', True), include_time_avg=take('include_time_avg', False),
        unroll_scan=take('unroll_scan', False), dtype=take('gen_dtype', 'float32'),
        z_device_seed=take('z_device_seed', None), shard=(reducer.rank, reducer.world),
        ssn_type=ssn_type, V=V, dist_in=dist_in, gen_kernel=gen_kernel, z_host_draw=take('z_host_draw', False))
    rate_penalty_threshold = gen_cfg.pop('rate_penalty_threshold')
    disc_rate_penalty_bound = disc_cfg.pop('rate_penalty_bound')
    seed = take('seed', 0)
    disc = UnConditionalDiscriminator(
        shape=gen.output_shape, loss_type='WD', layers=disc_cfg.pop('layers', []),
        normalization=disc_cfg.pop('normalization', 'none'), nonlinearity=disc_cfg.pop('nonlinearity', 'rectify'),
        net_options=disc_cfg.pop('net_options', None), precision=disc_cfg.pop('precision', 'fp32'),
        seed=disc_cfg.pop('init_seed', seed))
    upd_keys = ('learning_rate', 'update_name', 'update_config', 'reg_l2_penalty', 'reg_l2_decay', 'reg_l1_penalty',
                'reg_l1_decay')
    dynamics_cost = gen_cfg.pop('dynamics_cost', 1.0)
    rate_cost = gen_cfg.pop('rate_cost')
    bounds = {name: (gen_cfg.pop(name + '_min', 1e-3), gen_cfg.pop(name + '_max', 10.0)) for name in 'JDS'}
    bounds['V'] = _v_bounds(gen_cfg.pop('V_min', 0), gen_cfg.pop('V_max', 1), ssn_type)   # wgan.py:263-283
    gen_upd_cfg = {k: gen_cfg.pop(k) for k in list(gen_cfg) if k in upd_keys}
    gen_updaters = {name: Updater(**gen_upd_cfg) for name in 'VJDS'}
    disc_updater = Updater(**{k: disc_cfg.pop(k) for k in upd_keys if k in disc_cfg})
    if gen_cfg or disc_cfg:
        raise ValueError('Unknown trainer options: gen={} disc={}'.format(sorted(gen_cfg), sorted(disc_cfg)))
    gan = BPTTWassersteinGAN(
        gen, disc, gen_updaters, disc_updater, bandwidths, contrasts,
        include_inhibitory_neurons=include_inhibitory_neurons, rate_penalty_threshold=rate_penalty_threshold,
        batchsize=batchsize, critic_iters_init=take('critic_iters_init'), critic_iters=take('critic_iters'),
        lipschitz_cost=take('lipschitz_cost'), disc_rate_penalty_bound=disc_rate_penalty_bound,
        dynamics_cost=dynamics_cost, rate_cost=rate_cost, param_bounds=bounds, seed=seed)
    return gan, kwargs
