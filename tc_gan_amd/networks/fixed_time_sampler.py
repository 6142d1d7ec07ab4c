"""Fixed-time tuning-curve sampler (mirror of ``tc_gan/networks/fixed_time_sampler.py``): the generator
itself, run at given parameters with a fixed prober."""
import numpy as np

from .ssn import TuningCurveGenerator
from .utils import gridify_tc_samples
from .wgan import DEFAULT_PARAMS as _WGAN_DEFAULTS, grid_stimulator_inputs, probes_from_stim_space


def _make_new_JDS():
    """fixed_time_sampler.py:12-21 ("more stable" parameters)."""
    J = np.array([[.0957, .0638], [.1197, .0479]])
    D = np.array([[.7660, .5106], [.9575, .3830]])
    S = np.array([[.6667, .2], [1.333, .2]]) / 8
    D_new = D / 2
    J_new = J + D / 2 - D_new / 2
    return dict(J=J_new, D=D_new, S=S)


new_JDS = _make_new_JDS()

DEFAULT_PARAMS = dict(_WGAN_DEFAULTS, V=0.5, seed=0, norm_probes=[0], include_inhibitory_neurons=False, **new_JDS)
del DEFAULT_PARAMS['sample_sites']
del DEFAULT_PARAMS['gen']
del DEFAULT_PARAMS['disc']


class FixedTimeTuningCurveSampler(object):
    """fixed_time_sampler.py:38-246 (sampling surface)."""

    @classmethod
    def from_dict(cls, dct):
        cfg = dict(DEFAULT_PARAMS, **dct)
        cfg.pop('ssn_impl', None)
        if cfg.get('ssn_type', 'default') == 'default':
            for key in ('V', 'dist_in'):
                cfg.pop(key, None)
        bandwidths, contrasts = cfg.pop('bandwidths'), cfg.pop('contrasts')
        num_sites = cfg.pop('num_sites')
        probes = probes_from_stim_space(cfg.pop('norm_probes'), num_sites, cfg.pop('include_inhibitory_neurons'))
        seed = cfg.pop('seed')
        gen = TuningCurveGenerator(num_sites=num_sites, num_tcdom=len(bandwidths) * len(contrasts), probes=probes,
                                   include_rate_penalty=False, **cfg)
        return cls(gen, bandwidths, contrasts, seed)

    def __init__(self, gen, bandwidths, contrasts, seed):
        self.gen = gen
        self.bandwidths = np.asarray(bandwidths)
        self.contrasts = np.asarray(contrasts)
        assert self.bandwidths.ndim == 1 and self.contrasts.ndim == 1
        self.rng = np.random.RandomState(seed)
        self.stimulator_contrasts, self.stimulator_bandwidths = grid_stimulator_inputs(
            self.contrasts, self.bandwidths, self.batchsize)

    num_sites = property(lambda self: self.gen.num_sites)
    num_neurons = property(lambda self: self.gen.num_neurons)
    batchsize = property(lambda self: self.gen.batchsize)

    def forward(self, raw=True):
        return self.gen.forward(self.rng, stimulator_bandwidths=self.stimulator_bandwidths,
                                stimulator_contrasts=self.stimulator_contrasts)

    def sample(self, repeat=1):
        return np.concatenate([self.forward().prober_tuning_curve.cpu().numpy() for _ in range(repeat)])

    def timepoints(self):
        return np.linspace(self.gen.dt, self.gen.dt * self.gen.seqlen, self.gen.seqlen)

    @property
    def include_inhibitory_neurons(self):
        return bool(np.any(self.gen.probes >= self.gen.num_sites))

    def tc_samples_as_grid(self, data):
        probes = [p for p in self.gen.probes if p < self.gen.num_sites]
        return gridify_tc_samples(data, num_contrasts=len(self.contrasts), num_bandwidths=len(self.bandwidths),
                                  num_cell_types=int(self.include_inhibitory_neurons) + 1, num_probes=len(probes))

    def prepare(self):
        """Nothing to compile."""

    @classmethod
    def from_learner(cls, learner, batchsize, seed, **override):
        """fixed_time_sampler.py:208-246: copy the learner's generator settings, override parameters."""
        cfg = learner.gen.to_config()
        for key in ('probes', 'include_rate_penalty', 'ssn_impl', 'num_tcdom', 'batchsize'):
            cfg.pop(key, None)
        cfg.update(override)
        bandwidths, contrasts = learner.bandwidths, learner.contrasts
        if learner.gen.probes is not None:         # fixed prober (WGAN, moment matcher): reuse its probes
            probes = learner.gen.probes
        else:                                      # conditional prober (cWGAN)
            probes = probes_from_stim_space(learner.norm_probes, learner.num_sites, learner.include_inhibitory_neurons)
        gen = TuningCurveGenerator(num_tcdom=len(bandwidths) * len(contrasts), probes=probes, batchsize=batchsize,
                                   include_rate_penalty=False, **cfg)
        return cls(gen, bandwidths, contrasts, seed)
