"""BPTT-based moment matching of the SSN tuning-curve generator on the GPU.

Mirror of ``tc_gan/networks/moment_matching.py``: the generator is trained so that the minibatch mean and
variance of every tuning-curve channel match the data's (`MOMENT_WEIGHT_TYPES` define the weights), plus
the dynamics / rate penalties.  Same forward and BPTT kernels as the WGAN path; the scalar loss and its
gradient w.r.t. the tuning curves come from ``ssn_moment_sums_f32`` / ``ssn_moment_loss_grad_f32``
(csrc/ssn_aux.hip).  With several ranks the minibatch is sharded over models: one all-reduce of the
per-channel sums before the loss, one of the parameter gradients after the adjoint sweep.
"""
import itertools

import numpy as np
import torch

from .. import clib
from ..clib import libssnode
from ..critic import Updater
from ..utils import Namespace, StopWatch, as_randomstate
from .cwgan import _v_bounds, GradientAllReducer
from .ssn import TuningCurveGenerator
from .wgan import DEFAULT_PARAMS as _WGAN_DEFAULTS, grid_stimulator_inputs, probes_from_stim_space

# moment_matching.py:17-23
DEFAULT_PARAMS = dict(_WGAN_DEFAULTS, moment_weight_type='mean', **_WGAN_DEFAULTS['gen'])
del DEFAULT_PARAMS['gen']
del DEFAULT_PARAMS['disc']

MOMENT_WEIGHT_TYPES = ('mean', 'ew_mean', 'ew_relative')


def sample_moments(samples):
    """moment_matching.py:91-104: (sample_size, channels) -> (2, channels) = (mean, population variance)."""
    samples = np.asarray(samples)
    return np.asarray([samples.mean(axis=0), samples.var(axis=0)])


def calc_moment_weights(data, moment_weight_type='mean', moment_weights_regularization=1e-3, lam=1.0):
    """`BPTTMomentMatcher.set_dataset` (moment_matching.py:348-367) -> (data_moments, moment_weights)."""
    data = np.asarray(data)
    data_moments = sample_moments(data)
    eps = moment_weights_regularization
    num = np.broadcast_to([[1], [lam]], data_moments.shape)
    if moment_weight_type == 'mean':
        den = data.mean()
        weights = num / np.array([[den ** 2], [den ** 4]])
    elif moment_weight_type == 'ew_mean':
        den = data_moments[0] + eps
        weights = num / np.array([den ** 2, den ** 4])
    elif moment_weight_type == 'ew_relative':
        weights = num / (data_moments + eps) ** 2
    else:
        raise ValueError('Unknown moment_weight_type = {}'.format(moment_weight_type))
    return data_moments, np.array(weights, dtype='float64')


class BPTTMomentMatcher(object):
    """moment_matching.py:260-399 (with the trainer of 107-257 folded in)."""

    def __init__(self, gen, gen_updaters, bandwidths, contrasts, lam, moment_weights_regularization,
                 include_inhibitory_neurons, rate_penalty_threshold, moment_weight_type, dynamics_cost, rate_cost,
                 param_bounds, seed=0):
        assert moment_weight_type in MOMENT_WEIGHT_TYPES
        self.gen = gen
        self.gen_updaters = gen_updaters
        self.lam = lam
        self.moment_weights_regularization = moment_weights_regularization
        self.moment_weight_type = moment_weight_type
        self.rng = as_randomstate(seed)          # (the reference's stream; device draws hand their state back lazily)
        self.bandwidths = bandwidths
        self.contrasts = contrasts
        self.reducer = GradientAllReducer()
        self.global_batchsize = gen.batchsize * self.reducer.world
        # the host noise stream is drawn for the GLOBAL minibatch and sliced per rank (as in the cWGAN)
        self.stimulator_contrasts, self.stimulator_bandwidths = grid_stimulator_inputs(
            contrasts, bandwidths, gen.batchsize)
        self.include_inhibitory_neurons = include_inhibitory_neurons
        self.rate_penalty_threshold = rate_penalty_threshold
        self.dynamics_cost = dynamics_cost
        self.rate_cost = rate_cost
        self.param_bounds = param_bounds
        self._pnames = [name for name, _ in gen.get_all_params()]
        self._gparams = {name: torch.zeros(int(np.size(value)), device='cuda', dtype=torch.float32)
                         for name, value in gen.get_all_params()}

    batchsize = property(lambda self: self.global_batchsize)
    num_neurons = property(lambda self: self.gen.num_neurons)
    num_sites = property(lambda self: self.gen.num_sites)

    @property
    def num_mom_conds(self):
        """Number of conditions in which moments are evaluated (moment_matching.py:325-330)."""
        return self.gen.num_tcdom * len(self.gen.probes)

    @property
    def sample_sites(self):
        probes = list(self.gen.probes)
        return probes[:len(probes) // 2] if self.include_inhibitory_neurons else probes

    def get_gen_param(self):
        return [self.gen.J.copy(), self.gen.D.copy(), self.gen.S.copy()]

    def set_dataset(self, data):
        self.data_moments, self.moment_weights = calc_moment_weights(
            data, self.moment_weight_type, self.moment_weights_regularization, self.lam)
        self._dm = torch.as_tensor(np.ascontiguousarray(self.data_moments), device='cuda', dtype=torch.float64)
        self._w = torch.as_tensor(np.ascontiguousarray(self.moment_weights), device='cuda', dtype=torch.float64)

    def prepare(self):
        """Nothing to compile."""

    def _draw_noise(self):
        if self.gen._zgen is not None:
            return {}
        rows = None
        if self.reducer.on:
            per = self.gen.batchsize
            rows = (self.reducer.rank * per, (self.reducer.rank + 1) * per)
        return self.gen.gen_noise(self.rng, stimulator_bandwidths=np.empty((self.global_batchsize, 1)), rows=rows)

    def moment_loss_grad(self, x):
        """x (B_local, D) fp32 CUDA -> (gx, L0, gen_moments (2, D) numpy)."""
        x = x.to(torch.float32).contiguous()
        B, D = x.shape
        stream = torch.cuda.current_stream().cuda_stream
        sums = torch.empty((2, D), device='cuda', dtype=torch.float64)
        clib.check(libssnode.ssn_moment_sums_f32(x.data_ptr(), B, D, sums.data_ptr(), stream), 'ssn_moment_sums_f32')
        if self.reducer.on:
            self.reducer.dist.all_reduce(sums, op=self.reducer.dist.ReduceOp.SUM)
        gx = torch.empty_like(x)
        out = torch.empty(1 + 2 * D, device='cuda', dtype=torch.float64)
        clib.check(libssnode.ssn_moment_loss_grad_f32(
            x.data_ptr(), sums.data_ptr(), float(self.global_batchsize), self._dm.data_ptr(), self._w.data_ptr(),
            B, D, gx.data_ptr(), out.data_ptr(), stream), 'ssn_moment_loss_grad_f32')
        host = out.cpu().numpy()
        return gx, float(host[0]), host[1:].reshape(2, D)

    def train_generator(self, info):
        with self.train_watch:
            noise = self._draw_noise()
            out = self.gen.forward(rng=self.rng, save=True, stimulator_bandwidths=self.stimulator_bandwidths,
                                   stimulator_contrasts=self.stimulator_contrasts,
                                   model_rate_penalty_threshold=self.rate_penalty_threshold, **noise)
            gx, l0, gen_moments = self.moment_loss_grad(out.prober_tuning_curve)
            # L0 is a function of the GLOBAL minibatch: the local adjoint sweeps add up over ranks, whereas the
            # penalties are per-rank means -> scale gx so that the rank MEAN below is right for both
            gdict = self.gen.backward(gx * self.reducer.world, self.dynamics_cost, self.rate_cost)
            pens = torch.stack([out.model_dynamics_penalty.reshape(()).to(torch.float32),
                                out.model_rate_penalty.reshape(()).to(torch.float32)])
            grads = torch.as_tensor(np.concatenate([np.ravel(gdict[name]) for name in self._pnames]), device='cuda',
                                    dtype=torch.float32)
            self.reducer.mean_(grads, pens)
            off = 0
            for name in self._pnames:                                  # moment_matching.py:245-257 (clip_params)
                value = np.asarray(getattr(self.gen, name))
                p = self._gparams[name]
                p.copy_(torch.as_tensor(np.array(value).ravel(), dtype=torch.float32))
                self.gen_updaters[name](p, grads[off:off + p.numel()], clip=self.param_bounds[name])
                off += p.numel()
                setattr(self.gen, name, p.cpu().numpy().astype('float64').reshape(value.shape))
            info.dynamics_penalty = float(pens[0])
            info.rate_penalty = float(pens[1])
            info.loss = l0 + self.dynamics_cost * info.dynamics_penalty + self.rate_cost * info.rate_penalty
            info.gen_moments = gen_moments
        info.train_time = self.train_watch.sum()
        return info

    def learning(self):
        for step in itertools.count():
            self.train_watch = StopWatch()
            yield self.train_generator(Namespace(step=step))


def make_moment_matcher(config):
    """make_moment_matcher(config: dict) -> (BPTTMomentMatcher, dict of unconsumed config) (moment_matching.py:437-472)."""
    kwargs = dict(DEFAULT_PARAMS, **config)
    take = kwargs.pop
    bandwidths, contrasts, num_sites = take('bandwidths'), take('contrasts'), take('num_sites')
    include_inhibitory_neurons = take('include_inhibitory_neurons')
    probes = probes_from_stim_space(take('sample_sites'), num_sites, include_inhibitory_neurons)
    ssn_type = take('ssn_type', 'default')
    ssn_impl = take('ssn_impl', 'default')
    if ssn_impl not in ('default', 'mapclone'):
        raise ValueError('Unknown ssn_impl: {}'.format(ssn_impl))
    if 'V0' in kwargs:
        kwargs['V'] = kwargs.pop('V0')
    V = kwargs.pop('V', 0)
    dist_in = kwargs.pop('dist_in', 'bernoulli')
    reducer = GradientAllReducer()
    batchsize = take('batchsize')
    assert batchsize % reducer.world == 0, 'batchsize must be divisible by the number of ranks'
    gen = TuningCurveGenerator(
        num_sites=num_sites, num_tcdom=len(bandwidths) * len(contrasts), smoothness=take('smoothness'),
        J=take('J0'), D=take('D0'), S=take('S0'), k=take('k'), n=take('n'), tau_E=take('tau_E'), tau_I=take('tau_I'),
        dt=take('dt'), io_type=take('io_type'), seqlen=take('seqlen'), skip_steps=take('skip_steps'),
        batchsize=batchsize // reducer.world, probes=probes, include_rate_penalty=True,
        include_time_avg=take('include_time_avg', False), unroll_scan=take('unroll_scan', False),
        dtype=take('gen_dtype', 'float32'), z_device_seed=take('z_device_seed', None),
        shard=(reducer.rank, reducer.world),
        ssn_type=ssn_type, V=V, dist_in=dist_in, gen_kernel=take('gen_kernel', 'auto'), z_host_draw=take('z_host_draw', False))
    bounds = {name: (take(name + '_min', 1e-3), take(name + '_max', 10.0)) for name in 'JDS'}
    bounds['V'] = _v_bounds(take('V_min', 0), take('V_max', 1), ssn_type)
    upd_cfg = {k: take(k) for k in ('learning_rate', 'update_name', 'update_config', 'reg_l2_penalty', 'reg_l2_decay',
                                    'reg_l1_penalty', 'reg_l1_decay') if k in kwargs}
    mm = BPTTMomentMatcher(
        gen, {name: Updater(**upd_cfg) for name in 'VJDS'}, bandwidths, contrasts,
        lam=take('lam'), moment_weights_regularization=take('moment_weights_regularization'),
        include_inhibitory_neurons=include_inhibitory_neurons,
        rate_penalty_threshold=take('rate_penalty_threshold'), moment_weight_type=take('moment_weight_type'),
        dynamics_cost=take('dynamics_cost', 1.0), rate_cost=take('rate_cost'), param_bounds=bounds,
        seed=take('seed', 0))
    return mm, kwargs
