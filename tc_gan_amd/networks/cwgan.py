"""Conditional BPTT Wasserstein GAN on the GPU -- host-side mirror of ``tc_gan/networks/cwgan.py``.

Same components, names and schedule as the reference:

* `ConditionalMinibatch`, `RandomChoiceSampler` (cwgan.py:217-391) -- host index selection with the
  reference's RandomState consumption order (SURVEY.md section 8a gotcha 1);
* `ConditionalBPTTWassersteinGAN` (cwgan.py:410-552) -- ``learning()`` is a Python generator yielding a
  `Namespace` per critic step (``is_discriminator=True``) and per generator step, with the fields the
  driver/recorders read;
* `make_gan(config) -> (gan, rest)` (cwgan.py:555-614).

What differs is where the arithmetic runs: generator forward / BPTT in ``csrc/ssn_gen.hip``, critic
passes and optimizers in ``csrc/ssn_critic.hip``; tuning curves stay on the device between the
generator and the critic (the reference round-trips them through numpy, cwgan.py:478-505).

Data parallelism (new; the reference is single device): with ``torch.distributed`` initialised every
rank keeps a replica of the critic and of (J, D, S), works on its contiguous slice of the
``num_models`` weight draws, and ONE all-reduce per update (flat buffer: critic grads | 12 generator
grads | loss scalars) makes the replicas take identical optimizer steps.
"""
from logging import getLogger

import os

import time

import numpy as np
import torch

from .. import clib
from ..critic import Critic, Updater
from ..execution import KnownError
from ..gradient_expressions.utils import sample_sites_from_stim_space
from ..utils import Namespace, StopWatch, as_randomstate, cartesian_product, to_device, to_device_packed
from .ssn import TuningCurveGenerator
from .utils import gridify_tc_samples
from ._common import DEFAULT_PARAMS as _WGAN_DEFAULTS

# (A/B switch: TCGAN_PREQUEUE=0 queues the first critic forward of an iteration only after the generator step's record was read)
_PREQUEUE = os.environ.get('TCGAN_PREQUEUE', '1') != '0'
# (TCGAN_SUBSET_RETRY=0: a refused generator step is recomputed on the fp32 kernels for ALL draws, not only the refused ones)
_SUBSET_RETRY = os.environ.get('TCGAN_SUBSET_RETRY', '1') != '0'

logger = getLogger(__name__)

# cwgan.py:24-32
DEFAULT_PARAMS = dict(
    _WGAN_DEFAULTS,
    num_models=1,
    probes_per_model=1,
    e_ratio=0.8,
    hide_cell_type=False,
)
del DEFAULT_PARAMS['batchsize']
del DEFAULT_PARAMS['sample_sites']


class ConditionalMinibatch(object):
    """cwgan.py:217-274."""

    def __init__(self, tc_md, conditions_md, bandwidths, contrasts):
        self.tc_md = tc_md
        self.conditions_md = np.asarray(conditions_md)
        self.bandwidths = bandwidths
        self.contrasts = contrasts
        assert self.tc_md.shape[:-1] == self.conditions_md.shape[1:]
        assert self.tc_md.shape[-1] == len(bandwidths)

    num_models = property(lambda self: self.tc_md.shape[0])
    probes_per_model = property(lambda self: self.tc_md.shape[1])
    num_bandwidths = property(lambda self: self.tc_md.shape[2])

    @property
    def batchsize(self):
        return self.num_models * self.probes_per_model

    @property
    def gen_kwargs(self):
        contrasts, bandwidths = np.broadcast_arrays(self.contrasts.reshape((-1, 1)),
                                                    self.bandwidths.reshape((1, -1)))
        _, norm_probes, cell_types = self._conditions_T
        return dict(
            stimulator_bandwidths=bandwidths.astype('float32'),
            stimulator_contrasts=contrasts.astype('float32'),
            prober_norm_probes=norm_probes.astype('float32'),
            prober_cell_types=cell_types.astype('uint16'),
            prober_model_ids=self.model_ids.astype('uint16'))

    @property
    def tuning_curves(self):
        return self.tc_md.reshape((self.batchsize, self.num_bandwidths))

    @property
    def conditions(self):
        return self._conditions_T.T

    @property
    def _conditions_T(self):
        return self.conditions_md.reshape((-1, self.batchsize))

    @property
    def model_ids(self):
        ids = np.arange(self.num_models, dtype='uint16').reshape((-1, 1))
        return np.broadcast_to(ids, self.conditions_md.shape[1:]).flatten()

    def shard(self, rank, world):
        """The contiguous block of models of one data-parallel rank (same arrays, fewer rows)."""
        per = self.num_models // world
        sl = slice(rank * per, (rank + 1) * per)
        return ConditionalMinibatch(self.tc_md[sl], self.conditions_md[:, sl], self.bandwidths, self.contrasts[sl])


class RandomChoiceSampler(object):
    """Minibatch sampler based on random choice (cwgan.py:277-391)."""

    @classmethod
    def from_grid_data(cls, data, bandwidths, contrasts, norm_probes, include_inhibitory_neurons, **kwargs):
        cell_types = [0, 1] if include_inhibitory_neurons else [0]
        nested = gridify_tc_samples(data, num_contrasts=len(contrasts), num_bandwidths=len(bandwidths),
                                    num_cell_types=len(cell_types), num_probes=len(norm_probes))
        cond_values = [cell_types, norm_probes, contrasts, bandwidths]
        assert nested.shape == (len(data),) + tuple(map(len, cond_values))
        return cls(nested, cond_values, **kwargs)

    def __init__(self, nested, cond_values, e_ratio, seed=0):
        self.nested = np.asarray(nested)
        self.cond_values = cond_values = list(map(np.asarray, cond_values))
        self.e_ratio = e_ratio
        self.cell_types, self.norm_probes, self.contrasts, self.bandwidths = cond_values
        self.rng = as_randomstate(seed)
        assert tuple(self.cell_types) in [(0,), (0, 1)]

    def random_cells(self, num_models, probes_per_model):
        """Every (cell type, probe) pair at most once per model (cwgan.py:328-355)."""
        cellids = cartesian_product(np.arange(len(self.cell_types)), np.arange(len(self.norm_probes)), dtype=int).T
        if len(self.cell_types) == 2:
            probs = np.zeros(len(cellids))
            probs[:len(self.norm_probes)] = self.e_ratio
            probs[len(self.norm_probes):] = 1 - self.e_ratio
            probs /= probs.sum()
        else:
            probs = None
        if len(cellids) == 1:
            # one (cell type, probe) pair: `choice(1, 1, replace=False)` is permutation(1)[:1] = [0] and draws nothing
            # from the RandomState, so the per-model loop (5 us per model: 40 ms per critic step for the 8192 models
            # of an 8-GPU job, on every rank) can be skipped without changing the stream
            assert probes_per_model == 1
            ids = np.broadcast_to(cellids[0], (num_models, 1, 2))
        else:
            ids = np.asarray([cellids[self.rng.choice(len(cellids), probes_per_model, replace=False, p=probs)]
                              for _ in range(num_models)])
        ids_cell_type, ids_norm_probes = ids.transpose((2, 0, 1))
        return ids_cell_type, ids_norm_probes

    def select_minibatch(self, num_models, probes_per_model):
        """RandomState consumption order: samples, cells (per model), contrasts (cwgan.py:362-367)."""
        shape = (num_models, probes_per_model)
        ids_sample = self.rng.choice(len(self.nested), shape)
        ids_cell_type, ids_norm_probes = self.random_cells(*shape)
        ids_flat_contrast = self.rng.choice(len(self.contrasts), num_models)
        ids_contrast = np.broadcast_to(ids_flat_contrast.reshape((-1, 1)), shape)
        tc_md = self.nested[ids_sample, ids_cell_type, ids_norm_probes, ids_contrast]
        assert tc_md.shape == (num_models, probes_per_model, len(self.bandwidths))
        return ConditionalMinibatch(
            tc_md,
            [self.contrasts[ids_contrast], self.norm_probes[ids_norm_probes], self.cell_types[ids_cell_type]],
            self.bandwidths,
            self.contrasts[ids_flat_contrast])

    def random_minibatches(self, *args, **kwargs):
        while True:
            yield self.select_minibatch(*args, **kwargs)


class GradientAllReducer(object):
    """One flat-buffer all-reduce (mean over ranks) per update; identity without torch.distributed."""

    def __init__(self):
        import torch.distributed as dist
        self.dist = dist
        # (TCGAN_DIST_SINGLE_RANK=1: run the collectives in a one-rank group as well -- the mean over one rank is the
        # identity; bench.py uses it to put the all-reduce path through RCCL on a one-GPU box)
        self.on = dist.is_available() and dist.is_initialized() and (
            dist.get_world_size() > 1 or os.environ.get('TCGAN_DIST_SINGLE_RANK') == '1')
        self.world = dist.get_world_size() if self.on else 1
        self.rank = dist.get_rank() if self.on else 0
        # per-phase timing for multi-GPU diagnosis (bench.py --gpus N): device time between two events around every
        # collective, summed on demand; off by default (two event records per update)
        self.timed = False
        self._events = []
        self.calls = 0                         # collectives issued so far

    def collective_ms(self):
        """Device milliseconds spent in the all-reduces since the last call (synchronises)."""
        if not self._events:
            return 0.0
        torch.cuda.synchronize()
        total = sum(a.elapsed_time(b) for a, b in self._events)
        self._events = []
        return total

    def _sum(self, t):
        """One all-reduce (sum) of `t`, in stream order; timed between two events when `self.timed`."""
        self.calls += 1
        if self.timed and t.is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
            ev[1].record()
            self._events.append(ev)
        else:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        if self.world > 1:
            t /= self.world

    def mean_(self, *tensors):
        """In-place mean over ranks of several tensors through ONE collective.  A single contiguous fp32 tensor is reduced
        where it is; several go through a persistent flat fp32 buffer (one per total size: a step alternates between a
        few sizes, nothing is allocated per update)."""
        if not self.on:
            return
        if len(tensors) == 1 and tensors[0].dtype == torch.float32 and tensors[0].is_contiguous():
            self._sum(tensors[0])
            return
        n = sum(t.numel() for t in tensors)
        flats = self.__dict__.setdefault('_flats', {})
        key = (n, tensors[0].device)
        flat = flats.get(key)
        if flat is None:
            flat = flats[key] = torch.empty(n, device=tensors[0].device, dtype=torch.float32)
        off = 0
        for t in tensors:
            flat[off:off + t.numel()].copy_(t.reshape(-1))
            off += t.numel()
        self._sum(flat)
        off = 0
        for t in tensors:
            t.copy_(flat[off:off + t.numel()].reshape(t.shape))
            off += t.numel()


class ConditionalBPTTWassersteinGAN(object):
    """cwgan.py:410-552."""

    loss_type = 'WD'

    def __init__(self, gen, disc, gen_updaters, disc_updater, bandwidths, contrasts, norm_probes,
                 e_ratio, include_inhibitory_neurons, rate_penalty_threshold,
                 num_models, probes_per_model, critic_iters_init, critic_iters, lipschitz_cost,
                 disc_rate_penalty_bound, dynamics_cost, rate_cost, param_bounds, seed=0):
        self.gen = gen
        self.disc = disc
        self.gen_updaters = gen_updaters          # {'J': Updater, 'D': ..., 'S': ...}
        self.disc_updater = disc_updater
        self.bandwidths = np.asarray(bandwidths)
        self.contrasts = np.asarray(contrasts)
        self.norm_probes = np.asarray(norm_probes)
        self.e_ratio = e_ratio
        self.include_inhibitory_neurons = include_inhibitory_neurons
        self.rate_penalty_threshold = rate_penalty_threshold
        self.num_models = num_models
        self.probes_per_model = probes_per_model
        self.critic_iters_init = critic_iters_init
        self.critic_iters = critic_iters
        self.lipschitz_cost = lipschitz_cost
        self.disc_rate_penalty_bound = disc_rate_penalty_bound
        self.dynamics_cost = dynamics_cost
        self.rate_cost = rate_cost
        self.param_bounds = param_bounds          # {'J': (min, max), ...}  (wgan.py:244-251)
        self._init_loop_state(seed)
        assert self.probes_per_model < gen.num_neurons
        assert num_models % self.reducer.world == 0, 'num_models must be divisible by the number of ranks'

    def _init_loop_state(self, seed):
        """What the update loop keeps between steps (shared with the unconditional GAN of networks/wgan.py)."""
        gen = self.gen
        self.rng = as_randomstate(seed)
        self._predrawn = None          # host draws of the NEXT critic step, made early (see train_generator)
        self._next_ctx = None          # ... and that step itself, its forward queued behind the generator update (`_prequeue_next_disc`)
        self._gparams_host = {}        # host values the device copies of the generator parameters correspond to
        self._rng_before_predraw = None
        self._acc_carry = None         # data-parallel runs: this rank's accuracy of the last critic step, waiting for a collective
        self._arrived_with_gen = None
        self.reducer = GradientAllReducer()
        self._pnames = [name for name, _ in gen.get_all_params()]        # ['V',] 'J', 'D', 'S'
        self._gparams = {name: torch.zeros(int(np.size(value)), device='cuda', dtype=torch.float32)
                         for name, value in gen.get_all_params()}

    batchsize = property(lambda self: self.num_models * self.probes_per_model)
    num_sites = property(lambda self: self.gen.num_sites)
    num_neurons = property(lambda self: self.gen.num_neurons)
    NZ = property(lambda self: self.batchsize)
    discriminator = property(lambda self: self.disc)

    @property
    def sample_sites(self):
        return sample_sites_from_stim_space(self.norm_probes, self.num_sites)

    def get_gen_param(self):
        return [self.gen.J.copy(), self.gen.D.copy(), self.gen.S.copy()]

    def set_gen_param(self, **params):
        self.gen.set_params(params)

    def set_dataset(self, data, **kwargs):
        kwargs.setdefault('seed', self.rng)       # the sampler SHARES the GAN's RandomState (cwgan.py:452)
        self.sampler = RandomChoiceSampler.from_grid_data(
            data, bandwidths=self.bandwidths, contrasts=self.contrasts, norm_probes=self.norm_probes,
            e_ratio=self.e_ratio, include_inhibitory_neurons=self.include_inhibitory_neurons, **kwargs)
        self.dataset = self.sampler.random_minibatches(self.num_models, self.probes_per_model)

    def next_minibatch(self):
        return next(self.dataset)

    def prepare(self):
        """Nothing to compile."""

    # ---------------------------------------------------------------------------------------
    def _local(self, batch):
        return batch.shard(self.reducer.rank, self.reducer.world) if self.reducer.on else batch

    def _draw_noise(self, batch, keep_z=None):
        """Noise in the reference's stream order (ssn.py:434-439, 764-767: zs, then zs_in): the GLOBAL draw is consumed from
        the shared RandomState, this rank's rows are returned -- a data-parallel run consumes the RandomState exactly like a
        single-GPU run.  z itself is generated on the device (`ssn.device_rand`: only this rank's rows; the host never sees
        it) unless the generator was built with `z_host_draw`.  Empty in Philox mode (`z_device_seed`)."""
        if self.gen._zgen is not None:
            return {}
        rows = None
        if self.reducer.on:
            per = batch.num_models // self.reducer.world
            rows = (self.reducer.rank * per, (self.reducer.rank + 1) * per)
            if self.gen.z_host_draw and not getattr(self, '_warned_host_noise', False):
                # host draw under data parallelism: EVERY rank draws the global (num_models, 2N, 2N) fp64 tensor on the host
                # and keeps 1/world of it, so the host cost per step grows with the job.  Meant for equivalence tests.
                self._warned_host_noise = True
                logger.warning('host-side noise with %d ranks: every rank draws all %d models\' z on the host each step',
                               self.reducer.world, batch.num_models)
        # keep_z given: the forward follows at once with the current parameters, so W is formed in the draw's own launch
        return self.gen.gen_noise(self.rng, stimulator_bandwidths=np.empty((batch.num_models, 1)), rows=rows, keep_z=keep_z)

    def gen_forward(self, batch, noise=None, save=False, params_dev=None):
        local = self._local(batch)
        kw = local.gen_kwargs
        kw.update(noise or {})
        return self.gen.forward(rng=self.rng, save=save, model_rate_penalty_threshold=self.rate_penalty_threshold,
                                params_dev=params_dev, **kw), local

    # A critic step is split in two so that the NEXT generator forward (which does not depend on the critic update)
    # is already queued on the stream when the host blocks on this step's scalars: `_prepare_disc` draws from the
    # host RNG (same order as the reference: minibatch, eps, zs) and launches the forward; `_finish_disc` runs the
    # critic update and reads loss / accuracy / penalties back with ONE device-to-host copy.
    def _draw_disc(self, early=False):
        """The RNG draws of one critic step, in the reference's order (minibatch, eps, zs).  `early`: the draws `train_generator`
        makes ahead of time, BEFORE its parameter update -- zs is drawn as numbers only (W is formed from them at the forward,
        with the updated parameters); otherwise the forward follows at once and W is formed in the draw's own launch."""
        t0 = time.perf_counter()
        batch = self.next_minibatch()
        eps_full = self.rng.rand(batch.batchsize, 1)
        noise = self._draw_noise(batch, keep_z=None if early else False)
        self.host_draw_seconds = getattr(self, 'host_draw_seconds', 0.0) + (time.perf_counter() - t0)
        return batch, eps_full, noise

    def _prepare_disc(self, params_dev=None):
        drawn = self._predrawn
        if params_dev is None:
            # (the regular path uses the early draws up; a step prepared ahead from them is obsolete with that)
            self._predrawn = self._rng_before_predraw = self._next_ctx = None
        batch, eps_full, noise = drawn if drawn is not None else self._draw_disc()
        with self.gen_forward_watch:
            gen_out, local = self.gen_forward(batch, noise, params_dev=params_dev)
        return Namespace(batch=batch, eps_full=eps_full, gen_out=gen_out, local=local, pens64=self.gen.last_penalties,
                         gen_time=self.gen_forward_watch.times[-1])

    # The first critic step of the NEXT iteration needs nothing from the host but the updated generator parameters -- and those
    # are on the device the moment the optimizer launch is queued.  So its forward is queued right behind that launch, with W
    # formed from the device values (`ssn_build_w_devparams_f32`), BEFORE the host waits for the step's record: the GPU no
    # longer idles through the host's way from the record to the next launch (0.16 - 0.24 ms per iteration).  The draws were
    # made early anyway (`train_generator`); `_predrawn` stays set until the prepared step is taken up, so a checkpoint in
    # between still stores the RandomState from before them and a resumed run repeats them.
    def _prequeue_next_disc(self, st):
        if self._predrawn is None or not _PREQUEUE or self.gen.dtype != 'float32':
            return None
        noise = self._predrawn[2]
        if not self.gen.accepts_params_dev(noise.get('model_zs')):
            return None
        names, offs = self._pnames, st['offs']
        j = names.index('J')
        if names[j:j + 3] != ['J', 'D', 'S'] or offs[j + 3] - offs[j] != 12:
            return None
        pd = dict(JDS=st['flat'][offs[j]:offs[j] + 12])
        if 'V' in names:
            v = names.index('V')
            pd['V'] = st['flat'][offs[v]:offs[v + 1]]
        ctx = self._prepare_disc(params_dev=pd)
        ctx.prequeued_time = self.gen_forward_watch.times.pop()       # (booked on the iteration that uses the step)
        return ctx

    def _take_prequeued(self):
        """The critic step `_prequeue_next_disc` prepared, unless the generator's attributes were changed from outside since
        the record of the update was read (then it is prepared again from the same draws, with the attributes' values)."""
        ctx, self._next_ctx = self.__dict__.get('_next_ctx'), None
        if ctx is None:
            return None
        for name in self._pnames:
            cached = self._gparams_host.get(name)
            if cached is None or not np.array_equal(cached, np.asarray(getattr(self.gen, name))):
                return None
        self._predrawn = self._rng_before_predraw = None
        self.gen_forward_watch.times.append(ctx.prequeued_time)
        return ctx

    def _launch_disc(self, ctx):
        """Queue the critic update of a prepared step (no host wait); the four scalars of the step go to pinned
        host memory with an asynchronous copy followed by an event."""
        gen_out, local = ctx.gen_out, ctx.local
        xg = gen_out.prober_tuning_curve
        per = local.batchsize
        r0 = self.reducer.rank * per if self.reducer.on else 0
        if local.conditions is None:              # the unconditional GAN of networks/wgan.py: no condition columns
            (xd, eps), cd = to_device_packed([local.tuning_curves, ctx.eps_full[r0:r0 + per]], torch.float32), None
        else:
            xd, cd, eps = to_device_packed([local.tuning_curves, local.conditions, ctx.eps_full[r0:r0 + per]], torch.float32)
        ctx.skipped = False
        ctx.snapshot = None
        ctx.gated = False
        ctx.acc_deferred = False
        if not self.reducer.on and xg.dtype == torch.float32 and self.disc.has_step:
            # single process: the whole step is one library call (same kernels, same order); the skip rule of cwgan.py:493-498
            # is honoured by the optimizer kernel itself, which reads the rate penalty where the forward left it
            ctx.gated = self.disc_rate_penalty_bound > 0
            with self.disc_train_watch:
                xp, tail = self.disc.step(self.disc_updater, xg, xd, cd, eps, self.lipschitz_cost, pens64=ctx.pens64,
                                          rate_penalty_bound=self.disc_rate_penalty_bound if ctx.gated else None)
            ctx.xd, ctx.xg, ctx.xp, ctx.cd = xd, xg, xp, cd
        else:
            pens = ctx.pens64.to(torch.float32)      # [dynamics_penalty, rate_penalty] of this step's forward (averaged over ranks below)
            xp = self.disc.interpolate(eps, xd, xg)                               # cwgan.py:481
            ctx.xd, ctx.xg, ctx.xp, ctx.cd = xd, xg, xp, cd
            # cwgan.py:493-498 skips the critic update when the rate penalty of the batch exceeds `disc_rate_penalty_bound`.
            # Waiting for that value here would stall the queue at every critic step, so the update is made
            # speculatively and rolled back in `_read_disc` (which learns the value before the next update is queued)
            # in the rare case that the bound was exceeded.
            ctx.snapshot = self.disc_updater.snapshot(self.disc.params) if self.disc_rate_penalty_bound > 0 else None
            # ONE collective per update: gradients, loss statistics, the two penalties -- and, riding along, the
            # accuracy of the critic as the PREVIOUS step left it.  That number (cwgan.py:505-507) is a recorded diagnostic
            # of the updated critic, so it only exists after this step's collective has gone; instead of its own one-float
            # all-reduce it travels in the next flat buffer (the next critic step's, or the generator's), and the step's
            # record is handed to the driver once it has arrived (`_single_gen_step`).
            carry = self._acc_carry if self.reducer.on else None
            with self.disc_train_watch:
                stats = self.disc.loss_grad(xg, cd, xd, cd, xp, cd, self.lipschitz_cost)
                self.reducer.mean_(self.disc.grads, stats, pens, *([carry] if carry is not None else []))
                self.disc_updater(self.disc.params, self.disc.grads)
            acc = self.disc.accuracy_device(xg, cd, xd, cd)
            if self.reducer.on:
                self._acc_carry = acc            # this rank's share; averaged by the next collective
                ctx.acc_deferred = True
                arrived = carry if carry is not None else torch.full_like(acc, float('nan'))
                tail = torch.cat([pens, stats[3:4], acc, self.disc.param_sqnorms_device(), arrived])
            else:
                tail = torch.cat([pens, stats[3:4], acc, self.disc.param_sqnorms_device()])
        # `tail`: the four scalars of the step [dynamics penalty, rate penalty, loss, accuracy] + the per-tensor sums of squares
        # of the updated critic (disc_param_stats); pinned buffers and events come from a small ring (a step is read before the
        # ring comes round: at most two steps are in flight)
        ring = self.__dict__.setdefault('_host_ring', [])
        if not ring or ring[0][0].numel() != tail.numel():
            ring[:] = [(torch.empty(tail.numel(), dtype=torch.float32, pin_memory=True), torch.cuda.Event()) for _ in range(4)]
        self._host_ring_pos = (getattr(self, '_host_ring_pos', -1) + 1) % len(ring)
        ctx.host, ctx.event = ring[self._host_ring_pos]
        ctx.host.copy_(tail, non_blocking=True)
        ctx.event.record()
        ctx.disc_time = self.disc_train_watch.times[-1]

    def _read_disc(self, info, ctx):
        """Wait for the scalars of a launched step (only for its own kernels: later launches are not waited for)."""
        if ctx.event is not None:
            ctx.event.synchronize()
        host = ctx.host.numpy().copy()               # (the pinned buffer goes back to the ring)
        ctx.arrived_accuracy = None
        if ctx.acc_deferred:
            # last word: the job-wide accuracy of the PREVIOUS critic step, which travelled in this step's collective
            ctx.arrived_accuracy, host = float(host[-1]), host[:-1]
        if (ctx.snapshot is not None or ctx.gated) and np.float32(host[1]) > np.float32(self.disc_rate_penalty_bound):
            if ctx.gated:
                self.disc_updater.uncommit_step()                                # (the kernel made no update: same test, same value)
            else:
                self.disc_updater.restore(self.disc.params, ctx.snapshot)        # the skipped step of cwgan.py:493-498
            ctx.skipped = True
            host = np.array([host[0], host[1], np.nan, np.nan], dtype='float32')
            info.param_sqnorms = None
        else:
            info.param_sqnorms = host[4:].copy()     # of the critic as THIS step left it; handed to the recorder when the record is
        self.disc.cache_param_nnorms(info.param_sqnorms)
        ctx.snapshot = None
        info.gen_out = ctx.gen_out
        info.xd, info.xg, info.xp = ctx.xd, ctx.xg, ctx.xp
        info.cd = info.cg = info.cp = ctx.cd
        info.batch = ctx.batch
        info.gen_time = ctx.gen_time
        info.dynamics_penalty, info.rate_penalty = float(host[0]), float(host[1])
        info.disc_loss, info.accuracy = float(host[2]), float(host[3])
        info.accuracy_pending = bool(ctx.acc_deferred and not ctx.skipped)   # (host[3] is this rank's share until the mean arrives)
        info.disc_time = np.nan if ctx.skipped else ctx.disc_time
        return info

    def _flush_accuracy(self):
        """The job-wide mean of the accuracy still waiting for a collective, through one of its own (outside the loop's
        schedule: `train_discriminator` called by hand, the end of a run)."""
        carry, self._acc_carry = self._acc_carry, None
        if carry is None:
            return None
        self.reducer.mean_(carry)
        return float(carry[0])

    def _finish_disc(self, info, ctx):
        self._acc_carry = None
        self._launch_disc(ctx)
        info = self._read_disc(info, ctx)
        if info.accuracy_pending:
            info.accuracy, info.accuracy_pending = self._flush_accuracy(), False
        self._acc_carry = None
        return info

    def train_discriminator(self, info):
        return self._finish_disc(info, self._prepare_disc())

    def _mean_scalar(self, t):
        t = t.reshape(1).to(torch.float32).clone()
        self.reducer.mean_(t)
        return float(t[0])

    def _prepare_gen(self, batch):
        """Noise draw + generator forward with the trajectory kept (independent of the critic: may be queued before
        the last critic step of the iteration has finished)."""
        noise = self._draw_noise(batch, keep_z=True)
        with self.gen_train_watch:
            gen_out, local = self.gen_forward(batch, noise, save=True)
        return gen_out, local

    def train_generator(self, info, batch, prepared=None):
        gen_out, local = prepared if prepared is not None else self._prepare_gen(batch)
        # The host draws of the NEXT critic step (per-model `rng.choice` loops: ~5 ms for 1024 models) are made now,
        # while the trajectory-saving forward is still running on the GPU -- after the parameter update the GPU would
        # sit idle for them.  Same RandomState order: nothing else draws between here and that step.  A checkpoint
        # taken in between stores the RandomState from BEFORE these draws (state_dict), so a resumed run repeats them.
        if self.critic_iters > 0 and self._predrawn is None:
            self._rng_before_predraw = self.rng.get_state()
            self._predrawn = self._draw_disc(early=True)
        fused = self._gen_tail_fused()
        with self.gen_train_watch:
            cd = None if local.conditions is None else to_device(np.ascontiguousarray(local.conditions), torch.float32)
            xg = gen_out.prober_tuning_curve.to(torch.float32)
            nb = xg.shape[0]
            gx, dmean = self.disc.input_grad(xg, cd, scale=-1.0 / nb)           # d(-mean D)/d tuning curve
            if fused is not None:
                self._gen_step_now = getattr(info, 'gen_step', '?')
                host = self._train_generator_tail(fused, gx, dmean)
                info.gen_loss = float(host[-1])
                if not np.isfinite(host).all():
                    self._report_poisoned(info, gen_out)
        if fused is not None:
            info.gen_forward_time = self.gen_forward_watch.sum()
            info.gen_train_time = self.gen_train_watch.sum()
            info.gen_time = info.gen_train_time + info.gen_forward_time
            info.disc_time = self.disc_train_watch.sum()
            return info
        with self.gen_train_watch:
            gdict = self.gen.backward(gx, self.dynamics_cost, self.rate_cost, as_tensor=True)
            loss = (-dmean.to(torch.float64) + self.dynamics_cost * gen_out.model_dynamics_penalty
                    + self.rate_cost * gen_out.model_rate_penalty).reshape(1).to(torch.float32)
            grads = torch.cat([gdict[name].reshape(-1) for name in self._pnames]).to(torch.float32)
            # (the accuracy of the last critic step rides in the generator's collective: see `_launch_disc`)
            carry, self._acc_carry = (self._acc_carry if self.reducer.on else None), None
            self.reducer.mean_(grads, loss, *([carry] if carry is not None else []))
            # Everything up to here is queued without a host wait; the device copies of the parameters are the
            # working values (re-uploaded only when somebody changed the generator's attributes from outside), and ONE
            # device-to-host copy at the end returns the new parameter values together with the loss.
            off = 0
            for name in self._pnames:                                            # wgan.py:218-260
                value = np.asarray(getattr(self.gen, name))
                p = self._gparams[name]
                cached = self._gparams_host.get(name)
                if cached is None or cached.shape != value.shape or not np.array_equal(cached, value):
                    p.copy_(to_device(np.ascontiguousarray(value, dtype='float32').ravel()))
                self.gen_updaters[name](p, grads[off:off + p.numel()], clip=self.param_bounds[name])
                off += p.numel()
            host = torch.cat([self._gparams[name] for name in self._pnames] + [loss]
                             + ([carry.to(torch.float32)] if carry is not None else [])).cpu().numpy()
            self._arrived_with_gen = None
            if carry is not None:
                self._arrived_with_gen, host = float(host[-1]), host[:-1]
            off = 0
            for name in self._pnames:
                shape = np.shape(getattr(self.gen, name))
                n = int(np.prod(shape, dtype=int))
                new = host[off:off + n].astype('float64').reshape(shape)
                setattr(self.gen, name, new)
                self._gparams_host[name] = new.copy()
                off += n
            info.gen_loss = float(host[-1])
            if not np.isfinite(host).all():
                self._report_poisoned(info, gen_out)
        info.gen_forward_time = self.gen_forward_watch.sum()
        info.gen_train_time = self.gen_train_watch.sum()
        info.gen_time = info.gen_train_time + info.gen_forward_time
        info.disc_time = self.disc_train_watch.sum()
        return info

    def _report_poisoned(self, info, gen_out):
        """(the drivers' NaN guards end the run; say why when it was the fp16 adjoint's range and not the model)"""
        bad = self.gen.poisoned_draws()
        if bad:
            logger.warning('generator step %s: %d of %d draws have an adjoint that grew by more than 2^8 within one '
                           'Euler step -- beyond the lagged scale of the fp16-split sweep, so their gradient is NaN; '
                           '--gen-kernel mfma-fp32 has no such limit', getattr(info, 'gen_step', '?'), bad,
                           gen_out.prober_tuning_curve.shape[0] // max(getattr(self, 'probes_per_model', 1), 1))

    # -- the end of a generator step in two library calls (`ssn_gen_grads_f32`, `ssn_gen_apply_f32`) --------------------------
    def _gen_tail_fused(self):
        """The flat device state of the generator's update (parameters, optimizer moments, clip bounds in `_pnames` order), or
        None when the step has to go parameter by parameter: a generator in float64, or updaters that are not in step with
        each other (same rule, hyper-parameters and step count: `make_gan` builds them that way)."""
        ups = [self.gen_updaters[name] for name in self._pnames]
        sig = {(u.kind, u.learning_rate, tuple(sorted(u.cfg.items())), tuple(u.reg), u.step) for u in ups}
        if len(sig) != 1 or self.gen.dtype != 'float32':
            return None
        st = self.__dict__.get('_gflat')
        if st is None:
            sizes = [int(self._gparams[name].numel()) for name in self._pnames]
            offs = np.concatenate([[0], np.cumsum(sizes)]).astype(int)
            n = int(offs[-1])
            flat = torch.cat([self._gparams[name] for name in self._pnames])
            lo = np.concatenate([np.broadcast_to(np.asarray(self.param_bounds[name][0], dtype='float32').ravel(), (k,))
                                 for name, k in zip(self._pnames, sizes)])
            hi = np.concatenate([np.broadcast_to(np.asarray(self.param_bounds[name][1], dtype='float32').ravel(), (k,))
                                 for name, k in zip(self._pnames, sizes)])
            st = self._gflat = dict(n=n, offs=offs, flat=flat, m=torch.zeros_like(flat), v=torch.zeros_like(flat),
                                    lo=to_device(lo), hi=to_device(hi), record=torch.empty(n + 1, device='cuda', dtype=torch.float32))
            for i, name in enumerate(self._pnames):          # the per-name device copies become views of the flat vector
                self._gparams[name] = flat[offs[i]:offs[i + 1]]
        for i, name in enumerate(self._pnames):               # ... and so do the updaters' moments (checkpoints read them there)
            u, a, b = self.gen_updaters[name], st['offs'][i], st['offs'][i + 1]
            mine = (st['m'][a:b], st['v'][a:b])
            if u._state is None:
                mine[0].zero_(); mine[1].zero_()
                u._state = mine
            elif u._state[0].data_ptr() != mine[0].data_ptr():           # (restored from a checkpoint: take the values over)
                mine[0].copy_(u._state[0].reshape(-1)); mine[1].copy_(u._state[1].reshape(-1))
                u._state = mine
        return st

    def _train_generator_tail(self, st, gx, dmean):
        """Adjoint sweep, dL/dW, chain rule; then gradient vector + loss in one launch, the job's all-reduce, and the update of
        all parameters with their bounds + the record in one more.  Returns the record on the host: new values, loss."""
        import ctypes
        pieces = self.gen.backward(gx, self.dynamics_cost, self.rate_cost, raw=True)
        from .. import genops
        gl = genops.gen_grads(pieces['parts'], dmean, self.gen.last_penalties, self.dynamics_cost, self.rate_cost,
                              nv=pieces['nv'], g_ext=pieces.get('g_ext'), ext_base=pieces.get('ext_base'), zin=pieces.get('zin'))
        # (the accuracy of the last critic step rides in the generator's collective: see `_launch_disc`)
        carry, self._acc_carry = (self._acc_carry if self.reducer.on else None), None
        self.reducer.mean_(gl, *([carry] if carry is not None else []))
        for name in self._pnames:                                                # wgan.py:218-260
            value = np.asarray(getattr(self.gen, name))
            cached = self._gparams_host.get(name)
            if cached is None or cached.shape != value.shape or not np.array_equal(cached, value):
                self._gparams[name].copy_(to_device(np.ascontiguousarray(value, dtype='float32').ravel()))
        u0 = self.gen_updaters[self._pnames[0]]

        def apply(gl, gate):
            _, _, opt = u0.begin_step(st['flat'][:self._gparams[self._pnames[0]].numel()])
            opt.clip = 0
            opt.reserved = int(gate)          # bit 0: no update at all when a gradient element is not finite (record[n] = NaN)
            clib.check(clib.libssnode.ssn_gen_apply_f32(st['flat'].data_ptr(), gl.data_ptr(), st['m'].data_ptr(), st['v'].data_ptr(),
                                                        st['n'], ctypes.byref(opt), st['lo'].data_ptr(), st['hi'].data_ptr(),
                                                        st['record'].data_ptr(), clib.stream_ptr()), 'ssn_gen_apply_f32')
            return opt

        opt = apply(gl, gate=True)
        pens = self.gen.last_penalties
        tail = [st['record']] + ([carry.to(torch.float32)] if carry is not None else [])
        rec = torch.cat(tail) if len(tail) > 1 else tail[0]
        # the record's copy is queued BEFORE the next step's forward and waited for by its own event: a plain .cpu() behind the
        # forward would wait for the forward too
        pin = self.__dict__.get('_gen_record_pin')
        if pin is None or pin[0].numel() != rec.numel():
            pin = self._gen_record_pin = (torch.empty(rec.numel(), dtype=torch.float32, pin_memory=True),
                                          torch.cuda.Event())
        pin[0].copy_(rec, non_blocking=True)
        pin[1].record()
        self._next_ctx = self._prequeue_next_disc(st)
        pin[1].synchronize()
        host = pin[0].numpy().copy()
        arrived = None
        if carry is not None:
            arrived, host = float(host[-1]), host[:-1]
        if np.isnan(host[-1]) and np.isfinite(host[:-1]).all() and self.gen.__dict__.get('_retry') is not None:
            # The update was withheld: the summed gradient is not finite.  With the fp16-split adjoint that is what a draw whose
            # adjoint grows by more than 2^8 within one Euler step gives (unstable dynamics: NaN by construction, never a
            # clamped value); the reference's fp32 arithmetic carries such a draw's large finite gradient.  So does the fp32
            # sweep: the refused draws are made again on it -- forward, adjoint, dL/dW; all draws when more than half were
            # refused -- and THAT gradient is applied, whatever it is.
            bad = self.gen.poisoned_draws()
            logger.warning('generator step %s: %d draws have an adjoint that grew by more than 2^8 within one Euler step, beyond '
                           'the lagged scale of the fp16-split sweep; they are recomputed on the fp32 kernels (gen_kernel '
                           'mfma-fp32 runs every step there)', getattr(self, '_gen_step_now', '?'), bad)
            self._next_ctx = None          # (prepared with the parameters the withheld update left: prepared again later)
            idx = self.gen.poisoned_draw_indices()
            nb = int(pieces['parts'].shape[0])
            if _SUBSET_RETRY and (0 < idx.numel() <= nb // 2 or (self.reducer.on and idx.numel() == 0)):
                # only the refused draws go through the fp32 kernels (a job-wide NaN may come from another rank's draws: then
                # this rank has nothing to redo); every other draw keeps the gradient pieces it has
                parts, g_ext = pieces['parts'], pieces.get('g_ext')
                if idx.numel():
                    sub = self.gen.backward(gx, self.dynamics_cost, self.rate_cost, raw=True, exact=True, subset=idx)
                    parts = parts.clone()
                    parts[idx] = sub['parts']
                    if g_ext is not None:
                        g_ext = g_ext.clone()
                        g_ext[idx] = sub['g_ext']
                pieces = dict(pieces, parts=parts, g_ext=g_ext)
            else:
                pieces = self.gen.backward(gx, self.dynamics_cost, self.rate_cost, raw=True, exact=True)
            gl = genops.gen_grads(pieces['parts'], dmean, pens, self.dynamics_cost, self.rate_cost,
                                  nv=pieces['nv'], g_ext=pieces.get('g_ext'), ext_base=pieces.get('ext_base'), zin=pieces.get('zin'))
            self.reducer.mean_(gl)
            opt = apply(gl, gate=False)
            host = st['record'].cpu().numpy()
        for name in self._pnames:
            self.gen_updaters[name].commit_step(opt)
        self._arrived_with_gen = arrived
        off = 0
        for name in self._pnames:
            shape = np.shape(getattr(self.gen, name))
            n = int(np.prod(shape, dtype=int))
            new = host[off:off + n].astype('float64').reshape(shape)
            setattr(self.gen, name, new)
            self._gparams_host[name] = new.copy()
            off += n
        return host

    def _single_gen_step(self, gen_step, critic_iters):
        self.gen_forward_watch = StopWatch()
        self.gen_train_watch = StopWatch()
        self.disc_train_watch = StopWatch()
        ctx = (self._take_prequeued() or self._prepare_disc()) if critic_iters > 0 else None
        prepared_gen = None
        self._acc_carry = None
        held = None            # data-parallel runs: the record of the step whose job-wide accuracy is still on its way
        for disc_step in range(critic_iters):
            last = disc_step + 1 == critic_iters
            # queue this step's critic update, THEN do the host work of the next step and queue its forward, and only
            # then wait for this step's scalars: the GPU always has the next forward queued (host RNG order unchanged)
            self._launch_disc(ctx)
            nxt = None if last else self._prepare_disc()
            if last:
                prepared_gen = self._prepare_gen(ctx.batch)    # reuses the LAST critic batch's conditions (cwgan.py:535-539)
            info = Namespace(is_discriminator=True, gen_step=gen_step, disc_step=disc_step)
            info = self._read_disc(info, ctx)
            if held is not None:
                if held.accuracy_pending:
                    held.accuracy, held.accuracy_pending = ctx.arrived_accuracy, False
                # (a held record is yielded after the NEXT step was read: the recorder must see its own norms, not that step's)
                self.disc.cache_param_nnorms(held.param_sqnorms)
                yield held
                held = None
            if info.accuracy_pending:
                held = info        # same order of records, one collective later
            else:
                if ctx.acc_deferred:
                    self._acc_carry = None      # (a skipped step has no accuracy: nothing to carry)
                self.disc.cache_param_nnorms(info.param_sqnorms)
                yield info
            ctx = nxt
        disc_info = info
        batch = info.batch
        info = Namespace(is_discriminator=False, gen_step=gen_step)
        info = self.train_generator(info, batch, prepared_gen)
        if held is not None:
            held.accuracy, held.accuracy_pending = self._arrived_with_gen, False
        logger.debug('[Loss] Acc: %-9.3g D: %-9.3g G: %-9.3g [Time] Fwd: %.3g D: %.3g G: %.3g',
                     disc_info.accuracy, disc_info.disc_loss, info.gen_loss, self.gen_forward_watch.mean(),
                     self.disc_train_watch.mean(), self.gen_train_watch.mean())
        if held is not None:
            self.disc.cache_param_nnorms(held.param_sqnorms)
            yield held
        yield info

    def learning(self, start_step=0):
        """wgan.py:439-444: `critic_iters_init` critic steps before generator step 0, then `critic_iters`.
        `start_step` > 0 continues a run restored with `load_state_dict` (no second initial critic phase)."""
        import itertools
        if start_step == 0:
            for info in self._single_gen_step(0, self.critic_iters_init):
                yield info
        for gen_step in itertools.count(max(start_step, 1)):
            for info in self._single_gen_step(gen_step, self.critic_iters):
                yield info

    # -- checkpoints (not in the reference: its runs cannot be resumed) -------------------------------------
    def state_dict(self):
        """Everything a bit-identical continuation needs: generator and critic parameters, the optimizer states,
        the host RandomState (shared with the minibatch sampler) and the device noise generator."""
        kind, keys, pos, has_gauss, cached = (self._rng_before_predraw if self._predrawn is not None
                                              else self.rng.get_state())
        d = dict(gen={name: np.array(value) for name, value in self.gen.get_all_params()},
                 disc=self.disc.get_flat(), disc_updater=self.disc_updater.state_dict(),
                 gen_updaters={name: self.gen_updaters[name].state_dict() for name in self._pnames},
                 rng=dict(kind=kind, keys=keys, pos=pos, has_gauss=has_gauss, cached=cached))
        if self.gen._zgen is not None:
            d['zgen'] = self.gen._zgen.get_state()
        # explicit state that changes what a continuation computes: the kernel family (operand precision) and the job size
        d['gen_kernel'] = self.gen.gen_kernel
        d['world'] = self.reducer.world
        return d

    def load_state_dict(self, d):
        self._predrawn = self._rng_before_predraw = self._next_ctx = None
        self.gen.set_params(d['gen'])
        self.disc.set_flat(np.asarray(d['disc']))
        self.disc_updater.load_state_dict(d['disc_updater'])
        for name in self._pnames:
            self.gen_updaters[name].load_state_dict(d['gen_updaters'][name])
        r = d['rng']
        self.rng.set_state((str(r['kind']), np.asarray(r['keys'], dtype='uint32'), int(r['pos']), int(r['has_gauss']),
                            float(r['cached'])))
        if self.gen._zgen is not None and 'zgen' in d:
            z = d['zgen']
            if not (isinstance(z, dict) and 'seed' in z and 'position' in z):
                # format 1 of round 1 kept a torch.Generator state (uint8 array) here; the stream is a different one now
                raise KnownError('this checkpoint holds the device-noise state of an earlier format (a torch.Generator state '
                                 'array, not a Philox seed/position pair); it cannot be continued with --z-device-seed',
                                 exit_code=5)
            self.gen._zgen.set_state(z)
        for key, mine in (('gen_kernel', self.gen.gen_kernel), ('world', self.reducer.world)):
            if key in d and d[key] != mine:
                logger.warning('checkpoint was written with %s = %r, this run has %r: the continuation will not reproduce the '
                               'uninterrupted run bit for bit%s', key, d[key], mine,
                               ' (each rank draws other rows of the noise stream)' if key == 'world' else '')

    CHECKPOINT_VERSION = 2          # 2: Philox (seed, position) device-noise state, gen_kernel and world recorded

    def save_checkpoint(self, path, gen_step):
        import pickle
        tmp = path + '.tmp'
        with open(tmp, 'wb') as f:
            pickle.dump(dict(version=self.CHECKPOINT_VERSION, gen_step=int(gen_step), state=self.state_dict()), f, protocol=4)
        os.replace(tmp, path)

    def load_checkpoint(self, path):
        """Restore a checkpoint (after `set_dataset`); returns the generator step to continue WITH."""
        import pickle
        with open(path, 'rb') as f:
            ck = pickle.load(f)
        if not isinstance(ck, dict) or ck.get('version') not in (1, self.CHECKPOINT_VERSION) or 'state' not in ck:
            raise KnownError('{} is not a checkpoint this version can read (format version {!r}; known: 1, {})'
                             .format(path, ck.get('version') if isinstance(ck, dict) else None, self.CHECKPOINT_VERSION),
                             exit_code=5)
        self.load_state_dict(ck['state'])
        return ck['gen_step'] + 1


def _v_bounds(v_min, v_max, ssn_type):
    """Clip bounds of the input-variability parameter V (wgan.py:244-251, 263-283): one (V_E, V_I) pair each for
    'heteroin' -- per element, like numpy's broadcasting clip in the reference -- a scalar each for 'deg-heteroin'."""
    if ssn_type == 'heteroin':
        return (np.broadcast_to(np.asarray(v_min, dtype='float64'), (2,)).copy(),
                np.broadcast_to(np.asarray(v_max, dtype='float64'), (2,)).copy())
    return float(np.min(v_min)), float(np.max(v_max))


def make_gan(config):
    """make_gan(config: dict) -> (GAN, dict): build the GAN and return the unconsumed part of `config`
    (cwgan.py:555-614).  ``config['gen']`` / ``config['disc']`` hold the trainer options with the reference's
    names (learning_rate, update_name, dynamics_cost, rate_cost, J_min..S_max, rate_penalty_threshold;
    layers, normalization, nonlinearity, rate_penalty_bound, ...)."""
    kwargs = dict(DEFAULT_PARAMS, **config)
    gen_cfg = dict(DEFAULT_PARAMS['gen'], **config.get('gen', {}))
    disc_cfg = dict(DEFAULT_PARAMS['disc'], **config.get('disc', {}))
    kwargs.pop('gen', None)
    kwargs.pop('disc', None)
    take = kwargs.pop

    num_models = take('num_models')
    probes_per_model = take('probes_per_model')
    bandwidths = take('bandwidths')
    contrasts = take('contrasts')
    num_sites = take('num_sites')
    ssn_type = take('ssn_type', 'default')
    ssn_impl = take('ssn_impl', 'default')
    if ssn_impl not in ('default', 'mapclone'):     # 'mapclone' is the same arithmetic organised differently (ssn.py:25-30)
        raise ValueError('Unknown ssn_impl: {}'.format(ssn_impl))
    if 'V0' in kwargs:                              # cwgan.py:570-571
        kwargs['V'] = kwargs.pop('V0')
    V = kwargs.pop('V', 0)
    dist_in = kwargs.pop('dist_in', 'bernoulli')
    for key in ('V_min', 'V_max'):
        kwargs.pop(key, None)
    reducer = GradientAllReducer()
    local_models = num_models // reducer.world
    # (the command line's --gen-kernel arrives as gen['kernel']: run scripts group their gen_* options; configs written
    # by hand may also say gen_kernel at the top level)
    gen_kernel = gen_cfg.pop('kernel', None) or take('gen_kernel', 'auto')
    kwargs.pop('gen_kernel', None)
    gen = TuningCurveGenerator(
        num_sites=num_sites, num_tcdom=len(bandwidths), smoothness=take('smoothness'),
        J=take('J0'), D=take('D0'), S=take('S0'), k=take('k'), n=take('n'),
        tau_E=take('tau_E'), tau_I=take('tau_I'), dt=take('dt'), io_type=take('io_type'),
        seqlen=take('seqlen'), skip_steps=take('skip_steps'),
        batchsize=local_models * probes_per_model,
        include_rate_penalty=take('include_rate_penalty', True),
        include_time_avg=take('include_time_avg', False),
        unroll_scan=take('unroll_scan', False),
        dtype=take('gen_dtype', 'float32'),
        z_device_seed=take('z_device_seed', None), shard=(reducer.rank, reducer.world),
        ssn_type=ssn_type, V=V, dist_in=dist_in, gen_kernel=gen_kernel, z_host_draw=take('z_host_draw', False))
    rate_penalty_threshold = gen_cfg.pop('rate_penalty_threshold')
    disc_rate_penalty_bound = disc_cfg.pop('rate_penalty_bound')
    seed = take('seed', 0)
    disc = Critic(nx=len(bandwidths), layers=disc_cfg.pop('layers', []),
                  normalization=disc_cfg.pop('normalization', 'none'),
                  nonlinearity=disc_cfg.pop('nonlinearity', 'rectify'),
                  hide_cell_type=take('hide_cell_type'),
                  precision=disc_cfg.pop('precision', 'fp32'),
                  seed=disc_cfg.pop('init_seed', seed), net_options=disc_cfg.pop('net_options', None))

    def updater_from(cfg):
        return Updater(**{k: cfg.pop(k) for k in ('learning_rate', 'update_name', 'update_config', 'reg_l2_penalty',
                                                   'reg_l2_decay', 'reg_l1_penalty', 'reg_l1_decay') if k in cfg})

    dynamics_cost = gen_cfg.pop('dynamics_cost', 1.0)
    rate_cost = gen_cfg.pop('rate_cost')
    bounds = {name: (gen_cfg.pop(name + '_min', 1e-3), gen_cfg.pop(name + '_max', 10.0)) for name in 'JDS'}
    bounds['V'] = _v_bounds(gen_cfg.pop('V_min', 0), gen_cfg.pop('V_max', 1), ssn_type)   # wgan.py:263-283
    gen_upd_cfg = {k: gen_cfg.pop(k) for k in list(gen_cfg) if k in ('learning_rate', 'update_name', 'update_config',
                                                                     'reg_l2_penalty', 'reg_l2_decay',
                                                                     'reg_l1_penalty', 'reg_l1_decay')}
    gen_updaters = {name: Updater(**gen_upd_cfg) for name in 'VJDS'}
    disc_updater = updater_from(disc_cfg)
    if gen_cfg or disc_cfg:
        raise ValueError('Unknown trainer options: gen={} disc={}'.format(sorted(gen_cfg), sorted(disc_cfg)))
    gan = ConditionalBPTTWassersteinGAN(
        gen, disc, gen_updaters, disc_updater, bandwidths, contrasts,
        norm_probes=take('norm_probes'), e_ratio=take('e_ratio'),
        include_inhibitory_neurons=take('include_inhibitory_neurons'),
        rate_penalty_threshold=rate_penalty_threshold,
        num_models=num_models, probes_per_model=probes_per_model,
        critic_iters_init=take('critic_iters_init'), critic_iters=take('critic_iters'),
        lipschitz_cost=take('lipschitz_cost'), disc_rate_penalty_bound=disc_rate_penalty_bound,
        dynamics_cost=dynamics_cost, rate_cost=rate_cost, param_bounds=bounds, seed=seed)
    return gan, kwargs
