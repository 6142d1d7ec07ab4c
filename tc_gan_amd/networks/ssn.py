"""Tuning-curve generator based on the SSN, on the GPU.

Host-side mirror of ``tc_gan/networks/ssn.py`` for the default (``ssn_type='default'``,
``ssn_impl='default'``) generator: `BandwidthContrastStimulator` -> `EulerSSNModel` ->
prober, bundled by `TuningCurveGenerator` / `ConditionalTuningCurveGenerator`
(networks/cwgan.py:107-120).  The reference compiles a Theano graph
(``forward_impl``, ssn.py:906-914); here `forward` launches the HIP kernels of
``csrc/ssn_aux.hip`` (stimulus, W from z) and ``csrc/ssn_gen.hip`` (Euler recurrence with
fused reductions), and `forward_backward` adds the BPTT adjoint sweep.

Input and output names follow the reference (``stimulator_bandwidths``, ``model_zs``,
``prober_model_ids`` ...; outputs ``model_dynamics_penalty``, ``model_rate_penalty``,
[``model_time_avg``,] ``prober_tuning_curve``).
"""
import collections
import ctypes

import numpy as np
import torch

from .. import clib, genops
from ..gradient_expressions.utils import sample_sites_from_stim_space_impl
from ..stimuli import stimulus_batch
from ..utils import to_device
from ..weight_gen import generate_weight_batch

ssn_impl_choices = ('default',)
ssn_type_choices = ('default', 'heteroin', 'deg-heteroin')
dist_in_choices = ('bernoulli', 'uniform')


def neu_array(pop_array, num_sites, pops=None):
    """ssn.py:60-83.

    >>> neu_array([0, 1], 2)
    array([0, 0, 1, 1])
    """
    pop_array = np.asarray(pop_array)
    return np.concatenate([np.tile(pop_array[i], num_sites) for i in range(pop_array.shape[0])])


def concat_flat(arrays):
    flat = []
    for a in arrays:
        flat.extend(np.asarray(a).flat)
    return flat


def make_flat_param_names(named_params):
    """ssn.py:105-123: ('J_EE', 'J_EI', 'J_IE', 'J_II', 'D_EE', ...)."""
    names = []
    for name, value in named_params:
        value = np.asarray(value)
        if value.ndim == 0:
            names.append(name)
        elif value.ndim == 1:
            names.extend([name + '_E', name + '_I'])
        elif value.ndim == 2:
            names.extend([name + '_' + pq for pq in ('EE', 'EI', 'IE', 'II')])
        else:
            raise ValueError('Only ndim<=2 is supported.')
    return tuple(names)


def _is_mt19937(rng):
    # (the bit generator's type, not get_state()[0]: that copies the 624-word key at every call)
    return isinstance(rng, np.random.RandomState) and isinstance(getattr(rng, '_bit_generator', None), np.random.MT19937)


TAIL_KINDS = {'bernoulli': 1, 'uniform': 2}


def _ticket_finisher(ticket, key, kind, has_gauss, cached):
    """What `DeviceContinuedRandomState` runs when its state is next needed: the state behind a draw begun on the device."""
    def finish(r):
        cpos = ctypes.c_int(0)
        clib.check(clib.libssnode.ssn_mt19937_random_sample_finish(ticket, key.ctypes.data, ctypes.byref(cpos)),
                   'ssn_mt19937_random_sample_finish')
        np.random.RandomState.set_state(r, (kind, key, cpos.value, has_gauss, cached))
    return finish


def device_rand(rng, shape, tdtype, rows=None, tail=None):
    """``rng.rand(*shape)`` of a ``numpy.random.RandomState`` -- the reference's ``zs = rng.rand(batchsize, 2N, 2N)``
    (ssn.py:434-439) -- generated on the device, bit for bit (`ssn_mt19937_random_sample_*`), as a tensor of `tdtype`
    (float32: each double rounded to nearest, like the reference's downcast to floatX).  `rng` is left in the state numpy
    would have left it in, so the host draws that follow (minibatch `choice`, `eps`) continue the reference's stream.
    `rows = (lo, hi)`: only rows lo..hi-1 of the draw are generated (a rank's share of the global draw); the state still
    advances by the whole draw.
    `tail = (dist_in, width)` (float32, a `DeviceContinuedRandomState`): the draw the heterogeneous-input models make right
    behind it, ``zs_in = rng.choice(2, (shape[0], width)) * 2 - 1`` ('bernoulli') or ``rng.rand(shape[0], width) * 2 - 1``
    ('uniform'; ssn.py:710-720), in the same call -- the host does not fetch the state in between; returns (z, zs_in)."""
    kind, key, pos, has_gauss, cached = rng.get_state()
    if kind != 'MT19937':
        raise ValueError('device_rand continues numpy RandomState (MT19937) streams only, got {!r}'.format(kind))
    clib.require_gpu()
    shape = tuple(int(n) for n in shape)
    key = np.ascontiguousarray(key, dtype=np.uint32)
    per_row = int(np.prod(shape[1:], dtype=np.int64))
    lo, hi = (0, shape[0]) if rows is None else (int(rows[0]), int(rows[1]))
    out = torch.empty((hi - lo,) + shape[1:], device='cuda', dtype=tdtype)
    f32 = tdtype == torch.float32
    if tail is not None:
        assert f32 and hasattr(rng, '_defer')
        width = int(tail[1])
        zin = torch.empty((hi - lo, width), device='cuda', dtype=torch.float32)
        ticket = ctypes.c_int(-1)
        clib.check(clib.libssnode.ssn_mt19937_random_sample_tail_begin_f32(
            key.ctypes.data, int(pos), shape[0] * per_row, lo * per_row, (hi - lo) * per_row, out.data_ptr(), TAIL_KINDS[tail[0]],
            shape[0] * width, lo * width, (hi - lo) * width, zin.data_ptr(), clib.stream_ptr(), ctypes.byref(ticket)),
            'ssn_mt19937_random_sample_tail_begin_f32')
        rng._defer(_ticket_finisher(ticket.value, key, kind, has_gauss, cached))
        return out, zin
    if hasattr(rng, '_defer'):
        # the state after the draw is fetched when `rng` is next used (utils.DeviceContinuedRandomState): the caller goes on
        # queuing its launches meanwhile
        ticket = ctypes.c_int(-1)
        fn = clib.libssnode.ssn_mt19937_random_sample_begin_f32 if f32 else clib.libssnode.ssn_mt19937_random_sample_begin_f64
        clib.check(fn(key.ctypes.data, int(pos), shape[0] * per_row, lo * per_row, (hi - lo) * per_row, out.data_ptr(),
                      clib.stream_ptr(), ctypes.byref(ticket)), 'ssn_mt19937_random_sample_begin')

        rng._defer(_ticket_finisher(ticket.value, key, kind, has_gauss, cached))
        return out
    cpos = ctypes.c_int(int(pos))
    fn = clib.libssnode.ssn_mt19937_random_sample_f32 if f32 else clib.libssnode.ssn_mt19937_random_sample_f64
    clib.check(fn(key.ctypes.data, ctypes.byref(cpos), shape[0] * per_row, lo * per_row, (hi - lo) * per_row,
                  out.data_ptr(), clib.stream_ptr()), 'ssn_mt19937_random_sample')
    rng.set_state((kind, key, cpos.value, has_gauss, cached))
    return out


# (A/B switch: TCGAN_MT_FUSE_W=0 keeps the draw and the W build in two launches)
_FUSE_W = __import__('os').environ.get('TCGAN_MT_FUSE_W', '1') != '0'
# (A/B switch: TCGAN_STIM_HETERO=0 forms 1 + v z_in of the heterogeneous-input models with torch operations in front of the
# stimulus launch, as until round 5, instead of inside it -- same bits)
_STIM_HETERO = __import__('os').environ.get('TCGAN_STIM_HETERO', '1') != '0'
# (A/B switch: TCGAN_MT_TAIL=0 draws zs_in of the heterogeneous-input models on the host, after fetching the state behind zs)
_TAIL = __import__('os').environ.get('TCGAN_MT_TAIL', '1') != '0'


class MTWeights(object):
    """W = make_W_with_x(z) of a draw that was made on the device from the caller's RandomState (`device_rand_weights`), with
    the draw itself (`z`) when it was asked for: what `TuningCurveGenerator._device_inputs` takes in place of z."""

    def __init__(self, z, W):
        self.z, self.W = z, W


def device_rand_weights(rng, num_models, N, J, D, S, rows=None, keep_z=True, tail=None):
    """`device_rand(rng, (num_models, 2N, 2N), float32, rows)` and `generate_weight_batch` of it in ONE launch
    (`ssn_build_w_mt19937_begin_f32`: W is formed where the numbers are, z goes to memory only when `keep_z`): same numbers,
    same bits, `rng` left as numpy would leave it.  Returns `MTWeights` -- with `tail` = 'bernoulli' / 'uniform' the pair
    (MTWeights, zs_in[rows][2N]): the heterogeneous-input draw behind zs in the same call (see `device_rand`)."""
    kind, key, pos, has_gauss, cached = rng.get_state()
    if kind != 'MT19937':
        raise ValueError('device_rand_weights continues numpy RandomState (MT19937) streams only, got {!r}'.format(kind))
    clib.require_gpu()
    key = np.ascontiguousarray(key, dtype=np.uint32)
    M = 2 * int(N)
    lo, hi = (0, int(num_models)) if rows is None else (int(rows[0]), int(rows[1]))
    W = torch.empty((hi - lo, M, M), device='cuda', dtype=torch.float32)
    z = torch.empty_like(W) if keep_z else None
    arrs = [(ctypes.c_float * 4)(*np.asarray(a, dtype='double').reshape(4)) for a in (J, D, S)]
    ticket = ctypes.c_int(-1)
    zin = None
    if tail is not None:
        zin = torch.empty((hi - lo, M), device='cuda', dtype=torch.float32)
        clib.check(clib.libssnode.ssn_build_w_mt19937_tail_begin_f32(
            key.ctypes.data, int(pos), int(num_models), lo, hi - lo, arrs[0], arrs[1], arrs[2], W.data_ptr(),
            z.data_ptr() if keep_z else None, int(N), TAIL_KINDS[tail], zin.data_ptr(), clib.stream_ptr(), ctypes.byref(ticket)),
            'ssn_build_w_mt19937_tail_begin_f32')
    else:
        clib.check(clib.libssnode.ssn_build_w_mt19937_begin_f32(
            key.ctypes.data, int(pos), int(num_models), lo, hi - lo, arrs[0], arrs[1], arrs[2], W.data_ptr(),
            z.data_ptr() if keep_z else None, int(N), clib.stream_ptr(), ctypes.byref(ticket)), 'ssn_build_w_mt19937_begin_f32')
    finish = _ticket_finisher(ticket.value, key, kind, has_gauss, cached)
    if hasattr(rng, '_defer'):
        rng._defer(finish)            # (fetched when `rng` is next used: utils.DeviceContinuedRandomState)
    else:
        finish(rng)
    return MTWeights(z, W) if tail is None else (MTWeights(z, W), zin)


class PhiloxDraw(object):
    """A slice of a `DeviceNoise` stream that has been reserved but not generated."""

    def __init__(self, seed, offset, shape, tdtype):
        self.seed, self.offset, self.shape, self.tdtype = int(seed), int(offset), tuple(shape), tdtype

    def materialize(self):
        out = torch.empty(self.shape, device='cuda', dtype=self.tdtype)
        fn = clib.libssnode.ssn_philox_uniform_f32 if self.tdtype == torch.float32 else clib.libssnode.ssn_philox_uniform_f64
        clib.check(fn(self.seed, self.offset, out.data_ptr(), out.numel(), clib.stream_ptr()), 'ssn_philox_uniform')
        return out

    def weights(self, N, J, D, S, keep_z):
        """(z or None, W): `make_W_with_x` of the draw in one launch; z is written only when `keep_z`."""
        import ctypes
        B, M, M2 = self.shape
        assert M == M2 == 2 * N
        W = torch.empty(self.shape, device='cuda', dtype=self.tdtype)
        z = torch.empty_like(W) if keep_z else None
        ct, fn = ((ctypes.c_float, clib.libssnode.ssn_build_w_philox_f32) if self.tdtype == torch.float32
                  else (ctypes.c_double, clib.libssnode.ssn_build_w_philox_f64))
        arrs = [(ct * 4)(*np.asarray(a, dtype='double').reshape(4)) for a in (J, D, S)]
        clib.check(fn(self.seed, self.offset, arrs[0], arrs[1], arrs[2], W.data_ptr(), z.data_ptr() if keep_z else None,
                      int(B), int(N), clib.stream_ptr()), 'ssn_build_w_philox')
        return z, W


class PhiloxAmp(object):
    """The heterogeneous-input noise of a forward (z = +-1 or 2 u - 1, and 1 + v z) as a reserved slice of the stream."""

    def __init__(self, seed, offset, shape, tdtype, v, bernoulli):
        self.seed, self.offset, self.shape, self.tdtype, self.v, self.bernoulli = int(seed), int(offset), tuple(shape), tdtype, v, bernoulli

    def materialize(self):
        zin = torch.empty(self.shape, device='cuda', dtype=self.tdtype)
        amp = torch.empty_like(zin)
        fn = clib.libssnode.ssn_philox_amp_f32 if self.tdtype == torch.float32 else clib.libssnode.ssn_philox_amp_f64
        clib.check(fn(self.seed, self.offset, self.v.data_ptr(), zin.data_ptr(), amp.data_ptr(), zin.numel(),
                      int(self.shape[-1]), int(self.bernoulli), clib.stream_ptr()), 'ssn_philox_amp')
        return zin, amp


class DeviceNoise(object):
    """Uniform [0, 1) noise drawn on the device from ONE counter-based stream (Philox4x32-10, `ssn_philox_uniform_*`).

    A draw of local shape (n, ...) stands for rows [rank * n, (rank + 1) * n) of the GLOBAL draw of shape
    (world * n, ...): every rank generates only its own rows, and all ranks advance the stream position by the
    global element count.  A data-parallel job therefore uses exactly the numbers a single process with the same seed
    uses, ranks never repeat each other's draws, and the state (seed, position) is the same on every rank -- one
    checkpoint restores all of them."""

    def __init__(self, seed, rank=0, world=1):
        self.seed, self.rank, self.world = int(seed), int(rank), int(world)
        self.position = 0                      # elements of the global stream consumed so far

    def take(self, n):
        """Stream offset of THIS rank's n elements of the next global draw of world * n elements; advances the position
        by the global count (pure bookkeeping: the same on every rank but for the rank's own offset)."""
        offset = self.position + self.rank * n
        self.position += self.world * n
        return offset

    def lazy_uniform(self, local_shape, tdtype):
        """The next `uniform` draw of this shape, NOT generated: its place in the stream (`PhiloxDraw`).  The consumer either
        materialises it or hands (seed, offset) to a kernel that generates the numbers where it uses them
        (`ssn_build_w_philox_*`: W from the draw without the draw ever being stored)."""
        n = int(np.prod(local_shape))
        return PhiloxDraw(self.seed, self.take(n), tuple(local_shape), tdtype)

    def uniform(self, local_shape, tdtype):
        out = torch.empty(tuple(local_shape), device='cuda', dtype=tdtype)
        n = out.numel()
        fn = clib.libssnode.ssn_philox_uniform_f32 if tdtype == torch.float32 else clib.libssnode.ssn_philox_uniform_f64
        clib.check(fn(self.seed, self.take(n), out.data_ptr(), n, clib.stream_ptr()), 'ssn_philox_uniform')
        return out

    def lazy_signs_and_amp(self, local_shape, tdtype, v, bernoulli):
        """The next `signs_and_amp` draw of this shape, reserved but not generated (`PhiloxAmp`)."""
        n = int(np.prod(local_shape))
        return PhiloxAmp(self.seed, self.take(n), tuple(local_shape), tdtype, v, bool(bernoulli))

    def signs_and_amp(self, local_shape, tdtype, v, bernoulli):
        """The heterogeneous-input noise z (+-1 or 2 u - 1 from the next `uniform` draw of this shape) and 1 + v * z,
        in one launch (`ssn_philox_amp_*`)."""
        zin = torch.empty(tuple(local_shape), device='cuda', dtype=tdtype)
        amp = torch.empty_like(zin)
        n = zin.numel()
        assert v.dtype == tdtype and v.numel() == local_shape[-1] and v.is_contiguous()
        fn = clib.libssnode.ssn_philox_amp_f32 if tdtype == torch.float32 else clib.libssnode.ssn_philox_amp_f64
        clib.check(fn(self.seed, self.take(n), v.data_ptr(), zin.data_ptr(), amp.data_ptr(), n,
                      int(local_shape[-1]), int(bool(bernoulli)),
                      clib.stream_ptr()), 'ssn_philox_amp')
        return zin, amp

    def get_state(self):
        return dict(seed=self.seed, position=self.position)

    def set_state(self, state):
        self.seed, self.position = int(state['seed']), int(state['position'])


class TuningCurveGenerator(object):
    """Generator with a conditional prober (cwgan.py:72-120) or a fixed prober (ssn.py:779-851).

    Parameters mirror the union of the reference components' constructor arguments
    (stimulator: num_sites, num_tcdom, smoothness; core: J, D, S, k, n, tau_E, tau_I, dt,
    io_type; model: seqlen, skip_steps, include_rate_penalty, include_time_avg; prober: probes).
    """

    def __init__(self, num_sites, num_tcdom, smoothness, J, D, S, k, n, tau_E, tau_I, dt, io_type,
                 seqlen, skip_steps, batchsize, probes=None, include_rate_penalty=True,
                 include_time_avg=False, unroll_scan=False, dtype='float32', z_device_seed=None, shard=(0, 1),
                 ssn_type='default', V=0, dist_in='bernoulli', gen_kernel='auto', z_host_draw=False):
        clib.require_gpu()
        # kernel family of the forward / adjoint launches (`ssn_gen_params.kernel`, names in clib.GEN_KERNELS): explicit
        # state of the generator, written to info.json and checkpoints -- 'auto' follows the library's operand-precision
        # setting (clib.set_operand_precision), 'mfma-fp32' / 'tile' keep fp32 operands whatever that setting is
        self.kernel = clib.gen_kernel_code(gen_kernel)
        if ssn_type not in ssn_type_choices:
            raise ValueError('Unknown ssn_type: {}'.format(ssn_type))
        assert dist_in in dist_in_choices
        self.ssn_type = ssn_type
        self.dist_in = dist_in
        # ssn.py:688-740: 'heteroin' has V = (V_E, V_I); 'deg-heteroin' one scalar V used for both populations
        if ssn_type == 'heteroin':
            self.V = np.ascontiguousarray(np.broadcast_to(np.asarray(V, dtype='float64'), 2))
        elif ssn_type == 'deg-heteroin':
            self.V = np.asarray(V, dtype='float64').reshape(())
        else:
            self.V = None
        self.num_sites = int(num_sites)
        self.num_tcdom = int(num_tcdom)
        self.smoothness = float(smoothness)
        self.J = np.array(J, dtype='float64').reshape(2, 2)
        self.D = np.array(D, dtype='float64').reshape(2, 2)
        self.S = np.array(S, dtype='float64').reshape(2, 2)
        self.k, self.n = float(k), float(n)
        self.tau_E, self.tau_I, self.dt = float(tau_E), float(tau_I), float(dt)
        self.io_type = io_type
        self.seqlen, self.skip_steps = int(seqlen), int(skip_steps)
        self.batchsize = int(batchsize)
        self.probes = None if probes is None else np.asarray(probes, dtype=np.int64)
        self.include_rate_penalty = include_rate_penalty
        self.include_time_avg = include_time_avg
        self.unroll_scan = unroll_scan            # accepted for config compatibility; no effect
        self.dtype = dtype
        self.tdtype = {'float32': torch.float32, 'float64': torch.float64}[dtype]
        if not clib.libssnode.ssn_gen_supported(2 * self.num_sites, 4 if dtype == 'float32' else 8):
            raise NotImplementedError('no generator kernel instantiation for num_sites={} ({})'
                                      .format(num_sites, dtype))
        names = ['model_dynamics_penalty']
        if include_rate_penalty:
            names.append('model_rate_penalty')
        if include_time_avg:
            names.append('model_time_avg')
        names.append('prober_tuning_curve')
        self.OutType = collections.namedtuple('OutType', names)
        # device-side noise (perf mode): one counter-based Philox stream per generator instead of host MT19937;
        # `shard` = (rank, world) of a data-parallel job: every rank fills ITS rows of the one global stream
        self._zgen = None
        if z_device_seed is not None:
            self._zgen = DeviceNoise(int(z_device_seed), *shard)
        # reference-stream mode (no z_device_seed): `zs = rng.rand(...)` of the caller's RandomState is generated on the device,
        # bit for bit (`device_rand`); z_host_draw = True keeps the draw on the host (numpy itself: the A/B of the tests)
        self.z_host_draw = bool(z_host_draw)

    num_neurons = property(lambda self: 2 * self.num_sites)
    conditional = property(lambda self: self.probes is None)
    output_shape = property(lambda self: (self.batchsize, self.num_tcdom if self.conditional
                                          else self.num_tcdom * len(self.probes)))
    cond_shape = property(lambda self: (self.batchsize, 3))

    # -- parameters (generator.get_all_params order: J, D, S) -----------------------------
    heteroin = property(lambda self: self.V is not None)

    @property
    def vpop(self):
        return np.broadcast_to(self.V, 2).astype('float64')

    def get_all_params(self):
        """Order of the reference's ``get_all_params`` (stimulator parameters first: ssn.py:255-259, 703-705)."""
        params = [('V', self.V)] if self.heteroin else []
        return params + [('J', self.J), ('D', self.D), ('S', self.S)]

    def get_flat_param_names(self):
        return make_flat_param_names(self.get_all_params())

    def get_flat_param_values(self):
        return concat_flat(v for _, v in self.get_all_params())

    def set_params(self, params):
        rest = dict(params)
        for name in ('J', 'D', 'S'):
            if name in rest:
                setattr(self, name, np.array(rest.pop(name), dtype='float64').reshape(2, 2))
        if 'V' in rest and self.heteroin:
            self.V = np.array(rest.pop('V'), dtype='float64').reshape(self.V.shape)
        if rest:
            raise ValueError('Unknown parameters: {}'.format(rest))

    @property
    def fused_backward(self):
        """'duo-fused': `backward` runs the one-launch sweep + dL/dW (`genops.gen_backward_fused`)."""
        return self.kernel == clib.GEN_KERNEL_FUSED

    @property
    def gen_kernel(self):
        return next(name for name, code in clib.GEN_KERNELS.items() if code == self.kernel)

    def forward_variant(self, num_models=None, save=False):
        """The forward kernel a call with `num_models` draws runs (`ssn_gen_forward_variant`; fp32 only, else 1)."""
        if self.dtype != 'float32':
            return 1
        return genops.forward_variant(self.batchsize if num_models is None else num_models, self.num_tcdom,
                                      self.num_neurons, self.gen_params(), save=save)

    def gen_params(self, rate_penalty_threshold=200.0):
        return genops.make_gen_params(io_type=self.io_type, k=self.k, n=self.n, tau_E=self.tau_E, tau_I=self.tau_I,
                                      dt=self.dt, seqlen=self.seqlen, skip_steps=self.skip_steps,
                                      rate_penalty_threshold=rate_penalty_threshold,
                                      kernel=8 if self.fused_backward else self.kernel)

    # -- noise -----------------------------------------------------------------------------
    def gen_noise(self, rng, stimulator_bandwidths, rows=None, keep_z=None, **_):
        """ssn.py:434-439: ``zs = rng.rand(batchsize, 2N, 2N)`` of the caller's RandomState -- the reference's stream, generated
        on the device bit for bit and leaving `rng` as numpy would (`device_rand`; `z_host_draw`: numpy draws it on the host) --
        or a device Philox draw when the generator was built with `z_device_seed` (another stream: perf mode).
        `rows = (lo, hi)`: this rank's models of the global draw (the whole draw is consumed, only these rows are returned).
        `keep_z` = True / False (callers that run the forward right away, with the CURRENT J, D, S): W is formed in the draw's
        own launch (`device_rand_weights`) and `model_zs` is an `MTWeights`; z itself is written only when kept (the generator
        update's chain rule reads it)."""
        num_models = np.shape(stimulator_bandwidths)[0]
        M = self.num_neurons
        if self._zgen is not None:
            # (reserved, not generated: _device_inputs forms W straight from the stream -- `PhiloxDraw.weights`)
            noise = dict(model_zs=self._zgen.lazy_uniform((num_models, M, M), self.tdtype))
            if self.heteroin:
                vs = self._input_variability()
                # (reserved like z: _device_inputs generates it together with the stimulus and W, `ssn_gen_inputs_philox_f32`)
                noise['model_zs_in'] = self._zgen.lazy_signs_and_amp((num_models, M), self.tdtype, vs, self.dist_in == 'bernoulli')
            return noise
        sl = slice(None) if rows is None else slice(int(rows[0]), int(rows[1]))
        # zs_in of the heterogeneous-input models follows zs in the stream (ssn.py:764-767): drawn by the same device call when
        # it can be (fp32, a generator whose state is fetched lazily), so that the host need not wait for the state in between
        tail = self.dist_in if (self.heteroin and _TAIL and self.tdtype == torch.float32 and hasattr(rng, '_defer')) else None
        if self.z_host_draw or not _is_mt19937(rng):
            noise = dict(model_zs=rng.rand(num_models, M, M)[sl])
        elif keep_z is not None and _FUSE_W and self.tdtype == torch.float32 and num_models * M * M < (1 << 28):
            got = device_rand_weights(rng, num_models, self.num_sites, self.J, self.D, self.S, rows=rows, keep_z=keep_z, tail=tail)
            if tail is not None:
                return dict(model_zs=got[0], model_zs_in=got[1])
            noise = dict(model_zs=got)
        else:
            got = device_rand(rng, (num_models, M, M), self.tdtype, rows=rows, tail=None if tail is None else (tail, M))
            if tail is not None:
                return dict(model_zs=got[0], model_zs_in=got[1])
            noise = dict(model_zs=got)
        if self.heteroin:                       # drawn AFTER zs (ssn.py:764-767), ssn.py:707-720; small: stays on the host
            shape = (num_models, M)
            noise['model_zs_in'] = (rng.choice(2, shape) * 2 - 1 if self.dist_in == 'bernoulli'
                                    else rng.rand(*shape) * 2 - 1)[sl]
        return noise

    # -- forward ------------------------------------------------------------------------------
    def _cached_upload(self, slot, host, dtype):
        """Device copy of a small host array, re-uploaded only when its contents change (the stimulus grid of a run is the
        same at every step when there is one contrast; the input-variability vector changes once per generator step)."""
        cache = self.__dict__.setdefault('_upload_cache', {})
        hit = cache.get(slot)
        if hit is not None and hit[0].shape == host.shape and hit[0].dtype == host.dtype and np.array_equal(hit[0], host):
            return hit[1]
        dev = to_device(host, dtype)
        cache[slot] = (np.array(host, copy=True), dev)
        return dev

    def _input_variability(self):
        pd = self.__dict__.get('_params_dev')
        if pd is not None:
            # (V where the optimizer launch leaves it: the host has not read the new value yet -- `forward(params_dev=...)`)
            v = pd['V'].to(self.tdtype).reshape(-1)
            return (v.expand(2) if v.numel() == 1 else v).repeat_interleave(self.num_sites)
        # (per-neuron vector from the population values; rebuilt only when V changed -- twice per forward otherwise)
        v = np.asarray(self.V, dtype='float64')
        hit = self.__dict__.get('_vs_of')
        if hit is None or hit[0].shape != v.shape or not np.array_equal(hit[0], v):
            hit = self._vs_of = (v.copy(), np.asarray(neu_array(self.vpop, self.num_sites), dtype='float64'))
        return self._cached_upload('vs', hit[1], self.tdtype)

    def _device_inputs(self, stimulator_bandwidths, stimulator_contrasts, model_zs, model_zs_in=None, save=True):
        bw = stimulator_bandwidths if torch.is_tensor(stimulator_bandwidths) else \
            self._cached_upload('bw', np.asarray(stimulator_bandwidths), self.tdtype)
        con = stimulator_contrasts if torch.is_tensor(stimulator_contrasts) else \
            self._cached_upload('con', np.asarray(stimulator_contrasts), self.tdtype)
        amp = None
        self._zin = None
        if (isinstance(model_zs, PhiloxDraw) and self.tdtype == torch.float32 and torch.is_tensor(bw) and torch.is_tensor(con)
                and bw.dtype == torch.float32 and con.dtype == torch.float32 and bw.is_contiguous() and con.is_contiguous()
                and (isinstance(model_zs_in, PhiloxAmp) if self.heteroin else model_zs_in is None)):
            # device noise: heterogeneous-input signs, stimulus and W in ONE library call (the launches of the branches below, in
            # their order, without the host between them)
            import ctypes
            B, NB = bw.shape
            N, M = self.num_sites, self.num_neurons
            W = torch.empty((B, M, M), device='cuda', dtype=torch.float32)
            z = torch.empty_like(W) if save else None
            ext = torch.empty((B, NB, M), device='cuda', dtype=torch.float32)
            zin = amp = None
            if self.heteroin:
                zin = torch.empty((B, M), device='cuda', dtype=torch.float32)
                amp = torch.empty_like(zin)
            arrs = [(ctypes.c_float * 4)(*np.asarray(a, dtype='double').reshape(4)) for a in (self.J, self.D, self.S)]
            a = clib.GenInputs(seed=model_zs.seed, off_z=model_zs.offset, off_zin=model_zs_in.offset if self.heteroin else 0,
                               J=arrs[0], D=arrs[1], S=arrs[2], bw=bw.data_ptr(), con=con.data_ptr(), smoothness=float(self.smoothness),
                               v=model_zs_in.v.data_ptr() if self.heteroin else None,
                               bernoulli=int(model_zs_in.bernoulli) if self.heteroin else 0, W=W.data_ptr(),
                               z=z.data_ptr() if save else None, zin=zin.data_ptr() if self.heteroin else None,
                               amp=amp.data_ptr() if self.heteroin else None, ext=ext.data_ptr(), B=int(B), NB=int(NB), N=int(N))
            clib.check(clib.libssnode.ssn_gen_inputs_philox_f32(ctypes.byref(a), clib.stream_ptr()), 'ssn_gen_inputs_philox_f32')
            self._zin = zin
            self._ext_base = (stimulus_batch(bw, con, self.smoothness, self.num_sites, dtype=self.dtype)
                              if self.heteroin and save else None)
            return ext, z, W
        zin = vsrc = None
        if self.heteroin:
            if isinstance(model_zs_in, PhiloxAmp):
                zin, amp = model_zs_in.materialize()                      # device noise outside the one-call path (fp64)
            else:
                zin = (model_zs_in.to(self.tdtype) if torch.is_tensor(model_zs_in)      # (drawn on the device: `gen_noise`)
                       else to_device(model_zs_in, self.tdtype))          # (pinned staging: no wait for queued kernels)
                if self.tdtype == torch.float32 and _STIM_HETERO:
                    # 1 + v z_in inside the stimulus launch (same two roundings): V where the optimizer launch left it (one
                    # value per population), or the cached per-neuron vector of the host's V
                    pd = self.__dict__.get('_params_dev')
                    vsrc = pd['V'].to(self.tdtype).reshape(-1) if pd is not None else self._input_variability()
                else:
                    amp = 1 + self._input_variability()[None, :] * zin    # ssn.py:679-684
            self._zin = zin
        if vsrc is not None:
            ext = stimulus_batch(bw, con, self.smoothness, self.num_sites, dtype=self.dtype, zin=zin, v=vsrc)
        else:
            ext = stimulus_batch(bw, con, self.smoothness, self.num_sites, dtype=self.dtype, amp=amp)
        # the un-amplified stimulus is only needed by the V gradient of a BPTT step
        self._ext_base = (stimulus_batch(bw, con, self.smoothness, self.num_sites, dtype=self.dtype)
                          if self.heteroin and save else None)
        if isinstance(model_zs, PhiloxDraw):
            z, W = model_zs.weights(self.num_sites, self.J, self.D, self.S, keep_z=save)
            return ext, z, W
        if isinstance(model_zs, MTWeights):            # (W was formed with the draw: `gen_noise(..., keep_z=...)`)
            assert model_zs.z is not None or not save, 'the generator step keeps z: draw with keep_z=True'
            return ext, model_zs.z, model_zs.W
        if torch.is_tensor(model_zs):
            z = model_zs.to('cuda', self.tdtype).contiguous()
        else:
            z = torch.as_tensor(np.ascontiguousarray(model_zs)).to('cuda', self.tdtype)
        pd = self.__dict__.get('_params_dev')
        if pd is not None:
            W = torch.empty_like(z)
            clib.check(clib.libssnode.ssn_build_w_devparams_f32(z.data_ptr(), pd['JDS'].data_ptr(), W.data_ptr(), int(z.shape[0]),
                                                                int(self.num_sites), clib.stream_ptr()), 'ssn_build_w_devparams_f32')
            return ext, z, W
        W = generate_weight_batch(self.num_sites, self.J, self.D, self.S, z, dtype=self.dtype)
        return ext, z, W

    def _probe_indices(self, prober_norm_probes=None, prober_model_ids=None, prober_cell_types=None):
        """(ids, probes) of the conditional prober as int64 device tensors (cwgan.py:91-93), cached while the probe set stays."""
        key = (np.asarray(prober_norm_probes, dtype='float64').tobytes(), np.asarray(prober_cell_types).tobytes(),
               np.asarray(prober_model_ids).tobytes())
        hit = self.__dict__.get('_probe_cache')
        if hit is None or hit[0] != key:
            probes = sample_sites_from_stim_space_impl(np.asarray(prober_norm_probes, dtype='float64'),
                                                       self.num_sites, type='uint16').astype(np.int64) \
                + np.asarray(prober_cell_types).astype(np.int64) * self.num_sites     # cwgan.py:91-93
            hit = self._probe_cache = (key, to_device(np.asarray(prober_model_ids).astype(np.int64)), to_device(probes))
        return hit[1], hit[2]

    def _probe(self, time_avg, prober_norm_probes=None, prober_model_ids=None, prober_cell_types=None):
        if self.conditional:
            # (the probe set of a run rarely changes from step to step: index arithmetic and uploads only when it does)
            ids, pr = self._probe_indices(prober_norm_probes, prober_model_ids, prober_cell_types)
            return time_avg[ids, :, pr], ids, pr                                       # cwgan.py:98
        if getattr(self, '_probes_dev', None) is None or self._probes_dev[0] is not self.probes:
            self._probes_dev = (self.probes, to_device(np.asarray(self.probes)))
        pr = self._probes_dev[1]
        tc = time_avg[:, :, pr].reshape(time_avg.shape[0], -1)                         # ssn.py:846-848
        return tc, None, pr

    def accepts_params_dev(self, model_zs):
        """Whether `forward(params_dev=...)` can form this draw's W from device-resident parameters: fp32, z given as numbers."""
        return self.tdtype == torch.float32 and (torch.is_tensor(model_zs) or isinstance(model_zs, np.ndarray))

    def forward(self, rng=None, save=False, params_dev=None, **kwargs):
        """ssn.py:910-914.  Keyword inputs: stimulator_bandwidths, stimulator_contrasts (num_models, num_tcdom),
        model_zs (optional when `rng` is given), model_rate_penalty_threshold, and for the conditional prober
        prober_norm_probes, prober_model_ids, prober_cell_types.  Outputs are torch CUDA tensors.
        `params_dev` = dict(JDS=device float32[12][, V=device float32[1 or 2]]): W (and the input variability) from these
        device values instead of the attributes J, D, S, V -- for a caller whose optimizer launch has just been queued and
        who has not read the new values back yet (`accepts_params_dev`; same values, same W bits)."""
        if rng is not None or self._zgen is not None:
            if 'model_zs' not in kwargs:
                kwargs.update(self.gen_noise(rng, **kwargs))
        theta = kwargs.pop('model_rate_penalty_threshold', 200.0)
        if params_dev is not None:
            assert self.accepts_params_dev(kwargs['model_zs'])
        self._params_dev = params_dev
        try:
            ext, z, W = self._device_inputs(kwargs.pop('stimulator_bandwidths'), kwargs.pop('stimulator_contrasts'),
                                            kwargs.pop('model_zs'), kwargs.pop('model_zs_in', None), save=save)
        finally:
            self._params_dev = None
        probe_kw = {k: kwargs.pop(k) for k in list(kwargs) if k.startswith('prober_')}
        assert not kwargs, 'unknown inputs: {}'.format(sorted(kwargs))
        gp = self.gen_params(theta)
        if self.conditional:
            # (the probe gather rides in the forward's reduction launch: `ssn_penalty_means_probe_*`)
            ids, pr = self._probe_indices(**probe_kw)
            fwd = genops.gen_forward(W, ext, gp, save=save, probe=(ids, pr))
            tc = fwd['tuning_curve']
        else:
            fwd = genops.gen_forward(W, ext, gp, save=save)
            tc, ids, pr = self._probe(fwd['time_avg'], **probe_kw)
        self.last_penalties = fwd['penalties']        # fp64 [dynamics_penalty, rate_penalty] of this call, one device tensor
        vals = [fwd['dynamics_penalty']]
        if self.include_rate_penalty:
            vals.append(fwd['rate_penalty'])
        if self.include_time_avg:
            vals.append(fwd['time_avg'])
        vals.append(tc)
        out = self.OutType(*vals)
        if save:
            self._saved = dict(fwd=fwd, W=W, z=z, gp=gp, ids=ids, probes=pr, zin=self._zin, ext_base=self._ext_base, ext=ext,
                               theta=theta)
        return out

    def backward(self, g_tuning_curve, dynamics_cost, rate_cost, as_tensor=False, raw=False, exact=False, subset=None):
        """BPTT: gradient of  sum(g_tuning_curve * tuning_curve) + dynamics_cost * dynamics_penalty
        + rate_cost * rate_penalty  w.r.t. the generator parameters (dict: J, D, S[, V]), for the last
        ``forward(save=True)`` call.  float64 numpy arrays by default; with ``as_tensor=True`` float64 CUDA tensors
        and no host synchronisation anywhere in the call (the GAN loop keeps queuing work behind it).
        ``raw=True``: the pieces `genops.gen_grads` turns into the flat gradient vector in one launch -- dict(parts (B, 4, 3)
        float64, nv[, g_ext, ext_base, zin]) -- instead of the sums.
        ``exact=True`` (after a ``raw=True`` call): the step again on the fp32 kernels, forward included; with ``subset`` (int64
        CUDA indices of draws) only for those draws -- the pieces come back with len(subset) rows, for the caller to put in
        place of the rows it could not use."""
        sv = self._saved
        full = None
        if exact:
            # the step again on the fp32 kernels, forward included (the split sweep has overwritten f' with its deltas): the
            # gradient the reference's fp32 arithmetic gives where the fp16-split adjoint refuses (a draw whose adjoint grows
            # by more than 2^8 within one step -- unstable dynamics -- is NaN there by construction)
            sv = self._retry
            full = (sv['shape'], sv['n_dyn'], sv['n_rate'])
            if subset is not None:
                sv = dict(sv, **{k: sv[k][subset].contiguous() for k in ('W', 'z', 'zin', 'ext_base', 'ext') if sv[k] is not None})
            gp = genops.make_gen_params(io_type=self.io_type, k=self.k, n=self.n, tau_E=self.tau_E, tau_I=self.tau_I, dt=self.dt,
                                        seqlen=self.seqlen, skip_steps=self.skip_steps, rate_penalty_threshold=sv['theta'],
                                        kernel=clib.GEN_KERNELS['mfma-fp32'])
            # (trajectory and f' are all the sweep needs of it: no probe gather)
            sv = dict(sv, gp=gp, fwd=genops.gen_forward(sv['W'], sv['ext'], gp, save=True))
        fwd = sv['fwd']
        g = g_tuning_curve.to(fwd['time_avg'].dtype)
        shape_full = full[0] if full is not None else tuple(fwd['time_avg'].shape)
        if self.conditional:
            # scatter-add of the probe gather (several samples may probe the same model/neuron):
            # tuning_curve[n, :] = time_avg[ids[n], :, probes[n]]; one launch, samples added in order, no host wait
            # (torch.index_add_ spends a millisecond on the host per call, index_put_(accumulate=True) synchronises)
            B, NB, M = shape_full
            g = g.contiguous()
            g_ta = torch.empty(shape_full, device=g.device, dtype=g.dtype)
            fn = clib.libssnode.ssn_probe_scatter_f32 if g.dtype == torch.float32 else clib.libssnode.ssn_probe_scatter_f64
            clib.check(fn(g.data_ptr(), sv['ids'].data_ptr(), sv['probes'].data_ptr(), g_ta.data_ptr(), int(g.shape[0]),
                          int(B), int(NB), int(M), clib.stream_ptr()),
                       'ssn_probe_scatter')
        else:
            g_ta = torch.zeros(shape_full, device=g.device, dtype=g.dtype)
            g_ta[:, :, sv['probes']] = g.reshape(g_ta.shape[0], g_ta.shape[1], -1)
        if subset is not None:
            g_ta = g_ta[subset].contiguous()
        # (the means of the loss run over the WHOLE batch, whatever part of it this call sweeps)
        n_dyn, n_rate = (full[1], full[2]) if full is not None else (fwd['n_dyn'], fwd['n_rate'])
        c_dyn, c_rate = dynamics_cost / max(n_dyn, 1), rate_cost / n_rate
        if self.fused_backward and not exact:
            B, NB, _, M = fwd['traj'].shape
            xmax = genops.rate_bound(sv['gp'])
            if self.dtype != 'float32' or not genops.gen_backward_fused_supported(B, NB, M, sv['gp'], xmax):
                raise ValueError("gen_kernel 'duo-fused' needs fp32, at most 8 stimuli, 2N <= 208 and the saturating I/O "
                                 "function with dt <= tau (a bound on the rates)")
            gW, g_ext, dmax = genops.gen_backward_fused(sv['W'], fwd['traj'], fwd['df'], g_ta, c_dyn, c_rate, sv['gp'], xmax,
                                                        want_g_ext=self.heteroin)
            self.last_dmax = dmax
        else:
            res = genops.gen_backward(sv['W'], fwd['traj'], fwd['df'], g_ta, c_dyn, c_rate, sv['gp'],
                                      want_g_ext=self.heteroin, want_dmax=True)
            delta, g_ext, dmax = res if self.heteroin else (res[0], None, res[1])
            # (kept for `poisoned_draws`: a draw whose adjoint outgrew the fp16 sweep's lagged scale has NaN here)
            if not exact:
                self.last_dmax = dmax
            # (fp16 two-part form of dL/dW where the sweep handed over max |delta| per draw and the rates are bounded)
            gW = genops.weight_grad(delta, fwd['traj'], dmax=dmax, xmax=genops.rate_bound(sv['gp']))
        if raw:
            pieces = dict(parts=genops.jds_grad_parts(gW, sv['z'], self.J, self.D, self.S), nv=0)
            if self.heteroin:
                pieces.update(nv=2 if self.ssn_type == 'heteroin' else 1, g_ext=g_ext, ext_base=sv['ext_base'], zin=sv['zin'])
            # (what a second pass on the fp32 kernels needs, should this one turn out poisoned: references, no copies)
            self._retry = None if exact else dict({k: sv[k] for k in ('W', 'z', 'ids', 'probes', 'zin', 'ext_base', 'ext', 'theta')},
                                                   shape=tuple(fwd['time_avg'].shape), n_dyn=fwd['n_dyn'], n_rate=fwd['n_rate'])
            self._saved = None
            return pieces
        gJ, gD, gS = genops.jds_grad(gW, sv['z'], self.J, self.D, self.S, as_tensor=as_tensor)
        grads = dict(J=gJ, D=gD, S=gS)
        if self.heteroin:
            # ext = (1 + v_pop z_in) ext_base  ->  dL/dv_pop = sum over the population of g_ext * ext_base * z_in
            B, NB, M = g_ext.shape
            per = (g_ext.to(torch.float64) * sv['ext_base'].to(torch.float64) * sv['zin'].to(torch.float64)[:, None, :])
            gv = per.reshape(B, NB, 2, M // 2).sum(dim=(0, 1, 3))
            if as_tensor:
                grads['V'] = gv if self.ssn_type == 'heteroin' else gv.sum().reshape(())
            else:
                gv = gv.cpu().numpy()
                grads['V'] = gv if self.ssn_type == 'heteroin' else np.asarray(gv.sum())
        self._saved = None
        return grads

    def poisoned_draw_indices(self):
        """int64 CUDA indices of the draws `poisoned_draws` counts (empty when the sweep tracks no max |delta|)."""
        dmax = getattr(self, 'last_dmax', None)
        if dmax is None:
            return torch.empty(0, device='cuda', dtype=torch.int64)
        return torch.isnan(dmax).nonzero().reshape(-1)

    def poisoned_draws(self):
        """Number of draws of the last `backward` whose adjoint outgrew the lagged power-of-two scale of the fp16-split
        sweeps (their gradient is NaN by construction, csrc/ssn_duo.hip `gen_backward_duo`); 0 for the fp32 sweeps, which
        have no such limit.  Synchronises: for the failure path of the loop, not for every step."""
        dmax = getattr(self, 'last_dmax', None)
        return 0 if dmax is None else int(torch.isnan(dmax).sum())

    def prepare(self):
        """Nothing to compile (the reference forces Theano compilation here)."""

    def to_config(self):
        return dict(num_sites=self.num_sites, num_tcdom=self.num_tcdom, smoothness=self.smoothness,
                    J=self.J.tolist(), D=self.D.tolist(), S=self.S.tolist(), k=self.k, n=self.n,
                    tau_E=self.tau_E, tau_I=self.tau_I, dt=self.dt, io_type=self.io_type,
                    seqlen=self.seqlen, skip_steps=self.skip_steps, batchsize=self.batchsize,
                    include_rate_penalty=self.include_rate_penalty, include_time_avg=self.include_time_avg,
                    unroll_scan=self.unroll_scan, ssn_type=self.ssn_type, ssn_impl='default', gen_kernel=self.gen_kernel,
                    **({} if not self.heteroin else dict(V=np.asarray(self.V).tolist(), dist_in=self.dist_in)),
                    **({} if self.probes is None else dict(probes=self.probes.tolist())))


def is_heteroin(gen):
    return gen.heteroin
