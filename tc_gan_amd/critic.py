"""WGAN-GP critic on the GPU (C ABI section 4 of include/ssnode_mi355x.h).

Host-side mirror of ``ConditionalDiscriminator`` + ``ConditionalCriticTrainer``
(networks/cwgan.py:123-214) and ``Updater`` (networks/wgan.py:111-165): parameters live
in ONE flat fp32 device buffer in Lasagne's ``get_all_params`` order
[W_1, b_1, ..., W_L, b_L, W_out]; every pass is a chain of MFMA GEMM launches in
``csrc/ssn_critic.hip``.
"""
import ctypes

import numpy as np
import torch

from . import clib
from .clib import libssnode

PRECISION = {'bf16': 0, 'fp32': 1}


def _stream():
    return clib.stream_ptr()


# lasagne.nonlinearities names -> slope below zero (rectify, LeakyRectify(0.01), LeakyRectify(1/3), identity)
LEAK = {'rectify': 0.0, 'leaky_rectify': 0.01, 'very_leaky_rectify': 1.0 / 3.0, 'linear': 1.0, 'identity': 1.0}
# ... -> activation code of the general entry points (`ssn_critic_*_act`, include/ssnode_mi355x.h)
ACT = {'rectify': 0, 'leaky_rectify': 1, 'very_leaky_rectify': 2, 'linear': 3, 'identity': 3, 'tanh': 4, 'sigmoid': 5,
       'softplus': 6, 'elu': 7}


class _NoCond(object):
    """Stands for "no condition columns" in the calls of an unconditional critic (the library reads NULL that way)."""

    @staticmethod
    def data_ptr():
        return None


class Critic(object):
    """MLP critic: input = [tuning curve (nx), contrast, |norm_probe|, cell_type] (`ConditionalDiscriminator`,
    networks/cwgan.py:123-175), or the tuning curve alone with ``conditional=False`` (`UnConditionalDiscriminator`,
    networks/wgan.py:66-97: every `cond` argument is then None)."""

    def __init__(self, nx, layers, seed=0, hide_cell_type=False, precision='fp32',
                 normalization='none', nonlinearity='rectify', device=None, conditional=True, net_options=None):
        norms = list(normalization) if isinstance(normalization, (list, tuple)) else [normalization] * len(layers)
        if len(norms) != len(layers) or any(n not in ('none', 'layer') for n in norms):
            raise ValueError('normalization must be none/layer (or one per layer): {!r}'.format(normalization))
        self.normalization = norms
        # hidden nonlinearity (simple_discriminator.py:139-152 takes any name of lasagne.nonlinearities): the piecewise-linear
        # ones are x > 0 ? x : leak * x and keep the rectify fast paths (fixed slopes instead of fixed masks); the smooth ones
        # (tanh, sigmoid, softplus, elu) run on the general layer-by-layer path, whose gradient-penalty double backward
        # carries their curvature.
        if nonlinearity not in ACT:
            raise NotImplementedError('critic nonlinearity {!r}: the GPU path has {}'.format(nonlinearity, sorted(ACT)))
        self.nonlinearity = nonlinearity
        self.act = ACT[nonlinearity]
        self.leak = LEAK.get(nonlinearity, 0.0)
        # simple_discriminator.py:57-75: a layer-normalised layer gains a learnable ScaleLayer (one factor per unit, after the
        # normalisation, before the bias) for every nonlinearity but rectify (`use_scale='auto'`); `net_options` =
        # {'layer': {'use_scale': True / False / 'auto'}} (or one dict per layer) overrides, as the reference's `options` do
        opts = self._layer_options(net_options, norms)
        self.scaled = []
        for n, o in zip(norms, opts):
            unknown = set(o) - {'use_scale'}
            if unknown:
                raise NotImplementedError('critic layer options {}'.format(sorted(unknown)))
            use = o.get('use_scale', 'auto')
            if use not in (True, False, 'auto'):
                raise ValueError('use_scale must be True, False or "auto": {!r}'.format(use))
            if n != 'layer' and 'use_scale' in o:
                raise ValueError('use_scale is an option of layer-normalised layers')
            self.scaled.append(n == 'layer' and (nonlinearity != 'rectify' if use == 'auto' else bool(use)))
        # the general entry points: a smooth nonlinearity anywhere, or a scale (the rectify / leaky fast paths have neither)
        self.general = self.act >= 4 or any(self.scaled)
        clib.require_gpu()
        self.nx = int(nx)
        self.layers = [int(w) for w in layers]
        self.conditional = bool(conditional)
        self.dims = [self.nx + (3 if self.conditional else 0)] + self.layers
        self.nlayers = len(self.layers)
        self.hide_cell_type = int(bool(hide_cell_type))
        self.precision = PRECISION[precision]
        self._dims_c = (ctypes.c_int * len(self.dims))(*self.dims)
        self.layer_norm = any(n == 'layer' for n in norms)
        self._norm_c = (ctypes.c_int * max(len(norms), 1))(*[int(n == 'layer') for n in norms])
        # (flags of the general entry points: 1 = layer normalisation, 3 = with the learnable scale after it)
        self._flags_c = (ctypes.c_int * max(len(norms), 1))(*[int(n == 'layer') + 2 * int(sc) for n, sc in zip(norms, self.scaled)])
        if self.leak and self.layer_norm and not self.general:
            self.general = True              # (leaky + layer normalisation without its scale: no fast path for that either)
        self.num_params = int(libssnode.ssn_critic_num_params_act(self._dims_c, self._flags_c, self.nlayers) if self.general
                              else libssnode.ssn_critic_num_params(self._dims_c, self.nlayers))
        assert self.num_params == sum(int(np.prod(shape)) for _, shape in self.param_shapes())
        self.device = device or torch.device('cuda', torch.cuda.current_device())
        self.params = torch.empty(self.num_params, device=self.device, dtype=torch.float32)
        self.grads = torch.zeros_like(self.params)
        self.stats = torch.zeros(4, device=self.device, dtype=torch.float32)
        self._ws = None
        self._ws_key = None
        self.init_params(np.random.RandomState(seed))

    @staticmethod
    def _layer_options(options, norms):
        """simple_discriminator.py:90-97 (`_validate_options`): None, a dict keyed by normalisation type, or one dict per layer."""
        if options is None:
            return [{}] * len(norms)
        if isinstance(options, dict):
            if not set(options) <= {'none', 'layer'}:
                raise ValueError('net_options keys must be normalisation types: {!r}'.format(sorted(options)))
            return [dict(options.get(n, {})) for n in norms]
        options = [dict(o) for o in options]
        if len(options) != len(norms):
            raise ValueError('net_options: one dict per hidden layer')
        return options

    has_step = property(lambda self: not self.general)     # `step`: the one-call critic update of the rectify / leaky fast paths

    # -- parameters ------------------------------------------------------------------
    def param_shapes(self):
        """(name, shape) in lasagne's get_all_params order: W, [scales,] b per hidden layer, then the output W."""
        shapes = []
        for l in range(self.nlayers):
            shapes.append(('W', (self.dims[l], self.dims[l + 1])))
            if self.scaled[l]:
                shapes.append(('scales', (self.dims[l + 1],)))
            shapes.append(('b', (self.dims[l + 1],)))
        shapes.append(('W', (self.dims[-1], 1)))
        return shapes

    def init_params(self, rng):
        """Lasagne defaults: W ~ GlorotUniform (plain layers) or Normal(std=1) (layer-normalised layers,
        simple_discriminator.py:53), hidden b ~ Normal(std=.01) (149-150), linear output layer without bias
        (160-161)."""
        flat = []
        layer = 0
        for kind, shape in self.param_shapes():
            if kind == 'W':
                if layer < self.nlayers and self.normalization[layer] == 'layer':
                    flat.append(rng.normal(0.0, 1.0, size=shape).ravel())
                else:
                    a = np.sqrt(6.0 / (shape[0] + shape[1]))
                    flat.append(rng.uniform(-a, a, size=shape).ravel())
            elif kind == 'scales':
                flat.append(np.ones(shape).ravel())            # lasagne.layers.ScaleLayer: scales = init.Constant(1)
            else:
                flat.append(rng.normal(0.0, 0.01, size=shape).ravel())
                layer += 1
        self.set_flat(np.concatenate(flat))

    def set_flat(self, flat):
        flat = np.asarray(flat, dtype=np.float32)
        assert flat.shape == (self.num_params,)
        self.params.copy_(torch.from_numpy(flat))

    def get_flat(self):
        return self.params.detach().cpu().numpy()

    def get_param_values(self):
        """List of arrays in ``lasagne.layers.get_all_param_values`` order."""
        flat = self.get_flat()
        out, off = [], 0
        for _, shape in self.param_shapes():
            n = int(np.prod(shape))
            out.append(flat[off:off + n].reshape(shape))
            off += n
        return out

    def get_param_names(self):
        return [kind for kind, _ in self.param_shapes()]

    # -- parameter statistics without a host wait (recorders.py:275-311 logs them after EVERY critic step) --------
    def param_sqnorms_device(self):
        """Sum of squares per parameter tensor (device tensor, `param_shapes` order; two launches: `ssn_segment_sqnorms2_f32`)."""
        self._ensure_segments()
        out = torch.empty(len(self._seg_sizes), device=self.device, dtype=torch.float32)
        clib.check(libssnode.ssn_segment_sqnorms2_f32(self.params.data_ptr(), self._seg_bounds.data_ptr(), int(out.numel()),
                                                      out.data_ptr(), self._seg_ws.data_ptr(), _stream()),
                   'ssn_segment_sqnorms2_f32')
        return out

    def _ensure_segments(self):
        """Bounds of the parameter tensors inside the flat vector (device) and the scratch of `ssn_segment_sqnorms2_f32`."""
        if getattr(self, '_seg_bounds', None) is None:
            sizes = [int(np.prod(shape)) for _, shape in self.param_shapes()]
            self._seg_sizes = np.asarray(sizes, dtype='float64')
            self._seg_bounds = torch.as_tensor(np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)).to(self.device)
            self._seg_ws = torch.empty(int(libssnode.ssn_segment_sqnorms_ws_doubles(len(sizes))), device=self.device,
                                       dtype=torch.float64)

    def cache_param_nnorms(self, sqnorms_host):
        """Store ||p|| / size per tensor for the NEXT `param_nnorms()` call (None: no valid cache)."""
        self._nnorm_cache = (None if sqnorms_host is None else
                             list(np.sqrt(np.asarray(sqnorms_host, dtype='float64')) / self._seg_sizes))

    def param_nnorms(self):
        """Normalised norm of every parameter tensor; uses the values the GAN loop fetched together with the step's
        scalars when there are any (one shot), else reads the parameters back."""
        cached, self._nnorm_cache = getattr(self, '_nnorm_cache', None), None
        if cached is not None:
            return cached
        return [float(np.linalg.norm(arr.flatten()) / arr.size) for arr in self.get_param_values()]

    # -- passes ----------------------------------------------------------------------
    def _workspace(self, bgd, bp):
        """Scratch for a call with these batch sizes: ONE buffer that only grows (a step alternates between the sizes of the
        loss pass and of the forwards; the sizes are asked for once per shape)."""
        key = (bgd, bp)
        need = self.__dict__.setdefault('_ws_need', {}).get(key)
        if need is None:
            fn = (libssnode.ssn_critic_norm_workspace_floats if self.layer_norm or self.general
                  else libssnode.ssn_critic_workspace_floats)
            need = self._ws_need[key] = int(fn(self._dims_c, self.nlayers, int(bgd), int(bp)))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, device=self.device, dtype=torch.float32)
        return self._ws

    @staticmethod
    def _f32(t):
        if t is None:
            return _NoCond
        if torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous():
            return t
        return torch.as_tensor(t).to('cuda', torch.float32).contiguous()

    def _checked(self, *pairs):
        """(x, cond) pairs as device tensors; refuses a condition the critic was not built for (the library reads a NULL
        `cond` as "no condition columns": on a conditional critic it would then read nx + 3 columns from rows of nx)."""
        out = []
        for x, cond in pairs:
            x, cond = self._f32(x), self._f32(cond)
            if (cond is _NoCond) != (not self.conditional):
                raise ValueError('this critic is {}conditional: cond must {}be given'.format(
                    '' if self.conditional else 'un', '' if self.conditional else 'not '))
            if x.dim() != 2 or x.shape[1] != self.nx or (self.conditional and tuple(cond.shape) != (x.shape[0], 3)):
                raise ValueError('critic input of shape {} / cond {}: expected (batch, {}) / (batch, 3)'.format(
                    tuple(x.shape), None if cond is _NoCond else tuple(cond.shape), self.nx))
            out += [x, cond]
        return out

    def forward(self, x, cond):
        x, cond = self._checked((x, cond))
        batch = x.shape[0]
        out = torch.empty(batch, device=self.device, dtype=torch.float32)
        ws = self._workspace(batch, 0)
        if self.general:
            clib.check(libssnode.ssn_critic_forward_act(
                self.params.data_ptr(), self._dims_c, self._flags_c, self.nlayers, self.act, x.data_ptr(), cond.data_ptr(), batch,
                self.hide_cell_type, out.data_ptr(), ws.data_ptr(), self.precision, _stream()), 'ssn_critic_forward_act')
            return out
        if self.layer_norm:
            clib.check(libssnode.ssn_critic_forward_norm(
                self.params.data_ptr(), self._dims_c, self._norm_c, self.nlayers, x.data_ptr(), cond.data_ptr(), batch,
                self.hide_cell_type, out.data_ptr(), ws.data_ptr(), self.precision, _stream()), 'ssn_critic_forward_norm')
            return out
        if self.leak:
            clib.check(libssnode.ssn_critic_forward_leaky(
                self.params.data_ptr(), self._dims_c, self.nlayers, x.data_ptr(), cond.data_ptr(), batch, self.hide_cell_type,
                self.leak, out.data_ptr(), ws.data_ptr(), self.precision, _stream()), 'ssn_critic_forward_leaky')
            return out
        clib.check(libssnode.ssn_critic_forward(self.params.data_ptr(), self._dims_c, self.nlayers, x.data_ptr(),
                                                cond.data_ptr(), batch, self.hide_cell_type, out.data_ptr(),
                                                ws.data_ptr(), self.precision, _stream()), 'ssn_critic_forward')
        return out

    def loss_grad(self, xg, cg, xd, cd, xp, cp, lmd):
        """Fills ``self.grads`` and ``self.stats`` = [mean D(xg), mean D(xd), penalty, loss] (device)."""
        xg, cg, xd, cd, xp, cp = self._checked((xg, cg), (xd, cd), (xp, cp))
        ng, nd, npn = xg.shape[0], xd.shape[0], xp.shape[0]
        ws = self._workspace(ng + nd, npn)
        self._dvals = torch.empty(ng + nd, device=self.device, dtype=torch.float32)
        if self.general:
            clib.check(libssnode.ssn_critic_loss_grad_act(
                self.params.data_ptr(), self._dims_c, self._flags_c, self.nlayers, self.act, xg.data_ptr(), cg.data_ptr(),
                xd.data_ptr(), cd.data_ptr(), xp.data_ptr(), cp.data_ptr(), ng, nd, npn, float(lmd), self.hide_cell_type,
                self.grads.data_ptr(), self.stats.data_ptr(), self._dvals.data_ptr(), ws.data_ptr(), self.precision,
                _stream()), 'ssn_critic_loss_grad_act')
            return self.stats
        if self.layer_norm:
            clib.check(libssnode.ssn_critic_loss_grad_norm(
                self.params.data_ptr(), self._dims_c, self._norm_c, self.nlayers, xg.data_ptr(), cg.data_ptr(),
                xd.data_ptr(), cd.data_ptr(), xp.data_ptr(), cp.data_ptr(), ng, nd, npn, float(lmd), self.hide_cell_type,
                self.grads.data_ptr(), self.stats.data_ptr(), self._dvals.data_ptr(), ws.data_ptr(), self.precision,
                _stream()), 'ssn_critic_loss_grad_norm')
            return self.stats
        if self.leak:
            clib.check(libssnode.ssn_critic_loss_grad_leaky(
                self.params.data_ptr(), self._dims_c, self.nlayers, xg.data_ptr(), cg.data_ptr(), xd.data_ptr(),
                cd.data_ptr(), xp.data_ptr(), cp.data_ptr(), ng, nd, npn, float(lmd), self.hide_cell_type, self.leak,
                self.grads.data_ptr(), self.stats.data_ptr(), self._dvals.data_ptr(), ws.data_ptr(), self.precision,
                _stream()), 'ssn_critic_loss_grad_leaky')
            return self.stats
        clib.check(libssnode.ssn_critic_loss_grad(
            self.params.data_ptr(), self._dims_c, self.nlayers, xg.data_ptr(), cg.data_ptr(), xd.data_ptr(),
            cd.data_ptr(), xp.data_ptr(), cp.data_ptr(), ng, nd, npn, float(lmd), self.hide_cell_type,
            self.grads.data_ptr(), self.stats.data_ptr(), self._dvals.data_ptr(), ws.data_ptr(), self.precision,
            _stream()), 'ssn_critic_loss_grad')
        return self.stats

    def interpolate(self, eps, xd, xg):
        """The gradient-penalty points eps * xd + (1 - eps) * xg (cwgan.py:476-481); eps: one value per row."""
        eps, xd, xg = self._f32(eps).reshape(-1), self._f32(xd), self._f32(xg)
        assert xd.shape == xg.shape and eps.numel() == xd.shape[0]
        xp = torch.empty_like(xd)
        clib.check(libssnode.ssn_interpolate_f32(eps.data_ptr(), xd.data_ptr(), xg.data_ptr(), xp.data_ptr(),
                                                 int(xd.shape[0]), int(xd.shape[1]), _stream()), 'ssn_interpolate_f32')
        return xp

    def input_grad(self, x, cond, scale):
        """gx = scale * dD/dx summed over nothing (per sample), stats[0] = mean D(x)."""
        x, cond = self._checked((x, cond))
        batch = x.shape[0]
        gx = torch.empty((batch, self.nx), device=self.device, dtype=torch.float32)
        ws = self._workspace(batch, batch)
        if self.general:
            clib.check(libssnode.ssn_critic_input_grad_act(
                self.params.data_ptr(), self._dims_c, self._flags_c, self.nlayers, self.act, x.data_ptr(), cond.data_ptr(), batch,
                self.hide_cell_type, float(scale), gx.data_ptr(), self.stats.data_ptr(), ws.data_ptr(), self.precision,
                _stream()), 'ssn_critic_input_grad_act')
            return gx, self.stats[0]
        if self.layer_norm:
            clib.check(libssnode.ssn_critic_input_grad_norm(
                self.params.data_ptr(), self._dims_c, self._norm_c, self.nlayers, x.data_ptr(), cond.data_ptr(), batch,
                self.hide_cell_type, float(scale), gx.data_ptr(), self.stats.data_ptr(), ws.data_ptr(), self.precision,
                _stream()), 'ssn_critic_input_grad_norm')
            return gx, self.stats[0]
        if self.leak:
            clib.check(libssnode.ssn_critic_input_grad_leaky(
                self.params.data_ptr(), self._dims_c, self.nlayers, x.data_ptr(), cond.data_ptr(), batch,
                self.hide_cell_type, self.leak, float(scale), gx.data_ptr(), self.stats.data_ptr(), ws.data_ptr(),
                self.precision, _stream()), 'ssn_critic_input_grad_leaky')
            return gx, self.stats[0]
        clib.check(libssnode.ssn_critic_input_grad(
            self.params.data_ptr(), self._dims_c, self.nlayers, x.data_ptr(), cond.data_ptr(), batch,
            self.hide_cell_type, float(scale), gx.data_ptr(), self.stats.data_ptr(), ws.data_ptr(), self.precision,
            _stream()), 'ssn_critic_input_grad')
        return gx, self.stats[0]

    def accuracy_device(self, xg, cg, xd, cd, out=None):
        """mean D(xg) - mean D(xd) (cwgan.py:139-147) as a 1-element device tensor (`out`, when given), no host wait:
        ONE library call (`ssn_critic_accuracy`: two forwards -- every output row depends on its own input row only, and
        stacking the rows would cost two more launches -- and one reduction in a fixed order)."""
        xg, cg, xd, cd = self._checked((xg, cg), (xd, cd))
        ng, nd = xg.shape[0], xd.shape[0]
        ws = self._workspace(max(ng, nd), 0)
        if out is None:
            out = torch.empty(1, device=self.device, dtype=torch.float32)
        dv = self.__dict__.get('_acc_dvals')
        if dv is None or dv.numel() < ng + nd:
            dv = self._acc_dvals = torch.empty(ng + nd, device=self.device, dtype=torch.float32)
        if self.general:
            clib.check(libssnode.ssn_critic_accuracy_act(
                self.params.data_ptr(), self._dims_c, self._flags_c, self.nlayers, self.act, xg.data_ptr(), cg.data_ptr(),
                xd.data_ptr(), cd.data_ptr(), ng, nd, self.hide_cell_type, out.data_ptr(), dv.data_ptr(), ws.data_ptr(),
                self.precision, _stream()), 'ssn_critic_accuracy_act')
            return out
        clib.check(libssnode.ssn_critic_accuracy(
            self.params.data_ptr(), self._dims_c, self._norm_c if self.layer_norm else None, self.nlayers, float(self.leak),
            xg.data_ptr(), cg.data_ptr(), xd.data_ptr(), cd.data_ptr(), ng, nd, self.hide_cell_type, out.data_ptr(),
            dv.data_ptr(), ws.data_ptr(), self.precision, _stream()), 'ssn_critic_accuracy')
        return out

    def accuracy(self, xg, cg, xd, cd):
        return float(self.accuracy_device(xg, cg, xd, cd)[0])

    def step(self, updater, xg, xd, cond, eps, lmd, pens64=None, rate_penalty_bound=None):
        """One critic step of the GAN loop in ONE library call (`ssn_critic_step_run`): penalty points, loss + gradient,
        `updater`'s step on the parameters, accuracy of the updated critic, per-tensor sums of squares -- the kernels of
        `interpolate`, `loss_grad`, `Updater.__call__` (plain clip-free form), `accuracy_device`, `param_sqnorms_device` in
        that order, so the same numbers.  Returns (xp, tail): tail = [penalties (2, from `pens64`), loss, accuracy, sums of
        squares] on the device."""
        if self.general:
            raise NotImplementedError('Critic.step is the one-call update of the rectify / leaky fast paths (has_step)')
        xg, xd, cond = self._f32(xg), self._f32(xd), self._f32(cond)
        eps = self._f32(eps).reshape(-1)
        n = xg.shape[0]
        assert xd.shape == xg.shape and (cond is _NoCond or cond.shape[0] == n) and eps.numel() == n
        assert (cond is _NoCond) == (not self.conditional)
        self._ensure_segments()
        nseg = len(self._seg_sizes)
        ws = self._workspace(2 * n, n)
        xp = torch.empty_like(xd)
        tail = torch.empty(4 + nseg, device=self.device, dtype=torch.float32)
        self._dvals = torch.empty(2 * n, device=self.device, dtype=torch.float32)
        dv = self.__dict__.get('_acc_dvals')
        if dv is None or dv.numel() < 2 * n:
            dv = self._acc_dvals = torch.empty(2 * n, device=self.device, dtype=torch.float32)
        s1, s2, opt = updater.begin_step(self.params)
        a = clib.CriticStep(
            params=self.params.data_ptr(), dims=self._dims_c, layer_norm=self._norm_c if self.layer_norm else None,
            nlayers=self.nlayers, leak=float(self.leak), xg=xg.data_ptr(), xd=xd.data_ptr(), cond=cond.data_ptr(),
            eps=eps.data_ptr(), n=n, hide_cell_type=self.hide_cell_type, precision=self.precision, lmd=float(lmd),
            xp=xp.data_ptr(), grads=self.grads.data_ptr(), stats=self.stats.data_ptr(), dvals=self._dvals.data_ptr(),
            workspace=ws.data_ptr(), opt_s1=s1.data_ptr(), opt_s2=s2.data_ptr(), opt=ctypes.pointer(opt),
            seg_bounds=self._seg_bounds.data_ptr(), nseg=nseg, seg_ws=self._seg_ws.data_ptr(),
            pens64=pens64.data_ptr() if pens64 is not None else None, acc_dvals=dv.data_ptr(), tail=tail.data_ptr())
        if rate_penalty_bound is not None and rate_penalty_bound > 0:
            # cwgan.py:493-498 on the device: no update (parameters, optimizer state) when the batch's rate penalty exceeds the
            # bound; the caller learns it from tail[1] and takes the step count back (`Updater.uncommit_step`)
            clib.check(libssnode.ssn_critic_step_gated_run(ctypes.byref(a), float(rate_penalty_bound), _stream()),
                       'ssn_critic_step_gated_run')
        else:
            clib.check(libssnode.ssn_critic_step_run(ctypes.byref(a), _stream()), 'ssn_critic_step_run')
        updater.commit_step(opt)          # (only now: a refused launch leaves the step count, hence Adam's bias correction, alone)
        return xp, tail


class Updater(object):
    """wgan.py:111-165: 'adam-wgan' = Adam(beta1=.5, beta2=.9); any of adam / rmsprop / sgd by name;
    L2/L1 penalty (through the loss) and decoupled L2/L1 decay; optional clipping of the new value."""

    _named = {'adam-wgan': ('adam', dict(beta1=0.5, beta2=0.9))}
    _kinds = {'sgd': 0, 'adam': 1, 'rmsprop': 2}

    def __init__(self, learning_rate=0.001, update_name='adam-wgan', update_config=None,
                 reg_l2_penalty=0.0, reg_l2_decay=0.0, reg_l1_penalty=0.0, reg_l1_decay=0.0):
        name, default = self._named.get(update_name, (update_name, {}))
        if name not in self._kinds:
            raise ValueError('Unknown update method: {}'.format(update_name))
        cfg = dict(dict(beta1=0.9, beta2=0.999, epsilon=1e-8 if name == 'adam' else 1e-6, rho=0.9), **default)
        cfg.update(update_config or {})
        self.update_name = update_name
        self.learning_rate = learning_rate
        self.kind = self._kinds[name]
        self.cfg = cfg
        self.reg = (reg_l2_penalty, reg_l1_penalty, reg_l2_decay, reg_l1_decay)
        self.step = 0
        self._state = None

    def state_dict(self):
        """Optimizer state for checkpoints (step counter and the two moment buffers)."""
        st = self._state
        return dict(step=self.step, m=None if st is None else st[0].cpu().numpy(),
                    v=None if st is None else st[1].cpu().numpy())

    def load_state_dict(self, d):
        self.step = int(d['step'])
        if d.get('m') is None:
            self._state = None
        else:
            self._state = tuple(torch.as_tensor(np.asarray(d[k]), device='cuda', dtype=torch.float32).contiguous()
                                for k in ('m', 'v'))

    def snapshot(self, params):
        """Device copies of everything one update changes (no host wait): see `restore`."""
        st = self._state
        return (params.clone(), self.step, None if st is None else (st[0].clone(), st[1].clone()))

    def restore(self, params, snap):
        """Undo the updates made since `snapshot`."""
        params.copy_(snap[0])
        self.step = snap[1]
        self._state = snap[2]

    def begin_step(self, params):
        """Bookkeeping of one clip-free update made by somebody else's launch (`Critic.step`): the state tensors and the
        `ssn_opt_params` of this step, numbered step + 1.  The count itself advances in `commit_step`, which the caller
        runs once its launch has been accepted -- a library call that returns an error leaves the updater where it was."""
        if self._state is None or self._state[0].shape != params.shape:
            self._state = (torch.zeros_like(params), torch.zeros_like(params))
        o = clib.OptParams(kind=self.kind, step=self.step + 1, clip=0, reserved=0,
                           learning_rate=self.learning_rate, beta1=self.cfg['beta1'], beta2=self.cfg['beta2'],
                           epsilon=self.cfg['epsilon'], rho=self.cfg['rho'],
                           reg_l2_penalty=self.reg[0], reg_l1_penalty=self.reg[1],
                           reg_l2_decay=self.reg[2], reg_l1_decay=self.reg[3], clip_lo=0.0, clip_hi=0.0)
        return self._state[0], self._state[1], o

    def commit_step(self, opt):
        self.step = int(opt.step)

    def uncommit_step(self):
        """The last committed update turned out not to have been made (the device-side gate of `Critic.step`)."""
        self.step -= 1

    def __call__(self, params, grads, clip=None):
        """In-place update of the flat device tensor `params` from `grads`.  `clip` = (lo, hi): scalars, or arrays of the
        parameter's shape for bounds per element (the reference clips with numpy broadcasting, wgan.py:244-251 -- its
        heteroin test pins V_I with V_min = [0, 0], V_max = [1, 0])."""
        if self._state is None or self._state[0].shape != params.shape:
            self._state = (torch.zeros_like(params), torch.zeros_like(params))
        elementwise = None
        if clip is not None and (np.ndim(clip[0]) > 0 or np.ndim(clip[1]) > 0):
            lo = np.broadcast_to(np.asarray(clip[0], dtype='float32').ravel(), (params.numel(),))
            hi = np.broadcast_to(np.asarray(clip[1], dtype='float32').ravel(), (params.numel(),))
            key = (lo.tobytes(), hi.tobytes())
            if getattr(self, '_clip_cache', (None,))[0] != key:
                self._clip_cache = (key, torch.as_tensor(np.array(lo)).to(params.device), torch.as_tensor(np.array(hi)).to(params.device))
            elementwise = self._clip_cache[1:]
            clip = (float(lo.min()), float(hi.max()))
        o = clib.OptParams(kind=self.kind, step=self.step + 1, clip=int(clip is not None), reserved=0,
                           learning_rate=self.learning_rate, beta1=self.cfg['beta1'], beta2=self.cfg['beta2'],
                           epsilon=self.cfg['epsilon'], rho=self.cfg['rho'],
                           reg_l2_penalty=self.reg[0], reg_l1_penalty=self.reg[1],
                           reg_l2_decay=self.reg[2], reg_l1_decay=self.reg[3],
                           clip_lo=clip[0] if clip else 0.0, clip_hi=clip[1] if clip else 0.0)
        clib.check(libssnode.ssn_optimizer_step(params.data_ptr(), grads.data_ptr(), self._state[0].data_ptr(),
                                                self._state[1].data_ptr(), params.numel(), ctypes.byref(o), _stream()),
                   'ssn_optimizer_step')
        self.step += 1                    # (after the launch was accepted)
        if elementwise is not None:
            torch.minimum(torch.maximum(params, elementwise[0], out=params), elementwise[1], out=params)
