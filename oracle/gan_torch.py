"""ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

fp64 torch-on-CPU restatement of the BPTT conditional-WGAN path of the reference
(the reference builds these as Theano 0.9 / Lasagne graphs, which cannot be
imported here -- SURVEY.md section 8c).  Every function cites the reference lines it
follows (paths relative to /root/reference/tc_gan/).  Gradients come from torch
autograd on THIS restatement and are cross-checked by finite differences in
tests/test_oracle_gan.py.

Parity status:
  * forward (stimulus, W, Euler recurrence, time average) -- PINNED through the
    reference's own cross-test design (networks/tests/test_euler_ssn.py:29-86): the
    time average at long seqlen equals the fixed point of the reference's C solver
    (tests/test_oracle_gan.py::test_time_avg_matches_reference_c_fixed_point).
  * probe indexing -- PINNED by the quenched values of
    networks/tests/test_conditional_prober.py:18-86.
  * dynamics/rate penalties, critic, WGAN-GP, BPTT gradients, Adam/RMSprop steps --
    PARITY UNPINNED against the reference (it holds no value tests for them); pinned
    only against this restatement and finite differences.  Third-party semantics
    assumed (not in /root/reference): Lasagne@20efd95 `CustomRecurrentLayer` (zero
    initial state, output t = state after step t+1), `DenseLayer` (x.W + b),
    `BatchNormLayer` epsilon 1e-4, `updates.adam` (eps 1e-8, bias-corrected step
    size), `updates.rmsprop` (rho 0.9, eps 1e-6), `init.GlorotUniform`.
"""
import numpy as np
import torch

DT = torch.float64


def t64(x):
    return torch.as_tensor(np.asarray(x, dtype='float64'))


# ------------------------------------------------------------------ generator pieces
def stimulus(bandwidths, contrasts, smoothness, num_sites):
    """networks/ssn.py:167-188.  bandwidths, contrasts: (B, NB) -> (B, NB, 2N)."""
    b = t64(bandwidths)[..., None] if not torch.is_tensor(bandwidths) else bandwidths[..., None]
    c = t64(contrasts)[..., None] if not torch.is_tensor(contrasts) else contrasts[..., None]
    x = torch.linspace(-0.5, 0.5, num_sites, dtype=DT).reshape(1, 1, -1)

    def sigm(u):
        return 1 / (1 + torch.exp(-u / smoothness))

    stim = c * sigm(x + b / 2) * sigm(b / 2 - x)
    return torch.cat([stim, stim], dim=-1)


def make_W(z, J, D, S, N):
    """gradient_expressions/make_w_batch.py:8-34.  z: (B, 2N, 2N); J, D, S: (2, 2) tensors."""
    sign = torch.tensor([[1.0, -1.0], [1.0, -1.0]], dtype=DT)
    j = (sign * J).reshape(1, 2, 1, 2, 1)
    d = (sign * D).reshape(1, 2, 1, 2, 1)
    s = S.reshape(1, 2, 1, 2, 1)
    zz = z.reshape(-1, 2, N, 2, N)
    x = torch.linspace(-0.5, 0.5, N, dtype=DT).reshape(1, -1)
    xx = (x - x.T).reshape(1, 1, N, 1, N)
    wnn = torch.exp(-xx ** 2 / (2 * s ** 2))
    return (wnn * (j + d * zz)).reshape(-1, 2 * N, 2 * N)


def io_fun(v, io_type, k=0.01, n=2.2, r0=200.0, r1=1000.0):
    """ssnode.py:129-149 + 276-292 (clip / where forms, as Theano evaluates them)."""
    v0 = (r0 / k) ** (1 / n)
    if io_type == 'asym_power':
        return k * torch.clamp(v, min=0) ** n
    vc = torch.clamp(v, 0, v0)
    r_pow = k * vc ** n
    if io_type == 'asym_linear':
        lin = k * (v0 ** (n - 1)) * n * (v - v0)
        return torch.where(v <= v0, r_pow, r_pow + lin)
    if io_type == 'asym_tanh':
        r_tanh = r0 + (r1 - r0) * torch.tanh(n * r0 / (r1 - r0) * (v - v0) / v0)
        return torch.where(v <= v0, r_pow, r_tanh)
    raise ValueError(io_type)


def euler_ssn(W, ext, io_type, k, n, tau_E, tau_I, dt, seqlen, skip_steps, rate_penalty_threshold,
              return_trajectory=False):
    """networks/ssn.py:555-576 (step) + 598-633 (reductions).

    W: (B, M, M) (NOT transposed; the reference keeps Wt and computes batched_dot(r, Wt)),
    ext: (B, NB, M).  r_0 = 0; trajectory[t] = state after t+1 steps;
    rs = trajectory[skip_steps:].  Returns time_avg (B, NB, M), dynamics_penalty, rate_penalty.
    """
    B, NB, M = ext.shape
    N = M // 2
    tau = torch.cat([torch.full((N,), float(tau_E), dtype=DT), torch.full((N,), float(tau_I), dtype=DT)])
    eps = (dt / tau).reshape(1, 1, -1)
    r = torch.zeros((B, NB, M), dtype=DT)
    traj = []
    Wt = W.transpose(1, 2)
    for _ in range(seqlen):
        u = torch.bmm(r, Wt) + ext
        r = (1 - eps) * r + eps * io_fun(u, io_type, k, n)
        traj.append(r)
    rates = torch.stack(traj, dim=1)            # (B, T, NB, M)
    rs = rates[:, skip_steps:]
    time_avg = rs.mean(dim=1)
    dynamics_penalty = ((rs[:, 1:] - rs[:, :-1]) ** 2).mean()
    rate_penalty = torch.relu(rs - rate_penalty_threshold).mean()
    if return_trajectory:
        return time_avg, dynamics_penalty, rate_penalty, rates
    return time_avg, dynamics_penalty, rate_penalty


def probes_from_norm(norm_probes, cell_types, num_sites):
    """cwgan.py:91-93 + gradient_expressions/utils.py:23-24 (truncating cast to uint16)."""
    p = ((np.asarray(norm_probes, dtype='float64') + 1) * (num_sites - 1) / 2).astype('uint16')
    return p.astype(np.int64) + np.asarray(cell_types).astype(np.int64) * num_sites


def conditional_probe(time_avg, model_ids, probes):
    """cwgan.py:98: time_avg[model_ids, :, probes] -> (batch, NB)."""
    return time_avg[torch.as_tensor(model_ids, dtype=torch.long), :, torch.as_tensor(probes, dtype=torch.long)]


# ------------------------------------------------------------------ critic
def layer_norm(x, eps=1e-4):
    """simple_discriminator.py:6-48 on top of Lasagne BatchNormLayer (beta=gamma=None, axes=(1,),
    batch statistics always): (x - mean) / sqrt(var + eps) over the feature axis (biased variance)."""
    mean = x.mean(dim=1, keepdim=True)
    var = x.var(dim=1, unbiased=False, keepdim=True)
    return (x - mean) / torch.sqrt(var + eps)


ACTIVATIONS = {     # lasagne.nonlinearities by name (leaky_rectify = LeakyRectify(0.01), very_leaky_rectify = LeakyRectify(1 / 3))
    'rectify': torch.relu, 'tanh': torch.tanh, 'linear': lambda t: t, 'identity': lambda t: t,
    'leaky_rectify': lambda t: torch.nn.functional.leaky_relu(t, 0.01),
    'very_leaky_rectify': lambda t: torch.nn.functional.leaky_relu(t, 1.0 / 3.0),
    'sigmoid': torch.sigmoid,
    'softplus': torch.nn.functional.softplus,                # theano.tensor.nnet.softplus: log1p(exp(x))
    'elu': torch.nn.functional.elu,                          # switch(x > 0, x, expm1(x))
}


def critic_layer_scales(normalization, nonlinearity, nlayers, use_scale='auto'):
    """simple_discriminator.py:57-75: a layer-normalised layer has a ScaleLayer iff `use_scale` (auto: every nonlinearity but
    rectify).  Returns one bool per hidden layer."""
    norms = normalization if isinstance(normalization, (list, tuple)) else [normalization] * nlayers
    auto = nonlinearity != 'rectify'
    return [n == 'layer' and (auto if use_scale == 'auto' else bool(use_scale)) for n in norms]


def critic_forward(params, x, cond, normalization='none', nonlinearity='rectify', use_scale='auto'):
    """cwgan.py:123-175 + simple_discriminator.py:51-75, 139-165.

    params: list of tensors in lasagne's get_all_params order.  'none' layer: W, b;  'layer': W, [scales,] b with
    Dense(no bias) -> LayerNorm -> [ScaleLayer] -> Bias -> nonlinearity; last: Wout.
    Input = concat(x, [contrast, |norm_probe|, cell_type]) (cwgan.py:164-170)."""
    if cond is None:           # UnConditionalDiscriminator (wgan.py:66-97): the tuning curve alone
        h = x
    else:
        c = torch.stack([cond[:, 0], cond[:, 1].abs(), cond[:, 2]], dim=1)
        h = torch.cat([x, c], dim=1)
    act = ACTIVATIONS[nonlinearity]
    # number of hidden layers from the parameter count: 2 per layer + 1 per scaled layer + the output W
    nl = 0
    while True:
        scales = critic_layer_scales(normalization if not isinstance(normalization, (list, tuple)) else list(normalization)[:nl],
                                     nonlinearity, nl, use_scale)
        if 2 * nl + sum(scales) + 1 == len(params):
            break
        nl += 1
        assert nl < 64, 'parameter list does not match the layer description'
    norms = normalization if isinstance(normalization, (list, tuple)) else [normalization] * nl
    it = iter(params)
    for l in range(nl):
        pre = h @ next(it)
        if norms[l] == 'layer':
            pre = layer_norm(pre)
            if scales[l]:
                pre = pre * next(it)
        h = act(pre + next(it))
    return h @ next(it)          # (batch, 1), linear, no bias


def critic_loss(params, xg, xd, xp, cg, cd, cp, lmd, **kw):
    """cwgan.py:190-214: mean D(xg) - mean D(xd) + lmd * mean((||dD(xp)/dxp||_2 - 1)^2)."""
    xp = xp.clone().requires_grad_(True)
    dp = critic_forward(params, xp, cp, **kw)[:, 0]
    g, = torch.autograd.grad(dp.sum(), xp, create_graph=True)
    penalty = ((g.norm(2, dim=1) - 1) ** 2).mean()
    return critic_forward(params, xg, cg, **kw).mean() - critic_forward(params, xd, cd, **kw).mean() + lmd * penalty


def glorot_uniform(rng, fan_in, fan_out):
    """Lasagne init.GlorotUniform(gain=1): U(-a, a), a = sqrt(6 / (fan_in + fan_out))."""
    a = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-a, a, size=(fan_in, fan_out))


# ------------------------------------------------------------------ generator loss
def generator_loss(J, D, S, z, bandwidths, contrasts, model_ids, norm_probes, cell_types, critic_params,
                   num_sites, smoothness, io_type, k, n, tau_E, tau_I, dt, seqlen, skip_steps,
                   rate_penalty_threshold, dynamics_cost, rate_cost, critic_kw=None, V=None, zs_in=None):
    """wgan.py:236-241: -mean D(G(z)) + dynamics_cost * dyn_pen + rate_cost * rate_pen, with the
    conditional generator output of cwgan.py:107-120 (conditions = contrast, norm_probe, cell_type)."""
    critic_kw = critic_kw or {}
    ext = stimulus(bandwidths, contrasts, smoothness, num_sites)
    if V is not None:
        # networks/ssn.py:679-686: stimulus * (1 + v_pop z_in); V of shape (2,) ('heteroin') or () ('deg-heteroin')
        vpop = V if V.dim() == 1 else torch.stack([V, V])
        vs = torch.cat([vpop[0].expand(num_sites), vpop[1].expand(num_sites)])
        ext = (1 + vs.reshape(1, 1, -1) * t64(zs_in)[:, None, :]) * ext
    W = make_W(z, J, D, S, num_sites)
    ta, dyn, rate = euler_ssn(W, ext, io_type, k, n, tau_E, tau_I, dt, seqlen, skip_steps, rate_penalty_threshold)
    probes = probes_from_norm(norm_probes, cell_types, num_sites)
    tc = conditional_probe(ta, model_ids, probes)
    con = t64(contrasts)[torch.as_tensor(model_ids, dtype=torch.long), 0]
    cond = torch.stack([con, t64(norm_probes), t64(np.asarray(cell_types, dtype='float64'))], dim=1)
    loss = -critic_forward(critic_params, tc, cond, **critic_kw).mean() + dynamics_cost * dyn + rate_cost * rate
    return loss, dict(time_avg=ta, dynamics_penalty=dyn, rate_penalty=rate, tuning_curve=tc, conditions=cond)


def fixed_probe(time_avg, probes):
    """ssn.py:846-848 (`FixedProber`): time_avg[:, :, probes] flattened over (stimulus, probe) -> (batch, NB * P)."""
    return time_avg[:, :, torch.as_tensor(np.asarray(probes), dtype=torch.long)].reshape(time_avg.shape[0], -1)


def unconditional_generator_loss(J, D, S, z, bandwidths, contrasts, probes, critic_params, num_sites, smoothness, io_type,
                                 k, n, tau_E, tau_I, dt, seqlen, skip_steps, rate_penalty_threshold, dynamics_cost,
                                 rate_cost, critic_kw=None, V=None, zs_in=None):
    """wgan.py:236-241 with the unconditional pieces of wgan.py:299-444: stimulus grid of `grid_stimulator_inputs`
    (bandwidths, contrasts: (B, NB) arrays), fixed prober, critic without condition columns."""
    critic_kw = critic_kw or {}
    ext = stimulus(bandwidths, contrasts, smoothness, num_sites)
    if V is not None:
        vpop = V if V.dim() == 1 else torch.stack([V, V])
        vs = torch.cat([vpop[0].expand(num_sites), vpop[1].expand(num_sites)])
        ext = (1 + vs.reshape(1, 1, -1) * t64(zs_in)[:, None, :]) * ext
    W = make_W(z, J, D, S, num_sites)
    ta, dyn, rate = euler_ssn(W, ext, io_type, k, n, tau_E, tau_I, dt, seqlen, skip_steps, rate_penalty_threshold)
    tc = fixed_probe(ta, probes)
    loss = -critic_forward(critic_params, tc, None, **critic_kw).mean() + dynamics_cost * dyn + rate_cost * rate
    return loss, dict(time_avg=ta, dynamics_penalty=dyn, rate_penalty=rate, tuning_curve=tc)


# ------------------------------------------------------------------ optimizers (Lasagne defaults)
def adam_step(p, g, state, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """lasagne.updates.adam: t += 1; a_t = lr*sqrt(1-b2^t)/(1-b1^t); m = b1 m + (1-b1) g;
    v = b2 v + (1-b2) g^2; p -= a_t * m / (sqrt(v) + eps).  'adam-wgan' = beta1 .5, beta2 .9
    (wgan.py:113-117)."""
    state['t'] = state.get('t', 0) + 1
    t = state['t']
    m = state.get('m', np.zeros_like(p)); v = state.get('v', np.zeros_like(p))
    a_t = lr * np.sqrt(1 - beta2 ** t) / (1 - beta1 ** t)
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    state['m'], state['v'] = m, v
    return p - a_t * m / (np.sqrt(v) + eps)


def rmsprop_step(p, g, state, lr, rho=0.9, eps=1e-6):
    """lasagne.updates.rmsprop: a = rho a + (1-rho) g^2; p -= lr * g / sqrt(a + eps)."""
    a = state.get('a', np.zeros_like(p))
    a = rho * a + (1 - rho) * g * g
    state['a'] = a
    return p - lr * g / np.sqrt(a + eps)


def sgd_step(p, g, state, lr):
    return p - lr * g


# ------------------------------------------------------------------ moment matching
def sample_moments(samples):
    """moment_matching.py:91-104: (sample_size, channels) -> (2, channels): mean and POPULATION variance."""
    return torch.stack([samples.mean(dim=0), samples.var(dim=0, unbiased=False)])


def moment_weights(data, moment_weight_type='mean', regularization=1e-3, lam=1.0):
    """`BPTTMomentMatcher.set_dataset` (moment_matching.py:348-367) -> (data_moments, weights), numpy fp64."""
    data = np.asarray(data, dtype='float64')
    dm = np.asarray([data.mean(axis=0), data.var(axis=0)])
    num = np.broadcast_to([[1.0], [lam]], dm.shape)
    if moment_weight_type == 'mean':
        den = data.mean()
        w = num / np.array([[den ** 2], [den ** 4]])
    elif moment_weight_type == 'ew_mean':
        den = dm[0] + regularization
        w = num / np.array([den ** 2, den ** 4])
    elif moment_weight_type == 'ew_relative':
        w = num / (dm + regularization) ** 2
    else:
        raise ValueError(moment_weight_type)
    return dm, np.array(w)


def moment_matching_loss(J, D, S, z, bandwidths, contrasts, probes, data_moments, weights, num_sites, smoothness,
                         io_type, k, n, tau_E, tau_I, dt, seqlen, skip_steps, rate_penalty_threshold,
                         dynamics_cost, rate_cost):
    """`MMGeneratorTrainer.post_init` (moment_matching.py:232-243): mean(w * (data_moments - gen_moments)^2)
    + dynamics_cost * dynamics_penalty + rate_cost * rate_penalty, with the fixed prober of ssn.py:838-851
    (tuning_curve = time_avg[:, :, probes] flattened over (stimulus, probe))."""
    ext = stimulus(bandwidths, contrasts, smoothness, num_sites)
    W = make_W(z, J, D, S, num_sites)
    ta, dyn, rate = euler_ssn(W, ext, io_type, k, n, tau_E, tau_I, dt, seqlen, skip_steps, rate_penalty_threshold)
    tc = ta[:, :, torch.as_tensor(np.asarray(probes), dtype=torch.long)].reshape(ta.shape[0], -1)
    gm = sample_moments(tc)
    loss = (t64(weights) * (t64(data_moments) - gm) ** 2).mean() + dynamics_cost * dyn + rate_cost * rate
    return loss, dict(tuning_curve=tc, gen_moments=gm, dynamics_penalty=dyn, rate_penalty=rate)
