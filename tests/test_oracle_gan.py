"""Pin oracle/gan_torch.py (CPU): forward through the reference's C solver, probe indexing through
the reference's quenched values, gradients through finite differences."""
import numpy as np
import torch

from oracle import gan_torch as og
from oracle import ssn_numpy as on

P = on.DEFAULT_PARAMS
GEN = dict(io_type='asym_tanh', k=0.01, n=2.2, tau_E=10., tau_I=1., dt=0.1)     # wgan.py:39-55


def test_time_avg_matches_reference_c_fixed_point(oracle_lib):
    """Design of networks/tests/test_euler_ssn.py:29-76: long fixed-time run == ssnode fixed point
    (reference tolerance rtol=atol=5e-4 at seqlen 4000)."""
    N, B = 10, 2
    jds = on.new_JDS()
    zs, fps, counter = on.sample_fixed_points(B, seed=N * B, N=N, io_type='asym_tanh', atol=1e-10, **jds)
    assert not counter
    bw = np.tile(np.asarray(P['bandwidths'])[None], (B, 1))
    con = np.full_like(bw, 20.0)
    ext = og.stimulus(bw, con, P['smoothness'], N)
    W = og.make_W(og.t64(zs), *(og.t64(jds[k]) for k in 'JDS'), N)
    np.testing.assert_allclose(W[0].numpy(), on.generate_weight(N, jds['J'], jds['D'], jds['S'], zs[0]), rtol=1e-12, atol=1e-300)
    np.testing.assert_allclose(ext[0].numpy(), on.stimulus_input(P['bandwidths'], np.linspace(-.5, .5, N),
                                                                 P['smoothness'], [20.]), rtol=1e-13, atol=1e-300)
    seqlen = 4000
    ta, dyn, rate = og.euler_ssn(W, ext, seqlen=seqlen, skip_steps=seqlen - 1, rate_penalty_threshold=200., **GEN)
    np.testing.assert_allclose(ta.numpy(), fps, rtol=5e-4, atol=5e-4)


def test_conditional_probe_quenched_values():
    # networks/tests/test_conditional_prober.py:18-86
    norm_probes = np.array([-1, -0.5, 0, 0.5, 1] * 2, dtype='float32')
    cell_types = np.array([0] * 5 + [1] * 5, dtype='uint16')
    probes = og.probes_from_norm(norm_probes, cell_types, 201)
    np.testing.assert_array_equal(probes, [0, 50, 100, 150, 200, 201, 251, 301, 351, 401])
    model_ids = np.array([0, 1] * 5)
    shape = (2, 3, 402)
    time_avg = torch.arange(np.prod(shape), dtype=torch.float64).reshape(shape)
    tc = og.conditional_probe(time_avg, model_ids, probes).numpy()
    desired = [[0, 402, 804], [1256, 1658, 2060], [100, 502, 904], [1356, 1758, 2160], [200, 602, 1004],
               [1407, 1809, 2211], [251, 653, 1055], [1507, 1909, 2311], [351, 753, 1155], [1607, 2009, 2411]]
    np.testing.assert_array_equal(tc, desired)


def _small_problem(seed=0, B=3, N=6, NB=3, T=30, skip=20, width=8):
    rs = np.random.RandomState(seed)
    jds = on.new_JDS()
    z = og.t64(rs.rand(B, 2 * N, 2 * N))
    bw = np.tile(np.array([0.0625, 0.25, 1.0])[None, :NB], (B, 1))
    con = np.full((B, NB), 20.0)
    model_ids = np.array([0, 1, 2, 1])
    norm_probes = np.array([0.0, 0.5, -0.5, 0.0])
    cell_types = np.array([0, 1, 0, 1])
    nin = NB + 3
    params = [og.t64(og.glorot_uniform(rs, nin, width)), og.t64(rs.randn(width) * 0.01),
              og.t64(og.glorot_uniform(rs, width, width)), og.t64(rs.randn(width) * 0.01),
              og.t64(og.glorot_uniform(rs, width, 1))]
    kw = dict(num_sites=N, smoothness=P['smoothness'], seqlen=T, skip_steps=skip, rate_penalty_threshold=5.0,
              dynamics_cost=1.0, rate_cost=0.01, **GEN)
    return jds, z, bw, con, model_ids, norm_probes, cell_types, params, kw


def test_generator_gradient_finite_differences():
    jds, z, bw, con, mid, nprobe, ctype, params, kw = _small_problem()
    J, D, S = (og.t64(jds[k]).clone().requires_grad_(True) for k in 'JDS')
    loss, aux = og.generator_loss(J, D, S, z, bw, con, mid, nprobe, ctype, params, **kw)
    gJ, gD, gS = torch.autograd.grad(loss, [J, D, S])
    assert float(aux['dynamics_penalty']) > 0 and float(aux['rate_penalty']) > 0
    h = 1e-6
    for name, g in (('J', gJ), ('D', gD), ('S', gS)):
        for idx in ((0, 0), (0, 1), (1, 0), (1, 1)):
            vals = []
            for sgn in (+1, -1):
                q = {k: og.t64(jds[k]).clone() for k in 'JDS'}
                q[name][idx] += sgn * h
                vals.append(float(og.generator_loss(q['J'], q['D'], q['S'], z, bw, con, mid, nprobe, ctype, params, **kw)[0]))
            fd = (vals[0] - vals[1]) / (2 * h)
            np.testing.assert_allclose(float(g[idx]), fd, rtol=2e-5, atol=1e-9)


def test_critic_gp_gradient_finite_differences():
    rs = np.random.RandomState(1)
    batch, nb, width = 6, 4, 7
    params = [og.t64(og.glorot_uniform(rs, nb + 3, width)), og.t64(rs.randn(width) * 0.1),
              og.t64(og.glorot_uniform(rs, width, width)), og.t64(rs.randn(width) * 0.1),
              og.t64(og.glorot_uniform(rs, width, 1))]
    xg, xd = og.t64(rs.rand(batch, nb) * 5), og.t64(rs.rand(batch, nb) * 5)
    e = og.t64(rs.rand(batch, 1))
    xp = e * xd + (1 - e) * xg
    cond = og.t64(np.stack([np.full(batch, 20.), rs.rand(batch) - .5, rs.randint(0, 2, batch)], axis=1))
    for norm in ('none', 'layer'):
        ps = [p.clone().requires_grad_(True) for p in params]
        loss = og.critic_loss(ps, xg, xd, xp, cond, cond, cond, 10.0, normalization=norm)
        grads = torch.autograd.grad(loss, ps)
        h = 1e-6
        for li, (p, g) in enumerate(zip(params, grads)):
            flat = p.reshape(-1)
            for j in (0, flat.numel() // 2, flat.numel() - 1):
                vals = []
                for sgn in (+1, -1):
                    q = [t.clone() for t in params]
                    q[li].reshape(-1)[j] += sgn * h
                    vals.append(float(og.critic_loss(q, xg, xd, xp, cond, cond, cond, 10.0, normalization=norm)))
                np.testing.assert_allclose(float(g.reshape(-1)[j]), (vals[0] - vals[1]) / (2 * h), rtol=1e-4, atol=1e-7)


def test_optimizer_steps_against_closed_forms():
    p = np.array([1.0, -2.0]); g = np.array([0.5, -0.25])
    st = {}
    p1 = og.adam_step(p, g, st, lr=0.01, beta1=0.5, beta2=0.9)
    # first Adam step moves every coordinate by lr * sign(g) (up to eps)
    np.testing.assert_allclose(p1, p - 0.01 * np.sign(g), rtol=1e-6)
    st = {}
    p2 = og.rmsprop_step(p, g, st, lr=0.01)
    np.testing.assert_allclose(p2, p - 0.01 * g / np.sqrt(0.1 * g * g + 1e-6))


def test_oracle_critic_scale_layers_and_smooth_nonlinearities():
    """oracle/gan_torch.critic_forward in its general form against a direct numpy restatement of
    simple_discriminator.py:57-75 (Dense(no bias) -> LayerNorm eps 1e-4 -> ScaleLayer -> Bias -> nonlinearity):
    parameter order W, scales, b; the ScaleLayer only for non-rectify nonlinearities (`use_scale='auto'`)."""
    import torch
    from oracle import gan_torch as og
    rs = np.random.RandomState(0)
    x = rs.rand(5, 4)
    W1, s1, b1, Wo = rs.randn(4, 6), rs.uniform(0.5, 1.5, 6), rs.randn(6) * 0.1, rs.randn(6, 1)
    a = x @ W1
    y = (a - a.mean(1, keepdims=True)) / np.sqrt(a.var(1, keepdims=True) + 1e-4)
    for name, f in (('tanh', np.tanh), ('sigmoid', lambda t: 1 / (1 + np.exp(-t))), ('softplus', lambda t: np.log1p(np.exp(t))),
                    ('elu', lambda t: np.where(t > 0, t, np.expm1(t)))):
        want = f(y * s1 + b1) @ Wo
        got = og.critic_forward([og.t64(p) for p in (W1, s1, b1, Wo)], og.t64(x), None, normalization='layer', nonlinearity=name)
        np.testing.assert_allclose(got.numpy(), want, rtol=1e-12, atol=1e-12)
    want = np.maximum(y + b1, 0) @ Wo                     # rectify: no scale
    got = og.critic_forward([og.t64(p) for p in (W1, b1, Wo)], og.t64(x), None, normalization='layer', nonlinearity='rectify')
    np.testing.assert_allclose(got.numpy(), want, rtol=1e-12, atol=1e-12)
    assert og.critic_layer_scales(['none', 'layer'], 'tanh', 2) == [False, True]
    assert og.critic_layer_scales('layer', 'rectify', 2) == [False, False]
