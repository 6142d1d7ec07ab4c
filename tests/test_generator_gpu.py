"""GPU parity of the fixed-time generator (forward reductions, BPTT adjoint, J/D/S gradients)
against the fp64 torch restatement oracle/gan_torch.py."""
import numpy as np
import pytest
import torch

from oracle import gan_torch as og
from oracle import ssn_numpy as on

pytestmark = pytest.mark.gpu
P = on.DEFAULT_PARAMS
GEN = dict(io_type='asym_tanh', k=0.01, n=2.2, tau_E=10., tau_I=1., dt=0.1)


def _problem(N, B, NB, seed, T, skip, theta):
    rs = np.random.RandomState(seed)
    jds = on.new_JDS()
    z = rs.rand(B, 2 * N, 2 * N)
    bws = np.tile(np.resize(np.asarray(P['bandwidths']), NB)[None, :] * (1 + 0.05 * (np.arange(NB) // 8)), (B, 1))
    con = np.tile(rs.choice([5., 20.], size=(B, 1)), (1, NB))
    return jds, z, bws, con


@pytest.mark.parametrize('io_type', ['asym_tanh', 'asym_power', 'asym_linear'])
@pytest.mark.parametrize('N,B,NB,dtype,kernel', [
    (10, 3, 8, 'float64', 0), (50, 2, 4, 'float32', 1), (100, 2, 3, 'float32', 0), (102, 2, 1, 'float32', 0),
    (26, 3, 5, 'float64', 0),
    # fp32 with NB >= 4: the MFMA kernels (kernel 0 = library default = 2 here), tile kernels forced with 1
    (50, 2, 4, 'float32', 2), (100, 2, 8, 'float32', 0), (100, 1, 8, 'float32', 1), (101, 2, 9, 'float32', 2),
    (76, 1, 11, 'float32', 0), (10, 3, 8, 'float32', 2), (33, 2, 5, 'float32', 2), (104, 1, 4, 'float32', 2),
    # K-split operand layout with a partly filled ladder size (2N = 180 in the 200 build, 2N = 130 in the 152 build)
    (90, 2, 8, 'float32', 2), (65, 1, 6, 'float32', 2),
    # sizes beyond the register-resident instantiations (2N > 208 fp32, > 104 fp64): streaming kernels
    (110, 1, 2, 'float32', 0), (60, 2, 2, 'float64', 0), (129, 1, 4, 'float32', 0), (201, 1, 2, 'float32', 0),
    # MFMA kernels with one 4-stimulus group per workgroup (kernel 3)
    (100, 2, 8, 'float32', 3), (101, 1, 6, 'float32', 3), (50, 2, 9, 'float32', 3),
    # fp16-split MFMA kernel (csrc/ssn_mfma16.hip; saturating I/O function only), two groups per workgroup (4) and one (5):
    # the three tile grids (2N <= 104, <= 152, <= 208), odd sizes, ragged stimulus groups
    (100, 2, 8, 'float32', 4), (101, 1, 6, 'float32', 5), (50, 2, 9, 'float32', 5), (76, 1, 11, 'float32', 4),
    (10, 3, 8, 'float32', 4), (33, 2, 5, 'float32', 4), (104, 1, 4, 'float32', 4), (90, 2, 8, 'float32', 4),
    (65, 1, 6, 'float32', 4), (102, 2, 8, 'float32', 4), (52, 2, 8, 'float32', 5), (100, 2, 8, 'float32', 5),
    # 6: two groups in the alternating form (state as three fp16 parts, exact); 4 is the wide form (two parts)
    (100, 2, 8, 'float32', 6), (76, 1, 11, 'float32', 6), (33, 2, 5, 'float32', 6), (101, 1, 6, 'float32', 6),
    # 8: two draws per workgroup, every wave both roles (csrc/ssn_duo.hip): odd and even unit counts, ragged stimulus
    # groups, all three tile grids
    (100, 2, 8, 'float32', 8), (100, 3, 8, 'float32', 8), (101, 1, 6, 'float32', 8), (76, 1, 11, 'float32', 8),
    (10, 3, 8, 'float32', 8), (33, 2, 5, 'float32', 8), (104, 1, 4, 'float32', 8), (90, 2, 8, 'float32', 8),
    (65, 1, 6, 'float32', 8), (102, 2, 8, 'float32', 8), (52, 5, 8, 'float32', 8), (50, 2, 9, 'float32', 8)])
def test_forward_reductions_vs_oracle(io_type, N, B, NB, dtype, kernel):
    from tc_gan_amd import genops, stimuli, weight_gen
    if kernel in (4, 5, 6, 8) and io_type != 'asym_tanh':
        pytest.skip('the fp16-split kernel needs the rate bound of asym_tanh (refusal: test_split_kernel_refuses_...)')
    T, skip, theta = 60, 40, 2.0
    jds, z, bws, con = _problem(N, B, NB, N + NB, T, skip, theta)
    gen = dict(GEN, io_type=io_type)
    ext_o = og.stimulus(bws, con, P['smoothness'], N)
    W_o = og.make_W(og.t64(z), *(og.t64(jds[k]) for k in 'JDS'), N)
    ta_o, dyn_o, rate_o, traj_o = og.euler_ssn(W_o, ext_o, seqlen=T, skip_steps=skip, rate_penalty_threshold=theta,
                                                return_trajectory=True, **gen)
    W = weight_gen.generate_weight_batch(N, jds['J'], jds['D'], jds['S'], z, dtype=dtype)
    ext = stimuli.stimulus_batch(bws, con, P['smoothness'], N, dtype=dtype)
    gp = genops.make_gen_params(seqlen=T, skip_steps=skip, rate_penalty_threshold=theta, kernel=kernel, **gen)
    out = genops.gen_forward(W, ext, gp, save=True)
    rtol = 1e-4 if dtype == 'float32' else 1e-9
    np.testing.assert_allclose(out['time_avg'].cpu().numpy(), ta_o.numpy(), rtol=rtol, atol=rtol * 1e-2)
    np.testing.assert_allclose(float(out['dynamics_penalty']), float(dyn_o), rtol=10 * rtol)
    np.testing.assert_allclose(float(out['rate_penalty']), float(rate_o), rtol=10 * rtol, atol=1e-12)
    # trajectory layout [B][NB][T][M] vs oracle (B, T, NB, M)
    np.testing.assert_allclose(out['traj'].cpu().numpy(), traj_o.permute(0, 2, 1, 3).numpy(), rtol=rtol, atol=rtol * 1e-2)


@pytest.mark.parametrize('io_type', ['asym_tanh', 'asym_power'])
@pytest.mark.parametrize('N,B,NB,dtype,kernel', [
    (6, 3, 3, 'float64', 0), (20, 2, 8, 'float64', 0), (50, 2, 2, 'float32', 0), (100, 2, 2, 'float32', 0),
    # fp32 with NB >= 4: MFMA forward + adjoint kernels (2), tile kernels (1)
    (100, 1, 8, 'float32', 2), (100, 1, 8, 'float32', 1), (50, 2, 5, 'float32', 2), (101, 1, 4, 'float32', 2),
    (20, 2, 9, 'float32', 2), (76, 1, 8, 'float32', 2), (90, 1, 5, 'float32', 2),
    (110, 1, 2, 'float32', 0), (60, 2, 2, 'float64', 0), (100, 1, 8, 'float32', 3), (76, 2, 5, 'float32', 3),
    (201, 1, 1, 'float32', 0),
    # trajectory-saving forward and adjoint sweep on the fp16-split MFMA kernels
    (100, 1, 8, 'float32', 4), (76, 2, 5, 'float32', 5), (101, 1, 8, 'float32', 4), (50, 2, 9, 'float32', 4),
    (33, 1, 4, 'float32', 5), (104, 1, 6, 'float32', 4), (100, 1, 8, 'float32', 6),
    # trajectory-saving forward and adjoint sweep in the two-draw form (8; adjoint: any I/O function)
    (100, 1, 8, 'float32', 8), (100, 3, 8, 'float32', 8), (76, 2, 5, 'float32', 8), (101, 1, 8, 'float32', 8),
    (50, 2, 9, 'float32', 8), (33, 1, 4, 'float32', 8), (104, 2, 6, 'float32', 8)])
def test_bptt_gradients_vs_oracle(io_type, N, B, NB, dtype, kernel):
    """dL/dJ, dL/dD, dL/dS for L = sum(G * time_avg) + c_d * dyn_pen + c_r * rate_pen."""
    from tc_gan_amd import genops, stimuli, weight_gen
    if kernel in (4, 5, 6, 8) and io_type != 'asym_tanh':
        pytest.skip('the fp16-split kernel needs the rate bound of asym_tanh')
    T, skip, theta = 50, 30, 1.0
    jds, z, bws, con = _problem(N, B, NB, 7 * N + NB, T, skip, theta)
    gen = dict(GEN, io_type=io_type)
    rs = np.random.RandomState(5)
    G = rs.randn(B, NB, 2 * N) * (rs.rand(B, NB, 2 * N) < 0.1)       # sparse, like a probe gather
    dyn_cost, rate_cost = 1.0, 0.01
    # oracle
    J, D, S = (og.t64(jds[k]).clone().requires_grad_(True) for k in 'JDS')
    ext_o = og.stimulus(bws, con, P['smoothness'], N)
    W_o = og.make_W(og.t64(z), J, D, S, N)
    ta_o, dyn_o, rate_o = og.euler_ssn(W_o, ext_o, seqlen=T, skip_steps=skip, rate_penalty_threshold=theta, **gen)
    loss = (og.t64(G) * ta_o).sum() + dyn_cost * dyn_o + rate_cost * rate_o
    gJ_o, gD_o, gS_o = torch.autograd.grad(loss, [J, D, S])
    # device
    zt = torch.as_tensor(z).to('cuda', getattr(torch, dtype))
    W = weight_gen.generate_weight_batch(N, jds['J'], jds['D'], jds['S'], zt, dtype=dtype)
    ext = stimuli.stimulus_batch(bws, con, P['smoothness'], N, dtype=dtype)
    gp = genops.make_gen_params(seqlen=T, skip_steps=skip, rate_penalty_threshold=theta, kernel=kernel, **gen)
    out = genops.gen_forward(W, ext, gp, save=True)
    Gd = torch.as_tensor(G).to('cuda', getattr(torch, dtype))
    delta = genops.gen_backward(W, out['traj'], out['df'], Gd, dyn_cost / out['n_dyn'], rate_cost / out['n_rate'], gp)
    gW = genops.weight_grad(delta, out['traj'])
    gJ, gD, gS = genops.jds_grad(gW, zt, jds['J'], jds['D'], jds['S'])
    rtol = 2e-3 if dtype == 'float32' else 1e-8
    for got, want in ((gJ, gJ_o), (gD, gD_o), (gS, gS_o)):
        want = want.numpy()
        np.testing.assert_allclose(got, want, rtol=rtol, atol=rtol * np.abs(want).max())


_HORIZON_ORACLE = {}


def _horizon_oracle(shape):
    """fp64 autograd of oracle/gan_torch.py at a production horizon (cached per shape: seconds on the CPU)."""
    if shape in _HORIZON_ORACLE:
        return _HORIZON_ORACLE[shape]
    if shape == 'c3':          # BASELINE config 3: 2N = 200, 8 stimuli, seqlen 1200 / skip 1000, tau_E = 10
        N, B, NB, T, skip, theta, gen, v = 100, 2, 8, 1200, 1000, 5.0, dict(GEN), None
        costs = (1.0, 0.01)
    else:                      # the paper's run (scripts/fig4/gan/run.json): 2N = 202, tau_E = 2, 240 / 200, deg-heteroin
        N, B, NB, T, skip, theta, gen, v = 101, 3, 8, 240, 200, 5.0, dict(GEN, tau_E=2.), 0.1
        costs = (0.0, 100.0)
    jds, z, bws, con = _problem(N, B, NB, 11 * N + NB, T, skip, theta)
    rs = np.random.RandomState(17)
    G = rs.randn(B, NB, 2 * N) * (rs.rand(B, NB, 2 * N) < 0.1)
    zin = rs.choice(2, (B, 2 * N)) * 2.0 - 1.0
    J, D, S = (og.t64(jds[k]).clone().requires_grad_(True) for k in 'JDS')
    V = None if v is None else og.t64(v).clone().requires_grad_(True)
    base = og.stimulus(bws, con, P['smoothness'], N)
    ext_o = base if v is None else (1 + V * og.t64(zin)[:, None, :]) * base             # networks/ssn.py:679-686
    W_o = og.make_W(og.t64(z), J, D, S, N)
    W_o.retain_grad()
    ta_o, dyn_o, rate_o = og.euler_ssn(W_o, ext_o, seqlen=T, skip_steps=skip, rate_penalty_threshold=theta, **gen)
    loss = (og.t64(G) * ta_o).sum() + costs[0] * dyn_o + costs[1] * rate_o
    loss.backward()
    res = dict(N=N, B=B, NB=NB, T=T, skip=skip, theta=theta, gen=gen, v=v, costs=costs, jds=jds, z=z, bws=bws, con=con,
               G=G, zin=zin, gW=W_o.grad.numpy(), gJ=J.grad.numpy(), gD=D.grad.numpy(), gS=S.grad.numpy(),
               gV=None if v is None else float(V.grad), ta=ta_o.detach().numpy())
    _HORIZON_ORACLE[shape] = res
    return res


@pytest.mark.parametrize('shape,kernel', [('c3', 2), ('c3', 6), ('c3', 8), ('c3', 4), ('paper', 5), ('paper', 2), ('paper', 8)])
def test_bptt_gradients_vs_oracle_at_production_horizons(shape, kernel):
    """`test_bptt_gradients_vs_oracle` stops at 50 steps; the adjoint of the fp16-split sweeps carries a LAGGED power-of-two
    scale and hands max |delta| per draw to the fp16 form of dL/dW -- both are only exercised by a long sweep through the
    transient (delta grows by orders of magnitude between the penalty window and the first steps).  Here: the C3 horizon
    (2N = 200, 8 stimuli, seqlen 1200 / skip 1000) and the paper's (2N = 202, tau_E = 2, 240 / 200, deg-heteroin input), a few
    draws, dL/dW itself AND dL/d(J, D, S[, V]) against fp64 autograd of oracle/gan_torch.py (reference semantics:
    networks/wgan.py:236-242, networks/ssn.py:566-576, 598-633), on both dL/dW forms (three bf16 parts; two fp16 parts
    under the handed-over bound).  Tolerances: dL/dW relative to the largest element of the draw 1e-4 (measured: see
    DESIGN 1); parameter gradients 2e-3 of the largest (fp32 sums over 2.4e6 products)."""
    from tc_gan_amd import genops, stimuli
    from tc_gan_amd import weight_gen
    o = _horizon_oracle(shape)
    N, B, NB = o['N'], o['B'], o['NB']
    jds = o['jds']
    zt = torch.as_tensor(o['z']).to('cuda', torch.float32)
    W = weight_gen.generate_weight_batch(N, jds['J'], jds['D'], jds['S'], zt, dtype='float32')
    base = stimuli.stimulus_batch(o['bws'], o['con'], P['smoothness'], N, dtype='float32')
    zin = torch.as_tensor(o['zin']).to('cuda', torch.float32)
    ext = base if o['v'] is None else ((1 + o['v'] * zin)[:, None, :] * base).contiguous()
    gp = genops.make_gen_params(seqlen=o['T'], skip_steps=o['skip'], rate_penalty_threshold=o['theta'], kernel=kernel, **o['gen'])
    out = genops.gen_forward(W, ext, gp, save=True)
    np.testing.assert_allclose(out['time_avg'].cpu().numpy(), o['ta'], rtol=1e-4, atol=1e-5)
    Gd = torch.as_tensor(o['G']).to('cuda', torch.float32)
    delta, g_ext, dmax = genops.gen_backward(W, out['traj'], out['df'], Gd, o['costs'][0] / out['n_dyn'],
                                             o['costs'][1] / out['n_rate'], gp, want_g_ext=True, want_dmax=True)
    assert (dmax is not None) == (kernel in (4, 5, 6, 8))
    forms = [dict(kernel=2)] + ([dict(dmax=dmax, xmax=genops.rate_bound(gp))] if dmax is not None else [])
    worst = {}
    for form in forms:
        gW = genops.weight_grad(delta, out['traj'], **form)
        got = gW.cpu().numpy().astype('float64')
        assert np.isfinite(got).all()
        err = np.abs(got - o['gW']).reshape(B, -1).max(axis=1) / np.abs(o['gW']).reshape(B, -1).max(axis=1)
        worst['bf16x3' if 'kernel' in form else 'fp16x2'] = err.max()
        assert err.max() < 1e-4, (form.keys(), err)
        gJ, gD, gS = genops.jds_grad(gW, zt, jds['J'], jds['D'], jds['S'])
        for g, w in ((gJ, o['gJ']), (gD, o['gD']), (gS, o['gS'])):
            np.testing.assert_allclose(g, w, rtol=2e-3, atol=2e-3 * np.abs(w).max())
    if o['v'] is not None:
        # ext = (1 + V z_in) base  ->  dL/dV = sum g_ext * base * z_in (networks/ssn.py:679-686, V one scalar: 'deg-heteroin')
        gV = float((g_ext.double() * base.double() * zin.double()[:, None, :]).sum())
        np.testing.assert_allclose(gV, o['gV'], rtol=2e-3)
    print('horizon %s kernel %d: max |dL/dW - fp64| / max |dL/dW| = %s' % (shape, kernel, worst))


@pytest.mark.parametrize('num_sites,batchsize,seqlen,tol', [(10, 1, 4000, 5e-4), (10, 2, 4000, 5e-4),
                                                            (100, 3, 10000, 1e-4)])
def test_compare_with_ssnode(num_sites, batchsize, seqlen, tol):
    """networks/tests/test_euler_ssn.py:29-86: fp32 fixed-time time_avg (skip = seqlen-1) equals the fp64
    fixed point of ssnode.sample_fixed_points(atol=1e-10); reference tolerances 5e-4 (seqlen 4000) and
    1e-4 (seqlen 10000).  Both sides run on the GPU here; the fixed points are also checked against the
    CPU oracle."""
    from tc_gan_amd import ssnode
    from tc_gan_amd.networks.ssn import TuningCurveGenerator
    from tc_gan_amd.networks.wgan import DEFAULT_PARAMS, grid_stimulator_inputs
    jds = on.new_JDS()
    seed = num_sites * batchsize
    bandwidths, contrasts = DEFAULT_PARAMS['bandwidths'], DEFAULT_PARAMS['contrasts']
    con, bw = grid_stimulator_inputs(contrasts, bandwidths, batchsize)
    zs, fps, info = ssnode.sample_fixed_points(batchsize, N=num_sites, bandwidths=bandwidths, contrast=contrasts,
                                               seed=seed, io_type='asym_tanh', atol=1e-10, **jds)
    if num_sites <= 10:
        wz, wfps, _ = on.sample_fixed_points(batchsize, N=num_sites, bandwidths=bandwidths, contrast=contrasts,
                                             seed=seed, io_type='asym_tanh', atol=1e-10, **jds)
        np.testing.assert_allclose(fps, wfps, rtol=1e-7, atol=1e-9)
    gen = TuningCurveGenerator(num_sites=num_sites, num_tcdom=len(bandwidths), smoothness=DEFAULT_PARAMS['smoothness'],
                               J=jds['J'], D=jds['D'], S=jds['S'], k=DEFAULT_PARAMS['k'], n=DEFAULT_PARAMS['n'],
                               tau_E=10, tau_I=1, dt=0.1, io_type='asym_tanh', seqlen=seqlen, skip_steps=seqlen - 1,
                               batchsize=batchsize, probes=[0], include_time_avg=True)
    out = gen.forward(stimulator_bandwidths=bw, stimulator_contrasts=con, model_zs=zs)
    np.testing.assert_allclose(out.model_time_avg.cpu().numpy(), fps, rtol=tol, atol=tol)


@pytest.mark.parametrize('kernel', [2, 3, 4, 5, 6, 8])
def test_compare_with_ssnode_on_every_matrix_core_kernel(kernel):
    """The reference's long-horizon cross-test (networks/tests/test_euler_ssn.py:79-86: num_sites 100, seqlen 10000,
    rtol = atol = 1e-4 against ssnode.sample_fixed_points(atol=1e-10)) with the forward kernel FORCED: with three draws the
    library's own choice is the VALU tile kernel, so the case above never reaches the kernels that run every full-size
    generator step -- fp32 MFMA (2 / 3), fp16-split wide form (4: W and state 22 bits), alternating forms (5 / 6: state
    exact) and the two-draw form (8: state 23 bits by round to nearest)."""
    from tc_gan_amd import genops, ssnode
    from tc_gan_amd.networks.ssn import TuningCurveGenerator
    from tc_gan_amd.networks.wgan import DEFAULT_PARAMS, grid_stimulator_inputs
    jds = on.new_JDS()
    num_sites, batchsize, seqlen, tol = 100, 3, 10000, 1e-4
    bandwidths, contrasts = DEFAULT_PARAMS['bandwidths'], DEFAULT_PARAMS['contrasts']
    con, bw = grid_stimulator_inputs(contrasts, bandwidths, batchsize)
    zs, fps, info = ssnode.sample_fixed_points(batchsize, N=num_sites, bandwidths=bandwidths, contrast=contrasts,
                                               seed=num_sites * batchsize, io_type='asym_tanh', atol=1e-10, **jds)
    gen = TuningCurveGenerator(num_sites=num_sites, num_tcdom=len(bandwidths), smoothness=DEFAULT_PARAMS['smoothness'],
                               J=jds['J'], D=jds['D'], S=jds['S'], k=DEFAULT_PARAMS['k'], n=DEFAULT_PARAMS['n'],
                               tau_E=10, tau_I=1, dt=0.1, io_type='asym_tanh', seqlen=seqlen, skip_steps=seqlen - 1,
                               batchsize=batchsize, probes=[0], include_time_avg=True)
    gen.kernel = kernel
    assert genops.forward_variant(batchsize, len(bandwidths), 2 * num_sites, gen.gen_params()) == kernel
    out = gen.forward(stimulator_bandwidths=bw, stimulator_contrasts=con, model_zs=zs)
    np.testing.assert_allclose(out.model_time_avg.cpu().numpy(), fps, rtol=tol, atol=tol)


def test_full_size_c3_forward_and_adjoint_agree_across_kernels():
    """C3 sizes (2N = 200, 8 stimuli, seqlen 1200 / skip 1000; 256 draws instead of 1024 to bound the 4 GB of
    trajectory per kernel): the fp32-MFMA kernels (two groups per workgroup and one), the fp16-split MFMA kernels (4, 5)
    and the VALU tile kernels compute the same recurrence in different summation orders -- outputs and adjoint results must agree to fp32 accuracy."""
    from tc_gan_amd import genops, stimuli, weight_gen
    N, B, NB, T, skip = 100, 256, 8, 1200, 1000
    rs = np.random.RandomState(7)
    jds = on.new_JDS()
    z = torch.rand((B, 2 * N, 2 * N), device='cuda', generator=torch.Generator(device='cuda').manual_seed(3))
    W = weight_gen.generate_weight_batch(N, jds['J'], jds['D'], jds['S'], z, dtype='float32')
    bws = np.tile(np.asarray(P['bandwidths'])[None, :], (B, 1))
    ext = stimuli.stimulus_batch(bws, np.full_like(bws, 20.0), P['smoothness'], N, dtype='float32')
    gta = torch.as_tensor(rs.rand(B, NB, 2 * N), device='cuda', dtype=torch.float32)
    res = {}
    for kernel in (1, 2, 3, 4, 5, 6, 8):
        gp = genops.make_gen_params(seqlen=T, skip_steps=skip, rate_penalty_threshold=5.0, kernel=kernel, **GEN)
        out = genops.gen_forward(W, ext, gp, save=True)
        delta, dmax = genops.gen_backward(W, out['traj'], out['df'], gta, 1e-3, 1e-3, gp, want_dmax=True)
        assert (dmax is not None) == (kernel in (4, 5, 6, 8))
        # (the fp16-split sweeps: dL/dW on the fp16 two-part form, as the generator update runs it)
        gW = genops.weight_grad(delta, out['traj'], dmax=dmax, xmax=genops.rate_bound(gp))
        res[kernel] = (out['time_avg'].cpu().numpy(), float(out['dynamics_penalty']), float(out['rate_penalty']),
                       gW.cpu().numpy())
        del out, delta, gW
        torch.cuda.empty_cache()
    assert np.isfinite(res[1][0]).all() and res[1][0].max() > 1.0
    for kernel in (2, 3, 4, 5, 6, 8):
        np.testing.assert_allclose(res[kernel][0], res[1][0], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(res[kernel][1], res[1][1], rtol=1e-3)
        np.testing.assert_allclose(res[kernel][2], res[1][2], rtol=1e-4)
        scale = np.abs(res[1][3]).max()
        np.testing.assert_allclose(res[kernel][3], res[1][3], rtol=1e-3, atol=1e-4 * scale)


@pytest.mark.parametrize('B,NB,T,M,dtype,kernel', [
    (3, 2, 5, 20, 'float64', 0), (2, 4, 7, 20, 'float32', 1), (2, 8, 30, 100, 'float32', 2), (3, 8, 50, 200, 'float32', 0),
    (2, 8, 37, 202, 'float32', 2), (2, 3, 11, 224, 'float32', 2), (1, 8, 20, 258, 'float32', 0), (2, 8, 1, 200, 'float32', 2),
    (2, 8, 1200, 200, 'float32', 0), (2, 8, 30, 100, 'float32', 3), (3, 8, 50, 200, 'float32', 3), (2, 8, 37, 202, 'float32', 3),
    (2, 3, 11, 224, 'float32', 3), (2, 8, 1, 200, 'float32', 3), (2, 8, 1200, 200, 'float32', 3)])
def test_weight_grad_kernels_vs_fp64_matmul(B, NB, T, M, dtype, kernel):
    """dL/dW[b] = delta[b]^T traj[b] (csrc/ssn_gw.hip) against numpy fp64.  The split-bf16 MFMA kernel must deliver fp32
    input precision: every fp32 operand is the exact sum of its three bf16 terms and only partial products below 2^-24
    are dropped, so its error is that of an fp32 dot product (~1e-7 * sqrt(K) * |d||x|), not bf16's 4e-3."""
    from tc_gan_amd import genops
    rs = np.random.RandomState(B * 1000 + T)
    # wide dynamic range, mixed signs: rates up to ~1e2, deltas down to ~1e-6
    d = (rs.randn(B, NB, T, M) * np.exp(rs.uniform(-12, 0, (B, NB, T, M)))).astype(dtype)
    x = (rs.rand(B, NB, T, M) * 100 * np.exp(rs.uniform(-6, 0, (B, NB, T, M)))).astype(dtype)
    want = np.einsum('bki,bkj->bij', d.reshape(B, NB * T, M).astype('float64'), x.reshape(B, NB * T, M).astype('float64'))
    # kernel 3 (fp16 two-part form) takes its scales from bounds: max |delta| per draw, and a loose one on the rates
    bounds = dict(dmax=torch.as_tensor(np.abs(d).reshape(B, -1).max(axis=1)).cuda(), xmax=1000.0) if kernel == 3 else {}
    got = genops.weight_grad(torch.as_tensor(d).cuda(), torch.as_tensor(x).cuda(), kernel=kernel, **bounds).cpu().numpy()
    assert got.shape == (B, M, M)
    # scale of one output element: sum_k |d||x| -- the bound every floating-point dot product is measured against
    scale = np.einsum('bki,bkj->bij', np.abs(d.reshape(B, NB * T, M)).astype('float64'),
                      np.abs(x.reshape(B, NB * T, M)).astype('float64'))
    err = np.abs(got - want) / (scale + 1e-300)
    tol = 1e-14 if dtype == "float64" else 2e-6            # fp32 dot product: ~2^-24 per product and per accumulation step, relative to sum |d||x|
    assert err.max() < tol, (err.max(), kernel)
    if dtype == 'float32' and M <= 224:
        e = {}
        for k in (1, 2):
            r = genops.weight_grad(torch.as_tensor(d).cuda(), torch.as_tensor(x).cuda(), kernel=k).cpu().numpy()
            e[k] = (np.abs(r - want) / (scale + 1e-300)).max()
        # the split-bf16 MFMA kernel is at least as accurate as an fp32 FMA chain over the same K
        assert e[2] < tol and e[2] <= 1.5 * e[1] + 1e-7, e
        if kernel == 3:         # ... and so is the fp16 two-part form (operands to 2^-24, the dropped m * m term 2^-24)
            assert err.max() <= 1.5 * e[1] + 1e-7, (err.max(), e)


def test_scaled_weight_grad_ranges_and_refusals():
    """`ssn_weight_grad_scaled_f32`: the per-draw power-of-two scale makes the result independent of the magnitude of delta
    (draws of 1e-30 and 1e+20 next to each other), an all-zero draw gives zeros, a bound far above the data still gives fp32
    accuracy (the parts keep 2^-40 of the bound), a bound BELOW the data overflows to inf / NaN instead of a wrong finite
    value, and NaN in an operand reaches the output."""
    from tc_gan_amd import genops
    rs = np.random.RandomState(5)
    B, NB, T, M = 4, 8, 40, 200
    d = rs.randn(B, NB, T, M).astype('float32')
    x = (rs.rand(B, NB, T, M) * 300).astype('float32')
    mags = np.array([1e-30, 1.0, 1e20, 0.0], dtype='float32')
    d *= mags[:, None, None, None]
    want = np.einsum('bki,bkj->bij', d.reshape(B, NB * T, M).astype('float64'), x.reshape(B, NB * T, M).astype('float64'))
    scale = np.einsum('bki,bkj->bij', np.abs(d.reshape(B, NB * T, M)).astype('float64'),
                      np.abs(x.reshape(B, NB * T, M)).astype('float64'))
    dt, xt = torch.as_tensor(d).cuda(), torch.as_tensor(x).cuda()
    dmax = torch.as_tensor(np.abs(d).reshape(B, -1).max(axis=1)).cuda()
    for loose in (1.0, 2.0 ** 12):
        got = genops.weight_grad(dt, xt, kernel=3, dmax=dmax * loose, xmax=300.0 * loose).cpu().numpy()
        assert np.isfinite(got).all() and (got[3] == 0).all()
        err = np.abs(got[:3] - want[:3]) / scale[:3]
        assert err.max() < 2e-6, (loose, err.max())
    bad = genops.weight_grad(dt, xt, kernel=3, dmax=dmax * 2.0 ** -6, xmax=300.0).cpu().numpy()
    assert not np.isfinite(bad[1]).any()
    dn = dt.clone(); dn[1, 3, 7, 11] = float('nan')
    got = genops.weight_grad(dn, xt, kernel=3, dmax=dmax, xmax=300.0).cpu().numpy()
    assert np.isnan(got[1, 11]).all() and np.isfinite(got[0]).all() and np.isfinite(got[2]).all()
    with pytest.raises(ValueError):
        genops.weight_grad(dt, xt, kernel=3)
    with pytest.raises(Exception):
        genops.weight_grad(dt, xt, kernel=3, dmax=dmax, xmax=float('inf'))


@pytest.mark.parametrize('bwd_kernel', [8, 4, 5])
@pytest.mark.parametrize('io_type', ['asym_tanh', 'asym_power'])
def test_two_draw_adjoint_hands_over_max_delta(io_type, bwd_kernel):
    """`ssn_gen_backward_max_f32`: the fp16-split sweeps (two-draw form 8, alternating forms 4 / 5) return max |delta| per draw
    -- a bound on everything they stored (at most the prologue's estimate of the first step above the true maximum) -- and
    dL/dW from the fp16 form with that bound agrees with the bf16 x 3 form; the other sweeps report `not tracked`."""
    from tc_gan_amd import genops, stimuli, weight_gen
    N, B, NB, T, skip = 100, 5, 8, 120, 80
    jds = on.new_JDS()
    rs = np.random.RandomState(11)
    z = torch.as_tensor(rs.rand(B, 2 * N, 2 * N), device='cuda', dtype=torch.float32)
    W = weight_gen.generate_weight_batch(N, jds['J'], jds['D'], jds['S'], z, dtype='float32')
    bws = np.tile(np.asarray(P['bandwidths'])[None, :], (B, 1))
    ext = stimuli.stimulus_batch(bws, np.full_like(bws, 20.0), P['smoothness'], N, dtype='float32')
    gta = torch.as_tensor(rs.randn(B, NB, 2 * N) * np.array([1e-6, 1.0, 1e4, 1.0, 0.0])[:, None, None], device='cuda',
                          dtype=torch.float32)
    fwd_kernel = 8 if io_type == 'asym_tanh' else 2
    gpf = genops.make_gen_params(seqlen=T, skip_steps=skip, rate_penalty_threshold=5.0, kernel=fwd_kernel, **dict(GEN, io_type=io_type))
    out = genops.gen_forward(W, ext, gpf, save=True)
    gp8 = genops.make_gen_params(seqlen=T, skip_steps=skip, rate_penalty_threshold=5.0, kernel=bwd_kernel, **dict(GEN, io_type=io_type))
    delta, dmax = genops.gen_backward(W, out['traj'], out['df'].clone(), gta, 0.0, 0.0, gp8, want_dmax=True)
    assert dmax is not None and dmax.shape == (B,)
    true = delta.abs().reshape(B, -1).max(dim=1).values.cpu().numpy()
    got = dmax.cpu().numpy()
    assert (got >= true).all() and (got <= np.maximum(true * 4, 1e-30)).all(), (got, true)
    assert got[4] == 0.0
    xmax = genops.rate_bound(gp8)
    assert (xmax is not None) == (io_type == 'asym_tanh')
    a = genops.weight_grad(delta, out['traj'], kernel=2).cpu().numpy()
    b = genops.weight_grad(delta, out['traj'], kernel=3, dmax=dmax, xmax=float(out['traj'].max()) + 1.0).cpu().numpy()
    for i in range(B):
        np.testing.assert_allclose(b[i], a[i], rtol=0, atol=2e-6 * max(np.abs(a[i]).max(), 1e-300) * np.sqrt(NB * T))
    gp2 = genops.make_gen_params(seqlen=T, skip_steps=skip, rate_penalty_threshold=5.0, kernel=2, **dict(GEN, io_type=io_type))
    res = genops.gen_backward(W, out['traj'], out['df'].clone(), gta, 0.0, 0.0, gp2, want_dmax=True)
    assert res[1] is None


def test_split_kernel_refuses_unbounded_io_functions():
    """The fp16-split kernel scales the state by the rate bound of asym_tanh; asked for explicitly with another I/O
    function it must fail, not overflow."""
    from tc_gan_amd import clib, genops, stimuli, weight_gen
    jds, z, bws, con = _problem(50, 2, 8, 3, 20, 10, 1.0)
    W = weight_gen.generate_weight_batch(50, jds['J'], jds['D'], jds['S'], z, dtype='float32')
    ext = stimuli.stimulus_batch(bws, con, P['smoothness'], 50, dtype='float32')
    for kernel in (4, 8):
        for io_type in ('asym_power', 'asym_linear'):
            gp = genops.make_gen_params(seqlen=20, skip_steps=10, kernel=kernel, **dict(GEN, io_type=io_type))
            with pytest.raises(clib.SSNLibraryError):
                genops.gen_forward(W, ext, gp)
        gp = genops.make_gen_params(seqlen=20, skip_steps=10, kernel=kernel, **dict(GEN, dt=1.5))     # dt > tau_I: no bound
        with pytest.raises(clib.SSNLibraryError):
            genops.gen_forward(W, ext, gp)


@pytest.mark.parametrize('case', ['plain', 'weak', 'tiny', 'strong-diagonal'])
def test_split_kernel_error_against_fp64_is_that_of_the_fp32_kernels(case):
    """csrc/ssn_mfma16.hip carries W as two fp16 parts (22 significant bits) and the state as two (wide form, kernel 4) or
    three (alternating form, kernels 5 / 6: exact); every product is exact and the accumulation is fp32.  Its distance from the fp64 oracle must be the distance of the fp32
    MFMA kernel (whose own error is dominated by the fast power law) -- also for weights far from 1 in magnitude (the
    operand scale is taken from max |W| per draw: 2^-20 J, and a -60 self-inhibition on top of |W| ~ 0.1)."""
    from tc_gan_amd import genops, stimuli, weight_gen
    N, B, NB, T, skip = 100, 4, 8, 400, 300
    jds, z, bws, con = _problem(N, B, NB, 11, T, skip, 2.0)
    scale = {'plain': 1.0, 'weak': 1e-3, 'tiny': 2.0 ** -20, 'strong-diagonal': 1.0}[case]
    jds = dict(jds, J=jds['J'] * scale, D=jds['D'] * scale)
    W = weight_gen.generate_weight_batch(N, jds['J'], jds['D'], jds['S'], z, dtype='float32')
    if case == 'strong-diagonal':
        W[:, torch.arange(2 * N), torch.arange(2 * N)] = -60.0
    ext = stimuli.stimulus_batch(bws, con, P['smoothness'], N, dtype='float32')
    # the oracle runs on the very fp32 inputs the kernels get: only the recurrence itself is compared
    ta_o, dyn_o, rate_o = og.euler_ssn(og.t64(W.cpu().numpy()), og.t64(ext.cpu().numpy()), seqlen=T, skip_steps=skip,
                                       rate_penalty_threshold=2.0, **GEN)
    err, ta = {}, {}
    for kernel in (2, 4, 5, 6, 8):
        gp = genops.make_gen_params(seqlen=T, skip_steps=skip, rate_penalty_threshold=2.0, kernel=kernel, **GEN)
        ta[kernel] = genops.gen_forward(W, ext, gp)['time_avg'].cpu().numpy().astype('float64')
        assert np.isfinite(ta[kernel]).all()
        err[kernel] = np.abs(ta[kernel] - ta_o.numpy()).max() / np.abs(ta_o.numpy()).max()
    print('%s: max error / max rate against fp64: fp32 MFMA %.2e, fp16-split wide (state 22 bits, truncated) %.2e, alternating '
          '(state exact) %.2e, two-draw form (state 23 bits, round to nearest) %.2e' % (case, err[2], err[4], err[6], err[8]))
    assert ta_o.numpy().max() > 0.1
    assert err[2] < 2e-5 and err[4] < 2e-5 and err[6] < 2e-5 and err[8] < 2e-5
    assert err[4] < 2.0 * err[2] + 2e-7 and err[6] < 2.0 * err[2] + 2e-7 and err[8] < 2.0 * err[2] + 2e-7
    np.testing.assert_allclose(ta[8], ta[6], rtol=1e-5, atol=2e-6 * ta[6].max())
    np.testing.assert_array_equal(ta[6], ta[5])           # the alternating form with two groups per workgroup or one
    np.testing.assert_allclose(ta[4], ta[6], rtol=1e-5, atol=2e-6 * ta[6].max())


@pytest.mark.parametrize('N,B,NB,T,skip,taus', [(100, 2, 8, 300, 200, (10., 1.)), (101, 1, 6, 120, 100, (10., 1.)),
                                                (50, 2, 8, 400, 390, (10., 1.)), (100, 2, 8, 300, 280, (0.5, 0.25))])
def test_split_adjoint_matches_fp32_adjoint_step_by_step(N, B, NB, T, skip, taus):
    """gen_backward_split_kernel (W^T as two fp16 parts, delta as three, scale following max |delta| with one step of
    lag) against the fp32 MFMA adjoint on the SAME trajectory: delta_t of every step, normalised by that step's own
    maximum.  With the short time constants of the last case the adjoint decays by many orders of magnitude before the
    penalty window, which is what the moving scale has to follow."""
    from tc_gan_amd import genops, stimuli, weight_gen
    gen = dict(GEN, tau_E=taus[0], tau_I=taus[1])
    jds, z, bws, con = _problem(N, B, NB, 5, T, skip, 1.0)
    W = weight_gen.generate_weight_batch(N, jds['J'], jds['D'], jds['S'], z, dtype='float32')
    ext = stimuli.stimulus_batch(bws, con, P['smoothness'], N, dtype='float32')
    rs = np.random.RandomState(2)
    G = torch.as_tensor(rs.randn(B, NB, 2 * N) * (rs.rand(B, NB, 2 * N) < 0.05), device='cuda', dtype=torch.float32)
    gp2 = genops.make_gen_params(seqlen=T, skip_steps=skip, rate_penalty_threshold=1.0, kernel=2, **gen)
    out = genops.gen_forward(W, ext, gp2, save=True)
    res = {}
    for kernel in (2, 4, 5, 6, 8):
        gp = genops.make_gen_params(seqlen=T, skip_steps=skip, rate_penalty_threshold=1.0, kernel=kernel, **gen)
        d, gx = genops.gen_backward(W, out['traj'], out['df'].clone(), G, 1e-3, 1e-3, gp, want_g_ext=True)
        res[kernel] = (d.cpu().numpy().astype('float64'), gx.cpu().numpy().astype('float64'))
    d2, d4 = res[2][0], res[4][0]
    assert np.isfinite(d4).all() and np.isfinite(d2).all()
    per_step = np.abs(d2).max(axis=(0, 1, 3))                       # [T]; the last slot of the shifted stream is zero
    assert per_step[:-1].max() > 0
    live = per_step > 1e-30
    err = np.abs(d4 - d2).max(axis=(0, 1, 3))[live] / per_step[live]
    print('adjoint: max over steps of |split - fp32| / max|delta_t| = %.2e, dynamic range of max|delta_t| %.1e'
          % (err.max(), per_step[live].max() / per_step[live].min()))
    assert err.max() < 5e-6
    if taus[0] < 1:
        assert per_step[live].max() / per_step[live].min() > 1e6
    np.testing.assert_allclose(res[4][1], res[2][1], rtol=1e-5, atol=1e-6 * np.abs(res[2][1]).max())
    # (the adjoint has one form: kernels 4, 5 and 6 differ only in the groups per workgroup)
    np.testing.assert_array_equal(res[5][0], res[6][0])
    np.testing.assert_array_equal(res[4][0], res[6][0])
    err6 = np.abs(res[6][0] - d2).max(axis=(0, 1, 3))[live] / per_step[live]
    assert err6.max() < 5e-6
    # the two-draw form: delta as two fp16 parts by round to nearest, one scale per draw (all 8 stimuli)
    err8 = np.abs(res[8][0] - d2).max(axis=(0, 1, 3))[live] / per_step[live]
    print('adjoint, two-draw form: max over steps of |duo - fp32| / max|delta_t| = %.2e' % err8.max())
    assert np.isfinite(res[8][0]).all() and err8.max() < 5e-6
    np.testing.assert_allclose(res[8][1], res[2][1], rtol=1e-5, atol=1e-6 * np.abs(res[2][1]).max())


@pytest.mark.parametrize('soft,hard,contrast', [(200., 1000., 2000.), (2000., 20000., 5e4), (0.2, 0.5, 20.), (200., 1000., 20.)])
def test_split_kernel_at_the_rate_bound_and_with_other_bounds(soft, hard, contrast):
    """The state scale of the fp16-split forward comes from rate_hard_bound: saturated networks (rates at the bound),
    a bound near the top of the fp16 range (scale 2^0) and a tiny one (scale 2^14) must all match the fp32 MFMA kernel
    and the fp64 oracle."""
    from tc_gan_amd import genops, stimuli, weight_gen
    N, B, NB, T, skip = 100, 2, 8, 200, 150
    jds, z, bws, con = _problem(N, B, NB, 3, T, skip, 2.0)
    con = np.full_like(con, contrast)
    gen = dict(GEN)
    W = weight_gen.generate_weight_batch(N, jds['J'], jds['D'], jds['S'], z, dtype='float32')
    ext = stimuli.stimulus_batch(bws, con, P['smoothness'], N, dtype='float32')
    ta_o = None
    if (soft, hard) == (200., 1000.):                      # (the oracle's I/O function has the default bounds built in)
        ta_o = og.euler_ssn(og.t64(W.cpu().numpy()), og.t64(ext.cpu().numpy()), seqlen=T, skip_steps=skip,
                            rate_penalty_threshold=2.0, **gen)[0]
    ta = {}
    for kernel in (2, 4, 6, 8):
        gp = genops.make_gen_params(seqlen=T, skip_steps=skip, rate_penalty_threshold=2.0, kernel=kernel,
                                    rate_soft_bound=soft, rate_hard_bound=hard, **gen)
        ta[kernel] = genops.gen_forward(W, ext, gp)['time_avg'].cpu().numpy().astype('float64')
    assert np.isfinite(ta[4]).all() and ta[4].max() <= hard * (1 + 1e-6)
    assert np.isfinite(ta[8]).all() and ta[8].max() <= hard * (1 + 1e-6)
    np.testing.assert_allclose(ta[8], ta[2], rtol=2e-5, atol=2e-6 * ta[2].max())
    if contrast >= 2000.:
        assert ta[4].max() > 0.9 * hard                    # really at the bound
    np.testing.assert_allclose(ta[4], ta[2], rtol=2e-5, atol=2e-6 * ta[2].max())
    np.testing.assert_allclose(ta[6], ta[2], rtol=2e-5, atol=2e-6 * ta[2].max())
    if ta_o is not None:
        np.testing.assert_allclose(ta[4], ta_o.numpy(), rtol=1e-4, atol=1e-5 * ta_o.numpy().max())


def test_split_kernels_propagate_nan_and_terminate():
    """A NaN in W must come out as NaN rates (the reference's drivers stop on it), not as finite garbage, and the
    kernels must still drain."""
    from tc_gan_amd import genops, stimuli, weight_gen
    N, B, NB, T, skip = 100, 2, 8, 40, 20
    jds, z, bws, con = _problem(N, B, NB, 3, T, skip, 2.0)
    W = weight_gen.generate_weight_batch(N, jds['J'], jds['D'], jds['S'], z, dtype='float32')
    W[1, 7, 9] = float('nan')
    ext = stimuli.stimulus_batch(bws, con, P['smoothness'], N, dtype='float32')
    for kernel in (4, 8):                                      # (8: the two draws share a workgroup, forward and adjoint)
        gp = genops.make_gen_params(seqlen=T, skip_steps=skip, kernel=kernel, **GEN)
        out = genops.gen_forward(W, ext, gp, save=True)
        ta = out['time_avg'].cpu().numpy()
        assert np.isfinite(ta[0]).all()                        # the other draw is untouched
        assert np.isnan(ta[1]).any()
        G = torch.ones((B, NB, 2 * N), device='cuda', dtype=torch.float32)
        d = genops.gen_backward(W, out['traj'], out['df'], G, 1e-3, 1e-3, gp).cpu().numpy()
        assert np.isfinite(d[0]).all() and np.isnan(d[1]).any()


@pytest.mark.parametrize('kernel', [4, 5, 8])
def test_split_adjoint_marks_a_draw_that_outgrows_its_lagged_scale(kernel):
    """The fp16-split sweeps scale step tau by max |delta| of step tau + 1; a delta that grows by more than 2^8 within ONE
    step cannot be represented and is poisoned (NaN), never clamped.  Here f' of one draw is multiplied by 1e5 at one step:
    that draw's delta, dL/dW and hand-over word `dmax` come out NaN (`TuningCurveGenerator.poisoned_draws` counts those
    words), the other draw is untouched, and the fp32 sweep (kernel 2) carries the large finite value instead."""
    from tc_gan_amd import genops, stimuli, weight_gen
    N, B, NB, T, skip = 100, 2, 8, 60, 40
    jds, z, bws, con = _problem(N, B, NB, 3, T, skip, 2.0)
    W = weight_gen.generate_weight_batch(N, jds['J'], jds['D'], jds['S'], z, dtype='float32')
    ext = stimuli.stimulus_batch(bws, con, P['smoothness'], N, dtype='float32')
    G = torch.ones((B, NB, 2 * N), device='cuda', dtype=torch.float32)
    res = {}
    for kern in (2, kernel):
        gp = genops.make_gen_params(seqlen=T, skip_steps=skip, kernel=kern, **GEN)
        out = genops.gen_forward(W, ext, gp, save=True)
        out['df'][1, :, 30, :] *= 1e5
        d, dmax = genops.gen_backward(W, out['traj'], out['df'], G, 1e-3, 1e-3, gp, want_dmax=True)
        gW = genops.weight_grad(d, out['traj'], **({} if dmax is None else dict(dmax=dmax, xmax=genops.rate_bound(gp))))
        res[kern] = (d.cpu().numpy(), None if dmax is None else dmax.cpu().numpy(), gW.cpu().numpy())
    d32, _, gW32 = res[2]
    assert np.isfinite(d32).all() and np.isfinite(gW32).all() and np.abs(d32[1]).max() > 1e3 * np.abs(d32[0]).max()
    d, dmax, gW = res[kernel]
    assert np.isfinite(d[0]).all() and np.isfinite(gW[0]).all() and np.isfinite(dmax[0])
    np.testing.assert_allclose(gW[0], gW32[0], rtol=1e-3, atol=1e-5 * np.abs(gW32[0]).max())
    assert np.isnan(dmax[1]) and np.isnan(d[1]).any() and np.isnan(gW[1]).any()
    assert int(np.isnan(dmax).sum()) == 1


@pytest.mark.parametrize('dtype', ['float32', 'float64'])
def test_probe_scatter_is_the_adjoint_of_the_gather(dtype):
    """`ssn_probe_scatter_*`: g_ta[ids[k], :, probes[k]] += g[k, :] with collisions, against numpy's add.at."""
    import ctypes
    from tc_gan_amd import clib
    rs = np.random.RandomState(4)
    n, B, NB, M = 300, 7, 5, 22
    ids, probes = rs.randint(0, B, n), rs.randint(0, M, n)
    probes[:40] = probes[0]; ids[:40] = ids[0]                       # many samples on one (model, neuron)
    g = rs.randn(n, NB).astype(dtype)
    want = np.zeros((B, NB, M), dtype='float64')
    np.add.at(want, (ids[:, None], np.arange(NB)[None, :], probes[:, None]), g.astype('float64'))
    td = getattr(torch, dtype)
    out = torch.full((B, NB, M), 7.0, device='cuda', dtype=td)        # the kernel writes the whole tensor
    fn = clib.libssnode.ssn_probe_scatter_f32 if dtype == 'float32' else clib.libssnode.ssn_probe_scatter_f64
    gd, idd, prd = torch.as_tensor(g).cuda(), torch.as_tensor(ids).cuda(), torch.as_tensor(probes).cuda()
    for _ in range(2):
        clib.check(fn(gd.data_ptr(), idd.data_ptr(), prd.data_ptr(), out.data_ptr(), n, B, NB, M, None), 'scatter')
        np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-5 if dtype == 'float32' else 1e-13, atol=1e-6 if dtype == 'float32' else 1e-14)


def test_split_kernels_random_shapes_against_the_fp32_mfma_kernels():
    """Seeded sweep over sizes the parametrised cases do not name: every even 2N class of the three tile grids (<= 104,
    <= 152, <= 208), 4 .. 11 stimuli (ragged last group), 1 .. 3 draws, windows that start at step 0 or end at the last
    step -- forward outputs, trajectory, f' and the adjoint's delta of the fp16-split kernels (4 / 5) against the fp32
    MFMA kernels (2 / 3) on identical inputs."""
    from tc_gan_amd import genops, stimuli, weight_gen
    rs = np.random.RandomState(2026)
    for case in range(14):
        N = int(rs.choice([rs.randint(2, 53), rs.randint(53, 77), rs.randint(77, 105)]))
        B, NB = int(rs.randint(1, 4)), int(rs.randint(4, 12))
        T = int(rs.randint(3, 40))
        skip = int(rs.choice([0, T - 1, rs.randint(0, T)]))
        jds, z, bws, con = _problem(N, B, NB, 100 + case, T, skip, 1.0)
        W = weight_gen.generate_weight_batch(N, jds['J'], jds['D'], jds['S'], z, dtype='float32')
        ext = stimuli.stimulus_batch(bws, con, P['smoothness'], N, dtype='float32')
        G = torch.as_tensor(rs.randn(B, NB, 2 * N), device='cuda', dtype=torch.float32)
        res = {}
        for kernel in (2, 4, 5, 6, 8):
            gp = genops.make_gen_params(seqlen=T, skip_steps=skip, rate_penalty_threshold=1.0, kernel=kernel, **GEN)
            out = genops.gen_forward(W, ext, gp, save=True)
            keep = [out[k].cpu().numpy().astype('float64') for k in ('time_avg', 'traj', 'df')]
            keep += [float(out['dynamics_penalty']) if T - skip > 1 else 0.0, float(out['rate_penalty'])]
            d, gx = genops.gen_backward(W, out['traj'], out['df'], G, 1e-2, 1e-2, gp, want_g_ext=True)
            res[kernel] = keep + [d.cpu().numpy().astype('float64'), gx.cpu().numpy().astype('float64')]
        tag = 'case %d: N=%d B=%d NB=%d T=%d skip=%d' % (case, N, B, NB, T, skip)
        for kernel in (4, 5, 6, 8):
            for got, want in zip(res[kernel], res[2]):
                got, want = np.asarray(got), np.asarray(want)
                assert np.isfinite(got).all(), tag
                scale = np.abs(want).max() if want.size else 0.0
                np.testing.assert_allclose(got, want, rtol=2e-5, atol=3e-6 * scale + 1e-30, err_msg=tag)


def test_two_draw_kernel_half_real_tail_tile_one_value_per_lane_gives_the_same_bits():
    """2N = 194 ... 200: the 13th row tile holds at most 8 real rows and the wave that finishes it takes them one value per
    lane (`v_permlane32_swap`, 7 values per lane instead of 8); `SSN_DUO_HT=0` keeps the row-pair form.  Per value the same
    arithmetic: time averages, trajectory and f' bit-identical for every such 2N (partly filled tail tiles included) and
    next to the range (192, 202: the form does not apply); the window penalties are per-lane sums added in another order."""
    import os
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np, torch; sys.path.insert(0, %r)\n"
        "from tc_gan_amd import genops\n"
        "res = {}\n"
        "for M in (192, 194, 196, 198, 200, 202):\n"
        "    g = torch.Generator(device='cuda'); g.manual_seed(M)\n"
        "    W = (torch.rand((3, M, M), device='cuda', generator=g) - 0.6) * 0.02\n"
        "    ext = torch.rand((3, 8, M), device='cuda', generator=g) * 40\n"
        "    gp = genops.make_gen_params(seqlen=60, skip_steps=40, kernel=8)\n"
        "    out = genops.gen_forward(W, ext, gp, save=True)\n"
        "    plain = genops.gen_forward(W, ext, gp)\n"
        "    res['ta%%d' %% M] = out['time_avg'].cpu().numpy(); res['traj%%d' %% M] = out['traj'].cpu().numpy()\n"
        "    res['df%%d' %% M] = out['df'].cpu().numpy(); res['plain%%d' %% M] = plain['time_avg'].cpu().numpy()\n"
        "    res['pen%%d' %% M] = np.array([float(out['dynamics_penalty']), float(out['rate_penalty'])])\n"
        "np.savez(sys.argv[1], **res)\n" % root)
    res = {}
    with tempfile.TemporaryDirectory() as tmp:
        for form in ('0', '1'):
            path = os.path.join(tmp, 'ht%s.npz' % form)
            subprocess.run([sys.executable, '-c', code, path], check=True, env=dict(os.environ, SSN_DUO_HT=form), timeout=300)
            res[form] = dict(np.load(path))
    for key, want in res['0'].items():
        assert np.isfinite(want).all(), key
        if key.startswith('pen'):
            np.testing.assert_allclose(res['1'][key], want, rtol=1e-5, err_msg=key)
        else:
            np.testing.assert_array_equal(res['1'][key], want, err_msg=key)
    assert res['1']['ta200'].max() > 1.0
    np.testing.assert_array_equal(res['1']['plain200'], res['1']['ta200'])           # with and without trajectory stores


def test_two_draw_kernel_free_running_form_gives_the_same_bits():
    """`SSN_DUO_FREE=1` runs the two-draw forward without a workgroup barrier in the time loop (per-draw LDS counters, double
    buffered B images, flagged partial sums; kept for A/B timing).  Same arithmetic in the same order: its outputs must be
    bit-identical to the lock-step form's (per value; the penalty totals to fp32 summation order), odd unit count included.  (The switch is read once per process: subprocess.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np, torch; sys.path.insert(0, %r)\n"
        "from tc_gan_amd import genops\n"
        "g = torch.Generator(device='cuda'); g.manual_seed(3)\n"
        "W = (torch.rand((5, 200, 200), device='cuda', generator=g) - 0.6) * 0.02\n"
        "ext = torch.rand((5, 11, 200), device='cuda', generator=g) * 40\n"
        "gp = genops.make_gen_params(seqlen=300, skip_steps=250, kernel=8)\n"
        "out = genops.gen_forward(W, ext, gp, save=True)\n"
        "np.savez(sys.argv[1], ta=out['time_avg'].cpu().numpy(), traj=out['traj'].cpu().numpy(), df=out['df'].cpu().numpy(),\n"
        "         pen=np.array([float(out['dynamics_penalty']), float(out['rate_penalty'])]))\n" % root)
    import tempfile
    res = {}
    with tempfile.TemporaryDirectory() as tmp:
        for form in ('0', '1'):
            path = os.path.join(tmp, 'form%s.npz' % form)
            env = dict(os.environ, SSN_DUO_FREE=form)
            subprocess.run([sys.executable, '-c', code, path], check=True, env=env, timeout=300)
            res[form] = dict(np.load(path))
    assert np.isfinite(res['0']['ta']).all() and res['0']['ta'].max() > 1.0
    for key in ('ta', 'traj', 'df'):
        np.testing.assert_array_equal(res['1'][key], res['0'][key], err_msg=key)
    # (the window penalties are per-lane sums; the lock-step form adds the row tile it finishes behind its chain first)
    np.testing.assert_allclose(res['1']['pen'], res['0']['pen'], rtol=1e-5)


def test_heterogeneous_input_stimulus_forms_its_amplification_in_the_launch():
    """`ssn_stimulus_hetero_f32`: the stimulus of the heterogeneous-input models (networks/ssn.py:679-686) with amp = 1 + v z_in
    formed inside the launch -- the same two fp32 roundings as the torch expression in front of `ssn_stimulus_amp_f32`, so the
    same bits; v per neuron, per population and as one value; refuses a v of any other length."""
    import torch
    from tc_gan_amd import clib
    from tc_gan_amd.stimuli import stimulus_batch
    rs = np.random.RandomState(3)
    B, NB, N = 37, 8, 101
    bw = torch.as_tensor(rs.rand(B, NB) * 0.9 + 0.05, device='cuda', dtype=torch.float32)
    con = torch.as_tensor(rs.rand(B, NB) * 30, device='cuda', dtype=torch.float32)
    for zin in (torch.as_tensor(rs.choice(2, (B, 2 * N)) * 2.0 - 1, device='cuda', dtype=torch.float32),
                torch.as_tensor(rs.rand(B, 2 * N) * 2 - 1, device='cuda', dtype=torch.float32)):
        for v in (torch.as_tensor(rs.rand(2 * N) * 0.7, device='cuda', dtype=torch.float32),
                  torch.as_tensor([0.31, 0.057], device='cuda', dtype=torch.float32),
                  torch.as_tensor([0.173], device='cuda', dtype=torch.float32)):
            vs = v if v.numel() == 2 * N else (v.expand(2) if v.numel() == 1 else v).repeat_interleave(N)
            want = stimulus_batch(bw, con, 0.1, N, amp=1 + vs[None, :] * zin)
            got = stimulus_batch(bw, con, 0.1, N, zin=zin, v=v)
            assert torch.equal(got, want)
    ext = torch.empty((B, NB, 2 * N), device='cuda')
    v3 = torch.zeros(3, device='cuda')
    rc = clib.libssnode.ssn_stimulus_hetero_f32(bw.data_ptr(), con.data_ptr(), 0.1, zin.data_ptr(), v3.data_ptr(), 3, ext.data_ptr(),
                                                B, NB, N, clib.stream_ptr())
    assert rc != 0
