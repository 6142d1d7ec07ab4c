"""GPU parity tests for the SSN fixed-point path (call through the C ABI).

Oracle: oracle/liboracle.so (fp64 C restatement, pinned by tests/test_oracle.py)
and the committed golden fixtures produced by the reference build.
Tolerances: fp64 kernels vs the reference -- 1e-9 relative (libm vs ocml pow/tanh
differ in the last ulps, the dynamics are contractive); fp32 kernels -- 1e-4
relative, the tolerance BASELINE.json's north_star states.
"""
import ctypes

import numpy as np
import pytest

from conftest import golden
from oracle import ssn_numpy as on

pytestmark = pytest.mark.gpu
P = on.DEFAULT_PARAMS
RTOL32 = 1e-4      # north_star: "within 1e-4 relative float32 tolerance"
RTOL64 = 1e-9


def _inputs(N, B, NB, seed, jds=None, contrast=20.):
    jds = jds or on.new_JDS()
    rs = np.random.RandomState(seed)
    zs = rs.rand(B, 2 * N, 2 * N)
    Ws = np.stack([on.generate_weight(N, jds['J'], jds['D'], jds['S'], z) for z in zs])
    bws = P['bandwidths'][:NB] if NB <= 8 else list(np.linspace(0, 1, NB))
    if NB == 1:
        bws = [1.0]
    exts = on.stimulus_input(bws, np.linspace(-.5, .5, N), P['smoothness'], [contrast])
    return Ws, exts


def _oracle_batch(oracle_lib, Ws, exts, io_type, max_iter, atol, dt=8e-4, hard=1000., soft=200., r0=None):
    B, M = Ws.shape[0], Ws.shape[1]
    NB = exts.shape[0]
    r = np.zeros((B, NB, M)) if r0 is None else np.array(np.broadcast_to(r0, (B, NB, M)), dtype=float)
    scratch = np.zeros_like(r)
    codes = np.zeros((B, NB), dtype=np.int32)
    steps = np.zeros((B, NB), dtype=np.int32)
    ip = ctypes.POINTER(ctypes.c_int)
    oracle_lib.oracle_solve_batch(
        on.IO_CODES[io_type], B, NB, M // 2, on.ptr(np.ascontiguousarray(Ws)), on.ptr(np.ascontiguousarray(exts)),
        P['k'], P['n'], on.ptr(r), on.ptr(scratch), P['tau'][0], P['tau'][1], dt, max_iter, atol, soft, hard,
        codes.ctypes.data_as(ip), steps.ctypes.data_as(ip))
    # newest state: for code 1 with odd step counts / code 2 it sits in the scratch ("r1") buffer
    newest = r.copy()
    odd1 = (codes == 1) & (steps % 2 == 1)
    two = (codes == 2) & ((steps - 1) % 2 == 0)
    newest[odd1 | two] = scratch[odd1 | two]
    return newest, codes, steps


# ------------------------------------------------------------------ drop-in symbols
def test_legacy_symbols_match_reference_end_states_and_buffer_parity():
    """Every golden case: same return code and the same contents of BOTH caller buffers."""
    from tc_gan_amd.clib import libssnode, double_ptr
    g = golden('solver_cases.npz')
    jds = on.new_JDS()
    for c in g['cases']:
        N = int(c['N'])
        key = 'W_%d' % c['id']
        if key in g.files:
            W, ext = g[key], g['ext_%d' % c['id']]
        else:
            z = np.random.RandomState(int(c['seed'])).rand(2 * N, 2 * N)
            W = on.generate_weight(N, jds['J'], jds['D'], jds['S'], z)
            ext = on.stimulus_input([float(c['bw'])], np.linspace(-.5, .5, N), P['smoothness'], [20.])[0]
        W = np.ascontiguousarray(W); ext = np.ascontiguousarray(ext)
        r0 = np.zeros(2 * N); r1 = np.full(2 * N, np.nan)
        name = {'asym_power': 'power', 'asym_linear': 'linear', 'asym_tanh': 'tanh'}[str(c['io_type'])]
        code = getattr(libssnode, 'solve_dynamics_asym_%s_euler' % name)(
            N, W.ctypes.data_as(double_ptr), ext.ctypes.data_as(double_ptr), P['k'], P['n'],
            r0.ctypes.data_as(double_ptr), r1.ctypes.data_as(double_ptr), P['tau'][0], P['tau'][1],
            float(c['dt']), int(c['max_iter']), float(c['atol']), float(c['soft']), float(c['hard']))
        assert code == int(c['code']), dict(zip(c.dtype.names, c))
        np.testing.assert_allclose(r0, g['r0_%d' % c['id']], rtol=RTOL64, atol=1e-12)
        np.testing.assert_allclose(r1, g['r1_%d' % c['id']], rtol=RTOL64, atol=1e-12)


def test_legacy_symbols_are_reentrant_from_16_threads():
    """The reference calls the solver from cpu_count() Python threads with disjoint buffers and the GIL released
    (ssnode.py:436-459).  Every golden case, submitted 4 times over from a 16-thread pool in shuffled order,
    together with interleaved scalar helper calls: each call must return exactly what a lone call returns (codes and
    both caller buffers vs the reference build's fixtures)."""
    from multiprocessing.dummy import Pool
    from tc_gan_amd.clib import libssnode, double_ptr
    g = golden('solver_cases.npz')
    jds = on.new_JDS()
    jobs = []
    for c in g['cases']:
        N = int(c['N'])
        key = 'W_%d' % c['id']
        if key in g.files:
            W, ext = g[key], g['ext_%d' % c['id']]
        else:
            z = np.random.RandomState(int(c['seed'])).rand(2 * N, 2 * N)
            W = on.generate_weight(N, jds['J'], jds['D'], jds['S'], z)
            ext = on.stimulus_input([float(c['bw'])], np.linspace(-.5, .5, N), P['smoothness'], [20.])[0]
        jobs.append((c, np.ascontiguousarray(W), np.ascontiguousarray(ext)))
    jobs = jobs * 4
    np.random.RandomState(3).shuffle(jobs)
    io_want = float(on.io_fun(1.7, 'asym_tanh', P['k'], P['n'], 200., 1000.))

    def run(job):
        c, W, ext = job
        N = int(c['N'])
        r0 = np.zeros(2 * N); r1 = np.full(2 * N, np.nan)
        name = {'asym_power': 'power', 'asym_linear': 'linear', 'asym_tanh': 'tanh'}[str(c['io_type'])]
        code = getattr(libssnode, 'solve_dynamics_asym_%s_euler' % name)(
            N, W.ctypes.data_as(double_ptr), ext.ctypes.data_as(double_ptr), P['k'], P['n'],
            r0.ctypes.data_as(double_ptr), r1.ctypes.data_as(double_ptr), P['tau'][0], P['tau'][1],
            float(c['dt']), int(c['max_iter']), float(c['atol']), float(c['soft']), float(c['hard']))
        io = libssnode.io_atanh(1.7, 200., 1000., on.rate_to_volt(200., P['k'], P['n']), P['k'], P['n'])
        return int(c['id']), code, r0, r1, io

    pool = Pool(16)
    try:
        results = pool.map(run, jobs, chunksize=1)
    finally:
        pool.close(); pool.join()
    cases = {int(c['id']): c for c in g['cases']}
    assert len(results) == len(jobs)
    for cid, code, r0, r1, io in results:
        assert code == int(cases[cid]['code'])
        np.testing.assert_allclose(r0, g['r0_%d' % cid], rtol=RTOL64, atol=1e-12)
        np.testing.assert_allclose(r1, g['r1_%d' % cid], rtol=RTOL64, atol=1e-12)
        np.testing.assert_allclose(io, io_want, rtol=1e-12)


def test_legacy_io_symbols_vs_reference_tables():
    # reference tolerance tests/test_ssn.py:21,63: atol 1e-12
    from tc_gan_amd.clib import libssnode
    g = golden('io_tables.npz')
    args = (float(g['r0']), float(g['r1']), float(g['v0']), float(g['k']), float(g['n']))
    idx = np.linspace(0, len(g['xs']) - 1, 60).astype(int)       # scalar calls are one launch each
    for name in ('io_pow', 'io_alin', 'io_atanh'):
        got = np.array([getattr(libssnode, name)(float(g['xs'][i]), *args) for i in idx])
        np.testing.assert_allclose(got, g[name][idx], rtol=1e-13, atol=1e-12)
    got = np.array([libssnode.rate_to_volt(float(g['rates'][i]), float(g['k']), float(g['n'])) for i in idx])
    np.testing.assert_allclose(got, g['rate_to_volt'][idx], rtol=1e-13, atol=1e-12)
    x = np.random.RandomState(0).randn(300); y = np.random.RandomState(1).randn(300)
    from tc_gan_amd.clib import double_ptr
    d = libssnode.dot(300, x.ctypes.data_as(double_ptr), y.ctypes.data_as(double_ptr))
    np.testing.assert_allclose(d, np.dot(x, y), rtol=1e-12)


def test_io_eval_arrays_all_tables():
    from tc_gan_amd import ssnode
    g = golden('io_tables.npz')
    kw = dict(k=float(g['k']), n=float(g['n']), rate_soft_bound=float(g['r0']), rate_hard_bound=float(g['r1']))
    for io_type, name in (('asym_power', 'io_pow'), ('asym_linear', 'io_alin'), ('asym_tanh', 'io_atanh')):
        got = ssnode.io_eval(g['xs'], io_type, **kw)
        np.testing.assert_allclose(got, g[name], rtol=1e-13, atol=1e-12)
        got32 = ssnode.io_eval(g['xs'].astype(np.float32), io_type, **kw)
        assert got32.dtype == np.float32
        np.testing.assert_allclose(got32, g[name], rtol=2e-6, atol=1e-6)
    f = ssnode.make_io_fun(io_type='asym_tanh', **kw)
    np.testing.assert_allclose(f(g['xs']), g['io_atanh'], rtol=1e-13, atol=1e-12)


def test_test_inf_message():
    # tests/test_dynamics.py:129-137
    from tc_gan_amd.ssnode import fixed_point
    sol = fixed_point(W=[[2, 0], [0, 0]], ext=[10, 10], k=1, n=1, r0=[0, 0],
                      max_iter=10000000, io_type='asym_linear')
    assert sol.message == "Reached to rate_stop_at"
    g = golden('solver_cases.npz')
    np.testing.assert_allclose(sol.x, g['inf_r0'], rtol=RTOL64)


# ------------------------------------------------------------------ batched kernels
@pytest.mark.parametrize('io_type', ['asym_power', 'asym_linear', 'asym_tanh'])
@pytest.mark.parametrize('N,NB,variant,dtype', [
    (50, 1, 1, 'float32'), (50, 8, 1, 'float32'), (100, 1, 1, 'float32'), (100, 8, 1, 'float32'),
    (100, 3, 1, 'float32'), (16, 5, 1, 'float32'), (100, 2, 0, 'float32'),
    (50, 2, 1, 'float64'), (100, 1, 0, 'float64'), (23, 1, 1, 'float64'),
    (50, 1, 2, 'float32'), (50, 8, 2, 'float32'), (100, 1, 2, 'float32'), (100, 8, 2, 'float32'),
    (100, 3, 2, 'float32'), (102, 5, 2, 'float32'), (16, 5, 2, 'float32'), (1, 1, 2, 'float32'),
    (75, 2, 2, 'float32'), (50, 2, 2, 'float64'), (23, 3, 2, 'float64'),
    # variant 3: tile kernel with split VGPR/LDS residency (2N = 200, 152); 4: whole tile in VGPRs
    (100, 1, 3, 'float32'), (100, 8, 3, 'float32'), (102, 3, 3, 'float32'), (50, 8, 3, 'float32'),
    (75, 2, 3, 'float32'), (16, 5, 3, 'float32'), (1, 1, 3, 'float32'), (100, 1, 4, 'float32'), (100, 2, 4, 'float32'),
    (76, 1, 3, 'float32'), (76, 3, 4, 'float32'), (90, 3, 3, 'float32'),
    # variant 5: fp32 MFMA kernel (NB >= 4: matrix + serial waves, two stimulus groups half a step apart)
    (100, 8, 5, 'float32'), (100, 4, 5, 'float32'), (50, 5, 5, 'float32'), (101, 9, 5, 'float32'), (16, 11, 5, 'float32'),
    (76, 8, 5, 'float32'),
    # variant 6: fp16-split MFMA kernel (asym_tanh only; W two fp16 parts, state three, exact products)
    (100, 8, 6, 'float32'), (100, 4, 6, 'float32'), (50, 5, 6, 'float32'), (101, 9, 6, 'float32'), (16, 11, 6, 'float32'),
    (76, 8, 6, 'float32'), (52, 8, 6, 'float32'),
    # variant 7: the same in the alternating two-group form (state as three fp16 parts)
    (100, 8, 7, 'float32'), (50, 5, 7, 'float32'), (101, 9, 7, 'float32'), (76, 8, 7, 'float32'),
    # variant 8: fp16-split kernel with two draws per workgroup, every wave both roles (csrc/ssn_duo.hip; B = 6 draws:
    # even unit count, 9 / 11 stimuli: two groups per draw, 5: ragged group)
    (100, 8, 8, 'float32'), (100, 4, 8, 'float32'), (50, 5, 8, 'float32'), (101, 9, 8, 'float32'), (16, 11, 8, 'float32'),
    (76, 8, 8, 'float32'), (52, 8, 8, 'float32'),
    # fp64 resident shapes beyond 2N = 104: 4 rows per lane, 5-7 waves, one workgroup per CU (2N = 204 is the reference's
    # default N = 102: the truth-data path of every CLI run)
    (102, 1, 2, 'float64'), (102, 8, 2, 'float64'), (100, 3, 2, 'float64'), (76, 2, 2, 'float64'), (60, 1, 2, 'float64'),
    (104, 1, 2, 'float64'),
    # mixed kernels: lighter tile for the last wave (2N = 114..152 at C = 19: 4- and 5-row tiles; 2N = 170..200 at C = 25)
    (58, 1, 3, 'float32'), (64, 2, 3, 'float32'), (72, 1, 3, 'float32'), (73, 3, 3, 'float32'), (86, 1, 3, 'float32'),
    (92, 2, 3, 'float32'), (99, 1, 3, 'float32'),
])
def test_fixed_step_batch_vs_oracle(oracle_lib, io_type, N, NB, variant, dtype):
    """atol=0 -> exactly T Euler steps (code 1): end states vs the fp64 C oracle."""
    from tc_gan_amd.ssnode import fixed_points_batch
    if variant in (6, 7, 8) and io_type != 'asym_tanh':
        pytest.skip('the fp16-split solver needs the rate bound of asym_tanh (refused otherwise: test below)')
    B, T = 6, 300
    Ws, exts = _inputs(N, B, NB, seed=N * 31 + NB)
    want, wcodes, wsteps = _oracle_batch(oracle_lib, Ws, exts, io_type, T, 0.0)
    res = fixed_points_batch(Ws, exts, P['k'], P['n'], max_iter=T, atol=0.0, io_type=io_type,
                             rate_stop_at=1000. if io_type != 'asym_tanh' else np.inf,
                             dtype=dtype, variant=variant)
    np.testing.assert_array_equal(res.codes, wcodes)
    np.testing.assert_array_equal(res.steps, wsteps)
    rtol = RTOL32 if dtype == 'float32' else RTOL64
    np.testing.assert_allclose(res.x, want, rtol=rtol, atol=rtol * 1e-2)


@pytest.mark.parametrize('dtype,variant', [('float64', 0), ('float64', 1), ('float64', 2), ('float32', 1),
                                           ('float32', 2), ('float32', 0), ('float32', 3), ('float32', 4), ('float32', 5),
                                           ('float32', 6), ('float32', 7), ('float32', 8)])
def test_converging_batch_codes_steps_states(oracle_lib, dtype, variant):
    """Default solver settings (atol 1e-5, dt 8e-4): per-pair convergence step and state."""
    from tc_gan_amd.ssnode import fixed_points_batch
    N, B, NB = 50, 5, 8
    Ws, exts = _inputs(N, B, NB, seed=7)
    for io_type in ('asym_power', 'asym_tanh'):
        if variant in (6, 7, 8) and io_type != 'asym_tanh':
            continue
        want, wcodes, wsteps = _oracle_batch(oracle_lib, Ws, exts, io_type, 100000, 1e-5,
                                             hard=np.inf if io_type != 'asym_tanh' else 1000.)
        assert (wcodes == 0).all()
        res = fixed_points_batch(Ws, exts, P['k'], P['n'], max_iter=100000, atol=1e-5, io_type=io_type,
                                 dtype=dtype, variant=variant, want_prev=True)
        np.testing.assert_array_equal(res.codes, wcodes)
        if dtype == 'float64':
            assert np.abs(res.steps - wsteps).max() <= 1
            np.testing.assert_allclose(res.x, want, rtol=1e-7, atol=1e-9)
        else:
            # fp32 rounding noise in |r1-r0| (~4e-6 at r~50) can delay or advance the stop step
            assert np.abs(res.steps - wsteps).max() <= 0.2 * wsteps.max()
            np.testing.assert_allclose(res.x, want, rtol=RTOL32, atol=1e-3)
        # previous state is one Euler step behind the newest one
        assert np.abs(res.x - res.x_prev).max() < 1e-4


def test_blowup_codes_match_oracle(oracle_lib):
    """rate_stop_at=200 with unstable parameters (dataset.py:46-49): code 2 cases, mixed with good ones."""
    from tc_gan_amd.ssnode import fixed_points_batch
    g = golden('solver_cases.npz')
    bad = [c for c in g['cases'] if ('W_%d' % c['id']) in g.files]
    Ws = np.stack([g['W_%d' % c['id']] for c in bad])
    exts = np.stack([g['ext_%d' % bad[0]['id']]])
    want, wcodes, wsteps = _oracle_batch(oracle_lib, Ws, exts, 'asym_power', 100000, 1e-5, dt=5e-4, hard=200.)
    assert set(wcodes.flat) >= {2}
    for dtype, variant in (('float64', 1), ('float64', 0), ('float64', 2), ('float32', 1), ('float32', 2),
                           ('float32', 3), ('float32', 4)):
        res = fixed_points_batch(Ws, exts, P['k'], P['n'], max_iter=100000, atol=1e-5, dt=5e-4,
                                 io_type='asym_power', rate_stop_at=200., dtype=dtype, variant=variant)
        np.testing.assert_array_equal(res.codes, wcodes)
        if dtype == 'float64':
            np.testing.assert_array_equal(res.steps, wsteps)
            np.testing.assert_allclose(res.x, want, rtol=1e-8)


def test_variants_agree_and_edge_shapes(oracle_lib):
    from tc_gan_amd.ssnode import fixed_points_batch
    # M = 2 (one E, one I), NB = 1; M = 402 (no register instantiation -> streaming kernel)
    for N, NB in ((1, 1), (201, 2)):
        Ws, exts = _inputs(N, 2, NB, seed=N)
        want, wcodes, _ = _oracle_batch(oracle_lib, Ws, exts, 'asym_tanh', 60, 0.0)
        res = fixed_points_batch(Ws, exts, P['k'], P['n'], max_iter=60, atol=0.0, dtype='float32')
        np.testing.assert_allclose(res.x, want, rtol=RTOL32, atol=1e-6)
    # per-draw stimuli + non-zero initial state
    N, B, NB = 20, 3, 4
    Ws, exts = _inputs(N, B, NB, seed=3)
    extb = np.stack([exts * (1 + 0.1 * b) for b in range(B)])
    r0 = np.random.RandomState(5).rand(B, NB, 2 * N) * 10
    a = fixed_points_batch(Ws, extb, P['k'], P['n'], r0=r0, max_iter=100, atol=0.0, dtype='float64', variant=1)
    b = fixed_points_batch(Ws, extb, P['k'], P['n'], r0=r0, max_iter=100, atol=0.0, dtype='float64', variant=0)
    np.testing.assert_allclose(a.x, b.x, rtol=1e-11)
    c = fixed_points_batch(Ws, extb, P['k'], P['n'], r0=r0, max_iter=100, atol=0.0, dtype='float64', variant=2)
    np.testing.assert_allclose(a.x, c.x, rtol=1e-11)
    for bb in range(B):
        want, _, _ = _oracle_batch(oracle_lib, Ws[bb:bb + 1], extb[bb], 'asym_tanh', 100, 0.0, r0=r0[bb:bb + 1])
        np.testing.assert_allclose(a.x[bb:bb + 1], want, rtol=RTOL64)
    # a batch large enough for the library to pick the MFMA solver by itself: same answers as the tile kernel
    Wl, extl = _inputs(70, 200, 8, seed=11)
    auto = fixed_points_batch(Wl, extl, P['k'], P['n'], max_iter=40, atol=0.0, dtype='float32')
    tile = fixed_points_batch(Wl, extl, P['k'], P['n'], max_iter=40, atol=0.0, dtype='float32', variant=2)
    np.testing.assert_array_equal(auto.codes, tile.codes)
    np.testing.assert_array_equal(auto.steps, tile.steps)
    np.testing.assert_allclose(auto.x, tile.x, rtol=1e-4, atol=1e-6)
    # empty batch, zero iterations
    e = fixed_points_batch(np.zeros((0, 4, 4)), np.ones((2, 4)), P['k'], P['n'])
    assert e.x.shape == (0, 2, 4)
    z = fixed_points_batch(Ws, exts, P['k'], P['n'], r0=r0, max_iter=0, dtype='float32')
    np.testing.assert_array_equal(z.codes, 1)
    np.testing.assert_array_equal(z.steps, 0)
    np.testing.assert_allclose(z.x, r0, rtol=1e-6)


# ------------------------------------------------------------------ callers
def test_build_w_and_stimulus_kernels():
    from tc_gan_amd import stimuli, weight_gen
    g = golden('weights_stimuli.npz')
    m = golden('matlab_ne51.npz')
    for N in (5, 50):
        W = weight_gen.generate_weight(N, g['new_J'], g['new_D'], g['new_S'], g['z_N%d' % N])
        np.testing.assert_allclose(W, g['W_N%d' % N], rtol=1e-12, atol=1e-15)
        x = np.linspace(-.5, .5, N)
        np.testing.assert_allclose(stimuli.input(P['bandwidths'], x, P['smoothness'], P['contrast']),
                                   g['stim_N%d' % N], rtol=1e-11, atol=1e-13)
        np.testing.assert_allclose(stimuli.input([0.25, 1.0], x, 0.1, [5., 20.], [0., 0.1]),
                                   g['stim2_N%d' % N], rtol=1e-11, atol=1e-13)
        W32 = weight_gen.generate_weight_batch(N, g['new_J'], g['new_D'], g['new_S'],
                                               g['z_N%d' % N][None], dtype='float32').cpu().numpy()[0]
        np.testing.assert_allclose(W32, g['W_N%d' % N], rtol=1e-5, atol=1e-7)
    # MATLAB known answer (tests/test_dynamics.py:43-67, atol 1e-6)
    N = int(m['Ne'])
    np.testing.assert_allclose(weight_gen.generate_weight(N, m['J'], m['D'], m['S'], m['Z']), m['W'], atol=1e-6)
    # odd 2N % 4 != 0 path (scalar accesses)
    z = np.random.RandomState(1).rand(3, 6, 6)
    got = weight_gen.generate_weight_batch(3, g['new_J'], g['new_D'], g['new_S'], z, dtype='float64').cpu().numpy()
    for b in range(3):
        np.testing.assert_allclose(got[b], on.generate_weight(3, g['new_J'], g['new_D'], g['new_S'], z[b]), rtol=1e-12)


@pytest.mark.parametrize('io_type', ['asym_linear', 'asym_power', 'asym_tanh'])
def test_tuning_curve_vs_matlab_known_answer_gpu(io_type):
    """tests/test_dynamics.py:76-126 through the GPU finder (reference rtol 0.1)."""
    from tc_gan_amd import ssnode, stimuli
    g = golden('matlab_ne51.npz')
    N = int(g['Ne'])
    exts = stimuli.input(g['bandwidths'], np.linspace(-0.5, 0.5, N), float(g['smoothness']), [float(g['contrast'])])
    for dtype in ('float64', 'float32'):
        (z,), (fps,), info = ssnode.find_fixed_points(
            1, iter([('zz', g['W'])]), exts, k=float(g['k']), n=float(g['n']),
            r0=np.zeros(2 * N), io_type=io_type, method='parallel', check=True, dtype=dtype)
        assert z == 'zz' and info.rejections == 0
        ET = g['E_Tuning']
        center, ofs = N // 2, len(ET) // 2
        actual = np.array([x[center - ofs:center + ofs + 1] for x in fps]).T
        np.testing.assert_allclose(actual, ET, rtol=2e-3)
        # inhibitory block, same asset (`I_Tuning`; see tests/test_oracle.py)
        actual_i = np.array([x[N + center - ofs:N + center + ofs + 1] for x in fps]).T
        np.testing.assert_allclose(actual_i, g['I_Tuning'], rtol=2e-3)


def test_find_fixed_points_rejection_semantics_vs_oracle():
    from tc_gan_amd import ssnode
    g = golden('solver_cases.npz')
    bad = [c for c in g['cases'] if ('W_%d' % c['id']) in g.files]
    badW = {int(c['id']): g['W_%d' % c['id']] for c in bad}
    codes = {int(c['id']): int(c['code']) for c in bad}
    jds = on.new_JDS()
    N = 10
    goodW = [on.generate_weight(N, jds['J'], jds['D'], jds['S'], np.random.RandomState(s).rand(2 * N, 2 * N))
             for s in range(6)]
    ids = sorted(badW)
    seq = [('g0', goodW[0]), ('b%d' % ids[0], badW[ids[0]]), ('g1', goodW[1]), ('b%d' % ids[1], badW[ids[1]]),
           ('b%d' % ids[2], badW[ids[2]]), ('g2', goodW[2]), ('g3', goodW[3]), ('g4', goodW[4]), ('g5', goodW[5])]
    exts = on.stimulus_input([0.0625, 0.25, 1.0], np.linspace(-.5, .5, N), P['smoothness'], [40.])
    kw = dict(k=P['k'], n=P['n'], io_type='asym_power', dt=5e-4, max_iter=100000, rate_stop_at=200.)
    wz, wx, wcounter = on.find_fixed_points(4, iter(seq), exts, **kw)
    zs, xs, info = ssnode.find_fixed_points(4, iter(seq), exts, **kw)
    assert list(zs) == list(wz)
    assert dict(info.counter) == dict(wcounter)
    assert info.rejections == sum(wcounter.values())
    np.testing.assert_allclose(xs, wx, rtol=1e-7, atol=1e-9)
    assert len(info.solutions) == 4 and all(s.success for sols in info.solutions for s in sols)
    with pytest.raises(ssnode.FixedPointError):
        ssnode.find_fixed_points(4, iter(seq), exts, check=True, **kw)


def test_sample_tuning_curves_matches_oracle_sampler():
    """ssnode.sample_tuning_curves (the 'ssnode' dataset provider, dataset.py:28-71) on a small net."""
    from tc_gan_amd import ssnode
    jds = on.new_JDS()
    kw = dict(NZ=5, seed=42, N=12, bandwidths=[0.0625, 0.25, 1.0], contrast=[20.],
              io_type='asym_power', dt=5e-4, max_iter=100000, rate_stop_at=200., **jds)
    data, (zs, rates, info) = ssnode.sample_tuning_curves(
        sample_sites=[5, 6], track_offset_identity=True, include_inhibitory_neurons=True, **kw)
    wz, wr, wc = on.sample_fixed_points(**kw)
    np.testing.assert_array_equal(zs, wz)
    np.testing.assert_allclose(rates, wr, rtol=1e-7, atol=1e-9)
    want = on.subsample_neurons(wr, [5, 6], track_offset_identity=True, include_inhibitory_neurons=True).T
    np.testing.assert_allclose(data, want, rtol=1e-7, atol=1e-9)


# ------------------------------------------------------------------ full-size properties
def test_full_size_c2_properties(oracle_lib):
    """BASELINE config 2 shape (2N=200, batch 4096, 2000 steps, fp32): batch independence (a
    draw's result does not depend on its batch), oracle spot checks, determinism."""
    import torch
    from tc_gan_amd.ssnode import fixed_points_batch
    from tc_gan_amd.weight_gen import generate_weight_batch
    N, B, T = 100, 4096, 2000
    jds = on.new_JDS()
    g = torch.Generator(device='cpu'); g.manual_seed(0)
    z = torch.rand((B, 2 * N, 2 * N), generator=g, dtype=torch.float64)
    W = generate_weight_batch(N, jds['J'], jds['D'], jds['S'], z, dtype='float32')
    exts = on.stimulus_input([1.0], np.linspace(-.5, .5, N), P['smoothness'], [20.])
    kw = dict(k=P['k'], n=P['n'], max_iter=T, atol=0.0, io_type='asym_tanh', dtype='float32', return_torch=True)
    full = fixed_points_batch(W, exts, **kw)
    again = fixed_points_batch(W, exts, **kw)
    assert torch.equal(full.x, again.x)                                   # deterministic
    assert int((full.codes != 1).sum()) == 0 and int((full.steps != T).sum()) == 0
    pick = [0, 1, 777, 4095]
    sub = fixed_points_batch(W[pick], exts, **kw)
    assert torch.equal(sub.x, full.x[pick])                               # batch independence, bitwise
    Wd = np.stack([on.generate_weight(N, jds['J'], jds['D'], jds['S'], z[i].numpy()) for i in pick])
    want, _, _ = _oracle_batch(oracle_lib, Wd, exts, 'asym_tanh', T, 0.0)
    np.testing.assert_allclose(sub.x.cpu().numpy(), want, rtol=RTOL32, atol=1e-5)
    assert torch.isfinite(full.x).all()


def test_default_size_fp64_resident_kernel_converges_like_the_oracle(oracle_lib):
    """2N = 204 (the reference's default N = 102, tc_gan/ssnode.py:28), fp64, default solver settings: the register/LDS
    resident tile kernel (the library's choice) and the streaming kernel against the C oracle -- codes, stop steps and
    fixed points to 1e-9."""
    from tc_gan_amd.ssnode import fixed_points_batch
    N, B, NB = 102, 3, 8
    Ws, exts = _inputs(N, B, NB, seed=11)
    want, wcodes, wsteps = _oracle_batch(oracle_lib, Ws, exts, 'asym_tanh', 100000, 1e-5)
    assert (wcodes == 0).all()
    from tc_gan_amd.clib import libssnode
    assert libssnode.ssn_solver_fast_path(2 * N, NB, 8) == 2              # the tile kernel covers this size
    for variant in (None, 0):
        res = fixed_points_batch(Ws, exts, P['k'], P['n'], max_iter=100000, atol=1e-5, io_type='asym_tanh',
                                 dtype='float64', variant=variant, want_prev=True)
        np.testing.assert_array_equal(res.codes, wcodes)
        assert np.abs(res.steps - wsteps).max() <= 1
        np.testing.assert_allclose(res.x, want, rtol=RTOL64, atol=1e-9)


def test_split_solver_initial_states_beyond_the_rate_bound_and_refusals():
    """Variant 6 takes its state scale from max(rate_hard_bound, max |r0|): a start far above the bound must relax like
    the fp32 kernels' (not saturate in fp16); asked for with an unbounded I/O function it is refused."""
    from tc_gan_amd import clib
    from tc_gan_amd.ssnode import fixed_points_batch
    N, B, NB = 100, 3, 8
    Ws, exts = _inputs(N, B, NB, seed=5)
    r0 = np.random.RandomState(1).rand(B, NB, 2 * N) * 9000.0          # hard bound 1000
    b = fixed_points_batch(Ws, exts, P['k'], P['n'], r0=r0, max_iter=200, atol=0.0, dtype='float32', variant=2)
    for variant in (6, 8):                       # (8: B = 3 draws, the second workgroup holds one draw and an idle half)
        a = fixed_points_batch(Ws, exts, P['k'], P['n'], r0=r0, max_iter=200, atol=0.0, dtype='float32', variant=variant,
                               want_prev=True)
        np.testing.assert_allclose(a.x, b.x, rtol=2e-5, atol=1e-4)
        assert np.isfinite(a.x).all() and np.isfinite(a.x_prev).all()
        with pytest.raises(clib.SSNLibraryError):
            fixed_points_batch(Ws, exts, P['k'], P['n'], max_iter=10, atol=0.0, io_type='asym_power', dtype='float32',
                               variant=variant)


def test_two_draw_solver_with_per_draw_stimuli_large_batch_and_mixed_stop_steps(oracle_lib):
    """Variant 8 on what the automatic dispatch gives it: > 256 (draw, 8 stimuli) units, stimuli per draw, start states
    that differ per draw, default tolerances -- so the two draws of a workgroup stop at different steps and one half idles
    until the other is done.  Codes and steps per pair as the tile kernel's; states within fp32 tolerance."""
    from tc_gan_amd.clib import SolverParams, libssnode
    from tc_gan_amd.ssnode import fixed_points_batch
    import ctypes
    N, B, NB = 50, 301, 8                     # odd unit count as well
    Ws, exts = _inputs(N, B, NB, seed=21)
    rs = np.random.RandomState(3)
    extb = np.stack([exts * (0.2 + 1.6 * rs.rand()) for _ in range(B)])
    r0 = rs.rand(B, NB, 2 * N) * np.where(np.arange(B) % 3 == 0, 30.0, 0.0)[:, None, None]
    # (atol 1e-4: well above the fp32 rounding noise of |r1 - r0| at these rates -- up to 8e-6 at r = 100 -- so that
    # whether a pair converges does not depend on a kernel's summation order)
    kw = dict(r0=r0, max_iter=20000, atol=1e-4, dtype='float32', want_prev=True)
    a = fixed_points_batch(Ws, extb, P['k'], P['n'], variant=8, **kw)
    b = fixed_points_batch(Ws, extb, P['k'], P['n'], variant=2, **kw)
    np.testing.assert_array_equal(a.codes, b.codes)
    assert (a.codes == 0).all() and a.steps.min() < 0.7 * a.steps.max()      # really different stop steps
    assert np.abs(a.steps - b.steps).max() <= 0.2 * b.steps.max()
    # two kernels may see |r1 - r0| < atol first hold at different steps (fp32 summation order); every step between the
    # two stop steps moves an element by less than about atol, so that is the tolerance the stop rule itself leaves
    slack = 1e-3 + 1e-4 * (np.abs(a.steps - b.steps)[..., None] + 2)
    assert (np.abs(a.x - b.x) <= RTOL32 * np.abs(b.x) + slack).all()
    assert np.abs(a.x - a.x_prev).max() < 1e-3
    # and it is what the library picks by itself for this shape
    p = SolverParams(io_type=2, max_iter=100, k=P['k'], n=P['n'], tau_E=P['tau'][0], tau_I=P['tau'][1], dt=8e-4, atol=1e-5,
                     rate_soft_bound=200., rate_hard_bound=1000.)
    assert libssnode.ssn_solve_batch_variant_for(B, NB, 2 * 100, 4, ctypes.byref(p)) == 8
    assert libssnode.ssn_solve_batch_variant_for(200, NB, 2 * 100, 4, ctypes.byref(p)) == 6
    assert libssnode.ssn_solve_batch_variant_for(B, 1, 2 * 100, 4, ctypes.byref(p)) == 2


@pytest.mark.parametrize('variant', [2, 5, 6, 7, 8])
def test_fp32_solver_kernels_reach_the_fp64_fixed_points(variant):
    """The reference pins its fixed-time network against `sample_fixed_points(atol=1e-10)` at 1e-4 after 10000 steps
    (networks/tests/test_euler_ssn.py:79-86).  The same bar for the fp32 fixed-point solver kernels themselves, each
    FORCED (tile 2, fp32 MFMA 5, fp16-split wide 6 / alternating 7, two-draw form 8): 10000 Euler steps of the
    solver's own form (ssnode.c:64-67) from r = 0 on the fp32 weights against the fp64 fixed points of the same draws."""
    from tc_gan_amd import ssnode, stimuli, weight_gen
    N, B = 100, 3
    jds = on.new_JDS()
    zs, fps, _ = ssnode.sample_fixed_points(B, N=N, seed=N * B, io_type='asym_tanh', atol=1e-10, **jds)
    exts = stimuli.input(P['bandwidths'], np.linspace(-0.5, 0.5, N), P['smoothness'], P['contrast'], P['offset'])
    Ws = np.stack([weight_gen.generate_weight(N, jds['J'], jds['D'], jds['S'], z) for z in zs])
    res = ssnode.fixed_points_batch(Ws, exts, P['k'], P['n'], max_iter=10000, atol=0.0, io_type='asym_tanh',
                                    dtype='float32', variant=variant)
    np.testing.assert_array_equal(res.codes, 1)
    np.testing.assert_allclose(res.x, np.asarray(fps), rtol=1e-4, atol=1e-4)
