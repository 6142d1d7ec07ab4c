"""Pin the oracle (oracle/) against the reference's golden vectors.

CPU only.  Sources of truth: tests/golden/*.npz, produced by
tests/golden/make_golden.py from the reference's compiled C file, its numpy
modules and its MATLAB-derived .mat known-answer data.
"""
import numpy as np
import pytest

from conftest import golden
from oracle import ssn_numpy as on

P = on.DEFAULT_PARAMS


def test_io_functions_c_oracle_vs_reference_tables(oracle_lib):
    g = golden('io_tables.npz')
    args = (float(g['r0']), float(g['r1']), float(g['v0']), float(g['k']), float(g['n']))
    for code, name in ((0, 'io_pow'), (1, 'io_alin'), (2, 'io_atanh')):
        got = np.array([oracle_lib.oracle_io(code, x, *args) for x in g['xs']])
        np.testing.assert_array_equal(got, g[name])          # bit-exact: same libm calls
    got = np.array([oracle_lib.oracle_rate_to_volt(x, float(g['k']), float(g['n'])) for x in g['rates']])
    np.testing.assert_array_equal(got, g['rate_to_volt'])


def test_io_functions_numpy_forms_vs_reference_tables():
    # reference tolerance: tests/test_ssn.py:21 (atol 1e-12)
    g = golden('io_tables.npz')
    kw = dict(k=float(g['k']), n=float(g['n']), rate_soft_bound=float(g['r0']), rate_hard_bound=float(g['r1']))
    for io_type, name in (('asym_power', 'io_pow'), ('asym_linear', 'io_alin'), ('asym_tanh', 'io_atanh')):
        np.testing.assert_allclose(on.io_fun(g['xs'], io_type, **kw), g[name], rtol=0, atol=1e-12)
    np.testing.assert_allclose(on.rate_to_volt(g['rates'], kw['k'], kw['n']), g['rate_to_volt'], rtol=0, atol=1e-12)


@pytest.mark.parametrize('N', [5, 50])
def test_weight_and_stimulus_vs_reference(N):
    g = golden('weights_stimuli.npz')
    J, D, S = g['new_J'], g['new_D'], g['new_S']
    np.testing.assert_array_equal(on.new_JDS()['J'], J)
    np.testing.assert_array_equal(on.new_JDS()['D'], D)
    np.testing.assert_array_equal(on.new_JDS()['S'], S)
    z = g['z_N%d' % N]
    np.testing.assert_allclose(on.generate_weight(N, J, D, S, z), g['W_N%d' % N], rtol=1e-14, atol=0)
    np.testing.assert_allclose(on.generate_weight(N, P['J'], P['D'], P['S'], z), g['W_orig_N%d' % N], rtol=1e-14, atol=0)
    x = np.linspace(-.5, .5, N)
    np.testing.assert_allclose(on.stimulus_input(P['bandwidths'], x, P['smoothness'], P['contrast']),
                               g['stim_N%d' % N], rtol=1e-14, atol=1e-300)
    np.testing.assert_allclose(on.stimulus_input([0.25, 1.0], x, 0.1, [5., 20.], [0., 0.1]),
                               g['stim2_N%d' % N], rtol=1e-14, atol=1e-300)


def test_weight_vs_matlab_known_answer():
    # reference: tests/test_dynamics.py:43-67, atol 1e-6
    g = golden('matlab_ne51.npz')
    N = int(g['Ne'])
    W = on.generate_weight(N, g['J'], g['D'], g['S'], g['Z'])
    np.testing.assert_allclose(W, g['W'], atol=1e-6)


def _case_inputs(c, g):
    N = int(c['N'])
    key = 'W_%d' % c['id']
    if key in g.files:
        return g[key], g['ext_%d' % c['id']]
    jds = on.new_JDS()
    z = np.random.RandomState(int(c['seed'])).rand(2 * N, 2 * N)
    W = on.generate_weight(N, jds['J'], jds['D'], jds['S'], z)
    ext = on.stimulus_input([float(c['bw'])], np.linspace(-.5, .5, N), P['smoothness'], [20.])[0]
    return W, ext


def test_solver_c_oracle_vs_reference_end_states(oracle_lib):
    """Codes identical; end states in BOTH caller buffers match the reference build.
    (bit-exact except for re-association inside the `omp simd` dot product, hence 1e-12 rel.)"""
    g = golden('solver_cases.npz')
    seen = set()
    for c in g['cases']:
        W, ext = _case_inputs(c, g)
        M = W.shape[0]
        r0 = np.zeros(M)
        r1 = np.full(M, np.nan)
        import ctypes
        steps = ctypes.c_int(0)
        code = oracle_lib.oracle_solve_euler(
            on.IO_CODES[str(c['io_type'])], M // 2, on.ptr(np.ascontiguousarray(W)), on.ptr(np.ascontiguousarray(ext)),
            P['k'], P['n'], on.ptr(r0), on.ptr(r1), P['tau'][0], P['tau'][1], float(c['dt']),
            int(c['max_iter']), float(c['atol']), float(c['soft']), float(c['hard']), ctypes.byref(steps))
        assert code == int(c['code']), c
        seen.add(code)
        np.testing.assert_allclose(r0, g['r0_%d' % c['id']], rtol=1e-12, atol=1e-300)
        np.testing.assert_allclose(r1, g['r1_%d' % c['id']], rtol=1e-12, atol=1e-300, equal_nan=True)
        if float(c['atol']) == 0.0:
            assert steps.value == int(c['max_iter'])
    assert seen == {0, 1, 2}


def test_inf_blowup_returns_code_2():
    # reference: tests/test_dynamics.py:129-137
    g = golden('solver_cases.npz')
    assert int(g['inf_code']) == 2
    sol = on.fixed_point([[2., 0.], [0., 0.]], [10., 10.], k=1, n=1, r0=[0., 0.],
                         max_iter=10000000, io_type='asym_linear')
    assert sol.error == 2 and sol.message == 'Reached to rate_stop_at'
    np.testing.assert_array_equal(sol.x, g['inf_r0'])


@pytest.mark.parametrize('io_type', ['asym_linear', 'asym_power', 'asym_tanh'])
def test_tuning_curve_vs_matlab_known_answer(io_type):
    """reference: tests/test_dynamics.py:76-126 (rtol 0.1; measured 7.6e-4)."""
    g = golden('matlab_ne51.npz')
    N = int(g['Ne'])
    X = np.linspace(-0.5, 0.5, N)
    exts = on.stimulus_input(g['bandwidths'], X, float(g['smoothness']), [float(g['contrast'])])
    zs, fps, counter = on.find_fixed_points(
        1, iter([(None, g['W'])]), exts, k=float(g['k']), n=float(g['n']),
        r0=np.zeros(2 * N), io_type=io_type)
    assert not counter
    ET = g['E_Tuning']
    center, ofs = N // 2, len(ET) // 2
    actual = np.array([x[center - ofs:center + ofs + 1] for x in fps[0]]).T
    np.testing.assert_allclose(actual, ET, rtol=2e-3)
    # the inhibitory block of the same MATLAB solve (`I_Tuning` lies beside `E_Tuning` in the reference's asset; its own
    # test does not read it): neurons N + centre - 1 .. N + centre + 1, measured 5.9e-4
    actual_i = np.array([x[N + center - ofs:N + center + ofs + 1] for x in fps[0]]).T
    np.testing.assert_allclose(actual_i, g['I_Tuning'], rtol=2e-3)


def test_oracle_matches_reference_build_live(oracle_lib, reference_lib):
    """When oracle/_ref is present: random seeded solves, oracle vs reference C, all io types."""
    if reference_lib is None:
        pytest.skip('oracle/_ref/libssnode.so not built')
    jds = on.new_JDS()
    for seed, io_type in enumerate(['asym_power', 'asym_linear', 'asym_tanh'] * 2):
        N = 12 + seed
        z = np.random.RandomState(seed).rand(2 * N, 2 * N)
        W = on.generate_weight(N, jds['J'], jds['D'], jds['S'], z)
        ext = on.stimulus_input([0.5], np.linspace(-.5, .5, N), P['smoothness'], [20.])[0]
        a = on.fixed_point(W, ext, P['k'], P['n'], io_type=io_type, lib=oracle_lib)
        b = on.fixed_point(W, ext, P['k'], P['n'], io_type=io_type, lib=reference_lib)
        assert a.error == b.error
        np.testing.assert_allclose(a.x, b.x, rtol=1e-12)


def test_index_helpers_vs_reference():
    g = golden('index_helpers.npz')
    np.testing.assert_array_equal(on.sample_sites_from_stim_space([0, 0.5, 1], 101), g['sites_101'])
    np.testing.assert_array_equal(on.sample_sites_from_stim_space([-1, -0.5, 0, 0.5, 1], 201), g['sites_201'])
    np.testing.assert_array_equal(on.sample_sites_from_stim_space([-0.5, 0, 0.25], 100), g['sites_100'])
    for track in (False, True):
        for inh in (False, True):
            got = on.subsample_neurons(g['sub_in'], [2, 3, 4], track_offset_identity=track,
                                       include_inhibitory_neurons=inh)
            np.testing.assert_array_equal(got, g['sub_t%d_i%d' % (track, inh)])


def test_find_fixed_points_rejection_order():
    """A rejected draw is skipped, accepted draws keep draw order, stimuli come back un-reversed
    (ssnode.py:390-420)."""
    g = golden('solver_cases.npz')
    bad = [c for c in g['cases'] if c['code'] == 2 and ('W_%d' % c['id']) in g.files]
    good = [c for c in g['cases'] if c['code'] == 0 and c['N'] == 10 and c['io_type'] == 'asym_power']
    assert bad and good
    cb = bad[0]
    Wb = g['W_%d' % cb['id']]
    Wg, _ = _case_inputs(good[0], g)
    exts = on.stimulus_input([0.0625, 1.0], np.linspace(-.5, .5, 10), P['smoothness'], [40.])
    draws = iter([('g0', Wg), ('bad', Wb), ('g1', Wg)])
    zs, xs, counter = on.find_fixed_points(2, draws, exts, k=P['k'], n=P['n'], io_type='asym_power',
                                           dt=5e-4, max_iter=100000, rate_stop_at=200.)
    assert list(zs) == ['g0', 'g1']
    assert counter == {2: 1}
    assert xs.shape == (2, 2, 20)
    # stimulus order restored: the narrow stimulus drives less total activity than the wide one
    assert xs[0, 0].sum() < xs[0, 1].sum()
