#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ from the REFERENCE itself.

Runs only in the build container (needs /root/reference); the fixtures it
writes are committed so that tests never read /root/reference at run time.

What is executed from the reference, and how:
  * tc_gan/ext/ssnode.c -- compiled unmodified by oracle/Makefile into
    oracle/_ref/libssnode.so and called through ctypes with the argtypes of
    tc_gan/clib.py:16-33.
  * tc_gan/stimuli.py, tc_gan/weight_gen.py,
    tc_gan/gradient_expressions/utils.py, tc_gan/networks/utils.py -- numpy-only
    modules, loaded by file path (importlib) where they lie.  The tc_gan
    package itself is not imported (its __init__ chain needs Theano, which is
    not installed; nothing is stubbed).
  * tc_gan/assets/*.mat -- the reference's own known-answer data (MATLAB
    model), read with scipy.io and re-saved as arrays.
Only inputs and expected outputs are stored -- no reference source text.
"""
import importlib.util
import os
import sys

import numpy as np
import scipy.io

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference/tc_gan'
sys.path.insert(0, ROOT)

from oracle import ssn_numpy as on  # noqa: E402  (bindings + constants only)


def load_by_path(name, relpath):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


ref_stimuli = load_by_path('ref_stimuli', 'stimuli.py')
ref_weight_gen = load_by_path('ref_weight_gen', 'weight_gen.py')
ref_ge_utils = load_by_path('ref_ge_utils', 'gradient_expressions/utils.py')
ref_net_utils = load_by_path('ref_net_utils', 'networks/utils.py')
ref = on.load_reference_lib()
assert ref is not None, 'run `make -C oracle` first'

P = on.DEFAULT_PARAMS
JDS = on.new_JDS()
SOLVERS = {'asym_power': ref.solve_dynamics_asym_power_euler,
           'asym_linear': ref.solve_dynamics_asym_linear_euler,
           'asym_tanh': ref.solve_dynamics_asym_tanh_euler}


def ref_solve(io_type, W, ext, r0, k=P['k'], n=P['n'], tau=P['tau'], dt=8e-4,
              max_iter=10000, atol=1e-5, soft=200., hard=1000.):
    W = np.ascontiguousarray(W, dtype='double')
    ext = np.ascontiguousarray(ext, dtype='double')
    r0 = np.array(r0, dtype='double')
    r1 = np.full_like(r0, np.nan)
    code = SOLVERS[io_type](W.shape[0] // 2, on.ptr(W), on.ptr(ext), k, n,
                            on.ptr(r0), on.ptr(r1), tau[0], tau[1], dt,
                            max_iter, atol, soft, hard)
    return code, r0, r1


def gen_io_tables():
    k, n, r0, r1 = P['k'], P['n'], 200., 1000.
    v0 = ref.rate_to_volt(r0, k, n)
    xs = np.linspace(-0.1, v0 * 3, 1000)          # tests/test_ssn.py:15
    out = dict(xs=xs, v0=v0, k=k, n=n, r0=r0, r1=r1)
    for name in ('io_pow', 'io_alin', 'io_atanh'):
        f = getattr(ref, name)
        out[name] = np.array([f(x, r0, r1, v0, k, n) for x in xs])
    rates = np.linspace(0, 1000, 1000)             # tests/test_ssn.py:59
    out['rates'] = rates
    out['rate_to_volt'] = np.array([ref.rate_to_volt(x, k, n) for x in rates])
    np.savez_compressed(os.path.join(HERE, 'io_tables.npz'), **out)


def gen_weights_stimuli():
    out = {}
    for N in (5, 50):
        z = np.random.RandomState(100 + N).rand(2 * N, 2 * N)
        out['z_N%d' % N] = z
        out['W_N%d' % N] = ref_weight_gen.generate_weight(N, JDS['J'], JDS['D'], JDS['S'], z)
        out['W_orig_N%d' % N] = ref_weight_gen.generate_weight(N, P['J'], P['D'], P['S'], z)
        x = np.linspace(-.5, .5, N)
        out['stim_N%d' % N] = ref_stimuli.input(P['bandwidths'], x, P['smoothness'], P['contrast'])
        out['stim2_N%d' % N] = ref_stimuli.input([0.25, 1.0], x, 0.1, [5., 20.], [0., 0.1])
    for key in 'JDS':
        out['new_' + key] = JDS[key]
    np.savez_compressed(os.path.join(HERE, 'weights_stimuli.npz'), **out)


def gen_solver_cases():
    """End states/codes of the reference C solver on seeded inputs."""
    out = {}
    cases = []
    cid = 0
    # (a) fixed-step runs (atol=0 -> exactly max_iter steps, code 1), even max_iter
    for N, T in ((1, 50), (10, 200), (50, 500), (100, 400)):
        for io_type in ('asym_power', 'asym_linear', 'asym_tanh'):
            seed = 1000 + cid
            z = np.random.RandomState(seed).rand(2 * N, 2 * N)
            W = ref_weight_gen.generate_weight(N, JDS['J'], JDS['D'], JDS['S'], z)
            x = np.linspace(-.5, .5, N)
            ext = ref_stimuli.input([1.0], x, P['smoothness'], [20.])[0]
            code, r0, r1 = ref_solve(io_type, W, ext, np.zeros(2 * N), max_iter=T, atol=0.0)
            cases.append((cid, N, io_type, seed, 1.0, T, 0.0, 8e-4, 200., 1000., code))
            out['r0_%d' % cid] = r0
            out['r1_%d' % cid] = r1
            cid += 1
    # (b) odd max_iter (buffer-parity quirk), N=10
    for io_type in ('asym_power', 'asym_tanh'):
        N, T, seed = 10, 101, 2000 + cid
        z = np.random.RandomState(seed).rand(2 * N, 2 * N)
        W = ref_weight_gen.generate_weight(N, JDS['J'], JDS['D'], JDS['S'], z)
        x = np.linspace(-.5, .5, N)
        ext = ref_stimuli.input([1.0], x, P['smoothness'], [20.])[0]
        code, r0, r1 = ref_solve(io_type, W, ext, np.zeros(2 * N), max_iter=T, atol=0.0)
        cases.append((cid, N, io_type, seed, 1.0, T, 0.0, 8e-4, 200., 1000., code))
        out['r0_%d' % cid] = r0
        out['r1_%d' % cid] = r1
        cid += 1
    # (c) converged runs with default solver settings (code 0) over the 8 default bandwidths
    for N in (10, 50):
        for io_type in ('asym_power', 'asym_linear', 'asym_tanh'):
            for bw in (0.0625, 1.0):
                seed = 3000 + cid
                z = np.random.RandomState(seed).rand(2 * N, 2 * N)
                W = ref_weight_gen.generate_weight(N, JDS['J'], JDS['D'], JDS['S'], z)
                x = np.linspace(-.5, .5, N)
                ext = ref_stimuli.input([bw], x, P['smoothness'], [20.])[0]
                hard = np.inf if io_type != 'asym_tanh' else 1000.
                code, r0, r1 = ref_solve(io_type, W, ext, np.zeros(2 * N), max_iter=100000,
                                         atol=1e-5, dt=8e-4, hard=hard)
                cases.append((cid, N, io_type, seed, bw, 100000, 1e-5, 8e-4, 200., hard, code))
                out['r0_%d' % cid] = r0
                out['r1_%d' % cid] = r1
                cid += 1
    # (d) blow-up with the ORIGINAL (less stable) J, D at large D: rate_stop_at=200 (dataset.py:46-49)
    N = 10
    for seed in range(4000, 4006):
        z = np.random.RandomState(seed).rand(2 * N, 2 * N)
        W = ref_weight_gen.generate_weight(N, P['J'] * 3, P['D'] * 3, P['S'], z)
        x = np.linspace(-.5, .5, N)
        ext = ref_stimuli.input([1.0], x, P['smoothness'], [40.])[0]
        code, r0, r1 = ref_solve('asym_power', W, ext, np.zeros(2 * N), max_iter=100000,
                                 atol=1e-5, dt=5e-4, hard=200.)
        cases.append((cid, N, 'asym_power', seed, 1.0, 100000, 1e-5, 5e-4, 200., 200., code))
        out['W_%d' % cid] = W
        out['ext_%d' % cid] = ext
        out['r0_%d' % cid] = r0
        out['r1_%d' % cid] = r1
        cid += 1
    out['cases'] = np.array(
        cases, dtype=[('id', int), ('N', int), ('io_type', 'U16'), ('seed', int), ('bw', float),
                      ('max_iter', int), ('atol', float), ('dt', float), ('soft', float),
                      ('hard', float), ('code', int)])
    # (e) tests/test_dynamics.py:129-137 (test_inf): linear blow-up must return code 2
    code, r0, r1 = ref_solve('asym_linear', [[2., 0.], [0., 0.]], [10., 10.], [0., 0.],
                             k=1., n=1., max_iter=10000000, atol=1e-5, hard=np.inf)
    out['inf_code'] = code
    out['inf_r0'] = r0
    out['inf_r1'] = r1
    np.savez_compressed(os.path.join(HERE, 'solver_cases.npz'), **out)
    print('solver cases:', len(cases), 'codes:', sorted(set(c[-1] for c in cases)), 'inf:', code)


def gen_matlab():
    """tests/test_dynamics.py:43-126: the MATLAB known-answer data."""
    cp = scipy.io.loadmat(os.path.join(REF, 'assets', 'target_parameters_GAN-SSN_Ne51-Zs.mat'))
    mp = scipy.io.loadmat(os.path.join(REF, 'assets', 'training_data_TCs_Ne51-Zs.mat'))
    mz = cp['Zs']
    N = mz.shape[0]
    Z = np.zeros((2 * N, 2 * N))
    Z[:N, :N] = mz[:, :, 0, 0]
    Z[N:, :N] = mz[:, :, 1, 0]
    Z[:N, N:] = mz[:, :, 0, 1]
    Z[N:, N:] = mz[:, :, 1, 1]
    mpar = mp['Modelparams'][0, 0]
    L = mpar['L'][0, 0]
    np.savez_compressed(
        os.path.join(HERE, 'matlab_ne51.npz'),
        Z=Z,
        J=cp['Targetparams']['Jlow'][0, 0],
        D=cp['Targetparams']['dJ'][0, 0],
        S=cp['Targetparams']['sigmas'][0, 0] / 8,
        W=cp['W'].toarray(),
        bandwidths=mpar['bandwidths'][0] / L,
        smoothness=mpar['l_margin'][0, 0] / L,
        contrast=float(mpar['c'][0, 0]),
        Ne=int(mpar['Ne'][0, 0]),
        k=float(mpar['k'][0, 0]),
        n=float(mpar['n'][0, 0]),
        E_Tuning=mp['E_Tuning'],
        # the inhibitory block of the same solve (the reference's own test reads only E_Tuning; same file, same layout)
        I_Tuning=mp['I_Tuning'],
    )


def gen_index_helpers():
    out = {}
    out['sites_101'] = np.array(ref_ge_utils.sample_sites_from_stim_space([0, 0.5, 1], 101))
    out['sites_201'] = np.array(ref_ge_utils.sample_sites_from_stim_space([-1, -0.5, 0, 0.5, 1], 201))
    out['sites_100'] = np.array(ref_ge_utils.sample_sites_from_stim_space([-0.5, 0, 0.25], 100))
    N, NZ, NB = 7, 5, 3
    rv = np.arange(NZ * NB * 2 * N, dtype=float).reshape((NZ, NB, 2 * N))
    out['sub_in'] = rv
    for track in (False, True):
        for inh in (False, True):
            out['sub_t%d_i%d' % (track, inh)] = ref_ge_utils.subsample_neurons(
                rv, [2, 3, 4], track_offset_identity=track, include_inhibitory_neurons=inh)
    shape = (11, 5, 7, 2, 3)
    data = np.arange(np.prod(shape)).reshape((11, -1))
    out['grid_in'] = data
    out['grid_out'] = ref_net_utils.gridify_tc_samples(
        data, num_contrasts=5, num_bandwidths=7, num_cell_types=2, num_probes=3)
    np.savez_compressed(os.path.join(HERE, 'index_helpers.npz'), **out)


if __name__ == '__main__':
    gen_io_tables()
    gen_weights_stimuli()
    gen_solver_cases()
    gen_matlab()
    gen_index_helpers()
    for f in sorted(os.listdir(HERE)):
        if f.endswith('.npz'):
            print(f, os.path.getsize(os.path.join(HERE, f)))
