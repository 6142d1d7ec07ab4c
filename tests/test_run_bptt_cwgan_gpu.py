"""CLI integration on the GPU, after run/tests/test_bptt_cwgan.py:7-51 + test_bptt_wgan.py:12-105: one
generator step through ``main([...])`` in a scratch directory; checks info.json, exit.json, truth.npy, the typed
tables and their column names."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _load_tables(directory, store):
    """Typed tables of a run: store.hdf5 / <table>.hdf5 with h5py, <table>.csv without (structured arrays)."""
    path_h5 = os.path.join(directory, store + '.hdf5')
    if os.path.exists(path_h5):
        import h5py
        with h5py.File(path_h5, 'r') as f:
            return {k: f[k][...] for k in f}
    names = ([store] if store != 'store' else
             [n[:-4] for n in os.listdir(directory) if n.endswith('.csv') and n[:-4] in
              ('learning', 'disc_learning', 'generator', 'disc_param_stats')])
    return {n: np.genfromtxt(os.path.join(directory, n + '.csv'), delimiter=',', names=True, dtype=None, ndmin=1)
            for n in names}


@pytest.mark.parametrize('args', [
    [],
    ['--num-models', '1'],
    ['--sample-sites', '0, 0.5', '--probes-per-model', '2'],
    ['--contrasts', '5, 20'],
    ['--include-inhibitory-neurons', '--dataset-provider', 'fixedtime'],
    ['--disc-normalization', 'layer'],
    ['--disc-nonlinearity', 'tanh'],
    ['--disc-nonlinearity', 'elu', '--disc-normalization', 'layer'],
    ['--ssn-type', 'heteroin', '--dataset-provider', 'fixedtime'],
    ['--ssn-type', 'deg-heteroin', '--dataset-provider', 'fixedtime', '--include-inhibitory-neurons'],
])
def test_single_g_step(args, tmp_path, monkeypatch):
    from tc_gan_amd.run import bptt_cwgan
    monkeypatch.chdir(tmp_path)
    bptt_cwgan.main(['--iterations', '1', '--truth_size', '1', '--num-models', '2', '--n_bandwidths', '1',
                     '--WGAN_n_critic0', '1', '--seqlen', '4', '--skip-steps', '2',
                     '--tc-stats-record-interval', '1', '--datastore', 'results', '--disc-layers', '[8]',
                     '--J0', '0.1', '--D0', '0.05', '--S0', '0.1', '--quiet'] + args)
    out = tmp_path / 'results'
    info = json.load(open(out / 'info.json'))
    assert info['extra_info']['script_file'] == bptt_cwgan.__file__
    assert info['run_config']['bandwidths'] == [0.0625]
    assert 'true_ssn_options' in info['run_config']
    assert json.load(open(out / 'exit.json')) == dict(reason='end_of_iteration', good=True)
    assert np.load(out / 'truth.npy').shape[0] == 1
    tables = _load_tables(str(out), 'store')
    assert set(tables) == {'learning', 'disc_learning', 'generator', 'disc_param_stats'}
    assert tables['learning'].dtype.names == ('gen_step', 'Gloss', 'Dloss', 'Daccuracy', 'gen_forward_time',
                                              'gen_train_time', 'disc_time', 'rate_penalty', 'dynamics_penalty')
    vnames = {'heteroin': ('V_E', 'V_I'), 'deg-heteroin': ('V',)}.get(
        args[args.index('--ssn-type') + 1] if '--ssn-type' in args else 'default', ())
    assert tables['generator'].dtype.names == ('gen_step',) + vnames + (
        'J_EE', 'J_EI', 'J_IE', 'J_II', 'D_EE', 'D_EI', 'D_IE', 'D_II', 'S_EE', 'S_EI', 'S_IE', 'S_II')
    assert len(tables['learning']) == 1 and len(tables['disc_learning']) == 1
    assert np.isfinite(tables['learning']['Gloss']).all()
    tc = _load_tables(str(out), 'tc_stats')['tc_stats']
    assert tc.dtype.names[:6] == ('gen_step', 'is_fake', 'contrast', 'norm_probe', 'cell_type', 'count')
    assert set(tc['is_fake']) == {0, 1}
    assert os.path.exists(out / 'TC_mean.csv')
    assert os.path.exists(out / 'disc_param' / 'last.npz')
    npz = np.load(out / 'disc_param' / 'last.npz')
    # (a layer-normalised layer with a non-rectify nonlinearity carries the reference's ScaleLayer: simple_discriminator.py:57-75)
    scaled = '--disc-normalization' in args and '--disc-nonlinearity' in args
    assert int(npz['version']) == 1 and list(npz['param_names']) == (['W', 'scales', 'b', 'W'] if scaled else ['W', 'b', 'W'])
    assert len(tables['disc_param_stats'].dtype.names) == 2 + (4 if scaled else 3)


def test_known_error_exit_code(tmp_path, monkeypatch):
    """--quit-JDS-threshold tiny -> KnownError(exit_code=4) and exit.json with reason JDS_distance
    (drivers.py:183-198), surfaced as the process exit code by run.py."""
    import run as run_script
    monkeypatch.chdir(tmp_path)
    code = run_script.main(['tc_gan.run.bptt_cwgan', '--', '--iterations', '2', '--truth_size', '1', '--num-models', '1',
                            '--n_bandwidths', '1', '--WGAN_n_critic0', '1', '--seqlen', '4', '--skip-steps', '2',
                            '--datastore', 'results', '--quit-JDS-threshold', '1e-9', '--quiet',
                            '--dataset-provider', 'fixedtime'])
    assert code == 4
    assert json.load(open(tmp_path / 'results' / 'exit.json'))['reason'] == 'JDS_distance'


def test_cli_checkpoint_and_resume(tmp_path, monkeypatch):
    from tc_gan_amd.run import bptt_cwgan
    monkeypatch.chdir(tmp_path)
    common = ['--truth_size', '2', '--num-models', '2', '--n_bandwidths', '1', '--WGAN_n_critic0', '1', '--seqlen', '4',
              '--skip-steps', '2', '--disc-layers', '[8]', '--quiet', '--dataset-provider', 'fixedtime']
    bptt_cwgan.main(['--iterations', '2', '--datastore', 'first', '--checkpoint-interval', '1'] + common)
    assert os.path.exists(tmp_path / 'first' / 'checkpoint.pkl')
    bptt_cwgan.main(['--iterations', '4', '--datastore', 'second', '--resume-from', str(tmp_path / 'first' / 'checkpoint.pkl')]
                    + common)
    tables = _load_tables(str(tmp_path / 'second'), 'store')
    assert list(tables['learning']['gen_step']) == [2, 3]
    assert json.load(open(tmp_path / 'second' / 'exit.json'))['good']


def test_two_rank_cli_run_follows_the_single_process_run(tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 2 run.py tc_gan.run.bptt_cwgan -- ...` (gloo, both ranks on
    the one card of the test box): models sharded over the ranks, one all-reduce per update, every rank logging the
    same run -- which is the run a single process makes with the same seeds (host-side noise)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ['tc_gan.run.bptt_cwgan', '--', '--iterations', '2', '--truth_size', '4', '--num-models', '4',
              '--n_bandwidths', '4', '--WGAN_n_critic0', '2', '--WGAN_n_critic', '2', '--seqlen', '30',
              '--skip-steps', '20', '--disc-layers', '[16]', '--disc-precision', 'fp32', '--quiet',
              '--dataset-provider', 'fixedtime', '--J0', '0.1', '--D0', '0.05', '--S0', '0.1']
    env = dict(os.environ, TCGAN_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('WORLD_SIZE', None)
    subprocess.run([sys.executable, os.path.join(root, 'run.py')] + common + ['--datastore', str(tmp_path / 'one')],
                   check=True, env=env, cwd=str(tmp_path), timeout=300)
    port = str(26700 + os.getpid() % 1000)
    subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                    '--master-addr', '127.0.0.1', '--master-port', port, os.path.join(root, 'run.py')] + common +
                   ['--datastore', str(tmp_path / 'two')], check=True, env=env, cwd=str(tmp_path), timeout=300)
    one = _load_tables(str(tmp_path / 'one'), 'store')
    for sub in ('two', os.path.join('two', 'rank1')):
        two = _load_tables(str(tmp_path / sub), 'store')
        for name in one['generator'].dtype.names:
            np.testing.assert_allclose(two['generator'][name], one['generator'][name], rtol=2e-4, atol=1e-7)
        np.testing.assert_allclose(two['learning']['Dloss'], one['learning']['Dloss'], rtol=2e-4, atol=2e-5)
        assert json.load(open(tmp_path / sub / 'exit.json'))['good'] is True


_ONE_STEP = ['--iterations', '1', '--truth_size', '1', '--num-models', '2', '--n_bandwidths', '1', '--WGAN_n_critic0', '1',
             '--seqlen', '4', '--skip-steps', '2', '--tc-stats-record-interval', '1', '--datastore', 'results', '--quiet']


@pytest.mark.parametrize('args, config', [
    ([], dict(ssn_type='heteroin')),
    (['--include-inhibitory-neurons'], dict(ssn_type='heteroin')),
    (['--include-inhibitory-neurons'], dict(ssn_type='heteroin', V=[0.3, 0])),
    (['--include-inhibitory-neurons'], dict(ssn_type='heteroin', gen_V_min=[0, 0], gen_V_max=[1, 0])),
    ([], dict(ssn_type='deg-heteroin')),
    (['--include-inhibitory-neurons'], dict(ssn_type='deg-heteroin', V=0.5)),
])
def test_single_g_step_with_load_config(args, config, tmp_path, monkeypatch):
    """run/tests/test_bptt_cwgan.py:36-55 + test_bptt_wgan.py:80-105: the configuration comes from a JSON file given with
    --load-config (keys that have no command-line option: V, gen_V_min, gen_V_max), merged over the command line."""
    from tc_gan_amd.run import bptt_cwgan
    monkeypatch.chdir(tmp_path)
    config = dict(config, dataset_provider='fixedtime')
    with open(tmp_path / 'run.json', 'w') as fp:
        json.dump(config, fp)
    bptt_cwgan.main(_ONE_STEP + args + ['--load-config', str(tmp_path / 'run.json')])
    out = tmp_path / 'results'
    info = json.load(open(out / 'info.json'))
    for key, value in config.items():
        assert info['run_config'][key] == value
    assert json.load(open(out / 'exit.json')) == dict(reason='end_of_iteration', good=True)
    gen = _load_tables(str(out), 'store')['generator']
    vnames = {'heteroin': ('V_E', 'V_I'), 'deg-heteroin': ('V',)}[config['ssn_type']]
    assert gen.dtype.names == ('gen_step',) + vnames + ('J_EE', 'J_EI', 'J_IE', 'J_II', 'D_EE', 'D_EI', 'D_IE', 'D_II',
                                                       'S_EE', 'S_EI', 'S_IE', 'S_II')
    assert len(gen) == 1 and all(np.isfinite(gen[n]).all() for n in gen.dtype.names)
    if 'V' in config:        # the configured start value, one adam-wgan step (lr 0.01: +-0.01 per entry) later
        np.testing.assert_allclose([gen[n][0] for n in vnames], np.atleast_1d(config['V']), rtol=0, atol=0.0101)
    if 'gen_V_max' in config:                                   # V_I is pinned to [0, 0]
        assert gen['V_I'][0] == 0.0


def test_paper_run_json_key_set_through_load_config(tmp_path, monkeypatch):
    """The reference's realistic entry point is `./run tc_gan.run.bptt_cwgan --load-config run.json --datastore .`
    (scripts/fig4/gan/run.sh) with the key set of scripts/fig4/gan/run.json: nested true_ssn_options with V and
    unroll_scan, truth_batchsize, per-layer disc_normalization, disc_reg_l2_decay, rmsprop for both players, ...
    Same keys here, sizes cut down for a two-step run (num_sites, num_models, truth sizes, iterations, seqlen)."""
    from tc_gan_amd.run import bptt_cwgan
    monkeypatch.chdir(tmp_path)
    config = {"dataset_provider": "fixedtime",
              "true_ssn_options": {"J": [[0.3, 0.5], [0.4, 0.2]], "D": [[0.3, 0.4], [0.6, 0.19]],
                                   "S": [[0.15, 0.025], [0.1, 0.025]], "V": 0.1, "unroll_scan": False},
              "truth_batchsize": 4, "truth_size": 8, "iterations": 2, "quiet": True, "S0": 0.3, "contrasts": [20],
              "gen_dynamics_cost": 0, "gen_learning_rate": 0.0001, "gen_rate_cost": 100, "gen_update_name": "rmsprop",
              "include_inhibitory_neurons": False, "n_bandwidths": 8, "norm_probes": [0], "num_models": 4, "num_sites": 21,
              "probes_per_model": 1, "ssn_type": "deg-heteroin", "unroll_scan": True, "seqlen": 24, "skip_steps": 20,
              "tau_E": 2, "disc_layers": [128, 128, 128, 128], "disc_learning_rate": 0.02,
              "disc_normalization": ["none", "layer", "layer", "layer"], "disc_param_save_on_error": True,
              "disc_rate_penalty_bound": 1, "disc_reg_l2_decay": 0.001, "disc_update_name": "rmsprop"}
    with open(tmp_path / 'run.json', 'w') as fp:
        json.dump(config, fp)
    bptt_cwgan.main(['--load-config', 'run.json', '--datastore', '.', '--WGAN_n_critic0', '2'])
    info = json.load(open(tmp_path / 'info.json'))
    for key, value in config.items():
        if key not in ('S0', 'n_bandwidths'):                       # (rewritten by the run script's preprocessing, below)
            assert info['run_config'][key] == value, key
    assert info['run_config']['S0'] == [[0.3, 0.3], [0.3, 0.3]] and len(info['run_config']['bandwidths']) == 8
    assert json.load(open(tmp_path / 'exit.json')) == dict(reason='end_of_iteration', good=True)
    tables = _load_tables(str(tmp_path), 'store')
    assert list(tables['learning']['gen_step']) == [0, 1]
    assert tables['generator'].dtype.names[:2] == ('gen_step', 'V')
    assert np.isfinite(tables['learning']['Gloss']).all() and np.isfinite(tables['learning']['Dloss']).all()
    assert np.load(tmp_path / 'truth.npy').shape == (8, 8)
    # the critic has the per-layer normalisation of the file: no bias-free difference in parameter names, 4 hidden layers
    npz = np.load(tmp_path / 'disc_param' / 'last.npz')
    assert list(npz['param_names']).count('W') == 5
