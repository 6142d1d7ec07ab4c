"""The option table of the three run scripts (tc_gan_amd/run/options.py): names, aliases and defaults are the reference's
(tc_gan/run/bptt_wgan.py:46-172, bptt_cwgan.py:17-53, bptt_moments.py:36-101, run/gan.py:1109-1152, execution.py:290-317)."""
import pytest


def _parsers():
    from tc_gan_amd.run import bptt_cwgan, bptt_moments, bptt_wgan
    return {'w': bptt_wgan.make_parser(), 'c': bptt_cwgan.make_parser(), 'm': bptt_moments.make_parser()}


def test_table_is_well_formed():
    from tc_gan_amd.run import options
    for o in options.OPTIONS:
        assert set(o['scripts']) <= set('wcm') and o['flags'] and all(f.startswith('--') for f in o['flags'])
        assert o.get('type', 'str') in options.TYPES
    for script in 'wcm':
        flags = [f for o in options.options_of(script) for f in o['flags']]
        assert len(flags) == len(set(flags)), script                    # no flag twice in one script


def test_reference_defaults_and_aliases():
    p = _parsers()
    c = vars(p['c'].parse_args([]))
    # bptt_wgan.py:75-172 / bptt_cwgan.py:17-53 defaults
    assert (c['num_models'], c['probes_per_model'], c['norm_probes'], c['tc_stats_record_interval']) == (15, 1, [0], 100)
    assert (c['truth_size'], c['truth_seed'], c['seqlen'], c['skip_steps'], c['contrasts']) == (1000, 42, 1200, 1000, [20])
    assert (c['gen_learning_rate'], c['disc_learning_rate'], c['gen_update_name'], c['disc_update_name']) == (0.01, 0.01, 'adam-wgan', 'adam-wgan')
    assert (c['lipschitz_cost'], c['critic_iters_init'], c['critic_iters'], c['disc_layers'], c['disc_normalization'],
            c['disc_nonlinearity']) == (10.0, 50, 5, [], 'none', 'rectify')
    assert (c['J0'], c['D0'], c['S0'], c['gen_J_min'], c['gen_S_max'], c['gen_dynamics_cost']) == (0.01, 0.01, 0.01, 1e-3, 10, 1)
    assert (c['iterations'], c['quit_JDS_threshold'], c['disc_param_save_interval'], c['disc_param_template'], c['n_bandwidths']) == \
        (100000, -1, 5, 'last.npz', 4)
    assert c['datastore_template'] == 'logfiles/BPTT_CWGAN_{layers_str}' and c['datastore'] is None and c['load_config'] is None
    assert (c['z_device_seed'], c['z_host_draw'], c['gen_kernel'], c['disc_precision']) == (None, False, 'auto', 'fp32')
    # the reference's aliases
    a = vars(p['c'].parse_args(['--WGAN_n_critic0', '7', '--WGAN_n_critic', '2', '--WGAN_lambda', '3', '--layers', '[8, 8]',
                                '--sample-sites', '0, 0.5', '--contrast', '5, 20', '--gen-learn-rate', '0.1', '--debug']))
    assert (a['critic_iters_init'], a['critic_iters'], a['lipschitz_cost'], a['disc_layers'], a['norm_probes'], a['contrasts'],
            a['gen_learning_rate'], a['datastore_template']) == (7, 2, 3.0, [8, 8], [0.0, 0.5], [5.0, 20.0], 0.1, 'logfiles/debug')
    w = vars(p['w'].parse_args(['--n_samples', '4']))
    assert w['batchsize'] == 4 and w['sample_sites'] == [0] and w['datastore_template'] == 'logfiles/BPTT_WGAN_{layers_str}'
    assert 'num_models' not in w and 'tc_stats_record_interval' not in w
    m = vars(p['m'].parse_args([]))
    assert (m['lam'], m['moment_weights_regularization'], m['learning_rate'], m['update_name'], m['dynamics_cost'],
            m['gen_moments_record_interval'], m['J_min'], m['datastore_template']) == (.1, 1e-3, 0.01, 'adam-wgan', 1, 100, 1e-3, 'logfiles/BPTT_MM_{lam}')
    assert 'disc_layers' not in m and 'critic_iters' not in m and 'gen_J_min' not in m


@pytest.mark.parametrize('script,bad', [('c', ['--disc-normalization', 'batch']), ('m', ['--moment-weight-type', 'nope']),
                                        ('w', ['--n_bandwidths', '3']), ('c', ['--gen-kernel', 'cuda'])])
def test_choices_are_enforced(script, bad):
    with pytest.raises(SystemExit):
        _parsers()[script].parse_args(bad)


def test_table_dtypes_are_the_reference_columns():
    """tc_gan/recorders.py:113-362: table names, column names and dtypes (what the reference's loaders read)."""
    import numpy as np
    from tc_gan_amd import recorders as r
    assert r.LearningRecorder.dtype.names == ('gen_step', 'Gloss', 'Dloss', 'Daccuracy', 'gen_forward_time', 'gen_train_time',
                                              'disc_time', 'rate_penalty', 'dynamics_penalty')
    assert r.LearningRecorder.tablename == r.MMLearningRecorder.tablename == 'learning'
    assert r.MMLearningRecorder.dtype.names == ('step', 'loss', 'rate_penalty', 'dynamics_penalty', 'train_time')
    assert r.DiscLearningRecorder.dtype == np.dtype([('gen_step', 'uint32'), ('disc_step', 'uint32'), ('Dloss', 'double'),
                                                     ('Daccuracy', 'double'), ('SSsolve_time', 'double'), ('gradient_time', 'double'),
                                                     ('model_convergence', 'uint32'), ('model_unused', 'uint32')])
    assert r.table_dtype('gen_moments', range(2)).names == ('step', 'mean_0', 'mean_1', 'var_0', 'var_1')
    tc = r.table_dtype('tc_stats', range(3))
    assert tc.names == ('gen_step', 'is_fake', 'contrast', 'norm_probe', 'cell_type', 'count', 'mean_0', 'mean_1', 'mean_2',
                        'var_0', 'var_1', 'var_2')
    assert tc['is_fake'] == np.dtype('b') and tc['cell_type'] == np.dtype('uint16') and tc['count'] == np.dtype('uint32')
    names = list(r.DiscParamStatsRecorder.disc_param_unique_names(['W', 'scales', 'b', 'W', 'b', 'W']))
    assert names == ['W.nnorm.0', 'scales.nnorm.0', 'b.nnorm.0', 'W.nnorm.1', 'b.nnorm.1', 'W.nnorm.2']
    assert r.table_dtype('disc_param_stats', names).names == ('gen_step', 'disc_step') + tuple(names)
    assert r.table_dtype('generator', ['J_EE', 'J_EI']).names == ('gen_step', 'J_EE', 'J_EI')
