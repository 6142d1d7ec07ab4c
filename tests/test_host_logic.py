"""Host-side driver logic on the CPU (no kernels): exit recording, abort guards, limiters, datastore naming, config
loading and the doctests of the host modules -- the scenarios of the reference's tests/test_exit.py:9-52,
tests/test_legacy_drivers.py:305-417, tests/test_execution.py and its ``--doctest-modules`` run."""
import doctest
import importlib
import json
from unittest import mock

import numpy as np
import pytest


def test_recording_exit_reason_cases():
    from tc_gan_amd.drivers import recording_exit_reason
    from tc_gan_amd.execution import KnownError
    ds = mock.Mock()
    with recording_exit_reason(ds):
        pass
    ds.save_exit_reason.assert_called_once_with(reason='end_of_iteration', good=True)

    ds = mock.Mock()
    with pytest.raises(KeyboardInterrupt):
        with recording_exit_reason(ds):
            raise KeyboardInterrupt
    ds.save_exit_reason.assert_called_once_with(reason='keyboard_interrupt', good=False)

    ds = mock.Mock()
    err = Exception('some exception')
    with pytest.raises(Exception):
        with recording_exit_reason(ds):
            raise err
    ds.save_exit_reason.assert_called_once_with(reason='uncaught_exception', good=False, exception=str(err))

    ds = mock.Mock()
    with pytest.raises(KnownError):
        with recording_exit_reason(ds):
            raise KnownError('message')
    ds.save_exit_reason.assert_not_called()


def test_check_disc_param_aborts_on_nan_parameters_only():
    from tc_gan_amd import drivers
    from tc_gan_amd.execution import KnownError

    class Net(object):
        def __init__(self, values):
            self.values = values

        def get_param_values(self):
            return self.values
    ds = mock.Mock()
    bad = Net([np.array([[np.nan, 1.0]]), np.zeros(3)])
    with pytest.raises(KnownError) as exc:
        drivers.check_disc_param(ds, bad, np.array([np.nan, 0.0]))
    assert exc.value.exit_code == 3
    (obj, fname), _ = ds.dump_json.call_args
    assert fname == 'exit.json' and obj['reason'] == 'disc_param_has_nan' and obj['good'] is False
    # finite parameters: a NaN norm alone (or finite norms) does not abort
    ds = mock.Mock()
    drivers.check_disc_param(ds, Net([np.ones(2)]), np.array([np.nan]))
    drivers.check_disc_param(ds, bad, np.array([1.0, 2.0]))
    ds.dump_json.assert_not_called()


def test_maybe_quit_on_JDS_distance():
    from tc_gan_amd import drivers, ssnode
    from tc_gan_amd.execution import KnownError
    true = [ssnode.DEFAULT_PARAMS[k] for k in 'JDS']
    far = [np.exp(np.array(m)) for m in (
        [[-2.725132882048388, -2.4531698490543286], [-2.1680251198864506, -3.1575330875403287]],
        [[-0.6751161156746839, -0.3826601625506246], [-0.34232003427022, -1.1335422836893538]],
        [[-2.7327504049935936, -4.210179719643937], [-1.9547447679855652, -3.5791486928972325]])]
    ds = mock.Mock()
    with pytest.raises(KnownError) as exc:
        drivers.maybe_quit(ds, JDS_fake=far, JDS_true=true, quit_JDS_threshold=0.4)
    assert exc.value.exit_code == 4
    ds.dump_json.assert_called_once()
    ds = mock.Mock()
    drivers.maybe_quit(ds, JDS_fake=list(np.array(true) + 0.01), JDS_true=true, quit_JDS_threshold=0.4)
    drivers.maybe_quit(ds, JDS_fake=far, JDS_true=true, quit_JDS_threshold=-1)       # disabled
    ds.dump_json.assert_not_called()


def test_rejection_limiter():
    from tc_gan_amd import drivers
    from tc_gan_amd.execution import KnownError
    lim = drivers.SSNRejectionLimiter(None, n_samples=10)
    over, under = 20, 0
    assert not lim.should_abort(over)
    lim.should_abort(under)                                    # reset
    assert not any(lim.should_abort(over) for _ in range(lim.max_consecutive_exceedings))
    assert not lim.should_abort(under)
    lim.should_abort(under)
    assert not any(lim.should_abort(over) for _ in range(lim.max_consecutive_exceedings))
    assert lim.should_abort(over)                              # one more in a row: abort
    ds = mock.Mock()
    lim = drivers.SSNRejectionLimiter(ds, n_samples=10)
    for _ in range(lim.max_consecutive_exceedings):
        lim(over)
    with pytest.raises(KnownError):
        lim(over)
    ds.dump_json.assert_called_once_with(dict(reason='too_many_rejections', good=False), 'exit.json')


@pytest.mark.parametrize('repeats, shifts, last_shift', [
    ([9], [+1], +1), ([9], [+1], -1), ([0, 5, 4], [-1, +1, -1], +1), ([1, 5, 3], [-1, +1, -1], +1),
    ([2, 5, 2], [-1, +1, -1], +1), ([3, 5, 1], [-1, +1, -1], +1), ([4, 5, 0], [-1, +1, -1], +1)])
def test_wgan_disc_loss_limiter(repeats, shifts, last_shift):
    from tc_gan_amd import drivers
    from tc_gan_amd.execution import KnownError
    ds = mock.Mock()
    lim = drivers.WGANDiscLossLimiter(ds, prob_limit=0.6 - 1e-5, hist_length=10)
    for num, shift in zip(repeats, shifts):
        for _ in range(num):
            lim(lim.wild_disc_loss + shift)                    # history not full yet: never aborts
    with pytest.raises(KnownError):
        lim(lim.wild_disc_loss + last_shift)
    ds.dump_json.assert_called_once_with(dict(reason='wild_disc_loss', good=False), 'exit.json')


def test_interval_and_datastore_helpers(tmp_path):
    from tc_gan_amd import drivers, execution
    assert [drivers.is_at_interval(s, 3) for s in range(5)] == [True, False, False, True, False]
    assert not drivers.is_at_interval(0, -1) and not drivers.is_at_interval(5, 0)
    assert execution.format_datastore('a={alpha}_L={layers_str}', dict(alpha=10, layers=[128, 64])) == 'a=10_L=128_64'
    assert execution.KnownError('x', exit_code=7).exit_code == 7 and execution.SuccessExit('ok').exit_code == 0
    cfg = tmp_path / 'run.json'
    cfg.write_text(json.dumps(dict(ssn_type='heteroin', V=[0.3, 0])))
    assert execution.load_any_file(str(cfg)) == dict(ssn_type='heteroin', V=[0.3, 0])
    with execution.DataStore(str(tmp_path)) as ds:
        ds.tables.saverow('TC_mean.csv', [1.5, 2.5])
        ds.tables.saverow('TC_mean.csv', '3.0,4.0')
        p = ds.path('disc_param', 'last.npz')
        ds.save_exit_reason(reason='end_of_iteration', good=True)
    assert (tmp_path / 'disc_param').is_dir() and p.endswith('last.npz')
    np.testing.assert_allclose(np.loadtxt(str(tmp_path / 'TC_mean.csv'), delimiter=','), [[1.5, 2.5], [3.0, 4.0]])
    assert json.load(open(str(tmp_path / 'exit.json'))) == dict(reason='end_of_iteration', good=True)


def test_gen_and_disc_option_prefixes():
    """utils/dicts.py:1-47 as used by run/bptt_cwgan.py:65-66."""
    from tc_gan_amd import utils
    rc = utils.subdict_by_prefix(dict(gen_learning_rate=0.1, gen_J_min=1e-3, disc_layers=[8], seqlen=4), 'gen_')
    assert rc == dict(gen=dict(learning_rate=0.1, J_min=1e-3), disc_layers=[8], seqlen=4)
    assert utils.csv_line(float)('0, 0.5') == [0.0, 0.5] and utils.csv_line(float)('') == []
    np.testing.assert_array_equal(utils.cartesian_product([1, 2], [3, 4, 5]),
                                  [[1, 1, 1, 2, 2, 2], [3, 4, 5, 3, 4, 5]])


@pytest.mark.parametrize('module', ['tc_gan_amd.execution', 'tc_gan_amd.networks.ssn', 'tc_gan_amd.networks.utils',
                                    'tc_gan_amd.gradient_expressions.utils', 'tc_gan_amd.utils'])
def test_doctests(module):
    """The reference runs ``--doctest-modules`` (pytest.ini:2-3); the host modules keep their doctests."""
    res = doctest.testmod(importlib.import_module(module), optionflags=doctest.NORMALIZE_WHITESPACE)
    assert res.failed == 0


def test_single_cell_minibatch_selection_keeps_the_random_stream():
    """With one (cell type, probe) pair the per-model `choice` loop is skipped: same minibatch, same RandomState."""
    import numpy as np
    from tc_gan_amd.networks import cwgan
    cls = [v for v in vars(cwgan).values() if isinstance(v, type) and hasattr(v, 'random_cells')][0]
    nested = np.random.RandomState(0).rand(50, 1, 1, 2, 8)
    cond_values = [[0], [0.0], [5.0, 20.0], np.linspace(0, 1, 8)]
    a = cls(nested, cond_values, e_ratio=0.8, seed=np.random.RandomState(7))
    b = cls(nested, cond_values, e_ratio=0.8, seed=np.random.RandomState(7))
    mb_a = a.select_minibatch(16, 1)
    # reference behaviour: the loop of rng.choice calls (cwgan.py:328-355)
    shape = (16, 1)
    ids_sample = b.rng.choice(len(nested), shape)
    ids = np.asarray([[[0, 0]][0:1] for _ in range(16)])
    for _ in range(16):
        assert b.rng.choice(1, 1, replace=False)[0] == 0
    ids_contrast = b.rng.choice(2, 16)
    np.testing.assert_array_equal(mb_a.tc_md, nested[ids_sample, 0, 0, ids_contrast.reshape(-1, 1)])
    sa, sb = a.rng.get_state(), b.rng.get_state()
    assert sa[2] == sb[2] and np.array_equal(sa[1], sb[1])


def test_init_distributed_is_a_no_op_for_a_single_process():
    from tc_gan_amd.execution import distributed_rank, init_distributed
    assert init_distributed({'WORLD_SIZE': '1'}) is False
    assert init_distributed({}) is False
    assert distributed_rank() == 0
