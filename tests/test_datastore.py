"""Run-directory formats on the CPU: typed tables (CSV backend when h5py is absent), info/exit JSON, and
`tc_gan_amd.loaders.load_records` reading them back under the reference's attribute names."""
import json

import numpy as np


def test_typed_tables_roundtrip_through_load_records(tmp_path):
    from tc_gan_amd.execution import DataStore
    from tc_gan_amd.loaders import load_records
    from tc_gan_amd.recorders import GenMomentsRecorder, MMLearningRecorder
    from tc_gan_amd.utils import Namespace
    d = str(tmp_path)
    with DataStore(d) as ds:
        ds.dump_json(dict(run_config=dict(truth_size=10, batchsize=5), extra_info={}, meta_info={}), 'info.json')
        np.save(ds.path('truth.npy'), np.arange(6.).reshape(2, 3))
        lr = MMLearningRecorder.make(ds)
        gm = GenMomentsRecorder.make(ds, 2)
        ds.h5.tables.create_table('generator', np.dtype([('gen_step', 'uint32')] + [(n, 'double') for n in (
            'V', 'J_EE', 'J_EI', 'J_IE', 'J_II', 'D_EE', 'D_EI', 'D_IE', 'D_II', 'S_EE', 'S_EI', 'S_IE', 'S_II')]))
        for step in range(3):
            info = Namespace(loss=1.5 / (step + 1), rate_penalty=0.0, dynamics_penalty=float('nan'), train_time=0.25,
                             gen_moments=np.array([[1., 2.], [3., 4.]]) * step)
            lr.record(step, info)
            gm.record(step, info)
            ds.h5.tables.saverow('generator', np.array(tuple([step, 0.5] + list(np.arange(12.) + step)),
                                                       dtype=ds.h5.tables._dtype['generator']))
        ds.flush_all()
        ds.save_exit_reason(reason='end_of_iteration', good=True)
    rec = load_records(d + '/info.json')
    assert rec.exit == dict(reason='end_of_iteration', good=True)
    assert list(rec.learning.columns) == ['step', 'loss', 'rate_penalty', 'dynamics_penalty', 'train_time', 'epoch']
    np.testing.assert_allclose(rec.learning['loss'], [1.5, 0.75, 0.5])
    assert np.isnan(rec.learning['dynamics_penalty']).all()
    np.testing.assert_allclose(rec.learning['epoch'], [0, 0.5, 1.0])
    assert list(rec.gen_moments.columns)[:5] == ['step', 'mean_0', 'mean_1', 'var_0', 'var_1']
    np.testing.assert_allclose(rec.gen_moments.iloc[2][['mean_0', 'mean_1', 'var_0', 'var_1']], [2, 4, 6, 8])
    p = rec.gen_params_at(-1)
    assert p['V'] == 0.5
    np.testing.assert_allclose(p['J'], np.array([[2., 3.], [4., 5.]]))
    np.testing.assert_allclose(p['S'], np.array([[10., 11.], [12., 13.]]))
    assert rec.truth.shape == (2, 3)
    assert json.load(open(d + '/info.json'))['run_config']['truth_size'] == 10
