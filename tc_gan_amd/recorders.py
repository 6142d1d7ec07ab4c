"""Typed-row recorders of the BPTT-cWGAN driver (mirror of ``tc_gan/recorders.py``: same table names,
column names and dtypes, so the reference's loaders can read the tables)."""
import collections
import itertools

import numpy as np


def _host(x):
    return x.detach().cpu().numpy() if hasattr(x, 'detach') else np.asarray(x)


#: The typed tables of a run as DATA (tc_gan/recorders.py:113-362: same table names, column names and dtypes, so that the
#: reference's loaders read them).  key -> (table name, leading columns, patterns of the tail).  A tail pattern is formatted
#: once per item of the table's variable part, pattern by pattern: the moment conditions, the bandwidths, the generator's
#: flat parameter names, the critic's parameter tensors (`table_dtype`).
TABLES = {
    'learning': ('learning', [('gen_step', 'uint32'), ('Gloss', 'double'), ('Dloss', 'double'), ('Daccuracy', 'double'),
                              ('gen_forward_time', 'double'), ('gen_train_time', 'double'), ('disc_time', 'double'),
                              ('rate_penalty', 'double'), ('dynamics_penalty', 'double')], ()),
    'mm_learning': ('learning', [('step', 'uint32'), ('loss', 'double'), ('rate_penalty', 'double'),
                                 ('dynamics_penalty', 'double'), ('train_time', 'double')], ()),
    'gen_moments': ('gen_moments', [('step', 'uint32')], ('mean_{}', 'var_{}')),
    'disc_learning': ('disc_learning', [('gen_step', 'uint32'), ('disc_step', 'uint32'), ('Dloss', 'double'),
                                        ('Daccuracy', 'double'), ('SSsolve_time', 'double'), ('gradient_time', 'double'),
                                        ('model_convergence', 'uint32'), ('model_unused', 'uint32')], ()),
    'generator': ('generator', [('gen_step', 'uint32')], ('{}',)),
    'disc_param_stats': ('disc_param_stats', [('gen_step', 'uint32'), ('disc_step', 'uint32')], ('{}',)),
    'tc_stats': ('tc_stats', [('gen_step', 'uint32'), ('is_fake', 'b'), ('contrast', 'double'), ('norm_probe', 'double'),
                              ('cell_type', 'uint16'), ('count', 'uint32')], ('mean_{}', 'var_{}')),
}


def table_dtype(key, items=()):
    """The numpy dtype of table `key`: its leading columns, then one 'double' column per (tail pattern, item)."""
    _, lead, tail = TABLES[key]
    return np.dtype(list(lead) + [(pat.format(it), 'double') for pat in tail for it in items])


class HDF5Recorder(object):
    """recorders.py:62-110."""

    dedicated = False

    def __init__(self, datastore, quiet=True):
        self.datastore = datastore
        self.quiet = quiet

    @property
    def column_names(self):
        return self.dtype.names

    def _saverow(self, row):
        typed_row = np.array(tuple(row), dtype=self.dtype)
        self.datastore.h5.tables.saverow(self.tablename, typed_row, echo=not self.quiet)

    def write_header(self):
        self.datastore.h5.tables.create_table(self.tablename, self.dtype, dedicated=self.dedicated)

    def record(self, *row):
        self._saverow(row)

    @classmethod
    def make(cls, *args, **kwargs):
        self = cls(*args, **kwargs)
        self.write_header()
        return self

    @classmethod
    def from_driver(cls, driver):
        return cls.make(driver.datastore)


class LearningRecorder(HDF5Recorder):
    """recorders.py:113-147."""

    tablename = TABLES['learning'][0]
    dtype = table_dtype('learning')

    def record(self, gen_step, update_result):
        info, disc_info = update_result.info, update_result.disc_info
        self._saverow([gen_step, info.gen_loss, disc_info.disc_loss, disc_info.accuracy, info.gen_forward_time,
                       info.gen_train_time, info.disc_time, disc_info.rate_penalty, disc_info.dynamics_penalty])

    @classmethod
    def from_driver(cls, driver):
        return cls.make(driver.datastore, quiet=driver.quiet)


class MMLearningRecorder(HDF5Recorder):
    """recorders.py:150-172."""

    tablename = TABLES['mm_learning'][0]
    dtype = table_dtype('mm_learning')

    def record(self, gen_step, update_result):
        self._saverow([gen_step, update_result.loss, update_result.rate_penalty, update_result.dynamics_penalty,
                       update_result.train_time])

    @classmethod
    def from_driver(cls, driver):
        return cls.make(driver.datastore, quiet=driver.quiet)


class GenMomentsRecorder(HDF5Recorder):
    """recorders.py:175-199: minibatch mean and variance of every moment condition."""

    tablename = TABLES['gen_moments'][0]
    dedicated = True

    def __init__(self, datastore, num_mom_conds):
        super(GenMomentsRecorder, self).__init__(datastore)
        self.num_mom_conds = num_mom_conds
        self.dtype = table_dtype('gen_moments', range(num_mom_conds))

    def record(self, gen_step, update_result):
        self._saverow([gen_step] + list(np.asarray(update_result.gen_moments).flat))

    @classmethod
    def from_driver(cls, driver):
        return cls.make(driver.datastore, driver.mmatcher.num_mom_conds)


class DiscLearningRecorder(HDF5Recorder):
    """recorders.py:202-214."""

    tablename = TABLES['disc_learning'][0]
    dtype = table_dtype('disc_learning')


class FlexGenParamRecorder(HDF5Recorder):
    """recorders.py:243-272: one column per flat generator parameter."""

    tablename = TABLES['generator'][0]

    def __init__(self, datastore, gan):
        self.gan = gan
        super(FlexGenParamRecorder, self).__init__(datastore)
        self.dtype = table_dtype('generator', gan.gen.get_flat_param_names())

    def record(self, gen_step):
        self._saverow([gen_step] + list(self.gan.gen.get_flat_param_values()))
        return self.gan.get_gen_param()

    @classmethod
    def from_driver(cls, driver):
        return cls.make(driver.datastore, driver.gan)


class DiscParamStatsRecorder(HDF5Recorder):
    """recorders.py:275-311: normalised norm of every critic parameter tensor per critic step."""

    tablename = TABLES['disc_param_stats'][0]

    def __init__(self, datastore, discriminator):
        self.discriminator = discriminator
        super(DiscParamStatsRecorder, self).__init__(datastore)
        self.dtype = table_dtype('disc_param_stats', self.disc_param_unique_names(discriminator.get_param_names()))

    @staticmethod
    def disc_param_unique_names(names):
        counter = collections.Counter()
        for n in names:
            yield '{}.nnorm.{}'.format(n, counter[n])
            counter[n] += 1

    def record(self, gen_step, disc_step):
        if hasattr(self.discriminator, 'param_nnorms'):
            nnorms = self.discriminator.param_nnorms()       # fetched with the step's scalars: no host wait here
        else:
            nnorms = [np.linalg.norm(arr.flatten()) / arr.size for arr in self.discriminator.get_param_values()]
        self._saverow([gen_step, disc_step] + nnorms)
        return nnorms

    @classmethod
    def from_driver(cls, driver):
        return cls.make(driver.datastore, driver.gan.discriminator)


class ConditionalTuningCurveStatsRecorder(HDF5Recorder):
    """recorders.py:314-362: per condition mean/variance of real and generated tuning curves."""

    tablename = TABLES['tc_stats'][0]
    dedicated = True

    def __init__(self, datastore, num_bandwidths):
        super(ConditionalTuningCurveStatsRecorder, self).__init__(datastore)
        self.num_bandwidths = num_bandwidths
        self.dtype = table_dtype('tc_stats', range(num_bandwidths))

    @staticmethod
    def analyze(tuning_curves, conditions):
        tuning_curves, conditions = _host(tuning_curves), _host(conditions)

        def key(i):
            return tuple(conditions[i])
        for cond, group in itertools.groupby(sorted(range(len(conditions)), key=key), key=key):
            tc = tuning_curves[list(group)]
            yield list(cond) + [len(tc)] + list(tc.mean(axis=0)) + list(tc.var(axis=0))

    def record(self, gen_step, info):
        for is_fake, x, c in [(0, info.xd, info.cd), (1, info.xg, info.cg)]:
            for cond_stats in self.analyze(x, c):
                self._saverow([gen_step, is_fake] + cond_stats)

    @classmethod
    def from_driver(cls, driver):
        return cls.make(driver.datastore, len(driver.gan.bandwidths))
