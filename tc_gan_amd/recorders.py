"""Typed-row recorders of the BPTT-cWGAN driver (mirror of ``tc_gan/recorders.py``: same table names,
column names and dtypes, so the reference's loaders can read the tables)."""
import collections
import itertools

import numpy as np


def _host(x):
    return x.detach().cpu().numpy() if hasattr(x, 'detach') else np.asarray(x)


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

    tablename = 'learning'
    dtype = np.dtype([('gen_step', 'uint32'), ('Gloss', 'double'), ('Dloss', 'double'), ('Daccuracy', 'double'),
                      ('gen_forward_time', 'double'), ('gen_train_time', 'double'), ('disc_time', 'double'),
                      ('rate_penalty', 'double'), ('dynamics_penalty', 'double')])

    def record(self, gen_step, update_result):
        info, disc_info = update_result.info, update_result.disc_info
        self._saverow([gen_step, info.gen_loss, disc_info.disc_loss, disc_info.accuracy, info.gen_forward_time,
                       info.gen_train_time, info.disc_time, disc_info.rate_penalty, disc_info.dynamics_penalty])

    @classmethod
    def from_driver(cls, driver):
        return cls.make(driver.datastore, quiet=driver.quiet)


class MMLearningRecorder(HDF5Recorder):
    """recorders.py:150-172."""

    tablename = 'learning'
    dtype = np.dtype([('step', 'uint32'), ('loss', 'double'), ('rate_penalty', 'double'),
                      ('dynamics_penalty', 'double'), ('train_time', 'double')])

    def record(self, gen_step, update_result):
        self._saverow([gen_step, update_result.loss, update_result.rate_penalty, update_result.dynamics_penalty,
                       update_result.train_time])

    @classmethod
    def from_driver(cls, driver):
        return cls.make(driver.datastore, quiet=driver.quiet)


class GenMomentsRecorder(HDF5Recorder):
    """recorders.py:175-199: minibatch mean and variance of every moment condition."""

    tablename = 'gen_moments'
    dedicated = True

    def __init__(self, datastore, num_mom_conds):
        super(GenMomentsRecorder, self).__init__(datastore)
        self.num_mom_conds = num_mom_conds
        self.dtype = np.dtype([('step', 'uint32')] +
                              [('mean_{}'.format(i), 'double') for i in range(num_mom_conds)] +
                              [('var_{}'.format(i), 'double') for i in range(num_mom_conds)])

    def record(self, gen_step, update_result):
        self._saverow([gen_step] + list(np.asarray(update_result.gen_moments).flat))

    @classmethod
    def from_driver(cls, driver):
        return cls.make(driver.datastore, driver.mmatcher.num_mom_conds)


class DiscLearningRecorder(HDF5Recorder):
    """recorders.py:202-214."""

    tablename = 'disc_learning'
    dtype = np.dtype([('gen_step', 'uint32'), ('disc_step', 'uint32'), ('Dloss', 'double'), ('Daccuracy', 'double'),
                      ('SSsolve_time', 'double'), ('gradient_time', 'double'), ('model_convergence', 'uint32'),
                      ('model_unused', 'uint32')])


class FlexGenParamRecorder(HDF5Recorder):
    """recorders.py:243-272: one column per flat generator parameter."""

    tablename = 'generator'

    def __init__(self, datastore, gan):
        self.gan = gan
        super(FlexGenParamRecorder, self).__init__(datastore)
        self.dtype = np.dtype([('gen_step', 'uint32')] + [(n, 'double') for n in gan.gen.get_flat_param_names()])

    def record(self, gen_step):
        self._saverow([gen_step] + list(self.gan.gen.get_flat_param_values()))
        return self.gan.get_gen_param()

    @classmethod
    def from_driver(cls, driver):
        return cls.make(driver.datastore, driver.gan)


class DiscParamStatsRecorder(HDF5Recorder):
    """recorders.py:275-311: normalised norm of every critic parameter tensor per critic step."""

    tablename = 'disc_param_stats'

    def __init__(self, datastore, discriminator):
        self.discriminator = discriminator
        super(DiscParamStatsRecorder, self).__init__(datastore)
        self.dtype = np.dtype([('gen_step', 'uint32'), ('disc_step', 'uint32')] +
                              [(name, 'double') for name in
                               self.disc_param_unique_names(discriminator.get_param_names())])

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

    tablename = 'tc_stats'
    dedicated = True

    def __init__(self, datastore, num_bandwidths):
        super(ConditionalTuningCurveStatsRecorder, self).__init__(datastore)
        self.num_bandwidths = num_bandwidths
        self.dtype = np.dtype([('gen_step', 'uint32'), ('is_fake', 'b'), ('contrast', 'double'),
                               ('norm_probe', 'double'), ('cell_type', 'uint16'), ('count', 'uint32')] +
                              [('mean_{}'.format(i), 'double') for i in range(num_bandwidths)] +
                              [('var_{}'.format(i), 'double') for i in range(num_bandwidths)])

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
