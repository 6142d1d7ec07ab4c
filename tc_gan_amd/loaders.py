"""Reading a run directory back (subset of ``tc_gan/loaders``: `load_records`, records_loader.py:569-578).

The tables are returned as pandas DataFrames under the reference's attribute names (`learning`, `generator`,
`disc_learning`, `disc_param_stats`, `tc_stats`, `gen_moments`), whichever backend wrote them (``store.hdf5`` /
``<table>.hdf5`` with h5py, ``<table>.csv`` without)."""
import json
import os

import numpy as np
import pandas


def _read_table(directory, name):
    csv = os.path.join(directory, name + '.csv')
    if os.path.exists(csv):
        return pandas.read_csv(csv)
    for fname in (name + '.hdf5', 'store.hdf5'):
        path = os.path.join(directory, fname)
        if os.path.exists(path):
            import h5py
            with h5py.File(path, 'r') as f:
                if name in f:
                    a = f[name][...]
                    return pandas.DataFrame({k: a[k] for k in a.dtype.names})
    raise RuntimeError('table {!r} not found in {}'.format(name, directory))


class Records(object):
    """`info.json` + `exit.json` + `truth.npy` + the typed tables of one run (lazy)."""

    table_names = ('learning', 'generator', 'disc_learning', 'disc_param_stats', 'tc_stats', 'gen_moments')

    def __init__(self, directory):
        self.directory = str(directory)
        with open(os.path.join(self.directory, 'info.json')) as f:
            self.info = json.load(f)
        self.run_config = self.rc = self.info['run_config']
        self._cache = {}

    def __getattr__(self, name):
        if name in type(self).table_names:
            if name not in self._cache:
                self._cache[name] = self.insert_epoch_column(_read_table(self.directory, name))
            return self._cache[name]
        raise AttributeError(name)

    def insert_epoch_column(self, df):
        """records_loader.py:236-247: epoch = gen_step * batchsize / truth_size where both are known."""
        step = 'gen_step' if 'gen_step' in df else ('step' if 'step' in df else None)
        size = self.rc.get('truth_size')
        batch = self.rc.get('batchsize') or self.rc.get('num_models')
        if step and size and batch and 'epoch' not in df:
            df['epoch'] = df[step] * batch / size
        return df

    @property
    def exit(self):
        path = os.path.join(self.directory, 'exit.json')
        return json.load(open(path)) if os.path.exists(path) else None

    @property
    def truth(self):
        return np.load(os.path.join(self.directory, 'truth.npy'))

    @property
    def param_element_names(self):
        return [c for c in self.generator.columns if c not in ('gen_step', 'epoch')]

    def gen_params_at(self, gen_step=-1):
        """records_loader.py:318-358: {'J': 2x2 array, 'D': ..., 'S': ...[, 'V': ...]} of one generator step."""
        row = self.generator.iloc[gen_step]
        params = {}
        for name in self.param_element_names:
            array_name, _, idx = name.partition('_')
            if not idx:
                params[array_name] = row[name]
            elif len(idx) == 1:
                params.setdefault(array_name, np.zeros(2))['EI'.index(idx)] = row[name]
            else:
                params.setdefault(array_name, np.zeros((2, 2)))['EI'.index(idx[0]), 'EI'.index(idx[1])] = row[name]
        return params

    def disc_param(self, name='last.npz'):
        return np.load(os.path.join(self.directory, 'disc_param', name))


def load_records(path):
    """Load the run directory `path` (or the directory of a file in it, e.g. ``.../info.json``)."""
    path = str(path)
    if os.path.isfile(path):
        path = os.path.dirname(path)
    return Records(path)
