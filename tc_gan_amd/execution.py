"""Run bookkeeping: datastore directory, typed tables, info.json / exit.json.

Host-side mirror of ``tc_gan/execution.py`` (no arithmetic).  Same on-disk names:
``info.json`` (run_config / extra_info / meta_info), ``exit.json``, CSV tables through
``datastore.tables.saverow``, typed tables ``learning``, ``disc_learning``, ``generator``,
``disc_param_stats`` in the shared store and ``tc_stats`` in a dedicated one.

The reference keeps typed tables in ``store.hdf5`` via h5py (execution.py:156-213).  When h5py is
importable the same files are written; otherwise (this image has no h5py) every table becomes a
``<table>.csv`` with a header row (same column names), appended row by row -- the format the reference's
own loader tries first (loaders/datastore_loader.py:55-59).  `tc_gan_amd.loaders.load_records` reads both.
"""
from getpass import getuser
from logging import getLogger
from socket import gethostname
import json
import os
import subprocess
import sys

import numpy

logger = getLogger(__name__)

PROJECT_ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))


class KnownError(Exception):
    """Exception with exit code (execution.py:20-26)."""

    def __init__(self, message, exit_code=1):
        self.exit_code = exit_code
        super(KnownError, self).__init__(message)


class SuccessExit(KnownError):
    def __init__(self, message):
        super(SuccessExit, self).__init__(message, exit_code=0)


def init_distributed(_environ=os.environ):
    """Join the torch.distributed job this process was launched in (``torchrun --nproc-per-node G run.py ...``): one
    process per GPU, RCCL (backend "nccl") unless TCGAN_DIST_BACKEND says otherwise (``gloo`` lets several ranks share
    one card, for tests).  No-op for a plain single-process run.  The models of a minibatch are then sharded over the
    ranks and every update does one all-reduce (networks/cwgan.py); every rank writes its own (identical) log, ranks
    above 0 under ``<datastore>/rank<r>``."""
    world = int(_environ.get('WORLD_SIZE', '1'))
    if world <= 1:
        return False
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        return True
    backend = _environ.get('TCGAN_DIST_BACKEND', 'nccl')
    local = int(_environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local % max(torch.cuda.device_count(), 1) if backend != 'nccl' else local)
    _environ.setdefault('MASTER_ADDR', '127.0.0.1')
    dist.init_process_group(backend, rank=int(_environ.get('RANK', '0')), world_size=world)
    return True


def distributed_rank():
    try:
        import torch.distributed as dist
        return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    except Exception:
        return 0


def makedirs_exist_ok(name):
    os.makedirs(name, exist_ok=True)


def _git(args):
    try:
        return subprocess.check_output(['git'] + args, cwd=PROJECT_ROOT, universal_newlines=True,
                                       stderr=subprocess.DEVNULL)
    except Exception:
        return ''


def relevant_environ(_environ=os.environ):
    keep = ('PATH', 'LD_LIBRARY_PATH', 'HOST', 'HOSTNAME', 'USER', 'USERNAME')
    prefixes = ('SLURM', 'PBS', 'OMP', 'MKL', 'GPU', 'HIP', 'ROCR', 'HSA', 'NCCL', 'RCCL')
    return {k: v for k, v in _environ.items() if k in keep or k.startswith(prefixes)}


def get_meta_info(packages=()):
    return dict(
        repository=dict(revision=_git(['rev-parse', 'HEAD']).rstrip(),
                        is_clean=_git(['status', '--short', '--untracked-files=no']).strip() == ''),
        python=sys.executable,
        packages={p.__name__: getattr(p, '__version__', '?') for p in packages},
        argv=sys.argv,
        environ=relevant_environ(),
        pid=os.getpid(),
        hostname=gethostname(),
        username=getuser(),
    )


class DataTables(object):
    """Plain-text row tables (execution.py:110-153)."""

    def __init__(self, directory):
        self.directory = directory
        self._files = {}

    def _open(self, name):
        return open(os.path.join(self.directory, name), 'w')

    def saverow(self, name, row, echo=False, flush=False):
        if isinstance(row, (list, tuple)):
            row = ','.join(map(str, row))
        if name not in self._files:
            self._files[name] = self._open(name)
        f = self._files[name]
        f.write(row)
        f.write('\n')
        if flush:
            f.flush()
        if echo:
            print(row)

    def flush_all(self):
        for f in self._files.values():
            f.flush()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        for name, f in self._files.items():
            try:
                f.close()
            except Exception as err:       # keep closing the others
                print('Error while closing', name, err)


class TypedTables(object):
    """Typed row tables: ``create_table(name, dtype, dedicated)``, ``saverow(name, typed_row)``
    (the interface of HDF5Tables, execution.py:156-191).

    With h5py: ``store.hdf5`` (+ one file per dedicated table), as the reference writes them.  Without h5py:
    one ``<table>.csv`` with a header row per table, appended row by row -- the format the reference's loader
    looks for FIRST (loaders/datastore_loader.py:55-59), so its `load_records` reads either."""

    shared_filename = 'store'

    def __init__(self, directory):
        self.directory = directory
        self._dtype = {}
        self._file_of = {}
        try:
            import h5py       # noqa: F401
            self.backend = 'hdf5'
        except ImportError:
            self.backend = 'csv'
        self._h5 = {}
        self._csv = {}

    def create_table(self, name, dtype, dedicated=False):
        assert name not in self._dtype
        self._dtype[name] = numpy.dtype(dtype)
        self._file_of[name] = name if dedicated else self.shared_filename
        if self.backend == 'hdf5':
            import h5py
            fname = self._file_of[name] + '.hdf5'
            if fname not in self._h5:
                self._h5[fname] = h5py.File(os.path.join(self.directory, fname), 'w')
            self._h5[fname].create_dataset(name, (0,), maxshape=(None,), dtype=self._dtype[name])
        else:
            f = open(os.path.join(self.directory, name + '.csv'), 'w')
            f.write(','.join(self._dtype[name].names) + '\n')
            self._csv[name] = f

    def saverow(self, name, row, echo=False, flush=False):
        if name not in self._dtype:
            self.create_table(name, row.dtype)
        if self.backend == 'hdf5':
            ds = self._h5[self._file_of[name] + '.hdf5'][name]
            ds.resize((len(ds) + 1,))
            ds[-1] = row
        else:
            self._csv[name].write(','.join(repr(v) for v in row.tolist()) + '\n')
        if flush:
            self.flush_all()
        if echo:
            print(*row.tolist(), sep=',')

    def flush_all(self):
        for f in list(self._h5.values()) + list(self._csv.values()):
            f.flush()

    def close(self):
        self.flush_all()
        for f in list(self._h5.values()) + list(self._csv.values()):
            f.close()


class _H5Facade(object):
    """``datastore.h5.tables`` as the recorders expect it (execution.py:194-213)."""

    def __init__(self, directory):
        self.tables = TypedTables(directory)

    def flush_all(self):
        self.tables.flush_all()


class DataStore(object):
    """execution.py:216-256."""

    def __init__(self, directory):
        self.directory = directory
        self.tables = DataTables(directory)
        self.h5 = _H5Facade(directory)

    def path(self, *subpaths):
        newpath = os.path.join(self.directory, *subpaths)
        makedirs_exist_ok(os.path.dirname(newpath))
        return newpath

    def dump_json(self, obj, filename):
        with open(self.path(filename), 'w') as fp:
            json.dump(obj, fp)

    def save_exit_reason(self, reason, good, **kwargs):
        logger.info('Recording reason=%s (%s) in exit.json', reason, 'good' if good else 'bad')
        self.dump_json(dict(reason=reason, good=good, **kwargs), 'exit.json')

    def flush_all(self):
        self.tables.flush_all()
        self.h5.flush_all()

    def __repr__(self):
        return '<DataStore: {}>'.format(self.directory)

    def __enter__(self):
        self.tables.__enter__()
        return self

    def __exit__(self, *exc):
        try:
            self.h5.tables.close()
        finally:
            self.tables.__exit__(*exc)


def format_datastore(datastore_template, run_config):
    """execution.py:272-287.

    >>> format_datastore('alpha={alpha}_L={layers_str}', dict(alpha=10, layers=[128, 64]))
    'alpha=10_L=128_64'
    """
    return datastore_template.format(layers_str='_'.join(map(str, run_config.get('layers', []))), **run_config)


def add_base_learning_options(parser):
    """execution.py:290-317 (same option names)."""
    parser.add_argument('--datastore', help='Directory for output files (created if missing).')
    parser.add_argument('--datastore-template', default='logfiles/{IO_type}_{loss}_{layers_str}_{rate_cost}',
                        help='Python format template for the datastore directory. (default: %(default)s)')
    parser.add_argument('--debug', dest='datastore_template', action='store_const', const='logfiles/debug',
                        help='A shorthand for --datastore-template=logfiles/debug.')
    parser.add_argument('--load-config', help='Load hyper parameters from a JSON/YAML/pickle file; they '
                                              'override the command line.')


def load_any_file(path):
    """utils/serializations.py: by extension."""
    ext = os.path.splitext(path)[1].lower()
    if ext == '.json':
        with open(path) as fp:
            return json.load(fp)
    if ext in ('.yaml', '.yml'):
        import yaml
        with open(path) as fp:
            return yaml.safe_load(fp)
    if ext in ('.pickle', '.pkl'):
        import pickle
        with open(path, 'rb') as fp:
            return pickle.load(fp)
    raise ValueError('Unsupported configuration file type: {}'.format(path))


def _jsonable(obj):
    if isinstance(obj, dict):
        return {k: _jsonable(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_jsonable(v) for v in obj]
    if isinstance(obj, numpy.ndarray):
        return obj.tolist()
    if isinstance(obj, numpy.generic):
        return obj.item()
    return obj


def pre_learn(packages, datastore, datastore_template, load_config, extra_info={}, preprocess=None, **run_config):
    """execution.py:320-346: merge --load-config, preprocess, create the directory, write info.json."""
    if load_config:
        run_config.update(load_any_file(load_config))
    if preprocess:
        preprocess(run_config)
    if not datastore:
        datastore = format_datastore(datastore_template, run_config)
    if distributed_rank() > 0:
        datastore = os.path.join(datastore, 'rank%d' % distributed_rank())
    makedirs_exist_ok(datastore)
    if run_config.get('resume_from'):
        # A resumed run starts its tables afresh (they are opened for writing, not appended to, and info.json is
        # rewritten): continuing INSIDE the directory that holds the earlier run would wipe its rows.  Refuse.
        old = sorted(f for f in os.listdir(datastore)
                     if f.endswith(('.csv', '.hdf5')) or f in ('info.json', 'exit.json'))
        if old:
            raise KnownError('--resume-from needs a NEW --datastore directory: {} already holds {} (a resumed run '
                             'rewrites its tables; the rows before the checkpoint stay in the earlier directory)'
                             .format(datastore, ', '.join(old[:4]) + (', ...' if len(old) > 4 else '')), exit_code=5)
    with open(os.path.join(datastore, 'info.json'), 'w') as fp:
        json.dump(_jsonable(dict(run_config=run_config, extra_info=extra_info,
                                 meta_info=get_meta_info(packages=packages))), fp)
    run_config['datastore'] = datastore
    return run_config


def do_learning(learn, run_config, extra_info={}, preprocess=None, packages=None):
    """execution.py:349-365: ``learn(datastore=DataStore(...), **run_config)`` after pre-processing."""
    import torch
    packages = [numpy, torch] if packages is None else packages
    logger.info('PID: %d', os.getpid())
    run_config = pre_learn(packages=packages, extra_info=extra_info, preprocess=preprocess, **run_config)
    with DataStore(run_config.pop('datastore')) as datastore:
        logger.info('Output directory: %s', datastore.directory)
        return learn(datastore=datastore, **run_config)
