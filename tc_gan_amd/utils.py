"""Small host-side helpers used by the BPTT-cWGAN path (mirror of the non-Theano members of
``tc_gan/utils``; index / bookkeeping only, no arithmetic on data)."""
import multiprocessing
import os
import time
import warnings

import numpy as np


class Namespace(object):
    """Attribute bag (utils/misc.py): ``Namespace(a=1).a == 1``."""

    def __init__(self, **kwargs):
        self.__dict__.update(kwargs)

    def __repr__(self):
        return 'Namespace({})'.format(', '.join('{}={!r}'.format(k, v) for k, v in sorted(vars(self).items())))


class StopWatch(object):
    """utils/misc.py:27-58: ``with watch: ...`` appends the elapsed seconds to ``watch.times``."""

    def __init__(self):
        self.times = []

    def __enter__(self):
        self._t0 = time.time()
        return self

    def __exit__(self, *exc):
        self.times.append(time.time() - self._t0)

    def sum(self):
        return float(np.sum(self.times)) if self.times else 0.0

    def mean(self):
        return float(np.mean(self.times)) if self.times else float('nan')


def cpu_count(_environ=os.environ):
    """utils/systems.py:5-37: OMP_NUM_THREADS > SLURM_CPUS_PER_TASK > (1 under Slurm) > PBS_NUM_PPN > all."""
    for key in ('OMP_NUM_THREADS', 'SLURM_CPUS_PER_TASK'):
        try:
            return int(_environ[key])
        except (KeyError, ValueError):
            pass
        if key == 'SLURM_CPUS_PER_TASK' and 'SLURM_JOB_ID' in _environ:
            return 1
    try:
        return int(_environ['PBS_NUM_PPN'])
    except (KeyError, ValueError):
        pass
    return multiprocessing.cpu_count()


def cartesian_product(*arrays, **kwargs):
    """utils/numerics.py:25-47.

    >>> cartesian_product([0, 1], [10, 20], dtype=int)
    array([[ 0,  0,  1,  1],
           [10, 20, 10, 20]])
    """
    dtype = kwargs.pop('dtype', 'float32')
    assert not kwargs
    arrays = [np.asarray(a) for a in arrays]
    grids = np.meshgrid(*arrays, indexing='ij')
    return np.stack([g.reshape(-1) for g in grids]).astype(dtype)


class DeviceContinuedRandomState(np.random.RandomState):
    """``numpy.random.RandomState`` (same stream, same methods) that may be told "your state after the draw the device is
    making for you will be ready later" (`tc_gan_amd.networks.ssn.device_rand`): the state is fetched -- and waited for -- the
    next time ANY public attribute of the generator is touched (`choice`, `rand`, `get_state`, pickling ...), not at the end of
    the draw, so the launches that follow a z draw are queued before the host ever waits."""

    def __getattribute__(self, name):
        if name not in ('_pending', '__dict__', '__class__'):
            d = object.__getattribute__(self, '__dict__')
            pend = d.get('_pending')
            if pend is not None:
                d['_pending'] = None
                pend(self)
        return object.__getattribute__(self, name)

    def _defer(self, finish):
        """`finish(rng)` sets the state (``RandomState.set_state(rng, ...)``) before the generator is next used."""
        object.__getattribute__(self, '__dict__')['_pending'] = finish


def as_randomstate(seed):
    """utils/numerics.py:50-54 (a seed becomes a RandomState that can wait lazily for device draws: same numbers)."""
    return seed if hasattr(seed, 'seed') else DeviceContinuedRandomState(seed)


def random_minibatches(batchsize, data, strict=False, seed=0):
    """utils/numerics.py:57-82: endless shuffled minibatches (one shuffle per epoch)."""
    n = len(data)
    if batchsize > n:
        raise ValueError('batchsize = {} > len(data) = {}'.format(batchsize, n))
    if n % batchsize != 0:
        msg = 'len(data) = {} not divisible by batchsize = {}'.format(n, batchsize)
        if strict:
            raise ValueError(msg)
        warnings.warn(msg)
    rng = as_randomstate(seed)

    def iterator():
        while True:
            idx = np.arange(n)
            rng.shuffle(idx)
            for i in range(n // batchsize):
                yield data[idx[i * batchsize:(i + 1) * batchsize]]
    return iterator()


def subdict_by_prefix(flat, prefix, key=None):
    """utils/dicts.py: move every ``prefix + name`` entry of `flat` into ``flat[key][name]``.

    >>> subdict_by_prefix(dict(a_x=1, a_y=2, b=3), 'a_') == {'a': {'x': 1, 'y': 2}, 'b': 3}
    True
    """
    key = prefix.rstrip('_') if key is None else key
    out, sub = {}, {}
    for k, v in flat.items():
        if k.startswith(prefix):
            sub[k[len(prefix):]] = v
        else:
            out[k] = v
    if key in out:
        raise ValueError('key {!r} already exists'.format(key))
    out[key] = sub
    return out


def csv_line(value_parser):
    """argparse type: comma separated values -> list (utils/misc.py)."""
    def convert(string):
        return [value_parser(v) for v in string.split(',')] if string else []
    return convert


def log_timing_message(name, seconds):
    return '{} done in {:.3g} sec'.format(name, seconds)


def to_device_packed(arrays, dtype):
    """Several small host arrays -> CUDA tensors of `dtype` through ONE pinned staging buffer and ONE asynchronous copy
    (the per-step inputs of a critic update: data minibatch, conditions, interpolation weights)."""
    import torch
    arrays = [np.ascontiguousarray(a) for a in arrays]
    starts, total = [], 0
    for a in arrays:                              # every piece starts on a 256-byte boundary, like a tensor of its own
        starts.append(total)
        total += -(-a.size // 64) * 64
    host = torch.empty(int(total), dtype=dtype).pin_memory()
    view = host.numpy()
    for a, off in zip(arrays, starts):
        view[off:off + a.size] = a.reshape(-1)
    dev = host.to('cuda', non_blocking=True)
    return [dev[off:off + a.size].reshape(a.shape) for a, off in zip(arrays, starts)]


def to_device(x, dtype=None):
    """Host array -> CUDA tensor through pinned staging memory and an ASYNCHRONOUS copy, so that the host does
    not wait for the kernels already queued on the stream (a pageable-memory copy does); CUDA tensors pass through."""
    import torch
    if torch.is_tensor(x) and x.is_cuda:
        return x if dtype is None else x.to(dtype)
    t = x if torch.is_tensor(x) else torch.as_tensor(np.ascontiguousarray(x))
    if dtype is not None:
        t = t.to(dtype)
    if t.numel() == 0:
        return t.to('cuda')
    return t.contiguous().pin_memory().to('cuda', non_blocking=True)
