""""Truth" tuning-curve datasets (mirror of ``tc_gan/networks/dataset.py``)."""
from logging import getLogger

import numpy as np

from .. import ssnode
from .fixed_time_sampler import DEFAULT_PARAMS, FixedTimeTuningCurveSampler

logger = getLogger(__name__)

dataset_provider_choices = ('ssnode', 'fixedtime')


def dataset_by_ssnode(num_sites, bandwidths, contrasts, truth_size, truth_seed, sample_sites,
                      include_inhibitory_neurons, true_ssn_options={}):
    """dataset.py:28-71: fixed points of the batched GPU solver; defaults dt=5e-4, max_iter=1e5,
    io_type='asym_power', rate_stop_at=200 (draws whose rates pass 200 are rejected)."""
    data, (_, _, fpinfo) = ssnode.sample_tuning_curves(
        sample_sites=sample_sites, NZ=truth_size, seed=truth_seed, bandwidths=bandwidths, contrast=contrasts,
        N=num_sites, track_offset_identity=True, include_inhibitory_neurons=include_inhibitory_neurons,
        **dict(dict(dt=5e-4, max_iter=100000, io_type='asym_power', rate_stop_at=200), **true_ssn_options))
    data = np.array(data.T)
    logger.info('ssnode.sample_tuning_curves: rejections=%s codes=%r', fpinfo.rejections, fpinfo.counter)
    return data


def dataset_by_fixedtime(learner, truth_size, truth_seed, truth_batchsize=50, true_ssn_options={}):
    """dataset.py:74-122."""
    if truth_size < truth_batchsize:
        repeat, truth_batchsize = 1, truth_size
    else:
        repeat, mod = divmod(truth_size, truth_batchsize)
        if mod:
            repeat += 1
            logger.warning('truth_size=%d is not divisible by truth_batchsize=%d.', truth_size, truth_batchsize)
    options = dict(true_ssn_options)
    for name, _ in learner.gen.get_all_params():
        options.setdefault(name, DEFAULT_PARAMS[name])
    sampler = FixedTimeTuningCurveSampler.from_learner(learner, batchsize=truth_batchsize, seed=truth_seed, **options)
    data = np.concatenate([sampler.forward().prober_tuning_curve.cpu().numpy() for _ in range(repeat)])
    return data[:truth_size]


def generate_dataset(learner, dataset_provider='ssnode', **kwargs):
    """dataset.py:133-185."""
    if dataset_provider == 'ssnode' and learner.gen.heteroin:
        raise NotImplementedError("ssnode does not support SSN with heterogeneous input (yet).")   # dataset.py:163-166
    logger.info('Generating the truth...')
    if dataset_provider == 'ssnode':
        return dataset_by_ssnode(num_sites=learner.gen.num_sites, bandwidths=learner.bandwidths,
                                 contrasts=learner.contrasts, sample_sites=learner.sample_sites,
                                 include_inhibitory_neurons=learner.include_inhibitory_neurons, **kwargs)
    elif dataset_provider == 'fixedtime':
        return dataset_by_fixedtime(learner, **kwargs)
    raise ValueError('Unknown dataset_provider: {}'.format(dataset_provider))
