"""Defaults and stimulus-grid helpers shared by the BPTT GANs (the non-Theano parts of ``tc_gan/networks/wgan.py``:
DEFAULT_PARAMS 39-63, grid_stimulator_inputs 293-296, probes_from_stim_space 447-451); `networks.wgan` re-exports them."""
import numpy as np

from .. import ssnode
from ..gradient_expressions.utils import sample_sites_from_stim_space
from ..utils import cartesian_product

# networks/wgan.py:39-63
DEFAULT_PARAMS = dict(
    bandwidths=ssnode.DEFAULT_PARAMS['bandwidths'],
    contrasts=ssnode.DEFAULT_PARAMS['contrast'],
    smoothness=ssnode.DEFAULT_PARAMS['smoothness'],
    sample_sites=[0],
    # Stimulator:
    num_sites=ssnode.DEFAULT_PARAMS['N'],
    # Model / SSN:
    k=ssnode.DEFAULT_PARAMS['k'],
    n=ssnode.DEFAULT_PARAMS['n'],
    io_type='asym_tanh',
    tau_E=10,
    tau_I=1,
    dt=0.1,
    seqlen=1200,
    batchsize=1,
    skip_steps=1000,
    gen=dict(
        rate_cost=0.01,
        rate_penalty_threshold=200.0,
    ),
    disc=dict(
        rate_penalty_bound=-1.0,
    ),
)


def grid_stimulator_inputs(contrasts, bandwidths, batchsize):
    """networks/wgan.py:293-296 -> (stimulator_contrasts, stimulator_bandwidths), each (batchsize, NC*NB)."""
    product = cartesian_product(contrasts, bandwidths)
    return np.tile(product.reshape((1,) + product.shape), (batchsize,) + (1,) * product.ndim).swapaxes(0, 1)


def probes_from_stim_space(stim_locs, num_sites, include_inhibitory_neurons):
    """networks/wgan.py:447-451."""
    probes = sample_sites_from_stim_space(stim_locs, num_sites)
    if include_inhibitory_neurons:
        probes.extend(np.array(probes) + num_sites)
    return probes
