"""Probe / subsample index helpers -- mirror of
``tc_gan/gradient_expressions/utils.py`` (index arithmetic only; works on numpy
arrays and torch tensors alike)."""
import numpy as np


def sample_slice(N, center_sites):
    """utils.py:4-20."""
    i_beg = N // 2 - center_sites // 2
    i_end = i_beg + center_sites
    return np.s_[i_beg:i_end]


def sample_sites_from_stim_space_impl(stim_locs, N, type=int):
    """utils.py:23-24; the cast truncates toward zero."""
    return ((stim_locs + 1) * (N - 1) / 2).astype(type)


def sample_sites_from_stim_space(stim_locs, N):
    """utils.py:27-71."""
    stim_locs = np.asarray(stim_locs)
    assert all(stim_locs >= -1)
    assert all(stim_locs <= 1)

    sample_sites = sample_sites_from_stim_space_impl(stim_locs, N)

    if len(sample_sites) != len(set(sample_sites)):
        raise ValueError(
            'Non-unique sample sites are specified.\n'
            'N (= {}) is not large enough for stim_locs (= {}) to'
            ' generate unique sample sites.'
            ' They generates sample_sites = {}'
            .format(N, list(stim_locs), list(sample_sites)))

    return list(sample_sites)


def subsample_neurons(rate_vector, sample_sites,
                      track_offset_identity=False,
                      include_inhibitory_neurons=False,
                      N=None, NZ=None, NB=None):
    """utils.py:74-149: (NZ, NB, 2N) -> (NZ*len(sites), NB) or (NZ, NB*len(sites))."""
    if isinstance(rate_vector, np.ndarray):
        NZ_, NB_, TN_ = rate_vector.shape
        if NZ is None:
            NZ = NZ_
        if NB is None:
            NB = NB_
        if N is None:
            N = TN_ // 2
        assert (NZ_, NB_, TN_) == (NZ, NB, 2 * N)
        assert 0 <= min(sample_sites)
        assert max(sample_sites) < N

    if include_inhibitory_neurons:
        sample_sites = list(sample_sites)  # copy
        sample_sites.extend(np.array(sample_sites) + N)

    subsample = rate_vector[:, :, sample_sites]
    if track_offset_identity:
        return subsample.reshape((NZ, -1))
    else:
        return subsample.swapaxes(1, 2).reshape((-1, NB))
