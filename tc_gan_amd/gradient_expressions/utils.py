"""Which neurons are read out: probe positions -> site indices, and the gather that turns solver output
(draw, stimulus, neuron) into the rows the critic sees.  Same public names, arguments and results as
``tc_gan/gradient_expressions/utils.py`` (cited per function); host-side index arithmetic only."""
import numpy as np


def sample_slice(N, center_sites):
    """The `center_sites` sites around the middle of an N-site ring, as a slice (utils.py:4-20): for odd counts the
    extra site falls on the right of the centre, `sample_slice(10, 3)` -> sites 4, 5, 6."""
    first = N // 2 - center_sites // 2
    return slice(first, first + center_sites)


def sample_sites_from_stim_space_impl(stim_locs, N, type=int):
    """Bandwidth coordinate -> site index without validation (utils.py:23-24).  -1 is site 0 and +1 is site N-1; the
    position in between is cut toward zero by the integer cast (not rounded), so the centre probe of an even ring is
    site N/2 - 1.  The product is formed before the halving, as the reference does, so the cut falls on the same side."""
    last = N - 1
    return (last * (stim_locs + 1) / 2).astype(type)


def sample_sites_from_stim_space(stim_locs, N):
    """Validated form (utils.py:27-71): positions must lie in [-1, 1] and land on distinct sites; returns a list."""
    locs = np.asarray(stim_locs)
    assert all(locs >= -1)
    assert all(locs <= 1)
    sites = sample_sites_from_stim_space_impl(locs, N)
    if len(set(sites)) < len(sites):
        raise ValueError('Non-unique sample sites are specified.\n'
                         'N (= {}) is not large enough for stim_locs (= {}) to generate unique sample sites. '
                         'They generates sample_sites = {}'.format(N, list(locs), list(sites)))
    return list(sites)


def subsample_neurons(rate_vector, sample_sites, track_offset_identity=False, include_inhibitory_neurons=False,
                      N=None, NZ=None, NB=None):
    """Read `sample_sites` out of rates shaped (NZ draws, NB stimuli, 2N neurons) (utils.py:74-149).

    With `include_inhibitory_neurons` every site contributes its E neuron (index s) and its I neuron (index s + N), all
    E columns first.  `track_offset_identity=False`: one row per (draw, neuron), NB columns -- the neurons of a draw are
    treated as separate samples.  True: one row per draw, (stimulus, neuron) flattened with the neuron varying fastest."""
    columns = list(sample_sites)
    if isinstance(rate_vector, np.ndarray):
        draws, stimuli, neurons = rate_vector.shape
        NZ = draws if NZ is None else NZ
        NB = stimuli if NB is None else NB
        N = neurons // 2 if N is None else N
        assert (draws, stimuli, neurons) == (NZ, NB, 2 * N)
        assert min(columns) >= 0
        assert max(columns) < N
    if include_inhibitory_neurons:
        columns = columns + [s + N for s in columns]
    picked = rate_vector[:, :, columns]
    if track_offset_identity:
        return picked.reshape((NZ, -1))
    return picked.swapaxes(1, 2).reshape((-1, NB))
