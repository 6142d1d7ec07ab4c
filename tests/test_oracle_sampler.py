"""The minibatch sampler: the product's vectorised RandomChoiceSampler against the loop-form restatement of
cwgan.py:322-391 in oracle/sampler_numpy.py, stream for stream (`RandomState.get_state()` after every draw), so that the
RandomState consumption order of SURVEY.md section 8 (gotcha 1) is checked against an independent statement of it and
not against itself.  Runs without a GPU."""
import numpy as np
import pytest

from oracle import sampler_numpy as osn


def _state_equal(a, b):
    sa, sb = a.get_state(), b.get_state()
    return sa[0] == sb[0] and np.array_equal(sa[1], sb[1]) and sa[2:] == sb[2:]


def test_gridify_matches_the_reference_layout():
    """networks/utils.py:11-68: data varies (sample, contrast, bandwidth, cell type, probe); grid is
    (sample, cell type, probe, contrast, bandwidth) -- the doctest's shape contract plus every element."""
    from tc_gan_amd.networks.utils import gridify_tc_samples
    nb, nc, npb, nt = 7, 5, 3, 2
    shape = (11, nc, nb, nt, npb)
    data = np.arange(np.prod(shape)).reshape(shape)
    grid = osn.gridify(data, num_contrasts=nc, num_bandwidths=nb, num_cell_types=nt, num_probes=npb)
    assert grid.shape == (11, nt, npb, nc, nb)
    for idx in [(0, 0, 0, 0, 0), (3, 1, 2, 4, 6), (10, 0, 1, 2, 3)]:
        s, t, p, c, b = idx
        assert grid[idx] == data[s, c, b, t, p]
    got = gridify_tc_samples(data.reshape(11, -1), num_contrasts=nc, num_bandwidths=nb, num_cell_types=nt, num_probes=npb)
    np.testing.assert_array_equal(got, grid)


@pytest.mark.parametrize('inhib,norm_probes,contrasts,num_models,ppm', [
    (False, [0.0], [20.0], 6, 1),                 # one (cell type, probe) pair: the product skips the per-model loop
    (True, [0.0], [5.0, 20.0], 5, 1),             # two cells, weighted by e_ratio
    (True, [0.0, 0.5, -0.5], [5.0, 20.0], 4, 2),  # without replacement inside a model, p given
    (False, [0.0, 0.25, 0.5], [20.0], 7, 3),      # every probe of every model
])
def test_product_sampler_consumes_the_stream_like_the_restatement(inhib, norm_probes, contrasts, num_models, ppm):
    from tc_gan_amd.networks.cwgan import RandomChoiceSampler
    bandwidths = [0.0, 0.125, 0.5, 1.0]
    nt = 2 if inhib else 1
    data = np.random.RandomState(11).rand(9, len(contrasts) * len(bandwidths) * nt * len(norm_probes))
    r_prod, r_orc = np.random.RandomState(123), np.random.RandomState(123)
    sampler = RandomChoiceSampler.from_grid_data(data, bandwidths=bandwidths, contrasts=contrasts, norm_probes=norm_probes,
                                                 include_inhibitory_neurons=inhib, e_ratio=0.8, seed=r_prod)
    grid = osn.gridify(data, num_contrasts=len(contrasts), num_bandwidths=len(bandwidths), num_cell_types=nt,
                       num_probes=len(norm_probes))
    np.testing.assert_array_equal(sampler.nested, grid)
    for _ in range(5):
        got = sampler.select_minibatch(num_models, ppm)
        want = osn.select_minibatch(r_orc, grid, list(range(nt)), norm_probes, contrasts, bandwidths, 0.8, num_models, ppm)
        assert _state_equal(r_prod, r_orc)                       # same number and kind of draws, in the same order
        np.testing.assert_array_equal(got.tuning_curves, want['tuning_curves'])
        np.testing.assert_array_equal(got.conditions, want['conditions'])
        np.testing.assert_array_equal(got.model_ids, want['model_ids'])
        kw_got, kw_want = got.gen_kwargs, osn.gen_kwargs(want)
        assert set(kw_got) == set(kw_want)
        for k in kw_want:
            np.testing.assert_array_equal(np.asarray(kw_got[k]), kw_want[k], err_msg=k)
        # what follows the minibatch in a critic step (cwgan.py:476-481): eps, then zs -- still in step
        np.testing.assert_array_equal(r_prod.rand(got.batchsize, 1), r_orc.rand(num_models * ppm, 1))


def test_cells_are_unique_inside_a_model_and_follow_e_ratio():
    """random_cells (cwgan.py:328-355): at most once per model; E cells weigh e_ratio."""
    rng = np.random.RandomState(5)
    grid = np.zeros((3, 2, 4, 1, 2))
    picks = []
    for _ in range(400):
        mb = osn.select_minibatch(rng, grid, [0, 1], [0., .1, .2, .3], [20.], [0., 1.], 0.8, 1, 3)
        cells = {(c[2], c[1]) for c in mb['conditions']}
        assert len(cells) == 3
        picks.extend(c[2] for c in mb['conditions'])
    assert 0.6 < 1 - np.mean(picks) < 0.8          # E share: below e_ratio because draws are without replacement
