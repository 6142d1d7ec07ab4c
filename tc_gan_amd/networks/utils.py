"""networks/utils.py mirror: tuning-curve sample layouts (pure reshapes)."""

gridified_tc_axes = ('sample', 'cell_type', 'norm_probe', 'contrast', 'bandwidth')
sampled_tc_axes = ('sample', 'contrast', 'bandwidth', 'cell_type', 'norm_probe')


def gridify_tc_samples(data, num_contrasts, num_bandwidths, num_cell_types, num_probes):
    """(samples, contrasts*bandwidths*cell_types*probes) in `subsample_neurons` order ->
    (samples, cell_types, probes, contrasts, bandwidths)   (networks/utils.py:10-70)."""
    grid = data.reshape((len(data), num_contrasts, num_bandwidths, num_cell_types, num_probes))
    return grid.transpose((0, 3, 4, 1, 2))
