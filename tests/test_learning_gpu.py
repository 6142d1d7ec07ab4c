"""End-to-end learning evidence ("next" #3).  BPTT gradient, WGAN-GP and Adam VALUES have no reference-held
vectors and are pinned piecewise by finite differences / autograd on the restatement; the only
independent check that the pieces COMPOSE is that a seeded run actually moves the generator toward the parameters the
data were generated with.  Small shape (2N = 40, 64 models, 8 bandwidths, 200 steps of dynamics), truth from known
(J, D, S) through the product's own fixed-time sampler, start perturbed by +30 % / -25 %, everything fp32, through the
reference CLI (`bptt_cwgan`, `bptt_moments`).  Distance = the (J, D, S) distance of drivers.maybe_quit
(drivers.py:183-198: Euclidean norm over the 12 entries).

Measured with tools/learn_probe.py on MI355X (lr 0.01): start 0.2387; after 38 generator steps 0.157-0.158 (x 0.66) for
both learners and every kernel family, then a plateau at 0.16-0.18 (one probe site and 8 bandwidths do not identify D and S
any better).  Asserted: x <= 0.80 at step 60 and at the best step, and all kernel families within 0.02 of each other."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
STEPS = 60


def _distances(kind, kernel):
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        'learn_probe', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'learn_probe.py'))
    probe = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(probe)
    return probe.run(kind, kernel, STEPS)


@pytest.mark.parametrize('kind,kernels', [('moments', ('tile', 'mfma-fp32', 'split-wide', 'duo')),
                                          ('cwgan', ('tile', 'mfma-fp32', 'split-wide', 'duo', 'duo-fused'))])
def test_seeded_run_moves_the_generator_toward_the_truth(kind, kernels):
    final = {}
    for kernel in kernels:
        d0, dist = _distances(kind, kernel)
        assert abs(d0 - 0.2387) < 1e-3                      # the perturbation itself
        assert np.isfinite(dist).all()
        assert dist[-1] <= 0.80 * d0, (kind, kernel, d0, dist[-1])
        assert dist.min() <= 0.75 * d0, (kind, kernel, d0, dist.min())
        assert dist[0] < d0                                # already the first update goes the right way
        final[kernel] = dist[-1]
    # VALU tile, fp32 matrix-core, both fp16-split families and the one-launch backward learn the same thing
    assert max(final.values()) - min(final.values()) < 0.02, final
