"""The reference's quenched prober values asserted on the PRODUCT's probers (tc_gan_amd.networks.ssn), not only on the
oracle's restatement (tests/test_oracle_gan.py does that):
networks/tests/test_conditional_prober.py:18-86 (ConditionalProber) and the FixedProber doctest, networks/ssn.py:812-832."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GEN_KW = dict(smoothness=1 / 32, J=np.ones((2, 2)), D=np.ones((2, 2)), S=np.ones((2, 2)), k=0.01, n=2.2, tau_E=10, tau_I=1,
              dt=0.1, io_type='asym_tanh', seqlen=4, skip_steps=2)


def test_conditional_prober_quenched_values_on_the_product():
    from tc_gan_amd.networks.ssn import TuningCurveGenerator
    num_models, num_tcdom, num_sites = 2, 3, 201
    gen = TuningCurveGenerator(num_sites=num_sites, num_tcdom=num_tcdom, batchsize=10, probes=None, **GEN_KW)
    assert gen.conditional
    norm_probes = np.array([-1, -0.5, 0, 0.5, 1] * 2, dtype='float32')
    cell_types = np.array([0] * 5 + [1] * 5, dtype='uint16')
    model_ids = np.array([0, 1] * 5, dtype='uint16')
    shape = (num_models, num_tcdom, 2 * num_sites)
    time_avg = torch.arange(int(np.prod(shape)), dtype=torch.float32, device='cuda').reshape(shape)
    tc, ids, probes = gen._probe(time_avg, prober_norm_probes=norm_probes, prober_model_ids=model_ids,
                                 prober_cell_types=cell_types)
    # test_conditional_probes
    np.testing.assert_array_equal(probes.cpu().numpy(), [0, 50, 100, 150, 200, 201, 251, 301, 351, 401])
    np.testing.assert_array_equal(ids.cpu().numpy(), model_ids)
    # test_conditional_tuning_curve: the values the reference "quenched"
    desired = [[0, 402, 804], [1256, 1658, 2060], [100, 502, 904], [1356, 1758, 2160], [200, 602, 1004],
               [1407, 1809, 2211], [251, 653, 1055], [1507, 1909, 2311], [351, 753, 1155], [1607, 2009, 2411]]
    assert tuple(tc.shape) == (len(norm_probes), num_tcdom)
    np.testing.assert_array_equal(tc.cpu().numpy(), desired)
    # a second call with other probes must not serve the cached indices
    tc2, _, probes2 = gen._probe(time_avg, prober_norm_probes=norm_probes[::-1].copy(), prober_model_ids=model_ids,
                                 prober_cell_types=cell_types)
    np.testing.assert_array_equal(probes2.cpu().numpy(), [200, 150, 100, 50, 0, 401, 351, 301, 251, 201])


def test_conditional_probes_equal_the_sample_sites_of_the_gan():
    """test_conditional_probes_compare_with_sample_sites: for cell type 0 the conditional prober's indices are
    gan.sample_sites (sample_sites_from_stim_space, gradient_expressions/utils.py:23-24)."""
    from tc_gan_amd.gradient_expressions.utils import sample_sites_from_stim_space
    from tc_gan_amd.networks.ssn import TuningCurveGenerator
    num_sites = 50
    norm_probes = np.array([-0.75, -0.5, 0, 0.25, 0.5, 1.0])
    gen = TuningCurveGenerator(num_sites=num_sites, num_tcdom=2, batchsize=6, probes=None, **GEN_KW)
    ta = torch.zeros((1, 2, 2 * num_sites), device='cuda')
    _, _, probes = gen._probe(ta, prober_norm_probes=norm_probes, prober_model_ids=np.zeros(6, dtype='uint16'),
                              prober_cell_types=np.zeros(6, dtype='uint16'))
    assert list(probes.cpu().numpy()) == list(sample_sites_from_stim_space(norm_probes, num_sites))


def test_fixed_prober_doctest_values_on_the_product():
    from tc_gan_amd.networks.ssn import TuningCurveGenerator
    batchsize, num_tcdom, num_neurons = 3, 5, 7
    probes = np.array([0, 5])
    gen = TuningCurveGenerator(num_sites=4, num_tcdom=num_tcdom, batchsize=batchsize, probes=probes, **GEN_KW)
    assert not gen.conditional and gen.output_shape == (batchsize, num_tcdom * len(probes))
    time_avg = torch.arange(batchsize * num_tcdom * num_neurons, dtype=torch.float32, device='cuda').reshape(
        batchsize, num_tcdom, num_neurons)
    tc, _, _ = gen._probe(time_avg)
    np.testing.assert_array_equal(tc.cpu().numpy(),
                                  [[0, 5, 7, 12, 14, 19, 21, 26, 28, 33],
                                   [35, 40, 42, 47, 49, 54, 56, 61, 63, 68],
                                   [70, 75, 77, 82, 84, 89, 91, 96, 98, 103]])
    ta = time_avg.cpu().numpy()
    np.testing.assert_array_equal(ta[:, 0, probes], tc.cpu().numpy()[:, :len(probes)])
    np.testing.assert_array_equal(ta[:, 1, probes], tc.cpu().numpy()[:, len(probes):2 * len(probes)])
