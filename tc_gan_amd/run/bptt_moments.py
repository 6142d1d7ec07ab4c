"""
Run BPTT-based moment matching on MI355X.

Mirror of ``tc_gan/run/bptt_moments.py``: same options, same config flow and outputs
(``learning`` / ``generator`` tables in the store, a dedicated ``gen_moments`` table).
"""
from logging import getLogger

import numpy as np

from .. import clib, execution, utils
from ..drivers import MomentMatchingDriver
from ..networks.moment_matching import DEFAULT_PARAMS, MOMENT_WEIGHT_TYPES, make_moment_matcher
from .bptt_wgan import do_learning, generate_dataset_and_save

logger = getLogger(__name__)


def learn(driver, **generate_dataset_kwargs):
    """bptt_moments.py:19-33."""
    np.random.seed(0)
    mmatcher = driver.mmatcher
    mmatcher.prepare()
    data = generate_dataset_and_save(driver.datastore, mmatcher, **generate_dataset_kwargs)
    mmatcher.set_dataset(data)
    driver.run(mmatcher)


def make_parser():
    import argparse

    class CustomFormatter(argparse.RawDescriptionHelpFormatter, argparse.ArgumentDefaultsHelpFormatter):
        pass

    parser = argparse.ArgumentParser(formatter_class=CustomFormatter, description=__doc__)
    # Dataset
    parser.add_argument('--truth_size', default=1000, type=int, help='Number of SSNs used to generate ground truth data')
    parser.add_argument('--truth_seed', default=42, type=int, help='Seed for generating ground truth data')
    parser.add_argument('--dataset-provider', default='ssnode', choices=('ssnode', 'fixedtime'),
                        help='How the ground truth is generated (networks.dataset.generate_dataset)')
    # Driver
    parser.add_argument('--iterations', default=100000, type=int)
    parser.add_argument('--quiet', action='store_true')
    parser.add_argument('--gen-moments-record-interval', default=100, type=int,
                        help='Save tuning curve moments every given generator step. -1 means never.')
    # Generator
    parser.add_argument('--batchsize', '--n_samples', default=15, type=eval,
                        help='Number of samples to draw from G each step (aka NZ, minibatch size).')
    parser.add_argument('--seqlen', default=DEFAULT_PARAMS['seqlen'], type=int, help='Total time steps for SSN.')
    parser.add_argument('--skip-steps', default=DEFAULT_PARAMS['skip_steps'], type=int)
    parser.add_argument('--sample-sites', default=[0], type=utils.csv_line(float),
                        help='Locations (offsets) of the sampled neurons in the "bandwidth" space [-1, 1].')
    parser.add_argument('--contrasts', '--contrast', default=[20], type=utils.csv_line(float))
    parser.add_argument('--include-inhibitory-neurons', action='store_true')
    parser.add_argument('--unroll-scan', action='store_true', help='Accepted for compatibility; no effect.')
    for name in 'JDS':
        parser.add_argument('--{}-min'.format(name), default=1e-3, type=float)
        parser.add_argument('--{}-max'.format(name), default=10, type=float)
        parser.add_argument('--{}0'.format(name), default=0.01, type=eval)
    # Generator trainer
    parser.add_argument('--lam', default=.1, type=float, help='Weight for the variance')
    parser.add_argument('--moment-weights-regularization', default=1e-3, type=float)
    parser.add_argument('--moment-weight-type', default=DEFAULT_PARAMS['moment_weight_type'],
                        choices=MOMENT_WEIGHT_TYPES)
    parser.add_argument('--learning-rate', default=0.01, type=float)
    parser.add_argument('--update-name', default='adam-wgan')
    parser.add_argument('--dynamics-cost', type=float, default=1)
    parser.add_argument('--ssn-type', default='default', choices=('default', 'heteroin', 'deg-heteroin'),
                        help='SSN variant (the reference sets it through --load-config)')
    parser.add_argument('--gen-kernel', default='auto', choices=tuple(clib.GEN_KERNELS),
                        help='Kernel family of the generator forward / adjoint (new; see tc_gan.run.bptt_cwgan --help)')
    parser.add_argument('--z-device-seed', default=None, type=int,
                        help='Draw z from a Philox stream of this seed instead of the RandomState (new; another noise stream)')
    parser.add_argument('--z-host-draw', action='store_true',
                        help='Draw z with numpy on the host (default: the same RandomState stream continued on the device)')
    parser.add_argument('--n_bandwidths', default=4, type=int, choices=(1, 4, 5, 8))
    parser.add_argument('--load-gen-param', help='generator.csv whose last row is the starting point.')
    execution.add_base_learning_options(parser)
    parser.set_defaults(datastore_template='logfiles/BPTT_MM_{lam}')
    return parser


def init_driver(datastore, iterations, quiet, gen_moments_record_interval, quit_JDS_threshold=-1, **run_config):
    mmatcher, rest = make_moment_matcher(run_config)
    driver = MomentMatchingDriver(mmatcher, datastore, iterations=iterations, quiet=quiet,
                                  gen_moments_record_interval=gen_moments_record_interval,
                                  quit_JDS_threshold=quit_JDS_threshold)
    return dict(driver=driver, **rest)


def main(args=None):
    parser = make_parser()
    ns = parser.parse_args(args)
    do_learning(learn, vars(ns), init_driver=init_driver, script_file=__file__)


if __name__ == '__main__':
    main()
