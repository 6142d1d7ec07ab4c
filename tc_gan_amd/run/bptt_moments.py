"""
Run BPTT-based moment matching on MI355X.

Mirror of ``tc_gan/run/bptt_moments.py``: same options, same config flow and outputs
(``learning`` / ``generator`` tables in the store, a dedicated ``gen_moments`` table).
"""
from logging import getLogger

import numpy as np

from .. import utils
from ..drivers import MomentMatchingDriver
from ..networks.moment_matching import make_moment_matcher
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
    """bptt_moments.py:36-101: the option table lives in `run/options.py` (rows marked 'm')."""
    from . import options
    return options.build_parser('m', __doc__)


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
