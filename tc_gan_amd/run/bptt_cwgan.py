"""
Run SSN-BPTT conditional Wasserstein GAN learning on MI355X.

Mirror of ``tc_gan/run/bptt_cwgan.py``: same options, same config flow
(argparse -> run_config -> preprocess -> info.json -> init_driver -> learn), same outputs.
"""
from logging import getLogger

from .. import utils
from ..drivers import BPTTcWGANDriver
from ..networks.cwgan import make_gan
from .bptt_wgan import learn, do_learning

logger = getLogger(__name__)


def make_parser():
    """bptt_cwgan.py:17-53: the option table lives in `run/options.py` (rows marked 'c')."""
    from . import options
    return options.build_parser('c', __doc__)


def init_driver(datastore, iterations, quit_JDS_threshold, quiet, tc_stats_record_interval,
                disc_param_save_interval, disc_param_template, disc_param_save_on_error, layers,
                checkpoint_interval=-1, resume_from=None, **run_config):
    del layers                       # only used for the datastore name (execution.format_datastore)
    run_config = utils.subdict_by_prefix(run_config, 'disc_')
    run_config = utils.subdict_by_prefix(run_config, 'gen_')
    gan, rest = make_gan(run_config)
    driver = BPTTcWGANDriver(
        gan, datastore, iterations=iterations, quiet=quiet, tc_stats_record_interval=tc_stats_record_interval,
        disc_param_save_interval=disc_param_save_interval, disc_param_template=disc_param_template,
        disc_param_save_on_error=disc_param_save_on_error, quit_JDS_threshold=quit_JDS_threshold,
        checkpoint_interval=checkpoint_interval, resume_from=resume_from)
    return dict(driver=driver, **rest)


def main(args=None):
    parser = make_parser()
    ns = parser.parse_args(args)
    ns.layers = ns.disc_layers
    do_learning(learn, vars(ns), init_driver=init_driver, script_file=__file__)


if __name__ == '__main__':
    main()
