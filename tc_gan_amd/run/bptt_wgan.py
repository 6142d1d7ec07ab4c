"""
Run SSN-BPTT Wasserstein GAN learning on MI355X.

Mirror of ``tc_gan/run/bptt_wgan.py`` (and of the shared helpers of ``tc_gan/run/gan.py:1109-1254``): same option
names, defaults and config keys, same flow (argparse -> run_config -> preprocess -> info.json -> init_driver ->
learn), same outputs.  ``tc_gan_amd.run.bptt_cwgan`` reuses the options and `learn` from here.
"""
from logging import getLogger

import numpy as np

from .. import execution, ssnode, utils
from ..networks.dataset import generate_dataset
from ..networks.fixed_time_sampler import new_JDS
from ..drivers import BPTTWGANDriver
from ..networks.wgan import DEFAULT_PARAMS, make_gan

logger = getLogger(__name__)


def generate_dataset_and_save(datastore, learner, **kwargs):
    data = generate_dataset(learner, **kwargs)
    np.save(datastore.path('truth.npy'), data)        # bptt_wgan.py:19-27
    return data


def learn(driver, **generate_dataset_kwargs):
    """bptt_wgan.py:29-43."""
    np.random.seed(0)
    gan = driver.gan
    gan.prepare()
    data = generate_dataset_and_save(driver.datastore, gan, **generate_dataset_kwargs)
    gan.set_dataset(data)
    driver.run(gan)


def make_parser():
    """bptt_wgan.py:46-172 + run/gan.py:1109-1152: the option table lives in `run/options.py` (rows marked 'w')."""
    from . import options
    return options.build_parser('w', __doc__)


_BANDWIDTHS = {1: [0.0625], 4: [0.0625, 0.125, 0.25, 0.75], 5: [0.0625, 0.125, 0.25, 0.5, 0.75],
               8: [0, 0.0625, 0.125, 0.1875, 0.25, 0.5, 0.75, 1]}


def preprocess(run_config):
    """run/gan.py:1155-1214 + bptt_wgan.py:201-214: bandwidths from n_bandwidths, J0/D0/S0 broadcast to 2x2,
    true_ssn_options defaulting to the "new" J, D, S."""
    run_config['bandwidths'] = _BANDWIDTHS[run_config.pop('n_bandwidths')]
    load_gen_param = run_config.pop('load_gen_param')
    if load_gen_param:
        with open(load_gen_param) as f:                      # typed-table CSVs carry a header row, legacy ones do not
            has_header = not f.readline().split(',', 1)[0].strip().lstrip('-').replace('.', '', 1).replace('e', '', 1).isdigit()
        lastrow = np.atleast_2d(np.loadtxt(load_gen_param, delimiter=',', skiprows=int(has_header)))[-1]
        lastrow = lastrow[1:] if len(lastrow) == 13 else lastrow
        J0, D0, S0 = lastrow.reshape((3, 2, 2))
        run_config.update(J0=J0, D0=D0, S0=S0)
    else:
        for key in 'JDS':
            run_config.setdefault(key + '0', ssnode.DEFAULT_PARAMS[key])
    for key in ('J0', 'D0', 'S0'):
        run_config[key] = np.broadcast_to(run_config[key], (2, 2)).tolist()
    true_ssn_options = run_config.setdefault('true_ssn_options', {})
    for key in ['J', 'D', 'S']:
        true_ssn_options.setdefault(key, new_JDS[key].tolist())
    if run_config.get('ssn_type') == 'heteroin':                  # bptt_wgan.py:211-214
        true_ssn_options.setdefault('V', [0.3, 0])
    elif run_config.get('ssn_type') == 'deg-heteroin':
        true_ssn_options.setdefault('V', 0.5)


def do_learning(learn, run_config, script_file, init_driver, preprocess=preprocess, **kwargs):
    """bptt_wgan.py:217-221 -> run/gan.py:1238-1254 -> execution.do_learning."""
    extra_info = dict(n_bandwidths=run_config['n_bandwidths'], load_gen_param=run_config['load_gen_param'],
                      data_version=1, script_file=script_file,
                      learn='{}.{}'.format(learn.__module__, learn.__name__),
                      init_driver='{}.{}'.format(init_driver.__module__, init_driver.__name__))
    execution.do_learning(lambda **rc: learn(**init_driver(**rc)), run_config, preprocess=preprocess,
                          extra_info=extra_info, **kwargs)


def init_driver(datastore, iterations, quit_JDS_threshold, quiet, disc_param_save_interval, disc_param_template,
                disc_param_save_on_error, layers, checkpoint_interval=-1, resume_from=None, **run_config):
    """bptt_wgan.py:175-198."""
    del layers                       # only used for the datastore name (execution.format_datastore)
    run_config = utils.subdict_by_prefix(run_config, 'disc_')
    run_config = utils.subdict_by_prefix(run_config, 'gen_')
    gan, rest = make_gan(run_config)
    driver = BPTTWGANDriver(
        gan, datastore, iterations=iterations, quiet=quiet, disc_param_save_interval=disc_param_save_interval,
        disc_param_template=disc_param_template, disc_param_save_on_error=disc_param_save_on_error,
        quit_JDS_threshold=quit_JDS_threshold, checkpoint_interval=checkpoint_interval, resume_from=resume_from)
    return dict(driver=driver, **rest)


def main(args=None):
    """bptt_wgan.py:224-233."""
    parser = make_parser()
    ns = parser.parse_args(args)
    ns.layers = ns.disc_layers       # for the {layers_str} of the datastore name; dropped again in init_driver
    do_learning(learn, vars(ns), init_driver=init_driver, script_file=__file__)


if __name__ == '__main__':
    main()
