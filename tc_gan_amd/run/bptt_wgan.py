"""
Run SSN-BPTT Wasserstein GAN learning on MI355X.

Mirror of ``tc_gan/run/bptt_wgan.py`` (and of the shared helpers of ``tc_gan/run/gan.py:1109-1254``): same option
names, defaults and config keys, same flow (argparse -> run_config -> preprocess -> info.json -> init_driver ->
learn), same outputs.  ``tc_gan_amd.run.bptt_cwgan`` reuses the options and `learn` from here.
"""
from logging import getLogger

import numpy as np

from .. import clib, execution, ssnode, utils
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
    """bptt_wgan.py:46-72."""
    import argparse

    class CustomFormatter(argparse.RawDescriptionHelpFormatter, argparse.ArgumentDefaultsHelpFormatter):
        pass

    parser = argparse.ArgumentParser(formatter_class=CustomFormatter, description=__doc__)
    parser.add_argument('--batchsize', '--n_samples', default=15, type=eval,
                        help='Number of samples to draw from G each step (aka NZ, minibatch size).')
    parser.add_argument('--sample-sites', default=[0], type=utils.csv_line(float),
                        help='Locations (offsets) of neurons to be sampled from SSN in the "bandwidth" space [-1, 1].  '
                             '0 means the center of the network.')
    add_bptt_common_options(parser)
    add_learning_options(parser)
    parser.set_defaults(datastore_template='logfiles/BPTT_WGAN_{layers_str}')
    return parser


def add_bptt_common_options(parser):
    """bptt_wgan.py:75-172."""
    parser.add_argument('--truth_size', default=1000, type=int,
                        help='Number of SSNs used to generate ground truth data (default: %(default)s)')
    parser.add_argument('--truth_seed', default=42, type=int, help='Seed for the ground truth data')
    parser.add_argument('--dataset-provider', default='ssnode', choices=('ssnode', 'fixedtime'),
                        help='How the ground truth is generated (networks.dataset.generate_dataset)')
    for prefix in ['gen', 'disc']:
        parser.add_argument('--{}-learning-rate'.format(prefix), '--{}-learn-rate'.format(prefix), default=0.01,
                            type=float, help='{} learning rate (default: %(default)s)'.format(prefix))
        parser.add_argument('--{}-update-name'.format(prefix), default='adam-wgan',
                            help='{} update method (default: %(default)s)'.format(prefix))
    parser.add_argument('--seqlen', default=DEFAULT_PARAMS['seqlen'], type=int, help='Total time steps for SSN.')
    parser.add_argument('--skip-steps', default=DEFAULT_PARAMS['skip_steps'], type=int,
                        help='First time steps excluded from tuning curve and dynamics penalty.')
    parser.add_argument('--contrasts', '--contrast', default=[20], type=utils.csv_line(float))
    parser.add_argument('--include-inhibitory-neurons', action='store_true')
    parser.add_argument('--unroll-scan', action='store_true', help='Accepted for compatibility; no effect.')
    for name in 'JDS':
        parser.add_argument('--gen-{}-min'.format(name), default=1e-3, type=float)
        parser.add_argument('--gen-{}-max'.format(name), default=10, type=float)
        parser.add_argument('--{}0'.format(name), default=0.01, type=eval,
                            help='Initial value of the generator parameter {}.'.format(name))
    parser.add_argument('--gen-dynamics-cost', type=float, default=1)
    parser.add_argument('--disc-layers', '--layers', default=[], type=eval)
    parser.add_argument('--disc-normalization', default='none', choices=('none', 'layer'))
    parser.add_argument('--disc-nonlinearity', default='rectify')
    parser.add_argument('--disc-precision', default='fp32', choices=('bf16', 'fp32'),
                        help='MFMA operand precision of the critic GEMMs (new).  fp32 (default) keeps the reference\'s '
                             'floatX arithmetic; bf16 is the explicit fast mode (fp32 accumulation, ~1e-2 relative on '
                             'the critic loss and gradients)')
    parser.add_argument('--lipschitz-cost', '--WGAN_lambda', default=10.0, type=float)
    parser.add_argument('--critic-iters-init', '--WGAN_n_critic0', default=50, type=int)
    parser.add_argument('--critic-iters', '--WGAN_n_critic', default=5, type=int)
    parser.add_argument('--ssn-type', default='default', choices=('default', 'heteroin', 'deg-heteroin'),
                        help='SSN variant (the reference sets it through --load-config)')
    parser.add_argument('--gen-kernel', default='auto', choices=tuple(clib.GEN_KERNELS),
                        help='Kernel family of the generator forward / adjoint (new; recorded in info.json).  auto (default): '
                             'the library\'s choice -- for float32 with >= 4 bandwidths and enough models the fp16-split '
                             'matrix-core kernels (W and the state enter the products with 23 significant bits, exact '
                             'products, fp32 accumulation: within the fp32 kernels\' own distance from fp64); mfma-fp32 or '
                             'tile: fp32 operands (the reference\'s floatX arithmetic); the others name one kernel.  '
                             'The fp16-split adjoint scales each step by the previous step\'s largest |delta|; a draw whose '
                             'adjoint grows more than 2^8 within one step makes its gradient NaN (never a clamped finite value) '
                             'and the run logs how many draws did -- mfma-fp32 has no such limit')
    parser.add_argument('--z-device-seed', default=None, type=int,
                        help='Draw z from a Philox4x32-10 stream of this seed (sharded over the ranks) instead of the '
                             'RandomState the reference draws it from (new; ANOTHER noise stream, no host round trip at all)')
    parser.add_argument('--z-host-draw', action='store_true',
                        help='Draw z = rng.rand(batch, 2N, 2N) with numpy on the host, as the reference does.  Default: the '
                             'same RandomState stream continued on the device, bit for bit (ssn_mt19937_random_sample_*)')


def add_learning_options(parser):
    """run/gan.py:1109-1152."""
    parser.add_argument('--iterations', default=100000, type=int)
    parser.add_argument('--quit-JDS-threshold', default=-1, type=float)
    parser.add_argument('--quiet', action='store_true')
    parser.add_argument('--disc-param-save-interval', default=5, type=int)
    parser.add_argument('--disc-param-template', default='last.npz')
    parser.add_argument('--disc-param-save-on-error', action='store_true')
    parser.add_argument('--checkpoint-interval', default=-1, type=int,
                        help='Write <datastore>/checkpoint.pkl (parameters, optimizer states, RNG states) every given '
                             'generator step; -1 never (new)')
    parser.add_argument('--resume-from', default=None,
                        help='checkpoint.pkl of an earlier run to continue from: --iterations stays the TOTAL count (new)')
    parser.add_argument('--n_bandwidths', default=4, type=int, choices=(1, 4, 5, 8))
    parser.add_argument('--load-gen-param', help='generator.csv whose last row is the starting point.')
    execution.add_base_learning_options(parser)


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
