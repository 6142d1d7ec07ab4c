"""The command-line surface of the three BPTT run modules as DATA.

One table, `OPTIONS`: a row per option -- flags, default, value type, help, and the scripts that take it
(`w` = tc_gan.run.bptt_wgan, `c` = bptt_cwgan, `m` = bptt_moments) -- from which `build_parser` makes each script's
``argparse`` parser.  The names, aliases and defaults are the reference's (tc_gan/run/bptt_wgan.py:46-172,
bptt_cwgan.py:17-53, bptt_moments.py:36-101, run/gan.py:1109-1152, execution.py:290-317); `run_config` keys follow from the
first long flag as argparse derives them, and the CLI tests (tests/test_run_bptt_cwgan_gpu.py: option cases, the key set of
the paper's run.json through --load-config) are the contract.  Options this build adds say "(new)".
"""
import argparse

from .. import clib, utils
from ..networks import moment_matching, wgan

#: value types by name (argparse `type=`)
TYPES = {'int': int, 'float': float, 'eval': eval, 'csv_float': utils.csv_line(float), 'str': None}

_GEN_KERNEL_HELP = (
    'Kernel family of the generator forward / adjoint (new; recorded in info.json).  auto (default): '
    'the library\'s choice -- for float32 with >= 4 bandwidths and enough models the fp16-split '
    'matrix-core kernels (W and the state enter the products with 23 significant bits, exact '
    'products, fp32 accumulation: within the fp32 kernels\' own distance from fp64); mfma-fp32 or '
    'tile: fp32 operands (the reference\'s floatX arithmetic); the others name one kernel.  '
    'The fp16-split adjoint scales each step by the previous step\'s largest |delta|; a draw whose '
    'adjoint grows more than 2^8 within one step has no representable gradient there: such a step is '
    'recomputed on the fp32 kernels (logged) -- mfma-fp32 runs every step on them')


def _opt(scripts, *flags, **kw):
    return dict(scripts=scripts, flags=flags, **kw)


def _per_name(scripts, names, template, **kw):
    return [_opt(scripts, *[t.format(n) for t in template], **kw) for n in names]


OPTIONS = (
    # -- what each script samples ------------------------------------------------------------------------------------------
    [_opt('wm', '--batchsize', '--n_samples', default=15, type='eval',
          help='Number of samples to draw from G each step (aka NZ, minibatch size).'),
     _opt('w', '--sample-sites', default=[0], type='csv_float',
          help='Locations (offsets) of neurons to be sampled from SSN in the "bandwidth" space [-1, 1].  '
               '0 means the center of the network.'),
     _opt('m', '--sample-sites', default=[0], type='csv_float',
          help='Locations (offsets) of the sampled neurons in the "bandwidth" space [-1, 1].'),
     _opt('c', '--num-models', default=15, type='int', help='Number of SSN to be instantiated (aka NZ).'),
     _opt('c', '--probes-per-model', default=1, type='int'),
     _opt('c', '--norm-probes', '--sample-sites', default=[0], type='csv_float',
          help='Probe offsets in [-1, 1] "bandwidth coordinate".'),
     _opt('c', '--tc-stats-record-interval', default=100, type='int'),
     # -- data set ---------------------------------------------------------------------------------------------------------
     _opt('wcm', '--truth_size', default=1000, type='int', help='Number of SSNs used to generate ground truth data'),
     _opt('wcm', '--truth_seed', default=42, type='int', help='Seed for the ground truth data'),
     _opt('wcm', '--dataset-provider', default='ssnode', choices=('ssnode', 'fixedtime'),
          help='How the ground truth is generated (networks.dataset.generate_dataset)')]
    # -- updaters (GANs: one per player) -------------------------------------------------------------------------------------
    + [o for prefix in ('gen', 'disc') for o in (
        _opt('wc', '--{}-learning-rate'.format(prefix), '--{}-learn-rate'.format(prefix), default=0.01, type='float',
             help='{} learning rate'.format(prefix)),
        _opt('wc', '--{}-update-name'.format(prefix), default='adam-wgan', help='{} update method'.format(prefix)))]
    + [_opt('m', '--learning-rate', default=0.01, type='float'),
       _opt('m', '--update-name', default='adam-wgan'),
       # -- SSN ------------------------------------------------------------------------------------------------------------
       _opt('wc', '--seqlen', default=wgan.DEFAULT_PARAMS['seqlen'], type='int', help='Total time steps for SSN.'),
       _opt('m', '--seqlen', default=moment_matching.DEFAULT_PARAMS['seqlen'], type='int', help='Total time steps for SSN.'),
       _opt('wc', '--skip-steps', default=wgan.DEFAULT_PARAMS['skip_steps'], type='int',
            help='First time steps excluded from tuning curve and dynamics penalty.'),
       _opt('m', '--skip-steps', default=moment_matching.DEFAULT_PARAMS['skip_steps'], type='int'),
       _opt('wcm', '--contrasts', '--contrast', default=[20], type='csv_float'),
       _opt('wcm', '--include-inhibitory-neurons', action='store_true'),
       _opt('wcm', '--unroll-scan', action='store_true', help='Accepted for compatibility; no effect.')]
    + _per_name('wc', 'JDS', ['--gen-{}-min'], default=1e-3, type='float')
    + _per_name('wc', 'JDS', ['--gen-{}-max'], default=10, type='float')
    + _per_name('m', 'JDS', ['--{}-min'], default=1e-3, type='float')
    + _per_name('m', 'JDS', ['--{}-max'], default=10, type='float')
    + [_opt('wcm', '--{}0'.format(n), default=0.01, type='eval', help='Initial value of the generator parameter {}.'.format(n))
       for n in 'JDS']
    + [_opt('wc', '--gen-dynamics-cost', type='float', default=1),
       _opt('m', '--dynamics-cost', type='float', default=1),
       _opt('wcm', '--ssn-type', default='default', choices=('default', 'heteroin', 'deg-heteroin'),
            help='SSN variant (the reference sets it through --load-config)'),
       _opt('wcm', '--gen-kernel', default='auto', choices=tuple(clib.GEN_KERNELS), help=_GEN_KERNEL_HELP),
       _opt('wcm', '--z-device-seed', default=None, type='int',
            help='Draw z from a Philox4x32-10 stream of this seed (sharded over the ranks) instead of the '
                 'RandomState the reference draws it from (new; ANOTHER noise stream, no host round trip at all)'),
       _opt('wcm', '--z-host-draw', action='store_true',
            help='Draw z = rng.rand(batch, 2N, 2N) with numpy on the host, as the reference does (new).  Default: the '
                 'same RandomState stream continued on the device, bit for bit (ssn_mt19937_random_sample_*)'),
       # -- critic ---------------------------------------------------------------------------------------------------------
       _opt('wc', '--disc-layers', '--layers', default=[], type='eval'),
       _opt('wc', '--disc-normalization', default='none', choices=('none', 'layer')),
       _opt('wc', '--disc-nonlinearity', default='rectify',
            help='Hidden nonlinearity, a name of lasagne.nonlinearities: rectify, leaky_rectify, very_leaky_rectify, linear, '
                 'tanh, sigmoid, softplus, elu'),
       _opt('wc', '--disc-precision', default='fp32', choices=('bf16', 'fp32'),
            help='MFMA operand precision of the critic GEMMs (new).  fp32 (default) keeps the reference\'s '
                 'floatX arithmetic; bf16 is the explicit fast mode (fp32 accumulation, ~1e-2 relative on '
                 'the critic loss and gradients)'),
       _opt('wc', '--lipschitz-cost', '--WGAN_lambda', default=10.0, type='float'),
       _opt('wc', '--critic-iters-init', '--WGAN_n_critic0', default=50, type='int'),
       _opt('wc', '--critic-iters', '--WGAN_n_critic', default=5, type='int'),
       # -- moment matching ------------------------------------------------------------------------------------------------
       _opt('m', '--lam', default=.1, type='float', help='Weight for the variance'),
       _opt('m', '--moment-weights-regularization', default=1e-3, type='float'),
       _opt('m', '--moment-weight-type', default=moment_matching.DEFAULT_PARAMS['moment_weight_type'],
            choices=moment_matching.MOMENT_WEIGHT_TYPES),
       _opt('m', '--gen-moments-record-interval', default=100, type='int',
            help='Save tuning curve moments every given generator step. -1 means never.'),
       # -- the run --------------------------------------------------------------------------------------------------------
       _opt('wcm', '--iterations', default=100000, type='int'),
       _opt('wc', '--quit-JDS-threshold', default=-1, type='float'),
       _opt('wcm', '--quiet', action='store_true'),
       _opt('wc', '--disc-param-save-interval', default=5, type='int'),
       _opt('wc', '--disc-param-template', default='last.npz'),
       _opt('wc', '--disc-param-save-on-error', action='store_true'),
       _opt('wc', '--checkpoint-interval', default=-1, type='int',
            help='Write <datastore>/checkpoint.pkl (parameters, optimizer states, RNG states) every given '
                 'generator step; -1 never (new)'),
       _opt('wc', '--resume-from', default=None,
            help='checkpoint.pkl of an earlier run to continue from: --iterations stays the TOTAL count (new)'),
       _opt('wcm', '--n_bandwidths', default=4, type='int', choices=(1, 4, 5, 8)),
       _opt('wcm', '--load-gen-param', help='generator.csv whose last row is the starting point.'),
       # -- execution.py:290-317 -------------------------------------------------------------------------------------------
       _opt('wcm', '--datastore', help='Directory for output files (created if missing).'),
       _opt('wcm', '--datastore-template', default='logfiles/{IO_type}_{loss}_{layers_str}_{rate_cost}',
            help='Python format template for the datastore directory.'),
       _opt('wcm', '--debug', dest='datastore_template', action='store_const', const='logfiles/debug',
            help='A shorthand for --datastore-template=logfiles/debug.'),
       _opt('wcm', '--load-config', help='Load hyper parameters from a JSON/YAML/pickle file; they override the command line.')])

DATASTORE_TEMPLATES = {'w': 'logfiles/BPTT_WGAN_{layers_str}', 'c': 'logfiles/BPTT_CWGAN_{layers_str}', 'm': 'logfiles/BPTT_MM_{lam}'}


class _Formatter(argparse.RawDescriptionHelpFormatter, argparse.ArgumentDefaultsHelpFormatter):
    pass


def options_of(script):
    """The rows of `OPTIONS` the script takes ('w', 'c' or 'm'), in table order."""
    return [o for o in OPTIONS if script in o['scripts']]


def build_parser(script, description):
    parser = argparse.ArgumentParser(formatter_class=_Formatter, description=description)
    for o in options_of(script):
        kw = {k: v for k, v in o.items() if k not in ('scripts', 'flags', 'type')}
        if TYPES.get(o.get('type', 'str')) is not None:
            kw['type'] = TYPES[o['type']]
        parser.add_argument(*o['flags'], **kw)
    parser.set_defaults(datastore_template=DATASTORE_TEMPLATES[script])
    return parser
