#!/usr/bin/env python
"""Entry point mirroring the reference's ``./run <module> -- <args>`` (run.py:43-99):

    ./run tc_gan_amd.run.bptt_cwgan -- --iterations 10 --num-models 64 ...

``tc_gan.run.<name>`` module paths are accepted and mapped onto ``tc_gan_amd.run.<name>``.
Exit code = ``KnownError.exit_code`` for expected aborts (execution.KnownError)."""
import argparse
import importlib
import logging
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main(argv=None):
    parser = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    parser.add_argument('module')
    parser.add_argument('arguments', nargs='*')
    parser.add_argument('--log-level', default='INFO')
    parser.add_argument('--pidfile')
    ns = parser.parse_args(argv)
    logging.basicConfig(level=getattr(logging, ns.log_level), format='%(asctime)s %(levelname)s %(name)s: %(message)s')
    module = ns.module
    if os.path.isfile(module) and module.endswith('.py'):
        module = os.path.relpath(os.path.realpath(module), os.path.dirname(os.path.abspath(__file__)))[:-3].replace(os.sep, '.')
    if module.startswith('tc_gan.'):
        module = 'tc_gan_amd.' + module[len('tc_gan.'):]
    from tc_gan_amd.execution import KnownError, init_distributed
    init_distributed()          # torchrun / torch.distributed.run: one process per GPU, data parallel over weight draws
    loaded = importlib.import_module(module)
    if not hasattr(loaded, 'main'):
        print('Module', module, 'do not have main function.')
        return 1
    if ns.pidfile:
        with open(ns.pidfile, 'w') as f:
            f.write(str(os.getpid()))
    try:
        loaded.main(ns.arguments)
    except KnownError as err:
        print(err)
        return err.exit_code
    return 0


if __name__ == '__main__':
    sys.exit(main())
