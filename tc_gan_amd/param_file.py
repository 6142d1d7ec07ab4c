"""Critic parameter files (mirror of ``tc_gan/lasagne_toppings/param_file.py``: same ``.npz`` keys --
``pval_<i>``, ``version``, ``param_names``, ``layer_classes``)."""
import numpy as np

version = 1


def dump(discriminator, path):
    values = discriminator.get_param_values()
    dval = {'pval_{}'.format(i): p for i, p in enumerate(values)}
    dval.update(version=version, param_names=list(discriminator.get_param_names()),
                layer_classes=['tc_gan_amd.critic.Critic'])
    np.savez_compressed(path, **dval)


def load(path):
    npz = np.load(path)
    keys = sorted((k for k in npz if k.startswith('pval_')), key=lambda k: int(k[len('pval_'):]))
    return [npz[k] for k in keys]
