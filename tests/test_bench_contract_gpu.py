"""The bench.py contract the driver depends on: one JSON line with the agreed keys, for the default workload (C2) and
for the GAN loop (C3, shortened through the paper shape to keep the test fast)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {'metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
        'vs_baseline', 'dtype', 'data', 'config', 'roofline'}


def _bench(*args):
    env = dict(os.environ)
    env.pop('WORLD_SIZE', None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + list(args), check=True, env=env,
                         capture_output=True, text=True, timeout=600).stdout.strip().splitlines()
    assert len(out) == 1, out                     # exactly ONE line on stdout
    return json.loads(out[0])


def test_default_workload_line():
    d = _bench('--steps', '2', '--warmup', '1', '--cpu-sample', '32')
    assert KEYS | {'cpu_baseline'} <= set(d)
    assert d['metric'] == 'SSN-steps/sec' and d['n_gpus'] == 1 and d['steps'] == 2 and d['warmup'] == 1
    assert d['higher_is_better'] is True and d['scaling'] == 'weak' and d['data'] == 'synthetic' and d['dtype'] == 'f32'
    assert 'workload' in d['config'] and d['config']['workload'].startswith('C2')
    r = d['roofline']
    assert {'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'} <= set(r)
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9 and 0.05 < r['frac'] < 1.0
    # value = neuron-steps of the batch / wall time per step
    assert abs(d['value'] - 200 * 4096 * 2000 / (d['ms_per_step'] * 1e-3)) / d['value'] < 1e-6
    c = d['cpu_baseline']
    assert {'value', 'unit', 'cores', 'kind', 'sample'} <= set(c) and c['kind'] in ('reference', 'port') and c['value'] > 0


def test_gan_loop_line():
    d = _bench('--workload', 'c3paper', '--steps', '3', '--warmup', '1')
    assert KEYS <= set(d)
    assert d['metric'] == 'GAN iters/sec' and d['value'] > 0 and d['roofline']['bound'] == 'mfma'
    assert abs(d['value'] - 1e3 / d['ms_per_step']) / d['value'] < 1e-6
