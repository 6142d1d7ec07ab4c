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
    # the baseline names the box it ran on: the host's CPUs, this process's share of them, and the pool size
    # that ran the reference fastest among the candidates (no fixed cap)
    assert c['threads'] == c['cores'] <= c['affinity_cores'] == len(os.sched_getaffinity(0)) and c['host_cpu_count'] == os.cpu_count()
    assert 'cgroup_quota_cores' in c and isinstance(c['thread_probe_seconds'], dict)
    # the GAN half of the metric and the other named workloads ride along
    sec = d['secondary']
    assert sec['metric'] == 'GAN iters/sec' and sec['value'] > 0 and sec['gen_kernel'] == 'auto'
    assert sec['forward_variant'] == 8 and 'duo' in sec['roofline']['kernel']         # what the library picked, by its own word
    assert sec['fp32_mfma']['steps'] == sec['steps'] and sec['fp32_mfma']['warmup'] == sec['warmup']
    # the GAN loop is timed on the REFERENCE's noise stream (the RandomState continued on the device), the Philox loop beside it
    assert sec['z_mode'] == 'refstream' and 'RandomState' in sec['config']['workload'] and sec['host_draw_ms'] >= 0
    assert sec['philox_z']['ms_per_step'] > 0 and sec['philox_z']['steps'] == sec['steps']
    ex = d['extras']
    assert set(ex) == {'c2nb8', 'c5', 'c1_dropin', 'c3paper', 'c3_refstream', 'c3paper_rccl1'}
    assert ex['c3_refstream']['ms_per_step'] == sec['ms_per_step'] and ex['c3_refstream']['numpy_host_draw_ms'] > 0
    assert ex['c3paper']['z_mode'] == 'refstream'
    rc = ex.pop('c3paper_rccl1')                  # the paper-shape loop as the single rank of an RCCL group
    assert 'error' not in rc, rc
    assert rc['dist_backend'] == 'nccl' and rc['world_size'] == 1 and rc['ms_per_step'] > 0
    assert rc['phases']['collectives_per_iteration'] == 6 and rc['phases']['allreduce_ms']['max'] > 0    # 5 critic + 1 generator update
    assert rc['last_gen_loss'] == ex['c3paper']['last_gen_loss']            # mean over one rank = identity, bit for bit
    assert d['dist_backend'] is None and d['world_size'] == 1              # (the line itself is a plain single process)
    for name, e in ex.items():
        assert e['value'] > 0 and e['ms_per_step'] > 0 and 'workload' in e['config'], name
        assert name == 'c1_dropin' or {'bound', 'achieved', 'peak', 'frac'} <= set(e['roofline']), name
    assert 'solve_duo_kernel' in ex['c2nb8']['config']['kernel'] and ex['c1_dropin']['cpu_baseline']['value'] > 0


def test_no_extras_flag():
    d = _bench('--steps', '1', '--warmup', '0', '--no-extras', '--secondary-steps', '0', '--no-cpu-baseline')
    assert 'extras' not in d and 'secondary' not in d and 'cpu_baseline' not in d


def test_gan_loop_line():
    d = _bench('--workload', 'c3paper', '--steps', '3', '--warmup', '1')
    assert KEYS <= set(d)
    assert d['metric'] == 'GAN iters/sec' and d['value'] > 0 and d['roofline']['bound'] == 'mfma'
    assert abs(d['value'] - 1e3 / d['ms_per_step']) / d['value'] < 1e-6


def test_gpus_2_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher: the parent (which never touches the GPU) starts two ranks; here they
    share the one card of the test box over gloo.  One JSON line, n_gpus = world_size = 2, the C3 secondary aboard."""
    env = dict(os.environ, BENCH_DIST_BACKEND='gloo')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
                          '--secondary-steps', '1'], check=True, env=env, capture_output=True, text=True,
                         timeout=900).stdout.strip().splitlines()
    lines = [l for l in out if l.startswith('{')]
    assert len(lines) == 1, out
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['world_size'] == 2 and d['dist_backend'] == 'gloo'
    assert d['metric'] == 'SSN-steps/sec' and d['scaling'] == 'weak'
    # value = all ranks' units / max-over-ranks time
    assert abs(d['value'] - 2 * 200 * 4096 * 2000 / (d['ms_per_step'] * 1e-3)) / d['value'] < 1e-6
    sec = d['secondary']
    assert sec['metric'] == 'GAN iters/sec' and sec['n_gpus'] == 2 and sec['value'] > 0
    assert 'cpu_baseline' not in d                      # rank 0 at N = 1 only


def test_single_rank_group_runs_the_collectives_on_rccl():
    """The one way to put RCCL under this code on a one-GPU box: BENCH_FORCE_DIST=1 joins a one-rank `nccl` (= RCCL) group,
    and every collective of the N > 1 path -- the flat-buffer all-reduce per critic / generator update, the barriers and the
    max-over-ranks reductions of the timing -- runs through it.  The mean over one rank is the identity: the run must end on
    the loss of the plain single-process run, bit for bit."""
    env = dict(os.environ, BENCH_FORCE_DIST='1')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'BENCH_DIST_BACKEND'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--workload', 'c3paper', '--steps', '3', '--warmup', '1']
    out = subprocess.run(cmd, check=True, env=env, capture_output=True, text=True, timeout=600).stdout.strip().splitlines()
    assert len(out) == 1, out                 # RCCL's version banner goes to stderr, not into the stream the driver reads
    d = json.loads(out[0])
    assert d['dist_backend'] == 'nccl' and d['world_size'] == 1 and d['n_gpus'] == 1
    ph = d['phases']
    # ONE collective per update: five critic updates and one generator update per iteration
    assert ph['collectives_per_iteration'] == 6 and ph['allreduce_ms']['max'] > 0
    plain = _bench('--workload', 'c3paper', '--steps', '3', '--warmup', '1')
    assert plain['dist_backend'] is None and 'phases' not in plain
    assert d['last_gen_loss'] == plain['last_gen_loss']


def test_gpus_mismatch_is_refused():
    """A launcher that started a different number of ranks than --gpus names: no line, non-zero exit."""
    env = dict(os.environ, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and '{' not in r.stdout


def test_c1_dropin_line():
    """BASELINE config 1 through the zero-change drop-in symbols, timed like the reference calls them (thread pool, one
    ctypes call per solve, host buffers) with the reference's own C beside it."""
    d = _bench('--workload', 'c1', '--via', 'dropin', '--steps', '1', '--cpu-sample', '64')
    assert d['metric'] == 'SSN-steps/sec' and d['dtype'] == 'f64' and d['roofline'] is None
    assert 'drop-in' in d['config']['workload'] and d['config']['calls'] == 64
    assert d['value'] > 0 and d['config']['us_per_call_single_thread'] > 0
    c = d['cpu_baseline']
    assert c['kind'] in ('reference', 'port') and abs(c['gpu_over_cpu'] - d['value'] / c['value']) < 1e-9
