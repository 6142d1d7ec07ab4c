#!/usr/bin/env python3
"""Seeded small learning runs through the reference CLI (bptt_cwgan / bptt_moments): start from perturbed (J, D, S), print
the (J, D, S)-distance of drivers.maybe_quit (Euclidean norm over the 12 entries) to the parameters the truth was generated
with, before and after.  Used to set the thresholds of tests/test_learning_gpu.py.
usage: tools/learn_probe.py [cwgan|moments] [--kernel NAME] [--steps K] [--seed S] ..."""
import argparse
import json
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(kind, kernel, steps, lr=0.01, num_sites=20, models=64, seqlen=200, skip=150, truth=512, scale=(1.3, 0.75),
        critic_iters=5, extra=()):
    from tc_gan_amd.networks.fixed_time_sampler import new_JDS
    from tc_gan_amd.run import bptt_cwgan, bptt_moments
    true = {k: np.asarray(new_JDS[k], dtype=float) for k in 'JDS'}
    pert = np.array([[scale[0], scale[1]], [scale[1], scale[0]]])
    start = {k: (true[k] * pert).tolist() for k in 'JDS'}
    with tempfile.TemporaryDirectory() as tmp:
        cfg = dict(num_sites=num_sites, dataset_provider='fixedtime', truth_size=truth, truth_seed=42,
                   true_ssn_options={k: true[k].tolist() for k in 'JDS'}, J0=start['J'], D0=start['D'], S0=start['S'])
        with open(os.path.join(tmp, 'cfg.json'), 'w') as fp:
            json.dump(cfg, fp)
        common = ['--iterations', str(steps), '--seqlen', str(seqlen), '--skip-steps', str(skip), '--n_bandwidths', '8',
                  '--datastore', os.path.join(tmp, 'out'), '--quiet', '--load-config', os.path.join(tmp, 'cfg.json'),
                  '--gen-kernel', kernel] + list(extra)
        if kind == 'cwgan':
            bptt_cwgan.main(common + ['--num-models', str(models), '--disc-layers', '[64, 64]', '--gen-learning-rate', str(lr),
                                      '--disc-learning-rate', str(lr), '--critic-iters-init', '20', '--critic-iters', str(critic_iters)])
        else:
            bptt_moments.main(common + ['--batchsize', str(models), '--learning-rate', str(lr)])
        gen = np.genfromtxt(os.path.join(tmp, 'out', 'generator.csv'), delimiter=',', names=True)
    names = [p + '_' + q for p in 'JDS' for q in ('EE', 'EI', 'IE', 'II')]
    truth_vec = np.concatenate([true[k].ravel() for k in 'JDS'])
    traj = np.stack([gen[n] for n in names], axis=1)
    dist = np.linalg.norm(traj - truth_vec[None], axis=1)
    d0 = float(np.linalg.norm(np.concatenate([np.asarray(start[k]).ravel() for k in 'JDS']) - truth_vec))
    return d0, dist


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('kind', choices=['cwgan', 'moments'])
    ap.add_argument('--kernel', default='auto')
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--lr', type=float, default=0.01)
    ap.add_argument('--models', type=int, default=64)
    ap.add_argument('--num-sites', type=int, default=20)
    a = ap.parse_args()
    d0, dist = run(a.kind, a.kernel, a.steps, lr=a.lr, models=a.models, num_sites=a.num_sites)
    pick = sorted(set([0, len(dist) // 4, len(dist) // 2, 3 * len(dist) // 4, len(dist) - 1]))
    print('%s kernel=%s steps=%d lr=%g: start %.4f; after step %s: %s' % (
        a.kind, a.kernel, a.steps, a.lr, d0, pick, ' '.join('%.4f' % dist[i] for i in pick)), flush=True)
