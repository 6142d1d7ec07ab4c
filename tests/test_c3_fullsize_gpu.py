"""BASELINE config 3 at FULL size (1024 weight draws x 8 bandwidths, 2N = 200, seqlen 1200, 3x512 critic) through the
GAN loop the bench times:

* two runs from the same seeds agree BIT FOR BIT after a whole GAN iteration (5 critic updates + 1 generator update):
  nothing on the path adds floating-point numbers in a run-dependent order (the critic's split-K slabs are reduced in
  slice order; round 1 used fp32 atomics there and identical-seed runs drifted apart by orders of magnitude within a few
  Adam steps at lr 0.01, which is where the erratic `last_gen_loss` of round 1's bench lines came from);
* the bf16-MFMA critic against the fp32-MFMA critic on ONE critic update + ONE generator update from the same state:
  losses and gradients within the bf16 tolerance (8-bit mantissa operands, fp32 accumulation).  Longer horizons are not
  compared: Adam normalises every gradient entry to a step of ~lr whatever its size, so an entry whose sign differs in
  the last bits moves its parameter by 2 lr, and the two trajectories separate chaotically (both stay finite).
"""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _run(precision, critic_iters, n_events):
    import bench
    gan, shape, _ = bench.make_c3_gan(disc_precision=precision, critic_iters_init=critic_iters, critic_iters=critic_iters)
    it = gan.learning()
    events = [next(it) for _ in range(n_events)]
    torch.cuda.synchronize()
    return gan, events


def test_identical_seeds_give_identical_runs_at_full_size():
    runs = []
    for _ in range(2):
        gan, ev = _run('bf16', 5, 6)
        assert [e.is_discriminator for e in ev] == [True] * 5 + [False]
        runs.append((np.array([e.disc_loss for e in ev[:5]]), ev[5].gen_loss, gan.disc.get_flat(),
                     np.concatenate([np.ravel(p) for p in gan.get_gen_param()])))
    a, b = runs
    assert np.all(np.isfinite(a[0])) and np.isfinite(a[1])
    np.testing.assert_array_equal(a[0], b[0])          # five critic losses
    assert a[1] == b[1]                                 # generator loss
    np.testing.assert_array_equal(a[2], b[2])          # every critic parameter after 5 Adam steps
    np.testing.assert_array_equal(a[3], b[3])          # J, D, S after the generator step


def test_bf16_critic_tracks_fp32_critic_for_one_update_at_full_size():
    g32, e32 = _run('fp32', 1, 2)
    g16, e16 = _run('bf16', 1, 2)
    d32, d16 = e32[0], e16[0]
    # the same minibatch, the same generated curves (the generator is fp32 in both runs)
    np.testing.assert_array_equal(d32.xd.cpu().numpy(), d16.xd.cpu().numpy())
    np.testing.assert_array_equal(d32.xg.cpu().numpy(), d16.xg.cpu().numpy())
    # critic loss on identical parameters: bf16 operands -> ~1e-2 relative on O(1) terms (tests/test_critic_gpu.py)
    assert abs(d16.disc_loss - d32.disc_loss) <= 3e-2 * max(1.0, abs(d32.disc_loss)), (d16.disc_loss, d32.disc_loss)
    assert abs(d16.accuracy - d32.accuracy) <= 3e-2 * max(1.0, abs(d32.accuracy))
    # first Adam step: every parameter moves by lr * sign(g) (lasagne adam at t = 1); the parameters of the two critics
    # agree except where a gradient entry is so small that bf16 rounding flips its sign
    p32, p16 = g32.disc.get_flat(), g16.disc.get_flat()
    moved = np.abs(p32 - p16) > 1e-3
    assert moved.mean() < 0.05, moved.mean()
    # generator step against (slightly different) critics: loss and the updated J, D, S
    assert abs(e16[1].gen_loss - e32[1].gen_loss) <= 5e-2 * max(1.0, abs(e32[1].gen_loss)), (e16[1].gen_loss, e32[1].gen_loss)
    j32 = np.concatenate([np.ravel(p) for p in g32.get_gen_param()])
    j16 = np.concatenate([np.ravel(p) for p in g16.get_gen_param()])
    assert np.all(np.isfinite(j16))
    # 12 parameters, each moved by +-lr = +-0.001 (first adam-wgan step; a sign disagreement shows as 0.002): allow one
    assert (np.abs(j32 - j16) > 1e-3).sum() <= 1, (j32, j16)
