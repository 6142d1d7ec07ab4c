"""Data-parallel host logic on CPU with gloo, world_size 2: model sharding of a minibatch and the
single flat-buffer all-reduce that keeps the replicas in step."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from tc_gan_amd.networks.cwgan import GradientAllReducer, RandomChoiceSampler
    red = GradientAllReducer()
    assert red.on and red.world == world and red.rank == rank
    # every rank draws the SAME global minibatch from the same seed and keeps its block of models
    rs = np.random.RandomState(0)
    data = rs.rand(11, 2 * 3 * 2 * 2)              # contrasts(2) x bandwidths(3) x cell types(2) x probes(2)
    sampler = RandomChoiceSampler.from_grid_data(data, bandwidths=[.1, .2, .3], contrasts=[5., 20.],
                                                 norm_probes=[0., .5], include_inhibitory_neurons=True,
                                                 e_ratio=0.8, seed=7)
    batch = sampler.select_minibatch(2 * world, 2)
    local = batch.shard(rank, world)
    # "gradients": a deterministic function of the local shard; the mean over ranks must equal the
    # full-batch value computed on one process
    g_crit = torch.tensor(local.tuning_curves.sum(axis=0))
    g_gen = torch.tensor([local.conditions[:, 0].mean()])
    red.mean_(g_crit, g_gen)
    out.put((rank, local.tuning_curves, local.conditions, local.model_ids, g_crit.numpy(), g_gen.numpy(),
             batch.tuning_curves, batch.conditions))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 8])
def test_shard_and_flat_allreduce(world):
    """world 2, and world 8 = the node BASELINE config 4 names (8 x 1024 models; here 8 x 2)."""
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([out.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    full_tc, full_cond = res[0][6], res[0][7]
    for r in range(1, world):
        np.testing.assert_array_equal(res[r][6], full_tc)                   # same global draw on every rank
    np.testing.assert_array_equal(np.concatenate([res[r][1] for r in range(world)]), full_tc)      # shards tile the batch
    np.testing.assert_array_equal(np.concatenate([res[r][2] for r in range(world)]), full_cond)
    for r in range(world):
        np.testing.assert_array_equal(res[r][3], [0, 0, 1, 1])              # local model ids restart at 0 on every rank
    want_crit = full_tc.sum(axis=0) / world
    for r in range(world):
        np.testing.assert_allclose(res[r][4], want_crit, rtol=1e-6)          # mean over ranks, identical everywhere
        np.testing.assert_allclose(res[r][5], [full_cond[:, 0].mean()], rtol=1e-6)


def test_reducer_is_identity_without_process_group():
    from tc_gan_amd.networks.cwgan import GradientAllReducer
    red = GradientAllReducer()
    assert not red.on and red.world == 1
    t = torch.arange(4.)
    red.mean_(t)
    np.testing.assert_array_equal(t.numpy(), [0, 1, 2, 3])


@pytest.mark.parametrize('world', [2, 8])
def test_sharded_device_noise_offsets_tile_the_single_process_stream(world):
    """`DeviceNoise` bookkeeping without a GPU: rank r of `world` asks the Philox stream for rows [r n, (r + 1) n) of every
    global draw and all ranks advance by the global count, so the ranks' pieces (generated here by the CPU restatement of
    Philox4x32-10, oracle/philox_numpy.py) concatenate to exactly what a single process draws -- draw after draw, with
    draws of different sizes interleaved (z, then the heterogeneous-input noise), and the state is the same on all ranks."""
    from oracle import philox_numpy as ph
    from tc_gan_amd.networks.ssn import DeviceNoise
    seed, per_rank = 4321, [3 * 7 * 7, 3 * 7, 5 * 7 * 7, 5 * 7]          # local element counts of four consecutive draws
    single = DeviceNoise(seed)
    ranks = [DeviceNoise(seed, r, world) for r in range(world)]
    for n in per_rank:
        whole = ph.uniform(seed, single.take(world * n), world * n)
        pieces = [ph.uniform(seed, g.take(n), n) for g in ranks]
        np.testing.assert_array_equal(np.concatenate(pieces), whole)
        assert len({g.position for g in ranks} | {single.position}) == 1
    # one checkpoint restores every rank: the state carries no rank
    st = ranks[0].get_state()
    for g in ranks:
        assert g.get_state() == st
