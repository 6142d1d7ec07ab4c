"""Two gloo ranks on one card against the single process at the same GLOBAL size (2048 models): per-step losses and generator
parameters of the C3 loop.  usage: tools/dp_probe.py [z_mode] [gen_steps]   (diagnostic; prints, asserts nothing)"""
import os
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(world, rank, z_mode, steps, models_per_rank):
    import bench
    gan, shape, bandwidths = bench.make_c3_gan(world, rank, disc_precision='bf16', models=models_per_rank, z_mode=z_mode)
    # the same truth whatever the split: 2 x 1024 curves from RandomState(42), as the two-rank job's make_c3_gan draws them
    rs = np.random.RandomState(42)
    truth = []
    for _ in range(2):
        bw = np.tile(np.asarray(bandwidths, dtype='float32')[None], (1024, 1))
        out = gan.gen.forward(rng=rs, stimulator_bandwidths=bw, stimulator_contrasts=np.full_like(bw, 20.0),
                              prober_norm_probes=np.zeros(1024), prober_model_ids=np.arange(1024), prober_cell_types=np.zeros(1024))
        truth.append(out.prober_tuning_curve.cpu().numpy())
    gan.set_dataset(np.concatenate(truth))
    it = gan.learning()
    rows = []
    while len(rows) < steps:
        info = next(it)
        if info.is_discriminator:
            last = (info.disc_loss, info.rate_penalty, info.dynamics_penalty)
            continue
        rows.append((info.gen_step, info.gen_loss, last, np.concatenate([np.ravel(p) for p in gan.get_gen_param()]), gan.gen.poisoned_draws()))
    return rows


def worker(rank, world, port, z_mode, steps, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    rows = run(world, rank, z_mode, steps, 1024)
    if rank == 0:
        out.put(rows)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    z_mode = sys.argv[1] if len(sys.argv) > 1 else 'refstream'
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    single = run(1, 0, z_mode, steps, 2048)
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, 2, 29871, z_mode, steps, out)) for r in range(2)]
    for p in procs:
        p.start()
    two = out.get(timeout=600)
    for p in procs:
        p.join(timeout=60)
    np.set_printoptions(precision=5, linewidth=200)
    for a, b in zip(single, two):
        print('step', a[0], 'gen_loss', a[1], b[1], 'last critic (loss, rate, dyn)', a[2], b[2], 'poisoned', a[4], b[4])
        print('   params 1 proc :', a[3])
        print('   params 2 ranks:', b[3])
