#!/usr/bin/env python3
"""Benchmark of the SSN fixed-point hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload c2|c2nb8|c1]

Metric (BASELINE.json): SSN-steps/sec = neurons x batch x Euler-steps / s.
Workload at N=1 (BASELINE config 2, the one `metric` is quoted on): 2N=200 neurons,
batch 4096 independent weight draws, 2000 Euler steps, fp32, asym_tanh, atol=0 (so
exactly 2000 steps execute), synthetic z ~ U(0,1) resident in
HBM; W = make_W_with_x(z; new_JDS) is rebuilt on the device inside every step.
A "step" = one pass of the hot path over one batch: build W from z, then one batched
solve.  N>1: one process per GPU (torch.distributed / RCCL), every rank solves its own
batch of 4096 draws (the path partitions over draws, no data-path collective) ->
"scaling": "weak"; value = all ranks' units / max-over-ranks time.

Prints ONE JSON line on rank 0, with `roofline` (dominant kernel, HIP-event timed) and
`cpu_baseline` (the reference's own C file, oracle/_ref, on the host cores).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_VALU_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (vector)"
SPLIT_VARIANTS = (4, 5, 6, 7, 8)   # generator forward kernels on the fp16 matrix cores (`ssn_gen_forward_variant`)

WORKLOADS = {
    # name: (N, B, NB, T, description)
    'c2': (100, 4096, 1, 2000, 'C2: SSN fixed-point forward, 2N=200, batch 4096, NB=1 stimulus, 2000 Euler steps'),
    'c2nb8': (100, 4096, 8, 2000, 'C2/NB=8: 2N=200, batch 4096, 8 stimuli per draw, 2000 Euler steps'),
    'm152': (76, 4096, 1, 2000, 'probe: 2N=152, batch 4096, NB=1, 2000 Euler steps'),
    'c1': (50, 64, 1, 500, 'C1: 2N=100, batch 64, NB=1, 500 Euler steps'),
}


def new_jds():
    # tc_gan/networks/fixed_time_sampler.py:12-21
    J = np.array([[.0957, .0638], [.1197, .0479]])
    D = np.array([[.7660, .5106], [.9575, .3830]])
    S = np.array([[.6667, .2], [1.333, .2]]) / 8
    return J + D / 2 - D / 4, D / 2, S


def _bind_solve(fn):
    """ctypes signature of solve_dynamics_asym_*_euler (tc_gan/clib.py:16-26)."""
    dp = ctypes.POINTER(ctypes.c_double)
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_int, dp, dp, ctypes.c_double, ctypes.c_double, dp, dp, ctypes.c_double, ctypes.c_double,
                   ctypes.c_double, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double]
    return fn


def time_threaded_solves(fn, Ws, exts, N, T, threads, passes=2):
    """The reference's call pattern (ssnode.find_fixed_points_parallel, ssnode.py:423-510): a pool of Python
    threads, one task per weight draw, one blocking ctypes call per stimulus (largest bandwidth first) with the GIL
    released.  `fn` = a bound solve_dynamics_asym_tanh_euler.  Returns the best of 2 wall times after a warm-up."""
    from multiprocessing.dummy import Pool
    M = 2 * N
    dp = ctypes.POINTER(ctypes.c_double)

    def task(W):
        for ext in exts[::-1]:
            r0 = np.zeros(M)
            r1 = np.empty(M)
            fn(N, W.ctypes.data_as(dp), ext.ctypes.data_as(dp), 0.01, 2.2, r0.ctypes.data_as(dp), r1.ctypes.data_as(dp),
               0.01589, 0.002, 8e-4, T, 0.0, 200., 1000.)
        return r0[0]

    pool = Pool(threads)
    pool.map(task, Ws[:threads])                      # warm-up
    best = float('inf')
    for _ in range(passes):
        t0 = time.perf_counter()
        pool.map(task, Ws, chunksize=1)
        best = min(best, time.perf_counter() - t0)
    pool.close()
    pool.join()
    return best


def usable_cores():
    """(cores this process may run on, the cgroup's CPU quota in cores or None): os.sched_getaffinity names every logical CPU
    of the host on the GPU boxes (256) while the container's CPU share is a fraction of them (cpu.max)."""
    quota = None
    try:
        q, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if q != 'max':
            quota = max(1, int(round(int(q) / float(period))))
    except Exception:
        pass
    return len(os.sched_getaffinity(0)), quota


def pick_threads(fn, Ws, exts, N, T):
    """Thread count of the CPU baseline: the reference sizes its pool by cpu_count() (ssnode.py:436, utils/systems.py:5-37),
    which on a box whose container holds a share of the host's CPUs oversubscribes it (256 threads on a 16-core share ran at 0.7 x
    the 16-thread rate).  The baseline gets the BEST of: the cgroup quota, every core of the affinity mask, and 16 / 64 -- timed on
    a short probe of the same solves -- so that the GPU / CPU ratio is against the fastest this host gives the reference."""
    affinity, quota = usable_cores()
    cands = sorted({c for c in (16, 64, affinity, quota) if c and c <= affinity})
    if len(cands) == 1:
        return cands[0], {}
    probe = {}
    n = min(len(Ws), max(cands[-1], 128))
    for c in cands:
        probe[c] = time_threaded_solves(fn, Ws[:n], exts, N, T, c, passes=1)
    return min(probe, key=probe.get), {str(c): float(M2) for c, M2 in probe.items()}


def cpu_baseline(N, NB, T, sample_B, threads=None):
    """Time the reference's C solver (oracle/_ref/libssnode.so, built from tc_gan/ext/ssnode.c unmodified) driven as
    ssnode.find_fixed_points_parallel drives it.  Falls back to the oracle's C restatement ("port") when the
    prebuilt reference library is absent.  This is the only leg of bench.py that touches oracle/."""
    from oracle import ssn_numpy as on
    ref = on.load_reference_lib()
    kind = 'reference' if ref is not None else 'port'
    J, D, S = new_jds()
    M = 2 * N
    rs = np.random.RandomState(0)
    Ws = [on.generate_weight(N, J, D, S, rs.rand(M, M)) for _ in range(sample_B)]
    bws = [1.0] if NB == 1 else on.DEFAULT_PARAMS['bandwidths'][:NB]
    exts = on.stimulus_input(bws, np.linspace(-.5, .5, N), on.DEFAULT_PARAMS['smoothness'], [20.])
    if ref is not None:
        fn = _bind_solve(ref.solve_dynamics_asym_tanh_euler)
    else:
        lib = on.load_oracle_lib()

        def fn(N_, W, ext, k, n, r0, r1, tE, tI, dt, T_, atol, soft, hard):
            return lib.oracle_solve_euler(2, N_, W, ext, k, n, r0, r1, tE, tI, dt, T_, atol, soft, hard, None)
    probe = {}
    if threads is None:
        threads, probe = pick_threads(fn, Ws, exts, N, T)
    best = time_threaded_solves(fn, Ws, exts, N, T, threads)
    units = float(M) * sample_B * NB * T
    n1 = max(2, min(sample_B, 16))                      # the same call pattern on ONE thread, a few draws
    best1 = time_threaded_solves(fn, Ws[:n1], exts, N, T, 1)
    # the box the baseline ran on: `cores` = threads = the fastest of the candidate pool sizes (pick_threads), beside what the
    # host has (host_cpu_count), what this process may run on (affinity_cores) and the container's share (cgroup_quota_cores)
    affinity, quota = usable_cores()
    return dict(value=units / best, unit='neuron*batch*Euler-steps/s', cores=threads, threads=threads,
                affinity_cores=affinity, cgroup_quota_cores=quota, host_cpu_count=os.cpu_count(), thread_probe_seconds=probe, kind=kind,
                value_1thread=float(M) * n1 * NB * T / best1,
                sample='%d of the workload\'s weight draws x %d stimuli x %d steps, 2N=%d, fp64, '
                       '%d Python threads over ctypes (best of 2 after warm-up, %.2f s)' %
                       (sample_B, NB, T, M, threads, best))


def _traffic(key):
    """PMC-derived HBM bytes per launch of the workload's dominant kernel (profiles/hbm_traffic.json), or None."""
    try:
        return json.load(open(os.path.join(ROOT, 'profiles', 'hbm_traffic.json'))).get(key)
    except Exception:
        return None


def make_c3_gan(world=1, rank=0, paper=False, disc_precision='bf16', critic_iters_init=5, critic_iters=5, models=None,
                gen_kernel='auto', z_mode='philox', gen_learning_rate=None):
    """The GAN of BASELINE config 3/4 (or, `paper=True`, of scripts/fig4/gan/run.json) with its truth data set:
    returns (gan, (N, models_per_rank, NB, T, skip), bandwidths).  Shared by the bench and by the full-size parity test."""
    from tc_gan_amd.networks.cwgan import make_gan
    J, D, S = new_jds()
    N, default_models, NB, T, skip = (101, 128, 8, 240, 200) if paper else (100, 1024, 8, 1200, 1000)
    models = models or default_models
    bandwidths = [0, 0.0625, 0.125, 0.1875, 0.25, 0.5, 0.75, 1]
    cfg = dict(num_sites=N, num_models=models * world, probes_per_model=1, norm_probes=[0.0],
               include_inhibitory_neurons=False, bandwidths=bandwidths, contrasts=[20.0],
               seqlen=T, skip_steps=skip, J0=J, D0=D, S0=S, critic_iters_init=critic_iters_init, critic_iters=critic_iters,
               lipschitz_cost=10.0, gen_kernel=gen_kernel,
               # generator step size: the reference's default (networks/wgan.py:119).  Rounds 1-4 timed this loop at 0.01: with
               # S0 = 0.025 .. 0.17 three Adam steps of +-0.01 take S_EI / S_II to the lower bound, and at 2048 models and more
               # (the data-parallel sizes) this seed's dynamics explode at the second generator step -- rate penalty 700,
               # every draw refused by the fp16-split adjoint and redone in fp32.  Same kernels, same shapes,
               # same time per iteration at 1024 models; a run that trains instead of one that has diverged.
               gen=dict(learning_rate=float(os.environ.get('BENCH_GEN_LR', '0.001')) if gen_learning_rate is None else gen_learning_rate,
                        update_name='adam-wgan', dynamics_cost=1.0, rate_cost=0.01,
                        rate_penalty_threshold=200.0, J_min=1e-3, J_max=10, D_min=1e-3, D_max=10, S_min=1e-3, S_max=10),
               disc=dict(learning_rate=0.01, update_name='adam-wgan', layers=[512, 512, 512], normalization='none',
                         nonlinearity='rectify', precision=disc_precision))
    # z_mode: 'philox' = one Philox stream sharded over the ranks (--z-device-seed: NOT the reference's noise), 'refstream' = the
    # reference's own stream, rng.rand of the GAN's RandomState(seed 0), continued on the device bit for bit (the package
    # default), 'numpy' = the same stream drawn by numpy on the host (--z-host-draw: what the reference itself does)
    cfg.update({'philox': dict(z_device_seed=4321), 'refstream': {}, 'numpy': dict(z_host_draw=True)}[z_mode])
    if paper:
        cfg.update(tau_E=2, ssn_type='deg-heteroin', V=0.1)
        cfg['gen'].update(learning_rate=1e-4, update_name='rmsprop', dynamics_cost=0.0, rate_cost=100.0)
        cfg['disc'].update(learning_rate=0.02, update_name='rmsprop', layers=[128] * 4,
                           normalization=['none', 'layer', 'layer', 'layer'], reg_l2_decay=0.001, rate_penalty_bound=1.0)
    gan, _ = make_gan(cfg)
    # truth: 2048 curves from the generator itself at the true parameters (dataset_by_fixedtime), seed 42
    rs = np.random.RandomState(42)
    truth = []
    for _ in range(2):
        bw = np.tile(np.asarray(bandwidths, dtype='float32')[None], (models, 1))
        out = gan.gen.forward(rng=rs, stimulator_bandwidths=bw, stimulator_contrasts=np.full_like(bw, 20.0),
                              prober_norm_probes=np.zeros(models), prober_model_ids=np.arange(models),
                              prober_cell_types=np.zeros(models))
        truth.append(out.prober_tuning_curve.cpu().numpy())
    gan.set_dataset(np.concatenate(truth))
    return gan, (N, models, NB, T, skip), bandwidths


def _forward_roofline(variant, achieved, traffic, kernel_ms, M, steps_in_loop):
    """Roofline object of the generator forward.  `achieved` is ALGORITHMIC: (2 M + 8) flop per neuron-step.
    fp32 MFMA kernel (v_mfma_f32_4x4x1): peak = fp32 matrix peak = fp32 vector peak = 157.3 TFLOP/s.
    fp16-split kernels (v_mfma_f32_16x16x32_f16): priced against the fp16 dense peak, 16x that; they EXECUTE 4 fp16 flops
    per algorithmic flop in the wide form (W as 2 fp16 parts x 2 state parts, all 16 operand columns used by 8 stimuli),
    8 in the alternating form (2 x 4 columns per stimulus: 3 state parts + 1 unused), on tiles padded from M x M to
    16 ceil(M / 16) x 32 ceil(M / 32)."""
    groups = {2: 'two 4-stimulus groups per workgroup', 3: 'one 4-stimulus group per workgroup'}
    if variant in SPLIT_VARIANTS:
        peak = 16 * PEAK_FP32_VALU_TFLOPS
        pad = (16 * -(-M // 16)) * (32 * -(-M // 32)) / float(M * M)
        # executed fp16 flops per algorithmic flop: W parts x operand columns per stimulus
        form, per_flop, state = {4: ('all 8 stimuli in one chain per step', 2 * 2, '2 parts (22 bits)'),
                                 7: ('all 8 stimuli in one chain per step', 3 * 2, '3 parts (exact)'),
                                 5: (groups[3], 2 * 4, '3 parts (exact)'),
                                 6: (groups[2] + ', alternating', 2 * 4, '3 parts (exact)'),
                                 8: ('two draws per workgroup, every wave chain + serial part, all 8 stimuli in one chain per step',
                                     2 * 2, '2 parts (23 bits, round to nearest)')}[variant]
        executed = achieved * (2 * M) / (2 * M + 8) * per_flop * pad
        name = {4: 'wide', 7: 'wide', 8: 'duo'}.get(variant, 'split')
        return {'bound': 'mfma', 'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s', 'frac': achieved / peak,
                'traffic': traffic, 'kernel': 'gen_forward_%s_kernel (fp16-split MFMA, %s)' % (name, form),
                'mfma_dtype': 'f16 (W = 2 parts = 23 bits by round to nearest, state = %s, exact products, fp32 accumulate)' % state,
                'executed_mfma_tflops': executed, 'executed_frac': executed / peak,
                'frac_of_fp32_peak': achieved / PEAK_FP32_VALU_TFLOPS,
                'kernel_ms': kernel_ms, 'flops_per_unit': 2 * M + 8, 'ssn_steps_per_s_in_loop': steps_in_loop}
    return {'bound': 'mfma' if variant in (2, 3) else 'valu_fp32', 'achieved': achieved, 'peak': PEAK_FP32_VALU_TFLOPS,
            'unit': 'TFLOP/s', 'frac': achieved / PEAK_FP32_VALU_TFLOPS, 'traffic': traffic,
            'kernel': 'gen_forward_mfma_kernel (fp32 MFMA, %s)' % groups[variant] if variant in (2, 3) else 'gen_forward_kernel',
            'kernel_ms': kernel_ms, 'flops_per_unit': 2 * M + 8, 'ssn_steps_per_s_in_loop': steps_in_loop}


Z_MODES = {'philox': 'device-side z (Philox)',
           'refstream': "z = the reference's RandomState stream (seed 0, shared with the minibatch sampler) continued on the device "
                        'bit for bit (ssn_mt19937_random_sample_f32)',
           'numpy': "z = rng.rand drawn by numpy on the host, as the reference does (--z-host-draw)"}


def run_c3(args, rank, world, local_rank, paper=False, z_mode='refstream', comparisons=True):
    """``paper=True``: the shape of the reference's published run (scripts/fig4/gan/run.json): 2N=202, 128 models,
    seqlen 240 / skip 200, tau_E=2, deg-heteroin SSN, 4x128 critic with LayerNorm on layers 2-4, rmsprop.
    Default: BASELINE config 3/4: the bptt_cwgan loop.  One step = one GAN iteration = critic_iters (5) critic
    updates (each with a fresh generator forward) + one generator BPTT update; 1024 weight draws x 8
    bandwidths per GPU, 2N=200, seqlen 1200 / skip 1000, 3x512 critic on bf16 MFMA, adam-wgan.
    N>1: num_models = 1024*N sharded 1024 per rank, one RCCL all-reduce per update (weak scaling);
    value = N x iterations/s, i.e. 1024-model GAN iterations per second over the whole job."""
    import torch
    import torch.distributed as dist
    def timed_loop(gen_kernel, z_mode=z_mode):
        """warm-up + `steps` GAN iterations of a FRESH GAN (same seeds), barrier + synchronize on both sides, MAX over ranks."""
        gan, shape, bandwidths = make_c3_gan(world, rank, paper=paper, disc_precision=args.disc_precision, gen_kernel=gen_kernel,
                                             z_mode=z_mode)
        gan.reducer.timed = dist.is_initialized()
        it = gan.learning()

        def one_iter():
            while True:
                info = next(it)
                if not info.is_discriminator:
                    return info

        for _ in range(args.warmup):
            one_iter()
        gan.host_draw_seconds = 0.0
        gan.reducer.collective_ms()
        calls0 = gan.reducer.calls
        from tc_gan_amd import genops as _genops
        _genops.FORWARD_EVENTS = []             # every plain forward of the timed iterations, between two events on its stream
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            info = one_iter()
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        fwd_events, _genops.FORWARD_EVENTS = _genops.FORWARD_EVENTS, None
        timed_loop.forward_ms_in_loop = float(np.mean([a.elapsed_time(b) for a, b in fwd_events])) if fwd_events else None
        phases = None
        if dist.is_initialized():
            t = torch.tensor([elapsed], device='cuda', dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
            # where a rank's time goes besides its kernels: device time inside the all-reduces (includes waiting for the
            # slowest rank) and host time in the RNG draws, per GAN iteration, max / min over ranks
            mine = torch.tensor([gan.reducer.collective_ms() / args.steps, gan.host_draw_seconds * 1e3 / args.steps],
                                device='cuda', dtype=torch.float64)
            hi, lo = mine.clone(), mine.clone()
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            phases = {'allreduce_ms': {'max': float(hi[0]), 'min': float(lo[0])},
                      'host_draw_ms': {'max': float(hi[1]), 'min': float(lo[1])},
                      'collectives_per_iteration': (gan.reducer.calls - calls0) / args.steps}
        assert np.isfinite(info.gen_loss)
        timed_loop.host_draw_ms = gan.host_draw_seconds * 1e3 / args.steps
        return gan, shape, bandwidths, elapsed, info, phases

    gan, (N, models, NB, T, skip), bandwidths, elapsed, info, phases = timed_loop(os.environ.get('BENCH_GEN_KERNEL', 'auto'))    # (A/B runs only)
    forward_ms_in_loop = timed_loop.forward_ms_in_loop
    host_draw_ms = timed_loop.host_draw_ms
    # the same loop -- same seeds, fresh GAN, same warm-up, steps and max-over-ranks -- with the generator's W.r on the fp32
    # matrix instructions (W and state carried with all 24 bits), so that the line holds both numbers (`fp32_mfma` below)
    fp32_ms = None
    if comparisons and not paper and args.steps >= 2 and gan.gen.forward_variant(models) in SPLIT_VARIANTS:
        fp32_ms = timed_loop('mfma-fp32')[3] / args.steps * 1e3
    # and with the one-launch backward (gen_kernel duo-fused: adjoint sweep + dL/dW on chip) in place of the two launches
    fused_ms = None
    if comparisons and args.steps >= 2 and gan.gen.forward_variant(models) == 8:
        fused_ms = timed_loop('duo-fused')[3] / args.steps * 1e3
    # the same loop with z from a Philox stream (--z-device-seed: another noise stream than the reference's; no generator state
    # to hand back to the host, W straight from the counter)
    philox_ms = None
    if comparisons and z_mode == 'refstream' and args.steps >= 2:
        philox_ms = timed_loop(os.environ.get('BENCH_GEN_KERNEL', 'auto'), z_mode='philox')[3] / args.steps * 1e3
    # dominant kernel: gen_forward_kernel, timed alone with HIP events on the launch stream
    bw = np.tile(np.asarray(bandwidths, dtype='float32')[None], (models, 1))
    kw = dict(stimulator_bandwidths=bw, stimulator_contrasts=np.full_like(bw, 20.0), prober_norm_probes=np.zeros(models),
              prober_model_ids=np.arange(models), prober_cell_types=np.zeros(models))
    from tc_gan_amd import genops
    noise = gan.gen.gen_noise(gan.rng, bw)
    ext, z, W = gan.gen._device_inputs(bw, kw['stimulator_contrasts'], noise['model_zs'], noise.get('model_zs_in'))
    gp = gan.gen.gen_params(200.0)
    genops.gen_forward(W, ext, gp)
    out_extra_kernel = gan.gen.gen_kernel
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        genops.gen_forward(W, ext, gp)
    e1.record()
    torch.cuda.synchronize()
    kernel_ms_alone = e0.elapsed_time(e1) / 3
    # the roofline is priced on the kernel as it runs INSIDE the loop (mean over the plain forwards of the timed iterations:
    # the card is at its power limit there and the clock lower than for the kernel alone)
    kernel_ms = forward_ms_in_loop if forward_ms_in_loop else kernel_ms_alone
    M = 2 * N
    units = float(M) * models * NB * T                     # neuron-steps of one generator forward (per rank)
    achieved = units * (2 * M + 8) / (kernel_ms * 1e-3) * 1e-12
    iters_per_s = args.steps / elapsed
    variant = genops.forward_variant(models, NB, M, gp)
    traffic = None
    tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'hbm_traffic.json')
    if not paper and os.path.exists(tpath):
        traffic = json.load(open(tpath)).get('c3')
    out = {
        'metric': 'GAN iters/sec', 'value': iters_per_s * world, 'unit': '%d-model GAN iterations/s' % models,
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32 (critic GEMMs %s%s)' % (args.disc_precision, ('; generator W.r on fp16 matrix cores as an exact-product split of 23-bit operands' +
                                                                 ('; dL/dW on fp16 matrix cores, operands as two fp16 parts to 2^-24' if variant == 8 else ''))
                                                                if variant in SPLIT_VARIANTS else ''),
        'data': 'synthetic',
        'config': {'workload': ('C3 paper shape (scripts/fig4/gan/run.json): 2N=202, 128 models x 8 bandwidths per GPU, '
                                'seqlen 240 / skip 200, tau_E=2, deg-heteroin, 4x128 critic with LayerNorm on layers 2-4, '
                                'rmsprop, ' + Z_MODES[z_mode]) if paper else
                               'C3: bptt_cwgan loop, 2N=200, 1024 models x 8 bandwidths per GPU, seqlen 1200 / skip 1000, '
                               '5 critic updates + 1 generator BPTT update per iteration, 3x512 critic, adam-wgan '
                               '(generator step size %g, critic 0.01), ' % gan.gen_updaters[gan._pnames[0]].learning_rate
                               + Z_MODES[z_mode], 'parallelism': 'models sharded over %d GPU(s), one all-reduce per update' % world},
        'roofline': _forward_roofline(variant, achieved, traffic, kernel_ms, M, 7 * units * iters_per_s * world),
        'last_gen_loss': info.gen_loss, 'gen_kernel': out_extra_kernel, 'forward_variant': variant,
    }
    # (kernel_ms: mean duration of the plain forwards inside the timed loop; kernel_ms_alone: the same launch repeated by
    # itself after the loop, on an idle card)
    out['roofline']['kernel_ms_alone'] = kernel_ms_alone
    # host time per iteration inside the loop's RNG draws (minibatch choice, eps, and -- z_mode 'numpy' -- z itself; in 'refstream'
    # mode the part of ssn_mt19937_random_sample_f32 the host waits for: the state after the draw)
    out['z_mode'] = z_mode
    out['host_draw_ms'] = host_draw_ms
    if fp32_ms is not None:
        out['fp32_mfma'] = {'ms_per_step': fp32_ms, 'value': world * 1e3 / fp32_ms, 'steps': args.steps, 'warmup': args.warmup,
                            'note': 'same loop on a fresh GAN with the same seeds, generator forward and adjoint on the fp32 MFMA '
                                    'kernels (gen_kernel mfma-fp32: fp32 operands), same warm-up, steps and max-over-ranks timing'}
    if fused_ms is not None:
        out['fused_backward'] = {'ms_per_step': fused_ms, 'value': world * 1e3 / fused_ms, 'steps': args.steps, 'warmup': args.warmup,
                                 'note': 'same loop on a fresh GAN with the same seeds, generator backward as ONE launch (gen_kernel '
                                         'duo-fused: ssn_gen_backward_fused_f32) instead of adjoint sweep + dL/dW product'}
    if philox_ms is not None:
        out['philox_z'] = {'ms_per_step': philox_ms, 'value': world * 1e3 / philox_ms, 'steps': args.steps, 'warmup': args.warmup,
                           'note': 'same loop on a fresh GAN, z from one Philox4x32-10 stream sharded over the ranks (--z-device-seed): '
                                   "NOT the reference's noise stream; the difference to ms_per_step is what continuing the RandomState costs"}
    if phases is not None:
        out['phases'] = phases
    if not paper and rank == 0:
        out['critic'] = _critic_update_roofline(gan, models)
    out['world_size'] = world
    return out


def _critic_update_roofline(gan, rows):
    """north_star: "MFMA utilisation reported against gfx950 peak" for the critic's bf16 GEMMs.  Device time of ONE critic loss +
    gradient at the loop's shape (3 x `rows` input rows), calls back to back behind a long fill so that the queue stays ahead of
    the device (HIP events on torch's stream: the calls launch there), against the ALGORITHMIC flops of the update: forward of the
    3 x rows, backward chain of the 2 x rows of [xg; xd], input-gradient chain and second chain of the penalty rows, the
    weight gradients of both halves (2 flops per multiply-add of every layer GEMM)."""
    import numpy as np
    import torch
    from tc_gan_amd.critic import Critic
    disc = gan.discriminator if hasattr(gan, 'discriminator') else gan.disc
    c = getattr(disc, 'critic', disc)
    layers = [int(d) for d in c.dims[1:]]
    nx = c.nx
    cc = Critic(nx, layers, precision='bf16' if c.precision == 0 else 'fp32', normalization='none', nonlinearity=c.nonlinearity)
    rs = np.random.RandomState(0)
    xg, xd = (torch.as_tensor(rs.rand(rows, nx) * 5, device='cuda', dtype=torch.float32) for _ in range(2))
    xp = 0.5 * (xg + xd)
    cond = torch.as_tensor(np.stack([np.full(rows, 20.), rs.rand(rows), np.zeros(rows)], 1), device='cuda', dtype=torch.float32)
    big = torch.empty(1 << 27, device='cuda')

    def run(n):
        big.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            cc.loss_grad(xg, cond, xd, cond, xp, cond, 10.0)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    run(3)
    us = min(run(6), run(6))
    dims = [nx + 3] + layers
    per_row = sum(dims[l] * dims[l + 1] for l in range(len(layers))) + layers[-1]      # multiply-adds of one row's pass
    hidden = sum(dims[l] * dims[l + 1] for l in range(1, len(layers)))                 # ... of the backward chain (no input layer)
    macs = 3 * rows * per_row + 2 * rows * hidden + rows * (hidden + dims[0] * dims[1]) + rows * per_row + 3 * rows * per_row
    tf = 2.0 * macs / (us * 1e-6) * 1e-12
    peak = 2516.8
    return {'loss_grad_us': us, 'algorithmic_gflop': 2.0 * macs * 1e-9, 'achieved': tf, 'peak': peak, 'unit': 'TFLOP/s', 'frac': tf / peak,
            'bound': 'mfma', 'mfma_dtype': 'bf16 (fp32 master values rounded to nearest even, fp32 accumulate)',
            'rows': 3 * rows, 'layers': layers,
            'path': 'row-block kernel + batched weight gradients (ssn_critic_rows.hip, 8 launches)'
                    if os.environ.get('SSN_CRITIC_ROWS', '1') != '0' and c.precision == 0 else 'layer-by-layer GEMM chain',
            'note': 'device time of one loss + gradient (no optimizer step), back-to-back calls; the update is bound by the '
                    "CUs' 64 B/clock path from L2 (every 32-row block reads all packed weights per layer) and by launch count, not by the matrix pipe"}


def run_c5(args, rank, world, local_rank):
    """BASELINE config 5: the FF_lalazar feed-forward generator (get_FF_output), box_width 40 (64000 grid
    points), 27 stimuli, 1 hidden unit, 16384 samples per GPU; synthetic inputs as generate_samples draws them
    (uniform widths / strengths, box^3 / 100 = 640 connections per unit drawn with replacement).  One step = one forward
    over the batch through the connection-list entry point (`ssn_ff_forward_sparse_f32`: 4 B per (sample, grid point) +
    8 B per connection); the dense entry point (12 B per point, what round 3 timed) is timed beside it on the same inputs."""
    import torch
    import torch.distributed as dist
    from tc_gan_amd import ff_model
    nsam, box, nhid = 16384, 40, 1
    G = box ** 3
    nff = G // 100
    gen = torch.Generator(device='cuda'); gen.manual_seed(99 + rank)
    wid = torch.rand((nsam, G), device='cuda', generator=gen)
    # FF_lalazar_model.py:154-167: nff draws with replacement per unit; an index drawn twice counts once
    idx = torch.randint(0, G, (nsam, nhid, nff), device='cuda', generator=gen).sort(dim=2).values
    idx[:, :, 1:][idx[:, :, 1:] == idx[:, :, :-1]] = -1
    idx = idx.to(torch.int32).contiguous()
    strn = torch.rand((nsam, nhid, G), device='cuda', generator=gen)
    val = torch.gather(strn, 2, idx.clamp(min=0).long()).contiguous()
    ths = torch.rand((nsam, nhid), device='cuda', generator=gen) * 2 - 1
    stim = ff_model.default_stimuli()

    def fwd():
        return ff_model.ff_forward_sparse(ff_model.START_PARAMS, wid, idx, val, ths, stim, box)

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    for _ in range(args.warmup):
        out = fwd()
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        out = fwd()
        ev[k][1].record()
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist.is_initialized():
        t = torch.tensor([elapsed], device='cuda', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert bool(torch.isfinite(out).all())
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    # the dense entry point on the same inputs (FF_con / FF_str as arrays): same outputs, three per-point streams
    con = torch.zeros((nsam, nhid, G), device='cuda')
    rows = torch.arange(nsam * nhid, device='cuda').reshape(nsam, nhid, 1).expand_as(idx)
    ok = idx >= 0
    con.view(nsam * nhid, G)[rows[ok], idx[ok].long()] = 1.0
    dense = ff_model.ff_forward(ff_model.START_PARAMS, wid, con, strn, ths, stim, box)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        dense = ff_model.ff_forward(ff_model.START_PARAMS, wid, con, strn, ths, stim, box)
    e1.record()
    torch.cuda.synchronize()
    dense_ms = e0.elapsed_time(e1) / 3
    assert float((dense - out).abs().max()) <= 1e-4 * max(float(dense.abs().max()), 1.0)
    del con, dense
    # The pass over the widths is bound by the vector unit, not by HBM: per (sample, point) 27 FMAs (the 3 x 3 x 3 sums of
    # exp products) + 18 multiplies + 9 exponentials + ~6 = 87 flop against 4 bytes
    flops_alg = float(nsam) * G * 87.0
    bytes_alg = float(nsam) * (G * 4 + nhid * nff * 8)
    achieved = flops_alg / (kernel_ms * 1e-3) * 1e-12
    res = {'metric': 'FF tuning curves/sec', 'value': nsam * args.steps * world / elapsed, 'unit': 'samples (27-point curves)/s',
           'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
           'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
           'config': {'workload': 'C5: FF_lalazar get_FF_output, box_width 40 (64000 points), 27 stimuli, 1 hidden unit, '
                                  '16384 samples per GPU, connections as lists (640 slots per unit)',
                      'parallelism': 'samples sharded over %d GPU(s)' % world},
           'roofline': {'bound': 'valu_fp32', 'achieved': achieved, 'peak': 157.3, 'unit': 'TFLOP/s', 'frac': achieved / 157.3,
                        'traffic': _traffic('c5'), 'kernel': 'ff_forward_sparse_lattice_kernel', 'kernel_ms': kernel_ms,
                        'flops_per_unit': 87, 'algorithmic_hbm_bytes': bytes_alg,
                        'hbm_GBps': bytes_alg / (kernel_ms * 1e-3) * 1e-9, 'hbm_frac': bytes_alg / (kernel_ms * 1e-3) * 1e-9 / 8000.0,
                        'dense_entry_ms': dense_ms, 'dense_entry_hbm_bytes': float(nsam) * nhid * G * 12}}
    res['world_size'] = world
    return res


def run_c1_dropin(args):
    """BASELINE config 1 through the boundary a maintainer gets by copying libssnode.so into tc_gan/ext/:
    `find_fixed_points_parallel`'s pattern (ssnode.py:423-510) -- a pool of one Python thread per usable core, one task per
    weight draw, one `solve_dynamics_asym_tanh_euler` call per (draw, stimulus), host fp64 buffers -- timed on the
    GPU library and, beside it, on the reference's own C build.  PCIe, launch and synchronisation are all inside
    the timed region (this is the host-buffer entry point); `roofline` is null: a one-workgroup fp64 solve per
    call is latency-bound by construction."""
    from tc_gan_amd import stimuli, weight_gen
    from tc_gan_amd.clib import libssnode
    N, B, NB, T, desc = WORKLOADS['c1']
    M = 2 * N
    affinity, quota = usable_cores()
    threads = min(affinity, quota or affinity, 64)       # (the drop-in leg's own pool: the container's share, at most 64 callers)
    fn = _bind_solve(libssnode.solve_dynamics_asym_tanh_euler)
    sample = args.cpu_sample or B * max(args.steps, 1)
    J, D, S = new_jds()
    rs = np.random.RandomState(0)
    Ws = list(weight_gen.generate_weight_batch(N, J, D, S, rs.rand(sample, M, M), dtype='float64').cpu().numpy())
    exts = stimuli.input([1.0], np.linspace(-.5, .5, N), 0.25 / 8, [20.])
    t_pool = time_threaded_solves(fn, Ws, exts, N, T, threads)
    t_one = time_threaded_solves(fn, Ws[:64], exts, N, T, 1)
    units = float(M) * NB * T
    value = units * sample / t_pool
    out = {'metric': 'SSN-steps/sec', 'value': value, 'unit': 'neuron*batch*Euler-steps/s', 'n_gpus': 1,
           'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': t_pool / sample * B * 1e3, 'higher_is_better': True,
           'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
           'config': {'workload': desc + ', via the drop-in symbols: %d host threads, one solve_dynamics_asym_tanh_euler '
                                         'call per (draw, stimulus), host buffers (PCIe, launch and sync inside the timed '
                                         'region)' % threads,
                      'calls': sample * NB, 'us_per_call_pool': t_pool / (sample * NB) * 1e6,
                      'us_per_call_single_thread': t_one / (min(sample, 64) * NB) * 1e6,
                      'single_thread_value': units * min(sample, 64) / t_one},
           'roofline': None}
    if not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(N, NB, T, sample)
        out['cpu_baseline']['gpu_over_cpu'] = value / out['cpu_baseline']['value']
    return out


def self_launch(args, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves, as children of a parent that
    never touches the GPU (no HIP call, no torch.cuda call before the spawn; the parent is never replaced by
    another program).  The children are the driver's own command form (`python -m torch.distributed.run
    --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...`); rank 0's single JSON
    line goes to our stdout unchanged and we exit with the launcher's code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')        # dmabuf IPC: RCCL needs it on this driver
    env.setdefault('OMP_NUM_THREADS', '4')
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--workload', default='c2', choices=sorted(WORKLOADS) + ['c3', 'c3paper', 'c5'])
    ap.add_argument('--variant', type=int, default=-1, help='-1 auto, 0 streaming, 1 register-stationary DPP, 2 tile (library picks the shape), '
                    '3 tile/split residency, 4 tile/all-register, 5 fp32 MFMA (NB >= 4), 6 fp16-split MFMA (NB >= 4), 7 the same in the alternating form, '
                    '8 fp16-split MFMA with two draws per workgroup')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-sample', type=int, default=0, help='weight draws in the CPU baseline sample (0 = auto)')
    ap.add_argument('--secondary-steps', type=int, default=10,
                    help='GAN iterations of the C3 run whose line rides along as `secondary` with the default (C2) '
                         'workload, so that one command covers both halves of BASELINE.json.metric (0 = off)')
    ap.add_argument('--no-extras', dest='extras', action='store_false',
                    help='default (C2) workload on one GPU: skip the short c2nb8 / c5 / c1-dropin / c3paper runs that ride along '
                         'as `extras`')
    ap.add_argument('--z-mode', default='refstream', choices=sorted(Z_MODES),
                    help="c3 / c3paper: where z comes from (see make_c3_gan)")
    ap.add_argument('--disc-precision', default='bf16', choices=['bf16', 'fp32'],
                    help='c3: critic GEMM operand precision (BASELINE config 3 names bf16 MFMA)')
    ap.add_argument('--via', default='batched', choices=['batched', 'dropin'],
                    help="c1 only: 'dropin' times the reference's call pattern (one solve_dynamics_* call per "
                         "(draw, stimulus) from a pool of one thread per usable core) through the zero-change drop-in symbols")
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(args, sys.argv[1:]))            # before anything touches the GPU

    # Exactly ONE line on stdout: libraries write there as well (RCCL prints a five-line version banner when its first
    # communicator comes up), so the stream the caller reads is kept aside for the JSON line and file descriptor 1 points at
    # stderr for the rest of the run.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from tc_gan_amd import clib

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        sys.exit('bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks; refusing to report a line whose '
                 'n_gpus is not the number of ranks that ran' % (args.gpus, world))
    # BENCH_DIST_BACKEND=gloo lets several ranks share ONE card to rehearse the N>1 code path (never a result)
    backend = os.environ.get('BENCH_DIST_BACKEND', 'nccl')
    local_rank = local_rank % max(torch.cuda.device_count(), 1) if backend != 'nccl' else local_rank
    torch.cuda.set_device(local_rank)
    clib.require_gpu()
    # BENCH_FORCE_DIST=1: join a process group even as a single rank, so that the collectives of the N > 1 path (flat-buffer
    # all-reduce per update, barriers, max-over-ranks timing) run through the chosen backend -- RCCL -- on a one-GPU box
    force_dist = world == 1 and os.environ.get('BENCH_FORCE_DIST') == '1'
    if world > 1 or force_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if force_dist:
            import socket
            with socket.socket() as sk:
                sk.bind(('127.0.0.1', 0))
                os.environ.setdefault('MASTER_PORT', str(sk.getsockname()[1]))
            os.environ['TCGAN_DIST_SINGLE_RANK'] = '1'       # GradientAllReducer: reduce over the one rank, too
        dist.init_process_group(backend, rank=rank, world_size=world)
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)

    if args.workload == 'c3paper':
        out = run_c3(args, rank, world, local_rank, paper=True, z_mode=args.z_mode)
    elif args.workload == 'c3':
        out = run_c3(args, rank, world, local_rank, z_mode=args.z_mode)
    elif args.workload == 'c5':
        out = run_c5(args, rank, world, local_rank)
    elif args.workload == 'c1' and args.via == 'dropin':
        out = run_c1_dropin(args)
    else:
        out = run_solver(args, rank, world, local_rank)
        if args.workload == 'c2' and args.secondary_steps > 0:
            # the second half of BASELINE.json.metric ("GAN iters/sec"): a short C3 run in the same job
            sub = argparse.Namespace(**vars(args))
            sub.steps, sub.warmup = args.secondary_steps, 3
            sec = run_c3(sub, rank, world, local_rank)
            out['secondary'] = {k: sec[k] for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step',
                                                    'higher_is_better', 'scaling', 'dtype', 'data', 'config', 'roofline',
                                                    'last_gen_loss')}
            for key in ('fp32_mfma', 'fused_backward', 'philox_z', 'phases', 'gen_kernel', 'forward_variant', 'z_mode', 'host_draw_ms', 'critic'):
                if key in sec:
                    out['secondary'][key] = sec[key]
        if args.workload == 'c2' and args.extras and world == 1:
            # The other workloads SURVEY.md section 8(d) names, short runs in the same job so that the driver's record holds
            # them too: C2 with the 8 bandwidths every real caller uses, C5, C1 through the drop-in symbols, the paper's shape.
            out['extras'] = {}
            keep = ('metric', 'value', 'unit', 'steps', 'warmup', 'ms_per_step', 'dtype', 'config', 'roofline', 'cpu_baseline',
                    'gen_kernel', 'forward_variant', 'last_gen_loss', 'fused_backward', 'philox_z', 'z_mode', 'host_draw_ms', 'numpy_host_draw_ms')
            for name, workload, steps, warmup, kw in (('c2nb8', 'c2nb8', 3, 1, {}), ('c5', 'c5', 5, 1, {}),
                                                      ('c1_dropin', 'c1', 1, 0, dict(via='dropin', cpu_sample=128)),
                                                      ('c3paper', 'c3paper', 20, 5, {}),
                                                      ('c3_refstream', 'c3', max(args.secondary_steps, 3), 3, {})):
                sub = argparse.Namespace(**dict(vars(args), workload=workload, steps=steps, warmup=warmup, variant=-1,
                                                no_cpu_baseline=(name != 'c1_dropin'), **kw))
                t_extra = time.perf_counter()
                if name == 'c3_refstream':
                    # the C3 loop WITHOUT --z-device-seed -- the reference's noise stream (RandomState seed 0), z continued on the
                    # device -- is what `secondary` holds since round 5; repeated here under the name VERDICT r4 asked for, with
                    # what numpy takes to draw one iteration's six z tensors on this host beside it
                    if 'secondary' in out and out['secondary'].get('z_mode') == 'refstream':
                        res = dict(out['secondary'], config=dict(out['secondary']['config'], note='= `secondary` of this line'))
                    else:
                        res = run_c3(sub, rank, world, local_rank, z_mode='refstream', comparisons=False)
                    rs_t = np.random.RandomState(0)
                    t_np = time.perf_counter()
                    rs_t.rand(1024, 200, 200)
                    res['numpy_host_draw_ms'] = (time.perf_counter() - t_np) * 1e3 * 6     # six z draws per iteration on this host
                elif workload == 'c3paper':
                    res = run_c3(sub, rank, world, local_rank, paper=True)
                elif workload == 'c5':
                    res = run_c5(sub, rank, world, local_rank)
                elif name == 'c1_dropin':
                    res = run_c1_dropin(sub)
                else:
                    res = run_solver(sub, rank, world, local_rank)
                out['extras'][name] = dict({k: res[k] for k in keep if k in res}, wall_s=time.perf_counter() - t_extra)
                torch.cuda.empty_cache()
            if not dist.is_initialized() and backend == 'nccl':
                # RCCL on the one card there is: the paper-shape loop once more inside a ONE-rank nccl group, every collective
                # of the N > 1 path (flat-buffer all-reduce per update, barriers, max-over-ranks timing) going through RCCL.
                # Says nothing about xGMI; it shows that the communicator comes up and what a collective costs in stream order.
                t_extra = time.perf_counter()
                try:
                    out['extras']['c3paper_rccl1'] = dict(run_single_rank_group(args, local_rank), wall_s=time.perf_counter() - t_extra)
                except Exception as err:                      # (a rehearsal must not take the line down with it)
                    out['extras']['c3paper_rccl1'] = {'error': repr(err)[:300], 'wall_s': time.perf_counter() - t_extra}
                torch.cuda.empty_cache()
    # what actually ran: the process group's own size and backend (1 / none for a single process)
    out['world_size'] = dist.get_world_size() if dist.is_initialized() else 1
    out['dist_backend'] = dist.get_backend() if dist.is_initialized() else None
    assert out['n_gpus'] == args.gpus == out['world_size']
    if rank == 0:
        os.write(json_fd, (json.dumps(out) + '\n').encode())
    os.close(json_fd)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def run_single_rank_group(args, local_rank):
    """`--workload c3paper` for a few steps as the only rank of an `nccl` (= RCCL) process group (see BENCH_FORCE_DIST)."""
    import socket
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        os.environ['MASTER_PORT'] = str(sk.getsockname()[1])
    os.environ['TCGAN_DIST_SINGLE_RANK'] = '1'
    dist.init_process_group('nccl', rank=0, world_size=1)
    try:
        sub = argparse.Namespace(**dict(vars(args), workload='c3paper', steps=20, warmup=5, variant=-1, no_cpu_baseline=True))
        res = run_c3(sub, 0, 1, local_rank, paper=True)
        return {'ms_per_step': res['ms_per_step'], 'value': res['value'], 'unit': res['unit'], 'steps': 20, 'warmup': 5,
                'phases': res.get('phases'), 'dist_backend': dist.get_backend(), 'world_size': dist.get_world_size(),
                'last_gen_loss': res.get('last_gen_loss'),
                'config': {'workload': 'C3 paper shape as the single rank of an RCCL group: one flat all-reduce per update '
                                       '(mean over one rank = identity), barriers and timing reductions through RCCL'}}
    finally:
        dist.destroy_process_group()
        os.environ.pop('TCGAN_DIST_SINGLE_RANK', None)


def run_solver(args, rank, world, local_rank):
    import torch
    import torch.distributed as dist
    from tc_gan_amd import clib
    from tc_gan_amd.clib import libssnode
    N, B, NB, T, desc = WORKLOADS[args.workload]
    M = 2 * N
    J, D, S = new_jds()
    dev = torch.device('cuda', local_rank)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)                      # per-rank stream: independent weight draws
    z = torch.rand((B, M, M), device=dev, dtype=torch.float32, generator=gen)
    W = torch.empty_like(z)
    bws = [1.0] if NB == 1 else [0, 0.0625, 0.125, 0.1875, 0.25, 0.5, 0.75, 1][:NB]
    bw = torch.tensor([bws], device=dev, dtype=torch.float32)
    con = torch.full_like(bw, 20.0)
    ext = torch.empty((1, NB, M), device=dev, dtype=torch.float32)
    r = torch.zeros((B, NB, M), device=dev, dtype=torch.float32)
    codes = torch.empty((B, NB), device=dev, dtype=torch.int32)
    steps_out = torch.empty((B, NB), device=dev, dtype=torch.int32)
    p = clib.SolverParams(io_type=clib.SSN_IO_TANH, max_iter=T, k=0.01, n=2.2, tau_E=0.01589, tau_I=0.002,
                          dt=8e-4, atol=0.0, rate_soft_bound=200.0, rate_hard_bound=1000.0)
    f4 = ctypes.c_float * 4
    Jc, Dc, Sc = (f4(*np.asarray(a, dtype=float).reshape(4)) for a in (J, D, S))
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    clib.check(libssnode.ssn_stimulus_f32(bw.data_ptr(), con.data_ptr(), ctypes.c_float(0.25 / 8), ext.data_ptr(),
                                          1, NB, N, stream), 'ssn_stimulus_f32')

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]

    def one_step(k=None):
        r.zero_()
        clib.check(libssnode.ssn_build_w_f32(z.data_ptr(), Jc, Dc, Sc, W.data_ptr(), B, N, stream), 'ssn_build_w_f32')
        if k is not None:
            ev[k][0].record()
        if args.variant < 0:
            rc = libssnode.ssn_solve_batch_f32(W.data_ptr(), ext.data_ptr(), 0, r.data_ptr(), None, codes.data_ptr(),
                                               steps_out.data_ptr(), B, NB, M, ctypes.byref(p), stream)
        else:
            rc = libssnode.ssn_solve_batch_f32_variant(args.variant, W.data_ptr(), ext.data_ptr(), 0, r.data_ptr(), None,
                                                       codes.data_ptr(), steps_out.data_ptr(), B, NB, M,
                                                       ctypes.byref(p), stream)
        clib.check(rc, 'ssn_solve_batch_f32')
        if k is not None:
            ev[k][1].record()

    for _ in range(args.warmup):
        one_step()
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        one_step(k)
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist.is_initialized():
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # sanity: the work really ran (all pairs took exactly T steps with code 1, states finite)
    assert int((steps_out != T).sum()) == 0 and int((codes != 1).sum()) == 0
    assert bool(torch.isfinite(r).all())

    units_per_step = float(M) * B * NB * T                      # neuron x batch x Euler steps (per rank)
    value = units_per_step * args.steps * world / elapsed
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    flops_per_unit = 2 * M + 8
    achieved = units_per_step * flops_per_unit / (kernel_ms * 1e-3) * 1e-12
    traffic = None
    tpath = os.path.join(ROOT, 'profiles', 'hbm_traffic.json')   # PMC-derived bytes per launch, if collected
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(args.workload)
        except Exception:
            traffic = None
    # the variant that ran: the forced one, or what the library says it picks for this call shape and these parameters
    fast = args.variant if args.variant >= 0 else libssnode.ssn_solve_batch_variant_for(B, NB, M, 4, ctypes.byref(p))
    assert fast >= 0
    out = {
        'metric': 'SSN-steps/sec', 'value': value, 'unit': 'neuron*batch*Euler-steps/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': desc + ', asym_tanh, atol=0, fp32, W rebuilt from resident z each step',
                   'neurons': M, 'batch_per_gpu': B, 'stimuli_per_draw': NB, 'euler_steps': T,
                   'kernel': {8: 'solve_duo_kernel (fp16-split MFMA, two draws per workgroup)', 7: 'solve_split_kernel (fp16-split MFMA, alternating groups)', 6: 'solve_wide_kernel (fp16-split MFMA)', 5: 'solve_mfma_kernel', 4: 'solve_tile_kernel (all-register)', 3: 'solve_tile_kernel (split)', 2: 'solve_tile_kernel', 1: 'solve_regw_kernel', 0: 'solve_stream_kernel'}[int(fast)],
                   'parallelism': 'draws sharded over %d GPU(s), no data-path collective' % world},
        # fp32 VALU peak = fp32 MFMA (v_mfma_f32_4x4x1) peak = 157.3 TFLOP/s (MI355X_MICROARCH.md)
        'roofline': {'bound': 'mfma' if int(fast) in (5, 6, 7, 8) else 'valu_fp32', 'achieved': achieved, 'peak': PEAK_FP32_VALU_TFLOPS, 'unit': 'TFLOP/s',
                     'frac': achieved / PEAK_FP32_VALU_TFLOPS, 'traffic': traffic,
                     'kernel_ms': kernel_ms, 'flops_per_unit': flops_per_unit,
                     'algorithmic_hbm_bytes': B * (4 * M * M + 12 * M * NB)},
    }
    if int(fast) in (6, 7, 8):  # fp16-split MFMA solver: priced like the split generator forward (see _forward_roofline)
        wide = int(fast) == 6 and os.environ.get('SSN_FWD_WIDE', '2') != '0'
        rl = _forward_roofline(8 if int(fast) == 8 else (4 if wide else 6), achieved, traffic, kernel_ms, M, None)
        rl.pop('ssn_steps_per_s_in_loop')
        rl['kernel'] = (rl['kernel'].replace('gen_forward_wide_kernel', 'solve_wide_kernel').replace('gen_forward_split_kernel', 'solve_split_kernel')
                        .replace('gen_forward_duo_kernel', 'solve_duo_kernel'))
        rl['algorithmic_hbm_bytes'] = B * (4 * M * M + 12 * M * NB)
        out['roofline'] = rl
        out['dtype'] = 'f32 (W.r on fp16 matrix cores as an exact-product split of 23-bit operands)'
    out['world_size'] = world
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # ~10-20 s of host work at C2: the whole batch (4096 draws, ~5 s per pass on 16 threads), warm-up + best of 2, after a
        # short probe that picks the pool size (cgroup share / all cores / 16 / 64: whichever runs the reference fastest)
        sample = args.cpu_sample or (64 if args.workload == 'c1' else min(B, max(128, 4096 // NB)))
        out['cpu_baseline'] = cpu_baseline(N, NB, T, sample)
        cb = out['cpu_baseline']
        cb['gpu_over_cpu'] = value / cb['value']
        # what the WHOLE host could give the reference at best: the one-thread rate times every logical CPU (linear scaling, an
        # upper bound: the measured pool reaches 16 x 0.99 of it on its 16-core share) -- the ratio nobody can measure from
        # inside a container that holds a share of the machine, stated so that `gpu_over_cpu` is not read as more than it is
        cb['all_host_cpus_linear_bound'] = {'value': cb['value_1thread'] * cb['host_cpu_count'],
                                            'gpu_over_cpu': value / (cb['value_1thread'] * cb['host_cpu_count'])}
    return out




if __name__ == '__main__':
    main()
