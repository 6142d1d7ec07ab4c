#!/bin/bash
# Round-3 profile set (run on the GPU box via gpurun; summaries are then copied into profiles/r03/ by hand):
#   kernel-trace stats of the default bench command (C2) and of the C3 GAN loop, counter passes (tools/pmc_run.sh: MFMA / VALU /
#   LDS / wait counters, FETCH_SIZE, WRITE_SIZE, each its own rocprofv3 run) for the two-draw forward and solver kernels and
#   for the C2 tile kernel.
set -o pipefail
root=$(cd "$(dirname "$0")/.." && pwd)
cd $root
out=$root/gpurun_out/prof_r03
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/c2/trace -- python3 bench.py --steps 3 --warmup 1 --secondary-steps 0 --no-extras --no-cpu-baseline > $out/c2_bench.json 2> $out/c2_trace.log
rocprofv3 --kernel-trace --stats --output-format csv -d $out/c3/trace -- python3 bench.py --workload c3 --steps 3 --warmup 1 > $out/c3_bench.json 2> $out/c3_trace.log
rocprofv3 --kernel-trace --stats --output-format csv -d $out/c3paper/trace -- python3 bench.py --workload c3paper --steps 20 --warmup 3 > $out/c3paper_bench.json 2> $out/c3paper_trace.log
for t in c2 c3 c3paper; do python3 tools/summarize_profile.py $out/$t $out/$t; done
# steady-state iteration of the C3 loop: per-kernel totals, idle time between kernels (profiles/r03/c3_gan_loop_iteration.csv, _gaps.txt)
python3 tools/trace_gaps.py $out/c3/trace "gen_forward_duo_kernel<208, true" $out/c3_iteration.csv > $out/c3_gaps.txt 2>&1
bash tools/pmc_run.sh r03_fwd tools/time_fwd.py 8 4 > $out/pmc_fwd.log 2>&1
bash tools/pmc_run.sh r03_fwdsave tools/time_fwd.py 8 --save > $out/pmc_fwdsave.log 2>&1
bash tools/pmc_run.sh r03_solve tools/time_solver.py 8 6 > $out/pmc_solve.log 2>&1
bash tools/pmc_run.sh r03_adj tools/time_adj.py > $out/pmc_adj.log 2>&1
bash tools/pmc_run.sh r03_gw tools/time_gw.py MFMA > $out/pmc_gw.log 2>&1
bash tools/pmc_run.sh r03_c2 bench.py --steps 3 --warmup 1 --secondary-steps 0 --no-extras --no-cpu-baseline > $out/pmc_c2.log 2>&1
ls $out
