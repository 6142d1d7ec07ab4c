#!/bin/bash
# Round-5 profile set (run on the GPU box via gpurun, one section per call; tools/collect_r05.sh then copies the summaries into
# profiles/r05/ and regenerates profiles/hbm_traffic.json from them):
#   tools/profile_r05.sh traces   kernel-trace stats of the default bench command (C2), the C3 GAN loop on the reference's noise
#                                 stream (the package default), the paper shape, C5, C2 with 8 stimuli
#   tools/profile_r05.sh pmc1     counter passes (tools/pmc_run.sh) for the device MT19937 draw and the C3 forward (plain, saving)
#   tools/profile_r05.sh pmc2     counter passes for the two-draw solver, the backward, C5, C2
set -o pipefail
root=$(cd "$(dirname "$0")/.." && pwd)
cd $root
out=$root/gpurun_out/prof_r05
mkdir -p $out
export TMPDIR=/tmp
case "$1" in
traces)
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/c2/trace -- python3 bench.py --steps 3 --warmup 1 --secondary-steps 0 --no-extras --no-cpu-baseline > $out/c2_bench.json 2> $out/c2_trace.log || exit 1
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/c3/trace -- python3 bench.py --workload c3 --steps 3 --warmup 1 > $out/c3_bench.json 2> $out/c3_trace.log || exit 1
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/c3paper/trace -- python3 bench.py --workload c3paper --steps 20 --warmup 3 > $out/c3paper_bench.json 2> $out/c3paper_trace.log || exit 1
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/c5/trace -- python3 bench.py --workload c5 --steps 5 --warmup 1 --no-cpu-baseline > $out/c5_bench.json 2> $out/c5_trace.log || exit 1
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/c2nb8/trace -- python3 bench.py --workload c2nb8 --steps 3 --warmup 1 --no-cpu-baseline > $out/c2nb8_bench.json 2> $out/c2nb8_trace.log || exit 1
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/mt/trace -- python3 tools/time_mt.py > $out/mt_time.log 2> $out/mt_trace.log || exit 1
  for t in c2 c3 c3paper c5 c2nb8 mt; do python3 tools/summarize_profile.py $out/$t $out/$t; done
  # (iteration 2: the last of the first loop -- the reference-stream loop; the fp32 / one-launch-backward / Philox comparison loops follow)
  python3 tools/trace_gaps.py $out/c3/trace "gen_forward_duo_kernel<208, true" $out/c3_iteration.csv 2 > $out/c3_gaps.txt 2>&1
  # (paper shape: an iteration of the first loop, the reference-stream one; the Philox comparison loop follows it)
  python3 tools/trace_gaps.py $out/c3paper/trace "gen_forward_split_kernel<208, true" $out/c3paper_iteration.csv 15 > $out/c3paper_gaps.txt 2>&1
  ;;
pmc1)
  bash tools/pmc_run.sh r05_mt tools/time_mt.py > $out/pmc_mt.log 2>&1 || exit 1
  bash tools/pmc_run.sh r05_fwd tools/time_fwd.py 8 4 > $out/pmc_fwd.log 2>&1 || exit 1
  bash tools/pmc_run.sh r05_fwdsave tools/time_fwd.py 8 --save > $out/pmc_fwdsave.log 2>&1 || exit 1
  ;;
pmc2)
  bash tools/pmc_run.sh r05_solve tools/time_solver.py 8 6 > $out/pmc_solve.log 2>&1 || exit 1
  bash tools/pmc_run.sh r05_adj tools/time_adj.py > $out/pmc_adj.log 2>&1 || exit 1
  bash tools/pmc_run.sh r05_c5 tools/time_c5.py > $out/pmc_c5.log 2>&1 || exit 1
  bash tools/pmc_run.sh r05_c2 bench.py --steps 3 --warmup 1 --secondary-steps 0 --no-extras --no-cpu-baseline > $out/pmc_c2.log 2>&1 || exit 1
  ;;
*) echo "usage: $0 traces|pmc1|pmc2"; exit 2 ;;
esac
ls $out
