#!/bin/bash
# rocprofv3 kernel-trace stats of the C3 GAN loop (run on the GPU box via gpurun).
set -o pipefail
out=$PWD/gpurun_out/prof_$1
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --workload c3 --steps 3 --warmup 1 > $out/bench.json 2> $out/trace.log
cat $out/bench.json
cat $out/trace/*/*_kernel_stats.csv | cut -d, -f1-5 | cut -c1-150 | head -25
