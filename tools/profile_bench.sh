#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes for the
# bench command; summaries land in gpurun_out/prof_<tag>/ and are then copied into profiles/.
# usage: tools/profile_bench.sh <tag> [bench args...]
set -o pipefail
tag=$1; shift
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
args="--steps 3 --warmup 1 --no-cpu-baseline $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py $args > $out/bench_trace.json 2> $out/trace.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 bench.py $args > $out/bench_fetch.json 2> $out/fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 bench.py $args > $out/bench_write.json 2> $out/write.log
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $out/pmc_sq -- python3 bench.py $args > $out/bench_sq.json 2> $out/sq.log
find $out -name "*.csv" | head -30
