#!/bin/bash
# HBM byte counters (separate passes) for the C3 GAN loop kernels (run on the GPU box via gpurun).
set -o pipefail
out=$PWD/gpurun_out/prof_$1
mkdir -p $out
export TMPDIR=/tmp
args="--workload c3 --steps 1 --warmup 1"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 bench.py $args > $out/b1.json 2> $out/fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 bench.py $args > $out/b2.json 2> $out/write.log
ls $out/pmc_fetch/*/ | head
