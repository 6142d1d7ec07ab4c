#!/bin/bash
# MFMA pipe counters for the C3 generator kernels (run on the GPU box via gpurun; own pass, no trace domains).
set -o pipefail
out=$PWD/gpurun_out/prof_$1
mkdir -p $out
export TMPDIR=/tmp
args="--workload c3 --steps 1 --warmup 1"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d $out/pmc_mfma -- python3 bench.py $args > $out/b3.json 2> $out/mfma.log
ls $out/pmc_mfma/*/ | head
