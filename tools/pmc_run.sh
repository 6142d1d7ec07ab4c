#!/bin/bash
# Counter passes for an arbitrary python command (run on the GPU box via gpurun):
#   tools/pmc_run.sh <tag> <script.py> [args...]
# Writes gpurun_out/pmc_<tag>/pmc_{1,2,3,4}/... (one rocprofv3 run per counter set, --pmc only, no trace domains) and a
# per-kernel mean table gpurun_out/pmc_<tag>/summary_pmc_summary.csv.  Paths are resolved from this script's directory, so
# it can be started from anywhere inside the repository copy.
set -o pipefail
tag=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/pmc_$tag
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES \
  --output-format csv -d $out/pmc_1 -- python3 "$@" > $out/run1.log 2>&1 || { tail -5 $out/run1.log; exit 1; }
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU \
  --output-format csv -d $out/pmc_2 -- python3 "$@" > $out/run2.log 2>&1 || { tail -5 $out/run2.log; exit 1; }
rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE --output-format csv -d $out/pmc_3 -- python3 "$@" > $out/run3.log 2>&1 || { tail -5 $out/run3.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_4 -- python3 "$@" > $out/run4.log 2>&1 || { tail -5 $out/run4.log; exit 1; }
python3 $root/tools/summarize_profile.py $out $out/summary
cat $out/summary_pmc_summary.csv
