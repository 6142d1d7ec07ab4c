#!/bin/bash
# device time of the critic's loss + gradient for several builds of ssn_critic_rows.hip (tools/ab_one.sh), and the rows kernel's
# own duration from a kernel trace
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for lib in "$@"; do
  if [ $lib = main ]; then unset SSN_LIBDIR; else export SSN_LIBDIR=$PWD/tools/ab/$lib; fi
  python3 tools/time_critic_rows.py 2>&1 | grep ROWS
  rm -rf /tmp/rp_$lib
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$lib -- python3 tools/time_critic_rows.py > /dev/null 2>&1
  f=$(find /tmp/rp_$lib -name "*kernel_stats.csv" | head -1)
  grep -E "critic_rows_kernel|critic_pack|critic_stats|splitk_reduce|gemm_bf16_pipe_multi" $f | cut -d, -f1,2,4 | sed "s/^/   $lib: /"
done
SSN_CRITIC_ROWS=0 python3 tools/time_critic_rows.py 2>&1 | grep ROWS
