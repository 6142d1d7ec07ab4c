#!/bin/bash
# A/B on ONE box, builds alternating: the one-pass W prologue of the two-draw kernels against round 4's two passes.
#   main           = tc_gan_amd/ext (SSN_DUO_ONEPASS=1, solver two-pass)
#   tools/ab/twopass = -DSSN_DUO_ONEPASS=0        tools/ab/solve1 = -DSSN_DUO_ONEPASS_SOLVE=1
set -e
cd "$(dirname "$0")/.."
for rep in 1 2 3; do
  for lib in main twopass solve1; do
    if [ $lib = main ]; then unset SSN_LIBDIR; else export SSN_LIBDIR=$PWD/tools/ab/$lib; fi
    echo "== rep $rep lib $lib"
    python tools/time_fwd.py 8
    python tools/time_fwd.py 8 --save
    python tools/time_adj.py | head -1
    python tools/time_solver.py 8
  done
done
