#!/bin/bash
# Build a second copy of libssnode.so with extra compiler flags into tools/ab/<tag>/ for A/B timing on ONE box:
#   tools/ab_build.sh noasm -DSSN_PK_ASM=0     then   SSN_LIBDIR=tools/ab/noasm python bench.py ...
set -e
tag=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/tools/ab/$tag
mkdir -p $out/obj
cd $root/tc_gan_amd/csrc
for f in *.hip; do
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result "$@" -c $f -o $out/obj/${f%.hip}.o &
  if (( $(jobs -r | wc -l) >= 4 )); then wait -n; fi
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC $out/obj/*.o -o $out/libssnode.so
rm -rf $out/obj
echo built $out/libssnode.so
