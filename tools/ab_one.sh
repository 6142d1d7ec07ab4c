#!/bin/bash
# Like ab_build.sh, but recompiles ONE translation unit with the extra flags and links it with the objects of the normal
# build (make -C tc_gan_amd/csrc first):   tools/ab_one.sh nomma ssn_gw -DGW_ABLATE=1
set -e
tag=$1; unit=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/tools/ab/$tag
mkdir -p $out
cd $root/tc_gan_amd/csrc
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result "$@" -c $unit.hip -o $out/$unit.o
objs=$(ls *.o | grep -v "^$unit.o$")
hipcc --offload-arch=gfx950 -shared -fPIC $objs $out/$unit.o -o $out/libssnode.so
rm -f $out/$unit.o
echo built $out/libssnode.so
