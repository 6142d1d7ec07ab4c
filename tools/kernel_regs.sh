#!/bin/bash
# Compile the tile kernels to gfx950 assembly under /tmp/isa and print LDS / VGPR / spill counts per kernel
# (all kernels with spills, plus every kernel built for 3 waves/SIMD).  usage: tools/kernel_regs.sh [file ...]
set -e
cd "$(dirname "$0")/../tc_gan_amd/csrc"
mkdir -p /tmp/isa
files=${@:-ssn_gen ssn_tile}
for f in $files; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only $f.hip -o /tmp/isa/$f.s 2>&1 | grep -v "argument unused" || true
done
python3 - $files <<'PY'
import re, sys
for f in sys.argv[1:]:
    t = open('/tmp/isa/%s.s' % f).read()
    pat = r"\.group_segment_fixed_size:\s+(\d+).*?\.name:\s+(\S+)\n(.*?)\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)"
    for m in re.finditer(pat, t, re.S):
        name = m.group(2).replace('_ZN3ssn', '').split('EEEvNS')[0]
        if name.endswith('ELi3') or int(m.group(5)):
            print(name, 'lds', m.group(1), 'vgpr', m.group(4), 'spill', m.group(5))
PY
