#!/usr/bin/env python3
"""How much of the C2 launch is the partial last round?  Times solve_tile_kernel at batch sizes around multiples of
768 (256 CUs x 3 resident workgroups).  usage (GPU box): python tools/tail_probe.py"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = '''
import sys, json, io, contextlib
sys.path.insert(0, %r)
import bench
bench.WORKLOADS['c2'] = (100, int(sys.argv[1]), 1, 2000, 'probe')
sys.argv = ['bench.py', '--steps', '6', '--warmup', '2', '--no-cpu-baseline']
bench.main()
''' % ROOT
for B in (768, 1536, 3072, 3840, 4096, 4608):
    out = subprocess.run([sys.executable, '-c', code, str(B)], capture_output=True, text=True).stdout.strip().splitlines()[-1]
    d = json.loads(out)
    k = d['roofline']['kernel_ms']
    print('B = %4d  (%.2f rounds)  kernel %.3f ms  = %.3f ms per full round equivalent' % (B, B / 768., k, k / (B / 768.)))
