#!/usr/bin/env python3
"""Where the wall time of a profiled loop goes that no kernel accounts for.
usage: tools/trace_gaps.py <rocprofv3 output dir> [anchor kernel substring] [out.csv] [iteration]
Reads *_kernel_trace.csv, cuts the launch sequence into iterations at every launch of the anchor kernel (default: the
trajectory-saving generator forward, one per GAN iteration), and prints for the LAST full iteration (or the one that starts at
the anchor's launch number `iteration`, counted from 0: a command that runs several loops): wall span, summed kernel
time, idle time, and the largest idle gaps with the kernels on either side.  With out.csv: per-kernel totals of that
iteration (launches, summed us) -- a steady-state table, unlike --stats, which includes warm-up and set-up launches."""
import collections
import csv
import glob
import os
import sys


def short(name):
    name = name.replace('void ', '')
    name = name.split('(')[0]
    return name[:90]


def main(src, anchor='gen_forward_duo_kernel<208, true', out=None, which=None):
    files = glob.glob(os.path.join(src, '**', '*_kernel_trace.csv'), recursive=True)
    if not files:
        sys.exit('no kernel trace under ' + src)
    rows = []
    for r in csv.DictReader(open(files[0])):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
    rows.sort()
    cuts = [i for i, r in enumerate(rows) if anchor in r[2]]
    if len(cuts) < 3:
        sys.exit('anchor %r seen %d times' % (anchor, len(cuts)))
    lo, hi = (cuts[-2], cuts[-1]) if which is None else (cuts[int(which)], cuts[int(which) + 1])   # one full steady-state iteration
    it = rows[lo:hi]
    span = rows[hi][0] - it[0][0]
    busy = 0
    cur_end = it[0][0]
    gaps = []
    for k, (s, e, n) in enumerate(it):
        if s > cur_end:
            gaps.append((s - cur_end, short(it[k - 1][2]) if k else '-', short(n)))
        busy += max(0, e - max(s, cur_end))
        cur_end = max(cur_end, e)
    if rows[hi][0] > cur_end:
        gaps.append((rows[hi][0] - cur_end, short(it[-1][2]), short(rows[hi][2])))
    print('iteration: %d launches, span %.3f ms, kernels busy %.3f ms, idle %.3f ms (%.1f %%)' %
          (len(it), span / 1e6, busy / 1e6, (span - busy) / 1e6, 100.0 * (span - busy) / span))
    gaps.sort(reverse=True)
    for g, a, b in gaps[:12]:
        print('  %8.1f us   after %-60s before %s' % (g / 1e3, a[:60], b[:60]))
    print('  gaps: %d, of which > 20 us: %d (%.3f ms)' % (len(gaps), sum(1 for g in gaps if g[0] > 20000),
                                                        sum(g[0] for g in gaps if g[0] > 20000) / 1e6))
    tot = collections.OrderedDict()
    for s, e, n in it:
        t = tot.setdefault(short(n), [0, 0])
        t[0] += 1
        t[1] += e - s
    table = sorted(tot.items(), key=lambda kv: -kv[1][1])
    for n, (c, d) in table[:14]:
        print('  %5d x %-80s %9.1f us' % (c, n[:80], d / 1e3))
    if out:
        with open(out, 'w') as f:
            f.write('kernel,launches_per_iteration,total_us_per_iteration\n')
            for n, (c, d) in table:
                f.write('"%s",%d,%.1f\n' % (n, c, d / 1e3))
            f.write('"(idle between kernels)",%d,%.1f\n' % (len(gaps), (span - busy) / 1e3))
            f.write('"(iteration span)",%d,%.1f\n' % (len(it), span / 1e3))


if __name__ == '__main__':
    main(*sys.argv[1:])
