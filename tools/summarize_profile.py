#!/usr/bin/env python3
"""Condense rocprofv3 output of tools/profile_bench.sh / profile_c3.sh / profile_pmc_c3.sh into the two small CSVs
kept under profiles/: <name>_kernel_stats.csv (copy of the --stats table) and <name>_pmc_summary.csv (mean counter
value per kernel launch).  usage: tools/summarize_profile.py gpurun_out/prof_<tag> profiles/r01/<name>"""
import collections
import csv
import glob
import os
import shutil
import sys


def short(name):
    name = name.replace('void ', '')
    return name.split('(')[0] if name.startswith('ssn::') else name[:60]


def main(src, dst):
    stats = glob.glob(os.path.join(src, 'trace', '*', '*_kernel_stats.csv'))
    if stats:
        shutil.copy(stats[0], dst + '_kernel_stats.csv')
    rows = collections.OrderedDict()
    for f in sorted(glob.glob(os.path.join(src, 'pmc_*', '*', '*_counter_collection.csv'))):
        for r in csv.DictReader(open(f)):
            if not r['Kernel_Name'].startswith('void ssn::') and 'ssn::' not in r['Kernel_Name'][:12]:
                continue
            rows.setdefault((short(r['Kernel_Name']), r['Counter_Name']), []).append(float(r['Counter_Value']))
    if rows:
        with open(dst + '_pmc_summary.csv', 'w') as out:
            out.write('kernel,counter,launches,mean_per_launch\n')
            for (k, c), v in rows.items():
                out.write('"%s",%s,%d,%.6g\n' % (k, c, len(v), sum(v) / len(v)))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
