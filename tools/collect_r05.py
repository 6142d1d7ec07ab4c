#!/usr/bin/env python3
"""Copy the round's profile summaries from gpurun_out/ into profiles/r05/ and regenerate profiles/hbm_traffic.json from them.

usage: tools/collect_r05.py           (after tools/profile_r05.sh traces / pmc1 / pmc2 have run through gpurun)

Every entry of hbm_traffic.json is computed here from a *_pmc_summary.csv under profiles/ -- bytes per launch =
2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024 (FETCH_SIZE / WRITE_SIZE in KB; on gfx950 FETCH_SIZE counts half of a wide streaming
read: MI355X_MICROARCH.md, HBM section) -- and stamped with the file it came from and the commit the table was made at;
bench.py copies the numbers into `roofline.traffic`."""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DST = os.path.join(ROOT, 'profiles', 'r05')
SRC = os.path.join(ROOT, 'gpurun_out')

COPIES = [   # (source under gpurun_out/, name under profiles/r05/)
    ('prof_r05/c2_kernel_stats.csv', 'c2_kernel_stats.csv'), ('prof_r05/c3_kernel_stats.csv', 'c3_gan_loop_kernel_stats.csv'),
    ('prof_r05/c3_iteration.csv', 'c3_gan_loop_iteration.csv'), ('prof_r05/c3_gaps.txt', 'c3_gan_loop_gaps.txt'),
    ('prof_r05/c3paper_kernel_stats.csv', 'c3paper_kernel_stats.csv'), ('prof_r05/c3paper_iteration.csv', 'c3paper_iteration.csv'),
    ('prof_r05/c3paper_gaps.txt', 'c3paper_gaps.txt'), ('prof_r05/c5_kernel_stats.csv', 'c5_kernel_stats.csv'),
    ('prof_r05/c2nb8_kernel_stats.csv', 'c2nb8_kernel_stats.csv'), ('prof_r05/mt_kernel_stats.csv', 'mt19937_kernel_stats.csv'),
    ('prof_r05/mt_time.log', 'mt19937_draw_times.txt'),
    ('pmc_r05_mt/summary_pmc_summary.csv', 'mt19937_pmc_summary.csv'), ('pmc_r05_fwd/summary_pmc_summary.csv', 'duo_forward_pmc_summary.csv'),
    ('pmc_r05_fwdsave/summary_pmc_summary.csv', 'duo_forward_save_pmc_summary.csv'),
    ('pmc_r05_solve/summary_pmc_summary.csv', 'duo_solver_pmc_summary.csv'), ('pmc_r05_adj/summary_pmc_summary.csv', 'backward_pmc_summary.csv'),
    ('pmc_r05_c5/summary_pmc_summary.csv', 'c5_sparse_pmc_summary.csv'), ('pmc_r05_c2/summary_pmc_summary.csv', 'c2_mixed_pmc_summary.csv'),
]

# key of hbm_traffic.json -> (summary file under profiles/, kernel name prefix, note)
ENTRIES = {
    'c2': ('c2_mixed_pmc_summary.csv', 'ssn::solve_tile_mixed_kernel', 'solve_tile_mixed_kernel per launch at C2; algorithmic 665.2 MB'),
    'c3': ('duo_forward_pmc_summary.csv', 'ssn::gen_forward_duo_kernel<208, false',
           'gen_forward_duo_kernel<208, false, ...> per launch (critic-phase forwards, no trajectory stores); algorithmic: W 164 MB + '
           'ext / outputs 26 MB; W is read once since the one-pass prologue (401 MB before it); the excess is the scratch of the prologue '
           '(148-208 B per lane, written and read once: the fp32 units of W beside the lane constants)'),
    'c3_save': ('duo_forward_save_pmc_summary.csv', 'ssn::gen_forward_duo_kernel<208, true', 'the trajectory-saving forward of the generator step'),
    'c5': ('c5_sparse_pmc_summary.csv', 'ssn::ff_forward_sparse_lattice_kernel', 'ff_forward_sparse_lattice_kernel per launch; algorithmic 4.28 GB'),
    'c2nb8': ('duo_solver_pmc_summary.csv', 'ssn::solve_duo_kernel', 'solve_duo_kernel<208> per launch at C2 with 8 stimuli; algorithmic 734 MB'),
    'backward': ('backward_pmc_summary.csv', 'ssn::gen_backward_duo_kernel', 'gen_backward_duo_kernel<208, false> per launch at the C3 shape'),
    'gw': ('backward_pmc_summary.csv', 'ssn::gw_split_kernel', 'gw_split_kernel<7, true> per launch at the C3 shape; algorithmic 2 x 7.86 GB + 164 MB'),
    'backward_fused': ('backward_pmc_summary.csv', 'ssn::gen_backward_fused_kernel', 'gen_backward_fused_kernel<208, false> per launch at the C3 shape'),
    'mt19937_gen': ('mt19937_pmc_summary.csv', 'ssn::mt::mt_gen_kernel', 'mt_gen_kernel<float, ...> per launch, mean over the three shapes of tools/time_mt.py; '
                    'algorithmic: 4 B written per double drawn + the segment states read'),
    'mt19937_jump': ('mt19937_pmc_summary.csv', 'ssn::mt::mt_jump_kernel', 'mt_jump_kernel per launch (all rounds of the three shapes of tools/time_mt.py)'),
}


def traffic(path, prefix):
    fetch = write = None
    for r in csv.DictReader(open(path)):
        if r['kernel'].startswith(prefix):
            if r['counter'] == 'FETCH_SIZE':
                fetch = float(r['mean_per_launch'])
            elif r['counter'] == 'WRITE_SIZE':
                write = float(r['mean_per_launch'])
    if fetch is None or write is None:
        return None
    return 2 * fetch * 1024 + write * 1024


def main():
    os.makedirs(DST, exist_ok=True)
    for src, dst in COPIES:
        p = os.path.join(SRC, src)
        if os.path.exists(p):
            shutil.copy(p, os.path.join(DST, dst))
        else:
            print('missing', src, file=sys.stderr)
    commit = subprocess.check_output(['git', '-C', ROOT, 'rev-parse', '--short', 'HEAD']).decode().strip()
    table = {'_made_by': 'tools/collect_r05.py at commit %s: bytes per launch = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024 from the '
                         'named pmc summary (rocprofv3 --pmc passes of tools/pmc_run.sh)' % commit}
    old = {}
    try:
        old = json.load(open(os.path.join(ROOT, 'profiles', 'hbm_traffic.json')))
    except Exception:
        pass
    for key, (name, prefix, note) in ENTRIES.items():
        src = None
        for rnd in ('r05', 'r04', 'r03', 'r02'):          # the newest round that holds this kernel's counters
            cand = os.path.join(ROOT, 'profiles', rnd, name)
            if os.path.exists(cand) and traffic(cand, prefix) is not None:
                src = cand
                break
        if src is None:
            if key in old:                                  # (kept from an earlier table, said so)
                table[key] = old[key]
                table['_note_' + key] = 'carried over from the round-4 table (no counter summary for this kernel in profiles/)'
            continue
        table[key] = traffic(src, prefix)
        table['_note_' + key] = '%s; source profiles/%s' % (note, os.path.relpath(src, os.path.join(ROOT, 'profiles')))
    json.dump(table, open(os.path.join(ROOT, 'profiles', 'hbm_traffic.json'), 'w'), indent=1)
    print(json.dumps(table, indent=1))


if __name__ == '__main__':
    main()
