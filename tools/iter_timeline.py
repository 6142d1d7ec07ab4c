"""The kernels of one steady-state GAN iteration in launch order from a rocprofv3 kernel trace (cut at the trajectory-saving
forward): start offset, duration, name -- to see what the critic steps cost between the generator's forwards.
usage: tools/iter_timeline.py <rocprofv3 dir> [anchor substring]"""
import csv, glob, sys
path = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
anchor = sys.argv[2] if len(sys.argv) > 2 else 'gen_forward_duo_kernel<208, true'
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r['Start_Timestamp']))
cuts = [i for i, r in enumerate(rows) if anchor in r['Kernel_Name']]
lo, hi = cuts[-2], cuts[-1]
t0 = int(rows[lo]['Start_Timestamp'])
print('iteration span %.1f us, %d launches' % ((int(rows[hi]['Start_Timestamp']) - t0) / 1e3, hi - lo))
for r in rows[lo:hi]:
    print('%9.1f %8.1f  %s' % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3,
                             r['Kernel_Name'].replace('void ', '').split('(')[0][:80]))
