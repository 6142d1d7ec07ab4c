"""Timeline of one critic loss + gradient from a rocprofv3 kernel trace of tools/time_critic_rows.py: kernels of the LAST call
(from its memset to its splitk_reduce) with start offsets, durations and streams; and the span of each of the last calls."""
import csv, glob, sys
path = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r['Start_Timestamp']))
ends = [i for i, r in enumerate(rows) if 'splitk_reduce' in r['Kernel_Name']]
def first_of(e):                   # the call's first kernel: the one behind the previous call's last (torch's own kernels skipped)
    prev = max([i for i in ends if i < e], default=-1)
    s = prev + 1
    while 'at::native' in rows[s]['Kernel_Name']:
        s += 1
    return s
spans = []
for e in ends:
    s = first_of(e)
    spans.append((int(rows[e]['End_Timestamp']) - int(rows[s]['Start_Timestamp'])) / 1e3)
print('spans of the last calls (us):', ' '.join('%.0f' % x for x in spans[-8:]))
e = ends[-1]; s = first_of(e); t0 = int(rows[s]['Start_Timestamp'])
for r in rows[s:e + 1]:
    print('%8.1f %7.1f  q%-3s %s' % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3,
                                  r.get('Queue_Id', '?'), r['Kernel_Name'][:90]))
