import csv,glob,collections,sys
f=glob.glob(sys.argv[1]+'/**/*_kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
g=collections.defaultdict(list)
for r in rows:
    n=r['Kernel_Name'].replace('void ','').split('(')[0]
    key=(n[:50],r['Grid_Size_X'],r['Grid_Size_Y'],r['Grid_Size_Z'])
    g[key].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
tot=sum(sum(v) for v in g.values())
for k,v in sorted(g.items(), key=lambda kv:-sum(kv[1]))[:25]:
    print('%-52s %5s %4s %3s  n=%4d avg %6.1f us  share %4.1f%%'%(k+(len(v),sum(v)/len(v)/1e3,100*sum(v)/tot)))
