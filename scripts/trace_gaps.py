import csv,glob,sys
f=glob.glob(sys.argv[1]+'/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
names=[r['Kernel_Name'].split('(')[0].split('::')[-1][:28] for r in rows]
idx=[i for i,n in enumerate(names) if n.startswith('index_fused') or n.startswith('index_touch')]  # a round's first kernel
for i in idx[10:13]:
    t0=int(rows[i]['Start_Timestamp'])
    prev_end=int(rows[i-1]['End_Timestamp']) if i>0 else None
    for j in range(i,min(i+4,len(rows))):
        s,e=int(rows[j]['Start_Timestamp']),int(rows[j]['End_Timestamp'])
        print("  %-28s start %7.1f us dur %7.1f gap_before %6.1f"%(names[j],(s-t0)/1e3,(e-s)/1e3,0 if prev_end is None else (s-prev_end)/1e3))
        prev_end=e
    print()
