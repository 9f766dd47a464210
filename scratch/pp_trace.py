import csv,glob
rows=[]
for f in glob.glob('gpurun_out/lat/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)): rows.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'].replace('dcs::','').replace('void ','').replace('(anonymous namespace)::','')[:40]))
for f in glob.glob('gpurun_out/lat/*memory_copy_trace.csv'):
    for r in csv.DictReader(open(f)): rows.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),'COPY '+r.get('Direction','')[12:]))
rows.sort()
idx=[i for i,r in enumerate(rows) if 'k_dev_assemble' in r[2] and any('k_describe' in x[2] for x in rows[max(0,i-6):i])]
i0=idx[-3]
j=i0
while j>0 and 'k_track_finish' not in rows[j][2]: j-=1
t0=rows[j][0]; prev=rows[j][1]
for s,e,n in rows[j:i0+12]:
    print("%8.1f +%6.1f  %6.1f us %s"%((s-t0)/1e3,(s-prev)/1e3,(e-s)/1e3,n)); prev=max(prev,e)
