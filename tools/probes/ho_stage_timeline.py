import csv,sys,glob
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows=[r for r in rows if r['Kernel_Name'].startswith('void xd::k_ho') or 'k_ho' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# find a late k_ho_step_fast stage-0 launch: take the sequence starting at the 3rd-from-last group
starts=[i for i,r in enumerate(rows) if 'k_ho_step_fast' in r['Kernel_Name']]
nst=int(sys.argv[2])
i0=starts[-nst*5]
t0=int(rows[i0]['Start_Timestamp'])
for r in rows[i0:i0+4*nst+6]:
    n=r['Kernel_Name'].split('(')[0].replace('void xd::','')[:40]
    print('%-42s q%-3s start %8.1f us  dur %7.1f us  grid %s'%(n,r.get('Queue_Id','?'),(int(r['Start_Timestamp'])-t0)/1e3,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,r.get('Grid_Size_X', r.get('Grid_Size','?'))))
