import csv,sys,glob
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
t0=min(int(r['Start_Timestamp']) for r in rows)
rows.sort(key=lambda r:int(r['Start_Timestamp']))
for r in rows[-int(sys.argv[2]):]:
    s=int(r['Start_Timestamp']);e=int(r['End_Timestamp'])
    print("%-40s q%-3s start %10.3f dur %9.3f ms grid %s" % (r['Kernel_Name'][:40], r.get('Queue_Id','?'), (s-t0)/1e6,(e-s)/1e6, r.get('Grid_Size_X', r.get('Grid_Size','?'))))
