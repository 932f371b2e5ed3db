"""Kernel timeline of the last mbavo_lm_batch call in a rocprofv3 --kernel-trace run: python tools/lm_timeline.py <output dir>"""
import csv,sys,glob
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# find last k_lm_init and print the 60 kernels after it
idx=[i for i,r in enumerate(rows) if 'k_lm_init' in r['Kernel_Name']]
i0=idx[-1]
t0=int(rows[i0]['Start_Timestamp'])
prev_end=None
for r in rows[i0-6:i0+46]:
    s=int(r['Start_Timestamp'])-t0; e=int(r['End_Timestamp'])-t0
    nm=r['Kernel_Name'].split('(')[0][-40:]
    print("%9.2f %9.2f dur %7.2f gap %6.2f  %s grid %s"%(s/1e3,e/1e3,(e-s)/1e3,(s-prev_end)/1e3 if prev_end is not None else 0,nm,r.get('Grid_Size_X','')))
    prev_end=e
