"""Kernel timeline of the last mbavo_lm_batch call in a rocprofv3 --kernel-trace run: python tools/lm_timeline.py <output dir> [max kernels]"""
import csv, sys, glob
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last call: walk back from the last k_lm_solve over the kernels that follow each other within 150 us
last = max(i for i, r in enumerate(rows) if 'k_lm_solve' in r['Kernel_Name'])
i0 = last
while i0 > 0 and int(rows[i0]['Start_Timestamp']) - int(rows[i0 - 1]['End_Timestamp']) < 150000 and 'copyBuffer' not in rows[i0 - 1]['Kernel_Name']:
    i0 -= 1
t0 = int(rows[i0]['Start_Timestamp'])
prev_end = None
n = int(sys.argv[2]) if len(sys.argv) > 2 else 70
for r in rows[max(i0 - 3, 0):i0 + n]:
    s = int(r['Start_Timestamp']) - t0; e = int(r['End_Timestamp']) - t0
    nm = r['Kernel_Name'].split('(')[0][-40:]
    print("%9.2f %9.2f dur %7.2f gap %6.2f  %s grid %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3 if prev_end is not None else 0, nm, r.get('Grid_Size_X', '')))
    prev_end = e
