#!/bin/bash
# rocprofv3 kernel trace of the trackFrame bench: per-kernel totals for the whole run (2 x 9 frames: warm-up + timed)
export TMPDIR=/tmp
OUT=gpurun_out/prof_vo
rm -rf "$OUT"; mkdir -p "$OUT"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o vo -- python tools/vo_bench.py > "$OUT/out.txt" 2>&1
tail -1 "$OUT/out.txt" | cut -c1-300
f=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.3f ms in %d launches" % (tot / 1e6, sum(int(r["Calls"]) for r in rows)))
for r in rows[:16]:
    print("%-60s calls %6s  total %8.1f us  avg %7.2f us" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3))
PY
