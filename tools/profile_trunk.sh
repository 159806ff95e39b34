#!/bin/bash
# lab: per-kernel time of the conditioning trunk alone (7 passes under rocprofv3 --kernel-trace --stats)
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/trunk
mkdir -p $OUT
cd /tmp
PD_TRUNK_ONLY=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o p -- python $R/tools/trunk_time.py --samples 1 > $OUT/trace.log 2>&1
python - $OUT <<'PY'
import csv, glob, sys
out = sys.argv[1]
rows = list(csv.DictReader(open(glob.glob(out + "/trace/**/p_kernel_stats.csv", recursive=True)[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time per pass {tot / 7e6:.2f} ms, dispatches per pass {sum(int(r['Calls']) for r in rows) / 7:.0f}")
for r in rows[:45]:
    print("%-96s %6.0f x %8.1f us = %6.2f ms  %5.1f %%" % (r["Name"].replace("(anonymous namespace)::", "")[:96], int(r["Calls"]) / 7, float(r["AverageNs"]) / 1e3,
                                                       float(r["TotalDurationNs"]) / 7e6, float(r["Percentage"])))
PY
find $OUT -name "*.csv" -size +1M -delete
