#!/bin/bash
# kernel-trace stats of the bench at a small sample count (default 1): what bounds the latency-bound regime
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
B=${1:-1}
OUT=$R/gpurun_out/small_b$B
mkdir -p $OUT
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o p -- python $R/bench.py --samples $B --steps 3 --warmup 1 --no-roofline --no-cpu-baseline --no-extra > $OUT/trace.log 2>&1
python - $OUT <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + "/trace/**/p_kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(out + "/kernel_stats.txt", "w") as o:
    o.write("# total kernel time %.1f ms over %d dispatches (4 calls: 1 warm-up + 3 timed)\n" % (tot / 1e6, sum(int(r["Calls"]) for r in rows)))
    for r in rows[:45]:
        o.write("%-110s %7s %9.2f %6.2f %9.1f\n" % (r["Name"].replace("(anonymous namespace)::", "")[:110], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"]), float(r["AverageNs"]) / 1e3))
PY
cat $OUT/kernel_stats.txt
find $OUT -name "*.csv" -size +1M -delete
