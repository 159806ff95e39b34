#!/bin/bash
# lab: where a one-sample call spends its time - rocprofv3 kernel stats of `bench.py --samples 1` (1 eager call + 5 graph replays
# = 6 executions of every kernel of the call), per kernel: launches per call, average duration, ms per call
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
NS=${1:-1}
OUT=$R/gpurun_out/b$NS
mkdir -p $OUT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o p -- python $R/bench.py --samples $NS --steps 5 --warmup 1 --no-roofline --no-cpu-baseline --no-extra > $OUT/trace.log 2>&1
python - $OUT <<'PY'
import csv, glob, sys
out = sys.argv[1]
rows = list(csv.DictReader(open(glob.glob(out + "/trace/**/p_kernel_stats.csv", recursive=True)[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 6e6
n = sum(int(r["Calls"]) for r in rows) / 6
print(f"kernel time per call {tot:.2f} ms over {n:.0f} dispatches")
for r in rows[:48]:
    print("%-100s %7.1f x %7.1f us = %6.2f ms" % (r["Name"].replace("(anonymous namespace)::", "")[:100], int(r["Calls"]) / 6, float(r["AverageNs"]) / 1e3,
                                                  float(r["TotalDurationNs"]) / 6e6))
PY
find $OUT -name "*.csv" -size +1M -delete
