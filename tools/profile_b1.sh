#!/bin/bash
# lab: where a one-sample call spends its time - rocprofv3 kernel trace of `bench.py --samples 1` (graph replays), per kernel
# totals and the time NOT covered by kernels inside the replayed step loop
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/b1
mkdir -p $OUT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o p -- python $R/bench.py --samples 1 --steps 3 --warmup 1 --no-roofline --no-cpu-baseline --no-extra > $OUT/trace.log 2>&1
tail -2 $OUT/trace.log | cut -c1-300
python - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + "/trace/**/p_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last call = the last third of the dispatches (4 calls: 1 warm-up + 3 timed, the first is eager + capture)
n = len(rows)
per_call = n // 4
last = rows[-per_call:]
t0, t1 = int(last[0]["Start_Timestamp"]), int(last[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last)
print(f"dispatches per call ~{per_call}; last call: span {(t1 - t0) / 1e6:.2f} ms, kernel time {busy / 1e6:.2f} ms, gaps {(t1 - t0 - busy) / 1e6:.2f} ms")
agg = collections.defaultdict(lambda: [0, 0])
for r in last:
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:90]
    agg[k][0] += 1; agg[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"  {k:90s} {c:6d} x {t / c / 1e3:7.1f} us = {t / 1e6:7.2f} ms")
gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(last[:-1], last[1:])]
gaps.sort()
print(f"gap between consecutive kernels: median {gaps[len(gaps) // 2] / 1e3:.2f} us, mean {sum(gaps) / len(gaps) / 1e3:.2f} us, p90 {gaps[int(.9 * len(gaps))] / 1e3:.2f} us")
PY
find $OUT -name "*.csv" -size +1M -delete
