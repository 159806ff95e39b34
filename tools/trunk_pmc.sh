#!/bin/bash
# lab: matrix-pipe utilisation per kernel of the conditioning trunk (PD_TRUNK_ONLY passes of tools/trunk_time.py) - a --pmc pass of its own
# (counters only with --kernel-trace).  MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs), as tools/pmc_report.py
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/trunk_pmc
mkdir -p $OUT
cd /tmp
PD_TRUNK_ONLY=1 timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -o p -- python $R/tools/trunk_time.py --samples 64 > $OUT/sq.log 2>&1
python - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + "/sq/**/p_counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:70]
    acc[(k, r.get("Grid_Size", ""))][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        cnt[(k, r.get("Grid_Size", ""))] += 1
rows = []
for key, v in acc.items():
    gui = v.get("GRBM_GUI_ACTIVE", 0.0)
    if gui <= 0:
        continue
    rows.append((gui, key, v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8 * 1024), cnt[key]))
rows.sort(reverse=True)
print("# MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs), summed over all launches of the (kernel, grid) of 7 trunk passes")
print("%-72s %12s %7s %9s" % ("kernel", "grid", "n", "MfmaUtil"))
for gui, (k, g), u, n in rows[:24]:
    print("%-72s %12s %7d %9.3f" % (k, g, n, u))
PY
find $OUT -name "*.csv" -size +1M -delete
