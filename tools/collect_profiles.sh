#!/bin/bash
# Round profile bundle (GPU box): rocprofv3 kernel-trace stats of the default bench call + PMC passes (separate runs, counters
# only with --kernel-trace) -> gpurun_out/<tag>/ ; copy the summaries into profiles/ afterwards.   usage: bash tools/collect_profiles.sh r02
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python bench.py --steps 3 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err
# the traces below are about the step loop and the trunk: the first-call check of the fp16-format bounds (three eager denoiser passes with
# ATen reductions, engine.check_dit_bounds) is switched off for them - it runs once per (weights, schedule) and never inside a timed call
export PD_BOUND_CHECK=0
CMD="python $R/bench.py --steps 1 --warmup 0 --no-graph --no-roofline --no-cpu-baseline --no-extra --launch-log $OUT/launch_log.json"
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o p -- python $R/bench.py --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --no-extra > $OUT/trace.log 2>&1
python - $OUT <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + "/trace/**/p_kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(out + "/kernel_stats.txt", "w") as o:
    o.write("# rocprofv3 --kernel-trace --stats of `python bench.py --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --no-extra`\n")
    o.write("# (cfg1, B=64, 40 steps, physics; 1 eager capture pass + graph replays; total kernel time %.1f ms over %d dispatches)\n" % (tot / 1e6, sum(int(r["Calls"]) for r in rows)))
    o.write("%-100s %8s %10s %6s %10s %9s %9s\n" % ("kernel", "calls", "total_ms", "pct", "avg_us", "min_us", "max_us"))
    for r in rows[:40]:
        o.write("%-100s %8s %10.2f %6.2f %10.1f %9.1f %9.1f\n" % (r["Name"].replace("(anonymous namespace)::", "")[:100], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                                 float(r["Percentage"]), float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
cd /tmp
timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/sq -o p -- $CMD > $OUT/sq.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- $CMD > $OUT/fetch.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- $CMD > $OUT/write.log 2>&1
cd $R
python tools/pmc_report.py $OUT --by-symbol > $OUT/pmc_report.txt 2>&1
python tools/pmc_report.py $OUT --launch-log $OUT/launch_log.json > $OUT/pmc_by_shape.txt 2>&1
head -24 $OUT/kernel_stats.txt; head -30 $OUT/pmc_by_shape.txt | cut -c1-200
# steady-state per-kernel table of ONE 64-sample call (the last of three graph replays; marker = a kernel that runs once per call, first)
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/call_trace -o p -- python $R/bench.py --steps 3 --warmup 1 --no-roofline --no-cpu-baseline --no-extra > /dev/null 2>&1
cd $R
python tools/last_pass_stats.py $OUT/call_trace atom_pair_init_kernel 40 > $OUT/call_b64_steady_state.txt 2>&1
find $OUT -name "*.csv" -size +1M -delete
