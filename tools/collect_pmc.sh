#!/bin/bash
# Collect rocprofv3 PMC counters for one eager bench call (GPU box).  Counters go in their own runs
# (--kernel-trace + --pmc only).  usage: bash tools/collect_pmc.sh <outdir-under-gpurun_out>
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-pmc}
mkdir -p $OUT
cd $R
CMD="python bench.py --steps 1 --warmup 0 --no-graph --no-roofline --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/sq -o p -- $CMD > $OUT/sq.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- $CMD > $OUT/fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- $CMD > $OUT/write.log 2>&1
python tools/pmc_report.py $OUT > $OUT/report.txt 2>&1
tail -30 $OUT/report.txt
# keep only the report and small csv summaries (raw csvs are large)
find $OUT -name "*.csv" -size +8M -delete
