#!/bin/bash
# PMC counters of the split-operand GEMM at hot shapes (GPU box)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; OUT=$R/gpurun_out/probe_split; mkdir -p $OUT
echo "# MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs); SQ_* wave counters in quad-cycles" > $OUT/report.txt
for shape in "gemm 4096 4096 4096 0 1" "gemm 16384 512 1408 0 1" "gemm 131072 384 128 0 1" "gemm 16384 2816 512 1 1" "gemm 4096 4096 4096 0 0"; do
  tag=$(echo $shape | tr " " "_")
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/$tag -o p -- python tools/gemm_probe.py $shape > $OUT/$tag.log 2>&1
  python - "$shape" $OUT/$tag >> $OUT/report.txt <<'PY'
import csv, glob, sys
shape, d = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/p_counter_collection.csv", recursive=True)
t = glob.glob(d + "/**/p_kernel_trace.csv", recursive=True)
if not f:
    print(shape, "no counters"); sys.exit()
agg = {}
sel = lambda n: ("gemm_kernel" in n or "gemm_stream_kernel" in n or "gemm_split_kernel" in n)
for r in csv.DictReader(open(f[0])):
    if sel(r["Kernel_Name"]):
        agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
dur = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(t[0])) if sel(r["Kernel_Name"])]
m = {k: sum(v) / len(v) for k, v in agg.items()}
wc = m.get("SQ_WAVE_CYCLES", 1)
util = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8 * 1024)
print(f"{shape:30s} MfmaUtil {util:5.3f} | of wave cycles: wait_any {m.get('SQ_WAIT_ANY',0)/wc:5.3f} wait_inst {m.get('SQ_WAIT_INST_ANY',0)/wc:5.3f} (lds {m.get('SQ_WAIT_INST_LDS',0)/wc:5.3f}) active {m.get('SQ_ACTIVE_INST_ANY',0)/wc:5.3f} | lds_idx_active {m.get('SQ_LDS_IDX_ACTIVE',0):.3g} bank_conflict {m.get('SQ_LDS_BANK_CONFLICT',0):.3g} gui {m['GRBM_GUI_ACTIVE']:.3g} | dur_us {sum(dur)/len(dur)/1e3:8.1f}")
PY
done
cat $OUT/report.txt
find $OUT -name "*.csv" -delete
