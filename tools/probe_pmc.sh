#!/bin/bash
# PMC counters of single kernels at hot-path shapes (GPU box): MFMA utilisation evidence per kernel.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; OUT=$R/gpurun_out/probe_pmc; mkdir -p $OUT
echo "# single-kernel PMC probes: MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs)" > $OUT/report.txt
for shape in "attn 256 4 256" "attn 64 4 2048" "attn 64 16 256" "gemm 16384 1536 512" "gemm 16384 2816 512 1" "gemm 131072 384 128" "gemm 65536 512 128" "gemm 4096 4096 4096"; do
  tag=$(echo $shape | tr " " "_")
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/$tag -o p -- python tools/gemm_probe.py $shape > /dev/null 2>&1
  python - "$shape" $OUT/$tag >> $OUT/report.txt <<'PY'
import csv, glob, sys
shape, d = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/p_counter_collection.csv", recursive=True)
t = glob.glob(d + "/**/p_kernel_trace.csv", recursive=True)
agg = {}
for r in csv.DictReader(open(f[0])):
    if "gemm_kernel" in r["Kernel_Name"] or "gemm_stream_kernel" in r["Kernel_Name"] or "attn_kernel" in r["Kernel_Name"]:
        agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
dur = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(t[0])) if "gemm_kernel" in r["Kernel_Name"] or "gemm_stream_kernel" in r["Kernel_Name"] or "attn_kernel" in r["Kernel_Name"]]
m = {k: sum(v) / len(v) for k, v in agg.items()}
util = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8 * 1024)
print(f"{shape:28s} MfmaUtil {util:6.3f}  wait_any/wave_cycles {m['SQ_WAIT_ANY']/m['SQ_WAVE_CYCLES']:6.3f}  lds_bank_conflict {m['SQ_LDS_BANK_CONFLICT']:10.0f}  avg_dur_us {sum(dur)/len(dur)/1e3:9.1f}")
PY
done
cat $OUT/report.txt
find $OUT -name "*.csv" -delete
