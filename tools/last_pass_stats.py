"""Per-kernel statistics of the LAST pass of a repeated workload from a rocprofv3 --kernel-trace CSV: every dispatch from the last
occurrence of a marker kernel (one that runs once, first, in each pass) on.  Unlike `--stats` divided by the pass count this leaves
out what only the first pass does (weight packing, cache fills).  usage: python tools/last_pass_stats.py <dir> <marker> [rows]"""
import csv, glob, sys, collections

d, marker = sys.argv[1], sys.argv[2]
nrows = int(sys.argv[3]) if len(sys.argv) > 3 else 45
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
if not f:
    sys.exit("no kernel trace under " + d)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
if not idx:
    sys.exit("marker kernel not found: " + marker)
last = rows[idx[-1]:]
acc = collections.OrderedDict()
for r in last:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "")
    a = acc.setdefault(n, [0, 0.0])
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
tot = sum(v[1] for v in acc.values())
span = (int(last[-1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])) * 1e-6
aten = sum(v[0] for k, v in acc.items() if "at::native" in k)
copies = sum(v[0] for k, v in acc.items() if "copyBuffer" in k or "fillBuffer" in k)
print(f"last pass ({len(idx)} passes seen): {len(last)} dispatches, kernel time {tot / 1e3:.2f} ms, first start to last end {span:.2f} ms; "
      f"ATen kernels {aten}, copy / fill buffers {copies}")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1])[:nrows]:
    print("%-110s %5d x %8.1f us = %6.2f ms  %5.1f %%" % (k[:110], v[0], v[1] / v[0], v[1] / 1e3, 100 * v[1] / tot))
