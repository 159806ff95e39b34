#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel launches, total, average.
usage: python tools/prof_summary.py gpurun_out/prof/r01_results.db > profiles/r01_kernel_stats.txt"""
import re
import sqlite3
import sys

db = sys.argv[1]
c = sqlite3.connect(db)
rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
                 "max(vgpr_count), max(accum_vgpr_count), max(lds_size) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"# rocprofv3 --kernel-trace --stats summary of {db}")
print(f"# total kernel time {tot/1e6:.2f} ms over {sum(r[1] for r in rows)} dispatches")
print(f"{'kernel':92s} {'calls':>7s} {'total_ms':>10s} {'pct':>6s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'vgpr':>5s} {'agpr':>5s} {'lds':>7s}")
for r in rows:
    nm = re.sub(r"\(anonymous namespace\)::", "", r[0])
    nm = re.sub(r"\(.*\)$", "", nm)[:92]
    print(f"{nm:92s} {r[1]:7d} {r[2]/1e6:10.2f} {100*r[2]/tot:6.2f} {r[3]/1e3:10.1f} {r[4]/1e3:9.1f} {r[5]/1e3:9.1f} {r[6]:5d} {r[7]:5d} {r[8]:7d}")
