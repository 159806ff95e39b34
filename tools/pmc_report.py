#!/usr/bin/env python
"""Aggregate rocprofv3 --pmc CSVs (sq / fetch / write passes, each collected with --kernel-trace) per kernel symbol AND
launch shape (grid dimensions): atom / token / triangle / MSA attention launches of one symbol get their own rows, as do the
GEMM shapes, so that MfmaUtil, HBM traffic and duration can be set against the algorithmic work of THAT shape.

MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs) (busy matrix-pipe cycles per SIMD cycle at
the clock the kernel actually ran at); HBM bytes follow MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE x 1024 B, FETCH_SIZE reported
raw AND with the gfx950 read-side x2 correction (it under-reports wide coalesced reads by exactly 2x).

usage: python tools/pmc_report.py <dir with sq/ fetch/ write/> [--by-symbol] [--label-cfg1-b64]"""
import collections
import csv
import glob
import json
import re
import sys

root = sys.argv[1]
BY_SYMBOL = "--by-symbol" in sys.argv
#: bench.py --launch-log: the (symbol, shape) sequence of the GEMM / attention launches of the profiled call.  Persistent GEMM
#: kernels launch the same grid for every problem size, so their shapes are recovered by ORDER: the n-th dispatch of a symbol
#: in the trace is the n-th log entry of that symbol (used only when the counts agree)
LOG = None
if "--launch-log" in sys.argv:
    LOG = collections.defaultdict(list)
    for name, shape in json.load(open(sys.argv[sys.argv.index("--launch-log") + 1])):
        LOG[name].append(shape)
NXCD = 8      # GRBM_GUI_ACTIVE is reported summed over the 8 XCDs (checked: cycles/8 / duration = 1.7-2.1 GHz)


def clean(name):
    k = re.sub(r"\(anonymous namespace\)::", "", name)
    return re.sub(r"\(.*\)$", "", k).replace("void ", "")


def trace(sub):
    """Dispatch_Id -> (grid xyz in WORKGROUPS, duration ns) from the pass's own kernel trace"""
    f = glob.glob(f"{root}/{sub}/**/p_kernel_trace.csv", recursive=True)
    out = {}
    if f:
        for r in csv.DictReader(open(f[0])):
            wg = [max(int(r[f"Workgroup_Size_{a}"]), 1) for a in "XYZ"]
            grid = tuple(int(r[f"Grid_Size_{a}"]) // w for a, w in zip("XYZ", wg))
            out[r["Dispatch_Id"]] = (grid, float(r["End_Timestamp"]) - float(r["Start_Timestamp"]), wg[0] * wg[1] * wg[2])
    return out


def load(sub):
    f = glob.glob(f"{root}/{sub}/**/p_counter_collection.csv", recursive=True)
    if not f:
        return {}
    tr = trace(sub)
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(dict)
    rows_ = list(csv.DictReader(open(f[0])))
    shape_of = {}
    if LOG is not None:
        per_sym = collections.defaultdict(set)
        for r in rows_:
            per_sym[clean(r["Kernel_Name"])].add(int(r["Dispatch_Id"]))
        for sym, ids in per_sym.items():
            if sym in LOG and len(LOG[sym]) == len(ids):
                for d, shp in zip(sorted(ids), LOG[sym]):
                    shape_of[str(d)] = shp
    for r in rows_:
        grid, dur, wg = tr.get(r["Dispatch_Id"], ((int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1), 1, 1),
                                                  float(r["End_Timestamp"]) - float(r["Start_Timestamp"]), int(r["Workgroup_Size"])))
        if r["Dispatch_Id"] in shape_of:
            grid = shape_of[r["Dispatch_Id"]]
        key = (clean(r["Kernel_Name"]),) if BY_SYMBOL else (clean(r["Kernel_Name"]), grid, wg)
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[key][r["Dispatch_Id"]] = dur
    return {k: dict(v, _n=len(disp[k]), _dur=sum(disp[k].values()) / len(disp[k])) for k, v in agg.items()}


def label(key):
    """what a (symbol, grid) pair is in the benchmark call (cfg1: T 256 / A 2048 / S 128, medium model, B samples)"""
    if BY_SYMBOL or len(key) < 2:
        return ""
    if isinstance(key[1], str):
        return key[1]                      # shape from bench.py's launch log
    name, (gx, gy, gz), wg = key
    if name.startswith("attn_split_kernel") or name.startswith("attn_kernel"):
        # grid = (batch, query blocks, heads)
        if gz == 4 and gy >= 2:
            return f"atom attention: {gx} x {gz} heads x {gy} query blocks"
        if gz == 16:
            return f"token DiT attention: {gx} samples x 16 heads"
        if gz == 4 and gy == 1 and gx >= 128:
            return f"triangle / MSA-row attention: batch {gx} x 4 heads"
        return f"attention: batch {gx} x {gz} heads x {gy} query blocks"
    return f"grid {gx}x{gy}x{gz}"


sq, fe, wr = load("sq"), load("fetch"), load("write")
rows = []
for k, v in sq.items():
    n = v["_n"]
    gui = v.get("GRBM_GUI_ACTIVE", 0.0)
    mfma = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    util = mfma / (gui / NXCD * 1024) if gui else 0.0
    f_kb = fe.get(k, {}).get("FETCH_SIZE", 0.0) / max(fe.get(k, {}).get("_n", 1), 1)
    w_kb = wr.get(k, {}).get("WRITE_SIZE", 0.0) / max(wr.get(k, {}).get("_n", 1), 1)
    wave = max(v.get("SQ_WAVE_CYCLES", 1), 1)
    rows.append(dict(kernel=k[0], grid=(k[1] if isinstance(k[1], str) else list(k[1])) if len(k) > 1 else None, workgroup=k[2] if len(k) > 2 else None, what=label(k),
                     launches=n, gui_cycles_per_launch=gui / n, mfma_util=util,
                     wait_any_frac=v.get("SQ_WAIT_ANY", 0) / wave, wait_inst_frac=v.get("SQ_WAIT_INST_ANY", 0) / wave,
                     active_inst_frac=v.get("SQ_ACTIVE_INST_ANY", 0) / wave,
                     lds_bank_conflict=v.get("SQ_LDS_BANK_CONFLICT", 0) / n,
                     fetch_bytes_raw=f_kb * 1024, fetch_bytes_x2=2 * f_kb * 1024, write_bytes=w_kb * 1024,
                     total_gui=gui, dur_us=v["_dur"] / 1e3,
                     eff_clock_ghz=(gui / n / NXCD) / v["_dur"] if v["_dur"] else 0.0))
rows.sort(key=lambda r: -r["total_gui"])
tot = sum(r["total_gui"] for r in rows) or 1.0
print(f"# PMC summary of {root} (one eager sample_diffusion call of bench.py), grouped by kernel symbol"
      + ("" if BY_SYMBOL else " and launch grid (workgroups x, y, z)"))
print("# MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs); fetch = FETCH_SIZE*1024 B, x2 = gfx950 wide-read "
      "correction; pct = share of GRBM_GUI_ACTIVE")
print(f"{'kernel':58s} {'what':50s} {'n':>5s} {'pct':>5s} {'MfmaUtil':>8s} {'wait_any':>8s} {'fetchMB':>8s} {'(x2)':>8s} {'writeMB':>8s} "
      f"{'dur_us':>8s} {'GHz':>5s}")
for r in rows[:40]:
    print(f"{r['kernel'][:58]:58s} {r['what'][:50]:50s} {r['launches']:5d} {100 * r['total_gui'] / tot:5.1f} {r['mfma_util']:8.3f} "
          f"{r['wait_any_frac']:8.3f} {r['fetch_bytes_raw'] / 1e6:8.2f} {r['fetch_bytes_x2'] / 1e6:8.2f} {r['write_bytes'] / 1e6:8.2f} "
          f"{r['dur_us']:8.1f} {r['eff_clock_ghz']:5.2f}")
json.dump(rows, open(f"{root}/pmc_summary{'_by_symbol' if BY_SYMBOL else ''}.json", "w"), indent=1)
