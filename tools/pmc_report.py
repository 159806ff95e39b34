#!/usr/bin/env python
"""Aggregate rocprofv3 --pmc CSVs (sq / fetch / write passes) per kernel symbol.
MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 1024 SIMDs); HBM bytes follow
MI355X_MICROARCH.md: FETCH_SIZE/WRITE_SIZE are in KiB-like units of 1024 B... reported raw AND with the gfx950
read-side x2 correction (FETCH_SIZE under-reports wide coalesced reads by exactly 2x)."""
import csv, glob, json, re, sys, collections

root = sys.argv[1]

def load(sub):
    f = glob.glob(f"{root}/{sub}/**/p_counter_collection.csv", recursive=True)
    if not f:
        return {}
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(set)
    for r in csv.DictReader(open(f[0])):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        k = re.sub(r"\(.*\)$", "", k)
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k].add(r["Dispatch_Id"])
    return {k: dict(v, _n=len(cnt[k])) for k, v in agg.items()}

def durations(sub):
    f = glob.glob(f"{root}/{sub}/**/p_kernel_trace.csv", recursive=True)
    d = collections.defaultdict(list)
    if f:
        for r in csv.DictReader(open(f[0])):
            k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
            k = re.sub(r"\(.*\)$", "", k)
            d[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return {k: sum(v) / len(v) for k, v in d.items()}


sq, fe, wr = load("sq"), load("fetch"), load("write")
dur = durations("sq")
NXCD = 8      # GRBM_GUI_ACTIVE is reported summed over the 8 XCDs (checked: cycles/8 / duration = 1.7-2.1 GHz)
rows = []
for k, v in sq.items():
    n = v["_n"]
    gui = v.get("GRBM_GUI_ACTIVE", 0.0)
    mfma = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    util = mfma / (gui / NXCD * 1024) if gui else 0.0
    f_kb = fe.get(k, {}).get("FETCH_SIZE", 0.0) / max(fe.get(k, {}).get("_n", 1), 1)
    w_kb = wr.get(k, {}).get("WRITE_SIZE", 0.0) / max(wr.get(k, {}).get("_n", 1), 1)
    rows.append(dict(kernel=k, launches=n, gui_cycles_per_launch=gui / n, mfma_util=util,
                     wait_any_frac=v.get("SQ_WAIT_ANY", 0) / max(v.get("SQ_WAVE_CYCLES", 1), 1),
                     lds_bank_conflict=v.get("SQ_LDS_BANK_CONFLICT", 0) / n,
                     fetch_bytes_raw=f_kb * 1024, fetch_bytes_x2=2 * f_kb * 1024, write_bytes=w_kb * 1024,
                     total_gui=gui, dur_us=dur.get(k, 0.0) / 1e3,
                     eff_clock_ghz=(gui / n / NXCD) / dur[k] if dur.get(k) else 0.0))
rows.sort(key=lambda r: -r["total_gui"])
print(f"# PMC summary of {root} (one eager sample_diffusion call, cfg1 B=64 40 steps)")
print("# MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs); fetch = FETCH_SIZE*1024 B, x2 = gfx950 wide-read correction")
print(f"{'kernel':70s} {'n':>6s} {'MfmaUtil':>9s} {'wait_any':>9s} {'fetchMB':>9s} {'(x2)':>9s} {'writeMB':>9s} {'dur_us':>9s} {'GHz':>5s}")
for r in rows[:14]:
    print(f"{r['kernel'][:70]:70s} {r['launches']:6d} {r['mfma_util']:9.3f} {r['wait_any_frac']:9.3f} {r['fetch_bytes_raw']/1e6:9.2f} "
          f"{r['fetch_bytes_x2']/1e6:9.2f} {r['write_bytes']/1e6:9.2f} {r['dur_us']:9.1f} {r['eff_clock_ghz']:5.2f}")
json.dump(rows, open(f"{root}/pmc_summary.json", "w"), indent=1)
