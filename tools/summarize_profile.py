#!/usr/bin/env python3
"""Condense a tools/profile.sh output directory into a small JSON + text summary (for profiles/)."""
import collections
import csv
import glob
import json
import os
import sys

d = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "pt_trace_items"
out = {"dir": os.path.basename(d.rstrip("/")), "kernel": kern, "counters_mean_per_launch": {}}
for f in glob.glob(os.path.join(d, "trace", "*kernel_stats.csv")):
    out["kernel_stats"] = [r for r in csv.DictReader(open(f))][:6]
for f in glob.glob(os.path.join(d, "pmc_*", "*counter_collection.csv")):
    agg = collections.defaultdict(list)
    extra = {}
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            extra = {k: r[k] for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Grid_Size", "Workgroup_Size") if k in r}
    for k, v in agg.items():
        out["counters_mean_per_launch"][k] = sum(v) / len(v)
    if extra:
        out["dispatch"] = extra
c = out["counters_mean_per_launch"]
der = {}
if "SQ_BUSY_CYCLES" in c and "SQ_INSTS_VALU" in c:
    # SQ_BUSY_CYCLES is summed over the 32 shader engines; wave64 VALU op = 2 cycles on a SIMD32; 1024 SIMDs
    cyc = c["SQ_BUSY_CYCLES"] / 32.0
    der["busy_cycles_per_SE"] = cyc
    der["valu_issue_utilisation"] = c["SQ_INSTS_VALU"] * 2.0 / (cyc * 1024)
if "SQ_THREAD_CYCLES_VALU" in c and "SQ_ACTIVE_INST_VALU" in c:
    der["valu_thread_utilisation"] = c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64)
if "SQ_WAVE_CYCLES" in c and "SQ_BUSY_CYCLES" in c:
    der["avg_waves_per_SIMD"] = c["SQ_WAVE_CYCLES"] * 4 / (c["SQ_BUSY_CYCLES"] / 32.0 * 1024)
for k in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY"):
    if k in c and "SQ_WAVE_CYCLES" in c:
        der[k + "_frac_of_wave_cycles"] = c[k] / c["SQ_WAVE_CYCLES"]
if "TCC_HIT_sum" in c:
    der["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
if "FETCH_SIZE" in c:
    # MI355X_MICROARCH.md (HBM / rocprofv3 section): on gfx950 FETCH_SIZE under-reports WIDE, COALESCED streaming reads by 2x.
    # A traversal kernel's reads are 16-byte gathers out of 64-byte nodes and 48-byte leaf records, for which that factor is
    # not established: the raw figure is the lower bound, raw x 2 the upper, and both are printed.  argv[3] = "streaming"
    # marks a profile whose reads ARE wide streams (pt_accumulate); only then is the corrected figure the one to quote.
    streaming = len(sys.argv) > 3 and sys.argv[3] == "streaming"
    der["hbm_read_bytes_per_launch_raw(FETCH_SIZE KB x1024)"] = c["FETCH_SIZE"] * 1024
    der["hbm_read_bytes_per_launch_x2(gfx950 correction for wide coalesced reads)"] = c["FETCH_SIZE"] * 1024 * 2
    der["hbm_read_bytes_per_launch"] = c["FETCH_SIZE"] * 1024 * (2 if streaming else 1)
    der["hbm_read_note"] = "wide coalesced reads: x2 applied" if streaming else "scattered gathers: raw figure quoted, the x2 correction is an upper bound"
if "WRITE_SIZE" in c:
    der["hbm_write_bytes_per_launch(WRITE_SIZE KB x1024)"] = c["WRITE_SIZE"] * 1024
out["derived"] = der
print(json.dumps(out, indent=1))
