#!/usr/bin/env python3
"""Condenses rocprofv3 CSV output (gpurun_out/prof_<tag>_*) into the small summaries kept under profiles/.
Usage: python tools/summarize_prof.py <tag> <out_prefix>      e.g.  r01_pt_fast profiles/r01b_pt_fast"""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def short(name):
    m = re.search(r"(pathtrace_kernel<[^>]*>|pathtrace_pool_kernel<[^>]*>|mandelbrot_kernel<.*?, \d+>|convert_rgba8_kernel|deinterleave_rows_kernel)", name)
    return m.group(1) if m else None


def main():
    tag, out = sys.argv[1], sys.argv[2]
    stats = glob.glob(f"gpurun_out/prof_{tag}_stats/*/*_kernel_stats.csv")
    if stats:
        rows = [r for r in csv.DictReader(open(stats[0])) if short(r["Name"])]
        with open(out + "_kernel_stats.csv", "w") as f:
            w = csv.writer(f)
            w.writerow(["Kernel", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "StdDev"])
            for r in rows:
                w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["StdDev"]])
    summary = {}
    for kind in ("valu", "hbm", "mix", "busy"):
        files = glob.glob(f"gpurun_out/prof_{tag}_pmc_{kind}/*/*_counter_collection.csv")
        if not files:
            continue
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        meta = {}
        for r in csv.DictReader(open(files[0])):
            k = short(r["Kernel_Name"])
            if not k:
                continue
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r.get("End_Timestamp"):   # the launch's duration in the SAME pass: its clock
                agg[k]["_duration_ns_with_GRBM"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
            meta[k] = {"grid": r["Grid_Size"], "workgroup": r["Workgroup_Size"], "lds_bytes": r["LDS_Block_Size"],
                       "arch_vgpr": r["VGPR_Count"], "sgpr": r["SGPR_Count"]}
        for k, v in agg.items():
            e = summary.setdefault(k, {"dispatch": meta[k], "per_launch_mean": {}})
            for c, x in v.items():
                e["per_launch_mean"][c] = sum(x) / len(x)
    for k, e in summary.items():
        c = e["per_launch_mean"]
        d = {}
        if "SQ_INSTS_VALU" in c and "SQ_WAVES" in c:
            d["valu_insts_per_wave"] = c["SQ_INSTS_VALU"] / c["SQ_WAVES"]
        if "SQ_THREAD_CYCLES_VALU" in c and "SQ_ACTIVE_INST_VALU" in c:
            d["active_lanes_per_valu_inst"] = c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"]
        if "GRBM_GUI_ACTIVE" in c:
            d["gpu_cycles_per_xcd"] = c["GRBM_GUI_ACTIVE"] / 8.0
            if c.get("_duration_ns_with_GRBM"):
                # the shader clock the kernel ITSELF held (cycles per XCD / duration of the same launches) — not the clock a probe
                # kernel reads under its own load afterwards (profiles/r04_k3_clock.txt)
                d["kernel_clock_ghz"] = c["GRBM_GUI_ACTIVE"] / 8.0 / c["_duration_ns_with_GRBM"]
        if "WRITE_SIZE" in c:
            d["hbm_write_bytes"] = c["WRITE_SIZE"] * 1024.0    # WRITE_SIZE is in KB; exact for 16-B stores (MI355X_MICROARCH.md §HBM)
        if "SQ_INSTS_VALU_ADD_F32" in c:
            d["valu_other_insts"] = c["SQ_INSTS_VALU"] - sum(c.get(k, 0.0) for k in (
                "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_INT32"))
        if "SQ_INSTS_VALU" in c and "GRBM_GUI_ACTIVE" in c:
            # MEASURED issue rate: shader cycles of one SIMD per VALU wave-instruction it issued (1024 SIMDs).  Round 1 priced
            # the instruction classes with isolated microbenchmark costs instead ("modelled_valu_issue_utilisation") and
            # got > 1: in a real mix the classes overlap.  The floor these kernels reach is ~2.45 cycles per instruction.
            d["simd_cycles_per_valu_inst"] = (c["GRBM_GUI_ACTIVE"] / 8.0) / (c["SQ_INSTS_VALU"] / 1024.0)
        e["derived"] = d
    # Which build these counters were taken on: mc_build_id() of the library the profiled bench.py loaded (run this script in the same
    # gpurun call as the profile).  bench.py quotes a summary only when the id of its kernel family equals the loaded library's.
    import __graft_entry__ as entry
    B = entry.load_package().bindings
    summary["_build"] = dict(B.build_id(), library=os.path.relpath(B.LIB_PATH, ROOT))
    json.dump(summary, open(out + "_pmc_summary.json", "w"), indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
