#!/usr/bin/env python3
"""Condense rocprofv3 output under gpurun_out/ into small, tracked summaries under profiles/.

    python tools/summarize_profiles.py <round-tag> <kernel-stats-dir> <pmc-dir> [pairs points]

Writes profiles/<tag>_kernel_stats.csv (verbatim rocprofv3 --stats table),
profiles/<tag>_pmc_summary.json (per-kernel per-dispatch counter averages) and
profiles/<tag>_icp_kernel_traffic.json (HBM bytes per launch of the dominant kernel, with
the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md applied and stated).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag, stats_dir, pmc_dir = sys.argv[1:4]
pairs, points = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (256, 1024)
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(REPO, "profiles")
os.makedirs(out, exist_ok=True)

for f in glob.glob(os.path.join(stats_dir, "*kernel_stats.csv")):
    shutil.copy(f, os.path.join(out, f"{tag}_kernel_stats.csv"))

summary = {}
for p in sorted(glob.glob(os.path.join(pmc_dir, "pass*", "*counter_collection.csv"))):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"]
        if "icpflow" not in k:
            continue
        k = k.split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    for k in agg:
        e = summary.setdefault(k, {"dispatches": len(disp[k])})
        for c, v in agg[k].items():
            e[c + "_per_dispatch"] = v / len(disp[k])
json.dump(summary, open(os.path.join(out, f"{tag}_pmc_summary.json"), "w"), indent=1, sort_keys=True)

dom = next((k for k in summary if "icp_kernel" in k), None)
if dom:
    e = summary[dom]
    fetch_kb = e.get("FETCH_SIZE_per_dispatch", 0.0)
    write_kb = e.get("WRITE_SIZE_per_dispatch", 0.0)
    traffic = {
        "kernel": dom,
        "FETCH_SIZE_KiB_per_launch_raw": fetch_kb,
        "WRITE_SIZE_KiB_per_launch_raw": write_kb,
        "correction": "gfx950 rocprofv3 reports 1/2 of the bytes of wide (16 B/lane) coalesced reads "
                      "(FETCH_SIZE = TCC_EA0_RDREQ x 64 B with 128-B requests tallied at 64 B): read side x2; "
                      "WRITE_SIZE taken as is (the per-iteration history records and tallies; the kernel has no vector-register "
                      "spills and no scratch -- tools/kernel_resources.py -- and spilled SGPRs live in VGPR lanes, not in memory)",
        "hbm_bytes_per_launch": int(round((2.0 * fetch_kb + write_kb) * 1024)),
        "launch_shape": "all_iterations" if e["dispatches"] <= 16 else "one_iteration",
        "dispatches_profiled": e["dispatches"],
        "note": "average over all launches of the profiled run, including the few launches after the "
                "batch-global stop that return immediately",
    }
    json.dump(traffic, open(os.path.join(out, f"{tag}_icp_kernel_traffic.json"), "w"), indent=1)
    # what bench.py reads for roofline.traffic / roofline.valu: stamped with the hash of the library that was
    # profiled (icpflow_build_info()); bench.py refuses the file when its own library differs
    sys.path.insert(0, REPO)
    import importlib.util
    spec = importlib.util.spec_from_file_location("icpflow_build", os.path.join(REPO, "icp_flow_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    counters = dict(traffic)
    counters.update({"library_build": b.built_hash(), "pairs": pairs, "points": points,
                     "SQ_INSTS_VALU_per_launch": e.get("SQ_INSTS_VALU_per_dispatch"),
                     "SQ_INSTS_LDS_per_launch": e.get("SQ_INSTS_LDS_per_dispatch"),
                     "SQ_LDS_BANK_CONFLICT_per_launch": e.get("SQ_LDS_BANK_CONFLICT_per_dispatch"),
                     "SQ_ACTIVE_INST_LDS_per_launch": e.get("SQ_ACTIVE_INST_LDS_per_dispatch"),
                     "SQ_WAVE_CYCLES_per_launch": e.get("SQ_WAVE_CYCLES_per_dispatch"),
                     "SQ_BUSY_CYCLES_per_launch": e.get("SQ_BUSY_CYCLES_per_dispatch")})
    json.dump(counters, open(os.path.join(out, f"{tag}_icp_kernel_counters.json"), "w"), indent=1)
print(open(os.path.join(out, f"{tag}_icp_kernel_traffic.json")).read() if dom else "no dominant kernel found")
