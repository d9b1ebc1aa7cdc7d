#!/usr/bin/env python3
"""Condense tools/profile_ragged.sh's output (gpurun_out/ragged_<tag>/) into tracked summaries under profiles/:

    python tools/summarize_ragged.py r04

  profiles/<tag>_kernel_stats_ragged_128x10000_<sizes>.csv   rocprofv3 --kernel-trace --stats table, verbatim
  profiles/<tag>_ragged_counters.json                        per kernel and per batch: launches, average duration,
        SQ_INSTS_VALU, FETCH_SIZE / WRITE_SIZE (separate --pmc passes) with the gfx950 read-side correction of
        MI355X_MICROARCH.md (x2), stamped with the library build (bench.py refuses it for another build)
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(REPO, "gpurun_out", f"ragged_{tag}")
out = os.path.join(REPO, "profiles")
short = lambda k: k.split("(")[0].replace("void ", "").replace("icpflow::", "").replace("(anonymous namespace)::", "")
result = {"correction": "gfx950 rocprofv3 tallies 128-B read requests at 64 B (MI355X_MICROARCH.md, HBM section): read side x2, "
                        "WRITE_SIZE as reported; counters are per dispatch, averaged over the profiled run's dispatches",
          "workload": "bench.py extras.ragged_real_shape*: 128 cluster pairs, n ~ logUniform(20, 10^4) padded to 10000, <= 100 ICP iterations"}
for sizes in ("matched", "independent"):
    st = glob.glob(os.path.join(src, f"stats_{sizes}", "**", "*kernel_stats.csv"), recursive=True)
    if not st:
        continue
    shutil.copy(st[0], os.path.join(out, f"{tag}_kernel_stats_ragged_128x10000_{sizes}.csv"))
    run = {}
    for ln in open(os.path.join(src, f"stats_{sizes}.log")):
        if ln.startswith("{"):
            run = json.loads(ln)
    calls = run.get("calls", 11)
    kernels = {}
    for r in csv.DictReader(open(st[0])):
        if "icpflow" not in r["Name"]:
            continue
        kernels[short(r["Name"])] = {"launches_per_batch": int(r["Calls"]) / calls, "avg_us": float(r["AverageNs"]) / 1e3,
                                     "us_per_batch": float(r["TotalDurationNs"]) / 1e3 / calls}
    for p in sorted(glob.glob(os.path.join(src, f"pmc_{sizes}_pass*", "**", "*counter_collection.csv"), recursive=True)):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        disp = collections.defaultdict(set)
        for r in csv.DictReader(open(p)):
            if "icpflow" not in r["Kernel_Name"]:
                continue
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k].add(r["Dispatch_Id"])
        for k in agg:
            e = kernels.setdefault(k, {})
            for c, v in agg[k].items():
                e[c + "_per_dispatch"] = v / len(disp[k])
    tot_valu = tot_hbm = tot_us = 0.0
    for k, e in kernels.items():
        if "FETCH_SIZE_per_dispatch" in e or "WRITE_SIZE_per_dispatch" in e:
            e["hbm_bytes_per_dispatch"] = int(round((2.0 * e.get("FETCH_SIZE_per_dispatch", 0.0) + e.get("WRITE_SIZE_per_dispatch", 0.0)) * 1024))
        n = e.get("launches_per_batch", 0.0)
        tot_valu += n * e.get("SQ_INSTS_VALU_per_dispatch", 0.0)
        tot_hbm += n * e.get("hbm_bytes_per_dispatch", 0)
        tot_us += e.get("us_per_batch", 0.0)
    result[sizes] = {"run": run, "kernels": kernels,
                     "per_batch": {"kernel_us": round(tot_us, 1), "SQ_INSTS_VALU": tot_valu, "hbm_bytes": int(tot_hbm)}}
    result["library_build"] = run.get("library_build")
json.dump(result, open(os.path.join(out, f"{tag}_ragged_counters.json"), "w"), indent=1, sort_keys=True)
for sizes in ("matched", "independent"):
    if sizes in result:
        r = result[sizes]
        icp = next((v for k, v in r["kernels"].items() if k.startswith("icp_kernel")), {})
        print(sizes, r["run"].get("ms_per_batch"), "ms/batch; kernels", r["per_batch"]["kernel_us"], "us; ICP", round(icp.get("avg_us", 0), 1), "us, VALU insts",
              icp.get("SQ_INSTS_VALU_per_dispatch"), "hbm bytes", icp.get("hbm_bytes_per_dispatch"), "| batch VALU", r["per_batch"]["SQ_INSTS_VALU"], "hbm", r["per_batch"]["hbm_bytes"])
