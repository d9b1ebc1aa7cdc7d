#!/usr/bin/env python3
"""Condense tools/profile_workload.sh's output (gpurun_out/wl_<tag>_<name>/) into tracked summaries under profiles/:

    python tools/summarize_workload.py r05 config4_shard

  profiles/<tag>_kernel_stats_<name>.csv   rocprofv3 --kernel-trace --stats table, verbatim
  profiles/<tag>_<name>_counters.json      per kernel: launches per step, average duration, SQ_* counters, FETCH_SIZE / WRITE_SIZE
        (separate --pmc passes) with the gfx950 read-side correction of MI355X_MICROARCH.md (x2), executed VALU fraction
        (SQ_INSTS_VALU x 64 lanes / duration / 78.6 T lane-op/s); stamped with the library build
"""
import collections, csv, glob, json, os, shutil, sys
tag, name = sys.argv[1], sys.argv[2]
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(REPO, "gpurun_out", f"wl_{tag}_{name}")
out = os.path.join(REPO, "profiles")
VALU_PEAK = 78.6e12
short = lambda k: k.split("(")[0].replace("void ", "").replace("icpflow::", "").replace("(anonymous namespace)::", "")
st = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
assert st, f"no kernel stats under {src}"
shutil.copy(st[0], os.path.join(out, f"{tag}_kernel_stats_{name}.csv"))
run = {}
for ln in open(os.path.join(src, "stats.log")):
    if ln.startswith("{"):
        run = json.loads(ln)
calls = run.get("calls", 1)
kernels = {}
for r in csv.DictReader(open(st[0])):
    if "icpflow" not in r["Name"]:
        continue
    kernels[short(r["Name"])] = {"launches_per_step": int(r["Calls"]) / calls, "avg_us": float(r["AverageNs"]) / 1e3,
                                 "us_per_step": float(r["TotalDurationNs"]) / 1e3 / calls}
for p in sorted(glob.glob(os.path.join(src, "pmc_pass*", "**", "*counter_collection.csv"), recursive=True)):
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
tot_us = 0.0
for k, e in kernels.items():
    if "FETCH_SIZE_per_dispatch" in e or "WRITE_SIZE_per_dispatch" in e:
        e["hbm_bytes_per_dispatch"] = int(round((2.0 * e.get("FETCH_SIZE_per_dispatch", 0.0) + e.get("WRITE_SIZE_per_dispatch", 0.0)) * 1024))
        if e.get("avg_us"):
            e["hbm_GBps"] = round(e["hbm_bytes_per_dispatch"] / (e["avg_us"] * 1e-6) / 1e9, 1)
    if e.get("SQ_INSTS_VALU_per_dispatch") and e.get("avg_us"):
        e["executed_valu_frac"] = round(e["SQ_INSTS_VALU_per_dispatch"] * 64.0 / (e["avg_us"] * 1e-6) / VALU_PEAK, 4)
    if e.get("SQ_LDS_IDX_ACTIVE_per_dispatch"):
        e["lds_bank_conflict_frac"] = round(e.get("SQ_LDS_BANK_CONFLICT_per_dispatch", 0.0) / e["SQ_LDS_IDX_ACTIVE_per_dispatch"], 4)
    if e.get("SQ_WAVE_CYCLES_per_dispatch"):
        e["wait_frac"] = round(e.get("SQ_WAIT_ANY_per_dispatch", 0.0) / e["SQ_WAVE_CYCLES_per_dispatch"], 4)
    tot_us += e.get("us_per_step", 0.0)
result = {"correction": "gfx950 rocprofv3 tallies 128-B read requests at 64 B (MI355X_MICROARCH.md, HBM section): read side x2, WRITE_SIZE as "
                        "reported; counters are per dispatch, averaged over the profiled run's dispatches; executed_valu_frac = SQ_INSTS_VALU x 64 / "
                        "duration / 78.6e12 lane-op/s",
          "run": run, "library_build": run.get("library_build"), "kernels": kernels, "kernel_us_per_step": round(tot_us, 1)}
json.dump(result, open(os.path.join(out, f"{tag}_{name}_counters.json"), "w"), indent=1, sort_keys=True)
for k, e in sorted(kernels.items(), key=lambda kv: -kv[1].get("us_per_step", 0)):
    print(f"{k[:60]:60s} {e.get('us_per_step', 0):9.1f} us/step  valu {e.get('executed_valu_frac')}  hbm {e.get('hbm_GBps')} GB/s  wait {e.get('wait_frac')}  lds-conflict {e.get('lds_bank_conflict_frac')}")
