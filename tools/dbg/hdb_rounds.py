"""Print per-round kernel durations of one icpflow_hdbscan_mst call from a rocprofv3 kernel trace csv."""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
for name in ("hdb_core_kernel", "hdb_scan_kernel", "hdb_reduce_weight", "hdb_select", "hdb_flatten"):
    d = sorted((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
               for r in rows if name in r["Kernel_Name"])
    per = len(d) // 14 if len(d) >= 14 else len(d)
    print(name, [round(x[1]) for x in d[-per:]][:10])
