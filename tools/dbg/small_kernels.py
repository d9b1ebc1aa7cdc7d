"""Developer tool: durations of the small kernels in a rocprofv3 --stats csv (argument: directory)."""
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0])))
keys = ("team_plan", "count_pair", "select_kernel", "eval_epilogue", "score_pick", "gather_seg", "chunk_", "peaks", "transform", "table_", "flow_")
for r in rows:
    n = r["Name"].split("(")[0].replace("icpflow::", "")[-42:]
    if any(k in n for k in keys):
        print("%-44s calls %4s avg %7.1f min %7.1f max %7.1f us" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
