#!/usr/bin/env python3
"""Print a rocprofv3 *_kernel_stats.csv as a short table:  python tools/kstats.py <csv> [rows]"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:n]:
    print(f'{r["Name"][:72]:72s} calls={r["Calls"]:>5s} total_ms={float(r["TotalDurationNs"]) / 1e6:8.3f} '
          f'avg_us={float(r["AverageNs"]) / 1e3:9.1f} {r["Percentage"]:>6s}%')
print(f"total kernel time: {tot / 1e6:.3f} ms")
