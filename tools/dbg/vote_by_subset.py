"""Developer tool: durations of the vote / sort / sweep kernels per subset of tools/dbg/stage1_parts.py in a rocprofv3 rocpd database
(every subset launches the pre-ICP chain 22 times: 11 x estimate_init_pose, 11 x hist_icp_eval)."""
import sqlite3, glob, sys
c = sqlite3.connect(glob.glob(sys.argv[1] + '/*.db')[0])
rows = c.execute("select name,start,end from kernels order by start").fetchall()
names = ["all", "largest alone", "all but largest", "all but six largest", "six largest", "second largest alone"]
for key in ("hist_vote_sorted", "hist_peaks", "sweep_scan_kernel<0>", "chunk_sort", "chunk_merge", "zsort_kernel"):
    d = [(e - s) / 1e3 for n, s, e in rows if key in n]
    per = {"sweep_scan_kernel<0>": 66, "chunk_sort": 44, "chunk_merge": 44}.get(key, 22)
    d = d[len(d) - per * 6:] if len(d) >= per * 6 else d      # (the first frame pair run comes before the subsets)
    print(key, " | ".join(f"{names[k]}: {sum(d[k * per:(k + 1) * per]) / 22:.1f} us" for k in range(6) if len(d) >= (k + 1) * per), f"({len(d)} launches)")
