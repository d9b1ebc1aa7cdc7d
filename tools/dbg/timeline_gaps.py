"""Developer tool: kernel timeline of the LAST frame pair in a rocprofv3 rocpd database: kernels, gaps, busy time."""
import sqlite3, glob, sys
c = sqlite3.connect(glob.glob(sys.argv[1] + '/*.db')[0])
rows = c.execute("select name,start,end,stream_id from kernels order by start").fetchall()
names = [r[0] for r in rows]
idx = [i for i, n in enumerate(names) if 'table_key' in n]
f0 = idx[-2]
fr = rows[f0:]
# cut at the last flow_rigid kernel
last = max(i for i, r in enumerate(fr) if 'flow_rigid' in r[0])
fr = fr[:last + 1]
t0 = fr[0][1]; last_end = t0; busy = 0
for n, s, e, st in fr:
    short = n.split('(')[0].replace('icpflow::', '').replace('(anonymous namespace)::', '').replace('void ', '')[:44]
    gap = (s - last_end) / 1e3
    if gap > 8 or (e - s) > 40e3:
        print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f}  gap {gap:8.1f}  s{st} {short}")
    busy += max(0, e - max(s, last_end)); last_end = max(last_end, e)
print(f"span {(last_end - t0) / 1e3:.1f} us, GPU busy {busy / 1e3:.1f} us, idle {(last_end - t0 - busy) / 1e3:.1f} us, kernels {len(fr)}")
