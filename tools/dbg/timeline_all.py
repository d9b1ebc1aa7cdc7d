"""Developer tool: every kernel of the LAST frame pair in a rocprofv3 rocpd database (start, duration, gap before it)."""
import sqlite3, glob, sys
c = sqlite3.connect(glob.glob(sys.argv[1] + '/*.db')[0])
rows = c.execute("select name,start,end,stream_id from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if 'table_key' in r[0]]
fr = rows[idx[-2]:]
last = max(i for i, r in enumerate(fr) if 'flow_rigid' in r[0])
fr = fr[:last + 1]
t0 = fr[0][1]; last_end = t0
for n, s, e, st in fr:
    short = n.split('(')[0].replace('icpflow::', '').replace('(anonymous namespace)::', '').replace('void ', '')[:60]
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  gap {(s - last_end) / 1e3:7.1f}  s{st} {short}")
    last_end = max(last_end, e)
