"""Developer tool: average duration per kernel name in a rocprofv3 rocpd database (argument: directory holding the .db)."""
import sqlite3, glob, sys, collections
c = sqlite3.connect(glob.glob(sys.argv[1] + '/*.db')[0])
acc = collections.defaultdict(lambda: [0, 0.0])
for n, s, e in c.execute("select name,start,end from kernels"):
    short = n.split('(')[0].replace('icpflow::', '').replace('(anonymous namespace)::', '').replace('void ', '')[:50]
    acc[short][0] += 1; acc[short][1] += (e - s) / 1e3
for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    print(f"{k:52s} calls {n:5d} avg {t / n:9.1f} us total {t / 1e3:8.2f} ms")
