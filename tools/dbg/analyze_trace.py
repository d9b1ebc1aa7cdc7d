import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last 60% of the trace (the timed stream)
n = len(rows); rows = rows[int(n * 0.35):]
t0 = int(rows[0]['Start_Timestamp']); t1 = max(int(r['End_Timestamp']) for r in rows)
ev = []
for r in rows: ev.append((int(r['Start_Timestamp']), 1)); ev.append((int(r['End_Timestamp']), -1))
ev.sort()
busy = 0; depth = 0; last = t0; conc = collections.Counter()
for t, d in ev:
    if depth > 0: busy += t - last
    conc[min(depth, 4)] += t - last
    depth += d; last = t
print('span %.2f ms, GPU busy (any kernel) %.2f ms (%.0f%%), time by concurrent kernels:' % ((t1 - t0) / 1e6, busy / 1e6, 100 * busy / (t1 - t0)), {k: round(v / 1e6, 2) for k, v in sorted(conc.items())})
acc = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('icpflow::', '')[:48]
    acc[k][0] += 1; acc[k][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
for k, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:12]:
    print('  %-50s calls %4d avg %8.1f us total %7.2f ms' % (k, c, t / c, t / 1e3))
if len(sys.argv) > 2 and sys.argv[2] == "timeline":
    # the kernels of the LAST frame pair (K=1: one at a time), in order: start offset, duration, gap since the previous kernel's end
    per = len(rows) // max(1, int(sys.argv[3]) if len(sys.argv) > 3 else 16)
    last_rows = rows[-per:]
    base = int(last_rows[0]['Start_Timestamp']); prev_end = base
    for r in last_rows:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        name = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('icpflow::', '')[:60]
        print('  +%8.1f us  dur %7.1f  gap %7.1f  %s  grid %s wg %s' % ((s - base) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, name,
              r.get('Grid_Size', '?'), r.get('Workgroup_Size', '?')))
        prev_end = max(prev_end, e)
    print('  frame span %.1f us over %d kernels' % ((prev_end - base) / 1e3, len(last_rows)))
