#!/bin/bash
# Developer tool: kernel statistics of the demo frame pair (MP = max_points), to gpurun_out/demo_prof_<MP>/
MP=${1:-10000}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/demo_prof_$MP
mkdir -p $OUT
MP=$MP timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- python $GRAFT_REPO_ROOT/tools/dbg/demo_frame.py > $OUT/out.txt 2>/dev/null
cd $GRAFT_REPO_ROOT
python - <<PY
import csv,glob
f=glob.glob('gpurun_out/demo_prof_$MP/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
runs=6
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("kernel time per frame pair: %.3f ms" % (tot/runs/1e6))
for r in rows[:14]:
    print(f"{r['Name'][:64]:64s} calls/frame {int(r['Calls'])/runs:5.1f}  per frame {float(r['TotalDurationNs'])/runs/1e3:8.1f} us  avg {float(r['AverageNs'])/1e3:8.1f} us")
print(open('gpurun_out/demo_prof_$MP/out.txt').read())
PY
