#!/bin/bash
# Developer tool: kernel timeline of the last step of a bench.py run -> stdout
#   bash tools/dbg/step_timeline.sh [bench.py arguments]      (default: the headline config)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/step_timeline
rm -rf $OUT; mkdir -p $OUT
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $OUT -o run -- python $GRAFT_REPO_ROOT/bench.py --no-extras --cpu-pairs 0 --steps 6 --warmup 2 "$@" > $OUT/bench.json 2>$OUT/err.txt
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/step_timeline/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'zsort_kernel' in r['Kernel_Name'] or 'chunk_sort' in r['Kernel_Name']]
start = idx[-1]
t0 = int(rows[start]['Start_Timestamp'])
for r in rows[start:]:
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f}  {r['Kernel_Name'][:72]} g={r.get('Grid_Size_X')}")
PY
