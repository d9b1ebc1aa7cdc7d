#!/bin/bash
# Developer tool: kernel statistics of BASELINE config 4's per-GPU shard (1024 pairs x 2048 points) -> gpurun_out/c4shard/
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/c4shard
mkdir -p $OUT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- python $GRAFT_REPO_ROOT/bench.py --workload config4 --pairs 1024 --steps 6 --warmup 2 --no-extras --cpu-pairs 0 > $OUT/bench.json 2>$OUT/err.txt
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob,json
f=glob.glob('gpurun_out/c4shard/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:9]:
    print(f"{r['Name'][:60]:60s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us")
try:
    d=json.loads(open('gpurun_out/c4shard/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['workload'][:60])
except Exception as e: print(e, open('gpurun_out/c4shard/err.txt').read()[-500:])
PY
