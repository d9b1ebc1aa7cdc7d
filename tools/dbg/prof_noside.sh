cd /tmp && export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/noside
ICPFLOW_NO_SIDE_STREAM=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/noside -o run -- python $GRAFT_REPO_ROOT/bench.py --no-extras --cpu-pairs 0 --steps 20 > $GRAFT_REPO_ROOT/gpurun_out/noside/bench.json 2>/dev/null
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob,json
f=glob.glob('gpurun_out/noside/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(f"{r['Name'][:60]:60s} avg {float(r['AverageNs'])/1e3:8.1f} us")
print(open('gpurun_out/noside/bench.json').read()[:200])
PY
