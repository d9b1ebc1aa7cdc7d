#!/bin/bash
# one PMC pass for a quick look at the instruction mix of the bench kernels
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmcq
mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc ${1:-SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY} --output-format csv -d "$OUT" -o q -- python $ROOT/bench.py --steps 3 --warmup 1 --cpu-pairs 0 --no-extras > "$OUT/log.txt" 2>&1
python3 - <<PY
import csv, collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); disp=collections.defaultdict(set)
for r in csv.DictReader(open("$OUT/q_counter_collection.csv")):
    k=r['Kernel_Name'].split('(')[0][-50:]
    if 'icpflow' not in r['Kernel_Name']: continue
    agg[k][r['Counter_Name']]+=float(r['Counter_Value']); disp[k].add(r['Dispatch_Id'])
for k in agg:
    print(k, len(disp[k]), {c: round(v/len(disp[k])) for c,v in agg[k].items()})
PY
