#!/bin/bash
# one PMC pass (counters in $PMC) over an arbitrary python command:  PMC="..." bash tools/profile_pmc_cmd.sh <script> [filter]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmcq
rm -rf "$OUT"; mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
PMC=${PMC:-SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVES}
timeout 600 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d "$OUT" -o q -- python $ROOT/$1 > "$OUT/log.txt" 2>&1
FILTER=${2:-icpflow} python3 - <<PY
import csv, collections, os
agg=collections.defaultdict(lambda: collections.defaultdict(float)); disp=collections.defaultdict(set)
for r in csv.DictReader(open("$OUT/q_counter_collection.csv")):
    if os.environ["FILTER"] not in r['Kernel_Name']: continue
    k=r['Kernel_Name'].split('(')[0][-40:]+" grid="+r['Grid_Size']
    agg[k][r['Counter_Name']]+=float(r['Counter_Value']); disp[k].add(r['Dispatch_Id'])
for k in agg:
    print(k, len(disp[k]), {c: round(v/len(disp[k])) for c,v in agg[k].items()})
PY
