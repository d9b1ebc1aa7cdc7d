#!/bin/bash
# Developer tool: average kernel durations of the ragged batch (independent sizes) for several builds:
#   bash tools/dbg/kernel_avg_libs.sh "pattern" lib1.so lib2.so ...     ('-' = the product library)
PAT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  rm -rf /tmp/kavg; 
  if [ "$L" = "-" ]; then unset ICPFLOW_HIP_LIB; else export ICPFLOW_HIP_LIB=$ROOT/$L; fi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kavg -o run -- python $ROOT/tools/dbg/ragged_run.py ${SIZES:-independent} > /tmp/kavg.log 2>&1
  echo "== $L"
  python3 - "$PAT" <<PY
import csv,glob,sys,re
f=glob.glob("/tmp/kavg/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if re.search(sys.argv[1], r["Name"]): print("   ",r["Name"][:64].replace("void icpflow::",""), r["Calls"], round(float(r["AverageNs"])/1e3,1))
PY
done
