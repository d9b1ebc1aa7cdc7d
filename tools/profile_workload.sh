#!/bin/bash
# Kernel statistics + PMC passes (separate runs, --kernel-trace only beside --pmc) of ONE command, in one gpurun call:
#   bash tools/profile_workload.sh r05 config4_shard python tools/dbg/config4_run.py 1024
# Outputs land in gpurun_out/wl_<tag>_<name>/; tools/summarize_workload.py <tag> <name> condenses them into profiles/.
set -u
TAG=$1; NAME=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/wl_${TAG}_${NAME}
mkdir -p "$OUT"
CMD=""
for w in "$@"; do case "$w" in tools/*|bench.py) CMD="$CMD $ROOT/$w";; *) CMD="$CMD $w";; esac; done
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- $CMD > "$OUT/stats.log" 2>&1
echo "stats rc=$?"
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY" \
           "FETCH_SIZE" \
           "WRITE_SIZE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 900 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$OUT/pmc_pass$i" -o p -- $CMD > "$OUT/pmc_pass$i.log" 2>&1
  echo "pmc pass $i rc=$?"
done
