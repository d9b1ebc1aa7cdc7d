#!/bin/bash
# Collect PMC counters for the bench command in separate rocprofv3 passes (one --pmc set per
# run, --kernel-trace only: the pool refuses --pmc together with other trace domains).
# Usage (on the GPU box, via gpurun):  bash tools/profile_pmc.sh <tag>
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 3 --warmup 1 --cpu-pairs 0 --no-extras"
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY" \
           "FETCH_SIZE" \
           "WRITE_SIZE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$OUT/pass$i" -o p -- $CMD > "$OUT/pass$i.log" 2>&1
  echo "pass $i ($SET): rc=$?"
done
ls -R "$OUT" | head -40
