#!/bin/bash
# Kernel statistics + PMC passes of the ragged real-shape batch (bench.py extras.ragged_real_shape*), one gpurun call:
#   bash tools/profile_ragged.sh r04 [matched|independent|both] [nopmc]
# Outputs land in gpurun_out/ragged_<tag>/; tools/summarize_ragged.py condenses them into profiles/.
set -u
TAG=${1:-r04}
WHICH=${2:-both}
NOPMC=${3:-}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/ragged_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for SIZES in matched independent; do
  if [ "$WHICH" != "both" ] && [ "$WHICH" != "$SIZES" ]; then continue; fi
  CMD="python $ROOT/tools/dbg/ragged_run.py $SIZES"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$SIZES" -o run -- $CMD > "$OUT/stats_$SIZES.log" 2>&1
  echo "stats $SIZES rc=$?"
  [ -n "$NOPMC" ] && continue
  i=0
  for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY" \
             "FETCH_SIZE" \
             "WRITE_SIZE"; do
    i=$((i+1))
    timeout 600 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$OUT/pmc_${SIZES}_pass$i" -o p -- $CMD > "$OUT/pmc_${SIZES}_pass$i.log" 2>&1
    echo "pmc $SIZES pass $i rc=$?"
  done
done
