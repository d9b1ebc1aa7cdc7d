#!/bin/bash
# Developer tool: the extras of bench.py (ragged batches, frame pair, four frame pairs in flight, four batches in one call, config 4) run after
# run per setting of GPU_MAX_HW_QUEUES -- how stable the lines that use several streams are inside ONE busy process.
cd "$(dirname "$0")/../.."
for Q in ${QUEUES:-unset 8 16}; do for k in 1 2 3 4 5; do
  if [ $Q = unset ]; then unset GPU_MAX_HW_QUEUES; export ICPFLOW_KEEP_HW_QUEUES=1; else export GPU_MAX_HW_QUEUES=$Q; unset ICPFLOW_KEEP_HW_QUEUES; fi
  timeout 300 python bench.py --cpu-pairs 0 > /tmp/b.json 2>/dev/null; echo "Q=$Q $(python tools/dbg/benchsum.py /tmp/b.json)"; done; done
