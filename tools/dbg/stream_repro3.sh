#!/bin/bash
# Developer tool: bench.py --workload stream at the evidence's own length (20 timed steps, 3 warm-up), five runs per queue setting
cd "$(dirname "$0")/../.."
for Q in ${QUEUES:-unset 16 unset 16}; do for k in 1 2 3 4 5; do
  if [ $Q = unset ]; then unset GPU_MAX_HW_QUEUES; export ICPFLOW_KEEP_HW_QUEUES=1; else export GPU_MAX_HW_QUEUES=$Q; unset ICPFLOW_KEEP_HW_QUEUES; fi
  v=$(timeout 300 python bench.py --workload stream --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'])")
  echo "GPU_MAX_HW_QUEUES=$Q steps 20 run $k: $v ms/frame-pair"; done; done
