#!/bin/bash
# Developer tool: is the stream metric reproducible?  Five consecutive runs of `bench.py --workload stream` and of
# four batches in one call per setting of GPU_MAX_HW_QUEUES (HIP's pool of hardware queues: 4 by default; the streams of a process share them).
cd "$(dirname "$0")/../.."
OUT=${1:-gpurun_out/r6_stream_repro.txt}
: > $OUT
for Q in unset 4 8 16; do
  for k in 1 2 3 4 5; do
    if [ $Q = unset ]; then unset GPU_MAX_HW_QUEUES; export ICPFLOW_KEEP_HW_QUEUES=1; else export GPU_MAX_HW_QUEUES=$Q; fi
    v=$(timeout 300 python bench.py --workload stream --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'])")
    echo "GPU_MAX_HW_QUEUES=$Q stream run $k: $v ms/frame-pair" | tee -a $OUT
  done
  for k in 1 2 3; do timeout 300 python tools/dbg/many_repro.py 2>/dev/null | tee -a $OUT; done
done
