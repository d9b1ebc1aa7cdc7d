#!/bin/bash
cd /root/repo
OUT=gpurun_out/r6_stream_repro2.txt; : > $OUT
for Q in 16 32; do
  for k in 1 2 3 4 5 6; do
    export GPU_MAX_HW_QUEUES=$Q
    v=$(timeout 300 python bench.py --workload stream --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'])")
    echo "GPU_MAX_HW_QUEUES=$Q steps 10 run $k: $v ms/frame-pair" | tee -a $OUT
  done
done
