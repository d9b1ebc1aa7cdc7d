mkdir -p gpurun_out/r3o
for opt in "--steps 50 --warmup 10" "--steps 5 --warmup 2" "--steps 5 --warmup 2" "--steps 20 --warmup 3"; do
echo "== $opt"
ICPFLOW_BENCH_DEBUG=1 timeout 280 python bench.py --cpu-pairs 0 $opt 2>&1 >/dev/null | grep debug
done > gpurun_out/r3o/bench_variants2.log 2>&1
cat gpurun_out/r3o/bench_variants2.log
