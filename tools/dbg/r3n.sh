set -u
mkdir -p gpurun_out/r3n
bash tools/profile_round.sh r03 > gpurun_out/r3n/profile_round.log 2>&1
bash tools/dbg/prof_config4_shard.sh > gpurun_out/r3n/c4shard.log 2>&1
ICPFLOW_HIP_LIB=tools/dbg/libicpflow_tail.so timeout 600 python tools/dbg/tail_clock_big.py > gpurun_out/r3n/tail_big.log 2>&1
ICPFLOW_HIP_LIB=tools/dbg/libicpflow_tail.so ICPFLOW_NO_HELPERS=1 timeout 600 python tools/dbg/tail_clock_big.py > gpurun_out/r3n/tail_big_nohelp.log 2>&1
timeout 600 python tools/dbg/stream_bench_repeat.py > gpurun_out/r3n/stream_repeat.log 2>&1
timeout 300 python tools/dbg/stream_time.py > gpurun_out/r3n/stream_time.log 2>&1
tail -3 gpurun_out/r3n/*.log
