set -u
mkdir -p gpurun_out/r3n
bash tools/profile_round.sh r03 > gpurun_out/r3n/profile_round.log 2>&1
bash tools/dbg/prof_config4_shard.sh > gpurun_out/r3n/c4shard.log 2>&1
timeout 300 python tools/dbg/stream_time.py > gpurun_out/r3n/stream_time.log 2>&1
timeout 300 python tools/dbg/fused_ab.py > gpurun_out/r3n/fused_ab.log 2>&1
tail -n 3 gpurun_out/r3n/*.log
