#!/bin/bash
# Developer tool: everything a round's profiles/ is made of, in ONE gpurun call:  bash tools/dbg/round_evidence.sh r05
# (kernel statistics + PMC of the headline, of the ragged batches, of config 4's shard; the bench line; fuzzers; clocks of the
# pacing pairs with the -DICPFLOW_TAIL_CLOCK build; the frame pair's host time stamps and kernel timeline).  tools/dbg/install_round.sh
# copies the results into profiles/ afterwards (in the build container).
set -u
TAG=${1:-r06}
cd $GRAFT_REPO_ROOT
O=gpurun_out/evidence_$TAG; mkdir -p $O
bash tools/profile_round.sh $TAG > $O/profile_round.log 2>&1
bash tools/profile_ragged.sh $TAG both > $O/profile_ragged.log 2>&1
python tools/summarize_ragged.py $TAG > $O/summarize_ragged.log 2>&1
bash tools/profile_workload.sh $TAG config4_shard python tools/dbg/config4_run.py 1024 > $O/profile_config4.log 2>&1
python tools/summarize_workload.py $TAG config4_shard > $O/summarize_config4.log 2>&1
mkdir -p gpurun_out/profiles_$TAG; cp profiles/${TAG}_* gpurun_out/profiles_$TAG/ 2>/dev/null
python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
python bench.py --workload stream --steps 20 --warmup 3 > $O/${TAG}_bench_stream.json 2>> $O/${TAG}_bench.err
python bench.py --workload stream --steps 20 --warmup 3 --force-collective > $O/${TAG}_bench_stream_rccl.json 2>> $O/${TAG}_bench.err
REPS=6 python tools/dbg/stream_stress.py > $O/stress_default.txt 2>&1
for f in cert_fuzz frame_fuzz score_fuzz registration_fuzz native_fuzz vote_list_fuzz; do timeout 500 python tools/dbg/$f.py > $O/$f.txt 2>&1; echo "$f rc=$?"; done
# round 6: the stream's spread per setting of GPU_MAX_HW_QUEUES, the A/Bs of the round's switches, the direction keys' fuzz
bash tools/dbg/stream_repro.sh $O/stream_repro.txt > /dev/null 2>&1
bash tools/dbg/stream_repro2.sh > /dev/null 2>&1; cp gpurun_out/r6_stream_repro2.txt $O/stream_repro2.txt
bash tools/dbg/stream_repro3.sh > $O/stream_repro3.txt 2>&1
bash tools/dbg/four_queues.sh > $O/four_queues.txt 2>&1
SHAPES=256x1024,1024x2048,600x2048,1500x1500,m128x4000,m128x10000,r128x10000,8192x2048 python tools/dbg/dir_keys_ab.py > $O/dir_keys_ab.txt 2>&1
python tools/dbg/dir_keys_diff.py > $O/dir_keys_diff.txt 2>&1
timeout 900 python tools/dbg/dir_keys_fuzz.py > $O/dir_keys_fuzz.txt 2>&1
SHAPES=1024x2048,600x2048,1500x1500,2048x1100,r900x2048 python tools/dbg/two_launch_ab.py > $O/two_launch_ab.txt 2>&1
bash tools/dbg/build_debug.sh ICPFLOW_VOTE_STATS tools/dbg/libicpflow_vs.so > /dev/null 2>&1
ICPFLOW_HIP_LIB=tools/dbg/libicpflow_vs.so python tools/dbg/vote_stats.py > $O/vote_stats.txt 2>&1
python tools/dbg/frame_stamps.py > $O/frame_stamps.txt 2>&1
python tools/dbg/overlap_direct_ab.py > $O/overlap_ab.txt 2>&1
python tools/dbg/share_ab.py > $O/share_ab.txt 2>&1
(cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/frame_trace -o run -- python $GRAFT_REPO_ROOT/tools/dbg/frame_stamps.py > /dev/null 2>&1)
bash tools/dbg/build_debug.sh ICPFLOW_TAIL_CLOCK tools/dbg/libicpflow_dbg.so > /dev/null 2>&1
export ICPFLOW_HIP_LIB=tools/dbg/libicpflow_dbg.so
python tools/dbg/tail_clock.py > $O/tail_clock.txt 2>&1
python tools/dbg/stage1_tail.py > $O/stage1_tail.txt 2>&1
SIZES=matched TOP=14 python tools/dbg/ragged_tail.py > $O/ragged_tail_matched.txt 2>&1
SIZES=independent TOP=10 python tools/dbg/ragged_tail.py > $O/ragged_tail_independent.txt 2>&1
PAIRS=126,9 SHOW=3,15 python tools/dbg/ragged_units.py > $O/ragged_units.txt 2>&1
python tools/dbg/help_timeline.py > $O/help_timeline.txt 2>&1
python tools/dbg/two_launch_stats.py > $O/two_launch_stats.txt 2>&1
unset ICPFLOW_HIP_LIB
tail -3 $O/profile_round.log; tail -3 $O/summarize_ragged.log; tail -4 $O/summarize_config4.log; head -c 400 $O/${TAG}_bench.json; echo; head -c 300 $O/${TAG}_bench_stream.json; echo
for f in stress_default tail_clock stage1_tail frame_stamps; do tail -n 2 $O/$f.txt; done
for f in cert_fuzz frame_fuzz score_fuzz registration_fuzz native_fuzz vote_list_fuzz; do tail -1 $O/$f.txt; done
