#!/bin/bash
# Developer tool: latency breakdown of the demo frame pair (host phases, cProfile, kernel timeline)
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r3s; mkdir -p $O
for mp in 10000 2048; do MP=$mp timeout 300 python tools/dbg/frame_latency.py > $O/latency_$mp.log 2>&1; done
MP=10000 SORT=tottime ROWS=40 timeout 300 python tools/dbg/profile_host.py > $O/host_tottime.log 2>&1
MP=10000 SORT=cumulative ROWS=45 timeout 300 python tools/dbg/profile_host.py > $O/host_cum.log 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_tl; MP=10000 timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_tl -o demo -- python $GRAFT_REPO_ROOT/tools/dbg/demo_frame.py > $O/tl_run.log 2>&1
python $GRAFT_REPO_ROOT/tools/dbg/timeline_gaps.py $(dirname $(find /tmp/prof_tl -name "*.db" | head -1)) > $O/timeline.log 2>&1
python $GRAFT_REPO_ROOT/tools/dbg/timeline_all.py $(dirname $(find /tmp/prof_tl -name "*.db" | head -1)) > $O/timeline_all.log 2>&1
