#!/bin/bash
# Developer tool: copy what tools/dbg/round_evidence.sh <tag> left under gpurun_out/ into profiles/ (run in the build container
# after the gpurun call):  bash tools/dbg/install_round.sh r05
set -e
TAG=${1:-r06}
cd "$(dirname "$0")/../.."
E=gpurun_out/evidence_$TAG
P=gpurun_out/profiles_$TAG
B=$(python -c "from icp_flow_amd import _lib; print(_lib.BUILD_INFO)")
cp $P/${TAG}_* profiles/
cp $E/${TAG}_bench.json profiles/${TAG}_bench.json; grep -v amdgpu.ids $E/${TAG}_bench.err > profiles/${TAG}_bench.err || true
grep '^{' $E/${TAG}_bench_stream.json > profiles/${TAG}_bench_stream.json; grep '^{' $E/${TAG}_bench_stream_rccl.json > profiles/${TAG}_bench_stream_rccl.json   # (RCCL prints its version banner on stdout first)
grep -v amdgpu.ids $E/tail_clock.txt > profiles/${TAG}_icp_tail_clock.txt
grep -v amdgpu.ids $E/stage1_tail.txt > profiles/${TAG}_frame_stage1_tail.txt
{ grep -v amdgpu.ids $E/ragged_tail_matched.txt; grep -v amdgpu.ids $E/ragged_tail_independent.txt; grep -v amdgpu.ids $E/ragged_units.txt; } > profiles/${TAG}_ragged_tail_clocks.txt
{ echo "tools/dbg/share_ab.py on one MI355X, build $B: shared window scans of the team kernel against a build WITHOUT the code (-DICPFLOW_NO_SHARE),"
  echo "each library in its own process; transforms and iteration counts compared bit for bit; step / ICP launch in ms (base -> new)"
  grep -v amdgpu.ids $E/share_ab.txt; } > profiles/${TAG}_shared_scans_ab.txt
{ echo "tools/dbg/frame_stamps.py (host time stamps inside icpflow_track_frame, medians of 21 demo frame pairs, us) and tools/dbg/overlap_direct_ab.py"
  echo "(stage 2 beside stage 1's ICP on / off, alternating in one process; from the default stream and from a created stream that may share"
  echo "the second stream's hardware queue), one MI355X, build $B"
  grep -v amdgpu.ids $E/frame_stamps.txt; grep -v amdgpu.ids $E/overlap_ab.txt; } > profiles/${TAG}_frame_pair_host_stamps.txt
python tools/dbg/timeline_all.py $E/frame_trace > /tmp/tl.txt 2>/dev/null || python - "$E/frame_trace" > /tmp/tl.txt <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", "?")) for r in csv.DictReader(open(f))), key=lambda r: r[0])
idx = [i for i, r in enumerate(rows) if "table_dict_kernel" in r[2]]
fr = rows[idx[-1]:]
last = max(i for i, r in enumerate(fr) if "flow_rigid" in r[2])
fr = fr[:last + 1]
t0 = fr[0][0]; end = t0
print("the kernels of the LAST demo frame pair of tools/dbg/frame_stamps.py (max_points 10000) under rocprofv3 --kernel-trace: start, duration, gap to the end of everything before it (us), stream")
for s, e, n, st in fr:
    short = n.replace("(anonymous namespace)::", "").split("(")[0].replace("icpflow::", "").replace("void ", "")[:64]
    print(f"{(s - t0) / 1e3:9.1f}  +{(e - s) / 1e3:7.1f}  gap {(s - end) / 1e3:7.1f}  s{st}  {short}")
    end = max(end, e)
print(f"frame span {(end - t0) / 1e3:.1f} us over {len(fr)} kernels")
PY
cp /tmp/tl.txt profiles/${TAG}_frame_pair_native_timeline.txt
{ echo "tools/dbg/stream_stress.py on one MI355X, build $B: 6 passes of a 12-frame stream of the demo frame pair per setting (clouds uploaded per"
  echo "frame pair); every flow compared by torch.equal with the flow of the same host one frame pair at a time; a team that times out raises."
  grep -v amdgpu.ids $E/stress_default.txt; } > profiles/${TAG}_stream_stress.txt
{ echo "Developer fuzzers on the round's library (build $B), one MI355X; tools/dbg/*_fuzz.py"
  for f in cert_fuzz frame_fuzz score_fuzz registration_fuzz native_fuzz vote_list_fuzz dir_keys_fuzz; do [ -f $E/$f.txt ] || continue; echo; echo "== tools/dbg/$f.py (last lines)"; grep -v amdgpu.ids $E/$f.txt | tail -4; done; } > profiles/${TAG}_fuzz_final_build.txt
{ echo "bench.py --workload stream, five consecutive runs per setting of GPU_MAX_HW_QUEUES (tools/dbg/stream_repro.sh: three timed steps of 64 demo frame"
  echo "pairs each; then tools/dbg/stream_repro2.sh: ten timed steps, 16 and 32 queues), and four config-2 batches in one hist_icp_many call per process;"
  echo "one MI355X, build $B.  The package sets GPU_MAX_HW_QUEUES=16 at import (icp_flow_amd/__init__.py); 'unset' = ICPFLOW_KEEP_HW_QUEUES=1."
  cat $E/stream_repro.txt; cat $E/stream_repro2.txt
  echo; echo "tools/dbg/stream_repro3.sh: the same at the evidence's own length (20 timed steps, 3 warm-up), five runs per setting, twice, alternating"
  [ -f $E/stream_repro3.txt ] && cat $E/stream_repro3.txt
  echo; echo "tools/dbg/four_queues.sh: the default bench.py line five times per setting -- its extras that use several streams INSIDE one busy process"
  echo "(stream4 = four demo frame pairs in flight at max_points 2048 / 10000, four = four config-2 batches in one hist_icp_many call)"
  [ -f $E/four_queues.txt ] && cat $E/four_queues.txt; } > profiles/${TAG}_stream_repro.txt
{ echo "The sweeps' sort keys: the fixed cloud's longest axis (ICPFLOW_OPT_NO_DIR_KEYS) against the best of three axes and six horizontal directions"
  echo "(csrc/sortdir.hpp); tools/dbg/dir_keys_ab.py (step / ICP launch in ms, r = ragged independent sizes, m = matched sizes), dir_keys_diff.py and"
  echo "dir_keys_fuzz.py (what differs: the order of the fp64 moment sums); one MI355X, build $B"
  grep -v amdgpu.ids $E/dir_keys_ab.txt; grep -v amdgpu.ids $E/dir_keys_diff.txt; grep -v amdgpu.ids $E/dir_keys_fuzz.txt | tail -3; } > profiles/${TAG}_direction_keys.txt
{ echo "The ICP of batches of a few rounds in ONE launch (default) against two (ICPFLOW_OPT_TWO_LAUNCH: the grid of half-CU workgroups drained once the"
  echo "unfinished pairs fit one CU each, the rest resumed on whole CUs; icp.hip icp_split_kernel): tools/dbg/two_launch_ab.py, then -- library built with"
  echo "-DICPFLOW_TAIL_CLOCK -- tools/dbg/help_timeline.py (the one launch: resident owners over its span, the pairs that end it) and"
  echo "tools/dbg/two_launch_stats.py (when the drain happens, what the second launch's pairs take); config 4's shard, one MI355X, build $B"
  grep -v amdgpu.ids $E/two_launch_ab.txt; grep -v amdgpu.ids $E/help_timeline.txt; grep -v amdgpu.ids $E/two_launch_stats.txt; } > profiles/${TAG}_two_launch.txt
{ echo "What the sorted vote evaluates (library built with -DICPFLOW_VOTE_STATS, tools/dbg/vote_stats.py), one MI355X, build $B; VERDICT r5 item 6"
  grep -v amdgpu.ids $E/vote_stats.txt; } > profiles/${TAG}_vote_stats.txt
grep -o '"library_build": "[0-9a-f]*"' profiles/${TAG}_bench.json profiles/${TAG}_icp_kernel_counters.json profiles/${TAG}_ragged_counters.json profiles/${TAG}_config4_shard_counters.json | sort | uniq -c
