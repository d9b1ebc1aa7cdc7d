#!/bin/bash
# Developer tool: copy what tools/profile_round.sh, tools/profile_ragged.sh, tools/dbg/prof_config4_shard.sh and
# tools/dbg/final_evidence.sh left under gpurun_out/ into profiles/ (run in the build container after the gpurun call).
set -e
cd "$(dirname "$0")/../.."
E=gpurun_out/evidence
B=$(python -c "from icp_flow_amd import _lib; print(_lib.BUILD_INFO)")
cp gpurun_out/profiles_r04/r04_icp_kernel_counters.json gpurun_out/profiles_r04/r04_icp_kernel_traffic.json gpurun_out/profiles_r04/r04_pmc_summary.json \
   gpurun_out/profiles_r04/r04_kernel_stats.csv gpurun_out/profiles_r04/r04_ragged_counters.json gpurun_out/profiles_r04/r04_kernel_stats_ragged_* profiles/
cp gpurun_out/c4shard/run_kernel_stats.csv profiles/r04_kernel_stats_config4_shard_1024x2048.csv
cp $E/r04_bench.json profiles/r04_bench.json; cp $E/r04_bench.err profiles/r04_bench.err
grep -v amdgpu.ids $E/tail_clock.txt > profiles/r04_icp_tail_clock.txt
grep -v amdgpu.ids $E/tail_split.txt > profiles/r04_icp_tail_split.txt
grep -v amdgpu.ids $E/stage1_tail.txt > profiles/r04_frame_stage1_tail.txt
sed -n '/^(per-unit clocks of the 4614/,$p' profiles/r04_ragged_tail_clocks.txt > /tmp/units_part.txt
{ grep -v amdgpu.ids $E/ragged_tail_matched.txt; grep -v amdgpu.ids $E/ragged_tail_independent.txt; cat /tmp/units_part.txt; } > profiles/r04_ragged_tail_clocks.txt
grep -v amdgpu.ids $E/config2_units.txt > profiles/r04_config2_wave_units.txt
{ echo "tools/dbg/stream_stress.py on one MI355X, final build $B: 6 passes of a 12-frame stream of the demo frame pair per setting (clouds"
  echo "uploaded per frame pair), teams on half of the CUs, team launches chained two deep per device; every flow compared by torch.equal with"
  echo "the flow of the same host one frame pair at a time; a team that times out raises.  First the default host (icpflow_track_frame, a host"
  echo "thread per frame pair in flight), then DEVICE_ASSOC=1 NATIVE=0 (the Python scheduler with the device-side association) and"
  echo "DEVICE_ASSOC=0 (the Python scheduler with the host-side association)."
  grep -v amdgpu.ids $E/stress_default.txt; grep -v amdgpu.ids $E/stress_device.txt; [ -f $E/stress_host.txt ] && grep -v amdgpu.ids $E/stress_host.txt; } > profiles/r04_stream_stress.txt
{ echo "Developer fuzzers on the final round-4 library (build $B), one MI355X; tools/dbg/*_fuzz.py"
  for f in cert_fuzz frame_fuzz score_fuzz registration_fuzz native_fuzz; do [ -f $E/$f.txt ] || continue; echo; echo "== tools/dbg/$f.py (last lines)"; grep -v amdgpu.ids $E/$f.txt | tail -4; done; } > profiles/r04_fuzz_final_build.txt
grep -o '"library_build": "[0-9a-f]*"' profiles/r04_bench.json profiles/r04_icp_kernel_counters.json profiles/r04_ragged_counters.json | sort | uniq -c
