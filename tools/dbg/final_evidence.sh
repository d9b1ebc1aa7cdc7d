set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/evidence; mkdir -p $O
python bench.py > $O/r04_bench.json 2> $O/r04_bench.err
REPS=6 python tools/dbg/stream_stress.py > $O/stress_default.txt 2>&1
REPS=6 NATIVE=0 DEVICE_ASSOC=1 python tools/dbg/stream_stress.py > $O/stress_device.txt 2>&1
REPS=6 DEVICE_ASSOC=0 python tools/dbg/stream_stress.py > $O/stress_host.txt 2>&1
for f in cert_fuzz frame_fuzz score_fuzz registration_fuzz native_fuzz; do timeout 500 python tools/dbg/$f.py > $O/$f.txt 2>&1; echo "$f rc=$?"; done
bash tools/dbg/build_debug.sh ICPFLOW_TAIL_CLOCK tools/dbg/libicpflow_dbg.so > /dev/null 2>&1
export ICPFLOW_HIP_LIB=tools/dbg/libicpflow_dbg.so
python tools/dbg/tail_clock.py > $O/tail_clock.txt 2>&1
python tools/dbg/stage1_tail.py > $O/stage1_tail.txt 2>&1
SIZES=matched TOP=14 python tools/dbg/ragged_tail.py > $O/ragged_tail_matched.txt 2>&1
SIZES=independent TOP=10 python tools/dbg/ragged_tail.py > $O/ragged_tail_independent.txt 2>&1
python tools/dbg/config2_units.py > $O/config2_units.txt 2>&1
unset ICPFLOW_HIP_LIB
C=icp_flow_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-function -DICPFLOW_TAIL_CLOCK -DICPFLOW_TAIL_SPLIT -Iinclude -I$C -shared -o tools/dbg/libicpflow_dbg.so $C/api.hip $C/hist.hip $C/nn.hip $C/icp.hip $C/icp_fp32.hip $C/pose.hip $C/sort.hip $C/cluster.hip $C/hdbscan.hip $C/table.hip $C/assoc.hip $C/frame.hip $C/hdbscan_tree.cpp > /dev/null 2>&1
SPLIT=1 ICPFLOW_HIP_LIB=tools/dbg/libicpflow_dbg.so python tools/dbg/tail_clock.py > $O/tail_split.txt 2>&1
for f in stress_default stress_device stress_host tail_clock stage1_tail; do tail -n 2 $O/$f.txt; done
for f in cert_fuzz frame_fuzz score_fuzz registration_fuzz; do tail -1 $O/$f.txt; done
