#!/bin/bash
# Developer tool: build the library with each set of defines given ("A=1 B=2" per argument; "" = as shipped) and time the team
# launches (tools/dbg/team_ab.py) and the large shapes (tools/dbg/lib_ab.py) with each.  One gpurun call.
#   bash tools/dbg/define_sweep.sh "" "ICPFLOW_PROBE_STEPS_LONG=12" "ICPFLOW_PROBE_STEPS_LONG=16 ICPFLOW_PROBE_MAX_LONG=24"
cd "$(dirname "$0")/../.."
C=icp_flow_amd/csrc
i=0
for DEFS in "$@"; do
  i=$((i+1))
  FLAGS=""; for d in $DEFS; do FLAGS="$FLAGS -D$d"; done
  OUT=/tmp/libicpflow_sweep_$i.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
      -Wno-unused-function $FLAGS -Iinclude -I$C -shared -o $OUT \
      $C/api.hip $C/hist.hip $C/nn.hip $C/icp.hip $C/icp_fp32.hip $C/pose.hip $C/sort.hip $C/cluster.hip $C/hdbscan.hip $C/table.hip $C/assoc.hip $C/frame.hip $C/hdbscan_tree.cpp 2>&1 | grep -v warning | head -5 &
done
wait
i=0
for DEFS in "$@"; do
  i=$((i+1))
  echo "== [$DEFS]"
  ICPFLOW_HIP_LIB=/tmp/libicpflow_sweep_$i.so python tools/dbg/team_ab.py 2>&1 | tail -1
  ICPFLOW_HIP_LIB=/tmp/libicpflow_sweep_$i.so python tools/dbg/frame_native_ab.py 2>&1 | tail -1
  ICPFLOW_HIP_LIB=/tmp/libicpflow_sweep_$i.so python tools/dbg/stream_native_ab.py 2>&1 | tail -1
  [ -z "${SWEEP_NO_SHAPES:-}" ] && ICPFLOW_HIP_LIB=/tmp/libicpflow_sweep_$i.so python tools/dbg/lib_ab.py 2>&1 | grep "step" | tail -1
done
