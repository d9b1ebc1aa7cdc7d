#!/bin/bash
# Developer tool: variants of the library that differ in -D flags of ONE source (default icp.hip), built in the build container
# from the cached objects of the product build (icp_flow_amd/csrc/_obj) -- no GPU minutes spent compiling:
#   bash tools/dbg/icp_define_build.sh "ICPFLOW_SHARE_MIN_W=128 ICPFLOW_SHARE_PART_MIN=64" "ICPFLOW_TEAM_CHAIN=2" ...
# writes tools/dbg/sweep_<k>.so (+ sweep_<k>.txt with the flags); run them with tools/dbg/icp_define_run.sh in one gpurun call.
cd "$(dirname "$0")/../.."
SRC=${SWEEP_SRC:-icp.hip}
C=icp_flow_amd/csrc
BASE=$(basename $SRC .hip)
OTHERS=$(ls $C/_obj/*.o | grep -v "/$BASE\.[0-9a-f]*\.o")
rm -f tools/dbg/sweep_*.so tools/dbg/sweep_*.txt
k=0
for DEFS in "$@"; do
  k=$((k+1))
  FLAGS=""; for d in $DEFS; do FLAGS="$FLAGS -D$d"; done
  echo "$DEFS" > tools/dbg/sweep_$k.txt
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
      -Wno-unused-function $FLAGS -Iinclude -I$C -c $C/$SRC -o /tmp/sweep_$k.o 2>&1 | grep -v warning | head -5
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OTHERS /tmp/sweep_$k.o -o tools/dbg/sweep_$k.so && echo "built sweep_$k [$DEFS]" ) &
  [ $((k % ${SWEEP_JOBS:-6})) -eq 0 ] && wait
done
wait
