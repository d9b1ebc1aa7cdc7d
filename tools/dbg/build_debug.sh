#!/bin/bash
# Developer tool: build a debug variant of the library next to the product one.
#   tools/dbg/build_debug.sh ICPFLOW_CERT_STATS tools/dbg/libicpflow_dbg.so ; ICPFLOW_HIP_LIB=tools/dbg/libicpflow_dbg.so python tools/dbg/...
set -e
cd "$(dirname "$0")/../.."
DEF=$1; OUT=${2:-tools/dbg/libicpflow_dbg.so}
C=icp_flow_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
    -Wno-unused-function -D$DEF -Iinclude -I$C -shared -o $OUT \
    $C/api.hip $C/hist.hip $C/nn.hip $C/icp.hip $C/icp_fp32.hip $C/pose.hip $C/sort.hip $C/cluster.hip $C/hdbscan.hip $C/table.hip $C/assoc.hip $C/frame.hip $C/hdbscan_tree.cpp
echo built $OUT
