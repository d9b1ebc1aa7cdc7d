"""icp_flow_amd -- MI355X-native drop-in for ICP-Flow's cluster-pair registration
hot path (utils_hist / utils_icp / utils_match of yanconglin/ICP-Flow).

Device work is done by hand-written HIP kernels for gfx950 behind a C ABI
(`include/icpflow_hip.h`, built into `icp_flow_amd/libicpflow_hip.so`); this
package is the thin Python host side that mirrors the reference's function
names.  There is NO CPU fallback: importing the operator modules without the
HIP library raises.
"""
__version__ = "0.1.0"
