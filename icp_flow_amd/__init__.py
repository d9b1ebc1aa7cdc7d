"""icp_flow_amd -- MI355X-native drop-in for ICP-Flow's cluster-pair registration
hot path (utils_hist / utils_icp / utils_match of yanconglin/ICP-Flow).

Device work is done by hand-written HIP kernels for gfx950 behind a C ABI
(`include/icpflow_hip.h`, built into `icp_flow_amd/libicpflow_hip.so`); this
package is the thin Python host side that mirrors the reference's function
names.  There is NO CPU fallback: importing the operator modules without the
HIP library raises.
"""
__version__ = "0.1.0"

import os as _os

# The hardware queues of the process (round 6): HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES
# hardware queues, four by default.  A frame pair in flight uses two streams (stage 2's initial poses beside stage 1's ICP), four
# frame pairs in flight eight; with four queues, which streams end up SHARING a queue differs from process to process, and the
# overlap that shortens a frame pair by 0.2 ms costs 0.2 ms where the two streams share one (measured: ms / frame pair of the same
# stream 0.86-1.19 from run to run with four queues, 0.81-0.85 with sixteen; profiles/r06_stream_repro.txt).  So the package asks
# for sixteen -- before the HIP runtime initialises, which is why it happens at import; a process that has set the variable
# itself, or sets ICPFLOW_KEEP_HW_QUEUES=1, keeps its own choice.  Results do not depend on it.
if "GPU_MAX_HW_QUEUES" not in _os.environ and _os.environ.get("ICPFLOW_KEEP_HW_QUEUES", "0") in ("0", ""):
    _os.environ["GPU_MAX_HW_QUEUES"] = "16"
