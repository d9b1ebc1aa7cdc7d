"""CPU oracle for the ICP-Flow cluster-pair registration hot path.

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this package; the product (icp_flow_amd/) never does
and fails loudly when its HIP library is missing.
"""
