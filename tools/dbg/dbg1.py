import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from conftest import load_golden
from icp_flow_amd import utils_helper
g = load_golden("g3_nn")
for a, b, tag in ((g["src"], g["dst"], "fwd"), (g["dst"], g["src"], "bwd")):
    idx, dist = utils_helper.nearest_neighbor_batch(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda())
    d = dist.cpu().numpy(); w = g["dist_"+tag]
    bad = np.argwhere(d != w)
    print(tag, len(bad), bad[:10])
    for (bi, i) in bad[:10]:
        print(bi, i, d[bi,i], w[bi,i], d[bi,i]-w[bi,i], int(idx[bi,i]), g["idx_"+tag][bi,i], a[bi,i], b[bi,int(idx[bi,i])])
