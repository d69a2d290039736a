import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmrec_amd import hip_ops, synth
dev = torch.device("cuda:0")
nu, ni, eu, ei = synth.shaped_edges("c5", seed=0)
r, c, v = synth.sym_norm_coo(eu, ei, nu, ni)
n = nu + ni
g = hip_ops.CsrGraph.from_coo_host(np.stack([r, c]), v, n, n, dev, symmetric=True)
for d in (8, 16, 32):
    x = torch.rand(n, d, device=dev) - 0.5
    y = torch.empty_like(x)
    for on in (True, False):
        hip_ops.SLICE_WINDOWS = on
        g._wlists = {}
        for _ in range(6):
            hip_ops.spmm_raw(g, x, Y=y)
        torch.cuda.synchronize()
