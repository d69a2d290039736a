import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from mmrec_amd import hip_ops, synth
dev = torch.device("cuda:0")
nu, ni, eu, ei = synth.shaped_edges("c5", seed=0)
r, c, v = synth.sym_norm_coo(eu, ei, nu, ni)
n = nu + ni
x = torch.rand(n, 64, device=dev) - 0.5
y = torch.empty_like(x)
for thr in (16, 24, 32, 64):
    g = hip_ops.CsrGraph.from_coo_host(np.stack([r, c]), v, n, n, dev, symmetric=True, long_row_threshold=thr)
    for _ in range(5):
        hip_ops.spmm_raw(g, x, Y=y)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        hip_ops.spmm_raw(g, x, Y=y)
        hip_ops.spmm_raw(g, y, Y=x)
    e.record()
    torch.cuda.synchronize()
    print("c5 thr %d: %.1f us/layer (long rows %d, chunks %d)" % (thr, s.elapsed_time(e) / 40 * 1e3, g.n_long, g.n_chunks), flush=True)
