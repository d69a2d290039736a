#!/usr/bin/env python3
"""What locality is worth to the CSR SpMM (round-2 review: LDS-staged embedding tiles were never tried on a graph WITH
locality).  A config-5-sized bipartite graph (1M users, 500K items, 10M interactions) whose users belong to
communities of 2,000 users x 1,000 items (90 % of a user's interactions fall inside its community, item popularity zipf inside a
community), laid out two ways:
  * ids randomly permuted  -- what the synthetic benchmark graph looks like to the kernel: no structure to exploit;
  * ids grouped by community (a build-time relabelling, free at run time): the rows a workgroup gathers are shared with its
    neighbours in the grid.
Same kernel, same nonzeros, same per-row summation order up to the column permutation.  Prints ms per layer and, under
rocprofv3 (tools/pmc_kernels.py is not needed: run `rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python tools/spmm_locality_probe.py`
), the fetch bytes tell how much L2 absorbed.  If grouping alone brings the launch close to the compulsory traffic, L2 (4 MB per
XCD = 16K rows) already does what an LDS tile (160 KB = 640 rows) would."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmrec_amd import hip_ops, synth  # noqa: E402


def community_edges(nu, ni, ne, cu, ci, inside=0.9, seed=0):
    rng = np.random.default_rng(seed)
    n_comm = nu // cu
    u = rng.integers(0, nu, int(ne * 1.1))
    comm = u // cu
    pop = np.arange(1, ci + 1, dtype=np.float64) ** -0.8
    cdf = np.cumsum(pop) / pop.sum()
    local = np.searchsorted(cdf, rng.random(u.shape[0]), side="right").clip(0, ci - 1)
    far = rng.integers(0, ni, u.shape[0])
    it = np.where(rng.random(u.shape[0]) < inside, (comm % (ni // ci)) * ci + local, far)
    key = np.unique(u.astype(np.int64) * ni + it)
    if key.shape[0] > ne:
        key = key[np.sort(rng.choice(key.shape[0], ne, replace=False))]
    return key // ni, key % ni, n_comm


def main():
    dev = torch.device("cuda:0")
    nu, ni, ne = 1_000_000, 500_000, 10_000_000
    eu, ei, _ = community_edges(nu, ni, ne, 2000, 1000)
    rng = np.random.default_rng(1)
    pu, pi = rng.permutation(nu), rng.permutation(ni)
    gen = torch.Generator(device=dev).manual_seed(0)
    for name, (a, b) in (("grouped by community", (eu, ei)), ("randomly permuted ids", (pu[eu], pi[ei]))):
        o = np.lexsort((b, a))
        r, c, v = synth.sym_norm_coo(a[o], b[o], nu, ni)
        g = hip_ops.CsrGraph.from_coo_device(torch.from_numpy(r.astype(np.int32)).to(dev), torch.from_numpy(c.astype(np.int32)).to(dev),
                                             torch.from_numpy(v).to(dev), nu + ni, nu + ni, symmetric=True)
        x = torch.rand(nu + ni, 64, device=dev, generator=gen) - 0.5
        y = torch.empty_like(x)
        for _ in range(3):
            hip_ops.spmm_raw(g, x, Y=y)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            hip_ops.spmm_raw(g, x, Y=y)
            x, y = y, x
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 20 * 1e3
        alg = 264.0 * g.nnz + 260.0 * (nu + ni)
        print("%-24s nnz %d: %.3f ms per layer = %.1f G edges/s, gather model %.2f TB/s (%.2f of 8 TB/s), compulsory %.2f TB/s" %
              (name, g.nnz, ms, g.nnz / ms / 1e6, alg / ms / 1e9, alg / ms / 1e9 / 8.0, (8.0 * g.nnz + 516.0 * (nu + ni)) / ms / 1e9),
              flush=True)
        del g, x, y


if __name__ == "__main__":
    main()
