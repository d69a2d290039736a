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


def time_layer(g, x, y, reps=20):
    for _ in range(3):
        hip_ops.spmm_raw(g, x, Y=y)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        hip_ops.spmm_raw(g, x, Y=y)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def measure(dev, log=print):
    """-> dict: ms per layer with ids grouped by community / randomly permuted / randomly permuted then RELABELLED at build time
    (hip_ops.PermutedGraph: 'community' = label propagation, 'degree', 'rcm'), the relabelling's host time and the two
    permutation passes a propagation pays for it (d = 64; d = 8: one feature slice)."""
    nu, ni, ne = 1_000_000, 500_000, 10_000_000
    eu, ei, _ = community_edges(nu, ni, ne, 2000, 1000)
    rng = np.random.default_rng(1)
    pu, pi = rng.permutation(nu), rng.permutation(ni)
    gen = torch.Generator(device=dev).manual_seed(0)
    n = nu + ni
    out = {}

    def graph(a, b):
        o = np.lexsort((b, a))
        r, c, v = synth.sym_norm_coo(a[o], b[o], nu, ni)
        return hip_ops.CsrGraph.from_coo_host(np.stack([r, c]), v, n, n, dev, symmetric=True)
    x = {d: torch.rand(n, d, device=dev, generator=gen) - 0.5 for d in (64, 8)}
    y = {d: torch.empty_like(x[d]) for d in x}
    for name, (a, b) in (("grouped", (eu, ei)), ("scrambled", (pu[eu], pi[ei]))):
        g = graph(a, b)
        out[name] = {"ms_per_layer_d%d" % d: time_layer(g, x[d], y[d]) for d in x}
        log("%-28s d64 %.3f ms  d8 %.3f ms per layer" % (name, out[name]["ms_per_layer_d64"], out[name]["ms_per_layer_d8"]))
        if name == "scrambled":
            for how in ("community", "degree", "rcm"):
                t0 = time.time()
                pg = hip_ops.PermutedGraph(g, how, n_left=nu)
                host_s = time.time() - t0
                rec = {"ms_per_layer_d%d" % d: time_layer(pg.graph, x[d], y[d]) for d in x}
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(10):
                    pg.to_old(pg.to_new(x[64]))
                torch.cuda.synchronize()
                rec["ms_two_permutation_passes_d64"] = (time.perf_counter() - t0) / 10 * 1e3
                rec["relabel_host_s"] = host_s
                out["scrambled+" + how] = rec
                log("%-28s d64 %.3f ms  d8 %.3f ms per layer; permute in + out %.3f ms per propagation; relabelling %.1f s on the host"
                    % ("scrambled + " + how, rec["ms_per_layer_d64"], rec["ms_per_layer_d8"], rec["ms_two_permutation_passes_d64"], host_s))
                del pg
            # what the plugins do with config `reorder` (models/_base.py: RelabelledIdsMixin): users ranked among users, items
            # among items (two tables stay two tables), every table KEPT in the relabelled space -- the layer time below is
            # the whole cost, there are no permutation passes
            from mmrec_amd.graph import BipartiteRelabelling, relabel_graph
            t0 = time.time()
            rl = BipartiteRelabelling(g, nu, ni, "community", dev)
            gm = relabel_graph(g, rl.node_perm_host())
            host_s = time.time() - t0
            rec = {"ms_per_layer_d%d" % d: time_layer(gm, x[d], y[d]) for d in x}
            rec["ms_two_permutation_passes_d64"] = 0.0
            rec["relabel_host_s"] = host_s
            rec["what"] = "config `reorder: community` as FREEDOM applies it: tables live in the relabelled ids, nothing is permuted per step"
            out["scrambled+community_in_model"] = rec
            log("%-28s d64 %.3f ms  d8 %.3f ms per layer; no permutation passes; relabelling %.1f s on the host"
                % ("scrambled + reorder (model)", rec["ms_per_layer_d64"], rec["ms_per_layer_d8"], host_s))
            del gm
        del g
    return out


def main():
    measure(torch.device("cuda:0"))


if __name__ == "__main__":
    main()
