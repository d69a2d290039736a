#!/usr/bin/env python3
"""bench.py -- GCN-layer edges/s of the MI355X hot path (BASELINE.json metric), one JSON line.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over the whole graph: the 3-layer LightGCN-style user-item
propagation E <- A E (d = 64, fp32) on the synthetic 10M-edge graph of BASELINE.json configs[4]
(1M users x 500K items, nnz = 20M directed, N = 1.5M), the configuration BASELINE.md section 3 says
the HBM roofline is graded on (Amazon-Baby is cache resident; its numbers ride along in `extra`).
value = directed nnz x layers x steps / wall time (max over ranks), whole job.
At N > 1 the rows are sharded (users and items blockwise, nnz-balanced) and the blocks are all-gathered over
RCCL/xGMI after every layer, chunk by chunk under the next chunk's SpMM: the total work is fixed -> "scaling": "strong".

roofline: HIP events on the launch stream bracket every SpMM call inside the timed region.  The headline
`achieved` / `frac` are the COUNTER view: bytes the launch moves past L2 (rocprofv3 FETCH_SIZE, x2 on gfx950 for
16-B/lane reads, + WRITE_SIZE; collected in-run by re-running a few launches of the same graph under rocprofv3, one
--pmc pass per counter set) / mean launch duration / 8 TB/s -- <= 1 by construction.  The gather-model figure of
SURVEY.md 8d (264 B/nnz + 260 B/row: every nonzero fetches its X row; > 1 of peak because L2 absorbs hot rows), the
compulsory bound (8 B/nnz + 516 B/row) and the L2 hit rate ride next to it.
cpu_baseline (rank 0, N = 1): the reference's own operator, torch.sparse.mm on the uncoalesced COO
adjacency (freedom.py:172), on a bounded sample (a few layers) with all host cores.
"""
from __future__ import annotations

import argparse
import csv
import ctypes
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3   # fp32-input MFMA
MFMA_F16_PEAK_TF = 2500.0  # dense fp16 / bf16 MFMA (MI355X_MICROARCH.md; not the 2:1-sparsity figure)
N_LAYERS = 3
DSLICE_WORLDS = (1, 2, 4, 8)   # 64 / world = 64, 32, 16, 8 columns per rank (mmrec_spmm_csr_f32's slice widths)


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


class GpuTelemetry:
    """Shader / memory clock, socket power and junction temperature of one GPU, read IN-PROCESS through librocm_smi64 (ctypes; a
    read takes well under a millisecond -- a `rocm-smi` subprocess takes 0.3 s to start, by which time the chip idles again).
    What tells one lease's 0.68 ms launch from another's 0.78 ms: the chip runs this kernel power- / thermally limited."""

    class _Freq(ctypes.Structure):
        _fields_ = [("has_deep_sleep", ctypes.c_bool), ("num_supported", ctypes.c_uint32), ("current", ctypes.c_uint32),
                    ("frequency", ctypes.c_uint64 * 33)]

    def __init__(self, index=0):
        self.index, self.lib, self.error = int(index), None, None
        try:
            lib = ctypes.CDLL("librocm_smi64.so")
            if lib.rsmi_init(ctypes.c_uint64(0)) != 0:
                raise OSError("rsmi_init failed")
            self.lib = lib
        except Exception as ex:           # no library / no device: the fields read null
            self.error = repr(ex)

    def read(self):
        if self.lib is None:
            return {"error": self.error}
        out, lib, dv = {}, self.lib, ctypes.c_uint32(self.index)
        for name, clk in (("sclk_mhz", 0), ("mclk_mhz", 4)):            # RSMI_CLK_TYPE_SYS, RSMI_CLK_TYPE_MEM
            f = self._Freq()
            if lib.rsmi_dev_gpu_clk_freq_get(dv, ctypes.c_int(clk), ctypes.byref(f)) == 0 and f.current < 33:
                out[name] = f.frequency[f.current] / 1e6
            else:
                out[name] = None
        uw = ctypes.c_uint64(0)
        if lib.rsmi_dev_current_socket_power_get(dv, ctypes.byref(uw)) == 0 or \
                lib.rsmi_dev_power_ave_get(dv, ctypes.c_uint32(0), ctypes.byref(uw)) == 0:
            out["power_w"] = uw.value / 1e6
        else:
            out["power_w"] = None
        mc = ctypes.c_int64(0)
        out["temp_junction_c"] = (mc.value / 1e3 if lib.rsmi_dev_temp_metric_get(dv, ctypes.c_uint32(1), ctypes.c_int(0),
                                                                                  ctypes.byref(mc)) == 0 else None)
        return out


def graph_timeit(fn, reps=50, windows=5, warm=3, prepare=None):
    """`fn` captured ONCE as a hipGraph and replayed: `windows` windows of `reps` replays, each window bracketed by HIP events on
    the replay stream.  Returns per-replay seconds: median / min / max over the windows.  Why: the eager form of a 10-60-launch
    step is partly HOST time (Python, autograd bookkeeping, ~5 us per launch) and swung 2 x between leases of one build (round-4
    review: 117 / 175 / 216 us for the same 113 us of kernels); a replay is the kernels and nothing else.  Falls back to the
    eager loop (mode says so) when the step cannot be captured."""
    def stats(per, mode):
        per = sorted(per)
        return {"median": float(np.median(per)), "min": float(per[0]), "max": float(per[-1]), "windows": len(per), "reps": reps,
                "spread": float((per[-1] - per[0]) / max(np.median(per), 1e-12)), "mode": mode}
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warm):
                fn()
            if prepare is not None:
                prepare()
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        for _ in range(warm):
            graph.replay()
        per = []
        for _ in range(windows):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            a.record()
            for _ in range(reps):
                graph.replay()
            b.record()
            torch.cuda.synchronize()
            per.append(a.elapsed_time(b) * 1e-3 / reps)
        return stats(per, "hipgraph_replay")
    except Exception as ex:
        log("graph_timeit: capture failed (%r); eager timing" % (ex,))
        torch.cuda.synchronize()
        for _ in range(warm):
            fn()
        per = []
        for _ in range(windows):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            per.append((time.perf_counter() - t0) / reps)
        return stats(per, "eager (capture failed: %r)" % (ex,))


def alg_bytes(nnz, n_rows, d=64):
    """SURVEY.md 8(d) gather model: 4 col + 4 val + one X row per nonzero, 4 rowptr + one Y row per output row (d = 64: 264 / 260)"""
    return (8 + 4 * d) * nnz + (4 + 4 * d) * n_rows


def alg_compulsory_bytes(nnz, n_rows, d=64):
    return 8 * nnz + (4 + 8 * d) * n_rows  # colidx + vals once, rowptr + one read of X and one write of Y per row


def alg_line_bytes(nnz, n_rows, d=64):
    """what the fabric moves at best for RANDOM gathers: every L2 miss is a 128-B request (profiles/r04_spmm_pmc.txt:
    TCC_EA0_RDREQ_32B = 0), so a gathered row slice of 4 d < 128 bytes still costs a whole line"""
    line = max(128, 4 * d)
    return (8 + line) * nnz + (4 + 4 * d) * n_rows


_C5_HOST = {}


def build_c5(dev, rank, world, layout, multi, n_chunks=None):
    from mmrec_amd import hip_ops, synth
    from mmrec_amd.dist import BipartiteSharding
    if "g" not in _C5_HOST:                  # generated once per process (N > 1 measures two layouts of the same graph)
        t = time.time()
        nu, ni, eu, ei = synth.shaped_edges("c5", seed=0)
        r, c, v = synth.sym_norm_coo(eu, ei, nu, ni)
        _C5_HOST["g"] = (nu, ni, eu, ei, r, c, v)
        log("c5 graph generated on host in %.1fs (nnz %d)" % (time.time() - t, r.shape[0]))
    nu, ni, eu, ei, r, c, v = _C5_HOST["g"]
    if not multi or layout == "dslice":      # the whole graph on every rank (dslice: a rank owns 64 / world COLUMNS of every table)
        sh = BipartiteSharding(nu, ni, world)
        g = hip_ops.CsrGraph.from_coo_device(
            torch.from_numpy(r.astype(np.int32)).to(dev), torch.from_numpy(c.astype(np.int32)).to(dev),
            torch.from_numpy(v).to(dev), nu + ni, nu + ni, symmetric=True)
        return sh, g, None, None, r, c, v
    if layout == "allreduce":
        # users sharded, items replicated: rank r keeps R_r (its users' rows) and R_r^T
        sh = BipartiteSharding(nu, ni, world)
        ne = eu.shape[0]
        ub = -(-nu // world)
        u0, u1 = rank * ub, min((rank + 1) * ub, nu)
        s, e = np.searchsorted(eu, u0, "left"), np.searchsorted(eu, u1, "left")   # edges sorted by user
        w = v[:ne][s:e]                                      # first half of sym_norm_coo = user rows
        lu = torch.from_numpy((eu[s:e] - u0).astype(np.int32)).to(dev)
        li = torch.from_numpy(ei[s:e].astype(np.int32)).to(dev)
        wv = torch.from_numpy(w).to(dev)
        r_blk = hip_ops.CsrGraph.from_coo_device(lu, li, wv, u1 - u0, ni)
        rt_blk = hip_ops.CsrGraph.from_coo_device(li, lu, wv, ni, u1 - u0)
        return sh, None, r_blk, rt_blk, r, c, v
    # rows sharded, nnz-balanced cut points, `n_chunks` row chunks per rank (chunk-major padded id space)
    if n_chunks is None:      # a chunk should stay a >= ~1M-nnz SpMM (tens of us), else launches dominate
        n_chunks = int(min(4, max(1, r.shape[0] // max(world, 1) // 1_000_000)))
    sh = BipartiteSharding.from_coo(r, nu, ni, world, n_chunks=n_chunks)

    thr = hip_ops.default_long_row_threshold(nu + ni)        # the unsharded graph's plan, for every block (same summation order)

    def make(lr, pc, vals, n_rows, n_cols):
        if isinstance(dev, str) or dev.type != "cuda":       # the CPU stand-in of the block-construction test
            return hip_ops.CsrGraph.from_coo_host(np.stack([lr, pc]), vals, n_rows, n_cols, dev, long_row_threshold=thr)
        return hip_ops.CsrGraph.from_coo_device(torch.from_numpy(lr.astype(np.int32)).to(dev),
                                                torch.from_numpy(pc.astype(np.int32)).to(dev),
                                                torch.from_numpy(np.ascontiguousarray(vals)).to(dev), n_rows, n_cols,
                                                long_row_threshold=thr)
    ublocks, iblocks = sh.rank_blocks(r, c, v, rank, make)
    return sh, None, ublocks, iblocks, r, c, v


def cpu_baseline(r, c, v, n, max_seconds=25.0):
    """torch.sparse.mm on the reference-form (uncoalesced COO) adjacency, all host cores."""
    from oracle import mmrec_oracle as orc
    torch.set_num_threads(os.cpu_count() or 1)
    adj = orc.sparse_coo(np.stack([r, c]), v, n)
    x = torch.rand(n, 64) - 0.5
    t0 = time.time()
    y = orc.spmm(adj, x)          # warm-up layer
    warm = time.time() - t0
    reps = int(max(1, min(10, (max_seconds - warm) // max(warm, 1e-3))))
    t0 = time.time()
    for _ in range(reps):
        y = orc.spmm(adj, y)
    dt = (time.time() - t0) / reps
    return {"value": r.shape[0] / dt, "unit": "edges/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d layers of torch.sparse.mm (reference form: uncoalesced COO, nnz %d, N %d, d 64) "
                      "after 1 warm-up layer; %.0f ms/layer" % (reps, r.shape[0], n, dt * 1e3)}


def cpu_eval_baseline(nu, ni, eu, ei, r, c, v, batch=4096):
    """The second half of BASELINE.json's metric on the host cores, reference form, bounded sample: one evaluation
    batch of `batch` users -- the 3-layer propagation the reference repeats for every batch (freedom.py:212-220:
    full_sort_predict calls forward), U_b I^T, scores[mask] = -1e10, torch.topk(50), hit matrix + metrics
    (trainer.py:292-311, topk_evaluator.py:58-102) -- through the oracle's restatements."""
    from oracle import mmrec_oracle as orc
    torch.set_num_threads(os.cpu_count() or 1)
    n = nu + ni
    adj = orc.sparse_coo(np.stack([r, c]), v, n)
    gen = torch.Generator().manual_seed(0)
    ue, ie = (torch.rand(nu, 64, generator=gen) - 0.5) * 0.1, (torch.rand(ni, 64, generator=gen) - 0.5) * 0.1
    users = np.arange(min(batch, nu))
    sel = eu < users.shape[0]
    mask = np.stack([eu[sel], ei[sel]])
    rng = np.random.default_rng(0)
    pos_len = rng.integers(1, 9, users.shape[0])
    pos_flat = rng.integers(0, ni, int(pos_len.sum()))
    t0 = time.time()
    ua, ia = orc.lightgcn_forward(adj, ue, ie, N_LAYERS)
    scores = orc.full_sort_scores(ua, ia, users)
    _, idx = orc.mask_topk(scores, mask, 50)
    orc.topk_metrics(orc.hit_matrix(idx.numpy(), pos_flat, pos_len), pos_len)
    dt = time.time() - t0
    return {"value": users.shape[0] / dt, "unit": "users/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "one evaluation batch of %d users x %d items (3-layer propagation + scores + mask + top-50 + "
                      "metrics), %.0f ms" % (users.shape[0], ni, dt * 1e3)}


def weak_scaling_run(dev, rank, world, steps):
    """N > 1 companion number: the graph GROWS with the job -- every rank brings its own 1M users and
    10M interactions over the same 500K items (users sharded, items replicated, item sums all-reduced
    per layer).  Per-GPU work is what one GPU does on the c5 graph; only the 128 MB item all-reduce is
    added.  Degrees for the symmetric normalisation are global (item degrees are all-reduced)."""
    import torch.distributed as dist
    from mmrec_amd import hip_ops, synth
    from mmrec_amd.dist import ItemReplicatedPropagator
    nu, ni, eu, ei = synth.shaped_edges("c5", seed=1000 + rank)
    eu_d = torch.from_numpy(eu.astype(np.int64)).to(dev)
    ei_d = torch.from_numpy(ei.astype(np.int64)).to(dev)
    du = torch.bincount(eu_d, minlength=nu).to(torch.float64)
    di = torch.bincount(ei_d, minlength=ni).to(torch.float64)
    dist.all_reduce(di)
    w = ((du[eu_d] + 1e-7).pow(-0.5) * (di[ei_d] + 1e-7).pow(-0.5)).to(torch.float32)
    lu, li = eu_d.to(torch.int32), ei_d.to(torch.int32)
    r_blk = hip_ops.CsrGraph.from_coo_device(lu, li, w, nu, ni)
    rt_blk = hip_ops.CsrGraph.from_coo_device(li, lu, w, ni, nu)
    gen = torch.Generator(device=dev).manual_seed(7 + rank)
    xu = torch.rand(nu, 64, device=dev, generator=gen) - 0.5
    gi = torch.Generator(device=dev).manual_seed(7)          # same item table on every rank
    xi = torch.rand(ni, 64, device=dev, generator=gi) - 0.5
    prop = ItemReplicatedPropagator(r_blk, rt_blk, lambda blk, X, Y: hip_ops.spmm_raw(blk, X, Y=Y),
                                    world_size=world, force_collectives=True)

    def fence():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
    for _ in range(2):
        prop.propagate(xu, xi, N_LAYERS)
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        prop.propagate(xu, xi, N_LAYERS)
    fence()
    tt = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    nnz_rank = 2 * int(eu.shape[0])                          # R_r and R_r^T, as in the c5 count
    return {"edges_per_s": world * nnz_rank * N_LAYERS * steps / dt, "ms_per_step": dt / steps * 1e3,
            "graph": "%d x (1M users, 10M interactions) over 500K shared items, nnz %d" % (world, world * nnz_rank),
            "scaling": "weak"}


def sharded_train_run(dev, rank, world, sh, r_blk, rt_blk, steps):
    """N > 1 companion number: a whole LightGCN-style BPR training step on the c5 graph over the
    users-sharded / items-replicated layout (2 differentiable sharded layers forward + backward, fused
    BPR on the rank's own triplets of a 2048 batch, item-gradient all-reduce, fused Adam)."""
    import torch.distributed as dist
    from mmrec_amd import hip_ops
    from mmrec_amd.common.optim import HipAdam
    from mmrec_amd.dist import ItemReplicatedPropagator, ShardedLightGCNStep
    ub = -(-sh.n_users // world)
    u0, u1 = rank * ub, min((rank + 1) * ub, sh.n_users)
    gen = torch.Generator(device=dev).manual_seed(11)          # same tables / batches on every rank
    U = (torch.rand(sh.n_users, 64, device=dev, generator=gen) - 0.5) * 0.1
    I = (torch.rand(sh.n_items, 64, device=dev, generator=gen) - 0.5) * 0.1
    prop = ItemReplicatedPropagator(r_blk, rt_blk, lambda blk, X, Y: hip_ops.spmm_raw(blk, X, Y=Y),
                                    world_size=world, force_collectives=True)
    st = ShardedLightGCNStep(prop, U[u0:u1].contiguous(), I, 2,
                             lambda a, b, us, p, n: hip_ops.bpr_loss(a, b, us, p, n, reduction="sum"),
                             lr=1e-3, optimizer_cls=HipAdam)
    del U

    def one():
        users = torch.randint(0, sh.n_users, (2048,), device=dev, generator=gen)
        pos = torch.randint(0, sh.n_items, (2048,), device=dev, generator=gen)
        neg = torch.randint(0, sh.n_items, (2048,), device=dev, generator=gen)
        mine = (users >= u0) & (users < u1)
        return st.step((users[mine] - u0).contiguous(), pos[mine].contiguous(), neg[mine].contiguous(), 2048)

    def fence():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
    for _ in range(2):
        one()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = one()
    fence()
    tt = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    return {"ms_per_step": float(tt.item()) / steps * 1e3, "loss": float(loss),
            "what": "LightGCN L=2 + BPR(2048) + Adam on c5, users sharded x%d / items replicated" % world}


def c5_full_eval(dev, rank, world, user_emb, item_emb, eu, ei, n_users, barrier):
    """The second half of BASELINE.json's metric at config-5 size: full-sort evaluation of ALL 1M users against the 500K
    items (train positives masked, top-50), users sharded over the ranks, item table replicated -- no exchange in the data
    path (SURVEY.md 8e "P5 eval: shard users"), so it scales with the GPU count.  Every rank ranks its slice in blocks of
    65,536 users -- the Trainer's own `hip_eval_batch_size`; 256 query blocks x 16 candidate ranges = exactly 8 rounds of
    the 512 resident workgroups, where 50,000-user blocks ran 7 rounds for 6.1 rounds of work -- against ONE preparation
    of the item table (hip_ops.TopkCandidates: column means + centred fp16 copy, 0.56 ms, as the plugins' evaluation does
    it); the [65,536, 500,000] score block is never formed; users/s = all users / max-over-ranks time (preparation
    included)."""
    from mmrec_amd import hip_ops
    per = -(-n_users // world)
    lo, hi = min(rank * per, n_users), min((rank + 1) * per, n_users)
    s, e = np.searchsorted(eu, lo, "left"), np.searchsorted(eu, hi, "left")          # edges are sorted by user
    rp, col = hip_ops.mask_to_csr(np.stack([eu[s:e] - lo, ei[s:e]]), max(hi - lo, 1), dev)
    rp_host = rp.cpu().numpy().astype(np.int64)
    blocks = []
    for a in range(0, hi - lo, 65_536):
        b = min(a + 65_536, hi - lo)
        blocks.append((a, b, (rp[a:b + 1] - rp[a]).contiguous(), col[rp_host[a]:max(rp_host[b], rp_host[a] + 1)].contiguous()))

    # per block a list table ([users, 64] int32: top-50 + the runners-up the kernel ranked), written by the calls themselves
    lists = [torch.full((b - a, hip_ops.topk_hint_width(50)), -1, dtype=torch.int32, device=dev) for a, b, _, _ in blocks]
    counts = torch.zeros(2, dtype=torch.int32, device=dev)

    def run(warm=False, keep=False, item_table=None, user_table=None):
        """plain call; keep: cold call that leaves its lists; warm: threshold from the lists (left untouched: every timed
        repetition sees the same ones)"""
        ie = item_emb if item_table is None else item_table
        ue = user_emb if user_table is None else user_table
        cands = hip_ops.TopkCandidates(ie)
        for j, (a, b, brp, bcol) in enumerate(blocks):
            if warm or keep:
                hip_ops.score_topk(ue[lo + a:lo + b], cands, 50, brp, bcol, hint=lists[j], hint_cold=not warm,
                                   hint_update=not warm, queue_counts=counts if warm else None)
            else:
                hip_ops.score_topk(ue[lo + a:lo + b], cands, 50, brp, bcol)

    def timed(**kw):
        barrier()
        t0 = time.perf_counter()
        run(**kw)
        barrier()
        return time.perf_counter() - t0
    run()
    t_cold = timed()
    # WARM (round 6): the same ranking with each block's threshold taken from last time's lists (mmrec_score_topk_hinted_f32):
    # (a) the lists of these very tables -- the TEST pass after the VALID pass; (b) lists left by tables that have since moved
    # (every element by 5 % of its row's mean magnitude: a stand-in for "some training steps later")
    warm = {}
    try:
        run(keep=True)
        run(warm=True)
        counts.zero_()
        warm["seconds_same_tables"] = timed(warm=True)
        warm["queues_same_tables"] = counts.tolist()
        warm["seconds_cold_leaving_lists"] = timed(keep=True)
        gen = torch.Generator(device=dev).manual_seed(11)
        moved_u = user_emb + 0.05 * user_emb.abs().mean(1, keepdim=True) * torch.randn(user_emb.shape, device=dev, generator=gen)
        moved_i = item_emb + 0.05 * item_emb.abs().mean(1, keepdim=True) * torch.randn(item_emb.shape, device=dev, generator=gen)
        run(keep=True, item_table=moved_i, user_table=moved_u)          # "last epoch's" lists
        del moved_u, moved_i
        counts.zero_()
        warm["seconds_lists_of_moved_tables"] = timed(warm=True)
        warm["queues_lists_of_moved_tables"] = counts.tolist()
        warm["what"] = ("same call, threshold from the previous lists instead of pass 1; same_tables = TEST pass after VALID pass; "
                        "moved = lists of tables perturbed by 5 % per element; queues = [slow, overflow] queries of the pass")
    except Exception as ex:
        warm["error"] = repr(ex)
    return t_cold, warm


def make_freedom_step(dev, nu, ni, eu, ei, gen, lazy=False, capturable=False):
    """One FREEDOM training step (freedom.py:189-210 + Adam over all 33.6 M parameters incl. the
    trainable 7050 x 4096 / 7050 x 384 feature tables): masked-graph propagation, item-item SpMM,
    both projections, three BPR terms, backward, optimizer.  Reference on CPU: 165 ms.
    lazy: what the FREEDOM plugin does by default -- only the batch's pos / neg feature rows are projected (same
    function, same gradients) and the tables are updated by the row-lazy exact Adam (bit-identical parameters)."""
    from mmrec_amd import hip_ops
    import torch.nn as nn
    keep = torch.randperm(eu.shape[0], generator=torch.Generator().manual_seed(0))[:int(eu.shape[0] * 0.2)]
    masked = hip_ops.bipartite_graph_from_edges(torch.from_numpy(eu)[keep].to(dev), torch.from_numpy(ei)[keep].to(dev), nu, ni)
    knn = torch.randint(0, ni, (ni, 10), generator=torch.Generator().manual_seed(1))
    mm_rows = torch.arange(ni).repeat_interleave(10)
    mm = hip_ops.CsrGraph.from_coo_host(np.stack([mm_rows.numpy(), knn.reshape(-1).numpy()]),
                                        np.full(ni * 10, 0.1, np.float32), ni, ni, dev)
    mm.transpose()
    P = lambda *shape, s=0.05: nn.Parameter((torch.rand(*shape, device=dev, generator=gen) - 0.5) * s)
    ue, ie, vt, tt = P(nu, 64), P(ni, 64), P(ni, 4096, s=1.0), P(ni, 384, s=1.0)
    vw, vb, tw, tb = P(64, 4096), P(64), P(64, 384), P(64)
    from mmrec_amd.common.optim import HipAdam
    if lazy:
        from mmrec_amd.common.lazy_rows import LazyRowEmbedding
        vtab = LazyRowEmbedding.from_pretrained(vt.detach(), freeze=False)
        ttab = LazyRowEmbedding.from_pretrained(tt.detach(), freeze=False)
        vt, tt = vtab.weight, ttab.weight
    opt = HipAdam([ue, ie, vt, tt, vw, vb, tw, tb], lr=1e-3, capturable=capturable)
    gb = torch.Generator(device=dev).manual_seed(2)
    users = torch.randint(0, nu, (2048,), device=dev, generator=gb)
    pos = torch.randint(0, ni, (2048,), device=dev, generator=gb)
    neg = torch.randint(0, ni, (2048,), device=dev, generator=gb)

    rows, lp = torch.cat((pos, neg)), torch.arange(2048, device=dev)

    def freedom_step():
        opt.zero_grad(set_to_none=True)
        mean = hip_ops.lightgcn_mean(masked, torch.cat([ue, ie], 0), 2)
        ua, ia = mean[:nu].contiguous(), hip_ops.spmm(mm, ie, Z=mean[nu:].contiguous())
        if lazy:
            loss = hip_ops.bpr_loss(ua, ia, users, pos, neg) + 1e-3 * (
                hip_ops.bpr_loss(ua, hip_ops.linear(ttab.rows(rows), tw, tb), users, lp, lp + 2048) +
                hip_ops.bpr_loss(ua, hip_ops.linear(vtab.rows(rows), vw, vb), users, lp, lp + 2048))
            loss.backward()
            opt.step()
            return
        loss = hip_ops.bpr_loss(ua, ia, users, pos, neg) + 1e-3 * (
            hip_ops.bpr_loss(ua, hip_ops.linear(tt, tw, tb), users, pos, neg) +
            hip_ops.bpr_loss(ua, hip_ops.linear(vt, vw, vb), users, pos, neg))
        loss.backward()
        opt.step()
    freedom_step.opt = opt
    return freedom_step


def c5_fwd_bwd(dev, g, n_nodes, nnz, reps=10):
    """"report fwd+bwd separately" -- the propagation as the models run it (layer mean fused into the
    epilogue, Horner-form backward on the transposed = same symmetric graph) through autograd, config-5 graph"""
    from mmrec_amd import hip_ops
    gen = torch.Generator(device=dev).manual_seed(3)
    E = (torch.rand(n_nodes, 64, device=dev, generator=gen) - 0.5).requires_grad_()
    gO = torch.rand(n_nodes, 64, device=dev, generator=gen) - 0.5

    def fwd():
        with torch.no_grad():
            hip_ops.lightgcn_mean(g, E, N_LAYERS)

    def both():
        E.grad = None
        hip_ops.lightgcn_mean(g, E, N_LAYERS).backward(gO)
    out = {}
    for name, fn, passes in (("fwd", fwd, 1), ("fwd_bwd", both, 2)):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        out["ms_" + name] = dt * 1e3
        out["edges_per_s_" + name] = passes * N_LAYERS * nnz / dt
    out["what"] = ("hip_ops.lightgcn_mean over the config-5 graph, %d layers: forward alone (layer mean fused into the "
                   "SpMM epilogue) and forward + backward through autograd; edges/s counts nnz x layers x passes" % N_LAYERS)
    return out


def extra_baby(dev):
    """Amazon-Baby-shaped numbers (cache resident): 3-layer propagation, full-sort eval, projection."""
    from mmrec_amd import hip_ops, synth
    out = {}
    nu, ni, eu, ei = synth.shaped_edges("baby", seed=0)
    r, c, v = synth.sym_norm_coo(eu, ei, nu, ni)
    n = nu + ni
    g = hip_ops.CsrGraph.from_coo_host(np.stack([r, c]), v, n, n, dev, symmetric=True)
    gen = torch.Generator(device=dev).manual_seed(0)
    E0 = (torch.rand(n, 64, device=dev, generator=gen) - 0.5) * 0.1

    def timeit(fn, reps=50, warm=5, windows=3):
        """median over `windows` back-to-back windows of `reps` calls each: one stall of the box (allocator trim, clock
        ramp: a single 200-call window once read 217 us for a 46 us kernel) does not decide a companion number"""
        for _ in range(warm):
            fn()
        per = []
        for _ in range(windows):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            per.append((time.perf_counter() - t0) / reps)
        return float(np.median(per))

    with torch.no_grad():
        dt = timeit(lambda: hip_ops.lightgcn_mean(g, E0, N_LAYERS))
        out["baby_propagate_edges_per_s"] = g.nnz * N_LAYERS / dt
        out["baby_us_per_layer"] = dt / N_LAYERS * 1e6
        # full-sort eval: every user against every item, train positives masked, top-50
        mask = np.stack([eu, ei])
        rp, col = hip_ops.mask_to_csr(mask, nu, dev)
        emb = hip_ops.lightgcn_mean(g, E0, N_LAYERS)
        U, I = emb[:nu].contiguous(), emb[nu:].contiguous()

        def evaluate():
            e = hip_ops.lightgcn_mean(g, E0, N_LAYERS)
            return hip_ops.score_topk(e[:nu].contiguous(), e[nu:].contiguous(), 50, rp, col)
        dt = timeit(evaluate, reps=10, warm=2)
        out["baby_full_eval_users_per_s"] = nu / dt
        # the same two paths replayed as hipGraphs: at this size an eager call is partly HOST bound (3 + 9 launches of
        # 4-60 us kernels, ~5 us of Python + launch each); the replay shows what the kernels themselves take
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                gp, ge = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                with torch.cuda.graph(gp, stream=side):
                    hip_ops.lightgcn_mean(g, E0, N_LAYERS)
                with torch.cuda.graph(ge, stream=side):
                    evaluate()
            torch.cuda.current_stream().wait_stream(side)
            dt = timeit(gp.replay, reps=200, warm=10)
            out["baby_us_per_layer_graph_replay"] = dt / N_LAYERS * 1e6
            dt = timeit(ge.replay, reps=50, warm=5)
            out["baby_full_eval_users_per_s_graph_replay"] = nu / dt
        except Exception as ex:
            out["baby_graph_replay_error"] = repr(ex)
        dt = timeit(lambda: hip_ops.score_topk(U, I, 50, rp, col), reps=10, warm=2)
        out["baby_score_topk_ms"] = dt * 1e3
        # WARM call (round 6, mmrec_score_topk_hinted_f32): the threshold from the lists of the previous pass over the same
        # users -- the TEST pass after the VALID pass (same frozen tables), or the next epoch's evaluation -- instead of a
        # first pass over all products; identical output.  Replays: what the kernels themselves take.
        try:
            hint = torch.full((nu, hip_ops.topk_hint_width(50)), -1, dtype=torch.int32, device=dev)
            lists = hip_ops.score_topk(U, I, 50, rp, col, hint=hint, hint_cold=True)       # leaves its top-50 + runners-up
            assert torch.equal(hip_ops.score_topk(U, I, 50, rp, col, hint=hint), lists) and torch.equal(hint[:, :50], lists.to(torch.int32))
            out["baby_score_topk_warm_ms_eager"] = timeit(lambda: hip_ops.score_topk(U, I, 50, rp, col, hint=hint), reps=10, warm=2) * 1e3
            st_c = graph_timeit(lambda: hip_ops.score_topk(U, I, 50, rp, col), reps=50)
            st_w = graph_timeit(lambda: hip_ops.score_topk(U, I, 50, rp, col, hint=hint), reps=50)
            out["baby_score_topk_ms_graph_replay"] = st_c["median"] * 1e3
            out["baby_score_topk_warm_ms"] = st_w["median"] * 1e3
            out["baby_score_topk_warm_ms_min_max"] = [st_w["min"] * 1e3, st_w["max"] * 1e3]
            out["baby_score_topk_warm_mode"] = st_w["mode"]

            def evaluate_warm():
                e = hip_ops.lightgcn_mean(g, E0, N_LAYERS)
                return hip_ops.score_topk(e[:nu].contiguous(), e[nu:].contiguous(), 50, rp, col, hint=hint)
            st_e = graph_timeit(evaluate_warm, reps=50)
            out["baby_full_eval_users_per_s_warm_graph_replay"] = nu / st_e["median"]
        except Exception as ex:
            out["baby_score_topk_warm_error"] = repr(ex)
        # fp16 filter on the matrix cores + exact fp32 refinement of the survivors (topk_filter.hip); the rate is
        # the USEFUL work 2 nq nc 64 over the whole call (the fp32 score block it replaces is never formed)
        out["baby_score_topk_tflops"] = 2.0 * nu * ni * 64 / dt / 1e12
        # modal projection 4096 -> 64 over all items (P3)
        X = torch.rand(ni, 4096, device=dev, generator=gen)
        W = torch.rand(64, 4096, device=dev, generator=gen) - 0.5
        b = torch.zeros(64, device=dev)
        dt = timeit(lambda: hip_ops.linear(X, W, b), reps=100, warm=20, windows=5)
        out["baby_linear4096_fwd_us"] = dt * 1e6
        out["baby_linear4096_fwd_tflops"] = 2.0 * ni * 4096 * 64 / dt / 1e12
        out["baby_linear4096_fwd_frac_mfma_f32"] = out["baby_linear4096_fwd_tflops"] / MFMA_F32_PEAK_TF
        # the projection's roofline (north_star: "rocprof MFMA utilisation reported"): FLOP-derived fraction of the fp32-input
        # MFMA peak measured here, the X bytes it streams, and the COUNTER view (SQ_VALU_MFMA_BUSY_CYCLES over the busy CU
        # cycles, tools/pmc_kernels.py linear -> profiles/r03_mfma_pmc.json, collected in its own rocprofv3 passes)
        proj = {"bound": "the HBM stream of X since round 4: the forward runs on the 16-bit matrix cores with split operands "
                         "(x = hi + 2^-11 lo' in fp16, three fp16 MFMA products per 16 k, fp32 accumulators, fp32-accurate); the "
                         "fp32-MFMA form it replaces was bound by the 157.3 TFLOP/s fp32 matrix pipe",
                "peak_tflops": MFMA_F32_PEAK_TF, "kernel": "linear_fwd_dma_f16x3_kernel" if hip_ops.LINEAR_F16X3 else "linear_fwd_dma_kernel",
                "baby": {"n": ni, "F": 4096, "fwd_us": dt * 1e6, "fwd_tflops": out["baby_linear4096_fwd_tflops"],
                         "fwd_frac": out["baby_linear4096_fwd_frac_mfma_f32"], "x_bytes_streamed": 4.0 * ni * 4096,
                         "x_gbs": 4.0 * ni * 4096 / dt / 1e9, "frac_hbm_stream": 4.0 * ni * 4096 / dt / 1e9 / HBM_PEAK_GBS}}
        hip_ops.LINEAR_F16X3 = False                       # the fp32-MFMA kernel, for the A/B
        dt32 = timeit(lambda: hip_ops.linear(X, W, b), reps=100, warm=20, windows=5)
        hip_ops.LINEAR_F16X3 = True
        proj["baby"]["fwd_us_fp32_mfma_kernel"] = dt32 * 1e6
        proj["baby"]["fwd_frac_fp32_mfma_kernel"] = 2.0 * ni * 4096 * 64 / dt32 / 1e12 / MFMA_F32_PEAK_TF
        out["projection_roofline"] = proj
        out["_projection_items"] = ni            # (the counters are collected at the end: a child process under rocprofv3)
    # forward + backward (dW, db, dX) of the projection, as FREEDOM / BM3 run it every batch
    Xg, Wg, bg = X.clone().requires_grad_(), W.clone().requires_grad_(), b.clone().requires_grad_()
    G = torch.rand(ni, 64, device=dev, generator=gen) - 0.5

    def fwd_bwd():
        Xg.grad = Wg.grad = bg.grad = None
        hip_ops.linear(Xg, Wg, bg).backward(G)
    # Companions as hipGraph replays, median of 5 event-timed windows with min / max (round-4 review: the eager forms swung
    # 117-222 us / 0.57-0.88 ms between leases of one build -- host time around ~113 us of kernels); the eager wall time rides
    # along under *_eager_* so the host overhead stays visible.
    st = graph_timeit(fwd_bwd, reps=100, windows=5, warm=5)
    out["baby_linear4096_fwd_bwd_us"] = st["median"] * 1e6
    out["baby_linear4096_fwd_bwd_us_min_max"] = [st["min"] * 1e6, st["max"] * 1e6]
    out["baby_linear4096_fwd_bwd_mode"] = st["mode"]
    out["baby_linear4096_fwd_bwd_tflops"] = 3 * 2.0 * ni * 4096 * 64 / st["median"] / 1e12
    out["baby_linear4096_fwd_bwd_us_eager_wall"] = timeit(fwd_bwd, reps=100, warm=10) * 1e6
    hip_ops.LINEAR_F16X3 = False                           # A/B: forward, dW and dX on the fp32-MFMA kernels
    Xg.grad = Wg.grad = bg.grad = None
    st32 = graph_timeit(fwd_bwd, reps=100, windows=5, warm=5)
    hip_ops.LINEAR_F16X3 = True
    out["baby_linear4096_fwd_bwd_us_fp32_mfma_kernels"] = st32["median"] * 1e6
    # the backward's write roofline: dX is n x 4096 fp32 written once (kernel time from the in-run counters pass below)
    out["projection_roofline"]["baby"]["dx_bytes_written"] = 4.0 * ni * 4096
    Xg.grad = Wg.grad = bg.grad = None
    del Xg, G
    try:      # the same call at Amazon-Sports size (18,357 items: X = 301 MB streams from HBM, 144 row blocks -> three chunks each)
        ns = 18357
        Xs = torch.rand(ns, 4096, device=dev, generator=gen).requires_grad_()
        Gs = torch.rand(ns, 64, device=dev, generator=gen) - 0.5

        def fwd_bwd_sports():
            Xs.grad = Wg.grad = bg.grad = None
            hip_ops.linear(Xs, Wg, bg).backward(Gs)
        sts = graph_timeit(fwd_bwd_sports, reps=50, windows=5, warm=5)
        with torch.no_grad():
            dts = timeit(lambda: hip_ops.linear(Xs, Wg, bg), reps=50, warm=10, windows=3)
        out["projection_roofline"]["sports"] = {"n": ns, "F": 4096, "fwd_bwd_us": sts["median"] * 1e6,
                                                "fwd_bwd_us_min_max": [sts["min"] * 1e6, sts["max"] * 1e6], "mode": sts["mode"],
                                                "fwd_us_eager": dts * 1e6, "x_gbs_fwd_eager": 4.0 * ns * 4096 / dts / 1e9}
        Xs.grad = Wg.grad = bg.grad = None
        del Xs, Gs
    except Exception as ex:
        out["projection_roofline"]["sports"] = {"error": repr(ex)}
    for key, lazy in (("baby_freedom_train_step", False), ("baby_freedom_train_step_lazy", True)):
        reps, windows, warm = 20, 5, 3
        step = make_freedom_step(dev, nu, ni, eu, ei, gen, lazy=lazy, capturable=True)
        # (the row-lazy tables keep one pair of scalars per optimizer step in a device table a captured step cannot grow)
        st = graph_timeit(step, reps=reps, windows=windows, warm=warm,
                          prepare=lambda: (step.opt.init_state(2 * (reps * windows + 2 * warm + 8)), step.opt.zero_grad(set_to_none=True)))
        out[key + "_ms"] = st["median"] * 1e3
        out[key + "_ms_min_max"] = [st["min"] * 1e3, st["max"] * 1e3]
        out[key + "_mode"] = st["mode"]
        del step
        step = make_freedom_step(dev, nu, ni, eu, ei, gen, lazy=lazy)
        out[key + "_ms_eager_wall"] = timeit(step, reps=20, warm=3) * 1e3
        del step
    try:
        out["cpu_baseline_full_eval"] = cpu_eval_baseline(nu, ni, eu, ei, r, c, v)
    except Exception as ex:
        out["cpu_baseline_full_eval"] = {"error": repr(ex)}
    with torch.no_grad():   # last: its 557 MB score block evicts everything the measurements above keep in cache
        # the materialised fp32-MFMA path, for comparison
        out["baby_score_topk_materialised_ms"] = timeit(lambda: hip_ops.score_topk(U, I, 50, rp, col, use_filter=False),
                                                        reps=10, warm=2) * 1e3
    return out


def pmc_child(path):
    """rocprofv3 child: a few launches of mmrec_spmm_csr_f32 on the graph the parent saved, nothing else
    (`linear:<n>`: a few forward + backward passes of the 4096 -> 64 projection over n items instead)."""
    from mmrec_amd import hip_ops
    dev = torch.device("cuda", 0)
    if path.startswith("linear:"):
        n = int(path.split(":")[1])
        gen = torch.Generator(device=dev).manual_seed(0)
        X = torch.rand(n, 4096, device=dev, generator=gen).requires_grad_()
        W = (torch.rand(64, 4096, device=dev, generator=gen) - 0.5).requires_grad_()
        b = torch.zeros(64, device=dev, requires_grad=True)
        G = torch.rand(n, 64, device=dev, generator=gen) - 0.5
        for _ in range(4):
            X.grad = W.grad = b.grad = None
            hip_ops.linear(X, W, b).backward(G)
        torch.cuda.synchronize()
        return
    z = np.load(path)
    n = int(z["n"])
    g = hip_ops.CsrGraph(torch.from_numpy(z["rowptr"]).to(dev), torch.from_numpy(z["colidx"]).to(dev),
                         torch.from_numpy(z["vals"]).to(dev), n, n, symmetric=True, rowptr_host=z["rowptr"])
    x = torch.rand(n, 64, device=dev, generator=torch.Generator(device=dev).manual_seed(0)) - 0.5
    y = torch.empty_like(x)
    torch.cuda.synchronize()
    for _ in range(4):
        hip_ops.spmm_raw(g, x, Y=y)
        x, y = y, x
    torch.cuda.synchronize()


def measure_traffic(g, n_nodes):
    """Counters of the dominant kernel, IN THIS RUN: the same graph is handed to a child process that is run under
    `rocprofv3 --kernel-trace --pmc <set>` once per counter set (FETCH_SIZE and WRITE_SIZE do not fit one pass:
    MI355X_MICROARCH.md, rocprofv3 PMC slots).  Returns per-launch numbers (rows/chunks kernel + long-row reduce) or
    None when rocprofv3 is not usable here (the committed profile is quoted instead, and the line says so)."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        log("bench.py is itself running under a profiler: not nesting rocprofv3 (quoting the committed profile)")
        return None
    tmp = tempfile.mkdtemp(prefix="mmrec_pmc_", dir="/tmp")
    try:
        path = os.path.join(tmp, "graph.npz")
        np.savez(path, rowptr=g.rowptr_host, colidx=g.colidx.cpu().numpy(), vals=g.vals.cpu().numpy(), n=n_nodes)
        out = {}
        env = dict(os.environ, TMPDIR="/tmp")
        for tag, counters in (("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"]), ("l2", ["TCC_HIT_sum", "TCC_MISS_sum"])):
            d = os.path.join(tmp, tag)
            cmd = [exe, "--kernel-trace", "--pmc"] + counters + ["--output-format", "csv", "-d", d, "-o", "pm", "--",
                                                                 sys.executable, os.path.abspath(__file__), "--pmc-child", path]
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=100)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                log("rocprofv3 pass '%s' failed (rc %d): %s" % (tag, r.returncode, r.stderr[-300:]))
                return None
            per = {}
            launches = 0
            for row in csv.DictReader(open(files[0])):
                k = row["Kernel_Name"]
                if "spmm_rows_kernel" in k or "spmm_long_reduce_kernel" in k:
                    per[row["Counter_Name"]] = per.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                    if "spmm_rows_kernel" in k and row["Counter_Name"] == counters[0]:
                        launches += 1
            if not launches:
                return None
            for c in counters:
                out[c] = per.get(c, 0.0) / launches
        fetch = out["FETCH_SIZE"] * 1024.0 * 2.0        # KB -> B; x2: gfx950 tallies 16-B/lane reads at half size
        write = out["WRITE_SIZE"] * 1024.0
        return {"fetch_bytes": fetch, "write_bytes": write, "bytes": fetch + write,
                "l2_hit_rate": out["TCC_HIT_sum"] / max(out["TCC_HIT_sum"] + out["TCC_MISS_sum"], 1.0),
                "source": "in-run: rocprofv3 --kernel-trace --pmc {FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum}, one "
                          "pass each, 4 launches of this graph; FETCH_SIZE x2 (gfx950 wide-read correction), per launch"}
    except Exception as ex:   # the headline number must not be lost to the profiler
        log("in-run PMC collection failed: %r" % (ex,))
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def measure_mfma(n_items):
    """MFMA-pipe counters of the projection kernels IN THIS RUN (north_star: "rocprof MFMA utilisation reported"): the
    forward / dW / dX kernels of a 4096 -> 64 projection over `n_items` rows in a child process under
    `rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES` (one pass, its own run).
    MfmaUtil_busy_cu = MFMA busy cycles / (4 SIMDs x busy CU cycles); None when rocprofv3 is not usable here."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe) or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        return None
    tmp = tempfile.mkdtemp(prefix="mmrec_pmc_", dir="/tmp")
    try:
        d = os.path.join(tmp, "mfma")
        cmd = [exe, "--kernel-trace", "--pmc", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "--output-format", "csv",
               "-d", d, "-o", "pm", "--", sys.executable, os.path.abspath(__file__), "--pmc-child", "linear:%d" % n_items]
        r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=100)
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if r.returncode != 0 or not files:
            log("rocprofv3 MFMA pass failed (rc %d): %s" % (r.returncode, r.stderr[-300:]))
            return None
        per = {}          # kernel -> dispatch -> counter -> value (a dispatch has one row per counter and XCD: summed)
        for row in csv.DictReader(open(files[0])):
            k = row["Kernel_Name"]
            if "linear_" not in k and "slab_reduce" not in k and "f16x3" not in k and "gemm64_stream" not in k:
                continue
            short = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()
            cs = per.setdefault(short, {}).setdefault(row.get("Dispatch_Id", ""), {})
            cs[row["Counter_Name"]] = cs.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
        dur = {}
        for tf in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            for row in csv.DictReader(open(tf)):
                k = row["Kernel_Name"]
                if "linear_" in k or "slab_reduce" in k or "f16x3" in k or "gemm64_stream" in k:
                    short = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()
                    dur.setdefault(short, []).append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
        out = {}
        for k, disp in per.items():
            mf = np.mean([c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for c in disp.values()])
            cu = np.mean([c.get("SQ_BUSY_CU_CYCLES", 0.0) for c in disp.values()])
            if cu > 0 and mf > 0:
                out[k] = {"MfmaUtil_busy_cu": float(mf / (4.0 * cu)), "calls": len(disp),
                          "duration_us": float(np.mean(dur[k])) / 1e3 if k in dur else None}
        return out or None
    except Exception as ex:
        log("in-run MFMA counter collection failed: %r" % (ex,))
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def time_spmm(g, x, reps=20):
    from mmrec_amd import hip_ops
    y = torch.empty(g.n_rows, x.shape[1], device=x.device)
    for _ in range(3):
        hip_ops.spmm_raw(g, x, Y=y)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        hip_ops.spmm_raw(g, x, Y=y)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def dslice_projection(dev, g, n_nodes, ms64):
    """What the FEATURE-SLICED multi-GPU layout costs, measured on THIS one GPU (round-3 review, item 2): a rank of a P-GPU
    run owns 64 / P columns of every table and the whole graph, and nothing crosses xGMI inside the propagation -- so the
    time of ONE slice's layer here IS the P-GPU per-layer time of the layout, and ms(d = 64) / ms(d = 64 / P) its strong
    scaling.  Next to it: what the row-sharded layouts can reach at this size by their exchange volume."""
    gen = torch.Generator(device=dev).manual_seed(0)
    out = {"ms_per_layer_d64": ms64, "ms_per_layer_one_slice": {}, "implied_speedup": {}, "slice_frac_8d": {},
           "slice_frac_line_model": {}}
    for P in (2, 4, 8):
        d = 64 // P
        x = torch.rand(n_nodes, d, device=dev, generator=gen) - 0.5
        ms = time_spmm(g, x)
        out["ms_per_layer_one_slice"][str(d)] = ms
        out["implied_speedup"][str(P)] = ms64 / ms
        out["slice_frac_8d"][str(d)] = alg_bytes(g.nnz, g.n_rows, d) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        out["slice_frac_line_model"][str(d)] = alg_line_bytes(g.nnz, g.n_rows, d) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        del x
    out["row_sharded_predictions_at_8"] = {"allgather": "1.1-1.7 x (336 MB into every rank per layer over 7 xGMI links against "
                                                        "~0.1 ms of SpMM per rank)", "allreduce": "1.7-2.4 x (224 MB)"}
    out["what"] = ("one layer of the config-5 graph (20M nnz, 1.5M rows) on [N, 64 / P] slices through mmrec_spmm_csr_f32, on "
                   "this GPU; implied_speedup[P] = ms_per_layer_d64 / ms_per_layer_one_slice[64 / P]: the layout's strong "
                   "scaling at P GPUs (no exchange in the propagation; the slices are all-gathered once per evaluation). "
                   "slice_frac_line_model: against 128-B fabric requests per gathered row -- what bounds a slice "
                   "(profiles/r04_spmm_pmc.txt)")
    return out


def c5_train_step_roofline(dev, nu, ni, eu, ei):
    """The SpMM launches the config-5 FREEDOM step actually runs (freedom.py:128-143,164-177): the 80 %-pruned user-item
    graph (4M nnz over 1.5M rows: 2.7 per row; four launches per step, forward + backward of two layers) and the item-item
    kNN graph (10M nnz over 500K rows; two launches) -- ms per launch, SURVEY.md 8(d) bytes / time / 8 TB/s."""
    from mmrec_amd import hip_ops
    keep = torch.randperm(eu.shape[0], generator=torch.Generator().manual_seed(0))[:int(eu.shape[0] * 0.2)]
    pruned = hip_ops.bipartite_graph_from_edges(torch.from_numpy(eu)[keep].to(dev), torch.from_numpy(ei)[keep].to(dev), nu, ni)
    rng = np.random.default_rng(1)
    rows = np.repeat(np.arange(ni), 20)
    mm = hip_ops.CsrGraph.from_coo_host(np.stack([rows, rng.integers(0, ni, rows.shape[0])]),
                                        np.full(rows.shape[0], 0.05, np.float32), ni, ni, dev)
    gen = torch.Generator(device=dev).manual_seed(4)
    out = {}
    for name, g, n_x in (("pruned_user_item", pruned, nu + ni), ("item_item_knn", mm, ni)):
        x = torch.rand(n_x, 64, device=dev, generator=gen) - 0.5
        ms = time_spmm(g, x)
        ab = alg_bytes(g.nnz, g.n_rows)
        out[name] = {"nnz": g.nnz, "rows": g.n_rows, "nnz_per_row": g.nnz / g.n_rows, "ms_per_launch": ms,
                     "alg_bytes_per_launch": float(ab), "achieved": ab / (ms * 1e-3) / 1e9,
                     "frac": ab / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        del x
    out["what"] = ("mmrec_spmm_csr_f32, d = 64, on the two graphs of the config-5 training step; frac = (264 B x nnz + 260 B x "
                   "rows) / ms_per_launch / 8 TB/s (the headline's definition); counters: profiles/r04_spmm_pmc.txt")
    return out


def committed_traffic():
    pmc = os.path.join(ROOT, "profiles", "spmm_pmc.json")
    try:
        j = json.load(open(pmc))
        sha = subprocess.run(["git", "-C", ROOT, "log", "-1", "--format=%h", "--", "profiles/spmm_pmc.json"],
                             capture_output=True, text=True).stdout.strip() or "untracked"
        return {"fetch_bytes": j["fetch_bytes_corrected_x2_gfx950"], "write_bytes": j["write_bytes"],
                "bytes": j["hbm_bytes_per_launch"], "l2_hit_rate": j["l2_hit_rate"],
                "source": "profiles/spmm_pmc.json@%s (a committed profile of the same launch, NOT measured in this run)" % sha}
    except Exception:
        return None


def launch_ranks(args, argv):
    """`python bench.py --gpus N` without a torchrun environment: start the N ranks ourselves (the command line the driver
    contract names: torch.distributed.run, one process per GPU, rendezvous on 127.0.0.1) and pass rank 0's ONE JSON line
    through on stdout.  The reference has no launcher to mirror (src/main.py:16-27 is single-process)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // max(args.gpus, 1))))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    log("no torchrun environment: launching %d ranks: %s" % (args.gpus, " ".join(cmd)))
    return subprocess.run(cmd, env=env).returncode


def dry_run(args, rank, world):
    """--dry-run: everything of an N-rank run EXCEPT the device work -- process group (gloo), the nnz-balanced row cut
    of a small graph of the c5 kind, one collective, rank 0's one JSON line.  What tests/test_bench_launcher.py
    drives on a box without GPUs; no kernel runs and no number is reported."""
    import torch.distributed as dist
    from mmrec_amd import synth
    from mmrec_amd.dist import BipartiteSharding
    multi = world > 1
    saved = None
    if multi:
        sys.stdout.flush()
        saved = os.dup(1)                      # gloo / RCCL print banners on stdout: parked on stderr meanwhile
        os.dup2(2, 1)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    nu, ni = 3000, 1200
    eu, ei = synth.powerlaw_edges(nu, ni, 30000, seed=0)
    r, c, v = synth.sym_norm_coo(eu, ei, nu, ni)
    sh = BipartiteSharding.from_coo(r, nu, ni, world, n_chunks=args.chunks or 1)
    per_rank = sh.nnz_per_rank(r)
    seen = torch.zeros(world, dtype=torch.int64)
    seen[rank] = int(per_rank[rank])
    if multi:
        dist.all_reduce(seen)
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        os.dup2(saved, 1)
        os.close(saved)
    if rank == 0:
        print(json.dumps({"metric": "GCN-layer edges/sec (3-layer user-item CSR SpMM, d=64, fp32)", "value": None,
                          "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "dry_run": True, "ranks_in_process_group": dist.get_world_size() if multi else 1,
                          "extra": {"nnz_per_rank": [int(x) for x in seen.tolist()]}}), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="do not re-run launches under rocprofv3 for roofline.traffic")
    ap.add_argument("--no-configs", action="store_true", help="skip extra.configs_step_ms (Trainer-level epochs of VBPR and the five "
                                                              "north_star models at their BASELINE shapes, ~1 minute)")
    ap.add_argument("--headline-only", action="store_true",
                    help="the timed region and nothing else (no companions, no CPU baseline, no counters): what "
                         "`rocprofv3 --kernel-trace --stats -- python bench.py --headline-only` profiles, so that the summary's "
                         "average SpMM duration is the timed launches' (profiles/r03_bench_headline_kernel_stats.csv)")
    ap.add_argument("--pmc-child", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--force-dist", action="store_true",
                    help="testing aid: run the N > 1 code path (process group, sharded blocks, "
                         "collectives) with a single rank")
    ap.add_argument("--layout", choices=["dslice", "allgather", "allreduce"], default=None,
                    help="N > 1: 'dslice' (default for 2 / 4 / 8 ranks) = FEATURE-sliced: every rank owns 64 / N columns of "
                         "every table and the whole graph, Y[:, s] = A X[:, s] needs no exchange for any number of layers, "
                         "bit-exact vs 1 GPU; 'allgather' (north_star's wording) = rows sharded nnz-balanced, blocks "
                         "all-gathered per layer in chunks under the next chunk's SpMM, bit-exact vs 1 GPU; "
                         "'allreduce' = users sharded / items replicated, item partial sums all-reduced per "
                         "layer (2/3 of the all-gather volume, fp32-rounding-equal)")
    ap.add_argument("--chunks", type=int, default=None, help="row chunks per rank of the allgather layout (default: auto)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / process-group / sharding-plan self-test: no device work, no numbers (runs without a GPU)")
    args = ap.parse_args()
    if args.headline_only:
        args.no_extra = args.no_cpu_baseline = args.no_pmc = True
    if args.pmc_child:
        return pmc_child(args.pmc_child)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: start the N ranks (torchrun form keeps working: it sets WORLD_SIZE)
        sys.exit(launch_ranks(args, sys.argv[1:]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log("WORLD_SIZE %d != --gpus %d; using WORLD_SIZE" % (world, args.gpus))
    args.layout_given = args.layout is not None
    if args.layout is None:     # the single-layout fallback; at 2 / 4 / 8 ranks BOTH layouts are measured (run_layout below)
        args.layout = "dslice" if world in DSLICE_WORLDS else "allgather"
    if args.layout == "dslice" and world not in DSLICE_WORLDS:
        raise SystemExit("--layout dslice needs 1, 2, 4 or 8 ranks (64 columns / ranks = a slice width the kernel has)")
    if args.dry_run:
        return dry_run(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    multi = world > 1 or args.force_dist
    saved_stdout = None
    if multi:
        # RCCL prints a version banner on stdout when its first communicator comes up; the contract is
        # ONE JSON line on stdout, so stdout is parked on stderr until the warm-up collectives are done
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from mmrec_amd import hip_ops
    from mmrec_amd.dist import ShardedPropagator
    import types
    tele = GpuTelemetry(local_rank)
    parked = {"fd": saved_stdout}

    def fence():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    local_t = {}

    def timed_steps(fn, n):
        fence()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        local_t["own"] = time.perf_counter() - t0     # this rank's own clock, before it waits for the others
        fence()
        t = time.perf_counter() - t0
        if multi:
            tt = torch.tensor([t], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t = float(tt.item())
        return t

    def run_layout(layout):
        """One layout of the headline workload, start to finish: blocks of the graph for this rank, W warm-up steps, EXACTLY K timed
        steps between fences (max over ranks), HIP events around every SpMM call, every rank's own view gathered, and -- for the
        row-sharded all-gather layout -- what the exchange costs per layer and how much of it hides under the SpMMs."""
        L = types.SimpleNamespace(layout=layout, ev=[], timed=False, prop=None)
        L.sh, L.g, L.ublk, L.iblk, L.r, L.c, L.v = build_c5(dev, rank, world, layout, multi, args.chunks)
        sh, g, ublk, iblk = L.sh, L.g, L.ublk, L.iblk
        L.nnz_total, L.n_nodes = int(L.r.shape[0]), sh.n_users + sh.n_items
        gen = torch.Generator(device=dev).manual_seed(0)   # same seed -> same X0 on every rank
        L.dslice = dslice = multi and layout == "dslice"
        L.d_loc = d_loc = 64 // world if dslice else 64
        X0 = torch.rand(sh.N_pad if (multi and not dslice) else L.n_nodes, 64, device=dev, generator=gen) - 0.5
        if dslice and d_loc < 64:                           # this rank's columns of the same table
            X0 = X0[:, rank * d_loc:(rank + 1) * d_loc].contiguous()
        L.X0, L.bufs = X0, [torch.empty_like(X0), torch.empty_like(X0)]
        bufs = L.bufs

        def local_spmm(block, X, Y, **ep):
            if L.timed:
                s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s_.record()
                hip_ops.spmm_raw(block, X, Y=Y, **ep)
                e_.record()
                L.ev.append((s_, e_, block.nnz, block.n_rows, X.shape[1]))
            else:
                hip_ops.spmm_raw(block, X, Y=Y, **ep)
        L.local_spmm = local_spmm

        if not multi or dslice:      # dslice: the same three launches on this rank's columns -- nothing crosses xGMI
            def step():
                cur = X0
                for layer in range(N_LAYERS):
                    nxt = bufs[layer % 2]
                    local_spmm(g, cur, nxt)
                    cur = nxt
        elif layout == "allreduce":
            from mmrec_amd.dist import ItemReplicatedPropagator
            ubk = -(-sh.n_users // world)
            u0, u1 = rank * ubk, min((rank + 1) * ubk, sh.n_users)
            Xu0 = X0[u0:u1].contiguous()                                  # X0 is in padded id space:
            Xi0 = X0[sh.U_pad:sh.U_pad + sh.n_items].contiguous()         # users first, items at U_pad
            L.prop = ItemReplicatedPropagator(ublk, iblk, local_spmm, world_size=world, force_collectives=multi)

            def step():
                L.prop.propagate(Xu0, Xi0, N_LAYERS)
        else:
            L.prop = ShardedPropagator(sh, ublk, iblk, rank, local_spmm, force_collectives=multi)

            def step():
                L.prop.propagate(X0, N_LAYERS, bufs=bufs)     # ping-pong: only the last layer is kept, as a model would
        L.step = step

        for _ in range(args.warmup):
            step()
        fence()
        if parked["fd"] is not None:
            sys.stdout.flush()
            ctypes.CDLL(None).fflush(None)  # RCCL printf()s into libc's buffer: drain it while fd 1 is parked
            os.dup2(parked["fd"], 1)
            os.close(parked["fd"])
            parked["fd"] = None
        tele_start = tele.read()                      # right after the warm-up steps' fence
        L.timed = True
        L.dt = timed_steps(step, args.steps)
        L.timed = False
        tele_end = tele.read()                        # right after the closing fence
        own_s = local_t["own"]
        # ... and UNDER LOAD, outside the timed region: ~0.15 s of the same launches are enqueued (asynchronous), the sensors
        # are read while the GPU is still working through them -- the clocks the timed launches actually ran at
        for _ in range(max(1, min(200, int(0.15 / max(L.dt / args.steps, 1e-4))))):
            step()
        time.sleep(0.05)
        tele_load = tele.read()
        fence()
        L.telemetry = {"start_of_timed_region": tele_start, "end_of_timed_region": tele_end, "under_load_after": tele_load,
                       "source": "librocm_smi64 in-process (rsmi_dev_gpu_clk_freq_get SYS / MEM, socket power, junction "
                                 "temperature), GPU index %d" % local_rank}
        L.timed_events = list(L.ev)
        L.per_rank = None
        L.exchange = None
        if multi:
            # every rank's own view, gathered NOW (the emitting code may run on a watchdog thread: no collectives there):
            # own wall time per step, mean SpMM call duration, gather-model GB/s of its calls
            ms = np.array([s_.elapsed_time(e_) for s_, e_, _, _, _ in L.timed_events])
            ab = np.array([alg_bytes(nz, nr, dd) for _, _, nz, nr, dd in L.timed_events], dtype=np.float64)
            mine = torch.tensor([own_s / args.steps * 1e3, float(ms.mean()), float(ab.sum() / (ms.sum() * 1e-3) / 1e9),
                                 float(len(ms)), float(ms.sum() / args.steps)], device=dev, dtype=torch.float64)
            allr = torch.empty(world * 5, device=dev, dtype=torch.float64)
            dist.all_gather_into_tensor(allr, mine)
            allr = allr.view(world, 5).cpu().numpy()
            L.per_rank = {"ranks_in_process_group": dist.get_world_size(),
                          "ms_per_step": [float(x) for x in allr[:, 0]],
                          "spmm_ms_per_call": [float(x) for x in allr[:, 1]],
                          "spmm_ms_per_step": [float(x) for x in allr[:, 4]],            # compute: sum of this rank's SpMM calls
                          "spmm_ms_per_layer": [float(x) / N_LAYERS for x in allr[:, 4]],
                          "spmm_gather_model_gbs": [float(x) for x in allr[:, 2]],
                          "spmm_frac_gather_model": [float(x) / HBM_PEAK_GBS for x in allr[:, 2]],
                          "spmm_calls_timed": [int(x) for x in allr[:, 3]]}
        if multi and layout == "allgather":
            per_rank_nnz = sh.nnz_per_rank(L.r)
            ex = {"nnz_per_rank": [int(x) for x in per_rank_nnz],
                  "nnz_imbalance_max_over_mean": float(per_rank_nnz.max() / per_rank_nnz.mean()),
                  "chunks_per_rank": sh.n_chunks}
            b0 = L.prop.op.bytes_gathered
            step()
            fence()
            inbound = L.prop.op.bytes_gathered - b0                     # payload bytes this rank received per step
            compute_only = ShardedPropagator(sh, ublk, iblk, rank, local_spmm)
            compute_only.op.exchange = False
            t_comp = timed_steps(lambda: compute_only.propagate(X0, N_LAYERS, bufs=bufs), args.steps) / args.steps

            def comm_only():
                for layer in range(N_LAYERS):
                    works = []
                    for _, lo, hi, rlo, rhi in L.prop.op.entries:
                        works.append(dist.all_gather_into_tensor(bufs[layer % 2][rlo:rhi], bufs[layer % 2][lo:hi],
                                                                 async_op=True))
                    for w in works:
                        w.wait()
            comm_only()
            t_comm = timed_steps(comm_only, args.steps) / args.steps
            t_tot = L.dt / args.steps
            ex.update({
                "exchange_bytes_in_per_rank_per_step": int(inbound),
                "ms_per_step_compute_only": t_comp * 1e3, "ms_per_step_exchange_only": t_comm * 1e3,
                "ms_per_layer_compute_only": t_comp * 1e3 / N_LAYERS, "ms_per_layer_exchange_only": t_comm * 1e3 / N_LAYERS,
                "xgmi_gbps_achieved": inbound / t_tot / 1e9,            # per GPU, inbound, inside the timed step
                "xgmi_gbps_exchange_only": inbound / t_comm / 1e9,      # the same all-gathers with nothing else running
                "overlap_frac": max(0.0, min(1.0, (t_comp + t_comm - t_tot) / max(min(t_comp, t_comm), 1e-9))),
                "note_exchange": "per-GPU inbound payload of the per-layer all-gathers (fp32 rows); overlap_frac = share of "
                                 "the shorter of {SpMMs, exchange} that ran hidden under the other"})
            L.exchange = ex
        if dslice:
            L.exchange = {"columns_per_rank": d_loc, "exchange_bytes_in_per_rank_per_step": 0,
                          "ms_per_layer_exchange_only": 0.0,
                          "note_exchange": "feature-sliced: every rank runs the whole graph on its own %d columns; no collective "
                                           "inside the propagation (the slices meet once per evaluation: c5_full_eval)" % d_loc}
        L.summary = {"edges_per_s": L.nnz_total * N_LAYERS * args.steps / L.dt, "ms_per_step": L.dt / args.steps * 1e3,
                     "per_rank": L.per_rank, "exchange": L.exchange}
        return L

    # N > 1 (and no --layout): BOTH layouts in this one invocation -- the feature-sliced layout that needs no exchange, and
    # north_star's row-wise partition with an RCCL all-gather of the embedding blocks after each layer -- each with the full
    # warm-up + K timed steps; `value` is the faster one's (named in config.parallelism), both ride in `layouts`.  The
    # no-collective layout runs FIRST and a watchdog guards the second: chunked RCCL all-gathers under SpMMs have only ever run on
    # a one-rank group here, and a hang there must not cost the line.
    both = multi and not args.layout_given and world in DSLICE_WORLDS
    state = {"dist_extra": None, "c5_eval": None, "done": False, "per_rank": None, "telemetry": None, "layouts": None}
    runs = [run_layout("dslice" if both else args.layout)]

    def bind(L):
        state["per_rank"], state["telemetry"] = L.per_rank, L.telemetry
        state["layouts"] = {x.layout: x.summary for x in runs} if multi else None
        args.layout = L.layout
        return L
    best = bind(runs[0])
    if both:
        budget2 = float(os.environ.get("MMREC_BENCH_SECOND_LAYOUT_BUDGET_S", "300"))

        def give_up_second():
            if state["done"]:
                return
            state["done"] = True
            log("the all-gather layout did not finish in %.0f s: emitting the feature-sliced line alone" % budget2)
            try:
                state["layouts"]["allgather"] = {"error": "did not finish in %.0f s" % budget2}
                L = runs[0]
                emit_line(args, rank, world, multi, L.sh, L.g, L.r, L.c, L.v, dev, L.n_nodes, L.nnz_total, L.dt, L.timed_events, state)
            finally:
                os._exit(0)
        wd2 = threading.Timer(budget2, give_up_second)
        wd2.daemon = True
        wd2.start()
        failed = None
        try:
            runs.append(run_layout("allgather"))
        except Exception as ex:            # same code on every rank -> same exception on every rank
            log("all-gather layout failed: %r" % (ex,))
            failed = {"error": repr(ex)}
        wd2.cancel()
        if state["done"]:
            return
        # `value` is north_star's layout -- rows sharded, RCCL all-gather per layer -- whenever it finished (round-5 review 6);
        # the feature-sliced run is reported beside it under `layouts`, whichever is faster
        best = bind(next((L for L in runs if L.layout == "allgather"), runs[0]))
        if failed:
            state["layouts"]["allgather"] = failed
    sh, g, ublk, iblk, r, c, v = best.sh, best.g, best.ublk, best.iblk, best.r, best.c, best.v
    nnz_total, n_nodes, dt, timed_events = best.nnz_total, best.n_nodes, best.dt, best.timed_events
    X0, bufs, prop, step, dslice, d_loc, local_spmm = best.X0, best.bufs, best.prop, best.step, best.dslice, best.d_loc, best.local_spmm
    for L in runs:
        if L is not best:
            L.__dict__.clear()            # the other layout's blocks and buffers
    del runs
    if multi:
        torch.cuda.empty_cache()

    def emit():
        emit_line(args, rank, world, multi, sh, g, r, c, v, dev, n_nodes, nnz_total, dt, timed_events, state)
    watchdog = None
    if multi:
        # the headline is measured; the companions below run collectives this container could only exercise on a
        # one-rank group and gloo.  A rank stuck in one of them must not cost the line: after the budget every rank's
        # watchdog emits what exists (rank 0 prints) and ends the process
        budget = float(os.environ.get("MMREC_BENCH_COMPANION_BUDGET_S", "420"))

        def give_up():
            if state["done"]:
                return
            state["done"] = True
            log("companions exceeded %.0f s: emitting the headline without them" % budget)
            try:
                state["dist_extra"] = dict(state["dist_extra"] or {}, companions="stopped after %.0f s" % budget)
                emit()
            finally:
                os._exit(0)
        watchdog = threading.Timer(budget, give_up)
        watchdog.daemon = True
        watchdog.start()

    # N > 1: what the exchange costs and how much of it hides under the SpMMs (allgather layout), the nnz balance
    # of the cut, the no-exchange alternative (one full replica per GPU), a weak-scaling companion and a sharded
    # training step -- all AFTER the timed region
    dist_extra = None
    if multi:
        dist_extra = state["dist_extra"] = {}
        dist_extra.update(best.exchange or {})
        if dslice:
            full = g
            xa = torch.rand(n_nodes, 64, device=dev, generator=torch.Generator(device=dev).manual_seed(0)) - 0.5
        else:
            full = hip_ops.CsrGraph.from_coo_device(
                torch.from_numpy(r.astype(np.int32)).to(dev), torch.from_numpy(c.astype(np.int32)).to(dev),
                torch.from_numpy(v).to(dev), n_nodes, n_nodes, symmetric=True)
            xa = X0[:n_nodes].contiguous()
        xb, xc = torch.empty_like(xa), torch.empty_like(xa)

        def rep_step():
            cur, nxts = xa, (xb, xc)
            for layer in range(N_LAYERS):
                hip_ops.spmm_raw(full, cur, Y=nxts[layer % 2])
                cur = nxts[layer % 2]
        for _ in range(2):
            rep_step()
        dist_extra["replicas_edges_per_s"] = world * nnz_total * N_LAYERS * args.steps / timed_steps(rep_step, args.steps)
        del full, xa, xb, xc
        # companions: a failure here (same code on every rank -> same exception on every rank) must not
        # cost the headline number
        try:
            dist_extra["weak_scaling"] = weak_scaling_run(dev, rank, world, args.steps)
        except Exception as ex:
            dist_extra["weak_scaling"] = {"error": repr(ex)}
        if args.layout == "allreduce":
            try:
                dist_extra["sharded_train_step"] = sharded_train_run(dev, rank, world, sh, ublk, iblk, args.steps)
            except Exception as ex:
                dist_extra["sharded_train_step"] = {"error": repr(ex)}
        dist_extra["note"] = ("replicas = every GPU propagates its own full copy of the graph (how MMRec uses several "
                              "GPUs: independent hyper-parameter runs); `value` is the sharded layout named in "
                              "config.parallelism")

    # companion at every N: full-sort evaluation of all 1M users at config-5 size, users sharded over the ranks
    c5_eval = None
    try:
        if args.headline_only:
            raise RuntimeError("--headline-only")
        ne = nnz_total // 2                       # sym_norm_coo: the first half are the user rows, sorted by (user, item)
        eu_all, ei_all = r[:ne], c[:ne] - sh.n_users
        E = bufs[(N_LAYERS - 1) % 2] if (not multi or args.layout in ("allgather", "dslice")) else None
        if dslice and world > 1:
            # the ONE exchange of the feature-sliced layout: the ranks' column slices of the final tables -> the full rows the
            # ranking needs (an all-gather of [N, 64 / P] blocks, then columns interleaved back; once per evaluation)
            fence()
            t_x = time.perf_counter()
            parts = torch.empty(world * n_nodes, d_loc, device=dev)                  # rank-major row blocks
            dist.all_gather_into_tensor(parts, E)
            E = parts.view(world, n_nodes, d_loc).permute(1, 0, 2).reshape(n_nodes, 64).contiguous()
            del parts
            fence()
            dist_extra["eval_table_allgather_ms"] = (time.perf_counter() - t_x) * 1e3
            dist_extra["eval_table_allgather_bytes_in_per_rank"] = int((world - 1) * n_nodes * d_loc * 4)
        if E is None:
            ge = torch.Generator(device=dev).manual_seed(1)
            Ue = torch.rand(sh.n_users, 64, device=dev, generator=ge) - 0.5
            Ie = torch.rand(sh.n_items, 64, device=dev, generator=ge) - 0.5
        elif multi and not dslice:
            Ue, Ie = (t.contiguous() for t in sh.unpad(E))
        else:
            Ue, Ie = E[:sh.n_users].contiguous(), E[sh.n_users:].contiguous()
        t_eval, warm_eval = c5_full_eval(dev, rank, world, Ue, Ie, eu_all, ei_all, sh.n_users, fence)
        if multi:
            ts = [t_eval, warm_eval.get("seconds_same_tables", 0.0), warm_eval.get("seconds_lists_of_moved_tables", 0.0)]
            tt = torch.tensor(ts, device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t_eval = float(tt[0].item())
            for j, key in ((1, "seconds_same_tables"), (2, "seconds_lists_of_moved_tables")):
                if key in warm_eval:
                    warm_eval[key] = float(tt[j].item())
        for key in ("same_tables", "lists_of_moved_tables"):
            if "seconds_" + key in warm_eval:
                warm_eval["users_per_s_" + key] = sh.n_users / warm_eval["seconds_" + key]
        flop = 2.0 * sh.n_users * sh.n_items * 64
        c5_eval = {"users_per_s": sh.n_users / t_eval, "seconds": t_eval,
                   "useful_tflops": flop / t_eval / 1e12,            # one exact score per (user, item) pair
                   # what the filter EXECUTES on the matrix cores: pass 2 over every candidate + pass 1 over every second
                   # 64-candidate stage (>= 131,072 candidates: topk_filter.hip filter_plan) = 1.5 x the useful products
                   "executed_tflops_f16": 1.5 * flop / t_eval / 1e12,
                   "frac_mfma_f16_executed": 1.5 * flop / t_eval / 1e12 / MFMA_F16_PEAK_TF,
                   # the round-2 review's yardstick (two full passes' worth of products over the call), kept for comparison
                   "frac_mfma_f16_two_passes": 2 * flop / t_eval / 1e12 / MFMA_F16_PEAK_TF,
                   "warm": warm_eval,

                   "what": "score + mask + top-50 of all %d users x %d items (the propagated embeddings of the timed step), "
                           "users sharded x%d, item table replicated, no exchange" % (sh.n_users, sh.n_items, world)}
        del Ue, Ie
    except Exception as ex:
        c5_eval = {"error": repr(ex)}
    state["c5_eval"] = c5_eval
    if watchdog is not None:
        watchdog.cancel()
    if not state["done"]:
        state["done"] = True
        emit()
    if multi:
        dist.barrier()
        dist.destroy_process_group()


def emit_line(args, rank, world, multi, sh, g, r, c, v, dev, n_nodes, nnz_total, dt, ev, state):
    """rank 0: the ONE JSON line (headline + roofline + companions gathered so far)"""
    dist_extra, c5_eval = state["dist_extra"], state["c5_eval"]
    # roofline of the dominant kernel family (one SpMM call), from the events of this rank
    call_ms = np.array([s.elapsed_time(e) for s, e, _, _, _ in ev])
    call_alg = np.array([alg_bytes(nz, nr, dd) for _, _, nz, nr, dd in ev], dtype=np.float64)
    call_min = np.array([alg_compulsory_bytes(nz, nr, dd) for _, _, nz, nr, dd in ev], dtype=np.float64)
    call_line = np.array([alg_line_bytes(nz, nr, dd) for _, _, nz, nr, dd in ev], dtype=np.float64)
    ms_launch = float(call_ms.mean())
    alg_gbs = float(call_alg.sum() / (call_ms.sum() * 1e-3) / 1e9)
    traffic = None
    if rank == 0 and not multi:
        traffic = (None if args.no_pmc else measure_traffic(g, n_nodes)) or committed_traffic()
    d_call = int(ev[0][4]) if ev else 64
    # ONE definition of the headline fraction: SURVEY.md 8(d) -- achieved = algorithmic (gather-model) bytes per launch / mean
    # launch time, frac = achieved / 8 TB/s.  The counter view (bytes that really cross L2, <= what the fabric can carry)
    # rides along under its own names.
    # Key ORDER matters: the driver's record keeps the first two dozen keys of this object.  One gather-model figure (`frac`),
    # then the counter view -- bytes that really cross L2 -> fabric per launch (`traffic`), the <= 1 `frac_physical`, and the
    # two ratios that say whether bytes are wasted -- then the times and sizes they come from.
    roofline = {"bound": "hbm", "achieved": alg_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg_gbs / HBM_PEAK_GBS}
    if traffic is not None:      # what the launch really moves past L2 (<= the fabric can carry: <= 1)
        roofline.update({"traffic": traffic["bytes"],
                         # the physical (<= 1) fraction: bytes that crossed L2 -> fabric per launch / launch time / 8 TB/s.  It
                         # still contains reads served by the 256-MiB Infinity Cache (MALL): an HBM-side byte count is not
                         # reachable through rocprofv3 on this stack (no MALL / UMC counter in `rocprofv3 --list-avail` for
                         # gfx950: DESIGN.md 5), so HBM bytes proper are <= this
                         "frac_physical": traffic["bytes"] / (ms_launch * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "traffic_over_compulsory": traffic["bytes"] / float(call_min.mean()),
                         "traffic_over_algorithmic": traffic["bytes"] / float(call_alg.mean()),
                         "l2_hit_rate": traffic["l2_hit_rate"]})
    else:                        # N > 1 ranks / no profiler and no committed profile
        roofline.update({"traffic": None, "frac_physical": None})
    roofline.update({"ms_per_launch": ms_launch, "launches_timed": int(len(ev)),
                     "alg_bytes_per_launch": float(call_alg.mean()), "compulsory_bytes_per_launch": float(call_min.mean()),
                     "frac_compulsory": float(call_min.sum() / (call_ms.sum() * 1e-3) / 1e9) / HBM_PEAK_GBS,
                     "kernel": ("spmm_rows_kernel<1,false> + spmm_long_reduce_kernel (mmrec_spmm_csr_f32)" if d_call == 64 else
                                "spmm_narrow_rows_kernel on a %d-column slice + long-row reduce" % d_call),
                     "definition": "frac = SURVEY 8(d) GATHER MODEL bytes/launch / launch time / 8 TB/s (>1 possible: caches "
                                   "absorb re-gathers); frac_physical = counter bytes, <= 1"})
    if traffic is not None:
        roofline.update({"traffic_fetch_bytes": traffic["fetch_bytes"], "traffic_write_bytes": traffic["write_bytes"],
                         "traffic_source": traffic["source"]})
    if d_call != 64:             # what the fabric has to move for random gathers: every L2 miss is a 128-B request, whatever the slice width
        roofline.update({"line_model_bytes_per_launch": float(call_line.mean()),
                         "frac_line_model": float(call_line.sum() / (call_ms.sum() * 1e-3) / 1e9) / HBM_PEAK_GBS})
    roofline["definition_long"] = ("achieved = ((8 + 4 d) B x nnz + (4 + 4 d) B x rows) per launch / mean launch duration (HIP events "
                                   "on the launch stream over the timed region), d = %d columns; traffic = (FETCH_SIZE x2 + WRITE_SIZE) "
                                   "per launch from rocprofv3 --pmc passes of the same launch (Infinity-Cache hits included)" % d_call)

    if rank == 0:
        line = {
            "metric": "GCN-layer edges/sec (3-layer user-item CSR SpMM, d=64, fp32)",
            "value": nnz_total * N_LAYERS * args.steps / dt, "unit": "edges/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "c5: synthetic 1M-user/500K-item/10M-edge graph (nnz 20M, N 1.5M), "
                                   "LightGCN-style 3-layer propagation, d=64",
                       "layers": N_LAYERS, "nnz": nnz_total, "rows": n_nodes,
                       "parallelism": "single GPU" if not multi else
                       ("feature-sliced x%d: every rank owns %d of the 64 columns of every table and the whole graph; no "
                        "exchange in the propagation" % (world, 64 // world) if args.layout == "dslice" else
                        "users sharded x%d, items replicated, RCCL all-reduce of item sums per layer" % world
                        if args.layout == "allreduce" else
                        "rows sharded x%d (nnz-balanced, %d chunks/rank), RCCL all-gather per layer" % (world, sh.n_chunks))},
            "roofline": roofline,
            "telemetry": state.get("telemetry"),
        }
        if not multi and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(r, c, v, n_nodes)
        if not multi and not args.no_extra:
            try:
                line["extra"] = extra_baby(dev)
                # the second half of BASELINE.json's metric ("... + full-eval users/sec, Amazon-Baby d=64")
                # BASELINE.json's metric is worded on Amazon-Baby ("GCN-layer edges/sec + full-eval users/sec, Amazon-Baby d=64"):
                # its two numbers at the TOP level (the graded roofline stays on the config-5 graph)
                ex = line["extra"]
                line["secondary"] = [
                    {"metric": "GCN-layer edges/sec, Amazon-Baby shape (237k nnz, cache resident), 3-layer propagation, d=64",
                     "value": ex["baby_propagate_edges_per_s"], "unit": "edges/s", "us_per_layer": ex["baby_us_per_layer"],
                     "us_per_layer_graph_replay": ex.get("baby_us_per_layer_graph_replay")},
                    {"metric": "full-eval users/sec (3-layer propagation + score + mask + top-50), Amazon-Baby shape, d=64",
                     "value": ex["baby_full_eval_users_per_s"], "unit": "users/s",
                     "graph_replay": ex.get("baby_full_eval_users_per_s_graph_replay"),
                     "warm_graph_replay": ex.get("baby_full_eval_users_per_s_warm_graph_replay")}]
                line["config"]["metric_config_amazon_baby"] = {
                    "edges_per_s": ex["baby_propagate_edges_per_s"], "us_per_layer": ex["baby_us_per_layer"],
                    "full_eval_users_per_s": ex["baby_full_eval_users_per_s"],
                    "score_topk_ms_cold": ex.get("baby_score_topk_ms"), "score_topk_ms_warm": ex.get("baby_score_topk_warm_ms")}
            except Exception as ex:  # the headline number must not be lost to an auxiliary failure
                line["extra"] = {"error": repr(ex)}
            if isinstance(line["extra"].get("cpu_baseline_full_eval"), dict):
                line["cpu_baseline_full_eval"] = line["extra"]["cpu_baseline_full_eval"]
                if "cpu_baseline" in line and "value" in line["cpu_baseline_full_eval"]:      # (the driver's record keeps this object)
                    line["cpu_baseline"]["full_eval_users_per_s"] = line["cpu_baseline_full_eval"]["value"]
                    line["cpu_baseline"]["full_eval_sample"] = line["cpu_baseline_full_eval"].get("sample")
            try:
                line["extra"]["c5_propagate_fwd_bwd"] = c5_fwd_bwd(dev, g, n_nodes, nnz_total)
            except Exception as ex:
                line["extra"]["c5_propagate_fwd_bwd"] = {"error": repr(ex)}
            try:
                line["extra"]["dslice_projection"] = dslice_projection(dev, g, n_nodes, ms_launch)
            except Exception as ex:
                line["extra"]["dslice_projection"] = {"error": repr(ex)}
            # the projection's MFMA counters, IN THIS RUN (a child process under rocprofv3 --pmc, its own pass)
            pr = line["extra"].get("projection_roofline") if isinstance(line["extra"], dict) else None
            if pr is not None:
                n_proj = line["extra"].pop("_projection_items", None)
                got = None if (args.no_pmc or not n_proj) else measure_mfma(n_proj)
                if got:
                    pr["mfma_util_counters"] = got
                    for kname, rec in got.items():     # dX: n x 4096 fp32 written once -> fraction of the 8 TB/s (write) roofline
                        if kname.startswith(("bwd_x_f16x3", "gemm64_stream")) and rec.get("duration_us"):
                            rec["write_gbs"] = 4.0 * n_proj * 4096 / (rec["duration_us"] * 1e-6) / 1e9
                            rec["frac_write_roofline"] = rec["write_gbs"] / HBM_PEAK_GBS
                        if kname.startswith(("linear_bwd_w_bf16x3", "linear_bwd_w_dma", "linear_fwd_dma")) and rec.get("duration_us"):
                            rec["x_stream_gbs"] = 4.0 * n_proj * 4096 / (rec["duration_us"] * 1e-6) / 1e9
                    pr["counters_source"] = ("in-run: rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES, one "
                                             "pass, 4 forward + backward calls at %d items; MfmaUtil_busy_cu = MFMA busy cycles / "
                                             "(4 SIMDs x busy CU cycles)" % n_proj)
                else:
                    try:
                        pj = json.load(open(os.path.join(ROOT, "profiles", "r03_mfma_pmc.json")))
                        pr["mfma_util_counters"] = {k: {"MfmaUtil_busy_cu": v.get("MfmaUtil_busy_cu"),
                                                        "duration_us": v.get("duration_ns", 0) / 1e3}
                                                    for k, v in pj.items() if v.get("MfmaUtil_busy_cu")}
                        pr["counters_source"] = "profiles/r03_mfma_pmc.json (a committed profile of these kernels, NOT measured in this run)"
                    except Exception:
                        pr["mfma_util_counters"] = None
            try:      # graphs WITH locality (round-3 review item 5): what a build-time relabelling buys
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import spmm_locality_probe
                gi = spmm_locality_probe.measure(dev, log=log)
                gi["what"] = ("a config-5-sized graph with PLANTED communities (2,000 users x 1,000 items, 90 % of the interactions "
                              "inside), one SpMM layer: ids grouped by community / randomly permuted (what a dataset's arbitrary ids "
                              "look like) / randomly permuted and relabelled at build time by hip_ops.PermutedGraph (same bits as "
                              "the plain graph); the benchmark's own graph has no structure to find")
                line["extra"]["c5_grouped_ids"] = gi
            except Exception as ex:
                line["extra"]["c5_grouped_ids"] = {"error": repr(ex)}
            try:
                ne = nnz_total // 2
                line["extra"]["c5_train_step_roofline"] = c5_train_step_roofline(dev, sh.n_users, sh.n_items, r[:ne],
                                                                                 c[:ne] - sh.n_users)
            except Exception as ex:
                line["extra"]["c5_train_step_roofline"] = {"error": repr(ex)}
            try:      # round 6: the row-lazy Adam's catch-up of a late config-5 step, exact replay against the opt-in closed form
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import prof_adam_catchup
                key = "exponential gaps, mean 122 (the late config-5 step)"
                got = prof_adam_catchup.measure(dev, shapes=(key,), log=log)[key]
                line["extra"]["lazy_adam_catchup"] = {
                    "exact_replay_us": got["exact_us"], "closed_form_us": got["closed_form_us"], "rows": 4096, "columns": 4096,
                    "optimizer_step": 683, "bytes_streamed": 4096 * 4096 * 24.0,
                    "closed_form_gbs": 4096 * 4096 * 24.0 / (got["closed_form_us"] * 1e-6) / 1e9,
                    "what": "mmrec_adam_rows_catchup_f32 (the default: dense Adam's bits) against mmrec_adam_rows_fastforward_f32 (config "
                            "lazy_adam_fast_forward, opt-in: as close to float64 Adam as the dense kernel, not bit-identical) on 4,096 "
                            "listed rows of a [*, 4096] table whose gaps since the last visit are drawn like a config-5 run's after "
                            "680 steps"}
            except Exception as ex:
                line["extra"]["lazy_adam_catchup"] = {"error": repr(ex)}
            if not args.no_configs:
                # round-5 review, next 2: Trainer-level ms per batch of VBPR and the five models north_star names, at their
                # BASELINE shapes, through the plugin API (tools/run_config.py; the committed profiles/rNN_run_configs.json of
                # consecutive rounds are compared by tests/test_host_logic.py::test_run_configs_did_not_regress)
                try:
                    sys.path.insert(0, os.path.join(ROOT, "tools"))
                    import contextlib
                    import run_config
                    with contextlib.redirect_stdout(sys.stderr):          # (stdout carries the ONE JSON line)
                        got = {n: run_config.run(n, None, 2, verbose=False) for n in run_config.TIER}
                    line["extra"]["configs_step_ms"] = {n: round(v["ms_per_batch"], 4) for n, v in got.items()}
                    line["extra"]["configs_eval_users_per_s"] = {n: round(v["eval_users_per_s"]) for n, v in got.items()}
                    line["extra"]["configs_what"] = {n: "%s / %s-shaped synthetic data, %s" % (v["model"], v["dataset"], v["step_mode"])
                                                     for n, v in got.items()}
                except Exception as ex:
                    line["extra"]["configs_step_ms"] = {"error": repr(ex)}
        if multi:
            line["extra"] = dist_extra
            line["per_rank"] = state.get("per_rank")
            line["layouts"] = state.get("layouts")        # both layouts' edges/s, per-rank compute and per-layer exchange times
            line["roofline"]["per_rank_frac_gather_model"] = (state.get("per_rank") or {}).get("spmm_frac_gather_model")
        line.setdefault("extra", {})["c5_full_eval"] = c5_eval
        if isinstance(c5_eval, dict) and "users_per_s" in c5_eval:
            w = c5_eval.get("warm") or {}
            line["config"]["c5_full_eval_users_per_s"] = {"cold": c5_eval["users_per_s"], "warm_same_tables": w.get("users_per_s_same_tables"),
                                                          "warm_lists_of_moved_tables": w.get("users_per_s_lists_of_moved_tables")}
            if isinstance(line.get("secondary"), list):
                line["secondary"].append({"metric": "full-eval users/sec, config-5 shape (1M users x 500K items, mask + top-50), "
                                                    "cold / warm", "value": c5_eval["users_per_s"], "unit": "users/s",
                                          "warm_same_tables": w.get("users_per_s_same_tables"),
                                          "warm_lists_of_moved_tables": w.get("users_per_s_lists_of_moved_tables")})
        if multi:                # round-5 review 6: what ran, at the top level
            pr = state.get("per_rank") or {}
            line["ranks_in_process_group"] = pr.get("ranks_in_process_group")
            lay = (state.get("layouts") or {}).get("allgather") or {}
            line["nnz_per_rank"] = ((lay.get("exchange") or {}).get("nnz_per_rank") if isinstance(lay, dict) else None)
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
