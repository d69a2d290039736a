#!/usr/bin/env python3
"""Round 6: the fused full-sort evaluation COLD (two matrix-core passes: group maxima -> bound -> pass / fail bits) against WARM
(mmrec_score_topk_hinted_f32: threshold from the previous lists, one pass) on one MI355X.

    python tools/prof_eval_warm.py [baby] [c5block] [c5all]

  baby     19,445 x 7,050 LightGCN-propagated tables, train positives masked, k = 50: hipGraph replays, median of 5 windows
  c5block  one 65,536-user block against the 500,000 items of the config-5 graph (3 propagation layers), k = 50
  c5all    all 1,000,000 users in 65,536-user blocks against one preparation of the item table (what bench.py reports)

For each: cold ms, warm ms with the lists of the SAME tables (the TEST pass after the VALID pass), warm ms with the lists of
tables whose every element moved by 2 / 10 / 30 % of its row's mean magnitude (later epochs), the slow / overflow queue lengths,
and bit-equality of ids with the cold call.  Run under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmrec_amd import hip_ops, synth  # noqa: E402

K = 50


def med_ms(fn, reps, windows=5, warm=2):
    for _ in range(warm):
        fn()
    per = []
    for _ in range(windows):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        per.append((time.perf_counter() - t0) / reps * 1e3)
    return float(np.median(per)), float(min(per)), float(max(per))


def replay_ms(fn, reps=50):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    return med_ms(g.replay, reps)


def moved(U, I, rel, seed):
    gen = torch.Generator(device=U.device).manual_seed(seed)
    Un = U + rel * U.abs().mean(1, keepdim=True) * torch.randn(U.shape, device=U.device, generator=gen)
    In = I + rel * I.abs().mean(1, keepdim=True) * torch.randn(I.shape, device=I.device, generator=gen)
    return Un, In


def propagated(shape, dev, layers):
    nu, ni, eu, ei = synth.shaped_edges(shape, seed=0)
    r, c, v = synth.sym_norm_coo(eu, ei, nu, ni)
    g = hip_ops.CsrGraph.from_coo_host(np.stack([r, c]), v, nu + ni, nu + ni, dev, symmetric=True)
    gen = torch.Generator(device=dev).manual_seed(0)
    E0 = (torch.rand(nu + ni, 64, device=dev, generator=gen) - 0.5) * 0.1
    E = hip_ops.lightgcn_mean(g, E0, layers)
    return nu, ni, eu, ei, E[:nu].contiguous(), E[nu:].contiguous()


def block_case(tag, U, I, rp, col, timer):
    cands = hip_ops.TopkCandidates(I)
    cold = hip_ops.score_topk(U, cands, K, rp, col)
    counts = torch.zeros(2, dtype=torch.int32, device=U.device)
    W = hip_ops.topk_hint_width(K)
    t = timer(lambda: hip_ops.score_topk(U, cands, K, rp, col))
    print("%s cold: %.3f ms (min %.3f max %.3f)  %.2f M users/s" % (tag, t[0], t[1], t[2], U.shape[0] / t[0] / 1e3), flush=True)
    for name, rel in (("same tables", 0.0), ("moved 2 %", 0.02), ("moved 10 %", 0.1), ("moved 30 %", 0.3)):
        # the lists a cold call over the (moved) tables leaves behind: its top-k + the runners-up it ranked
        hint = torch.full((U.shape[0], W), -1, dtype=torch.int32, device=U.device)
        Un, In = (U, I) if rel == 0.0 else moved(U, I, rel, 7)
        hip_ops.score_topk(Un, In if rel else cands, K, rp, col, hint=hint, hint_cold=True)
        del Un, In
        for width in ((W, K) if rel in (0.0, 0.1) else (W,)):      # with / without the runners-up
            h = hint if width == W else hint[:, :K].contiguous()
            counts.zero_()
            out = hip_ops.score_topk(U, cands, K, rp, col, hint=h, queue_counts=counts, hint_update=False)
            q = counts.tolist()
            same = bool(torch.equal(out, cold))
            listed = float((h >= 0).sum(1).float().mean())
            kept = (float((h[:, :K].long().unsqueeze(2) == cold.unsqueeze(1)).any(2).float().sum(1).mean())
                    if U.shape[0] <= 70000 else float("nan"))
            t = timer(lambda: hip_ops.score_topk(U, cands, K, rp, col, hint=h, hint_update=False))
            print("%s warm, lists of %s, %d wide (%.1f ids listed, %.1f of the old top-%d still ranked): %.3f ms (min %.3f max %.3f)  "
                  "%.2f M users/s; slow queue %d, overflow queue %d of %d; ids == cold: %s" %
                  (tag, name, width, listed, kept, K, t[0], t[1], t[2], U.shape[0] / t[0] / 1e3, q[0], q[1], U.shape[0], same), flush=True)
    hint = torch.full((U.shape[0], W), -1, dtype=torch.int32, device=U.device)
    t = timer(lambda: hip_ops.score_topk(U, cands, K, rp, col, hint=hint, hint_cold=True))
    print("%s cold through the hinted entry (writes the lists): %.3f ms" % (tag, t[0]), flush=True)
    t = timer(lambda: hip_ops.score_topk(U, cands, K, rp, col, hint=hint))
    print("%s warm, same tables, lists updated in place by every call: %.3f ms" % (tag, t[0]), flush=True)


def main():
    what = sys.argv[1:] or ["baby", "c5block", "c5all"]
    dev = torch.device("cuda:0")
    with torch.no_grad():
        if "baby" in what:
            nu, ni, eu, ei, U, I = propagated("baby", dev, 3)
            rp, col = hip_ops.mask_to_csr(np.stack([eu, ei]), nu, dev)
            block_case("[baby 19445 x 7050, hipGraph replay]", U, I, rp, col, lambda fn: replay_ms(fn))
            block_case("[baby 19445 x 7050, eager]", U, I, rp, col, lambda fn: med_ms(fn, 20))
        if "c5block" in what or "c5all" in what:
            nu, ni, eu, ei, U, I = propagated("c5", dev, 3)
            order = np.lexsort((ei, eu))
            eu, ei = eu[order], ei[order]
        if "c5block" in what:
            nq = 65536
            e = np.searchsorted(eu, nq, "left")
            rp, col = hip_ops.mask_to_csr(np.stack([eu[:e], ei[:e]]), nq, dev)
            block_case("[c5 block 65536 x 500000]", U[:nq].contiguous(), I, rp, col, lambda fn: med_ms(fn, 5, windows=3, warm=1))
        if "c5all" in what:
            rp, col = hip_ops.mask_to_csr(np.stack([eu, ei]), nu, dev)
            block_case("[c5 all 1000000 x 500000]", U, I, rp, col, lambda fn: med_ms(fn, 1, windows=3, warm=1))


if __name__ == "__main__":
    main()
