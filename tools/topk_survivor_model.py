#!/usr/bin/env python3
"""Host model of the fp16 top-K filter's survivor counts (topk_filter.hip: group maxima of pass 1 -> bound -> candidates within
2 eps of it) on the embeddings the bench ranks with, for pass-1 stage strides 1 and 2.  No GPU needed.

    python tools/topk_survivor_model.py c5        # E = A^3 X0 of the config-5 graph (what bench.py's c5_full_eval ranks)
    python tools/topk_survivor_model.py sports    # layer mean of a 2-layer propagation at Amazon-Sports shape
    python tools/topk_survivor_model.py c5 tiles  # round 4: could pass 2 SKIP tiles?  share of (queries x candidates) tiles that
                                                  # hold a survivor, and what a norm bound |q||c'| < thr would prune

Round 3: at config 5 ONE candidate of norm 2.5 (median 0.11) sets eps, so ~120 candidates per query survive at
stride 1 (63 without the margin) and ~245 at stride 2 -- more than the 256 slots the lists had for a third of the queries."""
import sys

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from mmrec_amd import synth  # noqa: E402


def clip_threshold(cn):
    """tau of filter_clip_kernel: histogram of the norms over float bits >> 20, lowest bin with <= 32 rows from it upwards."""
    bins = np.ascontiguousarray(cn, dtype=np.float32).view(np.uint32) >> 20
    hist = np.bincount(bins, minlength=2048)
    cum, b = 0, 2048
    for j in range(2047, -1, -1):
        if cum + hist[j] > 32:
            break
        cum, b = cum + hist[j], j
    return (np.array([b << 20], dtype=np.uint32).view(np.float32)[0] if cum > 0 else np.inf), cum


def survivor_stats(U, I, m_per_query, k=50, tag="", clip=True):
    """U [nq, d] sampled query rows, I [ni, d] candidates, m_per_query masked items per query: survivors of the filter's bound
    at pass-1 strides 1 and 2 (as main() does), with the clipping of outlying candidate rows the kernels apply from 131,072
    candidates on."""
    ni = I.shape[0]
    mean = I.mean(0)
    Ic = I - mean
    cn = np.linalg.norm(Ic, axis=1)
    cmax = cn.max()
    tau, n_out = clip_threshold(cn) if clip and ni >= 131072 else (np.inf, 0)
    out = cn >= tau
    if n_out:
        Ic = Ic * np.minimum(1.0, tau / np.maximum(cn, 1e-30))[:, None]
    print("%scentred candidate norms: median %.3g  p99 %.3g  p99.99 %.3g  max %.3g; clipped rows %d at tau %.3g" % (
        tag, np.median(cn), np.percentile(cn, 99), np.percentile(cn, 99.99), cmax, n_out, tau))
    n_stages = (ni + 63) // 64
    n_ranges = 16
    spr = (-(-n_stages // n_ranges) + 3) // 4 * 4
    cm = min(cmax, tau)
    for S in (1, 2):
        surv, plain, low = [], [], 0
        for q in range(U.shape[0]):
            s = Ic @ U[q]
            m, qn = int(m_per_query[q]), np.linalg.norm(U[q])
            eps = qn * (1.0e-3 * cm + 4e-6 * (cmax + np.linalg.norm(mean))) + 2.4e-7 * (qn + cm)
            gm = np.full(32 * n_ranges, -np.inf)
            for rg in range(n_ranges):
                st = np.arange(rg * spr, min((rg + 1) * spr, ni // 64))[::S]
                if len(st):
                    gm[rg * 32:(rg + 1) * 32] = s[st[:, None] * 64 + np.arange(64)[None, :]].reshape(len(st), 2, 32).max(axis=(0, 1))
            bound = np.sort(gm)[::-1][k + m - 1]
            low += bound < eps
            surv.append(int(((s >= bound - 2 * eps) | out).sum()))
            plain.append(int((s >= bound).sum()))
        surv, plain = np.array(surv), np.array(plain)
        print("%sstride %d: survivors median %d p90 %d p99 %d max %d (without the 2 eps margin: median %d); > 256: %.3f  > 512: %.3f; "
              "bound < eps (slow queue when rows are clipped): %.3f" % (
                  tag, S, np.median(surv), np.percentile(surv, 90), np.percentile(surv, 99), surv.max(), np.median(plain),
                  (surv > 256).mean(), (surv > 512).mean(), low / U.shape[0]))


def warm_survivor_stats(U, I, U_old, I_old, mask_lists, k=50, tag=""):
    """Round 6: survivors of a WARM call (mmrec_score_topk_hinted_f32).  The threshold of query q is B = the smallest
    approximate score, under the CURRENT tables (U, I), of the k unmasked ids that ranked top under the tables of the PREVIOUS
    evaluation (U_old, I_old); survivors = candidates within 2 eps below B (+ the clipped rows).  U_old is U for the TEST pass
    after the VALID pass.  mask_lists[q]: the query's masked ids.  Prints the same line as survivor_stats for comparison."""
    ni = I.shape[0]
    mean = I.mean(0)
    Ic = I - mean
    cn = np.linalg.norm(Ic, axis=1)
    cmax = cn.max()
    tau, n_out = clip_threshold(cn) if ni >= 131072 else (np.inf, 0)
    out = cn >= tau
    if n_out:
        Ic = Ic * np.minimum(1.0, tau / np.maximum(cn, 1e-30))[:, None]
    cm = min(cmax, tau)
    surv, plain, low, kept = [], [], 0, []
    for q in range(U.shape[0]):
        s_old = I_old @ U_old[q]
        s_old[mask_lists[q]] = -np.inf
        hint = np.argpartition(-s_old, k - 1)[:k]
        s = Ic @ U[q]
        qn = np.linalg.norm(U[q])
        eps = qn * (1.0e-3 * cm + 4e-6 * (cmax + np.linalg.norm(mean))) + 2.4e-7 * (qn + cm)
        B = s[hint].min()
        low += B < eps
        surv.append(int(((s >= B - 2 * eps) | out).sum()))
        plain.append(int((s >= B).sum()))
        s_m = s.copy()
        s_m[mask_lists[q]] = -np.inf
        kept.append(len(np.intersect1d(np.argpartition(-s_m, k - 1)[:k], hint)))
    surv, plain = np.array(surv), np.array(plain)
    print("%swarm: survivors median %d p90 %d p99 %d max %d (without the 2 eps margin: median %d); > 256: %.3f  > 512: %.3f  > 1024 "
          "(overflow queue's list): %.3f; B < eps (slow queue when rows are clipped): %.3f; ids of the old list still in the top-%d: "
          "median %d" % (tag, np.median(surv), np.percentile(surv, 90), np.percentile(surv, 99), surv.max(), np.median(plain),
                         (surv > 256).mean(), (surv > 512).mean(), (surv > 1024).mean(), low / U.shape[0], k, np.median(kept)))


def main(shape):
    nu, ni, eu, ei = synth.shaped_edges(shape, seed=0)
    r, c, v = synth.sym_norm_coo(eu, ei, nu, ni)
    A = sp.coo_matrix((v, (r, c)), shape=(nu + ni, nu + ni)).tocsr()
    rng = np.random.default_rng(0)
    if shape == "c5":
        E = rng.random((nu + ni, 64), dtype=np.float32) - 0.5
        for _ in range(3):
            E = A @ E
    else:
        b = np.sqrt(6.0 / (nu + 64)), np.sqrt(6.0 / (ni + 64))
        E0 = np.concatenate([rng.uniform(-b[0], b[0], (nu, 64)), rng.uniform(-b[1], b[1], (ni, 64))]).astype(np.float32)
        E, acc = E0, E0.copy()
        for _ in range(2):
            E = A @ E
            acc += E
        E = acc / 3
    U, I = E[:nu], E[nu:]
    mean = I.mean(0)
    Ic = I - mean
    cn = np.linalg.norm(Ic, axis=1)
    cmax = cn.max()
    print("centred candidate norms: median %.3g  p99 %.3g  max %.3g" % (np.median(cn), np.percentile(cn, 99), cmax))
    qs = rng.choice(nu, 200, replace=False)
    k = 50
    deg = np.bincount(eu, minlength=nu)
    n_stages = (ni + 63) // 64
    n_ranges = 16
    spr = (-(-n_stages // n_ranges) + 3) // 4 * 4
    for S in (1, 2):
        surv, plain, words = [], [], []
        for q in qs:
            s = Ic @ U[q]
            m, qn = deg[q], np.linalg.norm(U[q])
            eps = qn * (1.0e-3 * cmax + 4e-6 * (cmax + np.linalg.norm(mean))) + 2.4e-7 * (qn + cmax)
            gm = np.full(32 * n_ranges, -np.inf)
            for rg in range(n_ranges):
                st = np.arange(rg * spr, min((rg + 1) * spr, ni // 64))[::S]
                if len(st):
                    gm[rg * 32:(rg + 1) * 32] = s[st[:, None] * 64 + np.arange(64)[None, :]].reshape(len(st), 2, 32).max(axis=(0, 1))
            bound = np.sort(gm)[::-1][k + m - 1]
            keep = s >= bound - 2 * eps
            surv.append(int(keep.sum()))
            plain.append(int((s >= bound).sum()))
            words.append(len(np.unique(np.nonzero(keep)[0] // 64)))
        surv, plain, words = np.array(surv), np.array(plain), np.array(words)
        print("%s stride %d: survivors median %d p90 %d max %d (without the 2 eps margin: median %d); non-zero words median %d "
              "max %d; k + m median %d; > 256: %.3f  > 512: %.3f" % (
                  shape, S, np.median(surv), np.percentile(surv, 90), surv.max(), np.median(plain), np.median(words), words.max(),
                  np.median(k + deg[qs]), (surv > 256).mean(), (surv > 512).mean()))


def tiles(shape):
    """Round-3 review, lever (ii): let pass 2 skip the MFMA tiles without a survivor.  A tile can only be skipped for ALL the
    queries that share its products (32 per MFMA, 64 per wave, 256 per LDS tile); this prints the share of tiles that hold at
    least one candidate >= thr for blocks of consecutive queries, with thr from the stride-2 bound as the kernels compute it --
    and the share of candidates a Cauchy-Schwarz bound (|q| |c'| < thr, candidates sorted by norm) would let a block skip."""
    nu, ni, eu, ei = synth.shaped_edges(shape, seed=0)
    r, c, v = synth.sym_norm_coo(eu, ei, nu, ni)
    A = sp.coo_matrix((v, (r, c)), shape=(nu + ni, nu + ni)).tocsr()
    rng = np.random.default_rng(0)
    E = rng.random((nu + ni, 64), dtype=np.float32) - 0.5
    for _ in range(3):
        E = A @ E
    U, I = E[:nu], E[nu:]
    mean = I.mean(0)
    Ic = I - mean
    cn = np.linalg.norm(Ic, axis=1)
    tau, n_out = clip_threshold(cn) if ni >= 131072 else (np.inf, 0)
    cm = min(cn.max(), tau)
    Icc = Ic * np.minimum(1.0, tau / np.maximum(cn, 1e-30))[:, None]
    deg = np.bincount(eu, minlength=nu)
    k, n_ranges, S = 50, 16, 2 if ni >= 131072 else 1
    n_stages = ni // 64
    spr = (-(-((ni + 63) // 64) // n_ranges) + 3) // 4 * 4
    share = {(32, 32): [], (32, 64): [], (64, 64): [], (256, 64): []}
    prune, surv = {32: [], 64: [], 256: []}, []
    for b0 in rng.integers(0, nu - 256, 6):
        Sc = U[b0:b0 + 256] @ Icc.T                                            # [256, ni]
        keep = np.zeros((256, n_stages * 64), dtype=bool)
        rho = np.empty(256)
        for j in range(256):
            s, m, qn = Sc[j], int(deg[b0 + j]), np.linalg.norm(U[b0 + j])
            eps = qn * (1.0e-3 * cm + 4e-6 * (cn.max() + np.linalg.norm(mean))) + 2.4e-7 * (qn + cm)
            gm = np.full(32 * n_ranges, -np.inf)
            for rg in range(n_ranges):
                st = np.arange(rg * spr, min((rg + 1) * spr, n_stages))[::S]
                if len(st):
                    gm[rg * 32:(rg + 1) * 32] = s[st[:, None] * 64 + np.arange(64)[None, :]].reshape(len(st), 2, 32).max(axis=(0, 1))
            thr = np.sort(gm)[::-1][min(k + m, gm.size) - 1] - 2 * eps
            keep[j] = s[:n_stages * 64] >= thr
            rho[j] = thr / max(qn, 1e-30)
            surv.append(int(keep[j].sum()))
        for (tq, tc) in share:
            kk = keep.reshape(256 // tq, tq, n_stages * 64 // tc, tc).any(axis=(1, 3))
            share[(tq, tc)].append(kk.mean())
        for tq in prune:
            prune[tq].append(np.mean([(cn < rho[a:a + tq].min()).mean() for a in range(0, 256, tq)]))
    print("%s: survivors per query median %d; tiles with >= 1 survivor:" % (shape, np.median(surv)))
    for (tq, tc), x in share.items():
        print("   %3d queries x %2d candidates: %.3f" % (tq, tc, np.mean(x)))
    print("   candidates with |q||c'| < thr for EVERY query of a block (prunable by a norm bound): " +
          "  ".join("%d queries %.4f" % (tq, np.mean(x)) for tq, x in prune.items()))


if __name__ == "__main__":
    (tiles if len(sys.argv) > 2 and sys.argv[2] == "tiles" else main)(sys.argv[1] if len(sys.argv) > 1 else "c5")
