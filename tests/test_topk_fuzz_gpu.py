"""GPU: RANDOMIZED DIFFERENTIAL TEST of score + mask + top-K (a10 / a11; trainer.py:302-310) against the oracle.

The fp16-filter path (csrc/topk_filter.hip) has five result paths (final / overflow / slow / slow-split / merge), two list
capacities, two row widths, clipping and a subsampled first pass; the hand-written cases of tests/test_hip_parity.py found
neither of its two real bugs (subnormal query rows, overflowing lists) except by accident.  This file draws N_CASES seeded
cases over

    nq in [1, 3000], nc in [1, 300,000] (with extra weight on both sides of every plan switch: 4096 / 32,768 / 65,536 /
    131,072 candidates), k in [1, 128], kd in
    {64, 128} and, in a quarter of the cases, {192, 384} (the fp16 pass + exact refinement of csrc/topk_wide.h for k <= 32,
    the fp32 block path above),
    mask density (none / sparse / the train-positive shape / heavy users with 600 and 5,000 masked items, optionally the
    query's BEST candidates masked), per-row norm spread up to 2^+-40 on the queries and 2^+-12 on the candidates, a common
    component (what LightGCN smoothing produces), outlying candidate rows (x 5..25: trained item tables), all-zero query
    rows, duplicated candidate rows (exact ties: the lower id must win) and duplicated query rows,

and checks every case against `orc.mask_topk` (the reference's trainer step on torch-CPU fp32 scores) with the near-tie
rule: ids may differ from the oracle's only where the float64 score is within fp32 summation noise of the k-th score
(2e-6 |q| |c|, per candidate), values equal the float64 scores to the same noise, rows are sorted, free of duplicates and
of masked ids (which only fill a tail when fewer than k candidates are unmasked, at exactly -1e10 like the reference).
A failing case prints its seed: `pytest tests/test_topk_fuzz_gpu.py -k "seed17]"` re-runs it.

Every case the fp16 filter serves is then ranked three more times as a WARM call (mmrec_score_topk_hinted_f32: threshold from a
caller-supplied list, one matrix-core pass) with the cold call's own lists, a stale ranking read through `hint_rows`, and lists
full of unusable ids: ids and values must equal the cold call's bit for bit (`hint_variants`).

tests/test_topk_fuzz_cpu.py runs generator + checker on scaled-down cases against the torch-CPU stand-in op."""
import numpy as np
import pytest
import torch

from oracle import mmrec_oracle as orc

N_CASES = 200
WORK = 1.5e7         # nq * nc per case: the CPU oracle forms the [nq, nc] block (fp32 and float64)
SCALE = 1.0          # the CPU twin shrinks the candidate counts

_NC_EDGES = (4096, 32768, 65536, 131072)


def gen_case(seed, scale=None, work=None, meta_only=False):
    scale = SCALE if scale is None else scale
    work = WORK if work is None else work
    rng = np.random.default_rng(1_000_003 * seed + 17)
    kd = int(rng.choice([64, 128, 64, 128, 64, 128, 192, 384]))      # wide rows (csrc/topk_wide.h) in a quarter of the cases
    kind = int(rng.integers(0, 10))
    if kind == 0:
        nc = int(rng.integers(1, 200))
    elif kind == 1:
        nc = int(rng.integers(200, 4096))
    elif kind in (2, 3):                                  # right at a plan switch
        nc = int(rng.choice(_NC_EDGES)) + int(rng.integers(-70, 70))
    elif kind == 4:
        nc = int(rng.integers(4096, 32768))
    elif kind == 5:
        nc = int(rng.integers(32768, 65536))
    elif kind == 6:
        nc = int(rng.integers(65536, 131072))
    else:
        nc = int(rng.integers(131072, 300_001))
    nc = max(1, int(nc * scale))
    nq = int(rng.integers(1, 3001))
    if rng.random() < 0.15:
        nq = int(rng.choice([1, 31, 32, 33, 255, 256, 257, 511, 513]))
    nq = max(1, min(nq, int(work // nc)))
    kmax = min(nc, 128)          # every kd here is a multiple of 32: k <= 128 on all paths since ABI 9
    k = int(rng.choice([1, 5, 10, 20, 50, 64, 65, 100, 128])) if rng.random() < 0.6 else int(rng.integers(1, 129))
    k = max(1, min(k, kmax))
    if meta_only:
        return dict(seed=seed, k=k, kd=kd, nq=nq, nc=nc)
    # ---- embeddings
    common = rng.standard_normal(kd) * rng.choice([0.0, 0.05, 0.5])
    Q = (rng.standard_normal((nq, kd)) * 0.2 + common).astype(np.float32)
    C = (rng.standard_normal((nc, kd)) * 0.2 + common * rng.choice([1.0, -1.0, 0.0])).astype(np.float32)
    tags = []
    if rng.random() < 0.5:                                # query rows of very different norms (scaled row by row)
        Q *= np.exp2(rng.uniform(-40, 40, (nq, 1))).astype(np.float32)
        tags.append("qspread")
    if rng.random() < 0.3:                                # candidates of different norms (ONE scale serves them all)
        C *= np.exp2(rng.uniform(-12, 12, (nc, 1)) * rng.choice([0.25, 1.0])).astype(np.float32)
        tags.append("cspread")
    if rng.random() < 0.3:                                # a few rows of outlying norm (clipped from 131,072 candidates on)
        rows = rng.choice(nc, max(1, min(nc, int(rng.choice([1, 8, 40, max(1, nc // 1000)])))), replace=False)
        C[rows] *= rng.uniform(5, 25, (rows.shape[0], 1)).astype(np.float32)
        tags.append("outliers")
    if rng.random() < 0.2 and nq > 2:
        Q[rng.choice(nq, max(1, nq // 50), replace=False)] = 0.0
        tags.append("zeroq")
    if rng.random() < 0.35 and nc > 4:                    # duplicated candidates: exact ties
        n_src = int(rng.integers(1, 4))
        for _ in range(n_src):
            src = int(rng.integers(0, nc))
            dup = rng.choice(nc, min(nc - 1, int(rng.choice([2, 30, 300, 1500]))), replace=False)
            C[dup] = C[src]
            if rng.random() < 0.5 and nq > 1:             # ... that ARE somebody's best candidates
                Q[int(rng.integers(0, nq))] = C[src] * np.float32(rng.uniform(0.5, 2.0))
        tags.append("ties")
    if rng.random() < 0.2 and nq > 3:
        Q[rng.choice(nq, min(nq, 8), replace=False)] = Q[0]
        tags.append("dupq")
    # scores stay far inside (-1e10, 1e10): beyond the reference's mask sentinel its own ranking is an artefact
    bound = float(np.abs(Q).max()) * float(np.abs(C).max()) * kd
    if bound > 1e8:
        Q *= np.float32(np.exp2(-np.ceil(np.log2(bound / 1e8))))
    # ---- mask
    mode = int(rng.integers(0, 5))
    rows, cols = [np.zeros(0, np.int64)], [np.zeros(0, np.int64)]
    if mode >= 1:
        per = {1: 1, 2: 16, 3: 16, 4: 40}[mode]
        n = min(nq * per, 400_000)
        rows.append(rng.integers(0, nq, n)), cols.append(rng.integers(0, nc, n))
    if mode >= 3:                                         # heavy users
        for m in (600, 5000) if mode == 4 else (600,):
            q = int(rng.integers(0, nq))
            m = min(m, nc)
            rows.append(np.full(m, q)), cols.append(rng.choice(nc, m, replace=False))
        tags.append("heavy")
    if mode >= 2 and rng.random() < 0.5:                  # train positives score HIGH: mask some queries' best candidates
        qs = rng.choice(nq, min(nq, 16), replace=False)
        s = Q[qs].astype(np.float64) @ C.astype(np.float64).T
        nb = min(nc, int(rng.choice([3, 20, 100])))
        best = np.argpartition(-s, nb - 1, axis=1)[:, :nb]
        rows.append(np.repeat(qs, nb)), cols.append(best.reshape(-1))
        tags.append("bestmasked")
    if rng.random() < 0.1 and nq > 1:                     # fewer than k unmasked candidates for one query
        q = int(rng.integers(0, nq))
        keep = int(rng.integers(0, k + 1))
        if nc - keep <= 200_000:
            drop = rng.permutation(nc)[:nc - keep]
            rows.append(np.full(drop.shape[0], q)), cols.append(drop)
            tags.append("starved")
    key = np.unique(np.concatenate(rows).astype(np.int64) * nc + np.concatenate(cols).astype(np.int64))
    mask = np.stack([key // nc, key % nc])
    mask = mask[:, rng.permutation(mask.shape[1])]        # unsorted, as the loader hands it over
    return dict(seed=seed, Q=Q, C=C, k=k, kd=kd, nq=nq, nc=nc, mask=mask, tags=tags)


def describe(case):
    return "seed %d: nq %d nc %d k %d kd %d masked %d %s" % (case["seed"], case["nq"], case["nc"], case["k"], case["kd"],
                                                             case["mask"].shape[1], "+".join(case["tags"]))


def _check_row(what, r, got, vals, ref_row, s64_r, masked_r, qn_r, cn, k, has_val=True):
    """one query, any number of unmasked candidates (the general statement of the rules; the vectorised body below is the
    same rules for the rows with at least k unmasked candidates)"""
    nc = s64_r.shape[0]
    assert len(set(got.tolist())) == k, (what, r, "duplicate ids")
    n_real = min(k, int(nc - masked_r.sum()))
    head, tail = got[:n_real], got[n_real:]
    assert not masked_r[head].any(), (what, r, "masked id ranked")
    assert masked_r[tail].all() and np.all(vals[n_real:] == np.float64(np.float32(-1e10))), (what, r, "tail")
    assert not has_val or np.all(np.diff(vals) <= 0), (what, r, "not sorted")
    assert np.all(np.abs(vals[:n_real] - s64_r[head]) <= 2e-6 * qn_r * cn[head] + 1e-37), (what, r, "values")
    if n_real == 0:
        return
    tie = (vals[1:n_real] == vals[:n_real - 1]) & has_val
    assert np.all(got[1:n_real][tie] > got[:n_real - 1][tie]), (what, r, "tie order")
    s64m = np.where(masked_r, -np.inf, s64_r)
    order = np.argpartition(-s64m, n_real - 1)[:n_real]
    kth_c = order[np.argmin(s64m[order])]
    kth = s64m[kth_c]
    for c in set(head.tolist()) ^ set(ref_row[:n_real].tolist()):
        assert not masked_r[c], (what, r, c, "oracle ranked a masked id")
        assert abs(s64_r[c] - kth) <= 2e-6 * qn_r * (cn[c] + cn[kth_c]) + 1e-37, (what, r, c, s64_r[c], kth)
    wc = head[np.argmin(s64_r[head])]
    better = np.flatnonzero(s64m > s64_r[wc] + 2e-6 * qn_r * (cn + cn[wc]) + 1e-37)
    assert np.isin(better, head).all(), (what, r, "a better candidate was dropped")


def check_case(ops, dev, case):
    Q, C, k, mask = case["Q"], case["C"], case["k"], case["mask"]
    nq = Q.shape[0]
    rp, col = ops.mask_to_csr(mask, nq, dev)
    Qt, Ct = torch.from_numpy(Q), torch.from_numpy(C)
    Qd, Cd = Qt.to(dev), Ct.to(dev)
    idx, val = ops.score_topk(Qd, Cd, k, rp, col, return_values=True)
    check_lists(describe(case), idx.cpu(), val.cpu(), Qt, Ct, mask, k)
    check_warm_calls(ops, case, Qd, Cd, rp, col, idx, val)


def hint_variants(case, idx):
    """the lists a WARM call (mmrec_score_topk_hinted_f32) may be handed for this case, as (name, int32 [rows, hk], hint_rows):
    the cold call's own output (the TEST pass after the VALID pass), a STALE ranking (the lists of perturbed tables: an earlier
    epoch), and lists that bound nothing for some or all queries (-1 padding, out-of-range ids, duplicates, masked ids, random
    ids) -- the result must be the cold call's bit for bit in every case."""
    rng = np.random.default_rng(77 * case["seed"] + 5)
    Q, C, k, nq, nc, mask = case["Q"], case["C"], case["k"], case["nq"], case["nc"], case["mask"]
    out = [("own", idx.to(torch.int32).contiguous(), None)]
    # stale: rankings of tables that moved (relative noise 0.3 on both sides), unmasked -- some ids are masked / low now
    qs, cs = torch.from_numpy(Q), torch.from_numpy(C)
    qn = qs + 0.3 * qs.abs().mean(dim=1, keepdim=True) * torch.from_numpy(rng.standard_normal(Q.shape).astype(np.float32))
    cn = cs + 0.3 * cs.abs().mean(dim=1, keepdim=True) * torch.from_numpy(rng.standard_normal(C.shape).astype(np.float32))
    hk = int(min(128, nc, k + int(rng.integers(0, 9))))
    stale = torch.topk(torch.nan_to_num(qn @ cn.t()), hk, dim=1)[1].to(torch.int32)
    # ... stored in a table with more rows than queries, read through hint_rows
    table = torch.full((nq + 7, hk), -1, dtype=torch.int32)
    rows = torch.from_numpy(rng.permutation(nq + 7)[:nq].astype(np.int64))
    table[rows] = stale
    out.append(("stale+rows", table, rows))
    junk = idx.to(torch.int32).clone()
    sel = torch.from_numpy(rng.random((nq, k)) < 0.3)
    repl = torch.from_numpy(rng.integers(-3, nc + 3, (nq, k)).astype(np.int32))
    junk[sel] = repl[sel]
    if k > 1:
        dup = torch.from_numpy(rng.random(nq) < 0.3)
        junk[dup, -1] = junk[dup, 0]                       # a duplicate: the row has k - 1 usable ids
    if mask.shape[1]:
        m = torch.from_numpy(mask)
        pick = torch.from_numpy(rng.integers(0, mask.shape[1], min(64, mask.shape[1])))
        junk[m[0, pick], 0] = m[1, pick].to(torch.int32)   # a masked id heads the list
    junk[torch.from_numpy(rng.random(nq) < 0.1)] = -1      # users without a list
    out.append(("junk", junk, None))
    return out


def check_warm_calls(ops, case, Qd, Cd, rp, col, idx, val):
    if not hasattr(ops, "topk_hint_served") or not ops.topk_hint_served(case["nc"], case["kd"], case["k"]) or case["nc"] < case["k"]:
        return
    dev = Qd.device
    counts = torch.zeros(2, dtype=torch.int32, device=dev)
    cands = ops.TopkCandidates(Cd)
    k, nc = case["k"], case["nc"]
    for name, hint, rows in hint_variants(case, idx.cpu()):
        hd, rd = hint.to(dev), None if rows is None else rows.to(dev)
        widx, wval = ops.score_topk(Qd, cands if name != "junk" else Cd, k, rp, col, return_values=True,
                                    hint=hd, hint_rows=rd, queue_counts=counts)
        assert torch.equal(widx, idx), (describe(case), "warm call (%s hint): ids differ from the cold call" % name,
                                        torch.nonzero((widx != idx).any(1)).flatten()[:4].tolist())
        assert torch.equal(wval, val), (describe(case), "warm call (%s hint): values differ from the cold call" % name)
        # the call left ITS ranking in the rows: the top-k, then runners-up (distinct, in range) or -1
        left = (hd if rd is None else hd[rd]).long()
        assert torch.equal(left[:, :k], idx), (describe(case), name, "the call's lists do not start with its top-k")
        tail = left[:, k:]
        if tail.shape[1]:
            assert int(tail.min()) >= -1 and int(tail.max()) < nc
            full = torch.where(left >= 0, left, -1 - torch.arange(left.shape[1], device=dev)[None, :])     # (-1s made distinct)
            assert not (torch.sort(full, dim=1)[0].diff(dim=1) == 0).any(), (describe(case), name, "duplicate ids in a list")
    assert int(counts.min()) >= 0
    # the FIRST evaluation: a cold call through the same entry point writes the lists a warm call then reads (64 / 128 wide)
    wide = torch.full((case["nq"], 64 if k <= 64 else 128), -7, dtype=torch.int32, device=dev)
    cidx = ops.score_topk(Qd, cands, k, rp, col, hint=wide, hint_cold=True)
    assert torch.equal(cidx, idx) and torch.equal(wide[:, :k].long(), idx) and int(wide.min()) >= -1
    kept = wide.clone()
    widx = ops.score_topk(Qd, cands, k, rp, col, hint=wide, hint_update=False)
    assert torch.equal(widx, idx) and torch.equal(wide, kept), (describe(case), "warm call from the cold call's lists / KEEP flag")


def check_lists(what, idx, val, Qt, Ct, mask, k, s64=None):
    # the GPU box has 256 host cores: torch's CPU ops on these mid-sized blocks spend their time waking threads (83 min of user
    # time for an 8-minute suite), so the checker runs on a bounded pool
    threads = torch.get_num_threads()
    torch.set_num_threads(min(threads, 32))
    try:
        return _check_lists(what, idx, val, Qt, Ct, mask, k, s64)
    finally:
        torch.set_num_threads(threads)


def _check_lists(what, idx, val, Qt, Ct, mask, k, s64=None):
    """Device lists `idx` [nq, k] (+ optional values `val`) of the queries Qt against the candidates Ct under `mask` ([2, n],
    rows relative to Qt) vs the oracle's trainer step, with the rules of this file's header.  CPU tensors.  `s64`: the
    float64 score block when the caller has formed it already (wide rows, in chunks)."""
    nq, nc = Qt.shape[0], Ct.shape[0]
    idx = idx.cpu()
    has_val = val is not None
    assert tuple(idx.shape) == (nq, k) and int(idx.min()) >= 0 and int(idx.max()) < nc, what
    # float64 scores to judge near-ties by: fp32 summation noise of <q, c> is below 2e-6 |q| |c|
    if s64 is None:
        s64 = Qt.double() @ Ct.double().t()
    # the reference's step on fp32 CPU scores (trainer.py:304-309)
    ref_i = orc.mask_topk(Qt @ Ct.t() if Qt.shape[1] <= 512 else s64.float(), mask, k)[1]
    qn, cn = Qt.double().norm(dim=1), Ct.double().norm(dim=1)
    is_masked = torch.zeros(nq, nc, dtype=torch.bool)
    is_masked[torch.as_tensor(mask[0]), torch.as_tensor(mask[1])] = True
    if has_val:
        val = val.cpu().double()
    else:           # ids only (the kNN build): the value rules are vacuous on the float64 scores of the ids themselves
        val = torch.gather(s64.masked_fill(is_masked, float(np.float32(-1e10))), 1, idx)
    full = (nc - is_masked.sum(1)) >= k
    rows = torch.nonzero(full).flatten()
    loop_rows = torch.nonzero(~full).flatten().tolist()      # fewer than k unmasked candidates: masked ids fill the tail
    if rows.numel() <= 8:                                    # (also keeps the general statement exercised on ordinary rows)
        loop_rows += rows.tolist()
        rows = rows[:0]
    for r in loop_rows:
        _check_row(what, r, idx[r].numpy(), val[r].numpy(), ref_i[r].numpy(), s64[r].numpy(), is_masked[r].numpy(),
                   float(qn[r]), cn.numpy(), k, has_val)
    if rows.numel() == 0:
        return
    sel = (lambda t: t) if rows.numel() == nq else (lambda t: t[rows])
    idx_f, val_f, ref_f, s_f, m_f, qn_f = sel(idx), sel(val), sel(ref_i), sel(s64), sel(is_masked), sel(qn)[:, None]

    def none(cond, msg, extra=None):
        bad = torch.nonzero(cond.any(1) if cond.dim() == 2 else cond).flatten()
        assert bad.numel() == 0, (what, "rows", rows[bad[:4]].tolist(), msg, extra(int(bad[0])) if extra else None)

    if k > 1:
        none(torch.sort(idx_f, dim=1)[0].diff(dim=1) == 0, "duplicate ids")
    if k > 1 and has_val:
        none(val_f.diff(dim=1) > 0, "not sorted")
        none((val_f[:, 1:] == val_f[:, :-1]) & (idx_f[:, 1:] <= idx_f[:, :-1]), "tie order")
    none(torch.gather(m_f, 1, idx_f), "masked id ranked")
    s_got = torch.gather(s_f, 1, idx_f)
    none((val_f - s_got).abs() > 2e-6 * qn_f * cn[idx_f] + 1e-37, "values")
    s_m = s_f.masked_fill(m_f, float("-inf"))
    top_v, top_i = torch.topk(s_m, k, dim=1)                 # sorted: the k-th best unmasked float64 score and its candidate
    kth, kth_c = top_v[:, -1:], top_i[:, -1]
    in_got = torch.zeros_like(m_f).scatter_(1, idx_f, True)
    in_ref = torch.zeros_like(m_f).scatter_(1, ref_f, True)
    one_side = in_got ^ in_ref
    none(one_side & m_f, "oracle ranked a masked id")
    far = one_side & ((s_f - kth).abs() > 2e-6 * qn_f * (cn[None, :] + cn[kth_c][:, None]) + 1e-37)
    none(far, "ids differ from the oracle's beyond near-ties at the k-th score",
         lambda b: [(int(c), float(s_f[b, c]), float(kth[b])) for c in torch.nonzero(far[b]).flatten()[:4]])
    worst, wpos = s_got.min(1, keepdim=True)
    wc = torch.gather(idx_f, 1, wpos)[:, 0]
    none((s_m > worst + 2e-6 * qn_f * (cn[None, :] + cn[wc][:, None]) + 1e-37) & ~in_got, "a better candidate was dropped")


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(N_CASES), ids=lambda s: "seed%d" % s)
def test_score_topk_random_case_vs_oracle(seed):
    from mmrec_amd import hip_ops
    assert torch.cuda.is_available()
    check_case(hip_ops, torch.device("cuda", 0), gen_case(seed))
