"""GPU: the two pieces of BASELINE config 5 that the round-3 review found unchecked AT SIZE.

(a) P6 / SURVEY.md 8a a7 -- the frozen item-item graph config 5's FREEDOM consumes: kNN(10) over 500,000 row-normalised
    items x 4096 (image) and x 384 (text) features, built by the product function FREEDOM calls
    (`mmrec_amd.graph.knn_normalized_coo`; freedom.py:79-100) from features generated on the device the way
    tests/test_c5_e2e_gpu.py does.  The [I, I] similarity block is 1 TB; a row of it depends on its own query row only, so
    512 sampled rows are the reference's rows: `orc.knn_rows` forms them on the CPU (fp32 like the reference's `torch.mm`,
    and float64 to judge near-ties by).  Checked: every item is its own first neighbour, rows / columns are in the
    reference's row-major order, every value equals `compute_normalized_laplacian`'s (freedom.py:93-100), and the sampled
    rows' neighbour sets equal the oracle's up to near-ties at the 10th score (2e-6 relative, per candidate).

(b) a10 / a11 -- a 65,536 x 500,000 evaluation block in the state that broke the filter in round 3: TRAINED-shaped tables
    (LightGCN-propagated embeddings of the 10M-interaction graph, then 0.1 % of the item rows scaled by 5..25 x: heavy-tailed
    item norms), users with 0 / 16 / 600 / 5,000 masked items, k = 50, candidates prepared once as the Trainer does
    (`hip_ops.TopkCandidates`).  1,024 sampled users -- every heavy one among them -- vs `orc.mask_topk` on CPU scores with
    the differential rules of tests/test_topk_fuzz_gpu.py; the block also equals the unprepared call bit for bit on its
    first 4096 users.

tests/test_c5_pieces_cpu.py runs the same bodies on a miniature shape with the torch-CPU stand-in ops."""
import time

import numpy as np
import pytest
import torch

from oracle import mmrec_oracle as orc
from tests.test_topk_fuzz_gpu import check_lists

pytestmark = pytest.mark.gpu

USE_GPU = True
SHAPE = dict(n_users=1_000_000, n_items=500_000, n_edges=10_000_000, block=65_536, sample_rows=512, sample_users=1024,
             image_dim=4096, text_dim=384, heavy=((600, 48), (5000, 8)))


def log(*a):
    print("[c5-pieces]", *a, flush=True)


def _dev():
    return torch.device("cuda", 0) if USE_GPU else torch.device("cpu")


def _sync():
    if USE_GPU:
        torch.cuda.synchronize()


@pytest.mark.parametrize("modality", ["text", "image"])
def test_knn_graph_at_c5_item_count(modality):
    from mmrec_amd.graph import knn_normalized_coo
    dev, n, k = _dev(), SHAPE["n_items"], 10
    F = SHAPE[modality + "_dim"]
    g = torch.Generator(device=dev).manual_seed(7 + F)
    x = torch.randn(n, F, device=dev, generator=g)
    x = torch.relu(x) if modality == "image" else x / x.norm(dim=1, keepdim=True)
    _sync()
    t = time.time()
    idx, val = knn_normalized_coo(x, k)
    _sync()
    log("kNN(%d) of %d x %d on the device: %.2fs" % (k, n, F, time.time() - t))
    idx, val = idx.cpu(), val.cpu()
    knn = idx[1].view(n, k)
    assert torch.equal(idx[0], torch.arange(n).repeat_interleave(k))                  # row-major, k entries per row
    assert torch.equal(knn[:, 0], torch.arange(n))                                     # self-similarity ranks first
    assert int(knn.min()) >= 0 and int(knn.max()) < n
    assert bool((torch.sort(knn, dim=1)[0].diff(dim=1) > 0).all())                    # no neighbour twice
    ref_idx, ref_val = orc.knn_laplacian_values(knn)                                   # freedom.py:86-100 on OUR neighbours
    assert np.array_equal(idx.numpy(), ref_idx)
    np.testing.assert_allclose(val.numpy(), ref_val, rtol=1e-6, atol=0)
    np.testing.assert_allclose(val.numpy(), 1.0 / k, rtol=1e-6)                        # every row sums to k
    rows = np.sort(np.random.default_rng(F).choice(n, min(SHAPE["sample_rows"], n), replace=False))
    t = time.time()
    ref_knn, _, sim64, xn = orc.knn_rows(x.cpu(), rows, k)
    log("oracle rows on the CPU (%d x %d x %d, fp32 + float64): %.1fs" % (rows.shape[0], n, F, time.time() - t))
    same = float(np.mean([set(a) == set(b) for a, b in zip(knn[rows].tolist(), ref_knn.tolist())]))
    for a in range(0, rows.shape[0], 128):            # (the checker's float64 temporaries are [rows, I]: 128 rows at a time)
        check_lists("kNN %s rows" % modality, knn[rows[a:a + 128]], None, xn[rows[a:a + 128]], xn,
                    np.zeros((2, 0), dtype=np.int64), k, s64=sim64[a:a + 128])
    log("%s: identical neighbour sets on %.4f of the %d sampled rows (the rest: near-ties at the 10th score)" %
        (modality, same, rows.shape[0]))
    from tests._env import observed
    floor = 0.995 if USE_GPU else 0.9        # observed 1.0000 on 512 rows at both widths (profiles/r04_c5_pieces_knn.log)
    assert observed("c5_pieces.knn_%s_rows" % modality, same, floor) >= floor, same


def _trained_shaped_tables(dev):
    """LightGCN-propagated Xavier tables of the config-5 graph, then heavy-tailed item norms"""
    from mmrec_amd import hip_ops, synth
    nu, ni = SHAPE["n_users"], SHAPE["n_items"]
    t = time.time()
    eu, ei = synth.powerlaw_edges(nu, ni, SHAPE["n_edges"], seed=0)
    r, c, v = synth.sym_norm_coo(eu, ei, nu, ni)
    g = hip_ops.CsrGraph.from_coo_host(np.stack([r, c]), v, nu + ni, nu + ni, dev, symmetric=True)
    gen = torch.Generator().manual_seed(3)
    E0 = torch.empty(nu + ni, 64)
    torch.nn.init.xavier_uniform_(E0[:nu], generator=gen), torch.nn.init.xavier_uniform_(E0[nu:], generator=gen)
    E = hip_ops.lightgcn_mean(g, E0.to(dev), 2)
    U, I = E[:nu].contiguous(), E[nu:].clone()
    rng = np.random.default_rng(11)
    hot = rng.choice(ni, max(1, ni // 1000), replace=False)
    I[torch.as_tensor(hot).to(dev)] *= torch.as_tensor(rng.uniform(5, 25, (hot.shape[0], 1)).astype(np.float32)).to(dev)
    _sync()
    log("graph + propagation + heavy-tailed item norms: %.1fs" % (time.time() - t))
    return eu, ei, U, I.contiguous()


def test_trained_shaped_eval_block_vs_oracle():
    from mmrec_amd import hip_ops
    dev, ni, nb, k = _dev(), SHAPE["n_items"], SHAPE["block"], 50
    eu, ei, U, I = _trained_shaped_tables(dev)
    Q = U[:nb].contiguous()
    # masks: a third of the users none, the rest 16 random items (+ their train positives where they have some), and
    # heavy users with 600 / 5,000 masked items -- half of which are the user's BEST candidates (train positives score high)
    rng = np.random.default_rng(13)
    sel = (eu < nb)
    rows, cols = [eu[sel]], [ei[sel]]
    some = np.flatnonzero(rng.random(nb) < 2 / 3)
    rows.append(np.repeat(some, 16)), cols.append(rng.integers(0, ni, some.shape[0] * 16))
    none = np.setdiff1d(np.arange(nb), some)[:nb // 8]            # users with NO masked item at all
    heavy = []
    taken = rng.choice(some, sum(n for _, n in SHAPE["heavy"]), replace=False)
    at = 0
    for m, n_users in SHAPE["heavy"]:
        for u in taken[at:at + n_users]:
            m_eff = min(m, ni - k - 1)
            s = (Q[int(u)] @ I.t())
            best = torch.topk(s, m_eff // 2)[1].cpu().numpy()
            rows.append(np.full(m_eff, u)), cols.append(np.concatenate([best, rng.choice(ni, m_eff - best.shape[0], replace=False)]))
            heavy.append(int(u))
        at += n_users
    key = np.unique(np.concatenate(rows).astype(np.int64) * ni + np.concatenate(cols).astype(np.int64))
    mask = np.stack([key // ni, key % ni])
    mask = mask[:, ~np.isin(mask[0], none)]
    rp, col = hip_ops.mask_to_csr(mask, nb, dev)
    cands = hip_ops.TopkCandidates(I)
    _sync()
    t = time.time()
    idx, val = hip_ops.score_topk(Q, cands, k, rp, col, return_values=True)
    _sync()
    log("score + mask + top-%d of %d x %d (prepared candidates): %.1f ms" % (k, nb, ni, 1e3 * (time.time() - t)))
    sample = np.unique(np.concatenate([heavy, none[:64], rng.choice(nb, SHAPE["sample_users"], replace=False)]))
    pos = np.searchsorted(sample, mask[0])
    hit = (pos < sample.shape[0]) & (sample[np.minimum(pos, sample.shape[0] - 1)] == mask[0])
    local = np.stack([pos[hit], mask[1][hit]])
    cnt = np.bincount(local[0], minlength=sample.shape[0])
    assert cnt.min() == 0 and cnt.max() >= 0.9 * min(max(m for m, _ in SHAPE["heavy"]), ni - k - 1)
    Ic = I.cpu()
    for a in range(0, sample.shape[0], 256):          # (the checker's float64 temporaries are [rows, I]: 256 rows at a time)
        st = torch.as_tensor(sample[a:a + 256]).to(dev)
        part = (local[0] >= a) & (local[0] < a + 256)
        check_lists("trained-shaped block", idx[st].cpu(), val[st].cpu(), Q[st].cpu(), Ic,
                    np.stack([local[0][part] - a, local[1][part]]), k)
    # the same rows through the unprepared entry point: the very bits
    n0 = min(4096, nb)
    rp0 = rp[:n0 + 1].contiguous()
    col0 = col[:max(int(rp0[-1]), 1)].contiguous()
    i0, v0 = hip_ops.score_topk(Q[:n0].contiguous(), I, k, rp0, col0, return_values=True)
    assert torch.equal(i0, idx[:n0]) and torch.equal(v0, val[:n0])
