"""CPU oracle for the MMRec hot path -- TEST INFRASTRUCTURE ONLY.

This file restates, function by function, the algorithm of the reference hot path
 so that the HIP kernels can be checked without /root/reference being
present (it does not exist on the GPU box).  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import it -- never the product package `mmrec_amd`, which must
fail loudly when its HIP library is missing.

Pinning: the reference ships no tests or golden vectors, so this oracle is
pinned against outputs of the *unmodified reference run in the build container*:
`tests/golden/tiny.npz`, produced by `tests/golden/make_golden.py`; `tests/test_oracle_golden.py`
checks every function below against it.

Conventions: integer / structural work is numpy (bit-exact expected); floating-point work uses the
same torch-CPU fp32 operators the reference itself calls (the reference *is* torch on CPU), so
gradients come from autograd exactly as in the reference.  Citations are `file:line` under
/root/reference/src.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------------
# P1 -- adjacency construction (integer structure exact, values fp64 -> fp32)
# --------------------------------------------------------------------------------------------


def norm_adj_coo(train_rows, train_cols, n_users, n_items):
    """D^-1/2 [[0,R],[R^T,0]] D^-1/2 as row-major-sorted COO.

    Follows models/freedom.py:102-126 (identical copies: bm3.py:58-82, layergcn.py:91-115,
    lightgcn.py:65-101): A is *binary* (the dict de-duplicates repeated pairs), degree =
    (A>0).sum(1) + 1e-7 in float64, d = degree**-0.5, value = (d[r]*1)*d[c] in float64, then cast to
    float32.  Returns (idx[2,nnz] int64 sorted by (row,col), val[nnz] float32, n_nodes).
    """
    r = np.asarray(train_rows, dtype=np.int64)
    c = np.asarray(train_cols, dtype=np.int64)
    n = int(n_users) + int(n_items)
    key = np.unique(r * np.int64(n_items) + c)  # de-dup, as the python dict does
    ur, uc = key // n_items, key % n_items
    rows = np.concatenate([ur, uc + n_users])
    cols = np.concatenate([uc + n_users, ur])
    order = np.lexsort((cols, rows))
    rows, cols = rows[order], cols[order]
    deg = np.bincount(rows, minlength=n).astype(np.float64) + 1e-7
    d = np.power(deg, -0.5)
    val = ((d[rows] * 1.0) * d[cols]).astype(np.float32)
    return np.stack([rows, cols]), val, n


def edge_norm_values(edge_rows, edge_cols, n_users, n_items):
    """Per-edge (du+1e-7)^-1/2 (di+1e-7)^-1/2 in fp32 -- models/freedom.py:145-162 (`get_edge_info`
    + `_normalize_adj_m`), same at layergcn.py:72-89.  Degrees count *entries* (duplicates add)."""
    er = torch.as_tensor(np.asarray(edge_rows, dtype=np.int64))
    ec = torch.as_tensor(np.asarray(edge_cols, dtype=np.int64))
    ones = torch.ones(er.shape[0], dtype=torch.float32)
    row_sum = 1e-7 + torch.zeros(n_users, dtype=torch.float32).index_add_(0, er, ones)
    col_sum = 1e-7 + torch.zeros(n_items, dtype=torch.float32).index_add_(0, ec, ones)
    return (torch.pow(row_sum, -0.5)[er] * torch.pow(col_sum, -0.5)[ec]).numpy()


def masked_adj_coo(edge_indices, keep_idx, n_users, n_items):
    """Degree-sensitive edge dropout, given the sampled `keep_idx` (the multinomial draw itself is
    device RNG and is injected) -- models/freedom.py:128-143 / layergcn.py:51-70: kept edges are
    re-normalised on the kept sub-graph, then laid out as cat(edges, flipped edges) with cat(v, v).
    Returns (idx[2,2*keep] int64 in the reference's order, val fp32)."""
    ei = np.asarray(edge_indices, dtype=np.int64)
    keep = ei[:, np.asarray(keep_idx, dtype=np.int64)].copy()
    v = edge_norm_values(keep[0], keep[1], n_users, n_items)
    keep[1] += n_users
    idx = np.concatenate([keep, keep[::-1]], axis=1)
    return idx, np.concatenate([v, v])


def knn_item_graph(feats, k):
    """kNN(k) item-item graph from row-normalised features, symmetric-normalised by row sums:
    models/freedom.py:79-100.  Every row has exactly k entries so every value is ~1/k.
    Returns (idx[2, I*k] int64 row-major, val fp32, knn_ind[I,k])."""
    x = torch.as_tensor(feats, dtype=torch.float32)
    xn = x.div(torch.norm(x, p=2, dim=-1, keepdim=True))
    sim = torch.mm(xn, xn.t())
    _, knn = torch.topk(sim, k, dim=-1)
    idx, val = knn_laplacian_values(knn)
    return idx, val, knn.numpy()


def knn_laplacian_values(knn):
    """models/freedom.py:86-100 (`compute_normalized_laplacian`) for given neighbour lists knn [I, k]: (rows, cols) in row-major
    order and D^-1/2 A D^-1/2 values with D = row sums of the 0/1 adjacency + 1e-7 (fp32).  -> (idx [2, I*k] int64, val fp32)."""
    knn = torch.as_tensor(knn, dtype=torch.int64)
    n, k = knn.shape
    rows = torch.arange(n).unsqueeze(1).expand(-1, k).reshape(-1)
    cols = knn.reshape(-1)
    row_sum = 1e-7 + torch.zeros(n, dtype=torch.float32).index_add_(
        0, rows, torch.ones(rows.shape[0], dtype=torch.float32))
    r_inv = torch.pow(row_sum, -0.5)
    val = r_inv[rows] * r_inv[cols]
    return torch.stack([rows, cols]).numpy(), val.numpy()


def knn_rows(feats, rows, k, chunk=32768):
    """The rows `rows` of models/freedom.py:80-84 -- `sim = mm(context_norm, context_norm^T)`, `topk(sim, k)` -- for item
    counts whose [I, I] similarity block cannot be formed (config 5: 1 TB): a row of `sim` depends on its own query row
    only, so the sampled rows are the reference's rows.  Candidates are walked in chunks; besides the reference's fp32
    scores the float64 scores of the same normalised rows are returned (to judge near-ties by).
    Returns (knn [len(rows), k] int64, sim32 [len(rows), I] fp32, sim64 [len(rows), I] float64, xn [I, F] fp32)."""
    x = torch.as_tensor(feats, dtype=torch.float32)
    xn = x.div(torch.norm(x, p=2, dim=-1, keepdim=True))
    rows = torch.as_tensor(np.asarray(rows), dtype=torch.int64)
    q = xn[rows]
    q64 = q.double()
    sim32 = torch.empty(rows.shape[0], x.shape[0], dtype=torch.float32)
    sim64 = torch.empty(rows.shape[0], x.shape[0], dtype=torch.float64)
    for a in range(0, x.shape[0], chunk):
        c = xn[a:a + chunk]
        torch.mm(q, c.t(), out=sim32[:, a:a + chunk])
        torch.mm(q64, c.double().t(), out=sim64[:, a:a + chunk])
    _, knn = torch.topk(sim32, k, dim=-1)
    return knn, sim32, sim64, xn


def freedom_mm_adj(image_feat, text_feat, k, image_weight):
    """w*A_img + (1-w)*A_txt (models/freedom.py:64-77) as an *uncoalesced* COO (concatenation)."""
    ii, iv, _ = knn_item_graph(image_feat, k)
    ti, tv, _ = knn_item_graph(text_feat, k)
    idx = np.concatenate([ii, ti], axis=1)
    val = np.concatenate([np.float32(image_weight) * iv, np.float32(1.0 - image_weight) * tv])
    return idx, val.astype(np.float32)


def coalesce_coo(idx, val, n_rows, n_cols):
    """Sum duplicate (row,col) entries in first-seen order; returns row-major sorted COO.  Used to
    compare graphs irrespective of storage order (the reference stores uncoalesced COO)."""
    idx = np.asarray(idx, dtype=np.int64)
    key = idx[0] * np.int64(n_cols) + idx[1]
    uk, inv = np.unique(key, return_inverse=True)
    out = np.zeros(uk.shape[0], dtype=np.float32)
    np.add.at(out, inv, np.asarray(val, dtype=np.float32))
    return np.stack([uk // n_cols, uk % n_cols]), out


def coo_to_csr(idx, val, n_rows):
    """Stable COO -> CSR (int32 rowptr/colidx): entries of one row keep their COO order.  This is the
    structure the HIP SpMM consumes; int32 is enough for N <= 1.5M, nnz <= 20M."""
    idx = np.asarray(idx, dtype=np.int64)
    order = np.argsort(idx[0], kind="stable")
    rowptr = np.zeros(n_rows + 1, dtype=np.int64)
    np.cumsum(np.bincount(idx[0], minlength=n_rows), out=rowptr[1:])
    return rowptr.astype(np.int32), idx[1][order].astype(np.int32), np.asarray(val, dtype=np.float32)[order]


# --------------------------------------------------------------------------------------------
# P2 -- sparse propagation
# --------------------------------------------------------------------------------------------


def sparse_coo(idx, val, n_rows, n_cols=None):
    """The reference keeps adjacency as an uncoalesced torch sparse COO fp32 tensor (freedom.py:126)."""
    n_cols = n_rows if n_cols is None else n_cols
    return torch.sparse_coo_tensor(torch.as_tensor(np.asarray(idx), dtype=torch.int64),
                                   torch.as_tensor(np.asarray(val), dtype=torch.float32),
                                   (n_rows, n_cols))


def spmm(adj, x):
    """Y = A @ X, the reference's own call: torch.sparse.mm (freedom.py:172, bm3.py:90,
    layergcn.py:131, lightgcn.py:120).  Also the operator `bench.py` times as the CPU baseline."""
    return torch.sparse.mm(adj, x)


def lightgcn_forward(adj, user_emb, item_emb, n_layers):
    """mean over [E0, A E0, ..., A^L E0] -- models/lightgcn.py:115-128."""
    e = torch.cat([user_emb, item_emb], 0)
    layers = [e]
    for _ in range(n_layers):
        e = spmm(adj, e)
        layers.append(e)
    out = torch.mean(torch.stack(layers, dim=1), dim=1)
    return out[:user_emb.shape[0]], out[user_emb.shape[0]:]


def layergcn_forward(adj, user_emb, item_emb, n_layers):
    """per layer E <- A E; w = cos(E, E0) (eps 1e-8); E <- w*E; output = sum of layers, ego
    excluded -- models/layergcn.py:125-138."""
    ego = torch.cat([user_emb, item_emb], 0)
    e, layers = ego, []
    for _ in range(n_layers):
        e = spmm(adj, e)
        w = F.cosine_similarity(e, ego, dim=-1)
        e = torch.einsum('a,ab->ab', w, e)
        layers.append(e)
    out = torch.sum(torch.stack(layers, dim=0), dim=0)
    return out[:user_emb.shape[0]], out[user_emb.shape[0]:]


def freedom_forward(adj, mm_adj, user_emb, item_emb, n_ui_layers, n_mm_layers):
    """models/freedom.py:164-178: h = M^n_mm I ; LightGCN mean over n_ui layers ; items += h."""
    h = item_emb
    for _ in range(n_mm_layers):
        h = spmm(mm_adj, h)
    u, i = lightgcn_forward(adj, user_emb, item_emb, n_ui_layers)
    return u, i + h


def bm3_forward(adj, user_emb, item_emb, n_layers):
    """models/bm3.py:84-95: LightGCN mean + residual item id embedding."""
    u, i = lightgcn_forward(adj, user_emb, item_emb, n_layers)
    return u, i + item_emb


# --------------------------------------------------------------------------------------------
# P3 -- modal projection
# --------------------------------------------------------------------------------------------


def linear(x, w, b=None):
    """nn.Linear: X W^T + b over all items (freedom.py:205,208; bm3.py:102,104; vbpr.py:70)."""
    return F.linear(x, w, b)


# --------------------------------------------------------------------------------------------
# P4 -- sampled scoring / losses
# --------------------------------------------------------------------------------------------


def bpr_logsigmoid(u, p, n, reduction="mean"):
    """-mean / -sum of logsigmoid(<u,p> - <u,n>): freedom.py:180-187 (mean), layergcn.py:140-152 (sum)."""
    x = torch.sum(u * p, dim=1) - torch.sum(u * n, dim=1)
    ls = F.logsigmoid(x)
    return -(ls.mean() if reduction == "mean" else ls.sum())


def bpr_gamma(u, p, n, gamma=1e-10):
    """common/loss.py:33-35 BPRLoss: -mean log(gamma + sigmoid(pos - neg)) (VBPR, LightGCN)."""
    x = torch.sum(u * p, dim=1) - torch.sum(u * n, dim=1)
    return -torch.log(gamma + torch.sigmoid(x)).mean()


def emb_loss(*embs):
    """common/loss.py:46-51 EmbLoss: sum of *unsquared* Frobenius norms / rows of the last arg."""
    s = sum(torch.norm(e, p=2) for e in embs)
    return s / embs[-1].shape[0]


def l2_loss(*embs):
    """common/loss.py:58-62 L2Loss: sum of 0.5*||x||^2, undivided."""
    return sum(0.5 * torch.sum(e ** 2) for e in embs)


def lightgcn_loss(adj, user_emb, item_emb, n_layers, batch, reg_weight):
    """models/lightgcn.py:130-153."""
    u_all, i_all = lightgcn_forward(adj, user_emb, item_emb, n_layers)
    us, ps, ns = (torch.as_tensor(b) for b in batch)
    mf = bpr_gamma(u_all[us], i_all[ps], i_all[ns])
    reg = emb_loss(user_emb[us], item_emb[ps], item_emb[ns])
    return mf + reg_weight * reg


def layergcn_loss(adj, user_emb, item_emb, n_layers, batch, reg_weight):
    """models/layergcn.py:163-175: BPR *sum* + reg_weight * L2Loss on the ego rows of the batch."""
    u_all, i_all = layergcn_forward(adj, user_emb, item_emb, n_layers)
    us, ps, ns = (torch.as_tensor(b) for b in batch)
    mf = bpr_logsigmoid(u_all[us], i_all[ps], i_all[ns], "sum")
    reg = l2_loss(user_emb[us], item_emb[ps], item_emb[ns])
    return mf + reg_weight * reg


def freedom_loss(adj, mm_adj, user_emb, item_emb, image_feat, image_w, image_b, text_feat, text_w,
                 text_b, n_ui_layers, n_mm_layers, batch, reg_weight):
    """models/freedom.py:189-210: id BPR + reg_weight*(text BPR + image BPR); projections run over all
    items and are consumed only at the pos/neg rows."""
    ua, ia = freedom_forward(adj, mm_adj, user_emb, item_emb, n_ui_layers, n_mm_layers)
    us, ps, ns = (torch.as_tensor(b) for b in batch)
    loss = bpr_logsigmoid(ua[us], ia[ps], ia[ns])
    tf = linear(text_feat, text_w, text_b)
    vf = linear(image_feat, image_w, image_b)
    mf_t = bpr_logsigmoid(ua[us], tf[ps], tf[ns])
    mf_v = bpr_logsigmoid(ua[us], vf[ps], vf[ns])
    return loss + reg_weight * (mf_t + mf_v)


def bm3_loss(adj, user_emb, item_emb, pred_w, pred_b, image_feat, image_w, image_b, text_feat,
             text_w, text_b, n_layers, batch2, reg_weight, cl_weight, dropout, masks):
    """models/bm3.py:97-147 with the four F.dropout keep-masks injected (u, i, t, v order of
    bm3.py:110-119; device RNG is not portable).  Targets carry no gradient."""
    u_on, i_on = bm3_forward(adj, user_emb, item_emb, n_layers)
    t_on = linear(text_feat, text_w, text_b)
    v_on = linear(image_feat, image_w, image_b)
    scale = 1.0 / (1.0 - dropout)
    mk = [torch.as_tensor(np.asarray(m), dtype=torch.float32) for m in masks]
    with torch.no_grad():
        u_t = u_on.detach() * mk[0] * scale
        i_t = i_on.detach() * mk[1] * scale
        t_t = t_on.detach() * mk[2] * scale
        v_t = v_on.detach() * mk[3] * scale
    users, items = torch.as_tensor(batch2[0]), torch.as_tensor(batch2[1])
    u_p, i_p = linear(u_on, pred_w, pred_b)[users], linear(i_on, pred_w, pred_b)[items]
    u_t, i_t = u_t[users], i_t[items]
    t_p = linear(t_on, pred_w, pred_b)[items]
    v_p = linear(v_on, pred_w, pred_b)[items]
    cs = F.cosine_similarity
    loss_t = 1 - cs(t_p, i_t, dim=-1).mean()
    loss_tv = 1 - cs(t_p, t_t[items], dim=-1).mean()
    loss_v = 1 - cs(v_p, i_t, dim=-1).mean()
    loss_vt = 1 - cs(v_p, v_t[items], dim=-1).mean()
    loss_ui = 1 - cs(u_p, i_t, dim=-1).mean()
    loss_iu = 1 - cs(i_p, u_t, dim=-1).mean()
    return (loss_ui + loss_iu) + reg_weight * emb_loss(u_on, i_on) + \
        cl_weight * (loss_t + loss_v + loss_tv + loss_vt)


def vbpr_forward(u_emb, i_emb, raw_feats, w, b):
    """models/vbpr.py:69-75 (dropout 0): item = cat(id_emb, Linear(raw)); user emb is 2*d wide."""
    return u_emb, torch.cat((i_emb, linear(raw_feats, w, b)), -1)


def vbpr_loss(u_emb, i_emb, raw_feats, w, b, batch, reg_weight):
    """models/vbpr.py:77-98."""
    ue, ie = vbpr_forward(u_emb, i_emb, raw_feats, w, b)
    us, ps, ns = (torch.as_tensor(x) for x in batch)
    mf = bpr_gamma(ue[us], ie[ps], ie[ns])
    return mf + reg_weight * emb_loss(ue[us], ie[ps], ie[ns])


# --------------------------------------------------------------------------------------------
# P5 -- full-sort scoring + mask + top-K ; a12 metrics
# --------------------------------------------------------------------------------------------


def full_sort_scores(user_all, item_all, users, out=None):
    """`scores = U[users] @ I^T` -- freedom.py:212-220 and siblings.  `out`: a buffer to write the block into (walking 50,000
    users x 500,000 items block by block, fresh 4 GB allocations spend their time in page faults)."""
    return torch.matmul(user_all[torch.as_tensor(users)], item_all.t(), out=out)


def mask_topk(scores, mask, k, inplace=False):
    """common/trainer.py:304-309: scores[mask0, mask1] = -1e10 ; topk(k) sorted descending.
    Returns (values[b,k] fp32, indices[b,k] int64).  Order among exact ties is unspecified.  inplace: mask the caller's
    block itself, as the reference does (trainer.py:307)."""
    s = scores if inplace else scores.clone()
    m = torch.as_tensor(np.asarray(mask), dtype=torch.int64)
    s[m[0], m[1]] = -1e10
    v, i = torch.topk(s, k, dim=-1)
    return v, i


def hit_matrix(topk_idx, pos_flat, pos_len):
    """utils/topk_evaluator.py:88-93: hit[u, j] = topk_idx[u, j] in ground-truth set of user u."""
    topk_idx = np.asarray(topk_idx)
    off = np.concatenate([[0], np.cumsum(pos_len)])
    hit = np.zeros(topk_idx.shape, dtype=bool)
    for u in range(topk_idx.shape[0]):
        hit[u] = np.isin(topk_idx[u], pos_flat[off[u]:off[u + 1]])
    return hit


def topk_metrics(hit, pos_len, topk=(5, 10, 20, 50), metrics=("recall", "ndcg", "precision", "map")):
    """utils/metrics.py:12-105 + rounding of topk_evaluator.py:95-102 (4 dp).  Returns an ordered
    dict keyed like the reference ('recall@5', ...)."""
    hit = np.asarray(hit, dtype=bool)
    pos_len = np.asarray(pos_len, dtype=np.int64)
    n, kmax = hit.shape
    ranks = np.arange(1, kmax + 1, dtype=np.float64)
    cum = np.cumsum(hit, axis=1)
    res = {}
    curves = {}
    curves["recall"] = (cum / pos_len.reshape(-1, 1)).mean(axis=0)
    curves["precision"] = (cum / ranks).mean(axis=0)
    # ndcg: idcg is cumulated up to min(pos_len, K) then held flat (metrics.py:45-62)
    disc = 1.0 / np.log2(ranks + 1)
    idcg_full = np.cumsum(disc)
    cap = np.minimum(pos_len, kmax)
    pos = np.minimum(np.arange(kmax)[None, :], (cap - 1)[:, None])
    idcg = idcg_full[pos]
    dcg = np.cumsum(np.where(hit, disc[None, :], 0.0), axis=1)
    curves["ndcg"] = (dcg / idcg).mean(axis=0)
    # map: AP@N normalised by min(m, N) (metrics.py:65-89)
    pre = cum / ranks
    sum_pre = np.cumsum(pre * hit.astype(np.float64), axis=1)
    denom = np.minimum(np.arange(1, kmax + 1)[None, :], cap[:, None]).astype(np.float64)
    curves["map"] = (sum_pre / denom).mean(axis=0)
    for m in metrics:
        for k in topk:
            res["%s@%d" % (m, k)] = round(float(curves[m][k - 1]), 4)
    return res


# --------------------------------------------------------------------------------------------
# LATTICE (models/lattice.py) -- dense torch-CPU restatement, pinned by tests/golden/lattice.npz
# --------------------------------------------------------------------------------------------


def lattice_norm_adj_coo(train_rows, train_cols, n_users, n_items):
    """Row-normalised bipartite adjacency with self loops, D^-1 (A + I) -- lattice.py:100-122.
    float64 arithmetic then cast to float32; returns row-major sorted (idx, val, n)."""
    r = np.asarray(train_rows, dtype=np.int64)
    c = np.asarray(train_cols, dtype=np.int64)
    n = int(n_users) + int(n_items)
    key = np.unique(r * np.int64(n_items) + c)
    ur, uc = key // n_items, key % n_items
    rows = np.concatenate([ur, uc + n_users, np.arange(n)])
    cols = np.concatenate([uc + n_users, ur, np.arange(n)])
    order = np.lexsort((cols, rows))
    rows, cols = rows[order], cols[order]
    rowsum = np.bincount(rows, minlength=n).astype(np.float64)
    val = (np.power(rowsum, -1.0)[rows]).astype(np.float32)
    return np.stack([rows, cols]), val, n


def lattice_knn_dense(feats, k):
    """build_sim -> build_knn_neighbourhood -> compute_normalized_laplacian (utils/utils.py:119-137):
    cosine similarity, keep the top-k similarity VALUES per row, symmetric D^-1/2 . D^-1/2 by row sums."""
    x = torch.as_tensor(feats, dtype=torch.float32)
    xn = x.div(torch.norm(x, p=2, dim=-1, keepdim=True))
    sim = torch.mm(xn, xn.t())
    val, ind = torch.topk(sim, k, dim=-1)
    adj = torch.zeros_like(sim).scatter_(-1, ind, val)
    return lattice_sym_norm_dense(adj)


def lattice_sym_norm_dense(adj):
    rowsum = torch.sum(adj, -1)
    d = torch.pow(rowsum, -0.5)
    d[torch.isinf(d)] = 0.
    return d.unsqueeze(1) * adj * d.unsqueeze(0)


def lattice_item_adj(image_feats, text_feats, image_orig, text_orig, modal_weight, k, lambda_coeff):
    """Learned + original item graph, lattice.py:137-157 (differentiable in the projected features
    and the modal weights through the kept similarity values)."""
    w = torch.softmax(modal_weight, dim=0)

    def knn_weighted(f):
        fn = f.div(torch.norm(f, p=2, dim=-1, keepdim=True))
        sim = torch.mm(fn, fn.t())
        val, ind = torch.topk(sim, k, dim=-1)
        return torch.zeros_like(sim).scatter_(-1, ind, val)

    learned = w[0] * knn_weighted(image_feats) + w[1] * knn_weighted(text_feats)
    original = w[0] * image_orig + w[1] * text_orig
    return (1 - lambda_coeff) * lattice_sym_norm_dense(learned) + lambda_coeff * original


def lattice_forward(adj, item_adj, user_emb, item_emb, n_ui_layers, n_layers):
    """cf_model 'lightgcn' branch, lattice.py:161-195: h = item_adj^n_layers I ; layer mean ; items += normalize(h)."""
    h = item_emb
    for _ in range(n_layers):
        h = torch.mm(item_adj, h)
    u, i = lightgcn_forward(adj, user_emb, item_emb, n_ui_layers)
    return u, i + F.normalize(h, p=2, dim=1)


def lattice_loss(ua, ia, batch, reg_weight, batch_size):
    """lattice.py:199-211: -mean logsigmoid + reg_weight * 0.5*(|u|^2+|p|^2+|n|^2) / batch_size(config)."""
    us, ps, ns = (torch.as_tensor(b) for b in batch)
    u, p, n = ua[us], ia[ps], ia[ns]
    mf = bpr_logsigmoid(u, p, n)
    reg = 0.5 * ((u ** 2).sum() + (p ** 2).sum() + (n ** 2).sum()) / batch_size
    return mf + reg_weight * reg


# --------------------------------------------------------------------------------------------
# MMGCN (models/mmgcn.py) -- torch-CPU restatement, pinned by tests/golden/mmgcn.npz.  The reference
# relies on torch_geometric's MessagePassing(aggr='mean') (unpinned, absent here): messages flow
# edge_index[0] -> edge_index[1] and are averaged over the in-degree (0 for isolated nodes).
# --------------------------------------------------------------------------------------------


def mean_aggregate(x, edge_index):
    """BaseModel.propagate with aggr='mean' (mmgcn.py:205-213)."""
    src, dst = torch.as_tensor(edge_index[0]), torch.as_tensor(edge_index[1])
    n = x.shape[0]
    out = torch.zeros(n, x.shape[1], dtype=x.dtype).index_add_(0, dst, x[src])
    deg = torch.zeros(n, dtype=x.dtype).index_add_(0, dst, torch.ones(dst.numel(), dtype=x.dtype))
    return out / deg.clamp(min=1).unsqueeze(1)


def mmgcn_gcn(p, prefix, features, preference, id_embedding, edge_index, has_mlp):
    """GCN.forward (mmgcn.py:164-188) with concate truthy ('False' is a non-empty string) and has_id."""
    g = lambda name: p[prefix + name]
    temp = F.linear(features, g("MLP.weight"), g("MLP.bias")) if has_mlp else features
    x = F.normalize(torch.cat((preference, temp), dim=0))
    for layer in (1, 2, 3):
        h = F.leaky_relu(mean_aggregate(torch.matmul(x, g("conv_embed_%d.weight" % layer)), edge_index))
        x_hat = F.leaky_relu(F.linear(x, g("linear_layer%d.weight" % layer), g("linear_layer%d.bias" % layer))) + id_embedding
        x = F.leaky_relu(F.linear(torch.cat((h, x_hat), dim=1), g("g_layer%d.weight" % layer), g("g_layer%d.bias" % layer)))
    return x


def mmgcn_forward(p, v_feat, t_feat, v_pref, t_pref, id_embedding, edge_index):
    """MMGCN.forward (mmgcn.py:64-77): mean of the modality representations."""
    rep = mmgcn_gcn(p, "v_gcn.", v_feat, v_pref, id_embedding, edge_index, True)
    rep = rep + mmgcn_gcn(p, "t_gcn.", t_feat, t_pref, id_embedding, edge_index, False)
    return rep / 2


def mmgcn_loss(out, id_embedding, v_pref, batch, n_users, reg_weight):
    """MMGCN.calculate_loss (mmgcn.py:79-97): interleaved (pos, neg) pairs, -mean log sigmoid(pos-neg),
    reg on the (untrained) id embedding rows and the visual preference table."""
    users = torch.as_tensor(batch[0])
    pos, neg = torch.as_tensor(batch[1]) + n_users, torch.as_tensor(batch[2]) + n_users
    user_t = users.repeat_interleave(2)
    item_t = torch.stack((pos, neg)).t().contiguous().view(-1)
    score = torch.sum(out[user_t] * out[item_t], dim=1).view(-1, 2)
    loss = -torch.mean(torch.log(torch.sigmoid(torch.matmul(score, torch.tensor([[1.0], [-1.0]])))))
    reg = (id_embedding[user_t] ** 2 + id_embedding[item_t] ** 2).mean() + (v_pref ** 2).mean()
    return loss + reg_weight * reg


# --------------------------------------------------------------------------------------------
# MGCN (models/mgcn.py) -- torch-CPU restatement, pinned by tests/golden/mgcn.npz.  The reference's
# sparse kNN graphs go through torch_scatter.scatter_add (unpinned, absent here; = index_add).
# --------------------------------------------------------------------------------------------


def mgcn_norm_adj_coo(train_rows, train_cols, n_users, n_items):
    """D^-1/2 A D^-1/2 of the bipartite graph, mgcn.py:111-136: row sums and powers in float32
    (the dok matrix is float32), inf -> 0 for isolated nodes, no epsilon.  Returns (idx, val, n)
    sorted row-major; the normalised user-item block R is its rows < n_users with columns - n_users."""
    r = np.asarray(train_rows, dtype=np.int64)
    c = np.asarray(train_cols, dtype=np.int64)
    n = int(n_users) + int(n_items)
    key = np.unique(r * np.int64(n_items) + c)
    ur, uc = key // n_items, key % n_items
    rows = np.concatenate([ur, uc + n_users])
    cols = np.concatenate([uc + n_users, ur])
    order = np.lexsort((cols, rows))
    rows, cols = rows[order], cols[order]
    rowsum = np.bincount(rows, minlength=n).astype(np.float32)
    with np.errstate(divide="ignore"):
        d = np.power(rowsum, np.float32(-0.5)).astype(np.float32)
    d[np.isinf(d)] = 0.
    val = (d[rows] * np.float32(1.0)) * d[cols]
    return np.stack([rows, cols]), val.astype(np.float32), n


def mgcn_knn_graph(feats, k):
    """build_sim + build_knn_normalized_graph(is_sparse=True, norm_type='sym') -- utils/utils.py:131-134,
    136-146, 166-177: top-k cosine similarities per row as edge weights, w' = d^-1/2[row] w d^-1/2[col]
    with d = ROW sums of the kept weights.  Returns (idx [2, n*k] in (row, rank) order, val)."""
    x = torch.as_tensor(feats, dtype=torch.float32)
    xn = x.div(torch.norm(x, p=2, dim=-1, keepdim=True))
    sim = torch.mm(xn, xn.t())
    val, ind = torch.topk(sim, k, dim=-1)
    n = x.shape[0]
    row = torch.arange(n).repeat_interleave(k)
    col = ind.reshape(-1)
    w = val.reshape(-1)
    deg = torch.zeros(n, dtype=w.dtype).index_add_(0, row, w)
    dis = deg.pow(-0.5)
    dis[dis == float("inf")] = 0
    return torch.stack([row, col]).numpy(), (dis[row] * w * dis[col]).numpy()


def infonce(view1, view2, temperature):
    """MGCN.InfoNCE, mgcn.py:224-231 (no max-subtraction, all B columns as negatives)."""
    v1, v2 = F.normalize(view1, dim=1), F.normalize(view2, dim=1)
    pos = torch.exp((v1 * v2).sum(dim=-1) / temperature)
    ttl = torch.exp(torch.matmul(v1, v2.t()) / temperature).sum(dim=1)
    return torch.mean(-torch.log(pos / ttl))


def mgcn_forward(p, adj, R, image_adj, text_adj, n_users, n_ui_layers, n_layers):
    """MGCN.forward(train=True), mgcn.py:145-209.  `p`: parameter dict (reference names); adj, R,
    image_adj, text_adj: torch sparse tensors.  Returns (users, items, side_embeds, content_embeds)."""
    image_feats = F.linear(p["image_embedding.weight"], p["image_trs.weight"], p["image_trs.bias"])
    text_feats = F.linear(p["text_embedding.weight"], p["text_trs.weight"], p["text_trs.bias"])
    gate = lambda name, x: torch.sigmoid(F.linear(x, p[name + ".0.weight"], p[name + ".0.bias"]))
    item_w, user_w = p["item_id_embedding.weight"], p["user_embedding.weight"]
    image_item = item_w * gate("gate_v", image_feats)
    text_item = item_w * gate("gate_t", text_feats)
    ego = torch.cat([user_w, item_w], dim=0)
    layers = [ego]
    for _ in range(n_ui_layers):
        ego = torch.sparse.mm(adj, ego)
        layers.append(ego)
    content = torch.stack(layers, dim=1).mean(dim=1)
    for _ in range(n_layers):
        image_item = torch.sparse.mm(image_adj, image_item)
    image_embeds = torch.cat([torch.sparse.mm(R, image_item), image_item], dim=0)
    for _ in range(n_layers):
        text_item = torch.sparse.mm(text_adj, text_item)
    text_embeds = torch.cat([torch.sparse.mm(R, text_item), text_item], dim=0)

    def query(x):
        h = torch.tanh(F.linear(x, p["query_common.0.weight"], p["query_common.0.bias"]))
        return F.linear(h, p["query_common.2.weight"])
    w = torch.softmax(torch.cat([query(image_embeds), query(text_embeds)], dim=-1), dim=-1)
    common = w[:, 0].unsqueeze(1) * image_embeds + w[:, 1].unsqueeze(1) * text_embeds
    sep_image = gate("gate_image_prefer", content) * (image_embeds - common)
    sep_text = gate("gate_text_prefer", content) * (text_embeds - common)
    side = (sep_image + sep_text + common) / 3
    out = content + side
    return out[:n_users], out[n_users:], side, content


def mgcn_loss(ua, ia, side, content, batch, n_users, reg_weight, cl_weight, batch_size, tau=0.2):
    """MGCN.calculate_loss, mgcn.py:233-255: -mean logsigmoid + reg_weight * 0.5*(|u|^2+|p|^2+|n|^2) /
    batch_size(config) + cl_weight * (InfoNCE(items @ pos) + InfoNCE(users @ users))."""
    us, ps, ns = (torch.as_tensor(b) for b in batch)
    u, pp, nn_ = ua[us], ia[ps], ia[ns]
    mf = bpr_logsigmoid(u, pp, nn_)
    reg = 0.5 * ((u ** 2).sum() + (pp ** 2).sum() + (nn_ ** 2).sum()) / batch_size
    cl = infonce(side[n_users:][ps], content[n_users:][ps], tau) + infonce(side[:n_users][us], content[:n_users][us], tau)
    return mf + reg_weight * reg + cl_weight * cl


# --------------------------------------------------------------------------------------------
# SMORE (models/smore.py) -- torch-CPU restatement, pinned by tests/golden/smore.npz.  Graph build is
# MGCN's (same get_adj_mat arithmetic, smore.py:162-184; same build_knn_normalized_graph) plus the
# max-pooled fusion graph; the spectrum step is restated WITHOUT an FFT library, as the real DFT
# matrices it stands for, so that the device path can use plain GEMMs.
# --------------------------------------------------------------------------------------------


def smore_fusion_graph(image_idx, image_val, text_idx, text_val, n):
    """SMORE.max_pool_fusion, smore.py:139-160: union of the two (coalesced) kNN edge sets, value =
    max over the modalities that have the edge.  Returns (idx sorted row-major, val)."""
    ki = np.asarray(image_idx[0], np.int64) * n + np.asarray(image_idx[1], np.int64)
    kt = np.asarray(text_idx[0], np.int64) * n + np.asarray(text_idx[1], np.int64)
    keys, inv = np.unique(np.concatenate([ki, kt]), return_inverse=True)
    vi = np.full(keys.shape[0], -np.inf, np.float32)
    vt = np.full(keys.shape[0], -np.inf, np.float32)
    np.maximum.at(vi, inv[:ki.shape[0]], np.asarray(image_val, np.float32))   # kNN rows hold distinct columns
    np.maximum.at(vt, inv[ki.shape[0]:], np.asarray(text_val, np.float32))
    return np.stack([keys // n, keys % n]), np.maximum(vi, vt)


def rdft_matrices(d):
    """Real matrices of torch.fft.rfft / irfft (norm='ortho', n=d, d even) along the last dim:
    rfft(x) = x @ C + i x @ S  with C, S [d, d/2+1];  irfft(Re, Im) = Re @ Ci + Im @ Si  with Ci, Si
    [d/2+1, d] (the imaginary parts of the DC and Nyquist bins do not contribute, as in C2R FFTs)."""
    n = torch.arange(d, dtype=torch.float64).unsqueeze(1)
    k = torch.arange(d // 2 + 1, dtype=torch.float64).unsqueeze(0)
    ang = 2.0 * np.pi * n * k / d
    s = 1.0 / np.sqrt(d)
    C, S = torch.cos(ang) * s, -torch.sin(ang) * s
    wk = torch.full((d // 2 + 1, 1), 2.0, dtype=torch.float64)
    wk[0, 0] = wk[-1, 0] = 1.0
    Ci, Si = wk * torch.cos(ang).t() * s, -wk * torch.sin(ang).t() * s
    Si[0, :] = Si[-1, :] = 0.0
    return C.float(), S.float(), Ci.float(), Si.float()


def smore_spectrum(image_feats, text_feats, w_img, w_txt, w_fus):
    """SMORE.spectrum_convolution, smore.py:193-211; w_*: [1, d/2+1, 2] (real, imag)."""
    C, S, Ci, Si = rdft_matrices(image_feats.shape[1])
    ir, ii = image_feats @ C, image_feats @ S
    tr, ti = text_feats @ C, text_feats @ S

    def filt(re, im, w):
        wr, wi = w[0, :, 0], w[0, :, 1]
        return (re * wr - im * wi) @ Ci + (re * wi + im * wr) @ Si
    fr, fi = tr * ir - ti * ii, tr * ii + ti * ir            # text_fft * image_fft
    return filt(ir, ii, w_img), filt(tr, ti, w_txt), filt(fr, fi, w_fus)


def smore_forward(p, adj, R, image_adj, text_adj, fusion_adj, n_users, n_ui_layers, n_layers, drop=None):
    """SMORE.forward(train=True), smore.py:213-297.  `drop`: None (eval) or the three dropout
    multipliers (mask / (1 - p)) applied to the image / text / fusion preference gates."""
    image_feats = F.linear(p["image_embedding.weight"], p["image_trs.weight"], p["image_trs.bias"])
    text_feats = F.linear(p["text_embedding.weight"], p["text_trs.weight"], p["text_trs.bias"])
    gate = lambda name, x: torch.sigmoid(F.linear(x, p[name + ".0.weight"], p[name + ".0.bias"]))
    ic, tc, fc = smore_spectrum(image_feats, text_feats, p["image_complex_weight"], p["text_complex_weight"],
                                p["fusion_complex_weight"])
    item_w, user_w = p["item_id_embedding.weight"], p["user_embedding.weight"]
    views = [item_w * gate("gate_v", ic), item_w * gate("gate_t", tc), item_w * gate("gate_f", fc)]
    ego = torch.cat([user_w, item_w], dim=0)
    layers = [ego]
    for _ in range(n_ui_layers):
        ego = torch.sparse.mm(adj, ego)
        layers.append(ego)
    content = torch.stack(layers, dim=1).mean(dim=1)
    embeds = []
    for x, g in zip(views, (image_adj, text_adj, fusion_adj)):
        for _ in range(n_layers):
            x = torch.sparse.mm(g, x)
        embeds.append(torch.cat([torch.sparse.mm(R, x), x], dim=0))
    image_embeds, text_embeds, fusion_embeds = embeds

    def query(name, x):
        h = torch.tanh(F.linear(x, p[name + ".0.weight"], p[name + ".0.bias"]))
        return F.linear(h, p[name + ".2.weight"])
    agg_image = torch.softmax(query("query_v", fusion_embeds), dim=-1) * image_embeds
    agg_text = torch.softmax(query("query_t", fusion_embeds), dim=-1) * text_embeds
    prefer = [gate("gate_image_prefer", content), gate("gate_text_prefer", content), gate("gate_fusion_prefer", content)]
    if drop is not None:
        prefer = [a * m for a, m in zip(prefer, drop)]
    side = torch.stack([prefer[0] * agg_image, prefer[1] * agg_text, prefer[2] * fusion_embeds]).mean(dim=0)
    out = content + side
    return out[:n_users], out[n_users:], side, content


# --------------------------------------------------------------------------------------------
# SELFCFED_LGN (models/selfcfed_lgn.py, common/encoders.py) and BPR (models/bpr.py) -- torch-CPU
# restatements pinned by tests/golden/selfcf.npz.
# --------------------------------------------------------------------------------------------


def selfcf_sparse_dropout(idx, val, n, rate, keep):
    """LightGCN_Encoder.sparse_dropout, encoders.py:77-90, with the draw injected: keep = floor(1 - rate +
    rand(nnz)) as bool over the stored (row-major) entries; kept values * 1/(1 - rate)."""
    keep = torch.as_tensor(keep, dtype=torch.bool)
    i, v = torch.as_tensor(idx)[:, keep], torch.as_tensor(val)[keep]
    return torch.sparse_coo_tensor(i, v, (n, n)) * (1. / (1 - rate))


def selfcf_encoder(user_emb, item_emb, adj, n_layers):
    """LightGCN_Encoder.forward / get_embedding without the row gather, encoders.py:92-139."""
    ego = torch.cat([user_emb, item_emb], 0)
    layers = [ego]
    for _ in range(n_layers):
        ego = torch.sparse.mm(adj, ego)
        layers.append(ego)
    out = torch.stack(layers, dim=1).mean(dim=1)
    return out[:user_emb.shape[0]], out[user_emb.shape[0]:]


def selfcf_loss(p, adj_dropped, batch, n_layers, reg_weight, target_mult_u, target_mult_i):
    """SELFCFED_LGN.calculate_loss, selfcfed_lgn.py:40-68.  target_mult_*: the F.dropout multipliers
    (mask / (1 - p)) of the detached target branch."""
    u_all, i_all = selfcf_encoder(p["online_encoder.embedding_dict.user_emb"],
                                  p["online_encoder.embedding_dict.item_emb"], adj_dropped, n_layers)
    u_online, i_online = u_all[torch.as_tensor(batch[0])], i_all[torch.as_tensor(batch[1])]
    u_target, i_target = u_online.detach() * target_mult_u, i_online.detach() * target_mult_i
    reg = 0.5 * torch.sum(u_online ** 2) + 0.5 * torch.sum(i_online ** 2)             # L2Loss, loss.py:54-62
    pu = F.linear(u_online, p["predictor.weight"], p["predictor.bias"])
    pi = F.linear(i_online, p["predictor.weight"], p["predictor.bias"])
    loss_ui = -F.cosine_similarity(pu, i_target, dim=-1).mean() / 2
    loss_iu = -F.cosine_similarity(pi, u_target, dim=-1).mean() / 2
    return loss_ui + loss_iu + reg_weight * reg


def selfcf_scores(p, adj, users, n_layers):
    """SELFCFED_LGN.full_sort_predict, selfcfed_lgn.py:51-54,70-77."""
    u, i = selfcf_encoder(p["online_encoder.embedding_dict.user_emb"],
                          p["online_encoder.embedding_dict.item_emb"], adj, n_layers)
    pu = F.linear(u, p["predictor.weight"], p["predictor.bias"])
    pi = F.linear(i, p["predictor.weight"], p["predictor.bias"])
    us = torch.as_tensor(users)
    return torch.matmul(pu[us], i.t()) + torch.matmul(u[us], pi.t())


def bpr_mf_loss(user_w, item_w, batch, reg_weight):
    """BPR.calculate_loss, bpr.py:74-91: BPRLoss (loss.py:33-35) + reg_weight * EmbLoss (loss.py:46-51:
    unsquared Frobenius norms / rows of the last argument)."""
    us, ps, ns = (torch.as_tensor(b) for b in batch)
    u, pp, nn_ = user_w[us], item_w[ps], item_w[ns]
    mf = -torch.log(1e-10 + torch.sigmoid((u * pp).sum(1) - (u * nn_).sum(1))).mean()
    reg = (torch.norm(u, p=2) + torch.norm(pp, p=2) + torch.norm(nn_, p=2)) / nn_.shape[0]
    return mf + reg_weight * reg


# --------------------------------------------------------------------------------------------
# PGL, mode 'local' (models/pgl.py) -- torch-CPU restatement pinned by tests/golden/pgl.npz.  Graph
# build = FREEDOM's (norm_adj_coo, edge_norm_values, masked_adj_coo with 30 % kept, freedom_mm_adj).
# --------------------------------------------------------------------------------------------


def pgl_forward(p, adj, mm_adj, n_users, n_ui_layers, n_mm_layers):
    """PGL.forward, pgl.py:188-213: rows are [normalize(image_trs(V)) | normalize(text_trs(T))]."""
    image_feats = F.normalize(F.linear(p["image_embedding.weight"], p["image_trs.weight"], p["image_trs.bias"]))
    text_feats = F.normalize(F.linear(p["text_embedding.weight"], p["text_trs.weight"], p["text_trs.bias"]))
    user_embeds = torch.cat([p["user_image.weight"], p["user_text.weight"]], dim=1)
    item_embeds = torch.cat([image_feats, text_feats], dim=1)
    h = item_embeds
    for _ in range(n_mm_layers):
        h = torch.sparse.mm(mm_adj, h)
    ego = torch.cat((user_embeds, item_embeds), dim=0)
    layers = [ego]
    for _ in range(n_ui_layers):
        ego = torch.sparse.mm(adj, ego)
        layers.append(ego)
    out = torch.stack(layers, dim=1).mean(dim=1)
    return out[:n_users], out[n_users:] + h


def pgl_loss(ua, ia, batch, reg_weight, drop_mults, tau=0.2):
    """PGL.calculate_loss, pgl.py:233-250.  drop_mults: the four F.dropout multipliers (mask / (1 - p)) in
    call order: two views of the user rows, two views of the positive-item rows."""
    us, ps, ns = (torch.as_tensor(b) for b in batch)
    u, pp, nn_ = ua[us], ia[ps], ia[ns]
    mf = bpr_logsigmoid(u, pp, nn_)
    cl = (infonce(u * drop_mults[0], u * drop_mults[1], tau) + infonce(pp * drop_mults[2], pp * drop_mults[3], tau)) / 2
    return mf + reg_weight * cl


# --------------------------------------------------------------------------------------------
# LGMRec (models/lgmrec.py) -- torch-CPU restatement pinned by tests/golden/lgmrec.npz.  Gumbel noise and dropout
# multipliers are inputs (the reference draws them; the golden run records them).
# --------------------------------------------------------------------------------------------


def lgmrec_forward(p, adj_R, norm_adj, num_inters, n_users, n_ui_layers, n_mm_layers, n_hyper_layer, alpha, gumbel,
                   drop=None, tau=0.2):
    """LGMRec.forward, lgmrec.py:108-149.  adj_R: binary [U, I] sparse interaction matrix; num_inters: 1 / (degree +
    1e-7) per node (lgmrec.py:43-44); gumbel: the four noise tensors in call order (iv, uv, it, ut); drop: None or
    the four dropout multipliers in call order (iv, uv, it, ut).  Returns (users, items, [uv, iv, ut, it])."""
    def hyper(feat, w, g_i, g_u):
        i_h = torch.mm(feat, w)
        u_h = torch.sparse.mm(adj_R, i_h)
        return ((i_h + g_i) / tau).softmax(1), ((u_h + g_u) / tau).softmax(1)
    iv_h, uv_h = hyper(p["image_embedding.weight"], p["v_hyper"], gumbel[0], gumbel[1])
    it_h, ut_h = hyper(p["text_embedding.weight"], p["t_hyper"], gumbel[2], gumbel[3])
    ego = torch.cat((p["user_embedding.weight"], p["item_id_embedding.weight"]), dim=0)
    layers = [ego]
    for _ in range(n_ui_layers):
        ego = torch.sparse.mm(norm_adj, ego)
        layers.append(ego)
    cge = torch.stack(layers, dim=1).mean(dim=1)

    def mge(feat, trs):
        item_feats = torch.mm(feat, trs)
        user_feats = torch.sparse.mm(adj_R, item_feats) * num_inters[:n_users]
        x = torch.cat([user_feats, item_feats], dim=0)
        for _ in range(n_mm_layers):
            x = torch.sparse.mm(norm_adj, x)
        return x
    v_feats = mge(p["image_embedding.weight"], p["item_image_trs"])
    t_feats = mge(p["text_embedding.weight"], p["item_text_trs"])
    lge = cge + F.normalize(v_feats) + F.normalize(t_feats)
    if drop is not None:
        iv_h, uv_h, it_h, ut_h = iv_h * drop[0], uv_h * drop[1], it_h * drop[2], ut_h * drop[3]

    def hgnn(i_h, u_h, embeds):                      # HGNNLayer.forward, lgmrec.py:205-213
        i_ret = embeds
        for _ in range(n_hyper_layer):
            lat = torch.mm(i_h.t(), i_ret)
            i_ret = torch.mm(i_h, lat)
            u_ret = torch.mm(u_h, lat)
        return u_ret, i_ret
    uv, iv = hgnn(iv_h, uv_h, cge[n_users:])
    ut, it = hgnn(it_h, ut_h, cge[n_users:])
    ghe = torch.cat([uv, iv], dim=0) + torch.cat([ut, it], dim=0)
    out = lge + alpha * F.normalize(ghe)
    return out[:n_users], out[n_users:], [uv, iv, ut, it]


def lgmrec_ssl(emb1, emb2, all_emb, tau=0.2):
    """LGMRec.ssl_triple_loss, lgmrec.py:157-164: every row of all_emb is a negative; SUM over the batch."""
    n1, n2, na = F.normalize(emb1), F.normalize(emb2), F.normalize(all_emb)
    pos = torch.exp((n1 * n2).sum(dim=1) / tau)
    ttl = torch.exp(torch.matmul(n1, na.t()) / tau).sum(dim=1)
    return -torch.log(pos / ttl).sum()


def lgmrec_loss(ua, ia, hyper, batch, cl_weight, reg_weight):
    """LGMRec.calculate_loss, lgmrec.py:173-194."""
    us, ps, ns = (torch.as_tensor(b) for b in batch)
    u, pp, nn_ = ua[us], ia[ps], ia[ns]
    uv, iv, ut, it = hyper
    bpr = bpr_logsigmoid(u, pp, nn_)
    hcl = lgmrec_ssl(uv[us], ut[us], ut) + lgmrec_ssl(iv[ps], it[ps], it)
    reg = (torch.norm(u, p=2) + torch.norm(pp, p=2) + torch.norm(nn_, p=2)) / nn_.shape[0]
    return bpr + cl_weight * hcl + reg_weight * reg
