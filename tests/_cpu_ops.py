"""TEST-ONLY stand-in for `mmrec_amd.hip_ops` on a box without a GPU.

The product has no CPU path: every op in `mmrec_amd/hip_ops.py` raises on a CPU tensor and
`mmrec_amd/_lib.py` raises when `libmmrec_hip.so` is missing.  The HOST logic around the kernels --
graph construction, which op a model calls with which operands, loss assembly, parameter names, the
evaluation loop -- is still worth checking in the `-m "not gpu"` suite against the reference's golden
outputs.  The `cpu_ops` fixture below swaps the op entry points of `mmrec_amd.hip_ops` for plain
torch-CPU restatements of what each op is specified to compute (same signatures, same results up to
fp32 summation order, and with the same ARGUMENT CONTRACT: dtypes, contiguity of index vectors, row widths the
kernels accept -- a call the device would reject fails here too) for the duration of ONE test.  Nothing under `mmrec_amd/` imports this file,
and the `-m gpu` tests never use it: there the same model code runs on the HIP kernels.
"""
from __future__ import annotations

import numpy as np
import pytest
import torch
import torch.nn.functional as F

EMB_DIM = 64
SLICE_WIDTHS = (8, 16, 32)        # hip_ops.SLICE_WIDTHS: one feature slice of a 64-wide table (mmrec_spmm_csr_f32 accepts them)
BPR_LOGSIG, BPR_GAMMA = 0, 1


# ---- the argument contract of the real ops (mmrec_amd/hip_ops.py `_chk` + the C entry points' shape rules), so that a
# ---- model that would be rejected on the device is rejected here too
def _ids(t, name):
    assert isinstance(t, torch.Tensor) and t.dtype == torch.int64 and t.dim() == 1 and t.is_contiguous(), \
        "%s must be a contiguous 1-d int64 tensor" % name


def _mat(t, name, width_multiple=None, width=None):
    assert isinstance(t, torch.Tensor) and t.dtype == torch.float32 and t.dim() == 2, "%s must be 2-d fp32" % name
    if width_multiple:
        assert t.shape[1] % width_multiple == 0, "%s: row width %d is not a multiple of %d" % (name, t.shape[1], width_multiple)
    if width:
        assert t.shape[1] == width, "%s: row width %d != %d" % (name, t.shape[1], width)


def _spmm_args(g, X, Z=None):
    _mat(X, "X")
    assert X.shape[0] >= g.n_cols, "X has %d rows, the graph %d columns" % (X.shape[0], g.n_cols)
    assert -(-X.shape[1] // EMB_DIM) <= 6, "row width %d > 384" % X.shape[1]
    assert X.shape[1] % 4 == 0
    if Z is not None:
        _mat(Z, "Z")
        assert Z.shape[0] >= g.n_rows and Z.shape[1] == X.shape[1], "Z must be [>= n_rows, d]"


class CsrGraph:
    """Same constructors / attributes as hip_ops.CsrGraph, arithmetic by index_add on the CPU."""

    def __init__(self, rowptr, colidx, vals, n_rows, n_cols, symmetric=False, long_row_threshold=64,
                 rowptr_host=None):
        self.rowptr, self.colidx, self.vals = rowptr.to(torch.int32), colidx.to(torch.int32), vals.float()
        self.n_rows, self.n_cols, self.nnz = int(n_rows), int(n_cols), int(colidx.numel())
        self.symmetric, self.long_row_threshold = bool(symmetric), int(long_row_threshold)
        self.rowptr_host = np.ascontiguousarray(
            rowptr_host if rowptr_host is not None else self.rowptr.cpu().numpy(), dtype=np.int32)
        self._t = self if symmetric else None
        self.n_long = self.n_chunks = 0

    @classmethod
    def from_coo_host(cls, idx, val, n_rows, n_cols, device, symmetric=False, **kw):
        idx = np.asarray(idx, dtype=np.int64)
        val = np.asarray(val, dtype=np.float32)
        order = np.argsort(idx[0], kind="stable")
        rowptr = np.zeros(n_rows + 1, dtype=np.int64)
        np.cumsum(np.bincount(idx[0], minlength=n_rows), out=rowptr[1:])
        return cls(torch.from_numpy(rowptr.astype(np.int32)), torch.from_numpy(idx[1][order].astype(np.int32)),
                   torch.from_numpy(val[order]), n_rows, n_cols, symmetric=symmetric, **kw)

    @classmethod
    def from_coo_device(cls, rows, cols, vals, n_rows, n_cols, symmetric=False, **kw):
        idx = np.stack([rows.cpu().numpy(), cols.cpu().numpy()])
        return cls.from_coo_host(idx, vals.detach().cpu().numpy(), n_rows, n_cols, "cpu", symmetric=symmetric, **kw)

    def rows(self):
        return torch.repeat_interleave(torch.arange(self.n_rows), torch.from_numpy(np.diff(self.rowptr_host.astype(np.int64))))

    def to_coo_host(self):
        return (np.stack([self.rows().numpy(), self.colidx.numpy().astype(np.int64)]), self.vals.numpy())

    def transpose(self):
        if self._t is None:
            idx, val = self.to_coo_host()
            self._t = CsrGraph.from_coo_host(idx[::-1], val, self.n_cols, self.n_rows, "cpu")
            self._t._t = self
        return self._t

    def row_block(self, r0, r1):
        rp = self.rowptr_host.astype(np.int64)
        s, e = int(rp[r0]), int(rp[r1])
        rph = (rp[r0:r1 + 1] - s).astype(np.int32)
        return CsrGraph(torch.from_numpy(rph), self.colidx[s:e], self.vals[s:e], r1 - r0, self.n_cols, rowptr_host=rph)

    def matmul(self, X, vals=None):
        v = self.vals if vals is None else vals
        out = torch.zeros(self.n_rows, X.shape[1], dtype=torch.float32)
        return out.index_add(0, self.rows(), v.unsqueeze(1) * X[self.colidx.long()])


def spmm_raw(g, X, Y=None, Z=None, acc_in=None, acc_out=None, alpha=1.0, beta=1.0, acc_scale=1.0):
    with torch.no_grad():
        y = alpha * g.matmul(X[:g.n_cols] if X.shape[0] > g.n_cols else X)
        if Z is not None:
            y = y + beta * Z[:g.n_rows]
        if Y is not None:
            Y[:g.n_rows].copy_(y)
        if acc_out is not None:
            acc_out[:g.n_rows].copy_(acc_scale * ((acc_in[:g.n_rows] if acc_in is not None else 0) + y))
    return Y if Y is not None else acc_out


def spmm(g, X, Z=None):
    _spmm_args(g, X, Z)
    y = g.matmul(X[:g.n_cols])
    return y if Z is None else y + Z


def spmm_rows(g, X, rows, Z_rows=None):
    _ids(rows, "rows")
    y = spmm(g, X).index_select(0, rows)
    return y if Z_rows is None else y + Z_rows


def lightgcn_mean_parts_rows(g, parts, n_layers, rows):
    _ids(rows, "rows")
    return torch.cat(lightgcn_mean_parts(g, parts, n_layers), dim=0).index_select(0, rows)


def lightgcn_mean(g, E0, n_layers):
    _spmm_args(g, E0)
    _mat(E0, "E0", width_multiple=None if E0.shape[1] in SLICE_WIDTHS else EMB_DIM)
    assert g.n_rows == g.n_cols == E0.shape[0], "layer mean needs a square graph over the rows of E0"
    outs, cur = [E0], E0
    for _ in range(int(n_layers)):
        cur = g.matmul(cur)
        outs.append(cur)
    return torch.stack(outs, 1).mean(1)


def lightgcn_mean_parts(g, parts, n_layers):
    sizes = [p.shape[0] for p in parts]
    return tuple(lightgcn_mean(g, torch.cat(list(parts), dim=0), n_layers).split(sizes))


def layergcn_sum(g, E0, n_layers):
    _spmm_args(g, E0)
    _mat(E0, "E0", width=EMB_DIM)
    cur, outs = E0, []
    for _ in range(int(n_layers)):
        cur = g.matmul(cur)
        w = F.cosine_similarity(cur, E0, dim=-1)
        cur = w.unsqueeze(1) * cur
        outs.append(cur)
    return torch.stack(outs, 0).sum(0)


def layergcn_sum_parts(g, parts, n_layers):
    sizes = [p.shape[0] for p in parts]
    return tuple(layergcn_sum(g, torch.cat(list(parts), dim=0), n_layers).split(sizes))


def bpr_loss(U, I, users, pos, neg, variant=BPR_LOGSIG, reduction="mean"):
    _mat(U, "U", width_multiple=EMB_DIM), _mat(I, "I", width=U.shape[1])
    _ids(users, "users"), _ids(pos, "pos"), _ids(neg, "neg")
    assert users.numel() == pos.numel() == neg.numel()
    x = (U[users] * I[pos]).sum(1) - (U[users] * I[neg]).sum(1)
    per = -F.logsigmoid(x) if variant == BPR_LOGSIG else -torch.log(1e-10 + torch.sigmoid(x))
    if users.numel() == 0:
        return per.sum()
    return per.mean() if reduction == "mean" else per.sum()


class _SumOverRanks(torch.autograd.Function):
    """the caller's in-place sum over ranks of partial results whose consumer is replicated: the incoming gradient is the same
    on every rank and is each rank's own partial's gradient"""

    @staticmethod
    def forward(ctx, x, fn):
        out = x.contiguous().clone()
        fn(out)
        return out

    @staticmethod
    def backward(ctx, g):
        return g, None


def bpr_losses_shared_users(U, users, terms, variant=BPR_LOGSIG, reduction="mean", joint_grad=False, sum_over_ranks=None):
    if sum_over_ranks is None:
        return tuple(bpr_loss(U, I, users, pos, neg, variant, reduction) for I, pos, neg in terms)
    assert U.shape[1] % EMB_DIM == 0 or U.shape[1] in SLICE_WIDTHS, "slice width the kernels have"
    _ids(users, "users")
    u = U[users]
    dots = torch.stack([torch.stack(((u * t[p]).sum(1), (u * t[n]).sum(1))) for t, p, n in terms])
    dots = _SumOverRanks.apply(dots, sum_over_ranks)
    out = []
    for j in range(len(terms)):
        x = dots[j, 0] - dots[j, 1]
        per = -F.logsigmoid(x) if variant == BPR_LOGSIG else -torch.log(1e-10 + torch.sigmoid(x))
        out.append(per.mean() if reduction == "mean" and users.numel() else per.sum())
    return tuple(out)


def bpr_weighted_total(U, users, terms, weights, variant=BPR_LOGSIG, reduction="mean", joint_grad=False):
    """hip_ops.bpr_weighted_total: sum_t w_t bpr_loss(U, I_t, users, pos_t, neg_t)"""
    total = 0.0
    for w, (I, pos, neg) in zip(weights, terms):
        total = total + w * bpr_loss(U, I, users, pos, neg, variant, reduction)
    return total


def infonce(E1, E2, ids, tau):
    _mat(E1, "E1", width=EMB_DIM), _mat(E2, "E2", width=EMB_DIM), _ids(ids, "ids")
    assert E1.shape == E2.shape
    v1, v2 = F.normalize(E1[ids], dim=1), F.normalize(E2[ids], dim=1)
    pos = torch.exp((v1 * v2).sum(-1) / tau)
    ttl = torch.exp(v1 @ v2.t() / tau).sum(1)
    return -torch.log(pos / ttl).mean()


def cosine_mean(X, ix, Y, iy):
    _mat(X, "X", width_multiple=EMB_DIM), _mat(Y, "Y", width=X.shape[1])
    for t, nm in ((ix, "ix"), (iy, "iy")):
        if t is not None:
            _ids(t, nm)
    x = X if ix is None else X[ix]
    y = (Y if iy is None else Y[iy]).detach()
    return F.cosine_similarity(x, y, dim=-1).mean()


def lightgcn_mean_rows_then_item_rows(g, user_table, item_table, n_layers, users, item_rows, mm):
    b, nu = users.shape[0], user_table.shape[0]
    at = lightgcn_mean_parts_rows(g, (user_table, item_table), n_layers, torch.cat((users, item_rows + nu)))
    return at[:b], spmm_rows(mm, item_table, item_rows, Z_rows=at[b:])


def cosine_means(terms):
    """hip_ops.cosine_means: sum_t w_t mean cos(X_t[ix_t], Y_t[iy_t])"""
    total = 0.0
    for X, ix, Y, iy, w in terms:
        total = total + w * cosine_mean(X, ix, Y, iy)
    return total


def gather_sqnorm(E, ids):
    _mat(E, "E"), _ids(ids, "ids")
    return (E[ids] ** 2).sum()


def row_normalize(X, eps=1e-12):
    return F.normalize(X, p=2, dim=1, eps=eps)


def cat_leaky(A, B, R=None, slope=0.01):
    """hip_ops.cat_leaky: cat((leaky_relu(A), leaky_relu(B) + R), dim=1)"""
    x_hat = F.leaky_relu(B, slope)
    return torch.cat((F.leaky_relu(A, slope), x_hat if R is None else x_hat + R), dim=1)


def rows_reg(terms, mode, scale=1.0):
    """hip_ops.rows_reg (ABI 14): scale * sum_t ||E_t[ids_t]||_F^2 (mode 0) or scale * sum_t ||E_t[ids_t]||_F (mode 1)"""
    total = 0.0
    for E, ids in terms:
        s = (E ** 2).sum() if ids is None else gather_sqnorm(E, ids)
        total = total + (s if mode == 0 else torch.sqrt(s))
    return scale * total


def linear(X, W, b=None):
    _mat(X, "X"), _mat(W, "W")
    assert W.shape[1] == X.shape[1]
    if W.shape[0] == 64:
        assert X.shape[1] % 4 == 0, "projection kernel: input width %d is not a multiple of 4" % X.shape[1]
    else:
        assert W.shape[0] % 64 == 0 and X.shape[1] % 32 == 0, "wide linear needs W [64 j, F], F % 32 == 0"
    return F.linear(X, W, b)


class TopkCandidates:
    """hip_ops.TopkCandidates: a candidate table whose filter-side preparation is shared by many calls (here: the table)"""

    def __init__(self, C):
        _mat(C, "C")
        self.C, self.prepared = C.contiguous(), None


def topk_hint_served(nc, kd, k):
    return k <= 128 and kd in (64, 128) and 4096 <= nc <= 1000000


def hint_served(C, Q, k):
    from mmrec_amd import hip_ops
    Ct = C.C if isinstance(C, TopkCandidates) else C
    return hip_ops.topk_hint_served(Ct.shape[0], Q.shape[1], k) and Ct.shape[0] >= k


def topk_hint_width(k):
    return 64 if k <= 64 else 128


def score_topk(Q, C, k, mask_rowptr=None, mask_col=None, return_values=False, use_filter=True, hint=None, hint_rows=None,
               queue_counts=None, hint_cold=False, hint_update=True):
    """(a warm call's list never changes the result: the stand-in checks its shape, ignores its content and -- like the kernel --
    leaves the call's top-k in the rows, -1 beyond)"""
    if hint is not None:
        assert hint.dtype == torch.int32 and hint.dim() == 2 and k <= hint.shape[1] <= 128
        assert (hint_rows is None and hint.shape[0] == Q.shape[0]) or (hint_rows.dtype == torch.int64 and hint_rows.numel() == Q.shape[0])
        if hint_update and hint_served(C, Q, k):
            out = score_topk(Q, C, k, mask_rowptr, mask_col)
            rows = hint_rows if hint_rows is not None else torch.arange(Q.shape[0])
            hint[rows] = -1
            hint[rows, :k] = out.to(torch.int32)
    if isinstance(C, TopkCandidates):
        C = C.C
    _mat(Q, "Q"), _mat(C, "C", width=Q.shape[1])
    assert Q.shape[1] % 4 == 0, "inner dim %d is not a multiple of 4" % Q.shape[1]
    if k > 128 or (k > 64 and Q.shape[1] % 32):     # MMREC_TOPK_MAX / MMREC_TOPK_MAX_OTHER
        from mmrec_amd._lib import MMRecHipError
        raise MMRecHipError("score_topk: k = %d is served for row widths that are a multiple of 32 only (<= 128)" % k)
    if mask_rowptr is not None:
        assert mask_rowptr.dtype == torch.int32 and (mask_col is None or mask_col.dtype == torch.int32)
    s = Q @ C.t()
    if mask_rowptr is not None:
        rp = mask_rowptr.long()
        rows = torch.repeat_interleave(torch.arange(Q.shape[0]), rp[1:] - rp[:-1])
        if rows.numel():
            s[rows, mask_col.long()[:rows.numel()]] = -1e10
    val, idx = torch.sort(s, dim=1, descending=True, stable=True)      # ties: lower id first
    return (idx[:, :k].contiguous(), val[:, :k].contiguous()) if return_values else idx[:, :k].contiguous()


def degree_count(ids, n_bins):
    return torch.bincount(ids, minlength=n_bins).to(torch.int32)


def edge_norm_values(eu, ei, n_users, n_items):
    du = torch.bincount(eu, minlength=n_users).float() + 1e-7
    di = torch.bincount(ei, minlength=n_items).float() + 1e-7
    return torch.pow(du, -0.5)[eu] * torch.pow(di, -0.5)[ei]


def bipartite_graph_from_edges(eu, ei, n_users, n_items, long_row_threshold=64):
    w = edge_norm_values(eu, ei, n_users, n_items)
    n = n_users + n_items
    rows, cols = torch.cat([eu, ei + n_users]), torch.cat([ei + n_users, eu])
    return CsrGraph.from_coo_device(rows, cols, torch.cat([w, w]), n, n, symmetric=True)


class DynGraph:
    def __init__(self, rows, cols, n_rows, n_cols, long_row_threshold=64):
        _ids(rows, "rows"), _ids(cols, "cols")
        self.rows, self.cols, self.n_rows, self.n_cols = rows, cols, int(n_rows), int(n_cols)


def spmm_vals(dyn, X, vals):
    _mat(X, "X", width=EMB_DIM)
    assert X.shape[0] >= dyn.n_cols and vals.dim() == 1 and vals.numel() == dyn.rows.numel()
    out = torch.zeros(dyn.n_rows, X.shape[1], dtype=torch.float32)
    return out.index_add(0, dyn.rows, vals.unsqueeze(1) * X[dyn.cols])


_PATCHED = ("CsrGraph", "spmm_raw", "spmm", "spmm_rows", "lightgcn_mean", "lightgcn_mean_parts", "lightgcn_mean_parts_rows", "lightgcn_mean_rows_then_item_rows", "layergcn_sum", "layergcn_sum_parts",
            "bpr_loss",
            "bpr_losses_shared_users", "bpr_weighted_total", "infonce",
            "gather_sqnorm", "rows_reg", "cat_leaky", "row_normalize", "cosine_mean", "cosine_means", "linear", "score_topk", "topk_hint_served", "topk_hint_width", "TopkCandidates", "degree_count", "edge_norm_values",
            "bipartite_graph_from_edges", "DynGraph", "spmm_vals")


@pytest.fixture
def cpu_ops(monkeypatch):
    """mmrec_amd.hip_ops' op entry points -> the torch-CPU restatements above, for this test only."""
    import sys
    from mmrec_amd import hip_ops
    me = sys.modules[__name__]
    for name in _PATCHED:
        monkeypatch.setattr(hip_ops, name, getattr(me, name))
    return me


def install():
    """the same swap, permanently, for a process that exists only for a test (torch.multiprocessing workers of the gloo
    tests: pytest fixtures do not reach them)"""
    import sys
    from mmrec_amd import hip_ops
    me = sys.modules[__name__]
    for name in _PATCHED:
        setattr(hip_ops, name, getattr(me, name))
    return me

