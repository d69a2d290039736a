"""Host-side graph construction used at model init (the reference also builds these on the host,
with scipy dok/coo -- freedom.py:102-126 takes 0.44 s on Baby and ~40 s extrapolated to 20M nnz;
this is vectorised numpy, 2 s at 20M nnz).  Per-epoch rebuilds run on the device (hip_ops)."""
from __future__ import annotations

import numpy as np
import torch

from . import hip_ops


def sym_norm_coo(eu, ei, n_users, n_items):
    """D^-1/2 A D^-1/2 of UNIQUE (user,item) edges, float64 -> float32 exactly as get_norm_adj_mat
    (freedom.py:113-124): degree + 1e-7, pow(-0.5), (d_r * 1) * d_c.  Returns (rows, cols, vals) of the
    2E-entry symmetric COO sorted by (row, col): user rows first, then item rows."""
    eu = np.asarray(eu, dtype=np.int64)
    ei = np.asarray(ei, dtype=np.int64)
    du = np.bincount(eu, minlength=n_users).astype(np.float64) + 1e-7
    di = np.bincount(ei, minlength=n_items).astype(np.float64) + 1e-7
    v = (np.power(du, -0.5)[eu] * np.power(di, -0.5)[ei]).astype(np.float32)
    sorted_already = eu.shape[0] < 2 or not (np.any(np.diff(eu) < 0) or
                                             np.any((np.diff(eu) == 0) & (np.diff(ei) < 0)))
    o1 = np.arange(eu.shape[0]) if sorted_already else np.lexsort((ei, eu))
    eu1, ei1, v1 = eu[o1], ei[o1], v[o1]
    o2 = np.argsort(ei1, kind="stable")  # item-major; users stay ascending inside an item
    rows = np.concatenate([eu1, ei1[o2] + n_users])
    cols = np.concatenate([ei1 + n_users, eu1[o2]])
    return rows, cols, np.concatenate([v1, v1[o2]])


def unique_edges(rows, cols, n_items):
    """De-duplicate (user,item) pairs like the reference's python dict does (freedom.py:108-111)."""
    key = np.unique(np.asarray(rows, dtype=np.int64) * np.int64(n_items) + np.asarray(cols, dtype=np.int64))
    return key // n_items, key % n_items


def norm_adj_graph(inter_coo, n_users, n_items, device, **kw):
    """scipy COO train matrix (TrainDataLoader.inter_matrix) -> symmetric normalised CsrGraph."""
    eu, ei = unique_edges(inter_coo.row, inter_coo.col, n_items)
    r, c, v = sym_norm_coo(eu, ei, n_users, n_items)
    n = n_users + n_items
    return hip_ops.CsrGraph.from_coo_host(np.stack([r, c]), v, n, n, device, symmetric=True, **kw)


def knn_normalized_coo(feats, k):
    """kNN(k) graph of row-normalised features with the symmetric row-sum normalisation of
    freedom.py:79-100, neighbours found by the fused HIP score+top-K kernel (the [I, I] similarity
    matrix is never materialised).  Returns (indices[2, I*k] int64 device, values fp32 device)."""
    x = feats.detach().to(torch.float32)
    xn = x.div(torch.norm(x, p=2, dim=-1, keepdim=True)).contiguous()
    knn = hip_ops.score_topk(xn, xn, k)                      # [I, k], best first (self first)
    n = x.shape[0]
    rows = torch.arange(n, device=x.device).unsqueeze(1).expand(-1, k).reshape(-1)
    cols = knn.reshape(-1)
    row_sum = 1e-7 + torch.zeros(n, device=x.device).index_add_(0, rows, torch.ones_like(rows, dtype=torch.float32))
    r_inv = torch.pow(row_sum, -0.5)
    return torch.stack([rows, cols]), r_inv[rows] * r_inv[cols]


def sparse_coo_to_graph(sp, device, **kw):
    """torch sparse COO tensor (e.g. the reference's cached mm_adj_*.pt) -> CsrGraph, keeping
    duplicates and their order (uncoalesced sum of two kNN graphs, freedom.py:74)."""
    idx = sp._indices().cpu().numpy()
    val = sp._values().cpu().numpy().astype(np.float32)
    return hip_ops.CsrGraph.from_coo_host(idx, val, sp.shape[0], sp.shape[1], device, **kw)


def mask_to_csr_device(mask, n_rows, n_cols):
    """[2, n] (row, item) device mask -> (rowptr int32, sorted cols int32) without leaving the GPU."""
    if mask.shape[1] == 0:
        return (torch.zeros(n_rows + 1, dtype=torch.int32, device=mask.device),
                torch.zeros(1, dtype=torch.int32, device=mask.device))
    key, _ = torch.sort(mask[0] * n_cols + mask[1])
    rows = torch.div(key, n_cols, rounding_mode='floor')
    rowptr = torch.zeros(n_rows + 1, dtype=torch.int64, device=mask.device)
    rowptr[1:] = torch.cumsum(torch.bincount(rows, minlength=n_rows), 0)
    return rowptr.to(torch.int32), (key - rows * n_cols).to(torch.int32)


# ---- build-time relabelling for gather locality (config key `reorder`) -----------------------------
class BipartiteRelabelling:
    """A relabelling of the users and of the items of ONE dataset (`reorder`: 'community' | 'degree' | 'rcm';
    hip_ops.locality_order on the symmetric user-item graph, then each side ranked on its own so that users stay a block and
    items stay a block -- the two id tables stay two tables).  `perm_*[old] = new`, `inv_*[new] = old` (device int64).

    A model that keeps its tables in the relabelled space pays NOTHING per step for the locality (hip_ops.PermutedGraph pays
    two permutation passes over [N, 64] per propagation: more than a 3-layer propagation gains): ids are mapped where they
    enter the model (a batch's [3, B] ids, an evaluation batch's users and mask), graphs are relabelled once at build time
    with every row's nonzeros kept in their original order (relabel_graph: same sums, bit for bit), top-K ids and the
    state_dict leave in the ORIGINAL ids."""

    def __init__(self, base_graph, n_users, n_items, how, device):
        n = n_users + n_items
        idx, _ = base_graph.to_coo_host()
        perm = hip_ops.locality_order(base_graph.rowptr_host, idx[1], n, how, n_left=n_users, device=device)
        self.how, self.n_users, self.n_items = how, int(n_users), int(n_items)
        pu = np.argsort(np.argsort(perm[:n_users], kind="stable"), kind="stable").astype(np.int64)   # rank among the users
        pi = np.argsort(np.argsort(perm[n_users:], kind="stable"), kind="stable").astype(np.int64)
        self.perm_u_host, self.perm_i_host = pu, pi
        self.perm_u, self.perm_i = torch.from_numpy(pu).to(device), torch.from_numpy(pi).to(device)
        self.inv_u, self.inv_i = torch.argsort(self.perm_u), torch.argsort(self.perm_i)

    def node_perm_host(self):
        return np.concatenate([self.perm_u_host, self.n_users + self.perm_i_host])

    def tag(self):
        """names this relabelling in checkpoints: the mode and a checksum of the two permutations"""
        import zlib
        return "%s:%08x" % (self.how, zlib.crc32(self.perm_i_host.tobytes(), zlib.crc32(self.perm_u_host.tobytes())))

    def to(self, device):
        for k in ("perm_u", "perm_i", "inv_u", "inv_i"):
            setattr(self, k, getattr(self, k).to(device))
        return self


def relabel_graph(g, row_perm_host, col_perm_host=None):
    """CsrGraph with rows (and columns) renamed, every row's nonzeros in their ORIGINAL order (stable COO -> CSR): the row sums
    are the plain graph's bit for bit.  A non-symmetric graph's transpose is relabelled from the plain transpose for the same
    reason (the backward's sums keep their order too)."""
    col_perm_host = row_perm_host if col_perm_host is None else col_perm_host

    def one(src, rp, cp, symmetric):
        idx, val = src.to_coo_host()
        return hip_ops.CsrGraph.from_coo_host(np.stack([rp[idx[0]], cp[idx[1]]]), val, src.n_rows, src.n_cols, src.rowptr.device,
                                              symmetric=symmetric, long_row_threshold=src.long_row_threshold)
    out = one(g, row_perm_host, col_perm_host, g.symmetric)
    if not g.symmetric:
        t = one(g.transpose(), col_perm_host, row_perm_host, False)
        out._t, t._t = t, out
    return out


# ---- the reference's on-disk graph caches -----------------------------------------
def dense_adj_to_coo(adj):
    """LATTICE caches its kNN graphs as DENSE [I, I] tensors (`image_adj_{k}.pt`, lattice.py:64-87).
    -> (rows, cols, vals) of the non-zeros in row-major order (k per row)."""
    idx = torch.nonzero(adj, as_tuple=False)
    return idx[:, 0].contiguous(), idx[:, 1].contiguous(), adj[idx[:, 0], idx[:, 1]].contiguous()


def coo_to_dense_adj(rows, cols, vals, n):
    """inverse of dense_adj_to_coo: the tensor the reference would have saved (duplicates add)."""
    out = torch.zeros(n, n, dtype=vals.dtype)
    out.index_put_((rows.cpu(), cols.cpu()), vals.cpu(), accumulate=True)
    return out


DENSE_CACHE_LIMIT_BYTES = 1 << 30   # do not write [I, I] caches above 1 GiB (Sports would be 1.35 GB)


def load_cached_adj(path):
    """torch.load of a reference cache file, or None."""
    import os
    if not os.path.exists(path):
        return None
    return torch.load(path, map_location="cpu", weights_only=False)
