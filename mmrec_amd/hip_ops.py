"""Host side of the MI355X hot path: torch tensors in, HIP kernels (libmmrec_hip.so, C ABI) out.

Every op here enqueues hand-written gfx950 kernels on torch's current stream through ctypes and is
wrapped in a `torch.autograd.Function` so the reference's `Trainer` (one Adam over
`model.parameters()`, trainer.py:111-128) works unchanged.  There is no CPU or eager fallback: a
missing library or a CPU tensor raises.

Reference call sites each op replaces are cited in include/mmrec_hip.h.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _lib

EMB_DIM = 64
# The 4096 -> 64 projection's forward on the 16-bit matrix cores with split operands (x = hi + 2^-11 lo' in fp16, three fp16 MFMA
# products per 16 k; as accurate against float64 as the fp32-MFMA kernel, which is bound by the 1/16-rate fp32 matrix pipe and
# not by the stream of X).  Rows / weights outside fp16's range -- |.| >= 65520, inf, NaN, a whole row below 2^-10 -- are found on
# the device and recomputed by the fp32 kernel inside the same call (csrc/gemm.hip), so the result is fp32-accurate over all of
# fp32's range.  False (config `hip_linear_split: False`): the fp32 kernel for everything.
LINEAR_F16X3_DEFAULT = True       # what `hip_linear_split: null` means (the Trainer applies the config value on every build)
LINEAR_F16X3 = LINEAR_F16X3_DEFAULT
SLICE_WIDTHS = (8, 16, 32)    # one feature slice of a 64-wide table: 64 / P columns per rank of the feature-sliced layout (csrc/spmm_narrow.hip)
SPMM_CHUNK = 512
LONG_ROW_DEFAULT = None    # by graph size, see default_long_row_threshold


def default_long_row_threshold(n_cols):
    """Rows with more nonzeros than this go to the chunk blocks (16 groups share the row) instead of one 16-lane group.
    Measured (tools/spmm_sweep.py, profiles/r02_spmm_plan_sweep.log): cache-resident graphs are bound by their longest
    serial gather chains -- 16 beats 64 by 30-38 % at Baby / Sports / Clothing size; HBM-sized graphs prefer 32 (-4 % at
    C5; 16 costs 12 % there: too many 256-thread workgroups for 20-nonzero rows).  A function of the COLUMN count only, so
    a graph and its row blocks (same columns) get the same plan: row shards stay bit-identical to the whole graph."""
    import os
    forced = os.environ.get("MMREC_LONG_ROW_THRESHOLD")      # measurement aid (tools/, A/B runs of bench.py)
    if forced:
        return int(forced)
    return 16 if n_cols <= (1 << 18) else 32

TOPK_MAX = 128            # MMREC_TOPK_MAX: row widths that are a multiple of 32 with <= 2,097,152 candidates; 64 for every other shape (the library says so: MMREC_ERR_UNSUPPORTED)
BPR_LOGSIG, BPR_GAMMA = 0, 1

# `hip_deterministic` (config key; Trainer sets it): the backward scatters of the fused loss kernels (BPR, cosine, InfoNCE,
# gather-norm) add with hardware fp32 atomics, so a batch with duplicated ids is order-dependent in the last ulp and a
# training run is not bitwise repeatable, although the reference's CPU path is.  In deterministic mode the same
# kernels run on the batch's GATHERED rows with identity ids (every output row written once) and the rows are scattered into
# the tables by mmrec_scatter_add_rows_sorted_f32: duplicates summed in position order by one owner, no atomics.
DETERMINISTIC_DEFAULT = False     # what `hip_deterministic: null` means (the Trainer passes the config value on every build)
DETERMINISTIC = DETERMINISTIC_DEFAULT
_TORCH_DET_BY_US = False


def set_deterministic(flag=True):
    """Process-wide switch.  Also puts torch's own scatter ops around the kernels (index_add & co.) into their
    deterministic mode -- and takes them out of it again only if this function put them there."""
    global DETERMINISTIC, _TORCH_DET_BY_US
    DETERMINISTIC = bool(flag)
    if DETERMINISTIC and not torch.are_deterministic_algorithms_enabled():
        torch.use_deterministic_algorithms(True, warn_only=True)
        _TORCH_DET_BY_US = True
    elif not DETERMINISTIC and _TORCH_DET_BY_US:
        torch.use_deterministic_algorithms(False)
        _TORCH_DET_BY_US = False


def scatter_add_rows(ids, rows, out):
    """out[ids[b]] += rows[b], duplicates summed in position order (deterministic; ids < 0 skipped)"""
    lib = _lib.load()
    _chk(ids, torch.int64, "ids", 1)
    rows, out = _chk(rows.contiguous(), torch.float32, "rows", 2), _chk(out, torch.float32, "out", 2)
    order = torch.sort(ids, stable=True)[1]
    _lib.check(lib.mmrec_scatter_add_rows_sorted_f32(_p(order), _p(ids), _p(rows), ids.numel(), rows.shape[1], _p(out),
                                                     _stream()), "scatter_add_rows_sorted")
    return out


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(t, dtype, name, dim=None):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.MMRecHipError("%s must be a device tensor (the hot path has no CPU fallback)" % name)
    if t.dtype != dtype:
        raise _lib.MMRecHipError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise _lib.MMRecHipError("%s must be contiguous" % name)
    if dim is not None and t.dim() != dim:
        raise _lib.MMRecHipError("%s must be %d-d" % (name, dim))
    return t


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


# ------------------------------------------------------------------------------------------------
# CSR graph container (+ long-row plan)
# ------------------------------------------------------------------------------------------------
class CsrGraph:
    """Device CSR (int32 rowptr/colidx, fp32 vals) + the long-row plan the SpMM kernel wants.

    `symmetric=True` (structure and values; all D^-1/2 A D^-1/2 graphs) lets the
    backward reuse the same CSR; otherwise `transpose()` builds A^T once (graphs are frozen or
    rebuilt once per epoch)."""

    def __init__(self, rowptr, colidx, vals, n_rows, n_cols, symmetric=False,
                 long_row_threshold=LONG_ROW_DEFAULT, rowptr_host=None):
        self.rowptr = _chk(rowptr, torch.int32, "rowptr", 1)
        self.colidx = _chk(colidx, torch.int32, "colidx", 1)
        self.vals = _chk(vals, torch.float32, "vals", 1)
        self.n_rows, self.n_cols = int(n_rows), int(n_cols)
        self.nnz = int(colidx.numel())
        self.symmetric = bool(symmetric)
        self.long_row_threshold = int(default_long_row_threshold(self.n_cols) if long_row_threshold is None
                                      else long_row_threshold)
        self._t = self if symmetric else None
        self._plan(rowptr_host)

    # -- plan: rows longer than the threshold are cut into fixed-size chunks (host side, C helper)
    def _plan(self, rowptr_host):
        lib = _lib.load()
        rp = rowptr_host if rowptr_host is not None else self.rowptr.cpu().numpy()
        rp = np.ascontiguousarray(rp, dtype=np.int32)
        self.rowptr_host = rp
        n_long, n_chunks = ctypes.c_int32(0), ctypes.c_int32(0)
        _lib.check(lib.mmrec_spmm_plan_count(rp.ctypes.data_as(ctypes.c_void_p), self.n_rows,
                                             self.long_row_threshold, ctypes.byref(n_long),
                                             ctypes.byref(n_chunks)), "spmm_plan_count")
        self.n_long, self.n_chunks = n_long.value, n_chunks.value
        dev = self.rowptr.device
        if self.n_long > 0:
            lr = np.empty(self.n_long, dtype=np.int32)
            cp = np.empty(self.n_long + 1, dtype=np.int32)
            _lib.check(lib.mmrec_spmm_plan_fill(rp.ctypes.data_as(ctypes.c_void_p), self.n_rows,
                                                self.long_row_threshold,
                                                lr.ctypes.data_as(ctypes.c_void_p),
                                                cp.ctypes.data_as(ctypes.c_void_p)), "spmm_plan_fill")
            self.long_rows = torch.from_numpy(lr).to(dev)
            self.long_chunk_ptr = torch.from_numpy(cp).to(dev)
            self.max_row_chunks = int(np.diff(cp).max())          # chunks of the longest row (mmrec_spmm_rows_any_f32)
            # last-arriver counters of the multi-chunk rows (small graphs finish such a row inside the launch): zero now,
            # left at zero by every launch
            self.long_tickets = torch.zeros(self.n_long, dtype=torch.int32, device=dev)
        else:
            self.long_rows = self.long_chunk_ptr = self.long_tickets = None
            self.max_row_chunks = 1
        self._partials = {}

    def checked(self, rc, what):
        """_lib.check for a launch that uses the last-arriver tickets: they are left at zero by every COMPLETED launch; after a
        failed one they are re-zeroed here, or every later SpMM on this graph would finish its long rows on a stale count."""
        try:
            _lib.check(rc, what)
        except _lib.MMRecHipError:
            if self.long_tickets is not None:
                try:
                    self.long_tickets.zero_()
                except RuntimeError:
                    pass                         # (a faulted device: nothing more will run on it anyway)
            raise

    def partials_for(self, d):
        """long-row workspace (n_chunks x d fp32), allocated once per embedding width"""
        if self.n_long == 0:
            return None
        if d not in self._partials:
            self._partials[d] = torch.empty(self.n_chunks * d, dtype=torch.float32, device=self.rowptr.device)
        return self._partials[d]

    @classmethod
    def from_coo_host(cls, idx, val, n_rows, n_cols, device, symmetric=False, **kw):
        """Stable COO->CSR on the host (numpy): entries of a row keep their COO order."""
        idx = np.asarray(idx, dtype=np.int64)
        val = np.asarray(val, dtype=np.float32)
        order = np.argsort(idx[0], kind="stable")
        rowptr = np.zeros(n_rows + 1, dtype=np.int64)
        np.cumsum(np.bincount(idx[0], minlength=n_rows), out=rowptr[1:])
        rp = rowptr.astype(np.int32)
        return cls(torch.from_numpy(rp).to(device),
                   torch.from_numpy(idx[1][order].astype(np.int32)).to(device),
                   torch.from_numpy(val[order]).to(device), n_rows, n_cols, symmetric=symmetric,
                   rowptr_host=rp, **kw)

    @classmethod
    def from_coo_device(cls, rows, cols, vals, n_rows, n_cols, symmetric=False, **kw):
        """Stable COO->CSR on the device (rocPRIM radix sort by row key inside the library)."""
        lib = _lib.load()
        _chk(rows, torch.int32, "rows", 1), _chk(cols, torch.int32, "cols", 1)
        _chk(vals, torch.float32, "vals", 1)
        nnz, dev = rows.numel(), rows.device
        rowptr = torch.empty(n_rows + 1, dtype=torch.int32, device=dev)
        colidx = torch.empty(nnz, dtype=torch.int32, device=dev)
        vout = torch.empty(nnz, dtype=torch.float32, device=dev)
        ws = _ws(lib.mmrec_coo_to_csr_workspace_bytes(nnz, n_rows), dev)
        _lib.check(lib.mmrec_coo_to_csr(_p(rows), _p(cols), _p(vals), nnz, n_rows, _p(rowptr),
                                        _p(colidx), _p(vout), _p(ws), _stream()), "coo_to_csr")
        return cls(rowptr, colidx, vout, n_rows, n_cols, symmetric=symmetric, **kw)

    def to_coo_host(self):
        rp = self.rowptr_host.astype(np.int64)
        rows = np.repeat(np.arange(self.n_rows, dtype=np.int64), np.diff(rp))
        return np.stack([rows, self.colidx.cpu().numpy().astype(np.int64)]), self.vals.cpu().numpy()

    def transpose(self):
        if self._t is None:
            idx, val = self.to_coo_host()
            self._t = CsrGraph.from_coo_host(idx[::-1], val, self.n_cols, self.n_rows,
                                             self.rowptr.device,
                                             long_row_threshold=self.long_row_threshold)
            self._t._t = self
        return self._t

    def row_block(self, r0, r1):
        """Rows [r0, r1) as their own CSR (column ids stay global): one rank's shard of a row-sharded
        graph.  Per-row data and order are untouched, so results equal the unsharded rows bit for bit."""
        rp = self.rowptr_host.astype(np.int64)
        s, e = int(rp[r0]), int(rp[r1])
        rph = (rp[r0:r1 + 1] - s).astype(np.int32)
        dev = self.rowptr.device
        return CsrGraph(torch.from_numpy(rph).to(dev), self.colidx[s:e].contiguous(),
                        self.vals[s:e].contiguous(), r1 - r0, self.n_cols,
                        long_row_threshold=self.long_row_threshold, rowptr_host=rph)


def _mode_labels(rows, lab_c, n_rows, n_labels):
    """for every row the most frequent label among its neighbours (ties: the smallest label); -1 for rows without any"""
    key = rows.astype(np.int64) * n_labels + lab_c
    key.sort()
    start = np.flatnonzero(np.concatenate([[True], key[1:] != key[:-1]]))
    cnt = np.diff(np.concatenate([start, [key.shape[0]]]))
    k = key[start]
    r, lab = k // n_labels, k % n_labels
    o = np.lexsort((lab, -cnt, r))               # per row: highest count first, then the smallest label
    r_s = r[o]
    first = np.concatenate([[True], r_s[1:] != r_s[:-1]])
    out = np.full(n_rows, -1, dtype=np.int64)
    out[r_s[first]] = lab[o][first]
    return out


def _mode_labels_device(rows, lab_c, n_rows, n_labels):
    """_mode_labels on the device (torch sorts of int64 keys: integer exact, the same answer): two sorts of nnz keys per sweep,
    milliseconds where the numpy form takes ~1 s per sweep at 20M nonzeros (7.4 s of a config-5 model build)."""
    key, _ = torch.sort(rows * n_labels + lab_c)
    uk, cnt = torch.unique_consecutive(key, return_counts=True)
    r, lab = torch.div(uk, n_labels, rounding_mode='floor'), uk % n_labels
    cmax = int(cnt.max().item()) + 1
    # per row: highest count first, then the smallest label -- one more sort of (row, cmax - count, label) packed in 63 bits
    if n_rows * cmax * n_labels >= 2 ** 62:
        raise OverflowError("label-propagation key does not fit 63 bits")
    k2, _ = torch.sort(r * (cmax * n_labels) + (cmax - cnt) * n_labels + lab)
    r2 = torch.div(k2, cmax * n_labels, rounding_mode='floor')
    first = torch.ones_like(r2, dtype=torch.bool)
    first[1:] = r2[1:] != r2[:-1]
    out = torch.full((n_rows,), -1, dtype=torch.int64, device=rows.device)
    out[r2[first]] = (k2 % n_labels)[first]
    return out


def locality_order(rowptr_host, colidx_host, n, how, n_left=None, iters=4, device=None):
    """A relabelling of the n nodes of a SQUARE graph for gather locality (round-3 review item 5): new id = perm[old id].
      'degree'     ids by descending degree (hot rows share cache lines and pages);
      'rcm'        reverse Cuthill-McKee on the structure (scipy.sparse.csgraph): good for mesh-like graphs, not for
                   interaction graphs (a few random edges make every BFS level the whole graph);
      'community'  `iters` rounds of label propagation (a node takes its neighbours' most frequent label), nodes then
                   ordered by label: the members of a community get adjacent ids, so the rows a workgroup gathers are the
                   rows its neighbours in the grid gather.  Bipartite graphs (`n_left`: ids below it are one side) update one
                   side from the other in turn.  A graph without communities just gets some permutation.
    Integer, deterministic; `device` (a CUDA device): the label-propagation sweeps run there (same permutation)."""
    rp = np.asarray(rowptr_host, dtype=np.int64)
    deg = np.diff(rp)
    ci = np.asarray(colidx_host, dtype=np.int64)
    if how == "degree":
        order = np.argsort(-deg, kind="stable")               # order[new] = old
    elif how == "rcm":
        import scipy.sparse as sp
        from scipy.sparse.csgraph import reverse_cuthill_mckee
        a = sp.csr_matrix((np.ones(ci.shape[0], dtype=np.int8), ci.astype(np.int32), rp.astype(np.int32)), shape=(n, n))
        order = np.asarray(reverse_cuthill_mckee(a, symmetric_mode=True), dtype=np.int64)
    elif how == "community" and device is not None and torch.device(device).type == "cuda":
        dev = torch.device(device)
        rows = torch.repeat_interleave(torch.arange(n, dtype=torch.int64, device=dev), torch.from_numpy(deg).to(dev))
        cols = torch.from_numpy(ci).to(dev)
        lab = torch.arange(n, dtype=torch.int64, device=dev)
        sides = [None] if not n_left else [rows < n_left, rows >= n_left]
        for _ in range(int(iters)):
            for side in sides:                                 # (bipartite: left from right, then right from the new left)
                r_s, c_s = (rows, cols) if side is None else (rows[side], cols[side])
                new = _mode_labels_device(r_s, lab[c_s], n, n)
                lab = torch.where(new >= 0, new, lab)
        order = torch.sort(lab, stable=True)[1].cpu().numpy()      # = np.lexsort((arange(n), lab))
    elif how == "community":
        rows = np.repeat(np.arange(n, dtype=np.int64), deg)
        lab = np.arange(n, dtype=np.int64)
        sides = [slice(None)] if not n_left else [rows < n_left, rows >= n_left]
        for _ in range(int(iters)):
            for side in sides:                                 # (bipartite: left from right, then right from the new left)
                new = _mode_labels(rows[side], lab[ci[side]], n, n)
                lab = np.where(new >= 0, new, lab)
        order = np.lexsort((np.arange(n), lab))
    else:
        raise ValueError("reorder must be 'degree', 'rcm' or 'community', got %r" % (how,))
    perm = np.empty(n, dtype=np.int64)
    perm[order] = np.arange(n, dtype=np.int64)
    return perm


class PermutedGraph:
    """A square CsrGraph with its node ids RELABELLED at build time (`reorder`: 'degree' | 'rcm' | 'community'; locality_order) and the
    permutation kept next to it.  Rows keep their nonzeros in their original order (stable relabelling), so every row's sum
    is the unpermuted graph's bit for bit: `lightgcn_mean(pg, E0, L)` / `spmm(pg, X)` permute the input once, run all
    layers in the relabelled space and permute the result back -- equal to the plain graph's results BITWISE."""

    def __init__(self, g: CsrGraph, reorder, n_left=None):
        if g.n_rows != g.n_cols:
            raise _lib.MMRecHipError("PermutedGraph needs a square graph")
        dev, n = g.rowptr.device, g.n_rows
        idx, val = g.to_coo_host()
        perm = locality_order(g.rowptr_host, idx[1], n, reorder, n_left=n_left)
        self.base, self.reorder = g, reorder
        self.perm = torch.from_numpy(perm).to(dev)                     # new id of old node
        inv = np.empty(n, dtype=np.int64)
        inv[perm] = np.arange(n, dtype=np.int64)
        self.inv = torch.from_numpy(inv).to(dev)                       # old node of new id
        self.graph = CsrGraph.from_coo_host(np.stack([perm[idx[0]], perm[idx[1]]]), val, n, n, dev, symmetric=g.symmetric,
                                            long_row_threshold=g.long_row_threshold)     # stable: in-row order kept
        self.n_rows = self.n_cols = n
        self.nnz = g.nnz

    def to_new(self, X):
        return X.index_select(0, self.inv)

    def to_old(self, Xp):
        return Xp.index_select(0, self.perm)


def spmm_raw(g: CsrGraph, X, Y=None, Z=None, acc_in=None, acc_out=None, alpha=1.0, beta=1.0,
             acc_scale=1.0):
    """Y = alpha*A@X (+ beta*Z);  acc_out = acc_scale*(acc_in + Y).  No autograd."""
    lib = _lib.load()
    _chk(X, torch.float32, "X", 2)
    d = X.shape[1]
    if (d not in SLICE_WIDTHS and (d % EMB_DIM or d > 6 * EMB_DIM)) or X.shape[0] < g.n_cols:
        raise _lib.MMRecHipError("X must be [>=%d, 64*k <= 384 (or a feature slice of 8 / 16 / 32 columns)], got %s" %
                                 (g.n_cols, tuple(X.shape)))
    for t, nm in ((Y, "Y"), (Z, "Z"), (acc_in, "acc_in"), (acc_out, "acc_out")):
        if t is not None:
            _chk(t, torch.float32, nm, 2)
            if t.shape[0] < g.n_rows or t.shape[1] != d:
                raise _lib.MMRecHipError("%s must be [>=%d, %d]" % (nm, g.n_rows, d))
    g.checked(lib.mmrec_spmm_csr_f32(_p(g.rowptr), _p(g.colidx), _p(g.vals), _p(X), _p(Y), _p(Z),
                                     _p(acc_in), _p(acc_out), g.n_rows, d, float(alpha),
                                     float(beta), float(acc_scale), g.long_row_threshold,
                                     _p(g.long_rows), _p(g.long_chunk_ptr), g.n_long, g.n_chunks,
                                     _p(g.partials_for(d)), _p(g.long_tickets), _stream()), "spmm_csr_f32")
    return Y if Y is not None else acc_out


ROWS_MAX_CHUNKS = 480      # mmrec_spmm_rows_any_f32 keeps a listed row's chunk partials in LDS (256 B each)


def rows_servable(g: CsrGraph, d):
    """can the row-list kernels reproduce the full launch's bits on this graph?  Every width they have when no row spans
    several chunks (mmrec_spmm_rows_f32); d = 64 also with such rows, up to ROWS_MAX_CHUNKS chunks (mmrec_spmm_rows_any_f32)"""
    if d not in SLICE_WIDTHS + (EMB_DIM,):
        return False
    return g.n_chunks == g.n_long or (d == EMB_DIM and g.max_row_chunks <= ROWS_MAX_CHUNKS)


def spmm_rows_raw(g: CsrGraph, X, rows, Z=None, z_compact=False):
    """(A @ X)[rows] (+ Z[rows], or + Z for a compact Z [len(rows), d]) as a compact [len(rows), d] tensor -- the bits of
    spmm_raw(g, X, Z=Z)[rows], computed for the listed rows only (a training step that reads a propagated table at its batch
    rows).  No autograd."""
    lib = _lib.load()
    _chk(X, torch.float32, "X", 2), _chk(rows, torch.int64, "rows", 1)
    d = X.shape[1]
    if not rows_servable(g, d) or X.shape[0] < g.n_cols:
        raise _lib.MMRecHipError("spmm_rows: graph with multi-chunk rows, or X not [>=%d, 8 / 16 / 32 / 64]" % g.n_cols)
    if Z is not None and (_chk(Z, torch.float32, "Z", 2).shape[0] < (rows.numel() if z_compact else g.n_rows) or Z.shape[1] != d):
        raise _lib.MMRecHipError("Z must be [>=%d, %d]" % (rows.numel() if z_compact else g.n_rows, d))
    Y = torch.empty(rows.numel(), d, dtype=torch.float32, device=X.device)
    long_t = g.long_row_threshold if g.n_long > 0 else 2 ** 31 - 1
    if g.n_chunks == g.n_long:
        _lib.check(lib.mmrec_spmm_rows_f32(_p(g.rowptr), _p(g.colidx), _p(g.vals), _p(X), _p(Z), 1 if z_compact else 0, _p(rows),
                                           rows.numel(), d, long_t, _p(Y), _stream()), "spmm_rows_f32")
    else:       # rows of several chunks among the graph's: a workgroup per such listed row (ABI 12)
        _lib.check(lib.mmrec_spmm_rows_any_f32(_p(g.rowptr), _p(g.colidx), _p(g.vals), _p(X), _p(Z), 1 if z_compact else 0, _p(rows),
                                               rows.numel(), d, long_t, g.max_row_chunks, _p(Y), _stream()), "spmm_rows_any_f32")
    return Y


def spmm_push_rows_raw(g: CsrGraph, G, rows, dX=None, dZ=None, scale=1.0):
    """dX[c] += A[r, c] * scale * G[i] over the nonzeros of the listed rows r = rows[i]; dZ[r] += scale * G[i] (fp32 atomics,
    in place; dZ may be dX).  Any graph (one workgroup per listed row)."""
    lib = _lib.load()
    _chk(G, torch.float32, "G", 2), _chk(rows, torch.int64, "rows", 1)
    d = G.shape[1]
    if G.shape[0] != rows.numel() or d not in SLICE_WIDTHS + (EMB_DIM,):
        raise _lib.MMRecHipError("G must be [len(rows), 8 / 16 / 32 / 64]")
    for t, nm, n in ((dX, "dX", g.n_cols), (dZ, "dZ", g.n_rows)):
        if t is not None and (_chk(t, torch.float32, nm, 2).shape[0] < n or t.shape[1] != d):
            raise _lib.MMRecHipError("%s must be [>=%d, %d]" % (nm, n, d))
    _lib.check(lib.mmrec_spmm_push_rows_f32(_p(g.rowptr), _p(g.colidx), _p(g.vals), _p(G), float(scale), _p(rows), rows.numel(),
                                            d, _p(dX), _p(dZ), _stream()), "spmm_push_rows_f32")


class _SpmmRows(torch.autograd.Function):
    """(A @ X)[rows] + Z_rows for a batch's rows (spmm_rows_raw); backward: dX by the transpose-free push through the listed
    rows into a zero-filled table, dZ_rows = the incoming gradient"""

    @staticmethod
    def forward(ctx, X, Z_rows, g, rows):
        ctx.g, ctx.x_shape, ctx.has_z = g, tuple(X.shape), Z_rows is not None
        ctx.save_for_backward(rows)
        return spmm_rows_raw(g, X.contiguous(), rows, None if Z_rows is None else Z_rows.contiguous(), z_compact=True)

    @staticmethod
    def backward(ctx, dY):
        rows, = ctx.saved_tensors
        dY = dY.contiguous()
        dX = None
        if ctx.needs_input_grad[0]:
            dX = torch.zeros(ctx.x_shape, dtype=dY.dtype, device=dY.device)
            spmm_push_rows_raw(ctx.g, dY, rows, dX=dX)
        return dX, (dY if ctx.has_z and ctx.needs_input_grad[1] else None), None, None


def spmm_rows(g: CsrGraph, X, rows, Z_rows=None):
    """(g @ X)[rows] (+ Z_rows, compact) -- differentiable in X and Z_rows -- for a training step that reads a propagated
    table at its batch rows only (FREEDOM's item-item layer, freedom.py:173-177 read at :197-199): the listed rows carry the
    full launch's bits, the backward pushes through them with fp32 atomics.  `hip_deterministic` runs, and graphs with rows
    spanning several chunks, take the full launch."""
    if DETERMINISTIC or isinstance(g, PermutedGraph) or not rows_servable(g, X.shape[1]):
        out = spmm(g, X).index_select(0, rows)
        return out if Z_rows is None else out + Z_rows
    return _SpmmRows.apply(X, Z_rows, g, rows)


class _SpMM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, Z, g):
        ctx.g, ctx.has_z, ctx.x_rows = g, Z is not None, X.shape[0]
        X = X.contiguous()
        Y = torch.empty(g.n_rows, X.shape[1], dtype=torch.float32, device=X.device)
        spmm_raw(g, X, Y=Y, Z=None if Z is None else Z.contiguous(), beta=1.0)
        return Y

    @staticmethod
    def backward(ctx, dY):
        dY = dY.contiguous()
        gt = ctx.g.transpose()
        dX = None
        if ctx.needs_input_grad[0]:
            # X may carry more rows than the graph has columns (spmm_raw allows it): they receive no gradient
            alloc = torch.empty if ctx.x_rows == gt.n_rows else torch.zeros
            dX = alloc(ctx.x_rows, dY.shape[1], dtype=torch.float32, device=dY.device)
            spmm_raw(gt, dY, Y=dX)
        return dX, (dY if ctx.has_z and ctx.needs_input_grad[1] else None), None


def spmm(g: CsrGraph, X, Z=None):
    """A @ X (+ Z), differentiable in X and Z.  replaces torch.sparse.mm (freedom.py:167,172) and PyG's
    mean-aggregating propagate (mmgcn.py:205-213).  Row widths that are not a multiple of 64 are
    zero-padded for the kernel (aggregation is column-wise independent) and sliced back."""
    if isinstance(g, PermutedGraph):
        out = spmm(g.graph, g.to_new(X), None if Z is None else g.to_new(Z))
        return g.to_old(out)
    d = X.shape[1]
    if d % EMB_DIM and d not in SLICE_WIDTHS:
        pad = EMB_DIM - d % EMB_DIM
        Xp = torch.nn.functional.pad(X, (0, pad))
        Zp = None if Z is None else torch.nn.functional.pad(Z, (0, pad))
        return _SpMM.apply(Xp, Zp, g)[:, :d]
    return _SpMM.apply(X, Z, g)


class _LightGCNMean(torch.autograd.Function):
    """mean_l(A^l E0), l = 0..L, with the layer sum fused into the SpMM epilogue
    (lightgcn.py:115-128, freedom.py:169-176, bm3.py:87-93).  Backward is the Horner form
    dE0 = s*(g + A^T(g + A^T(...)))  with s = 1/(L+1), again L launches of the same kernel."""

    @staticmethod
    def forward(ctx, E0, g, n_layers):
        ctx.g, ctx.L = g, int(n_layers)
        E0 = E0.contiguous()
        L = ctx.L
        if L == 0:
            return E0.clone()
        acc = torch.empty_like(E0)
        bufs = [torch.empty_like(E0) if L > 1 else None, torch.empty_like(E0) if L > 2 else None]
        cur = E0
        for layer in range(1, L + 1):
            last = layer == L
            Y = None if last else bufs[(layer - 1) % 2]
            spmm_raw(g, cur, Y=Y, acc_in=E0 if layer == 1 else acc, acc_out=acc,
                     acc_scale=1.0 / (L + 1) if last else 1.0)
            cur = Y
        return acc

    @staticmethod
    def backward(ctx, dOut):
        L, gt = ctx.L, ctx.g.transpose()
        dOut = dOut.contiguous()
        if L == 0:
            return dOut, None, None
        s = 1.0 / (L + 1)
        bufs = [torch.empty_like(dOut), torch.empty_like(dOut) if L > 1 else None]
        t = dOut
        for j in range(L):  # t <- s*dOut + A^T t   (first step also scales the inner term)
            out = bufs[j % 2]
            spmm_raw(gt, t, Y=out, Z=dOut, alpha=s if j == 0 else 1.0, beta=s)
            t = out
        return t, None, None


def lightgcn_mean(g, E0, n_layers):
    if isinstance(g, PermutedGraph):      # relabelled graph: one permutation in, all layers in the new id space, one out
        return g.to_old(_LightGCNMean.apply(g.to_new(E0), g.graph, n_layers))
    return _LightGCNMean.apply(E0, g, n_layers)


def row_blocks_of_one_buffer(ts):
    """True if the tensors are consecutive contiguous row blocks of ONE allocation, i.e. their cat already exists"""
    if any(t is None or not t.is_contiguous() or t.dtype != ts[0].dtype for t in ts):
        return False
    base = ts[0].untyped_storage().data_ptr()
    end = ts[0].data_ptr()
    for t in ts:
        if t.untyped_storage().data_ptr() != base or t.data_ptr() != end:
            return False
        end += t.numel() * t.element_size()
    return end <= base + ts[0].untyped_storage().nbytes()


class _LightGCNMeanParts(torch.autograd.Function):
    """_LightGCNMean on the row-wise concatenation of several tables (user table, item table), returning the mean split
    back into the same row blocks.  The same launches; what goes away is autograd's bookkeeping around them at 1.5M rows:
    the zero-filled [N, 64] gradients of the two output slices and their sum, and the split of the cat's gradient."""

    @staticmethod
    def forward(ctx, g, n_layers, *parts):
        ctx.sizes = [p.shape[0] for p in parts]
        if row_blocks_of_one_buffer(parts):        # models/_base.py AdjacentTablesMixin: the cat already exists
            E0 = parts[0].detach().as_strided((sum(ctx.sizes), parts[0].shape[1]), (parts[0].shape[1], 1))
        else:
            E0 = torch.cat([p.detach() for p in parts], dim=0)
        ctx.g, ctx.L = g, int(n_layers)
        out = _LightGCNMean.forward(ctx, E0, g, n_layers)
        return tuple(out.split(ctx.sizes))

    @staticmethod
    def backward(ctx, *grads):
        like = next(x for x in grads if x is not None)
        n, d = sum(ctx.sizes), like.shape[1]
        if row_blocks_of_one_buffer(grads):       # e.g. bpr_losses_shared_users(joint_grad=True): already laid out as cat(grads)
            dOut = grads[0].as_strided((n, d), (d, 1))
        else:
            dOut = torch.empty((n, d), dtype=like.dtype, device=like.device)
            for dst, src in zip(dOut.split(ctx.sizes), grads):
                dst.zero_() if src is None else dst.copy_(src)
        t = _LightGCNMean.backward(ctx, dOut)[0]
        return (None, None) + tuple(t.split(ctx.sizes))


def lightgcn_mean_parts(g: CsrGraph, parts, n_layers):
    """mean_l(A^l cat(parts)) as a tuple of row blocks, one per input table (freedom.py:165-178: cat, propagate, split)"""
    return _LightGCNMeanParts.apply(g, n_layers, *parts)


ROWS_LAST_LAYER = True      # lightgcn_mean_parts_rows: last layer at the listed rows only (False: full launches + gather, A/B)


class _LightGCNMeanPartsRows(torch.autograd.Function):
    """lightgcn_mean_parts consumed at LISTED rows only (the batch's users and items): the forward is the full propagation
    (every layer feeds the next) followed by a gather of the listed rows; the BACKWARD starts from the compact gradient of those
    rows, so its first step -- A^T applied to a gradient that is zero outside <= 3B rows, a launch over all rows in
    _LightGCNMean.backward -- is a push through the listed rows into a zero-filled table (one workgroup per row: popular
    items' rows have thousands of nonzeros), and the dense [N, d] gradient of the mean never exists.  The remaining L - 1
    steps are the usual full launches."""

    @staticmethod
    def forward(ctx, g, n_layers, rows, *parts):
        ctx.sizes = [p.shape[0] for p in parts]
        if row_blocks_of_one_buffer(parts):
            E0 = parts[0].detach().as_strided((sum(ctx.sizes), parts[0].shape[1]), (parts[0].shape[1], 1))
        else:
            E0 = torch.cat([p.detach() for p in parts], dim=0)
        ctx.save_for_backward(rows)
        L = int(n_layers)
        if L >= 1 and ROWS_LAST_LAYER and rows_servable(g, E0.shape[1]) and E0.shape[0] == g.n_rows == g.n_cols:
            # Round 6: the LAST layer is read at the listed rows only -- it feeds no further layer -- so it is computed there
            # (spmm_rows_raw: the full launch's bits row by row, long rows included), with the epilogue's
            # s * (running layer sum + y) done on the compact rows: a launch over all rows less per step (0.30 of a
            # config-5 step's 2.35 ms).
            ctx.g, ctx.L = g, L
            E0 = E0.contiguous()
            cur, acc = E0, E0
            if L > 1:
                acc = torch.empty_like(E0)
                bufs = [torch.empty_like(E0), torch.empty_like(E0) if L > 2 else None]
                for layer in range(1, L):
                    Y = bufs[(layer - 1) % 2]
                    spmm_raw(g, cur, Y=Y, acc_in=E0 if layer == 1 else acc, acc_out=acc, acc_scale=1.0)
                    cur = Y
            return spmm_rows_raw(g, cur, rows, Z=acc) * (1.0 / (L + 1))
        out = _LightGCNMean.forward(ctx, E0, g, n_layers)
        return out.index_select(0, rows)

    @staticmethod
    def backward(ctx, dRows):
        rows, = ctx.saved_tensors
        L, g = ctx.L, ctx.g
        dRows = dRows.contiguous()
        n, d = sum(ctx.sizes), dRows.shape[1]
        s = 1.0 / (L + 1)
        t = torch.zeros((n, d), dtype=dRows.dtype, device=dRows.device)
        if L == 0:
            spmm_push_rows_raw(g, dRows, rows, dZ=t)
            return (None, None, None) + tuple(t.split(ctx.sizes))
        # t1 = s * (A^T G + G) with G = the listed rows' gradient scattered: pushed, never materialised
        spmm_push_rows_raw(g, dRows, rows, dX=t, dZ=t, scale=s)
        gt = g.transpose()
        for _ in range(1, L):          # t <- A^T t + s G
            out = torch.empty_like(t)
            spmm_raw(gt, t, Y=out)
            spmm_push_rows_raw(g, dRows, rows, dZ=out, scale=s)
            t = out
        return (None, None, None) + tuple(t.split(ctx.sizes))


class _MeanPartsRowsThenItemRows(torch.autograd.Function):
    """FREEDOM's training-step forward at the batch rows as ONE autograd node: at = lightgcn_mean_parts_rows(g, (user table, item
    table), L, cat(users, nu + item_rows)), then ia = (mm @ item table)[item_rows] + at[b:] (freedom.py:165-178 read at
    :197-199).  As two nodes the item table receives two dense gradients -- the propagation's block and a zero-filled
    [n_items, d] buffer the item-item layer's backward pushes into -- that autograd then adds: at config 5 a 128 MB fill and a
    384 MB add per step (17 + 60 us of a 2.1 ms step).  Here the push goes straight into the propagation's gradient.
    Inputs: g, n_layers, rows, mm, item_rows, n_user_rows (b), then the two tables.  Forward bits: the two ops'."""

    @staticmethod
    def forward(ctx, g, n_layers, rows, mm, item_rows, b, user_table, item_table):
        import types
        inner = types.SimpleNamespace(saved=None)
        inner.save_for_backward = lambda *t: setattr(inner, 'saved', t)
        at = _LightGCNMeanPartsRows.forward(inner, g, n_layers, rows, user_table, item_table)
        ia = spmm_rows_raw(mm, item_table.detach().contiguous(), item_rows, at[b:].contiguous(), z_compact=True)
        ctx.inner = types.SimpleNamespace(sizes=inner.sizes, g=inner.g, L=inner.L)
        ctx.mm, ctx.b = mm, int(b)
        ctx.save_for_backward(rows, item_rows)
        return at[:b].contiguous(), ia

    @staticmethod
    def backward(ctx, d_user_rows, d_ia):
        rows, item_rows = ctx.saved_tensors
        inner = ctx.inner
        inner.saved_tensors = (rows,)
        d_ia = d_ia.contiguous()
        d_rows = torch.cat((d_user_rows.contiguous(), d_ia), dim=0)        # d at = (d user rows, d ia: Z passes its gradient on)
        grads = _LightGCNMeanPartsRows.backward(inner, d_rows)
        d_user_table, d_item_table = grads[3], grads[4]
        spmm_push_rows_raw(ctx.mm, d_ia, item_rows, dX=d_item_table)       # (atomics into the propagation's item block)
        return None, None, None, None, None, None, d_user_table, d_item_table


def lightgcn_mean_rows_then_item_rows(g: CsrGraph, user_table, item_table, n_layers, users, item_rows, mm: CsrGraph):
    """-> (propagated user rows [b, d], (mm @ item_table)[item_rows] + propagated item rows [len(item_rows), d]) for a training
    step that reads its tables at the batch rows only; one autograd node when every piece has its row-list kernel, the two
    ops otherwise (`hip_deterministic`, relabelled graphs, rows spanning too many chunks)."""
    b, nu, d = users.shape[0], user_table.shape[0], user_table.shape[1]
    rows = torch.cat((users, item_rows + nu))
    fast = (not DETERMINISTIC and not isinstance(g, PermutedGraph) and not isinstance(mm, PermutedGraph) and d == EMB_DIM
            and int(n_layers) >= 1 and ROWS_LAST_LAYER and rows_servable(g, d) and rows_servable(mm, d)
            and user_table.shape[0] + item_table.shape[0] == g.n_rows == g.n_cols and mm.n_rows == mm.n_cols == item_table.shape[0])
    if not fast:
        at = lightgcn_mean_parts_rows(g, (user_table, item_table), n_layers, rows)
        return at[:b], spmm_rows(mm, item_table, item_rows, Z_rows=at[b:])
    return _MeanPartsRowsThenItemRows.apply(g, n_layers, rows, mm, item_rows, b, user_table, item_table)


def lightgcn_mean_parts_rows(g: CsrGraph, parts, n_layers, rows):
    """lightgcn_mean_parts(g, parts, n_layers) read at `rows` (int64 ids in the concatenated id space) -> [len(rows), d];
    same forward bits, the backward starts from the compact gradient (see the class).  `hip_deterministic`: the dense path."""
    if DETERMINISTIC or isinstance(g, PermutedGraph) or parts[0].shape[1] not in SLICE_WIDTHS + (EMB_DIM,):
        return torch.cat(lightgcn_mean_parts(g, parts, n_layers), dim=0).index_select(0, rows)
    return _LightGCNMeanPartsRows.apply(g, n_layers, rows, *parts)


class _LayerGCNSum(torch.autograd.Function):
    """sum_l w_l * (A E_{l-1}),  w_l = cos(A E_{l-1}, E0) per row, E_l = w_l * A E_{l-1}
    (layergcn.py:125-138).  SpMM kernel + fused cos-scale/accumulate kernel per layer."""

    @staticmethod
    def forward(ctx, E0, g, n_layers):
        lib = _lib.load()
        E0 = E0.contiguous()
        L, n = int(n_layers), E0.shape[0]
        if E0.shape[1] != EMB_DIM or E0.shape[0] != g.n_rows or g.n_cols != g.n_rows:
            raise _lib.MMRecHipError("layergcn_sum needs E0 [n, %d] over a square graph of n rows" % EMB_DIM)
        acc = torch.zeros_like(E0) if L == 0 else torch.empty_like(E0)
        need_y = any(ctx.needs_input_grad)        # the unscaled products are only read by the backward
        ys, ws = [], []
        cur = E0
        for layer in range(L):                    # SpMM + cosine re-weighting + layer sum: ONE launch per layer
            y = torch.empty_like(E0) if need_y else None
            out, w = torch.empty_like(E0), torch.empty(n, dtype=torch.float32, device=E0.device)
            g.checked(lib.mmrec_spmm_csr_f32_layergcn(
                _p(g.rowptr), _p(g.colidx), _p(g.vals), _p(cur), _p(y), _p(E0), _p(out), _p(w),
                _p(acc) if layer > 0 else None, _p(acc), g.n_rows, EMB_DIM, g.long_row_threshold, _p(g.long_rows),
                _p(g.long_chunk_ptr), g.n_long, g.n_chunks, _p(g.partials_for(EMB_DIM)), _p(g.long_tickets), _stream()),
                "spmm_layergcn")
            ys.append(y), ws.append(w)
            cur = out
        ctx.g, ctx.L = g, L
        if need_y:
            ctx.save_for_backward(E0, *ys, *ws)
        return acc

    @staticmethod
    def backward(ctx, dSum):
        lib = _lib.load()
        L, gt = ctx.L, ctx.g.transpose()
        saved = ctx.saved_tensors
        E0, ys, ws = saved[0], saved[1:1 + L], saved[1 + L:]
        dSum = dSum.contiguous()
        n = E0.shape[0]
        dEgo = torch.zeros_like(E0)
        dOut = dSum  # gradient w.r.t. the last layer's output
        for layer in range(L - 1, -1, -1):
            dY = torch.empty_like(E0)
            _lib.check(lib.mmrec_cos_scale_bwd_f32(_p(dOut), _p(ys[layer]), _p(E0), _p(ws[layer]),
                                                   _p(dY), _p(dEgo), n, EMB_DIM, _stream()),
                       "cos_scale_bwd")
            nxt = torch.empty_like(E0)
            # layer > 0: grad of the previous layer's output = A^T dY + dSum ; layer 0: input is E0
            spmm_raw(gt, dY, Y=nxt, Z=dSum if layer > 0 else dEgo, beta=1.0)
            dOut = nxt
        return dOut, None, None


def layergcn_sum(g: CsrGraph, E0, n_layers):
    return _LayerGCNSum.apply(E0, g, n_layers)


class _LayerGCNSumParts(torch.autograd.Function):
    """_LayerGCNSum on the row-wise concatenation of several tables, result split back into the same row blocks (see
    _LightGCNMeanParts: the same launches without autograd's cat / slice bookkeeping around them)."""

    @staticmethod
    def forward(ctx, g, n_layers, *parts):
        ctx.sizes = [p.shape[0] for p in parts]
        if row_blocks_of_one_buffer(parts):
            E0 = parts[0].detach().as_strided((sum(ctx.sizes), parts[0].shape[1]), (parts[0].shape[1], 1))
        else:
            E0 = torch.cat([p.detach() for p in parts], dim=0)
        return tuple(_LayerGCNSum.forward(ctx, E0, g, n_layers).split(ctx.sizes))

    @staticmethod
    def backward(ctx, *grads):
        like = next(x for x in grads if x is not None)
        n, d = sum(ctx.sizes), like.shape[1]
        if row_blocks_of_one_buffer(grads):
            dSum = grads[0].as_strided((n, d), (d, 1))
        else:
            dSum = torch.empty((n, d), dtype=like.dtype, device=like.device)
            for dst, src in zip(dSum.split(ctx.sizes), grads):
                dst.zero_() if src is None else dst.copy_(src)
        t = _LayerGCNSum.backward(ctx, dSum)[0]
        return (None, None) + tuple(t.split(ctx.sizes))


def layergcn_sum_parts(g: CsrGraph, parts, n_layers):
    """sum_l w_l (A E_{l-1}) over cat(parts) as a tuple of row blocks, one per input table (layergcn.py:125-138)"""
    return _LayerGCNSumParts.apply(g, n_layers, *parts)


# ------------------------------------------------------------------------------------------------
# P4  sampled scoring
# ------------------------------------------------------------------------------------------------
def _bpr_bwd_deterministic(U, I, users, pos, neg, coef, g, scale, dU, dI):
    """the fused BPR backward without atomics on the tables: the kernel runs on the gathered rows U[users], I[pos], I[neg]
    with identity ids (each per-sample gradient row is written once), the rows are then scattered in position order"""
    lib = _lib.load()
    B, d = users.numel(), U.shape[1]
    ar = torch.arange(B, device=U.device)
    Ug = U.index_select(0, users)
    Ig = torch.cat((I.index_select(0, pos), I.index_select(0, neg)), 0)
    dUg = torch.zeros_like(Ug) if dU is not None else None
    dIg = torch.zeros_like(Ig) if dI is not None else None
    arn = (ar + B).contiguous()
    _lib.check(lib.mmrec_bpr_bwd_f32(_p(Ug), _p(Ig), _p(Ig), _p(ar), _p(ar), _p(arn), B, d, _p(coef), _p(g), scale, _p(dUg),
                                     _p(dIg), _p(dIg), _stream()), "bpr_bwd")
    if dU is not None:
        scatter_add_rows(users, dUg, dU)
    if dI is not None:
        scatter_add_rows(torch.cat((pos, neg)), dIg, dI)


class _BprLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, U, I, users, pos, neg, variant, scale):
        lib = _lib.load()
        U, I = _chk(U.contiguous(), torch.float32, "U", 2), _chk(I.contiguous(), torch.float32, "I", 2)
        for t, nm in ((users, "users"), (pos, "pos"), (neg, "neg")):
            _chk(t, torch.int64, nm, 1)
        B, dev = users.numel(), U.device
        loss = torch.empty((), dtype=torch.float32, device=dev)
        coef = torch.empty(max(B, 1), dtype=torch.float32, device=dev)
        ws = _ws(lib.mmrec_bpr_workspace_bytes(B), dev)
        if U.shape[1] != I.shape[1] or U.shape[1] % EMB_DIM:
            raise _lib.MMRecHipError("U and I need the same row width, a multiple of %d" % EMB_DIM)
        _lib.check(lib.mmrec_bpr_fwd_f32(_p(U), _p(I), _p(I), _p(users), _p(pos), _p(neg), B, U.shape[1],
                                         int(variant), float(scale), _p(loss), _p(coef), _p(ws),
                                         _stream()), "bpr_fwd")
        ctx.save_for_backward(U, I, users, pos, neg, coef)
        ctx.scale = float(scale)
        return loss

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        U, I, users, pos, neg, coef = ctx.saved_tensors
        g = g.contiguous().to(torch.float32)
        dU = torch.zeros_like(U) if ctx.needs_input_grad[0] else None
        dI = torch.zeros_like(I) if ctx.needs_input_grad[1] else None
        if DETERMINISTIC:
            _bpr_bwd_deterministic(U, I, users, pos, neg, coef, g, ctx.scale, dU, dI)
            return dU, dI, None, None, None, None, None
        _lib.check(lib.mmrec_bpr_bwd_f32(_p(U), _p(I), _p(I), _p(users), _p(pos), _p(neg),
                                         users.numel(), U.shape[1], _p(coef), _p(g), ctx.scale, _p(dU),
                                         _p(dI), _p(dI), _stream()), "bpr_bwd")
        return dU, dI, None, None, None, None, None


class _BprLossShared(torch.autograd.Function):
    """Several BPR terms over the SAME user rows U[users] (FREEDOM's id term and its two modality terms,
    freedom.py:197-211), one loss scalar per term.  The same forward / backward kernels as _BprLoss, but ONE dense
    gradient buffer for U that the backward launches accumulate into, instead of one zero-filled [n_users, d] buffer per
    term and their sums."""

    @staticmethod
    def forward(ctx, U, users, variant, scale, n_terms, joint, sum_over_ranks, *flat):
        lib = _lib.load()
        ctx.joint = bool(joint)
        U = _chk(U.contiguous(), torch.float32, "U", 2)
        _chk(users, torch.int64, "users", 1)
        B, dev = users.numel(), U.device
        tables, ids, coefs, losses = [], [], [], []
        ws = _ws(lib.mmrec_bpr_workspace_bytes(B), dev)
        d = U.shape[1]
        sliced = sum_over_ranks is not None
        if sliced:          # column slices: partial <u, p>, <u, n> of every term, ONE sum over the ranks, then the losses
            dots = torch.empty((n_terms, 2, max(B, 1)), dtype=torch.float32, device=dev)
        for t in range(n_terms):
            I, pos, neg = flat[3 * t], flat[3 * t + 1], flat[3 * t + 2]
            I = _chk(I.contiguous(), torch.float32, "I", 2)
            _chk(pos, torch.int64, "pos", 1), _chk(neg, torch.int64, "neg", 1)
            if d != I.shape[1] or (d % EMB_DIM and not (sliced and d in SLICE_WIDTHS)):
                raise _lib.MMRecHipError("U and I need the same row width, a multiple of %d%s" %
                                         (EMB_DIM, " or a slice of 8 / 16 / 32 columns" if sliced else ""))
            tables.append(I), ids.extend((pos, neg))
            if sliced:
                _lib.check(lib.mmrec_bpr_dots_f32(_p(U), _p(I), _p(I), _p(users), _p(pos), _p(neg), B, d, _p(dots[t]),
                                                  _stream()), "bpr_dots")
                continue
            loss = torch.empty((), dtype=torch.float32, device=dev)
            coef = torch.empty(max(B, 1), dtype=torch.float32, device=dev)
            _lib.check(lib.mmrec_bpr_fwd_f32(_p(U), _p(I), _p(I), _p(users), _p(pos), _p(neg), B, d,
                                             int(variant), float(scale), _p(loss), _p(coef), _p(ws), _stream()), "bpr_fwd")
            coefs.append(coef), losses.append(loss)
        if sliced:
            sum_over_ranks(dots)                                  # in place
            for t in range(n_terms):
                loss = torch.empty((), dtype=torch.float32, device=dev)
                coef = torch.empty(max(B, 1), dtype=torch.float32, device=dev)
                _lib.check(lib.mmrec_bpr_loss_from_dots_f32(_p(dots[t]), B, int(variant), float(scale), _p(loss), _p(coef),
                                                            _p(ws), _stream()), "bpr_loss_from_dots")
                coefs.append(coef), losses.append(loss)
        ctx.save_for_backward(U, users, *tables, *ids, *coefs)
        ctx.scale, ctx.n_terms = float(scale), n_terms
        return tuple(losses)

    @staticmethod
    def backward(ctx, *gs):
        lib = _lib.load()
        n = ctx.n_terms
        saved = ctx.saved_tensors
        U, users = saved[0], saved[1]
        tables, ids, coefs = saved[2:2 + n], saved[2 + n:2 + 3 * n], saved[2 + 3 * n:]
        dU = dI0 = None
        if ctx.joint and ctx.needs_input_grad[0] and ctx.needs_input_grad[7]:
            # the gradients of U and of the first item table as adjacent row blocks of one zero-filled buffer: when both
            # came out of lightgcn_mean_parts, its backward takes the buffer as the gradient of its output, copy-free
            both = torch.zeros((U.shape[0] + tables[0].shape[0], U.shape[1]), dtype=U.dtype, device=U.device)
            dU, dI0 = both[:U.shape[0]], both[U.shape[0]:]
        elif ctx.needs_input_grad[0]:
            dU = torch.zeros_like(U)
        out = []
        for t in range(n):
            I, pos, neg = tables[t], ids[2 * t], ids[2 * t + 1]
            need_i = ctx.needs_input_grad[7 + 3 * t]
            own = dI0 if t == 0 and dI0 is not None else None
            if gs[t] is None or (dU is None and not need_i):
                out.extend(((own if own is not None else torch.zeros_like(I)) if need_i else None, None, None))
                continue
            dI = (own if own is not None else torch.zeros_like(I)) if need_i else None
            g = gs[t].contiguous().to(torch.float32)
            if DETERMINISTIC:       # terms in order, each scattered without atomics: the shared dU gets them one after the other
                _bpr_bwd_deterministic(U, I, users, pos, neg, coefs[t], g, ctx.scale, dU, dI)
            else:
                _lib.check(lib.mmrec_bpr_bwd_f32(_p(U), _p(I), _p(I), _p(users), _p(pos), _p(neg), users.numel(), U.shape[1],
                                                 _p(coefs[t]), _p(g), ctx.scale, _p(dU), _p(dI), _p(dI), _stream()), "bpr_bwd")
            out.extend((dI, None, None))
        return (dU, None, None, None, None, None, None) + tuple(out)


class _BprWeightedTotal(torch.autograd.Function):
    """sum_t w_t bpr_loss(U, I_t, users, pos_t, neg_t): mmrec_bpr_multi_fwd_f32 / _bwd_f32 (ABI 14) -- every term's per-sample
    work in ONE launch (grid.y = term), one finish launch for the weighted total, one backward launch.  Inputs: U, users, variant,
    scale, weights (tuple), joint, then I_0, pos_0, neg_0, I_1, ...; `joint`: the gradients of U and of the first table are
    adjacent row blocks of one zero-filled buffer (what lightgcn_mean_parts' backward takes copy-free)."""

    @staticmethod
    def forward(ctx, U, users, variant, scale, weights, joint, *flat):
        lib = _lib.load()
        n = len(weights)
        U = _chk(U.contiguous(), torch.float32, "U", 2)
        _chk(users, torch.int64, "users", 1)
        B, d, dev = users.numel(), U.shape[1], U.device
        tables = [_chk(flat[3 * t].contiguous(), torch.float32, "I", 2) for t in range(n)]
        pos, neg = [flat[3 * t + 1] for t in range(n)], [flat[3 * t + 2] for t in range(n)]
        for t in range(n):
            _chk(pos[t], torch.int64, "pos", 1), _chk(neg[t], torch.int64, "neg", 1)
            if tables[t].shape[1] != d or d % EMB_DIM or pos[t].numel() != B or neg[t].numel() != B:
                raise _lib.MMRecHipError("U and every table need the same row width (a multiple of %d), every id list B entries" % EMB_DIM)
        ptrs = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
        ctx.arrays = (ptrs(tables), ptrs(pos), ptrs(neg), (ctypes.c_float * n)(*[float(w) for w in weights]))
        total = torch.empty((), dtype=torch.float32, device=dev)
        coef = torch.empty(n, max(B, 1), dtype=torch.float32, device=dev)
        ws = _ws(lib.mmrec_bpr_multi_workspace_bytes(n, B), dev)
        _lib.check(lib.mmrec_bpr_multi_fwd_f32(_p(U), _p(users), *ctx.arrays, n, B, d, int(variant), float(scale), _p(total), None,
                                               _p(coef), _p(ws), _stream()), "bpr_multi_fwd")
        ctx.save_for_backward(U, users, coef, *tables, *pos, *neg)
        ctx.n, ctx.scale, ctx.joint = n, float(scale), bool(joint)
        return total

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        n = ctx.n
        U, users, coef = ctx.saved_tensors[:3]
        tables = ctx.saved_tensors[3:3 + n]
        g = g.contiguous().to(torch.float32)
        need_u = ctx.needs_input_grad[0]
        need_i = [ctx.needs_input_grad[6 + 3 * t] for t in range(n)]
        dU = None
        grads = [None] * n
        if ctx.joint and need_u and need_i[0]:
            both = torch.zeros((U.shape[0] + tables[0].shape[0], U.shape[1]), dtype=U.dtype, device=U.device)
            dU, grads[0] = both[:U.shape[0]], both[U.shape[0]:]
        elif need_u:
            dU = torch.zeros_like(U)
        shared = {}
        for t in range(n):
            if need_i[t] and grads[t] is None:
                key = tables[t].data_ptr()
                if key not in shared:
                    shared[key] = torch.zeros_like(tables[t])
                grads[t] = shared[key]
        dI = (ctypes.c_void_p * n)(*[None if x is None else x.data_ptr() for x in grads])
        _lib.check(lib.mmrec_bpr_multi_bwd_f32(_p(U), _p(users), *ctx.arrays, n, users.numel(), U.shape[1], _p(coef), _p(g), ctx.scale,
                                               _p(dU), dI, _stream()), "bpr_multi_bwd")
        out, seen = [dU, None, None, None, None, None], set()
        for t in range(n):
            gi = grads[t]
            if gi is not None and id(gi) in seen:
                gi = None                                  # (a table named by two terms: its one buffer goes to the first)
            elif gi is not None:
                seen.add(id(gi))
            out += [gi, None, None]
        return tuple(out)


def bpr_weighted_total(U, users, terms, weights, variant=BPR_LOGSIG, reduction="mean", joint_grad=False):
    """sum_t weights[t] * bpr_loss(U, I_t, users, pos_t, neg_t) as ONE scalar (freedom.py:197-211: the id term + reg_weight times
    the modality terms) -- two launches forward, one backward (ABI 14); `hip_deterministic` or more than MMREC_BPR_MAX_TERMS terms:
    the per-term form."""
    terms = list(terms)
    if DETERMINISTIC or not 1 <= len(terms) <= 4:
        losses = bpr_losses_shared_users(U, users, terms, variant, reduction, joint_grad=joint_grad)
        total = 0.0
        for w, l in zip(weights, losses):
            total = total + w * l
        return total
    B = users.numel()
    scale = 1.0 / max(B, 1) if reduction == "mean" else 1.0
    flat = [x for term in terms for x in term]
    return _BprWeightedTotal.apply(U, users, variant, scale, tuple(float(w) for w in weights), joint_grad, *flat)


def bpr_losses_shared_users(U, users, terms, variant=BPR_LOGSIG, reduction="mean", joint_grad=False, sum_over_ranks=None):
    """[bpr_loss(U, I_t, users, pos_t, neg_t) for (I_t, pos_t, neg_t) in terms] with one shared gradient buffer for U;
    joint_grad: the gradient of the FIRST term's table is the row block right after U's in the same buffer.
    sum_over_ranks (feature-sliced layout): U and the tables are this rank's COLUMNS (8 / 16 / 32 of them, or whole rows);
    the callable sums the [terms, 2, B] partial dot products over the ranks in place (one all-reduce), the losses come out
    replicated and every rank's backward scatters into its own columns -- the same kernels, no collective."""
    B = users.numel()
    scale = 1.0 / max(B, 1) if reduction == "mean" else 1.0
    flat = [x for term in terms for x in term]
    return _BprLossShared.apply(U, users, variant, scale, len(terms), joint_grad, sum_over_ranks, *flat)


def bpr_loss(U, I, users, pos, neg, variant=BPR_LOGSIG, reduction="mean"):
    """Fused gather-dot-(log)sigmoid BPR loss on rows U[users], I[pos], I[neg].
    variant LOGSIG/mean = FREEDOM.bpr_loss (freedom.py:180-187), LOGSIG/sum = LayerGCN
    (layergcn.py:140-152), GAMMA/mean = BPRLoss (common/loss.py:33-35)."""
    B = users.numel()
    scale = 1.0 / max(B, 1) if reduction == "mean" else 1.0
    return _BprLoss.apply(U, I, users, pos, neg, variant, scale)


class _InfoNCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, E1, E2, ids, tau):
        lib = _lib.load()
        E1, E2 = _chk(E1.contiguous(), torch.float32, "E1", 2), _chk(E2.contiguous(), torch.float32, "E2", 2)
        _chk(ids, torch.int64, "ids", 1)
        if E1.shape != E2.shape or E1.shape[1] != EMB_DIM:
            raise _lib.MMRecHipError("InfoNCE views must both be [n, %d]" % EMB_DIM)
        B = ids.numel()
        loss = torch.empty((), dtype=torch.float32, device=E1.device)
        ws = _ws(lib.mmrec_infonce_workspace_bytes(B), E1.device)
        _lib.check(lib.mmrec_infonce_fwd_f32(_p(E1), _p(E2), _p(ids), B, EMB_DIM, float(tau), _p(loss),
                                             _p(ws), _stream()), "infonce_fwd")
        ctx.save_for_backward(ids, ws)
        ctx.tau, ctx.shape = float(tau), E1.shape
        return loss

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        ids, ws = ctx.saved_tensors
        g = g.contiguous().to(torch.float32)
        dE1 = torch.zeros(ctx.shape, dtype=torch.float32, device=g.device) if ctx.needs_input_grad[0] else None
        dE2 = torch.zeros(ctx.shape, dtype=torch.float32, device=g.device) if ctx.needs_input_grad[1] else None
        if DETERMINISTIC:      # per-sample rows (identity ids: one write each), then the position-ordered scatter
            B = ids.numel()
            ar = torch.arange(B, device=g.device)
            r1 = torch.zeros(B, EMB_DIM, dtype=torch.float32, device=g.device) if dE1 is not None else None
            r2 = torch.zeros(B, EMB_DIM, dtype=torch.float32, device=g.device) if dE2 is not None else None
            _lib.check(lib.mmrec_infonce_bwd_f32(_p(ar), B, EMB_DIM, ctx.tau, _p(g), _p(r1), _p(r2), _p(ws), _stream()),
                       "infonce_bwd")
            if dE1 is not None:
                scatter_add_rows(ids, r1, dE1)
            if dE2 is not None:
                scatter_add_rows(ids, r2, dE2)
            return dE1, dE2, None, None
        _lib.check(lib.mmrec_infonce_bwd_f32(_p(ids), ids.numel(), EMB_DIM, ctx.tau, _p(g), _p(dE1),
                                             _p(dE2), _p(ws), _stream()), "infonce_bwd")
        return dE1, dE2, None, None


def infonce(E1, E2, ids, tau):
    """Fused in-batch InfoNCE between rows `ids` of two [n, 64] views (MGCN.InfoNCE, mgcn.py:224-231:
    F.normalize both, positives on the diagonal, all B columns as negatives, mean over the batch)."""
    return _InfoNCE.apply(E1, E2, ids, tau)


class _GatherSqNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, E, ids):
        lib = _lib.load()
        E = _chk(E.contiguous(), torch.float32, "E", 2)
        _chk(ids, torch.int64, "ids", 1)
        out = torch.empty((), dtype=torch.float32, device=E.device)
        ws = _ws(4 * ids.numel(), E.device)
        _lib.check(lib.mmrec_gather_sqnorm_fwd_f32(_p(E), _p(ids), ids.numel(), E.shape[1], _p(out),
                                                   _p(ws), _stream()), "gather_sqnorm_fwd")
        ctx.save_for_backward(E, ids)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        E, ids = ctx.saved_tensors
        coef = (2.0 * g).contiguous().to(torch.float32)
        dE = torch.zeros_like(E)
        if DETERMINISTIC:
            return scatter_add_rows(ids, E.index_select(0, ids) * coef, dE), None
        _lib.check(lib.mmrec_gather_scale_add_bwd_f32(_p(E), _p(ids), ids.numel(), E.shape[1], _p(coef),
                                                      _p(dE), _stream()), "gather_scale_add_bwd")
        return dE, None


def gather_sqnorm(E, ids):
    """sum_b ||E[ids[b]]||^2 (differentiable): building block of EmbLoss / L2Loss on batch rows
    (common/loss.py:46-62 as used at lightgcn.py:145-149, layergcn.py:154-161)."""
    return _GatherSqNorm.apply(E, ids)


class _RowsReg(torch.autograd.Function):
    """scale * sum_t f(sum_b ||E_t[ids_t[b]]||^2) for ALL terms of a step's regulariser: mmrec_rows_reg_fwd_f32 / _bwd_f32
    (ABI 14).  Inputs: mode, scale, then E_0, ids_0, E_1, ids_1, ...; a table that appears in several terms (the item table's
    positive and negative rows) gets ONE dense gradient buffer, returned for its first occurrence."""

    @staticmethod
    def forward(ctx, mode, scale, *flat):
        lib = _lib.load()
        tables = [_chk(flat[2 * t].contiguous(), torch.float32, "E", 2) for t in range(len(flat) // 2)]
        ids = [None if flat[2 * t + 1] is None else _chk(flat[2 * t + 1], torch.int64, "ids", 1) for t in range(len(flat) // 2)]
        n, d, dev = len(tables), tables[0].shape[1], tables[0].device
        if any(E.shape[1] != d for E in tables) or d % EMB_DIM:
            raise _lib.MMRecHipError("rows_reg: every table needs the same row width, a multiple of %d" % EMB_DIM)
        batch = (ctypes.c_int32 * n)(*[E.shape[0] if i is None else i.numel() for E, i in zip(tables, ids)])   # ids None: every row
        ctx.arrays = ((ctypes.c_void_p * n)(*[E.data_ptr() for E in tables]),
                      (ctypes.c_void_p * n)(*[None if i is None else i.data_ptr() for i in ids]), batch)
        out = torch.empty((), dtype=torch.float32, device=dev)
        coef = torch.empty(n, dtype=torch.float32, device=dev)
        ws = _ws(lib.mmrec_rows_reg_workspace_bytes(n, max(batch)), dev)
        _lib.check(lib.mmrec_rows_reg_fwd_f32(ctx.arrays[0], ctx.arrays[1], batch, n, d, int(mode), float(scale), _p(out), _p(coef),
                                              _p(ws), _stream()), "rows_reg_fwd")
        ctx.save_for_backward(coef, *tables, *[i for i in ids if i is not None])     # (the ids: kept alive for the backward's pointers)
        ctx.n = n
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        n = ctx.n
        coef, tables = ctx.saved_tensors[0], ctx.saved_tensors[1:1 + n]
        g = g.contiguous().to(torch.float32)
        grads, first, out = {}, {}, [None, None]
        for t, E in enumerate(tables):
            if E.data_ptr() not in grads:
                grads[E.data_ptr()], first[E.data_ptr()] = torch.zeros_like(E), t
        dE = (ctypes.c_void_p * n)(*[grads[E.data_ptr()].data_ptr() for E in tables])
        _lib.check(lib.mmrec_rows_reg_bwd_f32(ctx.arrays[0], ctx.arrays[1], ctx.arrays[2], n, tables[0].shape[1], _p(coef), _p(g), dE,
                                              _stream()), "rows_reg_bwd")
        for t, E in enumerate(tables):
            out += [grads[E.data_ptr()] if (first[E.data_ptr()] == t and ctx.needs_input_grad[2 + 2 * t]) else None, None]
        return tuple(out)


ROWS_REG_SQUARED, ROWS_REG_NORM = 0, 1


def rows_reg(terms, mode, scale=1.0):
    """The regulariser of a training step over batch rows, fused: scale * sum_t ||E_t[ids_t]||_F^2 (mode ROWS_REG_SQUARED: the L2
    regulariser of layergcn.py:154-161 / lattice.py:214-216) or scale * sum_t ||E_t[ids_t]||_F (ROWS_REG_NORM: EmbLoss,
    common/loss.py:46-51).  terms = [(table [n, 64 k], ids [B]) ...].  Two launches forward, one backward, whatever the number
    of terms; `hip_deterministic`, more than MMREC_ROWS_REG_MAX_TERMS terms or tables that are views of one another's storage
    take the per-term ops."""
    terms = list(terms)
    same_rows = len({E.shape[1] for E, _ in terms}) == 1
    if DETERMINISTIC or len(terms) > 6 or not terms or not same_rows or any(not E.is_contiguous() for E, _ in terms):
        total = 0.0
        for E, ids in terms:
            s = (E * E).sum() if ids is None else gather_sqnorm(E, ids)
            total = total + (s if mode == ROWS_REG_SQUARED else torch.sqrt(s))
        return scale * total
    flat = [x for E, ids in terms for x in (E, None if ids is None else ids.contiguous())]
    return _RowsReg.apply(mode, scale, *flat)


class _RowNormalize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, eps):
        lib = _lib.load()
        X = _chk(X.contiguous(), torch.float32, "X", 2)
        if X.shape[1] % 4:
            raise _lib.MMRecHipError("row_normalize: rows of a multiple of 4 floats")
        Y = torch.empty_like(X)
        inv = torch.empty(max(X.shape[0], 1), dtype=torch.float32, device=X.device)
        _lib.check(lib.mmrec_row_normalize_fwd_f32(_p(X), X.shape[0], X.shape[1], float(eps), _p(Y), _p(inv), _stream()),
                   "row_normalize_fwd")
        ctx.save_for_backward(Y, inv)
        return Y

    @staticmethod
    def backward(ctx, G):
        lib = _lib.load()
        Y, inv = ctx.saved_tensors
        G = G.contiguous()
        dX = torch.empty_like(Y)
        _lib.check(lib.mmrec_row_normalize_bwd_f32(_p(Y), _p(G), _p(inv), Y.shape[0], Y.shape[1], _p(dX), _stream()),
                   "row_normalize_bwd")
        return dX, None


def row_normalize(X, eps=1e-12):
    """F.normalize(X, p=2, dim=1) in one launch each way (lattice.py:165, mmgcn.py:167)."""
    return _RowNormalize.apply(X, eps)


class _CatLeaky(torch.autograd.Function):
    @staticmethod
    def forward(ctx, A, B, R, slope):
        lib = _lib.load()
        A, B = _chk(A.contiguous(), torch.float32, "A", 2), _chk(B.contiguous(), torch.float32, "B", 2)
        if R is not None:
            R = _chk(R.contiguous(), torch.float32, "R", 2)
        n, wa, wb = A.shape[0], A.shape[1], B.shape[1]
        if B.shape[0] != n or (R is not None and R.shape != B.shape) or wa % 4 or wb % 4:
            raise _lib.MMRecHipError("cat_leaky: A [n, wa], B [n, wb], R [n, wb] with wa, wb multiples of 4")
        out = torch.empty(n, wa + wb, dtype=torch.float32, device=A.device)
        _lib.check(lib.mmrec_cat_leaky_fwd_f32(_p(A), _p(B), _p(R), n, wa, wb, float(slope), _p(out), _stream()), "cat_leaky_fwd")
        ctx.save_for_backward(A, B)
        ctx.slope = float(slope)
        return out

    @staticmethod
    def backward(ctx, dOut):
        lib = _lib.load()
        A, B = ctx.saved_tensors
        dOut = dOut.contiguous()
        dA = torch.empty_like(A) if ctx.needs_input_grad[0] else None
        dB = torch.empty_like(B) if ctx.needs_input_grad[1] else None
        dR = torch.empty_like(B) if ctx.needs_input_grad[2] else None
        _lib.check(lib.mmrec_cat_leaky_bwd_f32(_p(A), _p(B), _p(dOut), A.shape[0], A.shape[1], B.shape[1], ctx.slope, _p(dA), _p(dB),
                                               _p(dR), _stream()), "cat_leaky_bwd")
        return dA, dB, dR, None


def cat_leaky(A, B, R=None, slope=0.01):
    """cat((leaky_relu(A), leaky_relu(B) + R), dim=1) in one launch each way (mmgcn.py:170-173: h, x_hat and their cat)."""
    return _CatLeaky.apply(A, B, R, slope)


class _CosineMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, ix, Y, iy):
        lib = _lib.load()
        X, Y = _chk(X.contiguous(), torch.float32, "X", 2), _chk(Y.contiguous(), torch.float32, "Y", 2)
        for t, nm in ((ix, "ix"), (iy, "iy")):
            if t is not None:
                _chk(t, torch.int64, nm, 1)
        B = ix.numel() if ix is not None else (iy.numel() if iy is not None else X.shape[0])
        if X.shape[1] != Y.shape[1] or X.shape[1] % EMB_DIM:
            raise _lib.MMRecHipError("X and Y need the same row width, a multiple of %d" % EMB_DIM)
        if (ix is None and X.shape[0] != B) or (iy is None and Y.shape[0] != B):
            raise _lib.MMRecHipError("an operand without an index must have one row per sample")
        out = torch.empty((), dtype=torch.float32, device=X.device)
        coef = torch.empty(max(B, 1), 2, dtype=torch.float32, device=X.device)
        ws = _ws(lib.mmrec_cosine_workspace_bytes(B), X.device)
        _lib.check(lib.mmrec_cosine_fwd_f32(_p(X), _p(ix), _p(Y), _p(iy), B, X.shape[1], 1.0 / max(B, 1), _p(out),
                                            _p(coef), _p(ws), _stream()), "cosine_fwd")
        ctx.save_for_backward(X, Y, coef, *[t for t in (ix, iy) if t is not None])
        ctx.has = (ix is not None, iy is not None)
        ctx.B = B
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        X, Y, coef, *idx = ctx.saved_tensors
        ix = idx.pop(0) if ctx.has[0] else None
        iy = idx.pop(0) if ctx.has[1] else None
        dX = torch.zeros_like(X)
        g = g.contiguous().to(torch.float32)
        if DETERMINISTIC and ix is not None:      # gathered operands, one gradient row per sample, position-ordered scatter
            Xg = X.index_select(0, ix)
            Yg = Y.index_select(0, iy) if iy is not None else Y
            rows = torch.zeros_like(Xg)
            _lib.check(lib.mmrec_cosine_bwd_f32(_p(Xg), None, _p(Yg), None, ctx.B, X.shape[1], _p(coef), _p(g),
                                                1.0 / max(ctx.B, 1), _p(rows), _stream()), "cosine_bwd")
            return scatter_add_rows(ix, rows, dX), None, None, None
        _lib.check(lib.mmrec_cosine_bwd_f32(_p(X), _p(ix), _p(Y), _p(iy), ctx.B, X.shape[1], _p(coef), _p(g),
                                            1.0 / max(ctx.B, 1), _p(dX), _stream()), "cosine_bwd")
        return dX, None, None, None


class _CosineMeans(torch.autograd.Function):
    """sum_t w_t mean_b cos(X_t[ix_t[b]], Y_t[iy_t[b]]): mmrec_cosine_multi_fwd_f32 / _bwd_f32 (ABI 14).  Inputs: the weights
    (tuple of floats), then X_0, ix_0, Y_0, iy_0, X_1, ... (indices may be None); terms that share X share its gradient."""

    @staticmethod
    def forward(ctx, weights, *flat):
        lib = _lib.load()
        n = len(weights)
        X = [_chk(flat[4 * t].contiguous(), torch.float32, "X", 2) for t in range(n)]
        Y = [_chk(flat[4 * t + 2].contiguous(), torch.float32, "Y", 2) for t in range(n)]
        ix, iy = [flat[4 * t + 1] for t in range(n)], [flat[4 * t + 3] for t in range(n)]
        d, dev = X[0].shape[1], X[0].device
        batch = []
        for t in range(n):
            for i, nm in ((ix[t], "ix"), (iy[t], "iy")):
                if i is not None:
                    _chk(i, torch.int64, nm, 1)
            B = ix[t].numel() if ix[t] is not None else (iy[t].numel() if iy[t] is not None else X[t].shape[0])
            if X[t].shape[1] != d or Y[t].shape[1] != d or d % EMB_DIM:
                raise _lib.MMRecHipError("cosine_means: every operand needs the same row width, a multiple of %d" % EMB_DIM)
            if (ix[t] is None and X[t].shape[0] != B) or (iy[t] is None and Y[t].shape[0] != B):
                raise _lib.MMRecHipError("an operand without an index must have one row per sample")
            batch.append(B)
        ptrs = lambda ts: (ctypes.c_void_p * n)(*[None if t is None else t.data_ptr() for t in ts])
        ctx.arrays = (ptrs(X), ptrs(ix), ptrs(Y), ptrs(iy), (ctypes.c_float * n)(*[float(w) for w in weights]), (ctypes.c_int32 * n)(*batch))
        out = torch.empty((), dtype=torch.float32, device=dev)
        coef = torch.empty(n, max(max(batch), 1), 2, dtype=torch.float32, device=dev)
        ws = _ws(lib.mmrec_cosine_multi_workspace_bytes(n, max(batch)), dev)
        _lib.check(lib.mmrec_cosine_multi_fwd_f32(*ctx.arrays, n, d, _p(out), _p(coef), _p(ws), _stream()), "cosine_multi_fwd")
        ctx.save_for_backward(coef, *X, *Y, *[i for i in ix + iy if i is not None])
        ctx.n = n
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        n = ctx.n
        coef, X = ctx.saved_tensors[0], ctx.saved_tensors[1:1 + n]
        g = g.contiguous().to(torch.float32)
        grads, first = {}, {}
        for t, x in enumerate(X):
            if ctx.needs_input_grad[1 + 4 * t] and x.data_ptr() not in grads:
                grads[x.data_ptr()], first[x.data_ptr()] = torch.zeros_like(x), t
        dX = (ctypes.c_void_p * n)(*[grads[x.data_ptr()].data_ptr() if x.data_ptr() in grads else None for x in X])
        _lib.check(lib.mmrec_cosine_multi_bwd_f32(*ctx.arrays, n, X[0].shape[1], _p(coef), _p(g), dX, _stream()), "cosine_multi_bwd")
        out = [None]
        for t, x in enumerate(X):
            out += [grads[x.data_ptr()] if first.get(x.data_ptr()) == t else None, None, None, None]
        return tuple(out)


def cosine_means(terms):
    """sum_t w_t mean_b cosine_similarity(X_t[ix_t[b]], Y_t[iy_t[b]]) for terms = [(X, ix, Y, iy, w), ...] in one launch pair
    (Y constant, indices may be None): BM3's six BYOL terms (bm3.py:129-144).  `hip_deterministic` or more than
    MMREC_COSINE_MAX_TERMS terms: the per-term op."""
    terms = list(terms)
    if DETERMINISTIC or not terms or len(terms) > 8:
        total = 0.0
        for X, ix, Y, iy, w in terms:
            total = total + w * cosine_mean(X, ix, Y, iy)
        return total
    flat = [a for X, ix, Y, iy, _ in terms for a in (X, ix, Y.detach(), iy)]
    return _CosineMeans.apply(tuple(float(w) for *_, w in terms), *flat)


def cosine_mean(X, ix, Y, iy):
    """mean_b cosine_similarity(X[ix[b]], Y[iy[b]]) as ONE fused gather-dot-norm kernel (+ a scatter backward into X;
    Y is treated as a constant, as BM3's detached targets are): bm3.py:129-144, selfcfed_lgn.py:57-58.  ix / iy may
    be None (row b of the operand)."""
    return _CosineMean.apply(X, ix, Y.detach(), iy)


# ------------------------------------------------------------------------------------------------
# P3  modal projection
# ------------------------------------------------------------------------------------------------
LINEAR_SPLIT_MIN_F = 1024      # narrower inputs take the fp32-MFMA forward


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, W, b):
        lib = _lib.load()
        X = _chk(X.contiguous(), torch.float32, "X", 2)
        W = _chk(W.contiguous(), torch.float32, "W", 2)
        n, F = X.shape
        if W.shape != (64, F):
            raise _lib.MMRecHipError("W must be [64, %d], got %s" % (F, tuple(W.shape)))
        if b is not None:
            b = _chk(b.contiguous(), torch.float32, "b", 1)
        Y = torch.empty(n, 64, dtype=torch.float32, device=X.device)
        ws = _ws(lib.mmrec_linear_workspace_bytes(n, F, 64), X.device)
        # narrow tables (the 384-wide text features) are launch bound, not stream bound: the fp32 kernel is ONE launch where the
        # split form is three or four (W split, kernel, slab sum, guard fix-up): 19 us against 24 at 7,050 x 384
        fwd = lib.mmrec_linear_fwd_split_f32 if (LINEAR_F16X3 and F >= LINEAR_SPLIT_MIN_F) else lib.mmrec_linear_fwd_f32
        _lib.check(fwd(_p(X), _p(W), _p(b), _p(Y), n, F, 64, _p(ws), _stream()), "linear_fwd")
        ctx.save_for_backward(X, W)
        ctx.has_b = b is not None
        return Y

    @staticmethod
    def backward(ctx, dY):
        lib = _lib.load()
        X, W = ctx.saved_tensors
        n, F = X.shape
        dY = dY.contiguous()
        dX = dW = db = None
        want_w = ctx.needs_input_grad[1] or (ctx.has_b and ctx.needs_input_grad[2])
        if want_w:
            dW = torch.empty_like(W)
            db = torch.empty(64, dtype=torch.float32, device=X.device) if ctx.has_b else None
        if ctx.needs_input_grad[0]:
            dX = torch.empty_like(X)
        if LINEAR_F16X3 and n > 0 and F % 128 == 0:
            # dW, db and dX in one call on the 16-bit matrix cores with split operands (ABI 11: dW on three bf16 parts, dX on fp16
            # halves scaled by exact powers of two)
            ws = _ws(lib.mmrec_linear_bwd_split_workspace_bytes(n, F, 64), X.device)
            _lib.check(lib.mmrec_linear_bwd_split_f32(_p(dY), _p(X), _p(W), _p(dW), _p(db), _p(dX), n, F, 64, _p(ws),
                                                      _stream()), "linear_bwd_split")
            return dX, dW, db
        if want_w:
            ws = _ws(lib.mmrec_linear_workspace_bytes(n, F, 64), X.device)
            _lib.check(lib.mmrec_linear_bwd_w_f32(_p(dY), _p(X), _p(dW), _p(db), n, F, 64, _p(ws),
                                                  _stream()), "linear_bwd_w")
        if dX is not None:
            _lib.check(lib.mmrec_linear_bwd_x_f32(_p(dY), _p(W), _p(dX), n, F, 64, _stream()),
                       "linear_bwd_x")
        return dX, dW, db


class _LinearWide(torch.autograd.Function):
    """F.linear with out = 64 j > 64 and F % 32 == 0: forward and dX on the 128 x 128 LDS-DMA GEMM
    (mmrec_gemm_nt_f32), dW / db on the split-over-rows kernel, 64 output columns at a time."""

    @staticmethod
    def forward(ctx, X, W, b):
        lib = _lib.load()
        X = _chk(X.contiguous(), torch.float32, "X", 2)
        W = _chk(W.contiguous(), torch.float32, "W", 2)
        n, F = X.shape
        out = W.shape[0]
        if W.shape[1] != F or out % 64 or F % 32:
            raise _lib.MMRecHipError("wide linear needs W [64 j, F], F %% 32 == 0; got W %s, X %s" %
                                     (tuple(W.shape), tuple(X.shape)))
        if b is not None:
            b = _chk(b.contiguous(), torch.float32, "b", 1)
        Y = torch.empty(n, out, dtype=torch.float32, device=X.device)
        _lib.check(lib.mmrec_gemm_nt_f32(_p(X), _p(W), _p(b), _p(Y), n, out, F, out, _stream()), "gemm_nt")
        ctx.save_for_backward(X, W)
        ctx.has_b = b is not None
        return Y

    @staticmethod
    def backward(ctx, dY):
        lib = _lib.load()
        X, W = ctx.saved_tensors
        n, F = X.shape
        out = W.shape[0]
        dY = dY.contiguous()
        dX = dW = db = None
        if ctx.needs_input_grad[1] or (ctx.has_b and ctx.needs_input_grad[2]):
            dW = torch.empty_like(W)
            db = torch.empty(out, dtype=torch.float32, device=X.device) if ctx.has_b else None
            ws = _ws(lib.mmrec_linear_workspace_bytes(n, F, out), X.device)
            _lib.check(lib.mmrec_linear_bwd_w_f32(_p(dY), _p(X), _p(dW), _p(db), n, F, out, _p(ws),
                                                  _stream()), "linear_bwd_w")
        if ctx.needs_input_grad[0]:
            dX = torch.empty_like(X)
            Wt = W.t().contiguous()                         # [F, out]: dX = dY W = dY (W^T)^T
            _lib.check(lib.mmrec_gemm_nt_f32(_p(dY), _p(Wt), None, _p(dX), n, F, out, F, _stream()), "gemm_nt")
        return dX, dW, db


def linear(X, W, b=None):
    """X @ W^T + b on the fp32 matrix cores.  out = 64: nn.Linear image_trs / text_trs / item_linear
    (freedom.py:205,208; bm3.py:102,104; vbpr.py:70) on the projection kernels; out = 64 j with
    F % 32 == 0 (MMGCN's 256 / 384-wide layers): the general LDS-DMA GEMM."""
    if W.shape[0] == 64:
        return _Linear.apply(X, W, b)
    return _LinearWide.apply(X, W, b)


# ------------------------------------------------------------------------------------------------
# P5 / P6  fused score + mask + top-K
# ------------------------------------------------------------------------------------------------
def mask_to_csr(mask, n_rows, device):
    """[2, n] (row, item) mask of EvalDataLoader (dataloader.py:359-368) -> CSR with item ids sorted
    inside each row (the kernel binary-searches them).  Host side, integer exact."""
    m = mask.detach().cpu().numpy() if isinstance(mask, torch.Tensor) else np.asarray(mask)
    m = m.astype(np.int64)
    order = np.lexsort((m[1], m[0]))
    rowptr = np.zeros(n_rows + 1, dtype=np.int64)
    np.cumsum(np.bincount(m[0], minlength=n_rows), out=rowptr[1:])
    return (torch.from_numpy(rowptr.astype(np.int32)).to(device),
            torch.from_numpy(m[1][order].astype(np.int32)).to(device))


TOPK_NO_FILTER = 1


class TopkCandidates:
    """A candidate table [nc, kd] together with the candidate side of the fp16 top-K filter (column means, centred fp16
    copy, norms: mmrec_topk_prepare_f32), computed ONCE for all the query blocks ranked against it -- the batches of one
    evaluation and its valid / test pair (trainer.py:262,271,298-310: the item table is frozen while evaluating).
    `score_topk(Q, TopkCandidates(C), ...)` == `score_topk(Q, C, ...)` bit for bit.  Valid while C is unchanged."""

    def __init__(self, C):
        lib = _lib.load()
        self.C = _chk(C.contiguous(), torch.float32, "C", 2)
        nc, kd = self.C.shape
        nbytes = lib.mmrec_topk_prepared_bytes(nc, kd)
        self.prepared = None
        if nbytes:          # 0: the filter does not serve this shape; calls take the plain entry point
            self.prepared = _ws(nbytes, self.C.device)
            _lib.check(lib.mmrec_topk_prepare_f32(_p(self.C), nc, kd, _p(self.prepared), _stream()), "topk_prepare")


def topk_hint_served(nc, kd, k):
    """does the warm entry point (mmrec_score_topk_hinted_f32) serve this candidate table / k?  (the fp16 filter's shapes)"""
    return k <= TOPK_MAX and _lib.load().mmrec_topk_prepared_bytes(int(nc), int(kd)) > 0


TOPK_HINT_COLD, TOPK_HINT_KEEP = 1, 2


def topk_hint_width(k):
    """natural width of a list table for top-k calls: the final kernel ranks 64 (k <= 64) or 128 survivors anyway"""
    return 64 if k <= 64 else 128


def score_topk(Q, C, k, mask_rowptr=None, mask_col=None, return_values=False, use_filter=True, hint=None, hint_rows=None,
               queue_counts=None, hint_cold=False, hint_update=True):
    """top-k over candidates c of <Q[q], C[c]> per query with masked candidates at -1e10; never
    materialises the score matrix.  Returns int64 [nq, k] sorted by score desc (ties: lower id).
    C: a tensor, or a TopkCandidates (the candidate-side preparation of the fp16 filter done once for many calls).
    use_filter=False keeps the materialised fp32 path where the fp16 filter would serve the call (A/B measurements).
    hint (int32 [rows, k <= hk <= 128], IN / OUT, with hint_rows int64 [nq] or one row per query): a WARM call -- per query a
    list of ids expected to rank high (what a previous call left in the row for the same user); the filter takes its threshold
    from them and runs ONE matrix-core pass instead of two, and (hint_update) leaves this call's ranking -- top-k + runners-up
    -- in the row.  hint_cold: the rows are only written (the first evaluation).  The result does not depend on the hint
    (bit-identical to the plain call); shapes the filter does not serve ignore it.  queue_counts (int32 [2], device): +=
    queries the slow / overflow queues served."""
    lib = _lib.load()
    Q = _chk(Q.contiguous(), torch.float32, "Q", 2)
    prepared = None
    if isinstance(C, TopkCandidates):
        C, prepared = C.C, (C.prepared if use_filter else None)
    C = _chk(C.contiguous(), torch.float32, "C", 2)
    nq, kd = Q.shape
    nc = C.shape[0]
    if C.shape[1] != kd:
        raise _lib.MMRecHipError("Q and C must share the inner dim")
    if mask_rowptr is not None:
        _chk(mask_rowptr, torch.int32, "mask_rowptr", 1)
        if mask_col is None or mask_col.numel() == 0:
            mask_col = torch.zeros(1, dtype=torch.int32, device=Q.device)
        _chk(mask_col, torch.int32, "mask_col", 1)
    if hint is not None and not (use_filter and nc >= k and topk_hint_served(nc, kd, k)):
        hint = None
    if hint is not None:
        _chk(hint, torch.int32, "hint", 2)
        if hint.shape[1] < k or hint.shape[1] > TOPK_MAX:
            raise _lib.MMRecHipError("hint rows hold k <= hk <= %d ids, got %d for k = %d" % (TOPK_MAX, hint.shape[1], k))
        if hint_rows is not None:
            _chk(hint_rows, torch.int64, "hint_rows", 1)
            if hint_rows.numel() != nq:
                raise _lib.MMRecHipError("hint_rows: one row index per query")
        elif hint.shape[0] != nq:
            raise _lib.MMRecHipError("hint: one row per query (or pass hint_rows)")
        if queue_counts is not None:
            _chk(queue_counts, torch.int32, "queue_counts", 1)
    if nq > 2 * TOPK_QUERY_BLOCK and use_filter and (prepared is not None or lib.mmrec_topk_prepared_bytes(nc, kd) > 0):
        return _score_topk_blocked(Q, C, prepared, k, mask_rowptr, mask_col, return_values, hint, hint_rows, queue_counts,
                                   hint_cold, hint_update)
    idx = torch.empty(nq, k, dtype=torch.int64, device=Q.device)
    val = torch.empty(nq, k, dtype=torch.float32, device=Q.device) if return_values else None
    ws = _ws(lib.mmrec_topk_workspace_bytes(nq, nc, kd, k), Q.device)
    if hint is not None:
        _lib.check(lib.mmrec_score_topk_hinted_f32(_p(Q), _p(C), _p(prepared), nq, nc, kd, _p(mask_rowptr), _p(mask_col), k,
                                                   _p(hint), hint.shape[1], _p(hint_rows), _p(idx), _p(val), _p(ws),
                                                   _p(queue_counts), (TOPK_HINT_COLD if hint_cold else 0) |
                                                   (0 if hint_update else TOPK_HINT_KEEP), _stream()), "score_topk_hinted")
    elif prepared is not None:
        _lib.check(lib.mmrec_score_topk_prepared_f32(_p(Q), _p(C), _p(prepared), nq, nc, kd, _p(mask_rowptr), _p(mask_col), k,
                                                     _p(idx), _p(val), _p(ws), 0, _stream()), "score_topk_prepared")
    else:
        _lib.check(lib.mmrec_score_topk_f32(_p(Q), _p(C), nq, nc, kd, _p(mask_rowptr), _p(mask_col), k,
                                            _p(idx), _p(val), _p(ws), 0 if use_filter else TOPK_NO_FILTER, _stream()),
                   "score_topk")
    return (idx, val) if return_values else idx


TOPK_QUERY_BLOCK = 65536     # the Trainer's `hip_eval_batch_size`: 256 query blocks x 16 candidate ranges = 8 exact rounds of workgroups


def _score_topk_blocked(Q, C, prepared, k, mask_rowptr, mask_col, return_values, hint=None, hint_rows=None, queue_counts=None,
                        hint_cold=False, hint_update=True):
    """Very many queries in ONE call (a script ranking all 1M users at once): the fp16 filter's workspace is per query
    (~16 KB of word list at 500K candidates: 16 GB for 1M queries), so the call is walked in blocks of TOPK_QUERY_BLOCK queries
    against ONE preparation of the candidates -- what the Trainer's evaluation batches amount to.  One small device -> host
    read of the blocks' mask offsets (not for use under graph capture)."""
    nq = Q.shape[0]
    cands = TopkCandidates.__new__(TopkCandidates)
    cands.C = C
    cands.prepared = prepared if prepared is not None else TopkCandidates(C).prepared
    starts = list(range(0, nq, TOPK_QUERY_BLOCK))
    offs = None
    if mask_rowptr is not None:
        offs = mask_rowptr[torch.tensor(starts + [nq], device=mask_rowptr.device)].cpu().tolist()
    idx = torch.empty(nq, k, dtype=torch.int64, device=Q.device)
    val = torch.empty(nq, k, dtype=torch.float32, device=Q.device) if return_values else None
    for j, a in enumerate(starts):
        b = min(a + TOPK_QUERY_BLOCK, nq)
        rp = col = None
        if mask_rowptr is not None:
            rp = (mask_rowptr[a:b + 1] - offs[j]).contiguous()
            col = mask_col[offs[j]:max(offs[j + 1], offs[j] + 1)].contiguous()
        h = hr = None
        if hint is not None:
            h, hr = (hint, hint_rows[a:b]) if hint_rows is not None else (hint[a:b], None)
        out = score_topk(Q[a:b], cands, k, rp, col, return_values=return_values, hint=h, hint_rows=hr, queue_counts=queue_counts,
                         hint_cold=hint_cold, hint_update=hint_update)
        if return_values:
            idx[a:b], val[a:b] = out
        else:
            idx[a:b] = out
    return (idx, val) if return_values else idx


# ------------------------------------------------------------------------------------------------
# P1  graph build on device
# ------------------------------------------------------------------------------------------------
def degree_count(ids, n_bins):
    lib = _lib.load()
    _chk(ids, torch.int64, "ids", 1)
    counts = torch.zeros(n_bins, dtype=torch.int32, device=ids.device)
    _lib.check(lib.mmrec_degree_count_i32(_p(ids), ids.numel(), _p(counts), n_bins, _stream()),
               "degree_count")
    return counts


def edge_norm_values(eu, ei, n_users, n_items):
    """(du+1e-7)^-1/2 (di+1e-7)^-1/2 per edge in fp32 (freedom.py:145-154)."""
    lib = _lib.load()
    du, di = degree_count(eu, n_users), degree_count(ei, n_items)
    val = torch.empty(eu.numel(), dtype=torch.float32, device=eu.device)
    _lib.check(lib.mmrec_edge_norm_f32(_p(eu), _p(ei), eu.numel(), _p(du), _p(di), _p(val), _stream()),
               "edge_norm")
    return val


def bipartite_graph_from_edges(eu, ei, n_users, n_items, long_row_threshold=LONG_ROW_DEFAULT):
    """Symmetric normalised adjacency of the given (user,item) edges as a CsrGraph, built on device:
    re-normalise on this edge set, expand to cat(edges, flipped) and stable-sort into CSR.
    replaces pre_epoch_processing's masked_adj build (freedom.py:136-143, layergcn.py:63-70)."""
    lib = _lib.load()
    eu, ei = _chk(eu.contiguous(), torch.int64, "eu", 1), _chk(ei.contiguous(), torch.int64, "ei", 1)
    E, dev = eu.numel(), eu.device
    w = edge_norm_values(eu, ei, n_users, n_items)
    rows = torch.empty(2 * E, dtype=torch.int32, device=dev)
    cols = torch.empty(2 * E, dtype=torch.int32, device=dev)
    vals = torch.empty(2 * E, dtype=torch.float32, device=dev)
    _lib.check(lib.mmrec_bipartite_expand(_p(eu), _p(ei), _p(w), E, n_users, _p(rows), _p(cols),
                                          _p(vals), _stream()), "bipartite_expand")
    n = n_users + n_items
    return CsrGraph.from_coo_device(rows, cols, vals, n, n, symmetric=True,
                                    long_row_threshold=long_row_threshold)


# ------------------------------------------------------------------------------------------------
# SpMM with differentiable VALUES (LATTICE's learned item graph, lattice.py:137-163)
# ------------------------------------------------------------------------------------------------
class DynGraph:
    """Structure (rows, cols) of a sparse matrix whose values carry gradient and change every build.

    The CSR structure and its transpose are derived on the device (stable sort of the row / column
    ids: plumbing), the HIP SpMM does the arithmetic; `perm` maps the caller's COO order to CSR order so
    a values tensor can be re-used without rebuilding anything."""

    def __init__(self, rows, cols, n_rows, n_cols, long_row_threshold=LONG_ROW_DEFAULT):
        _chk(rows, torch.int64, "rows", 1), _chk(cols, torch.int64, "cols", 1)
        self.rows, self.cols, self.n_rows, self.n_cols = rows, cols, int(n_rows), int(n_cols)
        dev = rows.device
        zeros = torch.zeros(rows.numel(), dtype=torch.float32, device=dev)

        def build(r, c, nr, nc):
            perm = torch.sort(r, stable=True)[1]
            rowptr = torch.zeros(nr + 1, dtype=torch.int64, device=dev)
            rowptr[1:] = torch.cumsum(torch.bincount(r, minlength=nr), 0)
            g = CsrGraph(rowptr.to(torch.int32), c[perm].to(torch.int32).contiguous(), zeros, nr, nc,
                         long_row_threshold=long_row_threshold)
            return g, perm
        self.fwd, self.perm = build(rows, cols, self.n_rows, self.n_cols)
        self.bwd, self.perm_t = build(cols, rows, self.n_cols, self.n_rows)


class _SpMMVals(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, vals, dyn):
        X = X.contiguous()
        dyn.fwd.vals = vals.detach()[dyn.perm].contiguous()
        Y = torch.empty(dyn.n_rows, EMB_DIM, dtype=torch.float32, device=X.device)
        spmm_raw(dyn.fwd, X, Y=Y)
        ctx.dyn = dyn
        ctx.save_for_backward(X, vals)
        return Y

    @staticmethod
    def backward(ctx, dY):
        dyn = ctx.dyn
        X, vals = ctx.saved_tensors
        dY = dY.contiguous()
        dX = dvals = None
        if ctx.needs_input_grad[0]:
            dyn.bwd.vals = vals.detach()[dyn.perm_t].contiguous()
            dX = torch.empty(dyn.n_cols, EMB_DIM, dtype=torch.float32, device=dY.device)
            spmm_raw(dyn.bwd, dY, Y=dX)
        if ctx.needs_input_grad[1]:
            dvals = (dY[dyn.rows] * X[dyn.cols]).sum(-1)   # d val_e = <dY[row_e], X[col_e]>
        return dX, dvals, None


def spmm_vals(dyn: DynGraph, X, vals):
    """A(vals) @ X, differentiable in X and in the per-entry values (COO order of `dyn`)."""
    return _SpMMVals.apply(X, vals, dyn)


# ------------------------------------------------------------------------------------------------
# Rows next to the hot path: device negative sampler, device ranking metrics
# ------------------------------------------------------------------------------------------------
def flat_to_csr(flat, lens, device):
    """row lists given as one concatenated id array + per-row lengths -> (rowptr int32, ids int32 sorted inside each row)
    on `device`; one global sort of (row, id) keys instead of a Python loop over the rows (1M evaluation users: 0.5 s)."""
    lens = np.asarray(lens, dtype=np.int64)
    flat = np.asarray(flat, dtype=np.int64)
    rowptr = np.zeros(lens.shape[0] + 1, dtype=np.int64)
    np.cumsum(lens, out=rowptr[1:])
    if flat.size:
        stride = np.int64(flat.max()) + 1
        key = np.repeat(np.arange(lens.shape[0], dtype=np.int64), lens) * stride + flat
        key.sort()
        flat = key % stride
    else:
        flat = np.zeros(1, np.int64)
    return (torch.from_numpy(rowptr.astype(np.int32)).to(device),
            torch.from_numpy(flat.astype(np.int32)).to(device))


def lists_to_csr(lists, device):
    """list of per-row id arrays -> (rowptr int32, ids int32 sorted inside each row) on `device`."""
    lens = np.fromiter((len(x) for x in lists), dtype=np.int64, count=len(lists))
    flat = np.concatenate([np.asarray(x, dtype=np.int64) for x in lists]) if len(lists) else np.zeros(0, np.int64)
    return flat_to_csr(flat, lens, device)


def sample_negatives(users, hist_rowptr, hist_col, cand_items, seed, counter):
    """One uniform negative per user id from `cand_items`, outside the user's history (f1)."""
    lib = _lib.load()
    _chk(users, torch.int64, "users", 1), _chk(hist_rowptr, torch.int32, "hist_rowptr", 1)
    _chk(hist_col, torch.int32, "hist_col", 1), _chk(cand_items, torch.int32, "cand_items", 1)
    out = torch.empty_like(users)
    _lib.check(lib.mmrec_sample_negatives_i64(_p(users), users.numel(), _p(hist_rowptr), _p(hist_col),
                                              _p(cand_items), cand_items.numel(), int(seed) & (2 ** 64 - 1),
                                              int(counter) & (2 ** 64 - 1), _p(out), _stream()),
               "sample_negatives")
    return out


def topk_metrics_per_user(topk_idx, gt_rowptr, gt_col, ks, want_hits=False):
    """Per-user Recall/NDCG/Precision/MAP at cut-offs `ks` -> float64 [n_users, 4, len(ks)] (f2)."""
    lib = _lib.load()
    topk_idx = _chk(topk_idx.contiguous(), torch.int64, "topk_idx", 2)
    _chk(gt_rowptr, torch.int32, "gt_rowptr", 1), _chk(gt_col, torch.int32, "gt_col", 1)
    n, k = topk_idx.shape
    dev = topk_idx.device
    ks = sorted(int(x) for x in ks)
    if ks[-1] > k or ks[0] < 1:
        raise _lib.MMRecHipError("cut-offs must lie in [1, k]")
    disc = 1.0 / np.log2(np.arange(1, k + 1, dtype=np.float64) + 1)
    d_disc = torch.from_numpy(disc).to(dev)
    d_idcg = torch.from_numpy(np.cumsum(disc)).to(dev)
    d_ks = torch.tensor(ks, dtype=torch.int32, device=dev)
    out = torch.empty(n, 4, len(ks), dtype=torch.float64, device=dev)
    hits = torch.empty(n, k, dtype=torch.uint8, device=dev) if want_hits else None
    _lib.check(lib.mmrec_topk_metrics_f64(_p(topk_idx), n, k, _p(gt_rowptr), _p(gt_col), _p(d_disc),
                                          _p(d_idcg), _p(d_ks), len(ks), _p(hits), _p(out), _stream()),
               "topk_metrics")
    return (out, hits) if want_hits else out
