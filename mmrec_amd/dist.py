"""Row-sharded propagation over the GPUs of one node.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests).
The normalised adjacency is bipartite: user rows read only item embeddings and item rows only user
embeddings.  Rank r owns one contiguous block of users and one of items, cut so that every rank holds the
same number of NONZEROS (not rows: SURVEY.md 8e), computes its rows of Y = A X with the HIP SpMM, and the
blocks are all-gathered over RCCL after every layer so that each rank holds the next layer's X
("all-gather of item embeddings ... after each GCN layer", BASELINE.json north_star).  A row partition does not
change any row's summation order: sharded == single GPU bit for bit.

Layout of the exchanged space.  Every rank's block is padded to the same capacity and cut into `n_chunks` row
chunks; ids are laid out CHUNK-MAJOR (chunk c of every rank next to each other), so the all-gather of chunk c
is one contiguous `all_gather_into_tensor`, issued as soon as the SpMM of chunk c is enqueued: it runs on RCCL's
stream under the SpMM of chunk c + 1 (and the user chunks' exchange under the item rows' SpMMs).  Only the last
chunk's exchange of a layer is exposed.

fp32 on the wire (the 1e-4 parity target forbids a bf16 exchange).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


# ------------------------------------------------------------------------------------------------
# partition
# ------------------------------------------------------------------------------------------------
def balanced_cuts(weight, parts):
    """Contiguous cut points [0 = c_0 <= c_1 <= ... <= c_parts = n] with ~equal sum(weight) per part."""
    w = np.asarray(weight, dtype=np.float64)
    cum = np.concatenate([[0.0], np.cumsum(w)])
    cuts = [0]
    for k in range(1, parts):
        cuts.append(max(cuts[-1], int(np.searchsorted(cum, cum[-1] * k / parts, "left"))))
    cuts.append(w.shape[0])
    return np.asarray(cuts, dtype=np.int64)


class RowSpace:
    """n rows dealt to P ranks in contiguous blocks [cuts[r], cuts[r+1]); every block padded to `cap` rows
    (a multiple of n_chunks), chunk-major positions starting at `base`:
        row = cuts[r] + c * cb + o   ->   base + c * (P * cb) + r * cb + o          (cb = cap / n_chunks)."""

    def __init__(self, n, world, weight=None, n_chunks=1, base=0):
        self.n, self.P, self.n_chunks, self.base = int(n), int(world), int(n_chunks), int(base)
        if weight is None:
            per = -(-self.n // self.P)
            self.cuts = np.minimum(np.arange(self.P + 1, dtype=np.int64) * per, self.n)
        else:
            assert len(weight) == self.n
            self.cuts = balanced_cuts(weight, self.P)
        big = int(np.diff(self.cuts).max()) if self.n else 0
        self.cb = max(-(-big // self.n_chunks), 1)
        self.cap = self.cb * self.n_chunks
        self.size = self.P * self.cap
        owner = np.repeat(np.arange(self.P, dtype=np.int64), np.diff(self.cuts))
        local = np.arange(self.n, dtype=np.int64) - self.cuts[owner]
        c, o = local // self.cb, local % self.cb
        self.pos = self.base + c * (self.P * self.cb) + owner * self.cb + o     # row -> padded position

    def block(self, rank):
        return int(self.cuts[rank]), int(self.cuts[rank + 1])

    def chunk_out(self, rank, c):
        """padded positions [lo, hi) of chunk c of `rank`"""
        lo = self.base + c * (self.P * self.cb) + rank * self.cb
        return lo, lo + self.cb

    def chunk_region(self, c):
        """padded positions [lo, hi) of chunk c of ALL ranks (what one all-gather fills)"""
        lo = self.base + c * (self.P * self.cb)
        return lo, lo + self.P * self.cb


class BipartiteSharding:
    """Users and items each dealt to P ranks (RowSpace); items sit behind the users: one padded space of
    N_pad = U_pad + I_pad positions.  `user_weight` / `item_weight` (per-row nonzero counts) give nnz-balanced
    blocks; without them blocks have equal row counts."""

    def __init__(self, n_users, n_items, world_size, user_weight=None, item_weight=None, n_chunks=1):
        self.n_users, self.n_items, self.P, self.n_chunks = int(n_users), int(n_items), int(world_size), int(n_chunks)
        self.users = RowSpace(n_users, world_size, user_weight, n_chunks, base=0)
        self.items = RowSpace(n_items, world_size, item_weight, n_chunks, base=self.users.size)
        self.ub, self.ib = self.users.cap, self.items.cap
        self.U_pad, self.I_pad = self.users.size, self.items.size
        self.N_pad = self.U_pad + self.I_pad
        self.pos = np.concatenate([self.users.pos, self.items.pos])        # node id (items offset by n_users) -> position
        self._pos_t = {}

    @classmethod
    def from_coo(cls, rows, n_users, n_items, world_size, n_chunks=1):
        """nnz-balanced: weights = nonzeros per row of the symmetric adjacency given by its COO row ids"""
        deg = np.bincount(np.asarray(rows, dtype=np.int64), minlength=n_users + n_items)
        return cls(n_users, n_items, world_size, deg[:n_users], deg[n_users:], n_chunks)

    # -- contiguous blocks (n_chunks == 1 only)
    def user_rows(self, r):
        assert self.n_chunks == 1
        return self.users.chunk_out(r, 0)

    def item_rows(self, r):
        assert self.n_chunks == 1
        return self.items.chunk_out(r, 0)

    def pos_tensor(self, device):
        key = str(device)
        if key not in self._pos_t:
            self._pos_t[key] = torch.from_numpy(self.pos).to(device)
        return self._pos_t[key]

    def pad(self, x):
        """[n_users + n_items, d] in node order -> [N_pad, d] (padding rows zero)"""
        out = x.new_zeros(self.N_pad, x.shape[1])
        out[self.pos_tensor(x.device)] = x
        return out

    def pad_embeddings(self, user_emb, item_emb):
        return self.pad(torch.cat([user_emb, item_emb], 0))

    def unpad_nodes(self, x):
        return x[self.pos_tensor(x.device)]

    def unpad(self, x):
        y = self.unpad_nodes(x)
        return y[:self.n_users], y[self.n_users:]

    def padded_coo(self, rows, cols):
        """Map node ids of the unpadded symmetric COO (items offset by n_users) into the padded space."""
        return self.pos[np.asarray(rows, dtype=np.int64)], self.pos[np.asarray(cols, dtype=np.int64)]

    def entries(self, rank):
        """[(kind, chunk, out_lo, out_hi, reg_lo, reg_hi)] of this rank: user chunks first, then item chunks"""
        out = []
        for kind, sp in (("u", self.users), ("i", self.items)):
            for c in range(sp.n_chunks):
                out.append((kind, c) + sp.chunk_out(rank, c) + sp.chunk_region(c))
        return out

    def rank_blocks(self, rows, cols, vals, rank, make_csr, node_cols=False):
        """This rank's row chunks of the adjacency given as COO over NODE ids (items offset by n_users), any order:
        -> (user chunk blocks, item chunk blocks), each `make_csr(local_rows, cols, vals, n_rows, n_cols)`.
        Entries keep their COO order inside a row (make_csr must be stable), so a row sums exactly as on one GPU.
        node_cols: column ids stay NODE ids (the block multiplies a table in node order: the first layer reads the
        replicated parameters without a permutation into the padded space); default: padded positions."""
        rows = np.asarray(rows, dtype=np.int64)
        pr, pc = self.padded_coo(rows, cols)
        if node_cols:
            pc = np.asarray(cols, dtype=np.int64)
        vals = np.asarray(vals)
        blocks = {"u": [], "i": []}
        for kind, c, lo, hi, _, _ in self.entries(rank):
            sel = np.nonzero((pr >= lo) & (pr < hi))[0]
            blocks[kind].append(make_csr(pr[sel] - lo, pc[sel], vals[sel], hi - lo,
                                         self.n_users + self.n_items if node_cols else self.N_pad))
        return blocks["u"], blocks["i"]

    def own_row_slices(self, table, rank):
        """Per entry of `rank`: the [chunk rows, d] rows of the node-order `table` ([n_users + n_items, d]) that the entry's
        padded rows stand for, as VIEWS where possible.  Rows past the end of the rank's block are whatever follows in
        the table (or zeros past its end): they land in padding positions no column id refers to."""
        out = []
        n = table.shape[0]
        for kind, sp, off in (("u", self.users, 0), ("i", self.items, self.n_users)):
            for c in range(sp.n_chunks):
                start = off + int(sp.cuts[rank]) + c * sp.cb
                if start + sp.cb <= n:
                    out.append(table[start:start + sp.cb])
                else:
                    head = table[min(start, n):n]
                    out.append(torch.cat([head, table.new_zeros(sp.cb - head.shape[0], table.shape[1])], 0))
        return out

    def nnz_per_rank(self, rows):
        owner_u = np.repeat(np.arange(self.P), np.diff(self.users.cuts))
        owner_i = np.repeat(np.arange(self.P), np.diff(self.items.cuts))
        owner = np.concatenate([owner_u, owner_i])
        return np.bincount(owner[np.asarray(rows, dtype=np.int64)], minlength=self.P)


def space_blocks(space, rows, cols, vals, rank, col_pos, n_cols, make_csr):
    """Row chunks of a matrix whose ROWS live in `space` (a RowSpace; row ids 0..space.n) and whose column ids are
    mapped through `col_pos` into a padded space of `n_cols` positions (item-item graphs: both are the item space)."""
    rows = np.asarray(rows, dtype=np.int64)
    pr = space.pos[rows]
    pc = np.asarray(cols, dtype=np.int64) if col_pos is None else col_pos[np.asarray(cols, dtype=np.int64)]   # None: as given
    vals = np.asarray(vals)
    out = []
    for c in range(space.n_chunks):
        lo, hi = space.chunk_out(rank, c)
        sel = np.nonzero((pr >= lo) & (pr < hi))[0]
        out.append(make_csr(pr[sel] - lo, pc[sel], vals[sel], hi - lo, n_cols))
    return out


# ------------------------------------------------------------------------------------------------
# sharded SpMM + chunked all-gather
# ------------------------------------------------------------------------------------------------
class RowShardedOp:
    """out[own rows] = A_own @ X (+ epilogue) chunk by chunk, each chunk all-gathered as soon as it is enqueued.

    entries: [(block, out_lo, out_hi, reg_lo, reg_hi)] -- positions are relative to the `out` tensor handed to
    `apply`.  `local_spmm(block, X, Y, **epilogue)` is the HIP kernel (hip_ops.spmm_raw) in the product and a scipy
    checker in the gloo tests; epilogue keys: Z, acc_in, acc_out (row slices), alpha, beta, acc_scale."""

    def __init__(self, entries, local_spmm, group=None, exchange=True):
        self.entries, self.local_spmm, self.group, self.exchange = list(entries), local_spmm, group, exchange
        self.bytes_gathered = 0     # payload this rank RECEIVED through the all-gathers (metrics)
        if exchange:
            # The all-gather of a chunk is IN PLACE: the input is the rank's own slot of the output region (what
            # ncclAllGather documents as sendbuff == recvbuff + rank * sendcount).  Any other overlap of input and output is
            # undefined for RCCL, so the layout contract is checked here, once, instead of being trusted per call
            world = dist.get_world_size(group) if dist.is_initialized() else 1
            rank = dist.get_rank(group) if dist.is_initialized() else 0
            for _, lo, hi, rlo, rhi in self.entries:
                cb = hi - lo
                if (rhi - rlo) != world * cb or lo != rlo + rank * cb:
                    raise ValueError("in-place all-gather contract violated: own rows [%d, %d) are not slot %d of the "
                                     "region [%d, %d) of %d equal slots" % (lo, hi, rank, rlo, rhi, world))

    def apply(self, X, out, Z=None, acc_in=None, acc_out=None, gather_acc=False, write_y=True, **scal):
        """Per entry: Y = out[o], optional Z[o] / acc_in[o] / acc_out[o] slices; the gathered tensor is `acc_out`
        when gather_acc else `out`.  Returns the list of pending collectives (call `.wait()` on each)."""
        works = []
        target = acc_out if gather_acc else out
        for k, (blk, lo, hi, rlo, rhi) in enumerate(self.entries):
            ep = dict(scal)
            if Z is not None:
                ep["Z"] = Z[lo:hi]
            if acc_out is not None:      # acc_in: a padded tensor (sliced like `out`) or one [rows, d] tensor per entry
                ep["acc_in"] = acc_in[k] if isinstance(acc_in, (list, tuple)) else acc_in[lo:hi]
                ep["acc_out"] = acc_out[lo:hi]
            y = out[lo:hi] if write_y else None
            if ep:
                self.local_spmm(blk, X, y, **ep)
            else:
                self.local_spmm(blk, X, y)
            if self.exchange:
                works.append(dist.all_gather_into_tensor(target[rlo:rhi], target[lo:hi], group=self.group, async_op=True))
                self.bytes_gathered += (rhi - rlo - (hi - lo)) * target.shape[1] * target.element_size()
        return works

    def run(self, X, out, **kw):
        for w in self.apply(X, out, **kw):
            w.wait()
        return out


class ShardedPropagator:
    """L-layer LightGCN propagation with the rows of the bipartite adjacency sharded over `group`.

    user_blocks / item_blocks: this rank's row chunks (BipartiteSharding.rank_blocks; a single block is accepted
    for n_chunks == 1).  `local_spmm(block, X, Y, **epilogue)` computes Y[:block.n_rows] = A_block @ X."""

    def __init__(self, sharding, user_blocks, item_blocks, rank, local_spmm, group=None, force_collectives=False):
        self.sh, self.rank, self.group = sharding, rank, group
        ub = list(user_blocks) if isinstance(user_blocks, (list, tuple)) else [user_blocks]
        ib = list(item_blocks) if isinstance(item_blocks, (list, tuple)) else [item_blocks]
        assert len(ub) == sharding.n_chunks and len(ib) == sharding.n_chunks
        self.exchange = sharding.P > 1 or force_collectives   # run the exchange even at world size 1 (tests)
        blocks = ub + ib
        self.op = RowShardedOp([(blk,) + e[2:] for blk, e in zip(blocks, sharding.entries(rank))], local_spmm, group,
                               self.exchange)
        self.entry_op = None     # set_entry_blocks: the same rows with NODE-order column ids (first layer of a model)

    def set_entry_blocks(self, user_blocks, item_blocks):
        """Blocks of the same rows whose column ids are node ids (BipartiteSharding.rank_blocks(node_cols=True)): the first
        layer then multiplies the replicated node-order parameter table directly, without permuting it into the padded
        space (sharded_lightgcn_mean(..., padded_out=True) uses them when present)."""
        blocks = list(user_blocks) + list(item_blocks)
        self.entry_op = RowShardedOp([(blk,) + e[2:] for blk, e in zip(blocks, self.sh.entries(self.rank))],
                                     self.op.local_spmm, self.group, self.exchange)

    def layer(self, X, X_next):
        """X_next = A @ X for the full (padded) id space; returns X_next."""
        return self.op.run(X, X_next)

    def propagate(self, X0, n_layers, bufs=None):
        """Runs n_layers layers; returns [X1..XL] (one buffer per layer unless `bufs` are given: with fewer than
        n_layers buffers earlier outputs are overwritten, which callers that only need the last layer may ask for)."""
        if bufs is None:
            bufs = [torch.empty_like(X0) for _ in range(n_layers)]
        cur, outs = X0, []
        for layer in range(n_layers):
            nxt = bufs[layer % len(bufs)]
            self.layer(cur, nxt)
            outs.append(nxt)
            cur = nxt
        return outs


class _ShardedLightGCNMean(torch.autograd.Function):
    """mean_l(A^l E0), l = 0..L, over the sharded rows: hip_ops._LightGCNMean with an all-gather after every layer.
    The layer sum rides in the SpMM epilogue on the rank's own rows; the LAST layer gathers the mean itself, so
    forward and backward each move L all-gathers.  E0 is the replicated [N, d] table in node order (every rank computes
    the same loss on the result, so the incoming gradient is replicated too); per-row arithmetic is the single-GPU
    kernel's: results are bit-identical to hip_ops.lightgcn_mean on the unsharded graph.
    padded_out=False: the result is in node order too (two permutations forward, two backward).
    padded_out=True : the result stays in the PADDED space ([N_pad, d]: consumers gather rows through the position map,
    `sharding.pos_tensor`); with the propagator's node-order entry blocks (`set_entry_blocks`) the first layer reads E0 as
    it is, so the forward needs no permutation at all and the backward one (of the final gradient)."""

    @staticmethod
    def forward(ctx, E0, prop, n_layers, padded_out):
        sh, L = prop.sh, int(n_layers)
        ctx.prop, ctx.L, ctx.padded = prop, L, bool(padded_out)
        E0 = E0.contiguous()
        if L == 0:
            return sh.pad(E0) if padded_out else E0.clone()
        direct = padded_out and prop.entry_op is not None
        X0 = E0 if direct else sh.pad(E0)
        first_acc = sh.own_row_slices(E0, prop.rank) if direct else X0
        acc = torch.zeros(sh.N_pad, E0.shape[1], dtype=E0.dtype, device=E0.device)
        bufs = [torch.empty_like(acc) if L > 1 else None, torch.empty_like(acc) if L > 2 else None]
        cur = X0
        for layer in range(1, L + 1):
            last = layer == L
            nxt = acc if last else bufs[(layer - 1) % 2]
            op = prop.entry_op if (direct and layer == 1) else prop.op
            op.run(cur, nxt, acc_in=first_acc if layer == 1 else acc, acc_out=acc, gather_acc=last, write_y=not last,
                   acc_scale=1.0 / (L + 1) if last else 1.0)
            cur = nxt
        return acc if padded_out else sh.unpad_nodes(acc)

    @staticmethod
    def backward(ctx, dOut):
        prop, L, sh = ctx.prop, ctx.L, ctx.prop.sh
        if L == 0:
            return (sh.unpad_nodes(dOut) if ctx.padded else dOut), None, None, None
        s = 1.0 / (L + 1)
        G = dOut.contiguous() if ctx.padded else sh.pad(dOut.contiguous())
        bufs = [torch.empty_like(G), torch.empty_like(G) if L > 1 else None]
        t = G
        for j in range(L):       # t <- s G + A^T t  (A symmetric; the first step also scales the inner term)
            out = bufs[j % 2]
            prop.op.run(t, out, Z=G, alpha=s if j == 0 else 1.0, beta=s)
            t = out
        return sh.unpad_nodes(t), None, None, None


def sharded_lightgcn_mean(prop: ShardedPropagator, E0, n_layers, padded_out=False):
    return _ShardedLightGCNMean.apply(E0, prop, n_layers, padded_out)


class ShardedSquareMatrix:
    """A square matrix over ONE row space (FREEDOM's frozen item-item graph over the item space) with its rows sharded
    like that space: forward = rows of A, backward = rows of A^T (the graph is directed), both all-gathered."""

    def __init__(self, space, fwd_blocks, bwd_blocks, rank, local_spmm, group=None, force_collectives=False):
        self.space = space
        ex = space.P > 1 or force_collectives
        ent = [(space.chunk_out(rank, c)[0] - space.base, space.chunk_out(rank, c)[1] - space.base,
                space.chunk_region(c)[0] - space.base, space.chunk_region(c)[1] - space.base)
               for c in range(space.n_chunks)]
        self.fwd = RowShardedOp([(b,) + e for b, e in zip(fwd_blocks, ent)], local_spmm, group, ex)
        self.bwd = RowShardedOp([(b,) + e for b, e in zip(bwd_blocks, ent)], local_spmm, group, ex)
        self.fwd_node = None     # set_entry_blocks: rows of A with NODE-order column ids (X given in node order)
        self._ent, self._ex = ent, ex
        self._pos_t = {}

    def set_entry_blocks(self, fwd_blocks_node_cols):
        self.fwd_node = RowShardedOp([(b,) + e for b, e in zip(fwd_blocks_node_cols, self._ent)], self.fwd.local_spmm,
                                     self.fwd.group, self._ex)

    def pos_tensor(self, device):
        key = str(device)
        if key not in self._pos_t:
            self._pos_t[key] = torch.from_numpy(self.space.pos - self.space.base).to(device)
        return self._pos_t[key]

    def pad(self, x):
        out = x.new_zeros(self.space.size, x.shape[1])
        out[self.pos_tensor(x.device)] = x
        return out

    def unpad(self, x):
        return x[self.pos_tensor(x.device)]


class _ShardedSpMM(torch.autograd.Function):
    """A @ X + Z with A's rows sharded (ShardedSquareMatrix); X, Z and the result replicated, node order."""

    @staticmethod
    def forward(ctx, X, Z, mat):
        ctx.mat, ctx.has_z = mat, Z is not None
        Xp = mat.pad(X.contiguous())
        out = torch.zeros_like(Xp)
        mat.fwd.run(Xp, out, **({"Z": mat.pad(Z.contiguous()), "beta": 1.0} if Z is not None else {}))
        return mat.unpad(out)

    @staticmethod
    def backward(ctx, dY):
        mat = ctx.mat
        dX = None
        if ctx.needs_input_grad[0]:
            Gp = mat.pad(dY.contiguous())
            out = torch.zeros_like(Gp)
            mat.bwd.run(Gp, out)
            dX = mat.unpad(out)
        return dX, (dY if ctx.has_z and ctx.needs_input_grad[1] else None), None


def sharded_spmm(mat: ShardedSquareMatrix, X, Z=None):
    return _ShardedSpMM.apply(X, Z, mat)


class _ShardedSpMMPadded(torch.autograd.Function):
    """A @ X + Zp with X in NODE order (read through the matrix's node-order-column entry blocks), Zp and the result in
    the PADDED space of the matrix's row space: no permutation forward, one (of dX) backward."""

    @staticmethod
    def forward(ctx, X, Zp, mat):
        ctx.mat = mat
        out = torch.empty(mat.space.size, X.shape[1], dtype=X.dtype, device=X.device)
        mat.fwd_node.run(X.contiguous(), out, Z=Zp.contiguous(), beta=1.0)
        return out

    @staticmethod
    def backward(ctx, dY):
        mat = ctx.mat
        dY = dY.contiguous()
        dX = None
        if ctx.needs_input_grad[0]:
            out = torch.empty_like(dY)
            mat.bwd.run(dY, out)
            dX = mat.unpad(out)
        return dX, (dY if ctx.needs_input_grad[1] else None), None


def sharded_spmm_padded(mat: ShardedSquareMatrix, X, Zp):
    return _ShardedSpMMPadded.apply(X, Zp, mat)


# ------------------------------------------------------------------------------------------------
# users sharded / items replicated (smaller exchange volume, not bit-exact)
# ------------------------------------------------------------------------------------------------
class ItemReplicatedPropagator:
    """Users sharded, items replicated ("1.5D"): the layout SURVEY.md 8(e) calls the smaller-volume
    variant, and the one the rest of the pipeline wants anyway (full-sort evaluation shards users and
    replicates the item matrix).

    Rank r keeps its users' rows R_r of the normalised interaction matrix (and R_r^T).  Per layer
        item partial  P_r = R_r^T  U_r          (items x local users)
        all-reduce(P_r) -> I'                   (128 MB at C5 instead of the 336 MB all-gather inbound)
        U_r' = R_r I                            (local users x items)  -- overlaps the all-reduce
    No user embedding ever crosses a link.  The item sums are combined by RCCL, so results equal the
    single-GPU ones to fp32 rounding (not bit for bit, unlike the all-gather layout).

    `local_spmm(block, X, Y)` as in ShardedPropagator."""

    def __init__(self, r_block, rt_block, local_spmm, group=None, world_size=None, force_collectives=False,
                 n_chunks=None):
        self.r_block, self.rt_block, self.local_spmm, self.group = r_block, rt_block, local_spmm, group
        self.P = world_size if world_size is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.force_collectives = force_collectives   # run the all-reduce even at world size 1 (tests)
        # The item partial sums are produced and all-reduced in `n_chunks` row blocks of R_r^T with equal
        # nnz: the all-reduce of chunk c runs on RCCL's stream while chunk c+1 is being computed, so
        # only the last chunk's exchange (plus whatever the user-side SpMM does not cover) is exposed.
        # (blocks are hip_ops.CsrGraph on the GPU, scipy CSR in the gloo tests)
        n_rows = rt_block.n_rows if hasattr(rt_block, "n_rows") else rt_block.shape[0]
        if n_chunks is None:   # a chunk should still be a >= ~0.1 ms SpMM (2.5M nnz), else launches dominate
            n_chunks = int(min(4, max(1, rt_block.nnz // 2_500_000)))
        self.chunks = [(0, n_rows, rt_block)]
        if (self.P > 1 or force_collectives) and n_chunks > 1 and n_rows >= 4 * n_chunks:
            rp = np.asarray(rt_block.rowptr_host if hasattr(rt_block, "rowptr_host") else rt_block.indptr,
                            dtype=np.int64)
            cuts = [0]
            for c in range(1, n_chunks):
                cuts.append(int(np.searchsorted(rp, rp[-1] * c // n_chunks, "left")))
            cuts.append(n_rows)
            if self.P > 1:   # every rank must cut at the same rows (same all-reduce sizes): rank 0 decides
                dev = rt_block.rowptr.device if hasattr(rt_block, "rowptr") else torch.device("cpu")
                ct = torch.tensor(cuts, dtype=torch.int64, device=dev)
                dist.broadcast(ct, src=0 if group is None else dist.get_global_rank(group, 0), group=group)
                cuts = [int(x) for x in ct.cpu().tolist()]
            cuts = sorted(set(min(max(x, 0), n_rows) for x in cuts))
            sub = rt_block.row_block if hasattr(rt_block, "row_block") else (lambda a, b: rt_block[a:b])
            self.chunks = [(a, b, sub(a, b)) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]

    def layer(self, u_local, items, u_next, items_next):
        exchange = self.P > 1 or self.force_collectives
        works = []
        for a, b, blk in self.chunks:
            part = items_next[a:b]                                    # contiguous row slice
            self.local_spmm(blk, u_local, part)                       # partial item sums from local users
            if exchange:
                works.append(dist.all_reduce(part, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self.local_spmm(self.r_block, items, u_next)                  # overlaps the last exchanges
        for w in works:
            w.wait()
        return u_next, items_next

    def propagate(self, u_local, items, n_layers):
        """-> [(U_1, I_1) .. (U_L, I_L)], one pair of buffers per layer (callers sum the layers)."""
        outs = []
        for _ in range(n_layers):
            u_local, items = self.layer(u_local, items, torch.empty_like(u_local), torch.empty_like(items))
            outs.append((u_local, items))
        return outs


class _IRLayerFn(torch.autograd.Function):
    """One propagation layer of the users-sharded / items-replicated layout WITH autograd, so that a
    model can train through it:  (U_r, I) -> (U_r' = R_r I,  I' = all_reduce_r(R_r^T U_r)).

    Backward, given this rank's gradients gU' (its own users) and gI' (its local contribution to the
    gradient of the replicated I'):  G = all_reduce(gI')  (every rank used its replica of I', so the
    gradient of the replicated tensor is the sum of the ranks' contributions), then
        dU_r = R_r G            dI (local contribution) = R_r^T gU'
    -- the same two local SpMMs and one all-reduce as the forward.  The gradient of a replicated LEAF
    (the item embedding table) is completed by the caller's usual gradient all-reduce."""

    @staticmethod
    def forward(ctx, u_local, items, prop):
        u_next, items_next = torch.empty_like(u_local), torch.empty_like(items)
        prop.layer(u_local.detach().contiguous(), items.detach().contiguous(), u_next, items_next)
        ctx.prop = prop
        return u_next, items_next

    @staticmethod
    def backward(ctx, gu, gi):
        prop = ctx.prop
        g_items = gi.contiguous().clone()
        if prop.P > 1 or prop.force_collectives:
            dist.all_reduce(g_items, op=dist.ReduceOp.SUM, group=prop.group)
        du = torch.empty_like(gu)
        prop.local_spmm(prop.r_block, g_items, du)                   # dU_r = R_r G
        di = torch.empty_like(gi)
        prop.local_spmm(prop.rt_block, gu.contiguous(), di)          # dI  = R_r^T gU'   (local part)
        return du, di, None


def item_replicated_layer(prop, u_local, items):
    """Differentiable ItemReplicatedPropagator.layer: returns (u_next_local, items_next)."""
    return _IRLayerFn.apply(u_local, items, prop)


class ShardedLightGCNStep:
    """LightGCN-style BPR training step over the users-sharded / items-replicated layout (the u-i
    propagation + sampled-scoring part every model in SURVEY.md 8a shares), one process per GPU:

      * this rank owns the embeddings of its users; the item table is replicated;
      * forward = `n_layers` differentiable sharded layers (`item_replicated_layer`), layer mean;
      * the batch is sharded by user ownership: a rank scores the triplets of its own users against
        its replica of the propagated item table with the fused BPR kernel; the loss is the sum of the
        ranks' partial sums divided by the global batch size;
      * backward runs the same layers in reverse (one item all-reduce per layer); the gradient of the
        replicated item table is completed by one more all-reduce; each rank then applies the same
        optimizer update to its replica (replicas stay bit-identical) and its own update to its users.

    `local_spmm(block, X, Y)` as in ItemReplicatedPropagator; `bpr_sum(U, I, users, pos, neg)` returns the
    SUM over the given triplets of -logsigmoid(<u,p> - <u,n>) (hip_ops.bpr_loss(..., reduction='sum'))."""

    def __init__(self, prop, user_emb_local, item_emb, n_layers, bpr_sum, lr=1e-3, group=None,
                 optimizer_cls=torch.optim.Adam):
        self.prop, self.n_layers, self.bpr_sum, self.group = prop, n_layers, bpr_sum, group
        self.user_emb = user_emb_local.detach().clone().requires_grad_()
        self.item_emb = item_emb.detach().clone().requires_grad_()
        self.opt = optimizer_cls([self.user_emb, self.item_emb], lr=lr)

    def forward(self):
        cu, ci = self.user_emb, self.item_emb
        su, si = cu, ci
        for _ in range(self.n_layers):
            cu, ci = item_replicated_layer(self.prop, cu, ci)
            su, si = su + cu, si + ci
        return su / (self.n_layers + 1), si / (self.n_layers + 1)

    def step(self, users_local, pos, neg, global_batch):
        """users_local: ids relative to this rank's user block; pos / neg: global item ids."""
        self.opt.zero_grad(set_to_none=True)
        ua, ia = self.forward()
        loss = self.bpr_sum(ua.contiguous(), ia.contiguous(), users_local, pos, neg) / global_batch
        loss.backward()
        multi = self.prop.P > 1 or self.prop.force_collectives
        if multi:
            dist.all_reduce(self.item_emb.grad, op=dist.ReduceOp.SUM, group=self.group)
        self.opt.step()
        total = loss.detach().clone()
        if multi:
            dist.all_reduce(total, op=dist.ReduceOp.SUM, group=self.group)
        return total


# ---- the other sharded pieces of SURVEY.md 8(e) ---------------------------------------------------
class _ShardedProjectionFn(torch.autograd.Function):
    """P3 over item shards: rank r owns the feature rows of its item block (`X_local` [ib, F], the last
    block zero padded) and projects only those; the [I_pad, out] result is all-gathered so that every
    rank holds the projected table (128 MB at 500K items instead of re-reading 8.2 GB of features).
    Backward: each rank keeps its own rows of dY (what the all-gather's transpose, a reduce-scatter of
    the ranks' dY contributions, leaves it with), computes its partial dW / db and dX_local; dW and db
    are summed over ranks (W and b are replicated parameters)."""

    @staticmethod
    def forward(ctx, X_local, W, b, local_linear, group, force):
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        y_local, vjp = local_linear(X_local, W, b)
        ctx.vjp, ctx.group, ctx.multi = vjp, group, (world > 1 or force)
        ctx.has_b = b is not None
        if not ctx.multi:
            return y_local
        out = torch.empty(world * y_local.shape[0], y_local.shape[1], dtype=y_local.dtype, device=y_local.device)
        dist.all_gather_into_tensor(out, y_local.contiguous(), group=group)
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        if ctx.multi:
            world = dist.get_world_size(ctx.group)
            g_local = torch.empty(g.shape[0] // world, g.shape[1], dtype=g.dtype, device=g.device)
            dist.reduce_scatter_tensor(g_local, g, op=dist.ReduceOp.SUM, group=ctx.group)
        else:
            g_local = g
        dX, dW, db = ctx.vjp(g_local)
        if ctx.multi:
            dist.all_reduce(dW, op=dist.ReduceOp.SUM, group=ctx.group)
            if db is not None:
                dist.all_reduce(db, op=dist.ReduceOp.SUM, group=ctx.group)
        return dX, dW, (db if ctx.has_b else None), None, None, None


def sharded_projection(X_local, W, b, local_linear, group=None, force_collectives=False):
    """`local_linear(X, W, b) -> (Y, vjp)` with `vjp(dY) -> (dX, dW, db)`: hip_ops-backed on the GPU
    (see `hip_local_linear`), a torch checker in the gloo test."""
    return _ShardedProjectionFn.apply(X_local, W, b, local_linear, group, force_collectives)


def hip_local_linear(X, W, b):
    """`local_linear` for sharded_projection on the HIP projection kernels."""
    from . import hip_ops
    Xd, Wd = X.detach().requires_grad_(X.requires_grad), W.detach().requires_grad_()
    bd = b.detach().requires_grad_() if b is not None else None
    with torch.enable_grad():
        Y = hip_ops.linear(Xd, Wd, bd)

    def vjp(dY):
        grads = torch.autograd.grad(Y, [t for t in (Xd, Wd, bd) if t is not None and t.requires_grad], dY,
                                    allow_unused=True)
        it = iter(grads)
        dX = next(it) if Xd.requires_grad else None
        dW = next(it)
        db = next(it) if bd is not None else None
        return dX, dW, db
    return Y.detach(), vjp


class _OwnedRowsExchange(torch.autograd.Function):
    """rows[b] of a [B, d] matrix are produced by the rank that OWNS row b's source (a batch item's projected
    feature row is computed where the item's feature row lives); every rank needs all B rows (the loss is
    replicated).  Forward: the ranks' disjoint contributions (zeros elsewhere) are summed by an all-reduce, which is
    exact (x + 0) and, at B = 4096 rows x 64 floats = 1 MB, latency sized -- the degenerate all-to-all of 8 x 128 KB
    pieces.  Backward: the replicated gradient is simply masked to the owned rows (every rank already holds it)."""

    @staticmethod
    def forward(ctx, rows_local, owned, group, multi):
        ctx.owned = owned
        out = rows_local * owned.unsqueeze(1).to(rows_local.dtype)
        if multi:
            dist.all_reduce(out, op=dist.ReduceOp.SUM, group=group)
        return out

    @staticmethod
    def backward(ctx, g):
        return g * ctx.owned.unsqueeze(1).to(g.dtype), None, None, None


def exchange_owned_rows(rows_local, owned_mask, group=None, multi=True):
    return _OwnedRowsExchange.apply(rows_local, owned_mask, group, multi)


def sharded_score_topk(Q_local, C, k, score_topk, mask_rowptr=None, mask_col=None, group=None, gather=True):
    """P5 / P6 over query shards: every rank scores its own block of queries (eval users, or kNN query
    items) against its replica of the candidates -- no exchange in the data path; `gather=True`
    all-gathers the [rows, k] id blocks (equal block sizes) for callers that want the whole table."""
    idx = score_topk(Q_local, C, k, mask_rowptr, mask_col)
    if not gather or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return idx
    world = dist.get_world_size(group)
    out = torch.empty(world * idx.shape[0], idx.shape[1], dtype=idx.dtype, device=idx.device)
    dist.all_gather_into_tensor(out, idx.contiguous(), group=group)
    return out
