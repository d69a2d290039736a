"""FREEDOM on the HIP hot path (reference: models/freedom.py).

init   : normalised u-i adjacency (host, fp64->fp32), per-edge pruning weights (device), frozen
         kNN item-item graph -- loaded from the reference's cache file `mm_adj_freedomdsp_{k}_{10w}.pt`
         when present, otherwise built with the fused score+top-K kernel and saved in that format
epoch  : degree-sensitive edge dropout: multinomial draw, re-normalise, rebuild CSR on the device
forward: fused LightGCN layer mean over the (masked) u-i graph + item-item SpMM with the residual
         add fused into its epilogue
loss   : three fused BPR terms (id / text projection / image projection); projections are fp32
         MFMA GEMMs over all items with dW, db and dX (the feature tables are trainable)
eval   : fused score + mask + top-K
"""
import os

import numpy as np
import torch
import torch.nn as nn

from mmrec_amd import hip_ops
from mmrec_amd.common.lazy_rows import LazyRowEmbedding, lazy_adam_enabled
from mmrec_amd.graph import (knn_normalized_coo, mask_to_csr_device, norm_adj_graph, relabel_graph,
                             sparse_coo_to_graph)
from mmrec_amd.models._base import AdjacentTablesMixin, FusedEvalMixin, GeneralRecommender, RelabelledIdsMixin


def build_mm_adj(v_feat, t_feat, knn_k, mm_image_weight, n_items):
    """w * kNN(image) + (1 - w) * kNN(text) as an UNCOALESCED torch sparse COO tensor, exactly the object the
    reference caches in `mm_adj_freedomdsp_{k}_{10w}.pt` (freedom.py:55-77, pgl.py:52-74)."""
    size, parts = (n_items, n_items), []
    if v_feat is not None:
        parts.append((mm_image_weight if t_feat is not None else 1.0, knn_normalized_coo(v_feat, knn_k)))
    if t_feat is not None:
        parts.append((1.0 - mm_image_weight if v_feat is not None else 1.0, knn_normalized_coo(t_feat, knn_k)))
    idx = torch.cat([p[0] for _, p in parts], dim=1)
    val = torch.cat([w * p[1] for w, p in parts])
    return torch.sparse_coo_tensor(idx, val, size)   # uncoalesced sum, like w*A_img + (1-w)*A_txt


def load_or_build_mm_coo(config, v_feat, t_feat, knn_k, mm_image_weight, n_items, cache_name=None, write=True):
    """the frozen item-item graph as the torch sparse COO tensor the reference caches (loaded when the file exists)"""
    cache = os.path.join(os.path.abspath(config['data_path'] + config['dataset']),
                         cache_name or 'mm_adj_freedomdsp_{}_{}.pt'.format(knn_k, int(10 * mm_image_weight)))
    if os.path.exists(cache):
        return torch.load(cache, weights_only=False)
    mm = build_mm_adj(v_feat, t_feat, knn_k, mm_image_weight, n_items)
    if write:
        # written under a temporary name and renamed into place: the ranks of a multi-process run all come through here
        # (every rank builds the same graph, rank 0 writes), and a rank that finds the file must never read a half-written one
        tmp = '%s.tmp.%d' % (cache, os.getpid())
        torch.save(mm.cpu(), tmp)
        os.replace(tmp, cache)
    return mm


def load_or_build_mm_adj(config, v_feat, t_feat, knn_k, mm_image_weight, n_items, device, cache_name=None):
    mm = load_or_build_mm_coo(config, v_feat, t_feat, knn_k, mm_image_weight, n_items, cache_name)
    g = sparse_coo_to_graph(mm, device)
    g.transpose()   # directed kNN graph: the backward needs A^T, built once (graph is frozen)
    return g


class FREEDOM(RelabelledIdsMixin, AdjacentTablesMixin, FusedEvalMixin, GeneralRecommender):
    graph_capturable = True       # the step is a fixed launch sequence: replayed as a hipGraph by default (hip_graph_step: auto)

    adjacent_tables = ('user_embedding.weight', 'item_id_embedding.weight')
    relabelled_tables = {'user_embedding.weight': 'u', 'item_id_embedding.weight': 'i', 'image_embedding.weight': 'i',
                         'text_embedding.weight': 'i'}

    def __init__(self, config, dataset):
        super().__init__(config, dataset)
        self.embedding_dim = config['embedding_size']
        self.feat_embed_dim = config['feat_embed_dim']
        self.knn_k = config['knn_k']
        self.lambda_coeff = config['lambda_coeff']
        self.cf_model = config['cf_model']
        self.n_layers = config['n_mm_layers']
        self.n_ui_layers = config['n_ui_layers']
        self.reg_weight = config['reg_weight']
        self.build_item_graph = True
        self.mm_image_weight = config['mm_image_weight']
        self.dropout = config['dropout']
        self.degree_ratio = config['degree_ratio']
        lazy = config['lazy_projection']
        self.lazy_projection = True if lazy is None else bool(lazy)   # new key; False = project all items
        # new key: row-lazy exact Adam on the trainable feature tables (common/lazy_rows.py) -- with the
        # gathered-rows projection a step then reads / writes only the <= 2B feature rows of its batch
        n_feat = max([0] + [int(f.numel()) for f in (self.v_feat, self.t_feat) if f is not None])
        self.lazy_feature_adam = lazy_adam_enabled(config, n_feat) and self.lazy_projection
        self.lazy_prefetch = config['lazy_prefetch'] is not False    # new key: catch-up on a side stream (default on)
        self.pull_batch_rows = _batch_rows_wanted(config, self.n_users + self.n_items)   # new key `hip_pull_batch_rows`
        self.n_nodes = self.n_users + self.n_items

        self.interaction_matrix = dataset.inter_matrix(form='coo').astype(np.float32)
        self.norm_adj = norm_adj_graph(self.interaction_matrix, self.n_users, self.n_items, self.device)
        # new key `reorder` (community | degree | rcm): every id-indexed table lives in an id space relabelled once, here, for
        # gather locality (models/_base.py: RelabelledIdsMixin); the graphs are relabelled with their rows' nonzero order kept
        rl = self._setup_relabelling(config, self.norm_adj)
        if rl is not None:
            self.norm_adj = relabel_graph(self.norm_adj, rl.node_perm_host())
        self.masked_adj, self.mm_adj = None, None
        rows = torch.from_numpy(self.interaction_matrix.row.astype(np.int64))
        cols = torch.from_numpy(self.interaction_matrix.col.astype(np.int64))
        self.edge_indices = torch.stack([rows, cols]).to(self.device)
        # (degrees, hence the values, do not depend on the labels; the EDGE ORDER -- what torch.multinomial draws from in
        # pre_epoch_processing -- is the dataset's either way)
        self.edge_values = hip_ops.edge_norm_values(self.edge_indices[0].contiguous(),
                                                    self.edge_indices[1].contiguous(),
                                                    self.n_users, self.n_items)
        if rl is not None:
            self.edge_indices = torch.stack([rl.perm_u[self.edge_indices[0]], rl.perm_i[self.edge_indices[1]]])

        self.user_embedding = nn.Embedding(self.n_users, self.embedding_dim)
        self.item_id_embedding = nn.Embedding(self.n_items, self.embedding_dim)
        nn.init.xavier_uniform_(self.user_embedding.weight)
        nn.init.xavier_uniform_(self.item_id_embedding.weight)
        if rl is not None:            # the plain model's initial values, row `old` at relabelled row perm[old]
            with torch.no_grad():
                self.user_embedding.weight.copy_(self.user_embedding.weight[rl.inv_u.cpu()])
                self.item_id_embedding.weight.copy_(self.item_id_embedding.weight[rl.inv_i.cpu()])
        table = LazyRowEmbedding if self.lazy_feature_adam else nn.Embedding
        in_space = (lambda f: f) if rl is None else (lambda f: f.index_select(0, rl.inv_i.to(f.device)))
        if self.v_feat is not None:
            self.image_embedding = table.from_pretrained(in_space(self.v_feat), freeze=False)
            self.image_trs = nn.Linear(self.v_feat.shape[1], self.feat_embed_dim)
        if self.t_feat is not None:
            self.text_embedding = table.from_pretrained(in_space(self.t_feat), freeze=False)
            self.text_trs = nn.Linear(self.t_feat.shape[1], self.feat_embed_dim)

        # (built / loaded in the dataset's ids: the cache file `mm_adj_freedomdsp_*.pt` is the reference's, and kNN ties are
        # broken by id)
        self.mm_adj = load_or_build_mm_adj(config, self.v_feat, self.t_feat, self.knn_k, self.mm_image_weight,
                                           self.n_items, self.device)
        if rl is not None:
            self.mm_adj = relabel_graph(self.mm_adj, rl.perm_i_host)

    def _build_mm_adj(self):
        return build_mm_adj(self.v_feat, self.t_feat, self.knn_k, self.mm_image_weight, self.n_items)

    def pre_epoch_processing(self):
        if self.dropout <= .0:
            self.masked_adj = self.norm_adj
            return
        keep_len = int(self.edge_values.size(0) * (1. - self.dropout))
        self.set_kept_edges(torch.multinomial(self.edge_values, keep_len))

    def set_kept_edges(self, keep_idx):
        """Rebuild the pruned, re-normalised graph from sampled edge ids (injectable for parity runs)."""
        kept = self.edge_indices[:, keep_idx]
        self.masked_adj = hip_ops.bipartite_graph_from_edges(kept[0].contiguous(), kept[1].contiguous(),
                                                             self.n_users, self.n_items)

    def forward(self, adj):
        # cat -> propagate -> split (freedom.py:165-178) in one autograd node: no zero-padded slice gradients
        u_g, i_g = hip_ops.lightgcn_mean_parts(adj, (self.user_embedding.weight, self.item_id_embedding.weight), self.n_ui_layers)
        h = self.item_id_embedding.weight
        if self.n_layers == 0:
            return u_g, i_g + h
        for _ in range(self.n_layers - 1):
            h = hip_ops.spmm(self.mm_adj, h)
        return u_g, hip_ops.spmm(self.mm_adj, h, Z=i_g)     # i_g + M h in one launch

    def eval_embeddings(self):
        return self.forward(self.norm_adj)

    def calculate_loss(self, interaction):
        interaction = self._map_batch(interaction)
        users, pos_items, neg_items = interaction[0], interaction[1], interaction[2]
        rows = None
        if self.lazy_projection:
            rows = torch.cat((pos_items, neg_items))
            if self.lazy_feature_adam and self.lazy_prefetch:      # the row catch-up streams through HBM under the propagation
                for emb in (getattr(self, 'text_embedding', None), getattr(self, 'image_embedding', None)):
                    if emb is not None:
                        emb.prefetch(rows)
        if self.lazy_projection and self.pull_batch_rows and self.n_layers == 1 and not hip_ops.DETERMINISTIC:
            return self._loss_at_batch_rows(users, pos_items, neg_items, rows)
        ua, ia = self.forward(self.masked_adj)
        self.build_item_graph = False
        if self.lazy_projection:
            # The reference projects ALL items every batch (freedom.py:205,208) but only the pos/neg rows
            # are ever consumed (:206,:209).  A projection row depends on its own
            # feature row only, so projecting the <= 2B gathered rows is the same function and the same
            # gradient (dW, db, and dX scattered back into the table): 2.1 GFLOP regardless of n_items
            # instead of 3.7 (Baby) / 9.6 (Sports) / 262 (500K items).
            terms = self._batch_terms(ia, pos_items, neg_items, rows)
            # bpr(id) + reg_weight (bpr(text) + bpr(image)) (freedom.py:211): all terms in one launch pair (ABI 14)
            return hip_ops.bpr_weighted_total(ua, users, terms, [1.0] + [self.reg_weight] * (len(terms) - 1), joint_grad=True)
        loss = hip_ops.bpr_loss(ua, ia, users, pos_items, neg_items)
        mf_t = mf_v = 0.0
        if self.t_feat is not None:
            text_feats = hip_ops.linear(self.text_embedding.weight, self.text_trs.weight, self.text_trs.bias)
            mf_t = hip_ops.bpr_loss(ua, text_feats, users, pos_items, neg_items)
        if self.v_feat is not None:
            image_feats = hip_ops.linear(self.image_embedding.weight, self.image_trs.weight, self.image_trs.bias)
            mf_v = hip_ops.bpr_loss(ua, image_feats, users, pos_items, neg_items)
        return loss + self.reg_weight * (mf_t + mf_v)

    def _batch_terms(self, first_table, first_pos, first_neg, rows):
        """the three BPR terms: (id table or rows, pos ids, neg ids) + the projections of the batch's <= 2B feature rows"""
        b = first_pos.shape[0]
        lp = torch.arange(b, device=rows.device)
        ln = lp + b
        gather = (lambda emb: emb.rows(rows)) if self.lazy_feature_adam else (lambda emb: emb.weight[rows])
        terms = [(first_table, first_pos, first_neg)]
        if self.t_feat is not None:
            terms.append((hip_ops.linear(gather(self.text_embedding), self.text_trs.weight, self.text_trs.bias), lp, ln))
        if self.v_feat is not None:
            terms.append((hip_ops.linear(gather(self.image_embedding), self.image_trs.weight, self.image_trs.bias), lp, ln))
        return terms

    def _loss_at_batch_rows(self, users, pos_items, neg_items, rows):
        """New key `hip_pull_batch_rows` (default on; n_mm_layers = 1): the loss reads the propagated tables at the batch's
        rows only (freedom.py:197-199), so
          * the item-item layer (freedom.py:173-177) is computed at the 2B pos / neg rows instead of all items
            (hip_ops.spmm_rows: the full launch's bits; 4096 of 500,000 rows at config 5) and its gradient is pushed through
            those rows -- no launch over all items in either direction;
          * the user-item layer mean is gathered at the 3B batch rows and its backward starts from their compact gradient
            (hip_ops.lightgcn_mean_parts_rows): the dense [N, 64] gradient of the mean and the first of the backward's
            launches over all rows go away.
        Forward values are bit-identical to the full computation; the backward's pushes use fp32 atomics."""
        b = users.shape[0]
        # (one autograd node: the item-item layer's backward pushes into the propagation's gradient of the item table instead
        #  of a zero-filled [n_items, 64] buffer of its own that autograd would add -- 128 MB + 384 MB of traffic at config 5)
        ua_rows, ia_rows = hip_ops.lightgcn_mean_rows_then_item_rows(self.masked_adj, self.user_embedding.weight,
                                                                     self.item_id_embedding.weight, self.n_ui_layers, users, rows,
                                                                     self.mm_adj)
        self.build_item_graph = False
        ar = torch.arange(b, device=rows.device)
        terms = self._batch_terms(ia_rows, ar, ar + b, rows)
        return hip_ops.bpr_weighted_total(ua_rows, ar, terms, [1.0] + [self.reg_weight] * (len(terms) - 1))


def _batch_rows_wanted(config, n_nodes):
    """`hip_pull_batch_rows`: True / False, or absent / 'auto' = from 2^18 nodes on.  The batch-rows step trades launches over
    all rows for a handful of small ones: config 5 (1.5 M nodes) 3.27 -> 2.49 ms per step, but Amazon-Sports shape 0.77 -> 0.82
    and Amazon-Baby shape 0.89 -> 1.05 ms (cache-resident graphs whose full launches take ~10 us:
    profiles/r04_batch_rows_small_ab.log)."""
    v = config['hip_pull_batch_rows']
    if v is None or str(v).lower() == 'auto':
        return n_nodes >= (1 << 18)
    return v is True or str(v).lower() in ('true', '1', 'yes', 'on')


def _combine(losses, has_text, reg_weight):
    """bpr + reg_weight * (mf_t + mf_v), in the reference's order of operations (freedom.py:211)"""
    mf = [0.0, 0.0]
    rest = list(losses[1:])
    if has_text:
        mf[0] = rest.pop(0)
    if rest:
        mf[1] = rest.pop(0)
    return losses[0] + reg_weight * (mf[0] + mf[1])


# ------------------------------------------------------------------------------------------------------------------
# n_gpus > 1: the same model, one process per GPU (BASELINE config 5)
# ------------------------------------------------------------------------------------------------------------------
class _SumGradOverRanks(torch.autograd.Function):
    """identity on a REPLICATED parameter whose gradient is produced in rank-local pieces (the projection weights: every
    rank projects only the batch rows whose feature rows it owns): backward sums the pieces over the ranks, so that every
    replica applies the same update."""

    @staticmethod
    def forward(ctx, p, group):
        ctx.group = group
        return p.view_as(p)

    @staticmethod
    def backward(ctx, g):
        import torch.distributed as tdist
        g = g.contiguous().clone()
        if tdist.is_initialized() and tdist.get_world_size(ctx.group) > 1:
            tdist.all_reduce(g, op=tdist.ReduceOp.SUM, group=ctx.group)
        return g, None


def _local_spmm(blk, X, Y, **ep):
    return hip_ops.spmm_raw(blk, X, Y=Y, **ep)


class RowShardedFREEDOM(FusedEvalMixin, GeneralRecommender):
    """FREEDOM over `n_gpus` processes, ROWS of the graphs sharded (config `dist_layout: rows`) (torch.distributed initialised by the launcher; utils/quick_start.py does it from
    the torchrun environment).  What is sharded and what is replicated:

      * user-item graph (and its per-epoch pruned version): ROWS sharded, nnz-balanced (dist.BipartiteSharding); every
        propagation layer = local HIP SpMM on the rank's row chunks + chunked RCCL all-gather (dist.sharded_lightgcn_mean)
        -- forward and backward, bit-identical to the single-GPU kernel;
      * frozen item-item graph: rows sharded like the items (dist.sharded_spmm);
      * id embedding tables: replicated parameters with replicated gradients (the loss is computed on replicated
        tables from the same batch on every rank), identical fused-Adam updates: replicas stay bit-identical;
      * raw feature tables (the 8.2 GB image table of config 5) and their Adam state: SHARDED by item block; a batch's
        pos / neg rows are projected by their owners and exchanged as [2B, 64] rows (dist.exchange_owned_rows); the
        projection weights are replicated, their partial gradients summed over ranks;
      * evaluation: the propagation once (sharded), then every rank ranks its slice of the users against its replica of
        the item table; the [users, k] ids are all-gathered (dist.sharded_score_topk).
    Loaders, seeds and the negative sampler run identically on every rank (same batches everywhere).
    Parameter names are FREEDOM's; `image_embedding.weight` / `text_embedding.weight` hold the rank's rows only
    (`gather_feature_tables()` rebuilds the full tables)."""

    def __init__(self, config, dataset):
        super().__init__(config, dataset)
        import torch.distributed as tdist
        from mmrec_amd.dist import BipartiteSharding, ShardedPropagator, ShardedSquareMatrix, space_blocks
        from mmrec_amd.graph import sym_norm_coo, unique_edges
        if not tdist.is_initialized():
            raise RuntimeError('RowShardedFREEDOM needs torch.distributed (launch with torchrun; config n_gpus)')
        self.group = None
        self.rank, self.world = tdist.get_rank(), tdist.get_world_size()
        self.force = bool(config['dist_force_collectives'])     # testing aid: run the collectives at world size 1 too
        self.embedding_dim = config['embedding_size']
        self.feat_embed_dim = config['feat_embed_dim']
        self.knn_k = config['knn_k']
        self.n_layers = config['n_mm_layers']
        self.n_ui_layers = config['n_ui_layers']
        self.reg_weight = config['reg_weight']
        self.mm_image_weight = config['mm_image_weight']
        self.dropout = config['dropout']
        n_feat = max([0] + [int(f.numel()) for f in (self.v_feat, self.t_feat) if f is not None])
        self.lazy_feature_adam = lazy_adam_enabled(config, n_feat)
        self.graph_capturable = False
        nu, ni = self.n_users, self.n_items
        self.n_nodes = nu + ni

        self.interaction_matrix = dataset.inter_matrix(form='coo').astype(np.float32)
        eu, ei = unique_edges(self.interaction_matrix.row, self.interaction_matrix.col, ni)
        r, c, v = sym_norm_coo(eu, ei, nu, ni)
        chunks = config['dist_chunks']
        if not chunks:      # a chunk should stay a >= ~2.5M-nnz SpMM, else launches dominate; one rank has nothing to overlap
            chunks = 1 if self.world == 1 else int(min(4, max(1, r.shape[0] // self.world // 2_500_000)))
        self.sharding = sh = BipartiteSharding.from_coo(r, nu, ni, self.world, n_chunks=chunks)
        self.nnz_per_rank = sh.nnz_per_rank(r)
        # ONE long-row plan for every block of the u-i graph, the single-GPU graph's (a function of ITS column count): padded
        # blocks have N_pad columns and node-order entry blocks n_nodes -- with the per-block default the first and the
        # later layers could cut rows differently and sum in another order than the unsharded kernel
        self.ui_threshold = hip_ops.default_long_row_threshold(nu + ni)
        make = lambda lr, pc, vals, n_rows, n_cols: hip_ops.CsrGraph.from_coo_host(  # noqa: E731
            np.stack([lr, pc]), vals, n_rows, n_cols, self.device, long_row_threshold=self.ui_threshold)
        ub, ib = sh.rank_blocks(r, c, v, self.rank, make)
        self.norm_prop = ShardedPropagator(sh, ub, ib, self.rank, _local_spmm, group=self.group, force_collectives=self.force)
        # the same rows with node-order column ids: the first layer reads the replicated parameter table as it is, and
        # everything downstream of the propagation lives in the padded space (batch rows are gathered through pos_u / pos_i)
        self.norm_prop.set_entry_blocks(*sh.rank_blocks(r, c, v, self.rank, make, node_cols=True))
        self.pos_u = torch.from_numpy(sh.users.pos).to(self.device)
        self.pos_i = torch.from_numpy(sh.items.pos - sh.U_pad).to(self.device)
        self.masked_prop = None
        rows = torch.from_numpy(self.interaction_matrix.row.astype(np.int64))
        cols = torch.from_numpy(self.interaction_matrix.col.astype(np.int64))
        self.edge_indices = torch.stack([rows, cols]).to(self.device)
        self.edge_values = hip_ops.edge_norm_values(self.edge_indices[0].contiguous(), self.edge_indices[1].contiguous(),
                                                    nu, ni)

        # parameters in FREEDOM's construction order (same generator consumption -> same initial values)
        self.user_embedding = nn.Embedding(nu, self.embedding_dim)
        self.item_id_embedding = nn.Embedding(ni, self.embedding_dim)
        nn.init.xavier_uniform_(self.user_embedding.weight)
        nn.init.xavier_uniform_(self.item_id_embedding.weight)
        self.item_lo, self.item_hi = sh.items.block(self.rank)
        table = LazyRowEmbedding if self.lazy_feature_adam else nn.Embedding

        def local_rows(feat):
            rows_ = feat[self.item_lo:self.item_hi]
            return rows_.clone() if rows_.shape[0] else feat.new_zeros(1, feat.shape[1])   # an empty block owns nothing
        if self.v_feat is not None:
            self.image_embedding = table.from_pretrained(local_rows(self.v_feat), freeze=False)
            self.image_trs = nn.Linear(self.v_feat.shape[1], self.feat_embed_dim)
        if self.t_feat is not None:
            self.text_embedding = table.from_pretrained(local_rows(self.t_feat), freeze=False)
            self.text_trs = nn.Linear(self.t_feat.shape[1], self.feat_embed_dim)

        mm = load_or_build_mm_coo(config, self.v_feat, self.t_feat, self.knn_k, self.mm_image_weight, ni,
                                  write=self.rank == 0)
        idx, val = mm._indices().cpu().numpy(), mm._values().cpu().numpy().astype(np.float32)
        order = np.argsort(idx[0], kind='stable')            # the single-GPU CSR order (transpose() starts from it)
        idx, val = idx[:, order], val[order]
        ipos = sh.items.pos - sh.items.base
        mm_threshold = hip_ops.default_long_row_threshold(ni)       # the single-GPU item-item graph has n_items columns
        make_i = lambda lr, pc, vals, n_rows, n_cols: hip_ops.CsrGraph.from_coo_host(  # noqa: E731
            np.stack([lr, pc]), vals, n_rows, n_cols, self.device, long_row_threshold=mm_threshold)
        fwd = space_blocks(sh.items, idx[0], idx[1], val, self.rank, ipos, sh.items.size, make_i)
        bwd = space_blocks(sh.items, idx[1], idx[0], val, self.rank, ipos, sh.items.size, make_i)
        self.mm_adj = ShardedSquareMatrix(sh.items, fwd, bwd, self.rank, _local_spmm, group=self.group,
                                          force_collectives=self.force)
        self.mm_adj.set_entry_blocks(space_blocks(sh.items, idx[0], idx[1], val, self.rank, None, ni, make_i))
        self.v_feat_dim = None if self.v_feat is None else self.v_feat.shape[1]
        # the full tables are only needed up to here (kNN graph + this rank's rows)
        if config['dist_keep_full_features'] is not True:
            self.v_feat = None if self.v_feat is None else self.v_feat[:0]
            self.t_feat = None if self.t_feat is None else self.t_feat[:0]

    # ---- per-epoch pruned graph: rank 0 draws, everybody builds its own rows
    def pre_epoch_processing(self):
        import torch.distributed as tdist
        if self.dropout <= .0:
            self.masked_prop = self.norm_prop
            return
        keep_len = int(self.edge_values.size(0) * (1. - self.dropout))
        keep = torch.multinomial(self.edge_values, keep_len)     # every rank draws (generators stay aligned) ...
        if self.world > 1:                                        # ... and rank 0's draw is the one everybody uses
            tdist.broadcast(keep, src=0, group=self.group)
        self.set_kept_edges(keep)

    def set_kept_edges(self, keep_idx):
        from mmrec_amd.dist import ShardedPropagator
        sh, nu = self.sharding, self.n_users
        kept = self.edge_indices[:, keep_idx]
        eu, ei = kept[0].contiguous(), kept[1].contiguous()
        w = hip_ops.edge_norm_values(eu, ei, nu, self.n_items)
        rows, cols, vals = torch.cat((eu, ei + nu)), torch.cat((ei + nu, eu)), torch.cat((w, w))   # freedom.py:139-143
        pos = sh.pos_tensor(rows.device)
        pr, pc = pos[rows], pos[cols]
        blocks, entry = [], []
        for _, _, lo, hi, _, _ in sh.entries(self.rank):
            sel = (pr >= lo) & (pr < hi)
            lr, vv = (pr[sel] - lo).to(torch.int32).contiguous(), vals[sel].contiguous()
            blocks.append(hip_ops.CsrGraph.from_coo_device(lr, pc[sel].to(torch.int32).contiguous(), vv, hi - lo, sh.N_pad,
                                                           long_row_threshold=self.ui_threshold))
            entry.append(hip_ops.CsrGraph.from_coo_device(lr, cols[sel].to(torch.int32).contiguous(), vv, hi - lo,
                                                          self.n_nodes, long_row_threshold=self.ui_threshold))
        nc = sh.n_chunks
        self.masked_prop = ShardedPropagator(sh, blocks[:nc], blocks[nc:], self.rank, _local_spmm, group=self.group,
                                             force_collectives=self.force)
        self.masked_prop.set_entry_blocks(entry[:nc], entry[nc:])

    def forward_padded(self, prop):
        """-> (user table [U_pad, d], item table [I_pad, d]) in the PADDED id space: row of user u = pos_u[u], of item i =
        pos_i[i].  No permutation of the [N, d] tables on the way (the first layers read the node-order parameters through
        node-order-column blocks); the backward permutes the two final gradients only."""
        from mmrec_amd.dist import sharded_lightgcn_mean, sharded_spmm, sharded_spmm_padded
        sh = self.sharding
        ego = torch.cat((self.user_embedding.weight, self.item_id_embedding.weight), dim=0)
        mean = sharded_lightgcn_mean(prop, ego, self.n_ui_layers, padded_out=True)
        u_p, i_gp = mean[:sh.U_pad], mean[sh.U_pad:]
        h = self.item_id_embedding.weight
        if self.n_layers == 0:
            return u_p, i_gp + self.mm_adj.pad(h)
        for _ in range(self.n_layers - 1):
            h = sharded_spmm(self.mm_adj, h)
        return u_p, sharded_spmm_padded(self.mm_adj, h, i_gp)

    def forward(self, prop):
        """node-order (user_all [n_users, d], item_all [n_items, d]), like FREEDOM.forward"""
        u_p, i_p = self.forward_padded(prop)
        return u_p[self.pos_u], i_p[self.pos_i]

    def eval_embeddings(self):
        return self.forward(self.norm_prop)

    def _owned_projection(self, emb, trs, rows):
        """[2B, 64] projected feature rows of the batch items, each computed by the rank that owns the item"""
        from mmrec_amd.dist import exchange_owned_rows
        owned = (rows >= self.item_lo) & (rows < self.item_hi)
        if self.lazy_feature_adam:        # slots of other ranks' items: -1 = "no row" (no catch-up, no gradient, no step)
            emb.allow_missing = True
            feats = emb.rows(torch.where(owned, rows - self.item_lo, torch.full_like(rows, -1)))
        else:
            feats = emb.weight[torch.where(owned, rows - self.item_lo, torch.zeros_like(rows))]
        w, b = _SumGradOverRanks.apply(trs.weight, self.group), _SumGradOverRanks.apply(trs.bias, self.group)
        return exchange_owned_rows(hip_ops.linear(feats, w, b), owned, group=self.group, multi=self.world > 1 or self.force)

    def calculate_loss(self, interaction):
        users, pos_items, neg_items = interaction[0], interaction[1], interaction[2]
        ua, ia = self.forward_padded(self.masked_prop)            # padded tables: batch rows through the position maps
        ua, ia = ua.contiguous(), ia.contiguous()
        users = self.pos_u[users]
        rows = torch.cat((pos_items, neg_items))
        b = pos_items.shape[0]
        lp = torch.arange(b, device=rows.device)
        ln = lp + b
        terms = [(ia, self.pos_i[pos_items], self.pos_i[neg_items])]
        if self.t_feat is not None:
            terms.append((self._owned_projection(self.text_embedding, self.text_trs, rows), lp, ln))
        if self.v_feat is not None:
            terms.append((self._owned_projection(self.image_embedding, self.image_trs, rows), lp, ln))
        return _combine(hip_ops.bpr_losses_shared_users(ua, users, terms, joint_grad=True), self.t_feat is not None, self.reg_weight)

    @torch.no_grad()
    def full_sort_topk(self, interaction, k):
        """every rank ranks its slice of the batch's users; ids all-gathered (no exchange in the scoring itself)"""
        import torch.distributed as tdist
        users, mask = interaction[0], interaction[1]
        u, cands = self._cached_eval_candidates()          # the replicated item table, prepared once per evaluation
        i = cands.C if isinstance(cands, hip_ops.TopkCandidates) else cands
        rl = self.relabelling                              # (SlicedFREEDOM under `reorder`; None otherwise)
        if rl is not None:
            users = rl.perm_u[users]
            if mask.shape[1]:
                mask = torch.stack([mask[0], rl.perm_i[mask[1]]])
        rowptr, cols = mask_to_csr_device(mask, users.shape[0], i.shape[0])
        b = users.shape[0]
        per = -(-b // self.world)
        lo, hi = min(self.rank * per, b), min((self.rank + 1) * per, b)
        out = torch.zeros(per, k, dtype=torch.int64, device=users.device)
        if hi > lo:
            rp = rowptr[lo:hi + 1]
            s, e = int(rp[0]), int(rp[-1])
            out[:hi - lo] = hip_ops.score_topk(u[users[lo:hi]].contiguous(), cands, k, (rp - rp[0]).contiguous(),
                                               cols[s:max(e, s + 1)].contiguous() if e > s else None)
        if self.world > 1:
            full = torch.empty(self.world * per, k, dtype=torch.int64, device=users.device)
            tdist.all_gather_into_tensor(full, out, group=self.group)
            out = full
        return out[:b] if rl is None else rl.inv_i[out[:b]]      # ranked in the tables' id space, reported in the dataset's

    @torch.no_grad()
    def gather_feature_tables(self):
        """-> {name: full [n_items, F] table} rebuilt from the ranks' blocks (collective; for export / state_dict)"""
        import torch.distributed as tdist
        from mmrec_amd.common.lazy_rows import flush_lazy_tables
        flush_lazy_tables(self)
        out, cuts = {}, self.sharding.items.cuts
        for name in ('image_embedding', 'text_embedding'):
            if not hasattr(self, name):
                continue
            w = getattr(self, name).weight
            cap = int(np.diff(cuts).max())
            buf = w.new_zeros(cap, w.shape[1])
            n = self.item_hi - self.item_lo
            buf[:n] = w[:n]
            parts = [torch.empty_like(buf) for _ in range(self.world)]
            if self.world > 1:
                tdist.all_gather(parts, buf, group=self.group)
            else:
                parts = [buf]
            out[name + '.weight'] = torch.cat([p[:int(cuts[r + 1] - cuts[r])] for r, p in enumerate(parts)], 0)
        return out


# ------------------------------------------------------------------------------------------------------------------
# n_gpus > 1, FEATURE-sliced (config `dist_layout: dslice`; the default at 2 / 4 / 8 ranks)
# ------------------------------------------------------------------------------------------------------------------
class _TakeColumns(torch.autograd.Function):
    """this rank's column slice of a REPLICATED [B, d] matrix whose producer needs the gradient of ALL columns (the owner of
    a projected feature row back-propagates through its whole row): backward places the rank's [B, d / P] gradient at its
    columns and sums over the ranks -- every rank ends with the full [B, d] gradient."""

    @staticmethod
    def forward(ctx, x, lo, hi, group, multi):
        ctx.lo, ctx.hi, ctx.width, ctx.group, ctx.multi = lo, hi, x.shape[1], group, multi
        return x[:, lo:hi].contiguous()

    @staticmethod
    def backward(ctx, g):
        full = g.new_zeros(g.shape[0], ctx.width)
        full[:, ctx.lo:ctx.hi] = g
        if ctx.multi:
            import torch.distributed as tdist
            tdist.all_reduce(full, op=tdist.ReduceOp.SUM, group=ctx.group)
        return full, None, None, None, None


class SlicedFREEDOM(RelabelledIdsMixin, FusedEvalMixin, GeneralRecommender):
    """FREEDOM over `n_gpus` = P in {2, 4, 8} processes, the FEATURE dimension sliced:

      * every rank holds the WHOLE user-item graph (and its per-epoch pruned version) and the whole item-item graph, and
        d / P = 32 / 16 / 8 COLUMNS of the id embedding tables, their gradients and their Adam state.  `Y[:, s] = A X[:, s]`:
        all propagation layers, forward and backward, are local launches of mmrec_spmm_csr_f32 on the slice
        (csrc/spmm_narrow.hip) -- NOTHING crosses xGMI there, and a column's values are the single-GPU kernel's bit for bit
        (the row-sharded layout moves 224-336 MB into every rank per layer at config 5 against 0.1 ms of SpMM);
      * the three BPR terms need <u, p> - <u, n> over all 64 columns: one all-reduce of the [3, 2, B] partial dot products;
      * raw feature tables and their Adam state: item-sharded as in RowShardedFREEDOM (owner-computed projections of the
        batch rows, exchanged as [2B, 64] rows; the owners get the full-width gradient back through one all-reduce);
      * evaluation: the slices of the final tables are all-gathered ONCE per evaluation into replicated [N, 64] tables, then
        every rank ranks its share of the users (no exchange in the scoring).
    Parameter names are FREEDOM's; `user_embedding.weight` / `item_id_embedding.weight` hold the rank's columns,
    `image_embedding.weight` / `text_embedding.weight` the rank's rows (`gather_tables()` rebuilds the full tensors).
    Config key `reorder` (round 6: the layout where it pays most -- a gathered row of a d / P slice is a fraction of a 128-byte
    line, and a relabelling that groups neighbours is worth 34 % per 8-column layer, bench.py extra.c5_grouped_ids): every
    rank derives the SAME relabelling from the full graph, all id-indexed state -- the column slices' rows, the item blocks
    the feature tables are cut into, both graphs -- lives in the relabelled ids, batches and evaluation lists are translated
    where they enter and leave, and `gather_tables()` hands the tables back in the dataset's order."""

    relabelled_tables = {'user_embedding.weight': 'u', 'item_id_embedding.weight': 'i'}   # (the row-sharded feature tables: gather_tables)
    row_order_partial = True      # ... whose optimizer state cannot be re-ordered rank by rank: a checkpoint loads under the same `reorder` only

    def __init__(self, config, dataset):
        super().__init__(config, dataset)
        import torch.distributed as tdist
        if not tdist.is_initialized():
            raise RuntimeError('SlicedFREEDOM needs torch.distributed (launch with torchrun; config n_gpus)')
        self.group = None
        self.rank, self.world = tdist.get_rank(), tdist.get_world_size()
        self.force = bool(config['dist_force_collectives'])
        self.multi = self.world > 1 or self.force
        self.embedding_dim = config['embedding_size']
        if self.embedding_dim % self.world or self.embedding_dim // self.world not in (8, 16, 32, 64):
            raise ValueError('dist_layout dslice: embedding_size %d over %d ranks is not a slice width the SpMM kernel has '
                             '(8, 16, 32 columns per rank); use dist_layout: rows' % (self.embedding_dim, self.world))
        self.d_loc = self.embedding_dim // self.world
        self.col_lo, self.col_hi = self.rank * self.d_loc, (self.rank + 1) * self.d_loc
        self.feat_embed_dim = config['feat_embed_dim']
        self.knn_k = config['knn_k']
        self.n_layers = config['n_mm_layers']
        self.n_ui_layers = config['n_ui_layers']
        self.reg_weight = config['reg_weight']
        self.mm_image_weight = config['mm_image_weight']
        self.dropout = config['dropout']
        n_feat = max([0] + [int(f.numel()) for f in (self.v_feat, self.t_feat) if f is not None])
        self.lazy_feature_adam = lazy_adam_enabled(config, n_feat)
        # new key `dist_graph_step`: replay the sliced step -- its small collectives included -- as a hipGraph (RCCL collectives
        # are capturable; common/graph_step.py falls back to eager launches if a capture fails).  Off by default: it could only
        # be exercised on a one-rank group so far.
        self.graph_capturable = bool(config['dist_graph_step'])
        self.pull_batch_rows = _batch_rows_wanted(config, self.n_users + self.n_items)
        nu, ni = self.n_users, self.n_items
        self.n_nodes = nu + ni

        self.interaction_matrix = dataset.inter_matrix(form='coo').astype(np.float32)
        self.norm_adj = norm_adj_graph(self.interaction_matrix, nu, ni, self.device)
        rl = self._setup_relabelling(config, self.norm_adj)     # a function of the full graph: the same on every rank
        if rl is not None:
            self.norm_adj = relabel_graph(self.norm_adj, rl.node_perm_host())
        self.masked_adj = None
        rows = torch.from_numpy(self.interaction_matrix.row.astype(np.int64))
        cols = torch.from_numpy(self.interaction_matrix.col.astype(np.int64))
        self.edge_indices = torch.stack([rows, cols]).to(self.device)
        self.edge_values = hip_ops.edge_norm_values(self.edge_indices[0].contiguous(), self.edge_indices[1].contiguous(),
                                                    nu, ni)
        self.edge_indices = self._map_edges(self.edge_indices)      # (edge ORDER kept: what the per-epoch draw indexes)

        # parameters in FREEDOM's construction order on FULL tables (same generator consumption -> the single-process
        # model's initial values), then this rank's columns
        full_u, full_i = nn.Embedding(nu, self.embedding_dim).weight.data, nn.Embedding(ni, self.embedding_dim).weight.data
        nn.init.xavier_uniform_(full_u)
        nn.init.xavier_uniform_(full_i)
        if rl is not None:            # row `old` of the plain model at relabelled row perm[old]
            full_u, full_i = full_u[rl.inv_u.cpu()], full_i[rl.inv_i.cpu()]
        cols = slice(self.col_lo, self.col_hi)
        self.user_embedding = nn.Embedding.from_pretrained(full_u[:, cols].contiguous(), freeze=False)
        self.item_id_embedding = nn.Embedding.from_pretrained(full_i[:, cols].contiguous(), freeze=False)
        del full_u, full_i
        per = -(-ni // self.world)
        self.item_lo, self.item_hi = min(self.rank * per, ni), min((self.rank + 1) * per, ni)
        self.item_cuts = [min(r * per, ni) for r in range(self.world + 1)]
        table = LazyRowEmbedding if self.lazy_feature_adam else nn.Embedding

        def local_rows(feat):
            if rl is None:
                rows_ = feat[self.item_lo:self.item_hi]
            else:             # the rank's block of RELABELLED items: relabelled row `new` = dataset row inv_i[new]
                rows_ = feat.index_select(0, rl.inv_i[self.item_lo:self.item_hi].to(feat.device))
            return rows_.clone() if rows_.shape[0] else feat.new_zeros(1, feat.shape[1])   # an empty block owns nothing
        if self.v_feat is not None:
            self.image_embedding = table.from_pretrained(local_rows(self.v_feat), freeze=False)
            self.image_trs = nn.Linear(self.v_feat.shape[1], self.feat_embed_dim)
        if self.t_feat is not None:
            self.text_embedding = table.from_pretrained(local_rows(self.t_feat), freeze=False)
            self.text_trs = nn.Linear(self.t_feat.shape[1], self.feat_embed_dim)
        mm = load_or_build_mm_coo(config, self.v_feat, self.t_feat, self.knn_k, self.mm_image_weight, ni,
                                  write=self.rank == 0)       # (built / cached in the dataset's ids, like FREEDOM's)
        self.mm_adj = sparse_coo_to_graph(mm, self.device)
        self.mm_adj.transpose()
        if rl is not None:
            self.mm_adj = relabel_graph(self.mm_adj, rl.perm_i_host)
        self.has_text, self.has_image = self.t_feat is not None, self.v_feat is not None
        if config['dist_keep_full_features'] is not True:
            self.v_feat = None if self.v_feat is None else self.v_feat[:0]
            self.t_feat = None if self.t_feat is None else self.t_feat[:0]

    def pre_epoch_processing(self):
        import torch.distributed as tdist
        if self.dropout <= .0:
            self.masked_adj = self.norm_adj
            return
        keep_len = int(self.edge_values.size(0) * (1. - self.dropout))
        keep = torch.multinomial(self.edge_values, keep_len)     # every rank draws (generators stay aligned) ...
        if self.world > 1:                                        # ... and rank 0's draw is the one everybody uses
            tdist.broadcast(keep, src=0, group=self.group)
        self.set_kept_edges(keep)

    def set_kept_edges(self, keep_idx):
        kept = self.edge_indices[:, keep_idx]
        self.masked_adj = hip_ops.bipartite_graph_from_edges(kept[0].contiguous(), kept[1].contiguous(),
                                                             self.n_users, self.n_items)

    def forward(self, adj):
        """this rank's COLUMNS of (user_all, item_all): FREEDOM.forward on [*, d / P] slices, no collective"""
        u_g, i_g = hip_ops.lightgcn_mean_parts(adj, (self.user_embedding.weight, self.item_id_embedding.weight), self.n_ui_layers)
        h = self.item_id_embedding.weight
        if self.n_layers == 0:
            return u_g, i_g + h
        for _ in range(self.n_layers - 1):
            h = hip_ops.spmm(self.mm_adj, h)
        return u_g, hip_ops.spmm(self.mm_adj, h, Z=i_g)

    def _all_columns(self, t):
        """[n, d / P] slices of all ranks -> the replicated [n, d] table (the layout's one bulk exchange: per evaluation)"""
        import torch.distributed as tdist
        if self.world == 1:
            return t
        parts = torch.empty(self.world * t.shape[0], t.shape[1], dtype=t.dtype, device=t.device)      # rank-major row blocks
        tdist.all_gather_into_tensor(parts, t.contiguous(), group=self.group)
        return parts.view(self.world, t.shape[0], t.shape[1]).permute(1, 0, 2).reshape(t.shape[0], -1).contiguous()

    def eval_embeddings(self):
        u, i = self.forward(self.norm_adj)
        return self._all_columns(u), self._all_columns(i)

    def _owned_projection(self, emb, trs, rows):
        """this rank's columns of the [2B, 64] projected feature rows of the batch items, each row computed by its owner"""
        from mmrec_amd.dist import exchange_owned_rows
        owned = (rows >= self.item_lo) & (rows < self.item_hi)
        if self.lazy_feature_adam:        # slots of other ranks' items: -1 = "no row" (no catch-up, no gradient, no step)
            emb.allow_missing = True
            feats = emb.rows(torch.where(owned, rows - self.item_lo, torch.full_like(rows, -1)))
        else:
            feats = emb.weight[torch.where(owned, rows - self.item_lo, torch.zeros_like(rows))]
        w, b = _SumGradOverRanks.apply(trs.weight, self.group), _SumGradOverRanks.apply(trs.bias, self.group)
        full = exchange_owned_rows(hip_ops.linear(feats, w, b), owned, group=self.group, multi=self.multi)
        return _TakeColumns.apply(full, self.col_lo, self.col_hi, self.group, self.multi)

    def calculate_loss(self, interaction):
        interaction = self._map_batch(interaction)              # (`reorder`: the dataset's ids -> the tables' rows)
        users, pos_items, neg_items = interaction[0], interaction[1], interaction[2]
        rows = torch.cat((pos_items, neg_items))
        b = pos_items.shape[0]
        lp = torch.arange(b, device=rows.device)
        ln = lp + b
        if self.pull_batch_rows and self.n_layers == 1 and not hip_ops.DETERMINISTIC:
            # the propagated tables at the batch's rows only (FREEDOM._loss_at_batch_rows), on this rank's columns
            at = hip_ops.lightgcn_mean_parts_rows(self.masked_adj, (self.user_embedding.weight, self.item_id_embedding.weight),
                                                  self.n_ui_layers, torch.cat((users, rows + self.n_users)))
            ua, users = at[:b], lp
            terms = [(hip_ops.spmm_rows(self.mm_adj, self.item_id_embedding.weight, rows, Z_rows=at[b:]), lp, ln)]
        else:
            ua, ia = self.forward(self.masked_adj)
            terms = [(ia, pos_items, neg_items)]
        if self.has_text:
            terms.append((self._owned_projection(self.text_embedding, self.text_trs, rows), lp, ln))
        if self.has_image:
            terms.append((self._owned_projection(self.image_embedding, self.image_trs, rows), lp, ln))
        return _combine(hip_ops.bpr_losses_shared_users(ua, users, terms, sum_over_ranks=self._sum_over_ranks),
                        self.has_text, self.reg_weight)

    def _sum_over_ranks(self, t):
        """the layout's per-step exchange: the [terms, 2, B] partial dot products, summed in place"""
        if self.multi:
            import torch.distributed as tdist
            tdist.all_reduce(t, op=tdist.ReduceOp.SUM, group=self.group)

    full_sort_topk = RowShardedFREEDOM.full_sort_topk        # users sharded over the ranks, replicated item table

    @torch.no_grad()
    def gather_tables(self):
        """-> {name: full tensor}: the id tables re-assembled from the ranks' columns, the feature tables from their rows
        (collective; for export / state_dict)"""
        import torch.distributed as tdist
        from mmrec_amd.common.lazy_rows import flush_lazy_tables
        flush_lazy_tables(self)
        rl = self.relabelling
        out = {'user_embedding.weight': self._all_columns(self.user_embedding.weight.detach()),
               'item_id_embedding.weight': self._all_columns(self.item_id_embedding.weight.detach())}
        if rl is not None:            # dataset row `old` = relabelled row perm[old]
            out['user_embedding.weight'] = out['user_embedding.weight'].index_select(0, rl.perm_u)
            out['item_id_embedding.weight'] = out['item_id_embedding.weight'].index_select(0, rl.perm_i)
        cuts = self.item_cuts
        for name in ('image_embedding', 'text_embedding'):
            if not hasattr(self, name):
                continue
            w = getattr(self, name).weight
            cap = max(cuts[r + 1] - cuts[r] for r in range(self.world))
            buf = w.new_zeros(max(cap, 1), w.shape[1])
            n = self.item_hi - self.item_lo
            buf[:n] = w[:n]
            parts = [torch.empty_like(buf) for _ in range(self.world)]
            if self.world > 1:
                tdist.all_gather(parts, buf, group=self.group)
            else:
                parts = [buf]
            full = torch.cat([p[:cuts[r + 1] - cuts[r]] for r, p in enumerate(parts)], 0)
            out[name + '.weight'] = full if rl is None else full.index_select(0, rl.perm_i.to(full.device))
        return out

    gather_feature_tables = gather_tables


def ShardedFREEDOM(config, dataset):
    """what `get_model('FREEDOM', sharded=True)` hands to quick_start (config `n_gpus` > 1): the layout named by the new key
    `dist_layout` -- 'dslice' (feature-sliced, no exchange in the propagation: SlicedFREEDOM), 'rows' (row-sharded graphs,
    one all-gather per layer: RowShardedFREEDOM, north_star's wording), or 'auto' / unset: dslice where the embedding width
    divides into slices the kernel has (2 / 4 / 8 ranks at d = 64), rows otherwise.  One GPU measures why
    (bench.py extra.dslice_projection): a slice's layer takes 0.29-0.39 ms at config 5 against 0.78 ms on one GPU, the
    row-sharded layouts are bound by 224-336 MB of exchange per rank and layer."""
    import torch.distributed as tdist
    layout = config['dist_layout'] or 'auto'
    world = tdist.get_world_size() if tdist.is_initialized() else 1
    if layout == 'auto':
        d = config['embedding_size']
        layout = 'dslice' if (world in (2, 4, 8) and d % world == 0 and d // world in (8, 16, 32)) else 'rows'
    if layout not in ('dslice', 'rows'):
        raise ValueError("dist_layout must be 'dslice', 'rows' or 'auto', got %r" % (layout,))
    return (SlicedFREEDOM if layout == 'dslice' else RowShardedFREEDOM)(config, dataset)
