"""BM3 on the HIP hot path (reference: models/bm3.py).

forward : fused LightGCN layer mean + residual item id embedding
loss    : BYOL-style cosine losses between predictor outputs and dropout targets; projections and
          the 64x64 predictor run on the fp32 MFMA GEMM, EmbLoss over all rows.  The reference projects ALL items
          every batch (bm3.py:102-104) but consumes only the batch's rows (:124-127): with `lazy_projection` (default)
          the <= B gathered feature rows are projected -- same function, same gradients; the dropout masks of the
          targets are still drawn per ITEM ([n_items, 64], as in the reference: duplicates of an item share a mask)
          and gathered.  `lazy_feature_adam`: row-lazy exact Adam on the feature tables (common/lazy_rows.py)
eval    : predictor on all users/items, then fused score + mask + top-K
No negative sampling (`use_neg_sampling: False`): batches are [2, B] (user, item).
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from mmrec_amd import hip_ops
from mmrec_amd.common.lazy_rows import LazyRowEmbedding, lazy_adam_enabled
from mmrec_amd.graph import norm_adj_graph, relabel_graph
from mmrec_amd.models._base import AdjacentTablesMixin, FusedEvalMixin, GeneralRecommender, RelabelledIdsMixin


class BM3(RelabelledIdsMixin, AdjacentTablesMixin, FusedEvalMixin, GeneralRecommender):
    graph_capturable = True       # the step is a fixed launch sequence: replayed as a hipGraph by default (hip_graph_step: auto)

    adjacent_tables = ('user_embedding.weight', 'item_id_embedding.weight')
    relabelled_tables = {'user_embedding.weight': 'u', 'item_id_embedding.weight': 'i', 'image_embedding.weight': 'i',
                         'text_embedding.weight': 'i'}     # config key `reorder` (models/_base.py)

    def __init__(self, config, dataset):
        super().__init__(config, dataset)
        self.embedding_dim = config['embedding_size']
        self.feat_embed_dim = config['embedding_size']
        self.n_layers = config['n_layers']
        self.reg_weight = config['reg_weight']
        self.cl_weight = config['cl_weight']
        self.dropout = config['dropout']
        self.n_nodes = self.n_users + self.n_items
        lazy = config['lazy_projection']
        self.lazy_projection = True if lazy is None else bool(lazy)
        n_feat = max([0] + [int(f.numel()) for f in (self.v_feat, self.t_feat) if f is not None])
        self.lazy_feature_adam = lazy_adam_enabled(config, n_feat) and self.lazy_projection
        self.lazy_prefetch = config['lazy_prefetch'] is not False    # new key: catch-up on a side stream (default on)
        table = LazyRowEmbedding if self.lazy_feature_adam else nn.Embedding
        self.norm_adj = norm_adj_graph(dataset.inter_matrix(form='coo').astype(np.float32),
                                       self.n_users, self.n_items, self.device)
        # new key `reorder`: the id-indexed tables live in an id space relabelled once, here (bm3.py:84-95's propagation gathers
        # with locality; the graph keeps every row's nonzero order, so its sums are the plain model's bit for bit)
        rl = self._setup_relabelling(config, self.norm_adj)
        if rl is not None:
            self.norm_adj = relabel_graph(self.norm_adj, rl.node_perm_host())
        self.user_embedding = nn.Embedding(self.n_users, self.embedding_dim)
        self.item_id_embedding = nn.Embedding(self.n_items, self.embedding_dim)
        nn.init.xavier_uniform_(self.user_embedding.weight)
        nn.init.xavier_uniform_(self.item_id_embedding.weight)
        if rl is not None:            # the plain model's initial values, row `old` at relabelled row perm[old]
            self._to_relabelled_rows_(self.user_embedding.weight, 'u')
            self._to_relabelled_rows_(self.item_id_embedding.weight, 'i')
        self.predictor = nn.Linear(self.embedding_dim, self.embedding_dim)
        nn.init.xavier_normal_(self.predictor.weight)
        in_space = (lambda f: f) if rl is None else (lambda f: f.index_select(0, rl.inv_i.to(f.device)))
        if self.v_feat is not None:
            self.image_embedding = table.from_pretrained(in_space(self.v_feat), freeze=False)
            self.image_trs = nn.Linear(self.v_feat.shape[1], self.feat_embed_dim)
            nn.init.xavier_normal_(self.image_trs.weight)
        if self.t_feat is not None:
            self.text_embedding = table.from_pretrained(in_space(self.t_feat), freeze=False)
            self.text_trs = nn.Linear(self.t_feat.shape[1], self.feat_embed_dim)
            nn.init.xavier_normal_(self.text_trs.weight)

    def forward(self):
        u, i = hip_ops.lightgcn_mean_parts(self.norm_adj, (self.user_embedding.weight, self.item_id_embedding.weight),
                                           self.n_layers)
        return u, i + self.item_id_embedding.weight

    def _predict(self, x):
        return hip_ops.linear(x, self.predictor.weight, self.predictor.bias)

    def eval_embeddings(self):
        u, i = self.forward()
        return self._predict(u.contiguous()), self._predict(i.contiguous())

    def _targets(self, *tensors, sides=None):
        """dropout(clone) targets without gradient (bm3.py:107-122; F.dropout is out of place, so the reference's clone is not
        repeated: four D2D copies per step); one F.dropout call per tensor, in the
        reference's order (u, i, t, v) so an injected dropout function replays the same masks.  Under `reorder` a mask is
        drawn for the rows in the DATASET's order (the plain model's draw, element for element) and carried to the relabelled
        rows: sides[j] = 'u' / 'i' names tensor j's id space, None = rows already in dataset order."""
        rl = self.relabelling
        with torch.no_grad():
            if rl is None:
                return [F.dropout(t.detach(), self.dropout) for t in tensors]
            out = []
            for t, side in zip(tensors, sides):
                if side is None:
                    out.append(F.dropout(t.detach(), self.dropout))
                    continue
                perm, inv = (rl.perm_u, rl.inv_u) if side == 'u' else (rl.perm_i, rl.inv_i)
                out.append(F.dropout(t.detach().index_select(0, perm), self.dropout).index_select(0, inv))
            return out

    def calculate_loss(self, interactions):
        if self.lazy_projection and self.lazy_feature_adam and self.lazy_prefetch:   # row catch-up on the side stream
            rows_pf = interactions[1] if self.relabelling is None else self.relabelling.perm_i[interactions[1]]
            for emb in (getattr(self, 'text_embedding', None), getattr(self, 'image_embedding', None)):
                if emb is not None:
                    emb.prefetch(rows_pf)
        items_ds = interactions[1]                  # the dataset's ids: what the per-item dropout masks are indexed with
        interactions = self._map_batch(interactions)
        u_ori, i_ori = self.forward()
        u_ori, i_ori = u_ori.contiguous(), i_ori.contiguous()
        users, items = interactions[0], interactions[1]
        lazy = self.lazy_projection
        if lazy:
            rows = (lambda emb: emb.rows(items)) if self.lazy_feature_adam else (lambda emb: emb.weight[items])
        else:
            rows = lambda emb: emb.weight
        t_on = v_on = None      # lazy: [B, 64] rows of the batch's items; else [n_items, 64]
        if self.t_feat is not None:
            t_on = hip_ops.linear(rows(self.text_embedding), self.text_trs.weight, self.text_trs.bias)
        if self.v_feat is not None:
            v_on = hip_ops.linear(rows(self.image_embedding), self.image_trs.weight, self.image_trs.bias)
        # dropout targets in the reference's order and shapes (u, i, t, v); lazy: the per-item masks of t and v
        ones = torch.ones_like(i_ori) if lazy else None
        # (lazy: the masks of t and v are drawn over `ones` per item in dataset order and gathered with the dataset's ids)
        cand = ((u_ori, 'u'), (i_ori, 'i'), (None if t_on is None else (ones if lazy else t_on), None if lazy else 'i'),
                (None if v_on is None else (ones if lazy else v_on), None if lazy else 'i'))
        targets = self._targets(*[t for t, _ in cand if t is not None], sides=[sd for t, sd in cand if t is not None])
        u_tgt, i_tgt = targets[0], targets[1]
        rest = targets[2:]
        t_tgt = rest.pop(0) if t_on is not None else None
        v_tgt = rest.pop(0) if v_on is not None else None
        # six BYOL terms 1 - mean cos(online, detached target): each ONE fused gather-dot-norm kernel (+ one scatter
        # kernel backward) instead of ~20 elementwise / reduction launches (bm3.py:129-144)
        # (round 6: ALL of them in one launch pair -- hip_ops.cosine_means, ABI 14 -- as const - sum_t w_t mean cos_t)
        # the ONE predictor (bm3.py:61,129-135: self.predictor on u, i, t, v) applied to the row-wise cat of its inputs: one
        # forward and one backward instead of four each, and no sums of four weight / bias gradients (rows are independent:
        # the same values)
        online = [u_ori, i_ori] + [x for x in (t_on, v_on) if x is not None]
        preds = list(self._predict(torch.cat(online, 0)).split([x.shape[0] for x in online]))
        u_pred, i_pred = preds.pop(0), preds.pop(0)
        cl = float(self.cl_weight)
        terms, const = [(u_pred, users, i_tgt, items, -1.0), (i_pred, items, u_tgt, users, -1.0)], 2.0      # loss_ui, loss_iu
        if t_on is not None:
            t_pred, t_idx = preds.pop(0), (None if lazy else items)
            t_tgt = t_on.detach() * t_tgt[items_ds, :] if lazy else t_tgt
            terms += [(t_pred, t_idx, i_tgt, items, -cl), (t_pred, t_idx, t_tgt, t_idx, -cl)]                # loss_t, loss_tv
            const += 2.0 * cl
        if v_on is not None:
            v_pred, v_idx = preds.pop(0), (None if lazy else items)
            v_tgt = v_on.detach() * v_tgt[items_ds, :] if lazy else v_tgt
            terms += [(v_pred, v_idx, i_tgt, items, -cl), (v_pred, v_idx, v_tgt, v_idx, -cl)]                # loss_v, loss_vt
            const += 2.0 * cl
        # EmbLoss(u, i) = (||u|| + ||i||) / n_items over the WHOLE tables (bm3.py:146), times reg_weight: one launch pair
        reg = hip_ops.rows_reg(((u_ori, None), (i_ori, None)), hip_ops.ROWS_REG_NORM, float(self.reg_weight) / i_ori.shape[0])
        return const + hip_ops.cosine_means(terms) + reg
