"""LATTICE on the HIP hot path (reference: models/lattice.py).

The reference keeps the item-item graph as a DENSE [n_items, n_items] matrix (199 MB on Baby; why
Elec is "-" in its results table) and rebuilds it on the first batch of every epoch and in every
eval call.  Here it is what it really is, a sparse kNN graph: neighbours come from the fused
score+top-K kernel, their similarity values are recomputed differentiably for the kept pairs only
(gradient reaches image_trs / text_trs / modal_weight exactly as in the reference), the original and
learned graphs are concatenated COO and propagated with the HIP SpMM (`spmm_vals`: gradient w.r.t.
embeddings AND values).  u-i propagation: row-normalised D^-1(A+I) (not symmetric: transposed CSR for
backward) through the fused layer-mean kernel.
"""
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from mmrec_amd import hip_ops
from mmrec_amd.graph import relabel_graph
from mmrec_amd.models._base import AdjacentTablesMixin, FusedEvalMixin, GeneralRecommender, RelabelledIdsMixin


def _sym_norm_values(rows, cols, vals, n):
    """D^-1/2 A D^-1/2 with D = row sums of the (possibly duplicated-entry) COO; inf -> 0."""
    rowsum = torch.zeros(n, dtype=vals.dtype, device=vals.device).index_add(0, rows, vals)
    d = torch.pow(rowsum, -0.5)
    d = torch.where(torch.isinf(d), torch.zeros_like(d), d)
    return d[rows] * vals * d[cols]


class LATTICE(RelabelledIdsMixin, AdjacentTablesMixin, FusedEvalMixin, GeneralRecommender):
    # the first batch of an epoch builds the learned item graph (lattice.py:137-159), later ones reuse it: that batch runs
    # eagerly, the step is captured on the second one and replayed for the rest of the epoch (common/graph_step.py)
    graph_capturable = True
    graph_eager_batches = 1
    adjacent_tables = ('user_embedding.weight', 'item_id_embedding.weight')     # the cat of lattice.py:184 already exists
    relabelled_tables = {'user_embedding.weight': 'u', 'item_id_embedding.weight': 'i', 'image_embedding.weight': 'i',
                         'text_embedding.weight': 'i'}     # config key `reorder` (models/_base.py)

    def __init__(self, config, dataset):
        super().__init__(config, dataset)
        self.embedding_dim = config['embedding_size']
        self.feat_embed_dim = config['feat_embed_dim']
        self.weight_size = config['weight_size']
        self.knn_k = config['knn_k']
        self.lambda_coeff = config['lambda_coeff']
        self.cf_model = config['cf_model']
        self.n_layers = config['n_layers']
        self.reg_weight = config['reg_weight']
        self.build_item_graph = True

        self.interaction_matrix = dataset.inter_matrix(form='coo').astype(np.float32)
        self.norm_adj = self._row_norm_graph()
        # new key `reorder`: the u-i propagation (lattice.py:184-195) and every id-indexed table in an id space relabelled once,
        # here.  The kNN graphs are still FOUND in the dataset's ids (ties between equal similarities go to the lower dataset
        # id, the reference's cache files keep their meaning) and their (row, col) pairs renamed in order, so every row of
        # every graph sums the plain model's terms in the plain model's order.
        rl = self._setup_relabelling(config, self.norm_adj)
        if rl is not None:
            self.norm_adj = relabel_graph(self.norm_adj, rl.node_perm_host())
        in_space = (lambda f: f) if rl is None else (lambda f: f.index_select(0, rl.inv_i.to(f.device)))
        self.item_adj = None      # (DynGraph, values) of the current item graph

        self.n_ui_layers = len(self.weight_size)
        self.weight_size = [self.embedding_dim] + self.weight_size
        self.user_embedding = nn.Embedding(self.n_users, self.embedding_dim)
        self.item_id_embedding = nn.Embedding(self.n_items, self.embedding_dim)
        nn.init.xavier_uniform_(self.user_embedding.weight)
        nn.init.xavier_uniform_(self.item_id_embedding.weight)
        if rl is not None:            # the plain model's initial values, row `old` at relabelled row perm[old]
            self._to_relabelled_rows_(self.user_embedding.weight, 'u')
            self._to_relabelled_rows_(self.item_id_embedding.weight, 'i')
        if self.cf_model == 'ngcf':
            self.GC_Linear_list, self.Bi_Linear_list, self.dropout_list = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
            for i in range(self.n_ui_layers):
                self.GC_Linear_list.append(nn.Linear(self.weight_size[i], self.weight_size[i + 1]))
                self.Bi_Linear_list.append(nn.Linear(self.weight_size[i], self.weight_size[i + 1]))
                self.dropout_list.append(nn.Dropout(config['mess_dropout'][i]))

        dataset_path = os.path.abspath(config['data_path'] + config['dataset'])
        if self.v_feat is not None:
            self.image_embedding = nn.Embedding.from_pretrained(in_space(self.v_feat), freeze=False)
            self.image_original = self._original_graph(
                self.v_feat, os.path.join(dataset_path, 'image_adj_{}.pt'.format(self.knn_k)))
        if self.t_feat is not None:
            self.text_embedding = nn.Embedding.from_pretrained(in_space(self.t_feat), freeze=False)
            self.text_original = self._original_graph(
                self.t_feat, os.path.join(dataset_path, 'text_adj_{}.pt'.format(self.knn_k)))
        if self.v_feat is not None:
            self.image_trs = nn.Linear(self.v_feat.shape[1], self.feat_embed_dim)
        if self.t_feat is not None:
            self.text_trs = nn.Linear(self.t_feat.shape[1], self.feat_embed_dim)
        self.modal_weight = nn.Parameter(torch.Tensor([0.5, 0.5]))
        self.softmax = nn.Softmax(dim=0)

    # ---- graphs -------------------------------------------------------------------------------
    def _row_norm_graph(self):
        """D^-1 (A + I), float64 -> float32 (lattice.py:100-122)."""
        from mmrec_amd.graph import unique_edges
        nu, ni = self.n_users, self.n_items
        eu, ei = unique_edges(self.interaction_matrix.row, self.interaction_matrix.col, ni)
        n = nu + ni
        rows = np.concatenate([eu, ei + nu, np.arange(n)])
        cols = np.concatenate([ei + nu, eu, np.arange(n)])
        order = np.lexsort((cols, rows))
        rows, cols = rows[order], cols[order]
        val = np.power(np.bincount(rows, minlength=n).astype(np.float64), -1.0)[rows].astype(np.float32)
        return hip_ops.CsrGraph.from_coo_host(np.stack([rows, cols]), val, n, n, self.device)

    def _knn_pairs(self, feats_normed, in_dataset_ids=False):
        """(rows, cols) of the kNN graph of the given rows, in the id space the rows are in.  Relabelled rows are ranked in
        the dataset's order and the pairs renamed (see __init__)."""
        rl = self.relabelling
        f = feats_normed.detach()
        if rl is not None and not in_dataset_ids:
            f = f.index_select(0, rl.perm_i)                     # row `old` of the dataset order = relabelled row perm[old]
        knn = hip_ops.score_topk(f.contiguous(), f.contiguous(), self.knn_k)
        rows = torch.arange(self.n_items, device=knn.device).repeat_interleave(self.knn_k)
        cols = knn.reshape(-1)
        if rl is not None and not in_dataset_ids:
            rows, cols = rl.perm_i[rows], rl.perm_i[cols]
        return rows, cols

    def _weighted_knn(self, feats, in_dataset_ids=False):
        """top-k cosine neighbours with their similarity as (differentiable) weight: build_sim +
        build_knn_neighbourhood (utils/utils.py:119-137) without the dense matrix."""
        fn = feats.div(torch.norm(feats, p=2, dim=-1, keepdim=True))
        rows, cols = self._knn_pairs(fn, in_dataset_ids)
        return rows, cols, (fn[rows] * fn[cols]).sum(-1)

    def _original_graph(self, raw_feats, cache):
        """Frozen per-modality graph.  The reference caches it as a dense [I, I] tensor
        (`image_adj_{k}.pt`, lattice.py:64-87): such a file is used when present, and written in that
        format when it stays under graph.DENSE_CACHE_LIMIT_BYTES."""
        from mmrec_amd.graph import DENSE_CACHE_LIMIT_BYTES, coo_to_dense_adj, dense_adj_to_coo, load_cached_adj
        rl = self.relabelling
        rename = (lambda r, c, v: (r, c, v)) if rl is None else (lambda r, c, v: (rl.perm_i[r], rl.perm_i[c], v))
        dense = load_cached_adj(cache)
        if dense is not None:
            rows, cols, vals = dense_adj_to_coo(dense.to(torch.float32))
            return rename(rows.to(self.device), cols.to(self.device), vals.to(self.device))
        with torch.no_grad():      # (raw_feats: the dataset's table, in the dataset's ids -- as the cache file is)
            rows, cols, sim = self._weighted_knn(raw_feats.to(torch.float32), in_dataset_ids=True)
            vals = _sym_norm_values(rows, cols, sim, self.n_items)
        if self.n_items * self.n_items * 4 <= DENSE_CACHE_LIMIT_BYTES and os.path.isdir(os.path.dirname(cache)):
            torch.save(coo_to_dense_adj(rows, cols, vals, self.n_items), cache)
        return rename(rows, cols, vals)

    def _build_item_adj(self, image_feats, text_feats):
        weight = self.softmax(self.modal_weight)
        lr, lc, lv, orr, oc, ov = [], [], [], [], [], []
        mods = []
        if self.v_feat is not None:
            mods.append((image_feats, self.image_original))
        if self.t_feat is not None:
            mods.append((text_feats, self.text_original))
        for m, (feats, orig) in enumerate(mods):
            w = weight[m] if len(mods) == 2 else 1.0
            r, c, v = self._weighted_knn(feats)
            lr.append(r), lc.append(c), lv.append(w * v)
            orr.append(orig[0]), oc.append(orig[1]), ov.append(w * orig[2])
        lr, lc, lv = torch.cat(lr), torch.cat(lc), torch.cat(lv)
        learned = _sym_norm_values(lr, lc, lv, self.n_items)
        rows = torch.cat([lr] + orr)
        cols = torch.cat([lc] + oc)
        vals = torch.cat([(1 - self.lambda_coeff) * learned] + [self.lambda_coeff * v for v in ov])
        return hip_ops.DynGraph(rows.contiguous(), cols.contiguous(), self.n_items, self.n_items), vals

    def pre_epoch_processing(self):
        self.build_item_graph = True

    # ---- forward ------------------------------------------------------------------------------
    def forward(self, adj, build_item_graph=False):
        if build_item_graph:
            image_feats = text_feats = None
            if self.v_feat is not None:
                image_feats = hip_ops.linear(self.image_embedding.weight, self.image_trs.weight, self.image_trs.bias)
            if self.t_feat is not None:
                text_feats = hip_ops.linear(self.text_embedding.weight, self.text_trs.weight, self.text_trs.bias)
            self.item_adj = self._build_item_adj(image_feats, text_feats)
        else:
            self.item_adj = (self.item_adj[0], self.item_adj[1].detach())
        dyn, vals = self.item_adj
        h = self.item_id_embedding.weight
        for _ in range(self.n_layers):
            h = hip_ops.spmm_vals(dyn, h, vals)
        h = hip_ops.row_normalize(h)              # F.normalize(h, p=2, dim=1) (lattice.py:165), one launch each way
        if self.cf_model == 'mf':
            return self.user_embedding.weight, self.item_id_embedding.weight + h
        if self.cf_model == 'lightgcn':       # (the two tables as row blocks: no cat forward, no split / zero-filled halves backward)
            u_g, i_g = hip_ops.lightgcn_mean_parts(adj, (self.user_embedding.weight, self.item_id_embedding.weight), self.n_ui_layers)
            return u_g, i_g + h
        ego = torch.cat((self.user_embedding.weight, self.item_id_embedding.weight), dim=0)
        if self.cf_model == 'ngcf':
            layers = [ego]
            for i in range(self.n_ui_layers):
                side = hip_ops.spmm(adj, ego)
                lin = hip_ops.linear if self.weight_size[i + 1] == 64 else None
                gc, bi = self.GC_Linear_list[i], self.Bi_Linear_list[i]
                s = lin(side, gc.weight, gc.bias) if lin else gc(side)
                b = lin((ego * side).contiguous(), bi.weight, bi.bias) if lin else bi(ego * side)
                ego = self.dropout_list[i](F.leaky_relu(s) + F.leaky_relu(b))
                layers.append(F.normalize(ego, p=2, dim=1))
            mean = torch.stack(layers, dim=1).mean(dim=1)
        else:
            raise ValueError('unknown cf_model {}'.format(self.cf_model))
        return mean[:self.n_users], mean[self.n_users:] + h

    def eval_embeddings(self):
        return self.forward(self.norm_adj, build_item_graph=True)

    def calculate_loss(self, interaction):
        interaction = self._map_batch(interaction)
        users, pos_items, neg_items = interaction[0], interaction[1], interaction[2]
        ua, ia = self.forward(self.norm_adj, build_item_graph=self.build_item_graph)
        self.build_item_graph = False
        ua, ia = ua.contiguous(), ia.contiguous()
        mf_loss = hip_ops.bpr_loss(ua, ia, users, pos_items, neg_items)
        reg = hip_ops.rows_reg(((ua, users), (ia, pos_items), (ia, neg_items)), hip_ops.ROWS_REG_SQUARED,
                               0.5 * self.reg_weight / self.batch_size)
        return mf_loss + reg
