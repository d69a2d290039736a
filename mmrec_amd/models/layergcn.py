"""LayerGCN on the HIP hot path (reference: models/layergcn.py).

Each layer is E <- A E followed by the per-row cosine re-weighting against the ego embedding and
the running layer sum (ego excluded), SpMM kernel + fused cos-scale kernel with a hand-written
backward; BPR is the *sum* variant plus 0.5*||.||^2 on the batch's ego rows.  Edge pruning
alternates degree-sensitive multinomial and uniform sampling per epoch like the reference; the
pruned graph is re-normalised and rebuilt as CSR on the device.
"""
import random

import numpy as np
import torch
import torch.nn as nn

from mmrec_amd import hip_ops
from mmrec_amd.graph import norm_adj_graph
from mmrec_amd.utils.utils import random_sample_range
from mmrec_amd.graph import relabel_graph
from mmrec_amd.models._base import AdjacentTablesMixin, FusedEvalMixin, GeneralRecommender, RelabelledIdsMixin


class LayerGCN(RelabelledIdsMixin, AdjacentTablesMixin, FusedEvalMixin, GeneralRecommender):
    graph_capturable = True       # the step is a fixed launch sequence: replayed as a hipGraph by default (hip_graph_step: auto)
    adjacent_tables = ('user_embeddings', 'item_embeddings')
    relabelled_tables = {'user_embeddings': 'u', 'item_embeddings': 'i'}     # config key `reorder`

    def __init__(self, config, dataset):
        super().__init__(config, dataset)
        self.interaction_matrix = dataset.inter_matrix(form='coo').astype(np.float32)
        self.latent_dim = config['embedding_size']
        self.n_layers = config['n_layers']
        self.reg_weight = config['reg_weight']
        self.dropout = config['dropout']
        self.n_nodes = self.n_users + self.n_items
        self.user_embeddings = nn.Parameter(nn.init.xavier_uniform_(torch.empty(self.n_users, self.latent_dim)))
        self.item_embeddings = nn.Parameter(nn.init.xavier_uniform_(torch.empty(self.n_items, self.latent_dim)))
        self.norm_adj_matrix = norm_adj_graph(self.interaction_matrix, self.n_users, self.n_items, self.device)
        self.masked_adj = None
        self.forward_adj = None
        self.pruning_random = False
        rows = torch.from_numpy(self.interaction_matrix.row.astype(np.int64))
        cols = torch.from_numpy(self.interaction_matrix.col.astype(np.int64))
        self.edge_indices = torch.stack([rows, cols]).to(self.device)
        self.edge_values = hip_ops.edge_norm_values(self.edge_indices[0].contiguous(),
                                                    self.edge_indices[1].contiguous(),
                                                    self.n_users, self.n_items)
        if self._setup_relabelling(config, self.norm_adj_matrix) is not None:      # new key `reorder` (models/_base.py)
            self.norm_adj_matrix = relabel_graph(self.norm_adj_matrix, self.relabelling.node_perm_host())
            self.edge_indices = self._map_edges(self.edge_indices)
            self._to_relabelled_rows_(self.user_embeddings, 'u')
            self._to_relabelled_rows_(self.item_embeddings, 'i')

    def pre_epoch_processing(self):
        if self.dropout <= .0:
            self.masked_adj = self.norm_adj_matrix
            return
        n_edges = self.edge_values.size(0)
        keep_len = int(n_edges * (1. - self.dropout))
        if self.pruning_random:
            keep_idx = torch.as_tensor(random_sample_range(n_edges, keep_len), device=self.device)   # == random.sample
        else:
            keep_idx = torch.multinomial(self.edge_values, keep_len)   # prunes high-degree nodes harder
        self.pruning_random = True ^ self.pruning_random
        self.set_kept_edges(keep_idx)

    def set_kept_edges(self, keep_idx):
        kept = self.edge_indices[:, keep_idx]
        self.masked_adj = hip_ops.bipartite_graph_from_edges(kept[0].contiguous(), kept[1].contiguous(),
                                                             self.n_users, self.n_items)

    def get_ego_embeddings(self):
        return torch.cat([self.user_embeddings, self.item_embeddings], 0)

    def forward(self):
        return hip_ops.layergcn_sum_parts(self.forward_adj, (self.user_embeddings, self.item_embeddings), self.n_layers)

    def eval_embeddings(self):
        self.forward_adj = self.norm_adj_matrix
        return self.forward()

    def calculate_loss(self, interaction):
        interaction = self._map_batch(interaction)
        user, pos, neg = interaction[0], interaction[1], interaction[2]
        self.forward_adj = self.masked_adj
        u_all, i_all = self.forward()
        mf_loss, = hip_ops.bpr_losses_shared_users(u_all, user, [(i_all, pos, neg)], hip_ops.BPR_LOGSIG, 'sum', joint_grad=True)
        # 0.5 * (||u||^2 + ||p||^2 + ||n||^2) * reg_weight over the batch's EGO rows (layergcn.py:154-161), one launch pair
        reg = hip_ops.rows_reg(((self.user_embeddings, user), (self.item_embeddings, pos), (self.item_embeddings, neg)),
                               hip_ops.ROWS_REG_SQUARED, 0.5 * self.reg_weight)
        return mf_loss + reg
