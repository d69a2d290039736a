"""LightGCN on the HIP hot path (reference: models/lightgcn.py).

forward  = mean_l(A^l E0): L launches of the CSR SpMM with the layer mean fused into its epilogue
loss     = fused gather-dot-log(1e-10+sigmoid) BPR + reg_weight * EmbLoss on the batch's ego rows
eval     = fused score + mask + top-K
"""
import numpy as np
import torch
import torch.nn as nn

from mmrec_amd import hip_ops
from mmrec_amd.graph import norm_adj_graph
from mmrec_amd.graph import relabel_graph
from mmrec_amd.models._base import AdjacentTablesMixin, FusedEvalMixin, GeneralRecommender, RelabelledIdsMixin, emb_loss_rows


class LightGCN(RelabelledIdsMixin, AdjacentTablesMixin, FusedEvalMixin, GeneralRecommender):
    graph_capturable = True       # the step is a fixed launch sequence: replayed as a hipGraph by default (hip_graph_step: auto)

    adjacent_tables = ('embedding_dict.user_emb', 'embedding_dict.item_emb')
    relabelled_tables = {'embedding_dict.user_emb': 'u', 'embedding_dict.item_emb': 'i'}     # config key `reorder`

    def __init__(self, config, dataset):
        super().__init__(config, dataset)
        self.interaction_matrix = dataset.inter_matrix(form='coo').astype(np.float32)
        self.latent_dim = config['embedding_size']
        self.n_layers = config['n_layers']
        self.reg_weight = config['reg_weight']
        init = nn.init.xavier_uniform_
        self.embedding_dict = nn.ParameterDict({
            'user_emb': nn.Parameter(init(torch.empty(self.n_users, self.latent_dim))),
            'item_emb': nn.Parameter(init(torch.empty(self.n_items, self.latent_dim)))})
        self.norm_adj_matrix = norm_adj_graph(self.interaction_matrix, self.n_users, self.n_items, self.device)
        if self._setup_relabelling(config, self.norm_adj_matrix) is not None:      # new key `reorder` (models/_base.py)
            self.norm_adj_matrix = relabel_graph(self.norm_adj_matrix, self.relabelling.node_perm_host())
            self._to_relabelled_rows_(self.embedding_dict['user_emb'], 'u')
            self._to_relabelled_rows_(self.embedding_dict['item_emb'], 'i')

    def get_ego_embeddings(self):
        return torch.cat([self.embedding_dict['user_emb'], self.embedding_dict['item_emb']], 0)

    def forward(self):
        return hip_ops.lightgcn_mean_parts(self.norm_adj_matrix, (self.embedding_dict['user_emb'],
                                                                  self.embedding_dict['item_emb']), self.n_layers)

    eval_embeddings = forward

    def calculate_loss(self, interaction):
        interaction = self._map_batch(interaction)
        user, pos, neg = interaction[0], interaction[1], interaction[2]
        u_all, i_all = self.forward()
        mf_loss = hip_ops.bpr_loss(u_all, i_all, user, pos, neg, hip_ops.BPR_GAMMA, 'mean')
        ue, ie = self.embedding_dict['user_emb'], self.embedding_dict['item_emb']
        return mf_loss + emb_loss_rows(((ue, user), (ie, pos), (ie, neg)), user.shape[0], self.reg_weight)
