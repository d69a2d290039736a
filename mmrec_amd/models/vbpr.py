"""VBPR on the HIP hot path (reference: models/vbpr.py).

item embedding = cat(id embedding [64], Linear(cat(text, image) raw features) [64]); users are 128-d.
The raw features are constants here (no dX), the projection is the fp32 MFMA GEMM, BPR + EmbLoss
run on the fused gather kernels at row width 128.
"""
import torch
import torch.nn as nn

from mmrec_amd import hip_ops
from mmrec_amd.common.init import xavier_normal_initialization
from mmrec_amd.models._base import FusedEvalMixin, GeneralRecommender, emb_loss_rows


class VBPR(FusedEvalMixin, GeneralRecommender):
    graph_capturable = True       # the step is a fixed launch sequence: replayed as a hipGraph by default (hip_graph_step: auto)

    def __init__(self, config, dataloader):
        super().__init__(config, dataloader)
        self.u_embedding_size = self.i_embedding_size = config['embedding_size']
        self.reg_weight = config['reg_weight']
        self.u_embedding = nn.Parameter(nn.init.xavier_uniform_(torch.empty(self.n_users, self.u_embedding_size * 2)))
        self.i_embedding = nn.Parameter(nn.init.xavier_uniform_(torch.empty(self.n_items, self.i_embedding_size)))
        if self.v_feat is not None and self.t_feat is not None:
            self.item_raw_features = torch.cat((self.t_feat, self.v_feat), -1)
        else:
            self.item_raw_features = self.v_feat if self.v_feat is not None else self.t_feat
        self.item_linear = nn.Linear(self.item_raw_features.shape[1], self.i_embedding_size)
        self.apply(xavier_normal_initialization)

    def forward(self, dropout=0.0):
        projected = hip_ops.linear(self.item_raw_features, self.item_linear.weight, self.item_linear.bias)
        items = torch.cat((self.i_embedding, projected), -1)
        if dropout > 0.0:
            return nn.functional.dropout(self.u_embedding, dropout), nn.functional.dropout(items, dropout)
        return self.u_embedding, items

    eval_embeddings = forward

    def calculate_loss(self, interaction):
        user, pos, neg = interaction[0], interaction[1], interaction[2]
        ue, ie = self.forward()
        mf_loss = hip_ops.bpr_loss(ue, ie, user, pos, neg, hip_ops.BPR_GAMMA, 'mean')
        return mf_loss + emb_loss_rows(((ue, user), (ie, pos), (ie, neg)), user.shape[0], self.reg_weight)
