"""BPR matrix factorisation on the HIP hot path (reference: models/bpr.py).

loss = fused gather-dot-log(1e-10+sigmoid) BPR (common/loss.py:33-35) + reg_weight * EmbLoss on the
       batch's rows (bpr.py:74-91; forward()'s dropout has p = 0 there)
eval = fused score + mask + top-K on the raw embedding tables (bpr.py:93-99)
"""
import torch.nn as nn

from mmrec_amd import hip_ops
from mmrec_amd.common.init import xavier_normal_initialization
from mmrec_amd.models._base import FusedEvalMixin, GeneralRecommender, emb_loss_rows


class BPR(FusedEvalMixin, GeneralRecommender):
    graph_capturable = True       # the step is a fixed launch sequence: replayed as a hipGraph by default (hip_graph_step: auto)

    def __init__(self, config, dataset):
        super().__init__(config, dataset)
        self.embedding_size = config['embedding_size']
        self.reg_weight = config['reg_weight']
        self.user_embedding = nn.Embedding(self.n_users, self.embedding_size)
        self.item_embedding = nn.Embedding(self.n_items, self.embedding_size)
        self.apply(xavier_normal_initialization)

    def get_user_embedding(self, user):
        return self.user_embedding(user)

    def get_item_embedding(self, item):
        return self.item_embedding(item)

    def forward(self, dropout=0.0):
        return self.user_embedding.weight, self.item_embedding.weight

    eval_embeddings = forward

    def calculate_loss(self, interaction):
        user, pos, neg = interaction[0], interaction[1], interaction[2]
        ue, ie = self.forward()
        mf_loss = hip_ops.bpr_loss(ue, ie, user, pos, neg, hip_ops.BPR_GAMMA, 'mean')
        return mf_loss + emb_loss_rows(((ue, user), (ie, pos), (ie, neg)), neg.shape[0], self.reg_weight)
