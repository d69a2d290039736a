"""MGCN on the HIP hot path (reference: models/mgcn.py).

init   : D^-1/2 A D^-1/2 of the u-i graph in float32 without epsilon (mgcn.py:111-136) and its
         user-rows block R; per-modality kNN graphs weighted by the kept cosine similarities and
         normalised by their row sums (utils/utils.py:166-177) -- neighbours AND values come from the
         fused score+top-K kernel, the [I, I] similarity matrix is never built
forward: modal projections on the fp32 MFMA GEMM, LightGCN layer mean, item-item and R SpMMs on the
         CSR kernel (R and the kNN graphs are not symmetric: transposed CSR for the backward)
loss   : fused BPR + fused gather-norm regulariser + two fused in-batch InfoNCE terms
         (`hip_ops.infonce`: the [B, B] logits are never materialised, forward or backward)
eval   : fused score + mask + top-K
The 64 -> 64 gate / attention layers run on the same fp32 MFMA projection kernels (the library's
skinny dW GEMMs measured 107-150 us each, 30 % of the step); only the 64 -> 1 attention head is a
library matvec.
"""
import os

import numpy as np
import torch
import torch.nn as nn

from mmrec_amd import hip_ops
from mmrec_amd.graph import unique_edges
from mmrec_amd.models._base import FusedEvalMixin, GeneralRecommender


def mgcn_norm_graphs(inter_coo, n_users, n_items, device):
    """(norm_adj [N, N] symmetric, R [U, I] = its user rows) -- mgcn.py:111-136: float32 row sums,
    pow(-0.5), inf -> 0."""
    eu, ei = unique_edges(inter_coo.row, inter_coo.col, n_items)
    du = np.bincount(eu, minlength=n_users).astype(np.float32)
    di = np.bincount(ei, minlength=n_items).astype(np.float32)
    with np.errstate(divide="ignore"):
        su, si = np.power(du, np.float32(-0.5)), np.power(di, np.float32(-0.5))
    su[np.isinf(su)] = 0.
    si[np.isinf(si)] = 0.
    v = (su[eu] * si[ei]).astype(np.float32)                 # edges are sorted by (user, item)
    o = np.argsort(ei, kind="stable")
    rows = np.concatenate([eu, ei[o] + n_users])
    cols = np.concatenate([ei + n_users, eu[o]])
    n = n_users + n_items
    adj = hip_ops.CsrGraph.from_coo_host(np.stack([rows, cols]), np.concatenate([v, v[o]]), n, n, device,
                                         symmetric=True)
    R = hip_ops.CsrGraph.from_coo_host(np.stack([eu, ei]), v, n_users, n_items, device)
    R.transpose()
    return adj, R


def knn_sym_graph(feats, k, cache=None):
    """build_sim + build_knn_normalized_graph(sparse, 'sym'): kNN(k) by cosine similarity, edge weight =
    similarity, w' = d^-1/2[row] w d^-1/2[col], d = row sums of the kept weights.  `cache`: the
    reference's `image_adj_{k}_True.pt` file (a torch sparse COO tensor, mgcn.py:43-70): loaded when
    present, written otherwise."""
    from mmrec_amd.graph import load_cached_adj, sparse_coo_to_graph
    sp_ = load_cached_adj(cache) if cache else None
    if sp_ is not None:
        g = sparse_coo_to_graph(sp_, feats.device)
        g.transpose()
        return g
    x = feats.detach().to(torch.float32)
    xn = x.div(torch.norm(x, p=2, dim=-1, keepdim=True)).contiguous()
    idx, val = hip_ops.score_topk(xn, xn, k, return_values=True)
    n = x.shape[0]
    rows = torch.arange(n, device=x.device).unsqueeze(1).expand(-1, k).reshape(-1)
    cols, w = idx.reshape(-1), val.reshape(-1)
    deg = torch.zeros(n, device=x.device).index_add_(0, rows, w)
    dis = deg.pow(-0.5)
    dis = torch.where(torch.isinf(dis), torch.zeros_like(dis), dis)
    vals = (dis[rows] * w * dis[cols]).contiguous()
    if cache and os.path.isdir(os.path.dirname(cache)):
        torch.save(torch.sparse_coo_tensor(torch.stack([rows, cols]).cpu(), vals.cpu(), (n, n)), cache)
    g = hip_ops.CsrGraph.from_coo_device(rows.to(torch.int32), cols.to(torch.int32), vals, n, n)
    g.transpose()
    return g


class MGCN(FusedEvalMixin, GeneralRecommender):
    def __init__(self, config, dataset):
        super().__init__(config, dataset)
        self.cl_loss = config['cl_loss']
        self.n_ui_layers = config['n_ui_layers']
        self.embedding_dim = config['embedding_size']
        self.knn_k = config['knn_k']
        self.n_layers = config['n_layers']
        self.reg_weight = config['reg_weight']
        self.tau = 0.5

        self.interaction_matrix = dataset.inter_matrix(form='coo').astype(np.float32)
        self.norm_adj, self.R = mgcn_norm_graphs(self.interaction_matrix, self.n_users, self.n_items, self.device)

        self.user_embedding = nn.Embedding(self.n_users, self.embedding_dim)
        self.item_id_embedding = nn.Embedding(self.n_items, self.embedding_dim)
        nn.init.xavier_uniform_(self.user_embedding.weight)
        nn.init.xavier_uniform_(self.item_id_embedding.weight)
        dataset_path = os.path.abspath(config['data_path'] + config['dataset'])
        if self.v_feat is not None:
            self.image_embedding = nn.Embedding.from_pretrained(self.v_feat, freeze=False)
            self.image_original_adj = knn_sym_graph(self.image_embedding.weight, self.knn_k,
                                                    os.path.join(dataset_path, 'image_adj_{}_True.pt'.format(self.knn_k)))
            self.image_trs = nn.Linear(self.v_feat.shape[1], self.embedding_dim)
        if self.t_feat is not None:
            self.text_embedding = nn.Embedding.from_pretrained(self.t_feat, freeze=False)
            self.text_original_adj = knn_sym_graph(self.text_embedding.weight, self.knn_k,
                                                   os.path.join(dataset_path, 'text_adj_{}_True.pt'.format(self.knn_k)))
            self.text_trs = nn.Linear(self.t_feat.shape[1], self.embedding_dim)

        d = self.embedding_dim
        self.query_common = nn.Sequential(nn.Linear(d, d), nn.Tanh(), nn.Linear(d, 1, bias=False))
        self.gate_v = nn.Sequential(nn.Linear(d, d), nn.Sigmoid())
        self.gate_t = nn.Sequential(nn.Linear(d, d), nn.Sigmoid())
        self.gate_image_prefer = nn.Sequential(nn.Linear(d, d), nn.Sigmoid())
        self.gate_text_prefer = nn.Sequential(nn.Linear(d, d), nn.Sigmoid())

    def pre_epoch_processing(self):
        pass

    @staticmethod
    def _gate(seq, x):
        """Sequential(Linear(64, 64), Sigmoid) on the MFMA projection kernel."""
        return torch.sigmoid(hip_ops.linear(x.contiguous(), seq[0].weight, seq[0].bias))

    def _query(self, x):
        h = torch.tanh(hip_ops.linear(x.contiguous(), self.query_common[0].weight, self.query_common[0].bias))
        return self.query_common[2](h)

    def forward(self, adj, train=False):
        image_feats = hip_ops.linear(self.image_embedding.weight, self.image_trs.weight, self.image_trs.bias)
        text_feats = hip_ops.linear(self.text_embedding.weight, self.text_trs.weight, self.text_trs.bias)
        item_w = self.item_id_embedding.weight
        image_item = item_w * self._gate(self.gate_v, image_feats)
        text_item = item_w * self._gate(self.gate_t, text_feats)
        content = hip_ops.lightgcn_mean(adj, torch.cat([self.user_embedding.weight, item_w], dim=0),
                                        self.n_ui_layers)
        for _ in range(self.n_layers):
            image_item = hip_ops.spmm(self.image_original_adj, image_item)
        image_embeds = torch.cat([hip_ops.spmm(self.R, image_item), image_item], dim=0)
        for _ in range(self.n_layers):
            text_item = hip_ops.spmm(self.text_original_adj, text_item)
        text_embeds = torch.cat([hip_ops.spmm(self.R, text_item), text_item], dim=0)

        w = torch.softmax(torch.cat([self._query(image_embeds), self._query(text_embeds)], dim=-1), dim=-1)
        common = w[:, 0].unsqueeze(1) * image_embeds + w[:, 1].unsqueeze(1) * text_embeds
        sep_image = self._gate(self.gate_image_prefer, content) * (image_embeds - common)
        sep_text = self._gate(self.gate_text_prefer, content) * (text_embeds - common)
        side = (sep_image + sep_text + common) / 3
        out = content + side
        users, items = out[:self.n_users], out[self.n_users:]
        if train:
            return users, items, side, content
        return users, items

    def eval_embeddings(self):
        return self.forward(self.norm_adj)

    def calculate_loss(self, interaction):
        users, pos_items, neg_items = interaction[0], interaction[1], interaction[2]
        ua, ia, side, content = self.forward(self.norm_adj, train=True)
        ua, ia = ua.contiguous(), ia.contiguous()
        mf_loss = hip_ops.bpr_loss(ua, ia, users, pos_items, neg_items)
        reg = hip_ops.rows_reg(((ua, users), (ia, pos_items), (ia, neg_items)), hip_ops.ROWS_REG_SQUARED, 0.5 / self.batch_size)
        nu = self.n_users
        cl = hip_ops.infonce(side[nu:].contiguous(), content[nu:].contiguous(), pos_items, 0.2) + \
            hip_ops.infonce(side[:nu].contiguous(), content[:nu].contiguous(), users, 0.2)
        return mf_loss + self.reg_weight * reg + self.cl_loss * cl
