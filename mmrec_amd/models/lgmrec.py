"""LGMRec on the HIP hot path (reference: models/lgmrec.py).

init   : binary user-item matrix R as a CSR (+ transpose for the backward), D^-1/2 A D^-1/2 (host, fp64 -> fp32,
         lgmrec.py:70-84), 1 / (degree + 1e-7) per node; modal feature tables are frozen (lgmrec.py:55,59)
forward: CGE = fused LightGCN layer mean; MGE = modal projection X W on the fp32 MFMA GEMM (W is stored [F, 64] as in
         the reference: transposed view per step), R X on the CSR SpMM, n_mm propagation layers on the same kernel;
         hyperedge assignment X V (F x 4: a skinny library GEMM), R (X V) on the SpMM (4 columns padded to a 64-float
         row), Gumbel-softmax and the 4-hyperedge HGNN products in torch (I x 4 and U x 4 operands)
loss   : fused BPR + EmbLoss on the batch rows; the hypergraph contrastive term scores the batch against EVERY
         user / item (B x N logits, lgmrec.py:157-164), not in-batch: torch matmul
eval   : fused score + mask + top-K.  The reference draws fresh Gumbel noise in every forward, i.e. once per
         evaluation batch (lgmrec.py:196-200); here once per evaluate (the propagation is cached across batches).
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from mmrec_amd import hip_ops
from mmrec_amd.graph import norm_adj_graph, unique_edges
from mmrec_amd.models._base import FusedEvalMixin, GeneralRecommender, emb_loss_rows


class HGNNLayer(nn.Module):
    def __init__(self, n_hyper_layer):
        super().__init__()
        self.h_layer = n_hyper_layer

    def forward(self, i_hyper, u_hyper, embeds):
        i_ret = embeds
        for _ in range(self.h_layer):
            lat = torch.mm(i_hyper.T, i_ret)
            i_ret = torch.mm(i_hyper, lat)
            u_ret = torch.mm(u_hyper, lat)
        return u_ret, i_ret


class LGMRec(FusedEvalMixin, GeneralRecommender):
    graph_capturable = False      # Gumbel noise and dropout are drawn inside the step
    eval_tables_deterministic = False   # ... and inside the evaluation forward: recomputed per evaluation pass like the reference

    def __init__(self, config, dataset):
        super().__init__(config, dataset)
        self.embedding_dim = config['embedding_size']
        self.feat_embed_dim = config['feat_embed_dim']
        self.cf_model = config['cf_model']
        self.n_mm_layer = config['n_mm_layers']
        self.n_ui_layers = config['n_ui_layers']
        self.n_hyper_layer = config['n_hyper_layer']
        self.hyper_num = config['hyper_num']
        self.keep_rate = config['keep_rate']
        self.alpha = config['alpha']
        self.cl_weight = config['cl_weight']
        self.reg_weight = config['reg_weight']
        self.tau = 0.2
        self.n_nodes = self.n_users + self.n_items
        self.hgnnLayer = HGNNLayer(self.n_hyper_layer)

        self.interaction_matrix = dataset.inter_matrix(form='coo').astype(np.float32)
        eu, ei = unique_edges(self.interaction_matrix.row, self.interaction_matrix.col, self.n_items)
        self.adj = hip_ops.CsrGraph.from_coo_host(np.stack([eu, ei]), np.ones(eu.shape[0], np.float32),
                                                  self.n_users, self.n_items, self.device)
        self.adj.transpose()
        self.norm_adj = norm_adj_graph(self.interaction_matrix, self.n_users, self.n_items, self.device)
        deg = np.concatenate([np.bincount(eu, minlength=self.n_users), np.bincount(ei, minlength=self.n_items)])
        self.num_inters = torch.from_numpy((1.0 / (deg.astype(np.float64) + 1e-7)).astype(np.float32)).view(-1, 1).to(self.device)

        self.user_embedding = nn.Embedding(self.n_users, self.embedding_dim)
        self.item_id_embedding = nn.Embedding(self.n_items, self.embedding_dim)
        nn.init.xavier_uniform_(self.user_embedding.weight)
        nn.init.xavier_uniform_(self.item_id_embedding.weight)
        xav = lambda r, c: nn.Parameter(nn.init.xavier_uniform_(torch.zeros(r, c)))
        self.image_embedding = nn.Embedding.from_pretrained(self.v_feat, freeze=True)
        self.item_image_trs = xav(self.v_feat.shape[1], self.feat_embed_dim)
        self.v_hyper = xav(self.v_feat.shape[1], self.hyper_num)
        self.text_embedding = nn.Embedding.from_pretrained(self.t_feat, freeze=True)
        self.item_text_trs = xav(self.t_feat.shape[1], self.feat_embed_dim)
        self.t_hyper = xav(self.t_feat.shape[1], self.hyper_num)

    def cge(self):
        ego = torch.cat((self.user_embedding.weight, self.item_id_embedding.weight), dim=0)
        if self.cf_model == 'mf':
            return ego
        return hip_ops.lightgcn_mean(self.norm_adj, ego, self.n_ui_layers)

    def mge(self, which='v'):
        feat, trs = ((self.image_embedding.weight, self.item_image_trs) if which == 'v'
                     else (self.text_embedding.weight, self.item_text_trs))
        item_feats = hip_ops.linear(feat, trs.t().contiguous(), None)
        user_feats = hip_ops.spmm(self.adj, item_feats) * self.num_inters[:self.n_users]
        x = torch.cat([user_feats, item_feats], dim=0)
        for _ in range(self.n_mm_layer):
            x = hip_ops.spmm(self.norm_adj, x)
        return x

    def _hyper(self, feat, w):
        i_h = torch.mm(feat, w)
        u_h = hip_ops.spmm(self.adj, i_h)
        return (F.gumbel_softmax(i_h, self.tau, dim=1, hard=False), F.gumbel_softmax(u_h, self.tau, dim=1, hard=False))

    def forward(self):
        iv_hyper, uv_hyper = self._hyper(self.image_embedding.weight, self.v_hyper)
        it_hyper, ut_hyper = self._hyper(self.text_embedding.weight, self.t_hyper)
        cge_embs = self.cge()
        lge_embs = cge_embs + F.normalize(self.mge('v')) + F.normalize(self.mge('t'))
        drop = lambda x: F.dropout(x, 1 - self.keep_rate, self.training)
        uv, iv = self.hgnnLayer(drop(iv_hyper), drop(uv_hyper), cge_embs[self.n_users:])
        ut, it = self.hgnnLayer(drop(it_hyper), drop(ut_hyper), cge_embs[self.n_users:])
        ghe_embs = torch.cat([uv, iv], dim=0) + torch.cat([ut, it], dim=0)
        all_embs = lge_embs + self.alpha * F.normalize(ghe_embs)
        return all_embs[:self.n_users], all_embs[self.n_users:], [uv, iv, ut, it]

    def eval_embeddings(self):
        u, i, _ = self.forward()
        return u, i

    def ssl_triple_loss(self, emb1, emb2, all_emb):
        n1, n2, na = F.normalize(emb1), F.normalize(emb2), F.normalize(all_emb)
        pos_score = torch.exp(torch.mul(n1, n2).sum(dim=1) / self.tau)
        ttl_score = torch.exp(torch.matmul(n1, na.T) / self.tau).sum(dim=1)
        return -torch.log(pos_score / ttl_score).sum()

    def calculate_loss(self, interaction):
        ua, ia, (uv, iv, ut, it) = self.forward()
        users, pos_items, neg_items = interaction[0], interaction[1], interaction[2]
        ua, ia = ua.contiguous(), ia.contiguous()
        bpr = hip_ops.bpr_loss(ua, ia, users, pos_items, neg_items)
        hcl = self.ssl_triple_loss(uv[users], ut[users], ut) + self.ssl_triple_loss(iv[pos_items], it[pos_items], it)
        reg = emb_loss_rows(((ua, users), (ia, pos_items), (ia, neg_items)), neg_items.shape[0])
        return bpr + self.cl_weight * hcl + self.reg_weight * reg
