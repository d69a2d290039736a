"""SMORE on the HIP hot path (reference: models/smore.py).

init   : MGCN's graphs (float32 D^-1/2 A D^-1/2 without epsilon, its user-rows block R, per-modality
         similarity-weighted kNN graphs from the fused score+top-K kernel, smore.py:39-75,162-184) with
         separate image / text k, plus the max-pooled fusion graph (smore.py:139-160) built on the device
forward: modal projections on the fp32 MFMA GEMM; the spectrum step (smore.py:193-211) is written as the
         real DFT matrices torch.fft.rfft / irfft (norm='ortho') stand for -- [I, 64] x [64, 33] GEMMs,
         no FFT library in the path; LightGCN layer mean and the three item-item views + their R
         products on the CSR SpMM (transposed CSR for the backward: none of these graphs is symmetric);
         64 -> 64 gate / query layers on the MFMA projection kernels
loss   : fused BPR + fused gather-norm regulariser + two fused in-batch InfoNCE terms
eval   : fused score + mask + top-K
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from mmrec_amd import hip_ops
from mmrec_amd.models._base import FusedEvalMixin, GeneralRecommender
from mmrec_amd.models.mgcn import knn_sym_graph, mgcn_norm_graphs


def rdft_matrices(d, device):
    """rfft(x) = x @ C + i x @ S, irfft(Re, Im) = Re @ Ci + Im @ Si for norm='ortho', n = d (even);
    the DC / Nyquist imaginary parts do not contribute to the inverse, as in a C2R transform."""
    n = torch.arange(d, dtype=torch.float64).unsqueeze(1)
    k = torch.arange(d // 2 + 1, dtype=torch.float64).unsqueeze(0)
    ang = 2.0 * math.pi * n * k / d
    s = 1.0 / math.sqrt(d)
    wk = torch.full((d // 2 + 1, 1), 2.0, dtype=torch.float64)
    wk[0, 0] = wk[-1, 0] = 1.0
    Si = -wk * torch.sin(ang).t() * s
    Si[0, :] = 0.0
    Si[-1, :] = 0.0
    mats = (torch.cos(ang) * s, -torch.sin(ang) * s, wk * torch.cos(ang).t() * s, Si)
    return tuple(m.to(torch.float32).to(device) for m in mats)


def max_pool_fusion(a, b, n_items):
    """Union of two kNN CsrGraphs' edges, value = max over the graphs that hold the edge
    (smore.py:139-160) -> CsrGraph sorted row-major, with its transpose."""
    dev = a.rowptr.device

    def coo(g):
        rows = torch.repeat_interleave(torch.arange(g.n_rows, device=dev),
                                       torch.diff(g.rowptr.to(torch.int64)))
        return rows * n_items + g.colidx.to(torch.int64), g.vals
    ka, va = coo(a)
    kb, vb = coo(b)
    keys, inv = torch.unique(torch.cat([ka, kb]), return_inverse=True)
    neg = torch.full((keys.numel(),), float('-inf'), device=dev)
    pa = neg.clone().scatter_reduce_(0, inv[:ka.numel()], va, reduce='amax')
    pb = neg.clone().scatter_reduce_(0, inv[ka.numel():], vb, reduce='amax')
    rows = torch.div(keys, n_items, rounding_mode='floor')
    g = hip_ops.CsrGraph.from_coo_device(rows.to(torch.int32), (keys - rows * n_items).to(torch.int32),
                                         torch.maximum(pa, pb).contiguous(), n_items, n_items)
    g.transpose()
    return g


class SMORE(FusedEvalMixin, GeneralRecommender):
    def __init__(self, config, dataset):
        super().__init__(config, dataset)
        self.sparse = True
        self.cl_loss = config['cl_loss']
        self.n_ui_layers = config['n_ui_layers']
        self.embedding_dim = config['embedding_size']
        self.n_layers = config['n_layers']
        self.reg_weight = config['reg_weight']
        self.image_knn_k = config['image_knn_k']
        self.text_knn_k = config['text_knn_k']
        self.dropout_rate = config['dropout_rate']

        self.interaction_matrix = dataset.inter_matrix(form='coo').astype(np.float32)
        self.norm_adj, self.R = mgcn_norm_graphs(self.interaction_matrix, self.n_users, self.n_items, self.device)

        self.user_embedding = nn.Embedding(self.n_users, self.embedding_dim)
        self.item_id_embedding = nn.Embedding(self.n_items, self.embedding_dim)
        nn.init.xavier_uniform_(self.user_embedding.weight)
        nn.init.xavier_uniform_(self.item_id_embedding.weight)
        dataset_path = os.path.abspath(config['data_path'] + config['dataset'])
        self.image_embedding = nn.Embedding.from_pretrained(self.v_feat, freeze=False)
        self.image_original_adj = knn_sym_graph(
            self.image_embedding.weight, self.image_knn_k,
            os.path.join(dataset_path, 'image_adj_{}_{}.pt'.format(self.image_knn_k, self.sparse)))
        self.text_embedding = nn.Embedding.from_pretrained(self.t_feat, freeze=False)
        self.text_original_adj = knn_sym_graph(
            self.text_embedding.weight, self.text_knn_k,
            os.path.join(dataset_path, 'text_adj_{}_{}.pt'.format(self.text_knn_k, self.sparse)))
        self.fusion_adj = max_pool_fusion(self.image_original_adj, self.text_original_adj, self.n_items)
        self.image_trs = nn.Linear(self.v_feat.shape[1], self.embedding_dim)
        self.text_trs = nn.Linear(self.t_feat.shape[1], self.embedding_dim)

        d = self.embedding_dim
        self.query_v = nn.Sequential(nn.Linear(d, d), nn.Tanh(), nn.Linear(d, d, bias=False))
        self.query_t = nn.Sequential(nn.Linear(d, d), nn.Tanh(), nn.Linear(d, d, bias=False))
        for name in ('gate_v', 'gate_t', 'gate_f', 'gate_image_prefer', 'gate_text_prefer', 'gate_fusion_prefer'):
            setattr(self, name, nn.Sequential(nn.Linear(d, d), nn.Sigmoid()))
        self.image_complex_weight = nn.Parameter(torch.randn(1, d // 2 + 1, 2, dtype=torch.float32))
        self.text_complex_weight = nn.Parameter(torch.randn(1, d // 2 + 1, 2, dtype=torch.float32))
        self.fusion_complex_weight = nn.Parameter(torch.randn(1, d // 2 + 1, 2, dtype=torch.float32))
        self._dft = None

    def pre_epoch_processing(self):
        pass

    @staticmethod
    def _gate(seq, x):
        return torch.sigmoid(hip_ops.linear(x.contiguous(), seq[0].weight, seq[0].bias))

    @staticmethod
    def _query(seq, x):
        h = torch.tanh(hip_ops.linear(x.contiguous(), seq[0].weight, seq[0].bias))
        return hip_ops.linear(h, seq[2].weight, None)

    def spectrum_convolution(self, image_embeds, text_embeds):
        """Uni-modal spectral filters and the cross-modal product filter (smore.py:193-211)."""
        if self._dft is None or self._dft[0].device != image_embeds.device:
            self._dft = rdft_matrices(image_embeds.shape[1], image_embeds.device)
        C, S, Ci, Si = self._dft
        ir, ii = image_embeds @ C, image_embeds @ S
        tr, ti = text_embeds @ C, text_embeds @ S

        def filt(re, im, w):
            wr, wi = w[0, :, 0], w[0, :, 1]
            return (re * wr - im * wi) @ Ci + (re * wi + im * wr) @ Si
        fr, fi = tr * ir - ti * ii, tr * ii + ti * ir
        return (filt(ir, ii, self.image_complex_weight), filt(tr, ti, self.text_complex_weight),
                filt(fr, fi, self.fusion_complex_weight))

    def _view(self, graph, x):
        for _ in range(self.n_layers):
            x = hip_ops.spmm(graph, x)
        return torch.cat([hip_ops.spmm(self.R, x), x], dim=0)

    def forward(self, adj, train=False):
        image_feats = hip_ops.linear(self.image_embedding.weight, self.image_trs.weight, self.image_trs.bias)
        text_feats = hip_ops.linear(self.text_embedding.weight, self.text_trs.weight, self.text_trs.bias)
        image_conv, text_conv, fusion_conv = self.spectrum_convolution(image_feats, text_feats)
        item_w = self.item_id_embedding.weight
        content = hip_ops.lightgcn_mean(adj, torch.cat([self.user_embedding.weight, item_w], dim=0),
                                        self.n_ui_layers)
        image_embeds = self._view(self.image_original_adj, item_w * self._gate(self.gate_v, image_conv))
        text_embeds = self._view(self.text_original_adj, item_w * self._gate(self.gate_t, text_conv))
        fusion_embeds = self._view(self.fusion_adj, item_w * self._gate(self.gate_f, fusion_conv))

        agg_image = torch.softmax(self._query(self.query_v, fusion_embeds), dim=-1) * image_embeds
        agg_text = torch.softmax(self._query(self.query_t, fusion_embeds), dim=-1) * text_embeds
        prefer = [self._gate(g, content) for g in (self.gate_image_prefer, self.gate_text_prefer, self.gate_fusion_prefer)]
        prefer = [F.dropout(x, self.dropout_rate, self.training) for x in prefer]
        side = (prefer[0] * agg_image + prefer[1] * agg_text + prefer[2] * fusion_embeds) / 3
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
