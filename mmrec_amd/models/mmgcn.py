"""MMGCN on the HIP hot path (reference: models/mmgcn.py, which needs torch_geometric).

Per modality a 3-layer GCN: `conv(x) = mean_{j in N(i)} (x W)_j` (PyG MessagePassing(aggr='mean'),
messages flow along the symmetric user<->item edge list) + Linear/LeakyReLU stacks.  The mean
aggregation is the HIP CSR SpMM with values 1/in-degree at row widths 256 (visual latent), 384 (text
features) and 64 -- no PyG, no scatter; its backward uses the transposed CSR.  The dense transforms are
plain library GEMMs (torch -> hipBLASLt).  Reference quirks kept on purpose:
`concate = 'False'` is a truthy string, so the concat branch is ON; `id_embedding` and `preference`
are plain tensors, not Parameters, hence never trained; evaluation reuses the last training forward.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from mmrec_amd import hip_ops
from mmrec_amd.graph import relabel_graph
from mmrec_amd.models._base import FusedEvalMixin, GeneralRecommender, RelabelledIdsMixin


def mean_aggregation_graph(inter_coo, n_users, n_items, device):
    """D_in^-1 A over the edge list cat(edges, flipped edges) of mmgcn.py:41-44 (duplicates counted),
    rows = message targets.  Not symmetric in its values: the backward transposes it once."""
    rows = np.concatenate([inter_coo.row, inter_coo.col + n_users]).astype(np.int64)   # sources
    cols = np.concatenate([inter_coo.col + n_users, inter_coo.row]).astype(np.int64)   # targets
    n = n_users + n_items
    deg = np.bincount(cols, minlength=n).astype(np.float32)
    val = (1.0 / np.maximum(deg, 1.0))[cols].astype(np.float32)
    g = hip_ops.CsrGraph.from_coo_host(np.stack([cols, rows]), val, n, n, device)
    g.transpose()
    return g


class _Conv(nn.Module):
    """BaseModel: x W then mean aggregation (weight init: U(+-1/sqrt(in)) then xavier_normal_)."""

    def __init__(self, dim):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(dim, dim))
        bound = 1.0 / math.sqrt(dim)
        self.weight.data.uniform_(-bound, bound)

    def forward(self, x, graph):
        d = self.weight.shape[0]
        if d % 64 == 0 and (d == 64 or d % 32 == 0):     # x @ W on the fp32 MFMA kernels (64 / 256 / 384 wide)
            return hip_ops.spmm(graph, hip_ops.linear(x.contiguous(), self.weight.t().contiguous(), None))
        return hip_ops.spmm(graph, torch.matmul(x, self.weight))


SLICED_WIDE_MLP = True


def _lin64(layer, x):
    """nn.Linear with 64 outputs and an input width that is a multiple of 4 -> hip_ops.linear."""
    if layer.out_features == hip_ops.EMB_DIM and layer.in_features % 4 == 0:
        return hip_ops.linear(x.contiguous(), layer.weight, layer.bias)
    if layer.out_features % 64 == 0 and layer.in_features % 32 == 0:      # the 4096 -> 256 MLP
        if SLICED_WIDE_MLP and layer.in_features >= hip_ops.LINEAR_SPLIT_MIN_F and hip_ops.LINEAR_F16X3:
            # 64 outputs at a time on the split-operand kernels (X streamed four times at 23 us each) instead of the 128 x 128
            # fp32-MFMA GEMM, which fills 112 of 256 CUs at 7,050 x 256 outputs (287 us)
            x = x.contiguous()
            return torch.cat([hip_ops.linear(x, layer.weight[z:z + 64], None if layer.bias is None else layer.bias[z:z + 64])
                              for z in range(0, layer.out_features, 64)], dim=1)
        return hip_ops.linear(x.contiguous(), layer.weight, layer.bias)
    return layer(x)


class GCN(nn.Module):
    def __init__(self, num_user, dim_feat, dim_id, dim_latent, device):
        super().__init__()
        self.dim_latent = dim_latent
        d = dim_latent if dim_latent else dim_feat
        self.preference = nn.init.xavier_normal_(torch.rand((num_user, d))).to(device)   # not a Parameter
        if dim_latent:
            self.MLP = nn.Linear(dim_feat, dim_latent)
        self.conv_embed_1 = _Conv(d)
        nn.init.xavier_normal_(self.conv_embed_1.weight)
        self.linear_layer1 = nn.Linear(d, dim_id)
        nn.init.xavier_normal_(self.linear_layer1.weight)
        self.g_layer1 = nn.Linear(d + dim_id, dim_id)
        nn.init.xavier_normal_(self.g_layer1.weight)
        self.conv_embed_2 = _Conv(dim_id)
        nn.init.xavier_normal_(self.conv_embed_2.weight)
        self.linear_layer2 = nn.Linear(dim_id, dim_id)
        nn.init.xavier_normal_(self.linear_layer2.weight)
        self.g_layer2 = nn.Linear(dim_id + dim_id, dim_id)
        self.conv_embed_3 = _Conv(dim_id)
        nn.init.xavier_normal_(self.conv_embed_3.weight)
        self.linear_layer3 = nn.Linear(dim_id, dim_id)
        nn.init.xavier_normal_(self.linear_layer3.weight)
        self.g_layer3 = nn.Linear(dim_id + dim_id, dim_id)

    def forward(self, features, id_embedding, graph):
        temp = _lin64(self.MLP, features) if self.dim_latent else features
        x = hip_ops.row_normalize(torch.cat((self.preference, temp), dim=0))      # F.normalize (mmgcn.py:167-168)
        for conv, lin, gl in ((self.conv_embed_1, self.linear_layer1, self.g_layer1),
                              (self.conv_embed_2, self.linear_layer2, self.g_layer2),
                              (self.conv_embed_3, self.linear_layer3, self.g_layer3)):
            # every 64-wide layer runs on the MFMA projection kernels (forward, dW + db, dX): the
            # library's skinny dW GEMMs (contraction over 26k nodes) were 47 % of the step
            # h = leaky_relu(conv(x)), x_hat = leaky_relu(linear(x)) + id_embedding and their cat (mmgcn.py:170-173): one launch
            x = F.leaky_relu(_lin64(gl, hip_ops.cat_leaky(conv(x, graph), _lin64(lin, x), id_embedding)))
        return x


class MMGCN(RelabelledIdsMixin, FusedEvalMixin, GeneralRecommender):
    graph_capturable = True       # the step is a fixed launch sequence (~300 launches): replayed as a hipGraph by default (hip_graph_step: auto)
    relabelled_tables = {}        # config key `reorder`: MMGCN's id-indexed state (preference, id_embedding, the feature tables)
                                  # is plain tensors, not Parameters -- nothing of it is in the state_dict (mmgcn.py:55,126,139)

    def __init__(self, config, dataset):
        super().__init__(config, dataset)
        self.num_user, self.num_item = self.n_users, self.n_items
        dim_x = config['embedding_size']
        self.reg_weight = config['reg_weight']
        self.weight = torch.tensor([[1.0], [-1.0]]).to(self.device)
        inter = dataset.inter_matrix(form='coo').astype(np.float32)
        self.graph = mean_aggregation_graph(inter, self.n_users, self.n_items, self.device)
        # new key `reorder` (models/_base.py): the modality graphs' node rows -- user preferences, item features, the id
        # embedding -- live in an id space relabelled once, here; the aggregation graph keeps every row's nonzero order
        rl = self._setup_relabelling(config, self.graph)
        if rl is not None:
            self.graph = relabel_graph(self.graph, rl.node_perm_host())
            if self.v_feat is not None:
                self.v_feat = self.v_feat.index_select(0, rl.inv_i.to(self.v_feat.device))
            if self.t_feat is not None:
                self.t_feat = self.t_feat.index_select(0, rl.inv_i.to(self.t_feat.device))
        self.num_modal = 0
        if self.v_feat is not None:
            self.v_gcn = GCN(self.n_users, self.v_feat.size(1), dim_x, 256, self.device)
            self.num_modal += 1
        if self.t_feat is not None:
            self.t_gcn = GCN(self.n_users, self.t_feat.size(1), dim_x, None, self.device)
            self.num_modal += 1
        n = self.n_users + self.n_items
        self.id_embedding = nn.init.xavier_normal_(torch.rand((n, dim_x))).to(self.device)   # never trained
        self.result = nn.init.xavier_normal_(torch.rand((n, dim_x))).to(self.device)
        if rl is not None:            # the plain model's draws, row `old` at relabelled row perm[old]
            node_inv = torch.cat([rl.inv_u, self.n_users + rl.inv_i]).to(self.id_embedding.device)
            self.id_embedding = self.id_embedding.index_select(0, node_inv)
            self.result = self.result.index_select(0, node_inv)
            for gcn in (getattr(self, 'v_gcn', None), getattr(self, 't_gcn', None)):
                if gcn is not None:
                    gcn.preference = gcn.preference.index_select(0, rl.inv_u.to(gcn.preference.device))

    def _apply(self, fn, *a, **k):
        """`.to(device)` must also move the plain-tensor state the reference keeps outside Parameters."""
        out = super()._apply(fn, *a, **k)
        self.id_embedding, self.result = fn(self.id_embedding), fn(self.result)
        for gcn in (getattr(self, 'v_gcn', None), getattr(self, 't_gcn', None)):
            if gcn is not None:
                gcn.preference = fn(gcn.preference)
        return out

    def forward(self):
        rep = None
        if self.v_feat is not None:
            rep = self.v_gcn(self.v_feat, self.id_embedding, self.graph)
        if self.t_feat is not None:
            t = self.t_gcn(self.t_feat, self.id_embedding, self.graph)
            rep = t if rep is None else rep + t
        rep = rep / self.num_modal
        # kept for evaluation only (mmgcn.py:99-101); detached so that no autograd graph of the previous
        # step stays alive (its AccumulateGrad nodes would pin the eager stream and break hipGraph capture)
        self.result = rep.detach()
        return rep

    def eval_embeddings(self):
        res = self.result.detach()      # the reference evaluates the LAST TRAINING forward (mmgcn.py:99-101)
        return res[:self.n_users], res[self.n_users:]

    def calculate_loss(self, interaction):
        interaction = self._map_batch(interaction)
        users = interaction[0]
        pos, neg = interaction[1] + self.n_users, interaction[2] + self.n_users
        out = self.forward()
        # interleaved (pos, neg) pairs == BPR with -mean log sigmoid(pos - neg): the fused kernel on one table
        loss = hip_ops.bpr_loss(out, out, users, pos.contiguous(), neg.contiguous(), hip_ops.BPR_LOGSIG, 'mean')
        # reg = (id[user_t] ** 2 + id[item_t] ** 2).mean() over the interleaved [2B, 64] rows (+ mean(preference ** 2)), mmgcn.py:
        # 118-124: (2 sum ||id[u]||^2 + sum ||id[p]||^2 + sum ||id[n]||^2) / (2 B 64) -- neither id_embedding nor preference is a
        # Parameter there (or here), so the term only shifts the reported loss: one fused pass over the batch rows, the
        # preference mean computed once
        e = self.id_embedding
        reg = hip_ops.rows_reg(((e, users), (e, users), (e, pos), (e, neg)), hip_ops.ROWS_REG_SQUARED,
                               self.reg_weight / (2.0 * users.shape[0] * e.shape[1]))
        if self.v_feat is not None:
            if getattr(self, '_pref_sq_mean', None) is None or self._pref_sq_mean[0] is not self.v_gcn.preference:
                self._pref_sq_mean = (self.v_gcn.preference, float((self.v_gcn.preference ** 2).mean()) * self.reg_weight)
            return loss + reg + self._pref_sq_mean[1]
        return loss + reg
