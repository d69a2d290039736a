"""PGL (mode 'local') on the HIP hot path (reference: models/pgl.py).

init   : FREEDOM's graphs -- normalised u-i adjacency, per-edge pruning weights, frozen kNN item-item
         graph shared through the same `mm_adj_freedomdsp_{k}_{10w}.pt` cache (pgl.py:38-74)
epoch  : degree-sensitive edge pruning to 30 % of the edges, re-normalised, CSR rebuilt on the device
         (pgl.py:151-166)
forward: both modal projections on the fp32 MFMA GEMM feed the graph (pgl.py:188-213): rows are
         [image | text] = 128 floats, so the LightGCN layer mean, the item-item SpMM (+ fused residual)
         and the fused BPR all run at row width 128
loss   : fused BPR + reg_weight * mean of two InfoNCE terms between two dropout views of the batch rows
         (pgl.py:233-250; `reg_weight` is 0 in the reference's PGL.yaml -- the term is then skipped, it
         contributes exactly zero loss and zero gradient)
eval   : fused score + mask + top-K at row width 128
mode 'global' needs `sparsesvd` (third-party, absent from the reference tree and this image): not built.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from mmrec_amd import hip_ops
from mmrec_amd.graph import norm_adj_graph
from mmrec_amd.models._base import FusedEvalMixin, GeneralRecommender
from mmrec_amd.models.freedom import load_or_build_mm_adj


def global_subgraph(norm_adj, q, device, block=None, tol=1e-3):
    """PGL's `global_subgraph_extraction` (pgl.py:138-153): rank-(q/4) spectral sub-graph
        S = U[:, :m] diag(s[:m] * s[q-m:]) V[:, :m]^T,  m = int(0.25 q),  entries with |S| < 1e-3 dropped,
    from the q leading singular triplets of the normalised adjacency.  The reference takes them from the third-party
    `sparsesvd` package (SVDLIBC's Lanczos `las2`; not vendored, not pinned in requirements.txt and absent here: parity with
    ITS rounding is unpinned).  Restated with its published contract -- the q largest singular values in DESCENDING order with
    their vectors -- on ARPACK (`scipy.sparse.linalg.svds`): S is the same matrix for any orthonormal basis of a repeated
    singular value's subspace (the symmetric bipartite adjacency has every singular value twice; U and V rotate together and
    the two weights of a pair are equal), so only entries within rounding of the 1e-3 cut can differ between solvers.
    That argument needs the three cuts -- after triplet m, before triplet q - m, after triplet q -- to fall BETWEEN pairs, which
    an odd m (embedding_size 36: m = 9) or a pair the solver returned one copy of breaks: two more triplets than needed are
    requested, the three boundaries are looked at, and a cut pair is reported (the sub-graph then depends on the solver's basis
    of that pair beyond the 1e-3 cut; round-5 advice).
    The [N, N] product is formed `block` rows at a time (default: ~1 GB of fp32 per block; the reference materialises it
    densely: 2.8 GB at Amazon-Baby size)."""
    import logging
    import scipy.sparse as sp
    from scipy.sparse.linalg import svds
    idx, val = norm_adj.to_coo_host()
    n = norm_adj.n_rows
    a = sp.csc_matrix((val.astype(np.float64), (idx[0], idx[1])), shape=(n, n))
    k = min(int(q), n - 2)
    k_req = min(k + 2, n - 2)
    u, s_, vt = svds(a, k=k_req, which='LM', tol=1e-10, v0=np.ones(n))  # ascending singular values, deterministic start vector
    order = np.argsort(-s_, kind='stable')
    u, s_, vt = u[:, order], s_[order], vt[order]
    m = int(0.25 * q)
    cut = [j for j in (m, k - m, k) if 0 < j < k_req and abs(s_[j - 1] - s_[j]) <= 1e-7 * max(abs(s_[j - 1]), 1e-30)]
    if cut:
        logging.getLogger().warning('PGL global sub-graph: the cut after singular triplet(s) %s of %d falls inside a pair of equal '
                                    'singular values (m = %d): the sub-graph depends on the solver\'s basis of that pair -- parity '
                                    'with the reference\'s sparsesvd is not expected beyond the 1e-3 cut' % (cut, k, m))
    u, s_, vt = u[:, :k], s_[:k], vt[:k]
    if block is None:
        block = int(max(16, min(2048, (1 << 28) // max(n, 1))))
    w = s_[:m] * s_[k - m:k]
    left = (u[:, :m] * w[None, :]).astype(np.float32)
    right = vt[:m].astype(np.float32)
    rows, cols, vals = [], [], []
    for r0 in range(0, n, block):
        blk = left[r0:r0 + block] @ right
        rr, cc = np.nonzero(np.abs(blk) >= tol)
        rows.append(rr + r0), cols.append(cc), vals.append(blk[rr, cc])
    rows, cols, vals = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)
    g = hip_ops.CsrGraph.from_coo_host(np.stack([rows, cols]), vals.astype(np.float32), n, n, device)
    g.transpose()          # not symmetric after the cut in floating point: the backward takes the explicit transpose
    return g


class PGL(FusedEvalMixin, GeneralRecommender):
    def __init__(self, config, dataset):
        super().__init__(config, dataset)
        self.mode = config['mode']
        if self.mode not in ('local', 'global'):
            raise ValueError("PGL mode %r: 'local' (edge-pruned sub-graph, pgl.py:160-182) or 'global' (spectral sub-graph, "
                             "pgl.py:138-153)" % (self.mode,))
        self.embedding_dim = config['embedding_size']
        self.feat_embed_dim = config['feat_embed_dim']
        self.knn_k = config['knn_k']
        self.lambda_coeff = config['lambda_coeff']
        self.n_layers = config['n_mm_layers']
        self.n_ui_layers = config['n_ui_layers']
        self.reg_weight = config['reg_weight']
        self.mm_image_weight = config['mm_image_weight']
        self.dropout = config['dropout']
        self.n_nodes = self.n_users + self.n_items

        self.interaction_matrix = dataset.inter_matrix(form='coo').astype(np.float32)
        self.norm_adj = norm_adj_graph(self.interaction_matrix, self.n_users, self.n_items, self.device)
        self.sub_graph = None
        rows = torch.from_numpy(self.interaction_matrix.row.astype(np.int64))
        cols = torch.from_numpy(self.interaction_matrix.col.astype(np.int64))
        self.edge_indices = torch.stack([rows, cols]).to(self.device)
        self.edge_values = hip_ops.edge_norm_values(self.edge_indices[0].contiguous(),
                                                    self.edge_indices[1].contiguous(), self.n_users, self.n_items)

        self.user_text = nn.Embedding(self.n_users, self.embedding_dim)
        self.user_image = nn.Embedding(self.n_users, self.embedding_dim)
        nn.init.xavier_uniform_(self.user_image.weight)
        nn.init.xavier_uniform_(self.user_text.weight)
        self.image_embedding = nn.Embedding.from_pretrained(self.v_feat, freeze=False)
        self.image_trs = nn.Linear(self.v_feat.shape[1], self.feat_embed_dim)
        self.text_embedding = nn.Embedding.from_pretrained(self.t_feat, freeze=False)
        self.text_trs = nn.Linear(self.t_feat.shape[1], self.feat_embed_dim)
        self.mm_adj = load_or_build_mm_adj(config, self.v_feat, self.t_feat, self.knn_k, self.mm_image_weight,
                                           self.n_items, self.device)
        if self.mode == 'global':
            self.sub_graph = global_subgraph(self.norm_adj, self.embedding_dim, self.device)

    def pre_epoch_processing(self):
        if self.mode == 'global':     # the spectral sub-graph is fixed (pgl.py:160: only 'local' re-samples per epoch)
            return
        keep_len = int(self.edge_values.size(0) * 0.3)
        self.set_kept_edges(torch.multinomial(self.edge_values, keep_len))

    def set_kept_edges(self, keep_idx):
        """Rebuild the pruned, re-normalised sub-graph from sampled edge ids (injectable for parity runs)."""
        kept = self.edge_indices[:, keep_idx]
        self.sub_graph = hip_ops.bipartite_graph_from_edges(kept[0].contiguous(), kept[1].contiguous(),
                                                            self.n_users, self.n_items)

    def forward(self, adj):
        image_feats = hip_ops.linear(self.image_embedding.weight, self.image_trs.weight, self.image_trs.bias)
        text_feats = hip_ops.linear(self.text_embedding.weight, self.text_trs.weight, self.text_trs.bias)
        item_embeds = torch.cat([F.normalize(image_feats), F.normalize(text_feats)], dim=1)
        user_embeds = torch.cat([self.user_image.weight, self.user_text.weight], dim=1)
        mean = hip_ops.lightgcn_mean(adj, torch.cat((user_embeds, item_embeds), dim=0), self.n_ui_layers)
        u_g, i_g = mean[:self.n_users], mean[self.n_users:]
        h = item_embeds
        if self.n_layers == 0:
            return u_g, i_g + h
        for _ in range(self.n_layers - 1):
            h = hip_ops.spmm(self.mm_adj, h)
        return u_g, hip_ops.spmm(self.mm_adj, h, Z=i_g)     # i_g + M h in one launch

    def eval_embeddings(self):
        return self.forward(self.norm_adj)

    @staticmethod
    def InfoNCE(view1, view2, temperature):
        view1, view2 = F.normalize(view1, dim=1), F.normalize(view2, dim=1)
        pos_score = torch.exp((view1 * view2).sum(dim=-1) / temperature)
        ttl_score = torch.exp(torch.matmul(view1, view2.transpose(0, 1)) / temperature).sum(dim=1)
        return torch.mean(-torch.log(pos_score / ttl_score))

    def calculate_loss(self, interaction):
        users, pos_items = interaction[0], interaction[1]
        ua, ia = self.forward(self.sub_graph)
        ua, ia = ua.contiguous(), ia.contiguous()
        loss = hip_ops.bpr_loss(ua, ia, users, pos_items, interaction[2])
        if not self.reg_weight:
            if self.training:     # the reference still draws its four dropout masks (then multiplies the term by 0):
                with torch.no_grad():                 # keep the generator stream of a seeded run aligned with it
                    for width in (ua.shape[1], ua.shape[1], ia.shape[1], ia.shape[1]):
                        F.dropout(torch.empty(users.shape[0], width, device=ua.device), self.dropout, True)
            return loss
        u_g, p_g = ua[users], ia[pos_items]
        drop = lambda x: F.dropout(x, self.dropout, self.training)
        cl_loss = (self.InfoNCE(drop(u_g), drop(u_g), 0.2) + self.InfoNCE(drop(p_g), drop(p_g), 0.2)) / 2
        return loss + self.reg_weight * cl_loss
