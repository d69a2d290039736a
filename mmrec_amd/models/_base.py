"""Shared plumbing of the accelerated model plugins.

The model files in this directory keep the reference's plugin contract (class `<Name>` in
`models/<name>.py`, `__init__(config, dataloader)`, `calculate_loss`, `full_sort_predict`,
`pre_epoch_processing`) and its parameter names, so a reference `state_dict` loads
unchanged.  They can live in this package or be dropped into the reference's `src/models/`: the base
class is taken from whichever `common.abstract_recommender` is importable.
"""
try:  # inside the reference tree (cwd = src/)
    from common.abstract_recommender import GeneralRecommender  # noqa: F401
except ImportError:
    from mmrec_amd.common.abstract_recommender import GeneralRecommender  # noqa: F401

import torch

from mmrec_amd import hip_ops
from mmrec_amd.graph import mask_to_csr_device


class FusedEvalMixin:
    """`full_sort_topk`: fused score + mask + top-K on the final embeddings, with the propagation
    computed once per evaluation instead of once per eval batch (weights are frozen while evaluating;
    SURVEY.md App. C.4).  The cache is dropped whenever the module goes back to train mode."""

    _eval_cache = None
    _tables_version = 0
    # False for a plugin whose evaluation forward DRAWS random numbers (LGMRec's Gumbel noise): the reference recomputes the
    # forward in every evaluation pass, and so must such a plugin, or the generator's stream -- and the next epoch -- differs
    eval_tables_deterministic = True

    def train(self, mode=True):
        # dropped when training starts or ends; eval() -> eval() (the VALID and the TEST pass of one evaluation,
        # trainer.py:262,271: nothing was trained in between) keeps the propagated tables and the prepared candidates
        if mode or self.training or not self.eval_tables_deterministic:
            self._eval_cache = self._eval_cands = None
        return super().train(mode)

    def load_state_dict(self, *args, **kwargs):
        self._eval_cache = self._eval_cands = None      # new weights: nothing cached in eval mode describes them
        return super().load_state_dict(*args, **kwargs)

    def eval_embeddings(self):
        """-> (user_all [n_users, d], item_all [n_items, d]) used by full_sort_*; override."""
        raise NotImplementedError

    def _cached_eval_embeddings(self):
        if self.training or self._eval_cache is None:
            with torch.no_grad():
                u, i = self.eval_embeddings()
                cache = (u.contiguous(), i.contiguous())
            if self.training:
                return cache
            self._eval_cache = cache
            self._eval_cands = None
            self._tables_version += 1           # (warm evaluation: lists written under THESE tables are exact for them)
        return self._eval_cache

    _eval_cands = None

    def _cached_eval_candidates(self):
        """the cached item table with the candidate side of the top-K filter prepared once per evaluation (every batch of
        the valid AND the test loader ranks against the same frozen table); dropped with the cache"""
        u, i = self._cached_eval_embeddings()
        if self.training:
            return u, i
        if self._eval_cands is None:
            self._eval_cands = hip_ops.TopkCandidates(i)
        return u, self._eval_cands

    # A model whose tables live in a RELABELLED id space (config `reorder`, RelabelledIdsMixin) maps ids where they enter and
    # leave; for every other model these are the identity.
    relabelling = None

    def full_sort_predict(self, interaction):
        u, i = self._cached_eval_embeddings()
        rl = self.relabelling
        if rl is None:
            return torch.matmul(u[interaction[0]], i.transpose(0, 1))
        # columns back in the ORIGINAL item order (the reference Trainer indexes them with the dataset's ids)
        return torch.matmul(u[rl.perm_u[interaction[0]]], i.transpose(0, 1)).index_select(1, rl.perm_i)

    @torch.no_grad()
    def full_sort_topk(self, interaction, k):
        users, mask = interaction[0], interaction[1]
        u, cands = self._cached_eval_candidates()
        i = cands.C if isinstance(cands, hip_ops.TopkCandidates) else cands
        rl = self.relabelling
        cache = getattr(interaction, 'cache', None)          # our EvalDataLoader: batches never change
        key = ('mask_csr', getattr(interaction, 'cache_key', None), i.shape[0], None if rl is None else rl.how)
        if cache is not None and key in cache:
            rowptr, cols = cache[key]
        else:
            if rl is not None and mask.shape[1]:
                mask = torch.stack([mask[0], rl.perm_i[mask[1]]])
            rowptr, cols = mask_to_csr_device(mask, users.shape[0], i.shape[0])
            if cache is not None:
                cache[key] = (rowptr, cols)
        q = u[users if rl is None else rl.perm_u[users]].contiguous()
        out = self._ranked_with_hint(interaction, users, q, cands, k, rowptr, cols)
        # ranked in the relabelled space (ties between EXACTLY equal scores go to the lower relabelled id), reported in the dataset's ids
        return out if rl is None else rl.inv_i[out]

    # ---- WARM evaluation (new config key `hip_eval_hint`, default on): the fp16 filter needs a per-user threshold before its
    # products; a cold call gets it from a first pass over ALL products, a warm call from the k ids ranked for that user LAST
    # time -- the VALID pass's list when the TEST pass follows on the same frozen tables (trainer.py:262,271), the previous
    # epoch's list otherwise -- rescored under the current tables (mmrec_score_topk_hinted_f32): one matrix-core pass instead
    # of two, the same exact top-k.  The lists live on the device ([n_users, 64 or 128] int32: the top-k and the runners-up the
    # kernel ranked anyway, written by the call itself, candidate ids as the kernel reports them); which users have one -- and under which evaluation tables it was written -- is tracked on the host, so a batch is
    # only ranked warm when every user of it has a list.
    eval_hint = True
    _hint = None
    HINT_STALE_FRAC = 0.02      # more than this share of an evaluation's queries in the overflow / slow queues: next one is cold
    HINT_PROBE_FRAC = 0.005     # ... and of the FIRST warm batch of a pass (large candidate sets): the rest of the pass is cold
    HINT_PROBE_MIN_CANDIDATES = 32768

    def _ranked_with_hint(self, interaction, users, q, cands, k, rowptr, cols):
        import numpy as np
        nc = cands.C.shape[0] if isinstance(cands, hip_ops.TopkCandidates) else cands.shape[0]
        if (not self.eval_hint or self.training or nc < k
                or not hip_ops.topk_hint_served(nc, q.shape[1], k)):
            return hip_ops.score_topk(q, cands, k, rowptr, cols)
        st = self._hint
        if st is None or st['k'] != k or st['nc'] != nc or st['table'].device != q.device:
            st = self._hint = dict(k=k, nc=nc, table=torch.full((self.n_users, hip_ops.topk_hint_width(k)), -1, dtype=torch.int32,
                                                                device=q.device),
                                   ver=np.zeros(self.n_users, dtype=np.int64), cold_from=0, warm=0, cold=0, queries=0,
                                   counts=torch.zeros(2, dtype=torch.int32, device=q.device))
        cache = getattr(interaction, 'cache', None)
        key = ('users_host', getattr(interaction, 'cache_key', None))
        if cache is not None and key in cache:
            users_np = cache[key]
        else:
            users_np = users.detach().cpu().numpy()
            if cache is not None:
                cache[key] = users_np
        # a list exists for every user of the batch; after a pass whose lists had gone stale (eval_hint_feedback) only lists
        # written under the CURRENT tables count (the TEST pass after a cold VALID pass), until a cold pass has refreshed them
        oldest = int(st['ver'][users_np].min()) if users_np.shape[0] else 0
        old_lists = oldest != self._tables_version       # lists of EARLIER tables: the ones that can be stale
        warm = oldest > 0 and (oldest >= st['cold_from'] or not old_lists) and not (old_lists and st.get('pass_cold'))
        # either way the call leaves its ranking (top-k + the runners-up it ranked) in the users' rows for the next one
        out = hip_ops.score_topk(q, cands, k, rowptr, cols, hint=st['table'], hint_rows=users, hint_cold=not warm,
                                 queue_counts=st['counts'] if warm else None)
        if warm and old_lists:
            st['queries'] += users_np.shape[0]
            # PROBE: against a large candidate set a query whose list has gone useless costs an exact scan of all candidates
            # (config 5, 200 training steps after the previous evaluation: 4.6 % of the users, 0.42 s for a pass that takes
            # 0.09 s cold).  The first warm batch of a pass tells: one 8-byte read, and the rest of the pass runs cold (and
            # refreshes the lists) when more than HINT_PROBE_FRAC of its queries needed the queues.
            if nc >= self.HINT_PROBE_MIN_CANDIDATES and not st.get('probed'):
                st['probed'] = True
                slow, over = st['counts'].tolist()
                st['pass_cold'] = (slow + over) > self.HINT_PROBE_FRAC * st['queries']
        st['warm' if warm else 'cold'] += 1
        st['ver'][users_np] = self._tables_version
        return out

    def eval_hint_feedback(self):
        """Called by the Trainer once an evaluation's lists are on the host anyway: how many warm queries needed the overflow /
        slow queues (one 8-byte read).  Above HINT_STALE_FRAC the rankings have moved too far for last time's lists -- from
        then on only lists written under the tables being evaluated are used (the TEST pass after the VALID pass), i.e. the
        next evaluation's first pass runs cold and refreshes them.  -> (warm batches, cold batches) of this pass."""
        st = self._hint
        if st is None:
            return 0, 0
        if st['queries']:                                # warm queries ranked from lists of earlier tables
            slow, over = st['counts'].tolist()
            stale = (slow + over) > self.HINT_STALE_FRAC * st['queries']
            # (a pass the probe turned cold has refreshed every list itself: the next evaluation probes again)
            st['cold_from'] = self._tables_version + 1 if (stale and not st.get('pass_cold')) else 0
            st['last_queues'] = (slow, over, st['queries'])
        st['counts'].zero_()
        st['probed'] = st['pass_cold'] = False
        done = (st['warm'], st['cold'])
        st['warm'] = st['cold'] = st['queries'] = 0
        return done


class RelabelledIdsMixin:
    """New config key `reorder: community | degree | rcm` (absent / null: off): the model keeps its id-indexed tables -- the
    user / item id embeddings, the raw feature tables and their Adam state -- in an id space relabelled ONCE at build time for
    gather locality (graph.BipartiteRelabelling), so the locality costs nothing per step.  What the plugin API sees is
    unchanged: `calculate_loss` / `full_sort_predict` / `full_sort_topk` take and return the dataset's ids, `state_dict()`
    holds every table in the ORIGINAL row order (a reference checkpoint loads, and a checkpoint written here loads into the
    reference), and results equal the plain model's -- per-row sums bit for bit (rows keep their nonzero order).
    Subclasses list their tables in `relabelled_tables = {parameter name: 'u' | 'i'}` and call `_map_batch` on the way in."""

    relabelled_tables = {}

    def _setup_relabelling(self, config, base_graph):
        """reads `reorder`; -> graph.BipartiteRelabelling from the model's full normalised user-item graph, or None (key absent / off)"""
        from mmrec_amd.graph import BipartiteRelabelling
        how = config['reorder']
        on = how and str(how).lower() not in ('none', 'false', 'off')
        self.relabelling = BipartiteRelabelling(base_graph, self.n_users, self.n_items, str(how).lower(), self.device) if on else None
        if self.relabelling is not None:
            self._register_relabelling_hooks()
        return self.relabelling

    def _to_relabelled_rows_(self, param, side):
        """in place: the plain model's initial values, row `old` at relabelled row perm[old] (same generator consumption as plain)"""
        rl = self.relabelling
        inv = rl.inv_u if side == 'u' else rl.inv_i
        with torch.no_grad():
            param.copy_(param[inv.to(param.device)])

    def _map_edges(self, edge_indices):
        """[2, E] (user, item) edge list in the relabelled ids, edge ORDER kept (what the per-epoch multinomial draws from)"""
        rl = self.relabelling
        return edge_indices if rl is None else torch.stack([rl.perm_u[edge_indices[0]], rl.perm_i[edge_indices[1]]])

    def _map_batch(self, interaction):
        rl = self.relabelling
        if rl is None:
            return interaction
        rows = [rl.perm_u[interaction[0]]] + [rl.perm_i[interaction[j]] for j in range(1, interaction.shape[0])]
        return torch.stack(rows)

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        if self.relabelling is not None:
            dev = next(self.parameters()).device
            self.relabelling.to(dev)
        return out

    # state_dict() / load_state_dict() see ORIGINAL row order through two hooks registered on the model itself (round-5 advice:
    # overriding the two methods only worked for the top-level module -- a parent's load_state_dict recurses through
    # _load_from_state_dict and never calls a child's override, so a checkpoint round trip through any wrapper scrambled the
    # tables).  Hooks run at any nesting depth with the right prefix, for the positional and the keyword call forms alike.
    def _register_relabelling_hooks(self):
        if getattr(self, '_relabelling_hooks', False):
            return
        self._relabelling_hooks = True

        def to_original(module, state_dict, prefix, local_metadata):
            rl = module.relabelling
            if rl is None:
                return
            for name, side in module.relabelled_tables.items():
                k = prefix + name
                if k in state_dict:                            # original row `old` = relabelled row perm[old]
                    perm = rl.perm_u if side == 'u' else rl.perm_i
                    state_dict[k] = state_dict[k].detach().index_select(0, perm.to(state_dict[k].device))

        def to_relabelled(module, state_dict, prefix, *unused):
            rl = module.relabelling
            if rl is None:
                return
            for name, side in module.relabelled_tables.items():
                k = prefix + name
                if k in state_dict:                            # relabelled row `new` = original row inv[new]
                    inv = rl.inv_u if side == 'u' else rl.inv_i
                    state_dict[k] = state_dict[k].index_select(0, inv.to(state_dict[k].device))

        if hasattr(self, 'register_state_dict_post_hook'):
            self.register_state_dict_post_hook(to_original)
        else:
            self._register_state_dict_hook(to_original)
        if hasattr(self, 'register_load_state_dict_pre_hook'):
            self.register_load_state_dict_pre_hook(to_relabelled)
        else:
            self._register_load_state_dict_pre_hook(to_relabelled, with_module=True)


class AdjacentTablesMixin:
    """Keeps the parameters named in `adjacent_tables` (the user and the item id table) as consecutive row blocks of ONE
    allocation, so that the `cat` every forward starts with (freedom.py:165, bm3.py:87, lightgcn.py:111) already exists:
    hip_ops.lightgcn_mean_parts takes the blocks as they lie (at 1.5M rows the cat is 0.28 ms of a 3.8 ms step).
    Names, shapes and values of the parameters are untouched (a reference state_dict loads as before); after anything
    that re-allocates them (`.to(device)`) the layout is re-established; without it the ops fall back to the cat."""

    adjacent_tables = ()

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self.lay_out_adjacent_tables()
        return out

    def lay_out_adjacent_tables(self):
        ps = [self.get_parameter(n) for n in self.adjacent_tables]
        if len(ps) < 2 or hip_ops.row_blocks_of_one_buffer(ps):
            return
        with torch.no_grad():
            buf = torch.cat([p.data for p in ps], dim=0)
            off = 0
            for p in ps:
                p.data = buf[off:off + p.shape[0]]
                off += p.shape[0]


def emb_loss_rows(tables_and_ids, denom, weight=1.0):
    """weight * EmbLoss over gathered rows: weight * sum_t ||T[ids]||_F / denom (common/loss.py:46-51; `weight` = the model's
    reg_weight, folded into the kernel's scale) -- every term in one launch pair (hip_ops.rows_reg, ABI 14)."""
    return hip_ops.rows_reg(tables_and_ids, hip_ops.ROWS_REG_NORM, float(weight) / denom)     # (all terms in one launch pair, ABI 14)
