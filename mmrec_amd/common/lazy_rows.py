"""Row-lazy exact Adam for trainable raw-feature tables (kernels in csrc/adam.hip).

FREEDOM / BM3 / LATTICE keep the raw item features trainable (`nn.Embedding.from_pretrained(..., freeze=False)`,
freedom.py:58,61), so torch's Adam streams the whole [n_items, 4096] table and both moments every step although a
step's gradient touches at most 2B rows.  `LazyRowEmbedding` is an nn.Embedding (same parameter name, same
state_dict) whose rows are brought up to date on demand and updated only where the gradient is non-zero:

    feats = table.rows(ids)        # 1. replay the postponed zero-gradient Adam steps of these rows (in place),
                                   # 2. gather them (differentiable; duplicates allowed)
    loss.backward()                # the row gradients are parked on the table, `weight.grad` stays None
    optimizer.step()               # HipAdam: one Adam step on the touched rows only

The result equals dense Adam bit for bit (tests/test_hip_parity.py::test_lazy_row_adam_equals_dense).  `flush()`
replays everything that is still postponed (before the table is read as a whole: state_dict, export).
Opt-in `fast_forward` (config `lazy_adam_fast_forward: True`): rows skipped for long are advanced in closed form
(`mmrec_adam_rows_fastforward_f32`) -- 1e-6-close to the replay, not bit-identical; off by default.
No host synchronisation anywhere: duplicates are resolved by an `owner` array on the device, not by sort / unique.
Under a capturable HipAdam (hipGraph replay of the training step, common/graph_step.py) nothing step-dependent comes from
the host either: the step count and the two bias-correction scalars are read from the optimizer's device counters
(`mmrec_adam_*_dev`), and the per-step scalar table is reserved per capture (`reserve`).
"""
import ctypes

import torch
import torch.nn as nn

from mmrec_amd import _lib

INT_MAX = 2 ** 31 - 1
# while a step is being captured into a hipGraph, the side-stream catch-up becomes a parallel branch of the graph
# (fork: side.wait_stream(capturing stream); join: wait_event in rows()); False = one stream, catch-up in line
PREFETCH_IN_CAPTURE = True
MAX_IDS = 16000           # MMREC_ADAM_ROWS_MAX_IDS (include/mmrec_hip.h)


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


_SIDE = {}


def _side_stream(device):
    key = str(device)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=device)
    return _SIDE[key]


class _GatherRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weight, ids, table):
        ctx.table, ctx.ids = table, ids
        if table.allow_missing:          # ids of -1 = "no row": a zero row forward, no gradient, no optimizer work
            present = ids >= 0
            return weight.detach().index_select(0, ids.clamp_min(0)) * present.unsqueeze(1).to(weight.dtype)
        return weight.detach().index_select(0, ids)

    @staticmethod
    def backward(ctx, dY):
        ctx.table._pending.append((ctx.ids, dY.contiguous()))
        return None, None, None


class LazyRowEmbedding(nn.Embedding):
    """nn.Embedding whose Adam update is applied per touched row (see module docstring).  Use `rows(ids)`."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._lazy_init()

    @classmethod
    def from_pretrained(cls, embeddings, freeze=False, **kwargs):
        assert not freeze, "a frozen table needs no optimizer"
        m = super().from_pretrained(embeddings, freeze=False, **kwargs)
        m.__class__ = cls
        m._lazy_init()
        return m

    def _lazy_init(self):
        self.weight._lazy_table = self          # how HipAdam finds us
        self._pending = []                      # (ids, row gradients) of this step's uses
        self._opt = None                        # (exp_avg, exp_avg_sq, hyper) once the optimizer has seen us
        self._t = 0                             # optimizer steps taken on this table
        self._last_step = self._owner = self._hist = None
        self._dev = None                        # (step_dev int64[1], hyper_dev fp32[2]) of a capturable HipAdam: graph replay
        self._overflow = None                   # device flag: a step beyond the capacity of `_hist` (checked per epoch)
        self._prefetched = None                 # (ids, event) of a catch-up running on the side stream
        self.allow_missing = False              # True: rows(ids) accepts -1 = "no row" (item-sharded tables: slots of other ranks)
        self.fast_forward = False               # True (config lazy_adam_fast_forward): skipped steps in closed form, NOT bit-identical

    # ---- device state, created on first use (the module may have been moved since construction)
    def _state(self):
        w = self.weight
        if not w.is_cuda or w.dtype != torch.float32 or not w.is_contiguous() or w.shape[1] % 4:
            raise _lib.MMRecHipError("LazyRowEmbedding needs a contiguous fp32 device table with F % 4 == 0")
        if self._last_step is None or self._last_step.device != w.device:
            self._last_step = torch.zeros(w.shape[0], dtype=torch.int32, device=w.device)
            self._owner = torch.full((w.shape[0],), INT_MAX, dtype=torch.int32, device=w.device)
            self._hist = torch.zeros(1024, 2, dtype=torch.float32, device=w.device)
        return w

    def _bind(self, exp_avg, exp_avg_sq, hyper, dev=None):
        self._opt = (exp_avg, exp_avg_sq, hyper)
        self._dev = dev

    # ---- hipGraph replay (capturable HipAdam): nothing step-dependent comes from the host
    def steps_on_device(self):
        """optimizer steps taken so far according to the device counter (one host sync)"""
        return int(self._dev[0].item()) if self._dev is not None else self._t

    def reserve(self, n_more):
        """Make room in the per-step scalar table for `n_more` further optimizer steps.  Called OUTSIDE a capture (once per
        capture, i.e. per epoch): a captured step cannot grow the table, it raises the overflow flag instead."""
        w = self._state()
        if self._dev is not None:
            self._t = max(self._t, self.steps_on_device())
            self.check_overflow()
        need = self._t + int(n_more) + 2
        if need > self._hist.shape[0]:
            grown = torch.zeros(max(need, 2 * self._hist.shape[0]), 2, dtype=torch.float32, device=w.device)
            grown[:self._hist.shape[0]] = self._hist
            self._hist = grown

    @torch.no_grad()
    def resume(self, n_steps):
        """State after `load_state_dict` of a checkpoint taken at optimizer step `n_steps`: a checkpoint holds FLUSHED tables
        (`_save_to_state_dict`), i.e. every row is up to date at that step -- nothing older than it is ever replayed, so
        only the counters and room for the steps to come are needed."""
        w = self._state()
        self._t = int(n_steps)
        self._last_step.fill_(self._t)
        if self._hist.shape[0] < self._t + 1024:
            self._hist = torch.zeros(self._t + 1024, 2, dtype=torch.float32, device=w.device)
        self._pending, self._prefetched = [], None
        self._owner.fill_(INT_MAX)              # no catch-up is in flight: every owner mark is free
        if self._overflow is not None:
            self._overflow.zero_()              # (sticky flag of the run the checkpoint replaces)

    def check_overflow(self):
        if self._overflow is not None and int(self._overflow.item()):
            raise _lib.MMRecHipError("row-lazy Adam: more replayed optimizer steps than reserve() made room for; the "
                                     "feature table is not up to date (GraphedTrainStep reserves per epoch)")

    def _catch_up(self, ids):
        """rows of `ids` (None: all) -> state after the `self._t` optimizer steps taken so far"""
        if self._opt is None or (self._t == 0 and self._dev is None):
            return
        w = self._state()
        m, v, (b1, b2, eps, wd) = self._opt
        lib = _lib.load()
        n = 0 if ids is None else ids.numel()
        if ids is not None:
            _lib.check(lib.mmrec_adam_rows_owner(_p(ids), n, _p(self._owner), _stream()), "adam_rows_owner")
        fast = getattr(self, 'fast_forward', False)
        if self._dev is not None:
            fn = lib.mmrec_adam_rows_fastforward_dev_f32 if fast else lib.mmrec_adam_rows_catchup_dev_f32
            _lib.check(fn(
                _p(w), _p(m), _p(v), None if ids is None else _p(ids), None if ids is None else _p(self._owner), n,
                w.shape[0], w.shape[1], _p(self._last_step), _p(self._hist), self._hist.shape[0], _p(self._dev[0]), b1, b2,
                eps, wd, _stream()), "adam_rows_catchup_dev")
            return
        fn = lib.mmrec_adam_rows_fastforward_f32 if fast else lib.mmrec_adam_rows_catchup_f32
        _lib.check(fn(
            _p(w), _p(m), _p(v), None if ids is None else _p(ids), None if ids is None else _p(self._owner), n,
            w.shape[0], w.shape[1], _p(self._last_step), _p(self._hist), self._t, b1, b2, eps, wd, _stream()),
            "adam_rows_catchup")       # (the kernel hands the owner marks back: all INT_MAX again)

    def prefetch(self, ids):
        """Start the catch-up of rows `ids` on a side stream.  The catch-up streams p, m, v of every listed row through HBM
        (~100 us per table and step at Sports size) and needs nothing the rest of the step produces, while the graph
        propagation the models run first is a chain of short latency-bound launches that leaves HBM idle: a model that
        knows its row ids before it propagates calls this first, and the matching `rows(ids)` only waits for the event."""
        if self._opt is None or (self._t == 0 and self._dev is None) or not ids.is_cuda:
            return
        if torch.cuda.is_current_stream_capturing() and not PREFETCH_IN_CAPTURE:
            return                               # the captured step keeps to one stream: rows() catches up in line
        ids = ids.contiguous()
        side = _side_stream(ids.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            self._catch_up(ids)
            done = torch.cuda.Event()
            done.record(side)
        self._prefetched = (ids, done)

    def rows(self, ids):
        """up-to-date rows `ids` [len(ids), F], differentiable w.r.t. the table"""
        ids = ids.contiguous()
        pf, self._prefetched = getattr(self, '_prefetched', None), None
        if pf is not None:
            torch.cuda.current_stream().wait_event(pf[1])       # (also when the ids differ: the table must be quiet)
        if pf is None or pf[0].data_ptr() != ids.data_ptr() or pf[0].numel() != ids.numel():
            with torch.no_grad():
                self._catch_up(ids)
        return _GatherRows.apply(self.weight, ids, self)

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        self.flush()                            # a checkpoint must hold dense Adam's values, not stale rows
        super()._save_to_state_dict(destination, prefix, keep_vars)

    @torch.no_grad()
    def flush(self):
        """apply every postponed update: afterwards `weight` (and the moments) equal dense Adam's"""
        if self._dev is not None:
            self.check_overflow()
        self._catch_up(None)

    # ---- called by HipAdam.step()
    @torch.no_grad()
    def _apply_step(self, lr, b1, b2, eps, wd):
        if not self._pending:
            if self._dev is not None and self.steps_on_device() > 0:
                # the device step counter is the optimizer's: a step that skips this table would leave a hole in its
                # per-step scalars (dense Adam skips a parameter without gradient; a later catch-up could not)
                raise _lib.MMRecHipError("a row-lazy table under a capturable (hipGraph) optimizer must be used in every "
                                         "training step; set lazy_feature_adam: False or hip_graph_step: False")
            return False
        w = self._state()
        m, v, _ = self._opt
        lib = _lib.load()
        ids = torch.cat([i for i, _ in self._pending]) if len(self._pending) > 1 else self._pending[0][0]
        dY = torch.cat([g for _, g in self._pending]) if len(self._pending) > 1 else self._pending[0][1]
        self._pending = []
        n = ids.numel()
        self._t += 1                              # (under graph replay: a mirror that advances per capture; see reserve)
        if self._dev is not None:
            if self._overflow is None:
                self._overflow = torch.zeros(1, dtype=torch.int32, device=w.device)
            if not torch.cuda.is_current_stream_capturing() and self._t + 2 > self._hist.shape[0]:
                self.reserve(1024)
            _lib.check(lib.mmrec_adam_hist_set_dev(_p(self._hist), self._hist.shape[0], _p(self._dev[0]), _p(self._dev[1]),
                                                   _p(self._overflow), _stream()), "adam_hist_set_dev")
        else:
            if self._t >= self._hist.shape[0]:
                grown = torch.zeros(2 * self._hist.shape[0], 2, dtype=torch.float32, device=w.device)
                grown[:self._hist.shape[0]] = self._hist
                self._hist = grown
            _lib.check(lib.mmrec_adam_hist_set(_p(self._hist), self._t, float(lr), b1, b2, _stream()), "adam_hist_set")
        _lib.check(lib.mmrec_adam_rows_owner(_p(ids), n, _p(self._owner), _stream()), "adam_rows_owner")
        # rows used through several calls of this step were caught up by the first one.  The workgroup of a row's first
        # occurrence sums the row's occurrences itself, in position order (deterministic; no zero-fill + index_add_ pass
        # over the [n, F] gradient); id lists too long for its LDS position list are pre-summed into the owner slots
        presummed = n > MAX_IDS
        if presummed:
            # no boolean-mask indexing here: `x[mask]` runs nonzero(), a host synchronisation -- illegal inside a hipGraph
            # capture and a stall in eager mode.  Rows of id -1 ("no row") add zeros into slot 0 instead of being dropped
            present = ids >= 0
            slots = self._owner.index_select(0, ids.clamp_min(0)).long()
            slots = torch.where(present, slots, torch.zeros_like(slots))
            g = torch.zeros_like(dY).index_add_(0, slots, torch.where(present.unsqueeze(1), dY, torch.zeros_like(dY)))   # (a NaN row of a skipped slot must not reach slot 0)
        else:
            g = dY
        if self._dev is not None:
            _lib.check(lib.mmrec_adam_rows_step_dev_f32(_p(w), _p(m), _p(v), _p(ids), _p(self._owner), _p(g), n, w.shape[1],
                                                        _p(self._last_step), self._hist.shape[0], _p(self._dev[0]),
                                                        _p(self._dev[1]), b1, b2, eps, wd, int(presummed), _stream()),
                       "adam_rows_step_dev")
            return True
        _lib.check(lib.mmrec_adam_rows_step_f32(_p(w), _p(m), _p(v), _p(ids), _p(self._owner), _p(g), n, w.shape[1],
                                                _p(self._last_step), self._t, float(lr), b1, b2, eps, wd, int(presummed),
                                                _stream()),
                   "adam_rows_step")          # (owner marks consumed by the kernel)
        return True


AUTO_MIN_ELEMENTS = 64 << 20   # automatic mode, eager steps: tables from 64 Mi elements (256 MB) up
AUTO_MIN_ELEMENTS_REPLAYED = 16 << 20   # ... steps replayed as a hipGraph (`hip_graph_step` auto / True: the default): from 16 Mi


def lazy_adam_enabled(config, n_elements=0):
    """`lazy_feature_adam`: True / False, or absent = automatic.  Possible whenever the fused HIP Adam runs the step
    (learner adam, hip_fused_adam on, GPU; eager or replayed as a hipGraph); the update is bit-identical either way.
    Automatic mode turns it on for large tables only: the ~20 extra small launches of a step cost ~0.25 ms, more than
    dense Adam over the Amazon-Baby tables (31.6 M elements: 0.17 ms; measured 0.76 -> 1.02 ms per step), less than
    it from Sports (75 M: 1.90 -> 1.73 ms) and Clothing (3.36 -> 2.50 ms) up, 4.6 x at 500K items.  Those were EAGER steps;
    replayed as a hipGraph (the default for the plugins that use these tables) the extra launches cost a few microseconds each
    and the row-lazy form wins at the Amazon-Baby tables too (Trainer level, FREEDOM: 0.87 -> 0.67 ms per batch, identical
    losses; profiles/r06_lazy_adam_auto_threshold.log), so the automatic threshold is 16 Mi elements when `hip_graph_step` is
    not switched off."""
    want = config['lazy_feature_adam']
    ok = (str(config['learner']).lower() == 'adam' and config['hip_fused_adam'] in (None, True) and
          not config['clip_grad_norm'] and    # clipping needs the dense .grad
          getattr(config['device'], 'type', str(config['device'])) == 'cuda')
    replayed = config['hip_graph_step'] is not False and str(config['hip_graph_step']).lower() not in ('false', 'off', 'none')
    need = AUTO_MIN_ELEMENTS_REPLAYED if replayed else AUTO_MIN_ELEMENTS
    return (ok and n_elements >= need) if want is None else (bool(want) and ok)


def flush_lazy_tables(module):
    for m in module.modules():
        if isinstance(m, LazyRowEmbedding):
            m.flush()
