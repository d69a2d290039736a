"""HipAdam: torch.optim.Adam's update (no amsgrad) as ONE fused HIP kernel per parameter group
(multi-tensor launch, `multi_tensor=False`: one per parameter tensor)
(SURVEY.md 8 f3).  Adam is 44-48 % of the reference's FREEDOM / BM3 CPU step because the raw feature
tables are trainable; on the GPU torch's foreach path makes ~12 passes over them.  Same state key
names (`step`, `exp_avg`, `exp_avg_sq`) as torch's, so LambdaLR and state_dict round-trips work.

`capturable=True` keeps the step count and learning rate in device memory (one tiny prepare kernel per
step derives the bias corrections), so a whole training step can be captured in a hipGraph and replayed."""
import ctypes

import torch

from mmrec_amd import _lib


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


class HipAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, capturable=False,
                 multi_tensor=True):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.capturable = capturable
        self.multi_tensor = multi_tensor   # one launch per parameter group (<= 24 tensors each) instead of one per tensor
        self._dev = {}   # id(group) -> (step int64[1], lr fp32[1], hyper fp32[2]) when capturable
        # config `reorder` (models/_base.py: RelabelledIdsMixin): the per-row state of a relabelled table -- both moments -- lives in
        # the table's row order.  The Trainer registers the tables' permutations here, and a checkpoint then holds that state in the
        # DATASET's row order like the model's state_dict does (it loads under any `reorder`, or none); the tag is saved with it
        # and checked where a table's rows could not be mapped (round-5 advice)
        self._row_orders, self._row_order_tag, self._row_order_complete = {}, None, True

    def set_row_order(self, orders, tag, complete=True):
        """orders: {parameter: (perm, inv)} -- dataset row `old` lives at table row perm[old], table row `new` holds dataset row
        inv[new]; tag: a string naming the relabelling (mode + checksum); complete: False when the model has id-indexed state
        these maps do not cover (row-sharded feature tables): such a checkpoint only loads under the same tag."""
        self._row_orders, self._row_order_tag, self._row_order_complete = dict(orders), tag, bool(complete)

    def _indexed_params(self):
        return [p for group in self.param_groups for p in group['params']]

    def _moments(self, p):
        st = self.state[p]
        if not st:
            st['step'] = 0
            st['exp_avg'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
        return st

    def init_state(self, steps_ahead=None):
        """Allocate all moments (and device scalars) up front -- needed before a graph capture.  steps_ahead: optimizer
        steps the capture about to be made may be replayed for (row-lazy tables keep one pair of scalars per step in a
        device table that a captured step cannot grow)."""
        for group in self.param_groups:
            for p in group['params']:
                if p.requires_grad:
                    self._moments(p)
                    table = getattr(p, '_lazy_table', None)
                    if table is not None and self.capturable and table._opt is not None:
                        table.reserve(steps_ahead if steps_ahead is not None else 1 << 16)
            if self.capturable:
                self._group_dev(group)

    def _group_dev(self, group):
        key = id(group)
        if key not in self._dev:
            dev = group['params'][0].device
            self._dev[key] = (torch.zeros(1, dtype=torch.int64, device=dev),
                              torch.full((1,), float(group['lr']), dtype=torch.float32, device=dev),
                              torch.zeros(2, dtype=torch.float32, device=dev))
        return self._dev[key]

    def sync_lr(self):
        """Push the (scheduler-updated) learning rates to the device scalars; call OUTSIDE a capture,
        before a replay."""
        for group in self.param_groups:
            if id(group) in self._dev:
                self._dev[id(group)][1].fill_(float(group['lr']))

    def state_dict(self):
        """Under hipGraph replay the host mirror of `step` advances once per CAPTURE, the device counter once per
        replay: the device counter is the truth (a resumed run computes its bias corrections from the saved count)."""
        for group in self.param_groups:          # a checkpoint holds dense Adam's values: replay what the lazy tables postponed
            for p in group['params']:
                table = getattr(p, '_lazy_table', None)
                if table is not None:
                    table.flush()
        if self.capturable:
            for group in self.param_groups:
                if id(group) in self._dev:
                    n = int(self._dev[id(group)][0].item())
                    for p in group['params']:
                        if p in self.state and self.state[p]:
                            self.state[p]['step'] = n
        sd = super().state_dict()
        if self._row_orders:
            sd['state'] = dict(sd['state'])
            for i, p in enumerate(self._indexed_params()):
                if p in self._row_orders and i in sd['state']:
                    perm = self._row_orders[p][0]
                    sd['state'][i] = {k: (v.index_select(0, perm.to(v.device)) if torch.is_tensor(v) and v.dim() >= 1 and
                                          v.shape[0] == p.shape[0] else v) for k, v in sd['state'][i].items()}
        sd['mmrec_row_order'] = self._row_order_tag
        return sd

    def load_state_dict(self, state_dict):
        """torch restores the moments and the host step counts; the DEVICE step counters of a capturable (hipGraph) optimizer
        and the row-lazy tables' bookkeeping are restored here, so that a resumed run continues with the bias corrections of
        step n + 1 (and lazy rows are known to be current at step n: a checkpoint holds flushed tables).  Per-row state of
        relabelled tables (config `reorder`) arrives in the dataset's row order and is brought into this run's order."""
        state_dict = dict(state_dict)
        saved_tag = state_dict.pop('mmrec_row_order', None)
        if saved_tag != self._row_order_tag and not self._row_order_complete:
            raise ValueError('HipAdam.load_state_dict: the checkpoint was written under row order %r, this run uses %r, and the '
                             'model keeps per-row optimizer state that cannot be re-ordered (row-sharded feature tables)'
                             % (saved_tag, self._row_order_tag))
        if self._row_orders:
            state_dict['state'] = dict(state_dict['state'])
            for i, p in enumerate(self._indexed_params()):
                if p in self._row_orders and i in state_dict['state']:
                    inv = self._row_orders[p][1]
                    state_dict['state'][i] = {k: (v.index_select(0, inv.to(v.device)) if torch.is_tensor(v) and v.dim() >= 1 and
                                                  v.shape[0] == p.shape[0] else v) for k, v in state_dict['state'][i].items()}
        super().load_state_dict(state_dict)
        for group in self.param_groups:
            n = 0
            for p in group['params']:
                st = self.state.get(p)
                if st and 'step' in st:
                    st['step'] = int(st['step'])
                    n = max(n, st['step'])
            if self.capturable and n:
                self._group_dev(group)[0].fill_(n)
            for p in group['params']:
                table = getattr(p, '_lazy_table', None)
                if table is not None and p in self.state and self.state[p]:
                    own = int(self.state[p].get('step', n))
                    if self.capturable and own != n:
                        # under hipGraph replay the lazy tables read the GROUP's device counter, so a table whose own saved
                        # count differed would replay zeroed scalar-table entries (a silent no-op)
                        raise ValueError('HipAdam.load_state_dict: a row-lazy table was saved at step %d, its parameter '
                                         'group at step %d' % (own, n))
                    # eager mode: a table that saw no gradient in some steps was skipped there (as dense Adam skips a
                    # parameter without .grad) and legitimately lags its group: it resumes at its own count
                    table.resume(n if self.capturable else own)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        for group in self.param_groups:
            b1, b2 = group['betas']
            hyper = None
            if self.capturable:
                step_dev, lr_dev, hyper = self._group_dev(group)
                if not torch.cuda.is_current_stream_capturing():
                    lr_dev.fill_(float(group['lr']))
                _lib.check(lib.mmrec_adam_prepare(_ptr(step_dev), _ptr(lr_dev), float(b1), float(b2),
                                                  _ptr(hyper), stream), "adam_prepare")
            todo = []
            for p in group['params']:
                table = getattr(p, '_lazy_table', None)
                if table is not None:   # row-lazy exact Adam (common/lazy_rows.py): only the touched rows are visited
                    st = self._moments(p)
                    table._bind(st['exp_avg'], st['exp_avg_sq'],
                                (float(b1), float(b2), float(group['eps']), float(group['weight_decay'])),
                                dev=(step_dev, hyper) if self.capturable else None)
                    if table._apply_step(group['lr'], float(b1), float(b2), float(group['eps']),
                                         float(group['weight_decay'])):
                        st['step'] += 1
                    continue
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise _lib.MMRecHipError("HipAdam needs contiguous fp32 device parameters")
                st = self._moments(p)
                st['step'] += 1   # host mirror (exact outside graph replays)
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                todo.append((p, g, st))
            if not todo:
                continue
            if not self.multi_tensor:
                for p, g, st in todo:
                    if self.capturable:
                        _lib.check(lib.mmrec_adam_step_dev_f32(
                            _ptr(p), _ptr(g), _ptr(st['exp_avg']), _ptr(st['exp_avg_sq']), p.numel(), _ptr(hyper),
                            float(b1), float(b2), float(group['eps']), float(group['weight_decay']), stream),
                            "adam_step_dev")
                    else:
                        _lib.check(lib.mmrec_adam_step_f32(
                            _ptr(p), _ptr(g), _ptr(st['exp_avg']), _ptr(st['exp_avg_sq']), p.numel(),
                            float(group['lr']), float(b1), float(b2), float(group['eps']),
                            float(group['weight_decay']), int(st['step']), stream), "adam_step")
                continue
            # one launch for the whole group: host arrays of device pointers, copied into the kernel arguments
            k = len(todo)
            ptrs = lambda ts: (ctypes.c_void_p * k)(*[t.data_ptr() for t in ts])
            pp, gg = ptrs([t[0] for t in todo]), ptrs([t[1] for t in todo])
            mm, vv = ptrs([t[2]['exp_avg'] for t in todo]), ptrs([t[2]['exp_avg_sq'] for t in todo])
            nn_ = (ctypes.c_int64 * k)(*[t[0].numel() for t in todo])
            if self.capturable:
                _lib.check(lib.mmrec_adam_multi_step_dev_f32(
                    pp, gg, mm, vv, nn_, k, _ptr(hyper), float(b1), float(b2), float(group['eps']),
                    float(group['weight_decay']), stream), "adam_multi_step_dev")
            else:
                lrs = (ctypes.c_float * k)(*[float(group['lr'])] * k)
                steps = (ctypes.c_int64 * k)(*[int(t[2]['step']) for t in todo])
                _lib.check(lib.mmrec_adam_multi_step_f32(
                    pp, gg, mm, vv, nn_, k, lrs, steps, float(b1), float(b2), float(group['eps']),
                    float(group['weight_decay']), stream), "adam_multi_step")
        return loss
