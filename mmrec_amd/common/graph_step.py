"""Training step as a replayed hipGraph (calculate_loss + backward + optimizer step).

On the Amazon-sized datasets a step is ~40-80 short kernels; eager launching leaves the GPU idle most
of the time (LayerGCN on Baby: ~0.35 ms of kernels in a 1.7 ms step).  The step is captured once per
epoch -- models rebuild their pruned graph in `pre_epoch_processing`, which changes buffers and launch
geometry -- with the batch ids in a static buffer, and replayed for every full-size batch.  Capturing
does not execute anything, so no extra optimizer step is taken; the kernels and their order are the
eager ones.  Requires the capturable HipAdam (device-side step count / learning rate); row-lazy feature tables
(common/lazy_rows.py) take their step-dependent scalars from the same device counters.
"""
import torch


class GraphedTrainStep:
    def __init__(self, model, optimizer, loss_func=None, steps_per_capture=None):
        self.model, self.opt = model, optimizer
        self.steps_per_capture = steps_per_capture   # replays one capture may see (an epoch): row-lazy tables reserve for it
        self.loss_func = loss_func or model.calculate_loss
        self.graph = None
        self.static_batch = None
        self.static_loss = None
        self._warm = False
        self.failed = False

    def invalidate(self):
        """Call when buffers the step reads were re-created (new epoch / rebuilt graph)."""
        self.graph = None
        # a plugin whose FIRST batch of an epoch does something the others do not (LATTICE builds its learned item graph
        # there, lattice.py:137-159) declares `graph_eager_batches = 1`: those batches run eagerly, the capture is taken on
        # the next one
        self._eager_left = int(getattr(self.model, 'graph_eager_batches', 0) or 0)

    def _capture(self, batch):
        self.static_batch = batch.clone()
        import inspect
        try:                                         # does this optimizer reserve per-capture room for row-lazy tables?
            takes_steps = len(inspect.signature(self.opt.init_state).parameters) >= 1
        except (TypeError, ValueError):
            takes_steps = False
        if takes_steps:
            self.opt.init_state(self.steps_per_capture)
        else:                                        # an optimizer without row-lazy tables to reserve for
            self.opt.init_state()
        self.opt.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            losses = self.loss_func(self.static_batch)
            loss = sum(losses) if isinstance(losses, tuple) else losses
            loss.backward()
            self.opt.step()
        self.graph, self.static_loss = g, loss

    def __call__(self, batch):
        """Runs one optimizer step on `batch`; returns the (static) loss tensor."""
        if not self._warm:
            # the very first step runs eagerly: library GEMMs (rocBLAS / hipBLASLt behind the 64x64
            # gate / predictor layers of BM3, LATTICE, MMGCN, MGCN) create handles and workspaces on
            # first use, which is not permitted while a stream is capturing
            self._warm = True
            return self._eager(batch)
        if self.failed:
            return self._eager(batch)
        if self.graph is None and getattr(self, '_eager_left', 0) > 0:
            self._eager_left -= 1
            return self._eager(batch)
        if self.graph is None or batch.shape != self.static_batch.shape:
            if self.graph is not None and batch.shape != self.static_batch.shape:
                return self._eager(batch)            # the short last batch of an epoch
            try:
                self._capture(batch)
            except Exception as ex:                  # something in this model's step cannot be captured: run eagerly
                import logging
                logging.getLogger().warning('hipGraph capture of the training step failed (%r); continuing eagerly' % (ex,))
                self.failed, self.graph = True, None
                torch.cuda.synchronize()
                return self._eager(batch)
        else:
            self.static_batch.copy_(batch)
        self.opt.sync_lr()
        self.graph.replay()
        return self.static_loss

    def _eager(self, batch):
        self.opt.zero_grad(set_to_none=True)
        losses = self.loss_func(batch)
        loss = sum(losses) if isinstance(losses, tuple) else losses
        loss.backward()
        self.opt.step()
        return loss
