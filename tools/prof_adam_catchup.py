#!/usr/bin/env python3
"""The row-lazy catch-up in isolation: exact replay (mmrec_adam_rows_catchup_f32) against the opt-in closed form
(mmrec_adam_rows_fastforward_f32) on a synthetic late-run state of the config-5 image table's shape -- 4,096 listed rows
x 4,096 columns, optimizer step 683, gaps drawn like the real run's (exponential, mean 122; profiles/r06_c5_plugin_run.log) or
all equal.    python tools/prof_adam_catchup.py [F]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmrec_amd import _lib  # noqa: E402


def measure(dev, F=4096, shapes=None, log=print):
    """-> {shape name: {"exact_us", "closed_form_us", "element_steps"}} (min of three timed launches each, state restored)"""
    lib = _lib.load()
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    n, t_now, b1, b2, eps, lr = 4096, 683, 0.9, 0.999, 1e-8, 1e-3
    tt = np.arange(t_now + 2, dtype=np.float64)
    tt[0] = 1
    hist = torch.tensor(np.stack([lr / (1 - b1 ** tt), 1 / np.sqrt(1 - b2 ** tt)], 1), dtype=torch.float32).to(dev)
    gen = torch.Generator(device=dev).manual_seed(0)
    p0 = torch.randn(n, F, device=dev, generator=gen) * 0.1
    v0 = 10.0 ** (torch.rand(n, F, device=dev, generator=gen) * 6 - 12)
    m0 = v0.sqrt() * torch.randn(n, F, device=dev, generator=gen)
    ids = torch.arange(n, device=dev, dtype=torch.int64)
    owner = torch.full((n,), 2 ** 31 - 1, dtype=torch.int32, device=dev)
    rng = np.random.default_rng(0)
    every = {"exponential gaps, mean 122 (the late config-5 step)": np.clip(rng.exponential(122, n), 1, t_now - 1).astype(np.int32),
             "every gap 122": np.full(n, 122, np.int32), "every gap 13": np.full(n, 13, np.int32),
             "every gap 600 (last touched at step 83: replayed to step 128, closed form from there)": np.full(n, 600, np.int32),
             "every gap 400": np.full(n, 400, np.int32)}
    out = {}
    for what, gaps in every.items():
        if shapes is not None and what not in shapes:
            continue
        last0 = torch.from_numpy(t_now - gaps).to(dev)
        res = {"element_steps": float(gaps.sum()) * F}
        for name, fn in (("exact", lib.mmrec_adam_rows_catchup_f32), ("closed_form", lib.mmrec_adam_rows_fastforward_f32)):
            times = []
            for rep in range(4):
                p, m, v, last = p0.clone(), m0.clone(), v0.clone(), last0.clone()
                _lib.check(lib.mmrec_adam_rows_owner(P(ids), n, P(owner), None), "owner")
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                _lib.check(fn(P(p), P(m), P(v), P(ids), P(owner), n, n, F, P(last), P(hist), t_now, b1, b2, eps, 0.0, None), name)
                e1.record()
                torch.cuda.synchronize()
                times.append(e0.elapsed_time(e1) * 1e3)
            res[name + "_us"] = min(times[1:])
        log("%-88s element-steps %.2e:  exact %.0f us  closed form %.0f us   (streaming p, m, v of the rows once each way: %.0f MB)"
            % (what, res["element_steps"], res["exact_us"], res["closed_form_us"], n * F * 24 / 1e6))
        out[what] = res
    return out


if __name__ == "__main__":
    measure(torch.device("cuda:0"), int(sys.argv[1]) if len(sys.argv) > 1 else 4096)
