#!/usr/bin/env python3
"""A/B of `hip_deterministic` (position-ordered gradient scatters instead of hardware fp32 atomics) on the training step of
BASELINE configs 2-5 through the plugin API + Trainer (round-3 review, item 8): ms per batch with the key off and on, same
data, same seed, alternating runs; config 5 is built in memory the way tests/test_c5_e2e_gpu.py builds it.

    python tools/det_ab.py [c2 c3 c4 c5] [--epochs 3] [--steps 60]"""
import argparse
import os
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def small(name, epochs):
    import run_config
    from mmrec_amd.common.trainer import Trainer
    root, out = None, {}
    for rep in range(2):
        for det in (False, True):
            config, train_data, _, _, model, root = run_config.setup(name, {"hip_deterministic": det}, epochs, root=root)
            trainer = Trainer(config, model)
            times = []
            for epoch in range(epochs):
                model.pre_epoch_processing()
                torch.cuda.synchronize()
                t0 = time.time()
                trainer._train_epoch(train_data, epoch)
                torch.cuda.synchronize()
                times.append((time.time() - t0) / len(train_data) * 1e3)
            out.setdefault(det, []).append(min(times[1:]) if len(times) > 1 else times[0])
            del trainer, model
    return {k: min(v) for k, v in out.items()}


def c5(steps):
    from mmrec_amd.common.trainer import Trainer
    from tests.test_c5_e2e_gpu import build_c5
    root, out = tempfile.mkdtemp(prefix="mmrec_c5ab_", dir="/tmp"), {}
    for det in (False, True):
        config, train_data, _, model = build_c5(root, sharded=False, hyper={"hip_deterministic": det})
        trainer = Trainer(config, model)
        keep = torch.multinomial(model.edge_values, int(model.edge_values.numel() * 0.2),
                                 generator=torch.Generator(device=model.edge_values.device).manual_seed(5))
        model.set_kept_edges(keep)
        batches = []
        for b in train_data:
            batches.append(b)
            if len(batches) == steps + 5:
                break
        trainer._train_epoch(batches[:5], 0)
        torch.cuda.synchronize()
        t0 = time.time()
        trainer._train_epoch(batches[5:], 0)
        torch.cuda.synchronize()
        out[det] = (time.time() - t0) / steps * 1e3
        del trainer, model, train_data
        torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("configs", nargs="*", default=["c2", "c3", "c4", "c5"])
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--steps", type=int, default=60)
    args = ap.parse_args()
    for name in args.configs:
        r = c5(args.steps) if name == "c5" else small(name, args.epochs)
        print("[det-ab] %s: atomics %.3f ms/batch | hip_deterministic %.3f ms/batch | cost %+.1f %%" %
              (name, r[False], r[True], (r[True] / r[False] - 1) * 100), flush=True)


if __name__ == "__main__":
    main()
