#!/usr/bin/env python3
"""Where the host spends a Trainer-level epoch (tools/run_config.py configurations): loader, step call, the final read.
    python tools/prof_trainer_epoch.py freedom_baby"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import run_config  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "freedom_baby"
    from mmrec_amd.common.trainer import Trainer
    config, train_data, valid_data, test_data, model, root = run_config.setup(name, None, 3)
    trainer = Trainer(config, model)
    for epoch in range(4):
        model.pre_epoch_processing()
        torch.cuda.synchronize()
        if epoch < 2:
            trainer._train_epoch(train_data, epoch)
            continue
        graphed = trainer._graphed_step(model.calculate_loss)
        if epoch == 2:            # finer: where the loader and the step call spend their time (wrapped once)
            acc = {}

            def timed(obj, attr, key):
                fn = getattr(obj, attr)

                def wrap(*a, **k):
                    t = time.perf_counter()
                    try:
                        return fn(*a, **k)
                    finally:
                        acc[key] = acc.get(key, 0.0) + time.perf_counter() - t
                setattr(obj, attr, wrap)
            timed(train_data, "_sample_neg_ids", "loader: negative sampling")
            timed(train_data, "_slice", "loader: slice")
            timed(train_data, "_pairs_with_negative", "loader: batch assembly incl. the H2D copy")
            timed(graphed.opt, "sync_lr", "step call: sync_lr")
            timed(torch.cuda.CUDAGraph, "replay", "step call: hipGraphLaunch")
            main.acc = acc
        graphed.invalidate()
        graphed.steps_per_capture = len(train_data) + 2
        model.train()
        t_load = t_step = 0.0
        t0 = time.perf_counter()
        it = iter(train_data)
        t_shuffle = time.perf_counter() - t0
        n = 0
        while True:
            a = time.perf_counter()
            try:
                b = next(it)
            except StopIteration:
                break
            c = time.perf_counter()
            graphed(b)
            d = time.perf_counter()
            t_load += c - a
            t_step += d - c
            n += 1
        e = time.perf_counter()
        torch.cuda.synchronize()
        f = time.perf_counter()
        print("[%s] epoch %d: %d batches, wall %.2f ms = shuffle %.2f + loader %.2f + step calls %.2f + final wait %.2f  (per batch: "
              "loader %.3f, step call %.3f, total %.3f ms)" % (name, epoch, n, (f - t0) * 1e3, t_shuffle * 1e3, t_load * 1e3, t_step * 1e3,
                                                              (f - e) * 1e3, t_load / n * 1e3, t_step / n * 1e3, (f - t0) / n * 1e3), flush=True)
        if getattr(main, "acc", None):
            for k, v in main.acc.items():
                print("    %-60s %.3f ms per batch" % (k, v / n * 1e3), flush=True)
            main.acc.clear()


if __name__ == "__main__":
    main()
