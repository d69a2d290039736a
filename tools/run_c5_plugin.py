#!/usr/bin/env python3
"""BASELINE config 5 THROUGH THE PLUGIN API on one MI355X: the synthetic 1M-user / 500K-item / 10M-interaction dataset
(mmrec_amd/synth.py, reference on-disk format, 8.2 GB image + 0.77 GB text features) -> Config -> RecDataset -> loaders ->
FREEDOM / ShardedFREEDOM -> Trainer steps -> evaluation.

    python tools/run_c5_plugin.py            # plain FREEDOM, then ShardedFREEDOM (n_gpus code path, single-rank RCCL group,
                                             # collectives forced): ms per training step, evaluation users/s, same losses

The sharded run exercises on the device, at full size, what world-size 2 / 3 runs are tested for on the CPU (gloo): the
nnz-balanced chunked row sharding, the all-gather per layer forward and backward, the item-sharded feature tables with
the row-lazy Adam, the owner-computed projection exchange and the sharded evaluation."""
import os
import socket
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmrec_amd import synth  # noqa: E402


def log(*a):
    print("[c5]", *a, flush=True)


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    root = os.environ.get("MMREC_C5_ROOT") or tempfile.mkdtemp(prefix="mmrec_c5_", dir="/tmp")
    t0 = time.time()
    if not os.path.exists(os.path.join(root, "c5", "image_feat.npy")):
        nu, ni, ne = synth.write_dataset(root, "c5", seed=0)
        log("dataset written: %d users, %d items, %d interactions (%.0fs)" % (nu, ni, ne, time.time() - t0))
    from mmrec_amd.common.trainer import Trainer
    from mmrec_amd.utils.configurator import Config
    from mmrec_amd.utils.dataloader import EvalDataLoader, TrainDataLoader
    from mmrec_amd.utils.dataset import RecDataset
    from mmrec_amd.utils.utils import eval_batch_size, get_model, init_seed
    import torch.distributed as dist
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    losses = {}
    for sharded in ((False,) if os.environ.get("MMREC_C5_PLAIN_ONLY") else (False, True)):
        cd = dict(gpu_id=0, use_gpu=True, data_path=root + "/", epochs=1, save_recommended_topk=False, dropout=0.8,
                  reg_weight=1e-3, dist_force_collectives=True)
        if os.environ.get("MMREC_C5_EAGER"):
            cd["hip_graph_step"] = False
        if os.environ.get("MMREC_C5_NO_PREFETCH"):
            cd["lazy_prefetch"] = False
        if os.environ.get("MMREC_C5_FAST_FORWARD"):      # opt-in closed-form catch-up of the row-lazy tables (not bit-identical)
            cd["lazy_adam_fast_forward"] = True
        config = Config("FREEDOM", "c5", cd)
        for k, v in cd.items():
            config[k] = v
        config["seed"] = 999
        t = time.time()
        data = RecDataset(config)
        str(data)
        tr, va, te = data.split()
        str(tr), str(va), str(te)
        train_data = TrainDataLoader(config, tr, batch_size=config["train_batch_size"], shuffle=True)
        valid_data = EvalDataLoader(config, va, additional_dataset=tr, batch_size=eval_batch_size(config))
        init_seed(999)
        train_data.pretrain_setup()
        log("sharded=%s: dataset + loaders %.1fs" % (sharded, time.time() - t))
        if sharded and not dist.is_initialized():
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", torch.cuda.current_device()))
        t = time.time()
        model = get_model("FREEDOM", sharded=sharded)(config, train_data).to(config["device"])
        torch.cuda.synchronize()
        log("sharded=%s: model built in %.1fs (%d parameters; kNN graph cached after the first build)" %
            (sharded, time.time() - t, sum(p.numel() for p in model.parameters())))
        trainer = Trainer(config, model)
        t = time.time()
        keep = torch.multinomial(model.edge_values, int(model.edge_values.numel() * 0.2),
                                 generator=torch.Generator(device=model.edge_values.device).manual_seed(5))
        model.set_kept_edges(keep)
        torch.cuda.synchronize()
        log("sharded=%s: pruned graph rebuilt in %.2fs" % (sharded, time.time() - t))
        batches = []
        for b in train_data:
            batches.append(b)
            if len(batches) == steps + 3:
                break
        trainer._train_epoch(batches[:3], 0)                        # warm-up
        torch.cuda.synchronize()
        t = time.time()
        total, per = trainer._train_epoch(batches[3:], 0)
        torch.cuda.synchronize()
        dt = time.time() - t
        losses[sharded] = [float(x) for x in per]
        log("sharded=%s: %d training steps, %.2f ms/step (loss first %.6f last %.6f)" %
            (sharded, steps, dt / steps * 1e3, losses[sharded][0], losses[sharded][-1]))
        late = int(os.environ.get("MMREC_C5_LATE_STEPS", "0"))
        if late:     # the row-lazy Adam replays the steps a row sat out: its cost grows until every row has been touched
            more = []
            for b in train_data:
                more.append(b)
                if len(more) == late + steps:
                    break
            trainer._train_epoch(more[:late], 0)
            torch.cuda.synchronize()
            t = time.time()
            trainer._train_epoch(more[late:], 0)
            torch.cuda.synchronize()
            log("sharded=%s: after %d more steps: %.2f ms/step over %d steps" %
                (sharded, late, (time.time() - t) / max(len(more) - late, 1) * 1e3, len(more) - late))
            if os.environ.get("MMREC_C5_ADAM_HIST") and not sharded:      # what the row-lazy catch-up of the NEXT step will replay
                sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
                from lazy_adam_histogram import replay_histogram
                nxt = next(iter(train_data)).to(config["device"])
                ids = torch.cat((nxt[1], nxt[2]))
                if getattr(model, "relabelling", None) is not None:
                    ids = model.relabelling.perm_i[ids]
                for nm in ("image_embedding", "text_embedding"):
                    emb = getattr(model, nm, None)
                    if emb is not None and hasattr(emb, "_last_step"):
                        replay_histogram(emb, ids, float(config["learning_rate"]), tag="[c5] lazy Adam replay, %s: " % nm)
        for which in ("first (builds the per-batch mask CSRs, cached on the loader)", "second"):
            t = time.time()
            res = trainer.evaluate(valid_data)
            torch.cuda.synchronize()
            dt = time.time() - t
            log("sharded=%s: %s evaluation of %d users in %.2fs (%.0f users/s incl. metrics), recall@20 %.4f" %
                (sharded, which, valid_data.pr_end, dt, valid_data.pr_end / dt, res["recall@20"]))
        warm_steps = int(os.environ.get("MMREC_C5_WARM_STEPS", "0"))
        if warm_steps:      # round 6: evaluation WARM -- TEST pass after VALID pass, then again after more training steps
            log("sharded=%s: the second evaluation above ran warm/cold batches %s, queues (slow, overflow, warm queries) %s" %
                (sharded, trainer.eval_warm, (model._hint or {}).get("last_queues")))
            model.eval_hint = False                     # (the same Trainer: a second one would build a second optimizer)
            for _ in range(2):
                t = time.time()
                res_c = trainer.evaluate(valid_data)
                torch.cuda.synchronize()
                dt = time.time() - t
            log("sharded=%s: COLD evaluation (eval_hint off, same tables) in %.3fs (%.0f users/s), equal metrics: %s" %
                (sharded, dt, valid_data.pr_end / dt, res_c == res))
            model.eval_hint = True
            u_old, i_old = [t.float().cpu().numpy() for t in model._cached_eval_embeddings()]
            more = []
            for b in train_data:
                more.append(b)
                if len(more) == warm_steps:
                    break
            trainer._train_epoch(more, 0)
            torch.cuda.synchronize()
            for which in ("first after %d more steps (lists of the previous tables)" % warm_steps, "second (TEST pass: fresh lists)"):
                t = time.time()
                res_w = trainer.evaluate(valid_data)
                torch.cuda.synchronize()
                dt = time.time() - t
                log("sharded=%s: WARM evaluation, %s: %.3fs (%.0f users/s incl. metrics), warm/cold batches %s, queues (slow, "
                    "overflow, warm queries) %s, recall@20 %.4f" % (sharded, which, dt, valid_data.pr_end / dt, trainer.eval_warm,
                                                                     model._hint.get("last_queues"), res_w["recall@20"]))
            model.eval_hint = False
            t = time.time()
            res_c = trainer.evaluate(valid_data)
            torch.cuda.synchronize()
            log("sharded=%s: COLD evaluation of the same tables: %.3fs, equal metrics: %s" % (sharded, time.time() - t, res_c == res_w))
            model.eval_hint = True
            if os.environ.get("MMREC_C5_DIAG"):
                sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
                from topk_survivor_model import survivor_stats, warm_survivor_stats
                ue, ie = [t.float().cpu().numpy() for t in model._cached_eval_embeddings()]
                rs = np.random.default_rng(0).choice(model.n_users, 256, replace=False)
                inter = model.interaction_matrix.tocsr() if hasattr(model.interaction_matrix, "tocsr") else None
                lists = [inter.indices[inter.indptr[u]:inter.indptr[u + 1]] for u in rs]
                deg = np.array([len(l) for l in lists])
                survivor_stats(ue[rs], ie, deg, tag="[c5] diag after %d steps, cold: " % warm_steps)
                warm_survivor_stats(ue[rs], ie, u_old[rs], i_old, lists, tag="[c5] diag after %d steps, lists of the previous tables, " % warm_steps)
                warm_survivor_stats(ue[rs], ie, ue[rs], ie, lists, tag="[c5] diag, lists of the SAME tables (TEST after VALID), ")
        if os.environ.get("MMREC_C5_DIAG") and not warm_steps:   # survivor statistics of the top-K filter on the embeddings just ranked
            sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
            from topk_survivor_model import survivor_stats
            ue, ie = model._cached_eval_embeddings() if hasattr(model, "_cached_eval_embeddings") else model.eval_embeddings()
            rs = np.random.default_rng(0).choice(model.n_users, 256, replace=False)
            deg = np.bincount(model.interaction_matrix.row, minlength=model.n_users)
            survivor_stats(ue[torch.as_tensor(rs, device=ue.device)].float().cpu().numpy(), ie.float().cpu().numpy(), deg[rs],
                           tag="[c5] diag: ")
        if os.environ.get("MMREC_C5_PROFILE_EVAL"):          # where a Trainer evaluation's host time goes
            import cProfile
            import io
            import pstats
            pr = cProfile.Profile()
            pr.enable()
            trainer.evaluate(valid_data)
            torch.cuda.synchronize()
            pr.disable()
            buf = io.StringIO()
            pstats.Stats(pr, stream=buf).sort_stats("cumulative").print_stats(35)
            log("cProfile of a third evaluation:\n" + buf.getvalue())
        log("sharded=%s: peak device memory %.1f GB" % (sharded, torch.cuda.max_memory_allocated() / 2 ** 30))
        del model, trainer, train_data, valid_data, data
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
    if True not in losses:
        return
    a, b = np.array(losses[False]), np.array(losses[True])
    log("max relative loss difference sharded vs plain over %d steps: %.2e" % (steps, np.abs(a / b - 1).max()))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
