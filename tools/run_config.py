#!/usr/bin/env python3
"""End-to-end run of a BASELINE.json configuration on synthetic data of the named dataset's shape,
through the plugin API (Config -> RecDataset -> loaders -> model -> Trainer), on one MI355X:

    python tools/run_config.py c2      # LayerGCN on Amazon-Baby shape
    python tools/run_config.py c3      # FREEDOM on Amazon-Sports shape
    python tools/run_config.py c4      # BM3 on Amazon-Clothing shape
    python tools/run_config.py c1      # VBPR on Amazon-Baby shape (same plumbing, GPU kernels)

Prints per-epoch train time, eval time (valid + test), users/s and the metrics; `--epochs N`."""
import argparse
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmrec_amd import synth  # noqa: E402

CONFIGS = {
    "c1": ("VBPR", "baby", {"reg_weight": 1e-3}),
    "c2": ("LayerGCN", "baby", {"n_layers": 4, "dropout": 0.1, "reg_weight": 1e-3}),
    "c3": ("FREEDOM", "sports", {"dropout": 0.8, "reg_weight": 1e-3}),
    "freedom_baby": ("FREEDOM", "baby", {"dropout": 0.8, "reg_weight": 1e-3}),
    "c4": ("BM3", "clothing", {"n_layers": 2, "dropout": 0.3, "reg_weight": 0.1}),
    "lattice": ("LATTICE", "baby", {"reg_weight": 1e-3, "learning_rate": 1e-3}),
    "lightgcn": ("LightGCN", "baby", {"n_layers": 3, "reg_weight": 1e-4}),
    "mgcn": ("MGCN", "baby", {"cl_loss": 0.01}),
    "mmgcn": ("MMGCN", "baby", {"reg_weight": 1e-3, "learning_rate": 1e-3}),
    "smore": ("SMORE", "baby", {"n_ui_layers": 3, "image_knn_k": 10, "text_knn_k": 10, "reg_weight": 1e-4,
                                "dropout_rate": 0.1}),
    "selfcf": ("SELFCFED_LGN", "baby", {"n_layers": 2, "dropout": 0.2, "reg_weight": 1e-3}),
    "pgl": ("PGL", "baby", {"dropout": 0.2, "reg_weight": 0, "mode": "local"}),
    "bpr": ("BPR", "baby", {"reg_weight": 1e-2}),
    "mmgcf": ("MMGCF", "baby", {"n_ui_layers": 2, "reg_weight": 1e-3, "fusion_mode": "mean", "weighting": "equal",
                                "dropout": 0.5}),
    "slmrec": ("SLMRec", "baby", {"learning_rate": 1e-3, "ssl_temp": 0.5, "ssl_alpha": 0.1, "reg": 1e-3}),
    "grcn": ("GRCN", "baby", {"reg_weight": 1e-3, "learning_rate": 1e-3}),
    "itemknn": ("ItemKNNCBF", "baby", {"knn_k": 10, "shrink": 10}),
    "mvgae": ("MVGAE", "baby", {"learning_rate": 1e-3, "beta": 0.1}),
    "damrs": ("DAMRS", "baby", {"kl_weight": 1, "neighbor_weight": 0.001, "n_mm_layers": 1, "n_ui_layers": 2,
                                "learning_rate": 1e-3}),
    "dualgnn": ("DualGNN", "baby", {"aggr_mode": "add", "reg_weight": 1e-3, "learning_rate": 1e-3}),
    "dragon": ("DRAGON", "baby", {"aggr_mode": "add", "reg_weight": 1e-3, "learning_rate": 1e-3}),
    "lgmrec": ("LGMRec", "baby", {"n_ui_layers": 2, "n_mm_layers": 2, "n_hyper_layer": 1, "hyper_num": 4,
                                  "keep_rate": 0.5, "alpha": 0.3, "cl_weight": 1e-4, "reg_weight": 1e-6}),
}


def setup(name, cd_extra=None, epochs=3, root=None):
    """dataset on disk (synthetic, the named shape) -> Config -> loaders -> model; returns (config, train, valid, test, model, root)"""
    model_name, ds, hyper = CONFIGS[name]
    fresh = root is None
    root = root or tempfile.mkdtemp(prefix="mmrec_%s_" % ds)
    t0 = time.time()
    if fresh:
        nu, ni, ne = synth.write_dataset(root, ds, seed=0)
        print("[%s] synthetic %s-shaped data: %d users, %d items, %d interactions (%.1fs)" %
              (name, ds, nu, ni, ne, time.time() - t0), flush=True)
    if fresh and model_name in ("DualGNN", "DRAGON"):          # the user co-occurrence file these two load
        from mmrec_amd.utils.user_graph import write_user_graph_file
        t0 = time.time()
        write_user_graph_file(os.path.join(root, ds, ds + ".inter"), os.path.join(root, ds, "user_graph_dict.npy"))
        print("[%s] user_graph_dict.npy in %.1fs" % (name, time.time() - t0), flush=True)
    if fresh and model_name == "DAMRS":                        # its item graph: nothing in the reference writes one
        from mmrec_amd.utils.user_graph import write_item_graph_file
        write_item_graph_file(os.path.join(root, ds, ds + ".inter"), os.path.join(root, ds, "item_graph_dict_2.npy"))
    from mmrec_amd.utils.configurator import Config
    from mmrec_amd.utils.dataloader import EvalDataLoader, TrainDataLoader
    from mmrec_amd.utils.dataset import RecDataset
    from mmrec_amd.utils.utils import eval_batch_size, get_model, init_seed
    cd = dict(hyper, gpu_id=0, use_gpu=True, data_path=root + "/", epochs=epochs, save_recommended_topk=False)
    cd.update(cd_extra or {})
    config = Config(model_name, ds, cd)
    for k, v in cd.items():
        config[k] = v
    config["seed"] = 999
    data = RecDataset(config)
    str(data)
    tr, va, te = data.split()
    str(tr), str(va), str(te)
    train_data = TrainDataLoader(config, tr, batch_size=config["train_batch_size"], shuffle=True)
    valid_data = EvalDataLoader(config, va, additional_dataset=tr, batch_size=eval_batch_size(config))
    test_data = EvalDataLoader(config, te, additional_dataset=tr, batch_size=eval_batch_size(config))
    init_seed(999)
    train_data.pretrain_setup()
    t0 = time.time()
    model = get_model(model_name)(config, train_data).to(config["device"])
    torch.cuda.synchronize()
    print("[%s] model %s built in %.2fs (%d parameters)" % (name, model_name, time.time() - t0,
                                                            sum(p.numel() for p in model.parameters())), flush=True)
    return config, train_data, valid_data, test_data, model, root


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config", choices=sorted(CONFIGS) + ["tier"], help="a configuration, or `tier` (with --json): " + " ".join(TIER))
    ap.add_argument("--json", help="write {config: ms_per_batch, eval users/s} to this file (profiles/rNN_run_configs.json)")
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--device-neg-sampling", action="store_true")
    ap.add_argument("--graph-step", action="store_true", help="replay the training step as a hipGraph (every model that does not opt out)")
    ap.add_argument("--no-prefetch", action="store_true", help="row-lazy Adam: catch-up on the main stream (A/B of lazy_prefetch)")
    ap.add_argument("--eager", action="store_true", help="never replay (default: the plugins that declare graph_capturable)")
    ap.add_argument("--dense-adam", action="store_true", help="force the dense fused Adam on the trainable feature tables "
                                                               "(FREEDOM, BM3 default to the row-lazy exact Adam)")
    ap.add_argument("--deterministic", action="store_true", help="hip_deterministic: position-ordered gradient scatters")
    ap.add_argument("--no-batch-rows", action="store_true", help="FREEDOM: hip_pull_batch_rows False (launches over all rows)")
    ap.add_argument("--fp32-linear", action="store_true", help="hip_linear_split False: projection forward + backward on the fp32-MFMA kernels")
    ap.add_argument("--lazy-adam", action="store_true", help="force the row-lazy exact Adam on the feature tables (automatic from 64 Mi elements)")
    ap.add_argument("--fast-forward", action="store_true", help="lazy_adam_fast_forward: closed-form catch-up (opt-in, not bit-identical)")
    args = ap.parse_args()
    cd = dict(device_neg_sampling=args.device_neg_sampling)
    if args.graph_step or args.eager:
        cd['hip_graph_step'] = bool(args.graph_step)       # default: 'auto' (overall.yaml)
    if args.no_prefetch:
        cd['lazy_prefetch'] = False
    if args.dense_adam:
        cd['lazy_feature_adam'] = False
    if args.lazy_adam:
        cd['lazy_feature_adam'] = True
    if args.fast_forward:
        cd['lazy_adam_fast_forward'] = True
    if args.deterministic:
        cd['hip_deterministic'] = True
    if args.no_batch_rows:
        cd['hip_pull_batch_rows'] = False
    if args.fp32_linear:
        cd['hip_linear_split'] = False
    if args.json:
        names = TIER if args.config == "tier" else [args.config]
        out = {n: run(n, cd, args.epochs) for n in names}
        import json
        with open(args.json, "w") as f:
            json.dump({"what": "Trainer-level ms per training batch (best of epochs >= 1) and evaluation users/s (valid + test, metrics "
                               "included) on synthetic data of each configuration's dataset shape, default settings, one MI355X",
                       "epochs": args.epochs, "configs": out}, f, indent=1)
        return
    run(args.config, cd, args.epochs)


# BASELINE.json's configurations: VBPR and the five models north_star names, at their shapes (the regression guard of
# tests/test_host_logic.py::test_run_configs_did_not_regress compares profiles/rNN_run_configs.json of consecutive rounds)
TIER = ["c1", "c2", "c3", "c4", "lattice", "mmgcn"]


def run(name, cd=None, epochs=3, verbose=True):
    """-> {"model", "dataset", "ms_per_batch" (best epoch after the first), "ms_per_batch_epochs", "eval_users_per_s", "step_mode"}"""
    import shutil
    config, train_data, valid_data, test_data, model, root = setup(name, cd, epochs)
    from mmrec_amd.common.trainer import Trainer
    trainer = Trainer(config, model)
    n_eval = valid_data.pr_end + test_data.pr_end
    per_epoch, eval_rate = [], []
    for epoch in range(epochs):
        t0 = time.time()
        model.pre_epoch_processing()
        loss, _ = trainer._train_epoch(train_data, epoch)
        torch.cuda.synchronize()
        t1 = time.time()
        valid = trainer.evaluate(valid_data)
        warm_v = trainer.eval_warm
        test = trainer.evaluate(test_data)
        torch.cuda.synchronize()
        t2 = time.time()
        per_epoch.append((t1 - t0) / len(train_data) * 1e3)
        eval_rate.append(n_eval / (t2 - t1))
        if verbose:
            print("[%s] epoch %d: train %.3fs (%d batches, %.2f ms/batch, loss %.4f) | eval valid+test %.3fs "
                  "(%.0f users/s incl. metrics; warm/cold batches valid %s test %s) | valid recall@20 %.4f ndcg@20 %.4f | test recall@20 %.4f" %
                  (name, epoch, t1 - t0, len(train_data), per_epoch[-1], loss, t2 - t1, eval_rate[-1], warm_v, trainer.eval_warm,
                   valid["recall@20"], valid["ndcg@20"], test["recall@20"]), flush=True)
    gs = getattr(trainer, '_graphed', None)
    mode = "eager (no graphed step)" if gs is None else ("hipGraph capture FAILED -> eager" if gs.failed else "hipGraph replay")
    if verbose:
        print("[%s] training step: %s" % (name, mode))
        print("[%s] peak device memory %.2f GB" % (name, torch.cuda.max_memory_allocated() / 2 ** 30))
    model_name, ds, _ = CONFIGS[name]
    del trainer, model, train_data, valid_data, test_data
    torch.cuda.empty_cache()
    shutil.rmtree(root, ignore_errors=True)
    return {"model": model_name, "dataset": ds, "ms_per_batch": min(per_epoch[1:] or per_epoch), "ms_per_batch_epochs": per_epoch,
            "eval_users_per_s": max(eval_rate[1:] or eval_rate), "step_mode": mode}


if __name__ == "__main__":
    main()
