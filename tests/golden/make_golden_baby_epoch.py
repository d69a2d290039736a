#!/usr/bin/env python3
"""One TRAINED epoch of the unmodified reference at Amazon-Baby shape (north_star: "Recall@20 within 1e-4 of reference on
Amazon-Baby"):

    python tests/golden/make_golden_baby_epoch.py      # build container only (needs /root/reference); ~1 min on 8 cores

LayerGCN (config 2: L = 4, dropout 0.1, reg 1e-3) and FREEDOM (dropout 0.8, reg 1e-3) on the Baby-shaped synthetic dataset
(mmrec_amd/synth.py write_dataset("baby", seed=0)), seed 999: the reference's own Trainer runs `pre_epoch_processing()` +
`_train_epoch()` (58 optimizer steps on the loader's batches, torch.optim.Adam) and `evaluate(valid)`.  Recorded in
tests/golden/baby_epoch.npz: the epoch's multinomial draw (so that the device run prunes the same edges), FREEDOM's frozen
item-item graph, the epoch loss, the validation metric dict, and a fingerprint of every trained parameter (norm + sampled
rows).  tests/test_baby_trained_gpu.py replays the epoch on the HIP kernels and compares.
"""
import os
import sys
import tempfile

import numpy as np
import scipy.sparse as sp
import torch

REF_SRC = "/root/reference/src"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SEED = 999


def install_shims():
    np.float = float
    sp.dok_matrix._update = lambda self, d: self._dict.update(d)
    sys.path.insert(0, os.path.join(HERE, "_shims"))
    sys.path.insert(0, REF_SRC)
    import common  # noqa
    import common.abstract_recommender  # noqa
    import common.loss  # noqa
    for n in ("", ".abstract_recommender", ".loss"):
        sys.modules["models.common" + n] = sys.modules["common" + n]


def main():
    sys.path.insert(0, ROOT)
    from mmrec_amd import synth
    tmp = tempfile.mkdtemp(prefix="mmrec_golden_baby_")
    print(synth.write_dataset(tmp, "baby", seed=0), flush=True)
    install_shims()
    os.chdir(REF_SRC)
    from utils.configurator import Config
    from utils.dataset import RecDataset
    from utils.dataloader import TrainDataLoader, EvalDataLoader
    from utils.utils import init_seed, get_model
    from common.trainer import Trainer
    torch.set_num_threads(os.cpu_count() or 1)
    out = {}
    for tag, name, extra in (("lay_", "LayerGCN", {"n_layers": 4, "dropout": 0.1, "reg_weight": 1e-3}),
                             ("fr_", "FREEDOM", {"dropout": 0.8, "reg_weight": 1e-3})):
        cd = {"gpu_id": 0, "use_gpu": False, "data_path": tmp + "/", "save_recommended_topk": False, "epochs": 1}
        cd.update(extra)
        config = Config(name, "baby", cd)
        for k, v in extra.items():
            config[k] = v
        config["seed"] = SEED
        dataset = RecDataset(config)
        str(dataset)
        tr, va, te = dataset.split()
        str(tr), str(va), str(te)
        train_data = TrainDataLoader(config, tr, batch_size=config["train_batch_size"], shuffle=True)
        valid_data = EvalDataLoader(config, va, additional_dataset=tr, batch_size=config["eval_batch_size"])
        init_seed(SEED)
        train_data.pretrain_setup()
        model = get_model(name)(config, train_data)
        if name == "FREEDOM":
            mm = model.mm_adj
            out[tag + "mm_idx"] = mm._indices().numpy().astype(np.uint16)
            out[tag + "mm_vals"] = mm._values().numpy().astype(np.float32)
        trainer = Trainer(config, model)
        keep, real_multinomial = {}, torch.multinomial

        def recording(w, n, *a, **k):
            keep["idx"] = real_multinomial(w, n, *a, **k)
            return keep["idx"]
        torch.multinomial = recording
        model.pre_epoch_processing()
        torch.multinomial = real_multinomial
        out[tag + "keep_idx"] = keep["idx"].numpy().astype(np.int32)
        loss, _ = trainer._train_epoch(train_data, 0)
        out[tag + "epoch_loss"] = np.float64(loss)
        res = trainer.evaluate(valid_data)
        keys = sorted(res)
        out[tag + "metric_keys"], out[tag + "metrics"] = np.array(keys), np.array([res[k] for k in keys], dtype=np.float64)
        for pname, p in model.named_parameters():
            w = p.detach()
            out[tag + "p_" + pname + "_norm"] = np.float64(w.double().norm().item())
            rows = np.sort(np.random.default_rng(5).choice(w.shape[0], min(128, w.shape[0]), replace=False)) if w.dim() == 2 \
                else np.arange(w.shape[0])
            out[tag + "p_" + pname + "_rows"] = rows.astype(np.int64)
            out[tag + "p_" + pname + "_vals"] = (w[torch.as_tensor(rows)][:, :64] if w.dim() == 2 else w).numpy().copy()
        print(name, "epoch loss", loss, "recall@20", res["recall@20"], flush=True)
    np.savez_compressed(os.path.join(HERE, "baby_epoch.npz"), **out)
    print("wrote baby_epoch.npz", os.path.getsize(os.path.join(HERE, "baby_epoch.npz")) / 1e6, "MB")


if __name__ == "__main__":
    main()
