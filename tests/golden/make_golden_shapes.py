#!/usr/bin/env python3
"""Golden vectors from the *unmodified* reference AT THE BENCHMARK CONFIGS' SIZES (BASELINE configs 3 and 4):

    python tests/golden/make_golden_shapes.py          # build container only (needs /root/reference); ~3 min, ~4 GB RAM

FREEDOM on the Amazon-Sports-shaped and BM3 on the Amazon-Clothing-shaped synthetic dataset (mmrec_amd/synth.py,
write_dataset(seed=0): the generator is data, not code under test; the datasets are re-created bit for bit on the GPU
box by the test).  For each: the reference's own Config -> RecDataset -> loaders -> model (seed 999), the reference's own
random draws (the per-epoch multinomial of freedom.py:128-143 / the four F.dropout masks of bm3.py:110-119, recorded so
that the device run can replay them), ONE `calculate_loss` + backward on the loader's first batch, then the reference
Trainer's evaluation of the validation split.  Kept in `tests/golden/shapes.npz` (~2 MB): the draws, the batch, the loss,
sampled rows of the forward embeddings and of every gradient (+ each gradient's Frobenius norm and sum), the metric dict
and the top-50 lists of 512 sampled users.  The 300 MB feature tables and the full gradients are NOT stored: the test
regenerates the inputs and compares samples + norms.

Nothing here is imported by the product, the tests, bench.py or smoke(); it only *produces* data.
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
N_SAMPLE = 256


def install_shims():
    np.float = float  # utils/metrics.py:51,57,81,84
    sp.dok_matrix._update = lambda self, d: self._dict.update(d)  # freedom.py:111 & copies
    sys.path.insert(0, os.path.join(HERE, "_shims"))
    sys.path.insert(0, REF_SRC)
    import common  # noqa
    import common.abstract_recommender  # noqa
    import common.loss  # noqa
    for n in ("", ".abstract_recommender", ".loss"):  # layergcn.py:12-13
        sys.modules["models.common" + n] = sys.modules["common" + n]


def sample_rows(n, k, seed):
    return np.sort(np.random.default_rng(seed).choice(n, min(k, n), replace=False)).astype(np.int64)


def grad_fingerprint(out, prefix, model, batch_items):
    """per parameter: ||g||_F, sum(g), and sampled rows (for the trainable feature tables: rows of batch items)"""
    for name, p in model.named_parameters():
        g = p.grad
        if g is None:
            continue
        key = prefix + "g_" + name
        out[key + "_norm"] = np.float64(g.double().norm().item())
        out[key + "_sum"] = np.float64(g.double().sum().item())
        if g.dim() == 2 and g.shape[0] > 4096:
            rows = batch_items[:64] if g.shape[1] > 64 else sample_rows(g.shape[0], N_SAMPLE, 3)
            out[key + "_rows"] = np.asarray(rows, dtype=np.int64)
            out[key + "_vals"] = g[torch.as_tensor(rows)].numpy().copy()
        else:
            out[key + "_vals"] = g.numpy().copy()


def evaluate(out, prefix, config, model, valid_data, Trainer):
    trainer = Trainer(config, model)
    res = trainer.evaluate(valid_data)
    keys = sorted(res)
    out[prefix + "metric_keys"] = np.array(keys)
    out[prefix + "metrics"] = np.array([res[k] for k in keys], dtype=np.float64)
    model.eval()
    tops = []
    with torch.no_grad():
        for users, mask in valid_data:
            scores = model.full_sort_predict([users, mask])
            scores[mask[0], mask[1]] = -1e10
            tops.append(torch.topk(scores, 50, dim=-1)[1])
    top = torch.cat(tops).numpy()
    rows = sample_rows(top.shape[0], 512, 4)
    out[prefix + "topk_rows"], out[prefix + "topk"] = rows, top[rows].astype(np.int32)


def main():
    sys.path.insert(0, ROOT)
    from mmrec_amd import synth
    tmp = tempfile.mkdtemp(prefix="mmrec_golden_shapes_")
    for ds in ("sports", "clothing"):
        print(ds, synth.write_dataset(tmp, ds, seed=0), flush=True)
    install_shims()
    os.chdir(REF_SRC)  # Config reads ./configs (configurator.py:72-73); nothing is written here
    from utils.configurator import Config
    from utils.dataset import RecDataset
    from utils.dataloader import TrainDataLoader, EvalDataLoader
    from utils.utils import init_seed, get_model
    from common.trainer import Trainer
    torch.set_num_threads(os.cpu_count() or 1)
    out = {}

    def setup(model_name, ds, extra):
        cd = {"gpu_id": 0, "use_gpu": False, "data_path": tmp + "/", "save_recommended_topk": False, "epochs": 1}
        cd.update(extra)
        config = Config(model_name, ds, cd)
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
        return config, train_data, valid_data

    # ------------------------------------------------------------------ config 3: FREEDOM / Sports shape
    config, train_data, valid_data = setup("FREEDOM", "sports", {"dropout": 0.8, "reg_weight": 1e-3})
    model = get_model("FREEDOM")(config, train_data)
    mm = model.mm_adj                                  # the frozen item-item graph as the reference caches it (uncoalesced)
    mi, mv = mm._indices().numpy(), mm._values().numpy().astype(np.float32)
    assert model.n_items < 65536
    out["fr_mm_idx"], out["fr_mm_vals"] = mi.astype(np.uint16), mv              # entries in the reference's storage order
    keep = {}
    real_multinomial = torch.multinomial

    def recording_multinomial(w, n, *a, **k):
        keep["idx"] = real_multinomial(w, n, *a, **k)
        return keep["idx"]
    torch.multinomial = recording_multinomial
    model.pre_epoch_processing()                      # the reference's own draw (CPU generator, seed 999 stream)
    torch.multinomial = real_multinomial
    batch = next(iter(train_data))
    out["fr_keep_idx"] = keep["idx"].numpy().astype(np.int32)
    out["fr_batch"] = batch.numpy().astype(np.int64)
    urows, irows = sample_rows(model.n_users, N_SAMPLE, 1), sample_rows(model.n_items, N_SAMPLE, 2)
    with torch.no_grad():
        u, i = model.forward(model.masked_adj)
    out["fr_rows_u"], out["fr_rows_i"] = urows, irows
    out["fr_user_out"], out["fr_item_out"] = u[urows].numpy().copy(), i[irows].numpy().copy()
    out["fr_init_user"] = model.user_embedding.weight.detach()[urows].numpy().copy()     # same seed -> same init?
    out["fr_init_image_W"] = model.image_trs.weight.detach()[:, :64].numpy().copy()
    loss = model.calculate_loss(batch)
    loss.backward()
    out["fr_loss"] = np.float64(loss.item())
    grad_fingerprint(out, "fr_", model, np.unique(batch[1:].numpy().reshape(-1)))
    print("FREEDOM/sports loss", loss.item(), flush=True)
    model.zero_grad()
    evaluate(out, "fr_", config, model, valid_data, Trainer)
    print("FREEDOM/sports metrics", dict(zip(out["fr_metric_keys"], out["fr_metrics"])), flush=True)
    del model

    # ------------------------------------------------------------------ config 4: BM3 / Clothing shape
    config, train_data, valid_data = setup("BM3", "clothing", {"n_layers": 2, "dropout": 0.3, "reg_weight": 0.1})
    model = get_model("BM3")(config, train_data)
    import models.bm3 as bm3mod
    import torch.nn.functional as F
    masks, real_dropout = [], F.dropout

    def recording_dropout(x, p=0.5, training=True, inplace=False):
        y = real_dropout(x, p, training, inplace)
        masks.append((y != 0) | (x == 0))             # kept positions (an exact zero input stays kept)
        return y
    bm3mod.F.dropout = recording_dropout
    batch = next(iter(train_data))
    loss = model.calculate_loss(batch)
    bm3mod.F.dropout = real_dropout
    loss.backward()
    assert len(masks) == 4, len(masks)                 # u, i, t, v (bm3.py:110-119)
    for nm, mk in zip("uitv", masks):
        out["bm3_mask_" + nm] = np.packbits(mk.numpy().astype(np.uint8))
        out["bm3_mask_" + nm + "_shape"] = np.array(mk.shape, dtype=np.int64)
    out["bm3_batch"] = batch.numpy().astype(np.int64)
    out["bm3_loss"] = np.float64(loss.item())
    urows = sample_rows(model.n_users, N_SAMPLE, 1)
    out["bm3_rows_u"] = urows
    out["bm3_init_user"] = model.user_embedding.weight.detach()[urows].numpy().copy()
    grad_fingerprint(out, "bm3_", model, np.unique(batch[1].numpy()))
    print("BM3/clothing loss", loss.item(), flush=True)
    model.zero_grad()
    evaluate(out, "bm3_", config, model, valid_data, Trainer)
    print("BM3/clothing metrics", dict(zip(out["bm3_metric_keys"], out["bm3_metrics"])), flush=True)
    np.savez_compressed(os.path.join(HERE, "shapes.npz"), **out)
    print("wrote", os.path.join(HERE, "shapes.npz"), os.path.getsize(os.path.join(HERE, "shapes.npz")) / 1e6, "MB")


if __name__ == "__main__":
    main()
