"""Test helper: materialise the golden tiny dataset on disk and build config/loaders with OUR
plumbing (mmrec_amd.utils.*), mirroring what tests/golden/make_golden.py did with the reference's."""
import os

import numpy as np


def observed(name, value, floor):
    """A set-agreement fraction and the floor it is asserted against, appended to $MMREC_TEST_OBSERVED (a file; the build
    sets it on its GPU runs) so that floors stay within 10 x of the observed miss rate instead of drifting loose
    (round-3 review, weak 1d).  Returns `value`."""
    path = os.environ.get("MMREC_TEST_OBSERVED")
    if path:
        with open(path, "a") as f:
            f.write("%s\t%.6f\t%.6f\n" % (name, float(value), float(floor)))
    return value


def write_dataset(root, golden):
    ds = os.path.join(str(root), "baby")
    os.makedirs(ds, exist_ok=True)
    with open(os.path.join(ds, "baby.inter"), "w") as f:
        f.write("userID\titemID\trating\ttimestamp\tx_label\n")
        for u, i, lb in golden["inter"]:
            f.write("%d\t%d\t5.0\t0\t%d\n" % (u, i, lb))
    np.save(os.path.join(ds, "image_feat.npy"), golden["image_feat"])
    np.save(os.path.join(ds, "text_feat.npy"), golden["text_feat"])
    return str(root) + "/"


def setup(root, golden, model_name, extra, use_gpu=False, batch=256, seed=999):
    from mmrec_amd.utils.configurator import Config
    from mmrec_amd.utils.dataloader import EvalDataLoader, TrainDataLoader
    from mmrec_amd.utils.dataset import RecDataset
    from mmrec_amd.utils.utils import init_seed
    data_path = write_dataset(root, golden)
    cd = {"gpu_id": 0, "use_gpu": use_gpu, "data_path": data_path, "train_batch_size": batch,
          "save_recommended_topk": False, "epochs": 1}
    cd.update(extra)
    config = Config(model_name, "baby", cd)
    for k, v in extra.items():
        config[k] = v
    config["seed"] = seed
    dataset = RecDataset(config)
    str(dataset)
    tr, va, te = dataset.split()
    str(tr), str(va), str(te)
    train_data = TrainDataLoader(config, tr, batch_size=batch, shuffle=True)
    valid_data = EvalDataLoader(config, va, additional_dataset=tr, batch_size=config["eval_batch_size"])
    init_seed(seed)
    train_data.pretrain_setup()
    return config, train_data, valid_data


def _whole_runs():
    """the hyper-parameters tests/golden/make_golden_trajectories.py ran the reference with"""
    import ast
    src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "make_golden_trajectories.py")).read()
    return ast.literal_eval(src[src.index("RUNS = {") + 7:src.index("\n}\n", src.index("RUNS = {")) + 2])


WHOLE_RUNS = _whole_runs()


def whole_run(root, golden, run, use_gpu):
    """`Trainer.fit` (or, for an untrained model, two evaluations) of one entry of WHOLE_RUNS with OUR stack, set up in
    the order the golden script used with the reference's -> (losses, valid metrics, test metrics, reference dict)."""
    import torch
    from mmrec_amd.common.trainer import Trainer
    from mmrec_amd.utils.configurator import Config
    from mmrec_amd.utils.dataloader import EvalDataLoader, TrainDataLoader
    from mmrec_amd.utils.dataset import RecDataset
    from mmrec_amd.utils.utils import get_model, init_seed
    here = os.path.dirname(os.path.abspath(__file__))
    traj = dict(np.load(os.path.join(here, "golden", "trajectories.npz")))
    ug = dict(np.load(os.path.join(here, "golden", "dualgnn.npz")))
    name, mirror = run.split("+")[0], run.endswith("+mg")     # "+mg": the Mirror-Gradient trainer variant
    data_path = write_dataset(root, golden)
    ds = os.path.join(str(root), "baby")
    rp, ids, cnt = ug["ug_rowptr"], ug["ug_ids"].tolist(), ug["ug_cnt"].tolist()
    np.save(os.path.join(ds, "user_graph_dict.npy"),
            {u: [ids[rp[u]:rp[u + 1]], cnt[rp[u]:rp[u + 1]]] for u in range(len(rp) - 1)}, allow_pickle=True)
    np.save(os.path.join(ds, "item_graph_dict_2.npy"),
            {i: [[(i + 1) % 90, (i + 7) % 90], [1.0, 1.0]] for i in range(0, 90, 2)}, allow_pickle=True)
    if run.endswith("+img") or run.endswith("+txt"):            # single-modality dataset: only that feature file exists
        os.remove(os.path.join(ds, "text_feat.npy" if run.endswith("+img") else "image_feat.npy"))
    cd = dict(dict(epochs=3, train_batch_size=256), **dict(WHOLE_RUNS[run], gpu_id=0, use_gpu=use_gpu, data_path=data_path,
                                                           save_recommended_topk=False))
    config = Config(name, "baby", cd, mirror)
    for k, v in cd.items():
        config[k] = v
    config["seed"] = 999
    init_seed(999)
    dataset = RecDataset(config)
    str(dataset)
    tr, va, te = dataset.split()
    str(tr), str(va), str(te)
    train_data = TrainDataLoader(config, tr, batch_size=config["train_batch_size"], shuffle=True)
    valid_data = EvalDataLoader(config, va, additional_dataset=tr, batch_size=config["eval_batch_size"])
    test_data = EvalDataLoader(config, te, additional_dataset=tr, batch_size=config["eval_batch_size"])
    init_seed(999)
    train_data.pretrain_setup()
    model = get_model(name)(config, train_data).to(config["device"])
    assert config["device"].type == ("cuda" if use_gpu else "cpu")
    trainer = Trainer(config, model, mirror)
    keys = [str(k) for k in traj[run + "_metric_keys"]]
    ref = {"losses": traj[run + "_losses"], "valid": traj[run + "_valid"], "test": traj[run + "_test"]}
    if not config["req_training"]:
        res_v, res_t = trainer.evaluate(valid_data), trainer.evaluate(test_data)
        return np.zeros(0), np.array([res_v[k] for k in keys]), np.array([res_t[k] for k in keys]), ref
    _, best_valid, best_test = trainer.fit(train_data, valid_data=valid_data, test_data=test_data, saved=False, verbose=False)
    losses = np.array([float(trainer.train_loss_dict[e]) for e in sorted(trainer.train_loss_dict)])
    if use_gpu:
        torch.cuda.synchronize()
    return losses, np.array([best_valid[k] for k in keys]), np.array([best_test[k] for k in keys]), ref
