"""CPU: our config / dataset / loaders / evaluator reproduce the reference's outputs bit for bit
on the golden tiny dataset (ids, masks, interaction matrix: integer work, exact)."""
import numpy as np
import torch

from tests._env import setup


def test_config_merge_and_quirks(tmp_path, golden):
    config, _, _ = setup(tmp_path, golden, "FREEDOM", {"dropout": 0.8, "reg_weight": 1e-3})
    assert config["embedding_size"] == 64 and config["knn_k"] == 10 and config["n_ui_layers"] == 2
    assert config["no_such_key"] is None                       # missing -> None, not KeyError
    assert config["hyper_parameters"] == ["seed", "dropout", "reg_weight"]     # lists concatenated in file order
    assert config["valid_metric_bigger"] is True and config["topk"] == [5, 10, 20, 50]
    assert isinstance(config["learning_rate"], float) and config["USER_ID_FIELD"] == "userID"


def test_train_loader_batch_identical_to_reference(tmp_path, golden):
    config, train_data, valid_data = setup(tmp_path, golden, "LightGCN", {"n_layers": 3, "reg_weight": 1e-4})
    m = train_data.inter_matrix(form="coo")
    np.testing.assert_array_equal(m.row, golden["train_rows"])
    np.testing.assert_array_equal(m.col, golden["train_cols"])
    batch = next(iter(train_data))
    for _ in train_data:
        pass
    assert batch.dtype == torch.int64
    np.testing.assert_array_equal(batch.numpy(), golden["batch"])   # users, positives AND sampled negatives
    # second pass works (pointer reset) and yields a different shuffle
    assert not np.array_equal(next(iter(train_data)).numpy(), golden["batch"])


def test_eval_loader_masks_identical(tmp_path, golden):
    _, _, valid_data = setup(tmp_path, golden, "LightGCN", {"n_layers": 3, "reg_weight": 1e-4})
    users, masks = [], []
    for u, m in valid_data:
        users.append(u.numpy()), masks.append(m.numpy())
    np.testing.assert_array_equal(np.concatenate(users), golden["eval_users"])
    np.testing.assert_array_equal(np.concatenate(masks, axis=1), golden["eval_mask"])
    np.testing.assert_array_equal(valid_data.get_eval_len_list(), golden["eval_pos_len"])
    np.testing.assert_array_equal(np.concatenate(valid_data.get_eval_items()), golden["eval_pos_flat"])


def test_evaluator_metrics_identical(tmp_path, golden):
    from mmrec_amd.utils.topk_evaluator import TopKEvaluator
    config, _, valid_data = setup(tmp_path, golden, "LightGCN", {"n_layers": 3, "reg_weight": 1e-4})
    ev = TopKEvaluator(config)
    keys = [str(k) for k in golden["metric_keys"]]
    for pre in ("lgn", "fr"):
        res = ev.evaluate([torch.from_numpy(golden[pre + "_topk"])], valid_data)
        assert list(res.keys()) == keys
        np.testing.assert_array_equal([res[k] for k in keys], golden[pre + "_metrics"])


def test_bm3_loader_has_no_negatives(tmp_path, golden):
    _, train_data, _ = setup(tmp_path, golden, "BM3", {"n_layers": 2, "reg_weight": 0.1, "dropout": 0.3})
    b = next(iter(train_data))
    assert b.shape[0] == 2
    np.testing.assert_array_equal(b.numpy(), golden["batch"][:2])


def test_native_host_sampler_equals_python_loop(tmp_path, golden):
    """mmrec_host_sample_negatives (CPython's Mersenne Twister continued in C) == the reference-form Python loop:
    same negatives AND the same state of the global `random` generator afterwards, from several stream positions
    (incl. across a regeneration of the 624-word block)."""
    import random
    from tests._env import setup
    config, train_data, _ = setup(tmp_path, golden, "LightGCN", {"n_layers": 3, "reg_weight": 1e-4})
    rng = np.random.default_rng(0)
    n_users = int(golden["n_users"])
    for trial in range(6):
        random.seed(1000 + trial)
        for _ in range(211 * trial):
            random.random()
        start = random.getstate()
        users = rng.integers(0, n_users, 700)
        a = train_data._sample_neg_ids(users)
        state_a = random.getstate()
        assert train_data._native_sampler not in (None, False)
        random.setstate(start)
        b = train_data._sample_neg_ids_loop(users)
        assert np.array_equal(a, b) and random.getstate() == state_a
        hist = train_data.history_items_per_u
        assert all(int(n) not in hist[int(u)] for u, n in zip(users, a))


def test_sampler_draws_from_the_interpreters_generator_in_place(tmp_path, golden):
    """The C sampler works on the Mersenne Twister state INSIDE `random`'s global generator (probed layout of CPython's
    _random.Random object, utils/dataloader.py:_live_mt_state) instead of a getstate() / setstate() round trip per batch:
    same negatives and same generator state as the marshalling path and as the reference-form Python loop; interleaved
    draws by other code see a generator that is always current."""
    import random
    import sys
    import mmrec_amd.utils.dataloader as DL
    from tests._env import setup
    config, train_data, _ = setup(tmp_path, golden, "LightGCN", {"n_layers": 3, "reg_weight": 1e-4})
    DL._LIVE_MT.clear()
    live = DL._live_mt_state()
    if sys.implementation.name == "cpython" and sys.version_info[:2] == (3, 10):
        assert live is not None                      # the interpreter this image ships: the fast path must be the one in use
    users = np.random.default_rng(1).integers(0, int(golden["n_users"]), 900)
    outs = {}
    for mode in ("live", "marshal", "loop"):
        DL._LIVE_MT.clear()
        if mode != "live":
            DL._LIVE_MT.append(None)
        random.seed(4242)
        seq = []
        for _ in range(3):                           # sampler calls interleaved with ordinary draws
            fn = train_data._sample_neg_ids_loop if mode == "loop" else train_data._sample_neg_ids
            seq.append(fn(users).tolist())
            seq.append(random.random())
            seq.append(random.sample(range(1000), 5))
        outs[mode] = (seq, random.getstate())
    DL._LIVE_MT.clear()
    assert outs["live"] == outs["marshal"] == outs["loop"]


def test_user_cooccurrence_graph_matches_reference_script():
    """mmrec_amd.utils.user_graph (one sparse product) vs the dict the reference's preprocessing script
    (O(U^2) Python loop + torch.topk per user) wrote for the same interactions: same counts in the same order,
    same neighbour set at every count level (torch.topk leaves the order inside a tie undefined)."""
    import os
    from mmrec_amd.utils.user_graph import build_user_graph_dict, cooccurrence_topk, pack_user_graph_dict
    here = os.path.dirname(os.path.abspath(__file__))
    g = dict(np.load(os.path.join(here, "golden", "dualgnn.npz")))
    t = dict(np.load(os.path.join(here, "golden", "tiny.npz")))
    tr = t["inter"][t["inter"][:, 2] == 0]
    n = int(t["n_users"])
    rp, ids, cnt = cooccurrence_topk(tr[:, 0], tr[:, 1], n)
    np.testing.assert_array_equal(rp, g["ug_rowptr"])
    np.testing.assert_array_equal(cnt, g["ug_cnt"])
    rows = np.repeat(np.arange(n), np.diff(rp))
    assert set(zip(rows.tolist(), ids.tolist(), cnt.tolist())) == set(zip(rows.tolist(), g["ug_ids"].tolist(), g["ug_cnt"].tolist()))
    # truncation keeps the largest counts; the dict form and its padded first-k view
    rp5, ids5, cnt5 = cooccurrence_topk(tr[:, 0], tr[:, 1], n, top=5)
    assert np.diff(rp5).max() == 5
    for u in (0, 7, n - 1):
        np.testing.assert_array_equal(cnt5[rp5[u]:rp5[u + 1]], cnt[rp[u]:rp[u] + 5])
    d = build_user_graph_dict(tr[:, 0], tr[:, 1], n)
    assert isinstance(d[3][0][0], int) and isinstance(d[3][1][0], float) and len(d) == n
    pid, pcnt, plen = pack_user_graph_dict(d, 40)
    np.testing.assert_array_equal(plen, np.minimum(np.diff(rp), 40))
    np.testing.assert_array_equal(pid[3, :plen[3]], ids[rp[3]:rp[3] + plen[3]])
    # duplicate (user, item) pairs count once; a user without shared items has an empty list
    d2 = build_user_graph_dict([0, 0, 1, 2], [5, 5, 5, 9], 3)
    assert d2[0] == [[1], [1.0]] and d2[1] == [[0], [1.0]] and d2[2] == [[], []]


def test_item_cooccurrence_graph_min_count(tmp_path):
    """the item graph producer for DAMRS: co-occurrence with the id columns swapped, pairs below min_count dropped"""
    from mmrec_amd.utils.user_graph import write_item_graph_file
    f = tmp_path / "x.inter"
    rows = [(0, 0, 0), (0, 1, 0), (1, 0, 0), (1, 1, 0), (1, 2, 0), (2, 2, 0), (2, 3, 1), (3, 0, 2)]
    f.write_text("userID\titemID\tx_label\n" + "".join("%d\t%d\t%d\n" % r for r in rows))
    d = write_item_graph_file(str(f), str(tmp_path / "g.npy"), top=10, min_count=1)
    assert d[0] == [[1, 2], [2.0, 1.0]] and d[2] == [[0, 1], [1.0, 1.0]] and d[3] == [[], []]   # valid/test rows ignored
    d2 = write_item_graph_file(str(f), str(tmp_path / "g2.npy"), top=10, min_count=2)
    assert d2[0] == [[1], [2.0]] and d2[2] == [[], []]
    assert np.load(str(tmp_path / "g2.npy"), allow_pickle=True).item() == d2


def test_cooccurrence_topk_vs_pairwise_set_intersections():
    """random small interaction sets: the blocked sparse construction == the reference script's definition (size of
    the intersection of two users' item sets, every pair, diagonal excluded), incl. duplicates, empty users and
    truncation by `top`"""
    from mmrec_amd.utils.user_graph import cooccurrence_topk
    rng = np.random.default_rng(5)
    for trial in range(25):
        n_users, n_items = int(rng.integers(1, 40)), int(rng.integers(1, 15))
        m = int(rng.integers(0, 120))
        users, items = rng.integers(0, n_users, m), rng.integers(0, n_items, m)
        top = int(rng.integers(1, 8))
        sets = [set(items[users == u].tolist()) for u in range(n_users)]
        rp, ids, cnt = cooccurrence_topk(users, items, n_users, top=top)
        for u in range(n_users):
            pairs = sorted(((len(sets[u] & sets[v]), v) for v in range(n_users) if v != u and sets[u] & sets[v]),
                           key=lambda t: (-t[0], t[1]))[:top]
            np.testing.assert_array_equal(ids[rp[u]:rp[u + 1]], [v for _, v in pairs])
            np.testing.assert_array_equal(cnt[rp[u]:rp[u + 1]], [float(c) for c, _ in pairs])
