"""CPU: the host logic of tests/test_config_shapes_gpu.py's model-step and top-K cases on MINIATURE shapes, with the op
entry points of `mmrec_amd.hip_ops` swapped for the torch-CPU restatements of tests/_cpu_ops.py (test-only; the product
has no CPU path) -- so that the bodies the driver runs on the MI355X at full size are known to be sound."""
import pytest

import tests.test_config_shapes_gpu as S
from tests._cpu_ops import cpu_ops  # noqa: F401  (fixture)
from tests.test_config_shapes_gpu import (  # noqa: F401  (collected here without the module's gpu mark)
    test_bm3_step_at_clothing_shape, test_freedom_step_at_sports_shape, test_score_topk_c5_block_vs_oracle,
    test_knn_graph_at_sports_item_count)
from tests.test_config_shapes_gpu import (  # noqa: F401  (golden at full Amazon-Baby shape: small enough for the CPU stand-ins)
    test_lattice_step_vs_reference_golden_at_baby_shape, test_mmgcn_step_vs_reference_golden_at_baby_shape,
    test_eval_rows_of_128_at_baby_shape, test_trainer_warm_evaluation_equals_cold)


@pytest.fixture(autouse=True)
def _mini(cpu_ops, monkeypatch):  # noqa: F811
    from mmrec_amd import synth
    monkeypatch.setattr(S, "USE_GPU", False)
    monkeypatch.setitem(synth.SHAPES, "sports", (420, 150, 4200, 0.74))
    monkeypatch.setitem(synth.SHAPES, "clothing", (380, 170, 3600, 0.74))


def test_oracle_freedom_step_at_sports_shape_matches_the_reference(tmp_path, monkeypatch):
    """Pins the ORACLE at a benchmark config's size: orc.freedom_loss on the Amazon-Sports-shaped synthetic dataset (full
    size: 35,598 x 18,357, 4096-d features) reproduces what the unmodified reference recorded in tests/golden/shapes.npz --
    loss and every gradient (norm, sampled rows) -- from the same initial parameters, batch, multinomial draw and item-item
    graph.  (The device tests then compare the HIP path with the oracle AND with the same golden.)"""
    import numpy as np
    import torch
    from mmrec_amd import synth
    from oracle import mmrec_oracle as orc
    monkeypatch.setitem(synth.SHAPES, "sports", (35598, 18357, 296337, 0.74))       # undo the autouse miniature
    g = S._shapes_golden()
    config, train_data, valid_data, model = S.build_shape(tmp_path, "FREEDOM", "sports",
                                                          {"dropout": 0.8, "reg_weight": 1e-3, "lazy_feature_adam": False})
    batch = next(iter(train_data))
    np.testing.assert_array_equal(batch.numpy(), g["fr_batch"])
    nu, ni = model.n_users, model.n_items
    p = S.cpu_leaves(model)
    np.testing.assert_array_equal(p["user_embedding.weight"].detach()[g["fr_rows_u"]].numpy(), g["fr_init_user"])
    a_idx, a_val = orc.masked_adj_coo(model.edge_indices.numpy(), g["fr_keep_idx"].astype(np.int64), nu, ni)
    adj = orc.sparse_coo(a_idx, a_val, nu + ni)
    mm = orc.sparse_coo(g["fr_mm_idx"].astype(np.int64), g["fr_mm_vals"], ni, ni)
    loss = orc.freedom_loss(adj, mm, p["user_embedding.weight"], p["item_id_embedding.weight"], p["image_embedding.weight"],
                            p["image_trs.weight"], p["image_trs.bias"], p["text_embedding.weight"], p["text_trs.weight"],
                            p["text_trs.bias"], 2, 1, batch.numpy(), 1e-3)
    loss.backward()
    np.testing.assert_allclose(float(loss.detach()), float(g["fr_loss"]), rtol=1e-6)
    for name, t in p.items():
        key = "fr_g_" + name
        if name.endswith("trs.bias"):
            continue                                   # analytically zero (rounding noise on both sides)
        norm = float(t.grad.double().norm())
        assert abs(norm - g[key + "_norm"]) <= 1e-5 * g[key + "_norm"], (name, norm, g[key + "_norm"])
        vals = t.grad[torch.as_tensor(g[key + "_rows"])] if key + "_rows" in g else t.grad
        ref = g[key + "_vals"]
        np.testing.assert_allclose(vals.numpy(), ref, rtol=1e-4, atol=1e-5 * float(np.abs(ref).max()), err_msg=name)
