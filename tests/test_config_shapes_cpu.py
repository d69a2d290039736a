"""CPU: the host logic of tests/test_config_shapes_gpu.py's model-step and top-K cases on MINIATURE shapes, with the op
entry points of `mmrec_amd.hip_ops` swapped for the torch-CPU restatements of tests/_cpu_ops.py (test-only; the product
has no CPU path) -- so that the bodies the driver runs on the MI355X at full size are known to be sound."""
import pytest

import tests.test_config_shapes_gpu as S
from tests._cpu_ops import cpu_ops  # noqa: F401  (fixture)
from tests.test_config_shapes_gpu import (  # noqa: F401  (collected here without the module's gpu mark)
    test_bm3_step_at_clothing_shape, test_freedom_step_at_sports_shape, test_score_topk_c5_block_vs_oracle,
    test_knn_graph_at_sports_item_count)


@pytest.fixture(autouse=True)
def _mini(cpu_ops, monkeypatch):  # noqa: F811
    from mmrec_amd import synth
    monkeypatch.setattr(S, "USE_GPU", False)
    monkeypatch.setitem(synth.SHAPES, "sports", (420, 150, 4200, 0.74))
    monkeypatch.setitem(synth.SHAPES, "clothing", (380, 170, 3600, 0.74))
