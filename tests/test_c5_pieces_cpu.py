"""CPU: the bodies of tests/test_c5_pieces_gpu.py on a MINIATURE shape (3,000 users x 1,200 items) with the op entry points of
`mmrec_amd.hip_ops` swapped for the torch-CPU restatements of tests/_cpu_ops.py (test-only; the product has no CPU path):
the kNN-rows oracle, the trained-shaped tables, the mask construction and the differential checker are known to be sound
before the MI355X runs them at 500,000 items."""
import pytest

import tests.test_c5_pieces_gpu as P
from tests._cpu_ops import cpu_ops  # noqa: F401  (fixture)
from tests.test_c5_pieces_gpu import (test_knn_graph_at_c5_item_count,  # noqa: F401  (collected without the gpu mark)
                                      test_trained_shaped_eval_block_vs_oracle)

pytestmark = []


@pytest.fixture(autouse=True)
def _mini(cpu_ops, monkeypatch):  # noqa: F811
    monkeypatch.setattr(P, "USE_GPU", False)
    monkeypatch.setattr(P, "SHAPE", dict(n_users=3000, n_items=1200, n_edges=30000, block=700, sample_rows=200,
                                         sample_users=150, image_dim=256, text_dim=64, heavy=((60, 6), (500, 2))))
