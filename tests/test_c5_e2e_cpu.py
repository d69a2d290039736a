"""CPU: the host logic of tests/test_c5_e2e_gpu.py on a MINIATURE shape (1,500 users x 600 items), with the op entry points
of `mmrec_amd.hip_ops` swapped for the torch-CPU restatements of tests/_cpu_ops.py (test-only; the product has no CPU path)
and a one-rank gloo group in place of RCCL -- so that the body the driver runs on the MI355X at the full 1M / 500K / 10M
size is known to be sound: in-memory dataset -> loaders -> FREEDOM step vs the oracle composition -> sampled evaluation ->
ShardedFREEDOM."""
import pytest

import tests.test_c5_e2e_gpu as E
from tests._cpu_ops import cpu_ops  # noqa: F401  (fixture)
from tests.test_c5_e2e_gpu import test_freedom_c5_step_and_recall_vs_oracle  # noqa: F401  (collected without the gpu mark)


@pytest.fixture(autouse=True)
def _mini(cpu_ops, monkeypatch):  # noqa: F811
    monkeypatch.setattr(E, "USE_GPU", False)
    monkeypatch.setattr(E, "SHAPE", dict(n_users=1500, n_items=600, n_train=15000, n_eval=600, image_dim=128, text_dim=64,
                                         sample_users=400))
    monkeypatch.setattr(E, "_CACHE", {})
