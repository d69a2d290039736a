"""CPU: generator + checker of tests/test_topk_fuzz_gpu.py on scaled-down cases (candidate counts x 0.02) against the
torch-CPU stand-in op of tests/_cpu_ops.py -- the body the GPU box runs is known to be sound, and the checker is known to
have teeth: planted errors (a dropped best candidate, a masked id, a wrong tie order, a value off by 1e-5) are caught."""
import numpy as np
import pytest
import torch

import tests.test_topk_fuzz_gpu as F
from tests._cpu_ops import cpu_ops  # noqa: F401  (fixture)


@pytest.mark.parametrize("seed", range(48))
def test_fuzz_body_on_small_cases(cpu_ops, seed):  # noqa: F811
    from mmrec_amd import hip_ops             # its op entry points are the stand-ins for the duration of this test
    F.check_case(hip_ops, torch.device("cpu"), F.gen_case(seed, scale=0.02, work=3.0e5))


def test_generator_covers_every_plan_switch_and_is_deterministic():
    metas = [F.gen_case(s, meta_only=True) for s in range(F.N_CASES)]   # the shapes the GPU box will run
    ncs = np.array([c["nc"] for c in metas])
    assert all(c["nq"] * c["nc"] <= max(F.WORK, c["nc"]) for c in metas) and max(c["nq"] for c in metas) > 2000
    cases = [F.gen_case(s, scale=0.01, work=2.0e4) for s in range(F.N_CASES)]
    for edge in (4096, 32768, 65536, 131072):
        assert ((ncs < edge) & (ncs >= edge - 70)).any() and ((ncs >= edge) & (ncs < edge + 70)).any(), edge
    assert (ncs < 200).any() and (ncs > 200_000).any()
    assert {c["kd"] for c in metas} == {64, 128, 192, 384} and max(c["k"] for c in metas) == 128 and min(c["k"] for c in metas) == 1
    wide = [c for c in metas if c["kd"] > 128 and c["nc"] >= 4096]
    # wide rows: the fp16 pass (k <= 32), the fp32 block path above (incl. its k > 64 form) -- all occur
    assert any(c["k"] <= 32 for c in wide) and any(32 < c["k"] <= 64 for c in wide) and any(c["k"] > 64 for c in wide)
    assert any(c["k"] > 64 and c["nc"] < 4096 and c["kd"] <= 128 for c in metas)        # select_topk_kernel's second half
    tags = {t for c in cases for t in c["tags"]}
    assert {"qspread", "cspread", "outliers", "ties", "heavy", "bestmasked", "zeroq", "starved", "dupq"} <= tags, tags
    a, b = F.gen_case(7, scale=0.02, work=3.0e5), F.gen_case(7, scale=0.02, work=3.0e5)
    assert np.array_equal(a["Q"], b["Q"]) and np.array_equal(a["mask"], b["mask"]) and a["k"] == b["k"]


class _Tampered:
    """the stand-in op with ONE planted error in row 0"""

    def __init__(self, ops, how):
        self.ops, self.how, self.mask_to_csr = ops, how, ops.mask_to_csr

    def score_topk(self, Q, C, k, rp, col, return_values=True):
        idx, val = self.ops.score_topk(Q, C, k, rp, col, return_values=True)
        idx, val = idx.clone(), val.clone()
        s = Q[0] @ C.t()
        m = col[int(rp[0]):int(rp[1])].long()
        if self.how == "dropped":              # the best candidate replaced by a clearly worse one that was not ranked
            s2 = s.clone()
            s2[m] = float("inf")
            s2[idx[0]] = float("inf")
            worst = int(torch.argmin(s2))
            idx[0, 0], val[0, 0] = worst, s[worst]
            o = torch.argsort(val[0], descending=True, stable=True)
            idx[0], val[0] = idx[0][o], val[0][o]
        elif self.how == "masked":
            idx[0, -1], val[0, -1] = m[0], min(float(s[m[0]]), float(val[0, -1]))
        elif self.how == "value":
            val[0, 0] = val[0, 0] + 1e-5 * float(Q[0].norm() * C[idx[0, 0]].norm())
        elif self.how == "order":
            idx[0, :2] = idx[0, :2].flip(0)
        return idx, val


@pytest.mark.parametrize("how", ["dropped", "masked", "value", "order"])
def test_checker_catches_planted_errors(cpu_ops, how):  # noqa: F811
    rng = np.random.default_rng(5)
    nq, nc, k, kd = 12, 900, 20, 64
    Q = (rng.standard_normal((nq, kd)) * 0.2).astype(np.float32)
    C = (rng.standard_normal((nc, kd)) * 0.2).astype(np.float32)
    if how == "order":
        C[7] = C[3] = Q[0] * 4.0               # query 0's two best candidates tie exactly: 3 must come before 7
    mask = np.stack([np.repeat(np.arange(nq), 5), rng.choice(np.arange(10, nc), (nq, 5)).reshape(-1)])
    case = dict(seed=-1, Q=Q, C=C, k=k, kd=kd, nq=nq, nc=nc, mask=mask, tags=[how])
    from mmrec_amd import hip_ops
    F.check_case(hip_ops, torch.device("cpu"), case)                     # untampered: passes
    with pytest.raises(AssertionError):
        F.check_case(_Tampered(hip_ops, how), torch.device("cpu"), case)
