"""CPU: the C-ABI library loads and exports exactly what include/mmrec_hip.h declares; host-only
entry points (plan helpers, workspace sizes) behave.  No GPU compute is called here."""
import ctypes
import os
import re

import numpy as np
import pytest

from mmrec_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "mmrec_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mmrec_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        from mmrec_amd.build import build
        build(verbose=False)
    return _lib.load()


def test_header_symbols_exported_and_typed(lib):
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "library does not export %s" % n
        assert n in _lib.SIGNATURES, "ctypes binding lacks %s" % n
    assert sorted(_lib.SIGNATURES) == names
    assert lib.mmrec_abi_version() == _lib.ABI_VERSION
    assert lib.mmrec_error_string(0) == b"ok"
    assert b"bad argument" in lib.mmrec_error_string(10001)


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libmmrec_hip.so")
    with pytest.raises(_lib.MMRecHipError):
        _lib.load()


def test_cpu_tensor_is_rejected():
    import torch
    from mmrec_amd import hip_ops
    with pytest.raises(_lib.MMRecHipError):
        hip_ops.gather_sqnorm(torch.zeros(4, 64), torch.zeros(2, dtype=torch.int64))


def test_spmm_plan_host(lib):
    deg = np.array([0, 3, 300, 5000, 256, 257, 2048 + 1], dtype=np.int64)
    rp = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
    nl, nc = ctypes.c_int32(), ctypes.c_int32()
    assert lib.mmrec_spmm_plan_count(rp.ctypes.data_as(ctypes.c_void_p), len(deg), 256,
                                     ctypes.byref(nl), ctypes.byref(nc)) == 0
    assert nl.value == 4 and nc.value == 1 + 10 + 1 + 5   # ceil(deg / 512)
    lr = np.empty(nl.value, np.int32)
    cp = np.empty(nl.value + 1, np.int32)
    assert lib.mmrec_spmm_plan_fill(rp.ctypes.data_as(ctypes.c_void_p), len(deg), 256,
                                    lr.ctypes.data_as(ctypes.c_void_p),
                                    cp.ctypes.data_as(ctypes.c_void_p)) == 0
    assert lr.tolist() == [2, 3, 5, 6] and cp.tolist() == [0, 1, 11, 12, 17]
    assert lib.mmrec_spmm_plan_count(None, 3, 256, ctypes.byref(nl), ctypes.byref(nc)) == 10001


def test_workspace_sizes(lib):
    assert lib.mmrec_bpr_workspace_bytes(2048) == 2048 * 4
    assert lib.mmrec_linear_workspace_bytes(7050, 4096, 64) > 0
    assert lib.mmrec_linear_workspace_bytes(7050, 4096, 32) == 0


def test_argument_errors_without_gpu(lib):
    # argument validation happens on the host before any launch
    assert lib.mmrec_spmm_csr_f32(None, None, None, None, None, None, None, None, 10, 24, 1.0, 0.0, 1.0,
                                  256, None, None, 0, 0, None, None, None) == 10002   # d: not a multiple of 64, not a slice width
    assert lib.mmrec_spmm_csr_f32(None, None, None, None, None, None, None, None, 10, 32, 1.0, 0.0, 1.0,
                                  256, None, None, 0, 0, None, None, None) == 10001   # d = 32 is a feature slice: NULL pointers
    assert lib.mmrec_score_topk_f32(None, None, 4, 10, 64, None, None, 65, None, None, None, 0, None) == 10001
    assert lib.mmrec_score_topk_f32(None, None, 4, 10, 64, None, None, 5, None, None, None, 2, None) == 10001   # unknown flag
    assert lib.mmrec_linear_fwd_f32(None, None, None, None, 4, 6, 64, None, None) == 10002  # F % 4
    # the product at listed rows and its transposed push (ABI 10): widths 8 / 16 / 32 / 64
    assert lib.mmrec_spmm_rows_f32(None, None, None, None, None, 0, None, 5, 128, 32, None, None) == 10002
    assert lib.mmrec_spmm_rows_f32(None, None, None, None, None, 0, None, 5, 64, 32, None, None) == 10001
    assert lib.mmrec_spmm_rows_f32(None, None, None, None, None, 0, None, 0, 64, 32, None, None) == 0          # empty list
    assert lib.mmrec_spmm_push_rows_f32(None, None, None, None, 1.0, None, 5, 24, None, None, None) == 10002
    assert lib.mmrec_spmm_push_rows_f32(None, None, None, None, 1.0, None, 5, 16, None, None, None) == 10001
    # sampled scoring on a column slice: widths 8 / 16 / 32 / 64 k, nothing else
    assert lib.mmrec_bpr_dots_f32(None, None, None, None, None, None, 5, 24, None, None) == 10002
    assert lib.mmrec_bpr_dots_f32(None, None, None, None, None, None, 5, 16, None, None) == 10001
    assert lib.mmrec_bpr_dots_f32(None, None, None, None, None, None, 0, 8, None, None) == 0            # empty batch
    assert lib.mmrec_bpr_loss_from_dots_f32(None, 5, 7, 1.0, None, None, None, None) == 10001           # unknown variant
    assert lib.mmrec_bpr_bwd_f32(None, None, None, None, None, None, 5, 40, None, None, 1.0, None, None, None, None) == 10002
    assert lib.mmrec_bpr_bwd_f32(None, None, None, None, None, None, 5, 8, None, None, 1.0, None, None, None, None) == 10001


def test_topk_workspace_covers_both_kd64_paths(lib):
    """kd = 64 calls may be served by the fp16 filter path or the materialised path (the `flags` argument
    decides at launch): the workspace query must cover both, grow with the problem, and stay far below the 4 nq nc bytes of a
    full score matrix at evaluation scale."""
    prev = 0
    for nq, nc in ((100, 1500), (4096, 7050), (19445, 7050), (39387, 23033)):
        b = lib.mmrec_topk_workspace_bytes(nq, nc, 64, 50)
        assert b > 0 and b >= prev
        prev = b
    big = lib.mmrec_topk_workspace_bytes(20000, 500000, 64, 50)
    assert 0 < big <= 9 * 2 ** 30            # one 8 GB score block of the materialised path at most
    assert lib.mmrec_topk_workspace_bytes(0, 10, 64, 5) == 0
