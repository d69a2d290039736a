"""GPU: randomized differential test of the modal projection (nn.Linear, out = 64: freedom.py:205,208; bm3.py:102-104;
vbpr.py:70) -- forward, dW, db, dX through `hip_ops.linear` on the split-operand kernels (the default) against float64, on
shapes and magnitudes drawn per case: n in [1, 5000], F in {128, 256, 384, 512, 1024, 4096, 4480} (+ widths the split kernels
do not serve: 96, 40 -> fp32 kernels), rows / columns / single elements of X, W and dY scaled over 1e-20 ... 1e20, exact zeros,
sparse rows, an occasional inf / NaN.  Acceptance per output element: |err| <= tol * sum |a_k b_k| of that element (what a
correctly rounded fp32 accumulation is measured against), tol = 2e-6, or 3e-5 in the cases that scale something UP by more than
100 (a few products then dominate their sums and every later addition of ANY fp32 chain rounds at their ulp); non-finite outputs exactly where float64
(evaluated with fp32's range: |v| > 3.4e38 -> inf) has them.  The CPU twin below checks that the checker itself catches
planted errors."""
import numpy as np
import pytest
import torch

CASES = 150
WIDTHS = (128, 256, 384, 512, 1024, 4096, 4480, 96, 40)


def draw_case(seed):
    rng = np.random.default_rng(seed)
    F = int(WIDTHS[seed % len(WIDTHS)])
    n = int(rng.choice([1, 2, 31, 33, 127, 129, 500, 2049, 5000])) if seed % 3 else int(rng.integers(1, 3000))
    if F >= 4096:
        n = min(n, 2049)
    X = np.maximum(rng.standard_normal((n, F)), 0) if seed % 2 else rng.standard_normal((n, F))
    W = rng.standard_normal((64, F)) / np.sqrt(F)
    dY = rng.standard_normal((n, 64)) * 10.0 ** rng.uniform(-9, -1)
    b = rng.standard_normal(64)

    dominant = [False]

    def scale_some(a, axis, lo, hi):
        k = a.shape[axis]
        idx = rng.choice(k, size=min(k, int(rng.integers(0, 4))), replace=False)
        for i in idx:
            f = 10.0 ** rng.uniform(lo, hi)
            dominant[0] |= f > 100.0
            if axis == 0:
                a[i] *= f
            else:
                a[:, i] *= f
    mode = seed % 5
    if mode >= 1:
        scale_some(X, 0, -20, 4), scale_some(X, 1, -9, 4)
        scale_some(W, 0, -9, 2), scale_some(W, 1, -9, 3)
        scale_some(dY, 0, -12, 20), scale_some(dY, 1, -10, 6)
    if mode >= 2:
        X[rng.random(X.shape) < 0.3] = 0.0
        dY[rng.random(n) < 0.5] = 0.0
        if F > 3:
            W[:, int(rng.integers(0, F))] = 0.0
    if mode == 3 and n > 2:
        X[int(rng.integers(0, n)), int(rng.integers(0, F))] = rng.choice([1e5, 7e4, 3e38, -1e6])
        dominant[0] = True
    if mode == 4 and n > 2:
        which = int(rng.integers(0, 3))
        v = rng.choice([np.inf, -np.inf, np.nan])
        (X, W, dY)[which][int(rng.integers(0, (X, W, dY)[which].shape[0])), 0] = v
    return X.astype(np.float32), W.astype(np.float32), b.astype(np.float32), dY.astype(np.float32), dominant[0]


def check(got, A, B, extra=None, name="", dominant=False):
    """got ~= A @ B (+ extra) element-wise, in the sense of the module docstring; A [m, k], B [k, p] float32 arrays"""
    A64, B64 = A.astype(np.float64), B.astype(np.float64)
    with np.errstate(invalid="ignore", over="ignore"):
        ref = A64 @ B64
        mag = np.abs(A64) @ np.abs(B64)
        if extra is not None:
            ref = ref + extra.astype(np.float64)
            mag = mag + np.abs(extra.astype(np.float64))
    ref32 = np.where(np.abs(ref) > 3.4028234e38, np.sign(ref) * np.inf, ref)
    fin = np.isfinite(ref32) & np.isfinite(mag)
    got = np.asarray(got, dtype=np.float64)
    bad_pattern = np.isfinite(got) != fin
    # (a finite reference next to an overflowing intermediate: fp32 may legitimately overflow where float64 does not)
    near_overflow = mag > 1e37
    assert not (bad_pattern & ~near_overflow).any(), (name, "non-finite pattern", int((bad_pattern & ~near_overflow).sum()))
    ok = fin & ~near_overflow
    tol = 3e-5 if dominant else 2e-6
    with np.errstate(invalid="ignore"):
        err = np.abs(got - ref32)
        viol = ok & (err > tol * mag + 1e-40)        # (+ a floor far below fp32's smallest normal: sums in the denormal range round coarsely in ANY fp32 chain)
    assert not viol.any(), (name, int(viol.sum()), float(np.nanmax(np.where(ok, err / (mag + 1e-300), 0))))


def test_checker_catches_planted_errors():
    """CPU twin: the acceptance rule passes an fp32 GEMM and rejects a planted 1e-4 error, a dropped row, a wrong non-finite pattern"""
    X, W, b, dY, _ = draw_case(7)
    Y = (X.astype(np.float64) @ W.astype(np.float64).T + b).astype(np.float32)
    check(Y, X, W.T.copy(), extra=np.broadcast_to(b, Y.shape), name="clean")
    bad = Y.copy()
    bad[0, 0] *= 1.0 + 1e-4
    with pytest.raises(AssertionError):
        check(bad, X, W.T.copy(), extra=np.broadcast_to(b, Y.shape))
    bad = Y.copy()
    bad[-1] = 0
    if np.abs(Y[-1]).max() > 0:
        with pytest.raises(AssertionError):
            check(bad, X, W.T.copy(), extra=np.broadcast_to(b, Y.shape))
    bad = Y.copy()
    bad[0, 1] = np.inf
    with pytest.raises(AssertionError):
        check(bad, X, W.T.copy(), extra=np.broadcast_to(b, Y.shape))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(CASES))
def test_linear_fuzz(seed, monkeypatch):
    from mmrec_amd import hip_ops
    # the split-operand forward at every width it serves, as in rounds 4-5 (the default routes inputs narrower than 1024 columns
    # to the fp32 kernel)
    monkeypatch.setattr(hip_ops, "LINEAR_SPLIT_MIN_F", 0)
    dev = torch.device("cuda:0")
    X, W, b, dY, dom = draw_case(seed)
    use_b = seed % 4 != 0
    Xd = torch.from_numpy(X).to(dev).requires_grad_()
    Wd = torch.from_numpy(W).to(dev).requires_grad_()
    bd = torch.from_numpy(b).to(dev).requires_grad_() if use_b else None
    Y = hip_ops.linear(Xd, Wd, bd)
    Y.backward(torch.from_numpy(dY).to(dev))
    torch.cuda.synchronize()
    with np.errstate(invalid="ignore", over="ignore"):
        check(Y.detach().cpu().numpy(), X, W.T.copy(), extra=np.broadcast_to(b, (X.shape[0], 64)) if use_b else None,
              name="Y seed %d" % seed, dominant=dom)
        check(Wd.grad.cpu().numpy(), dY.T.copy(), X, name="dW seed %d" % seed, dominant=dom)
        check(Xd.grad.cpu().numpy(), dY, W, name="dX seed %d" % seed, dominant=dom)
        if use_b:
            check(bd.grad.cpu().numpy()[None, :], np.ones((1, dY.shape[0]), np.float32), dY, name="db seed %d" % seed, dominant=dom)
