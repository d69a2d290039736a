"""GPU parity tests proper: every HIP op (through the C ABI) against the CPU oracle on seeded inputs,
against the reference-generated golden vectors, and -- at BASELINE.json's full sizes -- through
size-independent properties.  Tolerances: index work exact; fp32 within 1e-4 relative (north_star).
"""
import numpy as np
import pytest
import torch

from oracle import mmrec_oracle as orc

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-4, 1e-6


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    from mmrec_amd import hip_ops
    return hip_ops


@pytest.fixture(autouse=True)
def _split_forward_at_every_width():
    """hip_ops routes inputs narrower than LINEAR_SPLIT_MIN_F (1024) to the fp32-MFMA forward (fewer launches: round 6); the
    tests of this file keep exercising the split-operand forward at EVERY width it serves, 128 and 384 included."""
    from mmrec_amd import hip_ops
    keep, hip_ops.LINEAR_SPLIT_MIN_F = hip_ops.LINEAR_SPLIT_MIN_F, 0
    yield
    hip_ops.LINEAR_SPLIT_MIN_F = keep


def close(a, b, rtol=RTOL, atol=ATOL):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def rel_fro(a, b):
    a = a.detach().cpu().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, dtype=np.float64)
    b = b.detach().cpu().double().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


def single_rank_rccl_group(dev):
    """one-rank RCCL process group on a free local port; a port the kernel hands out can still be taken by the time the
    store binds it (seen once: EADDRINUSE right after the previous test's group was torn down) -> another port"""
    import os
    import socket
    import torch.distributed as dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for attempt in range(5):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        try:
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
            return
        except dist.DistNetworkError:
            if attempt == 4:
                raise


def golden_graph(ops, g, dev, key="norm_adj", **kw):
    n = int(g["n_users"]) + int(g["n_items"])
    return ops.CsrGraph.from_coo_host(g[key + "_idx"], g[key + "_val"], n, n, dev, symmetric=True, **kw)


def D(a, dev, grad=False):
    t = torch.as_tensor(np.asarray(a)).to(dev)
    return t.requires_grad_() if grad else t


# ---------------------------------------------------------------------------------------- SpMM
def _random_csr(rng, n_rows, n_cols, degs):
    rows = np.repeat(np.arange(n_rows), degs)
    cols = rng.integers(0, n_cols, rows.shape[0])
    vals = rng.standard_normal(rows.shape[0]).astype(np.float32)
    return np.stack([rows, cols]), vals


@pytest.mark.parametrize("thr", [256, 8])
def test_spmm_vs_oracle_ragged(ops, dev, thr):
    rng = np.random.default_rng(0)
    n_rows, n_cols = 777, 500
    degs = rng.integers(0, 40, n_rows)
    degs[5] = 0
    degs[6] = 1
    degs[100] = 5000     # > 2 chunks of 2048
    degs[101] = 2048     # exactly one chunk
    degs[102] = 2049
    degs[776] = 300
    idx, val = _random_csr(rng, n_rows, n_cols, degs)
    g = ops.CsrGraph.from_coo_host(idx, val, n_rows, n_cols, dev, long_row_threshold=thr)
    assert g.n_long > 0
    X = rng.standard_normal((n_cols, 64)).astype(np.float32)
    ref = orc.spmm(orc.sparse_coo(idx, val, n_rows, n_cols), torch.from_numpy(X))
    Y = torch.empty(n_rows, 64, device=dev)
    ops.spmm_raw(g, D(X, dev), Y=Y)
    assert rel_fro(Y, ref) < 2e-6
    close(Y, ref, rtol=RTOL, atol=1e-4)   # atol: 5000-term rows of N(0,1) products
    assert torch.all(Y[5] == 0)
    # epilogue: Y = a*AX + b*Z ; acc = s*(acc_in + Y)
    Z = rng.standard_normal((n_rows, 64)).astype(np.float32)
    A0 = rng.standard_normal((n_rows, 64)).astype(np.float32)
    Y2, acc = torch.empty_like(Y), torch.empty_like(Y)
    ops.spmm_raw(g, D(X, dev), Y=Y2, Z=D(Z, dev), acc_in=D(A0, dev), acc_out=acc, alpha=0.5, beta=2.0,
                 acc_scale=0.25)
    ref2 = 0.5 * ref + 2.0 * torch.from_numpy(Z)
    close(Y2, ref2, atol=1e-4)
    close(acc, 0.25 * (torch.from_numpy(A0) + ref2), atol=1e-4)
    # run-to-run determinism (no float atomics)
    Y3 = torch.empty_like(Y)
    ops.spmm_raw(g, D(X, dev), Y=Y3)
    assert torch.equal(Y, Y3)


def test_spmm_row_shards_bitwise_equal(ops, dev):
    """multi-GPU invariant: computing a row block separately gives the same bits."""
    rng = np.random.default_rng(1)
    n = 1000
    degs = rng.integers(0, 30, n)
    degs[[10, 600]] = [3000, 700]
    idx, val = _random_csr(rng, n, n, degs)
    g = ops.CsrGraph.from_coo_host(idx, val, n, n, dev)
    X = D(rng.standard_normal((n, 64)).astype(np.float32), dev)
    Y = torch.empty(n, 64, device=dev)
    ops.spmm_raw(g, X, Y=Y)
    for r0, r1 in ((0, 333), (333, 700), (700, 1000)):
        blk = g.row_block(r0, r1)
        Yb = torch.empty(r1 - r0, 64, device=dev)
        ops.spmm_raw(blk, X, Y=Yb)
        assert torch.equal(Yb, Y[r0:r1])


@pytest.mark.parametrize("n_rows,thr,sorted_cols", [(777, 8, False), (777, 256, False), (300_000, None, False),
                                                    (300_000, None, True), (1_100_000, None, True)])
def test_spmm_feature_slices_equal_the_d64_launch_bitwise(ops, dev, n_rows, thr, sorted_cols):
    """The feature-sliced multi-GPU layout (csrc/spmm_narrow.hip): a rank owns 64 / P columns of every table
    and the whole graph, `Y[:, s] = A X[:, s]` needs no exchange.  World 1 here: the P = 2 / 4 / 8 slices ([n, 32 / 16 / 8]
    contiguous) through mmrec_spmm_csr_f32 one after the other == the columns of the d = 64 launch BIT FOR BIT -- short
    rows, empty rows, single-chunk and multi-chunk long rows (both sides of the chunk size), both row-finish forms of the
    d = 64 kernel (last-arriver up to 2^18 rows, two launches above), plain and full epilogue (alpha, beta Z, running sum);
    random and column-sorted rows, up to 1.1M rows."""
    rng = np.random.default_rng(n_rows)
    n_cols = n_rows if n_rows > 1000 else 500
    degs = rng.integers(0, 40, n_rows)
    degs[[5, 6, 100, 101, 102, 103, n_rows - 1]] = [0, 1, 5000, 512, 513, 20_000, 300]
    idx, val = _random_csr(rng, n_rows, n_cols, degs)
    if sorted_cols:      # column-sorted rows, as get_norm_adj_mat builds them (freedom.py:102-126)
        o = np.lexsort((idx[1], idx[0]))
        idx, val = idx[:, o], val[o]
    g = ops.CsrGraph.from_coo_host(idx, val, n_rows, n_cols, dev, long_row_threshold=thr)
    assert g.n_long > 0 and g.n_chunks > g.n_long
    X = D(rng.standard_normal((n_cols, 64)).astype(np.float32), dev)
    Z = D(rng.standard_normal((n_rows, 64)).astype(np.float32), dev)
    A0 = D(rng.standard_normal((n_rows, 64)).astype(np.float32), dev)
    Y, Y2, acc = (torch.empty(n_rows, 64, device=dev) for _ in range(3))
    ops.spmm_raw(g, X, Y=Y)
    ops.spmm_raw(g, X, Y=Y2, Z=Z, acc_in=A0, acc_out=acc, alpha=0.5, beta=2.0, acc_scale=0.25)
    for P in (2, 4, 8):
        w = 64 // P
        for s in range(P):
            cols = slice(s * w, (s + 1) * w)
            Xs, Zs, As = X[:, cols].contiguous(), Z[:, cols].contiguous(), A0[:, cols].contiguous()
            Ys, Y2s, accs = (torch.full((n_rows, w), float("nan"), device=dev) for _ in range(3))
            ops.spmm_raw(g, Xs, Y=Ys)
            assert torch.equal(Ys, Y[:, cols]), (P, s)
            ops.spmm_raw(g, Xs, Y=Y2s, Z=Zs, acc_in=As, acc_out=accs, alpha=0.5, beta=2.0, acc_scale=0.25)
            assert torch.equal(Y2s, Y2[:, cols]) and torch.equal(accs, acc[:, cols]), (P, s)
            ops.spmm_raw(g, Xs, acc_in=As, acc_out=accs)                  # Y = NULL: the running sum alone
            assert torch.equal(accs, As + Y[:, cols]), (P, s)
    assert torch.all(Y[5] == 0)
    with pytest.raises(Exception):
        ops.spmm_raw(g, X[:, :24].contiguous(), Y=torch.empty(n_rows, 24, device=dev))     # not a slice width


def test_lightgcn_mean_on_feature_slices_forward_and_backward_bitwise(ops, dev):
    """hip_ops.lightgcn_mean (freedom.py:169-176) per feature slice == its columns on the full tables, forward AND backward
    (the Horner recurrence on A^T is column-wise independent too), on the Amazon-Baby-shaped graph, P = 8."""
    from mmrec_amd import synth
    nu, ni, eu, ei = synth.shaped_edges("baby", seed=0)
    r, c, v = synth.sym_norm_coo(eu, ei, nu, ni)
    n = nu + ni
    g = ops.CsrGraph.from_coo_host(np.stack([r, c]), v, n, n, dev, symmetric=True)
    gen = torch.Generator().manual_seed(1)
    E0 = (torch.randn(n, 64, generator=gen) * 0.1).to(dev).requires_grad_()
    G = (torch.randn(n, 64, generator=gen)).to(dev)
    out = ops.lightgcn_mean(g, E0, 2)
    out.backward(G)
    for s in range(8):
        cols = slice(8 * s, 8 * s + 8)
        Es = E0.detach()[:, cols].contiguous().requires_grad_()
        o = ops.lightgcn_mean(g, Es, 2)
        o.backward(G[:, cols].contiguous())
        assert torch.equal(o.detach(), out.detach()[:, cols]) and torch.equal(Es.grad, E0.grad[:, cols]), s


@pytest.mark.parametrize("how", ["degree", "rcm", "community"])
def test_relabelled_graph_equals_the_plain_graph_bitwise(ops, dev, how):
    """hip_ops.PermutedGraph (round-3 review item 5): node ids relabelled at build time for gather locality ('degree': hot rows
    adjacent; 'rcm': reverse Cuthill-McKee, neighbours get nearby ids), the permutation kept on the graph, inputs / outputs
    permuted once per propagation.  A row keeps its nonzeros in their original order, so forward AND backward of
    lightgcn_mean and spmm equal the plain graph's BIT FOR BIT -- on the Baby-shaped graph (long rows, multi-chunk rows)."""
    from mmrec_amd import synth
    nu, ni, eu, ei = synth.shaped_edges("baby", seed=0)
    r, c, v = synth.sym_norm_coo(eu, ei, nu, ni)
    n = nu + ni
    g = ops.CsrGraph.from_coo_host(np.stack([r, c]), v, n, n, dev, symmetric=True)
    pg = ops.PermutedGraph(g, how, n_left=nu)
    assert sorted(pg.perm.cpu().tolist()) == list(range(n)) and torch.equal(pg.inv[pg.perm], torch.arange(n, device=dev))
    # structure: new row perm[r] holds old row r's entries, relabelled, in the same order
    idx_o, val_o = g.to_coo_host()
    idx_n, val_n = pg.graph.to_coo_host()
    perm = pg.perm.cpu().numpy()
    rp_o, rp_n = g.rowptr_host.astype(np.int64), pg.graph.rowptr_host.astype(np.int64)
    for row in (0, 17, nu + 3, int(np.argmax(np.diff(rp_o)))):
        a, b = slice(rp_o[row], rp_o[row + 1]), slice(rp_n[perm[row]], rp_n[perm[row] + 1])
        assert np.array_equal(perm[idx_o[1][a]], idx_n[1][b]) and np.array_equal(val_o[a], val_n[b])
    gen = torch.Generator().manual_seed(2)
    E0 = (torch.randn(n, 64, generator=gen) * 0.1).to(dev)
    G = torch.randn(n, 64, generator=gen).to(dev)
    outs = []
    for graph in (g, pg):
        E = E0.clone().requires_grad_()
        o = ops.lightgcn_mean(graph, E, 3)
        o.backward(G)
        Z = E0.clone().requires_grad_()
        o2 = ops.spmm(graph, Z, Z=Z)
        o2.backward(G)
        outs.append((o.detach(), E.grad, o2.detach(), Z.grad))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize("d", [64, 256])
def test_spmm_last_arriver_row_finish_equals_two_launches(ops, dev, d):
    """Graphs of <= 2^18 rows finish a multi-chunk row INSIDE the launch (mmrec_spmm_csr_f32 `long_tickets`, ABI 7: the chunk
    block that arrives last sums the row's partials) instead of in a second launch.  Same summation order => the same bits as
    the two-launch form (tickets withheld), launch after launch (the tickets are left at zero), in every epilogue -- plain,
    Z / layer sum, LayerGCN's cosine re-weighting -- and replayed inside a hipGraph.  Amazon-Baby-shaped graph (15 rows
    span several 512-nonzero chunks) plus rows of 5000 / 1025 / 513 nonzeros."""
    from mmrec_amd import synth
    rng = np.random.default_rng(3)
    nu, ni, eu, ei = synth.shaped_edges("baby", seed=0)
    r, c, v = synth.sym_norm_coo(eu, ei, nu, ni)
    n = nu + ni
    extra_rows = np.concatenate([np.full(5000, 7), np.full(1025, 9), np.full(513, n - 1)])
    idx = np.stack([np.concatenate([r, extra_rows]), np.concatenate([c, rng.integers(0, n, extra_rows.shape[0])])])
    val = np.concatenate([v, rng.standard_normal(extra_rows.shape[0]).astype(np.float32) * 0.01])
    g = ops.CsrGraph.from_coo_host(idx, val, n, n, dev)
    assert g.n_chunks > g.n_long > 0 and g.long_tickets is not None
    X = D((rng.random((n, d), dtype=np.float32) - 0.5) * 0.2, dev)
    Z = D(rng.standard_normal((n, d)).astype(np.float32), dev)
    A0 = D(rng.standard_normal((n, d)).astype(np.float32), dev)

    def run(fused):
        tickets, g.long_tickets = g.long_tickets, (g.long_tickets if fused else None)
        try:
            Y, acc = torch.empty(n, d, device=dev), torch.empty(n, d, device=dev)
            ops.spmm_raw(g, X, Y=Y, Z=Z, acc_in=A0, acc_out=acc, alpha=0.5, beta=2.0, acc_scale=0.25)
            return Y, acc
        finally:
            g.long_tickets = tickets
    ref = run(False)
    for _ in range(3):
        got = run(True)
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])
        assert int(g.long_tickets.abs().sum()) == 0
    if d == 64:
        ego = D(rng.standard_normal((n, 64)).astype(np.float32), dev)

        gs = ops.CsrGraph.from_coo_host(idx, val, n, n, dev, symmetric=True)
        a = ops.layergcn_sum(gs, ego, 2)
        tickets, gs.long_tickets = gs.long_tickets, None
        b = ops.layergcn_sum(gs, ego, 2)
        gs.long_tickets = tickets
        assert torch.equal(a, b)
        # replayed as a hipGraph: the tickets reset themselves, nothing comes from the host
        Y = torch.empty(n, d, device=dev)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            ops.spmm_raw(g, X, Y=Y)
            cg = torch.cuda.CUDAGraph()
            with torch.cuda.graph(cg, stream=side):
                ops.spmm_raw(g, X, Y=Y)
        torch.cuda.current_stream().wait_stream(side)
        Yref = torch.empty_like(Y)
        tickets, g.long_tickets = g.long_tickets, None
        ops.spmm_raw(g, X, Y=Yref)
        g.long_tickets = tickets
        for _ in range(3):
            Y.zero_()
            cg.replay()
            torch.cuda.synchronize()
            assert torch.equal(Y, Yref)


def test_spmm_empty_and_errors(ops, dev):
    from mmrec_amd._lib import MMRecHipError
    g = ops.CsrGraph.from_coo_host(np.zeros((2, 0), np.int64), np.zeros(0, np.float32), 10, 10, dev)
    Y = torch.full((10, 64), 7.0, device=dev)
    ops.spmm_raw(g, torch.ones(10, 64, device=dev), Y=Y)
    assert torch.all(Y == 0)
    with pytest.raises(MMRecHipError):
        ops.spmm_raw(g, torch.ones(10, 32, device=dev), Y=Y)      # d != 64
    X = torch.ones(10, 64, device=dev)
    with pytest.raises(MMRecHipError):
        ops.spmm_raw(g, X, Y=X)                                    # aliasing


def test_lightgcn_mean_golden(ops, dev, golden):
    g = golden
    graph = golden_graph(ops, g, dev)
    nu = int(g["n_users"])
    ue, ie = D(g["lgn_user_emb"], dev, True), D(g["lgn_item_emb"], dev, True)
    out = ops.lightgcn_mean(graph, torch.cat([ue, ie], 0), 3)
    close(out[:nu], g["lgn_user_out"])
    close(out[nu:], g["lgn_item_out"])
    # loss + grads of models/lightgcn.py:130-153 composed from the HIP ops
    b = D(g["batch"], dev)
    mf = ops.bpr_loss(out[:nu].contiguous(), out[nu:].contiguous(), b[0], b[1], b[2], ops.BPR_GAMMA)
    reg = (torch.sqrt(ops.gather_sqnorm(ue, b[0])) + torch.sqrt(ops.gather_sqnorm(ie, b[1])) +
           torch.sqrt(ops.gather_sqnorm(ie, b[2]))) / b.shape[1]
    loss = mf + 1e-4 * reg
    loss.backward()
    close(loss, g["lgn_loss"], rtol=1e-5)
    close(ue.grad, g["lgn_grad_user"], atol=1e-7)
    close(ie.grad, g["lgn_grad_item"], atol=1e-7)


def test_layergcn_golden(ops, dev, golden):
    g = golden
    nu = int(g["n_users"])
    ue, ie = D(g["lay_user_emb"], dev, True), D(g["lay_item_emb"], dev, True)
    out = ops.layergcn_sum(golden_graph(ops, g, dev), torch.cat([ue, ie], 0), 4)
    close(out[:nu], g["lay_user_out"])
    close(out[nu:], g["lay_item_out"])
    masked = golden_graph(ops, g, dev, key="lay_masked")
    out = ops.layergcn_sum(masked, torch.cat([ue, ie], 0), 4)
    b = D(g["batch"], dev)
    mf = ops.bpr_loss(out[:nu].contiguous(), out[nu:].contiguous(), b[0], b[1], b[2], ops.BPR_LOGSIG, "sum")
    reg = 0.5 * (ops.gather_sqnorm(ue, b[0]) + ops.gather_sqnorm(ie, b[1]) + ops.gather_sqnorm(ie, b[2]))
    loss = mf + 1e-3 * reg
    loss.backward()
    close(loss, g["lay_loss"], rtol=1e-5)
    close(ue.grad, g["lay_grad_user"], atol=2e-6)
    close(ie.grad, g["lay_grad_item"], atol=2e-6)


def test_freedom_golden(ops, dev, golden):
    """FREEDOM.calculate_loss (freedom.py:189-210) composed from the HIP ops: fused layer mean,
    item-item SpMM with residual, MFMA projections, three fused BPR terms -- loss and all grads."""
    g = golden
    nu, ni = int(g["n_users"]), int(g["n_items"])
    mm = ops.CsrGraph.from_coo_host(g["fr_mm_adj_idx"], g["fr_mm_adj_val"], ni, ni, dev)
    ue, ie = D(g["fr_user_emb"], dev, True), D(g["fr_item_emb"], dev, True)
    vf, tf = D(g["image_feat"], dev, True), D(g["text_feat"], dev, True)
    vw, vb = D(g["fr_image_W"], dev, True), D(g["fr_image_b"], dev, True)
    tw, tb = D(g["fr_text_W"], dev, True), D(g["fr_text_b"], dev, True)

    def forward(graph):
        mean = ops.lightgcn_mean(graph, torch.cat([ue, ie], 0), 2)
        return mean[:nu].contiguous(), ops.spmm(mm, ie, Z=mean[nu:].contiguous())

    ua, ia = forward(golden_graph(ops, g, dev))
    close(ua, g["fr_user_out"])
    close(ia, g["fr_item_out"])
    close(ops.linear(vf, vw, vb), g["fr_image_proj"], atol=1e-5)
    close(ops.linear(tf, tw, tb), g["fr_text_proj"], atol=1e-5)
    ua, ia = forward(golden_graph(ops, g, dev, key="fr_masked"))
    b = D(g["batch"], dev)
    loss = ops.bpr_loss(ua, ia, b[0], b[1], b[2]) + 1e-3 * (
        ops.bpr_loss(ua, ops.linear(tf, tw, tb), b[0], b[1], b[2]) +
        ops.bpr_loss(ua, ops.linear(vf, vw, vb), b[0], b[1], b[2]))
    loss.backward()
    close(loss, g["fr_loss"], rtol=1e-5)
    close(ue.grad, g["fr_grad_user"], atol=1e-8)
    close(ie.grad, g["fr_grad_item"], atol=1e-8)
    close(vw.grad, g["fr_grad_image_W"], atol=1e-9)
    close(vb.grad, g["fr_grad_image_b"], atol=1e-9)
    close(vf.grad, g["fr_grad_image_emb"], atol=1e-10)
    close(tw.grad, g["fr_grad_text_W"], atol=1e-9)


# ---------------------------------------------------------------------------------------- BPR
@pytest.mark.parametrize("variant,reduction", [(0, "mean"), (0, "sum"), (1, "mean")])
def test_bpr_variants_with_duplicates(ops, dev, variant, reduction):
    g = torch.Generator().manual_seed(3)
    U = (torch.randn(50, 64, generator=g) * 0.7).requires_grad_()
    I = (torch.randn(30, 64, generator=g) * 0.7).requires_grad_()
    B = 333  # not a multiple of 16; many duplicate ids
    us, ps, ns = (torch.randint(0, 50, (B,), generator=g), torch.randint(0, 30, (B,), generator=g),
                  torch.randint(0, 30, (B,), generator=g))
    if variant == 0:
        ref = orc.bpr_logsigmoid(U[us], I[ps], I[ns], reduction)
    else:
        ref = orc.bpr_gamma(U[us], I[ps], I[ns])
    (ref * 1.7).backward()
    Ud, Id = U.detach().to(dev).requires_grad_(), I.detach().to(dev).requires_grad_()
    loss = ops.bpr_loss(Ud, Id, us.to(dev), ps.to(dev), ns.to(dev), variant, reduction)
    (loss * 1.7).backward()
    close(loss, ref, rtol=1e-5)
    close(Ud.grad, U.grad, atol=1e-6)
    close(Id.grad, I.grad, atol=1e-6)


def _rows_test_graph(ops, dev, kind, rng):
    if kind == "knn":                     # 20 nonzeros in every row (two 10-NN graphs summed, uncoalesced): above the 16-nonzero
        n = 5000                          # threshold of a cache-resident graph -> every row is a single-chunk "long" row
        rows = np.repeat(np.arange(n), 20)
        cols = rng.integers(0, n, rows.shape[0])
    else:                                 # empty rows, short rows, single-chunk long rows (<= 512 nonzeros)
        n = 6000
        degs = np.minimum(rng.zipf(1.6, n), 500)
        degs[::7] = 0
        rows = np.repeat(np.arange(n), degs)
        cols = rng.integers(0, n, rows.shape[0])
    vals = rng.standard_normal(rows.shape[0]).astype(np.float32)
    return ops.CsrGraph.from_coo_host(np.stack([rows, cols]), vals, n, n, dev, long_row_threshold=None), n


@pytest.mark.parametrize("kind", ["knn", "mixed"])
@pytest.mark.parametrize("d", [8, 16, 32, 64])
def test_spmm_listed_rows_pull_is_bitwise_the_full_launch_and_push_is_its_transpose(ops, dev, d, kind):
    """mmrec_spmm_rows_f32 / mmrec_spmm_push_rows_f32 (ABI 10; FREEDOM's item-item layer read at the batch rows only,
    freedom.py:173-177, 197-199): the pulled rows carry the bits of the full launch -- short rows (one chain) and single-chunk
    long rows (the chunk block's 16-group order), with and without the residual Z, duplicated and unordered row lists -- and the
    push is the transposed product of the listed rows (fp32 atomics: compared with float64 to rounding)."""
    import scipy.sparse as sp
    rng = np.random.default_rng(d + len(kind))
    g, n = _rows_test_graph(ops, dev, kind, rng)
    assert ops.rows_servable(g, d) and (kind != "knn" or g.n_long == n)
    X = D(rng.standard_normal((n, d)).astype(np.float32), dev)
    Z = D(rng.standard_normal((n, d)).astype(np.float32), dev)
    rows = torch.from_numpy(np.concatenate([rng.integers(0, n, 3000), [0, n - 1, 5, 5, 5]])).to(dev)
    for z in (None, Z):
        full = torch.empty(n, d, device=dev)
        ops.spmm_raw(g, X, Y=full, Z=z)
        got = ops.spmm_rows_raw(g, X, rows, Z=z)
        assert torch.equal(got, full[rows])
    G = D(rng.standard_normal((rows.numel(), d)).astype(np.float32), dev)
    dX, dZ = torch.zeros(n, d, device=dev), torch.zeros(n, d, device=dev)
    ops.spmm_push_rows_raw(g, G, rows, dX, dZ, scale=0.25)
    idx, val = g.to_coo_host()
    A = sp.csr_matrix((val.astype(np.float64), (idx[0], idx[1])), shape=(n, n))
    r = rows.cpu().numpy()
    ref_x = A[r].T @ (0.25 * G.cpu().numpy().astype(np.float64))
    ref_z = np.zeros((n, d))
    np.add.at(ref_z, r, 0.25 * G.cpu().numpy().astype(np.float64))
    close(dX, ref_x, rtol=1e-5, atol=1e-5)
    close(dZ, ref_z, rtol=1e-6, atol=1e-6)
    both = torch.zeros(n, d, device=dev)                      # dZ may be dX: A^T G + G in one buffer
    ops.spmm_push_rows_raw(g, G, rows, both, both, scale=0.25)
    close(both, ref_x + ref_z, rtol=1e-5, atol=1e-5)
    zc = ops.spmm_rows_raw(g, X, rows, Z=G, z_compact=True)   # a compact residual: + Z[i]
    plain = ops.spmm_rows_raw(g, X, rows)
    assert torch.equal(zc, plain + G)
    # a graph with a row spanning several chunks: served at d = 64 since ABI 12 (mmrec_spmm_rows_any_f32, next test); the
    # narrower widths keep the full launch
    big = ops.CsrGraph.from_coo_host(np.stack([np.zeros(2000, np.int64), rng.integers(0, 3000, 2000)]),
                                     np.ones(2000, np.float32), 3000, 3000, dev, long_row_threshold=None)
    assert ops.rows_servable(big, d) == (d == 64)
    if d != 64:
        with pytest.raises(Exception):
            ops.spmm_rows_raw(big, X[:3000].contiguous(), rows[:10] % 3000)


def test_spmm_listed_rows_spanning_several_chunks_are_bitwise_the_full_launch(ops, dev):
    """mmrec_spmm_rows_any_f32 (ABI 12; the LAST user-item layer of a training step read at its batch rows, freedom.py:169-177 +
    197-199: popular items are in every batch and their rows span dozens of 512-nonzero chunks): rows of 2 ... 254 chunks,
    single-chunk long rows and short rows, listed several times and out of order, with the full and the compact residual --
    the full launch's bits for every listed row; the layer mean read at such rows (hip_ops.lightgcn_mean_parts_rows, last
    layer at the rows only) equals the full propagation read there, bit for bit, and its A/B switch reproduces the old path."""
    rng = np.random.default_rng(12)
    n = 40_000
    degs = rng.integers(0, 40, n)
    hubs = {7: 130_000, 8: 513, 9: 1024, 100: 1025, 101: 30_000, 39_999: 5_000, 20_000: 512}
    for r, k in hubs.items():
        degs[r] = k
    rows_coo = np.repeat(np.arange(n), degs)
    cols = rng.integers(0, n, rows_coo.shape[0])
    vals = rng.standard_normal(rows_coo.shape[0]).astype(np.float32) * 0.1
    g = ops.CsrGraph.from_coo_host(np.stack([rows_coo, cols]), vals, n, n, dev)
    assert g.n_chunks > g.n_long and g.max_row_chunks == -(-130_000 // 512) and ops.rows_servable(g, 64)
    X = D(rng.standard_normal((n, 64)).astype(np.float32), dev)
    Z = D(rng.standard_normal((n, 64)).astype(np.float32), dev)
    listed = torch.from_numpy(np.concatenate([list(hubs), rng.integers(0, n, 2000), [7, 101, 7, 0, n - 1]])).to(dev)
    for z in (None, Z):
        full = torch.empty(n, 64, device=dev)
        ops.spmm_raw(g, X, Y=full, Z=z)
        assert torch.equal(ops.spmm_rows_raw(g, X, listed, Z=z), full[listed])
    Gc = D(rng.standard_normal((listed.numel(), 64)).astype(np.float32), dev)
    assert torch.equal(ops.spmm_rows_raw(g, X, listed, Z=Gc, z_compact=True), ops.spmm_rows_raw(g, X, listed) + Gc)
    # the layer mean at listed rows (a square graph used as its own "user-item" graph: two row blocks)
    a, b = X[:n // 2].contiguous(), X[n // 2:].contiguous()
    for L in (1, 2, 3):
        want = torch.cat(ops.lightgcn_mean_parts(g, (a, b), L), dim=0)[listed]
        got = ops.lightgcn_mean_parts_rows(g, (a, b), L, listed)
        assert torch.equal(got, want), L
        try:
            ops.ROWS_LAST_LAYER = False
            assert torch.equal(ops.lightgcn_mean_parts_rows(g, (a, b), L, listed), want)
        finally:
            ops.ROWS_LAST_LAYER = True


@pytest.mark.parametrize("d", [64, 16])
@pytest.mark.parametrize("L", [0, 1, 2, 3])
def test_propagation_read_at_batch_rows_equals_the_dense_autograd(ops, dev, d, L):
    """hip_ops.lightgcn_mean_parts_rows + hip_ops.spmm_rows (FREEDOM._loss_at_batch_rows): the layer mean gathered at listed
    rows and the item-item layer pulled at listed rows == the dense ops read at those rows BIT FOR BIT in the forward; the
    backward (compact gradient -> push through the listed rows, then L - 1 full launches) == the dense autograd to rounding.
    A user-item graph with hub rows of thousands of nonzeros (one workgroup per pushed row), duplicated batch rows, a directed
    20-nonzero item-item graph."""
    rng = np.random.default_rng(10 * d + L)
    nu, ni = 900, 400
    from mmrec_amd import synth
    eu, ei = synth.powerlaw_edges(nu, ni, 30_000, seed=L)
    r, c, v = synth.sym_norm_coo(eu, ei, nu, ni)
    n = nu + ni
    g = ops.CsrGraph.from_coo_host(np.stack([r, c]), v, n, n, dev, symmetric=True)
    assert g.n_chunks > g.n_long > 0                           # rows spanning several chunks: the forward cannot be pulled here
    mm = ops.CsrGraph.from_coo_host(np.stack([np.repeat(np.arange(ni), 20), rng.integers(0, ni, 20 * ni)]),
                                    rng.random(20 * ni).astype(np.float32) * 0.1, ni, ni, dev, long_row_threshold=None)
    mm.transpose()
    U0, I0 = (rng.standard_normal((m, d)).astype(np.float32) * 0.3 for m in (nu, ni))
    B = 257
    users = torch.from_numpy(rng.integers(0, nu // 3, B)).to(dev)
    items = torch.from_numpy(np.concatenate([rng.integers(0, ni // 3, 2 * B - 3), [0, 0, ni - 1]])).to(dev)
    Gu, Gi = (D(rng.standard_normal((m, d)).astype(np.float32), dev) for m in (B, 2 * B))

    def run(rows_path):
        u, i = D(U0, dev, True), D(I0, dev, True)
        if rows_path:
            at = ops.lightgcn_mean_parts_rows(g, (u, i), L, torch.cat((users, items + nu)))
            ua_r, ia_r = at[:B], ops.spmm_rows(mm, i, items, Z_rows=at[B:])
        else:
            ua, ig = ops.lightgcn_mean_parts(g, (u, i), L)
            ua_r, ia_r = ua[users], ops.spmm(mm, i, Z=ig)[items]
        ((ua_r * Gu).sum() + (ia_r * Gi).sum()).backward()
        return [x.detach().cpu() for x in (ua_r, ia_r)], [x.grad.cpu() for x in (u, i)]

    (fa, ga), (fb, gb) = run(True), run(False)
    assert all(torch.equal(a, b) for a, b in zip(fa, fb))
    for a, b in zip(ga, gb):
        close(a, b, rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_sliced_shared_user_bpr_equals_the_full_width_op(ops, dev, world, variant):
    """the sampled-scoring kernels on COLUMN SLICES (feature-sliced layout): `world` ranks simulated in one
    process -- each slice's partial <u, p>, <u, n> (mmrec_bpr_dots_f32) against float64, their sum through
    mmrec_bpr_loss_from_dots_f32 == the full-width fused op's losses, and the slices' backward (mmrec_bpr_bwd_f32 at
    d = 64 / world) == the matching columns of its gradients.  Duplicate users / items in the batch on purpose."""
    gen = torch.Generator().manual_seed(7 + world)
    n_u, n_i, B, w = 500, 700, 333, 64 // world
    U, I, T = (torch.randn(n, 64, generator=gen) * 0.4 for n in (n_u, n_i, 2 * B))
    users, pos, neg = (torch.randint(0, n, (B,), generator=gen).to(dev) for n in (n_u // 3, n_i // 3, n_i))
    lp = torch.arange(B, device=dev)
    ln = lp + B

    def run(cols, sum_fn):
        u, i, t = (x[:, cols].contiguous().to(dev).requires_grad_() for x in (U, I, T))
        ls = ops.bpr_losses_shared_users(u, users, [(i, pos, neg), (t, lp, ln)], variant, sum_over_ranks=sum_fn)
        (ls[0] + 0.37 * ls[1]).backward()
        return [x.detach().cpu() for x in ls], [x.grad.cpu() for x in (u, i, t)]

    ref_l, ref_g = run(slice(0, 64), None)
    slices = [slice(r * w, (r + 1) * w) for r in range(world)]
    partial = []
    for cols in slices:                                                  # pass 1: every rank's partial dot products
        run(cols, lambda t: partial.append(t.clone()))
        u64, i64, t64 = (x[:, cols].double() for x in (U, I, T))
        uu = u64[users.cpu()]
        want = torch.stack((torch.stack(((uu * i64[pos.cpu()]).sum(1), (uu * i64[neg.cpu()]).sum(1))),
                            torch.stack(((uu * t64[:B]).sum(1), (uu * t64[B:]).sum(1)))))
        close(partial[-1], want, rtol=1e-5, atol=1e-6)
    total = partial[0].clone()
    for x in partial[1:]:
        total += x                                                       # what the all-reduce hands every rank
    grads = []
    for cols in slices:                                                  # pass 2: the replicated losses, the local backward
        ls, gs = run(cols, lambda t: t.copy_(total))
        for a, b in zip(ls, ref_l):
            close(a, b, rtol=2e-6, atol=1e-7)
        grads.append(gs)
    for j in range(3):
        close(torch.cat([g[j] for g in grads], 1), ref_g[j], rtol=1e-5, atol=1e-7)
    with pytest.raises(Exception):                                       # a width no kernel has
        run(slice(0, 24), lambda t: None)


def test_shared_user_bpr_and_split_mean_match_the_separate_ops(ops, dev, golden):
    """FREEDOM's fused autograd nodes (one [n_users, d] gradient buffer for the three BPR terms; cat -> propagate -> split
    as one node) against the per-op composition with torch's cat / slices / adds around it and against the CPU oracle:
    identical forward values, gradients within rounding of the changed summation order, an unused output handled."""
    g = golden
    graph = golden_graph(ops, g, dev)
    nu = int(g["n_users"])
    gen = torch.Generator().manual_seed(11)
    B = 233
    b = D(g["batch"], dev)[:, :B].contiguous()
    lp = (torch.arange(B, device=dev) % 77).contiguous()                 # duplicate slots on purpose
    ln = (lp + 17).contiguous()
    T = torch.randn(400, 64, generator=gen) * 0.3
    V = torch.randn(400, 64, generator=gen) * 0.3

    def run(fused, joint=False):
        ue, ie = D(g["lgn_user_emb"], dev, True), D(g["lgn_item_emb"], dev, True)
        t, v = T.to(dev).requires_grad_(), V.to(dev).requires_grad_()
        if fused:
            u, i = ops.lightgcn_mean_parts(graph, (ue, ie), 2)
            l0, lt, lv = ops.bpr_losses_shared_users(u, b[0], [(i, b[1], b[2]), (t, lp, ln), (v, lp, ln)], joint_grad=joint)
        else:
            out = ops.lightgcn_mean(graph, torch.cat([ue, ie], 0), 2)
            u, i = out[:nu].contiguous(), out[nu:].contiguous()
            l0, lt, lv = (ops.bpr_loss(u, i, b[0], b[1], b[2]), ops.bpr_loss(u, t, b[0], lp, ln),
                          ops.bpr_loss(u, v, b[0], lp, ln))
        loss = l0 + 0.37 * (lt + lv)
        loss.backward()
        return [x.detach().cpu() for x in (loss, u, i, ue.grad, ie.grad, t.grad, v.grad)]
    a, r, j = run(True), run(False), run(True, True)
    for k, (x, y) in enumerate(zip(a, j)):                               # the copy-free gradient hand-over changes nothing
        assert torch.equal(x, y) if k < 3 else torch.allclose(x, y, rtol=1e-5, atol=1e-7)   # (atomics: last-ulp order)
    for x, y in zip(a[:3], r[:3]):
        assert torch.equal(x, y)
    for x, y in zip(a[3:], r[3:]):
        close(x, y, atol=1e-7, rtol=1e-5)
    # CPU oracle for the same composite (lightgcn.py:115-128 + freedom.py:180-187 restated)
    n = graph.n_rows
    adj = orc.sparse_coo(g["norm_adj_idx"], g["norm_adj_val"], n)
    ue, ie = torch.tensor(g["lgn_user_emb"], requires_grad=True), torch.tensor(g["lgn_item_emb"], requires_grad=True)
    t, v = T.clone().requires_grad_(), V.clone().requires_grad_()
    u, i = orc.lightgcn_forward(adj, ue, ie, 2)
    bc, lpc, lnc = b.cpu(), lp.cpu(), ln.cpu()
    ref = (orc.bpr_logsigmoid(u[bc[0]], i[bc[1]], i[bc[2]]) +
           0.37 * (orc.bpr_logsigmoid(u[bc[0]], t[lpc], t[lnc]) + orc.bpr_logsigmoid(u[bc[0]], v[lpc], v[lnc])))
    ref.backward()
    close(a[0], ref, rtol=1e-5)
    for x, y in zip(a[3:], (ue.grad, ie.grad, t.grad, v.grad)):
        close(x, y, atol=1e-7, rtol=1e-4)
    # the two tables as adjacent row blocks of one allocation (models/_base.py AdjacentTablesMixin): no cat, same numbers
    both = torch.cat([D(g["lgn_user_emb"], dev), D(g["lgn_item_emb"], dev)], 0)
    ue, ie = both[:nu].detach().requires_grad_(), both[nu:].detach().requires_grad_()
    assert ops.row_blocks_of_one_buffer((ue, ie)) and not ops.row_blocks_of_one_buffer((ie, ue))
    u, i = ops.lightgcn_mean_parts(graph, (ue, ie), 2)
    l0, = ops.bpr_losses_shared_users(u, b[0], [(i, b[1], b[2])], joint_grad=True)
    l0.backward()
    assert torch.equal(u.detach().cpu(), a[1]) and torch.equal(i.detach().cpu(), a[2])
    ue2, ie2 = D(g["lgn_user_emb"], dev, True), D(g["lgn_item_emb"], dev, True)
    out = ops.lightgcn_mean(graph, torch.cat([ue2, ie2], 0), 2)
    ops.bpr_loss(out[:nu].contiguous(), out[nu:].contiguous(), b[0], b[1], b[2]).backward()
    close(ue.grad, ue2.grad, atol=1e-7, rtol=1e-5)
    close(ie.grad, ie2.grad, atol=1e-7, rtol=1e-5)
    # LayerGCN's propagation through the same kind of node (layergcn.py:125-152, BPR summed)
    ue, ie = both[:nu].detach().clone(), both[nu:].detach().clone()
    both2 = torch.cat([ue, ie], 0)
    ue, ie = both2[:nu].detach().requires_grad_(), both2[nu:].detach().requires_grad_()
    u, i = ops.layergcn_sum_parts(graph, (ue, ie), 3)
    l1, = ops.bpr_losses_shared_users(u, b[0], [(i, b[1], b[2])], ops.BPR_LOGSIG, "sum", joint_grad=True)
    l1.backward()
    ue2, ie2 = D(g["lgn_user_emb"], dev, True), D(g["lgn_item_emb"], dev, True)
    out = ops.layergcn_sum(graph, torch.cat([ue2, ie2], 0), 3)
    l2 = ops.bpr_loss(out[:nu].contiguous(), out[nu:].contiguous(), b[0], b[1], b[2], ops.BPR_LOGSIG, "sum")
    l2.backward()
    assert torch.equal(l1, l2) and torch.equal(u, out[:nu]) and torch.equal(i, out[nu:])
    close(ue.grad, ue2.grad, atol=1e-6, rtol=1e-5)
    close(ie.grad, ie2.grad, atol=1e-6, rtol=1e-5)
    # an output nobody uses: its gradient slot arrives as None
    ue, ie = D(g["lgn_user_emb"], dev, True), D(g["lgn_item_emb"], dev, True)
    u, i = ops.lightgcn_mean_parts(graph, (ue, ie), 2)
    u.square().sum().backward()
    ue2, ie2 = D(g["lgn_user_emb"], dev, True), D(g["lgn_item_emb"], dev, True)
    ops.lightgcn_mean(graph, torch.cat([ue2, ie2], 0), 2)[:nu].square().sum().backward()
    close(ue.grad, ue2.grad.cpu(), atol=1e-7, rtol=1e-5)
    close(ie.grad, ie2.grad.cpu(), atol=1e-7, rtol=1e-5)
    l = ops.bpr_losses_shared_users(u.detach().requires_grad_(), b[0], [(i.detach(), b[1], b[2]), (T.to(dev), lp, ln)])
    l[1].backward()                                                      # only the second term is used


@pytest.mark.parametrize("d", [64, 384])
def test_scatter_add_rows_sorted_is_deterministic_and_exact(ops, dev, d):
    """mmrec_scatter_add_rows_sorted_f32 (`hip_deterministic`): out[ids[b]] += rows[b] with duplicates summed in POSITION
    order by one owner -- equals the sequential fp32 loop of the reference's CPU scatter bit for bit, skips ids < 0, adds to
    what `out` holds; one id repeated 3000 times, ids that occur once, an empty batch."""
    rng = np.random.default_rng(d)
    n, n_rows = 5000, 400
    ids = rng.integers(0, n_rows, n)
    ids[:3000] = 7
    rng.shuffle(ids)
    ids[[5, 77]] = -1
    rows = rng.standard_normal((n, d)).astype(np.float32)
    base = rng.standard_normal((n_rows, d)).astype(np.float32)
    ref = np.zeros((n_rows, d), dtype=np.float32)
    for b in range(n):                                   # position order, fp32, like index_add on the CPU
        if ids[b] >= 0:
            ref[ids[b]] += rows[b]
    ref = base + ref
    out = D(base.copy(), dev)
    ops.scatter_add_rows(D(ids.astype(np.int64), dev), D(rows, dev), out)
    np.testing.assert_array_equal(out.cpu().numpy(), ref)
    out2 = D(base.copy(), dev)
    ops.scatter_add_rows(D(ids.astype(np.int64), dev), D(rows, dev), out2)
    assert torch.equal(out, out2)
    ops.scatter_add_rows(torch.zeros(0, dtype=torch.int64, device=dev), torch.zeros(0, d, device=dev), out2)
    assert torch.equal(out, out2)


def test_bpr_extreme_scores(ops, dev):
    """logsigmoid / log(1e-10+sigmoid) differ for x << 0: both exact."""
    U = torch.zeros(4, 64)
    I = torch.zeros(4, 64)
    U[:, 0] = torch.tensor([1.0, 1.0, 1.0, 1.0])
    I[:, 0] = torch.tensor([40.0, -40.0, 0.0, 100.0])
    us = torch.tensor([0, 1, 2, 3])
    ps = torch.tensor([0, 1, 2, 1])
    ns = torch.tensor([1, 0, 2, 3])
    for variant, fn in ((0, lambda: orc.bpr_logsigmoid(U[us], I[ps], I[ns])), (1, lambda: orc.bpr_gamma(U[us], I[ps], I[ns]))):
        got = ops.bpr_loss(U.to(dev), I.to(dev), us.to(dev), ps.to(dev), ns.to(dev), variant)
        close(got, fn(), rtol=1e-5)


def test_cosine_mean_fwd_bwd_vs_torch(ops, dev):
    """a9': fused mean cosine similarity of gathered rows == F.cosine_similarity (clamp 1e-8 per norm) incl. a zero
    row, duplicate ids, identity indexing, row width 128; gradient w.r.t. X only (targets are detached in BM3)."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(5)
    for d in (64, 128):
        X = torch.randn(50, d, generator=g)
        Y = torch.randn(40, d, generator=g)
        X[7] = 0.0
        ix = torch.randint(0, 50, (333,), generator=g)
        ix[:4] = 7
        iy = torch.randint(0, 40, (333,), generator=g)
        for use_ix, use_iy in ((True, True), (False, True), (False, False)):
            Xc = (X if use_ix else X[ix]).clone().requires_grad_()
            Yc = Y if use_iy else Y[iy].clone()
            ref = F.cosine_similarity(Xc[ix] if use_ix else Xc, Yc[iy] if use_iy else Yc, dim=-1).mean()
            ref.backward()
            Xd = Xc.detach().to(dev).requires_grad_()
            out = ops.cosine_mean(Xd, ix.to(dev) if use_ix else None, Yc.to(dev), iy.to(dev) if use_iy else None)
            (3.0 * out).backward()
            close(out, ref.item(), rtol=1e-5)
            close(Xd.grad, 3.0 * Xc.grad, rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("variant", [0, 1])
def test_bpr_weighted_total_fused_terms(ops, dev, variant):
    """ABI 14 mmrec_bpr_multi_fwd/bwd_f32 (hip_ops.bpr_weighted_total): FREEDOM's bpr(id) + reg_weight (bpr(text) + bpr(image))
    (freedom.py:197-211) as one launch pair against the per-term op: value to fp32 rounding, gradients of the user table and of
    every term's table (one of them named twice: one buffer), duplicate ids, a batch that is no multiple of 16, both loss
    variants, mean and sum reductions, an upstream gradient != 1, a table without gradient."""
    g = torch.Generator().manual_seed(31 + variant)
    nU, nI, B, d = 80, 55, 203, 64
    mk = lambda *s: torch.randn(*s, generator=g) * 0.3
    U0, I0, T0, V0 = mk(nU, d), mk(nI, d), mk(2 * B, d), mk(2 * B, d)
    users, pos, neg = (torch.randint(0, nU, (B,), generator=g), torch.randint(0, nI, (B,), generator=g),
                       torch.randint(0, nI, (B,), generator=g))
    users[:30] = users[30:60]
    lp, ln = torch.arange(B), torch.arange(B) + B
    w = [1.0, 0.13, 0.13, 0.5]
    for reduction in ("mean", "sum"):
        res = []
        for fused in (True, False):
            U, I, T = (x.clone().to(dev).requires_grad_() for x in (U0, I0, T0))
            V = V0.clone().to(dev)                                    # no gradient for this one
            terms = [(I, pos.to(dev), neg.to(dev)), (T, lp.to(dev), ln.to(dev)), (V, lp.to(dev), ln.to(dev)),
                     (I, neg.to(dev), pos.to(dev))]
            if fused:
                out = ops.bpr_weighted_total(U, users.to(dev), terms, w, variant, reduction)
            else:
                out = sum(wt * l for wt, l in zip(w, ops.bpr_losses_shared_users(U, users.to(dev), terms, variant, reduction)))
            (1.9 * out).backward()
            res.append((out.detach(), U.grad.clone(), I.grad.clone(), T.grad.clone()))
        close(res[0][0], res[1][0], rtol=2e-6)
        for a, b in zip(res[0][1:], res[1][1:]):
            close(a, b, rtol=1e-5, atol=1e-7 * (B if reduction == "sum" else 1))


def test_cosine_means_fused_terms(ops, dev):
    """ABI 14 mmrec_cosine_multi_fwd/bwd_f32 (hip_ops.cosine_means): BM3's six BYOL terms (bm3.py:129-144) as one launch pair --
    sum_t w_t mean_b cos(X_t[ix_t[b]], Y_t[iy_t[b]]) against torch: value and gradients; indexed and un-indexed operands, an X
    shared by two terms (one dense gradient), duplicate ids, a zero row (the 1e-8 clamp), batches that are no multiple of 16,
    128-wide rows, an upstream gradient != 1; rows_reg over WHOLE tables (ids None: bm3.py:146's EmbLoss) rides along."""
    g = torch.Generator().manual_seed(21)
    for d in (64, 128):
        nU, nI, B = 60, 45, 101
        Up, Ip, Tp = (torch.randn(nU, d, generator=g).requires_grad_(), torch.randn(nI, d, generator=g).requires_grad_(),
                      torch.randn(B, d, generator=g).requires_grad_())
        It, Ut, Tt = torch.randn(nI, d, generator=g), torch.randn(nU, d, generator=g), torch.randn(B, d, generator=g)
        with torch.no_grad():
            Ip[3].zero_()
        users, items = torch.randint(0, nU, (B,), generator=g), torch.randint(0, nI, (B,), generator=g)
        items[:10] = 3
        spec = [(0, users, It, items, -1.0), (1, items, Ut, users, -1.0), (2, None, It, items, -0.4), (2, None, Tt, None, -0.4)]
        cpu_x = (Up, Ip, Tp)
        cosf = torch.nn.functional.cosine_similarity
        ref = sum(w * cosf(cpu_x[k] if ix is None else cpu_x[k][ix], Y if iy is None else Y[iy], dim=-1).mean() for k, ix, Y, iy, w in spec)
        reg_ref = 0.02 * (torch.norm(Up) + torch.norm(Ip))
        (1.3 * (ref + reg_ref)).backward()
        dx = [t.detach().to(dev).requires_grad_() for t in cpu_x]
        mv = lambda t: None if t is None else t.to(dev)
        terms = [(dx[k], mv(ix), Y.to(dev), mv(iy), w) for k, ix, Y, iy, w in spec]
        out = ops.cosine_means(terms)
        reg = ops.rows_reg(((dx[0], None), (dx[1], None)), ops.ROWS_REG_NORM, 0.02)
        (1.3 * (out + reg)).backward()
        close(out, ref, rtol=1e-5, atol=1e-6)
        close(reg, reg_ref, rtol=2e-6)
        for a, b in zip(dx, cpu_x):
            close(a.grad, b.grad, rtol=1e-4, atol=2e-7)
        per = sum(w * ops.cosine_mean(x.detach(), ix, Y, iy) for x, ix, Y, iy, w in terms)
        close(out, per, rtol=1e-5, atol=1e-6)


def test_gather_sqnorm(ops, dev):
    g = torch.Generator().manual_seed(4)
    E = torch.randn(40, 64, generator=g).requires_grad_()
    ids = torch.randint(0, 40, (100,), generator=g)
    ref = torch.sum(E[ids] ** 2)
    (0.5 * ref).backward()
    Ed = E.detach().to(dev).requires_grad_()
    out = ops.gather_sqnorm(Ed, ids.to(dev))
    (0.5 * out).backward()
    close(out, ref, rtol=1e-5)
    close(Ed.grad, E.grad, atol=1e-6)


@pytest.mark.parametrize("mode", [0, 1])
def test_rows_reg_fused_regulariser(ops, dev, mode):
    """ABI 14 mmrec_rows_reg_fwd_f32 / _bwd_f32 (hip_ops.rows_reg): the three terms of a step's regulariser in one launch pair --
    scale * sum_t ||E_t[ids_t]||_F^2 (mode 0: layergcn.py:154-161) or scale * sum_t ||E_t[ids_t]||_F (mode 1: EmbLoss,
    common/loss.py:46-51) -- against torch: value and gradients, the item table named by two terms (one dense gradient), duplicate
    ids, batches of different lengths that are no multiple of 16, 128-wide rows, an upstream gradient != 1, and (mode 1) a term
    whose rows are all zero: gradient 0 there, as torch.norm's backward gives.  Equal to the per-term ops it replaces."""
    g = torch.Generator().manual_seed(6 + mode)
    for d in (64, 128):
        U, I, Z = (torch.randn(50, d, generator=g).requires_grad_(), torch.randn(70, d, generator=g).requires_grad_(),
                   torch.zeros(9, d, requires_grad=True))
        iu, ip, in_, iz = (torch.randint(0, 50, (101,), generator=g), torch.randint(0, 70, (101,), generator=g),
                           torch.randint(0, 70, (37,), generator=g), torch.randint(0, 9, (5,), generator=g))
        ip[:20] = ip[20:40]
        scale = 0.37
        terms = [(U, iu), (I, ip), (I, in_), (Z, iz)]
        f = (lambda x: x) if mode == 0 else torch.sqrt
        ref = scale * sum(f((E[i] ** 2).sum()) for E, i in terms[:3]) + scale * (Z[iz] ** 2).sum() * (1.0 if mode == 0 else 0.0)
        (1.7 * ref).backward()
        dev_t = [t.detach().to(dev).requires_grad_() for t in (U, I, Z)]
        d_terms = [(dev_t[0], iu.to(dev)), (dev_t[1], ip.to(dev)), (dev_t[1], in_.to(dev)), (dev_t[2], iz.to(dev))]
        out = ops.rows_reg(d_terms, mode, scale)
        (1.7 * out).backward()
        close(out, ref, rtol=2e-6)
        close(dev_t[0].grad, U.grad, rtol=1e-5, atol=1e-7)
        close(dev_t[1].grad, I.grad, rtol=1e-5, atol=1e-7)
        assert float(dev_t[2].grad.abs().max()) == 0.0 and torch.isfinite(dev_t[2].grad).all()
        # the per-term ops it replaces
        per = scale * sum((lambda x: x if mode == 0 else torch.sqrt(x))(ops.gather_sqnorm(E.detach(), i)) for E, i in d_terms[:3])
        close(out, per, rtol=2e-6)


@pytest.mark.parametrize("n,wa,wb,with_r", [(300, 64, 64, True), (1001, 384, 64, True), (77, 256, 64, False), (5, 4, 8, True)])
def test_cat_leaky_fused_layer_tail(ops, dev, n, wa, wb, with_r):
    """ABI 14 mmrec_cat_leaky_fwd/bwd_f32 (hip_ops.cat_leaky): cat((leaky_relu(A), leaky_relu(B) + R), dim=1) of an MMGCN layer
    (mmgcn.py:170-173) == the four torch ops, values and gradients BIT FOR BIT (elementwise: nothing to reorder), exact zeros and
    negative inputs included, R absent / without gradient."""
    g = torch.Generator().manual_seed(n + wa)
    A, B = torch.randn(n, wa, generator=g), torch.randn(n, wb, generator=g)
    A[0, :4] = 0.0
    B[n - 1] = 0.0
    R = torch.randn(n, wb, generator=g) if with_r else None
    G = torch.randn(n, wa + wb, generator=g)
    a, b = A.clone().requires_grad_(), B.clone().requires_grad_()
    r = R.clone().requires_grad_() if with_r else None
    x_hat = torch.nn.functional.leaky_relu(b)
    ref = torch.cat((torch.nn.functional.leaky_relu(a), x_hat + r if with_r else x_hat), dim=1)
    ref.backward(G)
    ad, bd = A.to(dev).requires_grad_(), B.to(dev).requires_grad_()
    rd = R.to(dev).requires_grad_() if with_r else None
    out = ops.cat_leaky(ad, bd, rd)
    out.backward(G.to(dev))
    assert torch.equal(out.cpu(), ref.detach())
    assert torch.equal(ad.grad.cpu(), a.grad) and torch.equal(bd.grad.cpu(), b.grad)
    if with_r:
        assert torch.equal(rd.grad.cpu(), r.grad)
        out2 = ops.cat_leaky(A.to(dev).requires_grad_(), B.to(dev), R.to(dev))       # R, B without gradient
        out2.sum().backward()


@pytest.mark.parametrize("n,d", [(333, 64), (50, 256), (1000, 384), (7, 4)])
def test_row_normalize_fused(ops, dev, n, d):
    """ABI 14 mmrec_row_normalize_fwd/bwd_f32 (hip_ops.row_normalize) == F.normalize(x, p=2, dim=1) (lattice.py:165, mmgcn.py:167):
    values and gradients to fp32 rounding, an all-zero row and a row below the 1e-12 clamp (norm treated as a constant) included."""
    g = torch.Generator().manual_seed(n + d)
    X = torch.randn(n, d, generator=g)
    X[1] = 0.0
    X[2] *= 1e-20
    X[3] *= 1e4
    G = torch.randn(n, d, generator=g)
    x = X.clone().requires_grad_()
    ref = torch.nn.functional.normalize(x, p=2, dim=1)
    ref.backward(G)
    xd = X.to(dev).requires_grad_()
    out = ops.row_normalize(xd)
    out.backward(G.to(dev))
    close(out, ref.detach(), rtol=2e-6, atol=1e-7)
    row_scale = x.grad.abs().max(dim=1, keepdim=True)[0]          # (the clamped rows' gradients are g / 1e-12: per-row scales)
    assert bool(((xd.grad.cpu() - x.grad).abs() <= 4e-6 * row_scale + 1e-12).all())
    assert torch.isfinite(xd.grad).all() and float(out[1].abs().max()) == 0.0


def test_infonce_fwd_bwd_vs_oracle(ops, dev):
    """In-batch InfoNCE incl. duplicate ids (scatter-add), batch not a multiple of the 64-row tile,
    a zero row (normalisation eps) and asymmetric views."""
    g = torch.Generator().manual_seed(3)
    n, B = 500, 333
    E1 = torch.randn(n, 64, generator=g).requires_grad_()
    E2 = (torch.randn(n, 64, generator=g) * 0.5 + 0.3 * E1.detach()).requires_grad_()
    with torch.no_grad():
        E1[7].zero_()
    ids = torch.randint(0, n, (B,), generator=g)
    ids[:40] = ids[40:80]          # duplicates
    ids[5] = 7
    for tau in (0.2, 0.5):
        E1.grad = E2.grad = None
        ref = orc.infonce(E1[ids], E2[ids], tau)
        (3.0 * ref).backward()
        A, C = E1.detach().to(dev).requires_grad_(), E2.detach().to(dev).requires_grad_()
        out = ops.infonce(A, C, ids.to(dev), tau)
        (3.0 * out).backward()
        close(out, ref, rtol=1e-5)
        assert rel_fro(A.grad, E1.grad) < 2e-6 and rel_fro(C.grad, E2.grad) < 2e-6
        close(A.grad, E1.grad, atol=2e-6)
        close(C.grad, E2.grad, atol=2e-6)


# ---------------------------------------------------------------------------------------- linear
@pytest.mark.parametrize("n,F,bias", [(90, 96, True), (90, 40, True), (300, 384, False), (1000, 4096, True),
                                      (129, 4480, True), (1, 8, True)])
def test_linear_fwd_bwd(ops, dev, n, F, bias):
    g = torch.Generator().manual_seed(n + F)
    X = torch.randn(n, F, generator=g).requires_grad_()
    W = (torch.randn(64, F, generator=g) / F ** 0.5).requires_grad_()
    b = torch.randn(64, generator=g).requires_grad_() if bias else None
    G = torch.randn(n, 64, generator=g)   # asymmetric upstream gradient (transpose-detecting)
    ref = orc.linear(X, W, b)
    ref.backward(G)
    Xd, Wd = X.detach().to(dev).requires_grad_(), W.detach().to(dev).requires_grad_()
    bd = b.detach().to(dev).requires_grad_() if bias else None
    Y = ops.linear(Xd, Wd, bd)
    Y.backward(G.to(dev))
    assert rel_fro(Y, ref) < 1e-6
    close(Y, ref, atol=1e-5)
    assert rel_fro(Wd.grad, W.grad) < 1e-6 and rel_fro(Xd.grad, X.grad) < 1e-6
    close(Wd.grad, W.grad, atol=1e-4)
    close(Xd.grad, X.grad, atol=1e-5)
    if bias:
        close(bd.grad, b.grad, atol=1e-4)


@pytest.mark.parametrize("n,F", [(3000, 4096), (7050, 384), (513, 4480), (40_000, 4096), (18_357, 4096), (16_600, 1024)])
def test_linear_split_forward_is_as_accurate_as_the_fp32_kernel(ops, dev, n, F):
    """mmrec_linear_fwd_split_f32 (ABI 8; hip_ops.LINEAR_F16X3, the default): the projection's forward as three fp16 MFMA products
    of split operands x = hi + 2^-11 lo'.  Against float64 its error is within 2 x the fp32-MFMA kernel's (both ~1e-7 of the
    result's scale), on relu-like features, rows scaled by 1e4 and 1e-6, a zero row, weights of mixed magnitude -- and the
    two kernels agree to 1e-6 of the row scale."""
    g = torch.Generator().manual_seed(n + F)
    X = torch.relu(torch.randn(n, F, generator=g))
    X[1] *= 1e4
    X[2] *= 1e-6
    X[3] = 0.0
    W = torch.randn(64, F, generator=g) / F ** 0.5
    W[:8] *= 50.0
    W[8:16] *= 1e-3
    b = torch.randn(64, generator=g)
    ref = X.double() @ W.double().t() + b.double()
    Xd, Wd, bd = X.to(dev), W.to(dev), b.to(dev)
    out = {}
    try:
        for split in (True, False):
            ops.LINEAR_F16X3 = split
            out[split] = ops.linear(Xd, Wd, bd).cpu().double()
    finally:
        ops.LINEAR_F16X3 = True
    scale = X.double().abs() @ W.double().abs().t() + b.double().abs()      # sum |x_k w_k| + |b|: what rounding errors are relative to
    err = {k: float(((v - ref).abs() / (scale + 1e-30)).max()) for k, v in out.items()}
    assert err[True] <= 2.0 * err[False] + 2e-7 and err[True] < 1e-6, err
    assert float(((out[True] - out[False]).abs() / (scale + 1e-30)).max()) < 1e-6
    assert torch.equal(out[True][3], out[False][3])              # the zero row: bias only, exactly


@pytest.mark.parametrize("n,F", [(1000, 4096), (7050, 384), (40_000, 4096), (300_000, 128), (18_357, 4096), (16_600, 1024)])
def test_linear_split_domain_guard(ops, dev, n, F):
    """The split-operand forward is the DEFAULT projection, so it has to be a drop-in for fp32 `nn.Linear` (freedom.py:205,208;
    bm3.py:102-104) over ALL of fp32's range, not only where fp16 holds the two halves: rows of tiny magnitude (1e-7, 1e-8: fp16
    subnormals), of huge magnitude (5e4 < 65504 still inside; 1e5, 3e38 outside), inf and NaN.  No bias, error measured against
    sum |x_k w_k| ALONE (a bias of size 1 would hide a 1e-6-scaled row), <= 1e-6; non-finite rows propagate exactly as the
    fp32 kernel's (= F.linear's) do; rows in the same 128-row block as a flagged row and all other rows are unharmed.  Shapes:
    one slab per row block (300,000 x 128), split-K (1000 x 4096, 7050 x 384: maxima combined over slabs) and the stream-K grid
    from 129 row blocks (18,357 and 16,600 rows: a block's tiles in two or three segments; 40,000: runs that span two blocks)."""
    g = torch.Generator().manual_seed(n + F)
    X = torch.relu(torch.randn(n, F, generator=g))
    W = torch.randn(64, F, generator=g) / F ** 0.5
    scales = {1: 1e-7, 2: 1e-8, 5: 5e4, 130: 1e5, 260: 3e-5, 261: 1e-3, 400: 1e-12, 520: 1e-30}
    for r, sc in scales.items():
        X[r] *= sc
    X[390, 7] = 3e38                      # finite in fp32 and so is 3e38 * w: fp16's inf must not reach the result
    X[650] = 0.0                          # an all-zero row is exact in both kernels and must not be flagged
    X[651, :] = 0.0
    X[651, F - 1] = 1e-9                  # a single tiny element
    X[700, 3] = float("inf")
    X[701, 5] = float("nan")
    X[702, 0], X[702, 1] = float("inf"), -float("inf")
    Xd, Wd = X.to(dev), W.to(dev)
    out = {}
    try:
        for split in (True, False):
            ops.LINEAR_F16X3 = split
            out[split] = ops.linear(Xd, Wd, None).cpu()
    finally:
        ops.LINEAR_F16X3 = True
    finite = torch.isfinite(X).all(1)
    ref = X[finite].double() @ W.double().t()
    scale = X[finite].double().abs() @ W.double().abs().t()
    for k in (True, False):
        err = ((out[k][finite].double() - ref).abs() / (scale + 1e-300))
        err[scale == 0] = (out[k][finite].double() - ref).abs()[scale == 0]
        assert torch.isfinite(out[k][finite]).all()
        assert float(err.max()) <= 1e-6, (k, float(err.max()), int(err.max(1)[0].argmax()))
    # non-finite rows: the flagged blocks ARE the fp32 kernel's result (same nan / inf pattern as F.linear)
    bad = ~finite
    tr = torch.nn.functional.linear(X[bad], W)
    assert torch.equal(torch.isnan(out[True][bad]), torch.isnan(tr)) and torch.equal(torch.isinf(out[True][bad]), torch.isinf(tr))
    assert torch.equal(torch.isnan(out[True][bad]), torch.isnan(out[False][bad]))
    assert torch.equal(out[True][650], torch.zeros(64))


def test_linear_split_domain_guard_weights(ops, dev):
    """The same guard on W's side: a weight row of tiny magnitude (or a huge / non-finite weight) sends the whole call to the fp32
    kernel; with ordinary weights nothing is recomputed and the two kernels differ only at the 1e-7 level."""
    g = torch.Generator().manual_seed(5)
    n, F = 2000, 4096
    X = torch.relu(torch.randn(n, F, generator=g))
    for mod in ("tiny_row", "huge", "nan", "mixed_small"):
        W = torch.randn(64, F, generator=g) / F ** 0.5
        if mod == "tiny_row":
            W[17] *= 1e-7
        elif mod == "huge":
            W[3, 100] = 1e5
        elif mod == "nan":
            W[9, 0] = float("nan")
        else:
            W[:, ::3] *= 1e-9           # tiny elements inside rows of ordinary size: stays on the split kernel, error norm-wise small
        out = {}
        try:
            for split in (True, False):
                ops.LINEAR_F16X3 = split
                out[split] = ops.linear(X.to(dev), W.to(dev), None).cpu()
        finally:
            ops.LINEAR_F16X3 = True
        if mod == "nan":
            assert torch.isnan(out[True][:, 9]).all() and torch.isfinite(out[True][:, :9]).all()
            continue
        ref = X.double() @ W.double().t()
        scale = X.double().abs() @ W.double().abs().t()
        # 'huge': one product of 1e5 |x| sits among 4095 of ~5e-3 -- every later addition rounds at ulp(1e5): ANY fp32 accumulation
        # chain is off by ~sqrt(K) 2^-24 of the sum there (the flagged call runs the fp32 kernel with the whole K per workgroup)
        tol = 2e-5 if mod == "huge" else 1e-6
        for k in (True, False):
            assert float(((out[k].double() - ref).abs() / scale).max()) <= tol, (mod, k)


def test_linear_split_guard_is_capture_safe(ops, dev):
    """The guard decides on the device (flags in the workspace, a second launch that returns at once): the call can be captured
    in a hipGraph and replayed on inputs that flip a block in and out of the domain."""
    g = torch.Generator().manual_seed(11)
    X = torch.relu(torch.randn(1500, 4096, generator=g)).to(dev)
    W = (torch.randn(64, 4096, generator=g) / 64).to(dev)
    Y = ops.linear(X, W, None)                      # warm-up (workspace allocation) outside the capture
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        Y = ops.linear(X, W, None)
    for scale in (1.0, 1e-8, 1.0, 2e5):
        X[200].normal_(generator=None).abs_().mul_(scale)
        graph.replay()
        torch.cuda.synchronize()
        ref = X.double().cpu() @ W.double().cpu().t()
        sc = X.double().cpu().abs() @ W.double().cpu().abs().t()
        assert float(((Y.double().cpu() - ref).abs() / sc).max()) <= 1e-6, scale


@pytest.mark.parametrize("n,F", [(7050, 4096), (40_000, 4096), (513, 4480), (7050, 384), (2048, 4096)])
def test_linear_split_backward_is_as_accurate_as_the_fp32_kernels(ops, dev, n, F):
    """mmrec_linear_bwd_split_f32 (ABI 11; the default backward with hip_ops.LINEAR_F16X3): dW = dY^T X, db, dX = dY W on the 16-bit
    matrix cores with split operands (dW: three bf16 parts; dX: two fp16 halves, power-of-two scaled), against float64, next to
    the fp32-MFMA kernels.  Gradients of realistic and of hostile magnitude: rows of dY spread over 1e-2 ... 1e-12 (most rows
    exactly zero: items outside the batch), one output column of dY 1e-8 of the others, feature weights (columns of W) of 1e-8, a
    feature (column of X) of 1e-7 and one of 1e5 (outside what fp16 halves of X could hold: dW's parts carry fp32's exponent),
    errors measured against sum |a b| of each output."""
    g = torch.Generator().manual_seed(n + F)
    X = torch.relu(torch.randn(n, F, generator=g))
    X[:, 5] *= 1e-7
    X[:, 300] *= 1e5
    X[:, 301] = 0.0
    W = torch.randn(64, F, generator=g) / F ** 0.5
    W[:, 7] *= 1e-8
    W[:, 200:210] *= 1e4
    W[:, 333] = 0.0
    dY = torch.randn(n, 64, generator=g) * 10.0 ** (-2 - 10 * torch.rand(n, 1, generator=g))
    dY[torch.rand(n, generator=g) < 0.6] = 0.0
    dY[:, 11] *= 1e-8
    dY[:, 12] = 0.0
    Xd, Wd, bd = X.to(dev).requires_grad_(), W.to(dev).requires_grad_(), torch.zeros(64, device=dev, requires_grad=True)
    out = {}
    try:
        for split in (True, False):
            ops.LINEAR_F16X3 = split
            Xd.grad = Wd.grad = bd.grad = None
            ops.linear(Xd, Wd, bd).backward(dY.to(dev))
            out[split] = (Wd.grad.cpu().double(), bd.grad.cpu().double(), Xd.grad.cpu().double())
    finally:
        ops.LINEAR_F16X3 = True
    dY64, X64, W64 = dY.double(), X.double(), W.double()
    ref = (dY64.t() @ X64, dY64.sum(0), dY64 @ W64)
    scale = (dY64.abs().t() @ X64.abs(), dY64.abs().sum(0), dY64.abs() @ W64.abs())
    for k in (True, False):
        for j, what in enumerate(("dW", "db", "dX")):
            err = (out[k][j] - ref[j]).abs() / (scale[j] + 1e-300)
            err[scale[j] == 0] = (out[k][j] - ref[j]).abs()[scale[j] == 0]
            # dW's column 300 (|x| ~ 1e5 among ~1: every later addition of an fp32 chain rounds at the large term's ulp)
            tol = 1e-6 if what != "dW" else 2e-5
            assert float(err.max()) <= tol, (k, what, float(err.max()))
            if what == "dW":
                keep = torch.ones(F, dtype=torch.bool)
                keep[300] = False
                assert float(err[:, keep].max()) <= 1e-6, (k, what, float(err[:, keep].max()))
    assert torch.equal(out[True][0][12], torch.zeros(F, dtype=torch.float64)) and float(out[True][1][12]) == 0.0   # zero gradient column: exact
    assert torch.equal(out[True][2][dY.abs().sum(1) == 0], torch.zeros(int((dY.abs().sum(1) == 0).sum()), F, dtype=torch.float64))


def test_linear_split_backward_nonfinite_and_extremes(ops, dev):
    """inf / NaN gradients propagate through the split backward as through F.linear's (same pattern of non-finite outputs);
    gradients of 1e30 and 1e-31 magnitude keep fp32's relative accuracy (dX: rows scaled by exact powers of two; dW: the three
    bf16 parts carry fp32's exponent range).  Below 2^-110 (7.7e-34) the LOW parts of an entry fall under bf16's smallest
    denormal (2^-133): the entry is then carried to an absolute 2^-133 instead of a relative 2^-24 ('floor')."""
    g = torch.Generator().manual_seed(3)
    n, F = 700, 512
    X = torch.relu(torch.randn(n, F, generator=g))
    W = torch.randn(64, F, generator=g) / F ** 0.5
    for mod in ("nan", "inf", "huge", "tiny", "floor"):
        dY = torch.randn(n, 64, generator=g) * 1e-3
        if mod == "nan":
            dY[17, 3] = float("nan")
        elif mod == "inf":
            dY[17, 3] = float("inf")
        elif mod == "huge":
            dY *= 1e33
        elif mod == "tiny":
            dY *= 1e-28
        else:
            dY *= 1e-33
        Xd, Wd = X.to(dev).requires_grad_(), W.to(dev).requires_grad_()
        ops.linear(Xd, Wd, None).backward(dY.to(dev))
        dW, dX = Wd.grad.cpu(), Xd.grad.cpu()
        rW, rX = dY.t() @ X, dY @ W
        if mod in ("nan", "inf"):
            assert torch.equal(torch.isfinite(dX), torch.isfinite(rX))
            assert torch.equal(torch.isfinite(dW), torch.isfinite(rW))
            fin = torch.isfinite(rX)
            np.testing.assert_allclose(dX[fin].numpy(), rX[fin].numpy(), rtol=1e-4, atol=1e-9)
        else:
            r64W, r64X = dY.double().t() @ X.double(), dY.double() @ W.double()
            sW, sX = dY.double().abs().t() @ X.double().abs(), dY.double().abs() @ W.double().abs()
            floor = 2.0 ** -132 * X.double().abs().sum(0)[None, :] if mod == "floor" else 0.0
            assert bool(((dW.double() - r64W).abs() <= 1e-6 * sW + floor).all()), mod
            assert float(((dX.double() - r64X).abs() / sX).max()) <= 1e-6, mod


def test_linear_split_backward_is_capture_safe_and_deterministic(ops, dev):
    """the backward's scales live on the device: capturable, replayable on changed gradients, and two runs agree bit for bit (no
    float atomics: slabs and bias-gradient partials are summed in fixed order)."""
    g = torch.Generator().manual_seed(12)
    X = torch.relu(torch.randn(3000, 4096, generator=g)).to(dev).requires_grad_()
    W = (torch.randn(64, 4096, generator=g) / 64).to(dev).requires_grad_()
    b = torch.zeros(64, device=dev, requires_grad=True)
    G = (torch.randn(3000, 64, generator=g) * 1e-4).to(dev)
    ops.linear(X, W, b).backward(G)
    first = [t.grad.clone() for t in (X, W, b)]
    X.grad = W.grad = b.grad = None
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        ops.linear(X, W, b).backward(G)
    graph.replay()
    torch.cuda.synchronize()
    for a, t in zip(first, (X, W, b)):
        assert torch.equal(a, t.grad)
    G.mul_(1e-6)
    graph.replay()
    torch.cuda.synchronize()
    ref = G.double().cpu().t() @ X.detach().double().cpu()
    sc = G.double().cpu().abs().t() @ X.detach().double().cpu().abs()
    assert float(((W.grad.double().cpu() - ref).abs() / sc).max()) <= 1e-6


@pytest.mark.parametrize("n,F,out,bias", [(300, 96, 256, True), (1000, 256, 256, False), (129, 384, 384, True),
                                          (777, 4096, 256, True)])
def test_linear_wide_fwd_bwd(ops, dev, n, F, out, bias):
    """out = 64 j > 64 (MMGCN's MLP / convolution weights): general GEMM forward and dX, blocked dW / db."""
    g = torch.Generator().manual_seed(n + F + out)
    X = torch.randn(n, F, generator=g).requires_grad_()
    W = (torch.randn(out, F, generator=g) / F ** 0.5).requires_grad_()
    b = torch.randn(out, generator=g).requires_grad_() if bias else None
    G = torch.randn(n, out, generator=g)
    ref = orc.linear(X, W, b)
    ref.backward(G)
    Xd, Wd = X.detach().to(dev).requires_grad_(), W.detach().to(dev).requires_grad_()
    bd = b.detach().to(dev).requires_grad_() if bias else None
    Y = ops.linear(Xd, Wd, bd)
    Y.backward(G.to(dev))
    assert rel_fro(Y, ref) < 1e-6
    assert rel_fro(Wd.grad, W.grad) < 1e-6 and rel_fro(Xd.grad, X.grad) < 1e-6
    close(Y, ref, atol=1e-5)
    close(Wd.grad, W.grad, atol=1e-4)
    close(Xd.grad, X.grad, atol=1e-5)
    if bias:
        close(bd.grad, b.grad, atol=1e-4)


# ---------------------------------------------------------------------------------------- top-K
def _topk_check(ops, dev, Q, C, k, mask=None, exact_gap=1e-5):
    nq, nc = Q.shape[0], C.shape[0]
    scores = torch.from_numpy(Q) @ torch.from_numpy(C).t()
    if mask is None:
        mask = np.zeros((2, 0), dtype=np.int64)
    ref_v, ref_i = orc.mask_topk(scores, mask, k)
    rp, col = ops.mask_to_csr(mask, nq, dev)
    idx, val = ops.score_topk(D(Q, dev), D(C, dev), k, rp, col, return_values=True)
    idx, val = idx.cpu().numpy(), val.cpu().numpy()
    assert idx.min() >= 0 and idx.max() < nc
    np.testing.assert_allclose(val, ref_v.numpy(), rtol=1e-4, atol=1e-5)       # same sorted scores
    assert np.all(np.diff(val, axis=1) <= 0)                                    # sorted descending
    s = scores.clone()
    s[torch.as_tensor(mask[0]), torch.as_tensor(mask[1])] = -1e10
    got_scores = np.take_along_axis(s.numpy(), idx, axis=1)
    np.testing.assert_allclose(got_scores, val, rtol=1e-4, atol=1e-5)           # idx really has that score
    for r in range(nq):
        assert len(set(idx[r])) == k                                            # no duplicates
        diff = set(idx[r]) ^ set(ref_i[r].numpy())
        for j in diff:   # only near-ties at the k-th score may differ (fp32 accumulation order)
            assert abs(s[r, j].item() - ref_v[r, -1].item()) <= exact_gap * max(1.0, abs(ref_v[r, -1].item()))
    return idx


def test_topk_eval_shape_with_mask(ops, dev):
    rng = np.random.default_rng(0)
    nq, nc = 100, 1500     # nq not a multiple of 32, nc not a multiple of 32
    Q = rng.standard_normal((nq, 64)).astype(np.float32) * 0.3
    C = rng.standard_normal((nc, 64)).astype(np.float32) * 0.3
    rows = rng.integers(0, nq, 3000)
    cols = rng.integers(0, nc, 3000)
    key = np.unique(rows * nc + cols)
    mask = np.stack([key // nc, key % nc])
    mask = mask[:, rng.permutation(mask.shape[1])]    # unsorted, as the loader hands it over
    _topk_check(ops, dev, Q, C, 50, mask)
    _topk_check(ops, dev, Q, C, 1, mask)
    _topk_check(ops, dev, Q, C, 64, None)


@pytest.mark.parametrize("nq,nc,k", [(70, 5000, 50), (40, 40000, 20), (33, 3300, 64)])
def test_topk_two_pass_and_split(ops, dev, nq, nc, k):
    """kd = 64 and nc >= 64k candidates: group-max lower bound pass + candidate splits + merge."""
    rng = np.random.default_rng(nc)
    Q = rng.standard_normal((nq, 64)).astype(np.float32) * 0.3
    C = rng.standard_normal((nc, 64)).astype(np.float32) * 0.3
    rows = rng.integers(0, nq, 4000)
    cols = rng.integers(0, nc, 4000)
    # mask each query's best few candidates on purpose (train positives score high)
    best = np.argsort(-(Q @ C.T), axis=1)[:, :5]
    rows = np.concatenate([rows, np.repeat(np.arange(nq), 5)])
    cols = np.concatenate([cols, best.reshape(-1)])
    key = np.unique(rows * nc + cols)
    _topk_check(ops, dev, Q, C, k, np.stack([key // nc, key % nc]))
    _topk_check(ops, dev, Q, C, k, None)


def test_topk_materialised_blocks_and_heavy_masks(ops, dev):
    """kd = 64 path at a size that needs two score blocks; one user masks 600 items (more than the
    128 - k groups the lower bound can absorb: threshold-free fallback), one masks its whole top-300."""
    rng = np.random.default_rng(7)
    nq, nc, k = 17000, 5000, 50
    Q = rng.standard_normal((nq, 64)).astype(np.float32) * 0.3
    C = rng.standard_normal((nc, 64)).astype(np.float32) * 0.3
    rows = rng.integers(0, nq, 60000)
    cols = rng.integers(0, nc, 60000)
    heavy = rng.choice(nc, 600, replace=False)
    top300 = np.argsort(-(Q[9000] @ C.T))[:300]
    rows = np.concatenate([rows, np.full(600, 123), np.full(300, 9000)])
    cols = np.concatenate([cols, heavy, top300])
    key = np.unique(rows.astype(np.int64) * nc + cols)
    _topk_check(ops, dev, Q, C, k, np.stack([key // nc, key % nc]))


@pytest.mark.parametrize("kd", [96, 384, 4096])
def test_topk_general_kd_materialised(ops, dev, kd):
    """kd % 32 == 0, kd != 64: S = Q C^T by the 128x128 LDS-DMA GEMM (no transposes), bound from an
    in-kernel sweep; sizes that are not tile multiples, with masks."""
    rng = np.random.default_rng(kd)
    nq, nc, k = 300, 2100, 10
    Q = (rng.standard_normal((nq, kd)) / np.sqrt(kd)).astype(np.float32)
    C = (rng.standard_normal((nc, kd)) / np.sqrt(kd)).astype(np.float32)
    best = np.argsort(-(Q @ C.T), axis=1)[:, :3]
    rows = np.concatenate([rng.integers(0, nq, 2000), np.repeat(np.arange(nq), 3)])
    cols = np.concatenate([rng.integers(0, nc, 2000), best.reshape(-1)])
    key = np.unique(rows.astype(np.int64) * nc + cols)
    _topk_check(ops, dev, Q, C, k, np.stack([key // nc, key % nc]), exact_gap=1e-4)
    _topk_check(ops, dev, Q, C, 50, None, exact_gap=1e-4)


@pytest.mark.parametrize("kd,nc,k", [(64, 2100, 100), (64, 3000, 128), (96, 3000, 70), (256, 5000, 100), (64, 150, 128)])
def test_topk_k_up_to_128_on_the_fp32_block_path(ops, dev, kd, nc, k):
    """ABI 9: `select_topk_kernel` hands out up to 128 sorted entries (ranks 64 .. k - 1 from the second half of its
    128-entry sort), so every row width that is a multiple of 32 serves k <= 128 -- candidate sets below 4096 rows at kd = 64 /
    128 and the wide rows above k = 64 no longer send `topk: [.., 100]` runs to the dense torch path.  Masks incl. the best
    candidates, a candidate set smaller than 2 k, duplicated candidates (ties: lower id first)."""
    rng = np.random.default_rng(kd + nc + k)
    nq = 77
    Q = (rng.standard_normal((nq, kd)) / np.sqrt(kd)).astype(np.float32)
    C = (rng.standard_normal((nc, kd)) / np.sqrt(kd)).astype(np.float32)
    C[20:25] = C[3]
    best = np.argsort(-(Q @ C.T), axis=1)[:, :3]
    rows = np.concatenate([rng.integers(0, nq, 900), np.repeat(np.arange(nq), 3)])
    cols = np.concatenate([rng.integers(0, nc, 900), best.reshape(-1)])
    key = np.unique(rows.astype(np.int64) * nc + cols)
    mask = np.stack([key // nc, key % nc])
    idx = _topk_check(ops, dev, Q, C, k, mask, exact_gap=1e-4)
    _topk_check(ops, dev, Q, C, min(k, 65), None, exact_gap=1e-4)
    for r in range(nq):                                   # the five copies of candidate 3 appear in id order wherever they appear
        pos = [int(np.nonzero(idx[r] == c)[0][0]) for c in (3, 20, 21, 22, 23, 24) if c in idx[r]]
        assert pos == sorted(pos)


@pytest.mark.parametrize("kd,nq,nc,k", [(192, 500, 22_003, 10), (384, 300, 20_011, 32), (4096, 260, 4500, 10), (4096, 129, 9000, 1)])
def test_topk_wide_rows_fp16_pass_with_exact_refinement(ops, dev, kd, nq, nc, k):
    """kd = 192 ... 4096, >= 4096 candidates, k <= 32 (csrc/topk_wide.h: the kNN builds over raw features, freedom.py:79-91):
    fp16 matrix-core scores -> the 64 best approximate candidates per query -> margin test -> exact fp32 re-scoring, and a
    rescue queue through the exact fp32 path for queries whose scores are too closely packed.  Row-normalised features with a
    large common component (the relu image features), duplicated candidates (exact ties: lower id first), query rows scaled by
    2^-30 and 2^+20, masks incl. a query's best candidates, one query whose candidates are ALL within 1e-4 of each other
    (fails the margin test: rescue queue) and, in one case, 80 such queries (more than one rescue workgroup) -- vs
    orc.mask_topk with the near-tie rule and vs the fp32 path (use_filter=False); two calls are bitwise identical."""
    rng = np.random.default_rng(kd + nc + k)
    base = np.maximum(rng.standard_normal((nc, kd)), 0).astype(np.float32) if kd == 4096 else \
        rng.standard_normal((nc, kd)).astype(np.float32) + 0.5
    C = base / np.linalg.norm(base, axis=1, keepdims=True)
    Q = C[rng.choice(nc, nq, replace=False)].copy()            # kNN-like: the queries ARE candidates (self-similarity 1)
    Q[3] *= np.float32(2.0) ** -30
    Q[4] *= np.float32(2.0) ** 20
    C[200:260] = C[77]                                          # 60 identical candidates ...
    Q[5] = C[77]                                                # ... that are this query's best
    n_flat = 80 if kd == 384 else 1
    flat = rng.standard_normal(kd).astype(np.float32)
    flat /= np.linalg.norm(flat)
    C[1000:1400] = flat + (rng.standard_normal((400, kd)) * 1e-5).astype(np.float32)     # 400 candidates within ~1e-4 of each other
    Q[10:10 + n_flat] = flat + (rng.standard_normal((n_flat, kd)) * 1e-5).astype(np.float32)
    best = np.argsort(-(Q[:40].astype(np.float64) @ C.astype(np.float64).T), axis=1)[:, :3]
    rows = np.concatenate([rng.integers(0, nq, 5 * nq), np.repeat(np.arange(40), 3)])
    cols = np.concatenate([rng.integers(0, nc, 5 * nq), best.reshape(-1)])
    key = np.unique(rows.astype(np.int64) * nc + cols)
    mask = np.stack([key // nc, key % nc])
    from tests.test_topk_fuzz_gpu import check_lists
    rp, col = ops.mask_to_csr(mask, nq, dev)
    Qd, Cd = D(Q, dev), D(C, dev)
    idx, val = ops.score_topk(Qd, Cd, k, rp, col, return_values=True)
    check_lists("wide rows kd %d" % kd, idx.cpu(), val.cpu(), torch.from_numpy(Q), torch.from_numpy(C), mask, k)
    if k > 1:
        masked5 = set(mask[1][mask[0] == 5].tolist())
        ties = [c for c in [77] + list(range(200, 260)) if c not in masked5]
        assert idx[5].cpu().tolist()[:min(k, len(ties))] == sorted(ties)[:min(k, len(ties))]      # exact ties: lowest ids first
    again = ops.score_topk(Qd, Cd, k, rp, col, return_values=True)
    assert torch.equal(again[0], idx) and torch.equal(again[1], val)
    m_idx, m_val = ops.score_topk(Qd, Cd, k, rp, col, return_values=True, use_filter=False)
    np.testing.assert_allclose(val.cpu().numpy(), m_val.cpu().numpy(), rtol=2e-5, atol=1e-6)
    assert (m_idx == idx).float().mean() > 0.97           # (the flat queries' 400 near-identical candidates order differently)


def test_topk_adversarial_ascending_scores(ops, dev):
    """scores increase with the candidate id: every candidate beats the threshold (max compactions)."""
    nq, nc = 33, 7000     # two-pass path: every group maximum is its last candidate
    Q = np.zeros((nq, 64), np.float32)
    Q[:, 0] = 1.0
    C = np.zeros((nc, 64), np.float32)
    C[:, 0] = np.arange(nc, dtype=np.float32)
    idx = _topk_check(ops, dev, Q, C, 50)
    assert np.array_equal(idx[0], np.arange(nc - 1, nc - 51, -1))


def test_topk_more_k_than_unmasked_and_ties(ops, dev):
    """K > #unmasked items: masked items (-1e10) fill the tail like the reference; ties -> lower id."""
    nq, nc, k = 5, 40, 20
    Q = np.ones((nq, 64), np.float32)
    C = np.zeros((nc, 64), np.float32)   # all scores tie at 0
    mask = np.stack([np.repeat(np.arange(nq), 30), np.tile(np.arange(5, 35), nq)])
    rp, col = ops.mask_to_csr(mask, nq, dev)
    idx, val = ops.score_topk(D(Q, dev), D(C, dev), k, rp, col, return_values=True)
    idx, val = idx.cpu().numpy(), val.cpu().numpy()
    unmasked = [0, 1, 2, 3, 4, 35, 36, 37, 38, 39]
    for r in range(nq):
        assert idx[r, :10].tolist() == unmasked           # ties broken by lower id
        assert np.all(val[r, :10] == 0) and np.all(val[r, 10:] == np.float32(-1e10))
        assert idx[r, 10:].tolist() == list(range(5, 15))


def test_topk_filter_path_edge_cases(ops, dev, monkeypatch):
    """kd = 64, nc >= 4096: the fp16 filter + exact refinement path (topk_filter.hip) and its on-device slow
    queue.  (a) all scores tie and K > #unmasked for some queries: every list overflows -> streaming exact
    top-k, ties by lower id, masked items at -1e10 fill the tail; (b) random data, filter vs materialised
    path (use_filter=False): same ids, scores equal to fp32 rounding; (c) tiny score gaps (1e-6 relative,
    far below the fp16 filter's error): the exact refinement still orders them."""
    nq, nc, k = 70, 4500, 20
    Q = np.ones((nq, 64), np.float32)
    C = np.zeros((nc, 64), np.float32)
    rows = np.concatenate([np.repeat(np.arange(5), nc - 10), np.repeat(np.arange(5, nq), 3)])
    cols = np.concatenate([np.tile(np.arange(5, nc - 5), 5), np.tile(np.array([0, 1, 7]), nq - 5)])
    mask = np.stack([rows, cols])
    rp, col = ops.mask_to_csr(mask, nq, dev)
    idx, val = ops.score_topk(D(Q, dev), D(C, dev), k, rp, col, return_values=True)
    idx, val = idx.cpu().numpy(), val.cpu().numpy()
    for r in range(5):        # 10 unmasked items, then the lowest masked ids at -1e10
        assert idx[r, :10].tolist() == [0, 1, 2, 3, 4] + list(range(nc - 5, nc))
        assert np.all(val[r, :10] == 0) and np.all(val[r, 10:] == np.float32(-1e10))
        assert idx[r, 10:].tolist() == list(range(5, 15))
    want = [c for c in range(40) if c not in (0, 1, 7)][:k]
    for r in range(5, nq):
        assert idx[r].tolist() == want and np.all(val[r] == 0)
    # (b)
    rng = np.random.default_rng(3)
    nq, nc, k = 3000, 9000, 50
    Qr = rng.standard_normal((nq, 64)).astype(np.float32) * 0.2
    Cr = rng.standard_normal((nc, 64)).astype(np.float32) * 0.2
    key = np.unique(rng.integers(0, nq, 30000).astype(np.int64) * nc + rng.integers(0, nc, 30000))
    rp, col = ops.mask_to_csr(np.stack([key // nc, key % nc]), nq, dev)
    out = {}
    for mode in ("1", "0"):
        i_, v_ = ops.score_topk(D(Qr, dev), D(Cr, dev), k, rp, col, return_values=True, use_filter=mode == "1")
        out[mode] = (i_.cpu().numpy(), v_.cpu().numpy())
    np.testing.assert_allclose(out["1"][1], out["0"][1], rtol=2e-6, atol=2e-6)
    assert np.mean(out["1"][0] == out["0"][0]) > 0.999
    # (c) candidates c0..c63 = base * (1 + j * 1e-6): gaps of ~1e-6 relative
    nq, nc, k = 64, 6000, 32
    base = rng.standard_normal(64).astype(np.float32)
    Cg = rng.standard_normal((nc, 64)).astype(np.float32) * 0.01
    Cg[100:164] = base[None, :] * (1.0 + np.arange(64, dtype=np.float32)[:, None] * 1e-6)
    Qg = np.tile(base, (nq, 1)).astype(np.float32)
    _topk_check(ops, dev, Qg, Cg, k, None, exact_gap=1e-6)
    # (d) round-1 review: HETEROGENEOUS query norms.  With one global query scale a row 2^-30 below the largest one fell
    # into fp16 subnormals, the filter's error bound no longer covered its rounding error and pass 2 could drop a true
    # top-k item silently; queries are now scaled row by row.  Every row is checked in ITS OWN units (float64 scores).
    nq, nc, k = 300, 6000, 50
    Qh = (rng.standard_normal((nq, 64)) * 0.2).astype(np.float32)
    for r, e in enumerate((-20, -24, -28, -30, -34, -40, -60, -100)):
        Qh[r] *= np.float32(2.0) ** e
    Qh[20] = 0.0                                    # all scores tie at 0: the k lowest unmasked ids (final kernel's shortcut)
    Qh[21] *= np.float32(1e25)
    Qh[22, 1:] *= np.float32(2.0) ** -30            # one dominant element
    Ch = (rng.standard_normal((nc, 64)) * 0.2 + 0.5).astype(np.float32)     # common component: exercises the centring
    Ch[:1500] *= np.float32(2.0) ** -12              # candidates of very different norms
    key = np.unique(rng.integers(0, nq, 2000).astype(np.int64) * nc + rng.integers(0, nc, 2000))
    mask = np.stack([key // nc, key % nc])
    rp, col = ops.mask_to_csr(mask, nq, dev)
    idx, val = ops.score_topk(D(Qh, dev), D(Ch, dev), k, rp, col, return_values=True)
    idx, val = idx.cpu().numpy(), val.cpu().numpy().astype(np.float64)
    s64 = Qh.astype(np.float64) @ Ch.astype(np.float64).T
    s64[mask[0], mask[1]] = -np.inf
    for r in range(nq):
        unit = max(np.abs(s64[r][np.isfinite(s64[r])]).max(), 1e-300)
        order = np.lexsort((np.arange(nc), -s64[r]))[:k]
        assert len(set(idx[r])) == k and not np.isin(idx[r], mask[1][mask[0] == r]).any()
        if r == 20:
            assert idx[r].tolist() == order.tolist() and np.all(val[r] == 0)
            continue
        np.testing.assert_allclose(val[r], s64[r][idx[r]], rtol=0, atol=2e-6 * unit)
        assert np.all(np.diff(val[r]) <= 0)
        for j in set(idx[r]) ^ set(order):          # only fp32-rounding near-ties at the k-th score may differ
            assert abs(s64[r, j] - s64[r, order[-1]]) <= 2e-6 * unit, (r, j)


def test_topk_slow_queue_split_over_workgroups(ops, dev):
    """>= 65,536 candidates: a query the filter cannot serve is split over 16 workgroups whose partial top-k lists are
    merged (one workgroup streaming 500K candidates for one flagged query took as long as a whole 20,000-query block).
    (a) all scores tie: every query is flagged; lowest unmasked ids win, masked ids at -1e10 fill a short list;
    (b) random data, two heavy users (k + #masked > the 512 group maxima) among ordinary ones: the oracle's top-k."""
    nq, nc, k = 6, 70_000, 20
    Q, C = np.ones((nq, 64), np.float32), np.zeros((nc, 64), np.float32)
    rows = np.concatenate([np.repeat(0, nc - 7), np.repeat(np.arange(1, nq), 2)])
    cols = np.concatenate([np.arange(3, nc - 4), np.tile(np.array([0, 5]), nq - 1)])
    rp, col = ops.mask_to_csr(np.stack([rows, cols]), nq, dev)
    idx, val = ops.score_topk(D(Q, dev), D(C, dev), k, rp, col, return_values=True)
    idx, val = idx.cpu().numpy(), val.cpu().numpy()
    assert idx[0, :7].tolist() == [0, 1, 2] + list(range(nc - 4, nc)) and idx[0, 7:].tolist() == list(range(3, 16))
    assert np.all(val[0, :7] == 0) and np.all(val[0, 7:] == np.float32(-1e10))
    for r in range(1, nq):
        assert idx[r].tolist() == [c for c in range(30) if c not in (0, 5)][:k] and np.all(val[r] == 0)
    rng = np.random.default_rng(11)
    nq = 40
    Qr = rng.standard_normal((nq, 64)).astype(np.float32) * 0.2
    Cr = rng.standard_normal((nc, 64)).astype(np.float32) * 0.2
    heavy = {3: rng.choice(nc, 700, replace=False), 17: rng.choice(nc, 5000, replace=False)}
    rows = np.concatenate([np.repeat(q, len(v)) for q, v in heavy.items()] + [np.arange(nq)])
    cols = np.concatenate([v for v in heavy.values()] + [rng.integers(0, nc, nq)])
    key = np.unique(rows.astype(np.int64) * nc + cols)
    _topk_check(ops, dev, Qr, Cr, 50, np.stack([key // nc, key % nc]))


@pytest.mark.parametrize("nq,nc", [(300, 32768), (517, 40_037), (64, 131_072 + 5)])
def test_topk_filter_word_lists(ops, dev, nq, nc):
    """>= 32,768 candidates: pass 2 appends its non-zero 64-bit pass / fail words to per-query lists (atomic slot
    counters) instead of writing rows of nc / 8 bytes per query, and the final kernel decodes the lists (arbitrary order:
    the outputs stay deterministic through the final sort).  Exact boundary size, ragged last stage / tile, query
    counts that are not multiples of 256, realistic masks, one query whose candidates tie massively (> 256 non-zero
    words: more than the final kernel's slots -> overflow queue), one heavy user; checked against the oracle; two calls give identical results."""
    rng = np.random.default_rng(nq + nc)
    k = 50
    Q = rng.standard_normal((nq, 64)).astype(np.float32) * 0.2
    C = (rng.standard_normal((nc, 64)) * 0.2 + 0.3).astype(np.float32)
    C[1000:1000 + 20_000:40] = 1.0                     # 500 identical candidates ...
    Q[7] = 1.0                                         # ... that are this query's best: > 256 words pass -> overflow
    rows = np.concatenate([rng.integers(0, nq, 20 * nq), np.repeat(11, 3000)])
    cols = np.concatenate([rng.integers(0, nc, 20 * nq), rng.choice(nc, 3000, replace=False)])
    key = np.unique(rows.astype(np.int64) * nc + cols)
    mask = np.stack([key // nc, key % nc])
    idx = _topk_check(ops, dev, Q, C, k, mask)
    masked7 = set(mask[1][mask[0] == 7].tolist())
    assert idx[7].tolist() == [c for c in range(1000, 21_000, 40) if c not in masked7][:k]      # ties: lowest ids first
    rp, col = ops.mask_to_csr(mask, nq, dev)
    a = ops.score_topk(D(Q, dev), D(C, dev), k, rp, col, return_values=True)
    b = ops.score_topk(D(Q, dev), D(C, dev), k, rp, col, return_values=True)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert np.array_equal(a[0].cpu().numpy(), idx)


@pytest.mark.parametrize("nq,nc,k,kd", [(200, 200_003, 50, 64), (64, 131_072, 100, 64), (130, 150_001, 20, 128)])
def test_topk_filter_subsampled_pass1(ops, dev, nq, nc, k, kd):
    """>= 131,072 candidates: pass 1 walks every second 64-candidate stage (its maxima over half the candidates still bound
    the (k + m)-th best from below), so ~2 (k + m) candidates survive into 512-entry word lists, and the <= 32 candidate rows
    of outlying norm are stored CLIPPED and always rescored (eps follows the largest stored norm).  Embeddings with a common
    component, one candidate of 20 x the typical norm (what a propagated low-degree item is at config 5) and 40 more of 2.5-12 x,
    a query whose best 1,200 candidates tie (> 1,024 words: overflow queue, then slow queue), heavy users with k + m below and above what
    the lists hold, against the oracle; the materialised path agrees; two calls are bitwise identical."""
    rng = np.random.default_rng(nq + nc + k)
    Q = (rng.standard_normal((nq, kd)) * 0.2 + 0.1).astype(np.float32)
    C = (rng.standard_normal((nc, kd)) * 0.2 + 0.1).astype(np.float32)
    C[77] *= 20.0
    big = rng.choice(np.arange(100_000, nc), 40, replace=False)      # more rows of outlying norm than the 32 the filter clips
    C[big] *= rng.uniform(2.5, 12.0, (40, 1)).astype(np.float32)
    C[5000:5000 + 1200 * 80:80] = 0.7
    Q[7] = 1.0
    Q[3] = 0.0                                                       # all scores are exactly 0: the k lowest unmasked ids
    Q[5] = -Q[5]                                                     # (the outliers are this query's WORST candidates)
    heavy = {11: 150, 13: 420, 19: 3000}
    rows = np.concatenate([rng.integers(0, nq, 10 * nq)] + [np.repeat(q, n) for q, n in heavy.items()] + [np.repeat(21, 20)])
    cols = np.concatenate([rng.integers(0, nc, 10 * nq)] + [rng.choice(nc, n, replace=False) for n in heavy.values()] +
                          [big[:20]])                                # a query with half the outliers masked
    key = np.unique(rows.astype(np.int64) * nc + cols)
    mask = np.stack([key // nc, key % nc])
    idx = _topk_check(ops, dev, Q, C, k, mask, exact_gap=2e-5 if kd > 64 else 1e-5)
    masked7 = set(mask[1][mask[0] == 7].tolist())
    outl = set(big.tolist()) | {77}
    ties = [c for c in range(5000, 5000 + 1200 * 80, 80) if c not in masked7]
    rest = [c for c in idx[7].tolist() if c not in outl]           # behind the outliers that beat them: the ties, lowest ids first
    assert len(rest) >= k - 41 and rest == ties[:len(rest)]
    masked3 = set(mask[1][mask[0] == 3].tolist())
    assert idx[3].tolist() == [c for c in range(k + len(masked3)) if c not in masked3][:k]
    rp, col = ops.mask_to_csr(mask, nq, dev)
    Qd, Cd = D(Q, dev), D(C, dev)
    a = ops.score_topk(Qd, Cd, k, rp, col, return_values=True)
    b = ops.score_topk(Qd, ops.TopkCandidates(Cd), k, rp, col, return_values=True)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and np.array_equal(a[0].cpu().numpy(), idx)
    m = ops.score_topk(Qd, Cd, min(k, 64), rp, col, return_values=True, use_filter=False)
    np.testing.assert_allclose(a[1][:, :min(k, 64)].cpu().numpy(), m[1].cpu().numpy(), rtol=1e-4, atol=1e-5)


def test_topk_filter_overflow_queue(ops, dev):
    """Queries with more surviving words than the final kernel has slots are ranked from their (twice as long) list by
    filter_overflow_kernel instead of by the slow queue's scan of all candidates.  (a) A dozen such queries among ordinary
    ones -- 900 tied best candidates each (ties: lowest ids win), one of them with 300 masked items (half of them among the
    ties) -- and a heavy user whose k + m = 420 alone asks for ~900 survivors.  (b) 1,300 tied candidates: more words than
    the list holds (1,024), so those queries are handed on to the slow queue: same answers."""
    rng = np.random.default_rng(17)
    nc, k = 140_003, 20
    for nq, n_tie_q, n_tied in ((300, 12, 900), (200, 10, 1300)):
        C = (rng.standard_normal((nc, 64)) * 0.2 + 0.1).astype(np.float32)
        tied = np.arange(3000, 3000 + n_tied * 100, 100)
        C[tied] = 0.7
        Q = (rng.standard_normal((nq, 64)) * 0.2 + 0.1).astype(np.float32)
        Q[:n_tie_q] = 1.0
        rows = [rng.integers(0, nq, 5 * nq), np.repeat(3, 300), np.repeat(40, 400)]
        cols = [rng.integers(0, nc, 5 * nq), np.concatenate([tied[:150], rng.choice(nc, 150, replace=False)]),
                rng.choice(nc, 400, replace=False)]                                               # 40: a heavy user, ordinary scores
        key = np.unique(np.concatenate(rows).astype(np.int64) * nc + np.concatenate(cols))
        mask = np.stack([key // nc, key % nc])
        rp, col = ops.mask_to_csr(mask, nq, dev)
        idx, val = ops.score_topk(D(Q, dev), D(C, dev), k, rp, col, return_values=True)
        idx, val = idx.cpu().numpy(), val.cpu().numpy()
        sample = np.unique(np.concatenate([np.arange(n_tie_q), [40], rng.integers(0, nq, 24)]))
        scores = torch.from_numpy(Q[sample]) @ torch.from_numpy(C).t()
        ref_v, ref_i = orc.mask_topk(scores, local_mask_rows(mask, sample), k)
        np.testing.assert_allclose(val[sample], ref_v.numpy(), rtol=1e-4, atol=1e-5)
        for j, r in enumerate(sample):
            if r < n_tie_q:                   # all ties: exactly the lowest unmasked tied ids
                masked = set(mask[1][mask[0] == r].tolist())
                assert idx[r].tolist() == [c for c in tied.tolist() if c not in masked][:k], r
            else:
                assert set(idx[r].tolist()) == set(ref_i[j].tolist()), r


def test_topk_very_many_queries_walked_in_blocks(ops, dev):
    """150,000 queries in ONE call: hip_ops.score_topk walks them in 65,536-query blocks against one preparation of the
    candidates (the filter's workspace is per query); the rows at the block seams and a random sample equal the oracle's, and
    the call equals three separate calls on the blocks bit for bit."""
    rng = np.random.default_rng(23)
    nq, nc, k = 150_000, 4500, 10
    Q = (rng.standard_normal((nq, 64)) * 0.2).astype(np.float32)
    C = (rng.standard_normal((nc, 64)) * 0.2 + 0.05).astype(np.float32)
    key = np.unique(np.repeat(np.arange(nq, dtype=np.int64), 3) * nc + rng.integers(0, nc, 3 * nq))
    mask = np.stack([key // nc, key % nc])
    rp, col = ops.mask_to_csr(mask, nq, dev)
    Qd, Cd = D(Q, dev), D(C, dev)
    idx, val = ops.score_topk(Qd, Cd, k, rp, col, return_values=True)
    sample = np.unique(np.concatenate([[0, 65535, 65536, 65537, 131071, 131072, nq - 1], rng.integers(0, nq, 200)]))
    scores = torch.from_numpy(Q[sample]) @ torch.from_numpy(C).t()
    ref_v, ref_i = orc.mask_topk(scores, local_mask_rows(mask, sample), k)
    np.testing.assert_allclose(val.cpu().numpy()[sample], ref_v.numpy(), rtol=1e-4, atol=1e-5)
    assert (idx.cpu().numpy()[sample] == ref_i.numpy()).mean() > 0.999
    rph = rp.cpu().numpy().astype(np.int64)
    for a in (0, 65536, 131072):
        b = min(a + 65536, nq)
        brp = (rp[a:b + 1] - int(rph[a])).contiguous()
        bcol = col[int(rph[a]):int(rph[b])].contiguous()
        bi, bv = ops.score_topk(Qd[a:b], Cd, k, brp, bcol, return_values=True)
        assert torch.equal(bi, idx[a:b]) and torch.equal(bv, val[a:b])


def local_mask_rows(mask, rows):
    """the [2, n] mask restricted to the ascending `rows`, row ids relative to the sample"""
    pos = np.searchsorted(rows, mask[0])
    hit = (pos < rows.shape[0]) & (rows[np.minimum(pos, rows.shape[0] - 1)] == mask[0])
    return np.stack([pos[hit], mask[1][hit]])


@pytest.mark.parametrize("nq,nc,k", [(600, 7050, 100), (300, 40_037, 128), (64, 70_001, 65), (200, 4096, 100)])
def test_topk_filter_k_up_to_128(ops, dev, nq, nc, k):
    """k = 65..128 on the fp16 filter path (kd = 64, >= 4096 candidates; `topk: [10, 20, 50, 100]` evaluates fused instead
    of through rocBLAS + torch.topk): rows of bits and word lists, the final kernel's two-register rank order, the slow queue
    (a heavy user whose k + #masked exceeds the group maxima; a query whose candidates all tie; >= 65,536 candidates: split
    over workgroups and merged) -- against orc.mask_topk (trainer.py:304-309); smaller candidate sets take the fp32 block path,
    row widths that are not a multiple of 32 say UNSUPPORTED."""
    from mmrec_amd._lib import MMRecHipError
    rng = np.random.default_rng(nq + nc + k)
    Q = rng.standard_normal((nq, 64)).astype(np.float32) * 0.2
    C = (rng.standard_normal((nc, 64)) * 0.2 + 0.1).astype(np.float32)
    C[100:100 + 300 * 7:7] = 0.5                       # 300 identical candidates ...
    Q[5] = 1.0                                         # ... that are this query's best: ties broken by id, lists overflow
    heavy = rng.choice(nc, 900, replace=False)         # k + m > 512 group maxima: slow queue
    rows = np.concatenate([rng.integers(0, nq, 10 * nq), np.repeat(9, heavy.shape[0])])
    cols = np.concatenate([rng.integers(0, nc, 10 * nq), heavy])
    key = np.unique(rows.astype(np.int64) * nc + cols)
    mask = np.stack([key // nc, key % nc])
    idx = _topk_check(ops, dev, Q, C, k, mask)
    masked5 = set(mask[1][mask[0] == 5].tolist())
    assert idx[5].tolist() == [c for c in range(100, 100 + 300 * 7, 7) if c not in masked5][:k]
    rp, col = ops.mask_to_csr(mask, nq, dev)
    a = ops.score_topk(D(Q, dev), D(C, dev), k, rp, col, return_values=True)
    assert np.array_equal(a[0].cpu().numpy(), idx)                       # repeatable
    _topk_check(ops, dev, Q, C[:3000].copy(), k)                         # below 4096 candidates: the fp32 block path (k <= 128 since ABI 9)
    with pytest.raises(MMRecHipError):
        ops.score_topk(D(Q[:, :40].copy(), dev), D(C[:3000, :40].copy(), dev), k)     # kd % 32 != 0: the fused fp32 path stops at 64


@pytest.mark.parametrize("nq,nc,k", [(600, 7050, 50), (300, 40_037, 50), (64, 70_001, 100), (517, 4096, 20)])
def test_topk_filter_rows_of_128(ops, dev, nq, nc, k):
    """kd = 128 on the fp16 filter path (VBPR / PGL / SELFCFED_LGN rank 128-wide rows: vbpr.py:100-106; they went through the
    materialised fp32 path before): each 64-candidate stage is walked as two 64-column tiles into the same accumulators, the
    exact refinement sums chunk j + chunk j + 16 first and then the kd = 64 tree.  Rows of bits and word lists, ties, a heavy
    user (slow queue), embeddings with a large common component, query rows of very different norms -- vs orc.mask_topk and
    vs the materialised path; prepared candidates give the same bits."""
    rng = np.random.default_rng(nq + nc + k)
    Q = (rng.standard_normal((nq, 128)) * 0.2 + 0.3).astype(np.float32)
    C = (rng.standard_normal((nc, 128)) * 0.2 + 0.3).astype(np.float32)
    Q[11] *= np.float32(2.0) ** -40
    Q[12] *= np.float32(2.0) ** 20
    C[100:100 + 300 * 7:7] = 0.5                       # 300 identical candidates ...
    Q[5] = 1.0                                         # ... that are this query's best
    heavy = rng.choice(nc, 900, replace=False)
    rows = np.concatenate([rng.integers(0, nq, 10 * nq), np.repeat(9, heavy.shape[0])])
    cols = np.concatenate([rng.integers(0, nc, 10 * nq), heavy])
    key = np.unique(rows.astype(np.int64) * nc + cols)
    mask = np.stack([key // nc, key % nc])
    idx = _topk_check(ops, dev, Q, C, k, mask, exact_gap=2e-5)
    masked5 = set(mask[1][mask[0] == 5].tolist())
    assert idx[5].tolist() == [c for c in range(100, 100 + 300 * 7, 7) if c not in masked5][:k]
    rp, col = ops.mask_to_csr(mask, nq, dev)
    Qd, Cd = D(Q, dev), D(C, dev)
    a = ops.score_topk(Qd, Cd, k, rp, col, return_values=True)
    b = ops.score_topk(Qd, ops.TopkCandidates(Cd), k, rp, col, return_values=True)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    m = ops.score_topk(Qd, Cd, min(k, 64), rp, col, return_values=True, use_filter=False)
    np.testing.assert_allclose(a[1][:, :min(k, 64)].cpu().numpy(), m[1].cpu().numpy(), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("nq,nc,k", [(300, 40_037, 50), (300, 40_037, 10), (130, 149_900, 20)])
def test_topk_filter_rows_of_128_padding_steps_score_real_candidates(ops, dev, nq, nc, k):
    """Round-3 advice: with kd = 128 a candidate range that holds an ODD number of whole stages (40,037: 25 in the last of 16
    ranges; 149,900: 61 virtual stages at pass-1 stride 2) ends pass 1's four-micro-step loop with two padding steps.  They
    used to re-load tile (last stage, column block 1) for BOTH column blocks: c[64:128] . q[0:64] + c[64:128] . q[64:128] --
    no candidate's score -- went into the group maxima and so into the bound.  Adversarial data for exactly that: the last
    whole stages' candidates are large in columns 64..127 along the direction the queries are large in in columns 0..63
    (bogus score ~ +60 against true scores of ~ 1), several masked items per query (k + m above the 32 groups of a range,
    so the inflated maxima would shift the rank the bound is read at).  Against orc.mask_topk."""
    rng = np.random.default_rng(nc + k)
    v = rng.standard_normal(64).astype(np.float32)
    v /= np.linalg.norm(v)
    Q = (rng.standard_normal((nq, 128)) * 0.1).astype(np.float32)
    Q[:, :64] += 8.0 * v                                  # queries: large along v in the FIRST column block
    C = (rng.standard_normal((nc, 128)) * 0.1).astype(np.float32)
    whole = nc // 64 * 64
    C[whole - 192:whole, 64:] += 8.0 * v                  # last whole stages: large along v in the SECOND column block
    rows = np.concatenate([rng.integers(0, nq, 30 * nq), np.repeat(np.arange(0, nq, 7), 40)])
    cols = np.concatenate([rng.integers(0, nc, 30 * nq), rng.integers(0, nc, 40 * len(range(0, nq, 7)))])
    key = np.unique(rows.astype(np.int64) * nc + cols)
    mask = np.stack([key // nc, key % nc])
    idx = _topk_check(ops, dev, Q, C, k, mask, exact_gap=2e-5)
    rp, col = ops.mask_to_csr(mask, nq, dev)
    m = ops.score_topk(D(Q, dev), D(C, dev), k, rp, col, use_filter=False)
    assert np.mean(m.cpu().numpy() == idx) > 0.999       # the materialised fp32 path agrees


@pytest.mark.parametrize("nq,nc", [(700, 7050), (300, 40_037), (100, 3000)])
def test_topk_prepared_candidates_identical(ops, dev, nq, nc):
    """mmrec_score_topk_prepared_f32 (ABI 7): the candidate side of the fp16 filter computed ONCE (hip_ops.TopkCandidates)
    and shared by several query blocks -- the batches of one evaluation, trainer.py:298-310 -- gives the very ids and
    values of the plain entry point, for every block, in both pass-2 forms (rows of bits / word lists) and for a shape the
    filter does not serve (no preparation exists: the plain path answers)."""
    rng = np.random.default_rng(nq * 7 + nc)
    k = 50
    Q = D(rng.standard_normal((nq, 64)).astype(np.float32) * 0.2, dev)
    C = D((rng.standard_normal((nc, 64)) * 0.2 + 0.3).astype(np.float32), dev)
    key = np.unique(rng.integers(0, nq, 12 * nq).astype(np.int64) * nc + rng.integers(0, nc, 12 * nq))
    mask = np.stack([key // nc, key % nc])
    cands = ops.TopkCandidates(C)
    assert (cands.prepared is not None) == (nc >= 4096)
    for a, b in ((0, nq), (0, nq // 3), (nq // 3, nq)):
        sel = (mask[0] >= a) & (mask[0] < b)
        rp, col = ops.mask_to_csr(np.stack([mask[0][sel] - a, mask[1][sel]]), b - a, dev)
        plain = ops.score_topk(Q[a:b].contiguous(), C, k, rp, col, return_values=True)
        prep = ops.score_topk(Q[a:b].contiguous(), cands, k, rp, col, return_values=True)
        assert torch.equal(plain[0], prep[0]) and torch.equal(plain[1], prep[1])


def test_topk_knn_shape(ops, dev, golden):
    """P6: kNN(k=10) over row-normalised features == freedom.py:79-82 on the golden features."""
    for key, k in (("image_feat", 10), ("text_feat", 10)):
        f = golden[key]
        fn = (f / np.linalg.norm(f, axis=1, keepdims=True)).astype(np.float32)
        _, _, knn = orc.knn_item_graph(f, k)
        idx = _topk_check(ops, dev, fn, fn, k)
        same = np.mean([set(a) == set(b) for a, b in zip(idx, knn)])
        assert same > 0.98
        assert np.all(idx[:, 0] == np.arange(f.shape[0]))   # self-similarity ranks first


def test_fullsort_eval_golden(ops, dev, golden):
    """Trainer.evaluate (trainer.py:292-311) on the golden LightGCN embeddings: top-50 sets and the
    metric dict (Recall/NDCG/Precision/MAP @5/10/20/50) identical to the reference's."""
    g = golden
    U, I = g["lgn_user_out"], g["lgn_item_out"]
    users = g["eval_users"]
    rp, col = ops.mask_to_csr(g["eval_mask"], users.shape[0], dev)
    idx = ops.score_topk(D(U[users], dev), D(I, dev), 50, rp, col).cpu().numpy()
    ref = g["lgn_topk"]
    assert np.mean([set(a) == set(b) for a, b in zip(idx, ref)]) == 1.0
    res = orc.topk_metrics(orc.hit_matrix(idx, g["eval_pos_flat"], g["eval_pos_len"]), g["eval_pos_len"])
    np.testing.assert_allclose([res[str(k)] for k in g["metric_keys"]], g["lgn_metrics"], atol=1e-4)


# ---------------------------------------------------------------------------------------- graph build
def test_graph_build_device_vs_golden(ops, dev, golden):
    g = golden
    nu, ni = int(g["n_users"]), int(g["n_items"])
    ei = D(g["edge_indices"], dev)
    ev = ops.edge_norm_values(ei[0].contiguous(), ei[1].contiguous(), nu, ni)
    close(ev, g["edge_values"], rtol=1e-6, atol=0)
    for pre in ("lay", "fr"):
        keep = D(g[pre + "_keep_idx"], dev)
        graph = ops.bipartite_graph_from_edges(ei[0][keep].contiguous(), ei[1][keep].contiguous(), nu, ni)
        idx, val = graph.to_coo_host()
        a = orc.coalesce_coo(idx, val, nu + ni, nu + ni)
        b = orc.coalesce_coo(g[pre + "_masked_idx"], g[pre + "_masked_val"], nu + ni, nu + ni)
        np.testing.assert_array_equal(a[0], b[0])                  # structure: exact
        np.testing.assert_allclose(a[1], b[1], rtol=1e-6)
        # stable: inside a row the entries keep the COO (reference) order
        rp_ref, ci_ref, _ = orc.coo_to_csr(g[pre + "_masked_idx"], g[pre + "_masked_val"], nu + ni)
        np.testing.assert_array_equal(graph.rowptr.cpu().numpy(), rp_ref)
        np.testing.assert_array_equal(graph.colidx.cpu().numpy(), ci_ref)


# ---------------------------------------------------------------------------------------- full sizes
def test_spmm_baby_shape_vs_oracle(ops, dev):
    from mmrec_amd import synth
    nu, ni, eu, ei = synth.shaped_edges("baby", seed=0)
    r, c, v = synth.sym_norm_coo(eu, ei, nu, ni)
    n = nu + ni
    g = ops.CsrGraph.from_coo_host(np.stack([r, c]), v, n, n, dev, symmetric=True)
    rng = np.random.default_rng(0)
    X = (rng.random((n, 64), dtype=np.float32) - 0.5) * 0.2
    # the oracle's reference-form operand: the UNCOALESCED row-sorted COO the reference multiplies by (freedom.py:113-126,172)
    adj = orc.sparse_coo(np.stack([r, c]), v, n)
    cur, ref = torch.from_numpy(X).to(dev), torch.from_numpy(X)
    for _ in range(3):
        Y = torch.empty_like(cur)
        ops.spmm_raw(g, cur, Y=Y)
        ref = torch.sparse.mm(adj, ref)
        cur = Y
    assert rel_fro(cur, ref) < 1e-6
    close(cur, ref, atol=1e-7)


def test_spmm_c5_properties(ops, dev):
    """10M-edge graph (nnz 20M, N 1.5M, max row 143k): sampled rows against a float64 host sum,
    linearity, bitwise determinism, device-built CSR == host-built CSR."""
    from mmrec_amd import synth
    nu, ni, eu, ei = synth.shaped_edges("c5", seed=0)
    r, c, v = synth.sym_norm_coo(eu, ei, nu, ni)
    n = nu + ni
    g = ops.CsrGraph.from_coo_device(D(r.astype(np.int32), dev), D(c.astype(np.int32), dev), D(v, dev),
                                     n, n, symmetric=True)
    rp = g.rowptr_host.astype(np.int64)
    assert rp[-1] == 20_000_000 and int(np.diff(rp).max()) > 100_000
    # device CSR == stable host CSR (integer exact)
    rows_sorted = np.all(np.diff(r) >= 0)
    assert rows_sorted
    np.testing.assert_array_equal(g.colidx.cpu().numpy(), c.astype(np.int32))
    gen = torch.Generator(device=dev).manual_seed(0)
    X = torch.rand(n, 64, device=dev, generator=gen) - 0.5
    X2 = torch.rand(n, 64, device=dev, generator=gen) - 0.5
    Y, Y2, Ys, Yr = (torch.empty_like(X) for _ in range(4))
    ops.spmm_raw(g, X, Y=Y)
    ops.spmm_raw(g, X, Y=Yr)
    assert torch.equal(Y, Yr)                                   # deterministic
    ops.spmm_raw(g, X2, Y=Y2)
    ops.spmm_raw(g, X + X2, Y=Ys)
    assert rel_fro(Ys, Y + Y2) < 1e-6                           # linearity
    rng = np.random.default_rng(1)
    heavy = np.argsort(np.diff(rp))[-3:]
    sample = np.concatenate([rng.integers(0, n, 500), heavy])
    Xh = X.cpu().numpy().astype(np.float64)
    Yh = Y.cpu().numpy()
    for row in sample:
        s, e = rp[row], rp[row + 1]
        ref = (v[s:e].astype(np.float64)[:, None] * Xh[c[s:e]]).sum(0)
        np.testing.assert_allclose(Yh[row], ref, rtol=1e-4, atol=1e-6)
    # EXHAUSTIVE: all 1.5 M rows of the layer against a float64-accumulating CSR product on the host (torch CPU), each element
    # within 1e-4 relative of the row's sum |a_k x_k| scale (north_star's tolerance; fp32 summation of <= 143k terms observed
    # ~1e-6), and the oracle's own reference-form product (fp32 torch.sparse.mm on the uncoalesced COO) to the same bound
    A64 = torch.sparse_csr_tensor(torch.from_numpy(rp), torch.from_numpy(c.astype(np.int64)), torch.from_numpy(v.astype(np.float64)),
                                  size=(n, n))
    ref64 = (A64 @ torch.from_numpy(Xh)).numpy()
    Aabs = torch.sparse_csr_tensor(torch.from_numpy(rp), torch.from_numpy(c.astype(np.int64)), torch.from_numpy(np.abs(v).astype(np.float64)),
                                   size=(n, n))
    scale = (Aabs @ torch.from_numpy(np.abs(Xh))).numpy()
    err = np.abs(Yh.astype(np.float64) - ref64)
    worst = float((err / (scale + 1e-30)).max())
    print("spmm c5, all %d rows: max |err| / sum|a x| = %.3e" % (n, worst))
    assert worst < 1e-5, worst                                      # 10 x inside north_star's 1e-4
    assert np.all(err <= 1e-4 * np.abs(ref64) + 1e-5 * scale)       # element-wise relative, cancellation-safe
    ref32 = torch.sparse.mm(orc.sparse_coo(np.stack([r, c]), v, n), X.cpu()).numpy()
    assert float((np.abs(Yh - ref32) / (scale + 1e-30)).max()) < 1e-5


# ---------------------------------------------------------------------------------------- RCCL path
def test_sharded_propagator_rccl_single_rank(ops, dev):
    """The N > 1 code path (HIP SpMM on row blocks + in-place all_gather_into_tensor over RCCL) run at
    world size 1 on the GPU: same bits as the plain full-graph SpMM.  (World size 2 is covered on CPU
    with gloo in tests/test_dist_gloo.py; 8 GPUs are only available to the driver.)"""
    import os
    import socket
    import torch.distributed as dist
    from mmrec_amd import synth
    from mmrec_amd.dist import BipartiteSharding, ShardedPropagator
    single_rank_rccl_group(dev)
    try:
        nu, ni = 3000, 1200
        eu, ei = synth.powerlaw_edges(nu, ni, 40000, seed=2)
        r, c, v = synth.sym_norm_coo(eu, ei, nu, ni)
        n = nu + ni
        full = ops.CsrGraph.from_coo_host(np.stack([r, c]), v, n, n, dev, symmetric=True)
        # nnz-balanced cut, 3 row chunks per rank (chunk-major padded id space: one all-gather per chunk)
        sh = BipartiteSharding.from_coo(r, nu, ni, 1, n_chunks=3)
        make = lambda lr, pc, vals, nr, nc: ops.CsrGraph.from_coo_host(np.stack([lr, pc]), vals, nr, nc, dev)  # noqa: E731
        ub, ib = sh.rank_blocks(r, c, v, 0, make)
        prop = ShardedPropagator(sh, ub, ib, 0, lambda blk, X, Y, **ep: ops.spmm_raw(blk, X, Y=Y, **ep),
                                 force_collectives=True)
        gen = torch.Generator(device=dev).manual_seed(0)
        x0 = torch.rand(n, 64, device=dev, generator=gen) - 0.5
        outs = prop.propagate(sh.pad(x0), 3)
        torch.cuda.synchronize()
        cur = x0
        for layer in range(3):
            y = torch.empty_like(x0)
            ops.spmm_raw(full, cur, Y=y)
            assert torch.equal(sh.unpad_nodes(outs[layer]), y)
            cur = y
        # the autograd layer mean over the sharded rows (forward AND backward all-gathers) == hip_ops.lightgcn_mean, bitwise
        from mmrec_amd.dist import sharded_lightgcn_mean
        w = torch.rand(n, 64, device=dev, generator=gen) - 0.5
        for layers in (1, 2, 3):
            a, b = x0.clone().requires_grad_(), x0.clone().requires_grad_()
            m1 = sharded_lightgcn_mean(prop, a, layers)
            m2 = ops.lightgcn_mean(full, b, layers)
            (m1 * w).sum().backward(), (m2 * w).sum().backward()
            assert torch.equal(m1, m2) and torch.equal(a.grad, b.grad), layers
        assert prop.op.bytes_gathered == 0          # world size 1: nothing is received
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("layout", ["rows", "dslice", "dslice+reorder"])
def test_sharded_freedom_plugin_rccl_single_rank(tmp_path, golden, dev, layout):
    """config `n_gpus`: the sharded FREEDOM plugins on the HIP kernels through a single-rank RCCL group, collectives forced:
    one epoch of Trainer steps and an evaluation == the plain FREEDOM plugin.  `dist_layout: rows` -- row-sharded graphs +
    RCCL all-gather per layer forward and backward, item-sharded feature tables with the projected batch rows exchanged,
    sharded evaluation; `dslice` -- the feature-sliced plugin (whole graphs, column-sliced id tables, all-reduced partial dot
    products, full-width projection gradients summed back to the owners, tables all-gathered per evaluation).  World sizes
    2 / 3 / 4 / 8 run on gloo with the CPU stand-ins (tests/test_dist_gloo.py); the slice kernels themselves are checked bit
    for bit above (test_spmm_feature_slices_equal_the_d64_launch_bitwise).  `dslice+reorder`: the sliced plugin with config
    `reorder: community` (tables, item blocks and graphs in relabelled ids; the device label propagation) against the PLAIN
    plugin without the key."""
    import os
    import socket
    import torch.distributed as dist
    from mmrec_amd.common.trainer import Trainer
    from mmrec_amd.utils.utils import get_model
    from tests._env import setup
    layout, reorder = (layout.split("+") + [None])[:2]
    single_rank_rccl_group(dev)
    try:
        res = {}
        for sharded in (False, True):
            extra = {"dropout": 0.8, "reg_weight": 1e-3, "learning_rate": 0.01, "dist_chunks": 2,
                     "dist_force_collectives": True, "lazy_feature_adam": False, "dist_layout": layout,
                     "hip_pull_batch_rows": sharded}     # the sliced plugin at the batch rows against the plain one over all rows
            if reorder and sharded:
                extra["reorder"] = "community"
            config, train_data, valid_data = setup(tmp_path / ("s%d" % sharded), golden, "FREEDOM", extra, use_gpu=True)
            model = get_model("FREEDOM", sharded=sharded)(config, train_data).to(config["device"])
            assert type(model).__name__ == {(False, layout): "FREEDOM", (True, "rows"): "RowShardedFREEDOM",
                                            (True, "dslice"): "SlicedFREEDOM"}[(sharded, layout)]
            assert (model.relabelling is not None) == bool(reorder and sharded)
            gen = torch.Generator().manual_seed(3)
            keep = torch.multinomial(model.edge_values.detach().cpu(), int(model.edge_values.numel() * 0.2), generator=gen)
            model.set_kept_edges(keep.to(dev))
            trainer = Trainer(config, model)
            loss, _ = trainer._train_epoch(train_data, 0)
            metrics = trainer.evaluate(valid_data)
            params = {n: p.detach().clone() for n, p in model.named_parameters()}
            if sharded:
                params.update(model.gather_feature_tables())
            res[sharded] = (float(loss), metrics, params)
        np.testing.assert_allclose(res[True][0], res[False][0], rtol=1e-5)
        assert res[True][1] == res[False][1]
        for name, ref in res[False][2].items():
            # analytically-zero gradient (the biases): Adam-normalised noise; elsewhere the same noise on single elements whose
            # gradient nearly cancels -- the two runs sum the batch's duplicates by atomics in different launch shapes: one
            # element of 12,800 at 1.1e-5 (3.2e-4 relative) was observed (round 6)
            atol = 1e-4 if name.endswith("trs.bias") else 2e-5
            np.testing.assert_allclose(res[True][2][name].cpu().numpy(), ref.cpu().numpy(), rtol=2e-4, atol=atol, err_msg=name)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("graph_step", [False, True])
def test_freedom_plugin_pulled_item_rows_equals_full_item_item_layer(tmp_path, golden, dev, graph_step):
    """config `hip_pull_batch_rows` (default on): FREEDOM's training step computes the item-item layer at the batch rows only.
    One epoch through the Trainer (eager and replayed as a hipGraph) + an evaluation against the same run with the key off:
    the first step's loss bit for bit (same forward bits), the epoch's loss, parameters and metrics to rounding (the backward's
    push uses fp32 atomics)."""
    from mmrec_amd.common.trainer import Trainer
    from mmrec_amd.utils.utils import get_model
    from tests._env import setup
    res = {}
    for pull in (True, False):
        extra = {"dropout": 0.8, "reg_weight": 1e-3, "learning_rate": 0.01, "hip_pull_batch_rows": pull,
                 "hip_graph_step": graph_step}
        config, train_data, valid_data = setup(tmp_path / ("p%d" % pull), golden, "FREEDOM", extra, use_gpu=True)
        model = get_model("FREEDOM")(config, train_data).to(config["device"])
        assert model.pull_batch_rows == pull
        gen = torch.Generator().manual_seed(3)
        keep = torch.multinomial(model.edge_values.detach().cpu(), int(model.edge_values.numel() * 0.2), generator=gen)
        model.set_kept_edges(keep.to(dev))
        first = float(model.calculate_loss(next(iter(train_data))))
        trainer = Trainer(config, model)
        loss, _ = trainer._train_epoch(train_data, 0)
        metrics = trainer.evaluate(valid_data)
        res[pull] = (first, float(loss), metrics, {n: p.detach().clone() for n, p in model.named_parameters()})
    assert res[True][0] == res[False][0]
    np.testing.assert_allclose(res[True][1], res[False][1], rtol=1e-6)
    assert res[True][2] == res[False][2]
    for name, ref in res[False][3].items():
        atol = 1e-4 if name.endswith("trs.bias") else 2e-6       # analytically-zero gradient: Adam-normalised noise
        np.testing.assert_allclose(res[True][3][name].cpu().numpy(), ref.cpu().numpy(), rtol=2e-4, atol=atol, err_msg=name)


@pytest.mark.parametrize("multi", [True, False])
def test_fused_adam_matches_torch(dev, multi):
    """f3: HipAdam == torch.optim.Adam (same formula) over several steps, with and without weight decay,
    sizes that exercise the float4 body and the scalar tail; zero-gradient rows keep decaying moments.
    multi=True: one launch per <= 24 tensors (30 tensors here -> two launches, one tensor without a
    gradient in some steps so that per-tensor step counts differ); multi=False: one launch per tensor."""
    from mmrec_amd.common.optim import HipAdam
    for wd in (0.0, 1e-2):
        g = torch.Generator().manual_seed(1)
        shapes = [(7, 64), (1, 3), (1000, 37), (64,)] + [(5, 8 + j) for j in range(26)]
        ref = [torch.randn(*s, generator=g).to(dev).requires_grad_() for s in shapes]
        ours = [r.detach().clone().requires_grad_() for r in ref]
        o_ref = torch.optim.Adam(ref, lr=1e-2, weight_decay=wd)
        o_our = HipAdam(ours, lr=1e-2, weight_decay=wd, multi_tensor=multi)
        sched = torch.optim.lr_scheduler.LambdaLR(o_our, lr_lambda=lambda ep: 0.9 ** ep)
        sched_ref = torch.optim.lr_scheduler.LambdaLR(o_ref, lr_lambda=lambda ep: 0.9 ** ep)
        for step in range(5):
            for j, (r, o) in enumerate(zip(ref, ours)):
                if j == 5 and step % 2:          # this tensor skips steps: its own step count lags
                    r.grad = o.grad = None
                    continue
                grad = torch.randn(r.shape, generator=g).to(dev)
                if r.dim() == 2 and r.shape[0] > 4:
                    grad[::2] = 0          # row-sparse gradient, dense update
                r.grad, o.grad = grad.clone(), grad.clone()
            o_ref.step(), o_our.step(), sched.step(), sched_ref.step()
        for r, o in zip(ref, ours):
            np.testing.assert_allclose(o.detach().cpu().numpy(), r.detach().cpu().numpy(), rtol=2e-6, atol=1e-7)
        assert o_our.state[ours[0]]['step'] == 5 and o_our.state[ours[5]]['step'] == 3


def test_lazy_row_adam_equals_dense(dev):
    """f3: row-lazy exact Adam (LazyRowEmbedding + HipAdam) == dense fused Adam BIT FOR BIT: random row batches with
    duplicates, rows untouched for many steps, a row used through two calls of one step, a learning-rate schedule,
    with and without weight decay; state after flush() and the rows seen by every forward are compared."""
    from mmrec_amd.common.lazy_rows import LazyRowEmbedding
    from mmrec_amd.common.optim import HipAdam
    n, F = 300, 132
    for wd in (0.0, 1e-2):
        g = torch.Generator().manual_seed(3)
        w0 = torch.randn(n, F, generator=g)
        other0 = torch.randn(40, 8, generator=g)
        dense = torch.nn.Embedding.from_pretrained(w0.clone(), freeze=False).to(dev)
        lazy = LazyRowEmbedding.from_pretrained(w0.clone(), freeze=False).to(dev)
        od, ol = torch.nn.Parameter(other0.clone().to(dev)), torch.nn.Parameter(other0.clone().to(dev))
        opt_d = HipAdam([dense.weight, od], lr=1e-2, weight_decay=wd)
        opt_l = HipAdam([lazy.weight, ol], lr=1e-2, weight_decay=wd)
        sch_d = torch.optim.lr_scheduler.LambdaLR(opt_d, lr_lambda=lambda ep: 0.8 ** ep)
        sch_l = torch.optim.lr_scheduler.LambdaLR(opt_l, lr_lambda=lambda ep: 0.8 ** ep)
        for step in range(12):
            # step 10: an id list longer than MMREC_ADAM_ROWS_MAX_IDS (the owner workgroups' LDS position list): the
            # pre-summed fallback; every other step: the owners sum their duplicates themselves, in position order
            n_ids = 16500 if step == 10 else 37
            ids = torch.randint(0, 60 if step % 3 else n, (n_ids,), generator=g)       # rows >= 60 are touched rarely
            ids2 = torch.cat([ids[:5], torch.randint(0, n, (6,), generator=g)])        # second use in the same step
            # gradients on a coarse dyadic grid: a row's duplicates are summed in another order by the dense path's
            # atomics, and only exactly representable partial sums make that order-free (the test is about the optimizer)
            coef = (torch.randint(-16, 17, (n_ids, F), generator=g).float() / (16 if n_ids == 37 else 1024)).to(dev)
            coef2 = (torch.randint(-16, 17, (11, F), generator=g).float() / 16).to(dev)
            ids, ids2 = ids.to(dev), ids2.to(dev)
            rows_l, rows_l2 = lazy.rows(ids), lazy.rows(ids2)
            rows_d, rows_d2 = dense.weight[ids], dense.weight[ids2]
            assert torch.equal(rows_l, rows_d) and torch.equal(rows_l2, rows_d2)      # forwards see identical rows
            opt_d.zero_grad(), opt_l.zero_grad()
            ((rows_d * coef).sum() + (rows_d2 * coef2).sum() + (od ** 2).sum()).backward()
            ((rows_l * coef).sum() + (rows_l2 * coef2).sum() + (ol ** 2).sum()).backward()
            assert lazy.weight.grad is None
            opt_d.step(), opt_l.step()
            if step % 4 == 3:
                sch_d.step(), sch_l.step()
        assert not torch.equal(lazy.weight, dense.weight)       # updates of idle rows are still postponed ...
        lazy.flush()                                            # ... until they are asked for
        assert torch.equal(lazy.weight, dense.weight)
        for key in ("exp_avg", "exp_avg_sq"):
            assert torch.equal(opt_l.state[lazy.weight][key], opt_d.state[dense.weight][key])
        assert torch.equal(ol, od) and opt_l.state[lazy.weight]["step"] == 12


def test_lazy_row_adam_long_gaps_equal_dense_bitwise(dev):
    """the catch-up's settled-parameter path (adam.hip: once the largest possible update of the next <= 256 steps is below a
    quarter ulp of p, only the two moment decays are replayed): rows that sit out 7 ... 699 optimizer steps, parameters that
    are exact zeros, powers of two, 1e-6-sized and ordinary, a learning-rate schedule -- every touch and the flushed table
    == dense fused Adam BIT FOR BIT over 700 steps (parameter AND both moments)."""
    from mmrec_amd.common.lazy_rows import LazyRowEmbedding
    from mmrec_amd.common.optim import HipAdam
    n, F, T = 48, 260, 700
    g = torch.Generator().manual_seed(11)
    w0 = torch.randn(n, F, generator=g) * 0.3
    w0[:, ::7] = 0.0
    w0[:, 1::7] = 0.5
    w0[:, 2::7] *= 1e-6
    w0[:, 3::7] = -2.0
    gaps = torch.tensor([1, 7, 60, 150, 333, 699])[torch.arange(n) % 6]
    dense = torch.nn.Embedding.from_pretrained(w0.clone(), freeze=False).to(dev)
    lazy = LazyRowEmbedding.from_pretrained(w0.clone(), freeze=False).to(dev)
    opt_d, opt_l = HipAdam([dense.weight], lr=1e-3), HipAdam([lazy.weight], lr=1e-3)
    sch_d = torch.optim.lr_scheduler.LambdaLR(opt_d, lr_lambda=lambda ep: 0.9 ** ep)
    sch_l = torch.optim.lr_scheduler.LambdaLR(opt_l, lr_lambda=lambda ep: 0.9 ** ep)
    for step in range(T):
        ids = torch.nonzero(step % gaps == 0).flatten()
        coef = (torch.randint(-16, 17, (ids.numel(), F), generator=g).float() / 16).to(dev)
        ids = ids.to(dev)
        rows_l, rows_d = lazy.rows(ids), dense.weight[ids]
        assert torch.equal(rows_l, rows_d), step
        opt_d.zero_grad(), opt_l.zero_grad()
        (rows_d * coef).sum().backward()
        (rows_l * coef).sum().backward()
        opt_d.step(), opt_l.step()
        if step % 200 == 199:
            sch_d.step(), sch_l.step()
    lazy.flush()
    assert torch.equal(lazy.weight, dense.weight)
    for key in ("exp_avg", "exp_avg_sq"):
        assert torch.equal(opt_l.state[lazy.weight][key], opt_d.state[dense.weight][key])


def test_lazy_row_adam_fast_forward_one_call_against_the_exact_replay(dev):
    """ABI 13, opt-in: mmrec_adam_rows_fastforward_f32 against mmrec_adam_rows_catchup_f32 on copies of the same state --
    rows last visited at step s0, brought to step t_now; second moments over 20 decades (eps far above / near / far below
    sqrt(v)), first moments of either sign and exact zeros, parameters that are zero / tiny / ordinary, gaps 1 ... 3000.
    p within 2e-6 of the distance the replay moved it (+ the roundings the replay itself makes: half an ulp per step it
    really moved), m and v within 1e-6 sqrt(gap) relative.  Rows the series does not serve (short gaps; a row that is still
    inside the first 128 optimizer steps) come out BIT-identical: they took the exact replay inside the same launch; a row
    last visited before step 128 is replayed to there and advanced in closed form for the rest."""
    from mmrec_amd import _lib
    import ctypes
    lib = _lib.load()
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    b1, b2, eps, lr, F = 0.9, 0.999, 1e-8, 1e-3, 4104
    T = 6000
    tt = np.arange(T, dtype=np.float64)
    tt[0] = 1
    hist = torch.tensor(np.stack([lr / (1 - b1 ** tt), 1 / np.sqrt(1 - b2 ** tt)], 1), dtype=torch.float32).to(dev)
    gen = torch.Generator().manual_seed(5)
    cases = [(0, 40), (5, 200), (40, 13), (150, 1), (150, 12), (150, 13), (150, 60), (400, 122), (683, 500), (1000, 257),
             (2500, 3000), (5000, 999)]
    n = len(cases)
    p0 = torch.randn(n, F, generator=gen) * 0.1
    p0[:, ::9] = 0.0
    p0[:, 1::9] *= 1e-5
    v0 = 10.0 ** (torch.rand(n, F, generator=gen) * 20 - 20)
    m0 = v0.sqrt() * torch.randn(n, F, generator=gen) * 3
    m0[:, 2::9] = 0.0
    last = torch.tensor([c[0] for c in cases], dtype=torch.int32)
    out = {}
    for fast in (False, True):
        fn = lib.mmrec_adam_rows_fastforward_f32 if fast else lib.mmrec_adam_rows_catchup_f32
        res = []
        for r, (s0, gap) in enumerate(cases):            # one call per row: each row has its own t_now
            p, m, v = p0.clone().to(dev), m0.clone().to(dev), v0.clone().to(dev)
            ls = last.clone().to(dev)
            ids = torch.tensor([r], dtype=torch.int64, device=dev)
            owner = torch.full((n,), 2 ** 31 - 1, dtype=torch.int32, device=dev)
            _lib.check(lib.mmrec_adam_rows_owner(P(ids), 1, P(owner), None), "owner")
            _lib.check(fn(P(p), P(m), P(v), P(ids), P(owner), 1, n, F, P(ls), P(hist), s0 + gap, b1, b2, eps, 0.0, None),
                       "catchup")
            torch.cuda.synchronize()
            assert int(ls[r]) == s0 + gap and int(owner[r]) == 2 ** 31 - 1
            others = torch.arange(n) != r
            assert torch.equal(p.cpu()[others], p0[others])                       # only the listed row moved
            res.append((p[r].cpu().double(), m[r].cpu().double(), v[r].cpu().double()))
        out[fast] = res
    def remainder(s0, gap):                              # adam_fast_row_scalars' R in float64 (tests/test_host_logic.py)
        jp = np.arange(1, min(gap, 256) + 1, dtype=np.float64)
        w = hist[s0 + 1:s0 + 1 + jp.size, 0].cpu().double().numpy() * b1 ** jp
        d = hist[s0 + 1:s0 + 1 + jp.size, 1].cpu().double().numpy() * b2 ** (jp / 2)
        de = d / ((w * d).sum() / w.sum()) - 1
        return (np.abs(w) * np.abs(de) ** 7).sum() / w.sum()

    worst = 0.0
    for r, (s0, gap) in enumerate(cases):
        (pe, me, ve), (pf, mf, vf) = out[False][r], out[True][r]
        s_f = max(s0, min(s0 + gap, 128))                # adam.hip FAST_FROM_STEP: exact replay up to step 128, closed form from there
        R = remainder(s_f, s0 + gap - s_f) if s0 + gap - s_f > 12 else 1.0
        if s0 + gap - s_f <= 12 or R > 2e-7:
            assert torch.equal(pe, pf) and torch.equal(me, mf) and torch.equal(ve, vf), (s0, gap)   # exact replay inside
            continue
        moved = (pe - p0[r].double()).abs()
        ulp = torch.maximum(pe.abs(), p0[r].double().abs()) * 2.0 ** -24
        tol = 2e-6 * moved + ulp * min(gap, 150) * 0.5 + 1e-12
        rel = ((pf - pe).abs() / tol).max().item()
        errs = [((a - b).abs()[a.abs() > 1e-30] / a.abs()[a.abs() > 1e-30]).max().item() if bool((a.abs() > 1e-30).any()) else 0.0
                for a, b in ((me, mf), (ve, vf))]          # (below 1e-30 the replay's denormals and the closed form's zero differ)
        print("s0 %d gap %d R %.1e: p error / tolerance %.3f (max |dp| %.2e of moved %.2e), m %.1e v %.1e%s" %
              (s0, gap, R, rel, (pf - pe).abs().max().item(), moved.max().item(), errs[0], errs[1],
               "  [bit-identical: exact replay]" if torch.equal(pe, pf) else ""))
        if R > 5e-8 and torch.equal(pe, pf):
            continue                                     # (R within rounding of the kernel's 1e-7 line: either path)
        worst = max(worst, rel)
        assert rel <= 1.0, (s0, gap, rel)
        for a, b, err in ((me, mf, errs[0]), (ve, vf, errs[1])):
            assert err <= 1e-6 * gap ** 0.5 + 2e-7, (s0, gap, err)
            assert float((a - b).abs().max()) <= 1e-30 or err > 0
            assert bool((b[a == 0] == 0).all())
        assert not torch.equal(pe, pf)                   # (the closed form really ran)
    print("fast-forward, one call: worst error / tolerance %.3f" % worst)


def test_lazy_row_adam_fast_forward_training_run_close_to_dense(dev):
    """the opt-in fast-forward through LazyRowEmbedding + HipAdam (fast_forward = True) over 900 steps with rows that sit out
    1 ... 699 steps, gradient columns of size 1, 1e-3 and 1e-6 (sqrt(v) from far above eps to below it) and a
    learning-rate schedule, judged against torch.optim.Adam's recurrence in FLOAT64: the fast-forwarded table is as close to
    it as the dense fp32 kernel is (largest error <= 1.5 x the dense kernel's, element by element <= 1e-5 of the distance
    moved + 3 x the dense kernel's largest error; the fp32 kernels round p every step, ~5e-7 after 900 steps of a
    parameter of size 1) -- and the two fp32 tables differ by no more than that from each other.  The default table
    (fast_forward False) in the same run stays bit-identical to the dense kernel."""
    from mmrec_amd.common.lazy_rows import LazyRowEmbedding
    from mmrec_amd.common.optim import HipAdam
    n, F, T, lr0, b1, b2, eps = 56, 260, 900, 1e-3, 0.9, 0.999, 1e-8
    g = torch.Generator().manual_seed(12)
    w0 = torch.randn(n, F, generator=g) * 0.3
    w0[:, ::7] = 0.0
    scale = torch.tensor([1.0, 1e-3, 1e-6])[torch.arange(F) % 3].to(dev)
    gaps = torch.tensor([1, 7, 13, 60, 150, 333, 699])[torch.arange(n) % 7]
    dense = torch.nn.Embedding.from_pretrained(w0.clone(), freeze=False).to(dev)
    exact = LazyRowEmbedding.from_pretrained(w0.clone(), freeze=False).to(dev)
    fast = LazyRowEmbedding.from_pretrained(w0.clone(), freeze=False).to(dev)
    fast.fast_forward = True
    opts = [HipAdam([t.weight], lr=lr0) for t in (dense, exact, fast)]
    schs = [torch.optim.lr_scheduler.LambdaLR(o, lr_lambda=lambda ep: 0.9 ** ep) for o in opts]
    p64 = w0.double().to(dev)
    m64, v64 = torch.zeros_like(p64), torch.zeros_like(p64)
    lr = lr0
    for step in range(T):
        ids = torch.nonzero(step % gaps == 0).flatten()
        coef = (torch.randint(-16, 17, (ids.numel(), F), generator=g).float() / 16).to(dev) * scale
        ids = ids.to(dev)
        rows = [dense.weight[ids], exact.rows(ids), fast.rows(ids)]
        assert torch.equal(rows[0], rows[1]), step
        for o, r in zip(opts, rows):
            o.zero_grad()
            (r * coef).sum().backward()
            o.step()
        g64 = torch.zeros_like(p64).index_add_(0, ids, coef.double())
        m64 = b1 * m64 + (1 - b1) * g64
        v64 = b2 * v64 + (1 - b2) * g64 * g64
        t = step + 1
        p64 = p64 - (lr / (1 - b1 ** t)) * m64 / (v64.sqrt() / (1 - b2 ** t) ** 0.5 + eps)
        if step % 200 == 199:
            for sc in schs:
                sc.step()
            lr = lr0 * 0.9 ** ((step + 1) // 200)
    exact.flush(), fast.flush()
    assert torch.equal(exact.weight, dense.weight)
    e_dense, e_fast = (dense.weight.double() - p64).abs(), (fast.weight.double() - p64).abs()
    moved = (p64 - w0.double().to(dev)).abs()
    print("900 steps: largest error against float64 Adam: dense fp32 kernel %.2e, fast-forwarded table %.2e (largest move %.2e); "
          "fast against dense %.2e" % (e_dense.max().item(), e_fast.max().item(), moved.max().item(),
                                       (fast.weight - dense.weight).abs().max().item()))
    for gp in (1, 7, 13, 60, 150, 333, 699):
        sel = (gaps == gp).to(dev)
        print("  rows touched every %3d steps: dense %.2e  fast %.2e" % (gp, e_dense[sel].max().item(), e_fast[sel].max().item()))
    assert e_fast.max().item() <= 1.5 * e_dense.max().item()
    assert bool((e_fast <= 1e-5 * moved + 3 * e_dense.max()).all())
    assert (fast.weight - dense.weight).abs().max().item() <= 2.5 * e_dense.max().item()
    assert not torch.equal(fast.weight, dense.weight)
    for key, ref in (("exp_avg", m64), ("exp_avg_sq", v64)):
        a, bb = opts[0].state[dense.weight][key].double(), opts[2].state[fast.weight][key].double()
        big = ref.abs() > 1e-30
        ea, eb = ((a - ref).abs() / ref.abs())[big].max().item(), ((bb - ref).abs() / ref.abs())[big].max().item()
        print("  %s: largest relative error against float64: dense %.2e  fast %.2e" % (key, ea, eb))
        assert eb <= max(1.5 * ea, 2e-6), key


def test_lazy_row_adam_under_graph_replay_equals_dense(dev):
    """Round-1 review item 8: the row-lazy Adam inside a REPLAYED hipGraph step.  The step-dependent scalars come from the
    capturable HipAdam's device counters (mmrec_adam_*_dev entry points), the per-step table is reserved per capture.
    Three 'epochs' of a GraphedTrainStep (eager warm-up step, re-capture per epoch, a short last batch run eagerly, a
    learning-rate change between epochs, many more replays than the table's initial 1024 entries) on a model with a lazy
    table == the same run with a dense table under the same capturable optimizer, bit for bit (weights and both moments
    after flush()); the device step counter equals the number of steps taken."""
    from mmrec_amd.common.graph_step import GraphedTrainStep
    from mmrec_amd.common.lazy_rows import LazyRowEmbedding
    from mmrec_amd.common.optim import HipAdam
    n, F, B = 400, 64, 96
    g = torch.Generator().manual_seed(9)
    w0, lin0 = torch.randn(n, F, generator=g), torch.randn(F, 8, generator=g) * 0.1
    # dyadic gradient coefficients: duplicate rows are summed in different orders by the two paths (see the eager test)
    coefs = (torch.randint(-8, 9, (n, F), generator=g).float() / 8).to(dev)
    epochs = [[torch.randint(0, 50 if (e + b) % 3 else n, (1, B if b < 420 else 31), generator=g) for b in range(421)]
              for e in range(3)]                                      # 1263 steps > 1024; last batch of an epoch is short

    class Net(torch.nn.Module):
        def __init__(self, lazy):
            super().__init__()
            cls = LazyRowEmbedding if lazy else torch.nn.Embedding
            self.table = cls.from_pretrained(w0.clone(), freeze=False)
            self.lin = torch.nn.Parameter(lin0.clone())
            self.lazy = lazy

        def calculate_loss(self, batch):
            ids = batch[0]
            rows = self.table.rows(ids) if self.lazy else self.table.weight[ids]
            return (rows * coefs[ids]).sum() + (self.lin ** 2).sum()

    def run(lazy, fast=False):
        net = Net(lazy).to(dev)
        if fast:
            net.table.fast_forward = True
        opt = HipAdam(net.parameters(), lr=1e-2, capturable=True)
        step = GraphedTrainStep(net, opt)
        for e, batches in enumerate(epochs):
            for group in opt.param_groups:
                group["lr"] = 1e-2 * 0.7 ** e
            step.invalidate()
            step.steps_per_capture = len(batches) + 2
            for b in batches:
                step(b.to(dev))
        assert not step.failed and step.graph is not None             # the step really was captured and replayed
        if lazy:
            assert net.table.steps_on_device() == sum(len(b) for b in epochs)
            net.table.flush()
        torch.cuda.synchronize()
        st = opt.state[net.table.weight]
        return net.table.weight.detach().clone(), st["exp_avg"].clone(), st["exp_avg_sq"].clone(), net.lin.detach().clone()
    a, b = run(True), run(False)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    assert not torch.equal(a[0].cpu(), w0)
    # ABI 13: the opt-in fast-forward inside the replayed step (mmrec_adam_rows_fastforward_dev_f32: step count from the device)
    c = run(True, fast=True)
    moved = (b[0] - w0.to(dev)).abs()
    assert not torch.equal(c[0], b[0])
    assert bool(((c[0] - b[0]).abs() <= 1e-5 * moved + 4e-6).all()), ((c[0] - b[0]).abs().max().item(), moved.max().item())
    assert torch.allclose(c[2], b[2], rtol=1e-4, atol=1e-30) and torch.equal(c[3], b[3])


def test_lazy_row_adam_long_id_lists_under_capture(dev):
    """Round-2 advice: a step with more ids than the step kernel's LDS position list holds (n > MMREC_ADAM_ROWS_MAX_IDS =
    16,000, i.e. train_batch_size > 8,000) pre-sums the row gradients into the owner slots on the host side of the call; that
    pre-sum used boolean-mask indexing (`x[mask]` = nonzero() = a host synchronisation: illegal inside a hipGraph capture, a
    stall in eager mode).  It is sync-free now (rows of id -1 add zeros): the step is CAPTURED and replayed, with -1 slots in
    the list, and equals the dense table under the same optimizer (dyadic coefficients: bit for bit)."""
    from mmrec_amd.common.graph_step import GraphedTrainStep
    from mmrec_amd.common.lazy_rows import MAX_IDS, LazyRowEmbedding
    from mmrec_amd.common.optim import HipAdam
    n, F, B = 3000, 64, MAX_IDS + 1500
    g = torch.Generator().manual_seed(12)
    w0 = torch.randn(n, F, generator=g)
    coefs = (torch.randint(-8, 9, (n, F), generator=g).float() / 8).to(dev)
    batches = []
    for b in range(6):
        ids = torch.randint(0, n, (1, B), generator=g)
        ids[0, torch.randperm(B, generator=g)[:200]] = -1              # slots of rows another rank would own
        batches.append(ids)

    class Net(torch.nn.Module):
        def __init__(self, lazy):
            super().__init__()
            cls = LazyRowEmbedding if lazy else torch.nn.Embedding
            self.table = cls.from_pretrained(w0.clone(), freeze=False)
            self.lazy = lazy
            if lazy:
                self.table.allow_missing = True

        def calculate_loss(self, batch):
            ids = batch[0]
            present = (ids >= 0).unsqueeze(1).float()
            safe = ids.clamp_min(0)
            rows = self.table.rows(ids) if self.lazy else self.table.weight[safe] * present
            return (rows * coefs[safe] * present).sum()

    def run(lazy):
        net = Net(lazy).to(dev)
        opt = HipAdam(net.parameters(), lr=1e-2, capturable=True)
        step = GraphedTrainStep(net, opt, steps_per_capture=len(batches) + 2)
        for b in batches:
            step(b.to(dev))
        assert not step.failed and step.graph is not None, "the step with the long id list must be capturable"
        if lazy:
            net.table.flush()
        torch.cuda.synchronize()
        st = opt.state[net.table.weight]
        return net.table.weight.detach().clone(), st["exp_avg"].clone(), st["exp_avg_sq"].clone()
    a, b = run(True), run(False)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


def test_lazy_row_adam_skips_missing_rows(dev):
    """item-sharded feature tables (ShardedFREEDOM): a batch slot whose item another rank owns is id -1 = "no row" -- a
    zero row forward, no gradient, no catch-up / step work; the present rows evolve exactly as under dense Adam."""
    from mmrec_amd.common.lazy_rows import LazyRowEmbedding
    from mmrec_amd.common.optim import HipAdam
    g = torch.Generator().manual_seed(5)
    w0 = torch.randn(50, 64, generator=g)
    dense = torch.nn.Embedding.from_pretrained(w0.clone(), freeze=False).to(dev)
    lazy = LazyRowEmbedding.from_pretrained(w0.clone(), freeze=False).to(dev)
    lazy.allow_missing = True
    opt_d, opt_l = HipAdam([dense.weight], lr=1e-2), HipAdam([lazy.weight], lr=1e-2)
    for step in range(6):
        ids = torch.randint(0, 50, (40,), generator=g)
        ids[torch.rand(40, generator=g) < 0.6] = -1                    # most slots belong to other ranks
        coef = (torch.randint(-16, 17, (40, 64), generator=g).float() / 16).to(dev)
        ids = ids.to(dev)
        present = ids >= 0
        rows_l = lazy.rows(ids)
        rows_d = dense.weight[ids.clamp_min(0)] * present.unsqueeze(1)
        assert torch.equal(rows_l, rows_d) and torch.all(rows_l[~present] == 0)
        opt_d.zero_grad(), opt_l.zero_grad()
        (rows_d * coef).sum().backward(), (rows_l * coef).sum().backward()
        opt_d.step(), opt_l.step()
    lazy.flush()
    assert torch.equal(lazy.weight, dense.weight)


# ---------------------------------------------------------------------------------------- Baby shape, end to end
def test_baby_shape_forward_eval_recall_vs_oracle(ops, dev):
    """north_star's accuracy target at the full Amazon-Baby shape (19,445 x 7,050, 118,706 train
    edges, d = 64): 3-layer propagation + full-sort top-50 + metrics on the GPU against the CPU oracle
    (torch.sparse.mm on the reference-form COO, dense scores, mask, torch.topk, reference metrics):
    embeddings <= 1e-4 relative, Recall/NDCG/Precision/MAP @5/10/20/50 within 1e-4."""
    from mmrec_amd import synth
    nu, ni, eu, ei = synth.shaped_edges("baby", seed=0)
    r, c, v = synth.sym_norm_coo(eu, ei, nu, ni)
    n = nu + ni
    g = ops.CsrGraph.from_coo_host(np.stack([r, c]), v, n, n, dev, symmetric=True)
    gen = torch.Generator().manual_seed(999)
    ue = (torch.rand(nu, 64, generator=gen) * 2 - 1) * (6.0 / (nu + 64)) ** 0.5       # xavier_uniform-like
    ie = (torch.rand(ni, 64, generator=gen) * 2 - 1) * (6.0 / (ni + 64)) ** 0.5
    # held-out ground truth: 1-8 unseen items per user
    rng = np.random.default_rng(5)
    seen = set(zip(eu.tolist(), ei.tolist()))
    gt = []
    for u in range(nu):
        items = [int(x) for x in rng.choice(ni, 12, replace=False) if (u, int(x)) not in seen][:rng.integers(1, 9)]
        gt.append(np.asarray(items, dtype=np.int64))
    # oracle
    adj = orc.sparse_coo(np.stack([r, c]), v, n)
    u_ref, i_ref = orc.lightgcn_forward(adj, ue, ie, 3)
    mask = np.stack([eu, ei])
    _, idx_ref = orc.mask_topk(orc.full_sort_scores(u_ref, i_ref, np.arange(nu)), mask, 50)
    lens = np.array([len(x) for x in gt])
    ref = orc.topk_metrics(orc.hit_matrix(idx_ref.numpy(), np.concatenate(gt), lens), lens)
    # HIP path
    out = ops.lightgcn_mean(g, torch.cat([ue, ie]).to(dev), 3)
    assert rel_fro(out[:nu], u_ref) < 1e-6 and rel_fro(out[nu:], i_ref) < 1e-6
    close(out[:nu], u_ref, rtol=1e-4, atol=1e-7)
    rp, col = ops.mask_to_csr(mask, nu, dev)
    idx = ops.score_topk(out[:nu].contiguous(), out[nu:].contiguous(), 50, rp, col)
    grp, gcol = ops.lists_to_csr(gt, dev)
    per_user = ops.topk_metrics_per_user(idx, grp, gcol, [5, 10, 20, 50]).cpu().numpy()
    for m, name in enumerate(("recall", "ndcg", "precision", "map")):
        for t, k in enumerate((5, 10, 20, 50)):
            got = round(float(per_user[:, m, t].mean()), 4)
            assert abs(got - ref["%s@%d" % (name, k)]) <= 1e-4, (name, k, got, ref["%s@%d" % (name, k)])
    agree = np.mean([set(a) == set(b) for a, b in zip(idx.cpu().numpy(), idx_ref.numpy())])
    assert agree > 0.999      # near-ties at rank 50 may swap (fp32 accumulation order)


def test_item_replicated_propagator_rccl_single_rank(ops, dev):
    """users-sharded / items-replicated layout (all-reduce of item partial sums over RCCL) at world size 1
    on the GPU: equals the plain full-graph SpMM (one contributor => bit for bit)."""
    import os
    import socket
    import torch.distributed as dist
    from mmrec_amd import synth
    from mmrec_amd.dist import ItemReplicatedPropagator
    single_rank_rccl_group(dev)
    try:
        nu, ni = 3000, 1200
        eu, ei = synth.powerlaw_edges(nu, ni, 40000, seed=2)
        r, c, v = synth.sym_norm_coo(eu, ei, nu, ni)
        n = nu + ni
        full = ops.CsrGraph.from_coo_host(np.stack([r, c]), v, n, n, dev, symmetric=True)
        w = v[:eu.shape[0]]
        R = ops.CsrGraph.from_coo_host(np.stack([eu, ei]), w, nu, ni, dev)
        Rt = ops.CsrGraph.from_coo_host(np.stack([ei, eu]), w, ni, nu, dev)
        prop = ItemReplicatedPropagator(R, Rt, lambda blk, X, Y: ops.spmm_raw(blk, X, Y=Y), world_size=1,
                                        force_collectives=True)
        gen = torch.Generator(device=dev).manual_seed(0)
        x0 = torch.rand(n, 64, device=dev, generator=gen) - 0.5
        outs = prop.propagate(x0[:nu].contiguous(), x0[nu:].contiguous(), 3)
        torch.cuda.synchronize()
        cur = x0
        for layer in range(3):
            y = torch.empty_like(x0)
            ops.spmm_raw(full, cur, Y=y)
            cur = y
        close(outs[-1][0], cur[:nu], rtol=1e-5, atol=1e-7)
        close(outs[-1][1], cur[nu:], rtol=1e-5, atol=1e-7)
    finally:
        dist.destroy_process_group()


def test_sharded_training_step_rccl_single_rank(ops, dev):
    """ShardedLightGCNStep (differentiable sharded layers + fused BPR + item-gradient all-reduce) on the
    HIP kernels with a single-rank RCCL group == the plain single-GPU LightGCN step."""
    import os
    import socket
    import torch.distributed as dist
    from mmrec_amd.dist import ItemReplicatedPropagator, ShardedLightGCNStep
    from mmrec_amd import synth
    single_rank_rccl_group(dev)
    created = True
    try:
        nu, ni = 900, 400
        eu, ei = synth.powerlaw_edges(nu, ni, 9000, seed=2)
        r, c, v = synth.sym_norm_coo(eu, ei, nu, ni)
        ne = eu.shape[0]
        R = ops.CsrGraph.from_coo_host(np.stack([r[:ne], c[:ne] - nu]), v[:ne], nu, ni, dev)
        Rt = ops.CsrGraph.from_coo_host(np.stack([c[:ne] - nu, r[:ne]]), v[:ne], ni, nu, dev)
        full = ops.CsrGraph.from_coo_host(np.stack([r, c]), v, nu + ni, nu + ni, dev, symmetric=True)
        g = torch.Generator().manual_seed(0)
        U, I = (torch.randn(nu, 64, generator=g) * 0.1).to(dev), (torch.randn(ni, 64, generator=g) * 0.1).to(dev)
        prop = ItemReplicatedPropagator(R, Rt, lambda blk, X, Y: ops.spmm_raw(blk, X, Y=Y), world_size=1,
                                        force_collectives=True, n_chunks=2)
        st = ShardedLightGCNStep(prop, U, I, 2, lambda a, b, us, p, n: ops.bpr_loss(a, b, us, p, n, reduction="sum"), lr=1e-2)
        u_ref, i_ref = U.clone().requires_grad_(), I.clone().requires_grad_()
        opt = torch.optim.Adam([u_ref, i_ref], lr=1e-2)
        for k in range(3):
            users = torch.randint(0, nu, (512,), generator=g).to(dev)
            pos, neg = torch.randint(0, ni, (512,), generator=g).to(dev), torch.randint(0, ni, (512,), generator=g).to(dev)
            loss = st.step(users, pos, neg, 512)
            opt.zero_grad()
            mean = ops.lightgcn_mean(full, torch.cat([u_ref, i_ref]), 2)
            ref = ops.bpr_loss(mean[:nu].contiguous(), mean[nu:].contiguous(), users, pos, neg)
            ref.backward()
            opt.step()
            close(loss, ref, rtol=1e-5)
        close(st.user_emb, u_ref, rtol=1e-4, atol=1e-6)
        close(st.item_emb, i_ref, rtol=1e-4, atol=1e-6)
    finally:
        if created:
            dist.destroy_process_group()


def test_sharded_projection_and_topk_rccl_single_rank(ops, dev):
    """P3 over item shards (all-gather forward, reduce-scatter + dW all-reduce backward) and P5/P6 over
    query shards on the HIP kernels, single-rank RCCL group: equal the unsharded ops."""
    import os
    import socket
    import torch.distributed as dist
    from mmrec_amd.dist import hip_local_linear, sharded_projection, sharded_score_topk
    single_rank_rccl_group(dev)
    try:
        g = torch.Generator().manual_seed(5)
        X = torch.randn(700, 384, generator=g).to(dev)
        W, b = (torch.randn(64, 384, generator=g) * 0.05).to(dev), torch.randn(64, generator=g).to(dev)
        G = torch.randn(700, 64, generator=g).to(dev)
        Xa, Wa, ba = X.clone().requires_grad_(), W.clone().requires_grad_(), b.clone().requires_grad_()
        Y = sharded_projection(Xa, Wa, ba, hip_local_linear, force_collectives=True)
        (Y * G).sum().backward()
        Xb, Wb, bb = X.clone().requires_grad_(), W.clone().requires_grad_(), b.clone().requires_grad_()
        Yr = ops.linear(Xb, Wb, bb)
        (Yr * G).sum().backward()
        assert torch.equal(Y, Yr) and torch.equal(Xa.grad, Xb.grad) and torch.equal(Wa.grad, Wb.grad)
        assert torch.equal(ba.grad, bb.grad)
        Q, C = torch.randn(300, 64, generator=g).to(dev), torch.randn(2000, 64, generator=g).to(dev)
        a = sharded_score_topk(Q, C, 20, ops.score_topk)
        assert torch.equal(a, ops.score_topk(Q, C, 20))
    finally:
        dist.destroy_process_group()


def test_linear_narrow_inputs_take_the_fp32_forward(ops, dev):
    """Round-5 review 3: the 384-wide text projection (freedom.py:208, bm3.py:104) is launch bound -- its forward goes to the
    fp32-MFMA kernel (one launch) unless LINEAR_SPLIT_MIN_F says otherwise: bitwise the `hip_linear_split: False` result; the
    4096-wide image projection keeps the split-operand kernel."""
    g = torch.Generator().manual_seed(3)
    for F, same in ((384, True), (4096, False)):
        X, W, b = torch.randn(700, F, generator=g).to(dev), (torch.randn(64, F, generator=g) / F ** 0.5).to(dev), torch.randn(64, generator=g).to(dev)
        ops.LINEAR_SPLIT_MIN_F = 1024
        routed = ops.linear(X, W, b)
        try:
            ops.LINEAR_F16X3 = False
            fp32 = ops.linear(X, W, b)
        finally:
            ops.LINEAR_F16X3 = True
        assert torch.equal(routed, fp32) == same, F
        close(routed, (X.double() @ W.double().t() + b.double()).float(), rtol=1e-4, atol=1e-5)


def test_locality_order_on_device_equals_host(ops, dev):
    """`reorder: community` with the sweeps on the device (hip_ops.locality_order(device=...)): the permutation of the host form,
    on a bipartite graph with planted communities and on a structureless one."""
    from mmrec_amd import synth
    rng = np.random.default_rng(0)
    nu, ni, eu, ei = synth.shaped_edges("baby", seed=0)
    r, c, v = synth.sym_norm_coo(eu, ei, nu, ni)
    g = ops.CsrGraph.from_coo_host(np.stack([r, c]), v, nu + ni, nu + ni, dev, symmetric=True)
    idx, _ = g.to_coo_host()
    host = ops.locality_order(g.rowptr_host, idx[1], nu + ni, "community", n_left=nu)
    devp = ops.locality_order(g.rowptr_host, idx[1], nu + ni, "community", n_left=nu, device=dev)
    assert np.array_equal(host, devp) and np.array_equal(np.sort(devp), np.arange(nu + ni))
    n = 3000                                    # a square graph (no sides) with 30 planted groups
    grp = rng.integers(0, 30, n)
    rows = np.repeat(np.arange(n), 8)
    cols = np.array([rng.choice(np.flatnonzero(grp == grp[i]), 8) if rng.random() < 0.9 else rng.integers(0, n, 8) for i in range(n)]).reshape(-1)
    rp = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=n))])
    assert np.array_equal(ops.locality_order(rp, cols, n, "community"), ops.locality_order(rp, cols, n, "community", device=dev))
