"""CPU, world_size 2 / 3, gloo: the row-sharding + chunked all-gather schedule of mmrec_amd.dist reproduces the
single-process propagation bit for bit, forward and backward, and the sharded FREEDOM plugin (config `n_gpus`)
reproduces the single-process plugin.  The local SpMM is a scipy checker here (the product passes the HIP kernel);
these tests are about partitioning and exchange, not about the kernel."""
import os
import socket

import numpy as np
import pytest
import scipy.sparse as sp
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mmrec_amd import synth
from mmrec_amd.dist import BipartiteSharding, ShardedPropagator

NU, NI, NE, L = 37, 23, 260, 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _scipy_csr(local_rows, cols, vals, n_rows, n_cols):
    return sp.csr_matrix((vals, (local_rows, cols)), shape=(n_rows, n_cols), dtype=np.float32)


def _problem(world, n_chunks=1, balanced=False):
    eu, ei = synth.powerlaw_edges(NU, NI, NE, seed=5)
    r, c, v = synth.sym_norm_coo(eu, ei, NU, NI)
    sh = BipartiteSharding.from_coo(r, NU, NI, world, n_chunks) if balanced else BipartiteSharding(NU, NI, world, n_chunks=n_chunks)
    rp, cp = sh.padded_coo(r, c)
    A = sp.csr_matrix((v, (rp, cp)), shape=(sh.N_pad, sh.N_pad), dtype=np.float32)
    g = torch.Generator().manual_seed(1)
    x0 = sh.pad_embeddings(torch.randn(NU, 64, generator=g), torch.randn(NI, 64, generator=g))
    return sh, A, x0, (r, c, v)


def _local_spmm(block, X, Y, Z=None, acc_in=None, acc_out=None, alpha=1.0, beta=1.0, acc_scale=1.0):
    """the epilogue contract of hip_ops.spmm_raw on a scipy block: Y = alpha A X + beta Z; acc_out = s (acc_in + Y)"""
    y = torch.from_numpy(block @ X.numpy()) * np.float32(alpha)
    if Z is not None:
        y = y + np.float32(beta) * Z
    if Y is not None:
        Y.copy_(y)
    if acc_out is not None:
        acc_out.copy_(np.float32(acc_scale) * (acc_in + y))


def _worker(rank, world, port, out, n_chunks, balanced):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sh, A, x0, _ = _problem(world, n_chunks, balanced)
    ent = sh.entries(rank)
    blocks = [A[lo:hi] for _, _, lo, hi, _, _ in ent]          # rows of the full padded matrix: same per-row order
    nc = sh.n_chunks
    prop = ShardedPropagator(sh, blocks[:nc], blocks[nc:], rank, _local_spmm)
    outs = prop.propagate(x0, L)
    if rank == 0:
        torch.save([o.clone() for o in outs], out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_chunks,balanced", [(2, 1, False), (2, 2, True), (3, 3, True), (8, 2, True)])
def test_sharded_equals_single(tmp_path, world, n_chunks, balanced):
    """equal-row and nnz-balanced blocks, 1..3 chunks per rank (chunk-major padded space, one all-gather per chunk)"""
    out = str(tmp_path / "outs.pt")
    mp.spawn(_worker, args=(world, _free_port(), out, n_chunks, balanced), nprocs=world, join=True)
    got = torch.load(out)
    sh, A, x0, _ = _problem(world, n_chunks, balanced)
    cur = x0.numpy()
    for layer in range(L):
        cur = A @ cur
        assert np.array_equal(got[layer].numpy(), cur), "layer %d differs" % layer


def test_rank_blocks_partition_the_adjacency_nnz_balanced():
    """BipartiteSharding.from_coo: contiguous user / item blocks with ~equal nonzeros per rank, and
    rank_blocks() = exactly the rank's rows of the padded matrix (every nonzero in exactly one block)."""
    eu, ei = synth.powerlaw_edges(400, 150, 5000, seed=2, zipf=1.1)      # strongly skewed items
    r, c, v = synth.sym_norm_coo(eu, ei, 400, 150)
    for world, n_chunks in ((2, 1), (3, 2), (8, 4)):
        sh = BipartiteSharding.from_coo(r, 400, 150, world, n_chunks)
        rp, cp = sh.padded_coo(r, c)
        assert len(np.unique(sh.pos)) == 550 and sh.pos.max() < sh.N_pad
        A = sp.csr_matrix((v, (rp, cp)), shape=(sh.N_pad, sh.N_pad), dtype=np.float32)
        total = 0
        for rank in range(world):
            ub, ib = sh.rank_blocks(r, c, v, rank, _scipy_csr)
            for blk, (_, _, lo, hi, rlo, rhi) in zip(ub + ib, sh.entries(rank)):
                assert (blk != A[lo:hi]).nnz == 0 and rlo <= lo and hi <= rhi
                total += blk.nnz
        assert total == A.nnz
        per = sh.nnz_per_rank(r)
        eq = BipartiteSharding(400, 150, world).nnz_per_rank(r)
        assert per.sum() == r.shape[0] and per.max() / per.mean() <= eq.max() / eq.mean() + 1e-9
        assert per.max() / per.mean() < 1.25, per


# ---- the autograd layer mean over sharded rows == the single-process one, forward and backward ------------------
def _worker_mean(rank, world, port, out):
    from mmrec_amd.dist import sharded_lightgcn_mean
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sh, A, _, _ = _problem(world, 2, True)
    ent = sh.entries(rank)
    blocks = [A[lo:hi] for _, _, lo, hi, _, _ in ent]
    prop = ShardedPropagator(sh, blocks[:2], blocks[2:], rank, _local_spmm)
    g = torch.Generator().manual_seed(3)
    E0 = torch.randn(NU + NI, 64, generator=g).requires_grad_()
    Wt = torch.randn(NU + NI, 64, generator=g)
    res = {}
    for layers in (1, 2, 3):
        E0.grad = None
        mean = sharded_lightgcn_mean(prop, E0, layers)
        (mean * Wt).sum().backward()          # the same (replicated) loss on every rank
        res[layers] = (mean.detach().clone(), E0.grad.clone())
    if rank == world - 1:
        torch.save(res, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_lightgcn_mean_forward_backward(tmp_path, world):
    out = str(tmp_path / "mean.pt")
    mp.spawn(_worker_mean, args=(world, _free_port(), out), nprocs=world, join=True)
    res = torch.load(out)
    eu, ei = synth.powerlaw_edges(NU, NI, NE, seed=5)
    r, c, v = synth.sym_norm_coo(eu, ei, NU, NI)
    A = torch.sparse_coo_tensor(torch.as_tensor(np.stack([r, c])), torch.as_tensor(v), (NU + NI, NU + NI)).coalesce()
    g = torch.Generator().manual_seed(3)
    E0 = torch.randn(NU + NI, 64, generator=g).requires_grad_()
    Wt = torch.randn(NU + NI, 64, generator=g)
    for layers in (1, 2, 3):
        E0.grad = None
        cur, acc = E0, E0
        for _ in range(layers):
            cur = torch.sparse.mm(A, cur)
            acc = acc + cur
        mean = acc / (layers + 1)
        (mean * Wt).sum().backward()
        np.testing.assert_allclose(res[layers][0].numpy(), mean.detach().numpy(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(res[layers][1].numpy(), E0.grad.numpy(), rtol=1e-5, atol=1e-6)


# ---- users sharded / items replicated layout (all-reduce of item partial sums) -----------------
def _worker_ir(rank, world, port, out):
    from mmrec_amd.dist import ItemReplicatedPropagator
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    eu, ei = synth.powerlaw_edges(NU, NI, NE, seed=5)
    r, c, v = synth.sym_norm_coo(eu, ei, NU, NI)
    R = sp.csr_matrix((v[:eu.shape[0]], (r[:eu.shape[0]], c[:eu.shape[0]] - NU)), shape=(NU, NI), dtype=np.float32)
    ub = -(-NU // world)
    u0, u1 = rank * ub, min((rank + 1) * ub, NU)
    Rr = R[u0:u1]
    prop = ItemReplicatedPropagator(Rr, Rr.T.tocsr(), _local_spmm, world_size=world, n_chunks=3)
    g = torch.Generator().manual_seed(1)
    U, I = torch.randn(NU, 64, generator=g), torch.randn(NI, 64, generator=g)
    outs = prop.propagate(U[u0:u1].contiguous(), I.clone(), L)
    gathered = [torch.zeros(ub, 64) for _ in range(world)]
    pad = torch.zeros(ub, 64)
    pad[:u1 - u0] = outs[-1][0]
    dist.all_gather(gathered, pad)
    if rank == 0:
        torch.save((torch.cat(gathered)[:NU], outs[-1][1].clone()), out)
    dist.barrier()
    dist.destroy_process_group()


def test_item_replicated_layout_matches_single(tmp_path):
    out = str(tmp_path / "ir.pt")
    mp.spawn(_worker_ir, args=(2, _free_port(), out), nprocs=2, join=True)
    u_got, i_got = torch.load(out)
    eu, ei = synth.powerlaw_edges(NU, NI, NE, seed=5)
    r, c, v = synth.sym_norm_coo(eu, ei, NU, NI)
    A = sp.csr_matrix((v, (r, c)), shape=(NU + NI, NU + NI), dtype=np.float32)
    g = torch.Generator().manual_seed(1)
    x = torch.cat([torch.randn(NU, 64, generator=g), torch.randn(NI, 64, generator=g)]).numpy()
    for _ in range(L):
        x = A @ x
    np.testing.assert_allclose(u_got.numpy(), x[:NU], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(i_got.numpy(), x[NU:], rtol=1e-5, atol=1e-6)


# ---- training through the sharded layout: gradients equal the single-process ones ---------------
def _worker_grad(rank, world, port, out):
    from mmrec_amd.dist import ItemReplicatedPropagator, item_replicated_layer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    eu, ei = synth.powerlaw_edges(NU, NI, NE, seed=5)
    r, c, v = synth.sym_norm_coo(eu, ei, NU, NI)
    R = sp.csr_matrix((v[:eu.shape[0]], (r[:eu.shape[0]], c[:eu.shape[0]] - NU)), shape=(NU, NI), dtype=np.float32)
    ub = -(-NU // world)
    u0, u1 = rank * ub, min((rank + 1) * ub, NU)
    Rr = R[u0:u1]
    prop = ItemReplicatedPropagator(Rr, Rr.T.tocsr(), _local_spmm, world_size=world, n_chunks=2)
    g = torch.Generator().manual_seed(1)
    U, I = torch.randn(NU, 64, generator=g), torch.randn(NI, 64, generator=g)
    wu, wi = torch.randn(NU, 64, generator=g), torch.randn(NI, 64, generator=g)
    u = U[u0:u1].clone().requires_grad_()
    it = I.clone().requires_grad_()                       # replicated leaf
    cu, ci = u, it
    acc_u, acc_i = u, it
    for _ in range(L):                                    # LightGCN-style layer sum
        cu, ci = item_replicated_layer(prop, cu, ci)
        acc_u, acc_i = acc_u + cu, acc_i + ci
    # every rank's loss: its own users, and its share (1 / world) of the replicated item term
    loss = (acc_u * wu[u0:u1]).sum() + (acc_i * wi).sum() / world
    loss.backward()
    gi = it.grad.clone()
    dist.all_reduce(gi)                                   # replicated leaf: sum of the local contributions
    gu = [torch.zeros(ub, 64) for _ in range(world)]
    pad = torch.zeros(ub, 64)
    pad[:u1 - u0] = u.grad
    dist.all_gather(gu, pad)
    if rank == 0:
        torch.save((torch.cat(gu)[:NU], gi), out)
    dist.barrier()
    dist.destroy_process_group()


def test_item_replicated_layer_gradients_match_single(tmp_path):
    out = str(tmp_path / "grad.pt")
    mp.spawn(_worker_grad, args=(2, _free_port(), out), nprocs=2, join=True)
    gu_got, gi_got = torch.load(out)
    eu, ei = synth.powerlaw_edges(NU, NI, NE, seed=5)
    r, c, v = synth.sym_norm_coo(eu, ei, NU, NI)
    A = torch.sparse_coo_tensor(torch.as_tensor(np.stack([r, c])), torch.as_tensor(v), (NU + NI, NU + NI)).coalesce()
    g = torch.Generator().manual_seed(1)
    U, I = torch.randn(NU, 64, generator=g), torch.randn(NI, 64, generator=g)
    wu, wi = torch.randn(NU, 64, generator=g), torch.randn(NI, 64, generator=g)
    x = torch.cat([U, I]).requires_grad_()
    cur, acc = x, x
    for _ in range(L):
        cur = torch.sparse.mm(A, cur)
        acc = acc + cur
    (acc * torch.cat([wu, wi])).sum().backward()
    np.testing.assert_allclose(gu_got.numpy(), x.grad[:NU].numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(gi_got.numpy(), x.grad[NU:].numpy(), rtol=1e-4, atol=1e-5)


# ---- a whole sharded BPR training step == the single-process one ---------------------------------
def _bpr_sum_cpu(U, I, users, pos, neg):
    x = (U[users] * I[pos]).sum(1) - (U[users] * I[neg]).sum(1)
    return -torch.nn.functional.logsigmoid(x).sum()


def _train_problem():
    eu, ei = synth.powerlaw_edges(NU, NI, NE, seed=5)
    r, c, v = synth.sym_norm_coo(eu, ei, NU, NI)
    g = torch.Generator().manual_seed(4)
    U, I = torch.randn(NU, 64, generator=g) * 0.1, torch.randn(NI, 64, generator=g) * 0.1
    B = 192
    batches = [(torch.randint(0, NU, (B,), generator=g), torch.randint(0, NI, (B,), generator=g),
                torch.randint(0, NI, (B,), generator=g)) for _ in range(3)]
    return eu, r, c, v, U, I, batches


def _worker_train(rank, world, port, out):
    from mmrec_amd.dist import ItemReplicatedPropagator, ShardedLightGCNStep
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    eu, r, c, v, U, I, batches = _train_problem()
    R = sp.csr_matrix((v[:eu.shape[0]], (r[:eu.shape[0]], c[:eu.shape[0]] - NU)), shape=(NU, NI), dtype=np.float32)
    ub = -(-NU // world)
    u0, u1 = rank * ub, min((rank + 1) * ub, NU)
    Rr = R[u0:u1]
    prop = ItemReplicatedPropagator(Rr, Rr.T.tocsr(), _local_spmm, world_size=world, n_chunks=1)
    st = ShardedLightGCNStep(prop, U[u0:u1], I, L, _bpr_sum_cpu, lr=1e-2)
    losses = []
    for users, pos, neg in batches:
        mine = (users >= u0) & (users < u1)
        losses.append(float(st.step(users[mine] - u0, pos[mine], neg[mine], users.numel())))
    gu = [torch.zeros(ub, 64) for _ in range(world)]
    pad = torch.zeros(ub, 64)
    pad[:u1 - u0] = st.user_emb.detach()
    dist.all_gather(gu, pad)
    if rank == 0:
        torch.save((losses, torch.cat(gu)[:NU], st.item_emb.detach().clone()), out)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_bpr_training_step_matches_single(tmp_path):
    out = str(tmp_path / "train.pt")
    mp.spawn(_worker_train, args=(2, _free_port(), out), nprocs=2, join=True)
    losses, U_got, I_got = torch.load(out)
    eu, r, c, v, U, I, batches = _train_problem()
    A = torch.sparse_coo_tensor(torch.as_tensor(np.stack([r, c])), torch.as_tensor(v), (NU + NI, NU + NI)).coalesce()
    u, it = U.clone().requires_grad_(), I.clone().requires_grad_()
    opt = torch.optim.Adam([u, it], lr=1e-2)
    for k, (users, pos, neg) in enumerate(batches):
        opt.zero_grad()
        x = torch.cat([u, it])
        cur, acc = x, x
        for _ in range(L):
            cur = torch.sparse.mm(A, cur)
            acc = acc + cur
        acc = acc / (L + 1)
        loss = _bpr_sum_cpu(acc[:NU], acc[NU:], users, pos, neg) / users.numel()
        loss.backward()
        opt.step()
        np.testing.assert_allclose(losses[k], loss.item(), rtol=1e-5)
    np.testing.assert_allclose(U_got.numpy(), u.detach().numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(I_got.numpy(), it.detach().numpy(), rtol=1e-4, atol=1e-6)


# ---- P3 over item shards: projection forward + gradients == single process ------------------------
def _torch_local_linear(X, W, b):
    Xd, Wd = X.detach().requires_grad_(), W.detach().requires_grad_()
    bd = b.detach().requires_grad_()
    with torch.enable_grad():
        Y = torch.nn.functional.linear(Xd, Wd, bd)
    return Y.detach(), (lambda dY: torch.autograd.grad(Y, [Xd, Wd, bd], dY))


def _worker_proj(rank, world, port, out):
    from mmrec_amd.dist import sharded_projection
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(9)
    ni, F = 210, 48
    ib = -(-ni // world)
    X = torch.randn(ni, F, generator=g)
    W, b = torch.randn(64, F, generator=g) * 0.1, torch.randn(64, generator=g)
    G = torch.randn(ib * world, 64, generator=g)
    G[ni:] = 0                                            # padded rows carry no gradient
    Xl = torch.zeros(ib, F)
    rows = X[rank * ib:min((rank + 1) * ib, ni)]
    Xl[:rows.shape[0]] = rows
    Xl.requires_grad_()
    Wp, bp = W.clone().requires_grad_(), b.clone().requires_grad_()
    Y = sharded_projection(Xl, Wp, bp, _torch_local_linear)
    # every rank consumes the whole replicated table in its own loss term; the terms add up to <Y, G>
    (Y * G).sum().div(world).backward()
    gx = [torch.zeros(ib, F) for _ in range(world)]
    dist.all_gather(gx, Xl.grad)
    if rank == 0:
        torch.save((Y.detach()[:ni].clone(), torch.cat(gx)[:ni], Wp.grad.clone(), bp.grad.clone()), out)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_projection_matches_single(tmp_path):
    out = str(tmp_path / "proj.pt")
    mp.spawn(_worker_proj, args=(2, _free_port(), out), nprocs=2, join=True)
    Y, gX, gW, gb = torch.load(out)
    g = torch.Generator().manual_seed(9)
    ni, F = 210, 48
    X = torch.randn(ni, F, generator=g).requires_grad_()
    W, b = (torch.randn(64, F, generator=g) * 0.1).requires_grad_(), torch.randn(64, generator=g).requires_grad_()
    G = torch.randn(2 * 105, 64, generator=g)[:ni]
    ref = torch.nn.functional.linear(X, W, b)
    (ref * G).sum().backward()
    np.testing.assert_allclose(Y.numpy(), ref.detach().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(gX.numpy(), X.grad.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(gW.numpy(), W.grad.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(gb.numpy(), b.grad.numpy(), rtol=1e-4, atol=1e-5)


def test_bench_multi_gpu_blocks_partition_the_graph(monkeypatch):
    """bench.py's N > 1 block construction (never exercised with world_size > 1 on a GPU by the build: 8-GPU nodes are
    the driver's): for both layouts the ranks' blocks, applied rank by rank on the CPU stand-in of the SpMM, reproduce
    one layer of the full normalised adjacency -- users-sharded / items-replicated (partial item sums added up as the
    all-reduce would) and row-sharded over the padded id space (blocks concatenated as the all-gather would)."""
    import sys
    import bench
    from mmrec_amd import hip_ops, synth
    from tests import _cpu_ops
    for name in ("CsrGraph", "spmm_raw"):
        monkeypatch.setattr(hip_ops, name, getattr(_cpu_ops, name))
    nu, ni = 57, 23
    eu, ei = synth.powerlaw_edges(nu, ni, 400, seed=1)
    order = np.lexsort((ei, eu))                       # bench relies on edges sorted by user
    eu, ei = eu[order], ei[order]
    monkeypatch.setattr(synth, "shaped_edges", lambda *a, **k: (nu, ni, eu, ei))
    r, c, v = synth.sym_norm_coo(eu, ei, nu, ni)
    n = nu + ni
    full = _cpu_ops.CsrGraph.from_coo_host(np.stack([r, c]), v, n, n, "cpu", symmetric=True)
    X = torch.rand(n, 64, generator=torch.Generator().manual_seed(0)) - 0.5
    ref = full.matmul(X)
    world = 3
    # users sharded, items replicated
    items_sum, user_rows = torch.zeros(ni, 64), []
    for rank in range(world):
        sh, _, r_blk, rt_blk, *_ = bench.build_c5("cpu", rank, world, "allreduce", True)
        ub = -(-nu // world)
        u0, u1 = rank * ub, min((rank + 1) * ub, nu)
        user_rows.append(r_blk.matmul(X[nu:]))                       # U_r' = R_r I
        items_sum += rt_blk.matmul(X[u0:u1])                         # partial of I' = sum_r R_r^T U_r
    np.testing.assert_allclose(torch.cat(user_rows).numpy(), ref[:nu].numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(items_sum.numpy(), ref[nu:].numpy(), rtol=1e-5, atol=1e-6)
    # rows sharded (nnz-balanced cut, two chunks per rank) over the chunk-major padded id space
    Xp = None
    out = None
    for rank in range(world):
        sh, _, ublocks, iblocks, *_ = bench.build_c5("cpu", rank, world, "allgather", True, 2)
        if Xp is None:
            Xp, out = sh.pad_embeddings(X[:nu], X[nu:]), torch.zeros(sh.N_pad, 64)
        for blk, (_, _, lo, hi, _, _) in zip(ublocks + iblocks, sh.entries(rank)):
            out[lo:hi] = blk.matmul(Xp)
    uo, io = sh.unpad(out)
    np.testing.assert_allclose(uo.numpy(), ref[:nu].numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(io.numpy(), ref[nu:].numpy(), rtol=1e-5, atol=1e-6)


# ---- config `n_gpus`: the sharded FREEDOM plugin through the Trainer == the single-process plugin -------------------
def _freedom_run(root, golden, world, layout="rows", reorder=None):
    """two epochs of Trainer on the golden tiny dataset (edge dropout 0.8, both modalities) -> per-epoch losses, the
    parameters, the feature tables and the validation metrics"""
    from mmrec_amd.common.trainer import Trainer
    from mmrec_amd.utils.utils import get_model
    from tests._env import setup
    extra = {"dropout": 0.8, "reg_weight": 1e-3, "learning_rate": 0.01, "n_gpus": world, "dist_chunks": 2,
             "dist_layout": layout,
             "hip_pull_batch_rows": world > 1}     # the sharded runs read the tables at the batch rows (forced: 'auto' is off at
                                                   # this size), the single-process reference run launches over all rows
    if reorder:
        extra["reorder"] = reorder
    config, train_data, valid_data = setup(root, golden, "FREEDOM", extra, use_gpu=False)
    model = get_model("FREEDOM", sharded=world > 1)(config, train_data)
    assert (model.relabelling is not None) == bool(reorder)
    trainer = Trainer(config, model)
    losses = []
    for epoch in range(2):
        model.pre_epoch_processing()
        loss, _ = trainer._train_epoch(train_data, epoch)
        losses.append(float(loss))
    metrics = trainer.evaluate(valid_data)
    params = {n: p.detach().clone() for n, p in model.named_parameters()}
    if world > 1:
        params.update(model.gather_feature_tables())
        if hasattr(model, "nnz_per_rank"):
            params["_nnz_per_rank"] = torch.as_tensor(model.nnz_per_rank)
        params["_class"] = type(model).__name__
    return losses, params, metrics


def _worker_freedom(rank, world, port, root, out, layout="rows", reorder=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(max(1, (os.cpu_count() or 8) // world))     # `world` processes share the host's cores
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests import _cpu_ops
    _cpu_ops.install()
    golden = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tiny.npz")))
    res = _freedom_run(os.path.join(root, "rank%d" % rank), golden, world, layout, reorder)
    torch.save(res, out + ".%d" % rank)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_freedom_plugin_matches_single_process(tmp_path, golden, cpu_ops, world):
    """BASELINE config 5's model over `n_gpus` processes (row-sharded graphs + all-gather per layer, item-sharded feature
    tables with the batch's projected rows exchanged, replicated id tables): same batches, same injected draws (the
    per-epoch multinomial is rank 0's), two epochs of optimizer steps -> the single-process plugin's losses, parameters
    (feature tables re-assembled from the shards) and metrics; every rank ends with the same replicated parameters."""
    out = str(tmp_path / "freedom.pt")
    mp.spawn(_worker_freedom, args=(world, _free_port(), str(tmp_path), out), nprocs=world, join=True)
    got = [torch.load(out + ".%d" % r, weights_only=False) for r in range(world)]
    losses, params, metrics = _freedom_run(str(tmp_path / "single"), golden, 1)
    for r in range(world):
        np.testing.assert_allclose(got[r][0], losses, rtol=1e-5)
        assert got[r][2] == metrics and got[r][1].pop("_class") == "RowShardedFREEDOM"
        for name, ref in params.items():
            # the projection biases cancel in <u, p> - <u, n>: their gradient is rounding noise of the summed terms, which
            # Adam normalises into +-lr-sized steps whose signs depend on the summation order (single process too)
            atol = 1e-4 if name.endswith("trs.bias") else 1e-6
            np.testing.assert_allclose(got[r][1][name].numpy(), ref.numpy(), rtol=2e-4, atol=atol, err_msg=name)
    for name in ("user_embedding.weight", "item_id_embedding.weight", "image_trs.weight"):
        for r in range(1, world):
            assert torch.equal(got[r][1][name], got[0][1][name]), name        # replicas stay bit-identical
    nnz = got[0][1]["_nnz_per_rank"].numpy()
    assert nnz.max() / nnz.mean() < 1.3


@pytest.mark.parametrize("world", [2, 4, 8])
def test_feature_sliced_freedom_plugin_matches_single_process(tmp_path, golden, cpu_ops, world):
    """config `dist_layout: dslice` (the default at 2 / 4 / 8 ranks): every rank holds the whole graphs and 64 / world COLUMNS
    of the id tables, propagates them with no collective at all, all-reduces the [3, 2, B] partial dot products of the BPR
    terms, exchanges the owner-computed projections of the batch rows, all-gathers the final tables once per evaluation.
    Two epochs through the Trainer (edge dropout 0.8, both modalities, same batches and draws) -> the single-process plugin's
    losses, parameters (id tables re-assembled from the ranks' columns, feature tables from their rows) and metrics."""
    out = str(tmp_path / "freedom.pt")
    mp.spawn(_worker_freedom, args=(world, _free_port(), str(tmp_path), out, "dslice"), nprocs=world, join=True)
    got = [torch.load(out + ".%d" % r, weights_only=False) for r in range(world)]
    losses, params, metrics = _freedom_run(str(tmp_path / "single"), golden, 1)
    for r in range(world):
        np.testing.assert_allclose(got[r][0], losses, rtol=1e-5)
        assert got[r][2] == metrics and got[r][1].pop("_class") == "SlicedFREEDOM"
        assert got[r][1]["user_embedding.weight"].shape == params["user_embedding.weight"].shape
        for name, ref in params.items():
            atol = 1e-4 if name.endswith("trs.bias") else 1e-6
            np.testing.assert_allclose(got[r][1][name].numpy(), ref.numpy(), rtol=2e-4, atol=atol, err_msg=name)
    for name in ("user_embedding.weight", "item_id_embedding.weight", "image_trs.weight", "text_trs.weight"):
        for r in range(1, world):
            assert torch.equal(got[r][1][name], got[0][1][name]), name        # replicas / re-assembled tables agree bit for bit


@pytest.mark.parametrize("world,how", [(2, "degree"), (4, "community")])
def test_feature_sliced_freedom_plugin_with_relabelled_ids_matches_single_process(tmp_path, golden, cpu_ops, world, how):
    """config `reorder` on the feature-sliced layout (round-5 review, missing 3): every rank derives the same relabelling from
    the full graph; the column slices' rows, the item blocks of the feature tables and both graphs live in the relabelled ids;
    batches, evaluation masks and ranked lists are translated at the boundary.  Two epochs through the Trainer -> the PLAIN
    single-process plugin's losses, metrics and parameters (gather_tables() hands every table back in the dataset's order),
    and the same numbers as the sliced run WITHOUT the key to the same tolerances."""
    out = str(tmp_path / "freedom.pt")
    mp.spawn(_worker_freedom, args=(world, _free_port(), str(tmp_path), out, "dslice", how), nprocs=world, join=True)
    got = [torch.load(out + ".%d" % r, weights_only=False) for r in range(world)]
    losses, params, metrics = _freedom_run(str(tmp_path / "single"), golden, 1)
    for r in range(world):
        np.testing.assert_allclose(got[r][0], losses, rtol=1e-5)
        assert got[r][2] == metrics and got[r][1].pop("_class") == "SlicedFREEDOM"
        for name, ref in params.items():
            atol = 1e-4 if name.endswith("trs.bias") else 1e-6
            np.testing.assert_allclose(got[r][1][name].numpy(), ref.numpy(), rtol=2e-4, atol=atol, err_msg=name)
    for name in ("user_embedding.weight", "item_id_embedding.weight", "image_embedding.weight", "text_trs.weight"):
        for r in range(1, world):
            assert torch.equal(got[r][1][name], got[0][1][name]), name


def _worker_sliced_propagation(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests import _cpu_ops
    _cpu_ops.install()
    torch.set_num_threads(1)          # torch's CPU sparse product is only run-to-run bitwise with a fixed thread partition
    from mmrec_amd import hip_ops, synth
    nu, ni, eu, ei = 900, 400, *synth.powerlaw_edges(900, 400, 9000, seed=3)
    r, c, v = synth.sym_norm_coo(eu, ei, nu, ni)
    n = nu + ni
    g = hip_ops.CsrGraph.from_coo_host(np.stack([r, c]), v, n, n, "cpu", symmetric=True)
    gen = torch.Generator().manual_seed(0)                       # the same tables and upstream gradient on every rank
    E0, G = torch.randn(n, 64, generator=gen) * 0.1, torch.randn(n, 64, generator=gen)
    w = 64 // world
    Es = E0[:, rank * w:(rank + 1) * w].contiguous().requires_grad_()
    o = hip_ops.lightgcn_mean(g, Es, 3)                          # this rank's columns: no collective
    o.backward(G[:, rank * w:(rank + 1) * w].contiguous())
    parts = [torch.empty(2, n, w) for _ in range(world)]
    dist.all_gather(parts, torch.stack((o.detach(), Es.grad)))
    if rank == 0:
        torch.save(torch.cat(parts, dim=2), out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_feature_sliced_propagation_equals_one_process_bitwise(tmp_path, cpu_ops, world):
    """the propagation of the feature-sliced layout: world processes x 64 / world columns, no exchange -> forward AND backward
    equal the one-process result BIT FOR BIT (columns are independent; the device twin is tests/test_hip_parity.py
    test_spmm_feature_slices_equal_the_d64_launch_bitwise)."""
    from mmrec_amd import hip_ops, synth
    out = str(tmp_path / "sliced.pt")
    mp.spawn(_worker_sliced_propagation, args=(world, _free_port(), out), nprocs=world, join=True)
    got = torch.load(out)
    nu, ni, eu, ei = 900, 400, *synth.powerlaw_edges(900, 400, 9000, seed=3)
    r, c, v = synth.sym_norm_coo(eu, ei, nu, ni)
    n = nu + ni
    g = hip_ops.CsrGraph.from_coo_host(np.stack([r, c]), v, n, n, "cpu", symmetric=True)
    gen = torch.Generator().manual_seed(0)
    E0, G = (torch.randn(n, 64, generator=gen) * 0.1).requires_grad_(), torch.randn(n, 64, generator=gen)
    threads = torch.get_num_threads()
    torch.set_num_threads(1)          # as in the workers (the stand-in's sums depend on torch's thread partition)
    try:
        o = hip_ops.lightgcn_mean(g, E0, 3)
        o.backward(G)
    finally:
        torch.set_num_threads(threads)
    assert torch.equal(got[0], o.detach())
    # (the torch-CPU stand-in's autograd blocks its sparse products by row width: its backward agrees to rounding only; the
    # HIP kernels' backward is bitwise too -- test_lightgcn_mean_on_feature_slices_forward_and_backward_bitwise, on the device)
    np.testing.assert_allclose(got[1].numpy(), E0.grad.numpy(), rtol=1e-5, atol=1e-7)


from tests._cpu_ops import cpu_ops  # noqa: E402,F401  (fixture)


# ---- the run driver with `n_gpus: 2`: quick_start from the torchrun environment -------------------------------------
def _worker_quick_start(rank, world, port, root, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    from tests import _cpu_ops
    _cpu_ops.install()
    from mmrec_amd.utils.quick_start import quick_start
    from tests._env import write_dataset
    golden = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tiny.npz")))
    data_path = write_dataset(os.path.join(root, "rank%d" % rank), golden)
    os.chdir(os.path.join(root, "rank%d" % rank))                     # ./log/ is written relative to the cwd
    results, best = quick_start("FREEDOM", "baby", dict(n_gpus=world, use_gpu=False, data_path=data_path, epochs=2,
                                                        train_batch_size=256, save_recommended_topk=False, dropout=[0.8],
                                                        reg_weight=[1e-3], learning_rate=0.01), save_model=False)
    torch.save((results, best), out + ".%d" % rank)
    dist.barrier()
    dist.destroy_process_group()


def test_quick_start_with_n_gpus_matches_single_process(tmp_path, golden, cpu_ops, monkeypatch):
    """`n_gpus: 2` end to end through the run driver: ranks from RANK / LOCAL_RANK / WORLD_SIZE, gloo group (CPU), the
    `ShardedFREEDOM` class picked by get_model, Trainer.fit with early-stopping bookkeeping and the sharded evaluation --
    both ranks report the grid results of the single-process run (same best-valid / test metric dicts)."""
    from mmrec_amd.utils.quick_start import init_distributed, quick_start
    from tests._env import write_dataset
    with pytest.raises(RuntimeError):
        init_distributed({"n_gpus": 2})                              # not launched as 2 processes: says how to launch
    out = str(tmp_path / "qs.pt")
    mp.spawn(_worker_quick_start, args=(2, _free_port(), str(tmp_path), out), nprocs=2, join=True)
    got = [torch.load(out + ".%d" % r, weights_only=False) for r in range(2)]
    data_path = write_dataset(tmp_path / "single", golden)
    monkeypatch.chdir(tmp_path / "single")
    ref = quick_start("FREEDOM", "baby", dict(use_gpu=False, data_path=data_path, epochs=2, train_batch_size=256,
                                              save_recommended_topk=False, dropout=[0.8], reg_weight=[1e-3],
                                              learning_rate=0.01), save_model=False)
    for r in range(2):
        assert got[r][1] == ref[1] and len(got[r][0]) == len(ref[0]) == 1
        for a, b in zip(got[r][0], ref[0]):
            assert a[0] == b[0]
            for k in b[1]:
                assert abs(a[1][k] - b[1][k]) <= 1e-4 and abs(a[2][k] - b[2][k]) <= 1e-4, (k, a[1][k], b[1][k])
