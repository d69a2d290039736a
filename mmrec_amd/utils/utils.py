"""Small helpers: model registry, seeding, early stopping (reference: utils/utils.py:17-115)."""
import datetime
import importlib
import random

import numpy as np
import torch


def get_local_time():
    return datetime.datetime.now().strftime('%b-%d-%Y-%H-%M-%S')


def get_model(model_name, sharded=False):
    """`FREEDOM` -> class FREEDOM in models/freedom.py.  Looks in a top-level `models` package first
    (the reference layout, when these files are dropped into its src/) and then in mmrec_amd.models.
    sharded (config `n_gpus` > 1): the module's `Sharded<Name>` class -- the same model over one process per GPU."""
    last_err = None
    for pkg in ('models', 'mmrec_amd.models'):
        try:
            module = importlib.import_module('{}.{}'.format(pkg, model_name.lower()))
            return getattr(module, ('Sharded' if sharded else '') + model_name)
        except (ImportError, AttributeError) as err:
            last_err = err
    if sharded:
        raise ImportError('model {} has no multi-GPU (n_gpus > 1) variant Sharded{}: {}'.format(model_name, model_name, last_err))
    raise ImportError('model {} not found: {}'.format(model_name, last_err))


def get_trainer():
    from mmrec_amd.common.trainer import Trainer
    return Trainer


def init_seed(seed):
    random.seed(seed)
    np.random.seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
        torch.cuda.manual_seed_all(seed)
    torch.manual_seed(seed)


def early_stopping(value, best, cur_step, max_step, bigger=True):
    """-> (best, cur_step, stop_flag, update_flag); stops after more than max_step non-improving evals."""
    improved = value > best if bigger else value < best
    if improved:
        return value, 0, False, True
    cur_step += 1
    return best, cur_step, cur_step > max_step, False


def dict2str(result_dict):
    return ''.join('{}: {:.04f}    '.format(k, v) for k, v in result_dict.items())


def random_sample_range(n, k):
    """`random.sample(range(n), k)` -- same list, same consumption of Python's global generator -- through the
    library's host function (CPython 3.10's algorithm continued in C from `random.getstate()`); plain
    `random.sample` when the library is not built or the interpreter is not the 3.10 whose algorithm was restated."""
    import ctypes
    import random
    import sys
    import numpy as np
    if sys.version_info[:2] != (3, 10) or n.bit_length() > 31:
        return random.sample(range(n), k)
    try:
        from mmrec_amd import _lib
        fn = _lib.load().mmrec_host_random_sample_range
    except Exception:
        return random.sample(range(n), k)
    version, internal, gauss = random.getstate()
    mt = np.array(internal[:-1], dtype=np.uint32)
    idx = ctypes.c_int32(internal[-1])
    out, scratch = np.empty(k, dtype=np.int64), np.empty(max(n, 1), dtype=np.int32)
    ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    err = fn(ptr(mt), ctypes.byref(idx), n, k, ptr(out), ptr(scratch))
    if err != 0:
        raise RuntimeError("mmrec_host_random_sample_range failed: %d" % err)
    random.setstate((version, tuple(mt.tolist()) + (idx.value,), gauss))
    return out


def graph_step_mode(config):
    """`hip_graph_step`: True / False, or absent / 'auto' = replay the training step as a hipGraph for the plugins that
    declare `graph_capturable = True` (verified: same kernels in the same order, tests/test_models_gpu.py) and run every
    other model -- anything a user drops in -- eagerly."""
    v = config['hip_graph_step']
    if v is None or str(v).lower() == 'auto':
        return 'auto'
    return 'on' if v is True or str(v).lower() in ('true', '1', 'yes', 'on') else 'off'


def eval_batch_size(config):
    """Batch size of the evaluation loaders.  The reference's 4096 (overall.yaml: eval_batch_size) exists because
    `full_sort_predict` materialises a [batch, n_items] score matrix; the fused score + mask + top-K never does, and
    it is most efficient on all users at once (Amazon-Baby: one 0.17 ms call instead of five 0.07 ms calls plus their
    launches).  New key `hip_eval_batch_size` (default 65536) applies when the fused evaluation is on and the model is on
    the GPU; the per-user results are independent of the batching."""
    fused = config['hip_fused_eval']
    on_gpu = getattr(config['device'], 'type', str(config['device'])) == 'cuda'
    if (fused is None or fused) and on_gpu:
        big = config['hip_eval_batch_size']
        return max(int(config['eval_batch_size']), int(big) if big else 65536)
    return config['eval_batch_size']


# ---- dense / sparse kNN-graph helpers with the reference's names and results (utils/utils.py:117-184), for model code
# ---- written against them.  The plugins in mmrec_amd/models do not materialise [n, n] similarity matrices: they take
# ---- neighbours and values from the fused score + top-K kernel (mmrec_amd/graph.py).
def _inverse_power(total, power):
    """total ** power with the infinities of empty rows replaced by 0"""
    scale = torch.pow(total, power)
    return torch.where(torch.isinf(scale), torch.zeros_like(scale), scale)


def build_sim(context):
    """cosine similarity of every pair of rows, dense [n, n]"""
    unit = context / torch.norm(context, p=2, dim=-1, keepdim=True)
    return unit @ unit.t()


def build_knn_neighbourhood(adj, topk):
    """keep the `topk` largest entries of every row, zero the rest (dense)"""
    val, ind = torch.topk(adj, topk, dim=-1)
    return torch.zeros_like(adj).scatter_(-1, ind, val)


def get_dense_laplacian(adj, normalization='none'):
    """'sym': D^-1/2 A D^-1/2, 'rw': D^-1 A, 'none': A -- D = row sums; row / column scaling instead of diagonal
    matrix products (same products, same order, no [n, n] diagonal operands)"""
    if normalization == 'sym':
        d = _inverse_power(adj.sum(-1), -0.5)
        return (adj * d[:, None]) * d[None, :]
    if normalization == 'rw':
        return adj * _inverse_power(adj.sum(-1), -1)[:, None]
    if normalization == 'none':
        return adj
    raise ValueError("normalization %r" % (normalization,))


def compute_normalized_laplacian(adj):
    return get_dense_laplacian(adj, 'sym')


def get_sparse_laplacian(edge_index, edge_weight, num_nodes, normalization='none'):
    """the same normalisations on a COO edge list; degrees are sums of the weights leaving `row`"""
    row, col = edge_index[0], edge_index[1]
    deg = torch.zeros(num_nodes, dtype=edge_weight.dtype, device=edge_weight.device).index_add_(0, row, edge_weight)
    if normalization == 'sym':
        d = _inverse_power(deg, -0.5)
        edge_weight = d[row] * edge_weight * d[col]
    elif normalization == 'rw':
        inv = 1.0 / deg
        edge_weight = torch.where(torch.isinf(inv), torch.zeros_like(inv), inv)[row] * edge_weight
    return edge_index, edge_weight


def build_knn_normalized_graph(adj, topk, is_sparse, norm_type):
    """top-k neighbourhood of a dense similarity matrix, normalised; sparse COO (row-major entries) or dense"""
    val, ind = torch.topk(adj, topk, dim=-1)
    if not is_sparse:
        return get_dense_laplacian(torch.zeros_like(adj).scatter_(-1, ind, val), normalization=norm_type)
    n = adj.shape[0]
    rows = torch.arange(n, device=adj.device).repeat_interleave(topk)
    index, weight = get_sparse_laplacian(torch.stack((rows, ind.reshape(-1))), val.reshape(-1), n, normalization=norm_type)
    return torch.sparse_coo_tensor(index, weight, adj.shape)
