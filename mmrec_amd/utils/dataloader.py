"""Train / eval loaders with the reference's batch semantics (utils/dataloader.py:15-418).

What the kernels consume comes from here: the `[3, B]` int64 id batch (users, positives, one
uniform negative drawn from train-seen items outside the user's history), the train interaction
matrix as scipy COO, and per eval batch `[users, mask[2, n]]` (train positives to exclude).
Sampling stays on the host and consumes Python's / numpy's global RNGs in the same order as the
reference (`random.sample(items, 1)` == one `_randbelow(len(items))`; `DataFrame.sample` for the
epoch shuffle), so with the same seed the id batches are identical.  Loaders are stateful
single-pass iterators like the reference's (pointer reset on StopIteration).
"""
import math
import ctypes
import random
import sys
from logging import getLogger

import numpy as np
import torch
from scipy.sparse import coo_matrix


_LIVE_MT = []   # [] = not probed yet; [None] = unavailable; [(state pointer, index pointer)]


def _live_mt_state():
    """Pointers to the Mersenne Twister state INSIDE the interpreter's global `random` generator, so that the C sampler
    (include/mmrec_hip.h: mmrec_host_sample_negatives) draws from it in place instead of through a getstate() / setstate()
    round trip per batch.  CPython's `_random.Random` object is {PyObject_HEAD; int index; uint32_t state[624]}
    (Modules/_randommodule.c); the layout is PROBED, not assumed: the view must show exactly what random.getstate() reports,
    before and after a draw, or the marshalling path stays in use."""
    if _LIVE_MT:
        return _LIVE_MT[0]
    found = None
    try:
        inst = getattr(random, '_inst', None)
        if type(inst) is random.Random and sys.implementation.name == 'cpython':
            base = id(inst) + ctypes.sizeof(ctypes.c_ssize_t) + ctypes.sizeof(ctypes.c_void_p)      # PyObject_HEAD
            index = ctypes.c_int.from_address(base)
            state = (ctypes.c_uint32 * 624).from_address(base + ctypes.sizeof(ctypes.c_int))

            def agrees():
                internal = random.getstate()[1]
                return index.value == internal[-1] and tuple(state) == tuple(internal[:-1])
            saved = random.getstate()
            ok = agrees()
            for _ in range(700):              # across a regeneration of the 624-word block
                random.getrandbits(32)
            ok = ok and agrees()
            random.setstate(saved)
            if ok and agrees():
                found = (ctypes.cast(state, ctypes.c_void_p), ctypes.cast(ctypes.pointer(index), ctypes.c_void_p))
    except Exception:
        found = None
    _LIVE_MT.append(found)
    return found


class AbstractDataLoader(object):
    def __init__(self, config, dataset, additional_dataset=None, batch_size=1, neg_sampling=False,
                 shuffle=False):
        self.config = config
        self.logger = getLogger()
        self.dataset = dataset
        self.dataset_bk = dataset.copy(dataset.df)
        self.additional_dataset = additional_dataset
        self.batch_size = self.step = batch_size
        self.shuffle = shuffle
        self.neg_sampling = neg_sampling
        self.device = config['device']
        self.sparsity = 1 - dataset.inter_num / dataset.user_num / dataset.item_num
        self.pr = 0
        self.inter_pr = 0

    def pretrain_setup(self):
        pass

    def __len__(self):
        return math.ceil(self.pr_end / self.step)

    def __iter__(self):
        if self.shuffle:
            self._shuffle()
        return self

    def __next__(self):
        if self.pr >= self.pr_end:
            self.pr = self.inter_pr = 0
            raise StopIteration()
        return self._next_batch_data()

    @property
    def pr_end(self):
        raise NotImplementedError

    def _shuffle(self):
        raise NotImplementedError

    def _next_batch_data(self):
        raise NotImplementedError


class TrainDataLoader(AbstractDataLoader):
    def __init__(self, config, dataset, batch_size=1, shuffle=False):
        super().__init__(config, dataset, batch_size=batch_size, neg_sampling=True, shuffle=shuffle)
        uid, iid = dataset.uid_field, dataset.iid_field
        self.all_items = dataset.df[iid].unique().tolist()      # items seen in training
        self.all_uids = dataset.df[uid].unique()
        self.all_items_set, self.all_users_set = set(self.all_items), set(self.all_uids)
        self.all_item_len = len(self.all_items)
        self.use_full_sampling = config['use_full_sampling']
        if not config['use_neg_sampling']:
            self.sample_func = self._pairs_only
        elif self.use_full_sampling:
            self.sample_func = self._user_ids_only
        else:
            self.sample_func = self._pairs_with_negative
        # per-user training history: the reference keeps {user: set(items)} (dataloader.py:282-291, one groupby group per
        # user: ~20 s at 1M users); here the same sets live as a CSR (sorted unique items per user) built by one sort, and
        # the dict the reference's attribute name promises is materialised only if somebody asks for it
        u_arr, i_arr = dataset.df[uid].values.astype(np.int64), dataset.df[iid].values.astype(np.int64)
        stride = np.int64(i_arr.max() + 1) if i_arr.size else np.int64(1)
        key = np.unique(u_arr * stride + i_arr)
        n_users = int(dataset.user_num)
        self._hist_rowptr = np.zeros(n_users + 1, dtype=np.int64)
        np.cumsum(np.bincount(key // stride, minlength=n_users), out=self._hist_rowptr[1:])
        self._hist_items = (key % stride).astype(np.int64) if key.size else np.zeros(1, np.int64)
        self._history_dict = None
        self.neighborhood_loss_required = config['use_neighborhood_loss']
        if self.neighborhood_loss_required:
            raise NotImplementedError('use_neighborhood_loss is not on the accelerated path')
        # optional device-side sampler; the host sampler stays the bit-exact parity mode
        self.device_neg_sampling = bool(config['device_neg_sampling']) and str(self.device).startswith('cuda')
        self._dev_sampler = None
        self._sample_counter = 0
        self._native_sampler = None      # resolved on first use: C host sampler of the library, or False
        self._cols_of = self._cols = None
        self._items_arr = None

    @property
    def history_items_per_u(self):
        if self._history_dict is None:
            rp, it = self._hist_rowptr, self._hist_items
            self._history_dict = {u: set(it[rp[u]:rp[u + 1]].tolist()) for u in range(len(rp) - 1) if rp[u + 1] > rp[u]}
        return self._history_dict

    def pretrain_setup(self):
        """Called once per hyper-parameter combination after seeding: restores the unshuffled data and
        fixes the item order the negative sampler indexes into."""
        if self.shuffle:
            self.dataset = self.dataset_bk.copy(self.dataset_bk.df)
        self.all_items.sort()
        if self.use_full_sampling:
            self.all_uids.sort()
        random.shuffle(self.all_items)
        self._items_arr = None

    def inter_matrix(self, form='coo', value_field=None):
        """Train interactions as scipy sparse [n_users, n_items] with data 1.0 (float64)."""
        df, uid, iid = self.dataset.df, self.dataset.uid_field, self.dataset.iid_field
        if not uid or not iid:
            raise ValueError('dataset doesn\'t exist uid/iid, thus can not converted to sparse matrix')
        if value_field is None:
            data = np.ones(len(df))
        elif value_field in df.columns:
            data = df[value_field].values
        else:
            raise ValueError('value_field [{}] should be one of `df_feat`\'s features.'.format(value_field))
        mat = coo_matrix((data, (df[uid].values, df[iid].values)),
                         shape=(self.dataset.user_num, self.dataset.item_num))
        if form == 'coo':
            return mat
        if form == 'csr':
            return mat.tocsr()
        raise NotImplementedError('sparse matrix format [{}] has not been implemented.'.format(form))

    @property
    def pr_end(self):
        return len(self.all_uids) if self.use_full_sampling else len(self.dataset)

    def _shuffle(self):
        self.dataset.shuffle()
        if self.use_full_sampling:
            np.random.shuffle(self.all_uids)

    def _next_batch_data(self):
        return self.sample_func()

    def _slice(self):
        # the (user, item) columns of the current (shuffled) frame as arrays, fetched once per frame: a DataFrame
        # slice + two column lookups per batch were ~0.2 ms of host time
        df = self.dataset.df
        if self._cols_of is not df:
            self._cols = (df[self.config['USER_ID_FIELD']].values, df[self.config['ITEM_ID_FIELD']].values)
            self._cols_of = df
        lo, hi = self.pr, self.pr + self.step
        self.pr += self.step
        return self._cols[0][lo:hi], self._cols[1][lo:hi]

    def _device_negatives(self, users_dev):
        from mmrec_amd import hip_ops
        if self._dev_sampler is None:
            rowptr = torch.from_numpy(self._hist_rowptr.astype(np.int32)).to(self.device)    # items sorted per user
            col = torch.from_numpy(self._hist_items.astype(np.int32)).to(self.device)
            cand = torch.tensor(sorted(self.all_items_set), dtype=torch.int32, device=self.device)
            self._dev_sampler = (rowptr, col, cand)
        rowptr, col, cand = self._dev_sampler
        self._sample_counter += 1
        return hip_ops.sample_negatives(users_dev, rowptr, col, cand, self.config['seed'] or 0, self._sample_counter)

    def _pairs_with_negative(self):
        users, items = self._slice()
        if self.device_neg_sampling:
            pairs = torch.from_numpy(np.stack([users.astype(np.int64), items.astype(np.int64)])).to(self.device)
            return torch.cat([pairs, self._device_negatives(pairs[0].contiguous()).unsqueeze(0)])
        negs = self._sample_neg_ids(users)
        batch = np.stack([users.astype(np.int64), items.astype(np.int64), negs])
        return torch.from_numpy(batch).to(self.device)   # one H2D copy instead of three

    def _pairs_only(self):
        users, items = self._slice()
        return torch.from_numpy(np.stack([users.astype(np.int64), items.astype(np.int64)])).to(self.device)

    def _user_ids_only(self):
        users = torch.tensor(self.all_uids[self.pr: self.pr + self.step]).type(torch.LongTensor)
        self.pr += self.step
        return users.to(self.device)

    def _sample_neg_ids(self, users):
        """The reference's sampler through `mmrec_host_sample_negatives` (a HOST function of the library: CPython's
        Mersenne Twister continued in C from `random.getstate()`): same ids and same consumption of the global
        `random` stream as `_sample_neg_ids_loop`, which remains the fallback (library absent) and the check."""
        if self._native_sampler is None:
            try:
                from mmrec_amd import _lib
                self._native_sampler = _lib.load().mmrec_host_sample_negatives
            except Exception:        # host-only use without the built library (CPU plumbing runs)
                self._native_sampler = False
        if self._native_sampler is False or self.all_item_len.bit_length() > 32:
            return self._sample_neg_ids_loop(users)
        ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        if self._items_arr is None:
            self._items_arr = np.ascontiguousarray(self.all_items, dtype=np.int64)
            self._const_ptrs = (ptr(self._hist_rowptr), ptr(self._hist_items), ptr(self._items_arr))
        users = np.ascontiguousarray(users, dtype=np.int64)
        out = np.empty(users.shape[0], dtype=np.int64)
        live = _live_mt_state()
        if live is not None:       # the C sampler continues the interpreter's own generator in place
            mt_ptr, idx_ptr = live
        else:                      # marshal the state through random.getstate() / setstate() (36 us per batch)
            version, internal, gauss = random.getstate()
            mt = np.array(internal[:-1], dtype=np.uint32)
            idx = ctypes.c_int32(internal[-1])
            mt_ptr, idx_ptr = ptr(mt), ctypes.byref(idx)
        err = self._native_sampler(mt_ptr, idx_ptr, ptr(users), users.shape[0], *self._const_ptrs, self.all_item_len,
                                   ptr(out))
        if err != 0:
            raise RuntimeError("mmrec_host_sample_negatives failed: %d" % err)
        if live is None:
            random.setstate((version, tuple(mt.tolist()) + (idx.value,), gauss))
        return out

    def _sample_neg_ids_loop(self, users):
        """One uniform train-seen item per user, rejected while in the user's history.  Consumes the
        global `random` stream exactly like the reference's `random.sample(all_items, 1)[0]` loop:
        that call is one `_randbelow(n)`, i.e. `getrandbits(n.bit_length())` redrawn while >= n --
        inlined here (the per-draw Python call overhead is most of a small-dataset training step)."""
        items, n = self.all_items, self.all_item_len
        hist, bits, k = self.history_items_per_u, random.getrandbits, self.all_item_len.bit_length()
        out = np.empty(len(users), dtype=np.int64)
        for idx, u in enumerate(users):
            seen = hist[u]
            while True:
                r = bits(k)
                while r >= n:
                    r = bits(k)
                cand = items[r]
                if cand not in seen:
                    break
            out[idx] = cand
        return out


class EvalBatch(list):
    """[users, mask] exactly as the reference hands it over, plus a per-loader dict (`cache`, keyed by
    `cache_key`) in which a model may keep data derived from this never-changing batch -- the fused
    evaluation keeps the mask's CSR form there instead of re-sorting it at every evaluation."""
    cache = None
    cache_key = None


class EvalDataLoader(AbstractDataLoader):
    """Batches of eval users with the mask of their training positives (rows relative to the batch)."""

    def __init__(self, config, dataset, additional_dataset=None, batch_size=1, shuffle=False):
        super().__init__(config, dataset, additional_dataset=additional_dataset, batch_size=batch_size,
                         shuffle=shuffle)
        if additional_dataset is None:
            raise ValueError('Training datasets is nan')
        uid, iid = dataset.uid_field, dataset.iid_field
        eval_u = dataset.df[uid].unique()
        # the reference walks `groupby(...).get_group(u)` user by user (dataloader.py:359-407: 2 x 50 us per user, minutes
        # at 1M users); the same lists -- a user's items in frame order, users in order of first appearance -- come
        # out of one stable sort per frame
        train_flat, train_len = self._lists_in_user_order(additional_dataset.df[additional_dataset.uid_field].values,
                                                         additional_dataset.df[additional_dataset.iid_field].values, eval_u)
        if train_len.size and train_len.min() == 0:        # get_group raises for a user without training interactions
            raise KeyError(eval_u[int(np.argmin(train_len))])
        self.train_pos_len_list = train_len.tolist()
        rows = np.repeat(np.arange(len(eval_u), dtype=np.int64), train_len)
        self.pos_items_per_u = torch.from_numpy(np.stack([rows, train_flat.astype(np.int64)])).to(self.device)
        self._mask_offsets = np.concatenate([[0], np.cumsum(train_len)])
        eval_flat, eval_len = self._lists_in_user_order(dataset.df[uid].values, dataset.df[iid].values, eval_u)
        self.eval_items_per_u = np.split(eval_flat, np.cumsum(eval_len)[:-1]) if len(eval_u) else []
        self.eval_len_list = np.asarray(eval_len)
        self._eval_flat = eval_flat            # the same lists, concatenated: what the device metrics build their CSR from
        self.eval_u = torch.tensor(eval_u).type(torch.LongTensor).to(self.device)
        self._batch_cache = {}

    @staticmethod
    def _lists_in_user_order(uids, iids, users):
        """-> (items of users[0] ++ items of users[1] ++ ..., lengths): each user's items in their order in the frame"""
        uids, iids = np.asarray(uids), np.asarray(iids)
        order = np.argsort(uids, kind='stable')
        su, si = uids[order], iids[order]
        lo, hi = np.searchsorted(su, users, 'left'), np.searchsorted(su, users, 'right')
        lens = (hi - lo).astype(np.int64)
        total = int(lens.sum())
        starts = np.cumsum(lens) - lens
        idx = np.repeat(lo - starts, lens) + np.arange(total, dtype=np.int64)
        return si[idx], lens

    @property
    def pr_end(self):
        return self.eval_u.shape[0]

    def _shuffle(self):
        self.dataset.shuffle()

    def _next_batch_data(self):
        lo, hi = self._mask_offsets[self.pr], self._mask_offsets[min(self.pr + self.step, self.pr_end)]
        users = self.eval_u[self.pr: self.pr + self.step]
        mask = self.pos_items_per_u[:, lo:hi].clone()
        mask[0] -= self.pr
        batch = EvalBatch([users, mask])
        batch.cache, batch.cache_key = self._batch_cache, self.pr   # eval batches never change: models may
        self.inter_pr = hi                                           # keep per-batch derived data here
        self.pr += self.step
        return batch

    def get_eval_items(self):
        return self.eval_items_per_u

    def get_eval_len_list(self):
        return self.eval_len_list

    def get_eval_users(self):
        return self.eval_u.cpu()
