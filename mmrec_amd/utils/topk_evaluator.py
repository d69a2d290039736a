"""TopKEvaluator: hit matrix + Recall/NDCG/Precision/MAP@topk rounded to 4 dp (reference:
utils/topk_evaluator.py:19-149).  The reference builds the hit matrix with a Python double loop
(2 of the 3 s of a Baby evaluation); here it is one vectorised membership test
on (user, item) keys -- integer work, identical result."""
import os

import numpy as np
import pandas as pd
import torch

from .metrics import metrics_dict
from .utils import get_local_time

topk_metrics = {m.lower(): m for m in ['Recall', 'Recall2', 'Precision', 'NDCG', 'MAP']}


class TopKEvaluator(object):
    def __init__(self, config):
        self.config = config
        self.metrics = config['metrics']
        self.topk = config['topk']
        self.save_recom_result = config['save_recommended_topk']
        self._check_args()

    def collect(self, interaction, scores_tensor, full=False):
        """top-max(k) item ids of one evaluation batch from its scores (topk_evaluator.py:36-56): `full` = one row of
        scores per user; otherwise a flat score vector split by `interaction.user_len_list`, short rows padded with
        -inf.  Our Trainer takes the ids from the fused score + mask + top-K instead; kept for trainers written against
        the reference."""
        lens = interaction.user_len_list
        if full is True:
            matrix = scores_tensor.view(len(lens), -1)
        else:
            matrix = torch.nn.utils.rnn.pad_sequence(torch.split(scores_tensor, lens, dim=0), batch_first=True,
                                                      padding_value=-np.inf)
        return torch.topk(matrix, max(self.topk), dim=-1)[1]

    def evaluate(self, batch_matrix_list, eval_data, is_test=False, idx=0):
        pos_items = eval_data.get_eval_items()
        pos_len = np.asarray(eval_data.get_eval_len_list())
        topk_index = torch.cat(batch_matrix_list, dim=0).cpu().numpy()
        if self.save_recom_result and is_test:
            self._dump(topk_index, eval_data, idx)
        assert len(pos_len) == len(topk_index)
        hit = self.hit_matrix(topk_index, pos_items, pos_len)
        result = {}
        for metric in self.metrics:
            curve = metrics_dict[metric](hit, pos_len)
            for k in self.topk:
                result['{}@{}'.format(metric, k)] = round(curve[k - 1], 4)
        return result

    def evaluate_device(self, batch_matrix_list, eval_data, is_test=False, idx=0):
        """Same result dict as `evaluate`, with the hit test and the per-user metric sums done by the
        HIP kernel on the top-k ids where they already are (device), the user mean in float64 there as well."""
        from mmrec_amd import hip_ops
        topk_index = torch.cat(batch_matrix_list, dim=0)
        if self.save_recom_result and is_test:
            self._dump(topk_index.cpu().numpy(), eval_data, idx)
        gt = getattr(eval_data, '_gt_csr', None)
        if gt is None or gt[0].device != topk_index.device:
            flat = getattr(eval_data, '_eval_flat', None)
            if flat is not None:               # our EvalDataLoader: the lists exist concatenated already
                gt = hip_ops.flat_to_csr(flat, eval_data.get_eval_len_list(), topk_index.device)
            else:
                gt = hip_ops.lists_to_csr(eval_data.get_eval_items(), topk_index.device)
            eval_data._gt_csr = gt
        assert gt[0].numel() - 1 == topk_index.shape[0]
        ks = sorted(self.topk)
        # the user mean in float64 on the device too: 16 numbers cross PCIe (at 1M users the [n, 4, 4] float64 block was
        # 128 MB and numpy's strided mean over it 0.18 s of a 0.30 s evaluation -- more than the ranking kernels)
        per_user = hip_ops.topk_metrics_per_user(topk_index, gt[0], gt[1], ks)
        means = (per_user.sum(dim=0) / per_user.shape[0]).cpu().numpy()
        order = {'recall': 0, 'ndcg': 1, 'precision': 2, 'map': 3}
        result = {}
        for metric in self.metrics:
            if metric not in order:       # e.g. recall2: only the host path implements it
                return self.evaluate(batch_matrix_list, eval_data, is_test=False, idx=idx)
            curve = means[order[metric]]
            for k in self.topk:
                result['{}@{}'.format(metric, k)] = round(curve[ks.index(k)], 4)
        return result

    @staticmethod
    def hit_matrix(topk_index, pos_items, pos_len):
        n, k = topk_index.shape
        stride = np.int64(max(int(topk_index.max()) + 1, max((int(np.max(p)) for p in pos_items if len(p)),
                                                               default=0) + 1))
        owners = np.repeat(np.arange(n, dtype=np.int64), pos_len)
        truth = owners * stride + np.concatenate([np.asarray(p, dtype=np.int64) for p in pos_items])
        keys = np.arange(n, dtype=np.int64)[:, None] * stride + topk_index.astype(np.int64)
        return np.isin(keys, truth)

    def _dump(self, topk_index, eval_data, idx):
        max_k = max(self.topk)
        out_dir = os.path.abspath(self.config['recommend_topk'])
        os.makedirs(out_dir, exist_ok=True)
        path = os.path.join(out_dir, '{}-{}-idx{}-top{}-{}.csv'.format(
            self.config['model'], self.config['dataset'], idx, max_k, get_local_time()))
        frame = pd.DataFrame(topk_index)
        frame.insert(0, 'id', eval_data.get_eval_users())
        frame.columns = ['id'] + ['top_' + str(i) for i in range(max_k)]
        frame.astype(int).to_csv(path, sep='\t', index=False)

    def _check_args(self):
        if isinstance(self.metrics, str):
            self.metrics = [self.metrics]
        if not isinstance(self.metrics, list):
            raise TypeError('metrics must be str or list')
        for m in self.metrics:
            if m.lower() not in topk_metrics:
                raise ValueError('There is no user grouped topk metric named {}!'.format(m))
        self.metrics = [m.lower() for m in self.metrics]
        if isinstance(self.topk, int):
            self.topk = [self.topk]
        if not isinstance(self.topk, list):
            raise TypeError('The topk must be a integer, list')
        for k in self.topk:
            if k <= 0:
                raise ValueError('topk must be a positive integer or a list of positive integers, '
                                 'but get `{}`'.format(k))

    def __str__(self):
        return 'The TopK Evaluator Info:\n\tMetrics:[' + ', '.join(topk_metrics[m] for m in self.metrics) + \
            '], TopK:[' + ', '.join(map(str, self.topk)) + ']'
