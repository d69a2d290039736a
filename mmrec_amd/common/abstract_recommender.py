"""Plugin API of MMRec (the drop-in boundary).

Same contract as the reference's `common/abstract_recommender.py:10-103`: a model is an nn.Module
with hooks `pre_epoch_processing`, `post_epoch_processing`, `calculate_loss(interaction)`,
`predict`, `full_sort_predict([users, mask])`; `GeneralRecommender.__init__(config, dataloader)`
reads the user/item counts from `dataloader.dataset` and loads `image_feat.npy` / `text_feat.npy`
onto `config['device']`.

One *optional* extension: `full_sort_topk(interaction, k) -> LongTensor[b, k]`, which our Trainer
prefers when a model offers it (fused score + mask + top-K on the GPU); the reference Trainer
keeps working through `full_sort_predict`.
"""
import os

import numpy as np
import torch
import torch.nn as nn


class AbstractRecommender(nn.Module):
    def pre_epoch_processing(self):
        return None

    def post_epoch_processing(self):
        return None

    def calculate_loss(self, interaction):
        raise NotImplementedError

    def predict(self, interaction):
        raise NotImplementedError

    def full_sort_predict(self, interaction):
        raise NotImplementedError

    def __str__(self):
        n_params = sum(int(np.prod(p.size())) for p in self.parameters())
        return super().__str__() + '\nTrainable parameters: {}'.format(n_params)


class GeneralRecommender(AbstractRecommender):
    def __init__(self, config, dataloader):
        super().__init__()
        self.USER_ID = config['USER_ID_FIELD']
        self.ITEM_ID = config['ITEM_ID_FIELD']
        self.NEG_ITEM_ID = config['NEG_PREFIX'] + self.ITEM_ID
        self.n_users = dataloader.dataset.get_user_num()
        self.n_items = dataloader.dataset.get_item_num()
        self.batch_size = config['train_batch_size']
        self.device = config['device']
        self.v_feat = self.t_feat = None
        if not config['end2end'] and config['is_multimodal_model']:
            root = os.path.abspath(config['data_path'] + config['dataset'])
            # new key (additive): {'v': tensor [n_items, F], 't': tensor} handed over in memory instead of the .npy
            # files -- a 500K x 4096 table (8.2 GB) generated on the device need not travel through the disk
            feats = {k: t.to(device=self.device, dtype=torch.float32) for k, t in (config['in_memory_features'] or {}).items()}
            for key, cfg in (() if feats else (('v', 'vision_feature_file'), ('t', 'text_feature_file'))):
                path = os.path.join(root, config[cfg])
                if os.path.isfile(path):
                    arr = np.load(path, allow_pickle=True)
                    feats[key] = torch.from_numpy(arr).type(torch.FloatTensor).to(self.device)
            self.v_feat, self.t_feat = feats.get('v'), feats.get('t')
            assert self.v_feat is not None or self.t_feat is not None, 'Features all NONE'
