"""Run driver: config -> data -> loaders -> hyper-parameter grid of (seed, model, Trainer.fit)
(reference: utils/quick_start.py:19-108)."""
import os
import platform
from itertools import product
from logging import getLogger

from .configurator import Config
from .dataloader import EvalDataLoader, TrainDataLoader
from .dataset import RecDataset
from .logger import init_logger
from .utils import dict2str, eval_batch_size, get_model, get_trainer, init_seed


def init_distributed(config_dict):
    """New key `n_gpus` (default 1).  n_gpus > 1: this process is one rank of a `torchrun --nproc-per-node n_gpus`
    launch (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment); it takes GPU LOCAL_RANK.  Returns
    (rank, world) -- the process group itself is created by `start_process_group` once the device is known."""
    n = int((config_dict or {}).get('n_gpus') or 1)
    if n <= 1:
        return 0, 1
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != n:
        raise RuntimeError('n_gpus={} needs one process per GPU: launch with `python -m torch.distributed.run --nnodes=1 '
                           '--nproc-per-node {} --master-addr 127.0.0.1 ...` (WORLD_SIZE is {})'.format(n, n, world))
    config_dict['gpu_id'] = int(os.environ.get('LOCAL_RANK', '0'))
    return int(os.environ.get('RANK', '0')), world


def start_process_group(config, rank, world):
    import torch.distributed as tdist
    if world <= 1 or tdist.is_initialized():
        return
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')     # dmabuf IPC (RCCL across processes on this driver)
    on_gpu = getattr(config['device'], 'type', 'cpu') == 'cuda'
    kw = {}
    if on_gpu:          # Config pinned this process to GPU LOCAL_RANK through CUDA_VISIBLE_DEVICES: it is device 0 here
        import torch
        kw['device_id'] = torch.device('cuda', torch.cuda.current_device())
    tdist.init_process_group('nccl' if on_gpu else 'gloo', rank=rank, world_size=world, **kw)


def quick_start(model, dataset, config_dict, save_model=True, mg=False):
    config_dict = dict(config_dict or {})
    rank, world = init_distributed(config_dict)
    config = Config(model, dataset, config_dict, mg)
    init_logger(config)
    logger = getLogger()
    start_process_group(config, rank, world)
    if rank > 0:
        import logging
        logger.setLevel(logging.WARNING)          # one log: rank 0's (all ranks compute the same losses / metrics)
    logger.info('██Server: \t' + platform.node())
    logger.info('██Dir: \t' + os.getcwd() + '\n')
    logger.info(config)

    data = RecDataset(config)
    logger.info(str(data))
    train_set, valid_set, test_set = data.split()
    for title, part in (('Training', train_set), ('Validation', valid_set), ('Testing', test_set)):
        logger.info('\n===={}====\n{}'.format(title, part))   # str() also sets inter_num (loaders need it)

    train_data = TrainDataLoader(config, train_set, batch_size=config['train_batch_size'], shuffle=True)
    valid_data = EvalDataLoader(config, valid_set, additional_dataset=train_set, batch_size=eval_batch_size(config))
    test_data = EvalDataLoader(config, test_set, additional_dataset=train_set, batch_size=eval_batch_size(config))

    logger.info('\n\n=================================\n\n')
    if 'seed' not in config['hyper_parameters']:
        config['hyper_parameters'] = ['seed'] + config['hyper_parameters']
    names = config['hyper_parameters']
    grid = list(product(*[(config[n] or [None]) for n in names]))
    metric = config['valid_metric'].lower()
    results, best_value, best_idx = [], 0.0, 0
    for idx, combo in enumerate(grid):
        for n, v in zip(names, combo):
            config[n] = v
        init_seed(config['seed'])
        logger.info('========={}/{}: Parameters:{}={}======='.format(idx + 1, len(grid), names, combo))
        train_data.pretrain_setup()
        net = get_model(config['model'], sharded=world > 1)(config, train_data).to(config['device'])
        logger.info(net)
        trainer = get_trainer()(config, net, mg)
        _, best_valid, best_test = trainer.fit(train_data, valid_data=valid_data, test_data=test_data,
                                               saved=save_model)
        results.append((combo, best_valid, best_test))
        if best_test[metric] > best_value:
            best_value, best_idx = best_test[metric], idx
        logger.info('best valid result: {}'.format(dict2str(best_valid)))
        logger.info('test result: {}'.format(dict2str(best_test)))
        logger.info('████Current BEST████:\nParameters: {}={},\nValid: {},\nTest: {}\n\n\n'.format(
            names, results[best_idx][0], dict2str(results[best_idx][1]), dict2str(results[best_idx][2])))

    logger.info('\n============All Over=====================')
    for combo, valid, test in results:
        logger.info('Parameters: {}={},\n best valid: {},\n best test: {}'.format(
            names, combo, dict2str(valid), dict2str(test)))
    logger.info('\n\n█████████████ BEST ████████████████')
    logger.info('\tParameters: {}={},\nValid: {},\nTest: {}\n\n'.format(
        names, results[best_idx][0], dict2str(results[best_idx][1]), dict2str(results[best_idx][2])))
    return results, best_idx
