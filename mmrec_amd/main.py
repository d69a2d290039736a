"""CLI: `python -m mmrec_amd.main -m FREEDOM -d baby` (reference: src/main.py:16-27)."""
import argparse
import os

from mmrec_amd.utils.quick_start import quick_start

os.environ['NUMEXPR_MAX_THREADS'] = '48'


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument('--model', '-m', type=str, default='SELFCFED_LGN', help='name of models')
    parser.add_argument('--dataset', '-d', type=str, default='baby', help='name of datasets')
    parser.add_argument('--n_gpus', type=int, default=1,
                        help='new: > 1 = one process per GPU under torchrun (models with a Sharded<Name> variant)')
    args, _ = parser.parse_known_args()
    config_dict = {'gpu_id': 0}
    if args.n_gpus > 1:
        config_dict['n_gpus'] = args.n_gpus
    quick_start(model=args.model, dataset=args.dataset, config_dict=config_dict, save_model=True)


if __name__ == '__main__':
    main()
