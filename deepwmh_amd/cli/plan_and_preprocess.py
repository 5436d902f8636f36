"""`nnUNet_plan_and_preprocess -t <id> --verify_dataset_integrity` drop-in
(/root/reference/deepwmh/pipeline/DCNN_multistage.py:281,487,685); env `nnUNet_raw_data_base`, `nnUNet_preprocessed`
(DCNN_multistage.py:188-191)."""
import argparse
import os
import sys


def build_parser():
    ap = argparse.ArgumentParser(prog='nnUNet_plan_and_preprocess')
    ap.add_argument('-t', '--task_ids', nargs='+', required=True)
    ap.add_argument('--verify_dataset_integrity', action='store_true')
    ap.add_argument('-pl3d', '--planner3d', default='ExperimentPlanner3D_v21')
    ap.add_argument('-pl2d', '--planner2d', default='ExperimentPlanner2D_v21')
    ap.add_argument('-no_pp', action='store_true')
    ap.add_argument('-tl', type=int, default=8)
    ap.add_argument('-tf', type=int, default=8)
    return ap


def main(argv=None):
    args = build_parser().parse_args(argv)
    raw, pre = os.environ.get('nnUNet_raw_data_base'), os.environ.get('nnUNet_preprocessed')
    if not raw or not pre:
        sys.exit('nnUNet_raw_data_base / nnUNet_preprocessed are not set')
    from ..data import plan_and_preprocess
    for t in args.task_ids:
        plan_and_preprocess(raw, pre, t, args.verify_dataset_integrity)
    return 0


if __name__ == '__main__':
    sys.exit(main())
