"""End-to-end iteration rate of the `nnUNet_train` drop-in (plan + preprocess a synthetic 160x192x160 dataset, then one
epoch of B batches of 2 x 128^3 patches).  usage: python tools/train_cli_bench.py [batches]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    from test_train_pipeline import make_raw_dataset
    from deepwmh_amd.cli import plan_and_preprocess as pp, train
    nb = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 40
    tmp = tempfile.mkdtemp(prefix='wmh_train_')
    raw, pre, res = os.path.join(tmp, 'raw'), os.path.join(tmp, 'pre'), os.path.join(tmp, 'results')
    make_raw_dataset(raw, n=4, shape=(160, 192, 160))
    os.environ.update(nnUNet_raw_data_base=raw, nnUNet_preprocessed=pre, RESULTS_FOLDER=res)
    t = time.time(); pp.main(['-t', '1', '--verify_dataset_integrity']); print('plan_and_preprocess: %.1f s' % (time.time() - t))
    argv = ['3d_fullres', 'nnUNetTrainerV2', '1', 'all', '-e', '1', '-b', str(nb), '--noval', '--val_batches', '1',
            '--disable_postprocessing_on_folds']
    t = time.time(); train.main(argv); dt = time.time() - t
    print('nnUNet_train: %d iterations (+1 validation batch, checkpoint writes, engine set-up) in %.1f s -> %.1f ms per iteration all-in'
          % (nb, dt, 1e3 * dt / nb))


if __name__ == '__main__':
    main()
