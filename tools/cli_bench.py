"""End-to-end timing of the `nnUNet_predict` drop-in on BASELINE-sized cases (160x192x160, baseline plan, random-init
weights): NIfTI read + preprocess, sliding-window predict, export + write, per case and as cases/s for the whole command.
usage: python tools/cli_bench.py [ncases] [--tta] [--save_softmax] [--spacing sx sy sz]      (WMH_PREDICT_INFLIGHT: cases in flight without --tta)"""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from deepwmh_amd import checkpoint as ck, nifti
    from deepwmh_amd.plan import baseline_plan
    from deepwmh_amd.synthetic import he_init_state_dict, synthetic_flair
    from deepwmh_amd.cli import predict
    n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 4
    tta = '--tta' in sys.argv
    spacing = (1.0, 1.0, 1.0)
    if '--spacing' in sys.argv:
        i = sys.argv.index('--spacing'); spacing = tuple(float(v) for v in sys.argv[i + 1:i + 4])
    tmp = tempfile.mkdtemp(prefix='wmh_cli_')
    plan = baseline_plan()
    sd = he_init_state_dict(plan, 0)
    plans = ck.make_plans(plan, spacing=(1.0, 1.0, 1.0), use_mask_for_norm=True)
    folder = ck.trainer_folder(os.path.join(tmp, 'model'), 'Task002_FinalModel')
    import torch
    ck.save_checkpoint(folder, {k: torch.from_numpy(v) for k, v in sd.items()}, plans, 'all', 'model_best', epoch=1)
    img, out = os.path.join(tmp, 'in'), os.path.join(tmp, 'out')
    os.makedirs(img)
    for c in range(n):
        vol = synthetic_flair((160, 192, 160), c)[0] * 50 + 300
        vol[synthetic_flair((160, 192, 160), c)[0] == 0] = 0
        nifti.write(os.path.join(img, 'case%02d_0000.nii.gz' % c), vol.astype(np.float32), spacing=spacing)
    os.environ['RESULTS_FOLDER'] = os.path.join(tmp, 'model')
    argv = ['-i', img, '-o', out, '-tr', 'nnUNetTrainerV2', '-m', '3d_fullres', '-p', 'nnUNetPlansv2.1', '-t', 'Task002_FinalModel',
            '-f', 'all', '-chk', 'model_best', '--disable_post_processing'] + ([] if tta else ['--disable_tta']) + (['--save_softmax'] if '--save_softmax' in sys.argv else [])
    t0 = time.time()
    predict.main(argv)
    dt = time.time() - t0
    print('whole command: %.2f s for %d cases -> %.3f cases/s (includes model load and engine set-up)' % (dt, n, n / dt))


if __name__ == '__main__':
    main()
