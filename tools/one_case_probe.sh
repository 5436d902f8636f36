python - <<'PY'
import os, sys, time, tempfile, subprocess, numpy as np
sys.path.insert(0, os.getcwd())
import torch
from deepwmh_amd import checkpoint as ck, nifti
from deepwmh_amd.plan import baseline_plan
from deepwmh_amd.synthetic import he_init_state_dict, synthetic_flair
tmp = tempfile.mkdtemp()
plan = baseline_plan(); sd = he_init_state_dict(plan, 0)
folder = ck.trainer_folder(os.path.join(tmp, 'model'), 'Task002_FinalModel')
ck.save_checkpoint(folder, {k: torch.from_numpy(v) for k, v in sd.items()}, ck.make_plans(plan, spacing=(1.0, 1.0, 1.0), use_mask_for_norm=True), 'all', 'model_best', epoch=1)
img = os.path.join(tmp, 'in'); os.makedirs(img)
v = synthetic_flair((160, 192, 160), 1)[0]
nifti.write(os.path.join(img, 'case00_0000.nii.gz'), (v * 50 + 300 * (v != 0)).astype(np.float32), spacing=(1.0, 1.0, 1.0))
env = dict(os.environ, RESULTS_FOLDER=os.path.join(tmp, 'model'), PATH=os.path.join(os.getcwd(), 'bin') + os.pathsep + os.environ['PATH'], CUDA_VISIBLE_DEVICES='0')
cmd = ('nnUNet_predict -i %s -o %s -tr nnUNetTrainerV2 -m 3d_fullres -p nnUNetPlansv2.1 -t Task002_FinalModel -f all -chk model_best --disable_post_processing --selected_cases case00' % (img, os.path.join(tmp, 'out'))).split()
for rep in range(3):
    t0 = time.time(); p = subprocess.run(cmd + (['--overwrite_existing'] if rep else []), env=env, capture_output=True, text=True); dt = time.time() - t0
    print('run %d: wall %.2f s rc %d' % (rep, dt, p.returncode)); print(p.stdout[-600:])
t0 = time.time(); subprocess.run([sys.executable, '-c', 'import torch; torch.cuda.init(); torch.zeros(1, device="cuda")']); print('import torch + context: %.2f s' % (time.time() - t0))
PY
