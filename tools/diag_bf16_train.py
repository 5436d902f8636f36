import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo')); sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tests'))
import numpy as np, torch
from deepwmh_amd import _lib
from deepwmh_amd.trainer import UNetTrainer, downsample_targets
from deepwmh_amd.plan import tiny_plan
from deepwmh_amd.synthetic import he_init_state_dict
lib = _lib.load(); P = _lib.ptr
# kernel-level: mode-1 (data gradient) bf16 vs fp32 on the same dy
g = torch.Generator().manual_seed(0)
for (Cin_total, Cout, off, cnt, D) in [(32, 32, 0, 32, 32), (64, 32, 32, 32, 32), (16, 8, 0, 16, 32), (64, 64, 0, 64, 16)]:
    w = (torch.randn(Cout, Cin_total, 27, generator=g) * 0.1).cuda()
    dy = torch.randn(2, Cout, D, D, D, generator=g).cuda()
    cfg = _lib.check(lib.wmh_conv3d_k3_config(Cout, cnt, D, D, D, 1))
    wf = torch.empty(lib.wmh_conv3d_k3_packed_floats(cfg, Cout, cnt), device='cuda')
    _lib.check(lib.wmh_conv3d_k3_pack_weights_device(cfg, P(w), Cin_total, Cout, off, cnt, 1, P(wf), None))
    y32 = torch.empty(2, cnt, D, D, D, device='cuda'); ws = torch.empty(max(1, lib.wmh_conv3d_k3_workspace_floats(cfg, Cout, 2, D, D, D, cnt, 1)), device='cuda')
    _lib.check(lib.wmh_conv3d_k3_fwd(cfg, P(dy), None, Cout, None, None, 0, P(wf), None, P(y32), None, P(ws), 2, D, D, D, cnt, 1, 0.01, None))
    if lib.wmh_conv3d_k3_bf16_supported(cfg):
        wb = torch.empty(lib.wmh_conv3d_k3_packed_halfs_bf16(cfg, Cout, cnt), dtype=torch.int16, device='cuda')
        _lib.check(lib.wmh_conv3d_k3_pack_weights_bf16_device(cfg, P(w), Cin_total, Cout, off, cnt, 1, P(wb), None))
        y16 = torch.empty_like(y32)
        _lib.check(lib.wmh_conv3d_k3_fwd_bf16(cfg, P(dy), None, Cout, None, None, 0, P(wb), None, P(y16), None, 2, D, D, D, cnt, 0.01, None))
        torch.cuda.synchronize()
        print('cfg', cfg, (Cin_total, Cout, off, cnt, D), 'rel L2 err bf16 vs fp32: %.4f' % float(((y16 - y32) ** 2).sum().sqrt() / (y32 ** 2).sum().sqrt()))
    else:
        print('cfg', cfg, 'no bf16')
# trainer-level: which parameters carry the error
plan = tiny_plan()
sd = he_init_state_dict(plan, 0)
torch.manual_seed(0)
x = torch.randn(2, 1, 32, 32, 32, device='cuda'); lab = (torch.rand(2, 32, 32, 32, device='cuda') < 0.2).to(torch.uint8)
td = downsample_targets(lab, plan.num_pool)
res = {}
def cos(a, b):
    return float((a * b).sum() / (a.norm() * b.norm() + 1e-30))
for name, fwd, dg in (('fp32', False, False), ('bf16', True, True), ('bf16 forward only', True, False), ('bf16 dgrad only', False, True)):
    tr = UNetTrainer(plan, sd, 'cuda:0', precision='bf16' if (fwd or dg) else 'fp32')
    tr.bf16_forward, tr.bf16_dgrad = fwd, dg
    tr.forward(x); l = float(tr.backward(td).cpu())
    res[name] = (l, tr.grads.clone())
g0 = res['fp32'][1]
for name in res:
    g = res[name][1]
    print('%-20s loss %.6f  relative L2 gradient difference %.4f  cosine %.5f' % (name, res[name][0], float((g - g0).norm() / g0.norm()), cos(g, g0)))
