"""Diagnostic: cl_conv3 at a BASELINE layer shape against torch's own GPU convolution (checker only), with the
positions of the worst errors.   python tools/cl_check_big.py N S C0 C1 Cout"""
import ctypes as C
import os
import sys
import numpy as np
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepwmh_amd import _lib
if os.environ.get('WMH_CL_VARIANT'):          # diagnostic build of cl_conv3.hip with extra -D flags, loaded instead of the library's
    import subprocess
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'deepwmh_amd', 'csrc')
    so = '/tmp/libcl_variant.so'
    subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '--offload-arch=gfx950', '-fPIC', '-std=c++17', '-Wno-unused-result'] +
                          os.environ['WMH_CL_VARIANT'].split() + ['-shared', os.path.join(csrc, 'cl_conv3.hip'), os.path.join(csrc, 'wmh_kernels.hip'),
                           os.path.join(csrc, 'conv3d_k3_bf16.hip'), os.path.join(csrc, 'conv3d_k3_bf16_ws.hip'), os.path.join(csrc, 'conv3d_k3.hip'),
                           os.path.join(csrc, 'conv3d_k3_ws.hip'), '-o', so])
    _lib.LIB_PATH = so
    _lib.SIGNATURES = {k: v for k, v in _lib.SIGNATURES.items() if k.startswith(('wmh_cl_conv3', 'wmh_conv3d_k3_pack', 'wmh_conv3d_k3_packed', 'wmh_last_error'))
                       and 'first' not in k and 'device' not in k}
lib = _lib.load()
N, S, C0, C1, Cout = [int(v) for v in sys.argv[1:6]]
Cin = C0 + C1
torch.manual_seed(0)
x = (torch.randn(N, Cin, S, S, S, device='cuda') * 1.5 + 0.3).to(torch.bfloat16)
ss = torch.stack([1 + 0.3 * torch.randn(N, Cin, device='cuda'), 0.4 * torch.randn(N, Cin, device='cuda')], -1).contiguous()
w = (torch.randn(Cout, Cin, 3, 3, 3, device='cuda') / np.sqrt(27 * Cin))
bias = torch.randn(Cout, device='cuda')
wq = w.to(torch.bfloat16).float()
z = x.float() * ss[:, :, 0][:, :, None, None, None] + ss[:, :, 1][:, :, None, None, None]
xin = torch.maximum(z, z * 0.01).to(torch.bfloat16).float()
ref = F.conv3d(xin, wq, bias, padding=1)
# CL16 (cl_common.h): [N, C/8, D, H, W, 8]
to_cl = lambda t: t.reshape(t.shape[0], t.shape[1] // 8, 8, *t.shape[2:]).permute(0, 1, 3, 4, 5, 2).contiguous()
x0 = to_cl(x[:, :C0])
x1 = to_cl(x[:, C0:]) if C1 else None
ss0 = ss[:, :C0].contiguous()
ss1 = ss[:, C0:].contiguous() if C1 else None
wh = np.ascontiguousarray(w.cpu().numpy().reshape(Cout, Cin, 27))
wf = np.empty(lib.wmh_conv3d_k3_packed_halfs_bf16(0, Cin, Cout), dtype=np.uint16)
lib.wmh_conv3d_k3_pack_weights_bf16(0, wh.ctypes.data_as(C.c_void_p), Cin, Cout, wf.ctypes.data_as(C.c_void_p))
wfd = torch.from_numpy(wf.view(np.int16)).cuda()
npart = lib.wmh_cl_conv3_num_partials(S, S, S)
y = torch.full((N, Cout // 8, S, S, S, 8), float('nan'), dtype=torch.bfloat16, device='cuda')
part = torch.empty(N, Cout, npart, 4, device='cuda')
_lib.check(lib.wmh_cl_conv3_fwd(_lib.ptr(x0), _lib.ptr(ss0), C0, _lib.ptr(x1), _lib.ptr(ss1), C1, _lib.ptr(wfd), _lib.ptr(bias), _lib.ptr(y), _lib.ptr(part),
                                N, S, S, S, Cout, 0.01, None), 'cl_conv3')
torch.cuda.synchronize()
got = y.float().permute(0, 1, 5, 2, 3, 4).reshape(N, Cout, S, S, S)
err = (got - ref).abs()
tol = 2.0 ** -7 * ref.abs() + 2e-3 * float(ref.abs().max())
bad = (err > tol) | ~torch.isfinite(got)
print('shape N=%d S=%d %d+%d->%d: bad %d of %d, max err %.4g, |ref| max %.4g' % (N, S, C0, C1, Cout, int(bad.sum()), bad.numel(), float(err[torch.isfinite(err)].max()), float(ref.abs().max())))
if bad.any():
    idx = bad.nonzero()
    print('first bad (n, c, z, y, x):', idx[:12].tolist())
    key = torch.stack([idx[:, 0], idx[:, 2] // 4, idx[:, 3] // 4, idx[:, 2] % 4, idx[:, 3] % 4, idx[:, 1] // 32], 1)
    u, c = torch.unique(key, dim=0, return_counts=True)
    print('(n, tz, ty, wave, m, g) -> count of bad elements (32 x * 32 c = 1024 is a whole M-tile):')
    for row, cnt in zip(u.tolist()[:64], c.tolist()[:64]):
        print('   ', row, cnt)
    b0 = idx[0].tolist()
    print('got / ref along x at the first bad (n,c,z,y):', got[b0[0], b0[1], b0[2], b0[3], :8].tolist(), ref[b0[0], b0[1], b0[2], b0[3], :8].tolist())
    for d, name in enumerate('nczyx'):
        vals, cnt = torch.unique(idx[:, d], return_counts=True)
        print('  by %s:' % name, dict(zip(vals.tolist()[:40], cnt.tolist()[:40])))
