"""Where a chunk iteration of cl_conv3_kernel spends its cycles: builds a diagnostic copy of the kernel (-DWMH_CL_DIAG:
s_memtime stamps around the phases of both roles; the production library has none) and prints per-iteration averages
over the workgroups.  Read the SHARES, not the run time.   python tools/cl_diag.py [S C0 C1 Cout]"""
import ctypes as C
import os
import subprocess
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, 'deepwmh_amd', 'csrc')
so = '/tmp/libcl_diag.so'
subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '--offload-arch=gfx950', '-fPIC', '-std=c++17', '-Wno-unused-result', '-DWMH_CL_DIAG',
                       '-shared', os.path.join(csrc, 'cl_conv3.hip'), os.path.join(csrc, 'wmh_kernels.hip'), '-o', so])
from deepwmh_amd import _lib
plib = _lib.load()
lib = C.CDLL(so)
S, C0, C1, Cout = [int(v) for v in sys.argv[1:5]] if len(sys.argv) >= 5 else (128, 32, 0, 32)
N, Cin = 8, C0 + C1
x0 = torch.randn(N, S, S, S, C0, device='cuda').to(torch.bfloat16)
x1 = torch.randn(N, S, S, S, C1, device='cuda').to(torch.bfloat16) if C1 else None
ss0 = torch.randn(N, C0, 2, device='cuda') * 0.2 + torch.tensor([1.0, 0.0], device='cuda')
ss1 = (torch.randn(N, C1, 2, device='cuda') * 0.2 + torch.tensor([1.0, 0.0], device='cuda')) if C1 else None
w = (np.random.default_rng(0).standard_normal((Cout, Cin, 27)) / np.sqrt(27 * Cin)).astype(np.float32)
wf = np.empty(plib.wmh_conv3d_k3_packed_halfs_bf16(0, Cin, Cout), dtype=np.uint16)
plib.wmh_conv3d_k3_pack_weights_bf16(0, w.ctypes.data_as(C.c_void_p), Cin, Cout, wf.ctypes.data_as(C.c_void_p))
wfd = torch.from_numpy(wf.view(np.int16)).cuda()
y = torch.empty(N, S, S, S, Cout, dtype=torch.bfloat16, device='cuda')
npart = plib.wmh_cl_conv3_num_partials(S, S, S)
part = torch.empty(N, Cout, npart, 4, device='cuda')
stamps = torch.zeros(256 * 16, dtype=torch.int64, device='cuda')
lib.wmh_cl_diag_stamps.argtypes = [C.c_void_p]
lib.wmh_cl_conv3_fwd.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
import subprocess as _sp
for rep in range(3):
    stamps.zero_()
    lib.wmh_cl_diag_stamps(P(stamps))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = lib.wmh_cl_conv3_fwd(P(x0), P(ss0), C0, P(x1), P(ss1), C1, P(wfd), None, P(y), P(part), N, S, S, S, Cout, 0.01, None)
    e1.record()
    torch.cuda.synchronize()
    assert rc == 0
st = stamps.cpu().numpy().reshape(256, 16).astype(np.float64)
it = st[:, 6]
ok = it > 0
per = lambda col: float((st[ok, col] / it[ok]).mean())
print('%dx%d^3, %d+%d -> %d channels, WMH_CL_ABL=%s: %.3f ms (diagnostic build), %d iterations per workgroup'
      % (N, S, C0, C1, Cout, os.environ.get('WMH_CL_ABL', '0'), e0.elapsed_time(e1), int(it[ok].mean())))
print('  loader  : issue W+DMA %8.0f | wait DMA(it+1) %8.0f | transform %8.0f | barrier1 %8.0f | weights->LDS %8.0f | barrier2 %8.0f | sum %8.0f'
      % (per(0), per(1), per(2), per(3), per(4), per(5), sum(per(c) for c in range(6))))
print('  consumer: MFMA phase  %8.0f | barrier1 %8.0f | barrier2 %8.0f | epilogue/iter %8.0f | tile prologue/iter %8.0f | sum %8.0f'
      % (per(8), per(9), per(10), per(11), per(12), sum(per(c) for c in range(8, 13))))
