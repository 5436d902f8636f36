"""Where a chunk iteration of cl_conv3_kernel spends its cycles, and at which shader clock: builds two diagnostic copies of the
kernel (-DWMH_CL_DIAG=1: s_memtime stamps around the phases of the staging waves, =2: of the four consumer waves plus the ratio
s_memtime / s_memrealtime of the whole run = the clock the CUs actually ran at; the production library has no stamps) and prints
per-iteration averages over the workgroups.  WMH_CL_ABL=m: timing-only ablations (compile-time, results are wrong): 1 no DMA,
4 no transform, 8 no MFMAs, 16 no epilogue, 64 no operand reloads.      python tools/cl_diag.py [S C0 C1 Cout]"""
import ctypes as C
import os
import subprocess
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, 'deepwmh_amd', 'csrc')
ABL = int(os.environ.get('WMH_CL_ABL', '0'))                 # timing-only ablations: 1 no DMA, 4 no transform, 8 no MFMAs, 16 no epilogue, 64 no operand reloads
ROLE = int(os.environ.get('WMH_CL_DIAG_ROLE', '0'))          # 1: stamps in the staging waves only, 2: in the consumer waves only, 0: both builds
def build(role):
    so = '/tmp/libcl_diag%d.so' % role
    subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '--offload-arch=gfx950', '-fPIC', '-std=c++17', '-Wno-unused-result', '-DWMH_CL_DIAG=%d' % role,
                           '-DWMH_CL_ABL_MASK=%d' % (ABL << 1), '-shared', os.path.join(csrc, 'cl_conv3.hip'), os.path.join(csrc, 'wmh_kernels.hip'), '-o', so])
    return so
from deepwmh_amd import _lib
plib = _lib.load()
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
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
GRID = 256
print('%dx%d^3, %d+%d -> %d channels, WMH_CL_ABL=%d' % (N, S, C0, C1, Cout, ABL))
for role in ((1, 2) if ROLE == 0 else (ROLE,)):
    lib = C.CDLL(build(role))
    stamps = torch.zeros(GRID * 16 + GRID * 4 * 8, dtype=torch.int64, device='cuda')
    lib.wmh_cl_diag_stamps.argtypes = [C.c_void_p]
    lib.wmh_cl_conv3_fwd.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]
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
    allst = stamps.cpu().numpy().astype(np.float64)
    ms = e0.elapsed_time(e1)
    if role == 1:
        st = allst[:GRID * 16].reshape(GRID, 16)
        ok = st[:, 6] > 0
        per = lambda col: float((st[ok, col] / st[ok, 6]).mean())
        print('  build 1 (stamps in the staging waves), %.3f ms, %d chunk iterations per workgroup; shader-clock cycles per iteration:' % (ms, int(st[ok, 6].mean())))
        print('    staging wave: issue table reads + DMA %6.0f | wait DMA(it+1) %6.0f | transform %6.0f | barrier wait %6.0f | sum %6.0f'
              % (per(0), per(1), per(2), per(3) + per(4) + per(5), sum(per(c) for c in range(6))))
        print('    weight wave : loads issued .. slice 0 released %6.0f | slice 0 -> LDS (waits for its loads) %6.0f | .. slice 1 released %6.0f | slice 1 -> LDS %6.0f | '
              '.. slice 2 released %6.0f (then 9 KB -> LDS and the barrier)' % (per(8), per(9), per(10), per(11), per(12)))
    else:
        pw = allst[GRID * 16:].reshape(GRID, 4, 8)
        ok = pw[:, 0, 4] > 0
        it = pw[ok, 0, 4]
        mhz = float((pw[ok, :, 5] / np.maximum(pw[ok, :, 6], 1)).mean() * 100)
        print('  build 2 (stamps in the consumer waves), %.3f ms; shader clock over the run (s_memtime / s_memrealtime at 100 MHz): %.0f MHz' % (ms, mhz))
        print('    108 MFMAs per iteration = 3456 cycles of issue; at this clock the bf16 MFMA peak is %.0f TFLOP/s' % (2500.0 * mhz / 2400.0))
        for w in range(4):
            v = pw[ok, w, :4] / it[:, None]
            print('    consumer wave %d: MFMA phase %6.0f | barrier wait %6.0f | epilogue + tile prologue, per iteration %6.0f | sum %6.0f'
                  % (w, v[:, 0].mean(), v[:, 1].mean() + v[:, 2].mean(), v[:, 3].mean(), v.sum(1).mean()))
