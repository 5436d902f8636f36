"""HBM traffic of the dominant pair kernel per layer (run under rocprofv3 --kernel-trace --pmc FETCH_SIZE, or WRITE_SIZE):
a calibration launch of known size, then the three 128^3 stride-1 layers of the BASELINE plan at WMH_BATCH (16) patches, each launched
WMH_REPS times: enc0.1 (32 -> 32), dec4.0 (32 + 32 -> 32, two sources) and dec4.1 (32 -> 32).
Raw ctypes on purpose: WMH_LIB_PATH may point at a library built from an older commit (tools/build_old_lib.sh), which lacks newer entry
points.  The diagnostic library's switches apply when WMH_LIB_PATH points at a diagnostic build (WMH_SUPERTILE=0, WMH_XCD_SWIZZLE=0).
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/x -o p -- python3 tools/pmc_layer.py
    python3 tools/parse_pmc.py gpurun_out/x"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = os.environ.get('WMH_LIB_PATH', os.path.join(ROOT, 'deepwmh_amd', 'lib', 'libdeepwmh_amd.so'))
lib = C.CDLL(path)
vp, i32, f32 = C.c_void_p, C.c_int, C.c_float
lib.wmh_conv3d_k3_packed_halfs_f16x2.restype = C.c_size_t
lib.wmh_conv3d_k3_fwd_f16x2.argtypes = [i32, vp, vp, i32, vp, vp, i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, vp]
lib.wmh_conv3d_k3_pack_weights_f16x2.argtypes = [i32, vp, i32, i32, vp]
lib.wmh_instnorm_lrelu_apply.argtypes = [vp, vp, vp, i32, i32, C.c_size_t, f32, vp]
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())

n = 256 * 1024 * 1024                      # 1 GiB read + 1 GiB written, far beyond the 256 MiB Infinity Cache
a = torch.randn(n, device='cuda'); b = torch.empty_like(a)
ss = torch.tensor([[1.5, 0.25]], device='cuda')
for _ in range(2):
    assert lib.wmh_instnorm_lrelu_apply(P(a), P(ss), P(b), 1, 1, n, 0.01, None) == 0
torch.cuda.synchronize()
del a, b

N, S, reps = int(os.environ.get('WMH_BATCH', '16')), 128, int(os.environ.get('WMH_REPS', '3'))
g = torch.Generator(device='cuda').manual_seed(1)
for (C0, C1, Cout) in ((32, 0, 32), (32, 32, 32), (32, 0, 32)):
    Cin = C0 + C1
    x0 = torch.randn(N, C0, S, S, S, device='cuda', generator=g)
    x1 = torch.randn(N, C1, S, S, S, device='cuda', generator=g) if C1 else None
    mk = lambda Cc: torch.stack([1 + 0.3 * torch.randn(N, Cc, device='cuda', generator=g), 0.2 * torch.randn(N, Cc, device='cuda', generator=g)], -1).contiguous()
    ss0, ss1 = mk(C0), (mk(C1) if C1 else None)
    w = (np.random.default_rng(0).standard_normal((Cout, Cin, 27)) / np.sqrt(27 * Cin)).astype(np.float32)
    cfg = lib.wmh_conv3d_k3_bf16_config(lib.wmh_conv3d_k3_config(Cin, Cout, S, S, S, 1), Cout, S)
    wf = np.empty(lib.wmh_conv3d_k3_packed_halfs_f16x2(cfg, Cin, Cout), np.uint16)
    assert lib.wmh_conv3d_k3_pack_weights_f16x2(cfg, w.ctypes.data_as(vp), Cin, Cout, wf.ctypes.data_as(vp)) == 0
    wfd = torch.from_numpy(wf.view(np.int16)).cuda()
    y = torch.empty(N, Cout, S, S, S, device='cuda')
    part = torch.empty(N, Cout, lib.wmh_conv3d_k3_num_partials_f16x2(cfg, N, S, S, S), 4, device='cuda')
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        assert lib.wmh_conv3d_k3_fwd_f16x2(cfg, P(x0), P(ss0), C0, P(x1), P(ss1), C1, P(wfd), None, P(y), P(part), None, N, S, S, S, Cout, 0.01, None) == 0
    e1.record(); torch.cuda.synchronize()
    print('%d x %d^3, %d + %d -> %d (cfg %d): %.3f ms per launch; algorithmic input %.2f GB, output %.2f GB'
          % (N, S, C0, C1, Cout, cfg, e0.elapsed_time(e1) / reps, 4e-9 * N * Cin * S ** 3, 4e-9 * N * Cout * S ** 3), flush=True)
    del x0, x1, y
print('done')
