"""Per-layer timing of the conv kernels at the BASELINE shapes (batch 8), HIP events on the launch
stream, interleaved repeats.  python tools/conv_bench.py [layer-name-substring ...]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from deepwmh_amd import _lib  # noqa: E402

lib = _lib.load_diag() if (os.environ.get('WMH_STAMPS') or os.environ.get('WMH_DIAG_LIB')) else _lib.load()   # stamps and variant switches: diagnostic library only
B = int(os.environ.get('WMH_BATCH', '8'))
REPS = int(os.environ.get('WMH_REPS', '5'))
# name, C0, C1, Cout, D(in), stride
LAYERS = [('enc0.0', 1, 0, 32, 128, 1), ('enc0.1', 32, 0, 32, 128, 1), ('enc1.0', 32, 0, 64, 128, 2),
          ('enc1.1', 64, 0, 64, 64, 1), ('enc2.0', 64, 0, 128, 64, 2), ('enc2.1', 128, 0, 128, 32, 1),
          ('enc3.0', 128, 0, 256, 32, 2), ('enc3.1', 256, 0, 256, 16, 1), ('enc4.0', 256, 0, 320, 16, 2),
          ('enc4.1', 320, 0, 320, 8, 1), ('bott.0', 320, 0, 320, 8, 2), ('bott.1', 320, 0, 320, 4, 1),
          ('dec0.0', 320, 320, 320, 8, 1), ('dec0.1', 320, 0, 320, 8, 1), ('dec1.0', 256, 256, 256, 16, 1),
          ('dec1.1', 256, 0, 256, 16, 1), ('dec2.0', 128, 128, 128, 32, 1), ('dec2.1', 128, 0, 128, 32, 1),
          ('dec3.0', 64, 64, 64, 64, 1), ('dec3.1', 64, 0, 64, 64, 1), ('dec4.0', 32, 32, 32, 128, 1),
          ('dec4.1', 32, 0, 32, 128, 1)]
sel = sys.argv[1:]
tot_ms = tot_fl = 0.0
for name, C0, C1, Cout, D, s in LAYERS:
    if sel and not any(k in name for k in sel):
        continue
    Cin = C0 + C1
    Do = (D - 1) // s + 1
    cfg = lib.wmh_conv3d_k3_config(Cin, Cout, D, D, D, s)
    x0 = torch.randn(B, C0, D, D, D, device='cuda')
    x1 = torch.randn(B, C1, D, D, D, device='cuda') if C1 else None
    ss0 = torch.rand(B, C0, 2, device='cuda') + 0.5
    ss1 = torch.rand(B, C1, 2, device='cuda') + 0.5 if C1 else None
    w = (np.random.default_rng(0).standard_normal((Cout, Cin, 27)) / np.sqrt(27 * Cin)).astype(np.float32)
    wf = np.empty(lib.wmh_conv3d_k3_packed_floats(cfg, Cin, Cout), np.float32)
    lib.wmh_conv3d_k3_pack_weights(cfg, w.ctypes.data_as(C.c_void_p), Cin, Cout, wf.ctypes.data_as(C.c_void_p))
    wfd = torch.from_numpy(wf).cuda()
    bias = torch.zeros(Cout, device='cuda')
    y = torch.empty(B, Cout, Do, Do, Do, device='cuda')
    npart = lib.wmh_conv3d_k3_num_partials(cfg, B, Do, Do, Do)
    part = torch.empty(B, Cout, npart, 4, device='cuda') if not os.environ.get('WMH_NOPART') else None
    nws = lib.wmh_conv3d_k3_workspace_floats(cfg, Cin, B, D, D, D, Cout, s)
    ws = torch.empty(nws, device='cuda') if nws else None

    def run():
        _lib.check(lib.wmh_conv3d_k3_fwd(cfg, _lib.ptr(x0), _lib.ptr(ss0), C0, _lib.ptr(x1), _lib.ptr(ss1), C1, _lib.ptr(wfd),
                                         _lib.ptr(bias), _lib.ptr(y), _lib.ptr(part), _lib.ptr(ws), B, D, D, D, Cout, s, 0.01, None), 'conv')
    run(); torch.cuda.synchronize()
    ts = []
    for _ in range(REPS):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    if os.environ.get('WMH_STAMPS'):
        import ctypes
        nblk = 1 << 22
        stamps = torch.zeros(nblk * 4, dtype=torch.int64, device='cuda')
        lib.wmh_debug_set_conv_stamps(ctypes.c_void_p(stamps.data_ptr()))
        run(); torch.cuda.synchronize()
        lib.wmh_debug_set_conv_stamps(None)
        st = stamps.view(-1, 4).cpu().numpy()
        st = st[st[:, 3] > 0]
        print('   stamps over %d workgroups: prologue %.0f, loop %.0f (%.0f per chunk, %d chunks), epilogue %.0f cycles (100 MHz? see s_memtime)'
              % (len(st), st[:, 0].mean(), st[:, 1].mean(), (st[:, 1] / st[:, 3]).mean(), st[0, 3], st[:, 2].mean()))
    fl = 2.0 * Cin * Cout * 27 * B * Do ** 3
    ms = float(np.median(ts))
    tot_ms += ms; tot_fl += fl
    print('%-7s ws %5.1f MB cfg%d Cin %3d Cout %3d out %3d^3 s%d: %8.3f ms (min %8.3f)  %7.1f GFLOP  %6.1f TFLOP/s'
          % (name, nws * 4 / 1e6, cfg, Cin, Cout, Do, s, ms, min(ts), fl / 1e9, fl / ms / 1e9), flush=True)
    del x0, x1, y, part
if tot_ms > 0:
    print('TOTAL %.3f ms  %.1f GFLOP  %.1f TFLOP/s' % (tot_ms, tot_fl / 1e9, tot_fl / tot_ms / 1e9))

# transposed convolutions of the BASELINE decoder
if not sel or any('tu' in k for k in sel):
    import ctypes as C2
    for name, Cin, Cout, D in [('tu0', 320, 320, 4), ('tu1', 320, 256, 8), ('tu2', 256, 128, 16), ('tu3', 128, 64, 32), ('tu4', 64, 32, 64)]:
        x = torch.randn(B, Cin, D, D, D, device='cuda')
        ss = torch.rand(B, Cin, 2, device='cuda') + 0.5
        w = (np.random.default_rng(1).standard_normal((Cin, Cout, 2, 2, 2)) * 0.1).astype(np.float32)
        wf = np.empty(lib.wmh_convtranspose3d_k2s2_packed_floats(Cin, Cout), np.float32)
        lib.wmh_convtranspose3d_k2s2_pack_weights(w.ctypes.data_as(C2.c_void_p), Cin, Cout, wf.ctypes.data_as(C2.c_void_p))
        wd, wfd = torch.from_numpy(w).cuda(), torch.from_numpy(wf).cuda()
        y = torch.empty(B, Cout, 2 * D, 2 * D, 2 * D, device='cuda')
        for label, fn in (('valu', lambda: lib.wmh_convtranspose3d_k2s2_fwd(_lib.ptr(x), _lib.ptr(ss), _lib.ptr(wd), _lib.ptr(y), B, Cin, Cout, D, D, D, 0.01, None)),
                          ('mfma', lambda: lib.wmh_convtranspose3d_k2s2_mfma_fwd(_lib.ptr(x), _lib.ptr(ss), _lib.ptr(wfd), _lib.ptr(y), B, Cin, Cout, D, D, D, 0.01, None))):
            fn(); torch.cuda.synchronize()
            ts = []
            for _ in range(REPS):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); fn(); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            gb = 4.0 * B * D ** 3 * (Cin + 8 * Cout) / 1e9
            print('%-4s %s Cin %3d Cout %3d in %2d^3: %7.3f ms  %6.2f GB -> %5.2f TB/s, %6.1f TFLOP/s'
                  % (name, label, Cin, Cout, D, np.median(ts), gb, gb / np.median(ts), 2.0 * B * D ** 3 * Cin * Cout * 8 / np.median(ts) / 1e9))
