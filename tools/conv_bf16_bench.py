"""bf16-operand conv kernel at the BASELINE layer shapes it serves (batch WMH_BATCH, default 8): ms, effective TFLOP/s and
algorithmic HBM GB/s (fp32 activations read once + written once).  python tools/conv_bf16_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from deepwmh_amd import _lib  # noqa: E402

lib = _lib.load_diag() if (os.environ.get('WMH_STAMPS') or os.environ.get('WMH_DIAG_LIB')) else _lib.load()   # stamps and variant switches: diagnostic library only
P = _lib.ptr
B = int(os.environ.get('WMH_BATCH', '8'))
LAYERS = [('enc0.1', 32, 0, 32, 128), ('dec4.0', 32, 32, 32, 128), ('enc1.1', 64, 0, 64, 64), ('dec3.0', 64, 64, 64, 64),
          ('enc2.1', 128, 0, 128, 32), ('dec2.0', 128, 128, 128, 32), ('enc3.1', 256, 0, 256, 16), ('dec1.0', 256, 256, 256, 16)]
if os.environ.get('WMH_DIMS'):          # one custom shape: WMH_DIMS=120,128,128 [WMH_CH=32]
    dd = [int(v) for v in os.environ['WMH_DIMS'].split(',')]
    ch = int(os.environ.get('WMH_CH', '32'))
    LAYERS = [('custom', ch, 0, ch, dd)]
ST = int(os.environ.get('WMH_ST', '0'))        # bf16 tensors: bit 0 = sources, bit 1 = output (config 0 layers only)
IN16, OUT16 = ST & 1, (ST >> 1) & 1
tot = 0.0
for name, C0, C1, Cout, D in LAYERS:
    Cin = C0 + C1
    Dz, Dy, Dx = (D, D, D) if isinstance(D, int) else D
    cfg = lib.wmh_conv3d_k3_config(Cin, Cout, Dz, Dy, Dx, 1)
    st_layer = ST and cfg == 0
    if not st_layer:
        cfg = lib.wmh_conv3d_k3_bf16_config(cfg, Cout, Dx)      # 64-channel tile where it applies (WMH_BF16_WIDE=0: off)
    dt_in = torch.bfloat16 if (st_layer and IN16) else torch.float32
    x0 = torch.randn(B, C0, Dz, Dy, Dx, device='cuda').to(dt_in); x1 = torch.randn(B, C1, Dz, Dy, Dx, device='cuda').to(dt_in) if C1 else None
    ss0 = torch.rand(B, C0, 2, device='cuda') + 0.5; ss1 = torch.rand(B, C1, 2, device='cuda') + 0.5 if C1 else None
    w = (torch.randn(Cout, Cin, 27, device='cuda') / np.sqrt(27 * Cin)).contiguous()
    wf = torch.empty(lib.wmh_conv3d_k3_packed_halfs_bf16(cfg, Cin, Cout), dtype=torch.int16, device='cuda')
    _lib.check(lib.wmh_conv3d_k3_pack_weights_bf16_device(cfg, P(w), Cin, Cout, 0, Cin, 0, P(wf), None))
    bias = torch.zeros(Cout, device='cuda'); y = torch.empty(B, Cout, Dz, Dy, Dx, device='cuda', dtype=torch.bfloat16 if (st_layer and OUT16) else torch.float32)
    npart = lib.wmh_conv3d_k3_num_partials_bf16(cfg, B, Dz, Dy, Dx)
    part = torch.empty(B, Cout, npart, 4, device='cuda')
    if st_layer:
        run = lambda: _lib.check(lib.wmh_conv3d_k3_fwd_bf16_st(cfg, P(x0), P(ss0), C0, P(x1), P(ss1), C1, P(wf), P(bias), P(y), P(part), B, Dz, Dy, Dx, Cout,
                                                               0.01, IN16, OUT16, None), 'bf16 conv (bf16 tensors)')
    else:
        run = lambda: _lib.check(lib.wmh_conv3d_k3_fwd_bf16(cfg, P(x0), P(ss0), C0, P(x1), P(ss1), C1, P(wf), P(bias), P(y), P(part), B, Dz, Dy, Dx, Cout,
                                                            0.01, None), 'bf16 conv')
    run(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ms = float(np.median(ts)); tot += ms
    if os.environ.get('WMH_STAMPS'):
        import ctypes
        stamps = torch.zeros((1 << 20) * 4, dtype=torch.int64, device='cuda')
        lib.wmh_debug_set_conv_stamps(ctypes.c_void_p(stamps.data_ptr()))
        run(); torch.cuda.synchronize()
        lib.wmh_debug_set_conv_stamps(None)
        if os.environ.get('WMH_BF16_WS', '1') != '0':
            st = stamps.view(-1, 8).cpu().numpy(); st = st[st[:, 3] > 0]
            n = st[:, 3].mean()
            print('   per chunk iteration (ticks): consumer mfma %.0f + barrier wait %.0f, epilogue %.0f per iteration; loader work %.0f + barrier wait %.0f; %d workgroups x %.0f iterations'
                  % (st[:, 0].mean() / n, st[:, 1].mean() / n, st[:, 2].mean() / n, st[:, 4].mean() / n, st[:, 5].mean() / n, len(st), n))
            nchk = Cin // 16 if Cin % 16 == 0 else Cin // 16 + 1
            tiles = n / max(nchk, 1)
            print('   per tile (%d chunks): consumer barrier wait in the first chunk %.0f, in the others %.0f; loader work next to an epilogue %.0f, in the others %.0f'
                  % (nchk, st[:, 7].mean() / tiles, (st[:, 1] - st[:, 7]).mean() / tiles, st[:, 6].mean() / tiles, (st[:, 4] - st[:, 6]).mean() / tiles))
        else:
            st = stamps.view(-1, 4).cpu().numpy(); st = st[st[:, 3] > 0]
            print('   per workgroup (s_memtime ticks): staging %.0f, mfma %.0f, epilogue %.0f over %d chunks; %d workgroups'
                  % (st[:, 0].mean(), st[:, 1].mean(), st[:, 2].mean(), st[0, 3], len(st)))
    vox = Dz * Dy * Dx
    fl = 2.0 * Cin * Cout * 27 * B * vox
    gb = B * vox * ((2.0 if (st_layer and IN16) else 4.0) * Cin + (2.0 if (st_layer and OUT16) else 4.0) * Cout)
    print('%-7s cfg%d Cin %3d Cout %3d %s: %7.3f ms  %7.1f TFLOP/s  %6.0f GB/s algorithmic (%.2f of 8 TB/s)' % (name, cfg, Cin, Cout, 'x'.join(map(str, (Dz, Dy, Dx))), ms, fl / ms / 1e9, gb / ms / 1e6, gb / ms / 8e9))
print('TOTAL %.3f ms' % tot)
