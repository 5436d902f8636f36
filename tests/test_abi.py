"""CPU tests of the C-ABI library: it loads, exports every symbol include/deepwmh_amd.h declares,
and its host-side entry points (configuration, weight packing, argument checks) behave."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    from deepwmh_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib.load()


def test_header_symbols_all_exported_and_bound(lib):
    from deepwmh_amd import _lib
    hdr = open(os.path.join(ROOT, 'include', 'deepwmh_amd.h')).read()
    declared = set(re.findall(r'\b(wmh_[a-z0-9_]+)\s*\(', hdr))
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(lib, name), 'library does not export %s' % name
    assert declared == set(_lib.SIGNATURES), 'ctypes table and header disagree'
    assert lib.wmh_abi_version() == 6                   # bumped in round 6 (it had stayed 1 while entry points were added every round)
    # measurement aids live in the diagnostic library only (include/deepwmh_amd_diag.h): the product exports none of them
    dhdr = open(os.path.join(ROOT, 'include', 'deepwmh_amd_diag.h')).read()
    diag_declared = set(re.findall(r'\b(wmh_[a-z0-9_]+)\s*\(', dhdr))
    assert diag_declared == set(_lib.DIAG_SIGNATURES) and not (diag_declared & declared)
    for name in diag_declared:
        assert not hasattr(lib, name), 'product library exports the diagnostic entry point %s' % name
    if os.path.exists(_lib.DIAG_LIB_PATH):
        dlib = C.CDLL(_lib.DIAG_LIB_PATH)
        for name in declared | diag_declared:
            assert hasattr(dlib, name), 'diagnostic library does not export %s' % name


def test_config_selection(lib):
    assert lib.wmh_conv3d_k3_config(32, 32, 128, 128, 128, 1) == 0
    assert lib.wmh_conv3d_k3_config(32, 64, 128, 128, 128, 2) == 1
    assert lib.wmh_conv3d_k3_config(256, 256, 16, 16, 16, 1) == 2
    assert lib.wmh_conv3d_k3_config(128, 256, 32, 32, 32, 2) == 3
    assert lib.wmh_conv3d_k3_config(320, 320, 8, 8, 8, 1) == 4
    assert lib.wmh_conv3d_k3_config(320, 320, 4, 4, 4, 1) == 6
    assert lib.wmh_conv3d_k3_config(320, 320, 8, 8, 8, 2) == 7
    assert lib.wmh_conv3d_k3_config(32, 32, 128, 128, 128, 3) < 0
    assert b'bad shape' in lib.wmh_last_error()


def test_pack_weights_layout(lib):
    # fragment-major image: [G][chunk][tap][lane][CK/2], lane = (cin parity, cout row)
    cin, cout, cfg = 6, 40, 0          # cfg 0: CK = 4 -> 2 chunks (2nd half-empty), G = 2 (2nd partial)
    w = np.random.default_rng(0).standard_normal((cout, cin, 3, 3, 3)).astype(np.float32)
    n = lib.wmh_conv3d_k3_packed_floats(cfg, cin, cout)
    assert n == 2 * 2 * 27 * 64 * 2
    wf = np.full(n, np.nan, np.float32)
    rc = lib.wmh_conv3d_k3_pack_weights(cfg, w.ctypes.data_as(C.c_void_p), cin, cout, wf.ctypes.data_as(C.c_void_p))
    assert rc == 0 and not np.isnan(wf).any()
    wf = wf.reshape(2, 2, 27, 1, 64, 2)     # [G][chunk][tap][NT][lane][CK/2], NT = 1 for cfg 0
    w27 = w.reshape(cout, cin, 27)
    for g, ch, tap, lane, p in [(0, 0, 0, 0, 0), (0, 1, 13, 37, 0), (1, 0, 26, 7, 1), (1, 1, 5, 63, 1), (0, 1, 2, 33, 1)]:
        co, ci = g * 32 + (lane & 31), ch * 4 + 2 * p + (lane >> 5)
        exp = w27[co, ci, tap] if (co < cout and ci < cin) else 0.0
        assert wf[g, ch, tap, 0, lane, p] == exp
    # cfg 2 has NT = 2: 64 output channels per group, [G=1][chunk][tap][2][64][2]
    n2 = lib.wmh_conv3d_k3_packed_floats(2, cin, cout)
    assert n2 == 1 * 2 * 27 * 2 * 64 * 2
    wf2 = np.empty(n2, np.float32)
    assert lib.wmh_conv3d_k3_pack_weights(2, w.ctypes.data_as(C.c_void_p), cin, cout, wf2.ctypes.data_as(C.c_void_p)) == 0
    wf2 = wf2.reshape(1, 2, 27, 2, 64, 2)
    for ch, tap, nt, lane, p in [(0, 3, 1, 5, 0), (1, 20, 0, 40, 0), (0, 26, 1, 20, 1)]:
        co, ci = nt * 32 + (lane & 31), ch * 4 + 2 * p + (lane >> 5)
        exp = w27[co, ci, tap] if (co < cout and ci < cin) else 0.0
        assert wf2[0, ch, tap, nt, lane, p] == exp
    assert np.isclose(np.abs(wf).sum(), np.abs(w).sum(), rtol=1e-5)   # nothing lost, padding is zero


def test_num_partials(lib):
    # cfg 0 tiles 4x8x32 voxels, 4 waves per tile -> 128^3: 32*16*4 tiles * 4
    assert lib.wmh_conv3d_k3_num_partials(0, 8, 128, 128, 128) == 32 * 16 * 4 * 4
    assert lib.wmh_conv3d_k3_num_partials(6, 8, 4, 4, 4) == 1


def test_argument_checks_fail_loudly(lib):
    from deepwmh_amd import _lib
    rc = lib.wmh_conv3d_k3_fwd(0, None, None, 1, None, None, 0, None, None, None, None, None, 1, 8, 8, 8, 8, 1, 0.01, None)
    assert rc == -1
    with pytest.raises(_lib.WMHError):
        _lib.check(rc, 'wmh_conv3d_k3_fwd')
    assert lib.wmh_seghead_softmax_accumulate(None, None, None, 2, None, None, 4, 8, 8, 8, 8, 8, 8, None, 1, None, 0.01, None) == -1


def test_engine_refuses_cpu(tiny_net):
    from deepwmh_amd.network import UNetEngine
    plan, _, sd = tiny_net
    with pytest.raises(RuntimeError):
        UNetEngine(plan, sd, device='cpu')


def test_f16x2_weight_image_is_a_two_term_split():
    """Host-side packing of the f16x2 mode (no GPU needed): every fragment holds the high fp16 parts of an output channel's 8 weights
    (scaled by 2^8) on lanes 0-31 and the low parts on lanes 32-63; hi + lo reproduces 256 w to 21-22 significant bits while the
    low part is a normal fp16 number, and to 2^-25 absolute below that; out-of-range weights are refused."""
    import ctypes as C
    import numpy as np
    from deepwmh_amd import _lib
    lib = _lib.load()
    Cin, Cout, cfg = 20, 40, 0                       # ragged: 3 chunks of 8 channels, 2 groups of 32 output channels
    assert lib.wmh_conv3d_k3_f16x2_supported(cfg) and not lib.wmh_conv3d_k3_f16x2_supported(8)
    rng = np.random.default_rng(0)
    w = (rng.standard_normal((Cout, Cin, 27)) * np.exp(rng.uniform(-9, 3, (Cout, Cin, 27)))).astype(np.float32)
    w = np.clip(w, -200, 200)
    n = lib.wmh_conv3d_k3_packed_halfs_f16x2(cfg, Cin, Cout)
    assert n == 2 * 3 * 27 * 1 * 64 * 8
    wf = np.zeros(n, np.uint16)
    _lib.check(lib.wmh_conv3d_k3_pack_weights_f16x2(cfg, w.ctypes.data_as(C.c_void_p), Cin, Cout, wf.ctypes.data_as(C.c_void_p)), 'pack')
    img = wf.view(np.float16).astype(np.float64).reshape(2, 3, 27, 1, 64, 8)              # [g][chunk][tap][nt][lane][e]
    hi, lo = img[:, :, :, 0, :32, :], img[:, :, :, 0, 32:, :]
    full = np.zeros((2 * 32, 3 * 8, 27), np.float32)
    full[:Cout, :Cin] = w
    ref = 256.0 * full.reshape(2, 32, 3, 8, 27).transpose(0, 2, 4, 1, 3).astype(np.float64)   # [g][chunk][tap][cout lane][e]
    rec = hi + lo
    assert np.all(np.isfinite(rec))
    err = np.abs(rec - ref)
    assert np.all(err <= np.maximum(2.0 ** -21 * np.abs(ref), 2.0 ** -25))               # two fp16 terms (nearest, nearest)
    assert np.all(rec[ref == 0] == 0)                                                    # padding channels / rows are zero
    assert np.array_equal(hi, ref.astype(np.float16).astype(np.float64))                 # the high part is fp16(256 w)
    w[3, 4, 5] = 300.0                                                                   # 256 w beyond fp16's range
    with __import__('pytest').raises(_lib.WMHError):
        _lib.check(lib.wmh_conv3d_k3_pack_weights_f16x2(cfg, w.ctypes.data_as(C.c_void_p), Cin, Cout, wf.ctypes.data_as(C.c_void_p)), 'pack')


def test_wgrad_f16x2_plan_host_side(lib):
    """Host-side planning of the pair weight gradient (no GPU): which layers it serves, and the slab workspace it asks for
    (P slabs of Cout x Cin x 27 floats, P = the power of two that brings the grid of 32 co x 16 ci workgroups closest to 256 from below)."""
    sup, ws = lib.wmh_conv3d_k3_wgrad_f16x2_supported, lib.wmh_conv3d_k3_wgrad_f16x2_workspace_floats
    assert sup(32, 32, 2, 128, 128, 128, 1) == 1 and sup(256, 256, 2, 16, 16, 16, 1) == 1
    assert sup(320, 320, 2, 8, 8, 8, 1) == 1            # rows of 8 voxels (round 5)
    assert sup(320, 320, 2, 4, 4, 4, 1) == 0            # shorter rows: the fp32 kernel
    assert sup(32, 64, 2, 128, 128, 128, 2) == 1 and sup(128, 256, 2, 32, 32, 32, 2) == 1   # strided layers with output rows of >= 12 voxels (round 5)
    assert sup(256, 320, 2, 16, 16, 16, 2) == 0         # ... shorter ones: the fp32 kernel
    # 32 -> 64 at 2 x 128^3 (output 64^3): 2 x 2 channel groups -> P = 64
    assert lib.wmh_conv3d_k3s2_wgrad_f16x2_workspace_floats(32, 64, 2, 128, 128, 128) == 64 * 64 * 32 * 27
    assert ws(320, 320, 2, 4, 4, 4) == 0 and ws(320, 320, 2, 8, 8, 8) == 1 * 320 * 320 * 27
    # 32 -> 32 at 2 x 128^3: 1 x 2 channel groups -> P = 128
    assert ws(32, 32, 2, 128, 128, 128) == 128 * 32 * 32 * 27
    # 256 -> 128 at 2 x 32^3: 4 x 16 groups -> P = 4
    assert ws(256, 128, 2, 32, 32, 32) == 4 * 128 * 256 * 27
    # more groups than CUs: one slab
    assert ws(640, 320, 1, 16, 16, 16) == 1 * 320 * 640 * 27
    # a single tile: P cannot exceed the tile count
    assert ws(16, 32, 1, 2, 4, 32) == 1 * 32 * 16 * 27
    # argument checks fail loudly
    assert lib.wmh_conv3d_k3_wgrad_f16x2(None, None, 32, None, None, 0, None, None, None, 2, 128, 128, 128, 32, 0.01, None) == -1
