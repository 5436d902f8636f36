"""GPU parity of the whole-volume passes either side of the network against the oracle:
connected components / remove_sparks (reference deepwmh/analysis/image_ops.py:325-367), fill-holes, crop + z-score,
un-crop.  Integer results must be bit-exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def lib():
    from deepwmh_amd import _lib
    return _lib.load()


def P(t):
    from deepwmh_amd import _lib
    return _lib.ptr(t)


def check(rc, what=''):
    from deepwmh_amd import _lib
    return _lib.check(rc, what)


def _masks():
    rng = np.random.default_rng(5)
    out = []
    for shape, dens in [((7, 9, 11), 0.3), ((1, 1, 1), 1.0), ((1, 17, 5), 0.5), ((16, 16, 16), 0.05), ((33, 20, 27), 0.55),
                        ((40, 48, 40), 0.15), ((5, 1, 64), 0.7)]:
        out.append((rng.random(shape) < dens).astype(np.uint8))
    out.append(np.zeros((6, 5, 4), np.uint8))
    out.append(np.ones((6, 5, 4), np.uint8))
    snake = np.zeros((12, 12, 12), np.uint8)            # one long winding component: deep union-find chains
    for z in range(12):
        snake[z, :, 0 if z % 2 else 11] = 1
        snake[z, 0 if (z // 2) % 2 else 11, :] = 1
    out.append(snake)
    return out


def test_cc_label_partition_matches_scipy(lib):
    from oracle import postprocess as opp
    for m in _masks():
        md = torch.from_numpy(m).cuda()
        lab = torch.full(m.shape, -7, dtype=torch.int32, device='cuda')
        check(lib.wmh_cc_label(P(md), P(lab), *m.shape, None), 'cc_label')
        torch.cuda.synchronize()
        assert np.array_equal(lab.cpu().numpy().astype(np.int64), opp.components_partition(m)), m.shape


@pytest.mark.parametrize('min_volume', [2, 3, 5, 27])
def test_remove_sparks_matches_reference_semantics(lib, min_volume):
    from deepwmh_amd import postprocess as pp
    from oracle import postprocess as opp
    for m in _masks():
        got = pp.remove_sparks(m.astype(np.float32), min_volume=min_volume)
        ref = opp.remove_sparks(m.astype(np.float32), min_volume=min_volume)
        assert got.dtype == ref.dtype and np.array_equal(got, ref), (m.shape, min_volume)


def test_remove_3mm_sparks_voxel_size_rules(lib):
    from deepwmh_amd import postprocess as pp
    from oracle import postprocess as opp
    m = _masks()[5]
    for vs in ([1.0, 1.0, 1.0], [0.5, 0.5, 0.5], [0.9, 0.9, 3.0], [2.0, 2.0, 2.0], [0.43, 0.43, 1.2]):
        assert np.array_equal(pp.remove_3mm_sparks(m, vs), opp.remove_3mm_sparks(m, vs)), vs
    with pytest.raises(RuntimeError):
        pp.remove_3mm_sparks(m, (1.0, 1.0, 1.0))          # the reference insists on a list (image_ops.py:351-352)


def test_fill_holes_matches_scipy(lib):
    from oracle import postprocess as opp
    rng = np.random.default_rng(9)
    cases = _masks()
    shell = np.zeros((20, 22, 24), np.uint8); shell[3:17, 4:18, 5:19] = 1; shell[6:14, 7:15, 8:16] = 0      # closed cavity
    leak = shell.copy(); leak[10, 10, 16:24] = 0                                                            # cavity with a tunnel out
    cases += [shell, leak, (rng.random((30, 30, 30)) < 0.8).astype(np.uint8)]
    for m in cases:
        md = torch.from_numpy(m).cuda()
        out = torch.empty_like(md)
        lab = torch.empty(m.size, dtype=torch.int32, device='cuda'); fl = torch.empty_like(lab)
        check(lib.wmh_fill_holes(P(md), P(out), P(lab), P(fl), *m.shape, None), 'fill_holes')
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy().astype(bool), opp.fill_holes(m)), m.shape


def test_full_size_volume_remove_sparks_exact(lib):
    from deepwmh_amd import postprocess as pp
    from deepwmh_amd.synthetic import synthetic_flair
    from oracle import postprocess as opp
    v = synthetic_flair((160, 192, 160), 3)[0]
    m = (v > 1.2).astype(np.float32)
    got = pp.remove_3mm_sparks(m, [1.0, 1.0, 1.0])
    ref = opp.remove_3mm_sparks(m, [1.0, 1.0, 1.0])
    assert np.array_equal(got, ref) and 0 < ref.sum() < m.sum()
    # idempotent, and a checksum independent of scheduling
    assert np.array_equal(pp.remove_3mm_sparks(got, [1.0, 1.0, 1.0]), got)


@pytest.mark.parametrize('use_mask', [True, False])
def test_preprocess_case_device_matches_host(lib, use_mask):
    from deepwmh_amd import checkpoint as ck, preprocess
    from oracle import preprocess as opre
    from deepwmh_amd.plan import tiny_plan
    plans = ck.make_plans(tiny_plan(), spacing=(1.0, 1.0, 1.0), use_mask_for_norm=use_mask)
    rng = np.random.default_rng(3)
    vol = np.zeros((46, 52, 44), np.float32)
    vol[3:43, 2:50, 5:42] = rng.standard_normal((40, 48, 37)).astype(np.float32) * 40 + 200
    vol[10:14, 20:25, 20:26] = 0                             # a hole inside the head: filled by the mask, kept in the statistics
    vol[0, 0, 0] = 0
    ref, props_ref = opre.preprocess_case(vol, (1.0, 1.0, 1.0), plans)
    got, props = preprocess.preprocess_case_device(vol, (1.0, 1.0, 1.0), plans)
    torch.cuda.synchronize()
    assert props['bbox'] == props_ref['bbox'] and tuple(props['shape_after_crop']) == tuple(props_ref['shape_after_crop'])
    assert tuple(got.shape) == ref.shape
    assert float(np.abs(got.cpu().numpy() - ref).max()) <= 2e-5 * (1 + float(np.abs(ref).max()))
    # un-crop: label map back in the input's geometry
    seg = (torch.rand(props['shape_after_crop'], device='cuda') < 0.3).to(torch.uint8)
    full = preprocess.export_segmentation_device(seg, props)
    prob = np.stack([1 - seg.cpu().numpy().astype(np.float32), seg.cpu().numpy().astype(np.float32)])
    full_ref, _ = opre.export_prediction(prob, props_ref)
    assert np.array_equal(full, full_ref)


def test_preprocess_device_all_zero_volume(lib):
    from deepwmh_amd import checkpoint as ck, preprocess
    from oracle import preprocess as opre
    from deepwmh_amd.plan import tiny_plan
    plans = ck.make_plans(tiny_plan(), spacing=(1.0, 1.0, 1.0), use_mask_for_norm=True)
    vol = np.zeros((8, 9, 10), np.float32)
    got, props = preprocess.preprocess_case_device(vol, (1.0, 1.0, 1.0), plans)
    ref, props_ref = opre.preprocess_case(vol, (1.0, 1.0, 1.0), plans)
    assert props['bbox'] == props_ref['bbox'] and np.array_equal(got.cpu().numpy(), ref)


@pytest.mark.parametrize('order', [0, 1, 3])
def test_resample_matches_scipy_zoom(lib, order):
    """wmh_resample against the host restatement of skimage's resize (scipy zoom, float64): up- and down-sampling,
    anisotropic factors, an axis left untouched, tiny extents."""
    from deepwmh_amd import preprocess
    from oracle import preprocess as opre
    rng = np.random.default_rng(11)
    for shp, new in [((20, 24, 18), (31, 24, 13)), ((9, 10, 11), (18, 25, 7)), ((3, 40, 5), (4, 33, 5)), ((16, 16, 16), (16, 16, 16)),
                     ((2, 3, 30), (5, 2, 41))]:
        x = (rng.standard_normal((2,) + shp) * 30 + 100).astype(np.float32)
        got = preprocess._resample_device(torch.from_numpy(x).cuda(), new, order).cpu().numpy()
        ref = np.stack([opre._resize(x[c].astype(np.float64), new, order).astype(np.float32) for c in range(2)]) \
            if tuple(new) != shp else x
        if tuple(new) == shp and order == 3:
            tol = 2e-5 * float(np.abs(x).max())          # identity zoom still runs the spline round trip on the device
        else:
            tol = 2e-6 * float(np.abs(ref).max()) if order else 0.0
        assert got.shape == ref.shape and float(np.abs(got - ref).max()) <= tol, (shp, new, order, float(np.abs(got - ref).max()))


@pytest.mark.parametrize('spacing,tf', [((0.9, 0.9, 1.2), [0, 1, 2]), ((1.3, 0.8, 1.0), [0, 1, 2]), ((3.0, 0.9, 0.9), [0, 1, 2]),
                                        ((1.0, 1.0, 1.0), [1, 2, 0]),
                                        # thick slices (anisotropy > 3): nnU-Net resamples the low-resolution axis on its own
                                        ((5.0, 1.1, 1.1), [0, 1, 2]), ((0.9, 0.9, 4.0), [0, 1, 2]), ((1.0, 0.8, 3.6), [2, 0, 1])])
def test_preprocess_and_export_device_with_resampling(lib, spacing, tf):
    from deepwmh_amd import checkpoint as ck, preprocess
    from oracle import preprocess as opre
    from deepwmh_amd.plan import tiny_plan
    plans = ck.make_plans(tiny_plan(), spacing=(1.0, 1.0, 1.0), use_mask_for_norm=True)
    plans['transpose_forward'] = tf
    plans['transpose_backward'] = [tf.index(i) for i in range(3)]
    rng = np.random.default_rng(13)
    from scipy.ndimage import gaussian_filter
    vol = np.zeros((40, 46, 38), np.float32)
    vol[4:37, 3:44, 5:35] = (gaussian_filter(rng.standard_normal((33, 41, 30)), 1.5) * 200 + 300).astype(np.float32)
    vol[15:18, 20:24, 18:22] = 0
    ref, props_ref = opre.preprocess_case(vol, spacing, plans)
    got, props = preprocess.preprocess_case_device(vol, spacing, plans)
    assert props['bbox'] == props_ref['bbox'] and tuple(props['shape_after_crop']) == tuple(props_ref['shape_after_crop'])
    assert tuple(got.shape) == ref.shape
    assert float(np.abs(got.cpu().numpy() - ref).max()) <= 5e-5 * (1 + float(np.abs(ref).max()))
    # export: probabilities in network geometry -> label map + P(background) in the input's geometry
    K = 2
    logits = torch.from_numpy(gaussian_filter(rng.standard_normal((K,) + ref.shape[1:]), (0, 2, 2, 2)).astype(np.float32) * 8).cuda()
    prob = torch.softmax(logits, 0).contiguous()
    seg = prob.argmax(0).to(torch.uint8)
    full, p0 = preprocess.export_prediction_device(seg, prob, props, True)
    full_ref, pfull_ref = opre.export_prediction(prob.cpu().numpy(), props_ref)
    assert float(np.abs(p0 - pfull_ref[0]).max()) <= 2e-6
    sure = np.abs(pfull_ref[1] - pfull_ref[0]) > 1e-5
    assert full.shape == full_ref.shape and int((full != full_ref)[sure].sum()) == 0


@pytest.mark.parametrize('spacing,target,use_mask', [((1.0, 1.0, 1.0), (1.0, 1.0, 1.0), True), ((0.8, 1.0, 1.3), (1.0, 1.0, 1.0), True),
                                                    ((1.0, 1.0, 1.0), (1.5, 0.75, 1.2), False), ((4.0, 0.9, 0.9), (3.0, 1.0, 1.0), True)])
def test_training_case_preprocessing_device_matches_host(lib, spacing, target, use_mask):
    """`nnUNet_plan_and_preprocess` per case (crop, order-3 data / label-wise order-1 segmentation resampling incl. the thick-slice
    branch, z-score) on the device against the host restatement in oracle/preprocess.py."""
    from deepwmh_amd import preprocess
    from oracle import preprocess as opre
    from scipy.ndimage import gaussian_filter
    rng = np.random.default_rng(17)
    vol = np.zeros((1, 30, 44, 40), np.float32)
    vol[0, 3:28, 4:41, 5:36] = (gaussian_filter(rng.standard_normal((25, 37, 31)), 1.5) * 200 + 300).astype(np.float32)
    vol[0, 10:13, 20:24, 18:22] = 0                      # a hole: filled by the mask
    vol[0, 3:9, 4:12, 5:36] = 0                          # a notch open to the outside: stays outside the mask (segmentation -1)
    lab = np.zeros((30, 44, 40), np.float32)
    lab[9:14, 13:22, 9:20] = 1
    lab[18:22, 25:33, 22:30] = 1
    cropped, seg, nz, bbox = preprocess.crop_training_case_device(vol, lab)
    c_ref, nz_ref, bbox_ref = opre.crop_to_nonzero(vol)
    assert bbox == bbox_ref and np.array_equal(cropped, c_ref) and np.array_equal(nz, nz_ref)
    seg_ref = lab[tuple(slice(a, b) for a, b in bbox)].astype(np.int8)
    seg_ref[~nz_ref] = -1
    assert np.array_equal(seg, seg_ref) and (seg == -1).any()
    got, seg_r = preprocess.resample_normalize_training_case_device(cropped, seg, nz, spacing, target, use_mask)
    res = opre.resample_to_spacing(c_ref, spacing, target, 3)
    if res.shape[1:] != seg_ref.shape:
        sr = opre.resample_segmentation(seg_ref, res.shape[1:], opre.separate_z_axis(np.asarray(spacing, float), np.asarray(target, float)))
        nzr = sr >= 0
    else:
        sr, nzr = seg_ref, nz_ref
    ref = opre.normalize_noncT(res, nzr, use_mask)
    assert got.shape == ref.shape and seg_r.shape == sr.shape and seg_r.dtype == np.int8
    # labels: identical except where a label's linear indicator sits on the 0.5 threshold to rounding (none expected; allow 1e-4)
    assert float((seg_r != sr).mean()) <= 1e-4, float((seg_r != sr).mean())
    same = seg_r == sr
    mask_same = (seg_r >= 0) == (sr >= 0)
    assert float(np.abs(got - ref)[:, mask_same & same].max()) <= 5e-5 * (1 + float(np.abs(ref).max()))
