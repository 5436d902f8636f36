"""nnU-Net's separate-z resampling [preprocessing.resample_patient / resample_data_or_seg, not in tree], restated in
oracle/preprocess.py (host restatement; the device form in deepwmh_amd/preprocess.py is compared with it in tests/test_gpu_volume_ops.py) (thick-slice FLAIR -- the case /root/reference/deepwmh/analysis/image_ops.py:354-359
handles explicitly -- was resampled isotropically with a cubic spline along the slice axis)."""
import numpy as np
from scipy.ndimage import zoom

from deepwmh_amd import checkpoint as ck, preprocess as dpp
from oracle import preprocess as pp
from deepwmh_amd.plan import tiny_plan


def test_axis_rule():
    assert pp.separate_z_axis((5, 1, 1), (1, 1, 1)) == 0            # the case's own spacing decides first
    assert pp.separate_z_axis((1, 1, 1), (1, 1, 4)) == 2            # else the plan's
    assert pp.separate_z_axis((3, 1, 1), (1, 1, 1)) is None         # threshold is strictly greater than 3
    assert pp.separate_z_axis((1, 1, 1), (1, 1, 1)) is None
    for a, b in (((5, 1, 1), (1, 1, 1)), ((1, 1, 1), (1, 1, 4)), ((3, 1, 1), (1, 1, 1)), ((0.24, 1.25, 1.25), (1, 1, 1)), ((1, 6, 1), (2, 2, 2))):
        assert dpp.separate_z_axis(a, b) == pp.separate_z_axis(a, b)      # the product's copy of the rule
    assert pp.separate_z_axis((0.24, 1.25, 1.25), (1, 1, 1)) is None   # two low-resolution axes: joint resampling


def test_slices_are_resized_in_plane_and_repeated_along_the_axis():
    rng = np.random.default_rng(0)
    v = rng.standard_normal((4, 10, 12))
    out = pp.resample_volume(v, (20, 15, 18), 3, axis=0)
    assert out.shape == (20, 15, 18)
    planes = [zoom(v[k], (1.5, 1.5), order=3, mode='nearest', grid_mode=True) for k in range(4)]
    # nearest neighbour along z at (o + 0.5) * 4 / 20 - 0.5: five output slices per input slice
    for o in range(20):
        assert np.array_equal(out[o], planes[o // 5])
    joint = pp.resample_volume(v, (20, 15, 18), 3, axis=None)
    assert float(np.abs(joint - out).max()) > 0.1                      # and that is not what a cubic spline along z gives


def test_preprocess_and_export_round_trip_with_thick_slices():
    plans = ck.make_plans(tiny_plan(), spacing=(1.0, 1.0, 1.0), use_mask_for_norm=True)
    rng = np.random.default_rng(1)
    vol = np.zeros((8, 40, 36), np.float32)
    vol[1:7, 4:36, 3:33] = rng.uniform(100, 300, (6, 32, 30)).astype(np.float32)
    data, props = pp.preprocess_case(vol, (5.0, 1.0, 1.0), plans)
    assert data.shape == (1, 30, 32, 30)                                # 6 slices of 5 mm -> 30 of 1 mm
    assert props['original_spacing'] == (5.0, 1.0, 1.0)
    for o in range(30):                                                  # every 1 mm slice is a copy of its 5 mm slice
        assert np.array_equal(data[0, o], data[0, 5 * (o // 5)])
    prob = np.zeros((2,) + data.shape[1:], np.float32)
    prob[1, 10:20] = 0.9
    prob[0] = 1 - prob[1]
    seg, pfull = pp.export_prediction(prob, props)
    assert seg.shape == vol.shape and pfull.shape == (2,) + vol.shape
    assert seg[1:7].reshape(6, -1).max(1).tolist() == [0, 0, 1, 1, 0, 0]   # order 0 along z: 1 mm slices 10..19 = 5 mm slices 2, 3
