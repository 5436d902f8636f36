"""CPU tests: the oracle's remove_sparks / fill_holes against brute-force flood fills, and the host-side voxel-size rule
of remove_3mm_sparks (reference deepwmh/analysis/image_ops.py:347-367)."""
import numpy as np
import pytest


def _bfs_components(mask):
    mask = mask.astype(bool)
    lab = -np.ones(mask.shape, dtype=np.int64)
    comps = []
    for start in zip(*np.nonzero(mask)):
        if lab[start] >= 0:
            continue
        stack, members = [start], []
        lab[start] = len(comps)
        while stack:
            z, y, x = stack.pop()
            members.append((z, y, x))
            for dz, dy, dx in ((1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)):
                q = (z + dz, y + dy, x + dx)
                if all(0 <= q[a] < mask.shape[a] for a in range(3)) and mask[q] and lab[q] < 0:
                    lab[q] = len(comps)
                    stack.append(q)
        comps.append(members)
    return comps


def test_oracle_remove_sparks_against_flood_fill():
    from oracle import postprocess as opp
    rng = np.random.default_rng(1)
    for shape, dens in [((6, 7, 8), 0.3), ((10, 4, 9), 0.5), ((3, 3, 3), 0.9)]:
        m = (rng.random(shape) < dens).astype(np.float32)
        for mv in (2, 3, 6):
            ref = np.zeros(shape, dtype='int')
            for members in _bfs_components(m > 0.5):
                if len(members) >= mv:
                    for q in members:
                        ref[q] = 1
            assert np.array_equal(opp.remove_sparks(m, mv), ref)


def test_oracle_partition_is_min_index():
    from oracle import postprocess as opp
    rng = np.random.default_rng(2)
    m = (rng.random((5, 6, 7)) < 0.4)
    part = opp.components_partition(m)
    for members in _bfs_components(m):
        idx = [np.ravel_multi_index(q, m.shape) for q in members]
        assert all(part[q] == min(idx) for q in members)
    assert np.all(part[~m] == -1)


def test_oracle_fill_holes_against_flood_fill():
    from oracle import postprocess as opp
    rng = np.random.default_rng(4)
    m = (rng.random((9, 9, 9)) < 0.7)
    ref = m.copy()
    for members in _bfs_components(~m):
        touches = any(q[a] in (0, m.shape[a] - 1) for q in members for a in range(3))
        if not touches:
            for q in members:
                ref[q] = True
    assert np.array_equal(opp.fill_holes(m), ref)


def test_min_volume_rule():
    from deepwmh_amd.postprocess import min_volume_for
    assert min_volume_for([1.0, 1.0, 1.0]) == 3
    assert min_volume_for([0.5, 0.5, 0.5]) == 24
    assert min_volume_for([2.0, 2.0, 2.0]) == 2          # int(round(0.375)) = 0 -> floor of 2
    assert min_volume_for([0.9, 0.9, 3.0]) == 3          # anisotropy 3.33 > 3: three voxels
    assert min_volume_for([1.0, 1.0, 3.0]) == 2          # anisotropy exactly 3 is "thin": round(1.0) = 1 -> 2
    with pytest.raises(RuntimeError):
        min_volume_for((1.0, 1.0, 1.0))
