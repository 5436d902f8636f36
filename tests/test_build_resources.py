"""Register / scratch budget of the built kernels (CPU: reads what deepwmh_amd/build.py recorded from hipcc's kernel-resource-usage remarks).

Round 4's lesson: compiling the cycle stamps out of the strided pair kernel changed nothing in its source semantics and let the compiler
hoist the loader's coordinate divisions out of the chunk loop -- 60 spilled registers, every reload waiting for the whole prefetch, the
kernel 1.6x slower, the bench 7 % down -- and no test noticed.  Now the CPU suite does: the hot kernels must not use scratch at all, and no
kernel may start to (the few that always have are listed with their budgets)."""
import json
import os

import pytest

from deepwmh_amd import build

# kernels that have used scratch since before round 4, with the bytes per lane they may use (rarely launched, or by design)
ALLOWED = {
    'median_big_kernel': 4112,                 # per-thread window of the large-extent median (stage1_kernels.hip): scratch by design
    'cl_conv3s2_kernelILi2E': 24,              # configs[2], strided 32 -> 64 layer
    'conv3d_k3_wgrad_pipe_kernelILi1E': 12,    # exact-fp32 weight gradient (stride 1), outside its loop
    'ELb0EEv8ConvArgs': 80,                    # one-role bf16 (NOT pair: ...Lb0E) instances of conv3d_k3_bf16_kernel with 32- / 16-voxel rows: what the
                                               # opt-in bf16 mode falls back to when a row is not a multiple of 4 voxels (the wave-specialised kernel loads
                                               # aligned quads); 52 / 76 bytes.  Their 8-wave variants (300-416 bytes) exist in the diagnostic library only
}
# hot instances that may keep ONE or two values in scratch, outside their loops: the 16- and 8-voxel-row geometries of the strided pair kernel reload a
# store offset in the epilogue's ragged-tile branch (a 768-thread workgroup leaves a wave 168 registers, two of them hold spilled SGPRs)
HOT_BUDGET = {'conv3d_k3s2_f16x2_ws_kernelI5S2GeoILi2ELi4ELi16E': 8, 'conv3d_k3s2_f16x2_ws_kernelI5S2GeoILi4ELi4ELi8E': 8}
HOT = ['conv3d_k3_bf16_ws_kernel', 'conv3d_k3s2_f16x2_ws_kernel', 'conv3d_k3_ws_kernel', 'conv3d_k3_wgrad_f16x2_kernel', '15cl_conv3_kernel',
       'convT_k2s2_f16x2_kernel', 'conv3d_k3s2_dgrad_f16x2_kernel', 'seghead_softmax_accum', 'conv3d_first_kernel', 'conv3d_up_f16x2_kernel']


@pytest.fixture(scope='module')
def resources():
    path = build.LIB[:-3] + '.resources.json'
    if not os.path.exists(path):
        if not (os.path.isfile(build.HIPCC) and os.access(build.HIPCC, os.X_OK)):
            pytest.skip('%s is absent (git-ignored build product) and there is no hipcc here to produce it' % path)
        build.build_library(verbose=False)
    with open(path) as f:
        return json.load(f)


def test_every_hot_kernel_is_built_without_scratch(resources):
    for needle in HOT:
        hits = {k: v for k, v in resources.items() if needle in k}
        assert hits, 'no kernel named *%s* in the library' % needle
        for name, r in hits.items():
            budget = max([b for k, b in HOT_BUDGET.items() if k in name] or [0])
            assert r.get('scratch_bytes_per_lane', 0) <= budget, '%s spills %d bytes per lane (%d VGPRs)' % (name, r['scratch_bytes_per_lane'], r.get('vgprs', -1))


def test_no_kernel_outside_the_known_list_uses_scratch(resources):
    for name, r in resources.items():
        sc = r.get('scratch_bytes_per_lane', 0)
        if sc == 0:
            continue
        budget = max([b for k, b in list(ALLOWED.items()) + list(HOT_BUDGET.items()) if k in name and (k != 'ELb0EEv8ConvArgs' or 'conv3d_k3_bf16_kernel' in name)] or [0])
        assert sc <= budget, '%s uses %d bytes of scratch per lane (allowed: %d)' % (name, sc, budget)


def test_the_pair_kernels_keep_two_waves_per_simd(resources):
    """One 512-thread workgroup per CU = two waves per SIMD: at most 256 registers."""
    for name, r in resources.items():
        if 'conv3d_k3_bf16_ws_kernel' in name or 'conv3d_k3_ws_kernel' in name:
            assert r['vgprs'] <= 256 and r['waves_per_simd'] >= 2, (name, r)
