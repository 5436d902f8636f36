"""The 3-D experiment planner behind `nnUNet_plan_and_preprocess` (/root/reference/deepwmh/pipeline/DCNN_multistage.py:281,
487,685 run it with the default `ExperimentPlanner3D_v21`).  Restated from upstream nnU-Net v1 [not in tree:
experiment_planning/experiment_planner_baseline_3DUNet_v21.py, experiment_planning/common_utils.py::get_pool_and_conv_props,
network_architecture/generic_UNet.py::compute_approx_vram_consumption]; parity unpinned like everything on this path.

What matters for DeepWMH: thick-slice FLAIR (anisotropy > 3, which /root/reference/deepwmh/analysis/image_ops.py:354-359
treats as a normal case) gets the low-resolution axis moved to the front (`transpose_forward`), [1,3,3] kernels and [1,2,2]
pooling there until the spacings are within a factor 2 of each other, and the 10th percentile of that axis' spacings as its
target spacing.  The single 3d_fullres stage only (DeepWMH never uses the cascade)."""
from copy import deepcopy

import numpy as np

ANISO_THRESHOLD = 3.0                 # ExperimentPlanner.anisotropy_threshold
MIN_EDGE = 4                          # unet_featuremap_min_edge_length
MAX_POOL = 999                        # unet_max_numpool
BASE_FEATURES = 32                    # ExperimentPlanner3D_v21.unet_base_num_features
MAX_FEATURES = 320                    # unet_max_num_filters
REF_VOXELS = 520000000 * BASE_FEATURES / 30    # Generic_UNet.use_this_for_batch_size_computation_3D, scaled from 30 features
DEFAULT_BATCH = 2                     # Generic_UNet.DEFAULT_BATCH_SIZE_3D
MIN_BATCH = 2                         # unet_min_batch_size
BATCH_COVERS = 0.05                   # batch_size_covers_max_percent_of_dataset


def target_spacing_v21(spacings, sizes):
    """Median spacing; for an anisotropic dataset (worst axis spacing > 3 x the others AND 3 x fewer voxels along it) the 10th
    percentile of the worst axis' spacings, not below the largest of the other axes."""
    spacings, sizes = np.vstack(spacings).astype(float), np.vstack(sizes).astype(float)
    target = np.percentile(spacings, 50, 0)
    tsize = np.percentile(sizes, 50, 0)
    worst = int(np.argmax(target))
    others = [i for i in range(3) if i != worst]
    o_sp, o_sz = [target[i] for i in others], [tsize[i] for i in others]
    if target[worst] > ANISO_THRESHOLD * max(o_sp) and tsize[worst] * ANISO_THRESHOLD < min(o_sz):
        t = np.percentile(spacings[:, worst], 10)
        if t < max(o_sp):
            t = max(max(o_sp), t) + 1e-5
        target[worst] = t
    return target


def transpose_for(target_spacing):
    """The axis with the largest spacing goes first (networks never pool / convolve across it before the others catch up)."""
    worst = int(np.argmax(target_spacing))
    tf = [worst] + [i for i in range(3) if i != worst]
    return tf, [int(np.argwhere(np.array(tf) == i)[0][0]) for i in range(3)]


def get_pool_and_conv_props(spacing, patch_size, min_feature_map_size=MIN_EDGE, max_numpool=MAX_POOL, in_plane_together=False):
    """common_utils.get_pool_and_conv_props: pool an axis while its spacing is within a factor 2 of the finest one and its extent
    stays >= 2 * min size; 3-wide kernels along the largest group of axes whose spacings are within a factor 2 of each other.
    in_plane_together (not upstream; `plan_stage`'s supported-plan constraint): stop pooling as soon as the two in-plane axes
    cannot both be pooled, so every pooling is [2,2,2] or [1,2,2]."""
    dim = len(spacing)
    cur_sp, cur_sz = [float(s) for s in spacing], [float(s) for s in patch_size]
    pools, kernels, num_pool = [], [], [0] * dim
    while True:
        mn = min(cur_sp)
        valid = [i for i in range(dim) if cur_sp[i] / mn < 2]
        axes = []
        for a in range(dim):
            partners = [i for i in range(dim) if cur_sp[i] / cur_sp[a] < 2 and cur_sp[a] / cur_sp[i] < 2]
            if len(partners) > len(axes):
                axes = partners
        kernel = [3 if i in axes else 1 for i in range(dim)]
        valid = [i for i in valid if cur_sz[i] >= 2 * min_feature_map_size]
        valid = [i for i in valid if num_pool[i] < max_numpool]
        if not valid or (in_plane_together and not (1 in valid and 2 in valid)):
            break
        pool = [1] * dim
        for v in valid:
            pool[v] = 2
            num_pool[v] += 1
            cur_sp[v] *= 2
            cur_sz[v] = np.ceil(cur_sz[v] / 2)
        pools.append(pool)
        kernels.append(kernel)
    div = [2 ** n for n in num_pool]
    patch = [int(np.ceil(p / d) * d) if p % d else int(p) for p, d in zip(patch_size, div)]      # pad_shape
    kernels.append([3] * dim)                                                                  # bottleneck
    return num_pool, pools, kernels, patch, div


def approx_vram(patch, num_pool_per_axis, base, max_feat, nmod, nclass, pools, conv_per_stage=2):
    """Generic_UNet.compute_approx_vram_consumption (feature-map voxels, deep_supervision=False)."""
    npool = len(pools)
    m = np.array(patch, dtype=np.int64)
    tmp = np.int64((conv_per_stage * 2 + 1) * np.prod(m) * base + nmod * np.prod(m) + nclass * np.prod(m))
    feat = base
    for p in range(npool):
        m = m // np.array(pools[p])
        feat = min(feat * 2, max_feat)
        blocks = (conv_per_stage * 2 + 1) if p < npool - 1 else conv_per_stage
        tmp += blocks * np.prod(m) * feat
    return int(tmp)


def poolings_supported(pools, kernels):
    """What libdeepwmh_amd.so implements (UNetPlan.check_supported): the two in-plane axes pool together, the first axis with
    them or not at all; 3-wide kernels in plane."""
    return all(list(p) in ([2, 2, 2], [1, 2, 2]) for p in pools) and all(list(k) in ([3, 3, 3], [1, 3, 3]) for k in kernels)


def plan_stage(current_spacing, original_spacing, original_shape, num_cases, nmod, nclass, log=None):
    """`_plan_stage` as upstream plans it; when that plan is outside what the HIP kernels implement (an in-plane axis that
    runs out of voxels one pooling before the other: [2,2,1], or the first axis pooling alone: [2,1,1]) the stage is
    re-planned with pooling stopped where the two in-plane axes can no longer both be halved -- a shallower network on the same
    budget, every pooling [2,2,2] or [1,2,2].  The deviation from upstream is logged.  Raises ValueError if that still is not
    a supported plan (in-plane spacings a factor 2 or more apart)."""
    st = _plan_stage(current_spacing, original_spacing, original_shape, num_cases, nmod, nclass, False)
    if poolings_supported(st['pool_op_kernel_sizes'], st['conv_kernel_sizes']):
        return st
    first = st
    st = _plan_stage(current_spacing, original_spacing, original_shape, num_cases, nmod, nclass, True)
    if st['pool_op_kernel_sizes'] and poolings_supported(st['pool_op_kernel_sizes'], st['conv_kernel_sizes']):
        if log:
            log('WARNING: upstream\'s plan (patch %s, poolings %s) is outside the HIP kernels\' repertoire; re-planned with the in-plane '
                'axes pooled together: patch %s, poolings %s' % ([int(v) for v in first['patch_size']], first['pool_op_kernel_sizes'],
                                                                [int(v) for v in st['patch_size']], st['pool_op_kernel_sizes']))
        # recorded IN the plans (and so in every checkpoint's plans): models of this stage are not interchangeable with upstream's
        st['deviates_from_upstream'] = dict(reason='in-plane axes pooled together (HIP kernels implement [2,2,2] / [1,2,2] poolings)',
                                            upstream_pool_op_kernel_sizes=[[int(v) for v in p] for p in first['pool_op_kernel_sizes']],
                                            upstream_patch_size=[int(v) for v in first['patch_size']],
                                            upstream_batch_size=int(first['batch_size']))
        return st
    raise ValueError('no plan with [2,2,2] / [1,2,2] poolings for spacing %s, median shape %s (upstream plans %s)'
                     % (np.asarray(current_spacing).tolist(), np.asarray(first['median_patient_size_in_voxels']).tolist(),
                        first['pool_op_kernel_sizes']))


def _plan_stage(current_spacing, original_spacing, original_shape, num_cases, nmod, nclass, together):
    """ExperimentPlanner3D_v21.get_properties_for_stage (spacing / shape already in the transposed axis order).
    nclass = number of foreground classes + 1."""
    current_spacing = np.asarray(current_spacing, float)
    median = np.round(np.asarray(original_spacing, float) / current_spacing * np.asarray(original_shape)).astype(int)
    dataset_voxels = float(np.prod(median, dtype=np.float64)) * num_cases
    p = 1.0 / current_spacing
    p /= p.mean()
    p *= 1.0 / min(p) * 512
    p = np.round(p).astype(int)
    p = [min(int(i), int(j)) for i, j in zip(p, median)]
    npa, pools, kernels, shp, div = get_pool_and_conv_props(current_spacing, p, MIN_EDGE, MAX_POOL, together)
    here = approx_vram(shp, npa, BASE_FEATURES, MAX_FEATURES, nmod, nclass, pools)
    while here > REF_VOXELS:
        ax = int(np.argsort(np.array(shp) / median)[-1])
        tmp = deepcopy(shp)
        tmp[ax] -= div[ax]
        _, _, _, _, div_new = get_pool_and_conv_props(current_spacing, tmp, MIN_EDGE, MAX_POOL, together)
        shp[ax] -= div_new[ax]
        npa, pools, kernels, shp, div = get_pool_and_conv_props(current_spacing, shp, MIN_EDGE, MAX_POOL, together)
        here = approx_vram(shp, npa, BASE_FEATURES, MAX_FEATURES, nmod, nclass, pools)
    batch = int(np.floor(max(REF_VOXELS / here, 1) * DEFAULT_BATCH))
    max_batch = max(int(np.round(BATCH_COVERS * dataset_voxels / float(np.prod(shp, dtype=np.float64)))), MIN_BATCH)
    batch = max(1, min(batch, max_batch))
    return dict(batch_size=batch, num_pool_per_axis=npa, patch_size=np.array(shp), median_patient_size_in_voxels=median,
                current_spacing=current_spacing, original_spacing=np.asarray(original_spacing, float),
                do_dummy_2D_data_aug=bool(max(shp) / shp[0] > ANISO_THRESHOLD), pool_op_kernel_sizes=pools, conv_kernel_sizes=kernels)


def plan_3d_fullres(spacings, sizes, nmod, nclass, log=None):
    """spacings / sizes: per training case, after cropping, in the image's own axis order.  Returns
    (target_spacing [image order], transpose_forward, transpose_backward, stage dict [transposed order])."""
    target = target_spacing_v21(spacings, sizes)
    new_shapes = [np.asarray(sp, float) / target * np.asarray(sz, float) for sp, sz in zip(spacings, sizes)]
    tf, tb = transpose_for(target)
    median = np.median(np.vstack(new_shapes), 0)
    stage = plan_stage(target[tf], target[tf], median[tf], len(spacings), nmod, nclass, log)
    return target, tf, tb, stage
