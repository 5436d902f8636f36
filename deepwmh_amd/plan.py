"""Network plan: the subset of an nnU-Net `plans.pkl` the hot path needs.

[nnU-Net, not in tree] `plans['plans_per_stage'][stage]` carries `patch_size`,
`pool_op_kernel_sizes`, `conv_kernel_sizes`, `batch_size`; top-level keys carry
`base_num_features`, `num_classes` (foreground count), `num_modalities`, `conv_per_stage`.
The reference pins this file's location, not its content:
/root/reference/deepwmh/pipeline/DCNN_multistage.py:608-631 and
/root/reference/deepwmh/main/install_model.py:91-106.
"""
from dataclasses import dataclass, field
from typing import List

import numpy as np


@dataclass
class ConvSpec:
    name: str          # state-dict prefix, e.g. 'conv_blocks_context.0.blocks.0'
    cin: int
    cout: int
    stride: int        # in-plane stride s (1 or 2)
    level: int         # resolution level (0 = full patch resolution)
    kz: int = 3        # kernel extent along the first spatial axis (1 for nnU-Net's anisotropic [1,3,3])
    sz: int = 0        # stride along the first spatial axis (0 = same as `stride`)

    def __post_init__(self):
        if self.sz == 0:
            self.sz = self.stride


@dataclass
class UNetPlan:
    num_input_channels: int = 1
    base_num_features: int = 32
    num_classes: int = 2                      # including background
    pool_op_kernel_sizes: List[List[int]] = field(default_factory=lambda: [[2, 2, 2]] * 5)
    conv_kernel_sizes: List[List[int]] = field(default_factory=lambda: [[3, 3, 3]] * 6)
    conv_per_stage: int = 2
    max_num_features: int = 320
    patch_size: List[int] = field(default_factory=lambda: [128, 128, 128])

    # ---- derived -------------------------------------------------------------------------
    @property
    def num_pool(self):
        return len(self.pool_op_kernel_sizes)

    def features(self):
        """Feature count per encoder stage incl. bottleneck: 32,64,128,256,320,320."""
        f, out = self.base_num_features, []
        for _ in range(self.num_pool + 1):
            out.append(f)
            f = min(int(np.round(f * 2)), self.max_num_features)
        return out

    def check_supported(self):
        """The HIP path implements nnU-Net's 3-D plans: kernels [3,3,3] or [1,3,3] (anisotropic data: no convolution
        along the thick-slice axis), pooling [2,2,2] or [1,2,2], two convolutions per stage."""
        for k in self.conv_kernel_sizes:
            if list(k) not in ([3, 3, 3], [1, 3, 3]):
                raise NotImplementedError("conv_kernel_sizes %s: [3,3,3] and [1,3,3] are implemented in HIP" % (k,))
        for p in self.pool_op_kernel_sizes:
            if list(p) not in ([2, 2, 2], [1, 2, 2]):
                raise NotImplementedError("pool_op_kernel_sizes %s: [2,2,2] and [1,2,2] are implemented in HIP" % (p,))
        if self.conv_per_stage != 2:
            raise NotImplementedError("conv_per_stage must be 2")
        for ax, s in enumerate(self.patch_size):
            div = int(np.prod([p[ax] for p in self.pool_op_kernel_sizes]))
            if s % div:
                raise ValueError("patch_size %s not divisible by the pooling along axis %d" % (self.patch_size, ax))

    @property
    def isotropic(self):
        return all(list(k) == [3, 3, 3] for k in self.conv_kernel_sizes) and all(list(p) == [2, 2, 2] for p in self.pool_op_kernel_sizes)

    def encoder_convs(self):
        """(stage d, conv i) -> ConvSpec in forward order, bottleneck included."""
        feats = self.features()
        specs, cin = [], self.num_input_channels
        ks, ps = self.conv_kernel_sizes, self.pool_op_kernel_sizes
        for d in range(self.num_pool):
            pre = 'conv_blocks_context.%d.blocks' % d
            s, sz = (1, 1) if d == 0 else (ps[d - 1][1], ps[d - 1][0])
            specs.append(ConvSpec('%s.0' % pre, cin, feats[d], s, d, ks[d][0], sz))
            specs.append(ConvSpec('%s.1' % pre, feats[d], feats[d], 1, d, ks[d][0], 1))
            cin = feats[d]
        d = self.num_pool
        specs.append(ConvSpec('conv_blocks_context.%d.0.blocks.0' % d, cin, feats[d], ps[-1][1], d, ks[d][0], ps[-1][0]))
        specs.append(ConvSpec('conv_blocks_context.%d.1.blocks.0' % d, feats[d], feats[d], 1, d, ks[d][0], 1))
        return specs

    def decoder_stages(self):
        """list of dicts per decoder stage u: tu (cin,cout), two convs, level."""
        feats = self.features()
        out, cdown = [], feats[self.num_pool]
        for u in range(self.num_pool):
            level = self.num_pool - 1 - u
            cskip = feats[level]
            pre = 'conv_blocks_localization.%d' % u
            kz = self.conv_kernel_sizes[-(u + 1)][0]      # upstream: decoder u takes the kernel of the stage one deeper
            out.append(dict(
                u=u, level=level, tu=('tu.%d' % u, cdown, cskip), tu_kt=self.pool_op_kernel_sizes[-(u + 1)][0],
                convs=[ConvSpec('%s.0.blocks.0' % pre, 2 * cskip, cskip, 1, level, kz, 1),
                       ConvSpec('%s.1.blocks.0' % pre, cskip, cskip, 1, level, kz, 1)],
                seg='seg_outputs.%d' % u))
            cdown = cskip
        return out

    @staticmethod
    def from_nnunet_plans(plans, stage=None):
        """Read an unpickled nnU-Net plans dict (plans.pkl / model.pkl['plans'])."""
        stages = plans['plans_per_stage']
        if stage is None:
            stage = max(stages.keys()) if isinstance(stages, dict) else len(stages) - 1
        st = stages[stage]
        return UNetPlan(
            num_input_channels=int(plans['num_modalities']),
            base_num_features=int(plans['base_num_features']),
            num_classes=int(plans['num_classes']) + 1,
            pool_op_kernel_sizes=[[int(v) for v in p] for p in st['pool_op_kernel_sizes']],
            conv_kernel_sizes=[[int(v) for v in k] for k in st['conv_kernel_sizes']],
            conv_per_stage=int(plans.get('conv_per_stage', 2)),
            max_num_features=320,
            patch_size=[int(v) for v in st['patch_size']])


def baseline_plan():
    """BASELINE.json configuration: patch 128^3, 1 modality, 2 classes, 32..320 features, 5 pools."""
    return UNetPlan()


def tiny_plan(patch=32, base=8, num_pool=3, max_feat=64):
    """Small net for fast parity tests."""
    return UNetPlan(1, base, 2, [[2, 2, 2]] * num_pool, [[3, 3, 3]] * (num_pool + 1), 2, max_feat,
                    [patch] * 3)
