"""Oracle restatement of nnU-Net v1 `Generic_UNet` as instantiated by `nnUNetTrainerV2`.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py; parity unpinned).

[nnU-Net, not in tree] upstream: nnunet/network_architecture/generic_UNet.py
(`ConvDropoutNormNonlin`, `StackedConvLayers`, `Generic_UNet`), configured as in
nnunet/training/network_training/nnUNetTrainerV2.py::initialize_network:
  conv_op=Conv3d(bias=True), norm_op=InstanceNorm3d(eps=1e-5, affine=True),
  dropout p=0 (module absent), nonlin=LeakyReLU(1e-2), convolutional_pooling,
  convolutional_upsampling, deep supervision, final_nonlin=identity,
  seg_output_use_bias=False, InitWeights_He(1e-2).
Reference-side evidence that this is the configuration used: trainer
`nnUNetTrainerV2`, network `3d_fullres`, plans `nnUNetPlansv2.1` at
/root/reference/deepwmh/main/predict.py:134-149 and
/root/reference/deepwmh/pipeline/DCNN_multistage.py:194-202.

Module/attribute names reproduce upstream's exactly so that `state_dict()` keys
are the checkpoint keys (`conv_blocks_context.0.blocks.0.conv.weight`,
`conv_blocks_localization.0.0.blocks.0.instnorm.bias`, `tu.0.weight`,
`seg_outputs.0.weight`, ...): a real DeepWMH `model_best.model` loads into this
class with `load_state_dict(strict=True)`.
"""
from collections import OrderedDict

import numpy as np
import torch
from torch import nn


class ConvDropoutNormNonlin(nn.Module):
    """lrelu(instnorm(conv(x))) -- upstream ConvDropoutNormNonlin.forward (dropout p=0 -> None)."""

    def __init__(self, cin, cout, kernel, stride):
        super().__init__()
        pad = [1 if k == 3 else 0 for k in kernel]
        self.conv = nn.Conv3d(cin, cout, kernel, stride, pad, dilation=1, bias=True)
        self.instnorm = nn.InstanceNorm3d(cout, eps=1e-5, affine=True)
        self.lrelu = nn.LeakyReLU(negative_slope=1e-2, inplace=True)

    def forward(self, x):
        return self.lrelu(self.instnorm(self.conv(x)))


class StackedConvLayers(nn.Module):
    """num_convs blocks; only the first carries the stride (convolutional pooling)."""

    def __init__(self, cin, cout, num_convs, kernel, first_stride=None):
        super().__init__()
        self.input_channels = cin
        self.output_channels = cout
        s1 = [1, 1, 1]
        self.blocks = nn.Sequential(
            *([ConvDropoutNormNonlin(cin, cout, kernel, first_stride if first_stride is not None else s1)] +
              [ConvDropoutNormNonlin(cout, cout, kernel, s1) for _ in range(num_convs - 1)]))

    def forward(self, x):
        return self.blocks(x)


class GenericUNetOracle(nn.Module):
    """Upstream Generic_UNet.__init__/forward for the 3-D nnUNetTrainerV2 configuration."""

    MAX_NUM_FILTERS_3D = 320

    def __init__(self, input_channels, base_num_features, num_classes, pool_op_kernel_sizes,
                 conv_kernel_sizes, num_conv_per_stage=2, deep_supervision=True,
                 max_num_features=None):
        super().__init__()
        num_pool = len(pool_op_kernel_sizes)
        assert len(conv_kernel_sizes) == num_pool + 1
        self.num_classes = num_classes
        self.do_ds = deep_supervision
        self._deep_supervision = deep_supervision
        self.pool_op_kernel_sizes = [list(p) for p in pool_op_kernel_sizes]
        self.conv_kernel_sizes = [list(k) for k in conv_kernel_sizes]
        if max_num_features is None:
            max_num_features = self.MAX_NUM_FILTERS_3D

        self.conv_blocks_context = nn.ModuleList()
        self.conv_blocks_localization = nn.ModuleList()
        self.tu = nn.ModuleList()
        self.seg_outputs = nn.ModuleList()

        output_features = base_num_features
        input_features = input_channels
        for d in range(num_pool):
            first_stride = pool_op_kernel_sizes[d - 1] if d != 0 else None
            self.conv_blocks_context.append(
                StackedConvLayers(input_features, output_features, num_conv_per_stage,
                                  conv_kernel_sizes[d], first_stride))
            input_features = output_features
            output_features = int(np.round(output_features * 2))
            output_features = min(output_features, max_num_features)

        # bottleneck (convolutional_upsampling=True -> final_num_features = output_features)
        first_stride = pool_op_kernel_sizes[-1]
        final_num_features = output_features
        self.conv_blocks_context.append(nn.Sequential(
            StackedConvLayers(input_features, output_features, num_conv_per_stage - 1,
                              conv_kernel_sizes[num_pool], first_stride),
            StackedConvLayers(output_features, final_num_features, 1, conv_kernel_sizes[num_pool])))

        for u in range(num_pool):
            nfeatures_from_down = final_num_features
            nfeatures_from_skip = self.conv_blocks_context[-(2 + u)].output_channels
            n_features_after_tu_and_concat = nfeatures_from_skip * 2
            final_num_features = nfeatures_from_skip
            self.tu.append(nn.ConvTranspose3d(nfeatures_from_down, nfeatures_from_skip,
                                              pool_op_kernel_sizes[-(u + 1)],
                                              pool_op_kernel_sizes[-(u + 1)], bias=False))
            k = conv_kernel_sizes[-(u + 1)]  # upstream quirk: decoder u takes the kernel of the stage one deeper
            self.conv_blocks_localization.append(nn.Sequential(
                StackedConvLayers(n_features_after_tu_and_concat, nfeatures_from_skip,
                                  num_conv_per_stage - 1, k),
                StackedConvLayers(nfeatures_from_skip, final_num_features, 1, k)))

        for ds in range(len(self.conv_blocks_localization)):
            self.seg_outputs.append(
                nn.Conv3d(self.conv_blocks_localization[ds][-1].output_channels, num_classes,
                          1, 1, 0, 1, 1, bias=False))

        self.apply(init_weights_he)

    def forward(self, x):
        skips = []
        seg_outputs = []
        for d in range(len(self.conv_blocks_context) - 1):
            x = self.conv_blocks_context[d](x)
            skips.append(x)
        x = self.conv_blocks_context[-1](x)
        for u in range(len(self.tu)):
            x = self.tu[u](x)
            x = torch.cat((x, skips[-(u + 1)]), dim=1)
            x = self.conv_blocks_localization[u](x)
            seg_outputs.append(self.seg_outputs[u](x))  # final_nonlin = identity
        if self._deep_supervision and self.do_ds:
            # upscale_logits=False -> identity ops: (full, 1/2, 1/4, ...) resolution logits
            return tuple([seg_outputs[-1]] + seg_outputs[:-1][::-1])
        return seg_outputs[-1]


def init_weights_he(module, neg_slope=1e-2):
    """Upstream InitWeights_He(1e-2): kaiming_normal_(a=neg_slope) on conv weights, bias 0."""
    if isinstance(module, (nn.Conv3d, nn.ConvTranspose3d)):
        module.weight = nn.init.kaiming_normal_(module.weight, a=neg_slope)
        if module.bias is not None:
            module.bias = nn.init.constant_(module.bias, 0)


def unet_from_plan(plan):
    """Build the oracle network from a `deepwmh_amd.plan.UNetPlan`-like object / dict."""
    g = plan if isinstance(plan, dict) else plan.__dict__
    return GenericUNetOracle(g['num_input_channels'], g['base_num_features'], g['num_classes'],
                             g['pool_op_kernel_sizes'], g['conv_kernel_sizes'],
                             g.get('conv_per_stage', 2), deep_supervision=True,
                             max_num_features=g.get('max_num_features', 320))


def perturb_like_trained(net, seed=0):
    """Give IN affine params, biases non-trivial values so parity tests exercise them
    (He-init leaves gamma=1, beta=0, bias=0, which would hide bugs in those paths)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in net.named_parameters():
            if name.endswith('instnorm.weight'):
                p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
            elif name.endswith('instnorm.bias') or name.endswith('conv.bias'):
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
    return net


def state_dict_numpy(net):
    return OrderedDict((k, v.detach().cpu().numpy()) for k, v in net.state_dict().items())
