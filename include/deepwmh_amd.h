/*
 * deepwmh_amd.h -- C ABI of libdeepwmh_amd.so: the MI355X (gfx950) kernels of the nnU-Net
 * 3d_fullres hot path that DeepWMH reaches through `nnUNet_predict` / `nnUNet_train`.
 *
 * Boundary being replaced.  The reference has no in-process FFI for this path: it formats a
 * command line and Popen()s it (reference: deepwmh/utilities/external_call.py:50-52;
 * command lines at deepwmh/main/predict.py:153-156 and
 * deepwmh/pipeline/DCNN_multistage.py:292-309,331-344,501-520,531-535).  Inside that process
 * the arithmetic is torch.nn modules of the un-vendored nnU-Net fork [nnU-Net, not in tree].
 * Each entry point below names the nnU-Net/torch operation it stands in for; the process-level
 * drop-in (same argv, same folders) is `deepwmh_amd.cli` and is described in INTEGRATION.md.
 *
 * Conventions: all tensors are dense fp32 NCDHW in device memory (HBM) unless a parameter says
 * "host"; the caller owns every buffer; `stream` is a hipStream_t passed as void* (NULL = default
 * stream); every call is asynchronous on that stream and returns 0 on success or a negative
 * WMH_E* code (wmh_last_error() gives text).  No global state except the last-error string.
 * No exceptions cross this ABI.  No torch types appear in any signature.
 */
#ifndef DEEPWMH_AMD_H
#define DEEPWMH_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WMH_OK 0
#define WMH_EINVAL (-1)   /* bad argument / unsupported shape */
#define WMH_ELAUNCH (-2)  /* HIP launch failure */
#define WMH_ENODEV (-3)   /* no gfx950 device */

#define WMH_MAX_BATCH_ITEMS 16

const char* wmh_last_error(void);
int wmh_abi_version(void);        /* 6 (round 6); 1 through rounds 1-5 although entry points were added every round */
/* Returns 0 when device `dev` is a gfx950 part this library was built for. */
int wmh_device_check(int dev);

/* ---------------------------------------------------------------------------------------------
 * 3x3x3 convolution, padding 1, stride 1 or 2 on all axes, bias, with
 *   - the previous layer's InstanceNorm+LeakyReLU applied to the INPUT while it is staged into
 *     LDS (ss = per-(n,c) {scale, shift}; x' = lrelu(x*scale + shift); zero padding is applied
 *     after that, as torch pads the normalised tensor),
 *   - an optional second input tensor whose channels follow the first's (concat-free
 *     torch.cat((up, skip), 1)),
 *   - per-(n, cout) InstanceNorm partial statistics of the OUTPUT written from the epilogue.
 * Stands in for nnU-Net ConvDropoutNormNonlin.forward = lrelu(instnorm(conv(x)))
 * [nnU-Net generic_UNet.py; torch.nn.Conv3d + InstanceNorm3d + LeakyReLU] split at the conv
 * output: this call produces the raw conv output y and the statistics; wmh_instnorm_finalize
 * turns them into {scale, shift}; the consumer applies them on load.
 * Arithmetic: exact fp32 (v_mfma_f32_32x32x2_f32 accumulating in k order).
 * ------------------------------------------------------------------------------------------- */

/* Pick the kernel configuration for a layer shape (input spatial dims D,H,W). <0 on error. */
int wmh_conv3d_k3_config(int Cin, int Cout, int D, int H, int W, int stride);
/* General form for nnU-Net's anisotropic plans: kernel (kz,3,3) with kz in {1,3} (weights [Cout][Cin][kz][3][3]),
 * stride (sz, s, s) with sz <= s (pooling [1,2,2] = sz 1, s 2).  Every wmh_conv3d_k3_* call below takes the cfg it
 * returns; `stride` arguments are the in-plane stride s, Do = (D-1)/sz + 1. */
int wmh_conv3d_config(int Cin, int Cout, int D, int H, int W, int kz, int sz, int s);
/* Number of floats of the packed (MFMA fragment-major) weight image for that configuration. */
size_t wmh_conv3d_k3_packed_floats(int cfg, int Cin, int Cout);
/* Host-side repack: w is host memory [Cout][Cin][3][3][3] (torch Conv3d.weight), wf host out. */
int wmh_conv3d_k3_pack_weights(int cfg, const float* w, int Cin, int Cout, float* wf);
/* Partial-statistics records per (n, cout) for an output of Do x Ho x Wo (each record = 4 floats). */
int wmh_conv3d_k3_num_partials(int cfg, int N, int Do, int Ho, int Wo);
/* Floats of scratch the layer needs (0 for most): small, under-filled layers split the cin reduction
 * over workgroups (split-K) and sum the partial outputs in a fixed order afterwards. */
size_t wmh_conv3d_k3_workspace_floats(int cfg, int Cin, int N, int D, int H, int W, int Cout, int stride);

int wmh_conv3d_k3_fwd(int cfg,
                      const float* x0, const float* ss0, int C0, /* [N,C0,D,H,W]; ss0 [N,C0,2] or NULL */
                      const float* x1, const float* ss1, int C1, /* optional second source or NULL/0 */
                      const float* wf, const float* bias,        /* packed weights (device), bias [Cout] or NULL */
                      float* y,                                  /* [N,Cout,Do,Ho,Wo], Do = (D-1)/stride+1 */
                      float* partials,                           /* [N,Cout,num_partials,4] or NULL */
                      float* workspace,                          /* wmh_conv3d_k3_workspace_floats() floats, or NULL */
                      int N, int D, int H, int W, int Cout, int stride,
                      float lrelu_slope, void* stream);

/* The same convolution with bf16 matrix-core operands (BASELINE.json config 3): storage, bias, statistics and
 * outputs stay fp32; inputs (after the on-load InstanceNorm+LeakyReLU) and weights are rounded to bf16
 * (nearest even) and multiplied on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  Stride 1 only, for the
 * configurations wmh_conv3d_k3_bf16_supported() accepts (the layers holding ~90 % of the FLOPs); wf is the bf16
 * fragment-major image from wmh_conv3d_k3_pack_weights_bf16 (packed_halfs_bf16() uint16 values). */
int wmh_conv3d_k3_bf16_supported(int cfg);
/* InstanceNorm partial records per (n, c) row written by wmh_conv3d_k3_fwd_bf16 (its wave split differs from the fp32
 * kernel's, so this replaces wmh_conv3d_k3_num_partials for that call). */
int wmh_conv3d_k3_num_partials_bf16(int cfg, int N, int Do, int Ho, int Wo);
/* ... and by wmh_conv3d_k3_fwd_f16x2 (always the four-consumer-wave geometry). */
int wmh_conv3d_k3_num_partials_f16x2(int cfg, int N, int Do, int Ho, int Wo);
size_t wmh_conv3d_k3_packed_halfs_bf16(int cfg, int Cin, int Cout);
int wmh_conv3d_k3_pack_weights_bf16(int cfg, const float* w, int Cin, int Cout, uint16_t* wf);
/* The same image written on the device, with the channel sub-range / data-gradient modes of
 * wmh_conv3d_k3_pack_weights_device (mixed-precision training: forward and data-gradient convolutions in bf16). */
int wmh_conv3d_k3_pack_weights_bf16_device(int cfg, const float* w, int Cin_total, int Cout, int ci_off, int ci_cnt, int mode,
                                           uint16_t* wf, void* stream);
/* The configuration to call the bf16 entry points with, given the one wmh_conv3d_config chose for the layer: layers with
 * 32-voxel rows and a multiple of 64 output channels are served by a tile of 4 x 4 x 32 voxels x 64 channels
 * (configuration 21, this kernel only), which halves the staging work per output; everything else keeps its cfg.
 * packed_halfs / pack_weights / num_partials / fwd must all be called with the returned value. */
int wmh_conv3d_k3_bf16_config(int cfg, int Cout, int W);
/* The same for the f16x2 entry points (wmh_conv3d_k3_*_f16x2): the value above, except that rows of 8 voxels that are whole 16-byte
 * quads (W % 4 == 0) go to configuration 24 -- the wave-specialised pair kernel on tiles of 8 x 8 x 8 voxels x 64 channels, with
 * split-K slices when the tiles alone do not fill the chip (wmh_conv3d_k3_f16x2_workspace_floats).  f16x2 only. */
int wmh_conv3d_k3_f16x2_config(int cfg, int Cout, int W);
int wmh_conv3d_k3_fwd_bf16(int cfg, const float* x0, const float* ss0, int C0, const float* x1, const float* ss1,
                           int C1, const uint16_t* wf, const float* bias, float* y, float* partials, int N, int D,
                           int H, int W, int Cout, float lrelu_slope, void* stream);
/* "f16x2": fp32-grade products on the 16-bit matrix instruction.  Every operand is split into two FP16 numbers
 * (x = x_hi + x_lo: 21-22 significant bits together; round 2 used bf16 pairs, 16 bits) and the partial products are
 * accumulated in fp32: 4 v_mfma_f32_32x32x16_f16 per 16 input channels and tap instead of 1 -- in the wave-specialised kernels of
 * configurations 0 / 2 / 21 ten per three taps, w_lo x_lo (2^-22 of a product) being left out -- still a quarter of the fp32
 * instruction's cycles or less.  Measured against float64, a layer's error is below the exact-fp32 kernel's (tools/f16x2_check.py).
 * fp32 NCDHW tensors in and out, configurations 0 / 2 / 21 (stride 1, rows of >= 16 voxels), 1 (stride 2, output rows of 32 voxels:
 * its own wave-specialised kernel; one source, no split-K workspace) and 3 / 4 / 5; geometry and
 * epilogue of wmh_conv3d_k3_fwd_bf16, moments records: wmh_conv3d_k3_num_partials_f16x2.  wf: wmh_conv3d_k3_pack_weights_f16x2
 * ([G][cin/8][27][NT][64][8] fp16 of the weights x 2^8: lanes 0-31 the high parts, lanes 32-63 the low parts of an output
 * channel's 8 weights -- configuration 1: a lane holds 4 channels x (hi, lo), channels 0-3 of the chunk on lanes 0-31 and 4-7 on
 * lanes 32-63, hi = the value cut to 11 significant bits; the kernel scales the accumulators back.  Weights must be finite and
 * below 253 in magnitude: WMH_EINVAL otherwise; the device packer saturates). */
int wmh_conv3d_k3_f16x2_supported(int cfg);
size_t wmh_conv3d_k3_packed_halfs_f16x2(int cfg, int Cin, int Cout);
int wmh_conv3d_k3_pack_weights_f16x2(int cfg, const float* w, int Cin, int Cout, uint16_t* wf);
/* the same image from weights in HBM (training); modes as wmh_conv3d_k3_pack_weights_bf16_device, and mode 2 (stride-2 configurations
 * 1 / 3 / 5): w is a ConvTranspose3d(k2, s2) weight [Cout][ci_cnt][2][2][2] whose data gradient is a stride-2 convolution of dY */
int wmh_conv3d_k3_pack_weights_f16x2_device(int cfg, const float* w, int Cin_total, int Cout, int ci_off, int ci_cnt, int mode,
                                             uint16_t* wf, void* stream);
/* The same image builds as ONE launch (a training step rebuilds ~43 of them): wmh_conv3d_k3_pack_job_f16x2 validates one build exactly as
 * wmh_conv3d_k3_pack_weights_f16x2_device does and writes its record (wmh_conv3d_k3_pack_job_bytes() bytes of plain data, host memory) into
 * job_out; the caller copies an array of such records to the device once (the pointers inside are the w / wf it passed) and hands it to
 * wmh_conv3d_k3_pack_weights_f16x2_batch whenever the weights have changed.  Results are bit-identical to the single launches. */
size_t wmh_conv3d_k3_pack_job_bytes(void);
int wmh_conv3d_k3_pack_job_f16x2(int cfg, const float* w, int Cin_total, int Cout, int ci_off, int ci_cnt, int mode, uint16_t* wf, void* job_out);
int wmh_conv3d_k3_pack_weights_f16x2_batch(const void* jobs_device, int njobs, void* stream);
/* Configurations 3 / 4 / 5 (rows of 16 / 8 voxels, stride 2 / 1 / 2; D, H, W = the INPUT extent) run the one-role kernel and
 * take split-K slab memory like wmh_conv3d_k3_fwd_bf16_splitk (workspace >= wmh_conv3d_k3_f16x2_workspace_floats floats, or NULL). */
size_t wmh_conv3d_k3_f16x2_workspace_floats(int cfg, int Cin, int N, int D, int H, int W, int Cout);
int wmh_conv3d_k3_fwd_f16x2(int cfg, const float* x0, const float* ss0, int C0, const float* x1, const float* ss1, int C1,
                             const uint16_t* wf, const float* bias, float* y, float* partials, float* workspace,
                             int N, int D, int H, int W, int Cout, float lrelu_slope, void* stream);
/* A decoder stage's transposed convolution composed with the `up` half of the 3x3x3 convolution behind it (f16x2 mode, inference;
 * csrc/conv3d_up_f16x2.hip).  nnU-Net [Generic_UNet.forward, not in tree; reached through /root/reference/deepwmh/main/predict.py:153-156]
 * computes up = tu(x_low), y = conv(cat(up, skip)); with nothing non-linear in between, conv(tu(.)) is a convolution of the low-resolution
 * tensor with 2 x 2 x 2 taps whose weights depend on the parity class of the output voxel.  Calling sequence for such a stage:
 *   1. wmh_conv3d_k3_fwd_f16x2 on the skip channels alone (weight image of the channel sub-range [Cup, Cin_total), partials NULL):
 *      y = bias + conv(skip);
 *   2. wmh_conv3d_up_fwd_f16x2(x_low, ss_low, ..., accumulate = 1): y += conv(tu(act(x_low))), InstanceNorm moment records of the sum
 *      in `partials` ([N][Cout][wmh_conv3d_up_num_partials_f16x2(D, H, W)][4], as wmh_instnorm_finalize reads them).
 * (D, H, W) is the OUTPUT (high-resolution) extent: whole 8 x 8 x 32 regions, Cout a multiple of 32 (wmh_conv3d_up_supported_f16x2).
 * The weight image is built on the host in double precision from the two layers' weights: w3 [Cout][Cin_total][27] whose input channels
 * [0, Cup) are the transposed convolution's outputs (torch.cat((up, skip), 1)), wt = ConvTranspose3d(k2, s2) weight [Clow][Cup][2][2][2]. */
int wmh_conv3d_up_supported_f16x2(int Clow, int Cup, int Cout, int D, int H, int W);
size_t wmh_conv3d_up_packed_bytes_f16x2(int Clow, int Cout);
int wmh_conv3d_up_num_partials_f16x2(int D, int H, int W);
int wmh_conv3d_up_pack_weights_f16x2(const float* w3, int Cin_total, int Cup, const float* wt, int Clow, int Cout, void* wimg);
int wmh_conv3d_up_fwd_f16x2(const float* xlow, const float* sslow, int Clow, const void* wimg, float* y, float* partials,
                             int N, int D, int H, int W, int Cout, float lrelu_slope, int accumulate, void* stream);
/* The same with split-K for the small-row configurations (rows of 16 / 8 voxels: the 16^3 / 8^3 levels): the channel reduction
 * is cut into slabs in `workspace` (>= wmh_conv3d_k3_bf16_workspace_floats floats; 0 = this launch does not split) and summed in
 * fixed order with the bias; the moments then come as one record per (n, c).  workspace NULL = wmh_conv3d_k3_fwd_bf16. */
size_t wmh_conv3d_k3_bf16_workspace_floats(int cfg, int Cin, int N, int D, int H, int W, int Cout);
int wmh_conv3d_k3_fwd_bf16_splitk(int cfg, const float* x0, const float* ss0, int C0, const float* x1, const float* ss1,
                                  int C1, const uint16_t* wf, const float* bias, float* y, float* partials, float* workspace,
                                  int N, int D, int H, int W, int Cout, float lrelu_slope, void* stream);

/* ---------------------------------------------------------------------------------------------
 * bf16 TENSORS for the full-resolution level (opt-in storage mode of the bf16 inference configuration; what
 * torch.autocast does to the convolution outputs of nnU-Net's mixed-precision predict [nnU-Net network_architecture/
 * neural_network.py::predict_3D(mixed_precision=True), not in tree]).  The "_st" calls are the calls above with a
 * storage flag per side: x_bf16 != 0 -> every source tensor holds bf16 bit patterns (uint16, NCDHW, same element
 * offsets); y_bf16 != 0 -> y is written as bf16 (round to nearest even).  Arithmetic, InstanceNorm moments (from the
 * fp32 accumulators) and every other argument are unchanged.  Served combinations (anything else fails with
 * WMH_EINVAL, there is no fallback): wmh_conv3d_k3_fwd_bf16_st config 0 with W % 4 == 0; wmh_conv3d_k3_fwd_st
 * y_bf16 for config 8 (first layer) and x_bf16 for config 1 (first strided convolution, W % 4 == 0). */
int wmh_conv3d_k3_fwd_bf16_st(int cfg, const void* x0, const float* ss0, int C0, const void* x1, const float* ss1,
                              int C1, const uint16_t* wf, const float* bias, void* y, float* partials, int N, int D,
                              int H, int W, int Cout, float lrelu_slope, int x_bf16, int y_bf16, void* stream);
int wmh_conv3d_k3_fwd_st(int cfg, const void* x0, const float* ss0, int C0, const void* x1, const float* ss1, int C1,
                         const float* wf, const float* bias, void* y, float* partials, int N, int D, int H, int W,
                         int Cout, int stride, float lrelu_slope, int x_bf16, int y_bf16, void* stream);
int wmh_convtranspose3d_mfma_fwd_st(const float* x, const float* ss, const float* wf, void* y, int N, int Cin,
                                    int Cout, int D, int H, int W, int kt, float lrelu_slope, int y_bf16, void* stream);
int wmh_seghead_softmax_accumulate_st(const void* x, const float* ss, const float* wseg, int K,
                                      const float* gauss, float* agg, int C, int P0, int P1, int P2,
                                      int A0, int A1, int A2, const int* origin, int n,
                                      const int* flips, float lrelu_slope, int x_bf16, void* stream);
int wmh_seghead_logits_st(const void* x, const float* ss, const float* wseg, int K, float* out,
                          int n, int C, size_t spatial, float lrelu_slope, int x_bf16, void* stream);

/* ---------------------------------------------------------------------------------------------
 * InstanceNorm3d(eps, affine) statistics -> per-(n,c) {scale, shift} so that
 * instnorm(x) = x*scale + shift.  Merges the conv epilogue's partial (count, sum, M2) records
 * with Chan's parallel-variance update in fp64, in a fixed order (bit-reproducible).
 * Stands in for torch.nn.InstanceNorm3d (biased variance, no running stats).
 * mean_rstd (optional, [N,C,2]) receives {mean, rstd}.
 * ------------------------------------------------------------------------------------------- */
int wmh_instnorm_finalize(const float* partials, int num_partials, const float* gamma,
                          const float* beta, float eps, float* ss, float* mean_rstd, int N, int C,
                          void* stream);
/* Same result computed from the tensor itself (two-pass, for layers produced without partials). */
int wmh_instnorm_stats(const float* x, const float* gamma, const float* beta, float eps, float* ss,
                       float* mean_rstd, int N, int C, size_t spatial, void* stream);
/* y = lrelu(x*scale + shift): the explicit apply pass (tests, deep-supervision outputs). */
int wmh_instnorm_lrelu_apply(const float* x, const float* ss, float* y, int N, int C, size_t spatial,
                             float lrelu_slope, void* stream);

/* ---------------------------------------------------------------------------------------------
 * ConvTranspose3d(kernel 2, stride 2, bias=False) with the producer's InstanceNorm+LeakyReLU
 * applied to the input on load (ss may be NULL).  w is torch layout [Cin][Cout][2][2][2] (device).
 * Stands in for Generic_UNet.tu[u] [nnU-Net; torch.nn.ConvTranspose3d].
 * ------------------------------------------------------------------------------------------- */
int wmh_convtranspose3d_k2s2_fwd(const float* x, const float* ss, const float* w, float* y,
                                 int N, int Cin, int Cout, int D, int H, int W,
                                 float lrelu_slope, void* stream);
/* The same operation on the fp32 matrix cores (the path the network uses): weights repacked once into a
 * fragment-major image (host: w torch layout -> wf, packed_floats() floats); x 16-byte aligned, D*H*W % 4 == 0. */
size_t wmh_convtranspose3d_k2s2_packed_floats(int Cin, int Cout);
int wmh_convtranspose3d_k2s2_pack_weights(const float* w, int Cin, int Cout, float* wf);
int wmh_convtranspose3d_k2s2_mfma_fwd(const float* x, const float* ss, const float* wf, float* y,
                                      int N, int Cin, int Cout, int D, int H, int W,
                                      float lrelu_slope, void* stream);
/* General form: kernel = stride = (kt,2,2), kt in {1,2} (nnU-Net's anisotropic up-sampling is kt = 1);
 * w torch layout [Cin][Cout][kt][2][2]; output [N,Cout,kt*D,2H,2W]. */
size_t wmh_convtranspose3d_packed_floats(int Cin, int Cout, int kt);
int wmh_convtranspose3d_pack_weights(const float* w, int Cin, int Cout, int kt, float* wf);
/* The same image written on the device (w, wf in HBM): training repacks every step. */
int wmh_convtranspose3d_pack_weights_device(const float* w, int Cin, int Cout, int kt, float* wf, void* stream);
int wmh_convtranspose3d_mfma_fwd(const float* x, const float* ss, const float* wf, float* y, int N, int Cin,
                                 int Cout, int D, int H, int W, int kt, float lrelu_slope, void* stream);
/* The transposed convolution in the f16x2 "pair" mode (fp32 tensors and accumulation, products from fp16 pairs: the forward kernels'
 * scheme, two v_mfma_f32_32x32x16_f16 per 8 input channels; any D*H*W).  wf = wmh_convtranspose3d_packed_dwords_f16x2(Cin, Cout, kt)
 * dwords written by wmh_convtranspose3d_pack_weights_f16x2_device from the torch-layout weights [Cin][Cout][kt][2][2] in HBM
 * (|w| is clamped to 253.9). */
size_t wmh_convtranspose3d_packed_dwords_f16x2(int Cin, int Cout, int kt);
int wmh_convtranspose3d_pack_weights_f16x2_device(const float* w, int Cin, int Cout, int kt, void* wf, void* stream);
int wmh_convtranspose3d_fwd_f16x2(const float* x, const float* ss, const void* wf, float* y, int N, int Cin, int Cout,
                                  int D, int H, int W, int kt, float lrelu_slope, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Sliding-window glue [nnU-Net neural_network.py::_internal_predict_3D_3Dconv_tiled and
 * _internal_maybe_mirror_and_pred_3D].
 * A "patch list" is n <= WMH_MAX_BATCH_ITEMS items; item b reads/writes the window of the
 * (padded) volume starting at origins[3*b..3*b+2] and is mirrored along the axes in flips[b]
 * (bit 2 = first spatial axis, bit 1 = second, bit 0 = last: nnU-Net's m = 0..7 order).
 * ------------------------------------------------------------------------------------------- */

/* Gather n patches [n,C,P0,P1,P2] out of volume [C,V0,V1,V2]; voxels outside the volume (the
 * centred zero pad of batchgenerators.pad_nd_image: pad_lo[] voxels before index 0) read as 0. */
int wmh_extract_patches(const float* volume, float* patches, int C, int V0, int V1, int V2,
                        const int* pad_lo, int P0, int P1, int P2, int n, const int* origins,
                        const int* flips, void* stream);

/* Segmentation head (1x1x1 conv, no bias, K classes) on lrelu(instnorm(x)) applied on load,
 * softmax over classes, mean over the `n` mirrored items of ONE patch (weight 1/n each, summed
 * in item order), times the Gaussian importance map, accumulated (+=) into agg[K,A0,A1,A2] at
 * `origin`.  x is [n,C,P0,P1,P2]; gauss is [P0,P1,P2] or NULL (= 1).  K in {2,3,4}.
 * Stands in for seg_outputs[-1] + softmax_helper + the flip/average/`*= gaussian`/`+=` sequence. */
int wmh_seghead_softmax_accumulate(const float* x, const float* ss, const float* wseg, int K,
                                   const float* gauss, float* agg, int C, int P0, int P1, int P2,
                                   int A0, int A1, int A2, const int* origin, int n,
                                   const int* flips, float lrelu_slope, void* stream);
/* Same head, logits only (no softmax/aggregation): out [n,K,P0,P1,P2]. Deep-supervision / tests. */
int wmh_seghead_logits(const float* x, const float* ss, const float* wseg, int K, float* out,
                       int n, int C, size_t spatial, float lrelu_slope, void* stream);

/* cnt[A0,A1,A2] += gauss (or 1 when NULL) over one patch window: aggregated_nb_of_predictions. */
int wmh_accumulate_weight(const float* gauss, float* cnt, int P0, int P1, int P2, int A0, int A1,
                          int A2, const int* origin, void* stream);

/* class_probabilities = agg / cnt cropped to the un-padded volume, seg = argmax (first maximum).
 * prob [K,V0,V1,V2] (may be NULL), seg uint8 [V0,V1,V2]; crop_lo = pad_lo of the extract call. */
int wmh_finalize_argmax(const float* agg, const float* cnt, float* prob, uint8_t* seg, int K,
                        int A0, int A1, int A2, const int* crop_lo, int V0, int V1, int V2,
                        void* stream);

/* ---------------------------------------------------------------------------------------------
 * Training step [nnU-Net nnUNetTrainerV2.run_iteration, loss_functions/dice_loss.py DC_and_CE_loss,
 * deep_supervision.py MultipleOutputLoss2; torch.optim.SGD(nesterov), clip_grad_norm_; not in tree].
 * Reference call sites: deepwmh/pipeline/DCNN_multistage.py:292-309,501-520 (`nnUNet_train ...`).
 * All reductions are fixed-order trees (no atomics): a step is bit-reproducible.
 * ------------------------------------------------------------------------------------------- */

/* Device-side weight repack (weights change every step): w = torch layout [Cout][Cin_total][27] in HBM, input
 * channels [ci_off, ci_off+ci_cnt).  mode 0: forward image (ci_cnt -> Cout).  mode 1: image of the data-gradient
 * convolution (Cout -> ci_cnt; transposed, taps flipped), to be run through wmh_conv3d_k3_fwd on dy.  mode 2: w is a
 * ConvTranspose3d(k2,s2) weight [Cout][ci_cnt][8]; the image is its data gradient as a stride-2 convolution
 * ci_cnt -> Cout over dY (cfg from wmh_conv3d_k3_config(ci_cnt, Cout, 2D, 2H, 2W, 2)). */
int wmh_conv3d_k3_pack_weights_device(int cfg, const float* w, int Cin_total, int Cout, int ci_off, int ci_cnt,
                                      int mode, float* wf, void* stream);
/* dW[Cout][C0+C1][27] = sum over (n, voxel) of dy * lrelu(instnorm(x)) (transform on load as in the forward).
 * General form (nnU-Net's anisotropic stages, [not in tree] generic_UNet.py with conv_kernel_sizes [1,3,3] /
 * pool_op_kernel_sizes [1,2,2]): kernel (kz,3,3), stride (sz, stride, stride), the combinations wmh_conv3d_config serves;
 * dW = [Cout][C0+C1][kz*9]. */
size_t wmh_conv3d_wgrad_workspace_floats(int Cin, int Cout, int N, int D, int H, int W, int kz, int sz, int stride);
int wmh_conv3d_wgrad(const float* x0, const float* ss0, int C0, const float* x1, const float* ss1, int C1,
                     const float* dy, float* dW, float* workspace, int N, int D, int H, int W, int Cout,
                     int kz, int sz, int stride, float lrelu_slope, void* stream);
size_t wmh_conv3d_k3_wgrad_workspace_floats(int Cin, int Cout, int N, int D, int H, int W, int stride);
int wmh_conv3d_k3_wgrad(const float* x0, const float* ss0, int C0, const float* x1, const float* ss1, int C1,
                        const float* dy, float* dW, float* workspace, int N, int D, int H, int W, int Cout,
                        int stride, float lrelu_slope, void* stream);
/* 1 in libdeepwmh_amd_diag.so (the same sources compiled with -DWMH_DIAG: s_memtime stamps in the persistent kernels, WMH_* environment
 * switches that select kernel variants, and the extra entry points of include/deepwmh_amd_diag.h), 0 in libdeepwmh_amd.so, the product,
 * which contains none of them. */
int wmh_diag_build(void);

/* The same weight gradient in the f16x2 "pair" mode (fp32-grade training, the default of nnUNet_train): dy and the activated input
 * are each split into two fp16 numbers (hi toward zero, lo = the remainder: 21-22 significant bits) and kept as one dword per voxel;
 * two v_mfma_f32_32x32x16_f16 per 8 voxels and channel form dy_hi (x_hi + x_lo) + dy_lo (x_hi + x_lo) in fp32 accumulators.
 * |dy|, |x'| up to 131008 (callers scale the loss, as the f16x2 data gradient needs).  Stride 1, rows of >= 8 voxels (stride 2: below)
 * (wmh_conv3d_k3_wgrad_f16x2_supported); workspace >= wmh_conv3d_k3_wgrad_f16x2_workspace_floats(...) floats. */
int wmh_conv3d_k3_wgrad_f16x2_supported(int Cin, int Cout, int N, int D, int H, int W, int stride);
size_t wmh_conv3d_k3_wgrad_f16x2_workspace_floats(int Cin, int Cout, int N, int D, int H, int W);
int wmh_conv3d_k3_wgrad_f16x2(const float* x0, const float* ss0, int C0, const float* x1, const float* ss1, int C1, const float* dy,
                              float* dW, float* workspace, int N, int D, int H, int W, int Cout, float lrelu_slope, void* stream);
/* ... of the convolution with stride 2 (round 5; the fp32 kernel wmh_conv3d_k3_wgrad served it before): (D, H, W) is the INPUT extent, dy
 * has ((D - 1) / 2 + 1, ...); output rows of >= 12 voxels (wmh_conv3d_k3_wgrad_f16x2_supported(..., stride 2)).  The halo rows of the activated
 * input are kept parity-split in LDS (odd inputs | even inputs), so that a tap's operand is consecutive dwords as at stride 1. */
size_t wmh_conv3d_k3s2_wgrad_f16x2_workspace_floats(int Cin, int Cout, int N, int D, int H, int W);
int wmh_conv3d_k3s2_wgrad_f16x2(const float* x0, const float* ss0, int C0, const float* x1, const float* ss1, int C1, const float* dy,
                                float* dW, float* workspace, int N, int D, int H, int W, int Cout, float lrelu_slope, void* stream);
/* Data gradient of the stride-2 3x3x3 convolution in gather form (each of the 27 taps feeds exactly one of the 8 output
 * parity classes: the ideal MFMA count, 1/8 of the zero-insert route below).  w = torch layout [Cout][Cin][27] in HBM;
 * wp = its image, wmh_conv3d_k3s2_dgrad_packed_floats(Cin, Cout) floats; dy = [N][Cout][Do][Ho][Wo] with Do = (D-1)/2+1;
 * dx = [N][Cin][D][H][W], every element written. */
size_t wmh_conv3d_k3s2_dgrad_packed_floats(int Cin, int Cout);
int wmh_conv3d_k3s2_dgrad_pack_weights_device(const float* w, int Cin, int Cout, float* wp, void* stream);
int wmh_conv3d_k3s2_dgrad(const float* dy, const float* wp, float* dx, int N, int Cin, int Cout, int D, int H, int W, void* stream);
/* The same data gradient in the f16x2 "pair" mode (products from fp16 pairs, fp32 accumulation; dy loss-scaled into fp16's range by the
 * caller: |dy| < 65520).  wp = wmh_conv3d_k3s2_dgrad_packed_dwords_f16x2(Cin, Cout) dwords from
 * wmh_conv3d_k3s2_dgrad_pack_weights_f16x2_device (w = torch layout [Cout][Cin][27] in HBM). */
size_t wmh_conv3d_k3s2_dgrad_packed_dwords_f16x2(int Cin, int Cout);
int wmh_conv3d_k3s2_dgrad_pack_weights_f16x2_device(const float* w, int Cin, int Cout, void* wp, void* stream);
int wmh_conv3d_k3s2_dgrad_f16x2(const float* dy, const void* wp, float* dx, int N, int Cin, int Cout, int D, int H, int W, void* stream);
/* up[n,c,z,2y,2x] = dy[n,c,z,y,x], zero elsewhere: the data gradient of a stride (1,2,2) convolution is the stride-1
 * convolution of `up` with the mode-1 weight image. */
int wmh_upsample_zero_plane(const float* dy, float* up, int N, int C, int Ho, int Wo, int D, int H, int W, void* stream);
/* up[n,c,2z,2y,2x] = dy[n,c,z,y,x], zero elsewhere (up has the stride-2 conv's input size D,H,W). */
int wmh_upsample_zero2(const float* dy, float* up, int N, int C, int Do, int Ho, int Wo, int D, int H, int W,
                       void* stream);
/* Dice+CE at one deep-supervision scale: loss_accum[0] += weight*(CE + Dice); dlogits (optional) = weight * dLoss/dlogits.
 * target = uint8 labels [N,S]; workspace = wmh_dice_ce_loss_workspace_doubles() doubles. */
size_t wmh_dice_ce_loss_workspace_doubles(void);
int wmh_dice_ce_loss_fwd_bwd(const float* logits, const uint8_t* target, int N, int K, size_t S, float weight,
                             float smooth, float* dlogits, double* loss_accum, double* workspace, void* stream);
/* Online evaluation of a validation batch (nnUNetTrainer.run_online_evaluation [nnU-Net, not in tree]: the hard counts behind the
 * `model_best` criterion; reference: deepwmh/pipeline/DCNN_multistage.py:529-533, 608-611 use model_best).  seg = argmax_k logits[n,k,v]
 * (first index on ties); for each foreground class c = 1..K-1: counts[c-1] += tp, counts[(K-1)+c-1] += fp, counts[2(K-1)+c-1] += fn
 * (ADDED to what the caller's 3(K-1) 64-bit counters hold: zero them at the start of an epoch's validation). */
int wmh_online_eval_counts(const float* logits, const uint8_t* target, int N, int K, size_t S,
                           unsigned long long* counts, void* stream);
/* Backward of a = lrelu(instnorm(y)) given dA = g1 + g2 + wseg^T dlogits (any subset): writes dy (the gradient of
 * the raw conv output) and the parameter gradients dgamma[C], dbeta[C], dwseg[K][C] and dbias[C] = sum dy (any may
 * be NULL).  scratch = wmh_in_lrelu_bwd_scratch_doubles(N, C, S) doubles. */
size_t wmh_in_lrelu_bwd_scratch_doubles(int N, int C, size_t S);
int wmh_in_lrelu_bwd(const float* y, const float* ss, const float* mean_rstd, const float* gamma, const float* g1,
                     const float* g2, const float* dlogits, const float* wseg, int K, float* dy, float* dgamma,
                     float* dbeta, float* dwseg, float* dbias, double* scratch, int N, int C, size_t S, float slope,
                     void* stream);
/* out[c] = sum over (n, voxels) of x[n,c,:] (conv bias gradient). */
int wmh_channel_sum(const float* x, float* out, int N, int C, size_t S, void* stream);
/* ConvTranspose3d(k2,s2) backward: dX (gradient of the activated input; may be NULL) and dW (may be NULL). */
int wmh_convtranspose3d_k2s2_bwd(const float* x, const float* ss, const float* w, const float* dY, float* dX,
                                 float* dW, int N, int Cin, int Cout, int D, int H, int W, float slope, void* stream);
/* The same dW on the matrix cores (voxels as the MFMA K dimension, fixed-order slab reduction).
 * workspace >= wmh_convtranspose3d_k2s2_wgrad_workspace_floats(...) floats. */
size_t wmh_convtranspose3d_k2s2_wgrad_workspace_floats(int Cin, int Cout, int N, int D, int H, int W);
/* kt = kernel and stride along the first spatial axis: 2, or 1 for ConvTranspose3d((1,2,2)) (dW = [Cin][Cout][kt*4]). */
size_t wmh_convtranspose3d_wgrad_workspace_floats(int Cin, int Cout, int N, int D, int H, int W, int kt);
int wmh_convtranspose3d_wgrad(const float* x, const float* ss, const float* dY, float* dW, float* workspace,
                              int N, int Cin, int Cout, int D, int H, int W, int kt, float slope, void* stream);
int wmh_convtranspose3d_k2s2_wgrad(const float* x, const float* ss, const float* dY, float* dW, float* workspace,
                                   int N, int Cin, int Cout, int D, int H, int W, float slope, void* stream);
/* out2[0] = ||grads||_2, out2[1] = min(1, max_norm/(norm+1e-6)); workspace = 1024 doubles.  A non-finite norm (overflow in a
   loss-scaled backward pass) is reported in out2[0] with out2[1] = 0. */
int wmh_grad_clip_coef(const float* grads, size_t n, float max_norm, float* out2, double* workspace, void* stream);
/* torch.optim.SGD(momentum, nesterov=True, weight_decay) on a flat arena; clip = out2 of wmh_grad_clip_coef or NULL.  When
   clip[0] is not finite the step is skipped: parameters and momentum are left untouched (GradScaler.step semantics). */
int wmh_sgd_nesterov_step(float* params, const float* grads, float* momentum_buf, size_t n, float lr, float momentum,
                          float weight_decay, const float* clip, int first_step, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Whole-volume passes either side of the network.  Volumes are [D][H][W] (C order), masks uint8.
 * Integer results (labels, masks, boxes) are exact and independent of scheduling.
 * ------------------------------------------------------------------------------------------- */

/* 6-connected components of mask != 0: labels[v] = smallest voxel index of v's component, -1 for background.
 * Stands in for scipy.ndimage.label as used by deepwmh/analysis/image_ops.py:8,333 (same partition; label ids differ). */
int wmh_cc_label(const uint8_t* mask, int32_t* labels, int D, int H, int W, void* stream);
/* out = mask with every component of fewer than min_volume voxels removed: remove_sparks,
 * deepwmh/analysis/image_ops.py:325-345 (remove_3mm_sparks :347-367 picks min_volume).  labels, counts: D*H*W int32 scratch. */
int wmh_remove_small_components(const uint8_t* mask, uint8_t* out, int32_t* labels, int32_t* counts, int D, int H, int W,
                                int min_volume, void* stream);
/* scipy.ndimage.binary_fill_holes (default cross structure): background that cannot reach the border becomes 1
 * [nnU-Net cropping.create_nonzero_mask].  labels, flags: D*H*W int32 scratch. */
int wmh_fill_holes(const uint8_t* mask, uint8_t* out, int32_t* labels, int32_t* flags, int D, int H, int W, void* stream);
/* mask[v] = any channel of data[C][D][H][W] is non-zero at v. */
int wmh_nonzero_mask(const float* data, uint8_t* mask, int C, int D, int H, int W, void* stream);
/* bbox6 (device) = {z0, z1, y0, y1, x0, x1}, half-open box of mask != 0; the whole volume if the mask is empty. */
int wmh_mask_bbox(const uint8_t* mask, int32_t* bbox6, int D, int H, int W, void* stream);
/* nnU-Net non-CT normalisation fused with the crop: per channel, (x - mean) / (std + 1e-8) over the box (inside mask
 * when use_mask, 0 outside it); out = [C][box], mask_out = [box] (may be NULL), stats = [C][2] {mean, std}.
 * Moments accumulate in double in a fixed order.  workspace >= wmh_zscore_crop_workspace_doubles(C). */
size_t wmh_zscore_crop_workspace_doubles(int C);
int wmh_zscore_crop(const float* data, const uint8_t* mask, const int32_t* bbox6, float* out, uint8_t* mask_out, float* stats,
                    double* workspace, int C, int D, int H, int W, int use_mask, void* stream);
/* full[box] = crop, `fill` elsewhere (segmentation export un-crop). */
int wmh_uncrop_u8(const uint8_t* crop, const int32_t* bbox6, uint8_t* full, int D, int H, int W, int fill, void* stream);

/* Spline resampling between voxel grids: skimage.transform.resize(order, mode='edge', anti_aliasing=False) as nnU-Net's
 * preprocessing (order 3 data, order 0 mask) and export (order 1 softmax) use it [not in tree]; same arithmetic as
 * scipy.ndimage.zoom(order, mode='nearest', grid_mode=True) in double (u = (o + 0.5) * in/out - 0.5).  in = [C][D][H][W],
 * out = [C][Do][Ho][Wo]; orders 0, 1, 3.  workspace >= wmh_resample_workspace_doubles(...) doubles (order 3 only). */
size_t wmh_resample_workspace_doubles(int D, int H, int W, int order);
int wmh_resample(const float* in, float* out, double* workspace, int C, int D, int H, int W, int Do, int Ho, int Wo, int order,
                 void* stream);
int wmh_resample_nearest_u8(const uint8_t* in, uint8_t* out, int D, int H, int W, int Do, int Ho, int Wo, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Stage-II glue of DeepWMH's self-training loop.  fp32 in numpy's operation order without
 * fused multiply-add: bit-identical to the reference expressions cited.
 * ------------------------------------------------------------------------------------------- */

/* y = 1 - (m * (1 - x)): valid-mask the inverted (background) softmax -- DCNN_multistage.py:102-109. */
int wmh_softmax_mask(const float* x, const float* m, float* y, size_t n, void* stream);
/* field = (stack[0] + stack[1] + ... + stack[E-1]) / E in that order; label = field < 0.5 (lesion where the averaged
 * background probability is below one half) -- DCNN_multistage.py:115-119.  stack = [E][n]; field may be NULL. */
int wmh_ensemble_softmax(const float* stack, int E, float* field, uint8_t* label, size_t n, void* stream);
/* out3 = {|a>.5 & b>.5|, |a>.5|, |b>.5|}: the three counts hard_dice_binary needs (deepwmh/analysis/metrics.py:26-32). */
int wmh_binary_overlap(const float* a, const float* b, unsigned long long* out3, size_t n, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Stage I: negative-log-likelihood initial segmentation.  The reference is float64 numpy
 * (deepwmh/analysis/lesion_analysis.py:84-278 and the image_ops.py helpers cited per entry); volumes are flat [n] or
 * [D][H][W], stacks [R][n], masks uint8.  Scalars a later kernel consumes stay on the device (`*_dev` arguments).
 * ------------------------------------------------------------------------------------------- */

/* out7 (device) = {count, mean, population std, min inside mask, min over everything, max inside, max over everything};
 * mask NULL = all voxels.
 * Sums run about `shift`.  masked_mean / masked_std (image_ops.py:13-21), np.ma(...).min() (lesion_analysis.py:155). */
size_t wmh_moments_workspace_doubles(void);
int wmh_masked_moments_f32(const float* x, const uint8_t* mask, size_t n, double shift, double* out7, double* workspace, void* stream);
int wmh_masked_moments_f64(const double* x, const uint8_t* mask, size_t n, double shift, double* out7, double* workspace, void* stream);
/* z_score (image_ops.py:172-179): out = (x - mean) / std in double; with fill_mask, voxels outside it become *fill_value_dev
 * (lesion_analysis.py:156,166). */
int wmh_zscore_f64(const float* x, double mean, double stdv, const uint8_t* fill_mask, const double* fill_value_dev, double* out, size_t n,
                   void* stream);
int wmh_mask_fill_f64(double* x, const uint8_t* mask, const double* fill_value_dev, size_t n, void* stream);
/* out = more than half of the R maps are > 0.5 at v (group_mean(m_i) > 0.5, lesion_analysis.py:139-141; tissue vote :239-242). */
int wmh_vote_mask(const float* labels, int R, uint8_t* out, size_t n, void* stream);
/* average_contiguous_labels (image_ops.py:23-38): per-voxel argmax of the votes for ids 0..K-1, first maximum wins. */
int wmh_label_vote(const float* labels, int R, int K, uint8_t* out, size_t n, void* stream);
/* numpy.histogram(x[mask], bins=edges): edges_dev = nbins+1 increasing doubles, last bin closed, outside ignored.  With
 * replace_mask, voxels outside it count as *replace_value_dev (np.where(m < 0.5, x.min(), x), lesion_analysis.py:145). */
int wmh_histogram_f64(const double* x, const uint8_t* mask, const uint8_t* replace_mask, const double* replace_value_dev, size_t n,
                      const double* edges_dev, int nbins, unsigned long long* counts, void* stream);
/* mean_std_grid (image_ops.py:56-170), in two steps: block statistics of the float32-rounded, zero-padded data on the
 * coarse grid, then scipy.ndimage.zoom(order=1) of the zero-surrounded grid evaluated on the window the reference keeps. */
int wmh_block_stats(const double* x, const uint8_t* mask, int D, int H, int W, const int* patch3, const int* step3, const int* grid3,
                    double* mean_grid, double* std_grid, void* stream);
int wmh_zoom_linear_window(const double* grid, const int* grid3, const int* zoomed3, const int* offset3, int D, int H, int W, double* out,
                           void* stream);
int wmh_sub_add_f64(double* x, const double* a, const double* b, size_t n, void* stream);       /* x = x - a + b */
int wmh_mul_mask_f64(double* x, const uint8_t* mask, size_t n, void* stream);                  /* x *= (mask != 0) */
int wmh_select_f64(double* x, const double* y, const uint8_t* mask, size_t n, void* stream);   /* x = mask ? y : x */
/* nll (lesion_analysis.py:84-113) with unmasked group statistics: mu, sigma over the R references (sigma floored at
 * min_std, or + 1e-6 when min_std < 0), anomaly of x_prime and (optionally) of every reference; side = +1 / -1 / 0. */
int wmh_nll(const double* x_prime, const double* refs, int R, size_t n, double min_std, int side, double* anomaly, double* mu, double* sigma,
            double* anomaly_refs, void* stream);
/* nll(use_mask=True) (/root/reference/deepwmh/analysis/lesion_analysis.py:87-92): a reference counts at a voxel only where it
 * exceeds its own Otsu threshold (thresholds: R doubles on the device); NaN statistics where none counts (anomaly 0 there). */
int wmh_nll_masked(const double* x_prime, const double* refs, const double* thresholds, int R, size_t n, double min_std, int side,
                   double* anomaly, double* mu, double* sigma, void* stream);
/* scipy.ndimage.median_filter(size=(kz,ky,kx), mode='constant', cval=0) (image_ops.py:181-183 via median_3mm :379-421);
 * windows up to 27 voxels. */
int wmh_median_filter_f64(const double* in, double* out, int D, int H, int W, int kz, int ky, int kx, void* stream);
/* component_filtering (image_ops.py:256-308): per slice and orientation, cross erosion then the largest 4-connected
 * component; the three orientations OR-ed.  scratch_u8: n bytes; labels, counts: n int32; best: max(D,H,W) uint64. */
int wmh_component_filtering(const uint8_t* mask, uint8_t* out, uint8_t* scratch_u8, int32_t* labels, int32_t* counts,
                            unsigned long long* best, int D, int H, int W, const int* do_filter3, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Channels-last bf16 tensors ("CL16": [n][c / 8][z][y][x][c % 8] -- planes of 8 channels, a voxel's 8 channels of a plane contiguous
 * = one 16-byte piece; bf16 bit patterns, 16-byte aligned; C % 8 == 0; a plane must stay below 4 GB.  Rounds 1-2 kept all C channels of a
 * voxel together; with planes a wave's 64 lanes, which work on consecutive voxels, read and write contiguous runs) -- the
 * storage='bf16' form of the bf16 inference configuration (BASELINE.json configs[2]: "bf16 forward, fused
 * InstanceNorm+LeakyReLU"; the reference reaches its reduced-precision forward through nnU-Net's autocast predict,
 * [nnU-Net neural_network.py::predict_3D(mixed_precision=True), not in tree], called at
 * /root/reference/deepwmh/main/predict.py:153-156).  Same contract as the NCDHW calls above: a tensor holds RAW
 * convolution outputs, its InstanceNorm+LeakyReLU is a per-(n, c) {scale, shift} table applied by the consumer; the
 * moment records come from the fp32 accumulators and go through wmh_instnorm_finalize unchanged.  Served shapes:
 * rows (last axis) that are multiples of 32 voxels, input channels % 16 == 0, output channels % 32 == 0; anything else
 * fails with WMH_EINVAL (the levels with shorter rows stay on the NCDHW kernels, converted at the boundary).
 * ------------------------------------------------------------------------------------------- */
int wmh_cl_from_ncdhw(const float* x_ncdhw, void* y_cl16, int N, int C, size_t spatial, void* stream);      /* C % 8 == 0 */
int wmh_cl_to_ncdhw(const void* x_cl16, float* y_ncdhw, int N, int C, size_t spatial, void* stream);
/* First convolution of the network: ONE input channel, fp32 NCDHW patches (nothing pending on them) -> CL16. */
size_t wmh_cl_conv3_first_packed_halfs(int Cout);
int wmh_cl_conv3_first_pack_weights(const float* w /* [Cout][1][3][3][3] */, int Cout, uint16_t* wf);
int wmh_cl_conv3_first_num_partials(int D, int H, int W);
int wmh_cl_conv3_first_fwd(const float* x, const uint16_t* wf, const float* bias, void* y, float* partials, int N, int D,
                           int H, int W, int Cout, void* stream);
/* Conv3d 3x3x3, stride 1 (x1 == NULL) or the decoder's two-source form (the torch.cat of Generic_UNet.forward is never
 * materialised).  wf: the bf16 fragment image of wmh_conv3d_k3_pack_weights_bf16(cfg 0, ...). */
int wmh_cl_conv3_num_partials(int D, int H, int W);
int wmh_cl_conv3_fwd(const void* x0, const float* ss0, int C0, const void* x1, const float* ss1, int C1,
                     const uint16_t* wf, const float* bias, void* y, float* partials, int N, int D, int H, int W,
                     int Cout, float lrelu_slope, void* stream);
/* Conv3d 3x3x3, stride 2, padding 1 (the first convolution of an encoder stage): x CL16 of [N][Cin] channels x [D][H][W] ->
 * y CL16 of [N][Cout] x [(D+1)/2][(H+1)/2][W/2]; W % 64 == 0.  wf: image of wmh_conv3d_k3_pack_weights_bf16(cfg 0, ...). */
int wmh_cl_conv3s2_num_partials(int Do, int Ho, int Wo);
int wmh_cl_conv3s2_fwd(const void* x, const float* ss, int Cin, const uint16_t* wf, const float* bias, void* y,
                       float* partials, int N, int D, int H, int W, int Cout, float lrelu_slope, void* stream);
/* ConvTranspose3d kernel 2 stride 2 without bias: x CL16 [N][Cin] x [D][H][W] -> y CL16 [N][Cout] x [2D][2H][2W]; D*H*W % 32 == 0. */
size_t wmh_cl_convt_packed_halfs(int Cin, int Cout);
int wmh_cl_convt_pack_weights(const float* w /* [Cin][Cout][2][2][2] */, int Cin, int Cout, uint16_t* wf);
int wmh_cl_convt_fwd(const void* x, const float* ss, const uint16_t* wf, void* y, int N, int Cin, int Cout, int D, int H,
                     int W, float lrelu_slope, void* stream);
/* wmh_seghead_softmax_accumulate on a CL16 feature map x of [n][C] channels x [P0][P1][P2] (C = 32, K = 2). */
int wmh_cl_seghead_softmax_accumulate(const void* x, const float* ss, const float* wseg, int K, const float* gauss,
                                      float* agg, int C, int P0, int P1, int P2, int A0, int A1, int A2,
                                      const int* origin, int n, const int* flips, float lrelu_slope, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Self-test helper: D = A(32x2) * B(2x32) through one v_mfma_f32_32x32x2_f32, used by the GPU
 * tests to pin the fragment layout this library assumes (a: [32][2], b: [2][32], d: [32][32]).
 * ------------------------------------------------------------------------------------------- */
int wmh_selftest_mfma_32x32x2(const float* a, const float* b, float* d, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DEEPWMH_AMD_H */
