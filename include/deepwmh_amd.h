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
int wmh_abi_version(void);
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
 * Self-test helper: D = A(32x2) * B(2x32) through one v_mfma_f32_32x32x2_f32, used by the GPU
 * tests to pin the fragment layout this library assumes (a: [32][2], b: [2][32], d: [32][32]).
 * ------------------------------------------------------------------------------------------- */
int wmh_selftest_mfma_32x32x2(const float* a, const float* b, float* d, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DEEPWMH_AMD_H */
