/* deepwmh_amd_diag.h -- entry points that exist in libdeepwmh_amd_diag.so ONLY (deepwmh_amd/build.py: the product's sources compiled with
 * -DWMH_DIAG).  The product library libdeepwmh_amd.so exports none of them and its kernels hold no stamp code (csrc/wmh_common.h); everything
 * in include/deepwmh_amd.h is exported by both libraries.  Users: bench.py's in-kernel clock leg, tools/*_check.py, tools/up_stamps.py,
 * the "diagnostic library == product library" tests.  Nothing here corresponds to a reference interface: measurement aids of this build. */
#ifndef DEEPWMH_AMD_DIAG_H
#define DEEPWMH_AMD_DIAG_H
#include "deepwmh_amd.h"
#ifdef __cplusplus
extern "C" {
#endif
/* Diagnostics : with a non-NULL buffer (device memory, 8 x 8 bytes per workgroup and zero-filled by the caller; 2 x grid records for
 * the wave-specialised 16-bit kernels) the persistent convolution kernels launched afterwards add s_memtime cycle counts per workgroup:
 * [0] the first consumer wave's MFMA phases, [1] its barrier waits, [2] its epilogues, [3] chunk iterations, [4..6] a loader wave's work /
 * barrier wait / work beside an epilogue.  s_memtime counts shader-clock cycles, so [0]+[1]+[2] over the launch's duration is the clock the
 * CUs actually ran at (bench.py: roofline.in_kernel; tools/f16x2_check.py).  NULL (the default) switches the stamps off.  Not stream
 * ordered: set it, launch, synchronise, clear it. */
void wmh_debug_set_conv_stamps(void* stamps);
/* wmh_conv3d_k3_wgrad_f16x2 with wmh_conv3d_k3_wgrad_f16x2_stamp_count(...) cycle counters (4 per workgroup: consumer work / wait, loader
 * work / wait) -- tools/wgrad_check.py. */
size_t wmh_conv3d_k3_wgrad_f16x2_stamp_count(int Cin, int Cout, int N, int D, int H, int W);
int wmh_conv3d_k3_wgrad_f16x2_stamped(const float* x0, const float* ss0, int C0, const float* x1, const float* ss1, int C1, const float* dy,
                                      float* dW, float* workspace, int N, int D, int H, int W, int Cout, float lrelu_slope,
                                      unsigned long long* stamps, void* stream);
#ifdef __cplusplus
}
#endif
#endif
