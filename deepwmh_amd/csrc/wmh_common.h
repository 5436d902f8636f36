// wmh_common.h -- shared helpers of libdeepwmh_amd.so (error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/deepwmh_amd.h"

int wmh_fail(int code, const char* msg);           // records msg, returns code
int wmh_check_launch(const char* what);            // hipGetLastError() -> WMH_OK / WMH_ELAUNCH

__device__ __forceinline__ float wmh_lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }

// ---- diagnostics live in a second library -------------------------------------------------------------------------------------
// libdeepwmh_amd.so (the product) is compiled WITHOUT -DWMH_DIAG: no s_memtime stamp branch exists in any kernel and no environment
// variable selects a kernel variant -- the variant switches below are compile-time constants (their defaults).
// libdeepwmh_amd_diag.so (deepwmh_amd/build.py, same sources with -DWMH_DIAG; load it with WMH_LIB_PATH or _lib.load_diag()) keeps
// both: tools/*_check.py, tools/cl_diag.py, bench.py's in-kernel clock leg and the "wave-specialised == one-role" tests use it.
#include <stdlib.h>
#ifdef WMH_DIAG
#define WMH_STAMPS(a) ((a).stamps)
static inline int wmh_diag_env(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
static inline bool wmh_diag_env_set(const char* name) { return getenv(name) != nullptr; }
#else
#define WMH_STAMPS(a) (static_cast<unsigned long long*>(nullptr))
#define wmh_diag_env(name, dflt) (dflt)
#define wmh_diag_env_set(name) (false)

#endif

// Sum of P slabs of n floats each (the per-workgroup partial weight gradients), in a fixed order: dW[i] = sum_p part[p n + i].  A workgroup is 64
// outputs x 4 slab groups (one wave each): group g adds the slabs [g ceil(P / 4), (g + 1) ceil(P / 4)) in order with eight loads in flight, the
// groups' sums are combined as (g0 + g1) + (g2 + g3).  (Until round 5 one thread walked all P slabs of its output: 108 workgroups of dependent loads
// for the 14 MB a 32 -> 32 layer leaves -- 28 us where the bytes need 4, nineteen times per training iteration.)
#ifdef __HIPCC__
__device__ __forceinline__ void wmh_slab_reduce_body(const float* part, float* dW, size_t n, int P) {
    __shared__ float wmh_sr_sh[3][64];
    const int lane = threadIdx.x & 63, grp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int chunk = (P + 3) / 4;
    const int p0 = grp * chunk < P ? grp * chunk : P, p1 = p0 + chunk < P ? p0 + chunk : P;
    for (size_t base = (size_t)blockIdx.x * 64; base < n; base += (size_t)gridDim.x * 64) {
        const size_t i = base + lane, ic = i < n ? i : n - 1;
        float s = 0.f;
#pragma unroll 8
        for (int p = p0; p < p1; ++p) s += part[(size_t)p * n + ic];
        if (grp) wmh_sr_sh[grp - 1][lane] = s;
        __syncthreads();
        if (grp == 0 && i < n) dW[i] = (s + wmh_sr_sh[0][lane]) + (wmh_sr_sh[1][lane] + wmh_sr_sh[2][lane]);
        __syncthreads();
    }
}
static inline unsigned wmh_slab_reduce_blocks(size_t n) { const size_t b = (n + 63) / 64; return (unsigned)(b > 8192 ? 8192 : b); }
#endif
