// wmh_kernels.hip -- the bandwidth-bound kernels of the nnU-Net 3d_fullres hot path on gfx950:
// InstanceNorm statistics, ConvTranspose3d(k2,s2), patch gather, segmentation head + softmax +
// mirror averaging + Gaussian-weighted accumulation, normalise + argmax.  See include/deepwmh_amd.h
// for what each entry point stands in for in nnU-Net [not in tree] and the reference call sites.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <stdio.h>
#include "wmh_common.h"

// ---- error plumbing ------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
int wmh_fail(int code, const char* msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}
int wmh_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
        return WMH_ELAUNCH;
    }
    return WMH_OK;
}
extern "C" const char* wmh_last_error(void) { return g_err; }
extern "C" int wmh_abi_version(void) { return 6; }
extern "C" int wmh_device_check(int dev) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) return wmh_fail(WMH_ENODEV, "no HIP device");
    if (strncmp(p.gcnArchName, "gfx950", 6) != 0) {
        snprintf(g_err, sizeof(g_err), "device %d is %s, library is built for gfx950", dev, p.gcnArchName);
        return WMH_ENODEV;
    }
    return WMH_OK;
}
static inline unsigned cdivu(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

// ---- InstanceNorm --------------------------------------------------------------------------------
struct Moments { double n, mean, m2; };
__device__ __forceinline__ double shfl_xor_d(double v, int m) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, m); hi = __shfl_xor(hi, m);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void write_scale_shift(Moments t, const float* gamma, const float* beta, float eps,
                                                  float* ss, float* mean_rstd, int nc, int c) {
    const double var = t.n > 0 ? t.m2 / t.n : 0.0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    const double gm = gamma ? (double)gamma[c] : 1.0, bt = beta ? (double)beta[c] : 0.0;
    ss[2 * nc] = (float)(gm * rstd);
    ss[2 * nc + 1] = (float)(bt - t.mean * gm * rstd);
    if (mean_rstd) { mean_rstd[2 * nc] = (float)t.mean; mean_rstd[2 * nc + 1] = (float)rstd; }
}

// One workgroup of four waves per (n, c) row.  The records (count n_i, sum s_i, M2_i about the record's own mean) are dealt round-robin to
// the 256 threads, summed per thread, per wave (fixed butterfly) and across the four waves in wave order -- a fixed association whatever the
// record count (bit-reproducible).
// Round 6: in double precision the pooled moments need no pairwise merge --
//     N = sum n_i,   S = sum s_i,   M2 = sum M2_i + sum s_i^2 / n_i - S^2 / N
// (the cancellation in the last two terms costs ~(mean^2 / var) x 1e-16 relative: nothing at |mean| / std up to 1e4) -- and nearly every
// record of a launch has the same count (a wave's voxels of a full tile), so s_i^2 is summed as it is and divided once; only records with
// another count (ragged tiles) pay a division.  The Chan merge this replaces did two double divisions per record: 8192 records per row at 128^3
// made the 22 launches of a forward 0.39 ms, most of it arithmetic on a 67 MB read.
__global__ __launch_bounds__(256) void instnorm_finalize_kernel(const float4* __restrict__ part, int np, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float eps, float* __restrict__ ss,
                                                                float* __restrict__ mean_rstd, int C) {
    __shared__ double sh[4][5];
    const int nc = blockIdx.x, c = nc % C;
    const float4* p = part + (size_t)nc * np;
    const float n0 = p[0].x;                                     // the common count (uniform: every thread reads record 0)
    double aN = 0.0, aS = 0.0, aM = 0.0, aQ = 0.0, aR = 0.0;    // counts, sums, M2s, s^2 of the common-count records, s^2 / n of the others
    int i = threadIdx.x;
    for (; i + 3 * 256 < np; i += 4 * 256) {                     // four 16-byte loads in flight per lane
        float4 r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) r[u] = p[i + u * 256];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const double sd = (double)r[u].y;
            aN += (double)r[u].x; aS += sd; aM += (double)r[u].z;
            if (r[u].x == n0) aQ += sd * sd;
            else if (r[u].x > 0.f) aR += sd * sd / (double)r[u].x;
        }
    }
    for (; i < np; i += 256) {
        const float4 r = p[i];
        const double sd = (double)r.y;
        aN += (double)r.x; aS += sd; aM += (double)r.z;
        if (r.x == n0) aQ += sd * sd;
        else if (r.x > 0.f) aR += sd * sd / (double)r.x;
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        aN += shfl_xor_d(aN, m); aS += shfl_xor_d(aS, m); aM += shfl_xor_d(aM, m); aQ += shfl_xor_d(aQ, m); aR += shfl_xor_d(aR, m);
    }
    if ((threadIdx.x & 63) == 0) { double* o = sh[threadIdx.x >> 6]; o[0] = aN; o[1] = aS; o[2] = aM; o[3] = aQ; o[4] = aR; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) t[q] = ((sh[0][q] + sh[1][q]) + sh[2][q]) + sh[3][q];
        Moments tot{t[0], 0.0, 0.0};
        if (t[0] > 0.0) {
            tot.mean = t[1] / t[0];
            const double between = (n0 > 0.f ? t[3] / (double)n0 : 0.0) + t[4] - t[1] * t[1] / t[0];
            tot.m2 = t[2] + (between > 0.0 ? between : 0.0);
        }
        write_scale_shift(tot, gamma, beta, eps, ss, mean_rstd, nc, c);
    }
}

extern "C" int wmh_instnorm_finalize(const float* partials, int num_partials, const float* gamma, const float* beta,
                                     float eps, float* ss, float* mean_rstd, int N, int C, void* stream) {
    if (!partials || !ss || num_partials < 1 || N < 1 || C < 1) return wmh_fail(WMH_EINVAL, "instnorm_finalize: bad args");
    hipLaunchKernelGGL(instnorm_finalize_kernel, dim3(N * C), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(partials), num_partials, gamma, beta, eps, ss, mean_rstd, C);
    return wmh_check_launch("instnorm_finalize_kernel");
}

__global__ __launch_bounds__(256) void instnorm_stats_kernel(const float* x, const float* gamma, const float* beta,
                                                             float eps, float* ss, float* mean_rstd, int C,
                                                             size_t spatial) {
    __shared__ double sh[2][4];
    const int nc = blockIdx.x, c = nc % C;
    const float* p = x + (size_t)nc * spatial;
    double s = 0.0;
    for (size_t i = threadIdx.x; i < spatial; i += 256) s += (double)p[i];
    for (int m = 1; m < 64; m <<= 1) s += shfl_xor_d(s, m);
    if ((threadIdx.x & 63) == 0) sh[0][threadIdx.x >> 6] = s;
    __syncthreads();
    const double mean = (sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3]) / (double)spatial;
    double q = 0.0;
    for (size_t i = threadIdx.x; i < spatial; i += 256) { const double d = (double)p[i] - mean; q += d * d; }
    for (int m = 1; m < 64; m <<= 1) q += shfl_xor_d(q, m);
    if ((threadIdx.x & 63) == 0) sh[1][threadIdx.x >> 6] = q;
    __syncthreads();
    if (threadIdx.x == 0) {
        Moments t{(double)spatial, mean, sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3]};
        write_scale_shift(t, gamma, beta, eps, ss, mean_rstd, nc, c);
    }
}
extern "C" int wmh_instnorm_stats(const float* x, const float* gamma, const float* beta, float eps, float* ss,
                                  float* mean_rstd, int N, int C, size_t spatial, void* stream) {
    if (!x || !ss || N < 1 || C < 1 || spatial < 1) return wmh_fail(WMH_EINVAL, "instnorm_stats: bad args");
    hipLaunchKernelGGL(instnorm_stats_kernel, dim3(N * C), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, eps, ss,
                       mean_rstd, C, spatial);
    return wmh_check_launch("instnorm_stats_kernel");
}

__global__ __launch_bounds__(256) void instnorm_lrelu_apply_kernel(const float* x, const float* ss, float* y,
                                                                   size_t spatial, float slope) {
    const int nc = blockIdx.y;
    const float sc = ss[2 * nc], sh = ss[2 * nc + 1];
    const float* p = x + (size_t)nc * spatial;
    float* o = y + (size_t)nc * spatial;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < spatial; i += (size_t)gridDim.x * 256)
        o[i] = wmh_lrelu(fmaf(p[i], sc, sh), slope);
}
extern "C" int wmh_instnorm_lrelu_apply(const float* x, const float* ss, float* y, int N, int C, size_t spatial,
                                        float lrelu_slope, void* stream) {
    if (!x || !ss || !y || N < 1 || C < 1 || spatial < 1) return wmh_fail(WMH_EINVAL, "instnorm_lrelu_apply: bad args");
    unsigned gx = cdivu(spatial, 256); if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(instnorm_lrelu_apply_kernel, dim3(gx, N * C), dim3(256), 0, (hipStream_t)stream, x, ss, y,
                       spatial, lrelu_slope);
    return wmh_check_launch("instnorm_lrelu_apply_kernel");
}

// ---- ConvTranspose3d k=2 s=2 ----------------------------------------------------------------------
// out[n,co,2d+i,2h+j,2w+k] = sum_ci x'[n,ci,d,h,w] * W[ci,co,i,j,k]; one thread = one input voxel x COT
// output channels x 8 taps; weights are wave-uniform (scalar loads); stores are float2 (k = 0,1).
template <int COT>
__global__ __launch_bounds__(256) void convT_k2s2_kernel(const float* x, const float* ss, const float* w, float* y,
                                                         int Cin, int Cout, int D, int H, int W, float slope) {
    const int n = blockIdx.z, co0 = blockIdx.y * COT;
    const size_t DHW = (size_t)D * H * W;
    const size_t v = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool ok = v < DHW;
    const size_t vv = ok ? v : 0;
    float acc[COT][8];
#pragma unroll
    for (int c = 0; c < COT; ++c)
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[c][t] = 0.f;
    const float* xp = x + (size_t)n * Cin * DHW + vv;
    for (int ci = 0; ci < Cin; ++ci) {
        float xv = xp[(size_t)ci * DHW];
        if (ss) xv = wmh_lrelu(fmaf(xv, ss[2 * (n * Cin + ci)], ss[2 * (n * Cin + ci) + 1]), slope);
        const float* wp = w + ((size_t)ci * Cout + co0) * 8;
#pragma unroll
        for (int c = 0; c < COT; ++c) {
            if (co0 + c < Cout) {
#pragma unroll
                for (int t = 0; t < 8; ++t) acc[c][t] = fmaf(xv, wp[c * 8 + t], acc[c][t]);
            }
        }
    }
    if (!ok) return;
    const int wq = (int)(v % W), hq = (int)((v / W) % H), dq = (int)(v / ((size_t)W * H));
    const int H2 = 2 * H, W2 = 2 * W;
    const size_t O = (size_t)8 * DHW;
#pragma unroll
    for (int c = 0; c < COT; ++c) {
        if (co0 + c >= Cout) continue;
        float* yp = y + ((size_t)n * Cout + co0 + c) * O + ((size_t)(2 * dq) * H2 + 2 * hq) * W2 + 2 * wq;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
                *reinterpret_cast<float2*>(yp + ((size_t)i * H2 + jj) * W2) =
                    make_float2(acc[c][i * 4 + jj * 2], acc[c][i * 4 + jj * 2 + 1]);
    }
}
extern "C" int wmh_convtranspose3d_k2s2_fwd(const float* x, const float* ss, const float* w, float* y, int N, int Cin,
                                            int Cout, int D, int H, int W, float lrelu_slope, void* stream) {
    if (!x || !w || !y || N < 1 || Cin < 1 || Cout < 1 || D < 1 || H < 1 || W < 1)
        return wmh_fail(WMH_EINVAL, "convtranspose3d_k2s2_fwd: bad args");
    constexpr int COT = 8;
    const size_t DHW = (size_t)D * H * W;
    dim3 grid(cdivu(DHW, 256), (Cout + COT - 1) / COT, N);
    hipLaunchKernelGGL((convT_k2s2_kernel<COT>), grid, dim3(256), 0, (hipStream_t)stream, x, ss, w, y, Cin, Cout, D, H,
                       W, lrelu_slope);
    return wmh_check_launch("convT_k2s2_kernel");
}

// ---- sliding-window glue ----------------------------------------------------------------------------
struct PatchList {
    int n;
    int org[WMH_MAX_BATCH_ITEMS][3];
    int flip[WMH_MAX_BATCH_ITEMS];
};
static int fill_patchlist(PatchList& pl, int n, const int* origins, int norg, const int* flips) {
    if (n < 1 || n > WMH_MAX_BATCH_ITEMS) return -1;
    pl.n = n;
    for (int b = 0; b < n; ++b) {
        const int ob = (norg == 1) ? 0 : b;
        for (int k = 0; k < 3; ++k) pl.org[b][k] = origins ? origins[3 * ob + k] : 0;
        pl.flip[b] = flips ? flips[b] : 0;
    }
    return 0;
}

__global__ __launch_bounds__(256) void extract_patches_kernel(const float* vol, float* out, int C, int V0, int V1, int V2,
                                                              int l0, int l1, int l2, int P0, int P1, int P2,
                                                              const PatchList pl) {
    const int b = blockIdx.z, c = blockIdx.y;
    const size_t P = (size_t)P0 * P1 * P2;
    const int o0 = pl.org[b][0] - l0, o1 = pl.org[b][1] - l1, o2 = pl.org[b][2] - l2, f = pl.flip[b];
    float* op = out + ((size_t)b * C + c) * P;
    const float* vp = vol + (size_t)c * V0 * V1 * V2;
    for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < P; p += (size_t)gridDim.x * 256) {
        int p2 = (int)(p % P2), p1 = (int)((p / P2) % P1), p0 = (int)(p / ((size_t)P2 * P1));
        if (f & 4) p0 = P0 - 1 - p0;
        if (f & 2) p1 = P1 - 1 - p1;
        if (f & 1) p2 = P2 - 1 - p2;
        const int s0 = o0 + p0, s1 = o1 + p1, s2 = o2 + p2;
        const bool ok = s0 >= 0 && s0 < V0 && s1 >= 0 && s1 < V1 && s2 >= 0 && s2 < V2;
        op[p] = ok ? vp[((size_t)s0 * V1 + s1) * V2 + s2] : 0.f;
    }
}
extern "C" int wmh_extract_patches(const float* volume, float* patches, int C, int V0, int V1, int V2, const int* pad_lo,
                                   int P0, int P1, int P2, int n, const int* origins, const int* flips, void* stream) {
    PatchList pl;
    if (!volume || !patches || !origins || fill_patchlist(pl, n, origins, n, flips))
        return wmh_fail(WMH_EINVAL, "extract_patches: bad args");
    const size_t P = (size_t)P0 * P1 * P2;
    unsigned gx = cdivu(P, 256); if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL(extract_patches_kernel, dim3(gx, C, n), dim3(256), 0, (hipStream_t)stream, volume, patches, C, V0,
                       V1, V2, pad_lo ? pad_lo[0] : 0, pad_lo ? pad_lo[1] : 0, pad_lo ? pad_lo[2] : 0, P0, P1, P2, pl);
    return wmh_check_launch("extract_patches_kernel");
}

// logits_k = sum_c wseg[k][c] * lrelu(x[c]*scale[c]+shift[c]) for one voxel of item `b`
// IN16: x is a bf16 tensor (uint16 bit patterns); element offsets are the same, the bytes half
template <int K, bool IN16 = false>
__device__ __forceinline__ void head_logits(const float* xb, const float* ssb, const float* wseg, int C, size_t P,
                                            size_t src, float slope, float (&lg)[K]) {
#pragma unroll
    for (int k = 0; k < K; ++k) lg[k] = 0.f;
    for (int c = 0; c < C; ++c) {
        float v;
        if constexpr (IN16) v = __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(xb)[(size_t)c * P + src] << 16);
        else v = xb[(size_t)c * P + src];
        if (ssb) v = wmh_lrelu(fmaf(v, ssb[2 * c], ssb[2 * c + 1]), slope);
#pragma unroll
        for (int k = 0; k < K; ++k) lg[k] = fmaf(v, wseg[k * C + c], lg[k]);
    }
}

template <int K, bool IN16 = false>
__global__ __launch_bounds__(256) void seghead_softmax_accum_kernel(const float* x, const float* ss, const float* wseg,
                                                                    const float* gauss, float* agg, int C, int P0,
                                                                    int P1, int P2, int A0, int A1, int A2, int g0,
                                                                    int g1, int g2, const PatchList pl, float slope) {
    const size_t P = (size_t)P0 * P1 * P2, A = (size_t)A0 * A1 * A2;
    const float wgt = 1.f / (float)pl.n;
    for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < P; p += (size_t)gridDim.x * 256) {
        const int p2 = (int)(p % P2), p1 = (int)((p / P2) % P1), p0 = (int)(p / ((size_t)P2 * P1));
        float res[K];
#pragma unroll
        for (int k = 0; k < K; ++k) res[k] = 0.f;
        for (int b = 0; b < pl.n; ++b) {
            const int f = pl.flip[b];
            const int s0 = (f & 4) ? P0 - 1 - p0 : p0, s1 = (f & 2) ? P1 - 1 - p1 : p1, s2 = (f & 1) ? P2 - 1 - p2 : p2;
            const size_t src = ((size_t)s0 * P1 + s1) * P2 + s2;
            float lg[K];
            head_logits<K, IN16>(IN16 ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(x) + (size_t)b * C * P) : x + (size_t)b * C * P,
                                 ss ? ss + (size_t)b * C * 2 : nullptr, wseg, C, P, src, slope, lg);
            float mx = lg[0];
#pragma unroll
            for (int k = 1; k < K; ++k) mx = fmaxf(mx, lg[k]);
            float e[K], den = 0.f;
#pragma unroll
            for (int k = 0; k < K; ++k) { e[k] = expf(lg[k] - mx); den += e[k]; }
#pragma unroll
            for (int k = 0; k < K; ++k) res[k] += wgt * (e[k] / den);
        }
        const float gw = gauss ? gauss[p] : 1.f;
        const size_t dst = ((size_t)(g0 + p0) * A1 + (g1 + p1)) * A2 + (g2 + p2);
        // product and sum rounded separately (no fma contraction), as numpy's `patch *= gaussian; agg += patch`
        // does; it also makes "accumulate a tile computed elsewhere" (patch-sharded multi-GPU) bit-identical
        {
#pragma clang fp contract(off)
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const float t = res[k] * gw;
                agg[(size_t)k * A + dst] = agg[(size_t)k * A + dst] + t;
            }
        }
    }
}

// Same arithmetic, 4 consecutive voxels of the last axis per thread with 16-byte loads (P2 % 4 == 0, aligned x):
// a mirrored last axis reads the mirrored quad and reverses it in registers.
// Round 6: the loop over the channels is cut into groups of eight whose 16-byte loads are issued back to back before the first is used, with no
// branch between them (HAS_SS is a template parameter) and every table value (scale, shift, head weights) read through the scalar cache
// (__restrict__: the stores to `agg` cannot alias them).  As compiled until round 5 the `if (ss)` inside the loop and the vector loads of
// the tables put an `s_waitcnt vmcnt(0)` behind every tensor load -- ONE 16-byte load in flight per lane, 3.65 TB/s on a read-only stream
// (MI355X_MICROARCH.md, Indexed rows: 5.5-6.1 TB/s want >= 64 KiB in flight per CU).  Same operations in the same order: same bits.
// AGG4: the accumulator quads are 16-byte aligned (A2 % 4 == 0, g2 % 4 == 0).
template <int K, bool IN16, bool HAS_SS, bool AGG4>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8)))
void seghead_softmax_accum4_kernel(const float* __restrict__ x, const float* __restrict__ ss, const float* __restrict__ wseg,
                                   const float* __restrict__ gauss, float* __restrict__ agg, int C, int P0,
                                   int P1, int P2, int A0, int A1, int A2, int g0,
                                   int g1, int g2, const PatchList pl, float slope) {
    typedef float f4v __attribute__((ext_vector_type(4)));
    typedef unsigned u2v __attribute__((ext_vector_type(2)));
    const size_t P = (size_t)P0 * P1 * P2, A = (size_t)A0 * A1 * A2, Q = P / 4;
    const int Q2 = P2 / 4;
    const float wgt = 1.f / (float)pl.n;
    constexpr int G = 8;                                        // loads in flight per lane
    for (size_t qi = (size_t)blockIdx.x * 256 + threadIdx.x; qi < Q; qi += (size_t)gridDim.x * 256) {
        const int q2 = (int)(qi % Q2), p1 = (int)((qi / Q2) % P1), p0 = (int)(qi / ((size_t)Q2 * P1));
        const int p2 = 4 * q2;
        float res[K][4];
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i) res[k][i] = 0.f;
        for (int b = 0; b < pl.n; ++b) {
            const int f = pl.flip[b];
            const bool fx = f & 1;
            const int s0 = (f & 4) ? P0 - 1 - p0 : p0, s1 = (f & 2) ? P1 - 1 - p1 : p1, s2 = fx ? P2 - 4 - p2 : p2;
            const size_t src = ((size_t)s0 * P1 + s1) * P2 + s2;
            const float* xb = x + (size_t)b * C * P + src;
            const unsigned short* xb16 = reinterpret_cast<const unsigned short*>(x) + (size_t)b * C * P + src;
            const float* ssb = HAS_SS ? ss + (size_t)b * C * 2 : nullptr;
            float lg[K][4];
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
                for (int i = 0; i < 4; ++i) lg[k][i] = 0.f;
            auto load = [&](int c) __attribute__((always_inline)) -> f4v {
                if constexpr (IN16) {
                    const u2v u = __builtin_nontemporal_load(reinterpret_cast<const u2v*>(xb16 + (size_t)c * P));
                    return f4v{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xFFFF0000u)};
                } else {
                    return __builtin_nontemporal_load(reinterpret_cast<const f4v*>(xb + (size_t)c * P));   // read once
                }
            };
            auto use = [&](int c, const f4v v4) __attribute__((always_inline)) {
                float v[4] = {fx ? v4.w : v4.x, fx ? v4.z : v4.y, fx ? v4.y : v4.z, fx ? v4.x : v4.w};
                if constexpr (HAS_SS) {
                    const float sc = ssb[2 * c], sh = ssb[2 * c + 1];
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = wmh_lrelu(fmaf(v[i], sc, sh), slope);
                }
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const float wk = wseg[k * C + c];
#pragma unroll
                    for (int i = 0; i < 4; ++i) lg[k][i] = fmaf(v[i], wk, lg[k][i]);
                }
            };
            int c = 0;
            for (; c + G <= C; c += G) {
                f4v t[G];
#pragma unroll
                for (int j = 0; j < G; ++j) t[j] = load(c + j);
#pragma unroll
                for (int j = 0; j < G; ++j) use(c + j, t[j]);
            }
            for (; c < C; ++c) use(c, load(c));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float mx = lg[0][i];
#pragma unroll
                for (int k = 1; k < K; ++k) mx = fmaxf(mx, lg[k][i]);
                float e[K], den = 0.f;
#pragma unroll
                for (int k = 0; k < K; ++k) { e[k] = expf(lg[k][i] - mx); den += e[k]; }
#pragma unroll
                for (int k = 0; k < K; ++k) res[k][i] += wgt * (e[k] / den);
            }
        }
        const size_t p = ((size_t)p0 * P1 + p1) * P2 + p2;
        float gw[4] = {1.f, 1.f, 1.f, 1.f};
        if (gauss) { const float4 g4 = *reinterpret_cast<const float4*>(gauss + p); gw[0] = g4.x; gw[1] = g4.y; gw[2] = g4.z; gw[3] = g4.w; }
        const size_t dst = ((size_t)(g0 + p0) * A1 + (g1 + p1)) * A2 + (g2 + p2);
        {
#pragma clang fp contract(off)
            if constexpr (AGG4) {
                f4v old[K];
#pragma unroll
                for (int k = 0; k < K; ++k) old[k] = *reinterpret_cast<const f4v*>(agg + (size_t)k * A + dst);
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    f4v o = old[k];
#pragma unroll
                    for (int i = 0; i < 4; ++i) { const float t = res[k][i] * gw[i]; o[i] = o[i] + t; }
                    *reinterpret_cast<f4v*>(agg + (size_t)k * A + dst) = o;
                }
            } else {
#pragma unroll
                for (int k = 0; k < K; ++k)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float t = res[k][i] * gw[i];
                        agg[(size_t)k * A + dst + i] = agg[(size_t)k * A + dst + i] + t;
                    }
            }
        }
    }
}

static int seghead_softmax_accumulate(const float* x, const float* ss, const float* wseg, int K,
                                     const float* gauss, float* agg, int C, int P0, int P1, int P2, int A0,
                                     int A1, int A2, const int* origin, int n, const int* flips,
                                     float lrelu_slope, int in16, void* stream) {
    PatchList pl;
    if (!x || !wseg || !agg || !origin || fill_patchlist(pl, n, origin, 1, flips))
        return wmh_fail(WMH_EINVAL, "seghead_softmax_accumulate: bad args");
    if (origin[0] < 0 || origin[1] < 0 || origin[2] < 0 || origin[0] + P0 > A0 || origin[1] + P1 > A1 || origin[2] + P2 > A2)
        return wmh_fail(WMH_EINVAL, "seghead_softmax_accumulate: patch window outside the accumulator");
    const size_t P = (size_t)P0 * P1 * P2;
    hipStream_t st = (hipStream_t)stream;
    // 16-byte path: quads along the last axis must not straddle rows and every base must be 16-byte aligned
    const bool quad = (P2 % 4 == 0) && !(reinterpret_cast<uintptr_t>(x) & 15) && (!gauss || !(reinterpret_cast<uintptr_t>(gauss) & 15));
    unsigned gx = cdivu(quad ? P / 4 : P, 256);
    {   // diagnostic library only (WMH_HEAD_BPC): persistent grid of that many workgroups per CU for the quad kernel (grid-stride loop)
        static const int bpc = wmh_diag_env("WMH_HEAD_BPC", 0);
        if (quad && bpc > 0 && gx > 256u * bpc) gx = 256u * bpc;
    }
    const bool agg4 = (A2 % 4 == 0) && (origin[2] % 4 == 0) && !(reinterpret_cast<uintptr_t>(agg) & 15);
#define LAUNCH_HEAD_Q(KK, F, S, A4) hipLaunchKernelGGL((seghead_softmax_accum4_kernel<KK, F, S, A4>), dim3(gx), dim3(256), 0, st, x, ss, wseg, gauss, agg, \
                                 C, P0, P1, P2, A0, A1, A2, origin[0], origin[1], origin[2], pl, lrelu_slope)
#define LAUNCH_HEAD_F(KK, F)                                                                                             \
    if (quad) { if (ss) { if (agg4) LAUNCH_HEAD_Q(KK, F, true, true); else LAUNCH_HEAD_Q(KK, F, true, false); }           \
                else { if (agg4) LAUNCH_HEAD_Q(KK, F, false, true); else LAUNCH_HEAD_Q(KK, F, false, false); } }         \
    else hipLaunchKernelGGL((seghead_softmax_accum_kernel<KK, F>), dim3(gx), dim3(256), 0, st, x, ss, wseg, gauss, agg, C,    \
                            P0, P1, P2, A0, A1, A2, origin[0], origin[1], origin[2], pl, lrelu_slope)
#define LAUNCH_HEAD(KK) do { if (in16) { LAUNCH_HEAD_F(KK, true); } else { LAUNCH_HEAD_F(KK, false); } } while (0)
    switch (K) {
        case 2: LAUNCH_HEAD(2); break;
        case 3: LAUNCH_HEAD(3); break;
        case 4: LAUNCH_HEAD(4); break;
        default: return wmh_fail(WMH_EINVAL, "seghead_softmax_accumulate: K must be 2, 3 or 4");
    }
#undef LAUNCH_HEAD
#undef LAUNCH_HEAD_F
#undef LAUNCH_HEAD_Q
    return wmh_check_launch("seghead_softmax_accum_kernel");
}
extern "C" int wmh_seghead_softmax_accumulate(const float* x, const float* ss, const float* wseg, int K,
                                              const float* gauss, float* agg, int C, int P0, int P1, int P2, int A0,
                                              int A1, int A2, const int* origin, int n, const int* flips,
                                              float lrelu_slope, void* stream) {
    return seghead_softmax_accumulate(x, ss, wseg, K, gauss, agg, C, P0, P1, P2, A0, A1, A2, origin, n, flips, lrelu_slope, 0, stream);
}
// the same with x a bf16 tensor (uint16 bit patterns) when x_bf16 != 0
extern "C" int wmh_seghead_softmax_accumulate_st(const void* x, const float* ss, const float* wseg, int K,
                                                 const float* gauss, float* agg, int C, int P0, int P1, int P2, int A0,
                                                 int A1, int A2, const int* origin, int n, const int* flips,
                                                 float lrelu_slope, int x_bf16, void* stream) {
    return seghead_softmax_accumulate(static_cast<const float*>(x), ss, wseg, K, gauss, agg, C, P0, P1, P2, A0, A1, A2, origin, n, flips,
                                      lrelu_slope, x_bf16 != 0, stream);
}

template <int K, bool IN16 = false>
__global__ __launch_bounds__(256) void seghead_logits_kernel(const float* x, const float* ss, const float* wseg,
                                                             float* out, int C, size_t P, float slope) {
    const int b = blockIdx.y;
    for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < P; p += (size_t)gridDim.x * 256) {
        float lg[K];
        head_logits<K, IN16>(IN16 ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(x) + (size_t)b * C * P) : x + (size_t)b * C * P,
                             ss ? ss + (size_t)b * C * 2 : nullptr, wseg, C, P, p, slope, lg);
#pragma unroll
        for (int k = 0; k < K; ++k) out[((size_t)b * K + k) * P + p] = lg[k];
    }
}
// The same per voxel (identical operation order: bit-identical logits), four consecutive voxels per thread with 16-byte loads and
// stores, eight channels' loads in flight (P % 4 == 0, 16-byte aligned x and out).  The one-voxel form read the training step's
// five deep-supervision heads at ~1 TB/s: 0.7 ms of a 28 ms iteration.
template <int K, bool HAS_SS>
__global__ __launch_bounds__(256) void seghead_logits4_kernel(const float* __restrict__ x, const float* __restrict__ ss, const float* __restrict__ wseg,
                                                              float* __restrict__ out, int C, size_t P, float slope) {
    // (round 6: groups of eight channels whose loads are issued back to back, no branch between them, tables through the scalar cache --
    //  see seghead_softmax_accum4_kernel; same operations in the same order)
    const int b = blockIdx.y;
    const float* xb = x + (size_t)b * C * P;
    const float* ssb = HAS_SS ? ss + (size_t)b * C * 2 : nullptr;
    typedef float f4v __attribute__((ext_vector_type(4)));
    constexpr int G = 8;
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < P / 4; q += (size_t)gridDim.x * 256) {
        float lg[K][4];
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i) lg[k][i] = 0.f;
        auto use = [&](int c, const f4v v4) __attribute__((always_inline)) {
            float v[4] = {v4.x, v4.y, v4.z, v4.w};
            if constexpr (HAS_SS) {
                const float sc = ssb[2 * c], sh = ssb[2 * c + 1];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = wmh_lrelu(fmaf(v[i], sc, sh), slope);
            }
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const float wk = wseg[k * C + c];
#pragma unroll
                for (int i = 0; i < 4; ++i) lg[k][i] = fmaf(v[i], wk, lg[k][i]);
            }
        };
        int c = 0;
        for (; c + G <= C; c += G) {
            f4v t[G];
#pragma unroll
            for (int j = 0; j < G; ++j) t[j] = *reinterpret_cast<const f4v*>(xb + (size_t)(c + j) * P + 4 * q);
#pragma unroll
            for (int j = 0; j < G; ++j) use(c + j, t[j]);
        }
        for (; c < C; ++c) use(c, *reinterpret_cast<const f4v*>(xb + (size_t)c * P + 4 * q));
#pragma unroll
        for (int k = 0; k < K; ++k) *reinterpret_cast<f4v*>(out + ((size_t)b * K + k) * P + 4 * q) = f4v{lg[k][0], lg[k][1], lg[k][2], lg[k][3]};
    }
}
// The deep-supervision heads of the smaller stages (64^3 and below at the training batch): with a whole channel loop per thread the 16^3 head
// is 8 workgroups walking 256 channels in 32 dependent rounds of loads (40 us for 8 MB).  Here a workgroup is 64 voxel quads x 4 channel
// slices (one slice per wave: scale / shift / weights stay scalar), the slices' partial logits are summed through LDS in slice order:
// deterministic, but not the one-thread order of the kernels above (1e-7 relative).
template <int K>
__global__ __launch_bounds__(256) void seghead_logits4s_kernel(const float* x, const float* ss, const float* wseg,
                                                               float* out, int C, size_t P, float slope) {
    __shared__ float part[3][K][4][64];
    const int b = blockIdx.y, qi = threadIdx.x & 63, cs = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float* xb = x + (size_t)b * C * P;
    const float* ssb = ss ? ss + (size_t)b * C * 2 : nullptr;
    typedef float f4v __attribute__((ext_vector_type(4)));
    const size_t q = (size_t)blockIdx.x * 64 + qi, nq = P / 4;
    const size_t qc = q < nq ? q : nq - 1;                      // (lanes beyond the end repeat the last quad and store nothing)
    const int c0 = (int)(((long long)C * cs) / 4), c1 = (int)(((long long)C * (cs + 1)) / 4);
    float lg[K][4];
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) lg[k][i] = 0.f;
#pragma unroll 8
    for (int c = c0; c < c1; ++c) {
        const f4v v4 = *reinterpret_cast<const f4v*>(xb + (size_t)c * P + 4 * qc);
        float v[4] = {v4.x, v4.y, v4.z, v4.w};
        if (ssb) {
            const float sc = ssb[2 * c], sh = ssb[2 * c + 1];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = wmh_lrelu(fmaf(v[i], sc, sh), slope);
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float wk = wseg[k * C + c];
#pragma unroll
            for (int i = 0; i < 4; ++i) lg[k][i] = fmaf(v[i], wk, lg[k][i]);
        }
    }
    if (cs > 0) {
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i) part[cs - 1][k][i][qi] = lg[k][i];
    }
    __syncthreads();
    if (cs == 0 && q < nq) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
#pragma unroll
            for (int i = 0; i < 4; ++i) lg[k][i] = ((lg[k][i] + part[0][k][i][qi]) + part[1][k][i][qi]) + part[2][k][i][qi];
            *reinterpret_cast<f4v*>(out + ((size_t)b * K + k) * P + 4 * q) = f4v{lg[k][0], lg[k][1], lg[k][2], lg[k][3]};
        }
    }
}
static int seghead_logits(const float* x, const float* ss, const float* wseg, int K, float* out, int n, int C,
                          size_t spatial, float lrelu_slope, int in16, void* stream) {
    if (!x || !wseg || !out || n < 1 || C < 1 || spatial < 1) return wmh_fail(WMH_EINVAL, "seghead_logits: bad args");
    unsigned gx = cdivu(spatial, 256); if (gx > 65535) gx = 65535;
    hipStream_t st = (hipStream_t)stream;
    if (!in16 && spatial % 4 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) == 0 && K >= 2 && K <= 4) {
        // the smaller stages (>= 64 channels below the full-resolution level: the 32-channel head keeps the one-thread summation order, which its
        // bf16-tensor form shares bit for bit): channel slices per workgroup (seghead_logits4s_kernel)
        if (spatial / 4 <= 131072 && C >= 64) {
            const unsigned gs = cdivu(spatial / 4, 64);
            if (K == 2) hipLaunchKernelGGL((seghead_logits4s_kernel<2>), dim3(gs, n), dim3(256), 0, st, x, ss, wseg, out, C, spatial, lrelu_slope);
            else if (K == 3) hipLaunchKernelGGL((seghead_logits4s_kernel<3>), dim3(gs, n), dim3(256), 0, st, x, ss, wseg, out, C, spatial, lrelu_slope);
            else hipLaunchKernelGGL((seghead_logits4s_kernel<4>), dim3(gs, n), dim3(256), 0, st, x, ss, wseg, out, C, spatial, lrelu_slope);
            return wmh_check_launch("seghead_logits4s_kernel");
        }
        unsigned g4 = cdivu(spatial / 4, 256); if (g4 > 65535) g4 = 65535;
#define LAUNCH_L4(KK) do { if (ss) hipLaunchKernelGGL((seghead_logits4_kernel<KK, true>), dim3(g4, n), dim3(256), 0, st, x, ss, wseg, out, C, spatial, lrelu_slope); \
                           else hipLaunchKernelGGL((seghead_logits4_kernel<KK, false>), dim3(g4, n), dim3(256), 0, st, x, ss, wseg, out, C, spatial, lrelu_slope); } while (0)
        if (K == 2) LAUNCH_L4(2);
        else if (K == 3) LAUNCH_L4(3);
        else LAUNCH_L4(4);
#undef LAUNCH_L4
        return wmh_check_launch("seghead_logits4_kernel");
    }
    switch (K) {
        case 2: if (in16) hipLaunchKernelGGL((seghead_logits_kernel<2, true>), dim3(gx, n), dim3(256), 0, st, x, ss, wseg, out, C, spatial, lrelu_slope);
                else hipLaunchKernelGGL((seghead_logits_kernel<2>), dim3(gx, n), dim3(256), 0, st, x, ss, wseg, out, C, spatial, lrelu_slope);
                break;
        case 3: if (in16) hipLaunchKernelGGL((seghead_logits_kernel<3, true>), dim3(gx, n), dim3(256), 0, st, x, ss, wseg, out, C, spatial, lrelu_slope);
                else hipLaunchKernelGGL((seghead_logits_kernel<3>), dim3(gx, n), dim3(256), 0, st, x, ss, wseg, out, C, spatial, lrelu_slope);
                break;
        case 4: if (in16) hipLaunchKernelGGL((seghead_logits_kernel<4, true>), dim3(gx, n), dim3(256), 0, st, x, ss, wseg, out, C, spatial, lrelu_slope);
                else hipLaunchKernelGGL((seghead_logits_kernel<4>), dim3(gx, n), dim3(256), 0, st, x, ss, wseg, out, C, spatial, lrelu_slope);
                break;
        default: return wmh_fail(WMH_EINVAL, "seghead_logits: K must be 2, 3 or 4");
    }
    return wmh_check_launch("seghead_logits_kernel");
}
extern "C" int wmh_seghead_logits(const float* x, const float* ss, const float* wseg, int K, float* out, int n, int C,
                                  size_t spatial, float lrelu_slope, void* stream) {
    return seghead_logits(x, ss, wseg, K, out, n, C, spatial, lrelu_slope, 0, stream);
}
extern "C" int wmh_seghead_logits_st(const void* x, const float* ss, const float* wseg, int K, float* out, int n, int C,
                                     size_t spatial, float lrelu_slope, int x_bf16, void* stream) {
    return seghead_logits(static_cast<const float*>(x), ss, wseg, K, out, n, C, spatial, lrelu_slope, x_bf16 != 0, stream);
}

__global__ __launch_bounds__(256) void accumulate_weight_kernel(const float* gauss, float* cnt, int P0, int P1, int P2,
                                                                int A1, int A2, int g0, int g1, int g2) {
    const size_t P = (size_t)P0 * P1 * P2;
    for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < P; p += (size_t)gridDim.x * 256) {
        const int p2 = (int)(p % P2), p1 = (int)((p / P2) % P1), p0 = (int)(p / ((size_t)P2 * P1));
        cnt[((size_t)(g0 + p0) * A1 + (g1 + p1)) * A2 + (g2 + p2)] += gauss ? gauss[p] : 1.f;
    }
}
extern "C" int wmh_accumulate_weight(const float* gauss, float* cnt, int P0, int P1, int P2, int A0, int A1, int A2,
                                     const int* origin, void* stream) {
    if (!cnt || !origin) return wmh_fail(WMH_EINVAL, "accumulate_weight: bad args");
    if (origin[0] < 0 || origin[1] < 0 || origin[2] < 0 || origin[0] + P0 > A0 || origin[1] + P1 > A1 || origin[2] + P2 > A2)
        return wmh_fail(WMH_EINVAL, "accumulate_weight: patch window outside the accumulator");
    const size_t P = (size_t)P0 * P1 * P2;
    hipLaunchKernelGGL(accumulate_weight_kernel, dim3(cdivu(P, 256)), dim3(256), 0, (hipStream_t)stream, gauss, cnt, P0,
                       P1, P2, A1, A2, origin[0], origin[1], origin[2]);
    return wmh_check_launch("accumulate_weight_kernel");
}

__global__ __launch_bounds__(256) void finalize_argmax_kernel(const float* agg, const float* cnt, float* prob,
                                                              uint8_t* seg, int K, int A0, int A1, int A2, int c0,
                                                              int c1, int c2, int V0, int V1, int V2) {
    const size_t V = (size_t)V0 * V1 * V2, A = (size_t)A0 * A1 * A2;
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < V; v += (size_t)gridDim.x * 256) {
        const int v2 = (int)(v % V2), v1 = (int)((v / V2) % V1), v0 = (int)(v / ((size_t)V2 * V1));
        const size_t src = ((size_t)(c0 + v0) * A1 + (c1 + v1)) * A2 + (c2 + v2);
        const float w = cnt[src];
        float best = 0.f; int bi = 0;
        for (int k = 0; k < K; ++k) {
            const float pr = agg[(size_t)k * A + src] / w;
            if (prob) prob[(size_t)k * V + v] = pr;
            if (k == 0 || pr > best) { best = pr; bi = k; }
        }
        seg[v] = (uint8_t)bi;
    }
}
extern "C" int wmh_finalize_argmax(const float* agg, const float* cnt, float* prob, uint8_t* seg, int K, int A0, int A1,
                                   int A2, const int* crop_lo, int V0, int V1, int V2, void* stream) {
    if (!agg || !cnt || !seg || K < 1) return wmh_fail(WMH_EINVAL, "finalize_argmax: bad args");
    const int c0 = crop_lo ? crop_lo[0] : 0, c1 = crop_lo ? crop_lo[1] : 0, c2 = crop_lo ? crop_lo[2] : 0;
    if (c0 < 0 || c1 < 0 || c2 < 0 || c0 + V0 > A0 || c1 + V1 > A1 || c2 + V2 > A2)
        return wmh_fail(WMH_EINVAL, "finalize_argmax: crop window outside the accumulator");
    const size_t V = (size_t)V0 * V1 * V2;
    unsigned gx = cdivu(V, 256);
    hipLaunchKernelGGL(finalize_argmax_kernel, dim3(gx), dim3(256), 0, (hipStream_t)stream, agg, cnt, prob, seg, K, A0,
                       A1, A2, c0, c1, c2, V0, V1, V2);
    return wmh_check_launch("finalize_argmax_kernel");
}
