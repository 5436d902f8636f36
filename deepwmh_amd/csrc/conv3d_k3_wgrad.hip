// conv3d_k3_wgrad.hip -- weight gradient of the 3x3x3 convolution (pad 1, stride 1|2) on the exact-fp32 MFMA.
//
//   dW[co][ci][tap] = sum_{n, o} dy[n, co, o] * x'[n, ci, S*o + tap - 1]
//
// The contraction runs over voxels, so voxels are the K dimension of v_mfma_f32_32x32x2_f32:
//   D[co 32][tap 32(27 used)] += A[co 32][voxel pair] * B[voxel pair][tap],   one D tile per input channel.
// A workgroup owns 32 output channels x 32 input channels (wave w owns ci = 4c + w, c = 0..7: eight
// accumulators) and walks a strided subset of the spatial tiles, keeping dW in registers; per tile the dy tile
// [32 co][TV voxels] is staged once and the x' halo tile 4 channels at a time (with the producer's
// InstanceNorm+LeakyReLU applied on the way in, zero padding after it).  A lane's B address is
// (its tap offset) + (row base) + immediate, so the inner loop has no address arithmetic.
// Partial dW per workgroup goes to a slab; wgrad_reduce sums the slabs in fixed order (bit-reproducible).
// Stands in for the weight half of torch.nn.Conv3d's backward [nnU-Net training, not in tree].
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "wmh_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WgradArgs {
    const float* x0; const float* ss0; int C0;
    const float* x1; const float* ss1; int C1;
    const float* dy; float* part;
    int N, D, H, W, Do, Ho, Wo, Cout, Cin;
    int TZ, TY, IZ, IY, IX;
    int ntx, nty, ntz, ntiles;
    int NG, NCG, P;
    int dys, xcs;
    float slope;
};

template <int S, int TX, int TY_, int TZ_>
__global__ __launch_bounds__(256, 2) void conv3d_k3_wgrad_kernel(const WgradArgs a) {
    constexpr int TY = TY_, TZ = TZ_, IX = S * (TX - 1) + 3, IY = S * (TY - 1) + 3, IZ = S * (TZ - 1) + 3;   // compile time:
    constexpr int TV = TZ * TY * TX, ZS = IY * IX, SPX = IZ * ZS, DYS = TV + 1, XCS = SPX;                 // no runtime divisions
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* dyl = lds;                      // [32 co][dys]
    float* xl = lds + 32 * DYS;            // [4 ci][XCS]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, kh = lane >> 5;
    int bid = blockIdx.x;
    const int p = bid % a.P; bid /= a.P;
    const int g = bid % a.NG, cg = bid / a.NG;

    const int tapj = min(i, 26);
    const int LB = wave * XCS + kh * S + (tapj / 9) * ZS + ((tapj / 3) % 3) * IX + (tapj % 3);
    const int LA = i * DYS + kh;
    const int HW = a.H * a.W, HoWo = a.Ho * a.Wo;
    const size_t DHW = (size_t)a.D * HW, DoHoWo = (size_t)a.Do * HoWo;

    f32x16 acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

    for (int t = p; t < a.ntiles; t += a.P) {
        int b = t;
        const int tx = b % a.ntx; b /= a.ntx;
        const int ty = b % a.nty; b /= a.nty;
        const int tz = b % a.ntz; b /= a.ntz;
        const int n = b;
        const int ox0 = tx * TX, oy0 = ty * TY, oz0 = tz * TZ;
        const int ix0 = ox0 * S - 1, iy0 = oy0 * S - 1, iz0 = oz0 * S - 1;
        __syncthreads();                                   // previous tile fully consumed
        for (int e = tid; e < 32 * TV; e += 256) {         // dy tile
            const int col = e / TV, v = e - col * TV;
            const int x = v % TX, y = (v / TX) % TY, z = v / (TX * TY);
            const int co = g * 32 + col, oz = oz0 + z, oy = oy0 + y, ox = ox0 + x;
            const bool ok = co < a.Cout && oz < a.Do && oy < a.Ho && ox < a.Wo;
            const float val = ok ? a.dy[((size_t)n * a.Cout + co) * DoHoWo + (size_t)oz * HoWo + oy * a.Wo + ox] : 0.f;
            dyl[col * DYS + v] = val;
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if (cg * 32 + 4 * c >= a.Cin) continue;         // uniform: no such input channels
            if (c > 0) __syncthreads();                    // x' chunk of the previous iteration consumed
            for (int e = tid; e < 4 * SPX; e += 256) {     // x' halo tile, 4 input channels
                const int cc = e / SPX, s = e - cc * SPX;
                const int x = s % IX, y = (s / IX) % IY, z = s / ZS;
                const int ci = cg * 32 + 4 * c + cc;
                const int gx = ix0 + x, gy = iy0 + y, gz = iz0 + z;
                const bool ok = ci < a.Cin && gx >= 0 && gx < a.W && gy >= 0 && gy < a.H && gz >= 0 && gz < a.D;
                float v = 0.f;
                if (ok) {
                    const bool first = ci < a.C0;
                    const float* xs = first ? a.x0 : a.x1;
                    const float* ss = first ? a.ss0 : a.ss1;
                    const int Cs = first ? a.C0 : a.C1, cl = first ? ci : ci - a.C0;
                    v = xs[((size_t)n * Cs + cl) * DHW + (size_t)gz * HW + gy * a.W + gx];
                    if (ss) {
                        const float* q = ss + ((size_t)n * Cs + cl) * 2;
                        v = fmaf(v, q[0], q[1]);
                        v = v > 0.f ? v : v * a.slope;
                    }
                }
                xl[cc * XCS + s] = v;
            }
            __syncthreads();
#pragma unroll
            for (int z = 0; z < TZ; ++z)
#pragma unroll
                for (int y = 0; y < TY; ++y) {
                    const float* ap = dyl + LA + (z * TY + y) * TX;
                    const float* bp = xl + LB + S * (z * ZS + y * IX);
#pragma unroll
                    for (int xp = 0; xp < TX / 2; ++xp)
                        acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * xp], bp[S * 2 * xp], acc[c], 0, 0, 0);
                }
        }
    }
    // partial dW of this workgroup: rows = co (register index), columns = tap (lane)
    if (i < 27) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int ci = cg * 32 + 4 * c + wave;
            if (ci >= a.Cin) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = g * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (co < a.Cout) a.part[(((size_t)p * a.Cout + co) * a.Cin + ci) * 27 + i] = acc[c][r];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Pipelined form (stride 1).  Same mapping as above, but the staging no longer alternates with the MFMAs:
//  * wave w stages the halo tile of the ONE input channel it consumes (ci = 4c + w) into a wave-private,
//    double-buffered LDS region, so x' needs no workgroup barrier at all (LDS is in order within a wave);
//  * the loads of chunk c+1 are issued, branch-free with clamped addresses, before the MFMAs of chunk c and
//    land in registers underneath them; InstanceNorm+LeakyReLU and the zero padding are applied on the way
//    to LDS afterwards;
//  * the next tile's dy is fetched the same way under the last chunk; two barriers per tile swap it in.
// ------------------------------------------------------------------------------------------------
// KZ / SZ: kernel extent and stride along the first spatial axis (nnU-Net's anisotropic stages: kernel [1,3,3], pooling
// [1,2,2]); the tap index on the lanes is (kz, ky, kx) with KZ * 9 taps used.
template <int S, int TX, int TY, int TZ, int KZ = 3, int SZ = S>
__global__ __launch_bounds__(256, 2) void conv3d_k3_wgrad_pipe_kernel(const WgradArgs a) {
    constexpr int IX = S * (TX - 1) + 3, IY = S * (TY - 1) + 3, IZ = SZ * (TZ - 1) + KZ, NTAP = 9 * KZ;
    constexpr int TV = TZ * TY * TX, ZS = IY * IX, SPX = IZ * ZS, DYS = TV + 1;
    constexpr int NK = (SPX + 63) / 64, XB = NK * 64;          // x' elements per lane / floats per wave buffer
    constexpr int NDY = TV / 8, CPK = 256 / TV;                // dy elements per thread / co rows per pass
    static_assert(TV <= 256 && 256 % TV == 0, "dy tile must divide the workgroup");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* dyl = lds;                                          // [32 co][DYS]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* xw = lds + 32 * DYS + wave * 2 * XB;                // [2][XB], private to this wave
    const int i = lane & 31, kh = lane >> 5;
    int bid = blockIdx.x;
    const int p = bid % a.P; bid /= a.P;
    const int g = bid % a.NG, cg = bid / a.NG;
    const int nchunks = min(8, (a.Cin - cg * 32 + 3) / 4);     // uniform: chunks that hold any input channel

    const int tapj = min(i, NTAP - 1);
    const int LB = kh * S + (tapj / 9) * ZS + ((tapj / 3) % 3) * IX + (tapj % 3);
    const int LA = i * DYS + kh;
    const int HW = a.H * a.W, HoWo = a.Ho * a.Wo;
    const size_t DHW = (size_t)a.D * HW, DoHoWo = (size_t)a.Do * HoWo;
    const int dv = tid % TV, dcol0 = tid / TV;
    const int dvx = dv % TX, dvy = (dv / TX) % TY, dvz = dv / (TX * TY);

    f32x16 acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

    // descriptor of the tile whose loads are being issued (byte offsets: scalar base + 32-bit lane offset)
    unsigned coff[NK]; unsigned okm = 0; unsigned dvoff = 0; bool dok = false; int tn = 0;
    auto tile_desc = [&](int t) {
        int b = t;
        const int tx = b % a.ntx; b /= a.ntx;
        const int ty = b % a.nty; b /= a.nty;
        const int tz = b % a.ntz; b /= a.ntz;
        tn = __builtin_amdgcn_readfirstlane(b);                 // (scalar division runs on the VALU)
        const int ox0 = __builtin_amdgcn_readfirstlane(tx * TX), oy0 = __builtin_amdgcn_readfirstlane(ty * TY),
                  oz0 = __builtin_amdgcn_readfirstlane(tz * TZ);
        const int ix0 = ox0 * S - 1, iy0 = oy0 * S - 1, iz0 = oz0 * SZ - KZ / 2;
        okm = 0;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int s = lane + 64 * k;
            const int gx = ix0 + s % IX, gy = iy0 + (s / IX) % IY, gz = iz0 + s / ZS;
            const bool ok = (k * 64 + 63 < SPX || s < SPX) && (unsigned)gx < (unsigned)a.W && (unsigned)gy < (unsigned)a.H &&
                            (unsigned)gz < (unsigned)a.D;
            coff[k] = ok ? 4u * (unsigned)(gz * HW + gy * a.W + gx) : 0u;
            okm |= ok ? (1u << k) : 0u;
        }
        const int ox = ox0 + dvx, oy = oy0 + dvy, oz = oz0 + dvz;
        dok = ox < a.Wo && oy < a.Ho && oz < a.Do;
        dvoff = dok ? 4u * (unsigned)(oz * HoWo + oy * a.Wo + ox) : 0u;
    };
    float xr[NK], dyr[NDY];
    float sc = 1.f, sh = 0.f, sl = 1.f; unsigned xmask = 0;
    auto issue_x = [&](int c) {                                 // chunk c of tile `tn`: this wave's channel
        const int ci = cg * 32 + 4 * c + wave;
        const bool civ = ci < a.Cin;
        const int cic = civ ? ci : 0;
        const bool first = cic < a.C0;
        const float* xs = first ? a.x0 : a.x1;
        const float* ss = first ? a.ss0 : a.ss1;
        const int Cs = first ? a.C0 : a.C1, cl = first ? cic : cic - a.C0;
        const char* src = (const char*)(xs + ((size_t)tn * Cs + cl) * DHW);
#pragma unroll
        for (int k = 0; k < NK; ++k) xr[k] = *(const float*)(src + coff[k]);
        const bool has = ss != nullptr;
        sc = has ? ss[((size_t)tn * Cs + cl) * 2] : 1.f;
        sh = has ? ss[((size_t)tn * Cs + cl) * 2 + 1] : 0.f;
        sl = has ? a.slope : 1.f;                               // fmaf(v,1,0) and v*1 are exact
        xmask = civ ? okm : 0u;
    };
    auto write_x = [&](int buf) {
        float* dst = xw + buf * XB + lane;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            float v = fmaf(xr[k], sc, sh);
            v = v > 0.f ? v : v * sl;
            dst[64 * k] = ((xmask >> k) & 1u) ? v : 0.f;
        }
    };
    const unsigned cstride = 4u * (unsigned)DoHoWo;            // bytes between output channels of dy (host checks the range)
    auto issue_dy = [&]() {
        const char* src = (const char*)(a.dy + (size_t)tn * a.Cout * DoHoWo);
#pragma unroll
        for (int k = 0; k < NDY; ++k) {
            const unsigned co = (unsigned)min(g * 32 + k * CPK + dcol0, a.Cout - 1);
            dyr[k] = *(const float*)(src + (co * cstride + dvoff));
        }
    };
    auto write_dy = [&]() {
#pragma unroll
        for (int k = 0; k < NDY; ++k) {
            const int col = k * CPK + dcol0;
            dyl[col * DYS + dv] = (dok && g * 32 + col < a.Cout) ? dyr[k] : 0.f;
        }
    };

    tile_desc(p);
    issue_dy(); issue_x(0);
    write_dy(); write_x(0);
    __syncthreads();
    for (int t = p; t < a.ntiles; t += a.P) {
        const bool more = t + a.P < a.ntiles;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if (c < 7) { if (c + 1 < nchunks) issue_x(c + 1); }
            else if (more) { tile_desc(t + a.P); issue_dy(); issue_x(0); }
            __builtin_amdgcn_sched_barrier(0x38F);             // the loads stay here, above the MFMAs
            if (c < nchunks) {
                const float* a0 = dyl + LA;
                const float* b0 = xw + (c & 1) * XB + LB;
#pragma unroll
                for (int z = 0; z < TZ; ++z)
#pragma unroll
                    for (int y = 0; y < TY; ++y) {
                        const float* ap = a0 + (z * TY + y) * TX;
                        const float* bp = b0 + SZ * z * ZS + S * y * IX;
#pragma unroll
                        for (int xp = 0; xp < TX / 2; ++xp)
                            acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * xp], bp[S * 2 * xp], acc[c], 0, 0, 0);
                    }
            }
            __builtin_amdgcn_sched_barrier(0x38F);
            if (c < 7) {
                if (c + 1 < nchunks) {
                    write_x((c + 1) & 1);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // same wave reads it back: order only
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
            } else {
                __syncthreads();                               // every wave is done with this dy tile
                if (more) { write_dy(); write_x(0); }
                __syncthreads();
            }
        }
    }
    if (i < NTAP) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int ci = cg * 32 + 4 * c + wave;
            if (ci >= a.Cin) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = g * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (co < a.Cout) a.part[(((size_t)p * a.Cout + co) * a.Cin + ci) * NTAP + i] = acc[c][r];
            }
        }
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* part, float* dW, size_t n, int P) { wmh_slab_reduce_body(part, dW, n, P); }   // wmh_common.h

struct WgradPlan { int TX, TY, TZ, IZ, IY, IX, ntx, nty, ntz, ntiles, NG, NCG, P, dys, xcs, lds_bytes; };
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static bool wgrad_old_form() {
    static const bool v = wmh_diag_env_set("WMH_WGRAD_OLD");      // diagnostic: the unpipelined kernel
    return v;
}
static WgradPlan plan_wgrad(int Cin, int Cout, int N, int D, int H, int W, int S, int KZ = 3, int SZ = 0) {
    WgradPlan q;
    if (SZ == 0) SZ = S;
    const int Do = (D - 1) / SZ + 1, Ho = (H - 1) / S + 1, Wo = (W - 1) / S + 1;
    q.TX = Wo >= 24 ? 32 : Wo >= 12 ? 16 : Wo >= 6 ? 8 : 4;
    if (S == 2 && (!wgrad_old_form() || KZ != 3 || SZ != S)) {   // pipelined stride 2: halo tiles small enough to prefetch
        if (q.TX == 32) q.TX = 16;
        if (q.TX == 16) { q.TY = 2; q.TZ = 2; } else if (q.TX == 8) { q.TY = 4; q.TZ = 2; } else { q.TY = 4; q.TZ = 4; }
    }
    else if (q.TX == 32) { q.TY = 4; q.TZ = 2; }                 // keep in step with the WG()/WP() tables below
    else if (q.TX == 16) { q.TY = S == 1 ? 8 : 4; q.TZ = 2; }
    else if (q.TX == 8) { q.TY = 8; q.TZ = S == 1 ? 4 : 2; }
    else { q.TY = 4; q.TZ = 4; }
    q.IX = S * (q.TX - 1) + 3; q.IY = S * (q.TY - 1) + 3; q.IZ = SZ * (q.TZ - 1) + KZ;
    q.ntx = cdiv(Wo, q.TX); q.nty = cdiv(Ho, q.TY); q.ntz = cdiv(Do, q.TZ);
    q.ntiles = N * q.ntx * q.nty * q.ntz;
    q.NG = cdiv(Cout, 32); q.NCG = cdiv(Cin, 32);
    const int pairs = q.NG * q.NCG;
    q.P = cdiv(512, pairs); if (q.P > q.ntiles) q.P = q.ntiles; if (q.P < 1) q.P = 1;
    q.dys = q.TZ * q.TY * q.TX + 1;
    q.xcs = q.IZ * q.IY * q.IX;
    q.lds_bytes = (32 * q.dys + 4 * q.xcs) * 4;
    return q;
}

static bool wgrad_shape_ok(int kz, int sz, int stride) {
    return (stride == 1 || stride == 2) && (sz == 1 || sz == 2) && (kz == 1 || kz == 3) && sz <= stride && !(kz == 1 && sz == 2);
}
// General form: kernel (kz,3,3), stride (sz, stride, stride) -- the combinations wmh_conv3d_config has forward kernels for
extern "C" size_t wmh_conv3d_wgrad_workspace_floats(int Cin, int Cout, int N, int D, int H, int W, int kz, int sz, int stride) {
    if (!wgrad_shape_ok(kz, sz, stride)) return 0;
    const WgradPlan q = plan_wgrad(Cin, Cout, N, D, H, W, stride, kz, sz);
    return (size_t)q.P * Cout * Cin * 9 * kz;
}
extern "C" size_t wmh_conv3d_k3_wgrad_workspace_floats(int Cin, int Cout, int N, int D, int H, int W, int stride) {
    return wmh_conv3d_wgrad_workspace_floats(Cin, Cout, N, D, H, W, 3, stride, stride);
}

template <int S, int TX, int TY, int TZ, int KZ = 3, int SZ = S>
static int launch_wgrad_pipe(const WgradArgs& a, unsigned nblk, hipStream_t st) {
    constexpr int IX = S * (TX - 1) + 3, IY = S * (TY - 1) + 3, IZ = SZ * (TZ - 1) + KZ;
    constexpr int XB = ((IX * IY * IZ + 63) / 64) * 64;
    constexpr int lds = (32 * (TX * TY * TZ + 1) + 8 * XB) * 4;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_k3_wgrad_pipe_kernel<S, TX, TY, TZ, KZ, SZ>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((conv3d_k3_wgrad_pipe_kernel<S, TX, TY, TZ, KZ, SZ>), dim3(nblk), dim3(256), lds, st, a);
    return wmh_check_launch("conv3d_k3_wgrad_pipe_kernel");
}

template <int S, int TX, int TY, int TZ>
static int launch_wgrad(const WgradArgs& a, unsigned nblk, int lds, hipStream_t st) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_k3_wgrad_kernel<S, TX, TY, TZ>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((conv3d_k3_wgrad_kernel<S, TX, TY, TZ>), dim3(nblk), dim3(256), lds, st, a);
    return wmh_check_launch("conv3d_k3_wgrad_kernel");
}

extern "C" int wmh_conv3d_wgrad(const float* x0, const float* ss0, int C0, const float* x1, const float* ss1, int C1,
                                const float* dy, float* dW, float* workspace, int N, int D, int H, int W, int Cout,
                                int kz, int sz, int stride, float lrelu_slope, void* stream) {
    if (!x0 || !dy || !dW || !workspace || C0 < 1 || N < 1 || Cout < 1 || !wgrad_shape_ok(kz, sz, stride))
        return wmh_fail(WMH_EINVAL, "conv3d_wgrad: bad args");
    if (!x1) C1 = 0;
    const int Cin = C0 + C1;
    const bool iso = kz == 3 && sz == stride;
    const WgradPlan q = plan_wgrad(Cin, Cout, N, D, H, W, stride, kz, sz);
    if (q.lds_bytes > 160 * 1024) return wmh_fail(WMH_EINVAL, "conv3d_wgrad: tile does not fit LDS");
    if ((size_t)D * H * W * 4 >= ((size_t)1 << 32) || (size_t)Cout * D * H * W * 4 >= ((size_t)1 << 32))
        return wmh_fail(WMH_EINVAL, "conv3d_wgrad: volume too large for 32-bit lane offsets");
    WgradArgs a;
    a.x0 = x0; a.ss0 = ss0; a.C0 = C0; a.x1 = x1; a.ss1 = ss1; a.C1 = C1; a.dy = dy; a.part = workspace;
    a.N = N; a.D = D; a.H = H; a.W = W;
    a.Do = (D - 1) / sz + 1; a.Ho = (H - 1) / stride + 1; a.Wo = (W - 1) / stride + 1;
    a.Cout = Cout; a.Cin = Cin;
    a.TZ = q.TZ; a.TY = q.TY; a.IZ = q.IZ; a.IY = q.IY; a.IX = q.IX;
    a.ntx = q.ntx; a.nty = q.nty; a.ntz = q.ntz; a.ntiles = q.ntiles;
    a.NG = q.NG; a.NCG = q.NCG; a.P = q.P; a.dys = q.dys; a.xcs = q.xcs; a.slope = lrelu_slope;
    const unsigned nblk = (unsigned)(q.NG * q.NCG * q.P);
    hipStream_t st = (hipStream_t)stream;
    int rc;
#define WG(SS, TT, YY, ZZ) rc = launch_wgrad<SS, TT, YY, ZZ>(a, nblk, q.lds_bytes, st)
#define WP(SS, TT, YY, ZZ) rc = launch_wgrad_pipe<SS, TT, YY, ZZ>(a, nblk, st)
#define WA(SS, TT, YY, ZZ, KK, SZZ) rc = launch_wgrad_pipe<SS, TT, YY, ZZ, KK, SZZ>(a, nblk, st)
    const bool old_form = wgrad_old_form();
    if (!iso && stride == 1) {                       // kernel [1,3,3], stride 1
        if (q.TX == 32) WA(1, 32, 4, 2, 1, 1); else if (q.TX == 16) WA(1, 16, 8, 2, 1, 1); else if (q.TX == 8) WA(1, 8, 8, 4, 1, 1); else WA(1, 4, 4, 4, 1, 1);
    } else if (!iso && kz == 1) {                    // kernel [1,3,3], stride (1,2,2)
        if (q.TX == 16) WA(2, 16, 2, 2, 1, 1); else if (q.TX == 8) WA(2, 8, 4, 2, 1, 1); else WA(2, 4, 4, 4, 1, 1);
    } else if (!iso) {                               // kernel [3,3,3], stride (1,2,2)
        if (q.TX == 16) WA(2, 16, 2, 2, 3, 1); else if (q.TX == 8) WA(2, 8, 4, 2, 3, 1); else WA(2, 4, 4, 4, 3, 1);
    }
    else if (stride == 2 && !old_form) { if (q.TX == 16) WP(2, 16, 2, 2); else if (q.TX == 8) WP(2, 8, 4, 2); else WP(2, 4, 4, 4); }
    else if (stride == 1 && !old_form) { if (q.TX == 32) WP(1, 32, 4, 2); else if (q.TX == 16) WP(1, 16, 8, 2); else if (q.TX == 8) WP(1, 8, 8, 4); else WP(1, 4, 4, 4); }
    else if (stride == 1) { if (q.TX == 32) WG(1, 32, 4, 2); else if (q.TX == 16) WG(1, 16, 8, 2); else if (q.TX == 8) WG(1, 8, 8, 4); else WG(1, 4, 4, 4); }
    else             { if (q.TX == 32) WG(2, 32, 4, 2); else if (q.TX == 16) WG(2, 16, 4, 2); else if (q.TX == 8) WG(2, 8, 8, 2); else WG(2, 4, 4, 4); }
#undef WG
#undef WP
#undef WA
    if (rc != WMH_OK) return rc;
    const size_t n = (size_t)Cout * Cin * 9 * kz;
    const unsigned rb = wmh_slab_reduce_blocks(n);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(rb), dim3(256), 0, st, workspace, dW, n, q.P);
    return wmh_check_launch("wgrad_reduce_kernel");
}
extern "C" int wmh_conv3d_k3_wgrad(const float* x0, const float* ss0, int C0, const float* x1, const float* ss1, int C1,
                                   const float* dy, float* dW, float* workspace, int N, int D, int H, int W, int Cout,
                                   int stride, float lrelu_slope, void* stream) {
    return wmh_conv3d_wgrad(x0, ss0, C0, x1, ss1, C1, dy, dW, workspace, N, D, H, W, Cout, 3, stride, stride, lrelu_slope, stream);
}
