// conv3d_k3_wgrad_bf16.hip -- weight gradient of the 3x3x3 convolution (pad 1, stride 1) with bf16 matrix-core operands,
// for mixed-precision training (the reference trains under AMP): dy and the activated input x' are rounded to bf16 on
// their way into LDS, the products accumulate in fp32 on v_mfma_f32_32x32x16_bf16, everything else (slab reduction,
// the gradient itself) stays fp32.
//
//   D[co 32][tap 32 (27 used)] += A[co][16 voxels] * B[16 voxels][tap]        one D tile per input channel
//
// Voxels are the K dimension, 16 consecutive x positions per instruction, so a lane's B operand is 8 consecutive x of the
// halo row shifted by its tap's dx.  A 2-byte shift cannot be a 16-byte LDS read, hence the halo tile of a channel is
// kept as THREE copies shifted by dx = 0, 1, 2 (a few KB per channel): every operand is one aligned ds_read_b128.
// Same organisation as conv3d_k3_wgrad_pipe_kernel (wave w owns ci = 4c + w, c < 4: a workgroup covers 32 co x 16 ci --
// four accumulators per wave leave room for the prefetch registers --, wave-private double-buffered x' tiles,
// register prefetch of the next chunk under the MFMAs, dy tile shared by the workgroup, one slab per workgroup and a
// fixed-order reduction).  1/8 of the fp32 kernel's matrix instructions: the kernel is bound by staging, i.e. by HBM.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "wmh_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 wb_bf16x2 __attribute__((ext_vector_type(2)));
typedef float wb_f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned short wb_bf16(float v) {           // v_cvt_pk_bf16_f32: round to nearest even
    const wb_f32x2 f = {v, v};
    const wb_bf16x2 b = __builtin_convertvector(f, wb_bf16x2);
    return (unsigned short)(__builtin_bit_cast(unsigned, b) & 0xFFFFu);
}

struct WgradBArgs {
    const float* x0; const float* ss0; int C0;
    const float* x1; const float* ss1; int C1;
    const float* dy; float* part;
    int N, D, H, W, Cout, Cin;
    int ntx, nty, ntz, ntiles;
    int NG, NCG, P;
    float slope;
};

template <int TX, int TY, int TZ>
__global__ __launch_bounds__(256, 2) void conv3d_k3_wgrad_bf16_kernel(const WgradBArgs a) {
    constexpr int IX = TX + 2, IY = TY + 2, IZ = TZ + 2, R = IY * IZ;
    constexpr int TV = TZ * TY * TX, ZS = IY * IX, SPX = IZ * ZS;
    constexpr int NK = (SPX + 63) / 64;
    constexpr int NDY = TV / 8, CPK = 256 / TV;
    constexpr int DYSB = TV * 2 + 16;                          // bytes per dy row (one output channel)
    constexpr int RSB = TX * 2 + 16, CSTB = R * RSB + 16, BUFB = 3 * CSTB;   // x': row, shifted copy, buffer (bytes)
    static_assert(TX % 16 == 0 && TV <= 256 && 256 % TV == 0, "tile geometry");
    extern __shared__ __attribute__((aligned(16))) char ldsb[];
    char* dyl = ldsb;                                          // [32 co][DYSB]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* xw = ldsb + 32 * DYSB + wave * 2 * BUFB;             // [2][3 copies][R][RSB], private to this wave
    const int i = lane & 31, kh = lane >> 5;
    int bid = blockIdx.x;
    const int p = bid % a.P; bid /= a.P;
    const int g = bid % a.NG, cg = bid / a.NG;
    constexpr int NCH = 4;                                     // channels (accumulators) per wave
    const int nchunks = min(NCH, (a.Cin - cg * 4 * NCH + 3) / 4);

    const int tapj = min(i, 26);
    const int LB = (tapj % 3) * CSTB + ((tapj / 9) * IY + (tapj / 3) % 3) * RSB + kh * 16;
    const int LA = i * DYSB + kh * 16;
    const int HW = a.H * a.W;
    const size_t DHW = (size_t)a.D * HW;
    const int dv = tid % TV, dcol0 = tid / TV;
    const int dvx = dv % TX, dvy = (dv / TX) % TY, dvz = dv / (TX * TY);

    // halo element s = lane + 64k of a channel sits at (row, hx) = (s / IX, s % IX); walked incrementally in write_x
    const int hx0 = lane % IX, row0 = lane / IX;

    f32x16 acc[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

    unsigned coff[NK]; unsigned okm = 0; unsigned dvoff = 0; bool dok = false; int tn = 0;
    auto tile_desc = [&](int t) __attribute__((always_inline)) {
        int b = t;
        const int tx = b % a.ntx; b /= a.ntx;
        const int ty = b % a.nty; b /= a.nty;
        const int tz = b % a.ntz; b /= a.ntz;
        tn = __builtin_amdgcn_readfirstlane(b);
        const int ox0 = __builtin_amdgcn_readfirstlane(tx * TX), oy0 = __builtin_amdgcn_readfirstlane(ty * TY),
                  oz0 = __builtin_amdgcn_readfirstlane(tz * TZ);
        okm = 0;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int s = lane + 64 * k;
            const int gx = ox0 - 1 + s % IX, gy = oy0 - 1 + (s / IX) % IY, gz = oz0 - 1 + s / ZS;
            const bool ok = (k * 64 + 63 < SPX || s < SPX) && (unsigned)gx < (unsigned)a.W && (unsigned)gy < (unsigned)a.H &&
                            (unsigned)gz < (unsigned)a.D;
            coff[k] = ok ? 4u * (unsigned)(gz * HW + gy * a.W + gx) : 0u;
            okm |= ok ? (1u << k) : 0u;
        }
        const int ox = ox0 + dvx, oy = oy0 + dvy, oz = oz0 + dvz;
        dok = ox < a.W && oy < a.H && oz < a.D;
        dvoff = dok ? 4u * (unsigned)(oz * HW + oy * a.W + ox) : 0u;
    };
    float xr[NK], dyr[NDY];
    float sc = 1.f, sh = 0.f, sl = 1.f; unsigned xmask = 0;
    auto issue_x = [&](int c) __attribute__((always_inline)) {
        const int ci = cg * 4 * NCH + 4 * c + wave;
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
        sl = has ? a.slope : 1.f;
        xmask = civ ? okm : 0u;
    };
    auto write_x = [&](int buf) __attribute__((always_inline)) {
        char* dst = xw + buf * BUFB;
        int hx = hx0, row = row0;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            float v = fmaf(xr[k], sc, sh);
            v = fmaxf(v, v * sl);                               // LeakyReLU for 0 <= slope <= 1 (host checks)
            const unsigned short hv = ((xmask >> k) & 1u) ? wb_bf16(v) : (unsigned short)0;
            if (k > 0) {                                        // s += 64
                hx += 64 % IX;
                const int wrap = hx >= IX ? 1 : 0;
                hx -= wrap * IX;
                row += 64 / IX + wrap;
            }
            if (k * 64 + 63 < SPX || lane + 64 * k < SPX) {
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int x = hx - dx;
                    if (x >= 0 && x < TX) *reinterpret_cast<unsigned short*>(dst + dx * CSTB + row * RSB + x * 2) = hv;
                }
            }
        }
    };
    const unsigned cstride = 4u * (unsigned)DHW;
    auto issue_dy = [&]() __attribute__((always_inline)) {
        const char* src = (const char*)(a.dy + (size_t)tn * a.Cout * DHW);
#pragma unroll
        for (int k = 0; k < NDY; ++k) {
            const unsigned co = (unsigned)min(g * 32 + k * CPK + dcol0, a.Cout - 1);
            dyr[k] = *(const float*)(src + (co * cstride + dvoff));
        }
    };
    auto write_dy = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < NDY; ++k) {
            const int col = k * CPK + dcol0;
            const float v = (dok && g * 32 + col < a.Cout) ? dyr[k] : 0.f;
            *reinterpret_cast<unsigned short*>(dyl + col * DYSB + dv * 2) = wb_bf16(v);
        }
    };

    tile_desc(p);
    issue_dy(); issue_x(0);
    write_dy(); write_x(0);
    __syncthreads();
    for (int t = p; t < a.ntiles; t += a.P) {
        const bool more = t + a.P < a.ntiles;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (c < NCH - 1) { if (c + 1 < nchunks) issue_x(c + 1); }
            else if (more) { tile_desc(t + a.P); issue_dy(); issue_x(0); }
            __builtin_amdgcn_sched_barrier(0x38F);
            if (c < nchunks) {
                const char* a0 = dyl + LA;
                const char* b0 = xw + (c & 1) * BUFB + LB;
#pragma unroll
                for (int z = 0; z < TZ; ++z)
#pragma unroll
                    for (int y = 0; y < TY; ++y)
#pragma unroll
                        for (int s = 0; s < TX / 16; ++s) {
                            const bf16x8 av = *reinterpret_cast<const bf16x8*>(a0 + ((z * TY + y) * TX + 16 * s) * 2);
                            const bf16x8 bv = *reinterpret_cast<const bf16x8*>(b0 + (z * IY + y) * RSB + s * 32);
                            acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[c], 0, 0, 0);
                        }
            }
            __builtin_amdgcn_sched_barrier(0x38F);
            if (c < NCH - 1) {
                if (c + 1 < nchunks) {
                    write_x((c + 1) & 1);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
            } else {
                __syncthreads();
                if (more) { write_dy(); write_x(0); }
                __syncthreads();
            }
        }
    }
    if (i < 27) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int ci = cg * 4 * NCH + 4 * c + wave;
            if (ci >= a.Cin) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = g * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (co < a.Cout) a.part[(((size_t)p * a.Cout + co) * a.Cin + ci) * 27 + i] = acc[c][r];
            }
        }
    }
}

__global__ __launch_bounds__(256) void wgrad_bf16_reduce_kernel(const float* part, float* dW, size_t n, int P) { wmh_slab_reduce_body(part, dW, n, P); }   // wmh_common.h

struct WBPlan { int TX, TY, TZ, ntx, nty, ntz, ntiles, NG, NCG, P; };
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static bool plan_wb(int Cin, int Cout, int N, int D, int H, int W, WBPlan& q) {
    if (W >= 24) { q.TX = 32; q.TY = 4; q.TZ = 2; } else if (W >= 12) { q.TX = 16; q.TY = 8; q.TZ = 2; } else return false;
    q.ntx = cdiv(W, q.TX); q.nty = cdiv(H, q.TY); q.ntz = cdiv(D, q.TZ);
    q.ntiles = N * q.ntx * q.nty * q.ntz;
    q.NG = cdiv(Cout, 32); q.NCG = cdiv(Cin, 16);
    q.P = cdiv(512, q.NG * q.NCG); if (q.P > q.ntiles) q.P = q.ntiles; if (q.P < 1) q.P = 1;
    return true;
}
// 1 when the bf16 form serves this layer (stride 1, rows of >= 12 voxels); the small layers stay on the fp32 kernel
extern "C" int wmh_conv3d_k3_wgrad_bf16_supported(int Cin, int Cout, int N, int D, int H, int W, int stride) {
    WBPlan q;
    return stride == 1 && Cin >= 1 && Cout >= 1 && N >= 1 && plan_wb(Cin, Cout, N, D, H, W, q) ? 1 : 0;
}
extern "C" size_t wmh_conv3d_k3_wgrad_bf16_workspace_floats(int Cin, int Cout, int N, int D, int H, int W) {
    WBPlan q;
    if (!plan_wb(Cin, Cout, N, D, H, W, q)) return 0;
    return (size_t)q.P * Cout * Cin * 27;
}
template <int TX, int TY, int TZ>
static int launch_wb(const WgradBArgs& a, unsigned nblk, hipStream_t st) {
    constexpr int R = (TY + 2) * (TZ + 2), RSB = TX * 2 + 16, CSTB = R * RSB + 16;
    constexpr int lds = 32 * (TX * TY * TZ * 2 + 16) + 8 * 3 * CSTB;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_k3_wgrad_bf16_kernel<TX, TY, TZ>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((conv3d_k3_wgrad_bf16_kernel<TX, TY, TZ>), dim3(nblk), dim3(256), lds, st, a);
    return wmh_check_launch("conv3d_k3_wgrad_bf16_kernel");
}
extern "C" int wmh_conv3d_k3_wgrad_bf16(const float* x0, const float* ss0, int C0, const float* x1, const float* ss1, int C1, const float* dy,
                                        float* dW, float* workspace, int N, int D, int H, int W, int Cout, float lrelu_slope, void* stream) {
    if (!x0 || !dy || !dW || !workspace || C0 < 1 || N < 1 || Cout < 1) return wmh_fail(WMH_EINVAL, "conv3d_k3_wgrad_bf16: bad args");
    if (!(lrelu_slope >= 0.f && lrelu_slope <= 1.f)) return wmh_fail(WMH_EINVAL, "conv3d_k3_wgrad_bf16: LeakyReLU slope must be in [0, 1]");
    if (!x1) C1 = 0;
    const int Cin = C0 + C1;
    WBPlan q;
    if (!plan_wb(Cin, Cout, N, D, H, W, q)) return wmh_fail(WMH_EINVAL, "conv3d_k3_wgrad_bf16: layer not served (rows shorter than 12 voxels)");
    if ((size_t)Cout * D * H * W * 4 >= ((size_t)1 << 32)) return wmh_fail(WMH_EINVAL, "conv3d_k3_wgrad_bf16: volume too large for 32-bit lane offsets");
    WgradBArgs a;
    a.x0 = x0; a.ss0 = ss0; a.C0 = C0; a.x1 = x1; a.ss1 = ss1; a.C1 = C1; a.dy = dy; a.part = workspace;
    a.N = N; a.D = D; a.H = H; a.W = W; a.Cout = Cout; a.Cin = Cin;
    a.ntx = q.ntx; a.nty = q.nty; a.ntz = q.ntz; a.ntiles = q.ntiles; a.NG = q.NG; a.NCG = q.NCG; a.P = q.P; a.slope = lrelu_slope;
    const unsigned nblk = (unsigned)(q.NG * q.NCG * q.P);
    hipStream_t st = (hipStream_t)stream;
    const int rc = q.TX == 32 ? launch_wb<32, 4, 2>(a, nblk, st) : launch_wb<16, 8, 2>(a, nblk, st);
    if (rc != WMH_OK) return rc;
    const size_t n = (size_t)Cout * Cin * 27;
    const unsigned rb = wmh_slab_reduce_blocks(n);
    hipLaunchKernelGGL(wgrad_bf16_reduce_kernel, dim3(rb), dim3(256), 0, st, workspace, dW, n, q.P);
    return wmh_check_launch("wgrad_bf16_reduce_kernel");
}
