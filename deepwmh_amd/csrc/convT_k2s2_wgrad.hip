// convT_k2s2_wgrad.hip -- weight gradient of ConvTranspose3d(kernel 2, stride 2) on the exact-fp32 MFMA.
//
//   dW[ci, co, t] = sum_{n, m} x'[n, ci, m] * dY[n, co, 2m + t]          t = (kz, ky, kx) in {0,1}^3
//
// The contraction runs over the low-resolution voxels m, so they are the K dimension of
// v_mfma_f32_32x32x2_f32: D[ci 32][co 32] += A[ci][voxel pair] * B[voxel pair][co], one D tile per tap.
// A workgroup owns 32 ci x 32 co; wave w owns the taps (ky, kx) = (w >> 1, w & 1) for kz = 0, 1 (two
// accumulators).  Per tile of 64 low-resolution voxels the x' tile [32][64] (producer's InstanceNorm +
// LeakyReLU applied on the way in) and the dY tile, de-interleaved into [32 co][8 taps][64], are staged
// through LDS; the next tile's loads are issued before the MFMAs of the current one and land in registers
// underneath them.  The layer is HBM-bound (dY is read once per group of 32 input channels).
// Per-workgroup partial dW go to slabs that are summed in fixed order (bit-reproducible).
// Stands in for the weight half of torch.nn.ConvTranspose3d's backward [nnU-Net training, not in tree].
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "wmh_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct CTWArgs {
    const float* x; const float* ss; const float* dy; float* part;
    int N, Cin, Cout, D, H, W;
    int ntx, nty, ntz, ntiles, NG, NCG, P;
    float slope;
};

// KT: kernel = stride along the first spatial axis (2, or 1 for nnU-Net's [1,2,2] pooling stages: 4 taps, one accumulator)
template <int TX, int TY, int TZ, int KT = 2>
__global__ __launch_bounds__(256, 2) void convT_k2s2_wgrad_kernel(const CTWArgs a) {
    constexpr int TVL = TX * TY * TZ, XS = TVL + 1, NTAP = 4 * KT, YS = NTAP * TVL + 1;
    constexpr int RT = 128 * KT, CP = 256 / RT, NKY = 32 / CP;     // threads per dY channel, channels per pass, passes
    static_assert(TVL == 64, "64 low-resolution voxels per tile");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* xs = lds;                   // [32 ci][XS]
    float* ys = lds + 32 * XS;         // [32 co][8 taps][TVL] (+1 per co)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, kh = lane >> 5;
    int bid = blockIdx.x;
    const int p = bid % a.P; bid /= a.P;
    const int g = bid % a.NG, cg = bid / a.NG;
    const int H2 = 2 * a.H, W2 = 2 * a.W;
    const size_t DHW = (size_t)a.D * a.H * a.W;

    // staging geometry: dY -- thread = (x, hi-res row) of one output channel per pass; x' -- thread = voxel, wave = channel
    const int t2 = tid % RT, cpi = tid / RT;
    const int sx = t2 % TX, sr = t2 / TX, sy2 = sr % (2 * TY), sz2 = sr / (2 * TY);      // sz2: high-resolution plane of the tile
    const int szl = KT == 2 ? (sz2 >> 1) : sz2;                                           // its low-resolution plane
    const int stap = (KT == 2 ? (sz2 & 1) * 4 : 0) + (sy2 & 1) * 2, sm = (szl * TY + (sy2 >> 1)) * TX + sx;
    const int mx = lane % TX, my = (lane / TX) % TY, mz = lane / (TX * TY);

    f32x16 acc[KT];
#pragma unroll
    for (int c = 0; c < KT; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

    int tn = 0; unsigned yoff = 0, xoff = 0; bool yok = false, xok = false;
    auto desc = [&](int t) {
        int b = t;
        const int tx = b % a.ntx; b /= a.ntx;
        const int ty = b % a.nty; b /= a.nty;
        const int tz = b % a.ntz; b /= a.ntz;
        tn = __builtin_amdgcn_readfirstlane(b);
        const int x0 = __builtin_amdgcn_readfirstlane(tx * TX), y0 = __builtin_amdgcn_readfirstlane(ty * TY),
                  z0 = __builtin_amdgcn_readfirstlane(tz * TZ);
        yok = x0 + sx < a.W && y0 + (sy2 >> 1) < a.H && z0 + szl < a.D;
        yoff = yok ? 4u * (unsigned)(((KT * z0 + sz2) * H2 + 2 * y0 + sy2) * W2 + 2 * (x0 + sx)) : 0u;
        xok = x0 + mx < a.W && y0 + my < a.H && z0 + mz < a.D;
        xoff = xok ? 4u * (unsigned)(((z0 + mz) * a.H + y0 + my) * a.W + x0 + mx) : 0u;
    };
    float2 yr[NKY]; float xr[8]; int wn = 0; bool wyok = false, wxok = false;
    const unsigned ycs = 4u * (unsigned)NTAP * (unsigned)DHW;               // bytes between channels of dY (host checks the range)
    const unsigned xcs = 4u * (unsigned)DHW;
    auto issue = [&]() {
        const char* ysrc = (const char*)(a.dy + (size_t)tn * a.Cout * NTAP * DHW);
#pragma unroll
        for (int k = 0; k < NKY; ++k) {
            const unsigned co = (unsigned)min(g * 32 + k * CP + cpi, a.Cout - 1);
            yr[k] = *(const float2*)(ysrc + (co * ycs + yoff));
        }
        const char* xsrc = (const char*)(a.x + (size_t)tn * a.Cin * DHW);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned ci = (unsigned)min(cg * 32 + wave + 4 * k, a.Cin - 1);
            xr[k] = *(const float*)(xsrc + (ci * xcs + xoff));
        }
        wn = tn; wyok = yok; wxok = xok;
    };
    auto write = [&]() {
#pragma unroll
        for (int k = 0; k < NKY; ++k) {
            const int col = k * CP + cpi;
            const bool ok = wyok && g * 32 + col < a.Cout;
            ys[col * YS + stap * TVL + sm] = ok ? yr[k].x : 0.f;
            ys[col * YS + (stap + 1) * TVL + sm] = ok ? yr[k].y : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int ci = cg * 32 + wave + 4 * k, cic = min(ci, a.Cin - 1);
            float v = xr[k];
            if (a.ss) {
                const float* q = a.ss + ((size_t)wn * a.Cin + cic) * 2;
                v = fmaf(v, q[0], q[1]);
                v = v > 0.f ? v : v * a.slope;
            }
            xs[(wave + 4 * k) * XS + lane] = (wxok && ci < a.Cin) ? v : 0.f;
        }
    };

    desc(p); issue();
    for (int t = p; t < a.ntiles; t += a.P) {
        __syncthreads();                                        // the previous tile's MFMAs are done with LDS
        write();
        __syncthreads();
        if (t + a.P < a.ntiles) { desc(t + a.P); issue(); }
        __builtin_amdgcn_sched_barrier(0x38F);                  // the loads stay here, above the MFMAs
        const float* ap = xs + i * XS + kh;
        const float* bp = ys + i * YS + wave * TVL + kh;
#pragma unroll
        for (int pp = 0; pp < TVL / 2; ++pp) {
            const float av = ap[2 * pp];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bp[2 * pp], acc[0], 0, 0, 0);
            if constexpr (KT == 2) acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bp[4 * TVL + 2 * pp], acc[1], 0, 0, 0);
        }
    }
    const int co = g * 32 + i;
    if (co < a.Cout) {
#pragma unroll
        for (int kz = 0; kz < KT; ++kz)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = cg * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (ci < a.Cin) a.part[(((size_t)p * a.Cin + ci) * a.Cout + co) * NTAP + kz * 4 + wave] = acc[kz][r];
            }
    }
}

__global__ __launch_bounds__(256) void convT_wgrad_reduce_kernel(const float* part, float* dW, size_t n, int P) { wmh_slab_reduce_body(part, dW, n, P); }   // wmh_common.h

struct CTWPlan { int TX, TY, TZ, ntx, nty, ntz, ntiles, NG, NCG, P; };
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static CTWPlan plan_ctw(int Cin, int Cout, int N, int D, int H, int W) {
    CTWPlan q;
    if (W >= 12) { q.TX = 16; q.TY = 2; q.TZ = 2; } else if (W >= 6) { q.TX = 8; q.TY = 4; q.TZ = 2; } else { q.TX = 4; q.TY = 4; q.TZ = 4; }
    q.ntx = cdiv(W, q.TX); q.nty = cdiv(H, q.TY); q.ntz = cdiv(D, q.TZ);
    q.ntiles = N * q.ntx * q.nty * q.ntz;
    q.NG = cdiv(Cout, 32); q.NCG = cdiv(Cin, 32);
    q.P = cdiv(512, q.NG * q.NCG); if (q.P > q.ntiles) q.P = q.ntiles; if (q.P < 1) q.P = 1;
    return q;
}

// kt: kernel = stride along the first spatial axis (2, or 1 for the [1,2,2] stages); D, H, W: the INPUT (low-resolution) size
extern "C" size_t wmh_convtranspose3d_wgrad_workspace_floats(int Cin, int Cout, int N, int D, int H, int W, int kt) {
    if (kt != 1 && kt != 2) return 0;
    const CTWPlan q = plan_ctw(Cin, Cout, N, D, H, W);
    return (size_t)q.P * Cin * Cout * 4 * kt;
}
extern "C" size_t wmh_convtranspose3d_k2s2_wgrad_workspace_floats(int Cin, int Cout, int N, int D, int H, int W) {
    return wmh_convtranspose3d_wgrad_workspace_floats(Cin, Cout, N, D, H, W, 2);
}

template <int TX, int TY, int TZ, int KT>
static int launch_ctw(const CTWArgs& a, unsigned nblk, hipStream_t st) {
    constexpr int lds = (32 * (TX * TY * TZ + 1) + 32 * (4 * KT * TX * TY * TZ + 1)) * 4;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&convT_k2s2_wgrad_kernel<TX, TY, TZ, KT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((convT_k2s2_wgrad_kernel<TX, TY, TZ, KT>), dim3(nblk), dim3(256), lds, st, a);
    return wmh_check_launch("convT_k2s2_wgrad_kernel");
}

extern "C" int wmh_convtranspose3d_wgrad(const float* x, const float* ss, const float* dY, float* dW, float* workspace,
                                         int N, int Cin, int Cout, int D, int H, int W, int kt, float slope, void* stream) {
    if (!x || !dY || !dW || !workspace || N < 1 || Cin < 1 || Cout < 1 || D < 1 || H < 1 || W < 1 || (kt != 1 && kt != 2))
        return wmh_fail(WMH_EINVAL, "convtranspose3d_wgrad: bad args");
    if ((size_t)Cout * 4 * kt * D * H * W * 4 >= ((size_t)1 << 32) || (size_t)Cin * D * H * W * 4 >= ((size_t)1 << 32))
        return wmh_fail(WMH_EINVAL, "convtranspose3d_wgrad: volume too large for 32-bit lane offsets");
    const CTWPlan q = plan_ctw(Cin, Cout, N, D, H, W);
    CTWArgs a;
    a.x = x; a.ss = ss; a.dy = dY; a.part = workspace;
    a.N = N; a.Cin = Cin; a.Cout = Cout; a.D = D; a.H = H; a.W = W;
    a.ntx = q.ntx; a.nty = q.nty; a.ntz = q.ntz; a.ntiles = q.ntiles; a.NG = q.NG; a.NCG = q.NCG; a.P = q.P; a.slope = slope;
    const unsigned nblk = (unsigned)(q.NG * q.NCG * q.P);
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (kt == 2) {
        if (q.TX == 16) rc = launch_ctw<16, 2, 2, 2>(a, nblk, st);
        else if (q.TX == 8) rc = launch_ctw<8, 4, 2, 2>(a, nblk, st);
        else rc = launch_ctw<4, 4, 4, 2>(a, nblk, st);
    } else {
        if (q.TX == 16) rc = launch_ctw<16, 2, 2, 1>(a, nblk, st);
        else if (q.TX == 8) rc = launch_ctw<8, 4, 2, 1>(a, nblk, st);
        else rc = launch_ctw<4, 4, 4, 1>(a, nblk, st);
    }
    if (rc != WMH_OK) return rc;
    const size_t n = (size_t)Cin * Cout * 4 * kt;
    const unsigned rb = wmh_slab_reduce_blocks(n);
    hipLaunchKernelGGL(convT_wgrad_reduce_kernel, dim3(rb), dim3(256), 0, st, workspace, dW, n, q.P);
    return wmh_check_launch("convT_wgrad_reduce_kernel");
}
extern "C" int wmh_convtranspose3d_k2s2_wgrad(const float* x, const float* ss, const float* dY, float* dW, float* workspace,
                                              int N, int Cin, int Cout, int D, int H, int W, float slope, void* stream) {
    return wmh_convtranspose3d_wgrad(x, ss, dY, dW, workspace, N, Cin, Cout, D, H, W, 2, slope, stream);
}
