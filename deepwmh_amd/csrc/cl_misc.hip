// cl_misc.hip -- the bandwidth-bound kernels of the channels-last bf16 path (cl_common.h): layout converters, the first
// convolution (1 input channel, fp32 NCDHW patches in), the 2x2x2 transposed convolution and the segmentation head with
// softmax / mirror averaging / Gaussian accumulation.  All of them stream a CL16 tensor with 16-byte pieces per lane.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "cl_common.h"

static inline unsigned cl_blocks(size_t n, int per, unsigned cap) {
    size_t b = (n + per - 1) / per;
    return (unsigned)(b > cap ? cap : (b < 1 ? 1 : b));
}

// ----------------------------------------------------------------------------------------------------------------
// Layout converters between NCDHW fp32 and CL16 (raw values; a pending scale/shift pair stays valid).  Used where the
// channels-last levels meet the levels that stay on the NCDHW kernels (rows shorter than 32 voxels), and by the tests.
// ----------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cl_from_ncdhw_kernel(const float* x, uint16_t* y, int C, size_t S, size_t total) {
    const int C8 = C / 8;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t v = i % S;                     // consecutive lanes: consecutive voxels (coalesced channel-plane reads)
        size_t r = i / S;
        const int c8 = (int)(r % C8);
        const size_t n = r / C8;
        const float* xp = x + ((size_t)n * C + 8 * c8) * S + v;
        cl_u32x4 w;
#pragma unroll
        for (int p = 0; p < 4; ++p) w[p] = cl_pack2(xp[(size_t)(2 * p) * S], xp[(size_t)(2 * p + 1) * S]);
        *reinterpret_cast<cl_u32x4*>(y + cl_index(n, C, S, c8, v)) = w;
    }
}
__global__ __launch_bounds__(256) void cl_to_ncdhw_kernel(const uint16_t* x, float* y, int C, size_t S, size_t total) {
    const int C8 = C / 8;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t v = i % S;
        size_t r = i / S;
        const int c8 = (int)(r % C8);
        const size_t n = r / C8;
        const cl_u32x4 u = *reinterpret_cast<const cl_u32x4*>(x + cl_index(n, C, S, c8, v));
        float* yp = y + ((size_t)n * C + 8 * c8) * S + v;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            yp[(size_t)(2 * p) * S] = __builtin_bit_cast(float, u[p] << 16);
            yp[(size_t)(2 * p + 1) * S] = __builtin_bit_cast(float, u[p] & 0xFFFF0000u);
        }
    }
}
extern "C" int wmh_cl_from_ncdhw(const float* x, void* y, int N, int C, size_t spatial, void* stream) {
    if (!x || !y || N < 1 || C < 8 || C % 8 || spatial < 1) return wmh_fail(WMH_EINVAL, "cl_from_ncdhw: bad args (C must be a multiple of 8)");
    const size_t total = (size_t)N * (C / 8) * spatial;
    hipLaunchKernelGGL(cl_from_ncdhw_kernel, dim3(cl_blocks(total, 256, 8192)), dim3(256), 0, (hipStream_t)stream, x,
                       static_cast<uint16_t*>(y), C, spatial, total);
    return wmh_check_launch("cl_from_ncdhw_kernel");
}
extern "C" int wmh_cl_to_ncdhw(const void* x, float* y, int N, int C, size_t spatial, void* stream) {
    if (!x || !y || N < 1 || C < 8 || C % 8 || spatial < 1) return wmh_fail(WMH_EINVAL, "cl_to_ncdhw: bad args (C must be a multiple of 8)");
    const size_t total = (size_t)N * (C / 8) * spatial;
    hipLaunchKernelGGL(cl_to_ncdhw_kernel, dim3(cl_blocks(total, 256, 8192)), dim3(256), 0, (hipStream_t)stream,
                       static_cast<const uint16_t*>(x), y, C, spatial, total);
    return wmh_check_launch("cl_to_ncdhw_kernel");
}

// ----------------------------------------------------------------------------------------------------------------
// First convolution: 1 input channel (fp32 NCDHW patches, nothing pending on them) -> 32*G output channels, CL16.
// K = 27 taps padded to 32 = two k-steps of v_mfma_f32_32x32x16_bf16; the B operand is gathered from an fp32 halo tile
// in LDS (8 ds_read_b32 + 4 v_cvt_pk per k-step).  Output-bandwidth bound: 64 B written per voxel against 4 B read.
// Workgroup = 4 waves, tile 4 x 8 x 32 outputs, wave w owns z-slice w (8 rows = 8 M-tiles).
// ----------------------------------------------------------------------------------------------------------------
struct ClFirstArgs {
    const float* x; const uint16_t* wf; const float* bias; uint16_t* y; float* part;
    int N, D, H, W, Cout, ntx, nty, ntz, npart, G;
};
__global__ __launch_bounds__(256) void cl_conv3_first_kernel(const ClFirstArgs a) {
    constexpr int TZ = 4, TY = 8, IZ = 6, IY = 10, IX = 34, NH = IZ * IY * IX;
    __shared__ float halo[NH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, kh = lane >> 5;
    int t = blockIdx.x;
    const int g = t % a.G; t /= a.G;
    const int tx = t % a.ntx; t /= a.ntx;
    const int ty = t % a.nty; t /= a.nty;
    const int tz = t % a.ntz; t /= a.ntz;
    const int n = t;
    const int HW = a.H * a.W;
    const float* xn = a.x + (size_t)n * a.D * HW;
    const int z0 = tz * TZ - 1, y0 = ty * TY - 1, x0 = tx * 32 - 1;
    for (int e = tid; e < NH; e += 256) {
        const int iz = e / (IY * IX), rem = e - iz * (IY * IX), iy = rem / IX, ix = rem - iy * IX;
        const int gz = z0 + iz, gy = y0 + iy, gx = x0 + ix;
        const bool ok = (unsigned)gz < (unsigned)a.D && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
        halo[e] = ok ? xn[(size_t)gz * HW + gy * a.W + gx] : 0.f;
    }
    // A operand: this lane's row (output channel g*32 + j) of the [32][32] weight matrix, k = 16 s + 8 kh + e
    const cl_bf16x8* wp = reinterpret_cast<const cl_bf16x8*>(a.wf) + (size_t)g * 128 + lane;
    const cl_bf16x8 a0 = wp[0], a1 = wp[64];
    // element offsets of this lane's 16 taps relative to its M-tile's first halo voxel (taps 27..31 are padding: they
    // read a valid address and meet zero weights)
    int toff[16];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int kA = 16 * s + e, kB = 16 * s + 8 + e;
            const int tA = kA < 27 ? kA : 0, tB = kB < 27 ? kB : 0;
            const int oA = ((tA / 9) * IY + (tA / 3) % 3) * IX + tA % 3, oB = ((tB / 9) * IY + (tB / 3) % 3) * IX + tB % 3;
            toff[8 * s + e] = kh ? oB : oA;
        }
    f32x16 acc[TY];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float bv = a.bias ? a.bias[g * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh] : 0.f;
#pragma unroll
        for (int m = 0; m < TY; ++m) acc[m][r] = bv;
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < TY; ++m) {
        const float* hp = halo + (wave * IY + m) * IX + j;
        unsigned bw[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) bw[p] = cl_pack2(hp[toff[2 * p]], hp[toff[2 * p + 1]]);
        const cl_u32x4 b0 = {bw[0], bw[1], bw[2], bw[3]}, b1 = {bw[4], bw[5], bw[6], bw[7]};
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, __builtin_bit_cast(cl_bf16x8, b0), acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, __builtin_bit_cast(cl_bf16x8, b1), acc[m], 0, 0, 0);
    }
    const int oz = tz * TZ + wave, oy0 = ty * TY, ox = tx * 32 + j, cob = g * 32;
    unsigned vmask = 0; int cnt = 0;
#pragma unroll
    for (int m = 0; m < TY; ++m) {
        const bool valid = oz < a.D && oy0 + m < a.H;
        vmask |= valid ? (1u << m) : 0u;
        cnt += valid ? 32 : 0;
    }
    if (a.part != nullptr)
        cl_moments<TY>(a.part, a.npart, a.Cout, n, cob, ((tz * a.nty + ty) * a.ntx + tx) * 4 + wave, acc, vmask, cnt, j, kh);
#pragma unroll
    for (int m = 0; m < TY; ++m)
        if ((vmask >> m) & 1u)
            cl_store_mtile<true>(acc[m], a.y + cl_index(n, a.Cout, (size_t)a.D * HW, cob >> 3, ((size_t)oz * a.H + (oy0 + m)) * a.W + ox), kh, (size_t)a.D * HW * 8);
}
// weight image of the first convolution: [G][2 k-steps][64 lanes][8] bf16; w = torch layout [Cout][1][27] fp32 (host)
extern "C" size_t wmh_cl_conv3_first_packed_halfs(int Cout) { return (size_t)((Cout + 31) / 32) * 2 * 64 * 8; }
extern "C" int wmh_cl_conv3_first_pack_weights(const float* w, int Cout, uint16_t* wf) {
    if (!w || !wf || Cout < 32 || Cout % 32) return wmh_fail(WMH_EINVAL, "cl_conv3_first_pack_weights: Cout must be a multiple of 32");
    for (int g = 0; g < Cout / 32; ++g)
        for (int s = 0; s < 2; ++s)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const int co = g * 32 + (lane & 31), k = 16 * s + 8 * (lane >> 5) + e;
                    const float v = k < 27 ? w[(size_t)co * 27 + k] : 0.f;
                    unsigned u; memcpy(&u, &v, 4);
                    wf[(((size_t)g * 2 + s) * 64 + lane) * 8 + e] = (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
                }
    return WMH_OK;
}
extern "C" int wmh_cl_conv3_first_num_partials(int D, int H, int W) {
    if (D < 1 || H < 1 || W < 32 || W % 32) return wmh_fail(WMH_EINVAL, "cl_conv3_first: rows must be a multiple of 32 voxels");
    return (W / 32) * ((H + 7) / 8) * ((D + 3) / 4) * 4;
}
extern "C" int wmh_cl_conv3_first_fwd(const float* x, const uint16_t* wf, const float* bias, void* y, float* partials,
                                      int N, int D, int H, int W, int Cout, void* stream) {
    if (!x || !wf || !y || N < 1 || D < 1 || H < 1 || W < 32 || W % 32 || Cout < 32 || Cout % 32)
        return wmh_fail(WMH_EINVAL, "cl_conv3_first_fwd: bad args (one input channel, rows and Cout multiples of 32)");
    ClFirstArgs a;
    a.x = x; a.wf = wf; a.bias = bias; a.y = static_cast<uint16_t*>(y); a.part = partials;
    a.N = N; a.D = D; a.H = H; a.W = W; a.Cout = Cout;
    a.ntx = W / 32; a.nty = (H + 7) / 8; a.ntz = (D + 3) / 4; a.G = Cout / 32;
    a.npart = a.ntx * a.nty * a.ntz * 4;
    const long long T = (long long)a.ntx * a.nty * a.ntz * N * a.G;
    if (T > 0x7fffffffLL) return wmh_fail(WMH_EINVAL, "cl_conv3_first_fwd: too many tiles");
    hipLaunchKernelGGL(cl_conv3_first_kernel, dim3((unsigned)T), dim3(256), 0, (hipStream_t)stream, a);
    return wmh_check_launch("cl_conv3_first_kernel");
}

// ----------------------------------------------------------------------------------------------------------------
// ConvTranspose3d(kernel 2, stride 2, no bias) on CL16: out[n, 2z+a, 2y+b, 2x+c, co] = sum_ci f(x[n,z,y,x,ci]) W[ci,co,a,b,c],
// f = the producer's pending InstanceNorm + LeakyReLU.  A GEMM per input voxel with no halo: the B operand (8 channels
// of a lane's voxel) is ONE 16-byte global load, normalised in registers; the eight taps share it.  A wave owns 32
// consecutive input voxels and 32 output channels and writes 8 x 32 voxels x 64 bytes.
// ----------------------------------------------------------------------------------------------------------------
struct ClConvTArgs {
    const uint16_t* x; const float* ss; const uint16_t* wf; uint16_t* y;
    int N, Cin, Cout, D, H, W, G;
    long long tiles;             // M-tiles of 32 voxels over N*D*H*W
    float slope;
};
// NKS = Cin / 16 (4: the plan's 64-channel level; 0 = any, loads inside the k loop).  With a compile-time NKS a tile's
// NKS operand pieces (one per 8-channel plane pair and lane) are loaded together, and the NEXT tile's are issued before this tile's 16 stores
// per lane, so that they do not queue behind them.
template <int NKS>
__global__ __launch_bounds__(256, 2) void cl_convT_kernel(const ClConvTArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, kh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = blockIdx.y;
    const int nks = NKS ? NKS : a.Cin / 16;
    const size_t S = (size_t)a.D * a.H * a.W;
    const cl_bf16x8* wp = reinterpret_cast<const cl_bf16x8*>(a.wf) + (size_t)g * nks * 8 * 64 + lane;
    const long long step = (long long)gridDim.x * 4;
    cl_u32x4 pre[NKS ? NKS : 1];
    auto src_of = [&](long long mt) __attribute__((always_inline)) {
        const size_t v = (size_t)mt * 32 + j, n = v / S;
        return a.x + cl_index(n, a.Cin, S, kh, v - n * S);          // + two planes per k-step
    };
    long long mt = (long long)blockIdx.x * 4 + wave;
    if constexpr (NKS > 0) {
        if (mt < a.tiles) {
            const uint16_t* xv = src_of(mt);
#pragma unroll
            for (int s = 0; s < NKS; ++s) pre[s] = *reinterpret_cast<const cl_u32x4*>(xv + (size_t)(2 * s) * S * 8);
        }
    }
    for (; mt < a.tiles; mt += step) {
        const size_t v = (size_t)mt * 32 + j;                 // flattened (n, z, y, x); S % 32 == 0: a tile stays in one sample
        const size_t n = v / S;
        const int n_u = __builtin_amdgcn_readfirstlane((int)n);
        const size_t r = v - n * S;
        size_t rr = r;
        const int x = (int)(rr % a.W); rr /= a.W;
        const int y = (int)(rr % a.H);
        const int z = (int)(rr / a.H);
        const uint16_t* xv = a.x + cl_index(n, a.Cin, S, kh, r);           // + two planes per k-step
        const float* ssn = a.ss ? a.ss + (size_t)n_u * a.Cin * 2 + 16 * kh : nullptr;
        f32x16 acc[8];
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;
        auto kstep = [&](int s, const cl_u32x4 raw) __attribute__((always_inline)) {
            cl_u32x4 bq = raw;
            if (ssn) {                                        // uniform
                float sc[8], sh[8];
                const float4* sp = reinterpret_cast<const float4*>(ssn + 32 * s);     // {scale, shift} of 8 channels: 64 bytes
#pragma unroll
                for (int q = 0; q < 4; ++q) { const float4 f = sp[q]; sc[2 * q] = f.x; sh[2 * q] = f.y; sc[2 * q + 1] = f.z; sh[2 * q + 1] = f.w; }
                bq = cl_transform8(raw, sc, sh, a.slope);
            }
            const cl_bf16x8 bv = __builtin_bit_cast(cl_bf16x8, bq);
#pragma unroll
            for (int t = 0; t < 8; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wp[(size_t)(s * 8 + t) * 64], bv, acc[t], 0, 0, 0);
        };
        if constexpr (NKS > 0) {
#pragma unroll
            for (int s = 0; s < NKS; ++s) kstep(s, pre[s]);
            if (mt + step < a.tiles) {                        // uniform per wave
                const uint16_t* xn = src_of(mt + step);
#pragma unroll
                for (int s = 0; s < NKS; ++s) pre[s] = *reinterpret_cast<const cl_u32x4*>(xn + (size_t)(2 * s) * S * 8);
            }
        } else {
            for (int s = 0; s < nks; ++s) kstep(s, *reinterpret_cast<const cl_u32x4*>(xv + (size_t)(2 * s) * S * 8));
        }
        const size_t H2 = 2 * (size_t)a.H, W2 = 2 * (size_t)a.W;
        uint16_t* yb = a.y + cl_index(n, a.Cout, 8 * S, g * 4, (((size_t)2 * z) * H2 + 2 * y) * W2 + 2 * x);
#pragma unroll
        for (int t = 0; t < 8; ++t)
            cl_store_mtile(acc[t], yb + (((size_t)(t >> 2) * H2 + ((t >> 1) & 1)) * W2 + (t & 1)) * 8, kh, 8 * S * 8);
    }
}
// weight image: [G][Cin/16][8 taps][64 lanes][8] bf16 from torch layout [Cin][Cout][2][2][2] fp32 (host)
extern "C" size_t wmh_cl_convt_packed_halfs(int Cin, int Cout) { return (size_t)(Cout / 32) * (Cin / 16) * 8 * 64 * 8; }
extern "C" int wmh_cl_convt_pack_weights(const float* w, int Cin, int Cout, uint16_t* wf) {
    if (!w || !wf || Cin < 16 || Cin % 16 || Cout < 32 || Cout % 32) return wmh_fail(WMH_EINVAL, "cl_convT_pack_weights: Cin % 16, Cout % 32");
    const int nks = Cin / 16;
    for (int g = 0; g < Cout / 32; ++g)
        for (int s = 0; s < nks; ++s)
            for (int t = 0; t < 8; ++t)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 8; ++e) {
                        const int co = g * 32 + (lane & 31), ci = 16 * s + 8 * (lane >> 5) + e;
                        const float v = w[((size_t)ci * Cout + co) * 8 + t];
                        unsigned u; memcpy(&u, &v, 4);
                        wf[((((size_t)g * nks + s) * 8 + t) * 64 + lane) * 8 + e] = (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
                    }
    return WMH_OK;
}
extern "C" int wmh_cl_convt_fwd(const void* x, const float* ss, const uint16_t* wf, void* y, int N, int Cin, int Cout,
                                int D, int H, int W, float lrelu_slope, void* stream) {
    if (!x || !wf || !y || N < 1 || D < 1 || H < 1 || W < 1 || Cin < 16 || Cin % 16 || Cout < 32 || Cout % 32)
        return wmh_fail(WMH_EINVAL, "cl_convT_fwd: bad args (Cin % 16, Cout % 32)");
    const size_t S = (size_t)D * H * W;
    if (S % 32) return wmh_fail(WMH_EINVAL, "cl_convT_fwd: D*H*W must be a multiple of 32");
    ClConvTArgs a;
    a.x = static_cast<const uint16_t*>(x); a.ss = ss; a.wf = wf; a.y = static_cast<uint16_t*>(y);
    a.N = N; a.Cin = Cin; a.Cout = Cout; a.D = D; a.H = H; a.W = W; a.G = Cout / 32;
    a.tiles = (long long)(N * S / 32);
    a.slope = lrelu_slope;
    const unsigned gx = cl_blocks((size_t)a.tiles, 4, 4096);
    // (measured, batch 8: 64 -> 32 channels at 64^3 -> 128^3 0.49 -> 0.36 ms with the prefetch; 128 -> 64 at 32^3 -> 64^3 0.14 -> 0.29 ms with
    // eight pieces held per lane, so that level keeps its loads inside the k loop)
    if (Cin == 64) hipLaunchKernelGGL(cl_convT_kernel<4>, dim3(gx, a.G), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(cl_convT_kernel<0>, dim3(gx, a.G), dim3(256), 0, (hipStream_t)stream, a);
    return wmh_check_launch("cl_convT_kernel");
}

// ----------------------------------------------------------------------------------------------------------------
// Segmentation head (1x1x1, no bias) + softmax + mirror average + Gaussian weight + accumulate on a CL16 feature map.
// Same arithmetic, order and rounding as seghead_softmax_accum_kernel (wmh_kernels.hip); a voxel's C channels are
// C/8 16-byte loads of one lane.
// ----------------------------------------------------------------------------------------------------------------
struct ClPatchList { int n; int flip[16]; };
template <int K, int C>
__global__ __launch_bounds__(256) void cl_seghead_accum_kernel(const uint16_t* x, const float* ss, const float* wseg,
                                                               const float* gauss, float* agg, int P0, int P1, int P2,
                                                               int A0, int A1, int A2, int g0, int g1, int g2,
                                                               const ClPatchList pl, float slope) {
    __shared__ float s_ss[16][C][2];
    __shared__ float s_w[K][C];
    const size_t P = (size_t)P0 * P1 * P2, A = (size_t)A0 * A1 * A2;
    for (int i = threadIdx.x; i < pl.n * C * 2; i += 256) (&s_ss[0][0][0])[i] = ss ? ss[i] : ((i & 1) ? 0.f : 1.f);
    for (int i = threadIdx.x; i < K * C; i += 256) (&s_w[0][0])[i] = wseg[i];
    __syncthreads();
    const float wgt = 1.f / (float)pl.n;
    const float sl = ss ? slope : 1.f;
    for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < P; p += (size_t)gridDim.x * 256) {
        const int p2 = (int)(p % P2), p1 = (int)((p / P2) % P1), p0 = (int)(p / ((size_t)P2 * P1));
        float res[K];
#pragma unroll
        for (int k = 0; k < K; ++k) res[k] = 0.f;
        for (int b = 0; b < pl.n; ++b) {
            const int f = pl.flip[b];
            const int s0 = (f & 4) ? P0 - 1 - p0 : p0, s1 = (f & 2) ? P1 - 1 - p1 : p1, s2 = (f & 1) ? P2 - 1 - p2 : p2;
            const size_t src = ((size_t)s0 * P1 + s1) * P2 + s2;
            const cl_u32x4* xp = reinterpret_cast<const cl_u32x4*>(x + cl_index((size_t)b, C, P, 0, src));
            cl_u32x4 u[C / 8];
#pragma unroll
            for (int q = 0; q < C / 8; ++q) u[q] = __builtin_nontemporal_load(xp + (size_t)q * P);      // read once (plane q)
            float lg[K];
#pragma unroll
            for (int k = 0; k < K; ++k) lg[k] = 0.f;
#pragma unroll
            for (int q = 0; q < C / 8; ++q)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int c = 8 * q + e;
                    const unsigned w = u[q][e >> 1];
                    float v = __builtin_bit_cast(float, (e & 1) ? (w & 0xFFFF0000u) : (w << 16));
                    v = fmaf(v, s_ss[b][c][0], s_ss[b][c][1]);
                    v = v > 0.f ? v : v * sl;
#pragma unroll
                    for (int k = 0; k < K; ++k) lg[k] = fmaf(v, s_w[k][c], lg[k]);
                }
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
extern "C" int wmh_cl_seghead_softmax_accumulate(const void* x, const float* ss, const float* wseg, int K, const float* gauss,
                                                 float* agg, int C, int P0, int P1, int P2, int A0, int A1, int A2,
                                                 const int* origin, int n, const int* flips, float lrelu_slope, void* stream) {
    if (!x || !wseg || !agg || !origin || n < 1 || n > 16) return wmh_fail(WMH_EINVAL, "cl_seghead_softmax_accumulate: bad args");
    if (origin[0] < 0 || origin[1] < 0 || origin[2] < 0 || origin[0] + P0 > A0 || origin[1] + P1 > A1 || origin[2] + P2 > A2)
        return wmh_fail(WMH_EINVAL, "cl_seghead_softmax_accumulate: patch window outside the accumulator");
    if (C != 32 || K != 2) return wmh_fail(WMH_EINVAL, "cl_seghead_softmax_accumulate: implemented for 32 feature maps and 2 classes");
    ClPatchList pl;
    pl.n = n;
    for (int b = 0; b < 16; ++b) pl.flip[b] = (flips && b < n) ? flips[b] : 0;
    const size_t P = (size_t)P0 * P1 * P2;
    hipLaunchKernelGGL((cl_seghead_accum_kernel<2, 32>), dim3(cl_blocks(P, 256, 1u << 20)), dim3(256), 0, (hipStream_t)stream,
                       static_cast<const uint16_t*>(x), ss, wseg, gauss, agg, P0, P1, P2, A0, A1, A2, origin[0], origin[1], origin[2],
                       pl, lrelu_slope);
    return wmh_check_launch("cl_seghead_accum_kernel");
}
