// conv3d_k3s2_dgrad_f16x2.hip -- data gradient of the stride-2 3x3x3 convolution (pad 1) in gather form, in the f16x2 "pair" mode
// (fp32 tensors and accumulation, products from fp16 pairs: conv3d_common.h).
//
//   dx[n, ci, i] = sum_{co} sum_{t : (i + 1 - t) even} dy[n, co, (i + 1 - t) / 2] * W[co, ci, t]        per axis
//
// The mapping of conv3d_k3s2_dgrad.hip -- each of the 27 taps belongs to exactly one of the 8 parity classes of the output; a wave owns a
// row of 32 low-resolution positions m and the 8 class accumulators D[ci 32][m 32] of their 2x2x2 output blocks -- on
// v_mfma_f32_32x32x16_f16: K = 8 dy channels x (hi, lo).  Operands are one dword (hi | lo << 16) per (position | input channel, dy
// channel): a B operand is 4 channels of one halo voxel (one ds_read_b128 from the channels-last halo tile, 8 reads per chunk for the
// 2x2x2 neighbourhood), an A operand 4 channels of one ci row (weights x 2^8, staged through LDS in fragment order, shared by the 4
// waves); A as stored gives hi hi + lo lo, A rotated by 16 bits the cross terms: 54 instructions per 8 channels instead of 108 of
// twice the length.  Double-buffered as the fp32 kernel: the next chunk's dy halo and weights are loaded into registers before the
// MFMAs of the current one.  dy is expected loss-scaled (|dy| < 65520: the trainer's 2^16) -- the split is a mask, not a saturating
// conversion.
// Stands in for the data half of torch.nn.Conv3d(stride=2)'s backward [nnU-Net training, not in tree].
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "wmh_common.h"

typedef float dgp_f32x16 __attribute__((ext_vector_type(16)));
typedef float dgp_f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 dgp_f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 dgp_f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned dgp_u32x4 __attribute__((ext_vector_type(4)));
#define DGP_WSCALE 256.f

struct DgPArgs {
    const float* dy; const unsigned* wp; float* dx;
    int N, Cout, Cin, D, H, W, Do, Ho, Wo;
    int ntx, nty, ntz, G, nchunks;
};

namespace dgp {
constexpr int TX = 32, TY = 2, TZ = 2, HX = TX + 1, HY = TY + 1, HZ = TZ + 1, SP = HX * HY * HZ;   // 297 halo voxels
constexpr int CK = 8;                                    // dy channels per chunk
constexpr int SPP = (SP + 7) / 8 * 8;                    // voxels of the LDS halo image (32 bytes each)
constexpr int WQ = 27 * 64;                              // 16-byte weight pieces per chunk
constexpr int BUF = SPP * 32 + WQ * 16;                  // bytes per buffer
constexpr int NITV = (SP + 255) / 256, NW4 = (WQ + 255) / 256;
}

// v -> (hi | lo << 16): hi = v cut to fp16's 11 significant bits (a mask), lo = the remainder to nearest (conv3d_k3_wgrad_f16x2.hip)
__device__ __forceinline__ unsigned dgp_split(float v) {
    const float hi = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) & 0xFFFFE000u);
    const float lo = v - hi;
    return __builtin_bit_cast(unsigned, __builtin_convertvector(dgp_f32x2{hi, lo}, dgp_f16x2));
}

__global__ __launch_bounds__(256, 2) void conv3d_k3s2_dgrad_f16x2_kernel(const DgPArgs a) {
    using namespace dgp;
    extern __shared__ __attribute__((aligned(16))) char ldsb[];      // [2][BUF]: halo [SPP voxels][2 pieces of 4 channels], then weights [27][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, kh = lane >> 5;
    int b = blockIdx.x;
    const int g = b % a.G; b /= a.G;
    const int tx = b % a.ntx; b /= a.ntx;
    const int ty = b % a.nty; b /= a.nty;
    const int tz = b % a.ntz; b /= a.ntz;
    const int n = b;
    const int mx0 = tx * TX, my0 = ty * TY, mz0 = tz * TZ;
    const int HoWo = a.Ho * a.Wo;
    const size_t DoHoWo = (size_t)a.Do * HoWo;
    const unsigned cstride = 4u * (unsigned)DoHoWo;             // bytes between dy channel planes (host checks the range)

    // staging geometry of the dy halo: slot k of this thread = halo voxel e = tid + 256 k (low-resolution window [m0, m0 + T] per axis;
    // positions past the end are zero); its two 16-byte pieces (channels 0-3 | 4-7) are swapped where (x >> 3) & 1 (bank spread of the
    // B reads: consecutive lanes are 32 bytes apart)
    unsigned yoff[NITV], ykeep[NITV], lpos[NITV];
#pragma unroll
    for (int k = 0; k < NITV; ++k) {
        const int e = tid + 256 * k;
        const int x = e % HX, y = (e / HX) % HY, z = e / (HX * HY);
        const int gx = mx0 + x, gy = my0 + y, gz = mz0 + z;
        const bool ok = (e < SP) & (gx < a.Wo) & (gy < a.Ho) & (gz < a.Do);
        ykeep[k] = ok ? 0xFFFFFFFFu : 0u;
        yoff[k] = (4u * (unsigned)(gz * HoWo + gy * a.Wo + gx)) & ykeep[k];
        const int ew = e < SP ? e : SPP - 1;
        lpos[k] = (unsigned)(ew * 32 + (e < SP ? ((x >> 3) & 1) * 16 : 0));     // byte position of piece 0; piece 1 at ^16
    }
    float yr[NITV][CK]; dgp_u32x4 wr[NW4];
    auto issue = [&](int ch) __attribute__((always_inline)) {
        const int cb = ch * CK;
        const char* src = (const char*)(a.dy + ((size_t)n * a.Cout + cb) * DoHoWo);
#pragma unroll
        for (int k = 0; k < NITV; ++k)
#pragma unroll
            for (int c = 0; c < CK; ++c) {
                // channels past Cout inside the last chunk: their weights are zero in the image; keep the address in range
                const unsigned coff = (unsigned)min(c, a.Cout - 1 - cb) * cstride;
                yr[k][c] = *(const float*)(src + (coff + yoff[k]));
            }
        const dgp_u32x4* wsrc = reinterpret_cast<const dgp_u32x4*>(a.wp) + ((size_t)g * a.nchunks + ch) * WQ;
#pragma unroll
        for (int k = 0; k < NW4; ++k) {
            const int e = tid + 256 * k;
            wr[k] = wsrc[e < WQ ? e : 0];
        }
    };
    auto write = [&](int buf) __attribute__((always_inline)) {
        char* d = ldsb + buf * BUF;
#pragma unroll
        for (int k = 0; k < NITV; ++k) {
            dgp_u32x4 p0, p1;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                p0[c] = dgp_split(yr[k][c]) & ykeep[k];
                p1[c] = dgp_split(yr[k][4 + c]) & ykeep[k];
            }
            *reinterpret_cast<dgp_u32x4*>(d + lpos[k]) = p0;
            *reinterpret_cast<dgp_u32x4*>(d + (lpos[k] ^ 16u)) = p1;
        }
        dgp_u32x4* dw = reinterpret_cast<dgp_u32x4*>(d + SPP * 32);
#pragma unroll
        for (int k = 0; k < NW4; ++k) {
            const int e = tid + 256 * k;
            if (e < WQ) dw[e] = wr[k];
        }
    };

    dgp_f32x16 acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

    const int yl = wave & 1, zl = wave >> 1;
    // byte address of this lane's B operand for neighbour (dz, dy, dx): voxel (zl + dz, yl + dy, j + dx), piece kh (swapped with the key)
    int badr[2][2][2];
#pragma unroll
    for (int dz = 0; dz < 2; ++dz)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int x = j + dx, v = ((zl + dz) * HY + (yl + dy)) * HX + x;
                badr[dz][dy][dx] = v * 32 + 16 * (kh ^ ((x >> 3) & 1));
            }

    issue(0); write(0);
    __syncthreads();
    for (int ch = 0; ch < a.nchunks; ++ch) {
        const bool more = ch + 1 < a.nchunks;
        if (more) issue(ch + 1);
        __builtin_amdgcn_sched_barrier(0x38F);                  // the loads stay above the MFMAs
        const char* cur = ldsb + (ch & 1) * BUF;
        const dgp_u32x4* wl = reinterpret_cast<const dgp_u32x4*>(cur + SPP * 32) + lane;
        dgp_u32x4 bv[2][2][2];
#pragma unroll
        for (int dz = 0; dz < 2; ++dz)
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) bv[dz][dy][dx] = *reinterpret_cast<const dgp_u32x4*>(cur + badr[dz][dy][dx]);
#pragma unroll
        for (int t = 0; t < 27; ++t) {
            const int tz_ = t / 9, ty_ = (t / 3) % 3, tx_ = t % 3;
            const int cls = (tz_ != 1) * 4 + (ty_ != 1) * 2 + (tx_ != 1);
            // A = 4 dy channels x (w_hi, w_lo) of this lane's ci row; rotated by 16 bits: (w_lo, w_hi) -- all four products from two MFMAs
            const dgp_u32x4 av = wl[t * 64];
            dgp_u32x4 ar;
#pragma unroll
            for (int d = 0; d < 4; ++d) ar[d] = __builtin_amdgcn_alignbit(av[d], av[d], 16);
            const dgp_u32x4 bq = bv[tz_ == 0][ty_ == 0][tx_ == 0];
            acc[cls] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(dgp_f16x8, av), __builtin_bit_cast(dgp_f16x8, bq), acc[cls], 0, 0, 0);
            acc[cls] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(dgp_f16x8, ar), __builtin_bit_cast(dgp_f16x8, bq), acc[cls], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0x38F);
        if (more) write((ch + 1) & 1);
        __syncthreads();
    }

    // outputs of this wave: the 2x2x2 block of every m of its row; classes px = 0,1 are neighbours in x -> 8-byte stores
    const int mx = mx0 + j, my = my0 + yl, mz = mz0 + zl;
    const int ix = 2 * mx;
    const size_t HW = (size_t)a.H * a.W;
    const bool even_w = (a.W & 1) == 0;
    constexpr float inv = 1.f / DGP_WSCALE;
#pragma unroll
    for (int pz = 0; pz < 2; ++pz)
#pragma unroll
        for (int py = 0; py < 2; ++py) {
            const int iz = 2 * mz + pz, iy = 2 * my + py;
            if (iz >= a.D || iy >= a.H || ix >= a.W) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = g * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (ci >= a.Cin) continue;
                float* o = a.dx + ((size_t)n * a.Cin + ci) * a.D * HW + (size_t)iz * HW + (size_t)iy * a.W + ix;
                const float v0 = acc[pz * 4 + py * 2][r] * inv, v1 = acc[pz * 4 + py * 2 + 1][r] * inv;
                if (even_w) *reinterpret_cast<float2*>(o) = make_float2(v0, v1);      // W even: ix + 1 < W and 8-byte aligned
                else { o[0] = v0; if (ix + 1 < a.W) o[1] = v1; }
            }
        }
}

// weight image: [g][chunk][tap][lane][4 dwords]: dword e of lane (row, kh) = pair of 2^8 W[co = 8 chunk + 4 kh + e][ci = 32 g + row][tap]
// (clamped to +-65000: a weight of 254 is not a trained network)
__global__ __launch_bounds__(256) void dgrad_pack_f16x2_kernel(const float* w, unsigned* wp, int Cin, int Cout, int nchunks, size_t total) {
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        size_t r = idx;
        const int e = (int)(r % 4); r /= 4;
        const int lane = (int)(r % 64); r /= 64;
        const int t = (int)(r % 27); r /= 27;
        const int ch = (int)(r % nchunks);
        const int g = (int)(r / nchunks);
        const int co = ch * dgp::CK + 4 * (lane >> 5) + e, ci = g * 32 + (lane & 31);
        float v = (co < Cout && ci < Cin) ? DGP_WSCALE * w[((size_t)co * Cin + ci) * 27 + t] : 0.f;
        v = fminf(fmaxf(v, -65000.f), 65000.f);
        wp[idx] = dgp_split(v);
    }
}

static inline int dgp_cdiv(int a, int b) { return (a + b - 1) / b; }

extern "C" size_t wmh_conv3d_k3s2_dgrad_packed_dwords_f16x2(int Cin, int Cout) {
    if (Cin < 1 || Cout < 1) return 0;
    return (size_t)dgp_cdiv(Cin, 32) * dgp_cdiv(Cout, dgp::CK) * 27 * 64 * 4;
}
extern "C" int wmh_conv3d_k3s2_dgrad_pack_weights_f16x2_device(const float* w, int Cin, int Cout, void* wp, void* stream) {
    if (!w || !wp || Cin < 1 || Cout < 1) return wmh_fail(WMH_EINVAL, "k3s2_dgrad pack f16x2: bad args");
    const size_t total = wmh_conv3d_k3s2_dgrad_packed_dwords_f16x2(Cin, Cout);
    unsigned nb = (unsigned)((total + 255) / 256); if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(dgrad_pack_f16x2_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, w, static_cast<unsigned*>(wp), Cin, Cout, dgp_cdiv(Cout, dgp::CK), total);
    return wmh_check_launch("dgrad_pack_f16x2_kernel");
}
extern "C" int wmh_conv3d_k3s2_dgrad_f16x2(const float* dy, const void* wp, float* dx, int N, int Cin, int Cout, int D, int H, int W, void* stream) {
    if (!dy || !wp || !dx || N < 1 || Cin < 1 || Cout < 1 || D < 1 || H < 1 || W < 1) return wmh_fail(WMH_EINVAL, "k3s2_dgrad_f16x2: bad args");
    DgPArgs a;
    a.dy = dy; a.wp = static_cast<const unsigned*>(wp); a.dx = dx; a.N = N; a.Cout = Cout; a.Cin = Cin; a.D = D; a.H = H; a.W = W;
    a.Do = (D - 1) / 2 + 1; a.Ho = (H - 1) / 2 + 1; a.Wo = (W - 1) / 2 + 1;
    if ((size_t)dgp::CK * a.Do * a.Ho * a.Wo * 4 >= ((size_t)1 << 32)) return wmh_fail(WMH_EINVAL, "k3s2_dgrad_f16x2: volume too large for 32-bit lane offsets");
    a.ntx = dgp_cdiv(a.Wo, dgp::TX); a.nty = dgp_cdiv(a.Ho, dgp::TY); a.ntz = dgp_cdiv(a.Do, dgp::TZ);
    a.G = dgp_cdiv(Cin, 32); a.nchunks = dgp_cdiv(Cout, dgp::CK);
    const long long nblk = (long long)a.ntx * a.nty * a.ntz * N * a.G;
    if (nblk > 0x7fffffffLL) return wmh_fail(WMH_EINVAL, "k3s2_dgrad_f16x2: grid too large");
    constexpr int lds = 2 * dgp::BUF;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_k3s2_dgrad_f16x2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(conv3d_k3s2_dgrad_f16x2_kernel, dim3((unsigned)nblk), dim3(256), lds, (hipStream_t)stream, a);
    return wmh_check_launch("conv3d_k3s2_dgrad_f16x2_kernel");
}
