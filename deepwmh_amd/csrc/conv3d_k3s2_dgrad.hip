// conv3d_k3s2_dgrad.hip -- data gradient of the stride-2 3x3x3 convolution (pad 1) in gather form, exact-fp32 MFMA.
//
//   dx[n, ci, i] = sum_{co} sum_{t : (i + 1 - t) even} dy[n, co, (i + 1 - t) / 2] * W[co, ci, t]        per axis
//
// An output position i = 2m + p of parity p sees tap 1 at o = m when p = 0, and taps 0 (o = m + 1) and 2 (o = m) when
// p = 1: each of the 27 taps belongs to exactly ONE of the 8 parity classes of the output.  A wave owns a row of 32
// low-resolution positions m and the 8 class accumulators D[ci 32][m 32] that belong to it (the 2x2x2 block of outputs
// of every m); per pair of dy channels it issues the 27 MFMAs -- the ideal count, 1/8 of running the forward kernel
// over a zero-inserted dy -- from 8 B operands (the 2x2x2 low-resolution neighbourhood, read once from the LDS halo
// tile) and 27 A operands (weights, staged through LDS in fragment order, shared by the 4 waves).
// Double-buffered: the next chunk's dy halo and weights are loaded into registers before the MFMAs of the current one.
// Stands in for the data half of torch.nn.Conv3d(stride=2)'s backward [nnU-Net training, not in tree].
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "wmh_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct DgArgs {
    const float* dy; const float* wp; float* dx;
    int N, Cout, Cin, D, H, W, Do, Ho, Wo;
    int ntx, nty, ntz, G, nchunks;
};

namespace dg {
constexpr int TX = 32, TY = 2, TZ = 2, HX = TX + 1, HY = TY + 1, HZ = TZ + 1, SP = HX * HY * HZ;   // 297 halo voxels
constexpr int CS = 352;                                  // channel stride: = 32 mod 64, so the two k-halves hit disjoint banks
}

template <int CK>
__global__ __launch_bounds__(256, 2) void conv3d_k3s2_dgrad_kernel(const DgArgs a) {
    using namespace dg;
    constexpr int NP = CK / 2, WS = NP * 27 * 64;        // weight floats per chunk
    constexpr int BUF = CK * CS + WS;
    constexpr int NDY = (CK * SP + 255) / 256, NW4 = (WS / 4 + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) float lds[];     // [2][BUF]
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

    // staging geometry of the dy halo (low-resolution window [m0, m0 + T] per axis; positions past the end are zero)
    unsigned yoff[NDY]; unsigned ymask = 0;
#pragma unroll
    for (int k = 0; k < NDY; ++k) {
        const int e = tid + 256 * k;
        const int c = e / SP, s = e - c * SP;
        const int x = s % HX, y = (s / HX) % HY, z = s / (HX * HY);
        const int gx = mx0 + x, gy = my0 + y, gz = mz0 + z;
        const bool ok = e < CK * SP && gx < a.Wo && gy < a.Ho && gz < a.Do;
        yoff[k] = ok ? 4u * (unsigned)((size_t)c * DoHoWo + (size_t)gz * HoWo + gy * a.Wo + gx) : 0u;
        ymask |= ok ? (1u << k) : 0u;
    }
    // native vector type: an array of HIP's float4 (a struct with a union) was left in scratch memory by the compiler, and
    // every scratch reload waits for all loads in flight
    typedef float f32x4n __attribute__((ext_vector_type(4)));
    float yr[NDY]; f32x4n wr[NW4];
    auto issue = [&](int ch) {
        const int cb = ch * CK;
        const char* src = (const char*)(a.dy + ((size_t)n * a.Cout + cb) * DoHoWo);
        // channels past Cout inside the last chunk: their weights are zero in the packed image; keep the address in range
        const bool whole = cb + CK <= a.Cout;
#pragma unroll
        for (int k = 0; k < NDY; ++k) {
            const int c = (tid + 256 * k) / SP;
            const unsigned off = (whole || cb + c < a.Cout) ? yoff[k] : 0u;
            yr[k] = *(const float*)(src + off);
        }
        const f32x4n* wsrc = reinterpret_cast<const f32x4n*>(a.wp + ((size_t)g * a.nchunks + ch) * WS);
#pragma unroll
        for (int k = 0; k < NW4; ++k) {
            const int e = tid + 256 * k;
            wr[k] = wsrc[e < WS / 4 ? e : 0];
        }
    };
    auto write = [&](int buf) {
        float* d = lds + buf * BUF;
#pragma unroll
        for (int k = 0; k < NDY; ++k) {
            const int e = tid + 256 * k;
            const int c = e / SP, s = e - c * SP;
            if (e < CK * SP) d[c * CS + s] = ((ymask >> k) & 1u) ? yr[k] : 0.f;
        }
        f32x4n* dw = reinterpret_cast<f32x4n*>(d + CK * CS);
#pragma unroll
        for (int k = 0; k < NW4; ++k) {
            const int e = tid + 256 * k;
            if (e < WS / 4) dw[e] = wr[k];
        }
    };

    f32x16 acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

    const int yl = wave & 1, zl = wave >> 1;
    const int bbase = kh * CS + (zl * HY + yl) * HX + j;       // + pair*2*CS + (dz*HY + dy)*HX + dx

    issue(0); write(0);
    __syncthreads();
    for (int ch = 0; ch < a.nchunks; ++ch) {
        const bool more = ch + 1 < a.nchunks;
        if (more) issue(ch + 1);
        __builtin_amdgcn_sched_barrier(0x38F);                  // the loads stay above the MFMAs
        const float* cur = lds + (ch & 1) * BUF;
        const float* wl = cur + CK * CS + lane;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            float bv[2][2][2];
#pragma unroll
            for (int dz = 0; dz < 2; ++dz)
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) bv[dz][dy][dx] = cur[bbase + 2 * p * CS + (dz * HY + dy) * HX + dx];
#pragma unroll
            for (int t = 0; t < 27; ++t) {
                const int tz_ = t / 9, ty_ = (t / 3) % 3, tx_ = t % 3;
                const int cls = (tz_ != 1) * 4 + (ty_ != 1) * 2 + (tx_ != 1);
                acc[cls] = __builtin_amdgcn_mfma_f32_32x32x2f32(wl[(p * 27 + t) * 64], bv[tz_ == 0][ty_ == 0][tx_ == 0], acc[cls], 0, 0, 0);
            }
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
                const float v0 = acc[pz * 4 + py * 2][r], v1 = acc[pz * 4 + py * 2 + 1][r];
                if (even_w) *reinterpret_cast<float2*>(o) = make_float2(v0, v1);      // W even: ix + 1 < W and 8-byte aligned
                else { o[0] = v0; if (ix + 1 < a.W) o[1] = v1; }
            }
        }
}

// weight image: [g][chunk][pair][tap][lane] = W[co = chunk*CK + 2*pair + (lane>>5)][ci = 32g + (lane&31)][tap]
__global__ __launch_bounds__(256) void dgrad_pack_kernel(const float* w, float* wp, int Cin, int Cout, int CK, int nchunks, size_t total) {
    const int NP = CK / 2;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        size_t r = idx;
        const int lane = (int)(r % 64); r /= 64;
        const int t = (int)(r % 27); r /= 27;
        const int p = (int)(r % NP); r /= NP;
        const int ch = (int)(r % nchunks);
        const int g = (int)(r / nchunks);
        const int co = ch * CK + 2 * p + (lane >> 5), ci = g * 32 + (lane & 31);
        wp[idx] = (co < Cout && ci < Cin) ? w[((size_t)co * Cin + ci) * 27 + t] : 0.f;
    }
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
constexpr int DG_CK = 8;

extern "C" size_t wmh_conv3d_k3s2_dgrad_packed_floats(int Cin, int Cout) {
    return (size_t)cdiv(Cin, 32) * cdiv(Cout, DG_CK) * (DG_CK / 2) * 27 * 64;
}
extern "C" int wmh_conv3d_k3s2_dgrad_pack_weights_device(const float* w, int Cin, int Cout, float* wp, void* stream) {
    if (!w || !wp || Cin < 1 || Cout < 1) return wmh_fail(WMH_EINVAL, "k3s2_dgrad pack: bad args");
    const size_t total = wmh_conv3d_k3s2_dgrad_packed_floats(Cin, Cout);
    unsigned nb = (unsigned)((total + 255) / 256); if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(dgrad_pack_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, w, wp, Cin, Cout, DG_CK, cdiv(Cout, DG_CK), total);
    return wmh_check_launch("dgrad_pack_kernel");
}
extern "C" int wmh_conv3d_k3s2_dgrad(const float* dy, const float* wp, float* dx, int N, int Cin, int Cout, int D, int H, int W,
                                     void* stream) {
    if (!dy || !wp || !dx || N < 1 || Cin < 1 || Cout < 1 || D < 1 || H < 1 || W < 1) return wmh_fail(WMH_EINVAL, "k3s2_dgrad: bad args");
    DgArgs a;
    a.dy = dy; a.wp = wp; a.dx = dx; a.N = N; a.Cout = Cout; a.Cin = Cin; a.D = D; a.H = H; a.W = W;
    a.Do = (D - 1) / 2 + 1; a.Ho = (H - 1) / 2 + 1; a.Wo = (W - 1) / 2 + 1;
    if ((size_t)DG_CK * a.Do * a.Ho * a.Wo * 4 >= ((size_t)1 << 32)) return wmh_fail(WMH_EINVAL, "k3s2_dgrad: volume too large for 32-bit lane offsets");
    a.ntx = cdiv(a.Wo, dg::TX); a.nty = cdiv(a.Ho, dg::TY); a.ntz = cdiv(a.Do, dg::TZ);
    a.G = cdiv(Cin, 32); a.nchunks = cdiv(Cout, DG_CK);
    const long long nblk = (long long)a.ntx * a.nty * a.ntz * N * a.G;
    if (nblk > 0x7fffffffLL) return wmh_fail(WMH_EINVAL, "k3s2_dgrad: grid too large");
    constexpr int lds = 2 * (DG_CK * dg::CS + (DG_CK / 2) * 27 * 64) * 4;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_k3s2_dgrad_kernel<DG_CK>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((conv3d_k3s2_dgrad_kernel<DG_CK>), dim3((unsigned)nblk), dim3(256), lds, (hipStream_t)stream, a);
    return wmh_check_launch("conv3d_k3s2_dgrad_kernel");
}
