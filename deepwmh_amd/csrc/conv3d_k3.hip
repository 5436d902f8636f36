// conv3d_k3.hip -- 3x3x3 convolution (pad 1, stride 1|2) for gfx950 as an im2col-free implicit
// GEMM on the exact-fp32 matrix instruction v_mfma_f32_32x32x2_f32.
//
//   D[cout 32][voxel 32] += A[cout 32][k 2] * B[k 2][voxel 32],   k = (tap, cin) pairs
//
// * B (activations) comes from an LDS halo tile [CK cin][TN][IZ][IY][IX] staged once per cin
//   chunk with coalesced NCDHW row reads; a lane's 27 taps are immediate offsets from one LDS
//   address, so the inner loop carries no address arithmetic.  The producer layer's
//   InstanceNorm+LeakyReLU is applied while staging (x' = lrelu(x*scale+shift)); halo voxels
//   outside the tensor are written as 0 AFTER that transform (torch pads the normalised tensor).
// * A (weights) is read straight from a fragment-major image in HBM/L2 ([G][chunk][tap][lane][CK/2]),
//   one vector load per tap per lane, shared by all MT voxel tiles of the wave.
// * Two input tensors may be given (concat-free torch.cat((up, skip), 1)).
// * Epilogue: + bias, coalesced 128-B row-segment stores, and per-wave InstanceNorm partials
//   (count, sum, M2 about the wave's own mean) for every output channel.
//
// Stands in for torch.nn.Conv3d inside nnU-Net's ConvDropoutNormNonlin [nnU-Net, not in tree];
// call sites in the reference: deepwmh/main/predict.py:153-156 (nnUNet_predict).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "wmh_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvArgs {
    const float* x0; const float* ss0; int C0;
    const float* x1; const float* ss1; int C1;
    const float* wf; const float* bias; float* y; float* part;
    int N, D, H, W, Do, Ho, Wo, Cout;
    int nchunks, ntx, nty, ntz, npart;
    float slope;
};

template <int S_, int LX_, int LY_, int TZ_, int TY_, int TN_, int MT_, int CK_>
struct ConvCfg {
    static constexpr int S = S_, LX = LX_, LY = LY_, LZ = 32 / (LX_ * LY_);
    static constexpr int TX = LX_, TY = TY_, TZ = TZ_, TN = TN_, MT = MT_, CK = CK_;
    static constexpr int MY = TY / LY, MZ = TZ / LZ, MPS = MY * MZ;  // M-tiles per sample
    static constexpr int NMT = TN * MPS;                             // M-tiles per workgroup
    static constexpr int WM = NMT / MT;                              // waves per workgroup
    static constexpr int THREADS = WM * 64;
    static constexpr int WPS = MPS / MT;                             // waves per sample
    static constexpr int IX = S * (TX - 1) + 3, IY = S * (TY - 1) + 3, IZ = S * (TZ - 1) + 3;
    static constexpr int YS = IX, ZS = IY * YS, NS = IZ * ZS, CS = TN * NS;
    static constexpr int SP = CS;                                    // staged elements per channel
    static constexpr int NIT = (SP + THREADS - 1) / THREADS;
    static constexpr int LDS_FLOATS = CK * CS;
    static_assert(LX * LY * LZ == 32, "lane map must cover 32 voxels");
    static_assert(TY % LY == 0 && TZ % LZ == 0, "tile must be whole M-tiles");
    static_assert(MPS % MT == 0, "a wave's M-tiles must belong to one sample");
    static_assert(NMT % MT == 0 && WM >= 1 && WM <= 4, "1..4 waves");
    static_assert(NIT <= 32, "validity mask is 32 bits");
    static_assert(CK % 2 == 0, "k step is a cin pair");
    static_assert(((2 * ZS + 2 * YS + 2) + (CK - 2) * CS) * 4 < 65536, "ds_read immediate range");
};

template <int V> struct AVec;
template <> struct AVec<1> { typedef float T; };
template <> struct AVec<2> { typedef float2 T; };
template <> struct AVec<4> { typedef float4 T; };
__device__ __forceinline__ float avec_get(float v, int) { return v; }
__device__ __forceinline__ float avec_get(float2 v, int i) { return i == 0 ? v.x : v.y; }
__device__ __forceinline__ float avec_get(float4 v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }

__device__ __forceinline__ float sum32(float v) {   // sum over the 32 lanes of this half-wave
    v += __shfl_xor(v, 16); v += __shfl_xor(v, 8); v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 2);  v += __shfl_xor(v, 1);
    return v;
}

template <class C>
__global__ __launch_bounds__(C::THREADS, 2) void conv3d_k3_kernel(const ConvArgs a) {
    constexpr int S = C::S, MT = C::MT, CK = C::CK, NIT = C::NIT, THREADS = C::THREADS;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, kh = lane >> 5;
    const int dx = j % C::LX, dy = (j / C::LX) % C::LY, dz = j / (C::LX * C::LY);

    int b = blockIdx.x;
    const int tx = b % a.ntx; b /= a.ntx;
    const int ty = b % a.nty; b /= a.nty;
    const int tz = b % a.ntz; b /= a.ntz;
    const int n0 = b * C::TN;
    const int g = blockIdx.y;
    const int ox0 = tx * C::TX, oy0 = ty * C::TY, oz0 = tz * C::TZ;
    const int ix0 = ox0 * S - 1, iy0 = oy0 * S - 1, iz0 = oz0 * S - 1;
    const int HW = a.H * a.W;
    const size_t DHW = (size_t)a.D * HW;

    // ---- per-thread staging map: element e of the [TN][IZ][IY][IX] tile <-> global voxel ------
    int sp_off[NIT];          // offset inside one (n,c) plane volume, valid when bit set in okmask
    unsigned okmask = 0;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int e = tid + it * THREADS;
        const int x = e % C::IX, y = (e / C::IX) % C::IY, z = (e / (C::IX * C::IY)) % C::IZ;
        const int ni = e / C::NS;
        const int gx = ix0 + x, gy = iy0 + y, gz = iz0 + z;
        const bool ok = (e < C::SP) && (n0 + ni < a.N) && gx >= 0 && gx < a.W && gy >= 0 && gy < a.H &&
                        gz >= 0 && gz < a.D;
        sp_off[it] = ok ? gz * HW + gy * a.W + gx : 0;   // always a safe address: loads are unconditional
        okmask |= ok ? (1u << it) : 0u;
    }

    // ---- per-wave M-tile bases into the LDS tile -----------------------------------------------
    const int tn_w = (wave * MT) / C::MPS;          // sample (within the workgroup) of this wave
    const int ts0 = (wave * MT) % C::MPS;           // first M-tile (within the sample)
    int baddr[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int ts = ts0 + m;
        const int tyi = ts % C::MY, tzi = ts / C::MY;
        baddr[m] = kh * C::CS + tn_w * C::NS + S * ((tzi * C::LZ + dz) * C::ZS + (tyi * C::LY + dy) * C::YS + dx);
    }

    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

    typedef typename AVec<CK / 2>::T AV;
    const AV* wbase = reinterpret_cast<const AV*>(a.wf) + (size_t)g * a.nchunks * 27 * 64 + lane;

    float stg[CK][NIT];

    auto src_of = [&](int ch, const float*& xs, const float*& ss, int& Cs, int& c0) {
        const int cb = ch * CK;
        if (cb < a.C0) { xs = a.x0; ss = a.ss0; Cs = a.C0; c0 = cb; }
        else           { xs = a.x1; ss = a.ss1; Cs = a.C1; c0 = cb - a.C0; }
    };
    auto load_chunk = [&](int ch) {
        const float* xs; const float* ss; int Cs, c0;
        src_of(ch, xs, ss, Cs, c0);
#pragma unroll
        for (int cc = 0; cc < CK; ++cc) {
            const int c = min(c0 + cc, Cs - 1);          // clamped: the value is discarded when out of range
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int ni = (C::TN == 1) ? 0 : min((tid + it * THREADS) / C::NS, a.N - 1 - n0);
                stg[cc][it] = xs[((size_t)(n0 + ni) * Cs + c) * DHW + sp_off[it]];
            }
        }
    };
    auto write_chunk = [&](int ch) {
        const float* xs; const float* ss; int Cs, c0;
        src_of(ch, xs, ss, Cs, c0);
#pragma unroll
        for (int cc = 0; cc < CK; ++cc) {
            const bool cok = (c0 + cc) < Cs;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int e = tid + it * THREADS;
                const int ni = (C::TN == 1) ? 0 : min(e / C::NS, a.N - 1 - n0);
                const bool ok = cok && ((okmask >> it) & 1u);
                float v = stg[cc][it];
                if (ss != nullptr) {
                    const float* q = ss + ((size_t)(n0 + ni) * Cs + min(c0 + cc, Cs - 1)) * 2;
                    v = fmaf(v, q[0], q[1]);
                    v = v > 0.f ? v : v * a.slope;
                }
                v = ok ? v : 0.f;
                if (NIT * THREADS == C::SP || e < C::SP) lds[cc * C::CS + e] = v;
            }
        }
    };

    // ---- main loop over cin chunks -------------------------------------------------------------
    load_chunk(0);
    write_chunk(0);
    __syncthreads();
    for (int ch = 0; ch < a.nchunks; ++ch) {
        const bool more = (ch + 1) < a.nchunks;
        if (more) load_chunk(ch + 1);           // global loads in flight under the MFMAs below

        const AV* wp = wbase + (size_t)ch * 27 * 64;
        AV a_cur = wp[0];
#pragma unroll
        for (int tap = 0; tap < 27; ++tap) {
            const int toff = (tap / 9) * C::ZS + ((tap / 3) % 3) * C::YS + (tap % 3);
            AV a_nxt = a_cur;
            if (tap + 1 < 27) a_nxt = wp[(tap + 1) * 64];
#pragma unroll
            for (int p = 0; p < CK / 2; ++p) {
                const float av = avec_get(a_cur, p);
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const float bv = lds[baddr[m] + toff + 2 * p * C::CS];
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[m], 0, 0, 0);
                }
            }
            a_cur = a_nxt;
        }
        __syncthreads();                        // every wave is done reading this chunk's tile
        if (more) {
            write_chunk(ch + 1);
            __syncthreads();
        }
    }

    // ---- epilogue: bias, store, InstanceNorm partials -------------------------------------------
    const int HoWo = a.Ho * a.Wo;
    const size_t DoHoWo = (size_t)a.Do * HoWo;
    const int n = n0 + tn_w;
    float bias_r[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = g * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        bias_r[r] = (a.bias != nullptr && co < a.Cout) ? a.bias[co] : 0.f;
    }
    float s[16], q[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; q[r] = 0.f; }
    int cnt = 0;
    unsigned vmask = 0;
    const bool full_rows = (g * 32 + 32) <= a.Cout;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int ts = ts0 + m;
        const int oz = oz0 + (ts / C::MY) * C::LZ + dz, oy = oy0 + (ts % C::MY) * C::LY + dy, ox = ox0 + dx;
        const bool valid = (n < a.N) && oz < a.Do && oy < a.Ho && ox < a.Wo;
        vmask |= valid ? (1u << m) : 0u;
        cnt += __popcll(__ballot(valid) & 0xffffffffull);
        float* yp = a.y + ((size_t)n * a.Cout + g * 32 + 4 * kh) * DoHoWo + (size_t)oz * HoWo + oy * a.Wo + ox;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = acc[m][r] + bias_r[r];
            acc[m][r] = v;
            s[r] += valid ? v : 0.f;
        }
        if (valid) {
            if (full_rows) {                    // wave-uniform: all 32 output channels of this group exist
#pragma unroll
                for (int r = 0; r < 16; ++r) yp[(size_t)((r & 3) + 8 * (r >> 2)) * DoHoWo] = acc[m][r];
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2);
                    if ((g * 32 + 4 * kh + row) < a.Cout) yp[(size_t)row * DoHoWo] = acc[m][r];
                }
            }
        }
    }
    if (a.part != nullptr) {
        const float fc = (float)cnt;
        const float inv = cnt > 0 ? 1.f / fc : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = sum32(s[r]);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const bool valid = (vmask >> m) & 1u;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float d = acc[m][r] - s[r] * inv;
                q[r] += valid ? d * d : 0.f;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) q[r] = sum32(q[r]);
        if (j == 0 && n < a.N) {
            const int tile_sp = (tz * a.nty + ty) * a.ntx + tx;
            const int pidx = tile_sp * C::WPS + (wave % C::WPS);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = g * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (co < a.Cout) {
                    float4* pp = reinterpret_cast<float4*>(a.part) + ((size_t)n * a.Cout + co) * a.npart + pidx;
                    *pp = make_float4(fc, s[r], q[r], 0.f);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// configuration table
// ------------------------------------------------------------------------------------------------
//                      S  LX LY TZ TY TN MT CK
typedef ConvCfg<1, 32, 1, 4, 8, 1, 8, 4> Cfg0;   // stride 1, rows of >=32 voxels (128^3, 64^3, 32^3)
typedef ConvCfg<2, 32, 1, 2, 8, 1, 4, 2> Cfg1;   // stride 2, output rows of >=32
typedef ConvCfg<1, 16, 2, 2, 16, 1, 4, 4> Cfg2;  // stride 1, 16-wide
typedef ConvCfg<2, 16, 2, 2, 16, 1, 4, 2> Cfg3;  // stride 2, 16-wide output
typedef ConvCfg<1, 8, 4, 4, 8, 1, 2, 8> Cfg4;    // stride 1, 8-wide
typedef ConvCfg<2, 8, 4, 4, 8, 1, 2, 4> Cfg5;    // stride 2, 8-wide output
typedef ConvCfg<1, 4, 4, 4, 4, 4, 2, 8> Cfg6;    // stride 1, 4-wide, 4 samples per workgroup
typedef ConvCfg<2, 4, 4, 4, 4, 4, 2, 4> Cfg7;    // stride 2, 4-wide output
#define WMH_NUM_CONV_CFG 8

struct CfgInfo { int S, TX, TY, TZ, TN, CK, WPS, threads, lds_bytes; };
template <class C> static CfgInfo info_of() {
    return CfgInfo{C::S, C::TX, C::TY, C::TZ, C::TN, C::CK, C::WPS, C::THREADS, C::LDS_FLOATS * 4};
}
static const CfgInfo& cfg_info(int cfg) {
    static const CfgInfo t[WMH_NUM_CONV_CFG] = {info_of<Cfg0>(), info_of<Cfg1>(), info_of<Cfg2>(), info_of<Cfg3>(),
                                                info_of<Cfg4>(), info_of<Cfg5>(), info_of<Cfg6>(), info_of<Cfg7>()};
    return t[cfg];
}

template <class C>
static int launch_conv(const ConvArgs& a, dim3 grid, hipStream_t st) {
    static bool attr_set = false;
    const int lds = C::LDS_FLOATS * 4;
    if (!attr_set) {
        if (lds > 48 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_k3_kernel<C>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((conv3d_k3_kernel<C>), grid, dim3(C::THREADS), lds, st, a);
    return wmh_check_launch("conv3d_k3_kernel");
}

extern "C" int wmh_conv3d_k3_config(int Cin, int Cout, int D, int H, int W, int stride) {
    if (Cin < 1 || Cout < 1 || D < 1 || H < 1 || W < 1 || (stride != 1 && stride != 2))
        return wmh_fail(WMH_EINVAL, "wmh_conv3d_k3_config: bad shape");
    const int Wo = (W - 1) / stride + 1;
    int base = Wo >= 24 ? 0 : Wo >= 12 ? 2 : Wo >= 6 ? 4 : 6;
    return base + (stride == 2 ? 1 : 0);
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

extern "C" size_t wmh_conv3d_k3_packed_floats(int cfg, int Cin, int Cout) {
    if (cfg < 0 || cfg >= WMH_NUM_CONV_CFG) return 0;
    const int CK = cfg_info(cfg).CK;
    return (size_t)cdiv(Cout, 32) * cdiv(Cin, CK) * 27 * 64 * (CK / 2);
}

extern "C" int wmh_conv3d_k3_pack_weights(int cfg, const float* w, int Cin, int Cout, float* wf) {
    if (cfg < 0 || cfg >= WMH_NUM_CONV_CFG || !w || !wf) return wmh_fail(WMH_EINVAL, "pack_weights: bad args");
    const int CK = cfg_info(cfg).CK, G = cdiv(Cout, 32), NCH = cdiv(Cin, CK), V = CK / 2;
    for (int g = 0; g < G; ++g)
        for (int ch = 0; ch < NCH; ++ch)
            for (int tap = 0; tap < 27; ++tap)
                for (int lane = 0; lane < 64; ++lane)
                    for (int p = 0; p < V; ++p) {
                        const int co = g * 32 + (lane & 31), ci = ch * CK + 2 * p + (lane >> 5);
                        float v = 0.f;
                        if (co < Cout && ci < Cin) v = w[((size_t)co * Cin + ci) * 27 + tap];
                        wf[((((size_t)g * NCH + ch) * 27 + tap) * 64 + lane) * V + p] = v;
                    }
    return WMH_OK;
}

extern "C" int wmh_conv3d_k3_num_partials(int cfg, int N, int Do, int Ho, int Wo) {
    if (cfg < 0 || cfg >= WMH_NUM_CONV_CFG) return wmh_fail(WMH_EINVAL, "num_partials: bad cfg");
    const CfgInfo& c = cfg_info(cfg);
    (void)N;
    return cdiv(Wo, c.TX) * cdiv(Ho, c.TY) * cdiv(Do, c.TZ) * c.WPS;
}

extern "C" int wmh_conv3d_k3_fwd(int cfg, const float* x0, const float* ss0, int C0, const float* x1,
                                 const float* ss1, int C1, const float* wf, const float* bias, float* y,
                                 float* partials, int N, int D, int H, int W, int Cout, int stride,
                                 float lrelu_slope, void* stream) {
    if (cfg < 0 || cfg >= WMH_NUM_CONV_CFG) return wmh_fail(WMH_EINVAL, "conv3d_k3_fwd: bad cfg");
    const CfgInfo& c = cfg_info(cfg);
    if (!x0 || !wf || !y || C0 < 1 || N < 1 || Cout < 1) return wmh_fail(WMH_EINVAL, "conv3d_k3_fwd: null/empty argument");
    if (c.S != stride) return wmh_fail(WMH_EINVAL, "conv3d_k3_fwd: cfg/stride mismatch");
    if (x1 == nullptr) C1 = 0;
    if (C1 > 0 && (C0 % c.CK) != 0) return wmh_fail(WMH_EINVAL, "conv3d_k3_fwd: C0 must be a multiple of the cin chunk when two sources are given");
    ConvArgs a;
    a.x0 = x0; a.ss0 = ss0; a.C0 = C0; a.x1 = x1; a.ss1 = ss1; a.C1 = C1;
    a.wf = wf; a.bias = bias; a.y = y; a.part = partials;
    a.N = N; a.D = D; a.H = H; a.W = W;
    a.Do = (D - 1) / stride + 1; a.Ho = (H - 1) / stride + 1; a.Wo = (W - 1) / stride + 1;
    a.Cout = Cout;
    a.nchunks = (C1 > 0) ? (C0 / c.CK + cdiv(C1, c.CK)) : cdiv(C0, c.CK);
    a.ntx = cdiv(a.Wo, c.TX); a.nty = cdiv(a.Ho, c.TY); a.ntz = cdiv(a.Do, c.TZ);
    a.npart = a.ntx * a.nty * a.ntz * c.WPS;
    a.slope = lrelu_slope;
    const long long nblk = (long long)a.ntx * a.nty * a.ntz * cdiv(N, c.TN);
    if (nblk > 0x7fffffffLL) return wmh_fail(WMH_EINVAL, "conv3d_k3_fwd: grid too large");
    dim3 grid((unsigned)nblk, (unsigned)cdiv(Cout, 32));
    hipStream_t st = (hipStream_t)stream;
    switch (cfg) {
        case 0: return launch_conv<Cfg0>(a, grid, st);
        case 1: return launch_conv<Cfg1>(a, grid, st);
        case 2: return launch_conv<Cfg2>(a, grid, st);
        case 3: return launch_conv<Cfg3>(a, grid, st);
        case 4: return launch_conv<Cfg4>(a, grid, st);
        case 5: return launch_conv<Cfg5>(a, grid, st);
        case 6: return launch_conv<Cfg6>(a, grid, st);
        case 7: return launch_conv<Cfg7>(a, grid, st);
    }
    return wmh_fail(WMH_EINVAL, "conv3d_k3_fwd: bad cfg");
}

// ---- MFMA layout self-test -----------------------------------------------------------------------
__global__ void selftest_mfma_kernel(const float* a, const float* b, float* d) {
    const int lane = threadIdx.x & 63;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float av = a[(lane & 31) * 2 + (lane >> 5)];     // A[i = lane&31][k = lane>>5]
    const float bv = b[(lane >> 5) * 32 + (lane & 31)];    // B[k = lane>>5][j = lane&31]
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        d[i * 32 + (lane & 31)] = acc[r];
    }
}
extern "C" int wmh_selftest_mfma_32x32x2(const float* a, const float* b, float* d, void* stream) {
    hipLaunchKernelGGL(selftest_mfma_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, b, d);
    return wmh_check_launch("selftest_mfma_kernel");
}
