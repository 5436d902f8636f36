// conv3d_k3_ws.hip -- wave-specialised, persistent form of the fp32 3x3x3 convolution for the large layers
// (stride 1, rows of >= 32 voxels: the layers that hold ~85 % of the network's FLOPs).
//
// Timing-only ablations of the one-role kernel (conv3d_k3.hip) showed that a wave issuing nothing but LDS reads
// and v_mfma_f32_32x32x2_f32 runs at 99.6 % of the 64-cycle issue rate, and that the staging instructions (global
// loads, transform, LDS writes, weight loads) interleaved into the same wave cost ~20 % of it.  So the roles are
// split inside one 512-thread workgroup per CU:
//   waves 0-3  consumers: B operand by ds_read with immediate offsets, A operand by ds_read_b64, MFMA, epilogue;
//   waves 4-7  loaders:   coalesced global loads of the next cin chunk (activation halo tile AND its weight
//              fragment image), InstanceNorm+LeakyReLU on the way in, LDS writes.
// A workgroup's waves are dealt to the 4 SIMDs in turn, so every SIMD hosts one consumer and one loader.
// LDS: activations [2][CK][halo tile] + weights [2][27 taps][64 lanes][CK/2], one barrier per chunk.
// The workgroup is persistent (grid = number of CUs): the loaders run one chunk ahead ACROSS tile boundaries, so
// a tile has no prologue, and they prefetch the next tile's first chunk while the consumers store the current one.
// Same arithmetic, same k order, same epilogue as the one-role kernel: results are bit-identical to it.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "conv3d_common.h"

__device__ __forceinline__ bool getenv_dbg_perwave(int dbg) { return (dbg & 8) != 0; }      // diagnostic build: per-wave stamps
// Uniform read of a value no kernel in flight writes (InstanceNorm scale/shift come from the previous launch): through the
// constant address space = the scalar cache, on its own counter.  As (conditional) vector loads these reads were several
// serialised memory round trips per chunk: the loader needed a whole consumer period for ~300 instructions.
__device__ __forceinline__ float ws_ld_uniform(const float* p) {
    return *reinterpret_cast<const __attribute__((address_space(4))) float*>(reinterpret_cast<unsigned long long>(p));
}
static __device__ const float ws_ident2[2] = {1.f, 0.f};

// IN16: the sources are bf16 tensors (a quad is one 8-byte load per channel, widened by a shift in the loader)
template <class C, bool IN16 = false>
__global__ __launch_bounds__(512, 2) void conv3d_k3_ws_kernel(const ConvArgs a, const int T /*tiles*/, const int dbg /*timing experiments of the WMH_DIAG_ABLATIONS build; unused otherwise*/) {
    // Stamps and ablation switches exist in the diagnostic build only (-DWMH_DIAG_ABLATIONS): in the library they are
    // compile-time constants and every branch on them folds away.
#ifdef WMH_DIAG_ABLATIONS
    const int dbg_ = dbg;
    unsigned long long* const stamps_ = WMH_STAMPS(a);
#else
    constexpr int dbg_ = 0;
    constexpr unsigned long long* stamps_ = nullptr;
    (void)dbg;
#endif
    constexpr int S = C::S, MT = C::MT, NT = C::NT, CK = C::CK, V = CK / 2;
    constexpr int LTH = 256;                                   // loader threads
    // LDS halo rows are the 16-byte-aligned span [ox0-4, ox0+TX+4) of the row (QPR quads; the halo row [ox0-1, ox0+TX] sits
    // at positions 3..TX+4): a loader moves whole quads -- one 16-byte global load and one ds_write_b128 per quad and channel
    // instead of four dword loads and stores -- because every loader instruction has to find an issue slot on a SIMD whose
    // consumer wave always has an MFMA ready (measured: 7.7 k cycles per chunk alone, 28.6 k next to a busy consumer).
    constexpr int ROWS = C::IY * C::IZ, QPR = (C::IX + 3 + 3) / 4, RSX = 4 * QPR, NQ = ROWS * QPR;   // span = 3 + IX, rounded up to quads
    constexpr int NSL = (NQ + LTH - 1) / LTH;                  // quad slots per loader thread
    constexpr int CS = ROWS * RSX, TILE = CK * CS;             // channel stride, floats per buffer
    constexpr int YSL = RSX, ZSL = C::IY * RSX;                // row / plane strides of the LDS tile
    constexpr int WT = 27 * NT * 64 * V;                       // weight floats per chunk
    constexpr int WIT = (WT / 2 + LTH - 1) / LTH;              // float2 pieces per loader thread (the image is copied as is)
    typedef typename AVec<V>::T AV;                            // a lane's weights of one (tap, cout tile): V cin pairs
    static_assert(C::TN == 1 && C::WM == 4, "four consumer waves, one sample per workgroup");
    static_assert(C::SZ == S && C::KZ == 3, "isotropic 3x3x3, stride 1 or 2");
    static_assert(((2 * ZSL + 2 * YSL + 2) + (CK - 2) * CS + TILE) * 4 < 65536, "ds_read immediate range");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* X = lds;                       // [2][TILE]
    float* Wl = lds + 2 * TILE;           // [2][WT]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave >= 4;
    const int HW = a.H * a.W;
    const size_t DHW = (size_t)a.D * HW;
    const int nch = a.nchunks;

    // virtual block v -> tile, with the XCD-aware order of the one-role kernel (contiguous run per XCD)
    auto tile_of = [&](int v, int& g, int& tx, int& ty, int& tz, int& n0) {
        const int q = T >> 3, r = T & 7, xcd = v & 7;
        int t = a.xcd_swizzle ? (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3) : v;
        wmh_tile_coords(a, t, g, tx, ty, tz, n0);
    };
    int ntiles_mine = 0;
    for (int v = blockIdx.x; v < T; v += gridDim.x) ++ntiles_mine;
    const int total = ntiles_mine * nch;                    // chunk iterations of this workgroup

    if (loader) {
        // ================================ loader waves =====================================================
        // A loader shares its SIMD with a consumer that always has an MFMA ready; left at equal priority it got to issue about
        // one instruction per MFMA slot and needed a whole consumer period (29 k cycles) for ~300 instructions.
        if (!(dbg_ & 32)) __builtin_amdgcn_s_setprio(3);
        const int lt = tid - 256;
        typedef float f32x4_t __attribute__((ext_vector_type(4)));
        unsigned q_off[NSL]; unsigned q_ok = 0;               // byte offsets of this thread's quads in a channel plane
        int g = 0, n0 = 0;
        f32x4_t stg[IN16 ? 1 : NSL][CK];
        uint2 stg16[IN16 ? NSL : 1][CK];
        float2 wst[WIT];
        float sc[CK], sh[CK], slope_eff = 1.f;
        auto setup = [&](int v) __attribute__((always_inline)) {
            int tx, ty, tz;
            tile_of(v, g, tx, ty, tz, n0);
            g = __builtin_amdgcn_readfirstlane(g); n0 = __builtin_amdgcn_readfirstlane(n0);       // out of VALU divisions: scalar again
            tx = __builtin_amdgcn_readfirstlane(tx); ty = __builtin_amdgcn_readfirstlane(ty); tz = __builtin_amdgcn_readfirstlane(tz);
            const int ix0 = tx * C::TX * S - 4, iy0 = ty * C::TY * S - 1, iz0 = tz * C::TZ * S - 1;
            q_ok = 0;
#pragma unroll
            for (int i = 0; i < NSL; ++i) {
                const int sidx = lt + i * LTH;
                const int row = sidx / QPR, q = sidx - row * QPR;
                const int gx = ix0 + 4 * q, gy = iy0 + row % C::IY, gz = iz0 + row / C::IY;
                const bool ok = sidx < NQ && gx >= 0 && gx < a.W && gy >= 0 && gy < a.H && gz >= 0 && gz < a.D;
                q_off[i] = ok ? 4u * (unsigned)(gz * HW + gy * a.W + gx) : 0u;
                q_ok |= ok ? (1u << i) : 0u;
            }
        };
        const unsigned pstride = 4u * (unsigned)DHW;           // bytes between channel planes (host checks the 32-bit range)
        auto load_chunk = [&](int ch) __attribute__((always_inline)) {
            ch = __builtin_amdgcn_readfirstlane(ch);
            const int cb = ch * CK;
            const float* xs; const float* ss; int Cs, c0;
            if (cb < a.C0) { xs = a.x0; ss = a.ss0; Cs = a.C0; c0 = cb; }
            else           { xs = a.x1; ss = a.ss1; Cs = a.C1; c0 = cb - a.C0; }
            const int cmax = Cs - 1 - c0;                     // channels past the source's end re-read its last one (scale 0)
            if constexpr (IN16) {
                const char* base = reinterpret_cast<const char*>(xs) + ((size_t)n0 * Cs + c0) * DHW * 2;
#pragma unroll
                for (int i = 0; i < NSL; ++i)
#pragma unroll
                    for (int cc = 0; cc < CK; ++cc)
                        stg16[i][cc] = *reinterpret_cast<const uint2*>(base + ((q_off[i] >> 1) + (unsigned)min(cc, cmax) * (pstride >> 1)));
            } else {
                const char* base = reinterpret_cast<const char*>(xs + ((size_t)n0 * Cs + c0) * DHW);
#pragma unroll
                for (int i = 0; i < NSL; ++i)
#pragma unroll
                    for (int cc = 0; cc < CK; ++cc)
                        stg[i][cc] = *reinterpret_cast<const f32x4_t*>(base + (q_off[i] + (unsigned)min(cc, cmax) * pstride));
            }
            const bool has_ss = ss != nullptr;
            slope_eff = has_ss ? a.slope : 1.f;
            const float* tab = has_ss ? ss + ((size_t)n0 * Cs + c0) * 2 : ws_ident2;      // without a pending normalisation: {1, 0}
            const int mul = has_ss ? 2 : 0;
#pragma unroll
            for (int cc = 0; cc < CK; ++cc) {
                const int idx = mul * min(cc, cmax);
                const float s0 = ws_ld_uniform(tab + idx), s1 = ws_ld_uniform(tab + idx + 1);
                sc[cc] = cc <= cmax ? s0 : 0.f;                 // (x * 0 + 0 -> LeakyReLU(0) = 0: channels past the end stage zeros)
                sh[cc] = cc <= cmax ? s1 : 0.f;
            }
            const float2* wsrc = reinterpret_cast<const float2*>(a.wf) + ((size_t)g * nch + ch) * (WT / 2);
#pragma unroll
            for (int i = 0; i < WIT; ++i) {
                const int k = min(lt + i * LTH, WT / 2 - 1);
                wst[i] = wsrc[k];
            }
        };
        auto write_chunk = [&](int buf) __attribute__((always_inline)) {
            float* xb = X + buf * TILE;
#pragma unroll
            for (int i = 0; i < NSL; ++i) {
                const int sidx = lt + i * LTH;
                // quads outside the volume: scale = shift = 0 -> zeros (the clamped address reads finite activations of this plane)
                const bool in = (q_ok >> i) & 1u;
#pragma unroll
                for (int cc = 0; cc < CK; ++cc) {
                    const float scl = in ? sc[cc] : 0.f, shl = in ? sh[cc] : 0.f;
                    // packed fp32 math (v_pk_fma_f32 / v_pk_mul_f32: two voxels per instruction, same rounding as fmaf / *).  Beside the fp32
                    // matrix instruction the packed form is the faster one (0.887 of peak against 0.866 with unpacked instructions, round 3);
                    // beside the 16-bit matrix instructions it is the other way round (conv3d_common.h::wmh_norm_lrelu)
                    typedef float f32x2_t __attribute__((ext_vector_type(2)));
                    const f32x2_t sc2 = {scl, scl}, sh2 = {shl, shl};
                    f32x2_t xlo, xhi;
                    if constexpr (IN16) {
                        const unsigned u0 = stg16[i][cc].x, u1 = stg16[i][cc].y;
                        xlo = f32x2_t{__builtin_bit_cast(float, u0 << 16), __builtin_bit_cast(float, u0 & 0xFFFF0000u)};
                        xhi = f32x2_t{__builtin_bit_cast(float, u1 << 16), __builtin_bit_cast(float, u1 & 0xFFFF0000u)};
                    } else {
                        xlo = f32x2_t{stg[i][cc][0], stg[i][cc][1]};
                        xhi = f32x2_t{stg[i][cc][2], stg[i][cc][3]};
                    }
                    const f32x2_t lo = __builtin_elementwise_fma(xlo, sc2, sh2);
                    const f32x2_t hi = __builtin_elementwise_fma(xhi, sc2, sh2);
                    const f32x2_t lo_s = lo * slope_eff, hi_s = hi * slope_eff;
                    // LeakyReLU as max(t, slope t) for 0 <= slope <= 1
                    const f32x4_t v = {fmaxf(lo[0], lo_s[0]), fmaxf(lo[1], lo_s[1]), fmaxf(hi[0], hi_s[0]), fmaxf(hi[1], hi_s[1])};
                    if (NSL * LTH == NQ || sidx < NQ) *reinterpret_cast<f32x4_t*>(xb + cc * CS + 4 * sidx) = v;   // row * RSX + 4 q = 4 sidx
                }
            }
            float2* wb = reinterpret_cast<float2*>(Wl + buf * WT);
#pragma unroll
            for (int i = 0; i < WIT; ++i) {
                const int k = lt + i * LTH;
                if (WIT * LTH == WT / 2 || k < WT / 2) wb[k] = wst[i];
            }
        };
        // iteration it handles (tile k = it / nch, chunk it % nch); the loaders stay one iteration ahead
        int vnext = blockIdx.x;
        if (total > 0) { setup(vnext); load_chunk(0); write_chunk(0); }
        __syncthreads();
        unsigned long long lw = 0, lb = 0, l0 = 0;                 // diagnostics: loader work / barrier wait
        for (int it = 0; it < total; ++it) {
            const int nx = it + 1;
            if (stamps_) l0 = __builtin_amdgcn_s_memtime();
            if (nx < total && !(dbg_ & 1)) {
                const int ch = nx % nch;
                if (ch == 0) { vnext += gridDim.x; setup(vnext); }
                load_chunk(ch);
                write_chunk(nx & 1);
            }
            if (stamps_) { const unsigned long long t = __builtin_amdgcn_s_memtime(); lw += t - l0; l0 = t; }
            __syncthreads();
            if (stamps_) lb += __builtin_amdgcn_s_memtime() - l0;
        }
        if (stamps_ && (dbg_ & 16) && tid == 256) {
            unsigned long long* o = stamps_ + (size_t)blockIdx.x * 4;
            o[0] = lw; o[1] = lb; o[2] = 0; o[3] = (unsigned long long)total;
        }
    } else {
        // ================================ consumer waves ===================================================
        const int j = lane & 31, kh = lane >> 5;
        const int dx = j % C::LX, dy = (j / C::LX) % C::LY, dz = j / (C::LX * C::LY);
        const int ts0 = (wave * MT) % C::MPS;
        int baddr[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int ts = ts0 + m;
            const int tyi = ts % C::MY, tzi = ts / C::MY;
            baddr[m] = kh * CS + S * ((tzi * C::LZ + dz) * ZSL + (tyi * C::LY + dy) * YSL + dx) + 3;    // halo x = 0 sits at span position 3
        }
        __syncthreads();
        int it = 0;
        unsigned long long tm = 0, tb = 0, te = 0, t0 = 0;         // diagnostics (stamps_ != NULL)
        for (int v = blockIdx.x; v < T; v += gridDim.x) {
            int g, tx, ty, tz, n0;
            tile_of(v, g, tx, ty, tz, n0);
            f32x16 acc[NT][MT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = (g * NT + nt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    const float bv = (a.bias != nullptr && co < a.Cout) ? a.bias[co] : 0.f;
#pragma unroll
                    for (int m = 0; m < MT; ++m) acc[nt][m][r] = bv;
                }
            for (int ch = 0; ch < nch; ++ch, ++it) {
                const float* cur = X + (it & 1) * TILE;
                const AV* wcur = reinterpret_cast<const AV*>(Wl + (it & 1) * WT) + lane;
                if (stamps_) t0 = __builtin_amdgcn_s_memtime();
                if (!(dbg_ & 64))
#pragma unroll
                for (int tap = 0; tap < 27; ++tap) {
                    const int toff = (tap / 9) * ZSL + ((tap / 3) % 3) * YSL + (tap % 3);
                    AV av[NT];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) av[nt] = wcur[(tap * NT + nt) * 64];
#pragma unroll
                    for (int p = 0; p < V; ++p) {
#pragma unroll
                        for (int m = 0; m < MT; ++m) {
                            const float bv = cur[baddr[m] + toff + 2 * p * CS];
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                acc[nt][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(avec_get(av[nt], p), bv, acc[nt][m], 0, 0, 0);
                        }
                    }
                }
                if (stamps_) { const unsigned long long t = __builtin_amdgcn_s_memtime(); tm += t - t0; t0 = t; }
                __syncthreads();
                if (stamps_) tb += __builtin_amdgcn_s_memtime() - t0;
            }
            if (stamps_) t0 = __builtin_amdgcn_s_memtime();
            if (!(dbg_ & 2))
                conv_epilogue<C>(a, acc, g, 0, n0, wave, j, kh, dx, dy, dz, ts0, tz * C::TZ, ty * C::TY, tx * C::TX,
                                 (tz * a.nty + ty) * a.ntx + tx);
            else if (acc[0][0][0] == 12345.678f) a.y[0] = acc[NT - 1][1][3];     // keep the accumulators live
            if (stamps_) te += __builtin_amdgcn_s_memtime() - t0;
        }
        if (stamps_ && getenv_dbg_perwave(dbg_)) {                  // per consumer wave: its own MFMA-phase total
            if (lane == 0) stamps_[(size_t)blockIdx.x * 4 + wave] = wave == 3 ? (unsigned long long)total : tm;
        } else if (stamps_ && tid == 0 && !(dbg_ & 16)) {            // barrier wait, MFMA phases, epilogues, chunk iterations
            unsigned long long* o = stamps_ + (size_t)blockIdx.x * 4;
            o[0] = tb; o[1] = tm; o[2] = te; o[3] = (unsigned long long)total;
        }
    }
}

typedef ConvCfg<1, 32, 1, 4, 8, 1, 8, 1, 4, 6> WCfg0;
typedef ConvCfg<1, 16, 2, 2, 16, 1, 4, 2, 4, 6> WCfg2;
typedef ConvCfg<2, 32, 1, 2, 8, 1, 4, 2, 2, 6> WCfg1;       // stride 2, output rows of >= 32
typedef ConvCfg<2, 16, 2, 2, 16, 1, 4, 2, 2, 6> WCfg3;      // stride 2, 16-wide output

static int ws_num_cus() {
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
        if (ncu < 8) ncu = 8;
        ncu &= ~7;
    }
    return ncu;
}
template <class C, bool IN16 = false>
static int launch_ws(const ConvArgs& a_in, hipStream_t st) {
    constexpr int CSL = C::IY * C::IZ * (4 * ((C::IX + 6) / 4));            // channel stride of the quad-aligned LDS tile
    constexpr int lds = (2 * C::CK * CSL + 2 * 27 * C::NT * 64 * (C::CK / 2)) * 4;
    static_assert(lds <= 160 * 1024, "LDS");
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_k3_ws_kernel<C, IN16>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const long long T = (long long)a_in.ntx * a_in.nty * a_in.ntz * a_in.N * a_in.G;
    if (T > 0x7fffffffLL) return wmh_fail(WMH_EINVAL, "conv3d_k3 (wave-specialised): too many tiles");
    const int ncu = ws_num_cus();
    const int grid = T < ncu ? (int)T : ncu;
    ConvArgs a = a_in;
    static const int supertile = wmh_diag_env("WMH_SUPERTILE", 1);
    if (supertile && a.xcd_swizzle) wmh_pick_supertile(a, C::TY, C::TZ, grid / 8 > 0 ? grid / 8 : 1);
#ifdef WMH_DIAG_ABLATIONS
    static const int dbg = wmh_diag_env("WMH_WS_DBG", 0);
#else
    const int dbg = 0;
#endif
    hipLaunchKernelGGL((conv3d_k3_ws_kernel<C, IN16>), dim3(grid), dim3(512), lds, st, a, (int)T, dbg);
    return wmh_check_launch("conv3d_k3_ws_kernel");
}
// tiles a launch needs before the persistent one-workgroup-per-CU form is preferred over the (split-K) one-role kernel
int wmh_conv_ws_min_tiles() { return ws_num_cus() * 3 / 4; }
// a: prepared by wmh_conv3d_k3_fwd with KS = 1 (no split-K: a long cin loop is what this form likes)
int wmh_launch_conv_ws(int cfg, const ConvArgs& a, hipStream_t st) {
    switch (cfg) {
        case 0: return launch_ws<WCfg0>(a, st);
        case 1: return launch_ws<WCfg1>(a, st);
        case 2: return launch_ws<WCfg2>(a, st);
        case 3: return launch_ws<WCfg3>(a, st);
    }
    return wmh_fail(WMH_EINVAL, "conv3d_k3 (wave-specialised): configuration not served");
}
// bf16 sources (config 1: the first strided convolution reads the full-resolution level)
int wmh_launch_conv_ws_in16(int cfg, const ConvArgs& a, hipStream_t st) {
    if (cfg != 1) return wmh_fail(WMH_EINVAL, "conv3d_k3 (wave-specialised): bf16 sources are served for config 1 only");
    return launch_ws<WCfg1, true>(a, st);
}
