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

template <class C>
__global__ __launch_bounds__(512, 2) void conv3d_k3_ws_kernel(const ConvArgs a, const int T /*tiles*/, const int dbg /*timing experiments; 0 in production*/) {
    constexpr int S = C::S, MT = C::MT, NT = C::NT, CK = C::CK, V = CK / 2;
    constexpr int LTH = 256;                                   // loader threads
    constexpr int NIT = (C::SP + LTH - 1) / LTH;
    constexpr int CS = NIT * LTH, TILE = CK * CS;              // padded channel stride: every loader store in range
    constexpr int WT = 27 * NT * 64 * V;                       // weight floats per chunk
    constexpr int WIT = (WT / 2 + LTH - 1) / LTH;              // float2 pieces per loader thread
    static_assert(C::TN == 1 && NT == 1 && C::WM == 4, "geometry of the large-layer configuration");
    static_assert(((2 * C::ZS + 2 * C::YS + 2) + (CK - 2) * CS + TILE) * 4 < 65536, "ds_read immediate range");
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
        g = t % a.G; t /= a.G;
        tx = t % a.ntx; t /= a.ntx;
        ty = t % a.nty; t /= a.nty;
        tz = t % a.ntz; t /= a.ntz;
        n0 = t;
    };
    int ntiles_mine = 0;
    for (int v = blockIdx.x; v < T; v += gridDim.x) ++ntiles_mine;
    const int total = ntiles_mine * nch;                    // chunk iterations of this workgroup

    if (loader) {
        // ================================ loader waves =====================================================
        const int lt = tid - 256;
        int sp_off[NIT];
        unsigned okmask = 0;
        int g = 0, n0 = 0;
        float stg[CK][NIT];
        float2 wst[WIT];
        float sc[CK], sh[CK], slope_eff = 1.f;
        unsigned cmask = 0;
        auto setup = [&](int v) {
            int tx, ty, tz;
            tile_of(v, g, tx, ty, tz, n0);
            const int ix0 = tx * C::TX * S - 1, iy0 = ty * C::TY * S - 1, iz0 = tz * C::TZ * S - 1;
            okmask = 0;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int e = lt + it * LTH;
                const int x = e % C::IX, y = (e / C::IX) % C::IY, z = (e / (C::IX * C::IY)) % C::IZ;
                const int gx = ix0 + x, gy = iy0 + y, gz = iz0 + z;
                const bool ok = (e < C::SP) && gx >= 0 && gx < a.W && gy >= 0 && gy < a.H && gz >= 0 && gz < a.D;
                sp_off[it] = ok ? gz * HW + gy * a.W + gx : 0;
                okmask |= ok ? (1u << it) : 0u;
            }
        };
        auto load_chunk = [&](int ch) {
            const int cb = ch * CK;
            const float* xs; const float* ss; int Cs, c0;
            if (cb < a.C0) { xs = a.x0; ss = a.ss0; Cs = a.C0; c0 = cb; }
            else           { xs = a.x1; ss = a.ss1; Cs = a.C1; c0 = cb - a.C0; }
            cmask = 0;
#pragma unroll
            for (int cc = 0; cc < CK; ++cc) {
                const int c = min(c0 + cc, Cs - 1);
                cmask |= (c0 + cc) < Cs ? (1u << cc) : 0u;
#pragma unroll
                for (int it = 0; it < NIT; ++it) stg[cc][it] = xs[((size_t)n0 * Cs + c) * DHW + sp_off[it]];
            }
            const bool has_ss = ss != nullptr;
            slope_eff = has_ss ? a.slope : 1.f;
#pragma unroll
            for (int cc = 0; cc < CK; ++cc) {
                const int c = min(c0 + cc, Cs - 1);
                sc[cc] = has_ss ? ss[((size_t)n0 * Cs + c) * 2] : 1.f;
                sh[cc] = has_ss ? ss[((size_t)n0 * Cs + c) * 2 + 1] : 0.f;
            }
            const float2* wsrc = reinterpret_cast<const float2*>(a.wf) + ((size_t)g * nch + ch) * (WT / 2);
#pragma unroll
            for (int i = 0; i < WIT; ++i) {
                const int k = min(lt + i * LTH, WT / 2 - 1);
                wst[i] = wsrc[k];
            }
        };
        auto write_chunk = [&](int buf) {
            float* xb = X + buf * TILE;
#pragma unroll
            for (int cc = 0; cc < CK; ++cc)
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    float v = fmaf(stg[cc][it], sc[cc], sh[cc]);
                    v = v > 0.f ? v : v * slope_eff;
                    v = (((cmask >> cc) & 1u) && ((okmask >> it) & 1u)) ? v : 0.f;
                    xb[cc * CS + lt + it * LTH] = v;
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
        for (int it = 0; it < total; ++it) {
            const int nx = it + 1;
            if (nx < total && !(dbg & 1)) {
                const int ch = nx % nch;
                if (ch == 0) { vnext += gridDim.x; setup(vnext); }
                load_chunk(ch);
                write_chunk(nx & 1);
            }
            __syncthreads();
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
            baddr[m] = kh * CS + S * ((tzi * C::LZ + dz) * C::ZS + (tyi * C::LY + dy) * C::YS + dx);
        }
        __syncthreads();
        int it = 0;
        for (int v = blockIdx.x; v < T; v += gridDim.x) {
            int g, tx, ty, tz, n0;
            tile_of(v, g, tx, ty, tz, n0);
            f32x16 acc[NT][MT];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = g * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                const float bv = (a.bias != nullptr && co < a.Cout) ? a.bias[co] : 0.f;
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[0][m][r] = bv;
            }
            for (int ch = 0; ch < nch; ++ch, ++it) {
                const float* cur = X + (it & 1) * TILE;
                const float2* wcur = reinterpret_cast<const float2*>(Wl + (it & 1) * WT) + lane;
#pragma unroll
                for (int tap = 0; tap < 27; ++tap) {
                    const int toff = (tap / 9) * C::ZS + ((tap / 3) % 3) * C::YS + (tap % 3);
                    const float2 av = wcur[tap * 64];
#pragma unroll
                    for (int p = 0; p < V; ++p) {
#pragma unroll
                        for (int m = 0; m < MT; ++m) {
                            const float bv = cur[baddr[m] + toff + 2 * p * CS];
                            acc[0][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(p == 0 ? av.x : av.y, bv, acc[0][m], 0, 0, 0);
                        }
                    }
                }
                __syncthreads();
            }
            if (!(dbg & 2))
                conv_epilogue<C>(a, acc, g, 0, n0, wave, j, kh, dx, dy, dz, ts0, tz * C::TZ, ty * C::TY, tx * C::TX,
                                 (tz * a.nty + ty) * a.ntx + tx);
            else if (acc[0][0][0] == 12345.678f) a.y[0] = acc[0][1][3];     // keep the accumulators live
        }
    }
}

typedef ConvCfg<1, 32, 1, 4, 8, 1, 8, 1, 4, 6> WCfg0;

// returns 1 when this form applies to the launch described by `a` (cfg0 geometry, no split-K)
int wmh_launch_conv_ws_cfg0(const ConvArgs& a, hipStream_t st) {
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
        if (ncu < 8) ncu = 8;
        ncu &= ~7;
    }
    constexpr int NITL = (WCfg0::SP + 255) / 256;
    const int lds = (2 * WCfg0::CK * NITL * 256 + 2 * 27 * 64 * (WCfg0::CK / 2)) * 4;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_k3_ws_kernel<WCfg0>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr = true;
    }
    const long long T = (long long)a.ntx * a.nty * a.ntz * a.N * a.G;
    if (T > 0x7fffffffLL) return wmh_fail(WMH_EINVAL, "conv3d_k3 (wave-specialised): too many tiles");
    const int grid = T < ncu ? (int)T : ncu;
    static const int dbg = getenv("WMH_WS_DBG") ? atoi(getenv("WMH_WS_DBG")) : 0;
    hipLaunchKernelGGL((conv3d_k3_ws_kernel<WCfg0>), dim3(grid), dim3(512), lds, st, a, (int)T, dbg);
    return wmh_check_launch("conv3d_k3_ws_kernel");
}
