// conv3d_up_f16x2.hip -- the "up" half of a decoder stage's first convolution, composed with the transposed convolution in front of it
// (f16x2 pair mode, inference):
//
//     y += conv3x3x3( ConvTranspose3d_{k2,s2}(act(x_low)) )        restricted to the `up` input channels of the 3x3x3 kernel
//
// nnU-Net's decoder stage is  up = tu(x_low);  y = conv(cat(up, skip))  [Generic_UNet.forward; reference call site
// /root/reference/deepwmh/main/predict.py:153-156].  Nothing non-linear sits between the transposed convolution and the convolution, so
// their composition is itself a convolution of the LOW-resolution tensor whose weights depend on the parity class of the output voxel:
// with o = 2 L + p per axis, the three taps o-1, o, o+1 land on the low-resolution voxels L-1 | L, L (p = 0) or L, L | L+1 (p = 1) --
// two low-resolution taps per axis, eight per output voxel instead of 27 on a tensor eight times the size:
//
//     Wc[p][d][co][ci] = sum_cm sum_{t in group(p, d)} W3[co][cm][t] * Wt[ci][cm][q(p, t)]              (host, double precision)
//
// 512 MACs per output and low-resolution channel pair where transposed convolution + 27 taps are 864 + 64, the 4.3 GB `up` tensor of the
// full-resolution stage is neither written nor re-read 2.4 times through the halo, and out-of-volume taps need no special case: every `up`
// voxel depends on exactly one low-resolution voxel, so zero padding of x_low is zero padding of `up`.
// The skip half of the layer runs first as the ordinary pair kernel (conv3d_k3_bf16_ws.hip) on the skip channels alone and leaves
// y = bias + conv(skip); this kernel adds its part, takes the InstanceNorm moments of the sum and stores it.
//
// One persistent 512-thread workgroup per CU.  A workgroup's tile is the voxels of ONE z parity pz of an 8 x 8 x 32 region (4 x 8 x 32 = 1024
// voxels) x 32 output channels: it needs the weights of four of the eight parity classes only -- 32 KB per 8-channel chunk instead of 64 KB for
// the same 96 MFMAs per wave, and the weight stream through the CU's one vector-memory path is what bounds this kernel (with compact 4 x 8 x 32
// tiles the loaders took 5.5 k cycles per chunk against the consumers' 3.6 k, 2.7 k of them issuing the chunk's 24 KB per wave).
//   waves 0-3  consumers.  All 32 voxels of an MFMA's B operand must share their weights, so an M-tile is 32 voxels of ONE parity class:
//              wave w owns py = w & 1, the z half h = w >> 1 and both px; M-tile m = 2 mt + px with mt -> (z, y-pair) and lanes -> 16 x positions
//              x 2 rows.  Per 8-channel chunk and class: 8 taps [w_hi | w_hi] x [x_hi | x_lo] and 4 MFMAs that pair the w_lo x_hi terms of two taps
//              (dz = 0 | 1) -- three of the four partial products, 12 MFMAs per M-tile (the 3x3x3 pair kernel: 41).  B operands from the LDS image
//              of the low-resolution halo tile (4 x 6 x 18 voxels), A fragments from the chunk's 64 KB weight image in LDS (double
//              buffered).  (First version: fragments straight from L2 through an 8-deep register ring -- every wave has its own two classes, 96 KB
//              per chunk and CU, and the MFMA phase took 5.5 k cycles for 3.1 k of issue.)
//              (Round 6, measured and not kept: the chunk's weight image by LDS-DMA -- global_load_lds_dwordx4 issued by the consumer waves, 8 x 1 KB per
//              wave one chunk ahead into the other weight buffer, nothing of the weights on the loader waves; as inline asm, because with the builtin in
//              the loop every counted lgkmcnt of the operand ring became lgkmcnt(0).  Same bits, 9.87 against 9.51 ms for the 128^3 stage, 4.36 / 4.26
//              and 2.18 / 2.14 for the other two, same box, interleaved runs.)
//   waves 4-7  loaders: the next chunk's halo tile as aligned quads, producer's InstanceNorm + LeakyReLU, fp16 split, two 16-byte LDS
//              writes per voxel (the pair kernel's image: 32 bytes per voxel, hi | lo pieces, XOR-swizzled by bit 3 of x); and the next
//              chunk's weight image, 16 x 16 bytes per thread through registers.
// Epilogue: the two px classes of an (mt, r) are the even and odd x of one row: float2 loads of y, add, float2 stores.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <utility>
#include <atomic>
#include <thread>
#include <vector>
#include "conv3d_common.h"

typedef short up_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 up_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 up_f16x2 __attribute__((ext_vector_type(2)));
typedef float up_f32x2 __attribute__((ext_vector_type(2)));
typedef float up_f32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int UTZ = 4, UTY = 8, UTX = 32;                    // output tile (voxels): the 4 z planes of one parity of an 8-plane region
constexpr int LIZ = UTZ + 1, LIY = UTY / 2 + 2, LIX = UTX / 2 + 2;          // low-resolution halo tile: 5 x 6 x 18 (planes L + d - 1 + pz, L = 0..3, d = 0..1)
constexpr int LYS = 20, LZS = LIY * LYS, LVOX = LIZ * LZS;   // row stride padded to 20 voxels: two rows apart = 1280 B = 0 mod 256 (LDS banks)
constexpr int LQPR = (UTX / 2 + 8) / 4, LNQ = LIZ * LIY * LQPR;            // 6 aligned quads per halo row, 180 per chunk
constexpr int UP_RING = 8;                                   // B-operand ring (MFMAs ahead)
constexpr int UP_W_OFF = 40960, UP_W_BYTES = 32768;          // two weight buffers behind the two activation buffers: [4 classes py, px][8 taps][hi 32 | lo 32 pieces] x 16 B
constexpr int UP_WIT = UP_W_BYTES / 16 / 256;                // 16-byte pieces per loader thread and chunk
constexpr int UP_FRAG_BYTES = 8 * 1024;                      // image per (cout group, chunk, class): 8 taps x (32 hi + 32 lo pieces) x 16 B

// fp16(t0 - hi0) | fp16(t1 - hi1) << 16, hi = the halves of h2: two mixed-precision FMAs (conv3d_k3_bf16_ws.hip::ws_lo_pair; same bits as
// convert + subtract + convert)
__device__ __forceinline__ unsigned up_lo_pair(unsigned h2, float t0, float t1) {
    unsigned d;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h2), "v"(t0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(d) : "v"(h2), "v"(t1));
    return d;
}
__device__ __forceinline__ float up_ld_uniform(const float* p) {
    return *reinterpret_cast<const __attribute__((address_space(4))) float*>(reinterpret_cast<unsigned long long>(p));
}
template <int... I, class F>
__device__ __forceinline__ void up_static_for(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
}  // namespace

// ACCUM: y already holds bias + the skip half
template <bool ACCUM>
__global__ __launch_bounds__(512, 2) void conv3d_up_f16x2_kernel(const ConvArgs a, const int T /*tiles x cout groups*/) {
    extern __shared__ __attribute__((aligned(16))) uint4 ldsq[];      // [2][LVOX voxels][2 pieces]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nch = a.nchunks;
    // a.D / a.H / a.W: the OUTPUT (high-resolution) extent; the source is [N][C0][D/2][H/2][W/2]
    const int Dl = a.D >> 1, Hl = a.H >> 1, Wl = a.W >> 1;
    const int HWl = Hl * Wl;
    const size_t DHWl = (size_t)Dl * HWl;

    auto tile_of = [&](int v, int& g, int& tx, int& ty, int& tz, int& n0) __attribute__((always_inline)) {
        const int q = T >> 3, r = T & 7, xcd = v & 7;
        int t = (a.xcd_swizzle & 1) ? (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3) : v;
        wmh_tile_coords(a, t, g, tx, ty, tz, n0);
    };
    int ntiles_mine = 0;
    for (int v = blockIdx.x; v < T; v += gridDim.x) ++ntiles_mine;
    const int total = ntiles_mine * nch;
    // (Staggering every other workgroup by half a tile, so that one half's epilogues -- 256 KB of y traffic per tile and workgroup -- fall into the
    //  other half's MFMA phases, changed nothing: epilogue 18.9 k cycles per tile either way.)

    if (wave >= 4) {
        // ================================================ loaders =================================================================
        __builtin_amdgcn_s_setprio(3);
        const int lt = tid - 256;
        const int row = lt / LQPR, q = lt - row * LQPR;            // this thread's quad of the halo tile (lt < LNQ)
        const bool mine = lt < LNQ;
        int q_off = 0; unsigned q_ok = 0; int n0_cur = 0, g_cur = 0, pz_cur = 0;
        auto setup = [&](int v) __attribute__((always_inline)) {
            int g, tx, ty, tz, n0;
            tile_of(v, g, tx, ty, tz, n0);
            n0_cur = __builtin_amdgcn_readfirstlane(n0); g_cur = __builtin_amdgcn_readfirstlane(g);
            tx = __builtin_amdgcn_readfirstlane(tx); ty = __builtin_amdgcn_readfirstlane(ty); tz = __builtin_amdgcn_readfirstlane(tz);
            pz_cur = tz & 1;                                       // tile tz = (region tz >> 1, z parity tz & 1)
            const int gx = tx * (UTX / 2) - 4 + 4 * q, gy = ty * (UTY / 2) - 1 + row % LIY, gz = (tz >> 1) * UTZ - 1 + pz_cur + row / LIY;
            const bool ok = mine && gx >= 0 && gx < Wl && gy >= 0 && gy < Hl && gz >= 0 && gz < Dl;
            q_off = ok ? gz * HWl + gy * Wl + gx : 0;
            q_ok = ok ? 1u : 0u;
        };
        const unsigned pstride = 4u * (unsigned)DHWl;
        up_f32x4 v[3][8]; unsigned okv[3] = {0, 0, 0}; int set_n0[3] = {0, 0, 0}, set_c[3] = {0, 0, 0}, set_g[3] = {0, 0, 0}, set_pz[3] = {0, 0, 0};
        typedef unsigned up_u32x4 __attribute__((ext_vector_type(4)));
        up_u32x4 wst[UP_WIT];                                        // the weight image of the chunk being converted, on its way to LDS
        int pk = 0, pc = 0, pq = 0;                                  // (tile counter, chunk) of the NEXT issue; pq: its linear index (clamped)
        auto issue = [&](auto sc) __attribute__((always_inline)) {
            constexpr int s = decltype(sc)::value;
            const int c0 = pc * 8, cmax = a.C0 - 1 - c0;
            const char* base = reinterpret_cast<const char*>(a.x0 + ((size_t)n0_cur * a.C0 + c0) * DHWl);
            const unsigned qo = 4u * (unsigned)q_off;
#pragma unroll
            for (int c = 0; c < 8; ++c) v[s][c] = *reinterpret_cast<const up_f32x4*>(base + (qo + (unsigned)min(c, cmax) * pstride));
            okv[s] = q_ok; set_n0[s] = n0_cur; set_c[s] = pc; set_g[s] = g_cur; set_pz[s] = pz_cur;
            // advance the cursor (past the last chunk it stays there: the ring re-reads the last chunk, branch-free loads for exact vmcnt)
            if (pq + 1 < total) {
                ++pq;
                if (++pc == nch) { pc = 0; ++pk; setup(blockIdx.x + pk * gridDim.x); }
            }
        };
        // (the 16 scalar loads of a chunk's scale / shift issued an iteration ahead of their use: measured, no faster)
        float scv[8], shv[8], slope_eff = 1.f;
        static __device__ const float ident2[2] = {1.f, 0.f};
        auto load_ss = [&](auto sc) __attribute__((always_inline)) {
            constexpr int s = decltype(sc)::value;
            const int c0 = __builtin_amdgcn_readfirstlane(set_c[s]) * 8, sn0 = __builtin_amdgcn_readfirstlane(set_n0[s]);
            const bool has_ss = a.ss0 != nullptr;
            slope_eff = has_ss ? a.slope : 1.f;
            const float* tab = has_ss ? a.ss0 + ((size_t)sn0 * a.C0 + c0) * 2 : ident2;
            const int mul = has_ss ? 2 : 0, cmax = a.C0 - 1 - c0;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float sc1 = up_ld_uniform(tab + mul * min(c, cmax)), sh1 = up_ld_uniform(tab + mul * min(c, cmax) + 1);
                scv[c] = c <= cmax ? sc1 : 0.f;
                shv[c] = c <= cmax ? sh1 : 0.f;
            }
        };
#ifdef WMH_DIAG
        const int abl = a.KS;                                         // diagnostic library (timing only; WMH_UP_ABL): 2 = weights loaded at the start only, 4 = conflict-free LDS writes
#else
        constexpr int abl = 1;                                        // the product build has no such switch
#endif
        auto convert = [&](auto sc, int buf) __attribute__((always_inline)) {
            constexpr int s = decltype(sc)::value;
            uint4* dst = ldsq + (size_t)buf * LVOX * 2;
            const unsigned keep = okv[s] ? 0xFFFFFFFFu : 0u;
            unsigned w[4][4], wlo[4][4];
#pragma unroll
            for (int c = 0; c < 8; c += 2) {
                up_f32x4 t[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float sc1 = scv[c + e], sh1 = shv[c + e];
                    t[e] = up_f32x4{wmh_norm_lrelu(v[s][c + e][0], sc1, sh1, slope_eff), wmh_norm_lrelu(v[s][c + e][1], sc1, sh1, slope_eff),
                                    wmh_norm_lrelu(v[s][c + e][2], sc1, sh1, slope_eff), wmh_norm_lrelu(v[s][c + e][3], sc1, sh1, slope_eff)};
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {                      // fp16 pair: high part toward zero, the remainder to nearest (the pair kernel's split)
                    const unsigned h2 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(t[0][k], t[1][k]));
                    w[k][c >> 1] = h2 & keep;
                    wlo[k][c >> 1] = up_lo_pair(h2, t[0][k], t[1][k]) & keep;
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int x = 4 * q - 3 + k;                       // position in the halo row (0 .. LIX - 1)
                if (mine && x >= 0 && x < LIX) {
                    const int e = row * LYS + x;
                    const int ea = (abl == 4) ? (lt & 255) + 256 * (k & 1) : e;          // diagnostics (timing only): 4 = conflict-free write addresses
                    dst[ea * 2 + (0 ^ ((x >> 3) & 1))] = make_uint4(w[k][0], w[k][1], w[k][2], w[k][3]);
                    dst[ea * 2 + (1 ^ ((x >> 3) & 1))] = make_uint4(wlo[k][0], wlo[k][1], wlo[k][2], wlo[k][3]);
                }
            }
        };
        constexpr auto S0 = std::integral_constant<int, 0>{};
        constexpr auto S1 = std::integral_constant<int, 1>{};
        constexpr auto S2 = std::integral_constant<int, 2>{};
        auto w_issue = [&](auto sc) __attribute__((always_inline)) {   // the weight image (the four classes of the tile's pz) of the chunk in activation set s
            constexpr int s = decltype(sc)::value;
            if (abl == 2 && pq > 3) return;
            const int g = __builtin_amdgcn_readfirstlane(set_g[s]), c = __builtin_amdgcn_readfirstlane(set_c[s]), z1 = __builtin_amdgcn_readfirstlane(set_pz[s]);
            const up_u32x4* src = reinterpret_cast<const up_u32x4*>(reinterpret_cast<const char*>(a.wf) + (((size_t)g * nch + c) * 2 + z1) * UP_W_BYTES) + lt;
#pragma unroll
            for (int e = 0; e < UP_WIT; ++e) wst[e] = src[e * 256];
        };
        auto w_write = [&](int buf) __attribute__((always_inline)) {
            if (abl == 2 && pq > 4) return;
            up_u32x4* dst = reinterpret_cast<up_u32x4*>(reinterpret_cast<char*>(ldsq) + UP_W_OFF + buf * UP_W_BYTES) + lt;
#pragma unroll
            for (int e = 0; e < UP_WIT; ++e) dst[e * 256] = wst[e];
        };
        if (total > 0) {
            setup(blockIdx.x);
            issue(S0);                                             // chunk 0 (exposed once per workgroup)
            w_issue(S0);
            issue(S1);                                             // chunks 1 and 2 in flight across the barrier
            issue(S2);
            load_ss(S0);
            convert(S0, 0);
            w_write(0);
        }
        __syncthreads();
        // iteration it: the consumers are on chunk it (activation and weight buffers it & 1).  First the weights of chunk it + 1 (from L2; written
        // at the end of the iteration), then the activation loads of chunk it + 3 into the register set chunk it came from -- TWO iterations
        // ahead of their conversion: with one, `convert` waited 3 k cycles per chunk for them under the epilogues' traffic -- then chunk it + 1
        // is normalised, split and written into the other buffer.
        unsigned long long lw = 0, lb = 0, lt0 = 0, l_is = 0, l_cv = 0, l_ww = 0;   // diagnostics (WMH_STAMPS): loader work / barrier wait; issue / convert / weight write
        auto iteration = [&](auto s_conv, auto s_issue, int buf) __attribute__((always_inline)) {
            if (WMH_STAMPS(a)) lt0 = __builtin_amdgcn_s_memtime();
            unsigned long long p0 = lt0;
            w_issue(s_conv);
            load_ss(s_conv);
            issue(s_issue);
            if (WMH_STAMPS(a)) { const unsigned long long t = __builtin_amdgcn_s_memtime(); l_is += t - p0; p0 = t; }
            convert(s_conv, buf);
            if (WMH_STAMPS(a)) { const unsigned long long t = __builtin_amdgcn_s_memtime(); l_cv += t - p0; p0 = t; }
            w_write(buf);
            if (WMH_STAMPS(a)) { const unsigned long long t = __builtin_amdgcn_s_memtime(); l_ww += t - p0; lw += t - lt0; lt0 = t; }
            __syncthreads();
            if (WMH_STAMPS(a)) lb += __builtin_amdgcn_s_memtime() - lt0;
        };
        for (int it = 0; it < total; it += 3) {                    // register sets by it % 3, buffers by (it + 1) & 1
            iteration(S1, S0, (it + 1) & 1);
            if (it + 1 >= total) break;
            iteration(S2, S1, it & 1);
            if (it + 2 >= total) break;
            iteration(S0, S2, (it + 1) & 1);
        }
        if (WMH_STAMPS(a) && tid == 256) {
            unsigned long long* o = WMH_STAMPS(a) + (size_t)blockIdx.x * 8;
            o[4] = lw; o[5] = lb;
            unsigned long long* o2 = WMH_STAMPS(a) + (size_t)(gridDim.x + blockIdx.x) * 8;
            o2[0] = l_is; o2[1] = l_cv; o2[2] = l_ww;
        }
        return;
    }

    // ================================================== consumers =================================================================
    const int j = lane & 31, kh = lane >> 5;
    const int i = j & 15, yy = j >> 4;
    const int zh = wave >> 1, py = wave & 1;                      // this wave's z half (zc = 2 zh + (mt >> 1)) and y parity
    // B operand addresses (bytes inside a buffer), per mt = (zc, yh) and x shift s = px + dx (0 .. 2):
    //   bA: the lane half's piece (hi for kh 0, lo for kh 1) of halo voxel (zc, yh + 2 yy + py, i + s) -- [x_hi | x_lo]  (the tile's pz is in the halo's origin)
    //   bP: the hi piece of that voxel for kh 0 and of the voxel one slice deeper (dz + 1) for kh 1   -- [x_hi(dz 0) | x_hi(dz 1)]
    int bA[4][3], bP[4][3];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int zc = 2 * zh + (mt >> 1), yh = mt & 1;
        const int eb = (zc * LIY + (yh + 2 * yy + py)) * LYS + i;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int key = ((i + s) >> 3) & 1;
            bA[mt][s] = (eb + s) * 32 + 16 * (kh ^ key);
            bP[mt][s] = (eb + s + kh * LZS) * 32 + 16 * key;
        }
    }
    const char* const ldsb = reinterpret_cast<const char*>(ldsq);
    // A fragments: image [g][chunk][class = pz 4 + py 2 + px][tap tau = dz 4 + dy 2 + dx][hi pieces of cout 0..31 | lo pieces] x 16 B, a chunk's
    // 64 KB in LDS.
    //   f 0..7   [w_hi(tau f) | w_hi(tau f)]: both lane halves read hi piece j of tap f
    //   f 8..11  [w_lo(tau f - 8) | w_lo(tau f - 4)]: lo piece j of tap (dz 0, dy, dx) on lanes 0-31, of tap (dz 1, dy, dx) on lanes 32-63
    const int offH = j * 16, offL = 512 + j * 16 + kh * 4096;
    constexpr int NF = 24;                                       // fragments per chunk: 2 classes (px) x 12
    auto frag_off = [&](int f) __attribute__((always_inline)) {   // byte offset of fragment f (0 .. 23) inside this wave's two class images
        const int px = f / 12, ff = f % 12;
        return px * UP_FRAG_BYTES + (ff < 8 ? ff * 1024 + offH : (ff - 8) * 1024 + offL);
    };
    const int cls0 = UP_W_OFF + (py * 2) * UP_FRAG_BYTES;            // this wave's first class (px 0) among the buffer's four (py, px)
    __syncthreads();                                             // chunk 0 is staged
    int it = 0;
    unsigned long long tm = 0, tb = 0, te = 0, t0 = 0;           // diagnostics (WMH_STAMPS): MFMA phases / barrier waits / epilogues
    for (int k = 0; k < ntiles_mine; ++k) {
        int g, tx, ty, tz, n0;
        tile_of(blockIdx.x + k * gridDim.x, g, tx, ty, tz, n0);
        f32x16 acc[8];
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
        for (int ch = 0; ch < nch; ++ch, ++it) {
            if (WMH_STAMPS(a)) t0 = __builtin_amdgcn_s_memtime();
            const char* cur = ldsb + (size_t)(it & 1) * LVOX * 32;
            const char* wcur = ldsb + cls0 + (it & 1) * UP_W_BYTES;
            // step st = f * 4 + mt: fragment f (class px = f / 12) against M-tile 2 mt + px
            auto loadB = [&](auto sc) __attribute__((always_inline)) -> up_h8 {
                constexpr int st = decltype(sc)::value, f = st >> 2, mt = st & 3;
                constexpr int px = f / 12, ff = f % 12;
                if constexpr (ff < 8) {
                    constexpr int dz = ff >> 2, dy = (ff >> 1) & 1, dx = ff & 1;
                    return *reinterpret_cast<const up_h8*>(cur + bA[mt][px + dx] + (dz * LZS + dy * LYS) * 32);
                } else {
                    constexpr int dy = ((ff - 8) >> 1) & 1, dx = (ff - 8) & 1;
                    return *reinterpret_cast<const up_h8*>(cur + bP[mt][px + dx] + (dy * LYS) * 32);
                }
            };
            up_h8 ring[UP_RING], aF[2];
            aF[0] = *reinterpret_cast<const up_h8*>(wcur + frag_off(0));
            up_static_for(std::make_integer_sequence<int, UP_RING>{}, [&](auto ic) __attribute__((always_inline)) { ring[decltype(ic)::value] = loadB(ic); });
            __builtin_amdgcn_sched_barrier(0);
            up_static_for(std::make_integer_sequence<int, NF * 4>{}, [&](auto sc) __attribute__((always_inline)) {
                constexpr int st = decltype(sc)::value, f = st >> 2, mt = st & 3, px = f / 12;
                if constexpr (mt == 0 && f + 1 < NF) aF[(f + 1) & 1] = *reinterpret_cast<const up_h8*>(wcur + frag_off(f + 1));     // one group ahead
                const up_h8 bv = ring[st % UP_RING];
                acc[2 * mt + px] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(up_f16x8, aF[f & 1]), __builtin_bit_cast(up_f16x8, bv),
                                                                           acc[2 * mt + px], 0, 0, 0);
                if constexpr (st + UP_RING < NF * 4) ring[st % UP_RING] = loadB(std::integral_constant<int, st + UP_RING>{});
                __builtin_amdgcn_sched_barrier(0);
            });
            if (WMH_STAMPS(a)) { const unsigned long long t = __builtin_amdgcn_s_memtime(); tm += t - t0; t0 = t; }
            __syncthreads();                                       // this chunk is consumed, the next one is staged
            if (WMH_STAMPS(a)) tb += __builtin_amdgcn_s_memtime() - t0;
        }
        if (WMH_STAMPS(a)) t0 = __builtin_amdgcn_s_memtime();
        // ---- epilogue: (acc / 2^8) [+ y], InstanceNorm moments of the result, stores ------------------------------------------------
        const int HW = a.H * a.W;
        const size_t DHW = (size_t)a.D * HW;
        const int cob = g * 32;
        float* const ych = a.y + ((size_t)n0 * a.Cout + cob) * DHW;
        const unsigned plane = 4u * (unsigned)DHW;
        const int oz0 = (tz >> 1) * (2 * UTZ) + (tz & 1) + 4 * zh, oy0 = ty * UTY + py + 4 * yy, ox0 = tx * UTX + 2 * i;   // z = 8 region + pz + 2 zc
        char* const ychb = reinterpret_cast<char*>(ych);
        // acc <- acc / 2^8 (+ y), stored at once (the two px classes of (mt, r) are the even and odd x of one row); with ACCUM the old values of
        // row group mt + 2 are requested while group mt is added and stored: two groups (64 registers) in flight instead of four exposed round trips
        auto lane_off = [&](int mt) __attribute__((always_inline)) {
            return (unsigned)(4 * kh) * plane + 4u * (unsigned)((oz0 + 2 * (mt >> 1)) * HW + (oy0 + 2 * (mt & 1)) * a.W + ox0);
        };
        up_f32x2 old[2][16];
        auto request = [&](auto mc) __attribute__((always_inline)) {
            constexpr int mt = decltype(mc)::value;
            if constexpr (ACCUM && mt < 4) {
                const unsigned loff = lane_off(mt);
#pragma unroll
                for (int r = 0; r < 16; ++r)                       // row r = channel plane (r & 3) + 8 (r >> 2) (+ 4 kh in the lane offset): uniform bases
                    old[mt & 1][r] = *reinterpret_cast<const up_f32x2*>(ychb + (size_t)((r & 3) + 8 * (r >> 2)) * plane + loff);
            }
        };
        request(std::integral_constant<int, 0>{});
        request(std::integral_constant<int, 1>{});
        up_static_for(std::make_integer_sequence<int, 4>{}, [&](auto mc) __attribute__((always_inline)) {
            constexpr int mt = decltype(mc)::value;
            const unsigned loff = lane_off(mt);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v0, v1;
                if constexpr (ACCUM) {
                    v0 = fmaf(acc[2 * mt][r], 1.f / WMH_PAIR_WSCALE, old[mt & 1][r][0]);
                    v1 = fmaf(acc[2 * mt + 1][r], 1.f / WMH_PAIR_WSCALE, old[mt & 1][r][1]);
                } else {
                    v0 = acc[2 * mt][r] * (1.f / WMH_PAIR_WSCALE); v1 = acc[2 * mt + 1][r] * (1.f / WMH_PAIR_WSCALE);
                }
                acc[2 * mt][r] = v0; acc[2 * mt + 1][r] = v1;
                *reinterpret_cast<up_f32x2*>(ychb + (size_t)((r & 3) + 8 * (r >> 2)) * plane + loff) = up_f32x2{v0, v1};
            }
            __builtin_amdgcn_sched_barrier(0);
            request(std::integral_constant<int, mt + 2>{});
            __builtin_amdgcn_sched_barrier(0);
        });
        if (a.part != nullptr) {
            // one-pass moments about a per-row shift K (conv3d_common.h::conv_epilogue, all 8 x 32 voxels of the wave valid)
            float vals[32], kshift[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v0 = acc[0][r];
                const float k0 = __builtin_amdgcn_readlane(v0, 0), k1 = __builtin_amdgcn_readlane(v0, 32);
                const float K = kh ? k1 : k0;
                kshift[r] = K;
                float sd = 0.f, sq = 0.f;
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    const float d = acc[m][r] - K;
                    sd += d;
                    sq = fmaf(d, d, sq);
                }
                vals[2 * r] = sd; vals[2 * r + 1] = sq;
            }
            // (DPP row adds instead of the recursive halving -- no permutes, no selects, the last lane writing all 16 records -- measured no faster
            //  here and made an unrelated instance of the shared epilogue spill: not kept)
            halve_step<16>(vals, j); halve_step<8>(vals, j); halve_step<4>(vals, j); halve_step<2>(vals, j); halve_step<1>(vals, j);
            const float mine_v = vals[0];
            const float other = __shfl_xor(mine_v, 1);
            float Kr = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) Kr = ((j >> 1) == r) ? kshift[r] : Kr;
            if ((j & 1) == 0) {
                const int r = j >> 1;
                const int co = cob + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (co < a.Cout) {
                    const float fc = 256.f;
                    const float sd = mine_v, sq = other;
                    const float m2 = fmaxf(sq - sd * sd / fc, 0.f);
                    const int pidx = ((tz * a.nty + ty) * a.ntx + tx) * 4 + wave;
                    float4* pp = reinterpret_cast<float4*>(a.part) + ((size_t)n0 * a.Cout + co) * a.npart + pidx;
                    *pp = make_float4(fc, fmaf(fc, Kr, sd), m2, 0.f);
                }
            }
        }
        if (WMH_STAMPS(a)) te += __builtin_amdgcn_s_memtime() - t0;
    }
    if (WMH_STAMPS(a) && tid == 0) {
        unsigned long long* o = WMH_STAMPS(a) + (size_t)blockIdx.x * 8;
        o[0] = tm; o[1] = tb; o[2] = te; o[3] = (unsigned long long)total;
    }
}

extern unsigned long long* g_conv_stamps;      // conv3d_k3.hip (diagnostics)
// ---- host side -----------------------------------------------------------------------------------------------------------------------
extern "C" int wmh_conv3d_up_supported_f16x2(int Clow, int Cup, int Cout, int D, int H, int W) {
    // whole 8 x 8 x 32 output regions, 32 output channels per group, quads along the low-resolution rows
    return (Clow >= 1 && Cup >= 1 && Cout >= 1 && (Cout % 32) == 0 && D > 0 && H > 0 && W > 0 && (D % (2 * UTZ)) == 0 && (H % UTY) == 0 && (W % UTX) == 0 &&
            (size_t)32 * D * H * W * 4 < ((size_t)1 << 32) && (size_t)Clow * (D / 2) * (H / 2) * (W / 2) * 4 < ((size_t)1 << 32)) ? 1 : 0;
}
extern "C" size_t wmh_conv3d_up_packed_bytes_f16x2(int Clow, int Cout) {
    return (size_t)((Cout + 31) / 32) * ((Clow + 7) / 8) * 8 * UP_FRAG_BYTES;
}
extern "C" int wmh_conv3d_up_num_partials_f16x2(int D, int H, int W) {
    return ((W + UTX - 1) / UTX) * ((H + UTY - 1) / UTY) * ((D + UTZ - 1) / UTZ) * 4;
}
// w3: the 3x3x3 convolution's weight [Cout][Cin_total][27] whose input channels [0, Cup) are the transposed convolution's outputs
// (torch.cat((up, skip), 1)); wt: ConvTranspose3d(k2, s2) weight [Clow][Cup][2][2][2].  wimg: host memory, wmh_conv3d_up_packed_bytes_f16x2 bytes.
extern "C" int wmh_conv3d_up_pack_weights_f16x2(const float* w3, int Cin_total, int Cup, const float* wt, int Clow, int Cout, void* wimg) {
    if (!w3 || !wt || !wimg || Cup < 1 || Cup > Cin_total || Clow < 1 || Cout < 1) return wmh_fail(WMH_EINVAL, "conv3d_up_pack_weights_f16x2: bad args");
    const int G = (Cout + 31) / 32, NCH = (Clow + 7) / 8;
    uint16_t* out = static_cast<uint16_t*>(wimg);
    memset(out, 0, wmh_conv3d_up_packed_bytes_f16x2(Clow, Cout));
    // the transposed convolution's weight as [q][cm][ci] doubles: the composition's inner loop then runs along contiguous ci (vectorisable);
    // as written first -- a stride of 8 Cup floats per step -- the 128-channel stage took seconds of `nnUNet_predict`'s start-up
    std::vector<double> wtT((size_t)8 * Cup * Clow);
    for (int ci = 0; ci < Clow; ++ci)
        for (int cm = 0; cm < Cup; ++cm)
            for (int q = 0; q < 8; ++q) wtT[((size_t)q * Cup + cm) * Clow + ci] = (double)wt[((size_t)ci * Cup + cm) * 8 + q];
    // One host thread per parity class (round 6: the classes' images are disjoint; every output is formed by one thread in the same order as before,
    // so the image does not depend on the thread count -- 0.18 s of a fresh process's first forward were this function, single-threaded).
    std::atomic<int> bad(0);
    auto one_class = [&](int cls) {
        std::vector<double> wc((size_t)Cout * Clow);
        const int p[3] = {(cls >> 2) & 1, (cls >> 1) & 1, cls & 1};
        for (int tau = 0; tau < 8; ++tau) {
            const int d[3] = {(tau >> 2) & 1, (tau >> 1) & 1, tau & 1};
            std::fill(wc.begin(), wc.end(), 0.0);
            // taps t (0..2 per axis, offset t - 1) whose voxel o + t - 1 lies in the low-resolution voxel L + d - 1 + p, and that voxel's parity q
            for (int tz = 0; tz < 3; ++tz) for (int ty = 0; ty < 3; ++ty) for (int tx = 0; tx < 3; ++tx) {
                const int t[3] = {tz, ty, tx};
                int q[3]; bool in = true;
                for (int ax = 0; ax < 3; ++ax) {
                    const int u = p[ax] + t[ax] - 1;                 // position relative to 2 L
                    const int l = (u >= 0) ? u / 2 : -1;             // floor(u / 2) for u in -1 .. 2
                    q[ax] = u - 2 * l;
                    if (l != d[ax] - 1 + p[ax]) in = false;
                }
                if (!in) continue;
                const int tap = (tz * 3 + ty) * 3 + tx, qq = (q[0] * 2 + q[1]) * 2 + q[2];
                for (int co = 0; co < Cout; ++co)
                    for (int cm = 0; cm < Cup; ++cm) {
                        const double w = w3[((size_t)co * Cin_total + cm) * 27 + tap];
                        double* __restrict dst = &wc[(size_t)co * Clow];
                        const double* __restrict src = &wtT[((size_t)qq * Cup + cm) * Clow];
                        for (int ci = 0; ci < Clow; ++ci) dst[ci] += w * src[ci];
                    }
            }
            for (int g = 0; g < G; ++g)
                for (int ch = 0; ch < NCH; ++ch)
                    for (int col = 0; col < 32; ++col) {
                        const int co = g * 32 + col;
                        uint16_t* hi = out + ((((size_t)g * NCH + ch) * 8 + cls) * UP_FRAG_BYTES + (size_t)tau * 1024 + col * 16) / 2;
                        uint16_t* lo = hi + 256;
                        for (int e = 0; e < 8; ++e) {
                            const int ci = ch * 8 + e;
                            float v = (co < Cout && ci < Clow) ? (float)wc[(size_t)co * Clow + ci] : 0.f;
                            v *= WMH_PAIR_WSCALE;
                            // a composed weight sums up to Cup products: it can leave fp16's range although both layers pass their own checks;
                            // refused like the other packers refuse (NaN fails the comparison too); the engine then keeps the two-kernel stage
                            if (!(fabsf(v) < 65000.f)) { bad.store(1); return; }
                            const _Float16 h = (_Float16)v;
                            const _Float16 l = (_Float16)(v - (float)h);
                            memcpy(&hi[e], &h, 2); memcpy(&lo[e], &l, 2);
                        }
                    }
        }
    };
    {
        std::vector<std::thread> th;
        for (int cls = 1; cls < 8; ++cls) th.emplace_back(one_class, cls);
        one_class(0);
        for (auto& t : th) t.join();
    }
    if (bad.load()) return wmh_fail(WMH_EINVAL, "conv3d_up_pack_weights_f16x2: composed weight not finite or beyond fp16's range (|256 w| >= 65000)");
    return WMH_OK;
}

// y (+)= the composed convolution of x_low; accumulate != 0: y already holds bias + the skip half (the pair kernel on the skip channels);
// part: InstanceNorm moment records of the result ([N][Cout][wmh_conv3d_up_num_partials_f16x2][4]) or NULL.
extern "C" int wmh_conv3d_up_fwd_f16x2(const float* xlow, const float* sslow, int Clow, const void* wimg, float* y, float* part,
                                        int N, int D, int H, int W, int Cout, float lrelu_slope, int accumulate, void* stream) {
    if (!xlow || !wimg || !y || N < 1) return wmh_fail(WMH_EINVAL, "conv3d_up_fwd_f16x2: null/empty argument");
    if (!wmh_conv3d_up_supported_f16x2(Clow, 1, Cout, D, H, W)) return wmh_fail(WMH_EINVAL, "conv3d_up_fwd_f16x2: needs whole 8 x 8 x 32 output regions and Cout % 32 == 0");
    if ((reinterpret_cast<uintptr_t>(xlow) & 15) != 0 || (reinterpret_cast<uintptr_t>(y) & 7) != 0 || ((W / 2) % 4) != 0)
        return wmh_fail(WMH_EINVAL, "conv3d_up_fwd_f16x2: 16-byte aligned source rows");
    if (!(lrelu_slope >= 0.f && lrelu_slope <= 1.f)) return wmh_fail(WMH_EINVAL, "conv3d_up_fwd_f16x2: LeakyReLU slope must be in [0, 1]");
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x0 = xlow; a.ss0 = sslow; a.C0 = Clow; a.wf = static_cast<const float*>(wimg); a.y = y; a.part = part;
    a.N = N; a.D = D; a.H = H; a.W = W; a.Do = D; a.Ho = H; a.Wo = W; a.Cout = Cout;
    a.nchunks = (Clow + 7) / 8;
    a.ntx = W / UTX; a.nty = H / UTY; a.ntz = D / UTZ;
    a.npart = a.ntx * a.nty * a.ntz * 4;
    static const int abl_env = wmh_diag_env("WMH_UP_ABL", 1);
    a.G = Cout / 32; a.KS = abl_env; a.xcd_swizzle = 1; a.by = a.bz = 1; a.slope = lrelu_slope; a.accum = accumulate ? 1 : 0;
    a.stamps = g_conv_stamps;
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
        if (ncu < 8) ncu = 8;
        ncu &= ~7;
    }
    const long long T = (long long)a.ntx * a.nty * a.ntz * N * a.G;
    if (T > 0x7fffffffLL) return wmh_fail(WMH_EINVAL, "conv3d_up_fwd_f16x2: too many tiles");
    const int grid = T < ncu ? (int)T : ncu;
    wmh_pick_supertile(a, UTY, UTZ, grid / 8 > 0 ? grid / 8 : 1);
    constexpr int lds = UP_W_OFF + 2 * UP_W_BYTES;
    static_assert(2 * LVOX * 32 <= UP_W_OFF && lds <= 160 * 1024, "LDS");
    auto launch = [&](auto kern) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, (hipStream_t)stream, a, (int)T);
    };
    if (accumulate) launch(&conv3d_up_f16x2_kernel<true>);
    else launch(&conv3d_up_f16x2_kernel<false>);
    return wmh_check_launch("conv3d_up_f16x2_kernel");
}
