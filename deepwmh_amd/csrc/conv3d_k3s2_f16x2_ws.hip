// conv3d_k3s2_f16x2_ws.hip -- the 3x3x3 convolution with stride 2 and output rows of 32 voxels (configuration 1: the first convolution of
// the 64^3 and 32^3 stages) in the f16x2 "pair" mode, wave-specialised and persistent.  Round 5: rebuilt on both sides at once.
//
// What the round-3/4 kernel was (git history): eight loader waves fetching single dwords of a 5 x 5 x 65 input box (x positions are not
// 16-byte aligned under stride 2), one dword (hi | lo << 16) per value, four consumer waves forming all FOUR partial products with a 16-bit
// rotate of the weight fragments, weights straight from L2 -- 216 KB of fragment loads per chunk and CU through the one vector-memory path,
// 216 v_alignbit per chunk and wave, 7 vector instructions per staged value: MFMA phase 6.2 k cycles for 3.5 k of issue, loaders 5.1 k;
// moving the weights to LDS alone, or the loader alone, changed nothing ("needs both sides rebuilt").
//
// Now the stride-1 pair kernel's structure (conv3d_k3_bf16_ws.hip) on a PARITY-SPLIT LDS tile.  With o the output voxel, tap k reads input
// 2 o + k - 1: k = 0 -> odd input o - 1, k = 1 -> even input o, k = 2 -> odd input o (per axis).  The 5 x 5 x 65 box is staged as its eight
// parity sub-tiles (even: T voxels per axis, odd: T + 1), so that
//   * the loaders fetch ALIGNED quads of the input rows (one 16-byte load per channel, 17 per row) and only the LDS address of a voxel
//     knows about parity: the stride-1 loader with another placement rule, 6 vector instructions per value;
//   * a tap's B operand is 32 CONSECUTIVE voxels of one sub-tile row: conflict-free ds_read_b128 under the same XOR swizzle, the image is
//     [hi piece | lo piece] per voxel, and the stride-1 kernel's THREE-partial-product schedule applies unchanged (41 MFMAs per output row
//     and chunk instead of 54, no rotates): every operand of its 41 groups is a per-lane byte address computed once per wave;
//   * the weights come through LDS (one buffer, dropped in between two barriers) in the stride-1 image.
// One 768-thread workgroup per CU: waves 0-3 consume (wave w = output row (z, y) = (w >> 1, w & 1) of the 2 x 2 x 32 tile x 64 output
// channels), waves 4-11 load (a stride-2 tile stages 12.7 voxels per output voxel: with four loader waves, two quads per thread and chunk, the
// loaders took 6.4 k cycles per chunk against the consumers' 3.3 k).  LDS: two activation buffers of 1625 voxels x 32 B + 55 KB of weight
// fragments = 159.3 KB.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include <utility>
#include "conv3d_common.h"

typedef short s2_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 s2_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 s2_f16x2 __attribute__((ext_vector_type(2)));
typedef float s2_f32x2 __attribute__((ext_vector_type(2)));
typedef float s2_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned s2_u32x4 __attribute__((ext_vector_type(4)));

// Tile geometries: four consumer waves, one M-tile (32 output voxels = LY rows of TX) x 64 output channels each.
//   configuration 1:  output rows of 32 voxels, tile 2 x 2 x 32 (z, y, x), input box 5 x 5 x 65
//   configuration 25: output rows of 16 voxels, tile 2 x 4 x 16, an M-tile = 2 rows,   input box 5 x 9 x 33   (round 5: the strided layer into the 16^3 level)
//   configuration 26: output rows of 8 voxels,  tile 4 x 4 x 8,  an M-tile = 4 rows,   input box 9 x 9 x 17   (... into the 8^3 level)
template <int TZ_, int TY_, int TX_>
struct S2Geo {
    static constexpr int TZ = TZ_, TY = TY_, TX = TX_, LX = TX_, LY = 32 / TX_, MY = TY_ / LY;
    typedef ConvCfg<2, TX_, LY, TZ_, TY_, 1, 1, 2, 2, 6> C;     // (= BCfg1 of conv3d_k3_bf16.hip for configuration 1)
    static_assert(LX * LY == 32 && TY % LY == 0 && TZ * MY == 4, "four M-tiles of 32 output voxels");
    // sub-tile s = pz 4 + py 2 + px holds the input voxels of parity (pz, py, px): even (p = 0): indices 0 .. T - 1 (taps k = 1); odd (p = 1):
    // indices 0 .. T (input o - 1 .. o + T - 1: taps k = 0 at index o, k = 2 at index o + 1)
    static constexpr int nz(int p) { return TZ + p; }
    static constexpr int ny(int p) { return TY + p; }
    static constexpr int nx(int p) { return TX + p; }
    static constexpr int base(int s) {
        int b = 0;
        for (int t = 0; t < s; ++t) b += nz((t >> 2) & 1) * ny((t >> 1) & 1) * nx(t & 1);
        return b;
    }
    static constexpr int BZ = 2 * TZ + 1, BY = 2 * TY + 1, BX = 2 * TX + 1;          // the input box
    static constexpr int VOX = base(8);
    static_assert(VOX == BZ * BY * BX, "input box");
    static constexpr int QPR = TX / 2 + 1, NQ = BZ * BY * QPR;                        // aligned quads per box row / per chunk
    static_assert(NQ <= 512, "one quad per loader thread");
    static constexpr int ACT_BYTES = VOX * 32;                                        // one activation buffer
    static constexpr int W_OFF = 2 * ACT_BYTES;
    static constexpr int WQ = 27 * 2 * 64;                                            // 16-byte weight pieces per chunk (NT = 2)
    static constexpr int LDS = W_OFF + WQ * 16;
    static_assert(LDS <= 160 * 1024, "LDS");
    // voxel index (without the lane's x) of tap (kd, kr, kw) for output row (tzi, tyi)
    static __device__ __forceinline__ int tap_row(int kd, int kr, int kw, int tzi, int tyi) {
        const int pz = kd == 1 ? 0 : 1, py = kr == 1 ? 0 : 1, px = kw == 1 ? 0 : 1;
        return base(pz * 4 + py * 2 + px) + ((tzi + (kd == 2 ? 1 : 0)) * ny(py) + (tyi + (kr == 2 ? 1 : 0))) * nx(px) + (kw == 2 ? 1 : 0);
    }
};
typedef S2Geo<2, 2, 32> S2Geo32;
typedef S2Geo<2, 4, 16> S2Geo16;
typedef S2Geo<4, 4, 8> S2Geo8;
static_assert(S2Geo32::VOX == 1625, "5 x 5 x 65 input box");

namespace {
// uniform read of a value no kernel in flight writes: a scalar-cache load (conv3d_k3_bf16_ws.hip::ld_uniform)
__device__ __forceinline__ float s2_ld_uniform(const float* p) {
    return *reinterpret_cast<const __attribute__((address_space(4))) float*>(reinterpret_cast<unsigned long long>(p));
}
// The low parts of two values as one dword: fp16(t0 - hi0) | fp16(t1 - hi1) << 16, hi = the two halves of h2.  v_fma_mixlo / mixhi_f16 take
// the fp16 operand as it is, subtract in fp32 and round ONCE to fp16: two instructions per pair where v_cvt_f32_f16 x 2 + v_sub_f32 x 2 +
// v_cvt_pk_f16_f32 are five -- the loaders are bound by their instruction count (two loader waves share a SIMD with a consumer).
__device__ __forceinline__ unsigned s2_lo_pair(unsigned h2, float t0, float t1) {
    unsigned d;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h2), "v"(t0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(d) : "v"(h2), "v"(t1));
    return d;
}
template <int... I, class F>
__device__ __forceinline__ void s2_static_for(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }

// ---- the parity-split tile: S2Geo above.  Tap k (0..2 along an axis) of output index o reads parity P(k) at sub-tile index o + SH(k) ----
constexpr int s2_shift(int k) { return k == 2 ? 1 : 0; }

// the stride-1 pair kernel's 41 groups (conv3d_k3_bf16_ws.hip::p3_grp): per (kd, kr) the three taps' [w_hi | w_hi] x [x_hi | x_lo] (pr 0..2),
// [w_lo(kw 0) | w_lo(kw 1)] x [x_hi(kw 0) | x_hi(kw 1)] (pr 3) and one more that pairs the remaining w_lo x_hi terms of kw = 2 across kr / kd
struct S2Grp { int kd, kr, pr; };
constexpr int S2_NG = 41;
constexpr S2Grp s2_grp(int gi) {
    int n = 0;
    for (int kd = 0; kd < 3; ++kd)
        for (int kr = 0; kr < 3; ++kr)
            for (int pr = 0; pr < 5; ++pr) {
                if (pr == 4 && (kr == 0 || (kr == 2 && kd == 0))) continue;
                if (n == gi) return S2Grp{kd, kr, pr};
                ++n;
            }
    return S2Grp{0, 0, 0};
}
}  // namespace

template <class GEO>
__global__ __launch_bounds__(768) void conv3d_k3s2_f16x2_ws_kernel(const ConvArgs a, const int T /*tiles x cout groups*/) {
    typedef typename GEO::C C;
    constexpr int NT = C::NT, MT = C::MT, LTH = 512;
    constexpr int S2_ACT_BYTES = GEO::ACT_BYTES, S2_W_OFF = GEO::W_OFF, S2_WQ = GEO::WQ;
    static_assert(MT == 1 && NT == 2 && C::WM == 4, "one output row per consumer wave, 64 output channels");
    extern __shared__ __attribute__((aligned(16))) uint4 ldsq[];   // [2][S2_VOX voxels][hi | lo piece, swapped where (ix >> 3) & 1], then [27][NT][64] weight pieces
    uint4* const wlds = ldsq + S2_W_OFF / 16;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int HW = a.H * a.W;
    const size_t DHW = (size_t)a.D * HW;
    const int nch = a.nchunks;

    auto tile_of = [&](int v, int& g, int& tx, int& ty, int& tz, int& n0) __attribute__((always_inline)) {
        const int q = T >> 3, r = T & 7, xcd = v & 7;
        int t = (a.xcd_swizzle & 1) ? (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3) : v;
        wmh_tile_coords(a, t, g, tx, ty, tz, n0);
    };
    int ntiles_mine = 0;
    for (int v = blockIdx.x; v < T; v += gridDim.x) ++ntiles_mine;
    const int total = ntiles_mine * nch;                        // chunk iterations of this workgroup

    if (wave >= 4) {
        // ================================================ loaders =================================================================
        __builtin_amdgcn_s_setprio(3);
        const int lt = tid - 256;
        // Staging unit = an aligned quad of 4 consecutive x of one row of the 5 x 5 x 65 box: the row [2 ox0 - 1, 2 ox0 + 63] lies in the
        // aligned span [2 ox0 - 4, 2 ox0 + 64), 17 quads; 425 per chunk, one per thread.
        constexpr int QPR = GEO::QPR, NQ = GEO::NQ;
        // this thread's quad: box row (bz, by) and quad q -> where its voxels go: LDS voxel index of sub-tile row (px = 0: odd x, px = 1: even x)
        const int row = lt / QPR, q = lt - row * QPR;
        const bool mine = lt < NQ;
        int rb0, rb1;
        {
            const int bz = (row / GEO::BY) % GEO::BZ, by = row % GEO::BY;
            const int pz = (bz & 1) ^ 1, py = (by & 1) ^ 1, iz = bz >> 1, iy = by >> 1;
            const int r0 = (iz * (GEO::TY + py) + iy);
            rb0 = (pz ? (py ? GEO::base(6) : GEO::base(4)) : (py ? GEO::base(2) : GEO::base(0))) + r0 * GEO::nx(0);
            rb1 = (pz ? (py ? GEO::base(7) : GEO::base(5)) : (py ? GEO::base(3) : GEO::base(1))) + r0 * GEO::nx(1);
        }
        int q_off = 0; unsigned q_ok = 0; int n0_cur = 0, g_cur = 0;
        auto setup = [&](int v) __attribute__((always_inline)) {
            int g, tx, ty, tz, n0;
            tile_of(v, g, tx, ty, tz, n0);
            n0_cur = __builtin_amdgcn_readfirstlane(n0); g_cur = __builtin_amdgcn_readfirstlane(g);
            tx = __builtin_amdgcn_readfirstlane(tx); ty = __builtin_amdgcn_readfirstlane(ty); tz = __builtin_amdgcn_readfirstlane(tz);
            const int gx = tx * (2 * GEO::TX) - 4 + 4 * q, gy = ty * (2 * GEO::TY) - 1 + row % GEO::BY, gz = tz * (2 * GEO::TZ) - 1 + (row / GEO::BY) % GEO::BZ;
            const bool ok = mine && gx >= 0 && gx < a.W && gy >= 0 && gy < a.H && gz >= 0 && gz < a.D;
            q_off = ok ? gz * HW + gy * a.W + gx : 0;
            q_ok = ok ? 1u : 0u;
        };
        constexpr int WIT = (S2_WQ + LTH - 1) / LTH;
        s2_u32x4 wst[WIT];                                         // the weight fragments of the chunk being converted (this thread's pieces)
        s2_f32x4 v[2][8]; unsigned okv[2] = {0, 0}; int set_n0[2] = {0, 0}, set_c[2] = {0, 0}, set_g[2] = {0, 0};
        int pk = 0, pc = 0, pq = 0;                                // (tile counter, chunk) of the NEXT issue; pq: its linear index (clamped)
        const unsigned pstride = 4u * (unsigned)DHW;
        auto issue = [&](auto sc) __attribute__((always_inline)) {
            constexpr int s_ = decltype(sc)::value;
            const int c0 = pc * 8, cmax = a.C0 - 1 - c0;          // channels past the end re-read the last one (scaled to zero)
            const char* base = reinterpret_cast<const char*>(a.x0 + ((size_t)n0_cur * a.C0 + c0) * DHW);
            const unsigned qo = 4u * (unsigned)q_off;
            // (lanes beyond the 425 quads re-read voxel 0: masking them off turns the loads into a divergent region and the compiler's counted
            //  waits for the ring into vmcnt(0); measured no faster either)
#pragma unroll
            for (int c = 0; c < 8; ++c) v[s_][c] = *reinterpret_cast<const s2_f32x4*>(base + (qo + (unsigned)min(c, cmax) * pstride));
            okv[s_] = q_ok; set_n0[s_] = n0_cur; set_c[s_] = pc; set_g[s_] = g_cur;
            if (pq + 1 < total) {                                  // past the last chunk the cursor stays: branch-free loads, exact vmcnt
                ++pq;
                if (++pc == nch) { pc = 0; ++pk; setup(blockIdx.x + pk * gridDim.x); }
            }
        };
        auto w_issue = [&](auto sc) __attribute__((always_inline)) {   // the weight fragments of the chunk in set s
            constexpr int s_ = decltype(sc)::value;
            const int g = __builtin_amdgcn_readfirstlane(set_g[s_]), c = __builtin_amdgcn_readfirstlane(set_c[s_]);
            const s2_u32x4* wsrc = reinterpret_cast<const s2_u32x4*>(a.wf) + ((size_t)g * nch + c) * S2_WQ;
#pragma unroll
            for (int i = 0; i < WIT; ++i) wst[i] = wsrc[min(lt + i * LTH, S2_WQ - 1)];
        };
        auto write_w = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < WIT; ++i) { const int k = lt + i * LTH; if (WIT * LTH == S2_WQ || k < S2_WQ) reinterpret_cast<s2_u32x4*>(wlds)[k] = wst[i]; }
        };
        float scv[8], shv[8], slope_eff = 1.f;
        static __device__ const float ident2[2] = {1.f, 0.f};
        auto load_ss = [&](auto sc) __attribute__((always_inline)) {
            constexpr int s_ = decltype(sc)::value;
            const bool has_ss = a.ss0 != nullptr;
            slope_eff = has_ss ? a.slope : 1.f;
            const int c0 = __builtin_amdgcn_readfirstlane(set_c[s_]) * 8, sn0 = __builtin_amdgcn_readfirstlane(set_n0[s_]);
            const float* tab = has_ss ? a.ss0 + ((size_t)sn0 * a.C0 + c0) * 2 : ident2;
            const int mul = has_ss ? 2 : 0, cmax = a.C0 - 1 - c0;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float sc1 = s2_ld_uniform(tab + mul * min(c, cmax)), sh1 = s2_ld_uniform(tab + mul * min(c, cmax) + 1);
                scv[c] = c <= cmax ? sc1 : 0.f;
                shv[c] = c <= cmax ? sh1 : 0.f;
            }
        };
        auto convert = [&](auto sc, int buf) __attribute__((always_inline)) {
            constexpr int s_ = decltype(sc)::value;
            uint4* dst = ldsq + (size_t)buf * (S2_ACT_BYTES / 16);
            const unsigned keep = okv[s_] ? 0xFFFFFFFFu : 0u;     // quad outside the volume: zeros
            // (interior tiles -- nearly all -- have every quad of the wave inside: the masks are skipped on a wave-uniform branch)
            const bool all_in = __builtin_amdgcn_readfirstlane((int)(__ballot(okv[s_] != 0u || !mine) == ~0ull)) != 0;
            unsigned w[4][4], wlo[4][4];                           // [voxel of the quad][channel pair]: high / low parts
#pragma unroll
            for (int c = 0; c < 8; c += 2) {
                s2_f32x4 t[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float sc1 = scv[c + e], sh1 = shv[c + e];
                    t[e] = s2_f32x4{wmh_norm_lrelu(v[s_][c + e][0], sc1, sh1, slope_eff), wmh_norm_lrelu(v[s_][c + e][1], sc1, sh1, slope_eff),
                                    wmh_norm_lrelu(v[s_][c + e][2], sc1, sh1, slope_eff), wmh_norm_lrelu(v[s_][c + e][3], sc1, sh1, slope_eff)};
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {                      // fp16 pair: high part toward zero, the remainder rounded once to nearest
                    const unsigned h2 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(t[0][k], t[1][k]));
                    w[k][c >> 1] = h2;
                    wlo[k][c >> 1] = s2_lo_pair(h2, t[0][k], t[1][k]);
                }
            }
            if (!all_in) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int e = 0; e < 4; ++e) { w[k][e] &= keep; wlo[k][e] &= keep; }
            }
            // voxel k of the quad sits at box x = 4 q - 3 + k: k = 0, 2 odd (px = 0, sub-row index 2 q - 2 | 2 q - 1), k = 1, 3 even (px = 1, 2 q - 1 | 2 q)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int x = 4 * q - 3 + k, ix = 2 * q - 2 + ((k + 1) >> 1);
                if (mine && x >= 0 && x < GEO::BX) {
                    const int e = ((k & 1) ? rb1 : rb0) + ix;
                    dst[e * 2 + (0 ^ ((ix >> 3) & 1))] = make_uint4(w[k][0], w[k][1], w[k][2], w[k][3]);
                    dst[e * 2 + (1 ^ ((ix >> 3) & 1))] = make_uint4(wlo[k][0], wlo[k][1], wlo[k][2], wlo[k][3]);
                }
            }
        };
        constexpr auto S0 = std::integral_constant<int, 0>{};
        constexpr auto S1 = std::integral_constant<int, 1>{};
        unsigned long long tw = 0, tb = 0, t0 = 0;
        if (total > 0) {
            setup(blockIdx.x);
            issue(S0);                                             // chunk 0 (exposed once per workgroup)
            w_issue(S0);
            issue(S1);                                             // chunk 1 in flight across the barrier
            load_ss(S0);
            convert(S0, 0);
            write_w();
        }
        __syncthreads();
        // iteration it: the consumers are on chunk it (buffer it & 1).  The weights of chunk it + 1 first (from L2), then the activation loads of
        // chunk it + 2 into the register set chunk it came from, then chunk it + 1 (in flight since the previous iteration) is normalised,
        // split and written into the other buffer; its weights drop into the single weight buffer between the two barriers.
        auto iteration = [&](auto s_conv, auto s_issue, int buf) __attribute__((always_inline)) {
            if (WMH_STAMPS(a)) t0 = __builtin_amdgcn_s_memtime();
            w_issue(s_conv);
            load_ss(s_conv);
            issue(s_issue);
            convert(s_conv, buf);
            if (WMH_STAMPS(a)) { const unsigned long long t = __builtin_amdgcn_s_memtime(); tw += t - t0; t0 = t; }
            __syncthreads();                                       // consumers are done with chunk `it`: its weights may go
            write_w();
            __syncthreads();
            if (WMH_STAMPS(a)) tb += __builtin_amdgcn_s_memtime() - t0;
        };
        for (int it = 0; it < total; it += 2) {
            iteration(S1, S0, 1);
            if (it + 1 >= total) break;
            iteration(S0, S1, 0);
        }
        if (WMH_STAMPS(a) && tid == 256) {
            unsigned long long* o = WMH_STAMPS(a) + (size_t)blockIdx.x * 8;
            o[4] = tw; o[5] = tb;
        }
        if (WMH_STAMPS(a) && lane == 0) {                          // every loader wave's work: record block (wave - 3), entry 6
            unsigned long long* o = WMH_STAMPS(a) + ((size_t)(wave - 3) * gridDim.x + blockIdx.x) * 8;
            o[6] = tw;
        }
        return;
    }

    // ================================================== consumers =================================================================
    const int j = lane & 31, kh = lane >> 5;
    const int dx = j % GEO::LX, dy = j / GEO::LX;                    // the lane's output voxel inside its M-tile: LY rows of LX
    const int tzi = wave / GEO::MY, tyi = (wave % GEO::MY) * GEO::LY + dy;
    // the 41 B operands of a chunk: byte addresses inside an activation buffer, computed once (they do not depend on the tile)
    int bG[S2_NG];
    s2_static_for(std::make_integer_sequence<int, S2_NG>{}, [&](auto gc) __attribute__((always_inline)) {
        constexpr int gi = decltype(gc)::value;
        constexpr S2Grp G = s2_grp(gi);
        auto full = [&](int kd, int kr, int kw) { const int ix = dx + s2_shift(kw); return (GEO::tap_row(kd, kr, kw, tzi, tyi) + dx) * 32 + 16 * (kh ^ ((ix >> 3) & 1)); };
        auto hi = [&](int kd, int kr, int kw) { const int ix = dx + s2_shift(kw); return (GEO::tap_row(kd, kr, kw, tzi, tyi) + dx) * 32 + 16 * ((ix >> 3) & 1); };
        if constexpr (G.pr < 3) bG[gi] = full(G.kd, G.kr, G.pr);                                          // [x_hi | x_lo](tap kw)
        else if constexpr (G.pr == 3) bG[gi] = kh ? hi(G.kd, G.kr, 1) : hi(G.kd, G.kr, 0);                // [x_hi(kw 0) | x_hi(kw 1)]
        else if constexpr (G.kr == 1) bG[gi] = kh ? hi(G.kd, 1, 2) : hi(G.kd, 0, 2);                      // [x_hi(kr 0, kw 2) | x_hi(kr 1, kw 2)]
        else if constexpr (G.kd == 1) bG[gi] = kh ? hi(1, 2, 2) : hi(0, 2, 2);                            // [x_hi(kd 0, kr 2, kw 2) | x_hi(kd 1, kr 2, kw 2)]
        else bG[gi] = full(2, 2, 2);                                                                       // [x_hi | x_lo](2, 2, 2) once more (A's upper half is zero)
    });
    const char* const ldsb = reinterpret_cast<const char*>(ldsq);
    // weight pieces: a tap's fragment = [hi of cout 0..31 | lo of cout 0..31] per nt; both lane halves read piece j
    const s2_h8* wl = reinterpret_cast<const s2_h8*>(wlds) + j;
    const s2_h8* wlp = wl + 32 + kh * (NT * 64);                  // lo piece of tap + kh
    const s2_h8* wlq = wl + 32 + kh * (3 * NT * 64);              // lo piece of tap + 3 kh (next kr)
    const s2_h8* wlr = wl + 32 + kh * (9 * NT * 64);              // lo piece of tap + 9 kh (next kd)
    const s2_h8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tm = 0, tb = 0, te = 0, t0 = 0;
    __syncthreads();                                             // chunk 0 is staged
    int it = 0;
    for (int k = 0; k < ntiles_mine; ++k) {
        int g, tx, ty, tz, n0;
        tile_of(blockIdx.x + k * gridDim.x, g, tx, ty, tz, n0);
        f32x16 acc[NT][MT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = (g * NT + nt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                acc[nt][0][r] = (a.bias != nullptr && co < a.Cout) ? a.bias[co] * WMH_PAIR_WSCALE : 0.f;
            }
        for (int ch = 0; ch < nch; ++ch, ++it) {
            if (WMH_STAMPS(a)) t0 = __builtin_amdgcn_s_memtime();
            const char* cur = ldsb + (size_t)(it & 1) * S2_ACT_BYTES;
            constexpr int RING = 5;
            s2_h8 ring[RING], aF[2][NT];
            auto loadA = [&](auto gc, s2_h8 (&dst)[NT]) __attribute__((always_inline)) {
                constexpr S2Grp G = s2_grp(decltype(gc)::value);
                constexpr int t0_ = G.kd * 9 + G.kr * 3;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    if constexpr (G.pr < 3) dst[nt] = wl[((t0_ + G.pr) * NT + nt) * 64];
                    else if constexpr (G.pr == 3) dst[nt] = wlp[(t0_ * NT + nt) * 64];
                    else if constexpr (G.kr == 1) dst[nt] = wlq[((t0_ - 3 + 2) * NT + nt) * 64];          // [w_lo(kr 0, kw 2) | w_lo(kr 1, kw 2)]
                    else if constexpr (G.kd == 1) dst[nt] = wlr[((6 + 2) * NT + nt) * 64];               // [w_lo(kd 0, kr 2, kw 2) | w_lo(kd 1, kr 2, kw 2)]
                    else dst[nt] = wl[((t0_ + 2) * NT + nt) * 64 + 32];                                   // w_lo(2, 2, 2): the upper half is zeroed at use
                }
            };
            loadA(std::integral_constant<int, 0>{}, aF[0]);
            s2_static_for(std::make_integer_sequence<int, RING>{}, [&](auto ic) __attribute__((always_inline)) {
                ring[decltype(ic)::value] = *reinterpret_cast<const s2_h8*>(cur + bG[decltype(ic)::value]);
            });
            __builtin_amdgcn_sched_barrier(0);
            s2_static_for(std::make_integer_sequence<int, S2_NG>{}, [&](auto sc) __attribute__((always_inline)) {
                constexpr int gi = decltype(sc)::value;
                constexpr S2Grp G = s2_grp(gi);
                if constexpr (gi + 1 < S2_NG) loadA(std::integral_constant<int, gi + 1>{}, aF[(gi + 1) & 1]);
                if constexpr (G.pr == 4 && G.kr == 2 && G.kd == 2) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) aF[gi & 1][nt] = kh ? zero8 : aF[gi & 1][nt];              // A = [w_lo | 0]
                }
                const s2_h8 bv = ring[gi % RING];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[nt][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(s2_f16x8, aF[gi & 1][nt]), __builtin_bit_cast(s2_f16x8, bv), acc[nt][0], 0, 0, 0);
                if constexpr (gi + RING < S2_NG) ring[gi % RING] = *reinterpret_cast<const s2_h8*>(cur + bG[gi + RING]);
                __builtin_amdgcn_sched_barrier(0);
            });
            if (WMH_STAMPS(a)) { const unsigned long long t = __builtin_amdgcn_s_memtime(); tm += t - t0; t0 = t; }
            __syncthreads();
            __syncthreads();                                       // the loaders swap the weight fragments in between
            if (WMH_STAMPS(a)) tb += __builtin_amdgcn_s_memtime() - t0;
        }
        if (WMH_STAMPS(a)) t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt][0] *= (1.f / WMH_PAIR_WSCALE);      // undo the weights' 2^8 (exact)
        // (the shared epilogue: 9 k cycles per tile, 6 k of them the moments' recursive halving, which costs the same for one row per wave as for
        //  eight.  A row-by-row version with DPP adds instead of LDS permutes took 15 k: dependent DPP steps wait out their hazards.)
        conv_epilogue<C>(a, acc, g, 0, n0, wave, j, kh, dx, dy, 0, wave, tz * C::TZ, ty * C::TY, tx * C::TX, (tz * a.nty + ty) * a.ntx + tx);
        if (WMH_STAMPS(a)) te += __builtin_amdgcn_s_memtime() - t0;
    }
    if (WMH_STAMPS(a) && tid == 0) {
        unsigned long long* o = WMH_STAMPS(a) + (size_t)blockIdx.x * 8;
        o[0] = tm; o[1] = tb; o[2] = te; o[3] = (unsigned long long)total;
    }
    if (WMH_STAMPS(a) && lane == 0) {                              // every consumer wave's MFMA phases: record block (9 + wave), entry 6
        unsigned long long* o = WMH_STAMPS(a) + ((size_t)(9 + wave) * gridDim.x + blockIdx.x) * 8;
        o[6] = tm; o[7] = te;
    }
}

// a: fully prepared by wmh_conv3d_k3_fwd_f16x2 (configurations 1, 25, 26: one source, no split-K; W % 4 == 0 and a 16-byte aligned source: checked there)
template <class G>
static int launch_s2(const ConvArgs& a_in, hipStream_t st) {
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
        if (ncu < 8) ncu = 8;
        ncu &= ~7;
    }
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_k3s2_f16x2_ws_kernel<G>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
    const long long T = (long long)a_in.ntx * a_in.nty * a_in.ntz * a_in.N * a_in.G;
    if (T > 0x7fffffffLL) return wmh_fail(WMH_EINVAL, "conv3d_k3 stride 2 f16x2 (wave-specialised): too many tiles");
    if ((size_t)8 * a_in.D * a_in.H * a_in.W * 4 >= ((size_t)1 << 32))
        return wmh_fail(WMH_EINVAL, "conv3d_k3 stride 2 f16x2 (wave-specialised): 8 source channel planes must stay below 4 GB (32-bit lane offsets)");
    const int grid = T < ncu ? (int)T : ncu;
    ConvArgs a = a_in;
    static const int supertile = wmh_diag_env("WMH_SUPERTILE", 1);
    if (supertile && (a.xcd_swizzle & 1)) wmh_pick_supertile(a, G::TY, G::TZ, grid / 8 > 0 ? grid / 8 : 1);
    hipLaunchKernelGGL(conv3d_k3s2_f16x2_ws_kernel<G>, dim3(grid), dim3(768), G::LDS, st, a, (int)T);
    return wmh_check_launch("conv3d_k3s2_f16x2_ws_kernel");
}
int wmh_launch_conv_s2_f16x2_ws(int cfg, const ConvArgs& a, hipStream_t st) {
    if (cfg == 1) return launch_s2<S2Geo32>(a, st);
    if (cfg == 25) return launch_s2<S2Geo16>(a, st);
    if (cfg == 26) return launch_s2<S2Geo8>(a, st);
    return wmh_fail(WMH_EINVAL, "conv3d_k3 stride 2 f16x2 (wave-specialised): configurations 1, 25 and 26");
}
