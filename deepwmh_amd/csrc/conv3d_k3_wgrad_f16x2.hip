// conv3d_k3_wgrad_f16x2.hip -- weight gradient of the 3x3x3 convolution (pad 1; stride 1, and since round 5 stride 2: WpGeo below) in the
// f16x2 "pair" mode: fp32-grade results from the 16-bit matrix instruction (conv3d_common.h, DESIGN.md sections 3.2, 3.6).
//
//   dW[co][ci][tap] = sum_{n, o} dy[n, co, o] * x'[n, ci, S o + tap - 1]          (S = the stride; the text below describes S = 1)
//
// Voxels are the K dimension of v_mfma_f32_32x32x16_f16, one D tile [co 32][tap 32 (27 used)] per input channel.  Both operands are
// split into two fp16 numbers (hi rounded toward zero, lo = the remainder rounded to nearest: 21-22 significant bits) and kept in LDS
// as ONE DWORD PER VOXEL, (hi | lo << 16).  The 16 k-slots of an instruction carry 8 voxels x (hi, lo) of x' on the B side against
// 8 voxels x (hi, hi) of dy on the A side (hi hi + hi lo), and every third instruction 16 voxels' dy_lo against their x_hi (the operands
// repacked in registers, one v_perm_b32 per dword): three of the four partial products -- dy_lo x_lo, 2^-22 of a product, is not
// formed, as in the forward kernel -- from 3 instructions per 16 voxels and channel (round 4; all four took 4).  What makes the layout
// work where round 2's attempt (three shifted 2-byte copies of every halo row, 2-byte LDS stores) did not: with a dword per voxel a
// tap's shift along x is a whole number of dwords, so a lane's B operand -- 4 consecutive voxels starting at its tap's kx -- is two
// ds_read2_b32 from ONE copy of the halo tile, written with plain 4-byte stores.
//
// Wave-specialised workgroup of 8 waves covering 32 co x 16 ci: consumer wave w (0..3) owns ci = 4w..4w+3 -- four accumulators
// that SHARE every dy fragment (10 LDS operand reads per 12 MFMAs; one channel at a time would need 16 and is LDS-bound) -- loader waves
// 4..7 stage.  The unit of hand-over is one z slice of a TX x TY x TZ tile (TY * TX = 128 voxels, 8 step pairs of 12 MFMAs per consumer):
// dy slices are double-buffered per unit, the 16 channels' halo tiles per tile; while the consumers work on unit u the loaders convert
// and write what they fetched during unit u - 1 (dy of unit u + 1, half of the next tile's channels) and issue the loads for the unit
// after; one workgroup barrier per unit.  A workgroup walks tiles z-fastest and the workgroups of one XCD hold neighbouring (x, y)
// tiles, so the two z planes a tile shares with its predecessor and the in-plane halos come from that XCD's L2.
// Partial dW per workgroup goes to a slab; a fixed-order reduction sums the slabs (bit-reproducible, no atomics).
// Stands in for the weight half of torch.nn.Conv3d's backward [nnU-Net training, not in tree].
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "wmh_common.h"

typedef float wp_f32x16 __attribute__((ext_vector_type(16)));
typedef float wp_f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 wp_f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 wp_f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned wp_u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) char* wp_lds_cp;

struct WgradPArgs {
    const float* x0; const float* ss0; int C0;
    const float* x1; const float* ss1; int C1;
    const float* dy; float* part;
    int N, D, H, W, Cout, Cin;          // (D, H, W): the INPUT extent
    int Do, Ho, Wo;                      // the output extent (= the input's for stride 1)
    int ntx, nty, ntz, ntiles;
    int NG, NCG, P;
    int order, m, q, nk;            // tile order (the loaders' cursor); nk: tiles per workgroup (orders 1, 2)
    float slope;
    unsigned long long* stamps;     // diagnostics only (NULL in production)
};

// v -> (hi | lo << 16): hi = v cut to fp16's 11 significant bits (toward zero; a mask -- exact in fp16's normal range), lo = the
// remainder rounded to nearest by v_cvt_pk_f16_f32.  Below fp16's normal range (|v| < 6.1e-5) the pair is off by at most half a
// subnormal step (3e-8 absolute); |v| >= 65520 gives a non-finite high part (loud).  Three instructions, none of them packed fp32
// (those do not overlap with the consumers' MFMAs: conv3d_common.h::wmh_norm_lrelu, profiles/r03_coissue.txt).
__device__ __forceinline__ unsigned wp_split(float v) {
    const float hi = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) & 0xFFFFE000u);
    const float lo = v - hi;
    return __builtin_bit_cast(unsigned, __builtin_convertvector(wp_f32x2{hi, lo}, wp_f16x2));
}
// buffer_load_dword: 128-bit scalar descriptor (base, size in bytes) + scalar offset + 32-bit lane offset
typedef __amdgpu_buffer_rsrc_t wp_rsrc;
__device__ __forceinline__ wp_rsrc wp_make_rsrc(const float* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float wp_load(wp_rsrc rs, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ float wp_norm_lrelu(float x, float sc, float sh, float slope) {
    // (plain scalar code: inline asm would pin every value's chain in program order -- no interleaving of the independent chains --
    //  and the compiler forms no packed fp32 instructions from it: checked in the ISA)
    const float z = __builtin_fmaf(x, sc, sh);
    return __builtin_fmaxf(z, z * slope);
}

// S: the convolution's stride (1 or 2).  Stride 2 (round 5: the first convolution of the 64^3 / 32^3 / 16^3 stages): tap k of output o reads
// input 2 o + k - 1, so a halo ROW is kept parity-split -- [odd inputs -1, 1, ... | even inputs 0, 2, ...] -- and a lane's B operand (4
// consecutive OUTPUT voxels of its tap) is again 4 consecutive dwords: the odd part at index o (k = 0) or o + 1 (k = 2), the even part at index
// o (k = 1); rows and planes are addressed with twice the step.  The halo of an output tile is 8 x as many voxels per output as at stride 1, so
// the tile is 16 x 2 x 2 outputs (33 x 5 x 5 inputs: the same 13 elements per loader lane and channel as the stride-1 tiles).
template <int S, int TX, int TY, int TZ, int NLW>
struct WpGeo {
    static constexpr int IX = S * (TX - 1) + 3, IY = S * (TY - 1) + 3, IZ = S * (TZ - 1) + 3, ZS = IY * IX, SPX = IZ * ZS;
    static constexpr int NK = (SPX + 63) / 64;                 // halo elements per loader lane and channel
    // x' in LDS: [channel][IZ planes][IY rows][IX voxels] dwords with the row stride RS and the plane stride PS padded so that the
    // 27 taps of a B operand read -- dword PS/4 kz + RS/4 ky + kx of the tile, one 32-lane group of a ds_read2_b32 (banks = dword
    // index mod 32) -- fall on 27 different banks; unpadded (RS = 4 IX, PS = IY RS) every such read is a 2-way conflict and the LDS,
    // not the matrix pipe, sets the pace.  The last dword of a channel takes the stores of the lanes beyond the tile.
    // Stride 2: XE = where the even part of a row starts; (RS, PS, XE) = (39, 203, 20) dwords give the 27 taps 27 different banks too.
    // Rows of 8 voxels (stride 1; round 5: the 8^3 level): (RS, PS) = (11, 131) dwords.
    static constexpr int RS = S == 2 ? 156 : (TX == 32 ? 140 : TX == 16 ? 72 : 44), PS = S == 2 ? 812 : (TX == 32 ? 852 : TX == 16 ? 724 : 524), XE = S == 2 ? 80 : 0;
    static constexpr int CS = (IZ * PS + 16 + 15) / 16 * 16;   // bytes per channel
    static constexpr int XB = 16 * CS;                         // one x' buffer: the workgroup's 16 channels
    static constexpr int UV = TY * TX;                         // voxels per unit (one z slice of the tile)
    static constexpr int DYSB = UV * 4 + 16, DYB = 32 * DYSB;  // bytes: dy row (one output channel), dy buffer
    static constexpr int LT = NLW * 64;                        // loader threads (NLW loader waves after the 4 consumer waves)
    static constexpr int NDY = 32 * UV / LT, CPK = LT / UV;    // dy values per loader lane and unit / co rows per pass
    static constexpr int LCH = 16 / NLW, XPU = LCH / TZ;       // x' channels per loader wave: of the group / staged per unit
    static constexpr int LDS = 2 * XB + 2 * DYB;
    // byte position of halo x index xi inside a row / of tap kx's first voxel
    static constexpr __host__ __device__ int xpos(int xi) { return S == 2 ? ((xi & 1) ? XE + (xi >> 1) * 4 : (xi >> 1) * 4) : xi * 4; }
    static constexpr __host__ __device__ int xtap(int kx) { return S == 2 ? (kx == 1 ? XE : kx == 2 ? 4 : 0) : kx * 4; }
    static_assert(S == 1 || S == 2, "stride");
    static_assert(TZ == 2 && (UV % 64 == 0 || 64 % UV == 0) && LT % UV == 0 && TX % 8 == 0 && XPU >= 1 && NDY >= 1 && (NLW == 4 || NLW == 8), "tile geometry");
    static_assert((TY - 1) * S * RS + (TX - 8) * 4 + 12 <= 1020, "ds_read2_b32 offset range");
    static_assert(RS >= (S == 2 ? XE + TX * 4 : IX * 4) && (S == 1 || XE >= (TX + 1) * 4) && PS >= IY * RS && (TX == 32 || TX == 16 || (TX == 8 && S == 1)) && CS % 16 == 0 && CS >= IZ * PS + 16, "x' layout");
    static_assert(LDS <= 160 * 1024, "LDS");
};

template <int S, int TX, int TY, int TZ, int NLW>
__global__ __launch_bounds__(256 + 64 * NLW) void conv3d_k3_wgrad_f16x2_kernel(const WgradPArgs a) {
    typedef WpGeo<S, TX, TY, TZ, NLW> G;
    constexpr int IX = G::IX, IY = G::IY, ZS = G::ZS, SPX = G::SPX, NK = G::NK, RS = G::RS, PS = G::PS, CS = G::CS, XB = G::XB, UV = G::UV;
    constexpr int DYSB = G::DYSB, DYB = G::DYB, NDY = G::NDY, CPK = G::CPK, XPU = G::XPU, LCH = G::LCH;
    extern __shared__ __attribute__((aligned(16))) char ldsb[];
    char* const xbuf = ldsb;                                   // [2][16 ci][CS]
    char* const dyl = ldsb + 2 * XB;                           // [2][32 co][DYSB]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = blockIdx.x;
    const int p = bid % a.P; bid /= a.P;
    const int g = bid % a.NG, cg = bid / a.NG;
    // workgroups p, p + 8, ... run on one XCD: give them neighbouring places in the step
    const int pp = (a.P % 8 == 0) ? (p % 8) * (a.P / 8) + p / 8 : p;
    const int nk = a.order == 0 ? (pp < a.ntiles ? (a.ntiles - pp + a.P - 1) / a.P : 0) : a.nk;
    const int HW = a.H * a.W, HWo = a.Ho * a.Wo;
    const size_t DHW = (size_t)a.D * HW, DHWo = (size_t)a.Do * HWo;

    unsigned long long st_work = 0, st_wait = 0, st_t = 0;
    const bool stamping = WMH_STAMPS(a) != nullptr;

    if (wave < 4) {
        // ---------------------------------------------------------------- consumers: MFMAs only
        const int i = lane & 31, kh = lane >> 5, tapj = min(i, 26);
        const int LB = (tapj / 9) * PS + ((tapj / 3) % 3) * RS + G::xtap(tapj % 3) + 16 * kh;
        const int LA = i * DYSB + kh * 16;
        const int ci0 = cg * 16 + 4 * wave;
        const bool any = ci0 < a.Cin;                          // uniform: this wave holds input channels (absent ones are staged as zeros)
        wp_f32x16 acc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
        __syncthreads();                                       // unit 0 and tile 0 are staged
        if (stamping) st_t = __builtin_amdgcn_s_memtime();
        for (int k = 0; k < nk; ++k) {
            const char* xb = xbuf + (k & 1) * XB + 4 * wave * CS + LB;
#pragma unroll
            for (int z = 0; z < TZ; ++z) {
                if (any) {
                    const char* ap = dyl + (z & 1) * DYB + LA;
                    // one base register per channel, opaque to the compiler (it would otherwise fold them into one base + an add per
                    // read: ds_read2_b32 immediates reach 1020 bytes, a channel is 3 KB): every operand read is base + immediate
                    wp_lds_cp bc[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        unsigned o = (unsigned)(uintptr_t)(wp_lds_cp)(xb + c * CS + z * S * PS);
                        asm("" : "+v"(o));
                        bc[c] = (wp_lds_cp)(uintptr_t)o;
                    }
                    // Three of the four partial products (dy_lo x_lo, 2^-22 of a product, is not formed -- as in the forward kernel),
                    // two steps of 8 voxels at a time.  With a = the dy dwords and b = the x' dwords of the pair's first and second step:
                    //      (dy_hi | dy_hi)(a0) . (x_hi | x_lo)(b0)      hi hi + hi lo of step 0
                    //      (dy_hi | dy_hi)(a1) . (x_hi | x_lo)(b1)      the same of step 1
                    //      (dy_lo(a0) | dy_lo(a1)) . (x_hi(b0) | x_hi(b1))   lo hi of both steps
                    // 3 instructions per 16 voxels and channel instead of 4; the repacked operands are one v_perm_b32 per dword
                    // (12 per pair on the A side, shared by the four channels; 4 per channel on the B side), which fit in the
                    // MFMAs' issue gaps.  A channel's three MFMAs run back to back on its accumulator (no interleaving needed for
                    // this instruction) and its operands of the NEXT pair are read right behind them, nine MFMAs before their use.
                    constexpr int NST = TY * (TX / 8), NP = NST / 2;
                    static_assert(NST % 2 == 0, "steps are taken in pairs");
                    wp_u32x4 a4[2][2], b[2][4];
                    auto fetch_a = [&](int st, int set) __attribute__((always_inline)) {
                        a4[set][st & 1] = *reinterpret_cast<const wp_u32x4*>(ap + ((st / (TX / 8)) * TX + 8 * (st % (TX / 8))) * 4);
                    };
                    auto fetch_b = [&](int st, int c) __attribute__((always_inline)) {
                        const auto* q = reinterpret_cast<const __attribute__((address_space(3))) unsigned*>(bc[c] + (st / (TX / 8)) * S * RS + 32 * (st % (TX / 8)));
                        b[st & 1][c] = wp_u32x4{q[0], q[1], q[2], q[3]};               // dword alignment: two ds_read2_b32
                    };
                    fetch_a(0, 0); fetch_a(1, 0);
#pragma unroll
                    for (int c = 0; c < 4; ++c) { fetch_b(0, c); fetch_b(1, c); }
#pragma unroll
                    for (int j = 0; j < NP; ++j) {
                        const int cur = j & 1, nxt = cur ^ 1;
                        wp_u32x4 ah0, ah1, al;
#pragma unroll
                        for (int d = 0; d < 4; ++d) {
                            ah0[d] = __builtin_amdgcn_perm(a4[cur][0][d], a4[cur][0][d], 0x01000100u);     // (hi | hi)
                            ah1[d] = __builtin_amdgcn_perm(a4[cur][1][d], a4[cur][1][d], 0x01000100u);
                            al[d] = __builtin_amdgcn_perm(a4[cur][1][d], a4[cur][0][d], 0x07060302u);      // (lo of step 0 | lo of step 1)
                        }
                        if (j + 1 < NP) { fetch_a(2 * j + 2, nxt); fetch_a(2 * j + 3, nxt); }
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            wp_u32x4 bh;
#pragma unroll
                            for (int d = 0; d < 4; ++d) bh[d] = __builtin_amdgcn_perm(b[1][c][d], b[0][c][d], 0x05040100u);   // (x_hi of step 0 | of step 1)
                            __builtin_amdgcn_sched_barrier(0);
                            acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(wp_f16x8, ah0), __builtin_bit_cast(wp_f16x8, b[0][c]), acc[c], 0, 0, 0);
                            acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(wp_f16x8, ah1), __builtin_bit_cast(wp_f16x8, b[1][c]), acc[c], 0, 0, 0);
                            acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(wp_f16x8, al), __builtin_bit_cast(wp_f16x8, bh), acc[c], 0, 0, 0);
                            __builtin_amdgcn_sched_barrier(0);
                            if (j + 1 < NP) { fetch_b(2 * j + 2, c); fetch_b(2 * j + 3, c); }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
                if (stamping) { const unsigned long long t = __builtin_amdgcn_s_memtime(); st_work += t - st_t; st_t = t; }
                __syncthreads();                               // this unit is consumed, the next one is staged
                if (stamping) { const unsigned long long t = __builtin_amdgcn_s_memtime(); st_wait += t - st_t; st_t = t; }
            }
        }
        if (i < 27) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int ci = ci0 + c;
                if (ci >= a.Cin) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = g * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    if (co < a.Cout) a.part[(((size_t)p * a.Cout + co) * a.Cin + ci) * 27 + i] = acc[c][r];
                }
            }
        }
        if (stamping && lane == 0 && wave == 0) { WMH_STAMPS(a)[blockIdx.x * 4 + 0] = st_work; WMH_STAMPS(a)[blockIdx.x * 4 + 1] = st_wait; }
        return;
    }

    // -------------------------------------------------------------------- loaders: fetch, InstanceNorm + LeakyReLU, split, stage
    // (default wave priority: raising the loaders', as the forward kernels do, costs 3 % here -- the consumers are the slower side)
    const int lw = wave - 4, ltid = tid - 256;
    const int dv = ltid % UV;
    constexpr bool DUNI = UV % 64 == 0;                       // a wave's lanes share their first dy row (UV is a multiple of 64)
    const int dcol0 = DUNI ? __builtin_amdgcn_readfirstlane(ltid / UV) : ltid / UV;
    const int dvx = dv % TX, dvy = dv / TX;
    const unsigned cstride = 4u * (unsigned)DHW;              // bytes between channels of x' (host checks the range)
    const unsigned dcstride = 4u * (unsigned)DHWo;            // ... of dy

    // Tile cursor: the workgroup's tiles in sequence -> sample and output origin (all scalar).  order 0: tiles pp, pp + P, ... in
    // x -> y -> z order (divisions).  orders 1 / 2: a step of the P workgroups covers (a part of | several) planes of tiles and the
    // steps walk z first: tz counts up, (tx, ty) change only when it wraps.
    int c_k = 0, c_tz = 0, c_part = 0, c_outer = 0;           // next tile
    int c_n = 0, c_ox0 = 0, c_oy0 = 0, c_oz0 = 0;             // the tile `advance` delivered last
    const int planeA = a.ntx * a.nty, c_e = pp / planeA;
    int c_tx = 0, c_ty = 0;
    if (a.order != 0) { const int idx = pp - c_e * planeA; c_tx = idx % a.ntx; c_ty = idx / a.ntx; }
    auto advance = [&]() __attribute__((always_inline)) {
        if (a.order == 0) {
            int b = pp + c_k * a.P;
            const int tx = b % a.ntx; b /= a.ntx;
            const int ty = b % a.nty; b /= a.nty;
            c_n = __builtin_amdgcn_readfirstlane(b / a.ntz);
            c_ox0 = __builtin_amdgcn_readfirstlane(tx * TX); c_oy0 = __builtin_amdgcn_readfirstlane(ty * TY);
            c_oz0 = __builtin_amdgcn_readfirstlane((b % a.ntz) * TZ);
        } else {
            c_n = c_outer * a.q + c_e; c_ox0 = c_tx * TX; c_oy0 = c_ty * TY; c_oz0 = c_tz * TZ;
            if (++c_tz == a.ntz) {
                c_tz = 0;
                if (++c_part == a.m) { c_part = 0; ++c_outer; }
                const int idx = __builtin_amdgcn_readfirstlane(c_part * a.P + pp - c_e * planeA);
                c_tx = __builtin_amdgcn_readfirstlane(idx % a.ntx); c_ty = __builtin_amdgcn_readfirstlane(idx / a.ntx);
            }
        }
        ++c_k;
    };

    // this lane's NK elements of a halo tile, s = lane + 64 j: where they go in a channel's LDS image (tile-invariant) ...
    unsigned loff[NK];
#pragma unroll
    for (int j = 0; j < NK; ++j) {
        const int s = lane + 64 * j;
        loff[j] = (j * 64 + 63 < SPX || s < SPX) ? (unsigned)((s / ZS) * PS + ((s / IX) % IY) * RS + G::xpos(s % IX)) : (unsigned)(CS - 4);
    }
    // ... and, for the cursor's tile, their byte offsets in the volume and validity masks (zero padding)
    unsigned coff[NK], xm[NK]; int x_tn = 0;
    auto x_desc = [&]() __attribute__((always_inline)) {
        x_tn = c_n;
#pragma unroll
        for (int j = 0; j < NK; ++j) {
            const int s = lane + 64 * j;
            const int gx = S * c_ox0 - 1 + s % IX, gy = S * c_oy0 - 1 + (s / IX) % IY, gz = S * c_oz0 - 1 + s / ZS;
            const bool ok = (j * 64 + 63 < SPX || s < SPX) & ((unsigned)gx < (unsigned)a.W) & ((unsigned)gy < (unsigned)a.H) &
                            ((unsigned)gz < (unsigned)a.D);                          // (bitwise: no branches)
            xm[j] = ok ? 0xFFFFFFFFu : 0u;
            coff[j] = (4u * (unsigned)(gz * HW + gy * a.W + gx)) & xm[j];
        }
    };
    // two register sets: what is fetched during unit u is converted and written during unit u + 1 (a whole unit for the loads to land)
    float xr[2][XPU][NK], xsc[2][XPU], xsh[2][XPU], xsl[2][XPU]; unsigned xmk[2][NK];
    auto issue_x = [&](int set, int c0) __attribute__((always_inline)) {   // channels LCH lw + c0 .. + XPU - 1 of the group, tile x_tn
#pragma unroll
        for (int e = 0; e < XPU; ++e) {
            // a channel beyond Cin reads channel 0: finite values into accumulators that are never stored
            const int ci = cg * 16 + LCH * lw + c0 + e;
            const int cic = ci < a.Cin ? ci : 0;
            const bool first = cic < a.C0;
            const float* xs = first ? a.x0 : a.x1;
            const float* ss = first ? a.ss0 : a.ss1;
            const int Cs = first ? a.C0 : a.C1, cl = first ? cic : cic - a.C0;
            // buffer loads: descriptor of the sample (scalar), channel offset (scalar), element offset (lane) in one instruction
            const wp_rsrc rs = wp_make_rsrc(xs + (size_t)x_tn * Cs * DHW, (unsigned)Cs * cstride);
            const unsigned soff = (unsigned)cl * cstride;
#pragma unroll
            for (int j = 0; j < NK; ++j) xr[set][e][j] = wp_load(rs, coff[j], soff);
            const bool has = ss != nullptr;
            xsc[set][e] = has ? ss[((size_t)x_tn * Cs + cl) * 2] : 1.f;
            xsh[set][e] = has ? ss[((size_t)x_tn * Cs + cl) * 2 + 1] : 0.f;
            xsl[set][e] = has ? a.slope : 1.f;                  // fma(v, 1, 0) and max(v, v * 1) are exact
        }
#pragma unroll
        for (int j = 0; j < NK; ++j) xmk[set][j] = xm[j];
    };
    auto write_x = [&](int set, int buf, int c0) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < XPU; ++e) {
            char* dst = xbuf + buf * XB + (LCH * lw + c0 + e) * CS;
#pragma unroll
            for (int j = 0; j < NK; ++j) {
                const float v = wp_norm_lrelu(xr[set][e][j], xsc[set][e], xsh[set][e], xsl[set][e]);
                *reinterpret_cast<unsigned*>(dst + loff[j]) = wp_split(__builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) & xmk[set][j]));   // zero padding comes after the activation
            }
        }
    };
    float dyr[2][NDY]; unsigned dmk[2] = {0u, 0u};
    auto issue_dy = [&](int set, int z) __attribute__((always_inline)) {   // unit z of the cursor's tile
        const int ox = c_ox0 + dvx, oy = c_oy0 + dvy, oz = c_oz0 + z;
        const bool dok = (ox < a.Wo) & (oy < a.Ho) & (oz < a.Do);
        dmk[set] = dok ? 0xFFFFFFFFu : 0u;
        const wp_rsrc rs = wp_make_rsrc(a.dy + (size_t)c_n * a.Cout * DHWo, (unsigned)a.Cout * dcstride);
        // rows walk down the channels by adding to the lane offset (one scalar live instead of NDY row offsets); a row beyond Cout
        // repeats the last channel: finite values into accumulator rows that are never stored
        int rows = a.Cout - g * 32 - dcol0;                   // rows of this lane's column that exist
        if constexpr (DUNI) asm volatile("" : "+s"(rows));    // (opaque: keeps the per-row selects out of the loop-invariant SGPRs)
        else asm volatile("" : "+v"(rows));
        unsigned voff = ((4u * (unsigned)(oz * HWo + oy * a.Wo + ox)) & dmk[set]) + (unsigned)(g * 32 + dcol0) * dcstride;
#pragma unroll
        for (int j = 0; j < NDY; ++j) {
            dyr[set][j] = wp_load(rs, voff, 0u);
            voff += (j + 1) * CPK < rows ? CPK * dcstride : 0u;
        }
    };
    auto write_dy = [&](int set, int buf) __attribute__((always_inline)) {
        char* dst = dyl + buf * DYB + dcol0 * DYSB + dv * 4;
#pragma unroll
        for (int j = 0; j < NDY; ++j)
            *reinterpret_cast<unsigned*>(dst + j * CPK * DYSB) = wp_split(__builtin_bit_cast(float, __builtin_bit_cast(unsigned, dyr[set][j]) & dmk[set]));
    };

    if (nk > 0) {                                              // unit (0, 0) and the whole of tile 0, one piece after the other
        advance();
        issue_dy(0, 0);
        x_desc();
#pragma unroll
        for (int c0 = 0; c0 < LCH; c0 += XPU) { issue_x(0, c0); write_x(0, 0, c0); }
        write_dy(0, 0);
        issue_dy(1, 1);                                        // in flight across the barrier: unit (0, 1) and the first channels of tile 1
        if (nk > 1) { advance(); x_desc(); }
        issue_x(1, 0);                                         // (a single tile: tile 0 once more, into a buffer nobody reads)
    }
    __syncthreads();
    if (stamping) st_t = __builtin_amdgcn_s_memtime();
    for (int k = 0; k < nk; ++k) {
#pragma unroll
        for (int z = 0; z < TZ; ++z) {
            // The consumers are on unit (k, z).  Register set z^1 holds what the previous unit fetched: dy of the next unit and
            // channels z XPU.. of tile k + 1.  First fetch, into set z, what the unit after the next needs (the cursor is on tile
            // k + 1 from the unit before) ...
            // BRANCH-FREE (round 5): every unit issues its loads and writes its previous fetch, past the last tile too -- the cursor then
            // stays on the last tile, whose data is fetched again and written into buffers nobody reads any more.  With the issues under
            // `if (k + 1 < nk)` the compiler's counted waits in front of the LDS writes were the minimum over both paths, i.e. they assumed
            // that nothing had been issued since: vmcnt(7) .. vmcnt(0) made every unit wait for the loads it had just issued -- a full memory
            // latency (~4.4 k cycles) per unit of 3.1 k cycles of MFMA work.  (The forward kernels learnt this in round 3.)
            if (z == 0) {                                      // next unit (k, 1): then dy of (k + 1, 0), channels XPU.. of tile k + 1
                issue_dy(0, 0); issue_x(0, XPU);
            } else {                                           // next unit (k + 1, 0): then dy of (k + 1, 1), channels 0.. of tile k + 2
                issue_dy(1, 1);
                if (k + 2 < nk) { advance(); x_desc(); }       // (scalar cursor and lane offsets only: no loads under the branch)
                issue_x(1, 0);
            }
            // ... then convert and write the previous fetch
            write_dy(z ^ 1, (z + 1) & 1);
            write_x(z ^ 1, (k + 1) & 1, z * XPU);
            if (stamping) { const unsigned long long t = __builtin_amdgcn_s_memtime(); st_work += t - st_t; st_t = t; }
            __syncthreads();
            if (stamping) { const unsigned long long t = __builtin_amdgcn_s_memtime(); st_wait += t - st_t; st_t = t; }
        }
    }
    if (stamping && lane == 0 && wave == 4) { WMH_STAMPS(a)[blockIdx.x * 4 + 2] = st_work; WMH_STAMPS(a)[blockIdx.x * 4 + 3] = st_wait; }
}

__global__ __launch_bounds__(256) void wgrad_f16x2_reduce_kernel(const float* part, float* dW, size_t n, int P) { wmh_slab_reduce_body(part, dW, n, P); }   // wmh_common.h

struct WPPlan { int TX, TY, TZ, ntx, nty, ntz, ntiles, NG, NCG, P, order, m, q, nk; };
static inline int wp_cdiv(int a, int b) { return (a + b - 1) / b; }
// (D, H, W): the OUTPUT extent
static bool plan_wp(int Cin, int Cout, int N, int D, int H, int W, int stride, WPPlan& q) {
    if (stride == 2) { if (W >= 12) { q.TX = 16; q.TY = 2; q.TZ = 2; } else return false; }
    else if (W >= 24) { q.TX = 32; q.TY = 4; q.TZ = 2; } else if (W >= 12) { q.TX = 16; q.TY = 8; q.TZ = 2; } else if (W >= 8) { q.TX = 8; q.TY = 8; q.TZ = 2; } else return false;
    q.ntx = wp_cdiv(W, q.TX); q.nty = wp_cdiv(H, q.TY); q.ntz = wp_cdiv(D, q.TZ);
    q.ntiles = N * q.ntx * q.nty * q.ntz;
    q.NG = wp_cdiv(Cout, 32); q.NCG = wp_cdiv(Cin, 16);
    // one workgroup per CU (140 KB of LDS): P = the power of two that brings the grid closest to 256 from below
    const int pairs = q.NG * q.NCG;
    int P = 1;
    while (2 * P * pairs <= 256 && 2 * P <= q.ntiles) P *= 2;
    q.P = P;
    const int A = q.ntx * q.nty;
    if (P <= A && A % P == 0) { q.order = 1; q.m = A / P; q.q = 1; q.nk = q.ntz * q.m * N; }
    else if (P > A && P % A == 0 && N % (P / A) == 0) { q.order = 2; q.m = 1; q.q = P / A; q.nk = q.ntz * (N / q.q); }
    else { q.order = 0; q.m = 1; q.q = 1; q.nk = wp_cdiv(q.ntiles, P); }
    return true;
}
static inline int wp_out(int v, int stride) { return (v - 1) / stride + 1; }
// 1 when the pair form serves this layer (stride 1: rows of >= 8 voxels; stride 2 -- round 5 -- output rows of >= 12 voxels); the other
// layers stay on the fp32 kernel.  (D, H, W): the input extent
extern "C" int wmh_conv3d_k3_wgrad_f16x2_supported(int Cin, int Cout, int N, int D, int H, int W, int stride) {
    WPPlan q;
    return (stride == 1 || stride == 2) && Cin >= 1 && Cout >= 1 && N >= 1 &&
           plan_wp(Cin, Cout, N, wp_out(D, stride), wp_out(H, stride), wp_out(W, stride), stride, q) ? 1 : 0;
}
extern "C" size_t wmh_conv3d_k3_wgrad_f16x2_workspace_floats(int Cin, int Cout, int N, int D, int H, int W) {
    WPPlan q;
    if (!plan_wp(Cin, Cout, N, D, H, W, 1, q)) return 0;
    return (size_t)q.P * Cout * Cin * 27;
}
extern "C" size_t wmh_conv3d_k3s2_wgrad_f16x2_workspace_floats(int Cin, int Cout, int N, int D, int H, int W) {
    WPPlan q;
    if (!plan_wp(Cin, Cout, N, wp_out(D, 2), wp_out(H, 2), wp_out(W, 2), 2, q)) return 0;
    return (size_t)q.P * Cout * Cin * 27;
}
template <int S, int TX, int TY, int TZ, int NLW>
static int launch_wp(const WgradPArgs& a, unsigned nblk, hipStream_t st) {
    constexpr int lds = WpGeo<S, TX, TY, TZ, NLW>::LDS;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_k3_wgrad_f16x2_kernel<S, TX, TY, TZ, NLW>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((conv3d_k3_wgrad_f16x2_kernel<S, TX, TY, TZ, NLW>), dim3(nblk), dim3(256 + 64 * NLW), lds, st, a);
    return wmh_check_launch("conv3d_k3_wgrad_f16x2_kernel");
}
static int wgrad_f16x2_impl(const float* x0, const float* ss0, int C0, const float* x1, const float* ss1, int C1, const float* dy, float* dW,
                            float* workspace, int N, int D, int H, int W, int Cout, int stride, float lrelu_slope, unsigned long long* stamps, void* stream) {
    if (!x0 || !dy || !dW || !workspace || C0 < 1 || N < 1 || Cout < 1) return wmh_fail(WMH_EINVAL, "conv3d_k3_wgrad_f16x2: bad args");
    if (!(lrelu_slope >= 0.f && lrelu_slope <= 1.f)) return wmh_fail(WMH_EINVAL, "conv3d_k3_wgrad_f16x2: LeakyReLU slope must be in [0, 1]");
    if (!x1) C1 = 0;
    const int Cin = C0 + C1;
    const int Do = wp_out(D, stride), Ho = wp_out(H, stride), Wo = wp_out(W, stride);
    WPPlan q;
    if (!plan_wp(Cin, Cout, N, Do, Ho, Wo, stride, q)) return wmh_fail(WMH_EINVAL, "conv3d_k3_wgrad_f16x2: layer not served (rows shorter than 8 voxels, or than 12 output voxels at stride 2)");
    if ((size_t)(C0 > C1 ? C0 : C1) * D * H * W * 4 >= ((size_t)1 << 32) || (size_t)Cout * Do * Ho * Wo * 4 >= ((size_t)1 << 32))
        return wmh_fail(WMH_EINVAL, "conv3d_k3_wgrad_f16x2: volume too large for 32-bit lane offsets");
    WgradPArgs a;
    a.x0 = x0; a.ss0 = ss0; a.C0 = C0; a.x1 = x1; a.ss1 = ss1; a.C1 = C1; a.dy = dy; a.part = workspace;
    a.N = N; a.D = D; a.H = H; a.W = W; a.Cout = Cout; a.Cin = Cin; a.Do = Do; a.Ho = Ho; a.Wo = Wo;
    a.ntx = q.ntx; a.nty = q.nty; a.ntz = q.ntz; a.ntiles = q.ntiles; a.NG = q.NG; a.NCG = q.NCG; a.P = q.P;
    a.order = q.order; a.m = q.m; a.q = q.q; a.nk = q.nk; a.slope = lrelu_slope; a.stamps = stamps;
    if (wmh_diag_env_set("WMH_WGRAD_PLAIN_ORDER")) { a.order = 0; }      // diagnostic: x -> y -> z tile order
    const unsigned nblk = (unsigned)(q.NG * q.NCG * q.P);
    hipStream_t st = (hipStream_t)stream;
    static const bool four = wmh_diag_env_set("WMH_WGRAD_LOADERS4");   // diagnostic: 4 loader waves instead of 8
    int rc;
    if (stride == 2) rc = launch_wp<2, 16, 2, 2, 8>(a, nblk, st);
    else if (q.TX == 8) rc = launch_wp<1, 8, 8, 2, 8>(a, nblk, st);
    else rc = four ? (q.TX == 32 ? launch_wp<1, 32, 4, 2, 4>(a, nblk, st) : launch_wp<1, 16, 8, 2, 4>(a, nblk, st))
                   : (q.TX == 32 ? launch_wp<1, 32, 4, 2, 8>(a, nblk, st) : launch_wp<1, 16, 8, 2, 8>(a, nblk, st));
    if (rc != WMH_OK) return rc;
    const size_t n = (size_t)Cout * Cin * 27;
    const unsigned rb = wmh_slab_reduce_blocks(n);
    hipLaunchKernelGGL(wgrad_f16x2_reduce_kernel, dim3(rb), dim3(256), 0, st, workspace, dW, n, q.P);
    return wmh_check_launch("wgrad_f16x2_reduce_kernel");
}
extern "C" int wmh_conv3d_k3_wgrad_f16x2(const float* x0, const float* ss0, int C0, const float* x1, const float* ss1, int C1, const float* dy,
                                         float* dW, float* workspace, int N, int D, int H, int W, int Cout, float lrelu_slope, void* stream) {
    return wgrad_f16x2_impl(x0, ss0, C0, x1, ss1, C1, dy, dW, workspace, N, D, H, W, Cout, 1, lrelu_slope, nullptr, stream);
}
// The same for the 3x3x3 convolution with stride 2 (pad 1): (D, H, W) is the INPUT extent, dy has ((D - 1) / 2 + 1, ...); workspace >=
// wmh_conv3d_k3s2_wgrad_f16x2_workspace_floats floats
extern "C" int wmh_conv3d_k3s2_wgrad_f16x2(const float* x0, const float* ss0, int C0, const float* x1, const float* ss1, int C1, const float* dy,
                                           float* dW, float* workspace, int N, int D, int H, int W, int Cout, float lrelu_slope, void* stream) {
    return wgrad_f16x2_impl(x0, ss0, C0, x1, ss1, C1, dy, dW, workspace, N, D, H, W, Cout, 2, lrelu_slope, nullptr, stream);
}
// diagnostics (libdeepwmh_amd_diag.so only, include/deepwmh_amd_diag.h): the same launch with per-workgroup cycle counts (4 per workgroup:
// consumer work / wait, loader work / wait)
#ifdef WMH_DIAG
extern "C" size_t wmh_conv3d_k3_wgrad_f16x2_stamp_count(int Cin, int Cout, int N, int D, int H, int W) {
    WPPlan q;
    if (!plan_wp(Cin, Cout, N, D, H, W, 1, q)) return 0;
    return (size_t)q.NG * q.NCG * q.P * 4;
}
extern "C" int wmh_conv3d_k3_wgrad_f16x2_stamped(const float* x0, const float* ss0, int C0, const float* x1, const float* ss1, int C1, const float* dy,
                                                 float* dW, float* workspace, int N, int D, int H, int W, int Cout, float lrelu_slope,
                                                 unsigned long long* stamps, void* stream) {
    return wgrad_f16x2_impl(x0, ss0, C0, x1, ss1, C1, dy, dW, workspace, N, D, H, W, Cout, 1, lrelu_slope, stamps, stream);
}
#endif
