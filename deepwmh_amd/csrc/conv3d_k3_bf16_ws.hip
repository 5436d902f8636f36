// conv3d_k3_bf16_ws.hip -- wave-specialised, persistent form of the bf16-operand 3x3x3 convolution.
//
// The one-role bf16 kernel (conv3d_k3_bf16.hip) runs its phases one after another -- stage a chunk (global loads,
// InstanceNorm+LeakyReLU, bf16 pack, LDS writes), then 27 taps of MFMA, then the stores -- and measured additive:
// 0.64 ms of staging instructions + 0.63 ms of exposed HBM latency + 0.6 ms of MFMA + 0.43 ms of stores for the
// 32->32 layer at 128^3 x 8 (tools/conv_bf16_bench.py, WMH_BF16_ABL).  Here the roles are split inside one
// 512-thread workgroup per CU so that they overlap:
//   waves 0-3  consumers: B operand = one ds_read_b128 per (tap, M-tile), A operand (weights) = one ds_read_b128 per
//              (tap, cout tile) -- 16-byte GLOBAL loads inside the MFMA loop cost ~13 cycles per MFMA even from cache
//              (tools/micro/bf16_inner.hip: 47-49 against 35.8 cycles per MFMA with the weights in LDS) --
//              v_mfma_f32_32x32x16_bf16, epilogue (bias, fp32 IN moments, stores);
//   waves 4-7  loaders: the next chunk's halo tile in two register halves, so that 64+ loads per lane stay in flight
//              while the other half is normalised, packed (v_cvt_pk_bf16_f32) and written to LDS.
// LDS: activations [2][halo voxels][16 cin] bf16 (channel-interleaved and XOR-swizzled as in the one-role kernel) and ONE
// buffer of the chunk's weight fragments (two do not fit in 160 KB): the loaders hold the next chunk's weights in
// registers and drop them into LDS between two barriers at the end of a chunk (~300 cycles of a ~7000-cycle chunk).  Persistent (grid = number of CUs): loaders run one chunk ahead across tile boundaries and the
// weight ring continues into the next tile, so a tile has no prologue.  Same arithmetic and k order as the one-role
// kernel: bit-identical results.  Round 5: split-K slices of a tile's channel chunks as separate parts of the persistent sequence (a.KS,
// for layers whose tiles do not fill the chip) and, in the pair mode, rows of 8 voxels (configuration 24: an M-tile = 4 rows of 8).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include <utility>
#include "conv3d_common.h"

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned ws_pack_bf16x2(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    const bf16x2_t b = __builtin_convertvector(v, bf16x2_t);
    return __builtin_bit_cast(unsigned, b);
}

// One v_max_f32.  fmaxf() on these values compiled to three (the two operands are "canonicalised" by a max with
// themselves first, although they come straight out of packed FMAs): 1153 instead of 384 v_max in the loader's code.
__device__ __forceinline__ float ws_vmax(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// Uniform read of a value no kernel in flight writes (InstanceNorm scale/shift come from the previous launch): through the
// constant address space, so that it is a scalar-cache load on its own counter instead of a vector load queued behind the
// activation prefetch on the in-order vmcnt.
__device__ __forceinline__ float ld_uniform(const float* p) {
    return *reinterpret_cast<const __attribute__((address_space(4))) float*>(reinterpret_cast<unsigned long long>(p));
}

// The low parts of two values as one dword: fp16(t0 - hi0) | fp16(t1 - hi1) << 16 with hi = the two halves of h2.  v_fma_mixlo / mixhi_f16 take the
// fp16 operand as it is, subtract in fp32 and round once to fp16: two instructions per pair where v_cvt_f32_f16 x 2 + v_sub_f32 x 2 +
// v_cvt_pk_f16_f32 were five.  Same bits: t - hi is exact in fp32 (hi is t cut to 11 significant bits), so both forms round the same number once.
__device__ __forceinline__ unsigned ws_lo_pair(unsigned h2, float t0, float t1) {
    unsigned d;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h2), "v"(t0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(d) : "v"(h2), "v"(t1));
    return d;
}

// ---- the pair kernel's MFMA sequence (three partial products), flattened ------------------------------------------------------------
// Per 8-channel chunk a consumer wave issues, for every output row m of its MT rows, 41 A fragments' worth of MFMAs ("groups"):
// per (kd, kr) the three taps' [w_hi | w_hi] against [x_hi | x_lo] (pr 0..2), [w_lo(kw 0) | w_lo(kw 1)] against [x_hi(v) | x_hi(v + 1)]
// (pr 3), and one more (pr 4) that pairs the remaining w_lo x_hi terms of kw = 2 across kr (kr 1: taps (kr 0, kw 2) | (kr 1, kw 2)),
// across kd (kd 1, kr 2: (kd 0, kr 2, kw 2) | (kd 1, kr 2, kw 2)) or stands alone half empty (kd 2, kr 2).
#ifndef WMH_PAIR3
#define WMH_PAIR3 1      /* 0: all four partial products, 108 MFMAs per 27 taps and 16 channels (the form until r03x) */
#endif
struct P3Grp { int kd, kr, pr; };
constexpr int P3_NG = 41;
constexpr P3Grp p3_grp(int gi) {
    int n = 0;
    for (int kd = 0; kd < 3; ++kd)
        for (int kr = 0; kr < 3; ++kr)
            for (int pr = 0; pr < 5; ++pr) {
                if (pr == 4 && (kr == 0 || (kr == 2 && kd == 0))) continue;
                if (n == gi) return P3Grp{kd, kr, pr};
                ++n;
            }
    return P3Grp{0, 0, 0};
}
static_assert(p3_grp(P3_NG - 1).kd == 2 && p3_grp(P3_NG - 1).kr == 2 && p3_grp(P3_NG - 1).pr == 4, "41 groups");
template <int... I, class F>
__device__ __forceinline__ void p3_static_for(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }

#ifndef WMH_BF16_WS_RING16
#define WMH_BF16_WS_RING16(NHS) ((NHS) % 3 == 0 ? 3 : 2)
#endif
#ifndef WMH_BF16_WS_RING
#define WMH_BF16_WS_RING(NHS) ((NHS) % 3 == 0 ? 3 : 2)   /* register sets per lane; a full-chunk ring (6 sets) spills 257 VGPRs */
#endif

// IN16: both sources are bf16 tensors (a quad is one 8-byte load per channel, widened by a shift); OUT16: y is stored as bf16.
// PAIR (the "pair" mode, fp32 tensors): every operand is split into TWO FP16 numbers, x = x_hi + x_lo with x_hi = fp16(x) (round
// toward zero: saturates instead of overflowing), x_lo = fp16(x - x_hi) (nearest) -- 21-22 significant bits together, against
// fp32's 24 -- and the product is formed from all four partial products on v_mfma_f32_32x32x16_f16 (same rate as the bf16
// instruction).  Round 2 split into bf16 pairs (16 bits together): 3.4e-6 relative error per layer against 0.5-1.4e-6 for
// the exact-fp32 kernel; fp16 pairs are at the fp32 kernel's level.  fp16's narrow exponent is no obstacle here: activations
// behind InstanceNorm + LeakyReLU are O(1) (a low part below fp16's normal range only loses bits that are < 2^-25 absolute),
// and the weights are scaled by 2^8 before the split (wmh_conv3d_k3_pack_weights_f16x2; |w| < 255 required), undone exactly
// on the accumulators before the epilogue.  A
// chunk is 8 input channels; the 16 k slots of an MFMA carry B = [x_hi | x_lo] of those channels (the bf16 kernel's LDS image:
// 32 bytes per voxel, the hi piece where channels 0-7 were, the lo piece where channels 8-15 were -- ONE operand read per
// voxel tile and tap, as before), and the tap is issued twice, with A = [w_hi | w_hi] and A = [w_lo | w_lo]:
// sum_k A B = w_hi (x_hi + x_lo), then w_lo (x_hi + x_lo).  (The other way round -- one A, two B -- costs an LDS read per
// MFMA, and with one 32-channel output tile per wave that is exactly the LDS bandwidth: 4.3 instead of 3.x ms per layer.)
// 4 MFMAs per 16 channels and tap instead of 1 -- still 4x fewer cycles than the fp32 instruction -- with fp32 accumulation,
// fp32 tensors and the fp32 epilogue.  Since the end of round 3 (WMH_PAIR3, the default) the fourth partial product, w_lo x_lo --
// 2^-22 of a product, below the split's own truncation -- is left out and the three w_lo x_hi terms of a (kd, kr) take two
// MFMAs instead of three: 10 MFMAs per three taps and 16 channels (see the consumer loop); a layer's error against float64 is
// unchanged to two digits (tools/f16x2_check.py: 3.5e-7 / 6.8e-7 relative L2 on the 32- / 128-channel test layers, the exact-fp32
// kernel's 5.3e-7 / 1.07e-6).
template <class C, bool IN16 = false, bool OUT16 = false, bool PAIR = false>
__global__ __launch_bounds__(512, 2) void conv3d_k3_bf16_ws_kernel(const ConvArgs a, const int T /*tiles*/) {
    constexpr int S = C::S, MT = C::MT, NT = C::NT, CKB = PAIR ? 8 : 16;
    static_assert(!(PAIR && (IN16 || OUT16)), "f16x2 works on fp32 tensors");
    constexpr int LTH = 256;                                   // loader threads
    constexpr int NIT = (C::SP + LTH - 1) / LTH;
    constexpr int CS = NIT * LTH;                              // padded voxel count: every loader store in range
    constexpr int WQ = 27 * NT * 64;                           // 16-byte weight pieces per chunk
    constexpr int WIT = (WQ + LTH - 1) / LTH;
    static_assert(C::TN == 1 && C::WM == 4 && S == 1, "four consumer waves, stride 1");
    extern __shared__ __attribute__((aligned(16))) uint4 ldsq[];      // [2][CS voxels][2] activations, then [27][NT][64] weights
    uint4* wlds = ldsq + 2 * CS * 2;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave >= 4;
    const int HW = a.H * a.W;
    const size_t DHW = (size_t)a.D * HW;
    const int nch = a.nchunks;

    // Split-K (a.KS > 1; T counts the parts): part t of the order below is slice t % KS of the channel chunks of tile t / KS -- the slices
    // of a tile run side by side on different workgroups and leave raw partial sums in slab ks of a.y (bias and moments come with
    // the reduce, wmh_launch_splitk_reduce).  For layers whose tiles alone do not fill the chip (16^3 / 8^3 levels at batch 2).
    const int KS = a.KS;
    auto tile_of = [&](int v, int& g, int& tx, int& ty, int& tz, int& n0, int& ks) {
        const int q = T >> 3, r = T & 7, xcd = v & 7;
        int t = (a.xcd_swizzle & 1) ? (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3) : v;
        ks = 0;
        if (KS > 1) { ks = t % KS; t /= KS; }
        wmh_tile_coords(a, t, g, tx, ty, tz, n0);
    };
    auto ch_lo_of = [&](int ks) { return KS > 1 ? (nch * ks) / KS : 0; };            // (nch * KS < 2^31: the host checks)
    auto ch_hi_of = [&](int ks) { return KS > 1 ? (nch * (ks + 1)) / KS : nch; };
    int ntiles_mine = 0, total = 0;                         // parts / chunk iterations of this workgroup
    for (int v = blockIdx.x; v < T; v += gridDim.x) {
        ++ntiles_mine;
        if (KS > 1) { int g, tx, ty, tz, n0, ks; tile_of(v, g, tx, ty, tz, n0, ks); total += ch_hi_of(ks) - ch_lo_of(ks); }
    }
    if (KS == 1) total = ntiles_mine * nch;

    if (loader) {
        // ================================ loader waves =====================================================
        __builtin_amdgcn_s_setprio(3);                        // ahead of the MFMA-saturating consumer on the same SIMD
        const int lt = tid - 256;
        // Staging unit = an aligned quad of 4 consecutive x voxels of one halo row (one 16-byte load per channel): the row
        // [ox0-1, ox0+TX] is covered by the aligned span [ox0-4, ox0+TX+4), QPR quads; quads are wholly inside or wholly
        // outside the volume (W % 4 == 0, checked by the host).  Voxels of the span outside the halo row are dropped.
        constexpr int ROWS = C::IY * C::IZ, QPR = (C::TX + 8) / 4, NQ = ROWS * QPR;
        constexpr int NSL = (NQ + LTH - 1) / LTH;                 // quad slots per loader thread (3 / 2)
        static_assert(NSL >= 2 && NSL <= 3, "slot pipeline");
        int q_off[NSL]; unsigned q_ok = 0;                        // of the tile the NEXT issue() reads
        int n0_cur = 0;
        auto setup = [&](int v) __attribute__((always_inline)) {
            int g, tx, ty, tz, n0, ks_;
            tile_of(v, g, tx, ty, tz, n0, ks_);
            // the tile coordinates come out of VALU divisions: make them scalar again, or everything indexed with them
            // (the InstanceNorm scale/shift reads above all) turns into vector loads on the in-order vmcnt queue
            n0_cur = __builtin_amdgcn_readfirstlane(n0);
            tx = __builtin_amdgcn_readfirstlane(tx); ty = __builtin_amdgcn_readfirstlane(ty); tz = __builtin_amdgcn_readfirstlane(tz);
            const int ix0 = tx * C::TX - 4, iy0 = ty * C::TY - 1, iz0 = tz * C::TZ - 1;
            q_ok = 0;
#pragma unroll
            for (int i = 0; i < NSL; ++i) {
                const int sidx = lt + i * LTH;
                const int row = sidx / QPR, q = sidx - row * QPR;
                const int gx = ix0 + 4 * q, gy = iy0 + row % C::IY, gz = iz0 + row / C::IY;
                const bool ok = sidx < NQ && gx >= 0 && gx < a.W && gy >= 0 && gy < a.H && gz >= 0 && gz < a.D;
                q_off[i] = ok ? gz * HW + gy * a.W + gx : 0;
                q_ok |= ok ? (1u << i) : 0u;
            }
        };
        // next chunk's weight fragments (this thread's pieces).  A native vector type: an array of HIP's uint4 (a struct with
        // a union inside) was left in scratch memory by the compiler, and its reload between the two barriers waited with
        // vmcnt(0) for the whole prefetch ring while every wave of the workgroup stood at the barrier (~5 k ticks per chunk)
        typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
        u32x4_t wst[WIT];
        // The chunk being produced: linear index pq of this workgroup's sequence = (tile counter pk, channel chunk pc), and the
        // cout group pg of that tile.  Kept as counters: `q / nch`, `q % nch` and tile_of() per chunk were ~400 scalar
        // instructions of the ~1700 a loader wave issues per chunk -- and the loader's instruction count is what bounds it.
        int pq = 0, pk = 0, pc = 0, pg = 0, p_hi = nch;   // p_hi: one past the last chunk of part pk
        unsigned long long t_issue = 0, t_conv = 0;      // diagnostics (WMH_STAMPS(a)): loads issued / conversion + LDS writes
        auto part_of = [&](int k, int& grp, int& lo, int& hi) __attribute__((always_inline)) {   // cout group and chunk range of my k-th part
            int g, tx, ty, tz, n0, ks;
            tile_of(blockIdx.x + k * gridDim.x, g, tx, ty, tz, n0, ks);
            ks = __builtin_amdgcn_readfirstlane(ks);
            grp = __builtin_amdgcn_readfirstlane(g);
            lo = ch_lo_of(ks); hi = ch_hi_of(ks);
        };
        auto issue_w = [&]() __attribute__((always_inline)) {                               // the weights of chunk (pk, pc)
            const u32x4_t* wsrc = reinterpret_cast<const u32x4_t*>(a.wf) + ((size_t)pg * nch + pc) * WQ;
#pragma unroll
            for (int i = 0; i < WIT; ++i) wst[i] = wsrc[min(lt + i * LTH, WQ - 1)];
        };
        auto write_w = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < WIT; ++i) { const int k = lt + i * LTH; if (WIT * LTH == WQ || k < WQ) reinterpret_cast<u32x4_t*>(wlds)[k] = wst[i]; }
        };
        // A half-slot = 8 channels of one quad (8 loads of 16 bytes, one 16-byte LDS piece per voxel).  The NHS half-slots of a
        // chunk go through a ring of RS register sets: while one set is normalised, packed and written, RS-1 are in flight.
        // bf16 sources: a set is half the registers, and with half the bytes per load the same ring would keep half the bytes
        // in flight -- the whole chunk rides in the ring instead
        constexpr int HPS = PAIR ? 1 : 2;                           // half-slots per quad slot (PAIR: the 8 channels of a chunk are one)
        constexpr int NHS = HPS * NSL, RS = IN16 ? WMH_BF16_WS_RING16(NHS) : WMH_BF16_WS_RING(NHS);   // RS must divide NHS (static ring indices)
        f32x4_t v[IN16 ? 1 : RS][8];
        uint2 v16[IN16 ? RS : 1][8];
        unsigned okh[RS];
        auto src_of = [&](int ch, const float*& xs, const float*& ss, int& Cs, int& c0) {
            const int cb = ch * CKB;
            if (cb < a.C0) { xs = a.x0; ss = a.ss0; Cs = a.C0; c0 = cb; }
            else           { xs = a.x1; ss = a.ss1; Cs = a.C1; c0 = cb - a.C0; }
        };
        // One scalar base per (tile, chunk) and a 32-bit lane offset per load (quad offset + channel * plane stride, host
        // checks the range): the loads are `global_load_dwordx4 v, v_off, s[base]`, no 64-bit arithmetic per channel.
        const unsigned pstride = 4u * (unsigned)DHW;
        auto issue = [&](auto hc, int ch) __attribute__((always_inline)) {                   // half-slot h of chunk ch of the tile set up last
            constexpr int h = decltype(hc)::value, i = h / HPS, half = h % HPS, r = h % RS;
            ch = __builtin_amdgcn_readfirstlane(ch);          // q % nch ran on the VALU
            const float* xs; const float* ss; int Cs, c0;
            src_of(ch, xs, ss, Cs, c0);
            const int cmax = Cs - 1 - c0;                     // channels past the source's end re-read its last one (zeroed later)
            if constexpr (IN16) {
                const char* base = reinterpret_cast<const char*>(xs) + ((size_t)n0_cur * Cs + c0) * DHW * 2;
                const unsigned qo = 2u * (unsigned)q_off[i];
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    v16[r][c] = *reinterpret_cast<const uint2*>(base + (qo + (unsigned)min(8 * half + c, cmax) * (pstride >> 1)));
            } else {
                const char* base = reinterpret_cast<const char*>(xs + ((size_t)n0_cur * Cs + c0) * DHW);
                const unsigned qo = 4u * (unsigned)q_off[i];
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    v[r][c] = *reinterpret_cast<const f32x4_t*>(base + (qo + (unsigned)min(8 * half + c, cmax) * pstride));
            }
            okh[r] = (q_ok >> i) & 1u;
        };
        // InstanceNorm scale / shift of the chunk being converted: 16 channels, scalar, read once per chunk.  Without a
        // pending normalisation the reads go to {1, 0} (index multiplier 0); channels past the end get scale = shift = 0.
        float scv[CKB], shv[CKB], slope_eff = 1.f;
        static __device__ const float ident2[2] = {1.f, 0.f};
        auto load_ss = [&]() __attribute__((always_inline)) {
            const int ch = pc;
            const float* xs; const float* ss; int Cs, c0;
            src_of(ch, xs, ss, Cs, c0);
            const bool has_ss = ss != nullptr;
            slope_eff = has_ss ? a.slope : 1.f;
            const float* tab = has_ss ? ss + ((size_t)n0_cur * Cs + c0) * 2 : ident2;
            const int mul = has_ss ? 2 : 0;
            const int cmax = Cs - 1 - c0;
#pragma unroll
            for (int c = 0; c < CKB; ++c) {
                const float sc = ld_uniform(tab + mul * min(c, cmax)), sh = ld_uniform(tab + mul * min(c, cmax) + 1);
                scv[c] = c <= cmax ? sc : 0.f;
                shv[c] = c <= cmax ? sh : 0.f;
            }
        };
        auto convert = [&](auto hc, int buf) __attribute__((always_inline)) {
            constexpr int h = decltype(hc)::value, i = h / HPS, half = h % HPS, r = h % RS;
            uint4* dst = ldsq + (size_t)buf * CS * 2;
            const int sidx = lt + i * LTH;
            const int row = sidx / QPR, q = sidx - row * QPR;
            const unsigned keep = okh[r] ? 0xFFFFFFFFu : 0u;      // quad outside the volume: zeros
            // InstanceNorm + LeakyReLU (unpacked fp32 instructions, conv3d_common.h), two
            // channels at a time straight into the packed words: the staging registers of the next chunk's weights must not
            // be pushed out to scratch (their reload sits between the two barriers, where every wave of the workgroup waits,
            // and a scratch reload waits for ALL loads in flight -- the whole prefetch ring)
            unsigned w[4][4];                                      // [voxel of the quad][channel pair]
            unsigned wlo[PAIR ? 4 : 1][4];                           // PAIR: the low parts
#pragma unroll
            for (int c = 0; c < 8; c += 2) {
                f32x4_t t[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int cc = c + e;
                    const float sc1 = scv[8 * half + cc], sh1 = shv[8 * half + cc];
                    f32x2_t xlo, xhi;
                    if constexpr (IN16) {
                        const unsigned u0 = v16[r][cc].x, u1 = v16[r][cc].y;
                        xlo = f32x2_t{__builtin_bit_cast(float, u0 << 16), __builtin_bit_cast(float, u0 & 0xFFFF0000u)};
                        xhi = f32x2_t{__builtin_bit_cast(float, u1 << 16), __builtin_bit_cast(float, u1 & 0xFFFF0000u)};
                    } else {
                        xlo = f32x2_t{v[r][cc][0], v[r][cc][1]};
                        xhi = f32x2_t{v[r][cc][2], v[r][cc][3]};
                    }
                    // unpacked fp32 instructions: packed ones do not overlap with the consumer's MFMAs (conv3d_common.h)
                    t[e] = f32x4_t{wmh_norm_lrelu(xlo[0], sc1, sh1, slope_eff), wmh_norm_lrelu(xlo[1], sc1, sh1, slope_eff),
                                   wmh_norm_lrelu(xhi[0], sc1, sh1, slope_eff), wmh_norm_lrelu(xhi[1], sc1, sh1, slope_eff)};
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if constexpr (PAIR) {                            // fp16 pair: high part toward zero, the remainder to nearest
                        const unsigned h2 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(t[0][k], t[1][k]));
                        w[k][c >> 1] = h2 & keep;
                        wlo[k][c >> 1] = ws_lo_pair(h2, t[0][k], t[1][k]) & keep;
                    } else {
                        w[k][c >> 1] = ws_pack_bf16x2(t[0][k], t[1][k]) & keep;
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int x = 4 * q - 3 + k;                       // position in the halo row
                if (sidx < NQ && x >= 0 && x < C::IX) {
                    const int e = row * C::IX + x;
                    dst[e * 2 + (half ^ ((x >> 3) & 1))] = make_uint4(w[k][0], w[k][1], w[k][2], w[k][3]);
                    if constexpr (PAIR) dst[e * 2 + (1 ^ ((x >> 3) & 1))] = make_uint4(wlo[k][0], wlo[k][1], wlo[k][2], wlo[k][3]);
                }
            }
        };
        // (tile, chunk) after (pk, pc) in the sequence; the last chunk is followed by itself
        auto next_of = [&](int& nk, int& nc, bool& newpart) __attribute__((always_inline)) {
            const bool last = pq + 1 >= total, wrap = pc + 1 == p_hi;
            newpart = wrap && !last;
            nk = newpart ? pk + 1 : pk;
            int lo = 0;
            if (KS > 1 && newpart) { int g_, hi_; part_of(nk, g_, lo, hi_); }
            nc = last ? pc : (wrap ? lo : pc + 1);
        };
        // one step: put half-slot (h + RS - 1) of the running sequence in flight, then convert half-slot h of chunk q.
        // Branch-free on purpose (past the last chunk the ring re-reads the last chunk): the weight fragments issued at the
        // top of produce() are waited for between the two barriers, and the compiler's vmcnt for them is the minimum over all
        // paths of the loads issued since -- with a conditional issue that minimum was 0, i.e. the whole ring had to land
        // while every wave of the workgroup stood at the barrier (~5 k ticks per chunk).
        auto step = [&](auto hc, int buf) __attribute__((always_inline)) {
            constexpr int h = decltype(hc)::value, hn = h + RS - 1;
            unsigned long long ts0 = 0;
            if (WMH_STAMPS(a)) ts0 = __builtin_amdgcn_s_memtime();
            if constexpr (hn < NHS) {
                issue(std::integral_constant<int, hn>{}, pc);
            } else {
                int nk, nc; bool newpart;
                next_of(nk, nc, newpart);
                if constexpr (hn == NHS) { if (newpart) setup(blockIdx.x + nk * gridDim.x); }   // first half-slot of the next chunk: maybe a new tile
                issue(std::integral_constant<int, hn - NHS>{}, nc);
            }
            if (WMH_STAMPS(a)) { const unsigned long long t = __builtin_amdgcn_s_memtime(); t_issue += t - ts0; ts0 = t; }
            convert(hc, buf);
            if (WMH_STAMPS(a)) t_conv += __builtin_amdgcn_s_memtime() - ts0;
        };
        auto produce = [&](int buf) __attribute__((always_inline)) {                          // chunk (pk, pc)
            load_ss();                                        // before the next tile's setup(): n0_cur is still this chunk's
            issue_w();
            step(std::integral_constant<int, 0>{}, buf); step(std::integral_constant<int, 1>{}, buf);
            if constexpr (NHS > 2) step(std::integral_constant<int, 2>{}, buf);
            if constexpr (NHS > 3) step(std::integral_constant<int, 3>{}, buf);
            if constexpr (NHS > 4) { step(std::integral_constant<int, 4>{}, buf); step(std::integral_constant<int, 5>{}, buf); }
        };
        auto advance = [&]() __attribute__((always_inline)) {                                  // to the next chunk of the sequence (clamped)
            int nk, nc; bool newpart;
            next_of(nk, nc, newpart);
            if (pq + 1 < total) ++pq;
            if (nk != pk) { int lo_; part_of(nk, pg, lo_, p_hi); }
            pk = nk; pc = nc;
        };
        if (total > 0) {
            part_of(0, pg, pc, p_hi);
            setup(blockIdx.x);
            issue(std::integral_constant<int, 0>{}, pc);
            if constexpr (RS > 2) issue(std::integral_constant<int, 1>{}, pc);
            if constexpr (RS > 3) issue(std::integral_constant<int, 2>{}, pc);
            if constexpr (RS > 4) issue(std::integral_constant<int, 3>{}, pc);
            if constexpr (RS > 5) issue(std::integral_constant<int, 4>{}, pc);
            produce(0);
            write_w();
        }
        __syncthreads();
        unsigned long long tw = 0, tb = 0, t0 = 0, twl = 0;  // diagnostics (WMH_STAMPS(a) != NULL): work / barrier wait / work next to an epilogue
        for (int it = 0; it < total; ++it) {
            advance();                                       // the chunk written during this iteration: it + 1 (the last iteration
                                                             // repeats the last chunk into the buffer nobody reads any more)
            if (WMH_STAMPS(a)) t0 = __builtin_amdgcn_s_memtime();
            produce((it + 1) & 1);
            if (WMH_STAMPS(a)) { const unsigned long long t = __builtin_amdgcn_s_memtime(); tw += t - t0; if (pc == (nch > 1 ? 1 : 0)) twl += t - t0; t0 = t; }
            __syncthreads();                                 // consumers are done with chunk `it`: its weights may go
            write_w();
            __syncthreads();
            if (WMH_STAMPS(a)) tb += __builtin_amdgcn_s_memtime() - t0;
        }
        if (WMH_STAMPS(a) && tid == 256) {
            unsigned long long* o = WMH_STAMPS(a) + (size_t)blockIdx.x * 8;
            o[4] = tw; o[5] = tb; o[6] = twl;
            unsigned long long* o2 = WMH_STAMPS(a) + (size_t)(gridDim.x + blockIdx.x) * 8;     // second record block
            o2[0] = t_issue; o2[1] = t_conv;               // twl: iterations whose consumers start with the previous tile's epilogue
        }
    } else {
        // ================================ consumer waves ===================================================
        const int j = lane & 31, kh = lane >> 5;
        const int dx = j % C::LX, dy = (j / C::LX) % C::LY, dz = j / (C::LX * C::LY);
        const int ts0 = (wave * MT) % C::MPS;
        int badr[3][MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int ts = ts0 + m;
            const int tyi = ts % C::MY, tzi = ts / C::MY;
            const int e0 = (tzi * C::LZ + dz) * C::ZS + (tyi * C::LY + dy) * C::YS + dx;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) badr[kw][m] = (e0 + kw) * 32 + 16 * (kh ^ (((dx + kw) >> 3) & 1));   // PAIR: hi piece (kh 0) / lo piece (kh 1)
        }
        const char* ldsb = reinterpret_cast<const char*>(ldsq);
        // PAIR: a fragment's 64 pieces are [hi of cout 0..31 | lo of cout 0..31]; both lane halves read the same piece
        const bf16x8* wl = reinterpret_cast<const bf16x8*>(wlds) + (PAIR ? j : lane);
        int g, tx, ty, tz, n0, ks;
        __syncthreads();
        int it = 0;
        unsigned long long tm = 0, tb = 0, te = 0, t0 = 0, tbl = 0;
        for (int vb = blockIdx.x; vb < T; vb += gridDim.x) {
            tile_of(vb, g, tx, ty, tz, n0, ks);
            const int ch_lo = ch_lo_of(ks), ch_hi = ch_hi_of(ks);
            f32x16 acc[NT][MT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = (g * NT + nt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    const float bv = (KS == 1 && a.bias != nullptr && co < a.Cout) ? a.bias[co] * (PAIR ? WMH_PAIR_WSCALE : 1.f) : 0.f;
#pragma unroll
                    for (int m = 0; m < MT; ++m) acc[nt][m][r] = bv;
                }
            for (int ch = ch_lo; ch < ch_hi; ++ch, ++it) {
                const char* cur = ldsb + (size_t)(it & 1) * CS * 32;
                if (WMH_STAMPS(a)) t0 = __builtin_amdgcn_s_memtime();
                // Taps are processed in the order (kd, kw, kr) -- wmh_tap_of(), the one-role kernel's order too -- three kr taps per
                // (kd, kw) group: their weight fragments are read together (9 exposed LDS latencies per chunk instead of 27: the
                // compiler reads a tap's fragments right in front of its first MFMA), and an input row is read once for the up to
                // three (kr, output row) pairs that use it.
                if constexpr (PAIR && WMH_PAIR3) {
                // THREE of the four partial products (w_lo x_lo, 2^-22 of a product, is not formed): the 41 groups of p3_grp().  Every operand
                // is one of the existing LDS images read with a lane-half-dependent address: no second layout, packer or loader.
                //   pr 0..2  A = [w_hi | w_hi](tap kw)                       B = [x_hi | x_lo](voxel + kw)
                //   pr 3     A = [w_lo(kw 0) | w_lo(kw 1)]                   B = [x_hi(voxel) | x_hi(voxel + 1)]
                //   pr 4     kr 1:        A = [w_lo(kr 0, kw 2) | w_lo(kr 1, kw 2)]            B = [x_hi(row, voxel + 2) | x_hi(row + 1, voxel + 2)]
                //            kd 1, kr 2:  A = [w_lo(kd 0, kr 2, kw 2) | w_lo(kd 1, kr 2, kw 2)]  B: the upper lane half one slice (ZS) deeper
                //            kd 2, kr 2:  A = [w_lo(2, 2, 2) | 0]                              B = [x_hi | x_lo](voxel + 2)   (half empty)
                // 41 MFMAs per output row and 8-channel chunk where 40.5 would be every k slot used (the four-product form: 54).
                int bP[MT], bQ[MT], bR[MT];
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    bP[m] = kh ? ((badr[1][m] & ~16) | (16 * ((((dx + 1) >> 3) & 1)))) : badr[0][m];                               // hi piece of voxel + kh
                    bQ[m] = ((badr[2][m] & ~16) | (16 * ((((dx + 2) >> 3) & 1)))) + (kh ? C::YS * 32 : 0);                          // hi piece of voxel + 2 in row + kh
                    bR[m] = ((badr[2][m] & ~16) | (16 * ((((dx + 2) >> 3) & 1)))) + 2 * C::YS * 32 + (kh ? C::ZS * 32 : 0);         // ... in row + 2 of slice + kh
                }
                const bf16x8* wlp = wl + 32 + kh * (NT * 64);                                                // lo piece of tap + kh
                const bf16x8* wlq = wl + 32 + kh * (3 * NT * 64);                                            // lo piece of tap + 3 kh (next kr)
                const bf16x8* wlr = wl + 32 + kh * (9 * NT * 64);                                            // lo piece of tap + 9 kh (next kd)
                const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
                // One flat sequence of P3_NG x MT steps, group-major: a group's A fragment(s) serve MT consecutive steps (one per output
                // row), every step has its own B operand.  A fragments are read one group (MT steps) ahead, B operands RING steps ahead
                // into a register ring; a step = its NT MFMAs, then the ring load that reuses its slot.  sched_barrier pins that order
                // (left alone the compiler hoists the reads of a whole (kd, kr): scratch).  Round 3's step pipeline (a step = one (kr, output
                // row): five A fragments and five B operands, both double buffered, 80 registers, operands read 5 MFMAs ahead) took 12.3 k
                // cycles per chunk at 8 rows x 32 output channels, this one 11.6 k (328 MFMAs = 10.5 k of issue); the 64-channel tile 13.1 k
                // -> 11.3 k.  What the faster issue buys in time is about half of that: the CUs run it at a lower clock (1578 -> 1519 MHz).
                // An epilogue interleaved with the last chunk's MFMAs (profiles/r04g_interleaved_epilogue_ab.txt) saved another 4.6 % of the
                // cycles and LOST time: clock 1440 -> 1310 MHz.  The kernel is where the part's power management lets it be.
                constexpr int RING = NT == 1 ? 8 : 5, NSTEP = P3_NG * MT;
                bf16x8 ring[RING], aF[2][NT];
                auto loadA = [&](auto gc, bf16x8 (&dst)[NT]) __attribute__((always_inline)) {
                    constexpr P3Grp G = p3_grp(decltype(gc)::value);
                    constexpr int t0 = G.kd * 9 + G.kr * 3;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        if constexpr (G.pr < 3) dst[nt] = wl[((t0 + G.pr) * NT + nt) * 64];
                        else if constexpr (G.pr == 3) dst[nt] = wlp[(t0 * NT + nt) * 64];
                        else if constexpr (G.kr == 1) dst[nt] = wlq[((t0 - 3 + 2) * NT + nt) * 64];          // [w_lo(kr 0, kw 2) | w_lo(kr 1, kw 2)]
                        else if constexpr (G.kd == 1) dst[nt] = wlr[((6 + 2) * NT + nt) * 64];               // [w_lo(kd 0, kr 2, kw 2) | w_lo(kd 1, kr 2, kw 2)]
                        else dst[nt] = wl[((t0 + 2) * NT + nt) * 64 + 32];                                    // w_lo(2, 2, 2): the upper half is zeroed at use
                    }
                };
                auto loadB = [&](auto sc) __attribute__((always_inline)) -> bf16x8 {
                    constexpr int st = decltype(sc)::value, m = st % MT;
                    constexpr P3Grp G = p3_grp(st / MT);
                    constexpr int off = (G.kd * C::ZS + G.kr * C::YS) * 32;
                    if constexpr (G.pr < 3) return *reinterpret_cast<const bf16x8*>(cur + badr[G.pr][m] + off);
                    else if constexpr (G.pr == 3) return *reinterpret_cast<const bf16x8*>(cur + bP[m] + off);
                    else if constexpr (G.kr == 1) return *reinterpret_cast<const bf16x8*>(cur + bQ[m] + G.kd * C::ZS * 32);
                    else if constexpr (G.kd == 1) return *reinterpret_cast<const bf16x8*>(cur + bR[m]);
                    else return *reinterpret_cast<const bf16x8*>(cur + badr[2][m] + off);                    // [x_hi | x_lo](voxel + 2) once more
                };
                loadA(std::integral_constant<int, 0>{}, aF[0]);
                p3_static_for(std::make_integer_sequence<int, RING>{}, [&](auto ic) __attribute__((always_inline)) { ring[decltype(ic)::value] = loadB(ic); });
                __builtin_amdgcn_sched_barrier(0);
                p3_static_for(std::make_integer_sequence<int, NSTEP>{}, [&](auto sc) __attribute__((always_inline)) {
                    constexpr int st = decltype(sc)::value, g = st / MT, m = st % MT;
                    constexpr P3Grp G = p3_grp(g);
                    if constexpr (m == 0 && g + 1 < P3_NG) loadA(std::integral_constant<int, g + 1>{}, aF[(g + 1) & 1]);
                    if constexpr (m == 0 && G.pr == 4 && G.kr == 2 && G.kd == 2) {
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) aF[g & 1][nt] = kh ? zero8 : aF[g & 1][nt];                  // A = [w_lo | 0]
                    }
                    const bf16x8 bv = ring[st % RING];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[nt][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, aF[g & 1][nt]), __builtin_bit_cast(f16x8, bv), acc[nt][m], 0, 0, 0);
                    if constexpr (st + RING < NSTEP) ring[st % RING] = loadB(std::integral_constant<int, st + RING>{});
                    __builtin_amdgcn_sched_barrier(0);
                });
                } else
#pragma unroll
                for (int grp = 0; grp < 9; ++grp) {
                    const int kw = grp % 3;
                    bf16x8 av3[3][NT], al3[3][PAIR ? NT : 1];
#pragma unroll
                    for (int kr = 0; kr < 3; ++kr)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            const int tap = (grp / 3) * 9 + kr * 3 + kw;
                            av3[kr][nt] = wl[(tap * NT + nt) * 64];
                            if constexpr (PAIR) al3[kr][nt] = wl[(tap * NT + nt) * 64 + 32];
                        }
#pragma unroll
                  for (int kr = 0; kr < 3; ++kr) {
                    const int rowoff = ((grp / 3) * C::ZS + kr * C::YS) * 32;
                    bf16x8 (&av)[NT] = av3[kr];
                    bf16x8 (&al)[PAIR ? NT : 1] = al3[kr];
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        const bf16x8 bv = *reinterpret_cast<const bf16x8*>(cur + badr[kw][m] + rowoff);
                        if constexpr (PAIR) {                        // high, then low parts of the weights against the same operand
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                acc[nt][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av[nt]), __builtin_bit_cast(f16x8, bv), acc[nt][m], 0, 0, 0);
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                acc[nt][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, al[nt]), __builtin_bit_cast(f16x8, bv), acc[nt][m], 0, 0, 0);
                        } else {
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                acc[nt][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[nt], bv, acc[nt][m], 0, 0, 0);
                        }
                    }
                  }
                }
                if (WMH_STAMPS(a)) { const unsigned long long t = __builtin_amdgcn_s_memtime(); tm += t - t0; t0 = t; }
                __syncthreads();
                __syncthreads();                             // the loaders swap the weight fragments in between
                if (WMH_STAMPS(a)) { const unsigned long long d = __builtin_amdgcn_s_memtime() - t0; tb += d; if (ch == ch_lo) tbl += d; }
            }
            if (WMH_STAMPS(a)) t0 = __builtin_amdgcn_s_memtime();
            if constexpr (PAIR) {                               // undo the weights' 2^8 (exact)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int m = 0; m < MT; ++m) acc[nt][m] *= (1.f / WMH_PAIR_WSCALE);
            }
            conv_epilogue<C, OUT16>(a, acc, g, ks, n0, wave, j, kh, dx, dy, dz, ts0, tz * C::TZ, ty * C::TY, tx * C::TX,
                             (tz * a.nty + ty) * a.ntx + tx);
            if (WMH_STAMPS(a)) te += __builtin_amdgcn_s_memtime() - t0;
        }
        if (WMH_STAMPS(a) && tid == 0) {
            unsigned long long* o = WMH_STAMPS(a) + (size_t)blockIdx.x * 8;
            o[0] = tm; o[1] = tb; o[2] = te; o[3] = (unsigned long long)total; o[7] = tbl;   // tbl: barrier wait of a tile's first chunk
        }
    }
}

typedef ConvCfg<1, 32, 1, 4, 8, 1, 8, 1, 4, 6> WBCfg0;
typedef ConvCfg<1, 16, 2, 2, 16, 1, 4, 2, 4, 6> WBCfg2;
typedef ConvCfg<1, 32, 1, 4, 4, 1, 4, 2, 4, 6> WBCfgW;      // configuration 21: 4 x 4 x 32 voxels x 64 output channels
typedef ConvCfg<1, 8, 4, 8, 8, 1, 4, 2, 4, 6> WBCfg8;       // configuration 24 (f16x2): 8 x 8 x 8 voxels x 64 output channels, an M-tile = 4 rows of 8

template <class C, bool IN16 = false, bool OUT16 = false, bool PAIR = false>
static int launch_bf16_ws(const ConvArgs& a_in, hipStream_t st) {
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
        if (ncu < 8) ncu = 8;
        ncu &= ~7;
    }
    constexpr int NITL = (C::SP + 255) / 256;
    constexpr int lds = 2 * NITL * 256 * 32 + 27 * C::NT * 64 * 16;
    static_assert(lds <= 160 * 1024, "LDS");
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_k3_bf16_ws_kernel<C, IN16, OUT16, PAIR>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const long long T = (long long)a_in.ntx * a_in.nty * a_in.ntz * a_in.N * a_in.G * (a_in.KS > 1 ? a_in.KS : 1);      // parts
    if (T > 0x7fffffffLL) return wmh_fail(WMH_EINVAL, "conv3d_k3 bf16 (wave-specialised): too many tiles");
    const int grid = T < ncu ? (int)T : ncu;
    ConvArgs a = a_in;
    static const int supertile = wmh_diag_env("WMH_SUPERTILE", 1);
    if (supertile && (a.xcd_swizzle & 1)) wmh_pick_supertile(a, C::TY, C::TZ, grid / 8 > 0 ? grid / 8 : 1);
    hipLaunchKernelGGL((conv3d_k3_bf16_ws_kernel<C, IN16, OUT16, PAIR>), dim3(grid), dim3(512), lds, st, a, (int)T);
    return wmh_check_launch("conv3d_k3_bf16_ws_kernel");
}
// a: fully prepared by wmh_conv3d_k3_fwd_bf16 (one-role geometry: npart, nchunks, G ... are the same)
// in16 / out16: bf16 storage of the sources / of y (cfg 0 only -- the 32-wide full-resolution level, where the bytes are)
int wmh_launch_conv_bf16_ws(int cfg, const ConvArgs& a, hipStream_t st, int in16, int out16) {
    if (in16 || out16) {
        if (cfg != 0) return wmh_fail(WMH_EINVAL, "conv3d_k3 bf16: bf16 storage is implemented for config 0 only");
        return in16 ? (out16 ? launch_bf16_ws<WBCfg0, true, true>(a, st) : launch_bf16_ws<WBCfg0, true, false>(a, st))
                    : launch_bf16_ws<WBCfg0, false, true>(a, st);
    }
    if (cfg == 21) return launch_bf16_ws<WBCfgW>(a, st);
    return cfg == 0 ? launch_bf16_ws<WBCfg0>(a, st) : launch_bf16_ws<WBCfg2>(a, st);
}

// f16x2: the same geometry, chunks of 8 input channels (a: prepared by wmh_conv3d_k3_fwd_f16x2)
int wmh_launch_conv_f16x2_ws(int cfg, const ConvArgs& a, hipStream_t st) {
    if (cfg == 21) return launch_bf16_ws<WBCfgW, false, false, true>(a, st);
    if (cfg == 0) return launch_bf16_ws<WBCfg0, false, false, true>(a, st);
    if (cfg == 2) return launch_bf16_ws<WBCfg2, false, false, true>(a, st);
    if (cfg == 24) return launch_bf16_ws<WBCfg8, false, false, true>(a, st);
    return wmh_fail(WMH_EINVAL, "conv3d_k3 f16x2: configurations 0, 2, 21 and 24");
}
