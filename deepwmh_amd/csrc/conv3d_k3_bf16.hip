// conv3d_k3_bf16.hip -- the 3x3x3 convolution with bf16 matrix-core operands (BASELINE.json config 3,
// "bf16 forward"): storage stays fp32 NCDHW and the epilogue (bias, fp32 InstanceNorm moments, stores) is the
// fp32 kernel's, but the operands are rounded to bf16 on their way into LDS and the product runs on
// v_mfma_f32_32x32x16_bf16 (fp32 accumulate; 8x the fp32 instruction's rate per k):
//
//   D[cout 32][voxel 32] += A[cout 32][16 cin of one tap] * B[16 cin][voxel 32]
//
// * LDS tile is channel-interleaved, [voxel][16 cin] bf16 = two 16-byte halves per voxel, so a lane's B operand
//   (8 consecutive cin of its voxel) is ONE ds_read_b128.  The half index is XOR-ed with bit 3 of the voxel's x
//   so that the 16 lanes of a ds_read_b128 group hit 16 different 16-byte slots (conflict-free for all three kw).
// * Staging gathers a voxel's 16 channels (16 coalesced dword loads across the wave), applies the producer's
//   InstanceNorm+LeakyReLU in fp32, converts with round-to-nearest-even and writes two 16-byte pieces.
// * Weights: fragment-major bf16 image [G][cin/16][tap][NT][lane][8], one 16-byte load per lane per tap.
// The kernel is latency/bandwidth bound, not MFMA bound (the matrix work is 1/8 of the fp32 kernel's).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include "conv3d_common.h"

typedef short bf16x8 __attribute__((ext_vector_type(8)));

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {     // v_cvt_pk_bf16_f32: round to nearest even
    const f32x2_t v = {lo, hi};
    const bf16x2_t b = __builtin_convertvector(v, bf16x2_t);
    return __builtin_bit_cast(unsigned, b);
}

// PAIR: the f16x2 form (conv3d_k3_bf16_ws.hip): chunks of 8 channels staged as [hi | lo] bf16 pairs, every tap issued with the
// high and with the low parts of the weights (a fragment's lanes 0-31 / 32-63)
template <class C, int RIT, bool PAIR = false>
__global__ __launch_bounds__(C::THREADS, (C::THREADS > 256 ? 4 : 2)) void conv3d_k3_bf16_kernel(const ConvArgs a) {
    constexpr int S = C::S, MT = C::MT, NT = C::NT, NIT = C::NIT, THREADS = C::THREADS, APF = 4, CKB = PAIR ? 8 : 16;
    static_assert(C::TN == 1, "one sample per workgroup");
    extern __shared__ __attribute__((aligned(16))) uint4 ldsq[];      // [CS voxels][2]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, kh = lane >> 5;
    const int dx = j % C::LX, dy = (j / C::LX) % C::LY, dz = j / (C::LX * C::LY);
    int bid = blockIdx.x;
    const int abl = a.xcd_swizzle >> 4;             // diagnostics: 1 skip MFMA phase, 2 skip epilogue, 4 all channels read plane 0
    if (a.xcd_swizzle & 1) {
        const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int ks = bid % a.KS; bid /= a.KS;
    const int g = bid % a.G;
    int b = bid / a.G;
    const int tx = b % a.ntx; b /= a.ntx;
    const int ty = b % a.nty; b /= a.nty;
    const int tz = b % a.ntz; b /= a.ntz;
    const int n0 = b;
    const int ox0 = tx * C::TX, oy0 = ty * C::TY, oz0 = tz * C::TZ;
    const int ix0 = ox0 * S - 1, iy0 = oy0 * S - 1, iz0 = oz0 * S - 1;
    const int HW = a.H * a.W;
    const size_t DHW = (size_t)a.D * HW;
    const int ch_lo = (int)(((long long)a.nchunks * ks) / a.KS);
    const int ch_hi = (int)(((long long)a.nchunks * (ks + 1)) / a.KS);

    int sp_off[NIT];
    unsigned okmask = 0, keymask = 0;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int e = tid + it * THREADS;
        const int x = e % C::IX, y = (e / C::IX) % C::IY, z = (e / (C::IX * C::IY)) % C::IZ;
        const int gx = ix0 + x, gy = iy0 + y, gz = iz0 + z;
        const bool ok = (e < C::SP) && (n0 < a.N) && gx >= 0 && gx < a.W && gy >= 0 && gy < a.H && gz >= 0 && gz < a.D;
        sp_off[it] = ok ? gz * HW + gy * a.W + gx : 0;
        okmask |= ok ? (1u << it) : 0u;
        keymask |= ((x >> 3) & 1) ? (1u << it) : 0u;
    }
    const int ts0 = (wave * MT) % C::MPS;
    // byte address of this lane's B operand for (m, kw) at tap row (kd, kh_) = badr[kw][m] + (kd*ZS + kh_*YS)*32
    int badr[3][MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int ts = ts0 + m;
        const int tyi = ts % C::MY, tzi = ts / C::MY;
        const int e0 = S * ((tzi * C::LZ + dz) * C::ZS + (tyi * C::LY + dy) * C::YS + dx);
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) badr[kw][m] = (e0 + kw) * 32 + 16 * (kh ^ (((S * dx + kw) >> 3) & 1));
    }

    f32x16 acc[NT][MT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = (g * NT + nt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            const float bv = (a.KS == 1 && a.bias != nullptr && co < a.Cout) ? a.bias[co] * (PAIR ? WMH_PAIR_WSCALE : 1.f) : 0.f;
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[nt][m][r] = bv;
        }

    // PAIR: a fragment of the image holds the high parts of its 32 rows on lanes 0-31 and the low parts on lanes 32-63.  ONE 16-byte
    // load per lane fetches both; v_permlane32_swap_b32 turns it into the two operands [w_hi | w_hi] and [w_lo | w_lo] when the tap is
    // issued.  (Loading the two halves separately, each into all 64 lanes, doubled the bytes through the vector-memory pipe: with
    // one voxel tile per wave the strided 32-row configuration moved 432 KB of weight fragments per chunk and CU at that pipe's
    // 64 bytes per clock -- twice its MFMA time.)
    const bf16x8* wl = reinterpret_cast<const bf16x8*>(a.wf) + (size_t)g * a.nchunks * 27 * NT * 64 + lane;
    bf16x8 aring[APF][NT];
    // fragments of step `st` (0..26, or beyond: the next chunk's) of chunk `chn`; the steps walk the taps in the order (kd, kw, kr)
    // (wmh_tap_of: the wave-specialised kernel's order).  Past the last chunk: the last fragment again (never used).
    auto loadA = [&](int chn, int st, bf16x8 (&dst)[NT]) {
        if (st >= 27) { st -= 27; ++chn; }
        const bool past = chn >= ch_hi;
        const int T = (past ? ch_hi - 1 : chn) * 27 + (past ? 26 : wmh_tap_of(st));
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) dst[nt] = wl[((size_t)T * NT + nt) * 64];
    };
#pragma unroll
    for (int d = 0; d < APF; ++d) loadA(ch_lo, d, aring[d]);

    const char* ldsb = reinterpret_cast<const char*>(ldsq);
    unsigned long long t_stage = 0, t_mfma = 0, t_mark = 0;          // diagnostics (WMH_STAMPS(a) != NULL only)
    for (int ch = ch_lo; ch < ch_hi; ++ch) {
        if (WMH_STAMPS(a)) t_mark = __builtin_amdgcn_s_memtime();
        // ---- stage 16 input channels of the halo tile -------------------------------------------------------
        const int cb = ch * CKB;
        const float* xs; const float* ss; int Cs, c0;
        if (cb < a.C0) { xs = a.x0; ss = a.ss0; Cs = a.C0; c0 = cb; }
        else           { xs = a.x1; ss = a.ss1; Cs = a.C1; c0 = cb - a.C0; }
        const bool has_ss = ss != nullptr;
        const float slope_eff = has_ss ? a.slope : 1.f;
        // per-channel scale / shift of the producer's InstanceNorm (1, 0 without one; 0, 0 for channels past the source's
        // end: they stage zeros).  Uniform over the workgroup: scalar registers.
        float scv[CKB], shv[CKB];
#pragma unroll
        for (int c = 0; c < CKB; ++c) {
            const bool cok = (c0 + c) < Cs;
            const int cc = min(c0 + c, Cs - 1);
            scv[c] = cok ? (has_ss ? ss[((size_t)n0 * Cs + cc) * 2] : 1.f) : 0.f;
            shv[c] = (cok && has_ss) ? ss[((size_t)n0 * Cs + cc) * 2 + 1] : 0.f;
        }
        if (ch != ch_lo) __syncthreads();                 // every wave is done reading the previous chunk
#pragma unroll
        for (int r0 = 0; r0 < NIT; r0 += RIT) {
            float v[RIT][CKB];
#pragma unroll
            for (int ii = 0; ii < RIT; ++ii)
#pragma unroll
                for (int c = 0; c < CKB; ++c) {
                    const int it = r0 + ii;
                    if (it < NIT) v[ii][c] = xs[((size_t)n0 * Cs + ((abl & 4) ? 0 : min(c0 + c, Cs - 1))) * DHW + sp_off[it]];
                }
#pragma unroll
            for (int ii = 0; ii < RIT; ++ii) {
                const int it = r0 + ii;
                if (it >= NIT) continue;
                const unsigned keep = ((okmask >> it) & 1u) ? 0xFFFFFFFFu : 0u;      // halo voxels outside the volume: zeros
                unsigned w[8];
#pragma unroll
                for (int c = 0; c < CKB; c += 2) {
                    // LeakyReLU as max(t, slope * t) (0 <= slope <= 1, checked by the host); v_cvt_pk_bf16_f32 rounds to nearest even
                    float t0 = fmaf(v[ii][c], scv[c], shv[c]);
                    float t1 = fmaf(v[ii][c + 1], scv[c + 1], shv[c + 1]);
                    t0 = fmaxf(t0, t0 * slope_eff);
                    t1 = fmaxf(t1, t1 * slope_eff);
                    if constexpr (PAIR) {                    // fp16 pair (conv3d_k3_bf16_ws.hip): words 0-3 high parts (toward zero), 4-7 the remainders (nearest)
                        const auto h2 = __builtin_amdgcn_cvt_pkrtz(t0, t1);
                        w[c >> 1] = __builtin_bit_cast(unsigned, h2) & keep;
                        typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
                        typedef float f32x2_t __attribute__((ext_vector_type(2)));
                        const f32x2_t lo2 = {t0 - (float)h2[0], t1 - (float)h2[1]};
                        w[4 + (c >> 1)] = __builtin_bit_cast(unsigned, __builtin_convertvector(lo2, f16x2_t)) & keep;
                    } else {
                        w[c >> 1] = pack_bf16x2(t0, t1) & keep;
                    }
                }
                const int e = tid + it * THREADS;
                const int key = (keymask >> it) & 1u;
                ldsq[e * 2 + (0 ^ key)] = make_uint4(w[0], w[1], w[2], w[3]);
                ldsq[e * 2 + (1 ^ key)] = make_uint4(w[4], w[5], w[6], w[7]);
            }
        }
        __syncthreads();
        if (WMH_STAMPS(a)) { const unsigned long long t = __builtin_amdgcn_s_memtime(); t_stage += t - t_mark; t_mark = t; }
        // ---- 27 taps x MT x NT MFMAs -----------------------------------------------------------------------------
        // The B operands of step + 1 are read BEFORE the MFMAs of step are issued (sched_barrier pins them): a wave issues in order and
        // an MFMA holds the issue until the pipe takes it, so a read placed in front of its own use -- the compiler's schedule -- waits
        // out the LDS latency with the matrix pipe empty.  With one voxel tile per wave (the strided 32-row configuration) that was
        // 356 cycles per tap against 128 of MFMA issue.
        auto loadB = [&](int st, bf16x8 (&dst)[MT]) __attribute__((always_inline)) {
            const int tap = wmh_tap_of(st < 27 ? st : 26);
            const int kw = tap % 3;
            const int rowoff = ((tap / 9) * C::ZS + ((tap / 3) % 3) * C::YS) * 32;
#pragma unroll
            for (int m = 0; m < MT; ++m) dst[m] = *reinterpret_cast<const bf16x8*>(ldsb + badr[kw][m] + rowoff);
        };
        bf16x8 bbuf[2][MT];                                  // static indices: the step loop is fully unrolled
        if (!(abl & 1)) loadB(0, bbuf[0]);
        if (!(abl & 1))
#pragma unroll
        for (int step = 0; step < 27; ++step) {
            bf16x8 anew[NT];
            loadA(ch, step + APF, anew);
            if (step + 1 < 27) loadB(step + 1, bbuf[(step + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
            typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
            u32x4_t ahi[PAIR ? NT : 1], alo[PAIR ? NT : 1];
            if constexpr (PAIR) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const u32x4_t raw = __builtin_bit_cast(u32x4_t, aring[0][nt]);
#pragma unroll
                    for (int d = 0; d < 4; ++d) {                        // lanes 32-63 of r[0] <- lanes 0-31, lanes 0-31 of r[1] <- lanes 32-63
                        const auto r = __builtin_amdgcn_permlane32_swap(raw[d], raw[d], false, false);
                        ahi[nt][d] = r[0]; alo[nt][d] = r[1];
                    }
                }
            }
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const bf16x8 bv = bbuf[step & 1][m];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    if constexpr (PAIR) {
                        acc[nt][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ahi[nt]), __builtin_bit_cast(f16x8, bv), acc[nt][m], 0, 0, 0);
                        acc[nt][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, alo[nt]), __builtin_bit_cast(f16x8, bv), acc[nt][m], 0, 0, 0);
                    } else {
                        acc[nt][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aring[0][nt], bv, acc[nt][m], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int d = 0; d + 1 < APF; ++d)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) aring[d][nt] = aring[d + 1][nt];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) aring[APF - 1][nt] = anew[nt];
            __builtin_amdgcn_sched_barrier(0);
        }
        if (WMH_STAMPS(a)) { const unsigned long long t = __builtin_amdgcn_s_memtime(); t_mfma += t - t_mark; }
    }
    if (WMH_STAMPS(a)) t_mark = __builtin_amdgcn_s_memtime();
    if constexpr (PAIR) {                                   // undo the weights' 2^8 (exact; split-K slabs hold unscaled partial sums)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[nt][m] *= (1.f / WMH_PAIR_WSCALE);
    }
    if (!(abl & 2)) conv_epilogue<C>(a, acc, g, ks, n0, wave, j, kh, dx, dy, dz, ts0, oz0, oy0, ox0, (tz * a.nty + ty) * a.ntx + tx);
    if (WMH_STAMPS(a) && tid == 0) {
        unsigned long long* o = WMH_STAMPS(a) + (size_t)blockIdx.x * 4;
        o[0] = t_stage; o[1] = t_mfma; o[2] = __builtin_amdgcn_s_memtime() - t_mark; o[3] = (unsigned long long)(ch_hi - ch_lo);
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------
extern unsigned long long* g_conv_stamps;
int wmh_launch_conv_bf16_ws(int cfg, const ConvArgs& a, hipStream_t st, int in16, int out16);      // conv3d_k3_bf16_ws.hip
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
typedef ConvCfg<1, 32, 1, 4, 8, 1, 8, 1, 4, 6> BCfg0;      // same tile geometry as the fp32 Cfg0 / Cfg2
typedef ConvCfg<1, 16, 2, 2, 16, 1, 4, 2, 4, 6> BCfg2;
// The kernel is bound by memory latency, not by the matrix cores: the same tiles cut into 8 waves of 4 (cfg 0) / 2 (cfg 2)
// M-tiles halve the accumulator registers, so two workgroups = 16 waves fit a CU and twice as many loads are in flight.
typedef ConvCfg<1, 32, 1, 4, 8, 1, 4, 1, 4, 6> BCfg0w;
typedef ConvCfg<1, 16, 2, 2, 16, 1, 2, 2, 4, 6> BCfg2w;
static int bf16_variant() { static const int v = wmh_diag_env("WMH_BF16_WAVES8", 0); return v; }

static inline unsigned short f2bf(float f) {
    unsigned u; memcpy(&u, &f, 4);
    return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
// Configuration 21 exists for this kernel only: the 32-wide stride-1 geometry with 64 output channels per tile
// (4 x 4 x 32 voxels, NT = 2).  A 32-channel tile re-stages the input once per 32 output channels, and the loaders are what
// bounds this kernel: with 64 or more output channels the wide tile halves their work per output (halo 1440 voxels per
// 512 x 64 outputs against 2400 per 1024 x 32).  Served by the wave-specialised kernel only.
#define WMH_BF16_CFG_WIDE 21
typedef ConvCfg<1, 32, 1, 4, 4, 1, 4, 2, 4, 6> BCfgW;
// The levels below 16-voxel rows and the strided convolutions into them (inference with bf16 operands: they hold ~2 ms of
// fp32 matrix work per batch-8 forward of the BASELINE network, 1/8 of it on the bf16 instruction): the one-role kernel on
// the fp32 kernel's tile geometries (configuration 3 with half the rows per tile: its stride-2 halo tile must fit LDS as
// [voxel][16 channels]).  Configurations 6 / 7 (4-voxel rows, four samples per workgroup) stay on the fp32 instruction.
typedef ConvCfg<2, 16, 2, 2, 8, 1, 2, 2, 2, 6> BCfg3;     // stride 2, 16-wide output, 2 x 8 x 16 outputs x 64 channels
typedef ConvCfg<1, 8, 4, 4, 8, 1, 2, 1, 8, 6> BCfg4;      // stride 1, 8-wide
typedef ConvCfg<2, 8, 4, 4, 8, 1, 2, 1, 2, 6> BCfg5;      // stride 2, 8-wide output
// Stride 2 with output rows of 32 voxels and more (the first convolution of the 64^3 and 32^3 stages; f16x2 mode only: round 3): tile
// geometry only -- the kernel is the wave-specialised one of conv3d_k3s2_f16x2_ws.hip (a 2 x 2 x 32 output tile x 64 channels reads a
// 5 x 5 x 65 input box; the one-role form that first served these layers spent half its time in its in-order staging).
typedef ConvCfg<2, 32, 1, 2, 2, 1, 1, 2, 2, 6> BCfg1;
// Configuration 24 (f16x2 only, round 5): rows of 8 voxels on the wave-specialised kernel -- a tile is a whole 8 x 8 x 8 block x 64 output
// channels, split-K slices fill the chip (the 8^3 level: 320 channels, 16 x 5 tiles).  The one-role kernel of configuration 4 ran these
// layers at 150-220 TFLOP/s (every chunk ends in a barrier that nothing hides); it keeps the rows that are no multiple of 4 voxels.
#define WMH_F16X2_CFG_ROWS8 24
typedef ConvCfg<1, 8, 4, 8, 8, 1, 4, 2, 4, 6> BCfg8;
// Configurations 25 / 26 (f16x2 only, round 5): stride 2 with output rows of 16 / 8 voxels on the wave-specialised strided kernel
// (conv3d_k3s2_f16x2_ws.hip: tiles of 2 x 4 x 16 / 4 x 4 x 8 outputs x 64 channels) -- configuration 1's kernel on two more geometries.
// The one-role kernels of configurations 3 / 5 ran these layers at 120-170 TFLOP/s; they keep the rows that are no multiple of 4 voxels.
#define WMH_F16X2_CFG_S2_ROWS16 25
#define WMH_F16X2_CFG_S2_ROWS8 26
static bool f16x2_s2_ws(int cfg) { return cfg == 1 || cfg == WMH_F16X2_CFG_S2_ROWS16 || cfg == WMH_F16X2_CFG_S2_ROWS8; }
static int bf16_nt(int cfg) {
    return (cfg == 0 || cfg == 4 || cfg == 5) ? 1
         : (cfg == 1 || cfg == 2 || cfg == 3 || cfg == WMH_BF16_CFG_WIDE || cfg == WMH_F16X2_CFG_ROWS8 || cfg == WMH_F16X2_CFG_S2_ROWS16 || cfg == WMH_F16X2_CFG_S2_ROWS8) ? 2 : 0;
}
static int bf16_stride(int cfg) { return (cfg == 1 || cfg == 3 || cfg == 5 || cfg == WMH_F16X2_CFG_S2_ROWS16 || cfg == WMH_F16X2_CFG_S2_ROWS8) ? 2 : 1; }
struct BGeo { int TX, TY, TZ, WPS; };
static BGeo bf16_geo(int cfg, bool w8) {
    if (cfg == WMH_BF16_CFG_WIDE) return BGeo{BCfgW::TX, BCfgW::TY, BCfgW::TZ, BCfgW::WPS};
    if (cfg == WMH_F16X2_CFG_ROWS8) return BGeo{BCfg8::TX, BCfg8::TY, BCfg8::TZ, BCfg8::WPS};
    if (cfg == WMH_F16X2_CFG_S2_ROWS16) return BGeo{16, 4, 2, 4};      // (TX, TY, TZ, one record per consumer wave)
    if (cfg == WMH_F16X2_CFG_S2_ROWS8) return BGeo{8, 4, 4, 4};
    if (cfg == 1) return BGeo{BCfg1::TX, BCfg1::TY, BCfg1::TZ, BCfg1::WPS};
    if (cfg == 3) return BGeo{BCfg3::TX, BCfg3::TY, BCfg3::TZ, BCfg3::WPS};
    if (cfg == 4) return BGeo{BCfg4::TX, BCfg4::TY, BCfg4::TZ, BCfg4::WPS};
    if (cfg == 5) return BGeo{BCfg5::TX, BCfg5::TY, BCfg5::TZ, BCfg5::WPS};
    if (cfg == 0) return BGeo{BCfg0::TX, BCfg0::TY, BCfg0::TZ, w8 ? BCfg0w::WPS : BCfg0::WPS};
    return BGeo{BCfg2::TX, BCfg2::TY, BCfg2::TZ, w8 ? BCfg2w::WPS : BCfg2::WPS};
}
// The configuration wmh_conv3d_k3_fwd_bf16 should be called with, given the one wmh_conv3d_config chose for the layer
extern "C" int wmh_conv3d_k3_bf16_config(int cfg, int Cout, int W) {
    static const int wide = wmh_diag_env("WMH_BF16_WIDE", 1);
    static const int use_ws = wmh_diag_env("WMH_BF16_WS", 1);
    if (cfg == 0 && wide && use_ws && bf16_variant() == 0 && Cout % 64 == 0 && W % 4 == 0) return WMH_BF16_CFG_WIDE;
    return cfg;
}

// The configuration wmh_conv3d_k3_fwd_f16x2 should be called with: the one above, and rows of 8 voxels go to configuration 24 when the
// wave-specialised kernel can load them (aligned quads)
extern "C" int wmh_conv3d_k3_f16x2_config(int cfg, int Cout, int W) {
    static const int rows8 = wmh_diag_env("WMH_F16X2_ROWS8", 1);
    const int c = wmh_conv3d_k3_bf16_config(cfg, Cout, W);
    if (c == 4 && rows8 && W % 4 == 0) return WMH_F16X2_CFG_ROWS8;
    static const int s2ws = wmh_diag_env("WMH_F16X2_S2_WS", 1);
    if (c == 3 && s2ws && W % 4 == 0) return WMH_F16X2_CFG_S2_ROWS16;      // (W: the INPUT row)
    if (c == 5 && s2ws && W % 4 == 0) return WMH_F16X2_CFG_S2_ROWS8;
    return c;
}

extern "C" int wmh_conv3d_k3_bf16_supported(int cfg) { return !f16x2_s2_ws(cfg) && cfg != WMH_F16X2_CFG_ROWS8 && bf16_nt(cfg) > 0; }      // configurations 1, 24, 25, 26: f16x2 only
// InstanceNorm partial records per (n, c) row the bf16 kernel writes (its wave split differs from the fp32 kernel's)
extern "C" int wmh_conv3d_k3_num_partials_bf16(int cfg, int N, int Do, int Ho, int Wo) {
    if (!bf16_nt(cfg)) return wmh_fail(WMH_EINVAL, "num_partials_bf16: configuration has no bf16 kernel");
    (void)N;
    const bool w8 = bf16_variant() != 0 && (cfg == 0 || cfg == 2);
    const BGeo ge = bf16_geo(cfg, w8);
    return cdiv(Wo, ge.TX) * cdiv(Ho, ge.TY) * cdiv(Do, ge.TZ) * ge.WPS;
}
// ... and by wmh_conv3d_k3_fwd_f16x2, which always runs the four-consumer-wave geometry (WMH_BF16_WAVES8 does not apply to it)
extern "C" int wmh_conv3d_k3_num_partials_f16x2(int cfg, int N, int Do, int Ho, int Wo) {
    if (!bf16_nt(cfg)) return wmh_fail(WMH_EINVAL, "num_partials_f16x2: configuration has no f16x2 kernel");
    (void)N;
    const BGeo ge = bf16_geo(cfg, false);
    return cdiv(Wo, ge.TX) * cdiv(Ho, ge.TY) * cdiv(Do, ge.TZ) * ge.WPS;
}
extern "C" size_t wmh_conv3d_k3_packed_halfs_bf16(int cfg, int Cin, int Cout) {
    const int NT = bf16_nt(cfg);
    if (!NT) return 0;
    return (size_t)cdiv(Cout, 32 * NT) * cdiv(Cin, 16) * 27 * NT * 64 * 8;
}
// w: host fp32 torch layout [Cout][Cin][27]; wf: host uint16 out, [G][cin/16][tap][NT][lane][8]
extern "C" int wmh_conv3d_k3_pack_weights_bf16(int cfg, const float* w, int Cin, int Cout, uint16_t* wf) {
    const int NT = bf16_nt(cfg);
    if (!NT || !w || !wf) return wmh_fail(WMH_EINVAL, "pack_weights_bf16: unsupported cfg or null");
    const int G = cdiv(Cout, 32 * NT), NCH = cdiv(Cin, 16);
    for (int g = 0; g < G; ++g)
        for (int ch = 0; ch < NCH; ++ch)
            for (int tap = 0; tap < 27; ++tap)
                for (int nt = 0; nt < NT; ++nt)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 8; ++e) {
                            const int co = (g * NT + nt) * 32 + (lane & 31), ci = ch * 16 + 8 * (lane >> 5) + e;
                            float v = 0.f;
                            if (co < Cout && ci < Cin) v = w[((size_t)co * Cin + ci) * 27 + tap];
                            wf[(((((size_t)g * NCH + ch) * 27 + tap) * NT + nt) * 64 + lane) * 8 + e] = f2bf(v);
                        }
    return WMH_OK;
}

// ---- the "pair" mode (entry points keep round 2's f16x2 names; conv3d_k3_bf16_ws.hip, PAIR): chunks of 8 input channels, a lane's
// 16 bytes are the HIGH fp16 parts of its output channel's 8 weights (lanes 0-31) or the LOW parts (lanes 32-63), of the
// weights scaled by 2^8: 256 w = hi + lo + O(2^-22 w).  |w| must stay below 255 (fp16's range after the scaling).
extern "C" int wmh_conv3d_k3_f16x2_supported(int cfg) { return cfg == 0 || cfg == 1 || cfg == 2 || cfg == WMH_BF16_CFG_WIDE || cfg == WMH_F16X2_CFG_ROWS8 || cfg == 3 || cfg == 4 || cfg == 5 || cfg == WMH_F16X2_CFG_S2_ROWS16 || cfg == WMH_F16X2_CFG_S2_ROWS8; }
extern "C" size_t wmh_conv3d_k3_packed_halfs_f16x2(int cfg, int Cin, int Cout) {
    const int NT = bf16_nt(cfg);
    if (!NT || !wmh_conv3d_k3_f16x2_supported(cfg)) return 0;
    return (size_t)cdiv(Cout, 32 * NT) * cdiv(Cin, 8) * 27 * NT * 64 * 8;
}
static inline uint16_t f2h(float v) { const _Float16 h = (_Float16)v; uint16_t u; memcpy(&u, &h, 2); return u; }     // nearest even
static inline float h2f(uint16_t u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; }
// w: host fp32 torch layout [Cout][Cin][27]; wf: host uint16 out, [G][cin/8][tap][NT][lane][8]
extern "C" int wmh_conv3d_k3_pack_weights_f16x2(int cfg, const float* w, int Cin, int Cout, uint16_t* wf) {
    const int NT = bf16_nt(cfg);
    if (!NT || !wmh_conv3d_k3_f16x2_supported(cfg) || !w || !wf) return wmh_fail(WMH_EINVAL, "pack_weights_f16x2: unsupported cfg or null");
    const int G = cdiv(Cout, 32 * NT), NCH = cdiv(Cin, 8);
    for (int g = 0; g < G; ++g)
        for (int ch = 0; ch < NCH; ++ch)
            for (int tap = 0; tap < 27; ++tap)
                for (int nt = 0; nt < NT; ++nt)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 8; ++e) {
                            const int co = (g * NT + nt) * 32 + (lane & 31), ci = ch * 8 + e;
                            float v = 0.f;
                            if (co < Cout && ci < Cin) v = w[((size_t)co * Cin + ci) * 27 + tap] * WMH_PAIR_WSCALE;
                            if (!(v > -65000.f && v < 65000.f)) return wmh_fail(WMH_EINVAL, "pack_weights_f16x2: a weight is not finite or exceeds +-253 (fp16 pair range)");
                            const uint16_t hi = f2h(v);
                            wf[(((((size_t)g * NCH + ch) * 27 + tap) * NT + nt) * 64 + lane) * 8 + e] = (lane >> 5) ? f2h(v - h2f(hi)) : hi;
                        }
    return WMH_OK;
}

// Device-side repack (training: the weights change every step).  Same argument meaning as
// wmh_conv3d_k3_pack_weights_device: w = torch layout [Cout][Cin_total][27] fp32 in HBM, channels [ci_off, ci_off+ci_cnt);
// mode 0 = forward image (ci_cnt -> Cout), mode 1 = data-gradient image (Cout -> ci_cnt, transposed, taps flipped).
__global__ __launch_bounds__(256) void pack_bf16_kernel(const float* w, uint16_t* wf, int Cin_total, int Cout, int ci_off, int ci_cnt, int mode,
                                                         int NT, int NCH, size_t total) {
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        size_t r = idx;
        const int e = (int)(r % 8); r /= 8;
        const int lane = (int)(r % 64); r /= 64;
        const int nt = (int)(r % NT); r /= NT;
        const int tap = (int)(r % 27); r /= 27;
        const int ch = (int)(r % NCH);
        const int g = (int)(r / NCH);
        const int eco = (g * NT + nt) * 32 + (lane & 31), eci = ch * 16 + 8 * (lane >> 5) + e;      // effective conv's (cout, cin)
        float v = 0.f;
        if (mode == 0) { if (eco < Cout && eci < ci_cnt) v = w[((size_t)eco * Cin_total + ci_off + eci) * 27 + tap]; }
        else           { if (eco < ci_cnt && eci < Cout) v = w[((size_t)eci * Cin_total + ci_off + eco) * 27 + (26 - tap)]; }
        const unsigned u = __float_as_uint(v);
        wf[idx] = (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
    }
}
extern "C" int wmh_conv3d_k3_pack_weights_bf16_device(int cfg, const float* w, int Cin_total, int Cout, int ci_off, int ci_cnt, int mode,
                                                      uint16_t* wf, void* stream) {
    const int NT = bf16_nt(cfg);
    if (!NT || !w || !wf || ci_off < 0 || ci_cnt < 1 || ci_off + ci_cnt > Cin_total || (mode != 0 && mode != 1))
        return wmh_fail(WMH_EINVAL, "pack_weights_bf16_device: bad args or configuration without a bf16 kernel");
    const int eCin = mode == 0 ? ci_cnt : Cout, eCout = mode == 0 ? Cout : ci_cnt;
    const int NCH = cdiv(eCin, 16);
    const size_t total = (size_t)cdiv(eCout, 32 * NT) * NCH * 27 * NT * 64 * 8;
    unsigned nb = (unsigned)((total + 255) / 256); if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(pack_bf16_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, w, wf, Cin_total, Cout, ci_off, ci_cnt, mode, NT, NCH, total);
    return wmh_check_launch("pack_bf16_kernel");
}

// The f16x2 image on the device (training: weights change every step): modes as above, chunks of 8 input channels,
// high parts on lanes 0-31 and low parts on lanes 32-63 of every fragment.
// il: the interleaved image of the strided 32-row configuration (a lane holds 4 channels x (hi, lo); conv3d_k3s2_f16x2_ws.hip)
struct PackJobF16x2 {                    // one weight image to (re)build: 64 bytes, plain data (the batch entry point takes an array of them)
    const float* w; uint16_t* wf;
    int Cin_total, Cout, ci_off, ci_cnt, mode, NT, NCH, il;
    unsigned long long total, pad_;
};
static_assert(sizeof(PackJobF16x2) == 64, "PackJobF16x2 is part of the C ABI (wmh_conv3d_k3_pack_job_bytes)");
// One thread per output row and fragment: (g, chunk, tap, nt, co) -> the 8 input channels of the chunk, written as the two 16-byte
// pieces of lanes co and co + 32.  (One thread per 2-byte element -- eight divisions each, every weight fetched twice -- made a
// training step's image rebuild 0.37 ms: arithmetic, not bytes.)
__device__ __forceinline__ void pack_f16x2_body(const PackJobF16x2& J, size_t first, size_t stride) {
    const float* w = J.w;
    uint4* wf = reinterpret_cast<uint4*>(J.wf);
    const int Cin_total = J.Cin_total, Cout = J.Cout, ci_off = J.ci_off, ci_cnt = J.ci_cnt, mode = J.mode, NT = J.NT, NCH = J.NCH, il = J.il;
    const size_t items = J.total / 16;
    for (size_t item = first; item < items; item += stride) {
        size_t r = item;
        const int col = (int)(r % 32); r /= 32;
        const int nt = (int)(r % NT); r /= NT;
        const int tap = (int)(r % 27); r /= 27;
        const int ch = (int)(r % NCH);
        const int g = (int)(r / NCH);
        const int eco = (g * NT + nt) * 32 + col;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int eci = ch * 8 + e;
            float t = 0.f;
            if (mode == 0) { if (eco < Cout && eci < ci_cnt) t = w[((size_t)eco * Cin_total + ci_off + eci) * 27 + tap]; }
            else if (mode == 2) {                // w = ConvTranspose3d(k2, s2) weight [Cout][ci_cnt][2][2][2]: its 2-window embedded in the 3-window
                const int tz = tap / 9, ty = (tap / 3) % 3, tx = tap % 3;
                if (eco < Cout && eci < ci_cnt && tz && ty && tx) t = w[((size_t)eco * Cin_total + ci_off + eci) * 8 + (tz - 1) * 4 + (ty - 1) * 2 + (tx - 1)];
            }
            else           { if (eco < ci_cnt && eci < Cout) t = w[((size_t)eci * Cin_total + ci_off + eco) * 27 + (26 - tap)]; }
            v[e] = fminf(fmaxf(t * WMH_PAIR_WSCALE, -65000.f), 65000.f);      // (training: a weight beyond +-253 saturates instead of overflowing)
        }
        uint16_t lo_lane[8], hi_lane[8];         // the pieces of lane col and of lane col + 32
        if (il) {                                // strided 32-row configuration: a lane holds 4 channels x (hi, lo); lanes 32-63 channels 4-7
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float h = __uint_as_float(__float_as_uint(v[e]) & 0xFFFFE000u);
                const uint16_t hb = __builtin_bit_cast(uint16_t, (_Float16)h), lb = __builtin_bit_cast(uint16_t, (_Float16)(v[e] - h));
                if (e < 4) { lo_lane[2 * e] = hb; lo_lane[2 * e + 1] = lb; }
                else       { hi_lane[2 * (e - 4)] = hb; hi_lane[2 * (e - 4) + 1] = lb; }
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const _Float16 hi = (_Float16)v[e];
                lo_lane[e] = __builtin_bit_cast(uint16_t, hi);
                hi_lane[e] = __builtin_bit_cast(uint16_t, (_Float16)(v[e] - (float)hi));
            }
        }
        const size_t piece = ((((size_t)g * NCH + ch) * 27 + tap) * NT + nt) * 64 + col;
        wf[piece] = make_uint4(lo_lane[0] | (unsigned)lo_lane[1] << 16, lo_lane[2] | (unsigned)lo_lane[3] << 16, lo_lane[4] | (unsigned)lo_lane[5] << 16,
                               lo_lane[6] | (unsigned)lo_lane[7] << 16);
        wf[piece + 32] = make_uint4(hi_lane[0] | (unsigned)hi_lane[1] << 16, hi_lane[2] | (unsigned)hi_lane[3] << 16, hi_lane[4] | (unsigned)hi_lane[5] << 16,
                                    hi_lane[6] | (unsigned)hi_lane[7] << 16);
    }
}
__global__ __launch_bounds__(256) void pack_f16x2_kernel(const PackJobF16x2 J) {
    pack_f16x2_body(J, (size_t)blockIdx.x * 256 + threadIdx.x, (size_t)gridDim.x * 256);
}
// every image of a training step in ONE launch: blockIdx.y = job (the step's 43 single launches were 0.6 ms of a 31 ms iteration)
__global__ __launch_bounds__(256) void pack_f16x2_batch_kernel(const PackJobF16x2* jobs) {
    const PackJobF16x2 J = jobs[blockIdx.y];
    pack_f16x2_body(J, (size_t)blockIdx.x * 256 + threadIdx.x, (size_t)gridDim.x * 256);
}
static int pack_job_f16x2(int cfg, const float* w, int Cin_total, int Cout, int ci_off, int ci_cnt, int mode, uint16_t* wf, PackJobF16x2& J) {
    const int NT = bf16_nt(cfg);
    if (!NT || !wmh_conv3d_k3_f16x2_supported(cfg) || !w || !wf || ci_off < 0 || ci_cnt < 1 || ci_off + ci_cnt > Cin_total || (mode != 0 && mode != 1 && mode != 2))
        return wmh_fail(WMH_EINVAL, "pack_weights_f16x2_device: bad args or configuration without a f16x2 kernel");
    if (f16x2_s2_ws(cfg) && mode == 1) return wmh_fail(WMH_EINVAL, "pack_weights_f16x2_device: configurations 1, 25, 26 (stride 2) have no data-gradient image");
    if (mode == 2 && bf16_stride(cfg) != 2) return wmh_fail(WMH_EINVAL, "pack_weights_f16x2_device: mode 2 (transposed convolution's data gradient) is a stride-2 image");
    const int eCin = mode == 1 ? Cout : ci_cnt, eCout = mode == 1 ? ci_cnt : Cout;
    J.w = w; J.wf = wf; J.Cin_total = Cin_total; J.Cout = Cout; J.ci_off = ci_off; J.ci_cnt = ci_cnt; J.mode = mode; J.NT = NT;
    J.NCH = cdiv(eCin, 8); J.il = 0;      // (il: the interleaved (hi | lo << 16) image of the round-3/4 strided kernel; since round 5 configuration 1 takes the common image)
    J.total = (unsigned long long)cdiv(eCout, 32 * NT) * J.NCH * 27 * NT * 64 * 8; J.pad_ = 0;
    return WMH_OK;
}
extern "C" int wmh_conv3d_k3_pack_weights_f16x2_device(int cfg, const float* w, int Cin_total, int Cout, int ci_off, int ci_cnt, int mode,
                                                        uint16_t* wf, void* stream) {
    PackJobF16x2 J;
    const int rc = pack_job_f16x2(cfg, w, Cin_total, Cout, ci_off, ci_cnt, mode, wf, J);
    if (rc != WMH_OK) return rc;
    unsigned nb = (unsigned)((J.total / 16 + 255) / 256); if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(pack_f16x2_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, J);
    return wmh_check_launch("pack_f16x2_kernel");
}
extern "C" size_t wmh_conv3d_k3_pack_job_bytes(void) { return sizeof(PackJobF16x2); }
extern "C" int wmh_conv3d_k3_pack_job_f16x2(int cfg, const float* w, int Cin_total, int Cout, int ci_off, int ci_cnt, int mode, uint16_t* wf, void* job_out) {
    if (!job_out) return wmh_fail(WMH_EINVAL, "pack_job_f16x2: no job record");
    return pack_job_f16x2(cfg, w, Cin_total, Cout, ci_off, ci_cnt, mode, wf, *static_cast<PackJobF16x2*>(job_out));
}
extern "C" int wmh_conv3d_k3_pack_weights_f16x2_batch(const void* jobs_device, int njobs, void* stream) {
    if (!jobs_device || njobs < 1 || njobs > 65535) return wmh_fail(WMH_EINVAL, "pack_weights_f16x2_batch: bad args");
    // 512 x 256 threads per image, one 2 x 16-byte fragment row per thread and step: the largest images (320 x 320 x 27: 346 k rows) take
    // three steps, the small ones' surplus workgroups leave at once
    hipLaunchKernelGGL(pack_f16x2_batch_kernel, dim3(512, (unsigned)njobs), dim3(256), 0, (hipStream_t)stream, static_cast<const PackJobF16x2*>(jobs_device));
    return wmh_check_launch("pack_f16x2_batch_kernel");
}

template <class C, int RIT, bool PAIR = false>
static int launch_bf16(const ConvArgs& a, unsigned nblk, hipStream_t st) {
    const int lds = C::CS * 32;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_k3_bf16_kernel<C, RIT, PAIR>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((conv3d_k3_bf16_kernel<C, RIT, PAIR>), dim3(nblk), dim3(C::THREADS), lds, st, a);
    return wmh_check_launch("conv3d_k3_bf16_kernel");
}

// split-K factor of the small-row configurations (3, 4, 5: rows of 16 / 8 voxels, the 16^3 and 8^3 levels at batch 8): their
// grids are one workgroup per CU or less and every chunk iteration ends in a barrier, so nothing hides the staging latency;
// with the channel reduction cut into slabs several workgroups share a CU.  Same rule as the fp32 path's pick_ksplit.
static int bf16_ksplit(long long nblk, int nchunks) {
    static const int force = wmh_diag_env("WMH_KSPLIT", 0);
    if (force > 0) return force < nchunks ? force : nchunks;
    if (nblk >= 512) return 1;
    int ks = (int)((1024 + nblk - 1) / nblk);
    if (ks > 16) ks = 16;
    if (ks > nchunks / 2) ks = nchunks / 2;
    return ks < 1 ? 1 : ks;
}
static long long bf16_small_blocks(int cfg, int N, int D, int H, int W, int Cout) {
    const int S = bf16_stride(cfg), NT = bf16_nt(cfg);
    const BGeo ge = bf16_geo(cfg, false);
    const int Do = (D - 1) / S + 1, Ho = (H - 1) / S + 1, Wo = (W - 1) / S + 1;
    return (long long)cdiv(Wo, ge.TX) * cdiv(Ho, ge.TY) * cdiv(Do, ge.TZ) * N * cdiv(Cout, 32 * NT);
}
extern "C" size_t wmh_conv3d_k3_bf16_workspace_floats(int cfg, int Cin, int N, int D, int H, int W, int Cout) {
    if (!(cfg == 3 || cfg == 4 || cfg == 5) || !bf16_nt(cfg)) return 0;
    const int S = bf16_stride(cfg);
    const int ks = bf16_ksplit(bf16_small_blocks(cfg, N, D, H, W, Cout), cdiv(Cin, 16));
    return ks > 1 ? (size_t)ks * N * Cout * ((D - 1) / S + 1) * ((H - 1) / S + 1) * ((W - 1) / S + 1) : 0;
}

static int conv_fwd_bf16(int cfg, const float* x0, const float* ss0, int C0, const float* x1,
                         const float* ss1, int C1, const uint16_t* wf, const float* bias, float* y,
                         float* partials, int N, int D, int H, int W, int Cout, float lrelu_slope,
                         int in16, int out16, void* stream, float* workspace = nullptr) {
    const int NT = bf16_nt(cfg);
    if (!NT || !wmh_conv3d_k3_bf16_supported(cfg)) return wmh_fail(WMH_EINVAL, "conv3d_k3_fwd_bf16: configuration has no bf16 kernel");
    if (!x0 || !wf || !y || C0 < 1 || N < 1 || Cout < 1) return wmh_fail(WMH_EINVAL, "conv3d_k3_fwd_bf16: null/empty argument");
    if (x1 == nullptr) C1 = 0;
    if (!(lrelu_slope >= 0.f && lrelu_slope <= 1.f)) return wmh_fail(WMH_EINVAL, "conv3d_k3_fwd_bf16: LeakyReLU slope must be in [0, 1]");
    if (C1 > 0 && (C0 % 16) != 0) return wmh_fail(WMH_EINVAL, "conv3d_k3_fwd_bf16: C0 must be a multiple of 16 with two sources");
    ConvArgs a;
    a.x0 = x0; a.ss0 = ss0; a.C0 = C0; a.x1 = x1; a.ss1 = ss1; a.C1 = C1;
    a.wf = reinterpret_cast<const float*>(wf); a.bias = bias; a.y = y; a.part = partials;
    const int S = bf16_stride(cfg);                 // configurations 3 and 5 are the stride-2 ones: (D, H, W) is the INPUT extent
    a.N = N; a.D = D; a.H = H; a.W = W; a.Do = (D - 1) / S + 1; a.Ho = (H - 1) / S + 1; a.Wo = (W - 1) / S + 1; a.Cout = Cout;
    a.nchunks = (C1 > 0) ? (C0 / 16 + cdiv(C1, 16)) : cdiv(C0, 16);
    const bool small = cfg == 3 || cfg == 4 || cfg == 5;
    const bool w8 = bf16_variant() != 0 && !small;
    const BGeo ge = bf16_geo(cfg, w8);
    a.ntx = cdiv(a.Wo, ge.TX); a.nty = cdiv(a.Ho, ge.TY); a.ntz = cdiv(a.Do, ge.TZ);
    a.npart = a.ntx * a.nty * a.ntz * ge.WPS;
    if ((size_t)a.Do * a.Ho * a.Wo * 4 * 32 >= ((size_t)1 << 32)) return wmh_fail(WMH_EINVAL, "conv3d_k3: 32 output channel planes must stay below 4 GB (epilogue lane offsets are 32-bit)");
    a.slope = lrelu_slope;
    a.G = cdiv(Cout, 32 * NT);
    a.KS = 1;
    static const int swz_env = wmh_diag_env("WMH_XCD_SWIZZLE", 1);
    static const int abl_env = wmh_diag_env("WMH_BF16_ABL", 0);
    a.xcd_swizzle = (swz_env & 1) | (abl_env << 4);
    a.stamps = g_conv_stamps;
    a.by = a.bz = 1;
    const long long nblk = (long long)a.ntx * a.nty * a.ntz * N * a.G;
    if (nblk > 0x7fffffffLL) return wmh_fail(WMH_EINVAL, "conv3d_k3_fwd_bf16: grid too large");
    hipStream_t st = (hipStream_t)stream;
    static const int use_ws = wmh_diag_env("WMH_BF16_WS", 1);
    // the wave-specialised form loads aligned 16-byte quads: rows must start 16-byte aligned
    const bool quad_ok = (W % 4) == 0 && (size_t)16 * D * H * W * 4 < ((size_t)1 << 32) && (reinterpret_cast<uintptr_t>(x0) & 15) == 0 && (!x1 || (reinterpret_cast<uintptr_t>(x1) & 15) == 0);
    if (small) {
        if (in16 || out16) return wmh_fail(WMH_EINVAL, "conv3d_k3_fwd_bf16_st: bf16 storage is served for configurations 0 and 21");
        a.KS = (workspace != nullptr && cfg != 1) ? bf16_ksplit(nblk, a.nchunks) : 1;
        if (a.KS > 1) a.y = workspace;                       // slabs [KS][N][Cout][Do][Ho][Wo]; bias and moments come with the reduce
        const unsigned nb = (unsigned)(nblk * a.KS);
        int rc;
        if (cfg == 3) rc = launch_bf16<BCfg3, 4>(a, nb, st);
        else if (cfg == 4) rc = launch_bf16<BCfg4, 3>(a, nb, st);
        else rc = launch_bf16<BCfg5, 4>(a, nb, st);
        if (rc != WMH_OK || a.KS == 1) return rc;
        return wmh_launch_splitk_reduce(workspace, bias, y, partials, a.npart, a.KS, N, Cout, (size_t)a.Do * a.Ho * a.Wo, st);
    }
    if (in16 || out16) {                          // bf16 tensors: only the wave-specialised kernel reads / writes them
        if (!quad_ok) return wmh_fail(WMH_EINVAL, "conv3d_k3_fwd_bf16_st: bf16 storage needs W % 4 == 0 and 16-byte aligned sources");
        return wmh_launch_conv_bf16_ws(cfg, a, st, in16, out16);
    }
    if (cfg == WMH_BF16_CFG_WIDE) {
        if (!quad_ok) return wmh_fail(WMH_EINVAL, "conv3d_k3_fwd_bf16: configuration 21 needs W % 4 == 0 and 16-byte aligned sources");
        return wmh_launch_conv_bf16_ws(cfg, a, st, 0, 0);
    }
    if (use_ws && abl_env == 0 && !w8 && quad_ok) return wmh_launch_conv_bf16_ws(cfg, a, st, 0, 0);
#ifdef WMH_DIAG        // the 8-wave one-role variants (WMH_BF16_WAVES8) spill 300-416 bytes per lane: diagnostic library only
    if (bf16_variant() == 2) {
        if (cfg == 0) return launch_bf16<BCfg0w, 1>(a, (unsigned)nblk, st);
        return launch_bf16<BCfg2w, 1>(a, (unsigned)nblk, st);
    }
    if (w8) {
        if (cfg == 0) return launch_bf16<BCfg0w, 2>(a, (unsigned)nblk, st);
        return launch_bf16<BCfg2w, 2>(a, (unsigned)nblk, st);
    }
#endif
    if (cfg == 0) return launch_bf16<BCfg0, 4>(a, (unsigned)nblk, st);
    return launch_bf16<BCfg2, 3>(a, (unsigned)nblk, st);
}

int wmh_launch_conv_f16x2_ws(int cfg, const ConvArgs& a, hipStream_t st);
// y = conv3x3x3 with both operands split into two bf16 numbers each and all four partial products accumulated in fp32
// (16 mantissa bits per operand): fp32 tensors in and out, stride 1, wf from wmh_conv3d_k3_pack_weights_f16x2.  Same
// geometry, moments records (wmh_conv3d_k3_num_partials_bf16) and epilogue as wmh_conv3d_k3_fwd_bf16.
// Split-K factor of the wave-specialised stride-1 pair kernels (configurations 0, 2, 21): one workgroup per CU, so a layer with fewer
// tiles than half the CUs (16^3 and 8^3 levels at batch 2: 64 tiles) leaves most of the chip idle for the whole launch.  KS slices of the
// channel chunks per tile, as many as fit the chip once, each at least two chunks.
static int ws_ksplit(long long tiles, int nchunks) {
    static const int force = wmh_diag_env("WMH_WS_KSPLIT", 0);
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
        if (ncu < 8) ncu = 8;
    }
    if (force > 0) return force < nchunks ? force : nchunks;
    if (tiles < 1 || tiles * 2 > ncu) return 1;
    int ks = (int)(ncu / tiles);
    if (ks > 16) ks = 16;
    if (ks > nchunks / 2) ks = nchunks / 2;
    return ks < 2 ? 1 : ks;
}
static long long ws_tiles(int cfg, int N, int D, int H, int W, int Cout) {
    const BGeo ge = bf16_geo(cfg, false);
    return (long long)cdiv(W, ge.TX) * cdiv(H, ge.TY) * cdiv(D, ge.TZ) * N * cdiv(Cout, 32 * bf16_nt(cfg));
}
extern "C" size_t wmh_conv3d_k3_f16x2_workspace_floats(int cfg, int Cin, int N, int D, int H, int W, int Cout) {
    if (cfg == 0 || cfg == 2 || cfg == WMH_BF16_CFG_WIDE || cfg == WMH_F16X2_CFG_ROWS8) {
        const int ks = ws_ksplit(ws_tiles(cfg, N, D, H, W, Cout), cdiv(Cin, 8));
        return ks > 1 ? (size_t)ks * N * Cout * D * H * W : 0;
    }
    if (!(cfg == 3 || cfg == 4 || cfg == 5)) return 0;           // (configuration 1: wave-specialised, no split-K)
    const int S = bf16_stride(cfg);
    const int ks = bf16_ksplit(bf16_small_blocks(cfg, N, D, H, W, Cout), cdiv(Cin, 8));
    return ks > 1 ? (size_t)ks * N * Cout * ((D - 1) / S + 1) * ((H - 1) / S + 1) * ((W - 1) / S + 1) : 0;
}
extern "C" int wmh_conv3d_k3_fwd_f16x2(int cfg, const float* x0, const float* ss0, int C0, const float* x1, const float* ss1, int C1,
                                        const uint16_t* wf, const float* bias, float* y, float* partials, float* workspace,
                                        int N, int D, int H, int W, int Cout, float lrelu_slope, void* stream) {
    const int NT = bf16_nt(cfg);
    if (!NT || !wmh_conv3d_k3_f16x2_supported(cfg)) return wmh_fail(WMH_EINVAL, "conv3d_k3_fwd_f16x2: configuration has no f16x2 kernel");
    if (!x0 || !wf || !y || C0 < 1 || N < 1 || Cout < 1) return wmh_fail(WMH_EINVAL, "conv3d_k3_fwd_f16x2: null/empty argument");
    if (x1 == nullptr) C1 = 0;
    if (!(lrelu_slope >= 0.f && lrelu_slope <= 1.f)) return wmh_fail(WMH_EINVAL, "conv3d_k3_fwd_f16x2: LeakyReLU slope must be in [0, 1]");
    if (C1 > 0 && (C0 % 8) != 0) return wmh_fail(WMH_EINVAL, "conv3d_k3_fwd_f16x2: C0 must be a multiple of 8 with two sources");
    const bool s2ws = f16x2_s2_ws(cfg);
    if (s2ws && C1 > 0) return wmh_fail(WMH_EINVAL, "conv3d_k3_fwd_f16x2: the wave-specialised strided configurations take one source");
    const bool small = s2ws || cfg == 3 || cfg == 4 || cfg == 5;           // the one-role kernel (3, 4, 5) / the strided wave-specialised one (1, 25, 26)
    if ((!small || s2ws) && ((W % 4) != 0 || (size_t)16 * D * H * W * 4 >= ((size_t)1 << 32) || (reinterpret_cast<uintptr_t>(x0) & 15) != 0 ||
                   (x1 && (reinterpret_cast<uintptr_t>(x1) & 15) != 0)))
        return wmh_fail(WMH_EINVAL, "conv3d_k3_fwd_f16x2: needs W % 4 == 0 and 16-byte aligned sources");
    ConvArgs a;
    a.x0 = x0; a.ss0 = ss0; a.C0 = C0; a.x1 = x1; a.ss1 = ss1; a.C1 = C1;
    a.wf = reinterpret_cast<const float*>(wf); a.bias = bias; a.y = y; a.part = partials;
    const int S = bf16_stride(cfg);                 // configurations 3 and 5: stride 2, (D, H, W) is the INPUT extent
    a.N = N; a.D = D; a.H = H; a.W = W; a.Do = (D - 1) / S + 1; a.Ho = (H - 1) / S + 1; a.Wo = (W - 1) / S + 1; a.Cout = Cout;
    a.nchunks = (C1 > 0) ? (C0 / 8 + cdiv(C1, 8)) : cdiv(C0, 8);
    const BGeo ge = bf16_geo(cfg, false);
    a.ntx = cdiv(a.Wo, ge.TX); a.nty = cdiv(a.Ho, ge.TY); a.ntz = cdiv(a.Do, ge.TZ);
    a.npart = a.ntx * a.nty * a.ntz * ge.WPS;
    if ((size_t)a.Do * a.Ho * a.Wo * 4 * 32 >= ((size_t)1 << 32)) return wmh_fail(WMH_EINVAL, "conv3d_k3: 32 output channel planes must stay below 4 GB (epilogue lane offsets are 32-bit)");
    a.slope = lrelu_slope;
    a.G = cdiv(Cout, 32 * NT);
    a.KS = 1;
    static const int swz_env = wmh_diag_env("WMH_XCD_SWIZZLE", 1);
    static const int abl_x = wmh_diag_env("WMH_BF16_ABL", 0);      // diagnostics (timing only): see the one-role kernel
    a.xcd_swizzle = (swz_env & 1) | (small ? abl_x << 4 : 0);
    a.stamps = g_conv_stamps;
    a.by = a.bz = 1;
    hipStream_t st = (hipStream_t)stream;
    if (!small) {
        a.KS = workspace != nullptr ? ws_ksplit(ws_tiles(cfg, N, D, H, W, Cout), a.nchunks) : 1;
        if (a.KS > 1) a.y = workspace;                       // slabs [KS][N][Cout][D][H][W]; bias and moments come with the reduce
        const int rc = wmh_launch_conv_f16x2_ws(cfg, a, st);
        if (rc != WMH_OK || a.KS == 1) return rc;
        return wmh_launch_splitk_reduce(workspace, bias, y, partials, a.npart, a.KS, N, Cout, (size_t)a.Do * a.Ho * a.Wo, st);
    }
    // rows of 16 / 8 voxels (stride 1 and 2): the one-role kernel, split-K when slab memory is given (wmh_conv3d_k3_fwd_bf16_splitk)
    const long long nblk = (long long)a.ntx * a.nty * a.ntz * N * a.G;
    if (nblk > 0x7fffffffLL) return wmh_fail(WMH_EINVAL, "conv3d_k3_fwd_f16x2: grid too large");
    a.KS = (workspace != nullptr && !s2ws) ? bf16_ksplit(nblk, a.nchunks) : 1;
    if (a.KS > 1) a.y = workspace;
    const unsigned nb = (unsigned)(nblk * a.KS);
    int rc;
    // stride 2 on the wave-specialised kernel of conv3d_k3s2_f16x2_ws.hip (one source, no split-K)
    if (s2ws) return wmh_launch_conv_s2_f16x2_ws(cfg, a, st);
    if (cfg == 3) rc = launch_bf16<BCfg3, 4, true>(a, nb, st);
    else if (cfg == 4) rc = launch_bf16<BCfg4, 3, true>(a, nb, st);
    else rc = launch_bf16<BCfg5, 4, true>(a, nb, st);
    if (rc != WMH_OK || a.KS == 1) return rc;
    return wmh_launch_splitk_reduce(workspace, bias, y, partials, a.npart, a.KS, N, Cout, (size_t)a.Do * a.Ho * a.Wo, st);
}

extern "C" int wmh_conv3d_k3_fwd_bf16(int cfg, const float* x0, const float* ss0, int C0, const float* x1,
                                      const float* ss1, int C1, const uint16_t* wf, const float* bias, float* y,
                                      float* partials, int N, int D, int H, int W, int Cout, float lrelu_slope,
                                      void* stream) {
    return conv_fwd_bf16(cfg, x0, ss0, C0, x1, ss1, C1, wf, bias, y, partials, N, D, H, W, Cout, lrelu_slope, 0, 0, stream);
}
// The same with split-K slab memory for the small-row configurations (>= wmh_conv3d_k3_bf16_workspace_floats floats, or NULL)
extern "C" int wmh_conv3d_k3_fwd_bf16_splitk(int cfg, const float* x0, const float* ss0, int C0, const float* x1,
                                             const float* ss1, int C1, const uint16_t* wf, const float* bias, float* y,
                                             float* partials, float* workspace, int N, int D, int H, int W, int Cout,
                                             float lrelu_slope, void* stream) {
    return conv_fwd_bf16(cfg, x0, ss0, C0, x1, ss1, C1, wf, bias, y, partials, N, D, H, W, Cout, lrelu_slope, 0, 0, stream, workspace);
}
// Same convolution with bf16 TENSORS either side (x_bf16: both sources, y_bf16: the output; uint16 bit patterns, NCDHW);
// the InstanceNorm moments are still taken from the fp32 accumulators.
extern "C" int wmh_conv3d_k3_fwd_bf16_st(int cfg, const void* x0, const float* ss0, int C0, const void* x1,
                                         const float* ss1, int C1, const uint16_t* wf, const float* bias, void* y,
                                         float* partials, int N, int D, int H, int W, int Cout, float lrelu_slope,
                                         int x_bf16, int y_bf16, void* stream) {
    return conv_fwd_bf16(cfg, static_cast<const float*>(x0), ss0, C0, static_cast<const float*>(x1), ss1, C1, wf, bias,
                         static_cast<float*>(y), partials, N, D, H, W, Cout, lrelu_slope, x_bf16 != 0, y_bf16 != 0, stream);
}
