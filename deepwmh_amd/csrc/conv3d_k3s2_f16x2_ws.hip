// conv3d_k3s2_f16x2_ws.hip -- the 3x3x3 convolution with stride 2 and output rows of 32 voxels (configuration 1: the first convolution of
// the 64^3 and 32^3 stages) in the f16x2 "pair" mode, wave-specialised and persistent.
//
// The one-role kernel that served these layers (conv3d_k3_bf16.hip, BCfg1) spends half its time staging: a 2 x 2 x 32 output tile reads
// a 5 x 5 x 65 input box -- 12.7 staged voxels per output voxel -- and each of its four waves issues the 673 instructions of a chunk's
// staging in order before it turns to the MFMAs (timing-only ablations: 0.85 of 1.74 ms for 32 -> 64 at 8 x 128^3 is the staging alone,
// DESIGN.md section 3.2).  Here the roles are split inside one workgroup of 12 waves per CU:
//   waves 0-3   consumers: wave w owns output row w of the tile (32 voxels) x 64 output channels; per tap one B operand from LDS (read
//               one tap ahead of its MFMAs) and two weight fragments straight from L2 (a ring four taps deep, one 16-byte load per lane and
//               fragment), four v_mfma_f32_32x32x16_f16; then the shared epilogue (conv3d_common.h: bias, fp32 InstanceNorm moments, stores);
//   waves 4-11  loaders: eight waves (two per SIMD beside the consumer) stage the next 8-channel chunk of the input box: buffer loads two
//               chunks ahead into alternating register sets, the producer's InstanceNorm + LeakyReLU, the fp16 split, two 16-byte LDS
//               writes per voxel.
// Operands are one dword per (voxel | output channel, input channel), (hi | lo << 16), as in the pair weight gradient and the pair
// transposed convolution: a k-slot pair holds a channel's (hi, lo), A as stored gives hi hi + lo lo and A rotated by 16 bits
// (v_alignbit_b32, which overlaps with the MFMAs) gives the cross terms.
// LDS: two activation buffers of 51 KB; the weights stay out of it.  (Round 4 built the other form -- one 54 KB weight buffer in LDS, the
// loaders carrying the next chunk's fragments in registers and dropping them in between two barriers, the consumers reading both operands
// through register rings -- 144 registers, no spills, bit-identical, and NOT faster: 1.79 against 1.78 ms for 32 -> 64 from 8 x 128^3,
// 0.68 against 0.66 ms for 64 -> 128 from 8 x 64^3, same box.  The L2 path of the fragments is not what bounds this kernel.)  One workgroup
// barrier per chunk; the loaders run ahead across tile boundaries, so the consumers' epilogue overlaps with the staging of the next tile's
// first chunks.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "conv3d_common.h"

typedef short s2_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 s2_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 s2_f16x2 __attribute__((ext_vector_type(2)));
typedef float s2_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned s2_u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t s2_rsrc;

// uniform read of a value no kernel in flight writes: a scalar-cache load (conv3d_k3_bf16_ws.hip::ld_uniform)
__device__ __forceinline__ float s2_ld_uniform(const float* p) {
    return *reinterpret_cast<const __attribute__((address_space(4))) float*>(reinterpret_cast<unsigned long long>(p));
}

// v -> (hi | lo << 16): hi = v cut to fp16's 11 significant bits (a mask: exact in fp16's normal range), lo = the remainder rounded to
// nearest by v_cvt_pk_f16_f32 (conv3d_k3_wgrad_f16x2.hip::wp_split; |v| >= 65520 gives a non-finite high part)
__device__ __forceinline__ unsigned s2_split(float v) {
    const float hi = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) & 0xFFFFE000u);
    const float lo = v - hi;
    return __builtin_bit_cast(unsigned, __builtin_convertvector(s2_f32x2{hi, lo}, s2_f16x2));
}

typedef ConvCfg<2, 32, 1, 2, 2, 1, 1, 2, 2, 6> S2Cfg;       // = BCfg1 of conv3d_k3_bf16.hip: tile 2 x 2 x 32, one row per wave, 64 channels

__global__ __launch_bounds__(768) void conv3d_k3s2_f16x2_ws_kernel(const ConvArgs a, const int T /*tiles x cout groups*/) {
    typedef S2Cfg C;
    constexpr int NT = C::NT, MT = C::MT, LTH = 512;
    constexpr int APF = 4;                                      // weight-fragment ring, taps ahead (2 and 8 measured the same)
    constexpr int SPP = (C::SP + 7) / 8 * 8;                    // voxels of an LDS buffer (the last one takes the stores of idle slots)
    constexpr int NITL = (C::SP + LTH - 1) / LTH;              // staged voxels per loader thread and chunk
    static_assert(MT == 1 && NT == 2 && C::WM == 4, "one output row per consumer wave, 64 output channels");
    extern __shared__ __attribute__((aligned(16))) uint4 ldsq[];   // [2][SPP voxels][2 pieces: channels 0-3 | 4-7 as (hi, lo) dwords, swapped where (x >> 3) & 1]

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

    if (wave < 4) {
        // ============================================ consumers ===========================================================
        const int j = lane & 31, kh = lane >> 5;
        const int dx = j;                                       // LX = 32, LY = LZ = 1
        const int ts0 = wave;                                   // M-tile (output row) of this wave
        const int tyi = ts0 % C::MY, tzi = ts0 / C::MY;
        int badr[3];
        {
            const int e0 = 2 * (tzi * C::ZS + tyi * C::YS + dx);
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) badr[kw] = (e0 + kw) * 32 + 16 * (kh ^ (((2 * dx + kw) >> 3) & 1));
        }
        const char* const ldsb = reinterpret_cast<const char*>(ldsq);
        // weight fragments: [g][nch][27 taps][NT][64 lanes] x 16 bytes; the ring runs APF taps ahead, across chunk and tile boundaries.
        // Everything about a ring load except the chunk's base is a compile-time constant of the unrolled step loop.
        const char* const wbytes = reinterpret_cast<const char*>(a.wf) + lane * 16;
        constexpr size_t CHB = (size_t)27 * NT * 64 * 16;       // bytes of a chunk's fragments
        s2_bf16x8 aring[APF][NT];
        auto group_of = [&](int k) __attribute__((always_inline)) {
            int g, tx, ty, tz, n0;
            tile_of(blockIdx.x + min(k, ntiles_mine - 1) * gridDim.x, g, tx, ty, tz, n0);
            return __builtin_amdgcn_readfirstlane(g);
        };
        auto frag = [&](const char* base, int st, s2_bf16x8 (&dst)[NT]) __attribute__((always_inline)) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) dst[nt] = *reinterpret_cast<const s2_bf16x8*>(base + ((size_t)wmh_tap_of(st) * NT + nt) * 64 * 16);
        };
        if (total > 0) {
            const char* w0 = wbytes + (size_t)group_of(0) * nch * CHB;
#pragma unroll
            for (int d = 0; d < APF; ++d) frag(w0, d, aring[d]);
        }
        unsigned long long tm = 0, tb = 0, te = 0, t0 = 0;
        __syncthreads();                                        // chunk 0 is staged
        int q = 0;
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
            const int g_next = group_of(k + 1);
            for (int ch = 0; ch < nch; ++ch, ++q) {
                if (WMH_STAMPS(a)) t0 = __builtin_amdgcn_s_memtime();
                const char* cur = ldsb + (size_t)(q & 1) * SPP * 32;
                // this chunk's fragments and the next one's (the next tile's first chunk behind the last; past the end: re-read, never used)
                const char* wc = wbytes + ((size_t)g * nch + ch) * CHB;
                const char* wn = ch + 1 < nch ? wc + CHB : wbytes + (size_t)g_next * nch * CHB;
                // B operands one tap ahead of their MFMAs (a wave issues in order: a read placed in front of its own use waits out the
                // LDS latency with the matrix pipe empty)
                s2_bf16x8 bbuf[2];
                auto loadB = [&](int st) __attribute__((always_inline)) {
                    const int tap = wmh_tap_of(st);
                    return *reinterpret_cast<const s2_bf16x8*>(cur + badr[tap % 3] + ((tap / 9) * C::ZS + ((tap / 3) % 3) * C::YS) * 32);
                };
                bbuf[0] = loadB(0);
#pragma unroll
                for (int step = 0; step < 27; ++step) {
                    s2_bf16x8 anew[NT];
                    if (step + APF < 27) frag(wc, step + APF, anew); else frag(wn, step + APF - 27, anew);
                    if (step + 1 < 27) bbuf[(step + 1) & 1] = loadB(step + 1);
                    __builtin_amdgcn_sched_barrier(0);
                    // A = 4 channels x (w_hi, w_lo) as stored; A' = the same dwords rotated by 16 bits.  With B = 4 channels x (x_hi, x_lo):
                    // A B = hi hi + lo lo, A' B = lo hi + hi lo -- all four products.  (The first version kept the one-role kernel's image
                    // -- high parts on lanes 0-31, low parts on lanes 32-63 -- and split it with v_permlane32_swap_b32: 8 swaps + 8 moves
                    // per tap that do not overlap with the MFMAs, 7.1 k instead of 4.2 k cycles per chunk.)
                    s2_u32x4 ahi[NT], alo[NT];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        ahi[nt] = __builtin_bit_cast(s2_u32x4, aring[0][nt]);
#pragma unroll
                        for (int d = 0; d < 4; ++d) alo[nt][d] = __builtin_amdgcn_alignbit(ahi[nt][d], ahi[nt][d], 16);
                    }
                    const s2_bf16x8 bv = bbuf[step & 1];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        acc[nt][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(s2_f16x8, ahi[nt]), __builtin_bit_cast(s2_f16x8, bv), acc[nt][0], 0, 0, 0);
                        acc[nt][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(s2_f16x8, alo[nt]), __builtin_bit_cast(s2_f16x8, bv), acc[nt][0], 0, 0, 0);
                    }
#pragma unroll
                    for (int d = 0; d + 1 < APF; ++d)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) aring[d][nt] = aring[d + 1][nt];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) aring[APF - 1][nt] = anew[nt];
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (WMH_STAMPS(a)) { const unsigned long long t = __builtin_amdgcn_s_memtime(); tm += t - t0; t0 = t; }
                __syncthreads();                                // this chunk is consumed, the next one is staged
                if (WMH_STAMPS(a)) tb += __builtin_amdgcn_s_memtime() - t0;
            }
            if (WMH_STAMPS(a)) t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt][0] *= (1.f / WMH_PAIR_WSCALE);      // undo the weights' 2^8 (exact)
            conv_epilogue<C>(a, acc, g, 0, n0, wave, j, kh, dx, 0, 0, ts0, tz * C::TZ, ty * C::TY, tx * C::TX, (tz * a.nty + ty) * a.ntx + tx);
            if (WMH_STAMPS(a)) te += __builtin_amdgcn_s_memtime() - t0;
        }
        if (WMH_STAMPS(a) && tid == 0) {
            unsigned long long* o = WMH_STAMPS(a) + (size_t)blockIdx.x * 8;
            o[0] = tm; o[1] = tb; o[2] = te; o[3] = (unsigned long long)total;
        }
        return;
    }

    // ================================================ loaders =================================================================
    const int lt = tid - 256;
    const unsigned cstride = 4u * (unsigned)DHW;                // bytes between channel planes (host checks the range)
    // slot i of this thread = voxel e = lt + 512 i of the 65 x 5 x 5 input box: where its two 16-byte pieces go (tile-invariant) ...
    unsigned lhi[NITL];                                        // (the second piece sits at lhi ^ 16)
#pragma unroll
    for (int i = 0; i < NITL; ++i) {
        const int e = lt + i * LTH;
        const int ew = e < C::SP ? e : SPP - 1;
        const int key = e < C::SP ? (((e % C::IX) >> 3) & 1) : 0;
        lhi[i] = (unsigned)((ew * 2 + (0 ^ key)) * 16);
    }
    // ... and, per tile, where it comes from (byte offset in a channel plane) and whether it lies in the volume
    unsigned exyz[NITL];                                        // x | y << 8 | z << 16 of the slot's voxel in the input box
#pragma unroll
    for (int i = 0; i < NITL; ++i) {
        const int e = lt + i * LTH;
        exyz[i] = (unsigned)(e % C::IX) | ((unsigned)((e / C::IX) % C::IY) << 8) | ((unsigned)(e / (C::IX * C::IY)) << 16);
    }
    unsigned voff[NITL], vbits = 0; int s_n0 = 0;               // vbits: bit i = slot i lies in the volume
    auto setup = [&](int v) __attribute__((always_inline)) {
        int g, tx, ty, tz, n0;
        tile_of(v, g, tx, ty, tz, n0);
        s_n0 = __builtin_amdgcn_readfirstlane(n0);
        const int ix0 = __builtin_amdgcn_readfirstlane(tx * C::TX * 2 - 1), iy0 = __builtin_amdgcn_readfirstlane(ty * C::TY * 2 - 1),
                  iz0 = __builtin_amdgcn_readfirstlane(tz * C::TZ * 2 - 1);
#pragma unroll
        for (int i = 0; i < NITL; ++i) {
            const int e = lt + i * LTH;
            // the slot's position in the box, unpacked from ONE register per slot behind an opaque copy: left visible, the loop-invariant
            // divisions of all four slots were hoisted out of the chunk loop into twelve registers -- the spills that followed made every
            // iteration reload them with a scratch load that waits for the whole prefetch
            unsigned pk = exyz[i];
            asm volatile("" : "+v"(pk));
            const int x = (int)(pk & 0xFFu), y = (int)((pk >> 8) & 0xFFu), z = (int)(pk >> 16);
            const int gx = ix0 + x, gy = iy0 + y, gz = iz0 + z;
            const bool ok = (e < C::SP) & ((unsigned)gx < (unsigned)a.W) & ((unsigned)gy < (unsigned)a.H) & ((unsigned)gz < (unsigned)a.D);
            vbits = i == 0 ? (ok ? 1u : 0u) : (vbits | (ok ? (1u << i) : 0u));
            voff[i] = ok ? 4u * (unsigned)(gz * HW + gy * a.W + gx) : 0u;
        }
    };
    // two register sets: the chunk fetched during iteration q is converted during iteration q + 1
    float xr[2][NITL][8]; unsigned kbits[2] = {0, 0}; int set_n0[2] = {0, 0}, set_ch[2] = {0, 0};
    int pk = 0, pc = 0;                                         // (tile counter, chunk) of the NEXT issue
    // issue(set, i): voxel slot i of the next chunk (8 channel loads); i == NITL - 1 also moves the cursor on.  convert(set, buf, i): slot i.
    // (The loaders' loads, not their conversions, are what slows the consumers down -- 4.3 k -> 6.5 k cycles of MFMA phase per chunk with
    //  the loads alone: they share the CU's vector-memory queue with the weight-fragment loads.  Issuing them slot by slot between the
    //  conversions instead of back to back made the loaders themselves slower (5.1 k -> 8.5 k) and the kernel no faster.)
    auto issue = [&](int set, int i) __attribute__((always_inline)) {
        const int c0 = pc * 8;
        const s2_rsrc rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x0 + (size_t)s_n0 * a.C0 * DHW), 0, (int)((unsigned)a.C0 * cstride), 0x00020000);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const unsigned soff = (unsigned)min(c0 + c, a.C0 - 1) * cstride;          // channels past the end re-read the last one (scaled to zero)
            xr[set][i][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)voff[i], (int)soff, 0));
        }
        if (i == 0) kbits[set] = vbits;
        if (i == NITL - 1) {
            set_n0[set] = s_n0; set_ch[set] = pc;
            if (++pc == nch) { pc = 0; ++pk; if (pk < ntiles_mine) setup(blockIdx.x + pk * gridDim.x); }
        }
    };
    float scv[8], shv[8], slope_eff = 1.f;
    auto load_ss = [&](int set) __attribute__((always_inline)) {
        // InstanceNorm scale / shift of the chunk's 8 channels (1, 0 without a pending normalisation; 0, 0 past the last channel).
        // Uniform reads through the constant address space: scalar-cache loads on their own counter.  As vector loads they queue
        // behind the prefetch on the in-order vmcnt, and waiting for them waits for the whole prefetch (HBM latency per chunk: that
        // was 2.5 k of the loaders' 7.6 k cycles).  Branch-free: without a pending normalisation the reads go to {1, 0}.
        const int c0 = __builtin_amdgcn_readfirstlane(set_ch[set]) * 8, sn0 = __builtin_amdgcn_readfirstlane(set_n0[set]);
        const bool has_ss = a.ss0 != nullptr;
        slope_eff = has_ss ? a.slope : 1.f;
        static __device__ const float ident2[2] = {1.f, 0.f};
        const float* tab = has_ss ? a.ss0 + ((size_t)sn0 * a.C0 + c0) * 2 : ident2;
        const int mul = has_ss ? 2 : 0, cmax = a.C0 - 1 - c0;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float sc = s2_ld_uniform(tab + mul * min(c, cmax)), sh = s2_ld_uniform(tab + mul * min(c, cmax) + 1);
            scv[c] = c <= cmax ? sc : 0.f;
            shv[c] = c <= cmax ? sh : 0.f;
        }
    };
    auto convert = [&](int set, int buf, int i) __attribute__((always_inline)) {
        char* dst = reinterpret_cast<char*>(ldsq) + (size_t)buf * SPP * 32;
        unsigned w[8];
        const unsigned keep = 0u - ((kbits[set] >> i) & 1u);   // all ones inside the volume
#pragma unroll
        for (int c = 0; c < 8; ++c) {                           // one dword per channel: (hi | lo << 16)
            float t = fmaf(xr[set][i][c], scv[c], shv[c]);
            t = fmaxf(t, t * slope_eff);
            w[c] = s2_split(t) & keep;
        }
        *reinterpret_cast<uint4*>(dst + lhi[i]) = make_uint4(w[0], w[1], w[2], w[3]);
        *reinterpret_cast<uint4*>(dst + (lhi[i] ^ 16u)) = make_uint4(w[4], w[5], w[6], w[7]);
    };
    unsigned long long tw = 0, tbw = 0, t0 = 0;
    if (total > 0) {
        setup(blockIdx.x);
#pragma unroll
        for (int i = 0; i < NITL; ++i) issue(0, i);             // chunk 0, exposed once per workgroup
        load_ss(0);
#pragma unroll
        for (int i = 0; i < NITL; ++i) convert(0, 0, i);
        if (total > 1) {                                        // chunk 1 in flight across the barrier
#pragma unroll
            for (int i = 0; i < NITL; ++i) issue(1, i);
        }
    }
    __syncthreads();
    for (int q = 0; q < total; q += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {                           // iteration q + h: the consumers are on chunk q + h (buffer h)
            const int qq = q + h;
            if (qq >= total) break;
            if (WMH_STAMPS(a)) t0 = __builtin_amdgcn_s_memtime();
            // chunk qq + 2 is fetched into the set chunk qq came from, then chunk qq + 1 (fetched during the previous iteration, set
            // h ^ 1) is converted into the other buffer
            const bool cv = qq + 1 < total, is = qq + 2 < total;
            if (cv) load_ss(h ^ 1);
            // (sched_barrier: one slot's conversion at a time.  Without these fences the whole iteration is one basic block -- it was not
            //  while cycle stamps were compiled in -- and the scheduler interleaves all four slots' conversions: 60 registers spilled, and
            //  every scratch reload waits for the whole prefetch; the kernel ran 1.6x slower)
            if (is) {
#pragma unroll
                for (int i = 0; i < NITL; ++i) issue(h, i);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (cv) {
#pragma unroll
                for (int i = 0; i < NITL; ++i) { convert(h ^ 1, h ^ 1, i); __builtin_amdgcn_sched_barrier(0); }
            }
            if (WMH_STAMPS(a)) { const unsigned long long t = __builtin_amdgcn_s_memtime(); tw += t - t0; t0 = t; }
            __syncthreads();
            if (WMH_STAMPS(a)) tbw += __builtin_amdgcn_s_memtime() - t0;
        }
    }
    if (WMH_STAMPS(a) && lane == 0 && wave == 4) {
        unsigned long long* o = WMH_STAMPS(a) + (size_t)blockIdx.x * 8;
        o[4] = tw; o[5] = tbw;
    }
}

// a: fully prepared by wmh_conv3d_k3_fwd_f16x2 (configuration 1, one source, no split-K)
int wmh_launch_conv_s2_f16x2_ws(const ConvArgs& a_in, hipStream_t st) {
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
        if (ncu < 8) ncu = 8;
        ncu &= ~7;
    }
    constexpr int lds = 2 * ((S2Cfg::SP + 7) / 8 * 8) * 32;
    static_assert(lds <= 160 * 1024, "LDS");
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_k3s2_f16x2_ws_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const long long T = (long long)a_in.ntx * a_in.nty * a_in.ntz * a_in.N * a_in.G;
    if (T > 0x7fffffffLL) return wmh_fail(WMH_EINVAL, "conv3d_k3 stride 2 f16x2 (wave-specialised): too many tiles");
    if ((size_t)a_in.C0 * a_in.D * a_in.H * a_in.W * 4 >= ((size_t)1 << 32))
        return wmh_fail(WMH_EINVAL, "conv3d_k3 stride 2 f16x2 (wave-specialised): a sample of the source must stay below 4 GB (32-bit buffer offsets)");
    const int grid = T < ncu ? (int)T : ncu;
    ConvArgs a = a_in;
    static const int supertile = wmh_diag_env("WMH_SUPERTILE", 1);
    if (supertile && (a.xcd_swizzle & 1)) wmh_pick_supertile(a, S2Cfg::TY, S2Cfg::TZ, grid / 8 > 0 ? grid / 8 : 1);
    hipLaunchKernelGGL(conv3d_k3s2_f16x2_ws_kernel, dim3(grid), dim3(768), lds, st, a, (int)T);
    return wmh_check_launch("conv3d_k3s2_f16x2_ws_kernel");
}
