// conv3d_k3.hip -- 3x3x3 convolution (pad 1, stride 1|2) for gfx950 as an im2col-free implicit
// GEMM on the exact-fp32 matrix instruction v_mfma_f32_32x32x2_f32.
//
//   D[cout 32][voxel 32] += A[cout 32][k 2] * B[k 2][voxel 32],   k = (tap, cin) pairs
//
// * B (activations) comes from an LDS halo tile [CK cin][TN][IZ][IY][IX], double buffered and staged
//   once per cin chunk with coalesced NCDHW row reads; a lane's 27 taps are immediate offsets from one
//   LDS address, so the inner loop carries no address arithmetic.  The producer layer's
//   InstanceNorm+LeakyReLU is applied while staging (x' = lrelu(x*scale+shift)); halo voxels outside
//   the tensor are written as 0 AFTER that transform (torch pads the normalised tensor).
//   The next chunk's global loads are issued at the top of a chunk and its LDS writes are spread over
//   the chunk's last taps, between MFMAs: one barrier per chunk, no exposed load latency.
// * A (weights) is read straight from a fragment-major image in HBM/L2 ([G][chunk][tap][NT][lane][CK/2]),
//   one vector load per tap per lane through a prefetch ring that runs APF taps ahead ACROSS chunk
//   boundaries: vmcnt retires in order, so a weight load issued behind the staging loads would make
//   every chunk's first MFMA wait a full HBM latency.
// * Each wave owns MT voxel tiles x NT cout tiles (MT*NT accumulators of 16 registers).
// * Two input tensors may be given (concat-free torch.cat((up, skip), 1)).
// * Split-K over cin chunks (KS > 1) writes raw partial sums to a slab per split; a reduce kernel adds
//   them in fixed order together with the bias and produces the InstanceNorm statistics.
// * Epilogue (KS == 1): + bias, coalesced 128-B row-segment stores, per-wave InstanceNorm partials
//   (count, sum, M2 about the wave's own mean) for every output channel.
//
// Stands in for torch.nn.Conv3d inside nnU-Net's ConvDropoutNormNonlin [nnU-Net, not in tree];
// call sites in the reference: deepwmh/main/predict.py:153-156 (nnUNet_predict).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "conv3d_common.h"

// ABL: timing-only ablations (diagnostics; results are wrong when != 0): 1 no LDS writes of the next chunk,
// 2 no staging loads, 4 no weight loads, 8 B operand not read from LDS, 16 no per-chunk barrier
template <class C, int ABL = 0, bool OUT16 = false>
__global__ __launch_bounds__(C::THREADS, 2) void conv3d_k3_kernel(const ConvArgs a) {
    constexpr int S = C::S, MT = C::MT, NT = C::NT, CK = C::CK, NIT = C::NIT, THREADS = C::THREADS, APF = C::APF;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, kh = lane >> 5;
    const int dx = j % C::LX, dy = (j / C::LX) % C::LY, dz = j / (C::LX * C::LY);

    // 1-D grid over (tile, cout group, k split).  XCD-aware order: blocks are dealt round-robin to the 8
    // XCDs (bid % 8), so give each XCD one contiguous run of the sequence: halo neighbours and the cout
    // groups of one tile then share that XCD's L2.  Placement only affects speed, never results.
    int bid = blockIdx.x;
    if (a.xcd_swizzle) {
        const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int ks = bid % a.KS; bid /= a.KS;
    const int g = bid % a.G;
    int b = bid / a.G;
    const int tx = b % a.ntx; b /= a.ntx;
    const int ty = b % a.nty; b /= a.nty;
    const int tz = b % a.ntz; b /= a.ntz;
    const int n0 = b * C::TN;
    const int ox0 = tx * C::TX, oy0 = ty * C::TY, oz0 = tz * C::TZ;
    const int ix0 = ox0 * S - 1, iy0 = oy0 * S - 1, iz0 = oz0 * C::SZ - C::PZ;
    const int HW = a.H * a.W;
    const size_t DHW = (size_t)a.D * HW;
    // this block's range of cin chunks (split-K)
    const int ch_lo = (int)(((long long)a.nchunks * ks) / a.KS);
    const int ch_hi = (int)(((long long)a.nchunks * (ks + 1)) / a.KS);

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
        baddr[m] = kh * C::CS + tn_w * C::NS + C::SZ * (tzi * C::LZ + dz) * C::ZS + S * ((tyi * C::LY + dy) * C::YS + dx);
    }

    // accumulators start at the bias (split-K slabs start at 0: the reduce kernel adds the bias once)
    f32x16 acc[NT][MT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = (g * NT + nt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            const float bv = (a.KS == 1 && a.bias != nullptr && co < a.Cout) ? a.bias[co] : 0.f;
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[nt][m][r] = bv;
        }

    // ---- weights: prefetch ring over the linear (chunk, tap) sequence ---------------------------
    typedef typename AVec<CK / 2>::T AV;
    const AV* wl = reinterpret_cast<const AV*>(a.wf) + (size_t)g * a.nchunks * C::NTAPS * NT * 64 + lane;
    const int Tlast = ch_hi * C::NTAPS - 1;
    AV aring[APF][NT];
    auto loadA = [&](int T, AV (&dst)[NT]) {
        T = min(T, Tlast);                           // past the end: harmless re-read of the last tap
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) dst[nt] = wl[((size_t)T * NT + nt) * 64];
    };

    // ---- activations: staging registers, per-chunk scale/shift ----------------------------------
    float stg[CK][NIT];
    float sc[CK], sh[CK];      // used when TN == 1 (one sample per workgroup: wave-uniform)
    float slope_eff = 1.f;     // 1 = identity when the source carries no pending InstanceNorm+LeakyReLU

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
        if (C::TN == 1) {
            const bool has_ss = (ss != nullptr);
            slope_eff = has_ss ? a.slope : 1.f;
#pragma unroll
            for (int cc = 0; cc < CK; ++cc) {
                const int c = min(c0 + cc, Cs - 1);
                sc[cc] = has_ss ? ss[((size_t)n0 * Cs + c) * 2] : 1.f;
                sh[cc] = has_ss ? ss[((size_t)n0 * Cs + c) * 2 + 1] : 0.f;
            }
        }
    };
    // write slice `it` (all CK channels) of chunk `ch` into LDS buffer `buf`
    auto write_slice = [&](int ch, int it, float* dstbuf) {
        const float* xs; const float* ss; int Cs, c0;
        src_of(ch, xs, ss, Cs, c0);
        const int e = tid + it * THREADS;
#pragma unroll
        for (int cc = 0; cc < CK; ++cc) {
            const bool ok = ((c0 + cc) < Cs) && ((okmask >> it) & 1u);
            float v = stg[cc][it];
            if (C::TN == 1) {          // fmaf(v,1,0) and v*1 are exact: no branch for the "no transform" case
                v = fmaf(v, sc[cc], sh[cc]);
                v = v > 0.f ? v : v * slope_eff;
            } else if (ss != nullptr) {
                const int ni = min(e / C::NS, a.N - 1 - n0);
                const float* q = ss + ((size_t)(n0 + ni) * Cs + min(c0 + cc, Cs - 1)) * 2;
                v = fmaf(v, q[0], q[1]);
                v = v > 0.f ? v : v * a.slope;
            }
            v = ok ? v : 0.f;
            dstbuf[cc * C::CS + e] = v;
        }
    };

    // ---- prologue ---------------------------------------------------------------------------------
    unsigned long long t_begin = 0, t_loop = 0, t_epi = 0;
    if (WMH_STAMPS(a)) t_begin = __builtin_amdgcn_s_memtime();
    load_chunk(ch_lo);
#pragma unroll
    for (int d = 0; d < APF; ++d) loadA(ch_lo * C::NTAPS + d, aring[d]);
#pragma unroll
    for (int it = 0; it < NIT; ++it) write_slice(ch_lo, it, lds);
    __syncthreads();

    // ---- one cin chunk: 27 taps x CK/2 cin pairs x NT x MT MFMAs out of LDS buffer BUF -------------
    auto chunk = [&](auto bufc, int ch) {
        constexpr int BUF = decltype(bufc)::value;
        const float* cur = lds + BUF * C::TILE;
        float* nxt = lds + (BUF ^ 1) * C::TILE;
        // The next chunk is staged unconditionally (the last chunk re-stages itself into the idle buffer):
        // the body below is one branch-free basic block.
        const int chn = min(ch + 1, ch_hi - 1);
        if (!(ABL & 2)) load_chunk(chn);            // global loads stay in flight under the MFMAs below
#pragma unroll
        for (int tap = 0; tap < C::NTAPS; ++tap) {
            const int toff = (tap / 9) * C::ZS + ((tap / 3) % 3) * C::YS + (tap % 3);
            AV anew[NT];
            if (!(ABL & 4)) loadA(ch * C::NTAPS + tap + APF, anew);
            else {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) anew[nt] = aring[0][nt];
            }
            // pin the prefetch here: left alone, the scheduler sinks the load to its use APF taps later
            // and the in-order vmcnt wait for it then drains the staging loads too
            // (mask 0x38F: ALU, VALU, SALU, MFMA and DS instructions may still cross; only VMEM is pinned)
            __builtin_amdgcn_sched_barrier(0x38F);
#pragma unroll
            for (int k = 0; k < C::WPT; ++k) {
                const int it = (tap - C::WT0) * C::WPT + k;
                if (!(ABL & 1) && tap >= C::WT0 && it < NIT) write_slice(chn, it, nxt);
            }
#pragma unroll
            for (int p = 0; p < CK / 2; ++p) {
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    float bv;
                    if (!(ABL & 8)) bv = cur[baddr[m] + toff + 2 * p * C::CS];
                    else { bv = __int_as_float(baddr[m] + toff); asm volatile("" : "+v"(bv)); }
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[nt][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(avec_get(aring[0][nt], p), bv, acc[nt][m], 0, 0, 0);
                }
            }
#pragma unroll
            for (int d = 0; d + 1 < APF; ++d)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) aring[d][nt] = aring[d + 1][nt];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) aring[APF - 1][nt] = anew[nt];
        }
        if (!(ABL & 16)) __syncthreads();     // next buffer fully written, this one fully read
    };
    if (WMH_STAMPS(a)) t_loop = __builtin_amdgcn_s_memtime();
    {
        int ch = ch_lo;
        for (; ch + 1 < ch_hi; ch += 2) {
            chunk(std::integral_constant<int, 0>{}, ch);
            chunk(std::integral_constant<int, 1>{}, ch + 1);
        }
        if (ch < ch_hi) chunk(std::integral_constant<int, 0>{}, ch);
    }

    // ---- epilogue ---------------------------------------------------------------------------------
    if (WMH_STAMPS(a)) t_epi = __builtin_amdgcn_s_memtime();
    conv_epilogue<C, OUT16>(a, acc, g, ks, n0 + tn_w, wave, j, kh, dx, dy, dz, ts0, oz0, oy0, ox0, (tz * a.nty + ty) * a.ntx + tx);
    if (WMH_STAMPS(a) && tid == 0) {
        const unsigned long long t_end = __builtin_amdgcn_s_memtime();
        unsigned long long* o = WMH_STAMPS(a) + (size_t)blockIdx.x * 4;
        o[0] = t_loop - t_begin; o[1] = t_epi - t_loop; o[2] = t_end - t_epi; o[3] = (unsigned long long)(ch_hi - ch_lo);
    }
}

// ------------------------------------------------------------------------------------------------
// First layer (ONE input channel, stride 1): the 27 taps are the K dimension.
//   D[cout 32][voxel 32] += A[cout 32][tap pair] * B[tap pair][voxel 32]      14 MFMAs per M-tile (27 taps + 1 zero column)
// The general kernel spends 27 MFMAs per M-tile here with half of every k pair a zero channel, and stages its halo through
// the two-buffer pipeline built for many chunks; with one channel the whole layer is a single chunk: halo tile
// (6 x 10 x 34 voxels) into LDS once, a lane's B operand is ONE float of it at (its voxel) + (its tap's offset), the 14 A
// operands (this lane's output channel, its tap of the pair) stay in registers.  What is left is the store of 32 output
// channels per voxel: the layer is HBM-bound.  Tile geometry, moments records and stores are configuration 8's
// (conv_epilogue<Cfg8>), so the two kernels are interchangeable record for record.
// ------------------------------------------------------------------------------------------------
template <class C, bool OUT16>
__global__ __launch_bounds__(256, 2) void conv3d_first_kernel(const ConvArgs a) {
    static_assert(C::TX == 32 && C::LY == 1 && C::MT == C::TY && C::WM == C::TZ && C::NT == 1 && C::S == 1, "one z-slice of TY rows per wave");
    constexpr int TY = C::TY, IY = C::IY, IX = C::IX, NH = C::IZ * C::IY * C::IX;
    __shared__ float halo[NH];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, kh = lane >> 5;
    int bid = blockIdx.x;
    if (a.xcd_swizzle) {
        const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int g = bid % a.G;
    int b = bid / a.G;
    const int tx = b % a.ntx; b /= a.ntx;
    const int ty = b % a.nty; b /= a.nty;
    const int tz = b % a.ntz; b /= a.ntz;
    const int n = b;
    const int ox0 = tx * C::TX, oy0 = ty * C::TY, oz0 = tz * C::TZ;
    const int HW = a.H * a.W;
    const float* xn = a.x0 + (size_t)n * a.D * HW;
    const bool has_ss = a.ss0 != nullptr;
    const float sc = has_ss ? a.ss0[(size_t)n * 2] : 1.f, sh = has_ss ? a.ss0[(size_t)n * 2 + 1] : 0.f, slope = has_ss ? a.slope : 1.f;
    for (int e = tid; e < NH; e += 256) {
        const int iz = e / (IY * IX), rem = e - iz * (IY * IX), iy = rem / IX, ix = rem - iy * IX;
        const int gz = oz0 - 1 + iz, gy = oy0 - 1 + iy, gx = ox0 - 1 + ix;
        const bool ok = (unsigned)gz < (unsigned)a.D && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
        float v = 0.f;
        if (ok) {
            v = fmaf(xn[(size_t)gz * HW + gy * a.W + gx], sc, sh);
            v = v > 0.f ? v : v * slope;
        }
        halo[e] = v;                                   // zero padding AFTER the producer's activation, as everywhere
    }
    // configuration 8's weight image: element (g, tap, lane) = w[g*32 + (lane & 31)][lane >> 5][tap]; channel 0 is lanes 0..31
    float aw[14]; int toff[14];
#pragma unroll
    for (int s = 0; s < 14; ++s) {
        const int tap = 2 * s + kh, tc = tap < 27 ? tap : 0;
        const float w = a.wf[((size_t)g * 27 + tc) * 64 + j];
        aw[s] = tap < 27 ? w : 0.f;
        toff[s] = ((tc / 9) * IY + (tc / 3) % 3) * IX + tc % 3;
    }
    f32x16 acc[1][TY];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = g * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        const float bv = (a.bias != nullptr && co < a.Cout) ? a.bias[co] : 0.f;
#pragma unroll
        for (int m = 0; m < TY; ++m) acc[0][m][r] = bv;
    }
    __syncthreads();
    // (Round 6, measured and not kept: this layer in the f16x2 pair mode -- the 27 taps as K of v_mfma_f32_32x32x16_f16, a lane's operand gathered by
    //  eight 2-byte LDS reads, 6 MFMAs of 32 cycles per row instead of 14 of 64; correct to 2e-6 of float64 -- 1.030 -> 1.010 ms at 16 patches: the
    //  layer is its 4.3 GB of stores at 4.25 TB/s, two 128-byte segments per instruction, and the matrix time was already hidden by the CU's other
    //  workgroup.)
    // (Round 6, measured and not kept: a row's stores issued right behind its 14 MFMAs, so that they drain under the next row's -- 1.05 -> 1.13 ms
    //  at 16 patches.  The wave issues in order: once the store queue is full it waits at a store, and the MFMAs behind it wait with it; the
    //  overlap this kernel has comes from the CU's other workgroup.)
#pragma unroll
    for (int m = 0; m < TY; ++m) {
        const float* hp = halo + (wave * IY + m) * IX + j;
#pragma unroll
        for (int s = 0; s < 14; ++s) acc[0][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[s], hp[toff[s]], acc[0][m], 0, 0, 0);
    }
    conv_epilogue<C, OUT16, true>(a, acc, g, 0, n, wave, j, kh, j, 0, 0, wave * TY, oz0, oy0, ox0, (tz * a.nty + ty) * a.ntx + tx);
}

// ------------------------------------------------------------------------------------------------
// split-K reduce: y[n,c,:] = bias[c] + sum_ks slab[ks][n,c,:] (fixed order), and InstanceNorm moments.
// One workgroup per (n,c); the layers that use split-K are the low-resolution ones (<= 16^3 voxels).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* slabs, const float* bias, float* y, float* part,
                                                            int npart, int KS, int C, size_t NC, size_t spatial) {
    __shared__ float sh_s[4], sh_q[4];
    const size_t nc = blockIdx.x;
    const int c = (int)(nc % C);
    const float bv = bias ? bias[c] : 0.f;
    const size_t slab = NC * spatial;
    const float* p = slabs + nc * spatial;
    float* o = y + nc * spatial;
    float s = 0.f;
    for (size_t i = threadIdx.x; i < spatial; i += 256) {
        float v = p[i];
        for (int k = 1; k < KS; ++k) v += p[(size_t)k * slab + i];
        v += bv;
        o[i] = v;
        s += v;
    }
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
    if ((threadIdx.x & 63) == 0) sh_s[threadIdx.x >> 6] = s;
    __syncthreads();
    const float tot = (sh_s[0] + sh_s[1]) + (sh_s[2] + sh_s[3]);
    const float mean = tot / (float)spatial;
    __syncthreads();
    float q = 0.f;
    for (size_t i = threadIdx.x; i < spatial; i += 256) { const float d = o[i] - mean; q += d * d; }   // own writes: visible
    for (int m = 32; m >= 1; m >>= 1) q += __shfl_xor(q, m);
    if ((threadIdx.x & 63) == 0) sh_q[threadIdx.x >> 6] = q;
    __syncthreads();
    if (part) {   // record 0 carries the moments, the remaining records of this (n,c) row are empty (count 0)
        float4* row = reinterpret_cast<float4*>(part) + nc * npart;
        for (int i = threadIdx.x; i < npart; i += 256)
            row[i] = (i == 0) ? make_float4((float)spatial, tot, (sh_q[0] + sh_q[1]) + (sh_q[2] + sh_q[3]), 0.f)
                              : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// ------------------------------------------------------------------------------------------------
// configuration table
// ------------------------------------------------------------------------------------------------
//                      S  LX LY TZ TY TN MT NT CK APF
typedef ConvCfg<1, 32, 1, 4, 8, 1, 8, 1, 4, 6> Cfg0;    // stride 1, rows of >=32 voxels (128^3, 64^3, 32^3)
typedef ConvCfg<2, 32, 1, 2, 8, 1, 4, 2, 2, 6> Cfg1;    // stride 2, output rows of >=32
typedef ConvCfg<1, 16, 2, 2, 16, 1, 4, 2, 4, 6> Cfg2;   // stride 1, 16-wide
typedef ConvCfg<2, 16, 2, 2, 16, 1, 4, 2, 2, 6> Cfg3;   // stride 2, 16-wide output
typedef ConvCfg<1, 8, 4, 4, 8, 1, 2, 1, 8, 6> Cfg4;     // stride 1, 8-wide
typedef ConvCfg<2, 8, 4, 4, 8, 1, 2, 1, 2, 6> Cfg5;     // stride 2, 8-wide output
typedef ConvCfg<1, 4, 4, 4, 4, 4, 2, 1, 8, 6> Cfg6;     // stride 1, 4-wide, 4 samples per workgroup
typedef ConvCfg<2, 4, 4, 4, 4, 4, 2, 1, 2, 6> Cfg7;     // stride 2, 4-wide output
typedef ConvCfg<1, 32, 1, 4, 8, 1, 8, 1, 2, 6> Cfg8;    // as Cfg0 for 1-2 input channels (first layer): one cin pair
// anisotropic plans (nnU-Net [1,3,3] kernels, [1,2,2] pooling): same tiles, 9 taps and/or no stride along z
typedef ConvCfg<1, 32, 1, 4, 8, 1, 8, 1, 4, 6, 1, 1> Cfg9;      // kernel 1x3x3, stride 1
typedef ConvCfg<1, 16, 2, 2, 16, 1, 4, 2, 4, 6, 1, 1> Cfg10;
typedef ConvCfg<1, 8, 4, 4, 8, 1, 2, 1, 8, 6, 1, 1> Cfg11;
typedef ConvCfg<1, 4, 4, 4, 4, 4, 2, 1, 8, 6, 1, 1> Cfg12;
typedef ConvCfg<2, 32, 1, 2, 8, 1, 4, 2, 2, 6, 1, 1> Cfg13;     // kernel 1x3x3, stride (1,2,2)
typedef ConvCfg<2, 16, 2, 2, 16, 1, 4, 2, 2, 6, 1, 1> Cfg14;
typedef ConvCfg<2, 8, 4, 4, 8, 1, 2, 1, 2, 6, 1, 1> Cfg15;
typedef ConvCfg<2, 4, 4, 4, 4, 4, 2, 1, 2, 6, 1, 1> Cfg16;
typedef ConvCfg<2, 32, 1, 2, 8, 1, 4, 2, 2, 6, 3, 1> Cfg17;     // kernel 3x3x3, stride (1,2,2)
typedef ConvCfg<2, 16, 2, 2, 16, 1, 4, 2, 2, 6, 3, 1> Cfg18;
typedef ConvCfg<2, 8, 4, 4, 8, 1, 2, 1, 2, 6, 3, 1> Cfg19;
typedef ConvCfg<2, 4, 4, 4, 4, 4, 2, 1, 2, 6, 3, 1> Cfg20;
#define WMH_NUM_CONV_CFG 21

struct CfgInfo { int S, TX, TY, TZ, TN, NT, CK, WPS, threads, lds_bytes, KZ, SZ; };
template <class C> static CfgInfo info_of() {
    return CfgInfo{C::S, C::TX, C::TY, C::TZ, C::TN, C::NT, C::CK, C::WPS, C::THREADS, C::LDS_FLOATS * 4, C::KZ, C::SZ};
}
static const CfgInfo& cfg_info(int cfg) {
    static const CfgInfo t[WMH_NUM_CONV_CFG] = {info_of<Cfg0>(), info_of<Cfg1>(), info_of<Cfg2>(), info_of<Cfg3>(),
                                                info_of<Cfg4>(), info_of<Cfg5>(), info_of<Cfg6>(), info_of<Cfg7>(),
                                                info_of<Cfg8>(), info_of<Cfg9>(), info_of<Cfg10>(), info_of<Cfg11>(),
                                                info_of<Cfg12>(), info_of<Cfg13>(), info_of<Cfg14>(), info_of<Cfg15>(),
                                                info_of<Cfg16>(), info_of<Cfg17>(), info_of<Cfg18>(), info_of<Cfg19>(),
                                                info_of<Cfg20>()};
    return t[cfg];
}
unsigned long long* g_conv_stamps = nullptr;       // diagnostics: also read by the bf16 launcher
// diagnostics: device buffer of 4 x uint64 per workgroup that the next conv launches fill (NULL = off)
// (libdeepwmh_amd_diag.so only: the product library has no stamp code in its kernels, wmh_common.h)
#ifdef WMH_DIAG
extern "C" void wmh_debug_set_conv_stamps(void* p) { g_conv_stamps = (unsigned long long*)p; }
#endif
extern "C" int wmh_diag_build(void) {
#ifdef WMH_DIAG
    return 1;
#else
    return 0;
#endif
}

template <class C>
static int launch_conv(const ConvArgs& a, unsigned nblk, hipStream_t st) {
    static bool attr_set = false;
    static const int lds_extra = wmh_diag_env("WMH_LDS_EXTRA", 0);   // experiments: limit residency
    const int lds = C::LDS_FLOATS * 4 + lds_extra;
    if (!attr_set) {
        if (lds > 48 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_k3_kernel<C>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((conv3d_k3_kernel<C>), dim3(nblk), dim3(C::THREADS), lds, st, a);
    return wmh_check_launch("conv3d_k3_kernel");
}

#ifdef WMH_DIAG_ABLATIONS
// tuning variants of Cfg0 (same results, different schedule / tile orientation)
typedef ConvCfg<1, 32, 1, 4, 8, 1, 8, 1, 4, 3> Cfg0v1;     // shorter weight prefetch ring
typedef ConvCfg<1, 32, 1, 4, 8, 1, 8, 1, 4, 10> Cfg0v2;    // longer ring
typedef ConvCfg<1, 32, 1, 8, 4, 1, 8, 1, 4, 6> Cfg0v3;     // tile 8 x 4 x 32 (z-major)
typedef ConvCfg<1, 32, 1, 4, 8, 1, 8, 1, 4, 1> Cfg0v4;     // no look-ahead beyond the next tap
template <class CV>
static int launch_variant(const ConvArgs& a0, hipStream_t st) {
    ConvArgs a = a0;
    a.ntx = (a.Wo + CV::TX - 1) / CV::TX; a.nty = (a.Ho + CV::TY - 1) / CV::TY; a.ntz = (a.Do + CV::TZ - 1) / CV::TZ;
    a.part = nullptr;
    const long long nblk = (long long)a.ntx * a.nty * a.ntz * a.N * a.G;
    return launch_conv<CV>(a, (unsigned)nblk, st);
}
template <int ABL>
static int launch_abl(const ConvArgs& a, unsigned nblk, hipStream_t st) {
    const int lds = Cfg0::LDS_FLOATS * 4 + wmh_diag_env("WMH_LDS_EXTRA", 0);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_k3_kernel<Cfg0, ABL>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((conv3d_k3_kernel<Cfg0, ABL>), dim3(nblk), dim3(Cfg0::THREADS), lds, st, a);
    return wmh_check_launch("conv3d_k3_kernel<abl>");
}
static int launch_cfg0_ablation(int abl, const ConvArgs& a, unsigned nblk, hipStream_t st) {
    switch (abl) {
        case 1: return launch_abl<1>(a, nblk, st);
        case 2: return launch_abl<2>(a, nblk, st);
        case 3: return launch_abl<3>(a, nblk, st);
        case 4: return launch_abl<4>(a, nblk, st);
        case 7: return launch_abl<7>(a, nblk, st);
        case 8: return launch_abl<8>(a, nblk, st);
        case 15: return launch_abl<15>(a, nblk, st);
        case 16: return launch_abl<16>(a, nblk, st);
        case 31: return launch_abl<31>(a, nblk, st);
    }
    return -1;
}
#endif

// General form: kernel (kz,3,3) with kz in {1,3}; stride (sz, s, s) with sz <= s.  <0 when the combination has no kernel.
extern "C" int wmh_conv3d_config(int Cin, int Cout, int D, int H, int W, int kz, int sz, int s) {
    if (Cin < 1 || Cout < 1 || D < 1 || H < 1 || W < 1 || (s != 1 && s != 2) || (sz != 1 && sz != 2) || (kz != 1 && kz != 3) || sz > s)
        return wmh_fail(WMH_EINVAL, "wmh_conv3d_config: bad shape");
    const int Wo = (W - 1) / s + 1;
    const int wc = Wo >= 24 ? 0 : Wo >= 12 ? 1 : Wo >= 6 ? 2 : 3;          // width class
    if (kz == 3 && sz == s) {
        if (wc == 0 && s == 1 && Cin <= 2) return 8;
        return 2 * wc + (s == 2 ? 1 : 0);
    }
    if (kz == 1 && sz == 1) return (s == 1 ? 9 : 13) + wc;
    if (kz == 3 && sz == 1 && s == 2) return 17 + wc;
    return wmh_fail(WMH_EINVAL, "wmh_conv3d_config: kernel 1 with stride 2 along the same axis is not implemented");
}
extern "C" int wmh_conv3d_k3_config(int Cin, int Cout, int D, int H, int W, int stride) {
    return wmh_conv3d_config(Cin, Cout, D, H, W, 3, stride, stride);
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

extern "C" size_t wmh_conv3d_k3_packed_floats(int cfg, int Cin, int Cout) {
    if (cfg < 0 || cfg >= WMH_NUM_CONV_CFG) return 0;
    const CfgInfo& c = cfg_info(cfg);
    return (size_t)cdiv(Cout, 32 * c.NT) * cdiv(Cin, c.CK) * (c.KZ * 9) * c.NT * 64 * (c.CK / 2);
}

extern "C" int wmh_conv3d_k3_pack_weights(int cfg, const float* w, int Cin, int Cout, float* wf) {
    if (cfg < 0 || cfg >= WMH_NUM_CONV_CFG || !w || !wf) return wmh_fail(WMH_EINVAL, "pack_weights: bad args");
    const CfgInfo& c = cfg_info(cfg);
    const int CK = c.CK, NT = c.NT, G = cdiv(Cout, 32 * NT), NCH = cdiv(Cin, CK), V = CK / 2, NTAPS = c.KZ * 9;
    for (int g = 0; g < G; ++g)
        for (int ch = 0; ch < NCH; ++ch)
            for (int tap = 0; tap < NTAPS; ++tap)
                for (int nt = 0; nt < NT; ++nt)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int p = 0; p < V; ++p) {
                            const int co = (g * NT + nt) * 32 + (lane & 31), ci = ch * CK + 2 * p + (lane >> 5);
                            float v = 0.f;
                            if (co < Cout && ci < Cin) v = w[((size_t)co * Cin + ci) * NTAPS + tap];
                            wf[(((((size_t)g * NCH + ch) * NTAPS + tap) * NT + nt) * 64 + lane) * V + p] = v;
                        }
    return WMH_OK;
}

// Device-side repack (training: the weights change every step).  w is torch layout [Cout][Cin_total][27] in
// HBM; input channels [ci_off, ci_off+ci_cnt) are used.  mode 0: the forward image of a conv ci_cnt -> Cout.
// mode 1: the image of the DATA-GRADIENT conv Cout -> ci_cnt, i.e. W'[ci][co][tap] = W[co][ci][26 - tap].
// mode 2: w is a ConvTranspose3d(k2,s2) weight [Cout][ci_cnt][2][2][2] (Cout = the transposed conv's INPUT channels);
// the image is its data gradient as a stride-2 3x3x3 convolution ci_cnt -> Cout over dY: taps (1|2) of the padded
// 3-window are the 2-window, tap 0 is zero  (out[m] = sum_t in[2m + t - 1] W3[t]).  With a kernel-[1,3,3] configuration:
// the same for a ConvTranspose3d((1,2,2)) weight [Cout][ci_cnt][1][2][2].
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* w, float* wf, int Cin_total, int Cout, int ci_off,
                                                           int ci_cnt, int mode, int CK, int NT, int NCH, int NTAPS, size_t total) {
    const int V = CK / 2;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        size_t r = idx;
        const int pp = (int)(r % V); r /= V;
        const int lane = (int)(r % 64); r /= 64;
        const int nt = (int)(r % NT); r /= NT;
        const int tap = (int)(r % NTAPS); r /= NTAPS;
        const int ch = (int)(r % NCH); r /= NCH;
        const int g = (int)r;
        const int eco = (g * NT + nt) * 32 + (lane & 31), eci = ch * CK + 2 * pp + (lane >> 5);   // effective conv's (cout, cin)
        float v = 0.f;
        if (mode == 0) {
            if (eco < Cout && eci < ci_cnt) v = w[((size_t)eco * Cin_total + ci_off + eci) * NTAPS + tap];
        } else if (mode == 2) {
            const int tz = tap / 9, ty = (tap / 3) % 3, tx = tap % 3;
            if (NTAPS == 9) {                      // (1,2,2) transposed convolution [Cout][ci_cnt][1][2][2] in a (1,3,3) window
                if (eco < Cout && eci < ci_cnt && ty && tx)
                    v = w[((size_t)eco * Cin_total + ci_off + eci) * 4 + (ty - 1) * 2 + (tx - 1)];
            } else if (eco < Cout && eci < ci_cnt && tz && ty && tx)
                v = w[((size_t)eco * Cin_total + ci_off + eci) * 8 + (tz - 1) * 4 + (ty - 1) * 2 + (tx - 1)];
        } else {
            if (eco < ci_cnt && eci < Cout) v = w[((size_t)eci * Cin_total + ci_off + eco) * NTAPS + (NTAPS - 1 - tap)];
        }
        wf[idx] = v;
    }
}
extern "C" int wmh_conv3d_k3_pack_weights_device(int cfg, const float* w, int Cin_total, int Cout, int ci_off, int ci_cnt,
                                                 int mode, float* wf, void* stream) {
    if (cfg < 0 || cfg >= WMH_NUM_CONV_CFG || !w || !wf || ci_off < 0 || ci_cnt < 1 || ci_off + ci_cnt > Cin_total)
        return wmh_fail(WMH_EINVAL, "pack_weights_device: bad args");
    const CfgInfo& c = cfg_info(cfg);
    if (mode < 0 || mode > 2) return wmh_fail(WMH_EINVAL, "pack_weights_device: bad mode");
    const int eCin = mode == 1 ? Cout : ci_cnt, eCout = mode == 1 ? ci_cnt : Cout;
    const int NCH = cdiv(eCin, c.CK);
    const size_t total = (size_t)cdiv(eCout, 32 * c.NT) * NCH * (c.KZ * 9) * c.NT * 64 * (c.CK / 2);
    unsigned nb = (unsigned)((total + 255) / 256); if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(pack_weights_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, w, wf, Cin_total, Cout, ci_off,
                       ci_cnt, mode, c.CK, c.NT, NCH, c.KZ * 9, total);
    return wmh_check_launch("pack_weights_kernel");
}

// split-K factor for a launch: only the small, under-filled layers split
static int pick_ksplit(const CfgInfo& c, long long nblk, int nchunks) {
    static const int force = wmh_diag_env("WMH_KSPLIT", 0);
    if (force > 0) return force < nchunks ? force : nchunks;
    if (nblk >= 512) return 1;
    int ks = (int)((1024 + nblk - 1) / nblk);
    if (ks > 16) ks = 16;
    if (ks > nchunks / 2) ks = nchunks / 2;
    return ks < 1 ? 1 : ks;
}

extern "C" int wmh_conv3d_k3_num_partials(int cfg, int N, int Do, int Ho, int Wo) {
    if (cfg < 0 || cfg >= WMH_NUM_CONV_CFG) return wmh_fail(WMH_EINVAL, "num_partials: bad cfg");
    const CfgInfo& c = cfg_info(cfg);
    (void)N;
    return cdiv(Wo, c.TX) * cdiv(Ho, c.TY) * cdiv(Do, c.TZ) * c.WPS;
}

extern "C" size_t wmh_conv3d_k3_workspace_floats(int cfg, int Cin, int N, int D, int H, int W, int Cout, int stride) {
    if (cfg < 0 || cfg >= WMH_NUM_CONV_CFG) return 0;
    const CfgInfo& c = cfg_info(cfg);
    const int Do = (D - 1) / c.SZ + 1, Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const long long nblk = (long long)cdiv(Wo, c.TX) * cdiv(Ho, c.TY) * cdiv(Do, c.TZ) * cdiv(N, c.TN) * cdiv(Cout, 32 * c.NT);
    const int ks = pick_ksplit(c, nblk, cdiv(Cin, c.CK));
    return ks > 1 ? (size_t)ks * N * Cout * Do * Ho * Wo : 0;
}

static int conv_fwd(int cfg, const float* x0, const float* ss0, int C0, const float* x1,
                    const float* ss1, int C1, const float* wf, const float* bias, float* y,
                    float* partials, float* workspace, int N, int D, int H, int W, int Cout, int stride,
                    float lrelu_slope, int in16, int out16, void* stream) {
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
    a.Do = (D - 1) / c.SZ + 1; a.Ho = (H - 1) / stride + 1; a.Wo = (W - 1) / stride + 1;   // stride = in-plane stride
    a.Cout = Cout;
    a.nchunks = (C1 > 0) ? (C0 / c.CK + cdiv(C1, c.CK)) : cdiv(C0, c.CK);
    a.ntx = cdiv(a.Wo, c.TX); a.nty = cdiv(a.Ho, c.TY); a.ntz = cdiv(a.Do, c.TZ);
    a.npart = a.ntx * a.nty * a.ntz * c.WPS;
    if ((size_t)a.Do * a.Ho * a.Wo * 4 * 32 >= ((size_t)1 << 32)) return wmh_fail(WMH_EINVAL, "conv3d_k3: 32 output channel planes must stay below 4 GB (epilogue lane offsets are 32-bit)");
    a.slope = lrelu_slope;
    a.G = cdiv(Cout, 32 * c.NT);
    static const int swz_env = wmh_diag_env("WMH_XCD_SWIZZLE", 1);
    a.xcd_swizzle = swz_env;
    a.stamps = g_conv_stamps;
    a.by = a.bz = 1;
    long long nblk = (long long)a.ntx * a.nty * a.ntz * cdiv(N, c.TN) * a.G;
    const long long tiles = nblk;
    a.KS = pick_ksplit(c, nblk, a.nchunks);
    if (a.KS > 1 && workspace == nullptr) a.KS = 1;          // no slab memory given: run unsplit
    nblk *= a.KS;
    if (nblk > 0x7fffffffLL) return wmh_fail(WMH_EINVAL, "conv3d_k3_fwd: grid too large");
    hipStream_t st = (hipStream_t)stream;
    if (in16 || out16) a.KS = 1, nblk = tiles;               // bf16 tensors: no split-K slabs
    if (a.KS > 1) a.y = workspace;
    static const int use_ws = wmh_diag_env("WMH_WS", 1);   // wave-specialised persistent form: +4 % on the large layers
    int rc;
    // the wave-specialised form moves aligned 16-byte quads on 32-bit lane offsets and writes LeakyReLU as max(t, slope t)
    const bool ws_ok = (W % 4) == 0 && (reinterpret_cast<uintptr_t>(x0) & 15) == 0 && (!x1 || (reinterpret_cast<uintptr_t>(x1) & 15) == 0) &&
                       (size_t)4 * D * H * W * 4 < ((size_t)1 << 32) && lrelu_slope >= 0.f && lrelu_slope <= 1.f;
    // 32-wide rows: whenever the launch is not split; 16-wide rows: when there are enough tiles for one persistent workgroup
    // per CU -- a long cin loop is what this form likes, so it replaces split-K there (16^3 layers at batch 8)
    // bf16 tensors either side are served where the full-resolution level needs them: the first layer writes them
    // (config 8, one-role kernel), the first strided convolution reads them (config 1, wave-specialised kernel)
    if (in16) {
        if (cfg != 1 || out16 || !ws_ok) return wmh_fail(WMH_EINVAL, "conv3d_k3_fwd_st: bf16 sources are implemented for config 1 (W % 4 == 0, aligned) only");
        return wmh_launch_conv_ws_in16(cfg, a, st);
    }
    static const int first_env = wmh_diag_env("WMH_FIRST_TAPS", 1);      // 0: the general kernel for the one-channel first layer too (diagnostics)
    const bool first = cfg == 8 && C0 == 1 && C1 == 0 && first_env;
    if (out16) {
        if (cfg != 8) return wmh_fail(WMH_EINVAL, "conv3d_k3_fwd_st: a bf16 output is implemented for config 8 only");
        if (first) hipLaunchKernelGGL((conv3d_first_kernel<Cfg8, true>), dim3((unsigned)nblk), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((conv3d_k3_kernel<Cfg8, 0, true>), dim3((unsigned)nblk), dim3(Cfg8::THREADS), Cfg8::LDS_FLOATS * 4, st, a);
        return wmh_check_launch("conv3d_k3_kernel (bf16 out)");
    }
    if (first) {
        hipLaunchKernelGGL((conv3d_first_kernel<Cfg8, false>), dim3((unsigned)nblk), dim3(256), 0, st, a);
        return wmh_check_launch("conv3d_first_kernel");
    }
    if (use_ws && ws_ok && ((cfg == 0 && a.KS == 1) || (cfg >= 1 && cfg <= 3 && tiles >= wmh_conv_ws_min_tiles()))) {
        a.KS = 1; a.y = y;
        return wmh_launch_conv_ws(cfg, a, st);
    }
#ifdef WMH_DIAG_ABLATIONS
    static const int abl = wmh_diag_env("WMH_ABL", 0);
    if (cfg == 0 && abl) return launch_cfg0_ablation(abl, a, (unsigned)nblk, st);
    static const int var = wmh_diag_env("WMH_VARIANT", 0);
    if (cfg == 0 && var == 1) return launch_variant<Cfg0v1>(a, st);
    if (cfg == 0 && var == 2) return launch_variant<Cfg0v2>(a, st);
    if (cfg == 0 && var == 3) return launch_variant<Cfg0v3>(a, st);
    if (cfg == 0 && var == 4) return launch_variant<Cfg0v4>(a, st);
#endif
    switch (cfg) {
        case 0: rc = launch_conv<Cfg0>(a, (unsigned)nblk, st); break;
        case 1: rc = launch_conv<Cfg1>(a, (unsigned)nblk, st); break;
        case 2: rc = launch_conv<Cfg2>(a, (unsigned)nblk, st); break;
        case 3: rc = launch_conv<Cfg3>(a, (unsigned)nblk, st); break;
        case 4: rc = launch_conv<Cfg4>(a, (unsigned)nblk, st); break;
        case 5: rc = launch_conv<Cfg5>(a, (unsigned)nblk, st); break;
        case 6: rc = launch_conv<Cfg6>(a, (unsigned)nblk, st); break;
        case 7: rc = launch_conv<Cfg7>(a, (unsigned)nblk, st); break;
        case 8: rc = launch_conv<Cfg8>(a, (unsigned)nblk, st); break;
        case 9: rc = launch_conv<Cfg9>(a, (unsigned)nblk, st); break;
        case 10: rc = launch_conv<Cfg10>(a, (unsigned)nblk, st); break;
        case 11: rc = launch_conv<Cfg11>(a, (unsigned)nblk, st); break;
        case 12: rc = launch_conv<Cfg12>(a, (unsigned)nblk, st); break;
        case 13: rc = launch_conv<Cfg13>(a, (unsigned)nblk, st); break;
        case 14: rc = launch_conv<Cfg14>(a, (unsigned)nblk, st); break;
        case 15: rc = launch_conv<Cfg15>(a, (unsigned)nblk, st); break;
        case 16: rc = launch_conv<Cfg16>(a, (unsigned)nblk, st); break;
        case 17: rc = launch_conv<Cfg17>(a, (unsigned)nblk, st); break;
        case 18: rc = launch_conv<Cfg18>(a, (unsigned)nblk, st); break;
        case 19: rc = launch_conv<Cfg19>(a, (unsigned)nblk, st); break;
        case 20: rc = launch_conv<Cfg20>(a, (unsigned)nblk, st); break;
        default: return wmh_fail(WMH_EINVAL, "conv3d_k3_fwd: bad cfg");
    }
    if (rc != WMH_OK || a.KS == 1) return rc;
    // split-K: fixed-order slab sum + bias + statistics.  The reduce writes ONE moments record per (n,c);
    // fill the rest of the caller's [npart] records with zeros so wmh_instnorm_finalize can be used unchanged.
    return wmh_launch_splitk_reduce(workspace, bias, y, partials, a.npart, a.KS, N, Cout, (size_t)a.Do * a.Ho * a.Wo, st);
}
int wmh_launch_splitk_reduce(const float* slabs, const float* bias, float* y, float* partials, int npart, int KS, int N, int Cout,
                             size_t spatial, hipStream_t st) {
    const size_t NC = (size_t)N * Cout;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)NC), dim3(256), 0, st, slabs, bias, y, partials, npart, KS, Cout, NC, spatial);
    return wmh_check_launch("splitk_reduce_kernel");
}

extern "C" int wmh_conv3d_k3_fwd(int cfg, const float* x0, const float* ss0, int C0, const float* x1,
                                 const float* ss1, int C1, const float* wf, const float* bias, float* y,
                                 float* partials, float* workspace, int N, int D, int H, int W, int Cout, int stride,
                                 float lrelu_slope, void* stream) {
    return conv_fwd(cfg, x0, ss0, C0, x1, ss1, C1, wf, bias, y, partials, workspace, N, D, H, W, Cout, stride, lrelu_slope, 0, 0, stream);
}
// The same convolution with bf16 TENSORS on one side (uint16 bit patterns, NCDHW): x_bf16 = both sources, y_bf16 = the output.
extern "C" int wmh_conv3d_k3_fwd_st(int cfg, const void* x0, const float* ss0, int C0, const void* x1,
                                    const float* ss1, int C1, const float* wf, const float* bias, void* y,
                                    float* partials, int N, int D, int H, int W, int Cout, int stride,
                                    float lrelu_slope, int x_bf16, int y_bf16, void* stream) {
    return conv_fwd(cfg, static_cast<const float*>(x0), ss0, C0, static_cast<const float*>(x1), ss1, C1, wf, bias,
                    static_cast<float*>(y), partials, nullptr, N, D, H, W, Cout, stride, lrelu_slope, x_bf16 != 0, y_bf16 != 0, stream);
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
