// conv3d_common.h -- definitions shared by the fp32 and bf16-operand forms of the 3x3x3 convolution kernel:
// launch arguments, the tile-geometry template, and the epilogue (bias is already in the accumulators; stores
// and one-pass InstanceNorm moments).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "wmh_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
// the "pair" mode (fp16 hi + lo per operand): the weights are split after scaling by this power of two (their low parts then stay
// in fp16's normal range down to |w| ~ 5e-4), the accumulators are scaled back before the epilogue
#define WMH_PAIR_WSCALE 256.f

struct ConvArgs {
    const float* x0; const float* ss0; int C0;
    const float* x1; const float* ss1; int C1;
    const float* wf; const float* bias; float* y; float* part;
    int N, D, H, W, Do, Ho, Wo, Cout;
    int nchunks, ntx, nty, ntz, npart;
    int G, KS, xcd_swizzle;
    int by, bz;                     // supertile of the persistent kernels' tile order (wmh_tile_coords); 1, 1 = plain x -> y -> z order
    float slope;
    unsigned long long* stamps;     // diagnostics only (NULL in production): per-workgroup cycle counts
    int accum;                      // conv3d_up_f16x2.hip only: y already holds bias + the skip half; add to it
};

// KZ: kernel extent along the first spatial axis (3, or 1 for nnU-Net's anisotropic [1,3,3] kernels); SZ: stride
// along that axis (differs from the in-plane stride S for [1,2,2] pooling)
template <int S_, int LX_, int LY_, int TZ_, int TY_, int TN_, int MT_, int NT_, int CK_, int APF_, int KZ_ = 3, int SZ_ = S_>
struct ConvCfg {
    static constexpr int S = S_, LX = LX_, LY = LY_, LZ = 32 / (LX_ * LY_), KZ = KZ_, SZ = SZ_, NTAPS = KZ_ * 9, PZ = KZ_ / 2;
    static constexpr int TX = LX_, TY = TY_, TZ = TZ_, TN = TN_, MT = MT_, NT = NT_, CK = CK_, APF = APF_;
    static constexpr int MY = TY / LY, MZ = TZ / LZ, MPS = MY * MZ;  // M-tiles per sample
    static constexpr int NMT = TN * MPS;                             // M-tiles per workgroup
    static constexpr int WM = NMT / MT;                              // waves per workgroup
    static constexpr int THREADS = WM * 64;
    static constexpr int WPS = MPS / MT;                             // waves per sample
    static constexpr int IX = S * (TX - 1) + 3, IY = S * (TY - 1) + 3, IZ = SZ * (TZ - 1) + KZ;
    static constexpr int YS = IX, ZS = IY * YS, NS = IZ * ZS;
    static constexpr int SP = TN * NS;                               // staged elements per channel
    static constexpr int NIT = (SP + THREADS - 1) / THREADS;
    static constexpr int CS = NIT * THREADS;                         // channel stride in LDS: padded so that every
                                                                     // thread's every store is in range (no branches)
    static constexpr int TILE = CK * CS;                             // floats per LDS buffer
    static constexpr int LDS_FLOATS = 2 * TILE;
    static constexpr int WT0 = (NIT >= NTAPS) ? 0 : NTAPS - NIT;           // tap at which the staged writes start
    static constexpr int WPT = (NIT + NTAPS - 1) / NTAPS;                      // write slices per tap
    static_assert(LX * LY * LZ == 32, "lane map must cover 32 voxels");
    static_assert(TY % LY == 0 && TZ % LZ == 0, "tile must be whole M-tiles");
    static_assert(MPS % MT == 0, "a wave's M-tiles must belong to one sample");
    static_assert(NMT % MT == 0 && WM >= 1 && WM <= 8, "1..8 waves");
    static_assert(NIT <= 32, "validity mask is 32 bits");
    static_assert(CK % 2 == 0, "k step is a cin pair");
    static_assert((((KZ - 1) * ZS + 2 * YS + 2) + (CK - 2) * CS + TILE) * 4 < 65536, "ds_read immediate range");
    static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS");
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

// lrelu(x * sc + sh) with UNPACKED fp32 instructions.  Measured (tools/micro/coissue.hip, profiles/r03_coissue.txt): a loader wave's
// v_pk_fma_f32 / v_pk_mul_f32 do not overlap at all with the MFMA stream of the consumer wave on the same SIMD (the kernel's wall
// time is the SUM of the two streams), while v_fma_f32 / v_mul_f32 / v_max_f32, the conversions and integer instructions do (wall =
// the longer stream): the packed form halves the instruction count and costs matrix time one for one.  Inline asm so that the
// compiler does not fuse neighbouring elements back into packed instructions; same roundings as the packed form (bit-identical).
__device__ __forceinline__ float wmh_norm_lrelu(float x, float sc, float sh, float slope) {
    float z, zs, r;
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(z) : "v"(x), "v"(sc), "v"(sh));
    asm("v_mul_f32 %0, %1, %2" : "=v"(zs) : "v"(z), "v"(slope));
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(z), "v"(zs));
    return r;
}

// The 16-bit-operand kernels (conv3d_k3_bf16.hip, conv3d_k3_bf16_ws.hip) walk the 27 taps in the order (kd, kw, kr): step p is tap
// wmh_tap_of(p) of the weight image (whose layout stays torch's kd, kr, kw).  The three kr taps of a (kd, kw) group read the same
// input rows shifted by one output row, which the wave-specialised kernel exploits; both kernels use the order so that their results
// stay bit-identical to each other.
__host__ __device__ constexpr int wmh_tap_of(int p) { return (p / 9) * 9 + (p % 3) * 3 + (p / 3) % 3; }

// Tile index (after the XCD-aware permutation) -> (cout group, tile coordinates, sample).  Tiles are enumerated supertile by
// supertile -- the ntx tiles of a row, a.by rows, a.bz slices -- instead of row by row through the whole slice: the ~32
// workgroups an XCD runs at one time then work on a compact block of the volume (128 x 16 x 16 voxels for the 128^3 layers)
// instead of half a slab (128 x 64 x 4), and the halos they share are found in that XCD's L2 instead of being fetched again.
// Placement only: results and moment records do not depend on it.  a.by | a.nty and a.bz | a.ntz (wmh_pick_supertile).
__device__ __forceinline__ void wmh_tile_coords(const ConvArgs& a, int t, int& g, int& tx, int& ty, int& tz, int& n0) {
    g = t % a.G; t /= a.G;
    const int per_sample = a.ntx * a.nty * a.ntz;
    n0 = t / per_sample;
    const int s = t - n0 * per_sample;
    const int blk = a.ntx * a.by * a.bz;
    const int q = s / blk;
    int r = s - q * blk;
    tx = r % a.ntx; r /= a.ntx;
    const int nby = a.nty / a.by;
    ty = (q % nby) * a.by + r % a.by;
    tz = (q / nby) * a.bz + r / a.by;
}
// by, bz: powers of two dividing nty / ntz with ntx * by * bz <= the workgroups an XCD runs at a time, grown towards a cube
// (TY, TZ: the tile's extent in voxels)
static inline void wmh_pick_supertile(ConvArgs& a, int TY, int TZ, int per_xcd) {
    a.by = a.bz = 1;
    for (;;) {
        const bool cy = a.nty % (2 * a.by) == 0, cz = a.ntz % (2 * a.bz) == 0;
        if (a.ntx * a.by * a.bz * 2 > per_xcd || (!cy && !cz)) break;
        const bool grow_y = cy && (!cz || a.by * TY <= a.bz * TZ);
        if (grow_y) a.by *= 2; else a.bz *= 2;
    }
}

// wave-specialised persistent form for the large-layer configuration (conv3d_k3_ws.hip)
int wmh_launch_conv_ws_in16(int cfg, const ConvArgs& a, hipStream_t st);   // config 1 with bf16 sources
int wmh_launch_conv_ws(int cfg, const ConvArgs& a, hipStream_t st);     // cfg 0..3: 32- and 16-wide rows, stride 1 and 2
int wmh_conv_ws_min_tiles();
int wmh_launch_conv_s2_f16x2_ws(int cfg, const ConvArgs& a, hipStream_t st);   // f16x2, stride 2: output rows of 32 / 16 / 8 voxels = configurations 1 / 25 / 26 (conv3d_k3s2_f16x2_ws.hip)
// split-K reduce (conv3d_k3.hip): y = bias + sum of KS slabs in fixed order, one moments record per (n, c), rest of the row zeroed
int wmh_launch_splitk_reduce(const float* slabs, const float* bias, float* y, float* partials, int npart, int KS, int N, int Cout,
                             size_t spatial, hipStream_t st);

// one step of the recursive-halving reduction (static indices only: a runtime index into a register array
// would be lowered to a compare/select chain per element)
template <int M>
__device__ __forceinline__ void halve_step(float (&vals)[32], int j) {
    const bool up = (j & M) != 0;
#pragma unroll
    for (int i = 0; i < M; ++i) {
        const float keep = up ? vals[i + M] : vals[i];
        const float send = up ? vals[i] : vals[i + M];
        vals[i] = keep + __shfl_xor(send, M);
    }
}

// ------------------------------------------------------------------------------------------------
// Epilogue of one workgroup tile.  acc[nt][m][r]: row (r&3)+8*(r>>2)+4*kh of cout tile nt, voxel j of M-tile m.
// ------------------------------------------------------------------------------------------------
// OUT16: y is a bf16 tensor (uint16 bits, round to nearest even); the moments still come from the fp32 accumulators.
// NTS: nontemporal (streaming) stores.  Measured both ways: they help the layer that does nothing but write a tensor far
// larger than the 256 MB Infinity Cache (first layer: 0.85 -> 0.66 ms) and HURT where the consumer finds its input still
// cached (64^3 / 32^3 levels: the transposed convolutions got 1.2-1.6x slower with them), so only the first layer asks for them.
template <class C, bool OUT16 = false, bool NTS = false>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x16 (&acc)[C::NT][C::MT], int g, int ks, int n,
                                              int wave, int j, int kh, int dx, int dy, int dz, int ts0, int oz0,
                                              int oy0, int ox0, int tile_sp) {
    constexpr int MT = C::MT, NT = C::NT;
    const int HoWo = a.Ho * a.Wo;
    const size_t DoHoWo = (size_t)a.Do * HoWo;
    const bool split = a.KS > 1;
    // split-K: raw partial sums into slab ks of y ([KS][N][Cout][Do][Ho][Wo]); bias/statistics come later
    float* ybase = a.y + (split ? (size_t)ks * a.N * a.Cout * DoHoWo : 0);

    unsigned vmask = 0;
    int cnt = 0;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int ts = ts0 + m;
        const int oz = oz0 + (ts / C::MY) * C::LZ + dz, oy = oy0 + (ts % C::MY) * C::LY + dy, ox = ox0 + dx;
        const bool valid = (n < a.N) && oz < a.Do && oy < a.Ho && ox < a.Wo;
        vmask |= valid ? (1u << m) : 0u;
        cnt += __popcll(__ballot(valid) & 0xffffffffull);
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int cob = (g * NT + nt) * 32;           // first output channel of this 32-row tile
        if (cob >= a.Cout) continue;                  // uniform over the workgroup
        const int nrows = min(32, a.Cout - cob);
        // ---- bias; InstanceNorm moments in ONE pass about a per-row shift K (one sample of the row) ----
        float vals[32];                               // [2r] = sum(v-K), [2r+1] = sum((v-K)^2) over this lane's voxels
        float kshift[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v0 = acc[nt][0][r];
            const float k0 = __builtin_amdgcn_readlane(v0, 0), k1 = __builtin_amdgcn_readlane(v0, 32);
            const float K = kh ? k1 : k0;
            kshift[r] = K;
            float sd = 0.f, sq = 0.f;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const float v = acc[nt][m][r];
                const float d = ((vmask >> m) & 1u) ? v - K : 0.f;
                sd += d;
                sq = fmaf(d, d, sq);
            }
            vals[2 * r] = sd; vals[2 * r + 1] = sq;
        }
        if (a.part != nullptr && !split) {
            // recursive halving over the 32 lanes of each half-wave: 16+8+4+2+1 shuffles; lane j ends with vals[j]
            halve_step<16>(vals, j); halve_step<8>(vals, j); halve_step<4>(vals, j); halve_step<2>(vals, j);
            halve_step<1>(vals, j);
            const float mine = vals[0];                       // lane j: (j&1) ? sum d^2 : sum d, of row j>>1
            const float other = __shfl_xor(mine, 1);
            float Kr = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) Kr = ((j >> 1) == r) ? kshift[r] : Kr;
            if ((j & 1) == 0 && n < a.N) {
                const int r = j >> 1;
                const int co = cob + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (co < a.Cout) {
                    const float fc = (float)cnt;
                    const float sd = mine, sq = other;
                    const float m2 = cnt > 0 ? fmaxf(sq - sd * sd / fc, 0.f) : 0.f;
                    const int pidx = tile_sp * C::WPS + (wave % C::WPS);
                    float4* pp = reinterpret_cast<float4*>(a.part) + ((size_t)n * a.Cout + co) * a.npart + pidx;
                    *pp = make_float4(fc, fmaf(fc, Kr, sd), m2, 0.f);
                }
            }
        }
        // ---- stores: one acc register = two 128-B row segments (two output channels) per wave-instruction.
        // (An LDS-transposed 16-byte-store variant measured slower: the epilogue is bound by the store path,
        // ~16 B/clk/CU, not by instruction count.)
        // bf16 output: 2-byte stores, 32 consecutive lanes = one 64-byte segment.  Packing the two voxels of an x pair into
        // one dword per lane (DPP swap, even lanes store row r, odd lanes row r+1) was 3x SLOWER: the address pattern across
        // consecutive lanes is what the store path cares about, not the number of instructions or the bytes per lane.
        if constexpr (OUT16) {
            unsigned short* ych = reinterpret_cast<unsigned short*>(a.y) + ((size_t)n * a.Cout + cob) * DoHoWo;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int ts = ts0 + m;
                const int oz = oz0 + (ts / C::MY) * C::LZ + dz, oy = oy0 + (ts % C::MY) * C::LY + dy, ox = ox0 + dx;
                if ((vmask >> m) & 1u) {
                    unsigned short* yp = ych + (size_t)(4 * kh) * DoHoWo + (size_t)oz * HoWo + oy * a.Wo + ox;
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        typedef __bf16 eb2 __attribute__((ext_vector_type(2)));
                        typedef float ef2 __attribute__((ext_vector_type(2)));
                        const ef2 f = {acc[nt][m][r], acc[nt][m][r + 1]};
                        const unsigned u = __builtin_bit_cast(unsigned, __builtin_convertvector(f, eb2));      // v_cvt_pk_bf16_f32
                        const int row0 = (r & 3) + 8 * (r >> 2), row1 = ((r + 1) & 3) + 8 * ((r + 1) >> 2);
                        if (4 * kh + row0 < nrows) {
                            if constexpr (NTS) __builtin_nontemporal_store((unsigned short)(u & 0xFFFFu), yp + (size_t)row0 * DoHoWo);
                            else yp[(size_t)row0 * DoHoWo] = (unsigned short)(u & 0xFFFFu);
                        }
                        if (4 * kh + row1 < nrows) {
                            if constexpr (NTS) __builtin_nontemporal_store((unsigned short)(u >> 16), yp + (size_t)row1 * DoHoWo);
                            else yp[(size_t)row1 * DoHoWo] = (unsigned short)(u >> 16);
                        }
                    }
                }
            }
            continue;
        }
        float* ych = ybase + ((size_t)n * a.Cout + cob) * DoHoWo;          // channel cob of sample n
        // A store's address is the wave-uniform base of the 32-channel group plus a 32-bit lane offset that walks the channel planes by
        // addition (the launchers check 32 planes < 4 GB): `global_store_dword v_off, v_data, s[base]`.  With 64-bit per-row offsets the
        // compiler kept sixteen scalar pairs, spilled them into vector lanes and read them back with two v_readlane per store.
        char* const ychb = reinterpret_cast<char*>(ych);
        const unsigned plane = 4u * (unsigned)DoHoWo;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int ts = ts0 + m;
            const int oz = oz0 + (ts / C::MY) * C::LZ + dz, oy = oy0 + (ts % C::MY) * C::LY + dy, ox = ox0 + dx;
            if ((vmask >> m) & 1u) {
                float* yp = ych + (size_t)(4 * kh) * DoHoWo + (size_t)oz * HoWo + oy * a.Wo + ox;
                if (nrows == 32) {                  // uniform: all 32 output channels of this tile exist
                    unsigned off = (unsigned)(4 * kh) * plane + 4u * (unsigned)(oz * HoWo + oy * a.Wo + ox);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {                    // rows (r & 3) + 8 (r >> 2): +1, +1, +1, +5 planes
                        float* p = reinterpret_cast<float*>(ychb + off);
                        if constexpr (NTS) __builtin_nontemporal_store(acc[nt][m][r], p);
                        else *p = acc[nt][m][r];
                        off += (r & 3) == 3 ? 5u * plane : plane;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (r & 3) + 8 * (r >> 2);
                        if (4 * kh + row < nrows) {
                            if constexpr (NTS) __builtin_nontemporal_store(acc[nt][m][r], yp + (size_t)row * DoHoWo);
                            else yp[(size_t)row * DoHoWo] = acc[nt][m][r];
                        }
                    }
                }
            }
        }
    }
}
