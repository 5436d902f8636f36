// cl_common.h -- shared pieces of the channels-last bf16 kernels (BASELINE.json configs[2], "bf16 forward").
//
// Tensor format "CL16": [n][c / 8][z][y][x][c % 8], bf16 bit patterns (round to nearest even), raw convolution
// outputs; the InstanceNorm + LeakyReLU of the producing layer is still pending as a per-(n, c) {scale, shift} pair and
// is applied by every consumer while it stages its operand (the normalised tensor is never written).
// Why 8 channels innermost: a voxel's 8 consecutive channels are ONE 16-byte piece = one lane's B operand of
// v_mfma_f32_32x32x16_bf16, so global -> LDS staging needs no transposition (LDS-DMA, no registers), and the epilogue
// writes 16 bytes per lane (an NCDHW bf16 tensor can only be written 2 bytes per lane: the store path, not HBM, bounds it).
// Why the 8-channel blocks are planes (round 3; rounds 1-2 kept all C channels of a voxel together, [n][z][y][x][c]): consecutive
// lanes of a wave work on consecutive voxels, so a wave-instruction of the LDS-DMA or of the epilogue's stores then covers one
// contiguous run of 16-byte pieces (a halo row of 34 voxels = 544 bytes, 5 cache lines) instead of one piece out of every voxel's
// 2C bytes (64 cache lines per instruction at C >= 64): the texture addresser takes a cache line per cycle, and at 42 DMA
// instructions per 16-channel chunk that was 2.7 k of the chunk's ~6 k cycles on the staging waves.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "conv3d_common.h"

typedef short cl_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned cl_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned cl_u32x2 __attribute__((ext_vector_type(2)));
typedef float cl_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 cl_bf2 __attribute__((ext_vector_type(2)));

#define CL_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define CL_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
// workgroup barrier that leaves LDS-DMA (vmcnt) in flight: __syncthreads() would drain it (its fence waits vmcnt(0))
#define CL_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

__device__ __forceinline__ unsigned cl_pack2(float lo, float hi) {           // v_cvt_pk_bf16_f32 (round to nearest even)
    const cl_f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, cl_bf2));
}
// max(a, b) as ONE v_max_f32: fmaxf() canonicalises both operands first (three instructions), and an inline-asm v_max costs
// a wait state after every statement; the median-of-three intrinsic with +infinity folds to the bare instruction
__device__ __forceinline__ float cl_vmax(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, __builtin_inff()); }
// Uniform read of a value no kernel in flight writes: scalar-cache load (its own counter, not the in-order vmcnt queue)
__device__ __forceinline__ float cl_ld_uniform(const float* p) {
    return *reinterpret_cast<const __attribute__((address_space(4))) float*>(reinterpret_cast<unsigned long long>(p));
}

// One 16-byte piece = 8 consecutive channels of one voxel: lrelu(x * scale + shift), back to bf16.
// sc / sh: the 8 channels' scale and shift; slope 1 with scale 1 / shift 0 = identity (a source without pending norm).
// UNPACKED: plain v_fma_f32 / v_mul_f32 / v_max_f32 (conv3d_common.h::wmh_norm_lrelu) -- for a staging wave that shares its SIMD with a
// consumer wave's MFMA stream: packed fp32 instructions do not overlap with it (cl_conv3_kernel's stride-1 layers 3-8 % faster,
// round 3).  The one-role kernels (stride 2, first layer, transposed convolution), where the same wave issues the MFMAs afterwards,
// keep the packed form: half the instructions (measured: 0.69 against 0.71 ms on the first strided layer).  Same roundings either way.
template <bool UNPACKED = false>
__device__ __forceinline__ cl_u32x4 cl_transform8(const cl_u32x4 u, const float (&sc)[8], const float (&sh)[8], float slope) {
    cl_u32x4 w;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        if constexpr (UNPACKED) {
            const float x0 = __builtin_bit_cast(float, u[p] << 16), x1 = __builtin_bit_cast(float, u[p] & 0xFFFF0000u);
            w[p] = cl_pack2(wmh_norm_lrelu(x0, sc[2 * p], sh[2 * p], slope), wmh_norm_lrelu(x1, sc[2 * p + 1], sh[2 * p + 1], slope));
        } else {
            const cl_f32x2 x = {__builtin_bit_cast(float, u[p] << 16), __builtin_bit_cast(float, u[p] & 0xFFFF0000u)};
            const cl_f32x2 s = {sc[2 * p], sc[2 * p + 1]}, t = {sh[2 * p], sh[2 * p + 1]};
            const cl_f32x2 z = __builtin_elementwise_fma(x, s, t);
            const cl_f32x2 zs = z * slope;
            w[p] = cl_pack2(cl_vmax(z[0], zs[0]), cl_vmax(z[1], zs[1]));
        }
    }
    return w;
}

// The same with plain (unpacked) fp32 instructions: scs / shs = scale * slope, shift * slope, so LeakyReLU is max of two FMAs
// (beside MFMAs on the same SIMD a packed fp32 instruction costs several times a plain one).
__device__ __forceinline__ cl_u32x4 cl_transform8_plain(const cl_u32x4 u, const float (&sc)[8], const float (&sh)[8], const float (&scs)[8],
                                                        const float (&shs)[8]) {
    cl_u32x4 w;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float x0 = __builtin_bit_cast(float, u[p] << 16), x1 = __builtin_bit_cast(float, u[p] & 0xFFFF0000u);
        float z0, z1, y0, y1;
        asm("v_fma_f32 %0, %1, %2, %3" : "=v"(z0) : "v"(x0), "v"(sc[2 * p]), "v"(sh[2 * p]));
        asm("v_fma_f32 %0, %1, %2, %3" : "=v"(y0) : "v"(x0), "v"(scs[2 * p]), "v"(shs[2 * p]));
        asm("v_fma_f32 %0, %1, %2, %3" : "=v"(z1) : "v"(x1), "v"(sc[2 * p + 1]), "v"(sh[2 * p + 1]));
        asm("v_fma_f32 %0, %1, %2, %3" : "=v"(y1) : "v"(x1), "v"(scs[2 * p + 1]), "v"(shs[2 * p + 1]));
        w[p] = cl_pack2(cl_vmax(z0, y0), cl_vmax(z1, y1));
    }
    return w;
}

// Channels of accumulator register r of v_mfma_f32_32x32x16_bf16's D tile: row (r&3) + 8*(r>>2) + 4*(lane>>5).
// One M-tile (32 voxels on the lanes, 32 output channels in the 16 registers of both lane halves) -> CL16 rows:
// registers pairwise to bf16, the two lane halves exchange 8-byte groups (v_permlane32_swap) so that every lane holds
// 16 contiguous bytes, two 16-byte stores per lane.  yv: address of this lane's voxel in the plane of channels cob .. cob+7
// (cl_index), plane: halfs from one 8-channel plane to the next (8 x voxels of the sample).
__device__ __forceinline__ size_t cl_index(size_t n, int C, size_t S, int c8, size_t v) { return ((n * (size_t)(C >> 3) + (size_t)c8) * S + v) * 8; }
template <bool NTS = false>     // NTS: streaming stores (conv3d_common.h): the first layer only
__device__ __forceinline__ void cl_store_mtile(const f32x16& acc, unsigned short* yv, int kh, size_t plane) {
    unsigned d[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) d[i] = cl_pack2(acc[2 * i], acc[2 * i + 1]);
    // d[0..1]: channels 0-3 (+4 on the upper half), d[2..3]: 8-11, d[4..5]: 16-19, d[6..7]: 24-27
#pragma unroll
    for (int gp = 0; gp < 2; ++gp) {
        const auto r0 = __builtin_amdgcn_permlane32_swap(d[4 * gp + 0], d[4 * gp + 2], false, false);
        const auto r1 = __builtin_amdgcn_permlane32_swap(d[4 * gp + 1], d[4 * gp + 3], false, false);
        // lower half: channels 16gp .. 16gp+7, upper half: 16gp+8 .. 16gp+15
        const cl_u32x4 v = {r0[0], r1[0], r0[1], r1[1]};
        if constexpr (NTS) __builtin_nontemporal_store(v, reinterpret_cast<cl_u32x4*>(yv + (size_t)(2 * gp + kh) * plane));
        else *reinterpret_cast<cl_u32x4*>(yv + (size_t)(2 * gp + kh) * plane) = v;
    }
}

// InstanceNorm moments of one wave's part of a tile (conv_epilogue's one-pass form about a per-row shift): acc[m][r]
// over the MT M-tiles whose bit is set in vmask; cnt = number of valid voxels of the wave (all lanes).  On return the
// even lanes j hold (count, sum, M2) of output channel cob + cl_moment_channel(j, kh).
__device__ __forceinline__ int cl_moment_channel(int j, int kh) { const int r = j >> 1; return (r & 3) + 8 * (r >> 2) + 4 * kh; }
template <int MT>
__device__ __forceinline__ float4 cl_wave_moments(const f32x16 (&acc)[MT], unsigned vmask, int cnt, int j, int kh) {
    float vals[32], kshift[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float v0 = acc[0][r];
        const float k0 = __builtin_amdgcn_readlane(v0, 0), k1 = __builtin_amdgcn_readlane(v0, 32);
        const float K = kh ? k1 : k0;
        kshift[r] = K;
        float sd = 0.f, sq = 0.f;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const float d = ((vmask >> m) & 1u) ? acc[m][r] - K : 0.f;
            sd += d;
            sq = fmaf(d, d, sq);
        }
        vals[2 * r] = sd; vals[2 * r + 1] = sq;
    }
    halve_step<16>(vals, j); halve_step<8>(vals, j); halve_step<4>(vals, j); halve_step<2>(vals, j); halve_step<1>(vals, j);
    const float mine = vals[0];                       // lane j: (j&1) ? sum d^2 : sum d, of register j>>1
    const float other = __shfl_xor(mine, 1);
    float Kr = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) Kr = ((j >> 1) == r) ? kshift[r] : Kr;
    const float fc = (float)cnt;
    const float sd = mine, sq = other;                // meaningful on even lanes
    const float m2 = cnt > 0 ? fmaxf(sq - sd * sd / fc, 0.f) : 0.f;
    return make_float4(fc, fmaf(fc, Kr, sd), m2, 0.f);
}
// ... written as record pidx of [N][Cout][npart][4]
template <int MT>
__device__ __forceinline__ void cl_moments(float* part, int npart, int Cout, int n, int cob, int pidx, const f32x16 (&acc)[MT],
                                           unsigned vmask, int cnt, int j, int kh) {
    const float4 rec = cl_wave_moments<MT>(acc, vmask, cnt, j, kh);
    if ((j & 1) == 0) {
        const int co = cob + cl_moment_channel(j, kh);
        if (co < Cout) reinterpret_cast<float4*>(part)[((size_t)n * Cout + co) * npart + pidx] = rec;
    }
}
// Chan's merge of two (count, sum, M2) records, a then b (fixed order: deterministic)
__device__ __forceinline__ float4 cl_merge_moments(const float4 a, const float4 b) {
    const float n = a.x + b.x;
    if (b.x == 0.f) return a;
    if (a.x == 0.f) return b;
    const float d = b.y / b.x - a.y / a.x;
    return make_float4(n, a.y + b.y, a.z + b.z + d * d * (a.x * b.x / n), 0.f);
}
