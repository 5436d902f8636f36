// convT_k2s2_f16x2.hip -- ConvTranspose3d(kernel (kt,2,2), stride (kt,2,2), bias=False) in the f16x2 "pair" mode: fp32 tensors and
// accumulation, every product formed from fp16 pairs (conv3d_common.h).
//
//   out[n, co, kt d + i, 2h + j, 2w + k] = sum_ci x'[n, ci, d, h, w] * W[ci, co, i, j, k]
//
// The same GEMM as convT_k2s2.hip -- rows = (co, tap) (R = 4 kt Cout), columns = input voxels, K = Cin -- on
// v_mfma_f32_32x32x16_f16.  Both operands are one dword per (row | voxel, ci), (hi | lo << 16): hi = the value cut to fp16's 11
// significant bits, lo = the remainder rounded to nearest; the weights are multiplied by 2^8 before the split (their low parts stay in
// fp16's normal range), undone on the accumulators.  The 16 k-slots of an instruction carry 8 input channels x (hi, lo):
//      A  = 8 ci x (w_hi, w_lo) as stored      ->  sum w_hi x_hi + w_lo x_lo
//      A' = the same dwords rotated by 16 bits  ->  sum w_lo x_hi + w_hi x_lo
// two instructions per 8 channels instead of four v_mfma_f32_32x32x2_f32 of twice the length: the fp32-instruction kernel is bound by
// its matrix time (1.03 ms for 64 -> 32 at 8 x 64^3 against an HBM floor of 0.6), this one by the tensors it reads and writes.
// * B: the input tile [CKC cin][128 voxels] goes through LDS as [cin / 4][voxel][4 ci x (hi, lo)] -- a lane's operand is one
//   ds_read_b128, consecutive lanes consecutive 16-byte pieces -- with the producer's InstanceNorm + LeakyReLU and the split applied
//   on the way in (a thread stages one voxel of 16 channels per chunk: scalar loads, coalesced across the lanes);
// * A: fragment-major weight image [row tile][cin / 8][lane][4 dwords], one 16-byte load per lane and 8 channels, straight from L2;
// * epilogue as in convT_k2s2.hip: rows 2t, 2t + 1 are the taps k = 0, 1 of one output position -> 8-byte stores, 256 contiguous
//   bytes per half-wave.
// Stands in for Generic_UNet.tu[u] = torch.nn.ConvTranspose3d [nnU-Net, not in tree].
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "wmh_common.h"

typedef float ctp_f32x16 __attribute__((ext_vector_type(16)));
typedef float ctp_f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 ctp_f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 ctp_f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned ctp_u32x4 __attribute__((ext_vector_type(4)));
#define CTP_WSCALE 256.f

struct ConvTPArgs {
    const float* x; const float* ss; const unsigned* wf; float* y;
    int N, Cin, Cout, D, H, W;
    int nrg;                 // row groups (256 rows each)
    long long nvox;          // N*D*H*W
    float slope;
};

// v -> (hi | lo << 16): hi = v cut to 11 significant bits (exact in fp16's normal range), lo = the remainder to nearest
// (conv3d_k3_wgrad_f16x2.hip::wp_split: three instructions, none of them packed fp32)
__device__ __forceinline__ unsigned ctp_split(float v) {
    const float hi = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) & 0xFFFFE000u);
    const float lo = v - hi;
    return __builtin_bit_cast(unsigned, __builtin_convertvector(ctp_f32x2{hi, lo}, ctp_f16x2));
}

// NTS: streaming stores (an output far larger than the 256 MB Infinity Cache: conv3d_common.h::conv_epilogue)
template <int NT, int MT, int CKC, int KT, bool NTS = false>
__global__ __launch_bounds__(256, MT <= 2 ? 4 : 2) void convT_k2s2_f16x2_kernel(const ConvTPArgs a) {
    constexpr int NTAP = 4 * KT;                  // rows per output channel
    constexpr int VT = MT * 32;                   // voxels per workgroup tile
    constexpr int NG = CKC / 4;                   // 4-channel groups per chunk
    constexpr int PPT = NG * VT / 256;            // 16-byte pieces staged per thread per chunk
    static_assert(NG * VT % 256 == 0 && 256 % VT == 0 && CKC % 8 == 0, "staging shape");
    extern __shared__ __attribute__((aligned(16))) ctp_u32x4 ldsq[];   // [2][NG][VT] pieces of 4 ci x (hi, lo)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, kh = lane >> 5;
    const int rg = blockIdx.x % a.nrg;
    const long long gv0 = (long long)(blockIdx.x / a.nrg) * VT;
    const long long DHW = (long long)a.D * a.H * a.W;
    const int nchunks = (a.Cin + CKC - 1) / CKC;

    // staging map: this thread's voxel (the same for all its pieces) and first group; piece it is group g0 + (256 / VT) it
    const int sv = tid % VT, g0 = tid / VT;
    const long long sgv = gv0 + sv;
    const bool s_ok = sgv < a.nvox;
    const long long sgvc = s_ok ? sgv : 0;
    const int s_n = (int)(sgvc / DHW);
    const float* xb = a.x + (size_t)s_n * a.Cin * DHW + (sgvc - (long long)s_n * DHW);
    float stg[PPT][4];
    auto load_chunk = [&](int ch) {
#pragma unroll
        for (int it = 0; it < PPT; ++it)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ci = min(ch * CKC + 4 * (g0 + (256 / VT) * it) + e, a.Cin - 1);
                stg[it][e] = xb[(size_t)ci * DHW];
            }
    };
    auto write_chunk = [&](int ch, ctp_u32x4* buf) {
#pragma unroll
        for (int it = 0; it < PPT; ++it) {
            const int g = g0 + (256 / VT) * it;
            ctp_u32x4 piece;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int cil = ch * CKC + 4 * g + e;
                const int ci = min(cil, a.Cin - 1);
                float v = stg[it][e];
                if (a.ss != nullptr) {
                    const float2 q = *reinterpret_cast<const float2*>(a.ss + ((size_t)s_n * a.Cin + ci) * 2);
                    v = wmh_lrelu(fmaf(v, q.x, q.y), a.slope);
                }
                piece[e] = (s_ok && cil < a.Cin) ? ctp_split(v) : 0u;
            }
            buf[g * VT + sv] = piece;
        }
    };

    ctp_f32x16 acc[NT][MT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][m][r] = 0.f;

    // weights: row tile rt = (rg*4 + wave)*NT + nt ; [rt][Cin8][lane][4 dwords], dword e of lane (row, kh) = channel 8 q + 4 kh + e
    const int Cin8 = (a.Cin + 7) / 8;
    const ctp_u32x4* wl = reinterpret_cast<const ctp_u32x4*>(a.wf) + ((size_t)(rg * 4 + wave) * NT * Cin8) * 64 + lane;
    ctp_u32x4 a_cur[NT], a_nxt[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) a_cur[nt] = wl[((size_t)nt * Cin8 + 0) * 64];

    load_chunk(0);
    write_chunk(0, ldsq);
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
        const ctp_u32x4* cur = ldsq + (ch & 1) * (NG * VT);
        ctp_u32x4* nxt = ldsq + ((ch & 1) ^ 1) * (NG * VT);
        const int chn = min(ch + 1, nchunks - 1);
        load_chunk(chn);
#pragma unroll
        for (int q = 0; q < CKC / 8; ++q) {
            const int Q = min(ch * (CKC / 8) + q + 1, Cin8 - 1);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) a_nxt[nt] = wl[((size_t)nt * Cin8 + Q) * 64];
            __builtin_amdgcn_sched_barrier(0x38F);       // keep the prefetch here (VMEM may not cross)
            ctp_u32x4 ar[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int d = 0; d < 4; ++d) ar[nt][d] = __builtin_amdgcn_alignbit(a_cur[nt][d], a_cur[nt][d], 16);
            // channels of the chunk that do not exist were staged as zeros, their weight dwords are zero in the image
            if (ch * CKC + 8 * q < a.Cin) {
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const ctp_u32x4 bv = cur[(2 * q + kh) * VT + 32 * m + j];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        acc[nt][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(ctp_f16x8, a_cur[nt]), __builtin_bit_cast(ctp_f16x8, bv), acc[nt][m], 0, 0, 0);
                        acc[nt][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(ctp_f16x8, ar[nt]), __builtin_bit_cast(ctp_f16x8, bv), acc[nt][m], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) a_cur[nt] = a_nxt[nt];
        }
        write_chunk(chn, nxt);
        __syncthreads();
    }

    // epilogue (as convT_k2s2.hip): accumulator rows r and r+1 (r even) are the taps k = 0, 1 of one output position -> float2
    const int H2 = 2 * a.H, W2 = 2 * a.W;
    const size_t O = (size_t)NTAP * DHW;
    constexpr float inv = 1.f / CTP_WSCALE;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const long long gv = gv0 + 32 * m + j;
        if (gv >= a.nvox) continue;
        const int n = (int)(gv / DHW);
        const long long v = gv - (long long)n * DHW;
        const int wq = (int)(v % a.W), hq = (int)((v / a.W) % a.H), dq = (int)(v / ((long long)a.W * a.H));
        const size_t base = ((size_t)(KT == 2 ? 2 * dq + kh : dq) * H2 + 2 * hq) * W2 + 2 * wq;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int co0 = ((rg * 4 + wave) * NT + nt) * (32 / NTAP);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int co = co0 + (KT == 2 ? c : 2 * c + kh);
                if (co >= a.Cout) continue;
                float* yp = a.y + ((size_t)n * a.Cout + co) * O + base;
                typedef float ctp_st2 __attribute__((ext_vector_type(2)));
                const ctp_st2 v0 = {acc[nt][m][4 * c + 0] * inv, acc[nt][m][4 * c + 1] * inv}, v1 = {acc[nt][m][4 * c + 2] * inv, acc[nt][m][4 * c + 3] * inv};
                if constexpr (NTS) {
                    __builtin_nontemporal_store(v0, reinterpret_cast<ctp_st2*>(yp));
                    __builtin_nontemporal_store(v1, reinterpret_cast<ctp_st2*>(yp + W2));
                } else {
                    *reinterpret_cast<ctp_st2*>(yp) = v0;
                    *reinterpret_cast<ctp_st2*>(yp + W2) = v1;
                }
            }
        }
    }
}

static inline int ctp_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// dwords of the weight image (the fp32 image's element count: one dword per (row, channel) of the padded tiles)
extern "C" size_t wmh_convtranspose3d_packed_dwords_f16x2(int Cin, int Cout, int kt) {
    if (Cin < 1 || Cout < 1 || (kt != 1 && kt != 2)) return 0;
    const int RT = ctp_cdiv((long long)Cout * 4 * kt, 256) * 8, Cin8 = (Cin + 7) / 8;
    return (size_t)RT * Cin8 * 64 * 4;
}
// w = torch layout [Cin][Cout][kt][2][2] in HBM -> wf, fragment-major [row tile][Cin/8][lane][4]: dword e of lane (row, kh) is the
// pair of 2^8 w[8 q + 4 kh + e][row] (row = co * 4 kt + tap); |2^8 w| is clamped to 65000 (a transposed-convolution weight of 254
// is not a trained network)
__global__ __launch_bounds__(256) void convT_pack_f16x2_kernel(const float* w, unsigned* wf, int Cin, int R, int Cin8, size_t total) {
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        size_t r = idx;
        const int e = (int)(r % 4); r /= 4;
        const int lane = (int)(r % 64); r /= 64;
        const int q = (int)(r % Cin8);
        const int rt = (int)(r / Cin8);
        const int row = rt * 32 + (lane & 31), ci = 8 * q + 4 * (lane >> 5) + e;
        float v = (row < R && ci < Cin) ? CTP_WSCALE * w[(size_t)ci * R + row] : 0.f;
        v = fminf(fmaxf(v, -65000.f), 65000.f);
        wf[idx] = ctp_split(v);
    }
}
extern "C" int wmh_convtranspose3d_pack_weights_f16x2_device(const float* w, int Cin, int Cout, int kt, void* wf, void* stream) {
    if (!w || !wf || Cin < 1 || Cout < 1 || (kt != 1 && kt != 2)) return wmh_fail(WMH_EINVAL, "convT pack f16x2 (device): bad args");
    const int R = Cout * 4 * kt, Cin8 = (Cin + 7) / 8;
    const size_t total = wmh_convtranspose3d_packed_dwords_f16x2(Cin, Cout, kt);
    unsigned nb = (unsigned)((total + 255) / 256); if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(convT_pack_f16x2_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, w, static_cast<unsigned*>(wf), Cin, R, Cin8, total);
    return wmh_check_launch("convT_pack_f16x2_kernel");
}

extern "C" int wmh_convtranspose3d_fwd_f16x2(const float* x, const float* ss, const void* wf, float* y, int N, int Cin, int Cout,
                                             int D, int H, int W, int kt, float lrelu_slope, void* stream) {
    if (!x || !wf || !y || N < 1 || Cin < 1 || Cout < 1 || D < 1 || H < 1 || W < 1 || (kt != 1 && kt != 2))
        return wmh_fail(WMH_EINVAL, "convtranspose3d_fwd_f16x2: bad args");
    if (reinterpret_cast<uintptr_t>(y) & 7) return wmh_fail(WMH_EINVAL, "convtranspose3d_fwd_f16x2: y must be 8-byte aligned");
    const long long DHW = (long long)D * H * W;
    constexpr int NT = 2, CKC = 32;
    // 64 voxels per workgroup, four workgroups per CU (measured: 0.78 / 0.22 / 0.09 ms for the three wide levels at batch 8 against
    // 0.82 / 0.26 / 0.10 with 128 voxels and two workgroups; WMH_CONVT_MT=4 selects the latter)
    static const int MT = wmh_diag_env("WMH_CONVT_MT", 2) == 4 ? 4 : 2;
    ConvTPArgs a;
    a.x = x; a.ss = ss; a.wf = static_cast<const unsigned*>(wf); a.y = y; a.N = N; a.Cin = Cin; a.Cout = Cout; a.D = D; a.H = H; a.W = W;
    a.nrg = ctp_cdiv((long long)Cout * 4 * kt, 256);
    a.nvox = (long long)N * DHW;
    a.slope = lrelu_slope;
    const long long nblk = (long long)ctp_cdiv(a.nvox, MT * 32) * a.nrg;
    if (nblk > 0x7fffffffLL) return wmh_fail(WMH_EINVAL, "convtranspose3d_fwd_f16x2: grid too large");
    const int lds = 2 * (CKC / 4) * MT * 32 * 16;
    // streaming stores: 0.82 -> 0.78 ms for the 2 GB output of the full-resolution level in isolation, nothing in the network (the
    // next convolution's loaders wait for HBM either way): off unless WMH_CONVT_NTS=1
    static const bool nts = wmh_diag_env("WMH_CONVT_NTS", 0) != 0;
    if (kt == 2 && MT == 4 && nts) hipLaunchKernelGGL((convT_k2s2_f16x2_kernel<NT, 4, CKC, 2, true>), dim3((unsigned)nblk), dim3(256), lds, (hipStream_t)stream, a);
    else if (kt == 2 && MT == 2 && nts) hipLaunchKernelGGL((convT_k2s2_f16x2_kernel<NT, 2, CKC, 2, true>), dim3((unsigned)nblk), dim3(256), lds, (hipStream_t)stream, a);
    else if (kt == 2 && MT == 2) hipLaunchKernelGGL((convT_k2s2_f16x2_kernel<NT, 2, CKC, 2>), dim3((unsigned)nblk), dim3(256), lds, (hipStream_t)stream, a);
    else if (kt == 2) hipLaunchKernelGGL((convT_k2s2_f16x2_kernel<NT, 4, CKC, 2>), dim3((unsigned)nblk), dim3(256), lds, (hipStream_t)stream, a);
    else         hipLaunchKernelGGL((convT_k2s2_f16x2_kernel<NT, 4, CKC, 1>), dim3((unsigned)ctp_cdiv(a.nvox, 128) * a.nrg), dim3(256), 2 * (CKC / 4) * 128 * 16, (hipStream_t)stream, a);
    return wmh_check_launch("convT_k2s2_f16x2_kernel");
}
