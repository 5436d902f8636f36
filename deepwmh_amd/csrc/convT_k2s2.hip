// convT_k2s2.hip -- ConvTranspose3d(kernel 2, stride 2, bias=False) for gfx950 on the exact-fp32 MFMA.
//
//   out[n, co, 2d+i, 2h+j, 2w+k] = sum_ci x'[n, ci, d, h, w] * W[ci, co, i, j, k]
//
// Non-overlapping taps make this a plain GEMM: rows = (co, tap) (R = 8*Cout), columns = input voxels,
// K = Cin.  D[row 32][voxel 32] += A[row 32][k 2] * B[k 2][voxel 32] (v_mfma_f32_32x32x2_f32):
// * B: the input tile [CKC cin][128 voxels] is staged through LDS (double buffered, 16-byte loads and LDS
//   writes, no halo) with the producer's InstanceNorm+LeakyReLU applied on the way in;
// * A: fragment-major packed weights [row tile][cin/8][lane][4]: one 16-byte load per lane feeds 4 k-steps;
// * rows 2t and 2t+1 of the accumulator are the taps k = 0,1 of one output position -> 8-byte stores that
//   are contiguous across the 32 lanes of a half-wave.
// Stands in for Generic_UNet.tu[u] = torch.nn.ConvTranspose3d [nnU-Net, not in tree].
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "wmh_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvTArgs {
    const float* x; const float* ss; const float* wf; float* y;
    int N, Cin, Cout, D, H, W;
    int nrg;                 // row groups (256 rows each)
    long long nvox;          // N*D*H*W
    float slope;
};

// KT: kernel/stride along the first spatial axis: 2, or 1 for nnU-Net's anisotropic (1,2,2) up-sampling
// OUT16: y is a bf16 tensor (the two taps of a position become one 4-byte store)
template <int NT, int MT, int CKC, int KT, bool OUT16 = false>
__global__ __launch_bounds__(256, 2) void convT_k2s2_mfma_kernel(const ConvTArgs a) {
    constexpr int NTAP = 4 * KT;                  // rows per output channel
    constexpr int VT = MT * 32;                  // voxels per workgroup tile
    constexpr int F4 = CKC * VT / 4 / 256;       // float4 staged per thread per chunk
    static_assert(CKC * VT / 4 % 256 == 0 && CKC % 8 == 0, "staging shape");
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [2][CKC][VT]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, kh = lane >> 5;
    const int rg = blockIdx.x % a.nrg;
    const long long gv0 = (long long)(blockIdx.x / a.nrg) * VT;
    const long long DHW = (long long)a.D * a.H * a.W;
    const int nchunks = a.Cin / CKC + ((a.Cin % CKC) ? 1 : 0);

    // staging map (per thread, chunk independent): float4 f -> (cin within chunk, 4 voxels)
    int s_ci[F4]; long long s_off[F4]; int s_n[F4]; bool s_ok[F4];
#pragma unroll
    for (int it = 0; it < F4; ++it) {
        const int f = tid + it * 256;
        s_ci[it] = f / (VT / 4);
        const long long gv = gv0 + 4 * (f % (VT / 4));
        s_ok[it] = gv < a.nvox;
        const long long gvc = s_ok[it] ? gv : 0;
        s_n[it] = (int)(gvc / DHW);
        s_off[it] = gvc - (long long)s_n[it] * DHW;
    }
    float4 stg[F4];
    auto load_chunk = [&](int ch) {
#pragma unroll
        for (int it = 0; it < F4; ++it) {
            const int ci = min(ch * CKC + s_ci[it], a.Cin - 1);
            stg[it] = *reinterpret_cast<const float4*>(a.x + ((size_t)s_n[it] * a.Cin + ci) * DHW + s_off[it]);
        }
    };
    auto write_chunk = [&](int ch, float* buf) {
#pragma unroll
        for (int it = 0; it < F4; ++it) {
            const int cil = ch * CKC + s_ci[it];
            const int ci = min(cil, a.Cin - 1);
            float4 v = stg[it];
            if (a.ss != nullptr) {
                const float2 q = *reinterpret_cast<const float2*>(a.ss + ((size_t)s_n[it] * a.Cin + ci) * 2);
                v.x = wmh_lrelu(fmaf(v.x, q.x, q.y), a.slope); v.y = wmh_lrelu(fmaf(v.y, q.x, q.y), a.slope);
                v.z = wmh_lrelu(fmaf(v.z, q.x, q.y), a.slope); v.w = wmh_lrelu(fmaf(v.w, q.x, q.y), a.slope);
            }
            if (!s_ok[it] || cil >= a.Cin) v = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(buf + (tid + it * 256) * 4) = v;      // [ci][VT] is linear in f
        }
    };

    f32x16 acc[NT][MT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][m][r] = 0.f;

    // weights: row tile rt = (rg*4 + wave)*NT + nt ; [rt][Cin8][lane][4]
    const int Cin8 = (a.Cin + 7) / 8;
    const float4* wl = reinterpret_cast<const float4*>(a.wf) + ((size_t)(rg * 4 + wave) * NT * Cin8) * 64 + lane;
    const int Qtot = Cin8;                         // number of 4-k-step groups overall
    float4 a_cur[NT], a_nxt[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) a_cur[nt] = wl[((size_t)nt * Cin8 + 0) * 64];

    load_chunk(0);
    write_chunk(0, lds);
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
        const float* cur = lds + (ch & 1) * (CKC * VT);
        float* nxt = lds + ((ch & 1) ^ 1) * (CKC * VT);
        const int chn = min(ch + 1, nchunks - 1);
        load_chunk(chn);
#pragma unroll
        for (int q = 0; q < CKC / 8; ++q) {
            const int Q = min(ch * (CKC / 8) + q + 1, Qtot - 1);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) a_nxt[nt] = wl[((size_t)nt * Cin8 + Q) * 64];
            __builtin_amdgcn_sched_barrier(0x38F);       // keep the prefetch here (VMEM may not cross)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const float bv = cur[(2 * (4 * q + e) + kh) * VT + 32 * m + j];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const float av = e == 0 ? a_cur[nt].x : e == 1 ? a_cur[nt].y : e == 2 ? a_cur[nt].z : a_cur[nt].w;
                        acc[nt][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[nt][m], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) a_cur[nt] = a_nxt[nt];
        }
        write_chunk(chn, nxt);
        __syncthreads();
    }

    // epilogue: accumulator rows r and r+1 (r even) are the taps k = 0, 1 of one output position -> float2;
    // lanes run along w -> 256 contiguous bytes per half-wave.
    //   KT = 2: row = co_l*8 + t, t = 4*kh + (r&3) -> tap i = kh, j = (r>>1)&1 ; co_l = r>>2          (4 couts per tile)
    //   KT = 1: row = co_l*4 + t, t = r&3          -> tap j = (r>>1)&1           ; co_l = 2*(r>>2) + kh (8 couts per tile)
    const int H2 = 2 * a.H, W2 = 2 * a.W;
    const size_t O = (size_t)NTAP * DHW;
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
                if constexpr (OUT16) {
                    typedef __bf16 eb2 __attribute__((ext_vector_type(2)));
                    typedef float ef2 __attribute__((ext_vector_type(2)));
                    unsigned short* yp = reinterpret_cast<unsigned short*>(a.y) + ((size_t)n * a.Cout + co) * O + base;
                    const ef2 f0 = {acc[nt][m][4 * c + 0], acc[nt][m][4 * c + 1]}, f1 = {acc[nt][m][4 * c + 2], acc[nt][m][4 * c + 3]};
                    *reinterpret_cast<unsigned*>(yp) = __builtin_bit_cast(unsigned, __builtin_convertvector(f0, eb2));
                    *reinterpret_cast<unsigned*>(yp + W2) = __builtin_bit_cast(unsigned, __builtin_convertvector(f1, eb2));
                    continue;
                }
                float* yp = a.y + ((size_t)n * a.Cout + co) * O + base;
                *reinterpret_cast<float2*>(yp) = make_float2(acc[nt][m][4 * c + 0], acc[nt][m][4 * c + 1]);
                *reinterpret_cast<float2*>(yp + W2) = make_float2(acc[nt][m][4 * c + 2], acc[nt][m][4 * c + 3]);
            }
        }
    }
}

static inline int cdivi(long long a, long long b) { return (int)((a + b - 1) / b); }

static size_t convT_packed_floats(int Cin, int Cout, int kt) {
    const int RT = cdivi((long long)Cout * 4 * kt, 256) * 8, Cin8 = (Cin + 7) / 8;
    return (size_t)RT * Cin8 * 64 * 4;
}
extern "C" size_t wmh_convtranspose3d_k2s2_packed_floats(int Cin, int Cout) { return convT_packed_floats(Cin, Cout, 2); }
extern "C" size_t wmh_convtranspose3d_packed_floats(int Cin, int Cout, int kt) { return (kt == 1 || kt == 2) ? convT_packed_floats(Cin, Cout, kt) : 0; }
// w: host torch layout [Cin][Cout][2][2][2]; wf: host out, fragment-major [row tile][Cin/8][lane][4]
// kt = kernel/stride along the first spatial axis (2, or 1 for (1,2,2)); w = torch layout [Cin][Cout][kt][2][2]
extern "C" int wmh_convtranspose3d_pack_weights(const float* w, int Cin, int Cout, int kt, float* wf) {
    if (!w || !wf || Cin < 1 || Cout < 1 || (kt != 1 && kt != 2)) return wmh_fail(WMH_EINVAL, "convT pack: bad args");
    const int R = Cout * 4 * kt, RT = cdivi(R, 256) * 8, Cin8 = (Cin + 7) / 8;
    for (int rt = 0; rt < RT; ++rt)
        for (int q = 0; q < Cin8; ++q)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 4; ++e) {
                    const int row = rt * 32 + (lane & 31), ci = 2 * (4 * q + e) + (lane >> 5);   // row = co*(4*kt) + tap
                    float v = 0.f;
                    if (row < R && ci < Cin) v = w[(size_t)ci * R + row];
                    wf[(((size_t)rt * Cin8 + q) * 64 + lane) * 4 + e] = v;
                }
    return WMH_OK;
}

// the same image written on the device (training: the weights change every step)
__global__ __launch_bounds__(256) void convT_pack_kernel(const float* w, float* wf, int Cin, int R, int Cin8, size_t total) {
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        size_t r = idx;
        const int e = (int)(r % 4); r /= 4;
        const int lane = (int)(r % 64); r /= 64;
        const int q = (int)(r % Cin8);
        const int rt = (int)(r / Cin8);
        const int row = rt * 32 + (lane & 31), ci = 2 * (4 * q + e) + (lane >> 5);
        wf[idx] = (row < R && ci < Cin) ? w[(size_t)ci * R + row] : 0.f;
    }
}
extern "C" int wmh_convtranspose3d_pack_weights_device(const float* w, int Cin, int Cout, int kt, float* wf, void* stream) {
    if (!w || !wf || Cin < 1 || Cout < 1 || (kt != 1 && kt != 2)) return wmh_fail(WMH_EINVAL, "convT pack (device): bad args");
    const int R = Cout * 4 * kt, Cin8 = (Cin + 7) / 8;
    const size_t total = convT_packed_floats(Cin, Cout, kt);
    unsigned nb = (unsigned)((total + 255) / 256); if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(convT_pack_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, w, wf, Cin, R, Cin8, total);
    return wmh_check_launch("convT_pack_kernel");
}

extern "C" int wmh_convtranspose3d_k2s2_pack_weights(const float* w, int Cin, int Cout, float* wf) {
    return wmh_convtranspose3d_pack_weights(w, Cin, Cout, 2, wf);
}

static int convT_mfma_fwd(const float* x, const float* ss, const float* wf, float* y, int N, int Cin,
                          int Cout, int D, int H, int W, int kt, float lrelu_slope, int out16, void* stream) {
    if (!x || !wf || !y || N < 1 || Cin < 1 || Cout < 1 || D < 1 || H < 1 || W < 1 || (kt != 1 && kt != 2))
        return wmh_fail(WMH_EINVAL, "convtranspose3d_mfma_fwd: bad args");
    const long long DHW = (long long)D * H * W;
    if (DHW % 4 != 0 || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(y) & 7))
        return wmh_fail(WMH_EINVAL, "convtranspose3d_k2s2_mfma_fwd: needs D*H*W % 4 == 0 and 16-byte aligned x");
    constexpr int NT = 2, MT = 4, CKC = 32;
    ConvTArgs a;
    a.x = x; a.ss = ss; a.wf = wf; a.y = y; a.N = N; a.Cin = Cin; a.Cout = Cout; a.D = D; a.H = H; a.W = W;
    a.nrg = cdivi((long long)Cout * 4 * kt, 256);
    a.nvox = (long long)N * DHW;
    a.slope = lrelu_slope;
    const long long nblk = (long long)cdivi(a.nvox, MT * 32) * a.nrg;
    if (nblk > 0x7fffffffLL) return wmh_fail(WMH_EINVAL, "convtranspose3d_k2s2_mfma_fwd: grid too large");
    const int lds = 2 * CKC * MT * 32 * 4;
    if (out16 && kt == 2) hipLaunchKernelGGL((convT_k2s2_mfma_kernel<NT, MT, CKC, 2, true>), dim3((unsigned)nblk), dim3(256), lds, (hipStream_t)stream, a);
    else if (out16)       hipLaunchKernelGGL((convT_k2s2_mfma_kernel<NT, MT, CKC, 1, true>), dim3((unsigned)nblk), dim3(256), lds, (hipStream_t)stream, a);
    else if (kt == 2) hipLaunchKernelGGL((convT_k2s2_mfma_kernel<NT, MT, CKC, 2>), dim3((unsigned)nblk), dim3(256), lds, (hipStream_t)stream, a);
    else         hipLaunchKernelGGL((convT_k2s2_mfma_kernel<NT, MT, CKC, 1>), dim3((unsigned)nblk), dim3(256), lds, (hipStream_t)stream, a);
    return wmh_check_launch("convT_k2s2_mfma_kernel");
}
extern "C" int wmh_convtranspose3d_mfma_fwd(const float* x, const float* ss, const float* wf, float* y, int N, int Cin,
                                            int Cout, int D, int H, int W, int kt, float lrelu_slope, void* stream) {
    return convT_mfma_fwd(x, ss, wf, y, N, Cin, Cout, D, H, W, kt, lrelu_slope, 0, stream);
}
// the same with y stored as a bf16 tensor (uint16 bit patterns) when y_bf16 != 0
extern "C" int wmh_convtranspose3d_mfma_fwd_st(const float* x, const float* ss, const float* wf, void* y, int N, int Cin,
                                               int Cout, int D, int H, int W, int kt, float lrelu_slope, int y_bf16, void* stream) {
    return convT_mfma_fwd(x, ss, wf, static_cast<float*>(y), N, Cin, Cout, D, H, W, kt, lrelu_slope, y_bf16 != 0, stream);
}
extern "C" int wmh_convtranspose3d_k2s2_mfma_fwd(const float* x, const float* ss, const float* wf, float* y, int N,
                                                 int Cin, int Cout, int D, int H, int W, float lrelu_slope,
                                                 void* stream) {
    return wmh_convtranspose3d_mfma_fwd(x, ss, wf, y, N, Cin, Cout, D, H, W, 2, lrelu_slope, stream);
}
