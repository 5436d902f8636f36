// train_kernels.hip -- backward-pass and optimizer kernels of the nnU-Net training step
// (nnUNetTrainerV2.run_iteration [nnU-Net, not in tree]; reference call sites
// deepwmh/pipeline/DCNN_multistage.py:292-309,501-520): Dice+CE loss (deep supervision), segmentation-head and
// InstanceNorm+LeakyReLU backward, zero-insert upsampling (stride-2 data gradient), transposed-conv backward,
// gradient-norm clipping and SGD with Nesterov momentum.  Every reduction is a fixed-order tree (no atomics):
// a training step is bit-reproducible.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "wmh_common.h"

static inline unsigned cdivu(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

__device__ __forceinline__ double shfl_xor_d(double v, int m) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, m); hi = __shfl_xor(hi, m);
    return __hiloint2double(hi, lo);
}
// sum over the 256 threads of a block in a fixed order; result valid in every thread
__device__ __forceinline__ double block_sum_256(double v, double* sh /*[4]*/) {
    for (int m = 1; m < 64; m <<= 1) v += shfl_xor_d(v, m);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// =================================================================================================
// Dice + cross-entropy loss at one deep-supervision scale  (DC_and_CE_loss: batch dice, do_bg=False)
// =================================================================================================
// partial sums per block: [k] sum p_k*y_k, [K+k] sum p_k, [2K+k] sum y_k, [3K] sum -log p_y
template <int K>
__global__ __launch_bounds__(256) void loss_partials_kernel(const float* logits, const uint8_t* target, double* part,
                                                            int N, size_t S) {
    __shared__ double sh[4];
    const size_t M = (size_t)N * S;
    double acc[3 * K + 1];
#pragma unroll
    for (int q = 0; q < 3 * K + 1; ++q) acc[q] = 0.0;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < M; idx += (size_t)gridDim.x * 256) {
        const size_t n = idx / S, v = idx - n * S;
        float l[K], mx;
#pragma unroll
        for (int k = 0; k < K; ++k) l[k] = logits[(n * K + k) * S + v];
        mx = l[0];
#pragma unroll
        for (int k = 1; k < K; ++k) mx = fmaxf(mx, l[k]);
        float e[K], den = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) { e[k] = expf(l[k] - mx); den += e[k]; }
        const int y = target[idx];
        const float logden = logf(den);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float p = e[k] / den;
            acc[K + k] += p;
            if (k == y) { acc[k] += p; acc[2 * K + k] += 1.0; acc[3 * K] += (double)(logden - (l[k] - mx)); }
        }
    }
#pragma unroll
    for (int q = 0; q < 3 * K + 1; ++q) {
        const double t = block_sum_256(acc[q], sh);
        if (threadIdx.x == 0) part[(size_t)blockIdx.x * (3 * K + 1) + q] = t;
    }
}
// coef: [0] ce mean, [1] dice loss, [2+2c] num_c, [3+2c] den_c ; loss_out[0] += weight * (ce + dice)
template <int K>
__global__ void loss_finalize_kernel(const double* part, int nblk, double M, float smooth, double weight, double* coef,
                                     double* loss_out) {
    __shared__ double tot[3 * K + 1];
    // one wave; per quantity a fixed-order tree: lane l sums blocks l, l + 64, ..., then the lanes pairwise (a single lane walking
    // all 1024 partial records -- dependent loads -- took 53 us, five times per iteration)
#pragma unroll
    for (int q = 0; q < 3 * K + 1; ++q) {
        double s = 0.0;
        for (int b = threadIdx.x; b < nblk; b += 64) s += part[(size_t)b * (3 * K + 1) + q];
        for (int m = 1; m < 64; m <<= 1) s += shfl_xor_d(s, m);
        if (threadIdx.x == 0) tot[q] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double ce = tot[3 * K] / M;
        double dc = 0.0;
        for (int c = 0; c < K; ++c) {
            const double tp = tot[c], fp = tot[K + c] - tp, fn = tot[2 * K + c] - tp;
            const double num = 2 * tp + smooth, den = 2 * tp + fp + fn + smooth + 1e-8;
            coef[2 + 2 * c] = num; coef[3 + 2 * c] = den;
            if (c >= 1) dc += num / den;
        }
        dc = -dc / (K - 1);
        coef[0] = ce; coef[1] = dc;
        loss_out[0] += weight * (ce + dc);
    }
}
template <int K>
__global__ __launch_bounds__(256) void loss_bwd_kernel(const float* logits, const uint8_t* target, const double* coef,
                                                       float weight, float* dlogits, int N, size_t S) {
    const size_t M = (size_t)N * S;
    float gc[K];      // d(dice loss)/d p_c for y_c = 0 and extra term for y_c = 1
    float gy[K];
#pragma unroll
    for (int c = 0; c < K; ++c) {
        const double num = coef[2 + 2 * c], den = coef[3 + 2 * c];
        // d(num/den)/dp = (2y*den - num)/den^2 ; loss = -(1/(K-1)) sum_{c>=1}
        gc[c] = c >= 1 ? (float)(num / (den * den) / (K - 1)) : 0.f;                  // y = 0:  -( -num/den^2 )/(K-1)
        gy[c] = c >= 1 ? (float)(-(2.0 * den - num) / (den * den) / (K - 1)) : 0.f;   // y = 1
    }
    const float invM = (float)(1.0 / (double)M);
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < M; idx += (size_t)gridDim.x * 256) {
        const size_t n = idx / S, v = idx - n * S;
        float l[K], mx;
#pragma unroll
        for (int k = 0; k < K; ++k) l[k] = logits[(n * K + k) * S + v];
        mx = l[0];
#pragma unroll
        for (int k = 1; k < K; ++k) mx = fmaxf(mx, l[k]);
        float p[K], den = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) { p[k] = expf(l[k] - mx); den += p[k]; }
        const int y = target[idx];
        float g[K], dot = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) { p[k] /= den; g[k] = (k == y) ? gy[k] : gc[k]; dot += g[k] * p[k]; }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float ce = (p[k] - (k == y ? 1.f : 0.f)) * invM;
            dlogits[(n * K + k) * S + v] = weight * (ce + p[k] * (g[k] - dot));
        }
    }
}

extern "C" int wmh_dice_ce_loss_fwd_bwd(const float* logits, const uint8_t* target, int N, int K, size_t S, float weight,
                                        float smooth, float* dlogits, double* loss_accum, double* workspace,
                                        void* stream) {
    if (!logits || !target || !loss_accum || !workspace || N < 1 || S < 1) return wmh_fail(WMH_EINVAL, "dice_ce_loss: bad args");
    hipStream_t st = (hipStream_t)stream;
    const size_t M = (size_t)N * S;
    unsigned nblk = cdivu(M, 256 * 8); if (nblk > 1024) nblk = 1024; if (nblk < 1) nblk = 1;
    double* coef = workspace;                  // [2 + 2K]
    double* part = workspace + 16;             // [nblk][3K+1]
#define LOSS_K(KK)                                                                                                   \
    hipLaunchKernelGGL((loss_partials_kernel<KK>), dim3(nblk), dim3(256), 0, st, logits, target, part, N, S);         \
    hipLaunchKernelGGL((loss_finalize_kernel<KK>), dim3(1), dim3(64), 0, st, part, (int)nblk, (double)M, smooth,       \
                       (double)weight, coef, loss_accum);                                                            \
    if (dlogits) hipLaunchKernelGGL((loss_bwd_kernel<KK>), dim3(cdivu(M, 256 * 4) > 4096 ? 4096 : cdivu(M, 256 * 4)), \
                                    dim3(256), 0, st, logits, target, coef, weight, dlogits, N, S)
    switch (K) {
        case 2: LOSS_K(2); break;
        case 3: LOSS_K(3); break;
        case 4: LOSS_K(4); break;
        default: return wmh_fail(WMH_EINVAL, "dice_ce_loss: K must be 2, 3 or 4");
    }
#undef LOSS_K
    return wmh_check_launch("dice_ce_loss kernels");
}
extern "C" size_t wmh_dice_ce_loss_workspace_doubles(void) { return 16 + 1024 * 13; }

// =================================================================================================
// Online evaluation of a validation batch (nnUNetTrainer.run_online_evaluation [nnU-Net, not in tree]; what decides `model_best`,
// which DeepWMH releases and predicts with: /root/reference/deepwmh/pipeline/DCNN_multistage.py:529-533, 608-611)
// =================================================================================================
// seg = argmax_k logits (== argmax of the softmax; first index on ties, as torch.argmax); per foreground class c:
// counts[c-1] += #(seg == c & y == c), counts[(K-1)+c-1] += #(seg == c & y != c), counts[2(K-1)+c-1] += #(seg != c & y == c).
// Integer sums: the atomics' order cannot change them.
template <int K>
__global__ __launch_bounds__(256) void online_eval_counts_kernel(const float* logits, const uint8_t* target,
                                                                 unsigned long long* counts, int N, size_t S) {
    __shared__ unsigned sh[4][3 * (K - 1)];
    const size_t M = (size_t)N * S;
    unsigned acc[3 * (K - 1)];
#pragma unroll
    for (int q = 0; q < 3 * (K - 1); ++q) acc[q] = 0u;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < M; idx += (size_t)gridDim.x * 256) {
        const size_t n = idx / S, v = idx - n * S;
        float best = logits[(n * K) * S + v];
        int seg = 0;
#pragma unroll
        for (int k = 1; k < K; ++k) {
            const float l = logits[(n * K + k) * S + v];
            if (l > best) { best = l; seg = k; }
        }
        const int y = target[idx];
#pragma unroll
        for (int c = 1; c < K; ++c) {
            acc[c - 1] += (seg == c && y == c);
            acc[(K - 1) + c - 1] += (seg == c && y != c);
            acc[2 * (K - 1) + c - 1] += (seg != c && y == c);
        }
    }
#pragma unroll
    for (int q = 0; q < 3 * (K - 1); ++q) {
        unsigned t = acc[q];
        for (int m = 1; m < 64; m <<= 1) t += __shfl_xor(t, m);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][q] = t;
    }
    __syncthreads();
    if (threadIdx.x < 3 * (K - 1)) {
        const unsigned long long t = (unsigned long long)sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
        if (t) atomicAdd(counts + threadIdx.x, t);
    }
}
extern "C" int wmh_online_eval_counts(const float* logits, const uint8_t* target, int N, int K, size_t S,
                                      unsigned long long* counts, void* stream) {
    if (!logits || !target || !counts || N <= 0 || S == 0) return wmh_fail(WMH_EINVAL, "online_eval_counts: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const size_t M = (size_t)N * S;
    unsigned nblk = cdivu(M, 256 * 8); if (nblk > 2048) nblk = 2048; if (nblk < 1) nblk = 1;
    switch (K) {
        case 2: hipLaunchKernelGGL((online_eval_counts_kernel<2>), dim3(nblk), dim3(256), 0, st, logits, target, counts, N, S); break;
        case 3: hipLaunchKernelGGL((online_eval_counts_kernel<3>), dim3(nblk), dim3(256), 0, st, logits, target, counts, N, S); break;
        case 4: hipLaunchKernelGGL((online_eval_counts_kernel<4>), dim3(nblk), dim3(256), 0, st, logits, target, counts, N, S); break;
        default: return wmh_fail(WMH_EINVAL, "online_eval_counts: K must be 2, 3 or 4");
    }
    return wmh_check_launch("online_eval_counts_kernel");
}

// =================================================================================================
// InstanceNorm + LeakyReLU backward (with the segmentation head folded in)
// =================================================================================================
// Upstream gradient of the activated tensor a = lrelu(z), z = y*scale+shift:
//   dA = g1 + g2 + sum_k wseg[k][c] * dl[k]      (any of the three sources may be absent)
// reduce: per (n,c)  sums[0] = sum dz, sums[1] = sum dz*xhat, sums[2+k] = sum a*dl[k]   (dz = dA * lrelu'(z))
// apply : dy = gamma*rstd * (dz - sums[0]/S - xhat * sums[1]/S)
struct BwdSrc {
    const float* g1; const float* g2; const float* dl; const float* wseg; int K;
};
__device__ __forceinline__ float upstream(const BwdSrc& s, size_t nc_off, size_t n, int c, int C, size_t S, size_t v) {
    float d = 0.f;
    if (s.g1) d += s.g1[nc_off + v];
    if (s.g2) d += s.g2[nc_off + v];
    if (s.dl) for (int k = 0; k < s.K; ++k) d = fmaf(s.wseg[k * C + c], s.dl[(n * s.K + k) * S + v], d);
    return d;
}
// grid (SPLIT, N*C): block (sp, nc) reduces the sp-th slice of the spatial range; partial sums per block.
// V4 (S % 4 == 0, checked by the launcher): a thread takes four consecutive voxels per step with 16-byte loads -- the dword form kept too
// few bytes in flight per thread and left these two passes at 60-70 % of what the HBM delivers (4.0 ms of a 31 ms training iteration).
template <bool V4>
__global__ __launch_bounds__(256) void in_lrelu_bwd_reduce_kernel(const float* y, const float* ss, const float* mr,
                                                                  BwdSrc src, double* psums, int C, size_t S, float slope,
                                                                  int nq) {
    __shared__ double sh[4];
    const size_t nc = blockIdx.y, n = nc / C;
    const int c = (int)(nc % C);
    size_t chunk = (S + gridDim.x - 1) / gridDim.x;
    if (V4) chunk = (chunk + 3) & ~(size_t)3;
    const size_t v0 = (size_t)blockIdx.x * chunk < S ? (size_t)blockIdx.x * chunk : S, v1 = v0 + chunk < S ? v0 + chunk : S;
    const float sc = ss[2 * nc], shf = ss[2 * nc + 1], mean = mr[2 * nc], rstd = mr[2 * nc + 1];
    const float* yp = y + nc * S;
    double a0 = 0.0, a1 = 0.0, ak[4] = {0.0, 0.0, 0.0, 0.0};
    if constexpr (V4) {
        float wk[4] = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < src.K && src.dl; ++k) wk[k] = src.wseg[k * C + c];
        for (size_t v = v0 + 4 * (size_t)threadIdx.x; v < v1; v += 1024) {
            const float4 y4 = *reinterpret_cast<const float4*>(yp + v);
            float4 d4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (src.g1) { const float4 t = *reinterpret_cast<const float4*>(src.g1 + nc * S + v); d4.x += t.x; d4.y += t.y; d4.z += t.z; d4.w += t.w; }
            if (src.g2) { const float4 t = *reinterpret_cast<const float4*>(src.g2 + nc * S + v); d4.x += t.x; d4.y += t.y; d4.z += t.z; d4.w += t.w; }
            float4 l4[4];
            if (src.dl)
                for (int k = 0; k < src.K; ++k) {
                    l4[k] = *reinterpret_cast<const float4*>(src.dl + (n * src.K + k) * S + v);
                    d4.x = fmaf(wk[k], l4[k].x, d4.x); d4.y = fmaf(wk[k], l4[k].y, d4.y); d4.z = fmaf(wk[k], l4[k].z, d4.z); d4.w = fmaf(wk[k], l4[k].w, d4.w);
                }
            const float yv[4] = {y4.x, y4.y, y4.z, y4.w}, dA[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float z = fmaf(yv[e], sc, shf);
                const float dz = z > 0.f ? dA[e] : dA[e] * slope;
                const float xh = (yv[e] - mean) * rstd;
                a0 += dz; a1 += (double)dz * xh;
                if (src.dl) {
                    const float act = z > 0.f ? z : z * slope;
                    for (int k = 0; k < src.K; ++k) ak[k] += (double)act * (e == 0 ? l4[k].x : e == 1 ? l4[k].y : e == 2 ? l4[k].z : l4[k].w);
                }
            }
        }
    } else {
    for (size_t v = v0 + threadIdx.x; v < v1; v += 256) {
        const float yv = yp[v];
        const float z = fmaf(yv, sc, shf);
        const float dA = upstream(src, nc * S, n, c, C, S, v);
        const float dz = z > 0.f ? dA : dA * slope;
        const float xh = (yv - mean) * rstd;
        a0 += dz; a1 += (double)dz * xh;
        if (src.dl) {
            const float act = z > 0.f ? z : z * slope;
            for (int k = 0; k < src.K; ++k) ak[k] += (double)act * src.dl[(n * src.K + k) * S + v];
        }
    }
    }
    double* out = psums + (nc * gridDim.x + blockIdx.x) * nq;
    double t = block_sum_256(a0, sh); if (threadIdx.x == 0) out[0] = t;
    t = block_sum_256(a1, sh);        if (threadIdx.x == 0) out[1] = t;
    for (int k = 0; k < src.K && src.dl; ++k) { t = block_sum_256(ak[k], sh); if (threadIdx.x == 0) out[2 + k] = t; }
}
// dy = gamma*rstd*(dz - mean(dz) - xhat*mean(dz*xhat)); each block also leaves sum(dy) of its slice (conv bias gradient)
template <bool V4>
__global__ __launch_bounds__(256) void in_lrelu_bwd_apply_kernel(const float* y, const float* ss, const float* mr,
                                                                 const float* gamma, BwdSrc src, const double* psums, int split,
                                                                 double* sums, float* dy, double* dysum, int C, size_t S, float slope,
                                                                 int nq) {
    __shared__ double sh[4];
    const size_t nc = blockIdx.y, n = nc / C;
    const int c = (int)(nc % C);
    const float sc = ss[2 * nc], shf = ss[2 * nc + 1], mean = mr[2 * nc], rstd = mr[2 * nc + 1];
    // the row's sums from the reduce pass's slices (split <= 256): every block of the row forms the same fixed-order tree; block 0
    // also forms the head's sums and leaves all of them for in_param_grads_kernel (this was a launch of its own: 22 per step)
    double tot0 = 0.0, tot1 = 0.0;
    {
        const int nused = 2 + (src.dl ? src.K : 0);
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            if (q >= nused || (q >= 2 && blockIdx.x != 0)) break;     // uniform
            const double part = (int)threadIdx.x < split ? psums[(nc * split + threadIdx.x) * nq + q] : 0.0;
            const double t = block_sum_256(part, sh);
            if (q == 0) tot0 = t;
            if (q == 1) tot1 = t;
            if (blockIdx.x == 0 && threadIdx.x == 0) sums[nc * nq + q] = t;
        }
    }
    const float m1 = (float)(tot0 / (double)S), m2 = (float)(tot1 / (double)S);
    const float gr = (gamma ? gamma[c] : 1.f) * rstd;
    const float* yp = y + nc * S;
    float* op = dy + nc * S;
    double acc = 0.0;
    if constexpr (V4) {
        float wk[4] = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < src.K && src.dl; ++k) wk[k] = src.wseg[k * C + c];
        for (size_t v = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; v < S; v += (size_t)gridDim.x * 1024) {
            const float4 y4 = *reinterpret_cast<const float4*>(yp + v);
            float4 d4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (src.g1) { const float4 t = *reinterpret_cast<const float4*>(src.g1 + nc * S + v); d4.x += t.x; d4.y += t.y; d4.z += t.z; d4.w += t.w; }
            if (src.g2) { const float4 t = *reinterpret_cast<const float4*>(src.g2 + nc * S + v); d4.x += t.x; d4.y += t.y; d4.z += t.z; d4.w += t.w; }
            if (src.dl)
                for (int k = 0; k < src.K; ++k) {
                    const float4 l = *reinterpret_cast<const float4*>(src.dl + (n * src.K + k) * S + v);
                    d4.x = fmaf(wk[k], l.x, d4.x); d4.y = fmaf(wk[k], l.y, d4.y); d4.z = fmaf(wk[k], l.z, d4.z); d4.w = fmaf(wk[k], l.w, d4.w);
                }
            const float yv[4] = {y4.x, y4.y, y4.z, y4.w}, dA[4] = {d4.x, d4.y, d4.z, d4.w};
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float z = fmaf(yv[e], sc, shf);
                const float dz = z > 0.f ? dA[e] : dA[e] * slope;
                const float xh = (yv[e] - mean) * rstd;
                o[e] = gr * (dz - m1 - xh * m2);
                acc += o[e];
            }
            *reinterpret_cast<float4*>(op + v) = make_float4(o[0], o[1], o[2], o[3]);
        }
    } else {
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < S; v += (size_t)gridDim.x * 256) {
        const float yv = yp[v];
        const float z = fmaf(yv, sc, shf);
        const float dA = upstream(src, nc * S, n, c, C, S, v);
        const float dz = z > 0.f ? dA : dA * slope;
        const float xh = (yv - mean) * rstd;
        const float o = gr * (dz - m1 - xh * m2);
        op[v] = o;
        acc += o;
    }
    }
    const double t = block_sum_256(acc, sh);
    if (threadIdx.x == 0) dysum[nc * gridDim.x + blockIdx.x] = t;
}
// ---- round 6: the two 16-byte passes without a branch between their loads --------------------------------------------------------------
// As compiled, the V4 kernels above test `src.g1`, `src.g2`, `src.dl` inside the loop; every test ends a basic block and the counted waits
// become `s_waitcnt vmcnt(0)` behind each load: one or two 16-byte loads in flight per lane (4.3 TB/s on the reduce pass; a read-only stream
// with eight in flight reads at 6.2-7 TB/s on this part: tools/micro/read_rate.hip).  Here the sources are a template parameter (SRC: bit 0 g1,
// bit 1 g2, bit 2 the head's dlogits with K classes), a step's loads are issued together and U steps are in flight per lane.  The sums are
// formed in the same order as above (a thread's steps in sequence): same bits.
template <int SRC, int K>
struct BwdStep {
    float4 y, g1, g2, l[K > 0 ? K : 1];
};
template <int SRC, int K>
__device__ __forceinline__ void bwd_step_load(BwdStep<SRC, K>& s, const float* __restrict__ yp, const float* __restrict__ g1p, const float* __restrict__ g2p,
                                              const float* __restrict__ dlp, size_t S, size_t v) {
    s.y = *reinterpret_cast<const float4*>(yp + v);
    if constexpr ((SRC & 1) != 0) s.g1 = *reinterpret_cast<const float4*>(g1p + v);
    if constexpr ((SRC & 2) != 0) s.g2 = *reinterpret_cast<const float4*>(g2p + v);
    if constexpr ((SRC & 4) != 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) s.l[k] = *reinterpret_cast<const float4*>(dlp + (size_t)k * S + v);
    }
}
template <int SRC, int K>
__device__ __forceinline__ void bwd_step_upstream(const BwdStep<SRC, K>& s, const float (&wk)[4], float (&dA)[4]) {
    float4 d4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr ((SRC & 1) != 0) { d4.x += s.g1.x; d4.y += s.g1.y; d4.z += s.g1.z; d4.w += s.g1.w; }
    if constexpr ((SRC & 2) != 0) { d4.x += s.g2.x; d4.y += s.g2.y; d4.z += s.g2.z; d4.w += s.g2.w; }
    if constexpr ((SRC & 4) != 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            d4.x = fmaf(wk[k], s.l[k].x, d4.x); d4.y = fmaf(wk[k], s.l[k].y, d4.y); d4.z = fmaf(wk[k], s.l[k].z, d4.z); d4.w = fmaf(wk[k], s.l[k].w, d4.w);
        }
    }
    dA[0] = d4.x; dA[1] = d4.y; dA[2] = d4.z; dA[3] = d4.w;
}
template <int SRC, int K>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void in_lrelu_bwd_reduce4_kernel(const float* __restrict__ y, const float* __restrict__ ss, const float* __restrict__ mr,
                                                                   BwdSrc src, double* __restrict__ psums, int C, size_t S, float slope, int nq) {
    __shared__ double sh[4];
    // steps in flight per lane (with the head's two logit planes a step is four loads: at U = 4 the register allocator, holding on to
    // eight waves per SIMD, serialised the loads again -- 1.9 ms for the 128^3 layer where the old form took 1.05)
    constexpr int U = (SRC & 4) ? 2 : 4;
    const size_t nc = blockIdx.y, n = nc / C;
    const int c = (int)(nc % C);
    size_t chunk = (S + gridDim.x - 1) / gridDim.x;
    chunk = (chunk + 3) & ~(size_t)3;
    const size_t v0 = (size_t)blockIdx.x * chunk < S ? (size_t)blockIdx.x * chunk : S, v1 = v0 + chunk < S ? v0 + chunk : S;
    const float sc = ss[2 * nc], shf = ss[2 * nc + 1], mean = mr[2 * nc], rstd = mr[2 * nc + 1];
    const float* yp = y + nc * S;
    const float* g1p = (SRC & 1) ? src.g1 + nc * S : nullptr;
    const float* g2p = (SRC & 2) ? src.g2 + nc * S : nullptr;
    const float* dlp = (SRC & 4) ? src.dl + n * K * S : nullptr;
    float wk[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr ((SRC & 4) != 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) wk[k] = src.wseg[k * C + c];
    }
    double a0 = 0.0, a1 = 0.0, ak[4] = {0.0, 0.0, 0.0, 0.0};
    auto use = [&](const BwdStep<SRC, K>& s) __attribute__((always_inline)) {
        float dA[4];
        bwd_step_upstream<SRC, K>(s, wk, dA);
        const float yv[4] = {s.y.x, s.y.y, s.y.z, s.y.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float z = fmaf(yv[e], sc, shf);
            const float dz = z > 0.f ? dA[e] : dA[e] * slope;
            const float xh = (yv[e] - mean) * rstd;
            a0 += dz; a1 += (double)dz * xh;
            if constexpr ((SRC & 4) != 0) {
                const float act = z > 0.f ? z : z * slope;
#pragma unroll
                for (int k = 0; k < K; ++k) ak[k] += (double)act * (e == 0 ? s.l[k].x : e == 1 ? s.l[k].y : e == 2 ? s.l[k].z : s.l[k].w);
            }
        }
    };
    size_t v = v0 + 4 * (size_t)threadIdx.x;
    for (; v + (U - 1) * 1024 < v1; v += U * 1024) {
        BwdStep<SRC, K> st[U];
#pragma unroll
        for (int u = 0; u < U; ++u) bwd_step_load<SRC, K>(st[u], yp, g1p, g2p, dlp, S, v + (size_t)u * 1024);
#pragma unroll
        for (int u = 0; u < U; ++u) use(st[u]);
    }
    for (; v < v1; v += 1024) {
        BwdStep<SRC, K> st;
        bwd_step_load<SRC, K>(st, yp, g1p, g2p, dlp, S, v);
        use(st);
    }
    double* out = psums + (nc * gridDim.x + blockIdx.x) * nq;
    double t = block_sum_256(a0, sh); if (threadIdx.x == 0) out[0] = t;
    t = block_sum_256(a1, sh);        if (threadIdx.x == 0) out[1] = t;
    if constexpr ((SRC & 4) != 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) { t = block_sum_256(ak[k], sh); if (threadIdx.x == 0) out[2 + k] = t; }
    }
}
template <int SRC, int K>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void in_lrelu_bwd_apply4_kernel(const float* __restrict__ y, const float* __restrict__ ss, const float* __restrict__ mr,
                                                                  const float* __restrict__ gamma, BwdSrc src, const double* __restrict__ psums, int split,
                                                                  double* __restrict__ sums, float* __restrict__ dy, double* __restrict__ dysum, int C, size_t S,
                                                                  float slope, int nq) {
    __shared__ double sh[4];
    constexpr int U = 2;
    const size_t nc = blockIdx.y, n = nc / C;
    const int c = (int)(nc % C);
    const float sc = ss[2 * nc], shf = ss[2 * nc + 1], mean = mr[2 * nc], rstd = mr[2 * nc + 1];
    double tot0 = 0.0, tot1 = 0.0;
    {
        constexpr int nused = 2 + ((SRC & 4) ? K : 0);
#pragma unroll
        for (int q = 0; q < nused; ++q) {
            if (q >= 2 && blockIdx.x != 0) break;                 // uniform
            const double part = (int)threadIdx.x < split ? psums[(nc * split + threadIdx.x) * nq + q] : 0.0;
            const double t = block_sum_256(part, sh);
            if (q == 0) tot0 = t;
            if (q == 1) tot1 = t;
            if (blockIdx.x == 0 && threadIdx.x == 0) sums[nc * nq + q] = t;
        }
    }
    const float m1 = (float)(tot0 / (double)S), m2 = (float)(tot1 / (double)S);
    const float gr = (gamma ? gamma[c] : 1.f) * rstd;
    const float* yp = y + nc * S;
    const float* g1p = (SRC & 1) ? src.g1 + nc * S : nullptr;
    const float* g2p = (SRC & 2) ? src.g2 + nc * S : nullptr;
    const float* dlp = (SRC & 4) ? src.dl + n * K * S : nullptr;
    float* op = dy + nc * S;
    float wk[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr ((SRC & 4) != 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) wk[k] = src.wseg[k * C + c];
    }
    double acc = 0.0;
    auto use = [&](const BwdStep<SRC, K>& s, size_t v) __attribute__((always_inline)) {
        float dA[4], o[4];
        bwd_step_upstream<SRC, K>(s, wk, dA);
        const float yv[4] = {s.y.x, s.y.y, s.y.z, s.y.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float z = fmaf(yv[e], sc, shf);
            const float dz = z > 0.f ? dA[e] : dA[e] * slope;
            const float xh = (yv[e] - mean) * rstd;
            o[e] = gr * (dz - m1 - xh * m2);
            acc += o[e];
        }
        *reinterpret_cast<float4*>(op + v) = make_float4(o[0], o[1], o[2], o[3]);
    };
    const size_t stride = (size_t)gridDim.x * 1024;
    size_t v = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    for (; v + (U - 1) * stride < S; v += U * stride) {
        BwdStep<SRC, K> st[U];
#pragma unroll
        for (int u = 0; u < U; ++u) bwd_step_load<SRC, K>(st[u], yp, g1p, g2p, dlp, S, v + (size_t)u * stride);
#pragma unroll
        for (int u = 0; u < U; ++u) use(st[u], v + (size_t)u * stride);
    }
    for (; v < S; v += stride) {
        BwdStep<SRC, K> st;
        bwd_step_load<SRC, K>(st, yp, g1p, g2p, dlp, S, v);
        use(st, v);
    }
    const double t = block_sum_256(acc, sh);
    if (threadIdx.x == 0) dysum[nc * gridDim.x + blockIdx.x] = t;
}

// dgamma[c] = sum_n sums[n,c,1], dbeta[c] = sum_n sums[n,c,0], dwseg[k][c] = sum_n sums[n,c,2+k], dbias[c] = sum dysum
__global__ void in_param_grads_kernel(const double* sums, const double* dysum, int gx, int N, int C, int nq, int K,
                                      float* dgamma, float* dbeta, float* dwseg, float* dbias) {
    // one wave per channel: the N * gx slice sums of dy as a fixed-order tree (lane l takes entries l, l + 64, ...), the N row sums
    // serially on lane 0
    const int c = blockIdx.x;
    double db = 0.0;
    for (int idx = threadIdx.x; idx < N * gx; idx += 64) db += dysum[((size_t)(idx / gx) * C + c) * gx + idx % gx];
    for (int m = 1; m < 64; m <<= 1) db += shfl_xor_d(db, m);
    if (threadIdx.x != 0) return;
    double g = 0.0, b = 0.0, w[4] = {0.0, 0.0, 0.0, 0.0};
    for (int n = 0; n < N; ++n) {
        const double* s = sums + ((size_t)n * C + c) * nq;
        b += s[0]; g += s[1];
        for (int k = 0; k < K; ++k) w[k] += s[2 + k];
    }
    if (dgamma) dgamma[c] = (float)g;
    if (dbeta) dbeta[c] = (float)b;
    if (dbias) dbias[c] = (float)db;
    if (dwseg) for (int k = 0; k < K; ++k) dwseg[k * C + c] = (float)w[k];
}

static void in_bwd_geometry(int N, int C, size_t S, unsigned& split, unsigned& gx) {
    const size_t NC = (size_t)N * C;
    split = (unsigned)((2048 + NC - 1) / NC);                 // ~2048 reducing blocks in total
    const unsigned maxs = cdivu(S, 2048); if (split > maxs) split = maxs; if (split < 1) split = 1; if (split > 256) split = 256;
    gx = cdivu(S, 256 * 4); if (gx > 512) gx = 512; if (gx < 1) gx = 1;
    const unsigned cap = (unsigned)((8192 + NC - 1) / NC); if (gx > cap) gx = cap < 1 ? 1 : cap;
}
extern "C" size_t wmh_in_lrelu_bwd_scratch_doubles(int N, int C, size_t S) {
    unsigned split, gx; in_bwd_geometry(N, C, S, split, gx);
    return (size_t)N * C * 6 * (1 + split) + (size_t)N * C * gx;
}
extern "C" int wmh_in_lrelu_bwd(const float* y, const float* ss, const float* mean_rstd, const float* gamma,
                                const float* g1, const float* g2, const float* dlogits, const float* wseg, int K,
                                float* dy, float* dgamma, float* dbeta, float* dwseg, float* dbias, double* scratch, int N,
                                int C, size_t S, float slope, void* stream) {
    if (!y || !ss || !mean_rstd || !dy || !scratch || N < 1 || C < 1 || S < 1 || K > 4 || (!g1 && !g2 && !dlogits))
        return wmh_fail(WMH_EINVAL, "in_lrelu_bwd: bad args");
    if (dlogits && (!wseg || K < 1)) return wmh_fail(WMH_EINVAL, "in_lrelu_bwd: head gradient without head weights");
    hipStream_t st = (hipStream_t)stream;
    BwdSrc src{g1, g2, dlogits, wseg, dlogits ? K : 0};
    const int nq = 6, NC = N * C;
    unsigned split, gx; in_bwd_geometry(N, C, S, split, gx);
    double* sums = scratch;
    double* psums = scratch + (size_t)NC * nq;
    double* dysum = psums + (size_t)NC * nq * split;
    // 16-byte form: four voxels per thread and step (every tensor is a whole number of 16-byte pieces per channel)
    const bool v4 = S % 4 == 0 && ((uintptr_t)y | (uintptr_t)dy | (uintptr_t)g1 | (uintptr_t)g2 | (uintptr_t)dlogits) % 16 == 0;
    // the branch-free 16-byte kernels (round 6): every combination of sources without the head, and with the head for K = 2
    const int srcmask = (g1 ? 1 : 0) | (g2 ? 2 : 0) | (dlogits ? 4 : 0);
    static const int old_form = wmh_diag_env("WMH_IN_BWD_OLD", 0);
    if (v4 && !old_form && (!dlogits || K == 2)) {
#define IN_BWD4(M, KK)                                                                                                                          \
        case M:                                                                                                                                 \
            hipLaunchKernelGGL((in_lrelu_bwd_reduce4_kernel<M, KK>), dim3(split, (unsigned)NC), dim3(256), 0, st, y, ss, mean_rstd, src, psums, C, S, slope, nq);   \
            hipLaunchKernelGGL((in_lrelu_bwd_apply4_kernel<M, KK>), dim3(gx, (unsigned)NC), dim3(256), 0, st, y, ss, mean_rstd, gamma, src, psums, (int)split, sums, dy, \
                               dysum, C, S, slope, nq);                                                                                         \
            break
        switch (srcmask) {
            IN_BWD4(1, 0); IN_BWD4(2, 0); IN_BWD4(3, 0); IN_BWD4(4, 2); IN_BWD4(5, 2); IN_BWD4(6, 2); IN_BWD4(7, 2);
            default: return wmh_fail(WMH_EINVAL, "in_lrelu_bwd: no gradient source");
        }
#undef IN_BWD4
        hipLaunchKernelGGL(in_param_grads_kernel, dim3(C), dim3(64), 0, st, sums, dysum, (int)gx, N, C, nq, src.K, dgamma,
                           dbeta, dlogits ? dwseg : nullptr, dbias);
        return wmh_check_launch("in_lrelu_bwd kernels");
    }
    if (v4) hipLaunchKernelGGL(in_lrelu_bwd_reduce_kernel<true>, dim3(split, (unsigned)NC), dim3(256), 0, st, y, ss, mean_rstd, src, psums, C, S, slope, nq);
    else hipLaunchKernelGGL(in_lrelu_bwd_reduce_kernel<false>, dim3(split, (unsigned)NC), dim3(256), 0, st, y, ss, mean_rstd, src, psums, C,
                       S, slope, nq);
    if (v4) hipLaunchKernelGGL(in_lrelu_bwd_apply_kernel<true>, dim3(gx, (unsigned)NC), dim3(256), 0, st, y, ss, mean_rstd, gamma, src, psums, (int)split, sums, dy, dysum, C, S, slope, nq);
    else hipLaunchKernelGGL(in_lrelu_bwd_apply_kernel<false>, dim3(gx, (unsigned)NC), dim3(256), 0, st, y, ss, mean_rstd, gamma, src, psums, (int)split,
                       sums, dy, dysum, C, S, slope, nq);
    hipLaunchKernelGGL(in_param_grads_kernel, dim3(C), dim3(64), 0, st, sums, dysum, (int)gx, N, C, nq, src.K, dgamma,
                       dbeta, dlogits ? dwseg : nullptr, dbias);
    return wmh_check_launch("in_lrelu_bwd kernels");
}

// per-channel sum over (n, spatial): the conv bias gradient  db[c] = sum_{n,v} dy[n,c,v]
__global__ __launch_bounds__(256) void channel_sum_kernel(const float* x, float* out, int N, int C, size_t S) {
    __shared__ double sh[4];
    const int c = blockIdx.x;
    double a = 0.0;
    for (int n = 0; n < N; ++n) {
        const float* p = x + ((size_t)n * C + c) * S;
        for (size_t v = threadIdx.x; v < S; v += 256) a += p[v];
    }
    const double t = block_sum_256(a, sh);
    if (threadIdx.x == 0) out[c] = (float)t;
}
extern "C" int wmh_channel_sum(const float* x, float* out, int N, int C, size_t S, void* stream) {
    if (!x || !out || N < 1 || C < 1 || S < 1) return wmh_fail(WMH_EINVAL, "channel_sum: bad args");
    hipLaunchKernelGGL(channel_sum_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, x, out, N, C, S);
    return wmh_check_launch("channel_sum_kernel");
}

// =================================================================================================
// zero-insert upsampling: up[n,c,2z,2y,2x] = dy[n,c,z,y,x], 0 elsewhere; up has the stride-2 conv's INPUT size
// (data gradient of a stride-2 3x3x3 conv = stride-1 conv of `up` with the flipped, transposed kernel)
// =================================================================================================
__global__ __launch_bounds__(256) void upsample_zero2_kernel(const float* dy, float* up, int Do, int Ho, int Wo, int D,
                                                             int H, int W) {
    const size_t nc = blockIdx.y, S = (size_t)D * H * W;
    const float* ip = dy + nc * Do * Ho * Wo;
    float* op = up + nc * S;
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < S; v += (size_t)gridDim.x * 256) {
        const int x = (int)(v % W), yy = (int)((v / W) % H), z = (int)(v / ((size_t)W * H));
        const bool on = !((x | yy | z) & 1);
        op[v] = on ? ip[((size_t)(z >> 1) * Ho + (yy >> 1)) * Wo + (x >> 1)] : 0.f;
    }
}
// the same for a stride (1,2,2) convolution: zeros inserted in the plane only (Do == D)
__global__ __launch_bounds__(256) void upsample_zero_plane_kernel(const float* dy, float* up, int Ho, int Wo, int D, int H, int W) {
    const size_t nc = blockIdx.y, S = (size_t)D * H * W;
    const float* ip = dy + nc * D * Ho * Wo;
    float* op = up + nc * S;
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < S; v += (size_t)gridDim.x * 256) {
        const int x = (int)(v % W), yy = (int)((v / W) % H), z = (int)(v / ((size_t)W * H));
        const bool on = !((x | yy) & 1);
        op[v] = on ? ip[((size_t)z * Ho + (yy >> 1)) * Wo + (x >> 1)] : 0.f;
    }
}
extern "C" int wmh_upsample_zero_plane(const float* dy, float* up, int N, int C, int Ho, int Wo, int D, int H, int W, void* stream) {
    if (!dy || !up || (H - 1) / 2 + 1 != Ho || (W - 1) / 2 + 1 != Wo || D < 1)
        return wmh_fail(WMH_EINVAL, "upsample_zero_plane: bad args");
    const size_t S = (size_t)D * H * W;
    unsigned gx = cdivu(S, 256 * 4); if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(upsample_zero_plane_kernel, dim3(gx, (unsigned)((size_t)N * C)), dim3(256), 0, (hipStream_t)stream, dy, up,
                       Ho, Wo, D, H, W);
    return wmh_check_launch("upsample_zero_plane_kernel");
}
extern "C" int wmh_upsample_zero2(const float* dy, float* up, int N, int C, int Do, int Ho, int Wo, int D, int H, int W,
                                  void* stream) {
    if (!dy || !up || (D - 1) / 2 + 1 != Do || (H - 1) / 2 + 1 != Ho || (W - 1) / 2 + 1 != Wo)
        return wmh_fail(WMH_EINVAL, "upsample_zero2: bad args");
    const size_t S = (size_t)D * H * W;
    unsigned gx = cdivu(S, 256 * 4); if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(upsample_zero2_kernel, dim3(gx, (unsigned)((size_t)N * C)), dim3(256), 0, (hipStream_t)stream, dy, up,
                       Do, Ho, Wo, D, H, W);
    return wmh_check_launch("upsample_zero2_kernel");
}

// =================================================================================================
// ConvTranspose3d(k2,s2) backward (vector ALU v1; the forward is on the matrix cores)
//   dX'[n,ci,d,h,w] = sum_{co,t} dY[n,co,2d+i,2h+j,2w+k] * W[ci,co,t]       (gradient of the ACTIVATED input x')
//   dW[ci,co,t]     = sum_{n,d,h,w} x'[n,ci,d,h,w] * dY[n,co,2d+i,2h+j,2w+k]
// =================================================================================================
template <int CIT>
__global__ __launch_bounds__(256) void convT_bwd_data_kernel(const float* dY, const float* w, float* dX, int Cin, int Cout,
                                                             int D, int H, int W) {
    const int n = blockIdx.z, ci0 = blockIdx.y * CIT;
    const size_t DHW = (size_t)D * H * W, v = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (v >= DHW) return;
    const int wq = (int)(v % W), hq = (int)((v / W) % H), dq = (int)(v / ((size_t)W * H));
    const int H2 = 2 * H, W2 = 2 * W;
    float acc[CIT];
#pragma unroll
    for (int c = 0; c < CIT; ++c) acc[c] = 0.f;
    for (int co = 0; co < Cout; ++co) {
        const float* yp = dY + ((size_t)n * Cout + co) * 8 * DHW + ((size_t)(2 * dq) * H2 + 2 * hq) * W2 + 2 * wq;
        float g[8];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const float2 t = *reinterpret_cast<const float2*>(yp + ((size_t)i * H2 + jj) * W2);
                g[i * 4 + jj * 2] = t.x; g[i * 4 + jj * 2 + 1] = t.y;
            }
#pragma unroll
        for (int c = 0; c < CIT; ++c) {
            if (ci0 + c < Cin) {
                const float* wp = w + ((size_t)(ci0 + c) * Cout + co) * 8;
#pragma unroll
                for (int t = 0; t < 8; ++t) acc[c] = fmaf(g[t], wp[t], acc[c]);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < CIT; ++c)
        if (ci0 + c < Cin) dX[((size_t)n * Cin + ci0 + c) * DHW + v] = acc[c];
}
// one block per (ci, co): reduces over n and space in a fixed order
__global__ __launch_bounds__(256) void convT_bwd_weight_kernel(const float* x, const float* ss, const float* dY, float* dW,
                                                               int N, int Cin, int Cout, int D, int H, int W,
                                                               float slope) {
    __shared__ double sh[4];
    const int co = blockIdx.x, ci = blockIdx.y;
    const size_t DHW = (size_t)D * H * W;
    const int H2 = 2 * H, W2 = 2 * W;
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int n = 0; n < N; ++n) {
        const float* xp = x + ((size_t)n * Cin + ci) * DHW;
        const float sc = ss ? ss[2 * ((size_t)n * Cin + ci)] : 1.f, shf = ss ? ss[2 * ((size_t)n * Cin + ci) + 1] : 0.f;
        const float* yb = dY + ((size_t)n * Cout + co) * 8 * DHW;
        for (size_t v = threadIdx.x; v < DHW; v += 256) {
            const int wq = (int)(v % W), hq = (int)((v / W) % H), dq = (int)(v / ((size_t)W * H));
            float xv = xp[v];
            if (ss) xv = wmh_lrelu(fmaf(xv, sc, shf), slope);
            const float* yp = yb + ((size_t)(2 * dq) * H2 + 2 * hq) * W2 + 2 * wq;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const float2 t = *reinterpret_cast<const float2*>(yp + ((size_t)i * H2 + jj) * W2);
                    acc[i * 4 + jj * 2] += (double)xv * t.x; acc[i * 4 + jj * 2 + 1] += (double)xv * t.y;
                }
        }
    }
    for (int t = 0; t < 8; ++t) {
        const double s = block_sum_256(acc[t], sh);
        if (threadIdx.x == 0) dW[((size_t)ci * Cout + co) * 8 + t] = (float)s;
    }
}
extern "C" int wmh_convtranspose3d_k2s2_bwd(const float* x, const float* ss, const float* w, const float* dY, float* dX,
                                            float* dW, int N, int Cin, int Cout, int D, int H, int W, float slope,
                                            void* stream) {
    if (!w || !dY || N < 1 || Cin < 1 || Cout < 1) return wmh_fail(WMH_EINVAL, "convtranspose3d_k2s2_bwd: bad args");
    hipStream_t st = (hipStream_t)stream;
    const size_t DHW = (size_t)D * H * W;
    if (dX) {
        constexpr int CIT = 8;
        hipLaunchKernelGGL((convT_bwd_data_kernel<CIT>), dim3(cdivu(DHW, 256), (Cin + CIT - 1) / CIT, N), dim3(256), 0, st, dY,
                           w, dX, Cin, Cout, D, H, W);
    }
    if (dW) {
        if (!x) return wmh_fail(WMH_EINVAL, "convtranspose3d_k2s2_bwd: dW needs x");
        hipLaunchKernelGGL(convT_bwd_weight_kernel, dim3(Cout, Cin), dim3(256), 0, st, x, ss, dY, dW, N, Cin, Cout, D, H, W, slope);
    }
    return wmh_check_launch("convT_bwd kernels");
}

// =================================================================================================
// gradient-norm clipping (torch.nn.utils.clip_grad_norm_) and SGD with Nesterov momentum + weight decay
// =================================================================================================
__global__ __launch_bounds__(256) void sqnorm_partials_kernel(const float* g, size_t n, double* part) {
    __shared__ double sh[4];
    double a = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a += (double)g[i] * g[i];
    const double t = block_sum_256(a, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}
// out[0] = total norm, out[1] = clip coefficient min(1, max_norm / (norm + 1e-6)).  A non-finite norm (an overflow somewhere in a
// loss-scaled fp16-pair backward pass) is reported as it is with coefficient 0: sgd_nesterov_kernel then skips the whole update,
// as torch.cuda.amp.GradScaler.step does in nnU-Net's own fp16 run_iteration [nnU-Net, not in tree].
__global__ void clip_coef_kernel(const double* part, int nblk, float max_norm, float* out) {
    double s = 0.0;                                        // one wave, fixed-order tree (as loss_finalize_kernel)
    for (int b = threadIdx.x; b < nblk; b += 64) s += part[b];
    for (int m = 1; m < 64; m <<= 1) s += shfl_xor_d(s, m);
    if (threadIdx.x == 0) {
        const double norm = sqrt(s);
        const double coef = max_norm / (norm + 1e-6);
        const bool finite = norm <= 1.7e308;               // false for inf and NaN
        out[0] = (float)norm;
        out[1] = finite ? (float)(coef < 1.0 ? coef : 1.0) : 0.f;
    }
}
extern "C" int wmh_grad_clip_coef(const float* grads, size_t n, float max_norm, float* out2, double* workspace, void* stream) {
    if (!grads || !out2 || !workspace || n < 1) return wmh_fail(WMH_EINVAL, "grad_clip_coef: bad args");
    unsigned nblk = cdivu(n, 256 * 16); if (nblk > 1024) nblk = 1024;
    hipLaunchKernelGGL(sqnorm_partials_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, grads, n, workspace);
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, workspace, (int)nblk, max_norm, out2);
    return wmh_check_launch("grad_clip kernels");
}
// torch.optim.SGD(nesterov=True): g = clip*grad + wd*p ; buf = first ? g : mom*buf + g ; p -= lr * (g + mom*buf)
__global__ __launch_bounds__(256) void sgd_nesterov_kernel(float* p, const float* g, float* m, size_t n, float lr, float mom,
                                                           float wd, const float* clip, int first) {
    const float cs = clip ? clip[1] : 1.f;
    // non-finite gradient norm: parameters and momentum stay as they are (one inf / NaN gradient would poison both for good)
    if (clip && !(fabsf(clip[0]) <= 3.4028234e38f)) return;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float pv = p[i];
        float gv = g[i] * cs;
        gv = fmaf(wd, pv, gv);
        const float b = first ? gv : fmaf(mom, m[i], gv);
        m[i] = b;
        p[i] = pv - lr * fmaf(mom, b, gv);
    }
}
extern "C" int wmh_sgd_nesterov_step(float* params, const float* grads, float* momentum_buf, size_t n, float lr,
                                     float momentum, float weight_decay, const float* clip, int first_step, void* stream) {
    if (!params || !grads || !momentum_buf || n < 1) return wmh_fail(WMH_EINVAL, "sgd_nesterov_step: bad args");
    unsigned nblk = cdivu(n, 256 * 4); if (nblk > 4096) nblk = 4096;
    hipLaunchKernelGGL(sgd_nesterov_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, params, grads, momentum_buf, n, lr,
                       momentum, weight_decay, clip, first_step);
    return wmh_check_launch("sgd_nesterov_kernel");
}
