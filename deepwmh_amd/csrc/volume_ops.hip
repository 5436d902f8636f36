// volume_ops.hip -- whole-volume integer/byte passes either side of the network, all HBM-bound:
//   * 6-connected component labelling (label equivalence with an atomic-min union-find; roots are the smallest
//     voxel index of a component, so the labelling does not depend on the order the atomics land in),
//   * remove_sparks: drop components smaller than min_volume   (reference deepwmh/analysis/image_ops.py:325-345),
//   * fill_holes: scipy.ndimage.binary_fill_holes = background components that do not touch the border
//     [nnU-Net cropping.create_nonzero_mask, not in tree],
//   * non-zero mask, bounding box, masked z-score + crop [nnU-Net GenericPreprocessor, not in tree], un-crop.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "wmh_common.h"

static inline unsigned vgrid(size_t n, int per = 256, unsigned cap = 65535u * 16u) {
    size_t b = (n + per - 1) / per; if (b < 1) b = 1; if (b > cap) b = cap; return (unsigned)b;
}

__device__ __forceinline__ int cc_load(const int* L, int a) {
    return __hip_atomic_load(L + a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // coherent at L2, like the atomics
}
__device__ __forceinline__ int cc_find(const int* L, int a) {
    int p;
    while ((p = cc_load(L, a)) != a) a = p;
    return a;
}
__device__ __forceinline__ void cc_union(int* L, int a, int b) {
    for (;;) {
        a = cc_find(L, a); b = cc_find(L, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }          // hang the larger root under the smaller
        const int old = atomicMin(L + a, b);
        if (old == a) return;                                   // a was still a root: linked
        a = old;                                                // somebody re-linked a meanwhile: retry from there
    }
}

// fg(v) = (mask[v] != 0) ^ invert.  Labels start as the first voxel of the x-run a voxel sits in (found with one ballot per
// wave: a wave covers 64 consecutive voxels), so the union-find only ever sees one node per run segment.
__global__ __launch_bounds__(256) void cc_init_kernel(const uint8_t* mask, int* L, unsigned n, int W, int invert) {
    const unsigned lane = threadIdx.x & 63;
    const unsigned nround = (n + 255u) & ~255u;                 // whole waves iterate together (ballots need all lanes)
    for (unsigned v = blockIdx.x * 256u + threadIdx.x; v < nround; v += gridDim.x * 256u) {
        const bool fg = v < n && ((((mask[v] != 0) ? 1 : 0) ^ invert) != 0);
        const unsigned x = v % (unsigned)W;
        const unsigned long long fgm = __ballot(fg);
        const bool prev_fg = lane > 0 && ((fgm >> (lane - 1)) & 1ull);
        const unsigned long long starts = __ballot(fg && (!prev_fg || x == 0));
        if (v < n) {
            int lab = -1;
            if (fg) {
                const unsigned long long below = starts & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
                const unsigned first = 63u - (unsigned)__builtin_clzll(below);     // this lane's run start (always exists)
                lab = (int)(v - (lane - first));
            }
            L[v] = lab;
        }
    }
}
// One union per place where two runs START touching: a voxel links to its y/z neighbour only if the pair to its left is
// not linked already (both left voxels foreground and in the same rows), and to its x neighbour only across a wave boundary.
__global__ __launch_bounds__(256) void cc_merge_kernel(int* L, int D, int H, int W) {
    const unsigned n = (unsigned)D * H * W, HW = (unsigned)H * W;
    for (unsigned v = blockIdx.x * 256u + threadIdx.x; v < n; v += gridDim.x * 256u) {
        if (L[v] < 0) continue;                                 // sign never changes after init
        const unsigned x = v % (unsigned)W, y = (v / (unsigned)W) % (unsigned)H, z = v / HW;
        const bool left = x > 0 && L[v - 1] >= 0;
        if (left && (v & 63u) == 0) cc_union(L, (int)v, (int)v - 1);
        if (y > 0 && L[v - W] >= 0 && !(left && L[v - W - 1] >= 0)) cc_union(L, (int)v, (int)(v - W));
        if (z > 0 && L[v - HW] >= 0 && !(left && L[v - HW - 1] >= 0)) cc_union(L, (int)v, (int)(v - HW));
    }
}
__global__ __launch_bounds__(256) void cc_compress_kernel(int* L, size_t n) {
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < n; v += (size_t)gridDim.x * 256)
        if (L[v] >= 0) L[v] = cc_find(L, (int)v);               // racing writers store ancestors only: still valid
}
// component sizes: lanes that follow each other with the same root add once (one atomic per run segment, integer: exact)
__global__ __launch_bounds__(256) void cc_count_kernel(const int* L, int* cnt, unsigned n) {
    const unsigned lane = threadIdx.x & 63;
    const unsigned nround = (n + 255u) & ~255u;
    for (unsigned v = blockIdx.x * 256u + threadIdx.x; v < nround; v += gridDim.x * 256u) {
        const int r = v < n ? L[v] : -1;
        const int rp = __shfl_up(r, 1, 64);
        const bool lead = r >= 0 && (lane == 0 || rp != r);
        const unsigned long long brk = __ballot(lead || r < 0);          // positions where a new segment (or background) begins
        if (lead) {
            const unsigned long long after = lane == 63 ? 0ull : (brk >> (lane + 1));
            const unsigned len = after ? (unsigned)__builtin_ctzll(after) + 1u : 64u - lane;
            atomicAdd(cnt + r, (int)len);
        }
    }
}
__global__ __launch_bounds__(256) void cc_filter_kernel(const int* L, const int* cnt, uint8_t* out, size_t n, int min_volume) {
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < n; v += (size_t)gridDim.x * 256)
        out[v] = (L[v] >= 0 && cnt[L[v]] >= min_volume) ? 1 : 0;
}
__global__ __launch_bounds__(256) void cc_border_kernel(const int* L, int* flag, int D, int H, int W) {
    const size_t n = (size_t)D * H * W; const int HW = H * W;
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < n; v += (size_t)gridDim.x * 256) {
        if (L[v] < 0) continue;
        const int x = (int)(v % W), y = (int)((v / W) % H), z = (int)(v / HW);
        if (x == 0 || x == W - 1 || y == 0 || y == H - 1 || z == 0 || z == D - 1) flag[L[v]] = 1;
    }
}
__global__ __launch_bounds__(256) void cc_fill_kernel(const int* L, const int* flag, uint8_t* out, size_t n) {
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < n; v += (size_t)gridDim.x * 256)
        out[v] = (L[v] < 0 || flag[L[v]] == 0) ? 1 : 0;         // foreground, or background that cannot reach the border
}

static int cc_label(const uint8_t* mask, int* labels, int D, int H, int W, int invert, hipStream_t st) {
    const size_t n = (size_t)D * H * W;
    hipLaunchKernelGGL(cc_init_kernel, dim3(vgrid(n)), dim3(256), 0, st, mask, labels, (unsigned)n, W, invert);
    hipLaunchKernelGGL(cc_merge_kernel, dim3(vgrid(n)), dim3(256), 0, st, labels, D, H, W);
    hipLaunchKernelGGL(cc_compress_kernel, dim3(vgrid(n)), dim3(256), 0, st, labels, n);
    return wmh_check_launch("cc_label kernels");
}
static bool bad_dims(int D, int H, int W) { return D < 1 || H < 1 || W < 1 || (size_t)D * H * W >= ((size_t)1 << 31); }

extern "C" int wmh_cc_label(const uint8_t* mask, int32_t* labels, int D, int H, int W, void* stream) {
    if (!mask || !labels || bad_dims(D, H, W)) return wmh_fail(WMH_EINVAL, "cc_label: bad args");
    return cc_label(mask, labels, D, H, W, 0, (hipStream_t)stream);
}
extern "C" int wmh_remove_small_components(const uint8_t* mask, uint8_t* out, int32_t* labels, int32_t* counts, int D, int H,
                                           int W, int min_volume, void* stream) {
    if (!mask || !out || !labels || !counts || bad_dims(D, H, W)) return wmh_fail(WMH_EINVAL, "remove_small_components: bad args");
    hipStream_t st = (hipStream_t)stream;
    const size_t n = (size_t)D * H * W;
    int rc = cc_label(mask, labels, D, H, W, 0, st);
    if (rc != WMH_OK) return rc;
    if (hipMemsetAsync(counts, 0, n * sizeof(int), st) != hipSuccess) return wmh_fail(WMH_ELAUNCH, "remove_small_components: memset");
    hipLaunchKernelGGL(cc_count_kernel, dim3(vgrid(n)), dim3(256), 0, st, labels, counts, (unsigned)n);
    hipLaunchKernelGGL(cc_filter_kernel, dim3(vgrid(n)), dim3(256), 0, st, labels, counts, out, n, min_volume);
    return wmh_check_launch("remove_small_components kernels");
}
extern "C" int wmh_fill_holes(const uint8_t* mask, uint8_t* out, int32_t* labels, int32_t* flags, int D, int H, int W, void* stream) {
    if (!mask || !out || !labels || !flags || bad_dims(D, H, W)) return wmh_fail(WMH_EINVAL, "fill_holes: bad args");
    hipStream_t st = (hipStream_t)stream;
    const size_t n = (size_t)D * H * W;
    int rc = cc_label(mask, labels, D, H, W, 1, st);            // components of the BACKGROUND
    if (rc != WMH_OK) return rc;
    if (hipMemsetAsync(flags, 0, n * sizeof(int), st) != hipSuccess) return wmh_fail(WMH_ELAUNCH, "fill_holes: memset");
    hipLaunchKernelGGL(cc_border_kernel, dim3(vgrid(n)), dim3(256), 0, st, labels, flags, D, H, W);
    hipLaunchKernelGGL(cc_fill_kernel, dim3(vgrid(n)), dim3(256), 0, st, labels, flags, out, n);
    return wmh_check_launch("fill_holes kernels");
}

// ---- non-zero mask, bounding box, masked z-score + crop, un-crop ---------------------------------------------------
__global__ __launch_bounds__(256) void nonzero_mask_kernel(const float* data, uint8_t* mask, int C, size_t n) {
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < n; v += (size_t)gridDim.x * 256) {
        bool nz = false;
        for (int c = 0; c < C; ++c) nz |= data[(size_t)c * n + v] != 0.f;
        mask[v] = nz ? 1 : 0;
    }
}
// bbox[6] = {z0, z1, y0, y1, x0, x1} (half-open), pre-set by the host wrapper to {D,0,H,0,W,0}
__global__ __launch_bounds__(256) void bbox_kernel(const uint8_t* mask, int* bbox, int D, int H, int W) {
    __shared__ int sh[6];
    if (threadIdx.x < 6) sh[threadIdx.x] = (threadIdx.x & 1) ? 0 : 0x7fffffff;
    __syncthreads();
    const size_t n = (size_t)D * H * W; const int HW = H * W;
    int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {0, 0, 0};
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < n; v += (size_t)gridDim.x * 256) {
        if (!mask[v]) continue;
        const int c[3] = {(int)(v / HW), (int)((v / W) % H), (int)(v % W)};
#pragma unroll
        for (int a = 0; a < 3; ++a) { lo[a] = min(lo[a], c[a]); hi[a] = max(hi[a], c[a] + 1); }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) { atomicMin(&sh[2 * a], lo[a]); atomicMax(&sh[2 * a + 1], hi[a]); }
    __syncthreads();
    if (threadIdx.x < 6) { if (threadIdx.x & 1) atomicMax(bbox + threadIdx.x, sh[threadIdx.x]); else atomicMin(bbox + threadIdx.x, sh[threadIdx.x]); }
}
__global__ void bbox_init_kernel(int* bbox, int D, int H, int W) {
    bbox[0] = D; bbox[1] = 0; bbox[2] = H; bbox[3] = 0; bbox[4] = W; bbox[5] = 0;
}
__global__ void bbox_fix_kernel(int* bbox, int D, int H, int W) {      // empty mask: the whole volume (as the host code does)
    if (bbox[1] == 0) { bbox[0] = 0; bbox[1] = D; bbox[2] = 0; bbox[3] = H; bbox[4] = 0; bbox[5] = W; }
}
extern "C" int wmh_nonzero_mask(const float* data, uint8_t* mask, int C, int D, int H, int W, void* stream) {
    if (!data || !mask || C < 1 || bad_dims(D, H, W)) return wmh_fail(WMH_EINVAL, "nonzero_mask: bad args");
    const size_t n = (size_t)D * H * W;
    hipLaunchKernelGGL(nonzero_mask_kernel, dim3(vgrid(n)), dim3(256), 0, (hipStream_t)stream, data, mask, C, n);
    return wmh_check_launch("nonzero_mask_kernel");
}
extern "C" int wmh_mask_bbox(const uint8_t* mask, int32_t* bbox6, int D, int H, int W, void* stream) {
    if (!mask || !bbox6 || bad_dims(D, H, W)) return wmh_fail(WMH_EINVAL, "mask_bbox: bad args");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bbox_init_kernel, dim3(1), dim3(1), 0, st, bbox6, D, H, W);
    hipLaunchKernelGGL(bbox_kernel, dim3(vgrid((size_t)D * H * W, 256 * 16, 1024)), dim3(256), 0, st, mask, bbox6, D, H, W);
    hipLaunchKernelGGL(bbox_fix_kernel, dim3(1), dim3(1), 0, st, bbox6, D, H, W);
    return wmh_check_launch("bbox kernels");
}

// Masked moments of one channel inside the box: fixed-order two-level reduction in double (sum, sum of squares, count).
constexpr int ZS_BLOCKS = 512;
__device__ __forceinline__ double zs_block_sum(double v, double* sh) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}
__global__ __launch_bounds__(256) void zscore_partial_kernel(const float* data, const uint8_t* mask, const int* bbox, double* part,
                                                              int D, int H, int W, int use_mask) {
    __shared__ double sh[4];
    const int c = blockIdx.y;
    const int z0 = bbox[0], y0 = bbox[2], x0 = bbox[4];
    const int cd = bbox[1] - z0, chh = bbox[3] - y0, cw = bbox[5] - x0;
    const size_t n = (size_t)cd * chh * cw, vol = (size_t)D * H * W;
    double s = 0, q = 0, k = 0;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
        const int x = (int)(e % cw), y = (int)((e / cw) % chh), z = (int)(e / ((size_t)cw * chh));
        const size_t v = ((size_t)(z0 + z) * H + (y0 + y)) * W + (x0 + x);
        if (use_mask && !mask[v]) continue;
        const double t = data[(size_t)c * vol + v];
        s += t; q += t * t; k += 1;
    }
    double* o = part + ((size_t)c * ZS_BLOCKS + blockIdx.x) * 3;
    double r = zs_block_sum(s, sh); if (threadIdx.x == 0) o[0] = r;
    r = zs_block_sum(q, sh);        if (threadIdx.x == 0) o[1] = r;
    r = zs_block_sum(k, sh);        if (threadIdx.x == 0) o[2] = r;
}
__global__ __launch_bounds__(64) void zscore_final_kernel(const double* part, float* stats) {     // stats[c] = {mean, std}
    const int c = blockIdx.x;
    double s = 0, q = 0, k = 0;
    for (int b = threadIdx.x; b < ZS_BLOCKS; b += 64) { const double* o = part + ((size_t)c * ZS_BLOCKS + b) * 3; s += o[0]; q += o[1]; k += o[2]; }
    for (int off = 32; off > 0; off >>= 1) { s += __shfl_down(s, off, 64); q += __shfl_down(q, off, 64); k += __shfl_down(k, off, 64); }
    if (threadIdx.x == 0) {
        const double mean = k > 0 ? s / k : 0.0;
        double var = k > 0 ? q / k - mean * mean : 0.0; if (var < 0) var = 0;
        stats[2 * c] = (float)mean; stats[2 * c + 1] = (float)sqrt(var);
    }
}
__global__ __launch_bounds__(256) void zscore_crop_kernel(const float* data, const uint8_t* mask, const int* bbox, const float* stats,
                                                           float* out, uint8_t* mask_out, int D, int H, int W, int use_mask) {
    const int c = blockIdx.y;
    const int z0 = bbox[0], y0 = bbox[2], x0 = bbox[4];
    const int cd = bbox[1] - z0, chh = bbox[3] - y0, cw = bbox[5] - x0;
    const size_t n = (size_t)cd * chh * cw, vol = (size_t)D * H * W;
    const float mean = stats[2 * c], den = stats[2 * c + 1] + 1e-8f;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
        const int x = (int)(e % cw), y = (int)((e / cw) % chh), z = (int)(e / ((size_t)cw * chh));
        const size_t v = ((size_t)(z0 + z) * H + (y0 + y)) * W + (x0 + x);
        const bool m = mask[v] != 0;
        const float t = (data[(size_t)c * vol + v] - mean) / den;
        out[(size_t)c * n + e] = (!use_mask || m) ? t : 0.f;
        if (c == 0 && mask_out) mask_out[e] = m ? 1 : 0;
    }
}
extern "C" size_t wmh_zscore_crop_workspace_doubles(int C) { return (size_t)C * ZS_BLOCKS * 3; }
extern "C" int wmh_zscore_crop(const float* data, const uint8_t* mask, const int32_t* bbox6, float* out, uint8_t* mask_out,
                               float* stats, double* workspace, int C, int D, int H, int W, int use_mask, void* stream) {
    if (!data || !mask || !bbox6 || !out || !stats || !workspace || C < 1 || bad_dims(D, H, W))
        return wmh_fail(WMH_EINVAL, "zscore_crop: bad args");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(zscore_partial_kernel, dim3(ZS_BLOCKS, C), dim3(256), 0, st, data, mask, bbox6, workspace, D, H, W, use_mask);
    hipLaunchKernelGGL(zscore_final_kernel, dim3(C), dim3(64), 0, st, workspace, stats);
    hipLaunchKernelGGL(zscore_crop_kernel, dim3(2048, C), dim3(256), 0, st, data, mask, bbox6, stats, out, mask_out, D, H, W, use_mask);
    return wmh_check_launch("zscore_crop kernels");
}
// full[z0:z1, y0:y1, x0:x1] = cropped, `fill` elsewhere (segmentation_export un-crop)
__global__ __launch_bounds__(256) void uncrop_u8_kernel(const uint8_t* crop, const int* bbox, uint8_t* full, int D, int H, int W, int fill) {
    const size_t n = (size_t)D * H * W; const int HW = H * W;
    const int z0 = bbox[0], z1 = bbox[1], y0 = bbox[2], y1 = bbox[3], x0 = bbox[4], x1 = bbox[5];
    const int chh = y1 - y0, cw = x1 - x0;
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < n; v += (size_t)gridDim.x * 256) {
        const int x = (int)(v % W), y = (int)((v / W) % H), z = (int)(v / HW);
        const bool in = z >= z0 && z < z1 && y >= y0 && y < y1 && x >= x0 && x < x1;
        full[v] = in ? crop[((size_t)(z - z0) * chh + (y - y0)) * cw + (x - x0)] : (uint8_t)fill;
    }
}
extern "C" int wmh_uncrop_u8(const uint8_t* crop, const int32_t* bbox6, uint8_t* full, int D, int H, int W, int fill, void* stream) {
    if (!crop || !bbox6 || !full || bad_dims(D, H, W)) return wmh_fail(WMH_EINVAL, "uncrop_u8: bad args");
    hipLaunchKernelGGL(uncrop_u8_kernel, dim3(vgrid((size_t)D * H * W)), dim3(256), 0, (hipStream_t)stream, crop, bbox6, full, D, H, W, fill);
    return wmh_check_launch("uncrop_u8_kernel");
}

// ---- Stage-II glue of the self-training loop (reference deepwmh/pipeline/DCNN_multistage.py:102-125) ----------------
// Element-wise fp32 in numpy's operation order, no fused multiply-add: results are bit-identical to the reference's.
#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void softmax_mask_kernel(const float* x, const float* m, float* y, size_t n) {
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < n; v += (size_t)gridDim.x * 256) {
        const float t = 1.0f - x[v];
        const float u = m[v] * t;
        y[v] = 1.0f - u;                                        // y = 1-(m*(1-x)), DCNN_multistage.py:108
    }
}
__global__ __launch_bounds__(256) void ensemble_kernel(const float* stack, int E, float* field, uint8_t* label, size_t n) {
    const float fe = (float)E;
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < n; v += (size_t)gridDim.x * 256) {
        float s = 0.f;
        for (int e = 0; e < E; ++e) s = s + stack[(size_t)e * n + v];    // scalar_field += ..., in file order (:115-117)
        s = s / fe;                                                    // :118
        if (field) field[v] = s;
        label[v] = s < 0.5f ? 1 : 0;                                   // inverted softmax: < 0.5 is lesion (:119)
    }
}
__global__ __launch_bounds__(256) void overlap_kernel(const float* a, const float* b, unsigned long long* out3, size_t n) {
    __shared__ unsigned int sh[3];
    if (threadIdx.x < 3) sh[threadIdx.x] = 0;
    __syncthreads();
    unsigned int tp = 0, sa = 0, sb = 0;
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < n; v += (size_t)gridDim.x * 256) {
        const bool pa = a[v] > 0.5f, pb = b[v] > 0.5f;
        tp += (pa && pb) ? 1u : 0u; sa += pa ? 1u : 0u; sb += pb ? 1u : 0u;
    }
    atomicAdd(&sh[0], tp); atomicAdd(&sh[1], sa); atomicAdd(&sh[2], sb);
    __syncthreads();
    if (threadIdx.x < 3) atomicAdd(out3 + threadIdx.x, (unsigned long long)sh[threadIdx.x]);
}
extern "C" int wmh_softmax_mask(const float* x, const float* m, float* y, size_t n, void* stream) {
    if (!x || !m || !y) return wmh_fail(WMH_EINVAL, "softmax_mask: bad args");
    if (n) hipLaunchKernelGGL(softmax_mask_kernel, dim3(vgrid(n, 1024)), dim3(256), 0, (hipStream_t)stream, x, m, y, n);
    return wmh_check_launch("softmax_mask_kernel");
}
extern "C" int wmh_ensemble_softmax(const float* stack, int E, float* field, uint8_t* label, size_t n, void* stream) {
    if (!stack || !label || E < 1) return wmh_fail(WMH_EINVAL, "ensemble_softmax: bad args");
    if (n) hipLaunchKernelGGL(ensemble_kernel, dim3(vgrid(n, 1024)), dim3(256), 0, (hipStream_t)stream, stack, E, field, label, n);
    return wmh_check_launch("ensemble_kernel");
}
extern "C" int wmh_binary_overlap(const float* a, const float* b, unsigned long long* out3, size_t n, void* stream) {
    if (!a || !b || !out3) return wmh_fail(WMH_EINVAL, "binary_overlap: bad args");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(out3, 0, 3 * sizeof(unsigned long long), st) != hipSuccess) return wmh_fail(WMH_ELAUNCH, "binary_overlap: memset");
    if (n) hipLaunchKernelGGL(overlap_kernel, dim3(vgrid(n, 256 * 16, 2048)), dim3(256), 0, st, a, b, out3, n);
    return wmh_check_launch("overlap_kernel");
}
