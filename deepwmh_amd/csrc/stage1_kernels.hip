// stage1_kernels.hip -- DeepWMH Stage I (negative-log-likelihood initial segmentation) on the GPU.
// The reference is float64 numpy (deepwmh/analysis/lesion_analysis.py:84-278, image_ops.py:56-231,256-308,379-421);
// every pass here is a whole-volume, HBM-bound kernel in double.  Reductions are two-level with a fixed order
// (no floating-point atomics); integer results (masks, votes, histograms, components) are exact.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "wmh_common.h"

#pragma clang fp contract(off)

namespace s1 {
constexpr int RB = 1024;                                     // partial blocks of the reductions
static inline unsigned grid(size_t n, unsigned cap = 1u << 20) { size_t b = (n + 255) / 256; if (b < 1) b = 1; if (b > cap) b = cap; return (unsigned)b; }
}

__device__ __forceinline__ double s1_block_sum(double v, double* sh) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}
__device__ __forceinline__ double s1_block_min(double v, double* sh) {
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_down(v, off, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmin(fmin(sh[0], sh[1]), fmin(sh[2], sh[3]));
}

// part[b] = {count, sum, sum of squares (about `shift`), min inside mask, min over everything, max inside, max over everything}
template <typename T>
__global__ __launch_bounds__(256) void moments_partial_kernel(const T* x, const uint8_t* mask, size_t n, double shift, double* part) {
    __shared__ double sh[4];
    double k = 0, s = 0, q = 0, mn = INFINITY, mna = INFINITY, mx = -INFINITY, mxa = -INFINITY;
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < n; v += (size_t)gridDim.x * 256) {
        const double t = (double)x[v];
        mna = fmin(mna, t); mxa = fmax(mxa, t);
        if (mask && !mask[v]) continue;
        const double d = t - shift;
        k += 1; s += d; q += d * d; mn = fmin(mn, t); mx = fmax(mx, t);
    }
    double* o = part + (size_t)blockIdx.x * 7;
    double r = s1_block_sum(k, sh); if (threadIdx.x == 0) o[0] = r;
    r = s1_block_sum(s, sh);        if (threadIdx.x == 0) o[1] = r;
    r = s1_block_sum(q, sh);        if (threadIdx.x == 0) o[2] = r;
    r = s1_block_min(mn, sh);       if (threadIdx.x == 0) o[3] = r;
    r = s1_block_min(mna, sh);      if (threadIdx.x == 0) o[4] = r;
    r = -s1_block_min(-mx, sh);     if (threadIdx.x == 0) o[5] = r;
    r = -s1_block_min(-mxa, sh);    if (threadIdx.x == 0) o[6] = r;
}
__global__ __launch_bounds__(64) void moments_final_kernel(const double* part, int nb, double shift, double* out5) {
    double k = 0, s = 0, q = 0, mn = INFINITY, mna = INFINITY, mx = -INFINITY, mxa = -INFINITY;
    for (int b = threadIdx.x; b < nb; b += 64) {
        const double* o = part + (size_t)b * 7; k += o[0]; s += o[1]; q += o[2]; mn = fmin(mn, o[3]); mna = fmin(mna, o[4]);
        mx = fmax(mx, o[5]); mxa = fmax(mxa, o[6]);
    }
    for (int off = 32; off > 0; off >>= 1) {
        k += __shfl_down(k, off, 64); s += __shfl_down(s, off, 64); q += __shfl_down(q, off, 64);
        mn = fmin(mn, __shfl_down(mn, off, 64)); mna = fmin(mna, __shfl_down(mna, off, 64));
        mx = fmax(mx, __shfl_down(mx, off, 64)); mxa = fmax(mxa, __shfl_down(mxa, off, 64));
    }
    if (threadIdx.x == 0) {
        const double mean_d = k > 0 ? s / k : 0.0;
        double var = k > 0 ? q / k - mean_d * mean_d : 0.0; if (var < 0) var = 0;
        out5[0] = k; out5[1] = mean_d + shift; out5[2] = sqrt(var); out5[3] = mn; out5[4] = mna; out5[5] = mx; out5[6] = mxa;
    }
}
// out7 (device) = {count, mean, population std, min inside the mask, min over all, max inside the mask, max over all}.  The sums run about
// `shift` (pass a rough mean, e.g. 0 for z-scored data) so that the variance does not cancel.
extern "C" size_t wmh_moments_workspace_doubles(void) { return (size_t)s1::RB * 7; }
extern "C" int wmh_masked_moments_f32(const float* x, const uint8_t* mask, size_t n, double shift, double* out7, double* workspace, void* stream) {
    if (!x || !out7 || !workspace) return wmh_fail(WMH_EINVAL, "masked_moments_f32: bad args");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL((moments_partial_kernel<float>), dim3(s1::RB), dim3(256), 0, st, x, mask, n, shift, workspace);
    hipLaunchKernelGGL(moments_final_kernel, dim3(1), dim3(64), 0, st, workspace, s1::RB, shift, out7);
    return wmh_check_launch("moments kernels");
}
extern "C" int wmh_masked_moments_f64(const double* x, const uint8_t* mask, size_t n, double shift, double* out7, double* workspace, void* stream) {
    if (!x || !out7 || !workspace) return wmh_fail(WMH_EINVAL, "masked_moments_f64: bad args");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL((moments_partial_kernel<double>), dim3(s1::RB), dim3(256), 0, st, x, mask, n, shift, workspace);
    hipLaunchKernelGGL(moments_final_kernel, dim3(1), dim3(64), 0, st, workspace, s1::RB, shift, out7);
    return wmh_check_launch("moments kernels");
}

// z_score (image_ops.py:172-179) followed by the "replace everything outside the rough brain mask" step
// (lesion_analysis.py:143,155-156,164-166):  out = (x - mean) / std;  if fill_mask: out = mask ? out : fill
__global__ __launch_bounds__(256) void zscore_fill_kernel(const float* x, double mean, double stdv, const uint8_t* fill_mask, const double* fill,
                                                           double* out, size_t n) {
    const double f = fill ? fill[0] : 0.0;
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < n; v += (size_t)gridDim.x * 256) {
        const double t = ((double)x[v] - mean) / stdv;
        out[v] = (fill_mask && !fill_mask[v]) ? f : t;
    }
}
extern "C" int wmh_zscore_f64(const float* x, double mean, double stdv, const uint8_t* fill_mask, const double* fill_value_dev, double* out,
                              size_t n, void* stream) {
    if (!x || !out || (fill_mask && !fill_value_dev)) return wmh_fail(WMH_EINVAL, "zscore_f64: bad args");
    if (n) hipLaunchKernelGGL(zscore_fill_kernel, dim3(s1::grid(n, 4096)), dim3(256), 0, (hipStream_t)stream, x, mean, stdv, fill_mask, fill_value_dev, out, n);
    return wmh_check_launch("zscore_fill_kernel");
}
__global__ __launch_bounds__(256) void mask_fill_kernel(double* x, const uint8_t* mask, const double* fill, size_t n) {
    const double f = fill[0];
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < n; v += (size_t)gridDim.x * 256)
        if (!mask[v]) x[v] = f;
}
extern "C" int wmh_mask_fill_f64(double* x, const uint8_t* mask, const double* fill_value_dev, size_t n, void* stream) {
    if (!x || !mask || !fill_value_dev) return wmh_fail(WMH_EINVAL, "mask_fill_f64: bad args");
    if (n) hipLaunchKernelGGL(mask_fill_kernel, dim3(s1::grid(n, 4096)), dim3(256), 0, (hipStream_t)stream, x, mask, fill_value_dev, n);
    return wmh_check_launch("mask_fill_kernel");
}

// out = 2 * |{r : labels[r][v] > 0.5}| > R   (group_mean(...) > 0.5 of 0/1 maps, lesion_analysis.py:139-141; majority vote :239-242)
__global__ __launch_bounds__(256) void vote_mask_kernel(const float* labels, int R, uint8_t* out, size_t n) {
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < n; v += (size_t)gridDim.x * 256) {
        int k = 0;
        for (int r = 0; r < R; ++r) k += labels[(size_t)r * n + v] > 0.5f ? 1 : 0;
        out[v] = 2 * k > R ? 1 : 0;
    }
}
extern "C" int wmh_vote_mask(const float* labels, int R, uint8_t* out, size_t n, void* stream) {
    if (!labels || !out || R < 1) return wmh_fail(WMH_EINVAL, "vote_mask: bad args");
    if (n) hipLaunchKernelGGL(vote_mask_kernel, dim3(s1::grid(n, 4096)), dim3(256), 0, (hipStream_t)stream, labels, R, out, n);
    return wmh_check_launch("vote_mask_kernel");
}
// average_contiguous_labels (image_ops.py:23-38): argmax over the per-id vote counts, first maximum wins
__global__ __launch_bounds__(256) void label_vote_kernel(const float* labels, int R, int K, uint8_t* out, size_t n) {
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < n; v += (size_t)gridDim.x * 256) {
        int best = 0, bestc = -1;
        for (int c = 0; c < K; ++c) {
            int k = 0;
            for (int r = 0; r < R; ++r) k += ((int)labels[(size_t)r * n + v] == c) ? 1 : 0;     // astype('int') truncates
            if (k > bestc) { bestc = k; best = c; }
        }
        out[v] = (uint8_t)best;
    }
}
extern "C" int wmh_label_vote(const float* labels, int R, int K, uint8_t* out, size_t n, void* stream) {
    if (!labels || !out || R < 1 || K < 1 || K > 255) return wmh_fail(WMH_EINVAL, "label_vote: bad args");
    if (n) hipLaunchKernelGGL(label_vote_kernel, dim3(s1::grid(n, 4096)), dim3(256), 0, (hipStream_t)stream, labels, R, K, out, n);
    return wmh_check_launch("label_vote_kernel");
}

// numpy.histogram with explicit increasing edges: bin i = [e[i], e[i+1]) and the last bin closed; values outside are ignored.
// The candidate index comes from the (uniform) spacing and is corrected against the actual edges, as numpy does.
__global__ __launch_bounds__(256) void histogram_kernel(const double* x, const uint8_t* mask, const uint8_t* replace_mask, const double* replace_value,
                                                         size_t n, const double* edges, int nbins, unsigned long long* counts) {
    extern __shared__ unsigned int shc[];
    for (int b = threadIdx.x; b < nbins; b += 256) shc[b] = 0;
    __syncthreads();
    const double e0 = edges[0], e1 = edges[nbins], inv = (double)nbins / (e1 - e0);
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < n; v += (size_t)gridDim.x * 256) {
        if (mask && !mask[v]) continue;
        double t = x[v];
        if (replace_mask && !replace_mask[v]) t = replace_value[0];
        if (!(t >= e0 && t <= e1)) continue;
        int b = (int)((t - e0) * inv);
        if (b >= nbins) b = nbins - 1;
        if (b < 0) b = 0;
        while (b > 0 && t < edges[b]) --b;
        while (b < nbins - 1 && t >= edges[b + 1]) ++b;
        atomicAdd(&shc[b], 1u);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < nbins; b += 256) if (shc[b]) atomicAdd(counts + b, (unsigned long long)shc[b]);
}
extern "C" int wmh_histogram_f64(const double* x, const uint8_t* mask, const uint8_t* replace_mask, const double* replace_value_dev, size_t n,
                                 const double* edges_dev, int nbins, unsigned long long* counts, void* stream) {
    if (!x || !edges_dev || !counts || nbins < 1 || nbins > 8192 || (replace_mask && !replace_value_dev)) return wmh_fail(WMH_EINVAL, "histogram_f64: bad args");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(counts, 0, (size_t)nbins * 8, st) != hipSuccess) return wmh_fail(WMH_ELAUNCH, "histogram_f64: memset");
    if (n) hipLaunchKernelGGL(histogram_kernel, dim3(s1::grid(n, 2048)), dim3(256), (size_t)nbins * 4, st, x, mask, replace_mask, replace_value_dev, n,
                              edges_dev, nbins, counts);
    return wmh_check_launch("histogram_kernel");
}

// mean_std_grid, block statistics (image_ops.py:105-148): windows of `patch` voxels every `step` over the zero-padded,
// float32-rounded data; masked mean / std (0 and 1e-5 for an empty window); without a mask std is floored at 1e-5.
__global__ __launch_bounds__(256) void block_stats_kernel(const double* x, const uint8_t* mask, int D, int H, int W, int pz, int py, int px,
                                                           int sz, int sy, int sx, int gz, int gy, int gx, double* mean_grid, double* std_grid) {
    __shared__ double sh[4];
    const int cell = blockIdx.x;
    const int k = cell % gx, j = (cell / gx) % gy, i = cell / (gx * gy);
    const int z0 = i * sz, y0 = j * sy, x0 = k * sx;
    const int nwin = pz * py * px;
    const int pD = pz * ((D + pz - 1) / pz), pH = py * ((H + py - 1) / py), pW = px * ((W + px - 1) / px);   // padded extents
    double cnt = 0, s = 0, q = 0;
    for (int e = threadIdx.x; e < nwin; e += 256) {
        const int xx = x0 + e % px, yy = y0 + (e / px) % py, zz = z0 + e / (px * py);
        if (zz >= pD || yy >= pH || xx >= pW) continue;             // numpy slicing clips the last half-windows
        const bool in = zz < D && yy < H && xx < W;                 // beyond the array: padding zeros (mask zero)
        const size_t v = in ? ((size_t)zz * H + yy) * W + xx : 0;
        if (mask) { if (!in || !mask[v]) continue; }
        const double t = in ? (double)(float)x[v] : 0.0;
        cnt += 1; s += t; q += t * t;
    }
    cnt = s1_block_sum(cnt, sh); s = s1_block_sum(s, sh); q = s1_block_sum(q, sh);
    if (threadIdx.x == 0) {
        double mu = 0.0, sigma = 0.00001;
        if (cnt > 0) {
            mu = s / cnt;
            double var = q / cnt - mu * mu; if (var < 0) var = 0;
            sigma = sqrt(var);
            if (!mask && sigma < 0.00001) sigma = 0.00001;
        }
        mean_grid[cell] = mu; std_grid[cell] = sigma;
    }
}
extern "C" int wmh_block_stats(const double* x, const uint8_t* mask, int D, int H, int W, const int* patch3, const int* step3, const int* grid3,
                               double* mean_grid, double* std_grid, void* stream) {
    if (!x || !patch3 || !step3 || !grid3 || !mean_grid || !std_grid) return wmh_fail(WMH_EINVAL, "block_stats: bad args");
    const long long cells = (long long)grid3[0] * grid3[1] * grid3[2];
    if (cells < 1 || cells > 0x7fffffffLL) return wmh_fail(WMH_EINVAL, "block_stats: bad grid");
    hipLaunchKernelGGL(block_stats_kernel, dim3((unsigned)cells), dim3(256), 0, (hipStream_t)stream, x, mask, D, H, W, patch3[0], patch3[1], patch3[2],
                       step3[0], step3[1], step3[2], grid3[0], grid3[1], grid3[2], mean_grid, std_grid);
    return wmh_check_launch("block_stats_kernel");
}
// scipy.ndimage.zoom(order=1) of the zero-surrounded grid, evaluated only on the window the reference keeps
// (image_ops.py:150-168): u = o * (in - 1) / (out - 1) per axis; out[z,y,x] for o = off + (z,y,x).
__global__ __launch_bounds__(256) void zoom_linear_window_kernel(const double* g, int gz, int gy, int gx, int oz, int oy, int ox, int offz, int offy,
                                                                  int offx, int D, int H, int W, double* out) {
    const size_t n = (size_t)D * H * W;
    const double rz = oz > 1 ? (double)(gz - 1) / (double)(oz - 1) : 0.0, ry = oy > 1 ? (double)(gy - 1) / (double)(oy - 1) : 0.0,
                 rx = ox > 1 ? (double)(gx - 1) / (double)(ox - 1) : 0.0;
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < n; v += (size_t)gridDim.x * 256) {
        const int x = (int)(v % W), y = (int)((v / W) % H), z = (int)(v / ((size_t)W * H));
        const double uz = (z + offz) * rz, uy = (y + offy) * ry, ux = (x + offx) * rx;
        const double fz = floor(uz), fy = floor(uy), fx = floor(ux);
        const double wz[2] = {1.0 - (uz - fz), uz - fz}, wy[2] = {1.0 - (uy - fy), uy - fy}, wx[2] = {1.0 - (ux - fx), ux - fx};
        double t = 0.0;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int zz = (int)fz + a, yy = (int)fy + b, xx = (int)fx + c;
                    double coeff = (zz < gz && yy < gy && xx < gx) ? g[((size_t)zz * gy + yy) * gx + xx] : 0.0;    // mode 'constant', cval 0
                    coeff *= wz[a]; coeff *= wy[b]; coeff *= wx[c];
                    t += coeff;
                }
        out[v] = t;
    }
}
extern "C" int wmh_zoom_linear_window(const double* grid, const int* grid3, const int* zoomed3, const int* offset3, int D, int H, int W, double* out,
                                      void* stream) {
    if (!grid || !grid3 || !zoomed3 || !offset3 || !out || D < 1 || H < 1 || W < 1) return wmh_fail(WMH_EINVAL, "zoom_linear_window: bad args");
    hipLaunchKernelGGL(zoom_linear_window_kernel, dim3(s1::grid((size_t)D * H * W, 8192)), dim3(256), 0, (hipStream_t)stream, grid, grid3[0], grid3[1],
                       grid3[2], zoomed3[0], zoomed3[1], zoomed3[2], offset3[0], offset3[1], offset3[2], D, H, W, out);
    return wmh_check_launch("zoom_linear_window_kernel");
}
// x = x - a + b  (align a reference's local mean to the target's, lesion_analysis.py:175)
__global__ __launch_bounds__(256) void sub_add_kernel(double* x, const double* a, const double* b, size_t n) {
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < n; v += (size_t)gridDim.x * 256) { const double t = x[v] - a[v]; x[v] = t + b[v]; }
}
extern "C" int wmh_sub_add_f64(double* x, const double* a, const double* b, size_t n, void* stream) {
    if (!x || !a || !b) return wmh_fail(WMH_EINVAL, "sub_add_f64: bad args");
    if (n) hipLaunchKernelGGL(sub_add_kernel, dim3(s1::grid(n, 4096)), dim3(256), 0, (hipStream_t)stream, x, a, b, n);
    return wmh_check_launch("sub_add_kernel");
}
// x *= mask (mask as 0/1 bytes), or x = mask ? y : x
__global__ __launch_bounds__(256) void mul_mask_kernel(double* x, const uint8_t* m, size_t n) {
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < n; v += (size_t)gridDim.x * 256) x[v] = x[v] * (m[v] ? 1.0 : 0.0);
}
__global__ __launch_bounds__(256) void select_kernel(double* x, const double* y, const uint8_t* m, size_t n) {
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < n; v += (size_t)gridDim.x * 256) if (m[v]) x[v] = y[v];
}
extern "C" int wmh_mul_mask_f64(double* x, const uint8_t* mask, size_t n, void* stream) {
    if (!x || !mask) return wmh_fail(WMH_EINVAL, "mul_mask_f64: bad args");
    if (n) hipLaunchKernelGGL(mul_mask_kernel, dim3(s1::grid(n, 4096)), dim3(256), 0, (hipStream_t)stream, x, mask, n);
    return wmh_check_launch("mul_mask_kernel");
}
extern "C" int wmh_select_f64(double* x, const double* y, const uint8_t* mask, size_t n, void* stream) {
    if (!x || !y || !mask) return wmh_fail(WMH_EINVAL, "select_f64: bad args");
    if (n) hipLaunchKernelGGL(select_kernel, dim3(s1::grid(n, 4096)), dim3(256), 0, (hipStream_t)stream, x, y, mask, n);
    return wmh_check_launch("select_kernel");
}

// nll (lesion_analysis.py:84-113), unmasked statistics: mu = mean_r refs, sigma = population std (two pass, as nanstd),
// floored at min_std (or + 1e-6 when min_std < 0), anomaly = (x-mu)^2 / (2 sigma^2) + log(sigma * 2.506), NaN -> 0,
// one-sided when side = +1 / -1.  Also the anomaly of every reference against the same statistics (:188-193).
__device__ __forceinline__ double nll_one(double x, double mu, double sigma, int side) {
    const double d = x - mu;
    double a = (d * d) / (2.0 * (sigma * sigma)) + log(sigma * 2.506);
    if (a != a) a = 0.0;                                        // nan_to_num(nan=0.0); +-inf -> largest finite double
    else if (isinf(a)) a = a > 0 ? 1.7976931348623157e308 : -1.7976931348623157e308;
    if (side > 0) a = a * (x > mu ? 1.0 : 0.0);
    else if (side < 0) a = a * (x < mu ? 1.0 : 0.0);
    return a;
}
// thr != NULL (use_mask=True, lesion_analysis.py:87-92): reference r counts at a voxel only where refs[r] > thr[r] (its Otsu
// threshold); nanmean / nanstd over the counted ones, NaN where none counts (the anomaly there becomes 0 by nan_to_num).
__global__ __launch_bounds__(256) void nll_kernel(const double* xp, const double* refs, const double* thr, int R, size_t n, double min_std, int side,
                                                   double* anomaly, double* mu_out, double* sigma_out, double* anomaly_refs) {
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < n; v += (size_t)gridDim.x * 256) {
        double s = 0.0;
        int cnt = 0;
        for (int r = 0; r < R; ++r) {
            const double x = refs[(size_t)r * n + v];
            const bool on = thr == nullptr || x > thr[r];
            s += on ? x : 0.0; cnt += on ? 1 : 0;
        }
        const double nan = __builtin_nan("");
        const double mu = cnt ? s / (double)cnt : nan;
        double q = 0.0;
        for (int r = 0; r < R; ++r) {
            const double x = refs[(size_t)r * n + v];
            const double d = x - mu;
            q += (thr == nullptr || x > thr[r]) ? d * d : 0.0;
        }
        double sigma = cnt ? sqrt(q / (double)cnt) : nan;
        if (min_std < 0) sigma += 1e-6; else sigma = sigma < min_std ? min_std : sigma;
        if (anomaly) anomaly[v] = nll_one(xp[v], mu, sigma, side);
        if (mu_out) mu_out[v] = mu;
        if (sigma_out) sigma_out[v] = sigma;
        if (anomaly_refs)
            for (int r = 0; r < R; ++r) anomaly_refs[(size_t)r * n + v] = nll_one(refs[(size_t)r * n + v], mu, sigma, side);
    }
}
extern "C" int wmh_nll(const double* x_prime, const double* refs, int R, size_t n, double min_std, int side, double* anomaly, double* mu,
                       double* sigma, double* anomaly_refs, void* stream) {
    if (!refs || R < 1 || (anomaly && !x_prime) || side < -1 || side > 1) return wmh_fail(WMH_EINVAL, "nll: bad args");
    if (n) hipLaunchKernelGGL(nll_kernel, dim3(s1::grid(n, 8192)), dim3(256), 0, (hipStream_t)stream, x_prime, refs, (const double*)nullptr, R, n, min_std,
                              side, anomaly, mu, sigma, anomaly_refs);
    return wmh_check_launch("nll_kernel");
}
// thresholds: R doubles on the device (each reference's Otsu threshold)
extern "C" int wmh_nll_masked(const double* x_prime, const double* refs, const double* thresholds, int R, size_t n, double min_std, int side,
                              double* anomaly, double* mu, double* sigma, void* stream) {
    if (!refs || !thresholds || R < 1 || (anomaly && !x_prime) || side < -1 || side > 1) return wmh_fail(WMH_EINVAL, "nll_masked: bad args");
    if (n) hipLaunchKernelGGL(nll_kernel, dim3(s1::grid(n, 8192)), dim3(256), 0, (hipStream_t)stream, x_prime, refs, thresholds, R, n, min_std, side,
                              anomaly, mu, sigma, (double*)nullptr);
    return wmh_check_launch("nll_kernel");
}

// scipy.ndimage.median_filter(size=(kz,ky,kx), mode='constant', cval=0): window offsets [-k/2, k - k/2 - 1] per axis,
// the element of rank N/2 of the N window values.  Rank selection by counting (N <= 27: the 3 mm kernels of 1 mm data).
template <int NMAX>
__global__ __launch_bounds__(256) void median_kernel(const double* in, double* out, int D, int H, int W, int kz, int ky, int kx) {
    const size_t n = (size_t)D * H * W;
    const int N = kz * ky * kx, rank = N / 2;
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < n; v += (size_t)gridDim.x * 256) {
        const int x = (int)(v % W), y = (int)((v / W) % H), z = (int)(v / ((size_t)W * H));
        double w[NMAX];
#pragma unroll
        for (int e = 0; e < NMAX; ++e) {
            double t = 0.0;
            if (e < N) {
                const int dx = e % kx - kx / 2, dy = (e / kx) % ky - ky / 2, dz = e / (kx * ky) - kz / 2;
                const int xx = x + dx, yy = y + dy, zz = z + dz;
                if (xx >= 0 && xx < W && yy >= 0 && yy < H && zz >= 0 && zz < D) t = in[((size_t)zz * H + yy) * W + xx];
            }
            w[e] = t;
        }
        double res = 0.0;
#pragma unroll
        for (int e = 0; e < NMAX; ++e) {
            if (e >= N) continue;
            int less = 0, eq = 0;
#pragma unroll
            for (int f = 0; f < NMAX; ++f) { if (f < N) { less += w[f] < w[e] ? 1 : 0; eq += w[f] == w[e] ? 1 : 0; } }
            if (less <= rank && rank < less + eq) res = w[e];
        }
        out[v] = res;
    }
}
// Windows of more than 27 voxels (voxels finer than 0.75 mm: 4 x 4 x 4 .. 8 x 8 x 8): the window in per-thread scratch memory,
// rank N/2 by quickselect (median-of-three pivot, three-way partition).  The rank-N/2 VALUE does not depend on the selection
// algorithm, so the result is the same as the counting form's.
constexpr int MEDIAN_BIG_MAX = 512;
__global__ __launch_bounds__(256) void median_big_kernel(const double* in, double* out, int D, int H, int W, int kz, int ky, int kx) {
    const size_t n = (size_t)D * H * W;
    const int N = kz * ky * kx, rank = N / 2;
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < n; v += (size_t)gridDim.x * 256) {
        const int x = (int)(v % W), y = (int)((v / W) % H), z = (int)(v / ((size_t)W * H));
        double w[MEDIAN_BIG_MAX];
        int e = 0;
        for (int dz = -(kz / 2); dz < kz - kz / 2; ++dz)
            for (int dy = -(ky / 2); dy < ky - ky / 2; ++dy)
                for (int dx = -(kx / 2); dx < kx - kx / 2; ++dx) {
                    const int xx = x + dx, yy = y + dy, zz = z + dz;
                    w[e++] = (xx >= 0 && xx < W && yy >= 0 && yy < H && zz >= 0 && zz < D) ? in[((size_t)zz * H + yy) * W + xx] : 0.0;
                }
        int lo = 0, hi = N - 1;
        double res = w[rank];
        while (lo < hi) {
            const double a = w[lo], b = w[(lo + hi) >> 1], c = w[hi];
            const double piv = a < b ? (b < c ? b : (a < c ? c : a)) : (a < c ? a : (b < c ? c : b));
            int lt = lo, i = lo, gt = hi;                          // [lo, lt) < piv, [lt, i) == piv, (gt, hi] > piv
            while (i <= gt) {
                const double t = w[i];
                if (t < piv) { w[i] = w[lt]; w[lt] = t; ++lt; ++i; }
                else if (t > piv) { w[i] = w[gt]; w[gt] = t; --gt; }
                else ++i;
            }
            if (rank < lt) hi = lt - 1;
            else if (rank > gt) lo = gt + 1;
            else { lo = hi = rank; w[rank] = piv; }
        }
        res = w[rank];
        out[v] = res;
    }
}
extern "C" int wmh_median_filter_f64(const double* in, double* out, int D, int H, int W, int kz, int ky, int kx, void* stream) {
    if (!in || !out || D < 1 || H < 1 || W < 1 || kz < 1 || ky < 1 || kx < 1) return wmh_fail(WMH_EINVAL, "median_filter_f64: bad args");
    const int N = kz * ky * kx;
    if (N > MEDIAN_BIG_MAX) return wmh_fail(WMH_EINVAL, "median_filter_f64: windows of more than 512 voxels are not implemented");
    const unsigned g = s1::grid((size_t)D * H * W, 1u << 16);
    if (N > 27) {
        hipLaunchKernelGGL(median_big_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, in, out, D, H, W, kz, ky, kx);
        return wmh_check_launch("median_big_kernel");
    }
    if (N <= 9) hipLaunchKernelGGL((median_kernel<9>), dim3(g), dim3(256), 0, (hipStream_t)stream, in, out, D, H, W, kz, ky, kx);
    else hipLaunchKernelGGL((median_kernel<27>), dim3(g), dim3(256), 0, (hipStream_t)stream, in, out, D, H, W, kz, ky, kx);
    return wmh_check_launch("median_kernel");
}

// component_filtering (image_ops.py:256-308): per 2-D slice of one orientation, binary erosion with the cross element
// (outside the slice counts as background) and then the largest 4-connected component (first in raster order on a
// tie); the three orientations are OR-ed.  `axis` is the slice axis; neighbours run along the other two.
__global__ __launch_bounds__(256) void erode2d_kernel(const uint8_t* m, uint8_t* out, int D, int H, int W, int axis) {
    const size_t n = (size_t)D * H * W; const int HW = H * W;
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < n; v += (size_t)gridDim.x * 256) {
        const int c[3] = {(int)(v / HW), (int)((v / W) % H), (int)(v % W)};
        const int dim[3] = {D, H, W}; const int str[3] = {HW, W, 1};
        bool keep = m[v] != 0;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            if (a == axis) continue;
            keep = keep && c[a] > 0 && c[a] < dim[a] - 1 && m[v - str[a]] && m[v + str[a]];
        }
        out[v] = keep ? 1 : 0;
    }
}
__device__ __forceinline__ int s1_find(const int* L, int a) {
    int p;
    while ((p = __hip_atomic_load(L + a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != a) a = p;
    return a;
}
__device__ __forceinline__ void s1_union(int* L, int a, int b) {
    for (;;) {
        a = s1_find(L, a); b = s1_find(L, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }
        const int old = atomicMin(L + a, b);
        if (old == a) return;
        a = old;
    }
}
__global__ __launch_bounds__(256) void cc2d_init_kernel(const uint8_t* m, int* L, size_t n) {
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < n; v += (size_t)gridDim.x * 256) L[v] = m[v] ? (int)v : -1;
}
__global__ __launch_bounds__(256) void cc2d_merge_kernel(int* L, int D, int H, int W, int axis) {
    const size_t n = (size_t)D * H * W; const int HW = H * W;
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < n; v += (size_t)gridDim.x * 256) {
        if (L[v] < 0) continue;
        const int c[3] = {(int)(v / HW), (int)((v / W) % H), (int)(v % W)};
        const int str[3] = {HW, W, 1};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            if (a == axis) continue;
            if (c[a] > 0 && L[v - str[a]] >= 0) s1_union(L, (int)v, (int)v - str[a]);
        }
    }
}
__global__ __launch_bounds__(256) void cc2d_compress_count_kernel(int* L, int* cnt, size_t n) {
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < n; v += (size_t)gridDim.x * 256)
        if (L[v] >= 0) { const int r = s1_find(L, (int)v); L[v] = r; atomicAdd(cnt + r, 1); }
}
// best[slice] = max over roots of (count << 32 | ~root): largest component, smallest root (first in raster order) on ties
__global__ __launch_bounds__(256) void cc2d_best_kernel(const int* L, const int* cnt, unsigned long long* best, int D, int H, int W, int axis) {
    const size_t n = (size_t)D * H * W; const int HW = H * W;
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < n; v += (size_t)gridDim.x * 256) {
        if (L[v] != (int)v) continue;                               // roots only
        const int c[3] = {(int)(v / HW), (int)((v / W) % H), (int)(v % W)};
        const unsigned long long key = ((unsigned long long)(unsigned)cnt[v] << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)v);
        atomicMax(best + c[axis], key);
    }
}
__global__ __launch_bounds__(256) void cc2d_pick_kernel(const int* L, const unsigned long long* best, uint8_t* acc, int D, int H, int W, int axis) {
    const size_t n = (size_t)D * H * W; const int HW = H * W;
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < n; v += (size_t)gridDim.x * 256) {
        if (L[v] < 0) continue;
        const int c[3] = {(int)(v / HW), (int)((v / W) % H), (int)(v % W)};
        const unsigned root = 0xFFFFFFFFu - (unsigned)(best[c[axis]] & 0xFFFFFFFFull);
        if ((unsigned)L[v] == root) acc[v] = 1;
    }
}
__global__ __launch_bounds__(256) void or_kernel(uint8_t* acc, const uint8_t* m, size_t n) {
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < n; v += (size_t)gridDim.x * 256) if (m[v]) acc[v] = 1;
}
// out = OR over the three orientations; do_filter3[a] = 0 passes the mask through for that orientation.
// scratch_u8: n bytes; labels, counts: n int32 each; best: max(D,H,W) uint64.
extern "C" int wmh_component_filtering(const uint8_t* mask, uint8_t* out, uint8_t* scratch_u8, int32_t* labels, int32_t* counts,
                                       unsigned long long* best, int D, int H, int W, const int* do_filter3, void* stream) {
    if (!mask || !out || !scratch_u8 || !labels || !counts || !best || !do_filter3 || D < 1 || H < 1 || W < 1 || (size_t)D * H * W >= ((size_t)1 << 31))
        return wmh_fail(WMH_EINVAL, "component_filtering: bad args");
    hipStream_t st = (hipStream_t)stream;
    const size_t n = (size_t)D * H * W;
    const unsigned g = s1::grid(n, 1u << 16);
    const int dim[3] = {D, H, W};
    if (hipMemsetAsync(out, 0, n, st) != hipSuccess) return wmh_fail(WMH_ELAUNCH, "component_filtering: memset");
    for (int a = 0; a < 3; ++a) {
        if (!do_filter3[a]) { hipLaunchKernelGGL(or_kernel, dim3(g), dim3(256), 0, st, out, mask, n); continue; }
        hipLaunchKernelGGL(erode2d_kernel, dim3(g), dim3(256), 0, st, mask, scratch_u8, D, H, W, a);
        hipLaunchKernelGGL(cc2d_init_kernel, dim3(g), dim3(256), 0, st, scratch_u8, labels, n);
        hipLaunchKernelGGL(cc2d_merge_kernel, dim3(g), dim3(256), 0, st, labels, D, H, W, a);
        if (hipMemsetAsync(counts, 0, n * sizeof(int), st) != hipSuccess || hipMemsetAsync(best, 0, (size_t)dim[a] * 8, st) != hipSuccess)
            return wmh_fail(WMH_ELAUNCH, "component_filtering: memset");
        hipLaunchKernelGGL(cc2d_compress_count_kernel, dim3(g), dim3(256), 0, st, labels, counts, n);
        hipLaunchKernelGGL(cc2d_best_kernel, dim3(g), dim3(256), 0, st, labels, counts, best, D, H, W, a);
        hipLaunchKernelGGL(cc2d_pick_kernel, dim3(g), dim3(256), 0, st, labels, best, out, D, H, W, a);
    }
    return wmh_check_launch("component_filtering kernels");
}
