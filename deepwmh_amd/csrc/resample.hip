// resample.hip -- spline resampling of volumes between voxel grids, as nnU-Net's preprocessing / export do it through
// skimage.transform.resize(order, mode='edge', anti_aliasing=False) [nnU-Net, not in tree], restated in
// deepwmh_amd/preprocess.py::_resize as scipy.ndimage.zoom(order, mode='nearest', grid_mode=True).
//   order 3: pad 12 voxels of edge values, cubic B-spline prefilter (pole sqrt(3)-2, mirror initialisation) along the
//            three axes, then a 64-tap gather per output voxel at u = (o + 0.5) * in/out - 0.5;
//   order 1: 8-tap linear gather with clamped indices;   order 0: nearest.
// All arithmetic in double, taps accumulated in scipy's order (z, y, x; coefficient * wz * wy * wx), so the result
// agrees with the float64 host computation to rounding.  HBM/L2-bound; one thread per line (prefilter) or per output voxel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "wmh_common.h"

namespace rs { constexpr int PAD = 12; }

__global__ __launch_bounds__(256) void pad_edge_kernel(const float* in, double* out, int D, int H, int W) {
    const int P = rs::PAD, Dp = D + 2 * P, Hp = H + 2 * P, Wp = W + 2 * P;
    const size_t n = (size_t)Dp * Hp * Wp;
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < n; v += (size_t)gridDim.x * 256) {
        const int x = (int)(v % Wp), y = (int)((v / Wp) % Hp), z = (int)(v / ((size_t)Wp * Hp));
        const int sx = min(max(x - P, 0), W - 1), sy = min(max(y - P, 0), H - 1), sz = min(max(z - P, 0), D - 1);
        out[v] = (double)in[((size_t)sz * H + sy) * W + sx];
    }
}
// in-place cubic B-spline prefilter of every line along `axis` (0 = z, 1 = y, 2 = x) of a [Dp][Hp][Wp] double volume
__global__ __launch_bounds__(256) void spline3_prefilter_kernel(double* c, int Dp, int Hp, int Wp, int axis) {
    const int n = axis == 0 ? Dp : axis == 1 ? Hp : Wp;
    const size_t stride = axis == 0 ? (size_t)Hp * Wp : axis == 1 ? (size_t)Wp : 1;
    const int nlines = axis == 0 ? Hp * Wp : axis == 1 ? Dp * Wp : Dp * Hp;
    const int l = blockIdx.x * 256 + threadIdx.x;
    if (l >= nlines) return;
    size_t base;
    if (axis == 0) base = (size_t)l;                                                    // (y, x)
    else if (axis == 1) base = (size_t)(l / Wp) * Hp * Wp + (size_t)(l % Wp);           // (z, x)
    else base = (size_t)l * Wp;                                                         // (z, y)
    double* p = c + base;
    if (n < 2) return;                                                                  // scipy leaves length-1 lines alone
    const double z = sqrt(3.0) - 2.0, gain = (1.0 - z) * (1.0 - 1.0 / z);
    for (int i = 0; i < n; ++i) p[i * stride] *= gain;
    // causal initialisation, mirror boundary (terms beyond ~60 samples are below double rounding but cost nothing to keep exact)
    const double zn1 = pow(z, (double)(n - 1));
    double zi = z, c0 = p[0] + zn1 * p[(size_t)(n - 1) * stride];
    for (int i = 1; i < n - 1; ++i) { c0 += zi * (p[i * stride] + zn1 * p[(size_t)(n - 1 - i) * stride]); zi *= z; }
    c0 /= 1.0 - zn1 * zn1;
    p[0] = c0;
    double prev = c0;
    for (int i = 1; i < n; ++i) { prev = p[i * stride] + z * prev; p[i * stride] = prev; }
    double next = (z / (z * z - 1.0)) * (z * p[(size_t)(n - 2) * stride] + p[(size_t)(n - 1) * stride]);
    p[(size_t)(n - 1) * stride] = next;
    for (int i = n - 2; i >= 0; --i) { next = z * (next - p[i * stride]); p[i * stride] = next; }
}
__device__ __forceinline__ void spline3_weights(double u, int& start, double w[4]) {
    const double f = floor(u);
    const double y = u - f, zc = 1.0 - y;
    w[1] = (y * y * (y - 2.0) * 3.0 + 4.0) / 6.0;
    w[2] = (zc * zc * (zc - 2.0) * 3.0 + 4.0) / 6.0;
    w[0] = zc * zc * zc / 6.0;
    w[3] = 1.0 - w[0] - w[1] - w[2];
    start = (int)f - 1;
}
#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void zoom3_kernel(const double* c, float* out, int D, int H, int W, int Do, int Ho, int Wo) {
    const int P = rs::PAD, Dp = D + 2 * P, Hp = H + 2 * P, Wp = W + 2 * P;
    const size_t n = (size_t)Do * Ho * Wo;
    const double sz = (double)D / (double)Do, sy = (double)H / (double)Ho, sx = (double)W / (double)Wo;
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < n; v += (size_t)gridDim.x * 256) {
        const int ox = (int)(v % Wo), oy = (int)((v / Wo) % Ho), oz = (int)(v / ((size_t)Wo * Ho));
        int z0, y0, x0; double wz[4], wy[4], wx[4];
        spline3_weights((oz + 0.5) * sz - 0.5 + P, z0, wz);
        spline3_weights((oy + 0.5) * sy - 0.5 + P, y0, wy);
        spline3_weights((ox + 0.5) * sx - 0.5 + P, x0, wx);
        double t = 0.0;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int zz = min(max(z0 + a, 0), Dp - 1);
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int yy = min(max(y0 + b, 0), Hp - 1);
                const double* row = c + ((size_t)zz * Hp + yy) * Wp;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const int xx = min(max(x0 + d, 0), Wp - 1);
                    double coeff = row[xx];
                    coeff *= wz[a]; coeff *= wy[b]; coeff *= wx[d];
                    t += coeff;
                }
            }
        }
        out[v] = (float)t;
    }
}
__global__ __launch_bounds__(256) void zoom1_kernel(const float* in, float* out, int D, int H, int W, int Do, int Ho, int Wo) {
    const size_t n = (size_t)Do * Ho * Wo;
    const double sz = (double)D / (double)Do, sy = (double)H / (double)Ho, sx = (double)W / (double)Wo;
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < n; v += (size_t)gridDim.x * 256) {
        const int ox = (int)(v % Wo), oy = (int)((v / Wo) % Ho), oz = (int)(v / ((size_t)Wo * Ho));
        const double uz = (oz + 0.5) * sz - 0.5, uy = (oy + 0.5) * sy - 0.5, ux = (ox + 0.5) * sx - 0.5;
        const double fz = floor(uz), fy = floor(uy), fx = floor(ux);
        const double wz[2] = {1.0 - (uz - fz), uz - fz}, wy[2] = {1.0 - (uy - fy), uy - fy}, wx[2] = {1.0 - (ux - fx), ux - fx};
        double t = 0.0;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int zz = min(max((int)fz + a, 0), D - 1);
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int yy = min(max((int)fy + b, 0), H - 1);
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const int xx = min(max((int)fx + d, 0), W - 1);
                    double coeff = (double)in[((size_t)zz * H + yy) * W + xx];
                    coeff *= wz[a]; coeff *= wy[b]; coeff *= wx[d];
                    t += coeff;
                }
            }
        }
        out[v] = (float)t;
    }
}
template <typename T>
__global__ __launch_bounds__(256) void zoom0_kernel(const T* in, T* out, int D, int H, int W, int Do, int Ho, int Wo) {
    const size_t n = (size_t)Do * Ho * Wo;
    const double sz = (double)D / (double)Do, sy = (double)H / (double)Ho, sx = (double)W / (double)Wo;
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < n; v += (size_t)gridDim.x * 256) {
        const int ox = (int)(v % Wo), oy = (int)((v / Wo) % Ho), oz = (int)(v / ((size_t)Wo * Ho));
        const int zz = min(max((int)floor((oz + 0.5) * sz - 0.5 + 0.5), 0), D - 1);
        const int yy = min(max((int)floor((oy + 0.5) * sy - 0.5 + 0.5), 0), H - 1);
        const int xx = min(max((int)floor((ox + 0.5) * sx - 0.5 + 0.5), 0), W - 1);
        out[v] = in[((size_t)zz * H + yy) * W + xx];
    }
}

static inline unsigned rgrid(size_t n) { size_t b = (n + 255) / 256; if (b < 1) b = 1; if (b > 1048576) b = 1048576; return (unsigned)b; }

extern "C" size_t wmh_resample_workspace_doubles(int D, int H, int W, int order) {
    if (order != 3) return 0;
    return (size_t)(D + 2 * rs::PAD) * (H + 2 * rs::PAD) * (W + 2 * rs::PAD);
}
extern "C" int wmh_resample(const float* in, float* out, double* workspace, int C, int D, int H, int W, int Do, int Ho, int Wo, int order,
                            void* stream) {
    if (!in || !out || C < 1 || D < 1 || H < 1 || W < 1 || Do < 1 || Ho < 1 || Wo < 1 || (order != 0 && order != 1 && order != 3))
        return wmh_fail(WMH_EINVAL, "resample: bad args (orders 0, 1 and 3 are implemented)");
    if (order == 3 && !workspace) return wmh_fail(WMH_EINVAL, "resample: order 3 needs a workspace");
    hipStream_t st = (hipStream_t)stream;
    const size_t nin = (size_t)D * H * W, nout = (size_t)Do * Ho * Wo;
    const int Dp = D + 2 * rs::PAD, Hp = H + 2 * rs::PAD, Wp = W + 2 * rs::PAD;
    for (int c = 0; c < C; ++c) {
        const float* src = in + (size_t)c * nin; float* dst = out + (size_t)c * nout;
        if (order == 3) {
            hipLaunchKernelGGL(pad_edge_kernel, dim3(rgrid((size_t)Dp * Hp * Wp)), dim3(256), 0, st, src, workspace, D, H, W);
            hipLaunchKernelGGL(spline3_prefilter_kernel, dim3((Hp * Wp + 255) / 256), dim3(256), 0, st, workspace, Dp, Hp, Wp, 0);
            hipLaunchKernelGGL(spline3_prefilter_kernel, dim3((Dp * Wp + 255) / 256), dim3(256), 0, st, workspace, Dp, Hp, Wp, 1);
            hipLaunchKernelGGL(spline3_prefilter_kernel, dim3((Dp * Hp + 255) / 256), dim3(256), 0, st, workspace, Dp, Hp, Wp, 2);
            hipLaunchKernelGGL(zoom3_kernel, dim3(rgrid(nout)), dim3(256), 0, st, workspace, dst, D, H, W, Do, Ho, Wo);
        } else if (order == 1) {
            hipLaunchKernelGGL(zoom1_kernel, dim3(rgrid(nout)), dim3(256), 0, st, src, dst, D, H, W, Do, Ho, Wo);
        } else {
            hipLaunchKernelGGL((zoom0_kernel<float>), dim3(rgrid(nout)), dim3(256), 0, st, src, dst, D, H, W, Do, Ho, Wo);
        }
    }
    return wmh_check_launch("resample kernels");
}
extern "C" int wmh_resample_nearest_u8(const uint8_t* in, uint8_t* out, int D, int H, int W, int Do, int Ho, int Wo, void* stream) {
    if (!in || !out || D < 1 || H < 1 || W < 1 || Do < 1 || Ho < 1 || Wo < 1) return wmh_fail(WMH_EINVAL, "resample_nearest_u8: bad args");
    hipLaunchKernelGGL((zoom0_kernel<uint8_t>), dim3(rgrid((size_t)Do * Ho * Wo)), dim3(256), 0, (hipStream_t)stream, in, out, D, H, W, Do, Ho, Wo);
    return wmh_check_launch("zoom0_kernel");
}
