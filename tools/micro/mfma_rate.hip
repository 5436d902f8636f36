// cycles per MFMA for the matrix instructions this project uses or considers (one wave per SIMD, 8 accumulators)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s4 __attribute__((ext_vector_type(4)));
typedef short s8 __attribute__((ext_vector_type(8)));
template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned long long* out, float* sink, int iters) {
    f32x16 acc[8];
    for (int m = 0; m < 8; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f;
    s4 a4 = {1, 2, 3, (short)threadIdx.x}, b4 = {4, 5, 6, (short)threadIdx.x};
    s8 a8 = {1, 2, 3, 4, 5, 6, 7, (short)threadIdx.x}, b8 = {4, 5, 6, 7, 8, 9, 1, (short)threadIdx.x};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (MODE == 0) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m], 0, 0, 0);
            if (MODE == 1) acc[m] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a4, b4, acc[m], 0, 0, 0);
            if (MODE == 2) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc[m], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int m = 0; m < 8; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}
int main() {
    unsigned long long* out; float* sink;
    hipMalloc(&out, 1024 * 8); hipMalloc(&sink, 1024 * 256 * 4);
    unsigned long long h[1024];
    const int iters = 2000;
    const char* names[3] = {"v_mfma_f32_32x32x2_f32", "v_mfma_f32_32x32x8_bf16", "v_mfma_f32_32x32x16_bf16"};
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, out, sink, iters);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, out, sink, iters);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, out, sink, iters);
            hipDeviceSynchronize();
        }
        hipMemcpy(h, out, 256 * 8, hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < 256; ++i) s += h[i];
        printf("%-28s %.1f cycles per MFMA (one wave per SIMD)\n", names[mode], s / 256 / (iters * 8.0));
    }
    return 0;
}
