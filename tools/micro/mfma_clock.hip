// What clock does the chip hold under a pure matrix-instruction stream?  One workgroup of 4 (or 8) waves per CU issues independent
// v_mfma_f32_32x32x16_f16 (mode 0), v_mfma_f32_32x32x16_bf16 (1) or v_mfma_f32_32x32x2_f32 (2) back to back for a few seconds, operands
// random fp16 / bf16 / fp32 values held in registers (no memory traffic at all).  Prints the achieved rate; run it under
// tools/clock_probe.sh to see the shader clock and socket power it is achieved at.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_clock mfma_clock.hip && ./mfma_clock <mode> <waves per CU: 4|8> [seconds]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters, unsigned seed) {
    unsigned h = (blockIdx.x * 512 + threadIdx.x) * 2654435761u + seed;
    auto rnd = [&]() { h = h * 1664525u + 1013904223u; return h; };
    f16x8 a, b; s16x8 ab, bb; float af = 0, bf = 0;
    for (int e = 0; e < 8; ++e) {
        a[e] = (_Float16)(((int)(rnd() >> 20) - 2048) * (1.f / 1024)); b[e] = (_Float16)(((int)(rnd() >> 20) - 2048) * (1.f / 1024));
        ab[e] = (short)(0x3f00 + (rnd() >> 24)); bb[e] = (short)(0xbf00 + (rnd() >> 24));
    }
    af = ((int)(rnd() >> 20) - 2048) * (1.f / 1024); bf = ((int)(rnd() >> 20) - 2048) * (1.f / 1024);
    f32x16 acc[4];
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (MODE == 0) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[c], 0, 0, 0);
                else if (MODE == 1) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[c], 0, 0, 0);
                else acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc[c], 0, 0, 0);
            }
    }
    float s = 0.f;
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0, waves = argc > 2 ? atoi(argv[2]) : 4;
    const double secs = argc > 3 ? atof(argv[3]) : 3.0;
    float* out; hipMalloc(&out, 256 * 512 * 4);
    const double flop_per_mfma = mode == 2 ? 2.0 * 32 * 32 * 2 : 2.0 * 32 * 32 * 16;
    const double cyc = mode == 2 ? 64 : 32;                                  // cycles per instruction on one SIMD's matrix pipe
    int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    double total_ms = 0, last = 0;
    for (int rep = 0; total_ms < secs * 1e3; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(waves * 64), 0, 0, out, iters, rep);
        else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(waves * 64), 0, 0, out, iters, rep);
        else hipLaunchKernelGGL(k<2>, dim3(256), dim3(waves * 64), 0, 0, out, iters, rep);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        total_ms += ms; last = ms;
    }
    const double n = (double)iters * 32 * 256 * waves;                       // instructions per launch
    printf("mode %d (%s), %d waves per CU: %.1f TFLOP/s over the last launch = the matrix pipes' rate at %.0f MHz\n", mode,
           mode == 0 ? "v_mfma_f32_32x32x16_f16" : mode == 1 ? "v_mfma_f32_32x32x16_bf16" : "v_mfma_f32_32x32x2_f32", waves,
           n * flop_per_mfma / (last * 1e-3) / 1e12, n * cyc / 1024.0 / (last * 1e-3) / 1e6);
    return 0;
}
