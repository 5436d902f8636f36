// Inner-loop study for the weight-gradient kernel: cycles per MFMA of D[co][tap] += dy[co][voxel pair] * x'[voxel pair][tap]
// out of LDS, for several operand-fetch structures (no global traffic: the tiles are filled once).
//   MODE 0  one accumulator chain, operands read where they are used (what the compiler does by itself)
//   MODE 1  one chain, operands of row r+1 fetched before the MFMAs of row r
//   MODE 2  two input channels at once: one A (dy) read feeds two MFMAs on independent accumulators
//   MODE 3  MODE 2 + row prefetch
//   MODE 4  registers only, one dependent chain (issue rate of back-to-back dependent MFMAs)
//   MODE 5  registers only, 8 independent accumulators
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int TX = 32, TY = 4, TZ = 2, IX = 34, IY = 6, IZ = 4, ZS = IX * IY, SPX = ZS * IZ, TV = 256, DYS = 257, XB = 832;

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(unsigned long long* out, float* sink, int iters) {
    extern __shared__ float lds[];
    float* dyl = lds;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* xw = lds + 32 * DYS + wave * 2 * XB;
    const int i = lane & 31, kh = lane >> 5;
    for (int e = tid; e < 32 * DYS + 8 * XB; e += 256) lds[e] = (float)(e % 13) * 0.01f;
    __syncthreads();
    const int tapj = min(i, 26);
    const int LB = kh + (tapj / 9) * ZS + ((tapj / 3) % 3) * IX + (tapj % 3);
    const int LA = i * DYS + kh;
    f32x16 acc[8];
    for (int m = 0; m < 8; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    const float* a0 = dyl + LA;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        asm volatile("" ::: "memory");      // LDS contents are opaque per iteration (nothing hoisted)
        if (MODE == 0) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float* b0 = xw + (c & 1) * XB + LB;
#pragma unroll
                for (int r = 0; r < TZ * TY; ++r) {
                    const float* ap = a0 + r * TX; const float* bp = b0 + (r / TY) * ZS + (r % TY) * IX;
#pragma unroll
                    for (int xp = 0; xp < 16; ++xp) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * xp], bp[2 * xp], acc[c], 0, 0, 0);
                }
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float* b0 = xw + (c & 1) * XB + LB;
                float av[2][16], bv[2][16];
#pragma unroll
                for (int xp = 0; xp < 16; ++xp) { av[0][xp] = a0[2 * xp]; bv[0][xp] = b0[2 * xp]; }
#pragma unroll
                for (int r = 0; r < TZ * TY; ++r) {
                    const int rn = (r + 1) % (TZ * TY);
                    const float* ap = a0 + rn * TX; const float* bp = b0 + (rn / TY) * ZS + (rn % TY) * IX;
#pragma unroll
                    for (int xp = 0; xp < 16; ++xp) { av[(r + 1) & 1][xp] = ap[2 * xp]; bv[(r + 1) & 1][xp] = bp[2 * xp]; }
                    __builtin_amdgcn_sched_barrier(0x0F);      // DS reads stay above
#pragma unroll
                    for (int xp = 0; xp < 16; ++xp) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[r & 1][xp], bv[r & 1][xp], acc[c], 0, 0, 0);
                }
            }
        } else if (MODE == 2) {
#pragma unroll
            for (int c = 0; c < 8; c += 2) {
                const float* b0 = xw + LB; const float* b1 = xw + XB + LB;
#pragma unroll
                for (int r = 0; r < TZ * TY; ++r) {
                    const float* ap = a0 + r * TX; const int bo = (r / TY) * ZS + (r % TY) * IX;
#pragma unroll
                    for (int xp = 0; xp < 16; ++xp) {
                        const float av = ap[2 * xp];
                        acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0[bo + 2 * xp], acc[c], 0, 0, 0);
                        acc[c + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1[bo + 2 * xp], acc[c + 1], 0, 0, 0);
                    }
                }
            }
        } else if (MODE == 3) {
#pragma unroll
            for (int c = 0; c < 8; c += 2) {
                const float* b0 = xw + LB; const float* b1 = xw + XB + LB;
                float av[2][16], bv[2][16], cv[2][16];
#pragma unroll
                for (int xp = 0; xp < 16; ++xp) { av[0][xp] = a0[2 * xp]; bv[0][xp] = b0[2 * xp]; cv[0][xp] = b1[2 * xp]; }
#pragma unroll
                for (int r = 0; r < TZ * TY; ++r) {
                    const int rn = (r + 1) % (TZ * TY);
                    const float* ap = a0 + rn * TX; const int bo = (rn / TY) * ZS + (rn % TY) * IX;
#pragma unroll
                    for (int xp = 0; xp < 16; ++xp) { av[(r + 1) & 1][xp] = ap[2 * xp]; bv[(r + 1) & 1][xp] = b0[bo + 2 * xp]; cv[(r + 1) & 1][xp] = b1[bo + 2 * xp]; }
                    __builtin_amdgcn_sched_barrier(0x0F);
#pragma unroll
                    for (int xp = 0; xp < 16; ++xp) {
                        acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[r & 1][xp], bv[r & 1][xp], acc[c], 0, 0, 0);
                        acc[c + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[r & 1][xp], cv[r & 1][xp], acc[c + 1], 0, 0, 0);
                    }
                }
            }
        } else if (MODE == 4) {
            float a = tid * 0.001f, b = tid * 0.002f;
#pragma unroll
            for (int m = 0; m < 64; ++m) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
        } else {
            float a = tid * 0.001f, b = tid * 0.002f;
#pragma unroll
            for (int m = 0; m < 64; ++m) acc[m & 7] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m & 7], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int m = 0; m < 8; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
    sink[blockIdx.x * 256 + tid] = s;
    if (tid == 0) out[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(int nblk, unsigned long long* out, float* sink) {
    const int lds = (32 * DYS + 8 * XB) * 4, iters = 200;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k<MODE>, dim3(nblk), dim3(256), lds, 0, out, sink, iters); hipDeviceSynchronize(); }
    static unsigned long long h[2048];
    hipMemcpy(h, out, nblk * 8, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < nblk; ++i) s += h[i];
    const double nm = (MODE >= 4) ? 64.0 * iters : 1024.0 * iters;
    // s_memtime ticks at 100 MHz; convert to core clocks at 2.385 GHz (s_memtime in fact counts shader cycles: DESIGN.md 4)
    printf("mode %d  blocks %4d  %.1f clk per MFMA per wave (x%d waves/SIMD -> %.1f clk per MFMA per SIMD)\n", MODE, nblk,
           s / nblk / nm * 23.85, nblk / 256, s / nblk / nm * 23.85 / (nblk / 256));
}
int main() {
    unsigned long long* out; float* sink;
    hipMalloc(&out, 2048 * 8); hipMalloc(&sink, 2048 * 256 * 4);
    for (int nblk : {256, 512}) {
        run<0>(nblk, out, sink); run<1>(nblk, out, sink); run<2>(nblk, out, sink); run<3>(nblk, out, sink); run<4>(nblk, out, sink); run<5>(nblk, out, sink);
    }
    return 0;
}
