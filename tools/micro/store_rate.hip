// Store-path throughput per CU for the epilogue's access patterns (gfx950): 256 workgroups x 256 threads, every wave writes its own
// region of a large buffer, (a) global_store_dword, one wave-instruction = two 128-byte row segments of two channels (the NCDHW
// fp32 epilogue), (b) global_store_dwordx4, one wave-instruction = 16 row pieces of 64 bytes... , (c) dwordx4 with each group of 8
// lanes covering one 128-byte line.  Prints bytes per clock per CU.
//   hipcc --offload-arch=gfx950 -O3 -o store_rate store_rate.hip && ./store_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* ticks, int iters, size_t plane) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, kh = lane >> 5;
    // region of this wave: `iters` rows of 32 voxels x 32 channel planes, row r at out[base + c * plane + r * 32]
    float* base = out + ((size_t)blockIdx.x * 4 + wave) * (size_t)iters * 32;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < iters; ++r) {
        if (MODE == 0) {          // 16 dword stores: register q -> channel (q&3) + 8(q>>2) + 4 kh, lane -> voxel j
#pragma unroll
            for (int q = 0; q < 16; ++q) base[(size_t)((q & 3) + 8 * (q >> 2) + 4 * kh) * plane + r * 32 + j] = (float)(q + lane);
        } else if (MODE == 1) {   // 4 dwordx4 stores: lane -> (channel lane&3 [+4 kh], voxels 4 (j>>2) ..), register group g -> channel + 8 g
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f4 v = {(float)g, (float)lane, 1.f, 2.f};
                *reinterpret_cast<f4*>(base + (size_t)((j & 3) + 8 * g + 4 * kh) * plane + r * 32 + 4 * (j >> 2)) = v;
            }
        } else if (MODE == 3) {   // transposed accumulator (lane -> channel j, registers -> voxels): 4 dwordx4 stores, lane -> (channel j, voxels 8 g + 4 kh ..):
                                  // 32 channel planes x 32 contiguous bytes per instruction
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f4 v = {(float)g, (float)lane, 1.f, 2.f};
                *reinterpret_cast<f4*>(base + (size_t)j * plane + r * 32 + 8 * g + 4 * kh) = v;
            }
        } else {                  // 4 dwordx4 stores: lane -> (channel (j>>3) [+4 kh], voxels 4 (j&7) ..): 8 consecutive lanes = one 128-byte line
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f4 v = {(float)g, (float)lane, 1.f, 2.f};
                *reinterpret_cast<f4*>(base + (size_t)((j >> 3) + 8 * g + 4 * kh) * plane + r * 32 + 4 * (j & 7)) = v;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) ticks[blockIdx.x * 4 + wave] = t1 - t0;
}

int main(int argc, char** argv) {
    const int iters = 512;
    const int grid = argc > 1 ? atoi(argv[1]) : 256;      // fewer workgroups than CUs: the per-CU store path instead of the HBM write rate
    const size_t plane = (size_t)256 * 4 * iters * 32;          // floats per channel plane
    float* out; unsigned long long* ticks;
    hipMalloc(&out, plane * 32 * sizeof(float)); hipMalloc(&ticks, 1024 * 8);
    const char* names[4] = {"global_store_dword  (2 x 128 B segments per instruction)", "global_store_dwordx4 (lanes of a quad on 4 channels)",
                            "global_store_dwordx4 (8 lanes = one 128 B line)", "global_store_dwordx4 (lane = channel: 32 planes x 32 B)"};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 4; ++mode) {
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, out, ticks, iters, plane);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, out, ticks, iters, plane);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, out, ticks, iters, plane);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, out, ticks, iters, plane);
            hipEventRecord(e1); hipDeviceSynchronize();
            hipEventElapsedTime(&ms, e0, e1);
        }
        const double bytes = (double)grid * 4 * iters * 32 * 32 * 4;
        printf("%-62s %.3f ms  %.0f GB/s  = %.1f B per 2.4 GHz clock per CU\n", names[mode], ms, bytes / ms / 1e6, bytes / (ms * 1e-3) / 2.4e9 / grid);
    }
    return 0;
}
