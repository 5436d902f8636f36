// Inner loop of a pair-form (f16x2) weight-gradient kernel, alone: what the LDS delivers when the 27 taps sit on the lanes.
//   D[co 32][tap 32] += A[co][8 voxels x (hi, hi)] * B[8 voxels x (hi, lo)][tap]      two MFMAs (A from dy_hi, from dy_lo) per B read
// LDS image: dy [32 co][256 voxels] and x' [4 channels per wave][24 rows][RS bytes], one dword (fp16 hi | fp16 lo << 16) per voxel.
// A lane's B operand is 4 consecutive voxels starting at (its tap's kx) -- a 16-byte read at 4-byte alignment.
// MODE 0: kx forced to 0 (aligned ds_read_b128: the reference rate), 1: two ds_read2_b32 (what the compiler emits for align 4),
// 2: ds_read_b128 at 4-byte alignment (inline asm).  Prints cycles per MFMA (32 = the matrix pipe's rate) and a checksum.
//   hipcc --offload-arch=gfx950 -O3 -o wgrad_pair_inner wgrad_pair_inner.hip && ./wgrad_pair_inner
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) U4A { unsigned a, b, c, d; };

template <int MODE, int RS, int NCH, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(float* out, unsigned long long* ticks, int ntiles) {
    constexpr int TX = 32, TY = 4, TZ = 2, IY = TY + 2, R = IY * (TZ + 2), CS = R * RS + 16, DYSB = TX * TY * TZ * 4 + 16;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    char* dyl = lds;
    char* xw = lds + 32 * DYSB + wave * NCH * CS;
    for (int i = tid; i < (32 * DYSB + WAVES * NCH * CS) / 4; i += WAVES * 64) {
        const _Float16 h = (_Float16)((i * 37 % 101) * 0.01f - 0.5f), l = (_Float16)((i * 11 % 13) * 1e-4f);
        ((unsigned*)lds)[i] = (unsigned)__builtin_bit_cast(unsigned short, h) | ((unsigned)__builtin_bit_cast(unsigned short, l) << 16);
    }
    __syncthreads();
    const int i = lane & 31, kh = lane >> 5, tapj = min(i, 26);
    const int kx = MODE == 0 ? 0 : tapj % 3;
    const int LB = ((tapj / 9) * IY + (tapj / 3) % 3) * RS + (kx + 4 * kh) * 4;
    const int LA = i * DYSB + kh * 16;
    f32x16 acc[NCH];
    for (int c = 0; c < NCH; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < ntiles; ++t) {
        asm volatile("" ::: "memory");            // the real kernel's LDS changes from tile to tile: no hoisting
#pragma unroll
        for (int z = 0; z < TZ; ++z)
#pragma unroll
            for (int y = 0; y < TY; ++y)
#pragma unroll
                for (int s = 0; s < TX / 8; ++s) {
                    const u32x4 a4 = *reinterpret_cast<const u32x4*>(dyl + LA + ((z * TY + y) * TX + 8 * s) * 4);
                    u32x4 ah, al;
#pragma unroll
                    for (int d = 0; d < 4; ++d) { ah[d] = __builtin_amdgcn_perm(a4[d], a4[d], 0x01000100u); al[d] = __builtin_amdgcn_perm(a4[d], a4[d], 0x03020302u); }
                    u32x4 b[NCH];
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        const char* p = xw + c * CS + LB + (z * IY + y) * RS + 8 * s * 4;
                        if (MODE == 0) b[c] = *reinterpret_cast<const u32x4*>(p);
                        else if (MODE == 1) { const U4A v = *reinterpret_cast<const U4A*>(p); b[c] = u32x4{v.a, v.b, v.c, v.d}; }
                        else asm volatile("ds_read_b128 %0, %1" : "=v"(b[c]) : "v"((unsigned)(size_t)p) : "memory");
                    }
                    if (MODE == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int c = 0; c < NCH; ++c)
                        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, b[c]), acc[c], 0, 0, 0);
#pragma unroll
                    for (int c = 0; c < NCH; ++c)
                        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, al), __builtin_bit_cast(f16x8, b[c]), acc[c], 0, 0, 0);
                }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int c = 0; c < NCH; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * WAVES * 64 + tid] = s;
    if (lane == 0) ticks[blockIdx.x * WAVES + wave] = t1 - t0;
}

template <int MODE, int RS, int NCH, int WAVES>
static void run(const char* name, float* out, unsigned long long* ticks) {
    constexpr int CS = 24 * RS + 16, lds = 32 * (256 * 4 + 16) + WAVES * NCH * CS;
    const int ntiles = 64, grid = 256;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE, RS, NCH, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, RS, NCH, WAVES>), dim3(grid), dim3(WAVES * 64), lds, 0, out, ticks, ntiles);
        hipEventRecord(e1); hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
    }
    static float h[256 * 512]; static unsigned long long ht[256 * 8];
    hipMemcpy(h, out, sizeof(float) * grid * WAVES * 64, hipMemcpyDeviceToHost);
    hipMemcpy(ht, ticks, 8 * grid * WAVES, hipMemcpyDeviceToHost);
    double cs = 0; for (int i = 0; i < grid * WAVES * 64; ++i) cs += h[i];
    double tk = 0; for (int i = 0; i < grid * WAVES; ++i) tk += (double)ht[i];
    const double nm = (double)ntiles * 32 * 2 * NCH;
    printf("%-46s RS %3d ch/wave %d waves %d lds %6d B: %.3f ms, %.1f ticks per MFMA (wall: %.1f cycles at 2.4 GHz), checksum %.6e%s\n", name, RS, NCH, WAVES, lds,
           ms, tk / (grid * WAVES) / nm, ms * 1e-3 * 2.4e9 / nm, cs, hipGetLastError() == hipSuccess ? "" : "  ERROR");
}

int main() {
    float* out; unsigned long long* ticks;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&ticks, 256 * 8 * 8);
    run<0, 144, 4, 4>("aligned ds_read_b128 (kx = 0)", out, ticks);
    run<1, 136, 4, 4>("2 x ds_read2_b32", out, ticks);
    run<1, 144, 4, 4>("2 x ds_read2_b32", out, ticks);
    run<1, 152, 4, 4>("2 x ds_read2_b32", out, ticks);
    run<1, 160, 4, 4>("2 x ds_read2_b32", out, ticks);
    run<2, 136, 4, 4>("ds_read_b128 at 4-byte alignment", out, ticks);
    run<2, 144, 4, 4>("ds_read_b128 at 4-byte alignment", out, ticks);
    run<2, 152, 4, 4>("ds_read_b128 at 4-byte alignment", out, ticks);
    run<1, 144, 2, 4>("2 x ds_read2_b32", out, ticks);
    run<1, 144, 1, 4>("2 x ds_read2_b32", out, ticks);
    run<1, 144, 2, 8>("2 x ds_read2_b32", out, ticks);
    run<2, 144, 2, 8>("ds_read_b128 at 4-byte alignment", out, ticks);
    return 0;
}
