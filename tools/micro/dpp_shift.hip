// "LDS reads per MFMA are a measurable energy term ... form the kw +- 1 operands of a row from ONE ds_read with DPP
// row shifts and report clock and time, not cycles".  The stride-1 pair kernel's consumer reads one 16-byte B operand per lane and MFMA
// (1 KB per wave-instruction) from LDS; the three kw taps of an input row are the same 32 voxels shifted by one lane.  This probe runs the
// consumer's instruction mix on every CU (one 256-thread workgroup per CU: one consumer wave per SIMD, as in the kernel) for long enough that
// the power management settles, and reports per variant the launch time, the shader cycles (s_memtime) and the clock they imply:
//   MODE 0  three ds_read_b128 per input row and M-tile (the kernel as it is: one LDS read per MFMA)
//   MODE 1  one ds_read_b128 per row (the kw = 1 operand); the kw = 0 / 2 operands by 2 x 4 v_mov_b32_dpp (row_shr:1 / row_shl:1) --
//           (the lanes at the row ends would need the halo voxels' pieces from a second, two-lane read; written as an exec-masked
//           ds_read inside the loop -- MODE 2 -- that alone took 98 cycles per MFMA: the branch ends the block and every wait becomes a full one)
//   MODE 3  registers only (no LDS, no DPP): the MFMA stream's own time and clock
// 8 M-tiles per wave, 27 taps, fp16 32x32x16, A operand constant (the kernel reads it from LDS once per 8 MFMAs).
//   hipcc --offload-arch=gfx950 -O3 -o dpp_shift dpp_shift.hip && ./dpp_shift
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int IX = 34, IY = 10, IZ = 6, YS = IX, ZS = IY * IX, CS = 2048, MT = 8;

template <int CTRL>
__device__ __forceinline__ u32x4 shift(u32x4 v) {
    u32x4 r;
    r.x = __builtin_amdgcn_update_dpp(0u, v.x, CTRL, 0xF, 0xF, true);
    r.y = __builtin_amdgcn_update_dpp(0u, v.y, CTRL, 0xF, 0xF, true);
    r.z = __builtin_amdgcn_update_dpp(0u, v.z, CTRL, 0xF, 0xF, true);
    r.w = __builtin_amdgcn_update_dpp(0u, v.w, CTRL, 0xF, 0xF, true);
    return r;
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(unsigned long long* out, float* sink, int iters) {
    extern __shared__ uint4 ldsq[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // pseudo-random fp16 values in (-2, 2): the clock the part grants under its power cap answers to the operands' bit activity
    // (with a constant image this probe ran at 2.34 GHz; the pair kernel, on activations, runs at 1.4-1.5)
    auto rnd = [](unsigned x) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; x *= 2654435761u; return ((x & 0x83ff83ffu) | 0x3c003c00u) ^ ((x >> 3) & 0x00ff00ffu); };
    for (int e = tid; e < CS * 2; e += 256) ldsq[e] = make_uint4(rnd(4 * e + 1), rnd(4 * e + 2), rnd(4 * e + 3), rnd(4 * e + 4));
    __syncthreads();
    const int j = lane & 31, kh = lane >> 5;
    int badr[MT];
    for (int m = 0; m < MT; ++m) {
        const int ts = wave * MT + m, tyi = ts % 8, tzi = ts / 8;
        badr[m] = (tzi * ZS + tyi * YS + j + 1) * 32 + 16 * (kh ^ (((j + 1) >> 3) & 1));
    }
    f32x16 acc[MT];
    for (int m = 0; m < MT; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    const char* ldsb = reinterpret_cast<const char*>(ldsq);
    const u32x4 a0u = {rnd(tid + 77), rnd(tid + 177), rnd(tid + 277), rnd(tid + 377)};
    const f16x8 a0 = __builtin_bit_cast(f16x8, a0u);
    const bool edge = (j == 0) || (j == 31);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int row = 0; row < 9; ++row) {                          // (kd, kr)
            const int rowoff = ((row / 3) * ZS + (row % 3) * YS) * 32;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                u32x4 b[3];
                if (MODE == 3) { b[0] = u32x4{rnd(tid + m), rnd(tid + 2 * m + 1), rnd(tid + 3), rnd(tid + 5)}; b[1] = u32x4{b[0].y, b[0].z, b[0].w, b[0].x}; b[2] = u32x4{b[0].z, b[0].w, b[0].x, b[0].y}; }
                else if (MODE == 0) {
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) b[kw] = *reinterpret_cast<const u32x4*>(ldsb + badr[m] + rowoff + (kw - 1) * 32);
                } else {
                    b[1] = *reinterpret_cast<const u32x4*>(ldsb + badr[m] + rowoff);
                    b[0] = shift<0x111>(b[1]);                          // row_shr:1
                    b[2] = shift<0x101>(b[1]);                          // row_shl:1
                    if (MODE == 2 && edge) {                            // the row ends: the halo voxels' pieces (two lanes per half-wave)
                        const u32x4 h = *reinterpret_cast<const u32x4*>(ldsb + badr[m] + rowoff + (j == 0 ? -32 : 32));
                        if (j == 0) b[0] = h; else b[2] = h;
                    }
                }
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, __builtin_bit_cast(f16x8, b[kw]), acc[m], 0, 0, 0);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int m = 0; m < MT; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
    sink[blockIdx.x * 256 + tid] = s;
    if (tid == 0) out[blockIdx.x] = t1 - t0;
}
template <int MODE>
static void run(int nblk, unsigned long long* out, float* sink, const char* what) {
    const int lds = CS * 32, iters = 4000;                           // 4000 x 216 MFMAs x 32 cycles ~ 28 M cycles ~ 15-20 ms per launch
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(nblk), dim3(256), lds, 0, out, sink, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    static unsigned long long h[1024];
    hipMemcpy(h, out, nblk * 8, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < nblk; ++i) s += h[i];
    const double cyc = s / nblk, nmfma = (double)iters * 27 * MT;
    printf("mode %d %-58s: %7.3f ms  %.1f cycles per MFMA  in-kernel clock %4.0f MHz  %.0f TFLOP/s\n", MODE, what, ms, cyc / nmfma, cyc / ms / 1e3,
           nmfma * 4 * nblk * 2.0 * 32 * 32 * 16 / ms / 1e9);
}
int main() {
    int ncu = 256; hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    unsigned long long* out; float* sink;
    hipMalloc(&out, 1024 * 8); hipMalloc(&sink, 1024 * 256 * 4);
    for (int pass = 0; pass < 2; ++pass) {
        run<3>(ncu, out, sink, "registers only (MFMA stream alone)");
        run<0>(ncu, out, sink, "one ds_read_b128 per MFMA (the kernel)");
        run<1>(ncu, out, sink, "one ds_read_b128 per row + 8 v_mov_dpp");
    }
    return 0;
}
