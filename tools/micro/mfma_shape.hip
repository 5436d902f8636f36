// Which 16-bit MFMA shape should the pair (f16x2) convolution use?  VERDICT r03 item 2(a) / MI355X_MICROARCH.md "DVFS give-back" item 7:
// on random data the chip holds a different clock under v_mfma_f32_16x16x32 than under v_mfma_f32_32x32x16 at equal cycles per FLOP.
// This measures both shapes for f16 and bf16 at the pair kernel's geometry: one consumer wave per SIMD owns 8 output rows x 32 voxels x
// 32 output channels (128 accumulator registers either way); per "tap step" it multiplies the row's B operand(s) with the tap's A operand(s).
//   32x32x16: 1 A fragment (32 cout x K16), per row 1 B fragment (K16 x 32 voxels)            -> 1 MFMA  (32 cycles)
//   16x16x32: 2 A fragments (16 cout x K32), per row 2 B fragments (K32 x 16 voxels) per TWO taps -> 4 MFMAs (64 cycles = two taps' worth)
// i.e. the same operand bytes per FLOP.  FEED 0: operands stay in registers; FEED 1: every B fragment is re-read from LDS by ds_read_b128
// (as the real kernel does); FEED 2: A fragments re-read from LDS as well.  Random fp16 / bf16 data.  Prints TFLOP/s, cycles per MFMA and
// the in-kernel clock (s_memtime / s_memrealtime).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_shape mfma_shape.hip && ./mfma_shape [seconds per case]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int ROWS = 8, TAPS = 28;          // 28 taps = 14 tap pairs (the real chunk: 27 taps + lo-part pairings, 41 MFMAs per row)
constexpr int ROWB = 36 * 32;               // bytes of one halo row in the LDS image (34 voxels x 32 B, padded)
constexpr int LDS_BYTES = 2 * 2048 * 32;    // two activation buffers as in the real kernel (128 KB)

// LOADER: waves 4-7 of a 512-thread workgroup (the same SIMDs, s_setprio 3) run the pair kernel's loader arithmetic beside the MFMA stream:
// per 8-channel chunk and lane 3 quad slots x 8 channels x 4 voxels = 96 values, each InstanceNorm + LeakyReLU (v_fma, v_mul, v_max),
// the fp16 split (v_cvt_pkrtz per 2 values, v_sub, v_cvt_f32_f16 x1, v_cvt_pk_f16 per 2) and 2 ds_write_b128 per voxel: ~1000 vector
// instructions per chunk of TAPS*ROWS 32x32x16-equivalents, i.e. the real loader's load on the SIMD's issue slots (no global loads here).
template <int SHAPE, int FEED, bool BF, bool LOADER = false>
__global__ __launch_bounds__(LOADER ? 512 : 256) void k(float* out, unsigned long long* stamps, int iters, unsigned seed) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (LOADER && wave >= 4) {
        __builtin_amdgcn_s_setprio(3);
        float x[8]; unsigned w[8];
        for (int i = 0; i < 8; ++i) x[i] = lane * 0.01f + i + (seed & 3);
        const float sc = 1.0003f, sh = 0.02f, slope = 0.01f;
        const unsigned addr = (unsigned)(size_t)(lds) + 96 * 1024 + (wave - 4) * 8192 + lane * 16;
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 12; ++q) {                   // 12 x (8 values -> one hi piece + one lo piece) = 96 values per chunk
#pragma unroll
                for (int c = 0; c < 8; c += 2) {
                    float z0, z1, t0_, t1_, r0, r1, l0, l1, h0f, h1f; unsigned hp, lp;
                    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(z0) : "v"(x[c]), "v"(sc), "v"(sh));
                    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(z1) : "v"(x[c + 1]), "v"(sc), "v"(sh));
                    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t0_) : "v"(z0), "v"(slope));
                    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t1_) : "v"(z1), "v"(slope));
                    asm volatile("v_max_f32 %0, %1, %2" : "=v"(r0) : "v"(z0), "v"(t0_));
                    asm volatile("v_max_f32 %0, %1, %2" : "=v"(r1) : "v"(z1), "v"(t1_));
                    asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(hp) : "v"(r0), "v"(r1));
                    asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(h0f) : "v"(hp));
                    asm volatile("v_cvt_f32_f16 %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(h1f) : "v"(hp));
                    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(l0) : "v"(r0), "v"(h0f));
                    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(l1) : "v"(r1), "v"(h1f));
                    asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(lp) : "v"(l0), "v"(l1));
                    w[c >> 1] = hp; w[4 + (c >> 1)] = lp;
                    x[c] = r0 + 0.5f; x[c + 1] = r1 - 0.25f;
                }
                asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(addr), "v"(*reinterpret_cast<const __attribute__((ext_vector_type(4))) unsigned*>(&w[0])), "n"((0) * 1024) : "memory");
                asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(addr), "v"(*reinterpret_cast<const __attribute__((ext_vector_type(4))) unsigned*>(&w[4])), "n"((1) * 1024) : "memory");
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        float sacc = 0.f;
        for (int i = 0; i < 8; ++i) sacc += x[i];
        out[blockIdx.x * 512 + threadIdx.x] = sacc;
        if (threadIdx.x == 256) stamps[512 + blockIdx.x] = t1 - t0;
        return;
    }
    unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + seed;
    auto rnd = [&]() { h = h * 1664525u + 1013904223u; return h; };
    auto rnd16 = [&]() -> unsigned {                 // one random 16-bit float in (-2, 2), full-range mantissa and sign
        const float v = ((int)(rnd() >> 19) - 4096) * (1.f / 2048);
        if (BF) return __builtin_bit_cast(unsigned, v) >> 16;
        return (unsigned)__builtin_bit_cast(unsigned short, (_Float16)v);
    };
    if (threadIdx.x < 256) for (int i = threadIdx.x; i < (LDS_BYTES - 32 * 1024) / 4; i += 256) lds[i] = rnd16() | (rnd16() << 16);
    s16x8 areg[4];
    for (int f = 0; f < 4; ++f) for (int e = 0; e < 8; ++e) areg[f][e] = (short)rnd16();
    s16x8 breg[2];
    for (int f = 0; f < 2; ++f) for (int e = 0; e < 8; ++e) breg[f][e] = (short)rnd16();
    if (!LOADER) __syncthreads();
    const char* ldsb = reinterpret_cast<const char*>(lds);
    // per-lane base: 32x32x16: voxel = lane & 31, 16-byte piece = lane >> 5; 16x16x32: voxel = lane & 15, piece group = lane >> 4
    // (the 32x32x16 image is XOR-swizzled by bit 3 of the voxel as in the real kernel: conflict-free ds_read_b128; the 16x16x32 lane map is
    // conflict-free on the plain image)
    const int vb32 = (wave * 2) * 10 * ROWB + (lane & 31) * 32 + ((lane >> 5) ^ (((lane & 31) >> 3) & 1)) * 16;
    const int vb16 = (wave * 2) * 10 * ROWB + (lane & 15) * 32 + ((lane >> 4) & 1) * 16 + ((lane >> 5) * 32);   // groups 2,3: the next voxel (second tap of the pair)
    const int ab = 64 * 1024 + lane * 16;            // A fragments: 1 KB each
    f32x16 acc32[ROWS];
    f32x4 acc16[ROWS][2][2];
    for (int m = 0; m < ROWS; ++m) {
        for (int r = 0; r < 16; ++r) acc32[m][r] = 0.f;
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 4; ++r) acc16[m][i][j][r] = 0.f;
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        const char* cur = ldsb + (it & 1) * 1024;     // loop-variant base: the reads cannot be hoisted
        if constexpr (SHAPE == 0) {
#pragma unroll
            for (int t = 0; t < TAPS; ++t) {
                s16x8 a = areg[t & 3];
                if constexpr (FEED == 2) a = *reinterpret_cast<const s16x8*>(cur + ab + (t % 27) * 1024);
#pragma unroll
                for (int m = 0; m < ROWS; ++m) {
                    s16x8 b = breg[m & 1];
                    if constexpr (FEED >= 1) b = *reinterpret_cast<const s16x8*>(cur + vb32 + ((t / 3) % 3 + m) * ROWB + (t % 3) * 32 + (t / 9) * 10 * ROWB);
                    if constexpr (BF) acc32[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc32[m], 0, 0, 0);
                    else acc32[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc32[m], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int tp = 0; tp < TAPS / 2; ++tp) {
                s16x8 a0 = areg[tp & 3], a1 = areg[(tp + 1) & 3];
                if constexpr (FEED == 2) {
                    a0 = *reinterpret_cast<const s16x8*>(cur + ab + (2 * tp % 27) * 1024);
                    a1 = *reinterpret_cast<const s16x8*>(cur + ab + ((2 * tp + 1) % 27) * 1024);
                }
#pragma unroll
                for (int m = 0; m < ROWS; ++m) {
                    s16x8 b0 = breg[0], b1 = breg[1];
                    if constexpr (FEED >= 1) {
                        const int off = ((tp / 2) % 3 + m) * ROWB + (tp % 2) * 32 + (tp / 5) * 10 * ROWB;
                        b0 = *reinterpret_cast<const s16x8*>(cur + vb16 + off);
                        b1 = *reinterpret_cast<const s16x8*>(cur + vb16 + off + 16 * 32);
                    }
                    if constexpr (BF) {
                        acc16[m][0][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0, acc16[m][0][0], 0, 0, 0);
                        acc16[m][1][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b0, acc16[m][1][0], 0, 0, 0);
                        acc16[m][0][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b1, acc16[m][0][1], 0, 0, 0);
                        acc16[m][1][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, acc16[m][1][1], 0, 0, 0);
                    } else {
                        const f16x8 A0 = __builtin_bit_cast(f16x8, a0), A1 = __builtin_bit_cast(f16x8, a1), B0 = __builtin_bit_cast(f16x8, b0), B1 = __builtin_bit_cast(f16x8, b1);
                        acc16[m][0][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A0, B0, acc16[m][0][0], 0, 0, 0);
                        acc16[m][1][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A1, B0, acc16[m][1][0], 0, 0, 0);
                        acc16[m][0][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A0, B1, acc16[m][0][1], 0, 0, 0);
                        acc16[m][1][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A1, B1, acc16[m][1][1], 0, 0, 0);
                    }
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int m = 0; m < ROWS; ++m) {
        if (SHAPE == 0) for (int r = 0; r < 16; ++r) s += acc32[m][r];
        else for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 4; ++r) s += acc16[m][i][j][r];
    }
    out[blockIdx.x * (LOADER ? 512 : 256) + threadIdx.x] = s;
    if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int SHAPE, int FEED, bool BF, bool LOADER = false>
static void run(const char* name, double secs, float* out, unsigned long long* stamps) {
    auto kern = k<SHAPE, FEED, BF, LOADER>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    double total = 0, last = 0;
    for (int rep = 0; total < secs * 1e3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(LOADER ? 512 : 256), LDS_BYTES, 0, out, stamps, iters, (unsigned)rep);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        total += ms; last = ms;
    }
    std::vector<unsigned long long> st(768);
    hipMemcpy(st.data(), stamps, 768 * 8, hipMemcpyDeviceToHost);
    std::vector<double> mhz, cyc;
    for (int b = 0; b < 256; ++b) { mhz.push_back((double)st[2 * b] / (double)st[2 * b + 1] * 100.0); cyc.push_back((double)st[2 * b]); }
    std::sort(mhz.begin(), mhz.end()); std::sort(cyc.begin(), cyc.end());
    const double flop = (double)iters * TAPS * ROWS * 2.0 * 32 * 32 * 16 * 4 * 256;     // both shapes: the same FLOPs per tap step
    const double n32 = (double)iters * TAPS * ROWS;                                      // 32x32x16-equivalents per wave
    std::vector<double> lc;
    for (int b = 0; b < 256; ++b) lc.push_back((double)st[512 + b]);
    std::sort(lc.begin(), lc.end());
    printf("%-44s %8.1f TFLOP/s  %6.2f cycles per 32x32x16-equivalent  in-kernel clock %5.0f MHz  (%.3f ms per launch)", name,
           flop / (last * 1e-3) / 1e12, cyc[128] / n32, mhz[128], last);
    if (LOADER) printf("  loader wave: %.0f k cycles per chunk (consumer: %.0f k)", lc[128] / iters / 1e3, cyc[128] / iters / 1e3);
    printf("\n");
    fflush(stdout);
}

int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 2.5;
    float* out; hipMalloc(&out, 256 * 256 * 4);
    unsigned long long* stamps; hipMalloc(&stamps, 768 * 8); hipMemset(stamps, 0, 768 * 8);
    float* out2; hipMalloc(&out2, 256 * 512 * 4);
    for (int round = 0; round < 2; ++round) {                  // two interleaved rounds: the order must not decide
        run<0, 0, false>("f16  32x32x16  registers", secs, out, stamps);
        run<1, 0, false>("f16  16x16x32  registers", secs, out, stamps);
        run<0, 1, false>("f16  32x32x16  B from LDS", secs, out, stamps);
        run<1, 1, false>("f16  16x16x32  B from LDS", secs, out, stamps);
        run<0, 2, false>("f16  32x32x16  A and B from LDS", secs, out, stamps);
        run<1, 2, false>("f16  16x16x32  A and B from LDS", secs, out, stamps);
        run<0, 0, true>("bf16 32x32x16  registers", secs, out, stamps);
        run<1, 0, true>("bf16 16x16x32  registers", secs, out, stamps);
        run<0, 1, true>("bf16 32x32x16  B from LDS", secs, out, stamps);
        run<1, 1, true>("bf16 16x16x32  B from LDS", secs, out, stamps);
        // the pair kernel's situation: a loader wave on every SIMD beside the consumer
        run<0, 2, false, true>("f16  32x32x16  A and B from LDS + loader waves", secs, out2, stamps);
        run<1, 2, false, true>("f16  16x16x32  A and B from LDS + loader waves", secs, out2, stamps);
    }
    return 0;
}
