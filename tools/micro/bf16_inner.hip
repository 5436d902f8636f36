// Consumer inner loop of the bf16 conv kernel in isolation (cfg0 geometry: 8 M-tiles per wave, one wave per SIMD = 256
// threads, LDS prefilled): cycles per v_mfma_f32_32x32x16_bf16 when the B operand comes from LDS by ds_read_b128
//   MODE 0  with the kernel's XOR-swizzled [voxel][2 halves] layout, A constant
//   MODE 1  registers only (no LDS): the instruction's issue rate
//   MODE 2  as 0 plus the A operand through a 12-deep ring of 16-byte global loads (L2-resident weights)
//   MODE 3  as 0 but every lane reads the SAME 16 bytes (broadcast: no bank pressure)
//   MODE 4  as 2 without the scheduling barrier;   MODE 5  A operand by ds_read_b128 from a weights region of LDS
//   MODE 7  A through a 9-slot register ring indexed statically (slot = tap % 9: no register moves)
//   MODE 6  as 2 with the loads issued but the ring not consumed (A constant): cost of the load instructions alone
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
constexpr int IX = 34, IY = 10, IZ = 6, YS = IX, ZS = IY * IX, CS = 2048, MT = 8;

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(unsigned long long* out, float* sink, const bf16x8* w, int iters) {
    extern __shared__ uint4 ldsq[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < CS * 2 + 27 * 64; e += 256) ldsq[e] = make_uint4(e * 3u, e * 5u, e * 7u, e * 11u);
    __syncthreads();
    const int j = lane & 31, kh = lane >> 5, dx = j;
    int badr[3][MT];
    for (int m = 0; m < MT; ++m) {
        const int ts = wave * MT + m, tyi = ts % 8, tzi = ts / 8;
        const int e0 = tzi * ZS + tyi * YS + dx;
        for (int kw = 0; kw < 3; ++kw) badr[kw][m] = MODE == 3 ? 0 : (e0 + kw) * 32 + 16 * (kh ^ (((dx + kw) >> 3) & 1));
    }
    f32x16 acc[MT];
    for (int m = 0; m < MT; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    const char* ldsb = reinterpret_cast<const char*>(ldsq);
    bf16x8 a0 = {1, 2, 3, 4, 5, 6, 7, (short)tid};
    constexpr int APF = 12;
    bf16x8 ring[APF];
    const bf16x8* wl = w + lane;
    if (MODE == 2 || MODE == 4 || MODE == 6) for (int d = 0; d < APF; ++d) ring[d] = wl[d * 64];
    bf16x8 r9[9];
    if (MODE == 7) for (int d = 0; d < 9; ++d) r9[d] = wl[d * 64];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int tap = 0; tap < 27; ++tap) {
            const int kw = tap % 3;
            const int rowoff = ((tap / 9) * ZS + ((tap / 3) % 3) * YS) * 32;
            bf16x8 anew;
            if (MODE == 2 || MODE == 4 || MODE == 6) { anew = wl[(((it * 27 + tap + APF) % 54)) * 64]; if (MODE != 4) __builtin_amdgcn_sched_barrier(0x38F); }
            bf16x8 alds;
            if (MODE == 5) alds = *reinterpret_cast<const bf16x8*>(ldsb + CS * 32 + (tap * 64 + lane) * 16);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                bf16x8 bv;
                if (MODE == 1) bv = a0; else bv = *reinterpret_cast<const bf16x8*>(ldsb + badr[kw][m] + rowoff);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16((MODE == 2 || MODE == 4) ? ring[0] : MODE == 5 ? alds : MODE == 7 ? r9[tap % 9] : a0, bv, acc[m], 0, 0, 0);
            }
            if (MODE == 7) { r9[tap % 9] = wl[(((it * 27 + tap + 9) % 54)) * 64]; __builtin_amdgcn_sched_barrier(0x38F); }
            if (MODE == 2 || MODE == 4 || MODE == 6) {
#pragma unroll
                for (int d = 0; d + 1 < APF; ++d) ring[d] = ring[d + 1];
                ring[APF - 1] = anew;
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    if (MODE == 6) for (int d = 0; d < APF; ++d) s += (float)ring[d][0];
    for (int m = 0; m < MT; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
    sink[blockIdx.x * 256 + tid] = s;
    if (tid == 0) out[blockIdx.x] = t1 - t0;
}
template <int MODE>
static void run(int nblk, unsigned long long* out, float* sink, const bf16x8* w) {
    const int lds = CS * 32 + 27 * 64 * 16, iters = 100;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k<MODE>, dim3(nblk), dim3(256), lds, 0, out, sink, w, iters); hipDeviceSynchronize(); }
    static unsigned long long h[1024];
    hipMemcpy(h, out, nblk * 8, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < nblk; ++i) s += h[i];
    printf("mode %d  blocks %4d: %.1f ticks per MFMA per wave\n", MODE, nblk, s / nblk / (100.0 * 27 * MT));
}
int main() {
    unsigned long long* out; float* sink; bf16x8* w;
    hipMalloc(&out, 1024 * 8); hipMalloc(&sink, 1024 * 256 * 4); hipMalloc(&w, 64 * 64 * 16); hipMemset(w, 1, 64 * 64 * 16);
    for (int nblk : {256}) { run<1>(nblk, out, sink, w); run<0>(nblk, out, sink, w); run<2>(nblk, out, sink, w); run<4>(nblk, out, sink, w); run<5>(nblk, out, sink, w); run<6>(nblk, out, sink, w); run<7>(nblk, out, sink, w); }
    return 0;
}
