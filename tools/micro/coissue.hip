// How do vector / LDS instructions of one wave share a SIMD with the MFMA stream of another wave?  (gfx950)
// 512-thread workgroups, one per CU: waves 0-3 issue v_mfma_f32_32x32x16_bf16 back to back (8 independent accumulators),
// waves 4-7 -- same SIMDs -- issue R instructions of one kind per MFMA of their partner.  Both roles time themselves with
// s_memtime; the host prints cycles per MFMA (consumer) and cycles per instruction (loader), alone and together.
//   hipcc --offload-arch=gfx950 -O3 -o coissue coissue.hip && ./coissue
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum { K_FMA = 0, K_PKFMA, K_CVTPK, K_INT, K_DSREAD, K_DSWRITE, K_TRANSFORM, K_MAX3, K_BFI, K_CNDMASK, K_ADD3, K_PERM, K_CVTF32F16, K_CVTPKRTZ, K_MOV, K_ADDF, K_TRANSFORM_PLAIN, K_PKMUL, K_NKINDS };
static const char* kind_name[K_NKINDS] = {"v_fma_f32", "v_pk_fma_f32", "v_cvt_pk_bf16_f32", "v_lshl/v_and", "ds_read_b128", "ds_write_b128",
                                          "transform (rd+28valu+wr)/30", "v_med3_f32", "v_bfi_b32", "v_cmp_gt_i32+v_cndmask", "v_add3_u32", "v_perm_b32", "v_cvt_f32_f16",
                                          "v_cvt_pkrtz_f16_f32", "v_mov_b32", "v_add_f32", "transform, plain fp32 (rd+36valu+wr)/38", "v_pk_mul_f32"};

template <int KIND, int PER>      // PER loader instructions per block of 8 MFMAs
__global__ __launch_bounds__(512) void k(unsigned long long* out, float* sink, int iters, int run_mfma, int run_loader, int prio) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[64 * 1024];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < 4) {
        if (!run_mfma) return;
        f32x16 acc[8];
        for (int m = 0; m < 8; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
        s8 a8 = {1, 2, 3, 4, 5, 6, 7, (short)threadIdx.x}, b8 = {4, 5, 6, 7, 8, 9, 1, (short)threadIdx.x};
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int m = 0; m < 8; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc[m], 0, 0, 0);
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        float s = 0.f;
        for (int m = 0; m < 8; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
        sink[blockIdx.x * 512 + threadIdx.x] = s;
        if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
    } else {
        if (!run_loader) return;
        if (prio) __builtin_amdgcn_s_setprio(3);
        float x[8]; f32x2 p[8]; unsigned u[8]; u32x4 q[4];
        for (int i = 0; i < 8; ++i) { x[i] = lane * 0.01f + i; p[i] = f32x2{x[i], x[i] + 1.f}; u[i] = lane * 77u + i; }
        for (int i = 0; i < 4; ++i) q[i] = u32x4{u[i], u[i + 1], u[i + 2], u[i + 3]};
        const unsigned addr = (unsigned)(size_t)(lds) + (wave - 4) * 16384 + lane * 16;
        const float c1 = 1.0001f, c2 = 0.5f;
        const f32x2 pc = {1.0001f, 0.9999f}, pd = {0.001f, 0.002f};
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const int r = j & 7;
                if (KIND == K_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[r]) : "v"(c1), "v"(c2));
                if (KIND == K_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[r]) : "v"(pc), "v"(pd));
                if (KIND == K_CVTPK) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[r]) : "v"(x[r]), "v"(x[(r + 1) & 7]));
                if (KIND == K_INT) { if (j & 1) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(u[r])); else asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(u[r])); }
                if (KIND == K_MAX3) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x[r]) : "v"(c1), "v"(c2));
                if (KIND == K_BFI) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(u[r]) : "v"(u[(r + 1) & 7]), "v"(u[(r + 2) & 7]));
                if (KIND == K_CNDMASK) { if (j & 1) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[r]) : "v"(u[(r + 1) & 7]) : "vcc"); else asm volatile("v_cmp_gt_i32 vcc, 0, %0" :: "v"(u[r]) : "vcc"); }
                if (KIND == K_ADD3) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(u[r]) : "v"(u[(r + 1) & 7]), "v"(u[(r + 2) & 7]));
                if (KIND == K_PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[r]) : "v"(u[(r + 1) & 7]), "v"(0x07060302u));
                if (KIND == K_CVTF32F16) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(x[r]) : "v"(u[r]));
                if (KIND == K_CVTPKRTZ) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(u[r]) : "v"(x[r]), "v"(x[(r + 1) & 7]));
                if (KIND == K_MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(u[r]) : "v"(u[(r + 1) & 7]));
                if (KIND == K_ADDF) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[r]) : "v"(c2));
                if (KIND == K_DSREAD) {
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[j & 3]) : "v"(addr), "n"((j & 7) * 1024));
                    if ((j & 3) == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                if (KIND == K_DSWRITE) {
                    asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(addr), "v"(q[j & 3]), "n"((j & 7) * 1024) : "memory");
                    if ((j & 3) == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                if (KIND == K_PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[r]) : "v"(pc));
                if (KIND == K_TRANSFORM_PLAIN && (j % 38) == 0) {
                    // one 16-byte piece with unpacked float instructions: read, 4 x (shl, and, 4 x v_fma (z, z', slope z, slope z'), med3, med3, cvt_pk), write
                    u32x4 w;
                    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(w) : "v"(addr + ((j / 38) & 7) * 1024) : "memory");
#pragma unroll
                    for (int pp = 0; pp < 4; ++pp) {
                        unsigned lo, hi; float z0, z1, y0, y1, m0, m1; unsigned o;
                        asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(lo) : "v"(w[pp]));
                        asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(hi) : "v"(w[pp]));
                        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(z0) : "v"(lo), "v"(c1), "v"(c2));
                        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(y0) : "v"(lo), "v"(c2), "v"(c1));
                        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(z1) : "v"(hi), "v"(c1), "v"(c2));
                        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(y1) : "v"(hi), "v"(c2), "v"(c1));
                        asm volatile("v_med3_f32 %0, %1, %2, %3" : "=v"(m0) : "v"(z0), "v"(y0), "v"(c1));
                        asm volatile("v_med3_f32 %0, %1, %2, %3" : "=v"(m1) : "v"(z1), "v"(y1), "v"(c1));
                        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(o) : "v"(m0), "v"(m1));
                        w[pp] = o;
                    }
                    asm volatile("ds_write_b128 %0, %1" :: "v"(addr + ((j / 38) & 7) * 1024), "v"(w) : "memory");
                }
                if (KIND == K_TRANSFORM && (j % 30) == 0) {
                    // one 16-byte piece: read, 4 x (shl, and, pk_fma, pk_mul, med3, med3, cvt_pk), write = 30 instructions
                    u32x4 w;
                    asm volatile("ds_read_b128 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(w) : "v"(addr), "n"(((j / 30) & 7) * 1024) : "memory");
#pragma unroll
                    for (int pp = 0; pp < 4; ++pp) {
                        unsigned lo, hi; f32x2 z, zs; float m0, m1; unsigned o;
                        asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(lo) : "v"(w[pp]));
                        asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(hi) : "v"(w[pp]));
                        f32x2 xin = {__builtin_bit_cast(float, lo), __builtin_bit_cast(float, hi)};
                        asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(z) : "v"(xin), "v"(pc), "v"(pd));
                        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(zs) : "v"(z), "v"(pc));
                        asm volatile("v_med3_f32 %0, %1, %2, %3" : "=v"(m0) : "v"(z[0]), "v"(zs[0]), "v"(c1));
                        asm volatile("v_med3_f32 %0, %1, %2, %3" : "=v"(m1) : "v"(z[1]), "v"(zs[1]), "v"(c1));
                        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(o) : "v"(m0), "v"(m1));
                        w[pp] = o;
                    }
                    asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(addr), "v"(w), "n"(((j / 30) & 7) * 1024) : "memory");
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += x[i] + p[i][0] + p[i][1] + (float)u[i];
        for (int i = 0; i < 4; ++i) s += (float)(q[i][0] ^ q[i][3]);
        sink[blockIdx.x * 512 + threadIdx.x] = s;
        if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
    }
}

static unsigned long long* d_out; static float* d_sink;
template <int KIND, int PER>
static void run(int iters, int prio = 0) {
    unsigned long long h[256 * 8];
    double res[3][2];
    float ms[3];
    const int modes[3][2] = {{1, 0}, {0, 1}, {1, 1}};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int md = 0; md < 3; ++md) {
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(d_out, 0, sizeof(h));
            hipEventRecord(e0);
            hipLaunchKernelGGL((k<KIND, PER>), dim3(256), dim3(512), 0, 0, d_out, d_sink, iters, modes[md][0], modes[md][1], prio);
            hipEventRecord(e1);
            hipDeviceSynchronize();
        }
        hipEventElapsedTime(&ms[md], e0, e1);
        hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
        double sm = 0, sl = 0;
        for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? sm : sl) += (double)h[b * 8 + w];
        res[md][0] = sm / (256 * 4) / (iters * 8.0);
        res[md][1] = sl / (256 * 4) / ((double)iters * PER);
    }
    // s_memtime ticks -> shader cycles: the MFMA-only run issues one MFMA per 32 cycles... report raw ticks and wall time
    if (prio) printf("[loader at s_setprio 3] ");
    printf("%-30s %4.1f per MFMA | alone: mfma %6.2f ticks (%.3f ms), loader %6.2f ticks/instr (%.3f ms) | together: mfma %6.2f, loader %6.2f (%.3f ms)\n",
           kind_name[KIND], PER / 8.0, res[0][0], ms[0], res[1][1], ms[1], res[2][0], res[2][1], ms[2]);
}

int main(int argc, char** argv) {
    hipMalloc(&d_out, 256 * 8 * 8); hipMalloc(&d_sink, 256 * 512 * 4);
    const int iters = 4000;
    if (argc > 1 && argv[1][0] == 'p') {      // loaders at raised priority, as in the real kernels
        run<K_FMA, 8>(iters, 1); run<K_FMA, 16>(iters, 1); run<K_FMA, 32>(iters, 1); run<K_PKFMA, 16>(iters, 1); run<K_CVTPK, 8>(iters, 1);
        run<K_INT, 32>(iters, 1); run<K_TRANSFORM, 30>(iters, 1); run<K_TRANSFORM, 60>(iters, 1); run<K_TRANSFORM, 120>(iters, 1);
        run<K_TRANSFORM_PLAIN, 38>(iters, 1); run<K_TRANSFORM_PLAIN, 76>(iters, 1); run<K_TRANSFORM_PLAIN, 152>(iters, 1);
        run<K_TRANSFORM_PLAIN, 38>(iters, 0);
        run<K_PKMUL, 16>(iters, 1); run<K_PKFMA, 8>(iters, 1); run<K_PKFMA, 32>(iters, 1); run<K_MAX3, 32>(iters, 1); run<K_CVTPK, 32>(iters, 1);
        run<K_CVTF32F16, 32>(iters, 1); run<K_CVTPKRTZ, 32>(iters, 1); run<K_DSREAD, 16>(iters, 1); run<K_DSWRITE, 8>(iters, 1);
        return 0;
    }
    if (argc > 1) {           // the first table (round 3, profiles/r03_coissue.txt)
    run<K_FMA, 4>(iters); run<K_FMA, 8>(iters); run<K_FMA, 16>(iters); run<K_FMA, 32>(iters); run<K_FMA, 64>(iters);
    run<K_PKFMA, 8>(iters); run<K_PKFMA, 16>(iters); run<K_PKFMA, 32>(iters);
    run<K_CVTPK, 8>(iters); run<K_CVTPK, 32>(iters);
    run<K_INT, 8>(iters); run<K_INT, 32>(iters);
    run<K_MAX3, 8>(iters); run<K_MAX3, 32>(iters);
    run<K_DSREAD, 8>(iters); run<K_DSREAD, 16>(iters); run<K_DSREAD, 32>(iters);
    run<K_DSWRITE, 8>(iters); run<K_DSWRITE, 16>(iters);
    }
    run<K_TRANSFORM, 30>(iters); run<K_TRANSFORM, 60>(iters); run<K_TRANSFORM, 120>(iters);
    run<K_INT, 64>(iters); run<K_INT, 128>(iters);
    run<K_BFI, 32>(iters); run<K_CNDMASK, 32>(iters); run<K_ADD3, 32>(iters); run<K_PERM, 32>(iters); run<K_MOV, 32>(iters);
    run<K_CVTF32F16, 32>(iters); run<K_CVTPKRTZ, 32>(iters); run<K_ADDF, 32>(iters);
    return 0;
}
