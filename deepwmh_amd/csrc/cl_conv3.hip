// cl_conv3.hip -- 3x3x3 convolution (stride 1) on channels-last bf16 tensors ("CL16", cl_common.h): the convolution of
// BASELINE.json configs[2] for the levels whose rows are at least 32 voxels long (where ~95 % of the bytes are).
//
//   D[cout 32][voxel 32] += A[cout 32][16 cin of one tap] * B[16 cin][voxel 32]       (v_mfma_f32_32x32x16_bf16)
//
// One persistent 768-thread workgroup per CU, twelve waves in three roles:
//   waves 4-10 staging.  A chunk (16 input channels of the tile's halo, 6 x 6 x 34 voxels) comes in by LDS-DMA
//              (global_load_lds_dwordx4: a lane's 16 bytes are 8 channels of one voxel, already the B operand) -- no
//              staging registers, so three chunks ride in LDS: one being consumed, one being normalised, one landing.
//              The producer's InstanceNorm + LeakyReLU is applied IN PLACE in LDS (read 16 B, FMAs + max + pack, write
//              16 B); voxels outside the volume become zeros there.  These waves issue LDS-DMA ONLY (see the weight wave).
//   wave 11    weights.  The next chunk's 27 KB ride in its registers (ordinary loads ONLY: a counted s_waitcnt vmcnt over a mix
//              of ordinary loads and LDS-DMA does not say which are still out) and go into the single LDS weight buffer in
//              three kd-slices under the consumers' MFMAs (an LDS counter per slice: a consumer wave counts itself out of a
//              slice when it has issued its last fragment read of it); one workgroup barrier per chunk iteration.
//   waves 0-3  consumers: one z-slice of the tile each (TY rows of 32 voxels).  LDS image is planar, [half][voxel] x 16 B
//              (half = channels 0-7 / 8-15 of the chunk): consecutive lanes read consecutive 16-byte slots, conflict-free
//              for every tap shift without a swizzle, and a DMA wave-instruction fills 64 consecutive voxels of one half.
//              An input row is read once per kd and feeds the up to three output rows that use it (54 + 27 LDS reads for 108
//              MFMAs per chunk).  Epilogue: 16-byte stores (cl_store_mtile), fp32 InstanceNorm moments kept as running sums
//              per (workgroup, sample) and flushed once per sample.
// DESIGN.md 3.4 has the measurements (what bounds it, the ablations, the vmcnt finding).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "cl_common.h"

struct ClConvArgs {
    const uint16_t* x0; const float* ss0; int C0;
    const uint16_t* x1; const float* ss1; int C1;
    const uint16_t* wf; const float* bias; uint16_t* y; float* part;
    int N, D, H, W, Cout;
    int nchunks, ntx, nty, ntz, npart, G, xcd_swizzle;
    unsigned mG, mtx, mty, mtz;      // ceil(2^32 / d) of the four tile-index divisors (host checked the range they are exact for)
    float slope;
#ifdef WMH_CL_DIAG
    unsigned long long* stamps;      // diagnostic build only (tools/cl_diag.py): per-workgroup cycle sums, [grid][16]
#endif
};
#ifdef WMH_CL_DIAG
// Diagnostic build (tools/cl_diag.py).  -DWMH_CL_DIAG=1: s_memtime stamps in the staging waves, =2: in the consumer waves -- both at
// once, or run-time ablation switches, cost registers this kernel does not have (244 bytes of scratch, an epilogue 10x slower).
// Timing-only ablations, compile-time (-DWMH_CL_ABL_MASK=m; results are wrong): bit 1 = skip the activation DMA, 3 the transform,
// 4 the MFMAs, 5 the epilogue, 7 the operand reloads, 8 the sliced weight hand-over, 9 the stores, 10 the moments.
#ifndef WMH_CL_ABL_MASK
#define WMH_CL_ABL_MASK 0
#endif
#define CL_ABL(bit) ((WMH_CL_ABL_MASK >> (bit)) & 1)
#define CL_STAMP_(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define CL_STAMP_ADD_(acc, a, b) acc += (unsigned)((b) - (a))
#if WMH_CL_DIAG == 2
#define CL_STAMP(var)
#define CL_STAMP_ADD(acc, a, b)
#define CL_CSTAMP(var) CL_STAMP_(var)
#define CL_CSTAMP_ADD(acc, a, b) CL_STAMP_ADD_(acc, a, b)
#else
#define CL_STAMP(var) CL_STAMP_(var)
#define CL_STAMP_ADD(acc, a, b) CL_STAMP_ADD_(acc, a, b)
#define CL_CSTAMP(var)
#define CL_CSTAMP_ADD(acc, a, b)
#endif
static unsigned long long* g_cl_stamps = nullptr;
extern "C" void wmh_cl_diag_stamps(unsigned long long* p) { g_cl_stamps = p; }
#else
#define CL_ABL(bit) 0
#define CL_STAMP(var)
#define CL_STAMP_ADD(acc, a, b)
#define CL_CSTAMP(var)
#define CL_CSTAMP_ADD(acc, a, b)
#endif

template <int TZ_, int TY_>
struct ClGeo {
    static constexpr int TZ = TZ_, TY = TY_, TX = 32;
    static constexpr int IZ = TZ + 2, IY = TY + 2, IX = TX + 2;
    static constexpr int NV = IZ * IY * IX;                       // halo voxels of a tile
    static constexpr int LTH = 448;                               // staging threads: 7 waves (the 8th loader wave moves the weights)
    static constexpr int NSLOT = (NV + LTH - 1) / LTH;            // voxels per loader lane (the last slot: only the waves below NVP)
    static constexpr int NVP = (NV + 63) / 64 * 64;               // plane stride: whole wave-instructions
    static constexpr int BUF = 2 * NVP * 16;                      // bytes of one activation buffer: [2 halves][NVP] x 16
    static constexpr int WQ = 27 * 64;                            // 16-byte weight pieces per chunk (one 32-cout tile)
    static constexpr int WIT = WQ / 64;                           // per lane of the weight wave
    static constexpr int SCR = 4 * 256;                           // per consumer wave: bias and moment shift of its 2 x 16 rows
    static constexpr int LDS = 3 * BUF + WQ * 16 + SCR;           // + 2 scale/shift tables of 8 bytes per input channel (launcher)
    static_assert((NSLOT - 1) * LTH < NVP && NVP <= NSLOT * LTH, "slot split");
    static constexpr int MT = TY;                                 // M-tiles (rows) per consumer wave
    static_assert(TZ == 4, "four consumer waves, one z-slice each");
    static_assert(LDS <= 160 * 1024, "LDS");
};

// t / d and t % d for a wave-uniform t through the host's reciprocal (two scalar multiplies instead of ~30 instructions)
__device__ __forceinline__ void cl_divmod(int t, int d, unsigned m, int& q, int& r) {
    const int qq = d == 1 ? t : (int)__umulhi((unsigned)t, m);          // (q may alias t; ceil(2^32 / 1) does not fit 32 bits)
    r = t - qq * d;
    q = qq;
}

template <class G>
__global__ __launch_bounds__(768, 3) void cl_conv3_kernel(const ClConvArgs a, const int T /*tiles*/) {
    constexpr int IY = G::IY, IX = G::IX, NV = G::NV, NSLOT = G::NSLOT, NVP = G::NVP, BUF = G::BUF, WQ = G::WQ, WIT = G::WIT, MT = G::MT, LTH = G::LTH;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* wlds = lds + 3 * BUF;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave >= 4;
    const int HW = a.H * a.W;
    const size_t DHW = (size_t)a.D * HW;
    const int nch = a.nchunks;

    // tile k of this workgroup -> (cout group, tile coordinates, sample); everything stays scalar
    auto tile_of = [&](int k, int& g, int& tx, int& ty, int& tz, int& n0) __attribute__((always_inline)) {
        const int v = blockIdx.x + k * gridDim.x;
        const int q = T >> 3, r = T & 7, xcd = v & 7;
        int t = (a.xcd_swizzle & 1) ? (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3) : v;
        t = __builtin_amdgcn_readfirstlane(t);
        cl_divmod(t, a.G, a.mG, t, g);
        cl_divmod(t, a.ntx, a.mtx, t, tx);
        cl_divmod(t, a.nty, a.mty, t, ty);
        cl_divmod(t, a.ntz, a.mtz, n0, tz);
    };
    int ntiles_mine = 0;
    for (int v = blockIdx.x; v < T; v += gridDim.x) ++ntiles_mine;
    const int total = ntiles_mine * nch;                    // chunk iterations of this workgroup

    if (wave == 11) {
        // ================================ weight wave ======================================================
        // One wave moves the weight fragments: chunk it+1's 27 KB ride in its registers (27 x 16 bytes per lane) while chunk it
        // is consumed, and drop into the single LDS weight buffer between the two barriers of the chunk.  It issues ordinary
        // loads ONLY and the staging waves LDS-DMA ONLY: s_waitcnt vmcnt(N) counts the two kinds together but they do not
        // complete in one order, so a counted wait over a mix does not say which operations are still out (as one stream,
        // "all but the N youngest" let a chunk's DMA pass as landed while it was not -- rare garbage tiles at full size).
        // The buffer is handed over in three kd-slices: the consumers count themselves out of slices 0 and 1 on an LDS counter as
        // soon as their last fragment of the slice has arrived (half way through the slice's MFMAs), and this wave refills
        // those two under the remaining MFMAs; only the last 9 KB are written between the barriers, where everybody waits.
// A/B switches of round 3 (same-box measurements, configs[2] bench): CL_WPRIO = priority of the weight wave (3: no change),
// CL_REL_IY = row step of a kd at whose end a consumer wave counts itself out of the kd's weight slice (1: right behind the issue of
// its last fragment read -- LDS operations of a wave complete in order; 2: one row step later), CL_ONE_BARRIER = the last slice is
// handed over through its counter like the first two, and a chunk iteration has ONE workgroup barrier instead of two
// (11.93 -> 12.07 volumes/s with CL_REL_IY 1).
#ifndef CL_WPRIO
#define CL_WPRIO 0
#endif
#ifndef CL_REL_IY
#define CL_REL_IY 1
#endif
#ifndef CL_ONE_BARRIER
#define CL_ONE_BARRIER 1
#endif
        if (CL_WPRIO) __builtin_amdgcn_s_setprio(CL_WPRIO);
        cl_u32x4 wst[WIT];
        int kw_ = 0, cw = 0, g_w = 0;
        const unsigned cnt_addr = (unsigned)(G::LDS + 2 * (a.C0 + a.C1) * 8);
        if (lane < 4) reinterpret_cast<unsigned*>(lds + G::LDS + 2 * (a.C0 + a.C1) * 8)[lane] = 0u;
        auto wait_slice = [&](int sl, unsigned target) __attribute__((always_inline)) {
            for (;;) {
                unsigned v;
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(cnt_addr + 4u * sl) : "memory");
                if (__builtin_amdgcn_readfirstlane(v) >= target) break;
                __builtin_amdgcn_s_sleep(2);
            }
        };
        auto write_slice = [&](int sl) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 9; ++i) reinterpret_cast<cl_u32x4*>(wlds)[(sl * 9 + i) * 64 + lane] = wst[sl * 9 + i];
        };
        auto locate_w = [&]() __attribute__((always_inline)) { int tx, ty, tz, n0; tile_of(kw_, g_w, tx, ty, tz, n0); };
        auto issue_w = [&]() __attribute__((always_inline)) {
            const cl_u32x4* wsrc = reinterpret_cast<const cl_u32x4*>(a.wf) + ((size_t)g_w * nch + cw) * WQ + lane;
#pragma unroll
            for (int i = 0; i < WIT; ++i) wst[i] = wsrc[i * 64];
        };
        auto write_w = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < WIT; ++i) reinterpret_cast<cl_u32x4*>(wlds)[i * 64 + lane] = wst[i];
        };
        locate_w();
        CL_BARRIER();                                        // (the staging waves' table barrier)
        issue_w();
        write_w();
        CL_BARRIER();
#ifdef WMH_CL_DIAG
        unsigned w_s0 = 0, w_w0 = 0, w_s1 = 0, w_w1 = 0, w_b1 = 0;
#endif
        for (int it = 0; it < total; ++it) {
            CL_STAMP(q0);
            if (kw_ * nch + cw + 1 < total && ++cw == nch) { cw = 0; ++kw_; locate_w(); }
            issue_w();                                       // chunk it+1 (the last iteration repeats the last chunk)
            if (!CL_ABL(8)) {
                wait_slice(0, 4u * (unsigned)(it + 1));
                CL_STAMP(q1);
                write_slice(0);
                asm volatile("" ::: "memory");
                CL_STAMP(q2);
                wait_slice(1, 4u * (unsigned)(it + 1));
                CL_STAMP(q3);
                write_slice(1);
                asm volatile("" ::: "memory");
                CL_STAMP(q4);
                if (CL_ONE_BARRIER) wait_slice(2, 4u * (unsigned)(it + 1));
                else CL_BARRIER();                           // consumers are done with chunk `it`: the last slice may go
                CL_STAMP(q5);
                CL_STAMP_ADD(w_s0, q0, q1); CL_STAMP_ADD(w_w0, q1, q2); CL_STAMP_ADD(w_s1, q2, q3); CL_STAMP_ADD(w_w1, q3, q4); CL_STAMP_ADD(w_b1, q4, q5);
                write_slice(2);
            } else {
                if (!CL_ONE_BARRIER) CL_BARRIER();
                write_w();
            }
            CL_BARRIER();
        }
#if defined(WMH_CL_DIAG) && WMH_CL_DIAG == 1
        if (a.stamps && lane == 0) {
            unsigned long long* o = a.stamps + (size_t)blockIdx.x * 16;
            o[8] = w_s0; o[9] = w_w0; o[10] = w_s1; o[11] = w_w1; o[12] = w_b1;
        }
#endif
    } else if (loader) {
        // ================================ staging waves ====================================================
        if (!CL_ABL(6)) __builtin_amdgcn_s_setprio(2);       // ahead of the MFMA-issuing consumer on the same SIMD
        const int lt = tid - 256;
        const int lw = wave - 4;
        const bool last_slot = (NSLOT - 1) * LTH + lw * 64 < NVP;     // uniform: does this wave take part in the last slot?
        // this lane's NSLOT halo voxels (slot k: voxel k*LTH + lt of the halo box, z-major): position code and the offset
        // from the box's first voxel; constant over tiles
        int rel[NSLOT], code[NSLOT];
#pragma unroll
        for (int k = 0; k < NSLOT; ++k) {
            const int e = min(k * LTH + lt, NV - 1);
            const int iz = e / (IY * IX), rem = e - iz * (IY * IX), iy = rem / IX, ix = rem - iy * IX;
            rel[k] = (iz * a.H + iy) * a.W + ix;
            code[k] = (iz << 16) | (iy << 8) | ix;
        }
        const int rel_center = HW + a.W + 1;                 // the tile's first output voxel: always inside the volume
        // Three streams run over this workgroup's chunk sequence (tile-major, chunk-minor), each with its own (tile, chunk)
        // counter -- no division in the loop: DMA (chunk it+2), transform and weights (chunk it+1).
        int gv[NSLOT];                                       // DMA: voxel index inside the sample, clamped when outside the volume
        int n_dma = 0, n_tr = 0;
        unsigned keep_tr = 0;                                // transform: bit k = slot k's voxel is inside the volume
        int kd = 0, cd = 0, kt = 0, ct = 0;                  // (tile, chunk) of the DMA stream and of the transform / weight stream
        auto inside = [&](int k, int z0, int y0, int x0) __attribute__((always_inline)) {
            const int gz = z0 + (code[k] >> 16), gy = y0 + ((code[k] >> 8) & 255), gx = x0 + (code[k] & 255);
            // unsigned compares fold the two-sided range checks; bitwise & keeps it branch-free
            return (int)(k * LTH + lt < NV) & (int)((unsigned)gz < (unsigned)a.D) & (int)((unsigned)gy < (unsigned)a.H) & (int)((unsigned)gx < (unsigned)a.W);
        };
        // Scale / shift of the pending InstanceNorm: kept in LDS for the sample being staged, one float4 {scale c, scale c+1,
        // shift c, shift c+1} per channel pair (the operand pairs of the packed FMAs), so that a chunk's 16 channels are 8
        // LDS reads into vector registers -- as scalar loads they cost 32 SGPRs, and the spills that followed put ~70
        // v_readlane into every chunk iteration.  Two tables (sample parity): the DMA stream, one chunk ahead of the transform
        // and separated from it by the chunk's barriers, fills the table slot of the tile it enters when the sample changed.
        const int Ctot = a.C0 + a.C1;
        float4* tab = reinterpret_cast<float4*>(lds + 3 * BUF + WQ * 16 + G::SCR);
        auto write_table = [&](int n, int slot) __attribute__((always_inline)) {
            float4* t = tab + slot * (Ctot / 2);
            for (int i = lt; i < Ctot / 2; i += LTH) {
                const int c = 2 * i;
                const float* ss = c < a.C0 ? a.ss0 : a.ss1;
                const int Cs = c < a.C0 ? a.C0 : a.C1, cc = c < a.C0 ? c : c - a.C0;
                float4 v = make_float4(1.f, 1.f, 0.f, 0.f);
                if (ss != nullptr) { const float4 q = *reinterpret_cast<const float4*>(ss + ((size_t)n * Cs + cc) * 2); v = make_float4(q.x, q.z, q.y, q.w); }
                t[i] = v;
            }
        };
        int n_tab0 = -1, n_tab1 = -1;                        // samples the two table slots hold (slot = tile counter parity)
        auto setup_dma = [&]() __attribute__((always_inline)) {
            int g, tx, ty, tz;
            tile_of(kd, g, tx, ty, tz, n_dma);
            // uniform; a handful of times per launch.  Slot = parity of the tile counter: the transform stream is at most one
            // tile behind, so the slot of tile kd is not read any more by anyone when tile kd is entered here
            if (kd & 1) { if (n_dma != n_tab1) { n_tab1 = n_dma; write_table(n_dma, 1); } }
            else        { if (n_dma != n_tab0) { n_tab0 = n_dma; write_table(n_dma, 0); } }
            const int z0 = tz * G::TZ - 1, y0 = ty * G::TY - 1, x0 = tx * G::TX - 1;
            const int vb = (z0 * a.H + y0) * a.W + x0;
#pragma unroll
            for (int k = 0; k < NSLOT; ++k) gv[k] = vb + (inside(k, z0, y0, x0) ? rel[k] : rel_center);
        };
        auto setup_tr = [&]() __attribute__((always_inline)) {
            int g, tx, ty, tz;
            tile_of(kt, g, tx, ty, tz, n_tr);
            const int z0 = tz * G::TZ - 1, y0 = ty * G::TY - 1, x0 = tx * G::TX - 1;
            keep_tr = 0;
#pragma unroll
            for (int k = 0; k < NSLOT; ++k) keep_tr |= inside(k, z0, y0, x0) ? (1u << k) : 0u;
        };
        auto src_of = [&](int ch, const uint16_t*& xs, const float*& ss, int& Cs, int& c0) __attribute__((always_inline)) {
            const int cb = ch * 16;
            if (cb < a.C0) { xs = a.x0; ss = a.ss0; Cs = a.C0; c0 = cb; }
            else           { xs = a.x1; ss = a.ss1; Cs = a.C1; c0 = cb - a.C0; }
        };
        // chunk cd of the DMA stream's tile into buffer b: 2 * NSLOT wave-instructions of 1 KB
        auto dma = [&](int b) __attribute__((always_inline)) {
            const uint16_t* xs; const float* ss; int Cs, c0;
            src_of(cd, xs, ss, Cs, c0);
            // planes (c0 / 8) and (c0 / 8 + 1) of sample n_dma: a lane's piece is 16 bytes at 16 x its voxel index
            const char* base = reinterpret_cast<const char*>(xs) + cl_index((size_t)n_dma, Cs, DHW, c0 >> 3, 0) * 2;
            const size_t plane = DHW * 16;
            unsigned char* dst = lds + b * BUF + lw * (64 * 16);
#pragma unroll
            for (int k = 0; k < NSLOT; ++k)
                if (k < NSLOT - 1 || last_slot) {             // uniform
                    const unsigned off = (unsigned)gv[k] * 16u;
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        __builtin_amdgcn_global_load_lds(CL_GLOBAL_PTR(base + h * plane + off), CL_LDS_PTR(dst + (h * NVP + k * LTH) * 16), 16, 0, 0);
                }
        };
        // scale / shift of chunk ct of the transform stream's tile (scalar loads, issued early: they share a counter with the
        // LDS operations and return out of order, so nothing of the transform's LDS traffic may be in flight beside them)
        // The chunk's 8 table reads are inline asm, issued at the top of an iteration and waited for together with the DMA of
        // the chunk (TQ_WAIT): a compiler-visible LDS read behind LDS-DMA in flight gets an s_waitcnt vmcnt(0) in front of it
        // (hipcc cannot tell the DMA's destination from the table), which drained the prefetch at every chunk.
        typedef float f32x4v __attribute__((ext_vector_type(4)));
        f32x4v tq[8];
        float slope_tr = 1.f;
        bool pending_tr = false;
        auto load_ss = [&]() __attribute__((always_inline)) {
            const uint16_t* xs; const float* ss; int Cs, c0;
            src_of(ct, xs, ss, Cs, c0);
            pending_tr = ss != nullptr;
            slope_tr = pending_tr ? a.slope : 1.f;
            const unsigned taddr = (unsigned)(3 * BUF + WQ * 16 + G::SCR) + (unsigned)(((kt & 1) * (Ctot / 2) + ct * 8) * 16);
#define CL_TQ(i) asm volatile("ds_read_b128 %0, %1 offset:" #i "*16" : "=v"(tq[i]) : "v"(taddr) : "memory")
            CL_TQ(0); CL_TQ(1); CL_TQ(2); CL_TQ(3); CL_TQ(4); CL_TQ(5); CL_TQ(6); CL_TQ(7);
#undef CL_TQ
        };
#define CL_TQ_OPS "+v"(tq[0]), "+v"(tq[1]), "+v"(tq[2]), "+v"(tq[3]), "+v"(tq[4]), "+v"(tq[5]), "+v"(tq[6]), "+v"(tq[7])
        // InstanceNorm + LeakyReLU of the chunk in buffer b, in place: all reads first, then piece by piece
        auto transform = [&](int b) __attribute__((always_inline)) {
            unsigned char* buf = lds + b * BUF + lt * 16;
            if (pending_tr) {                                 // uniform
                float sc[2][8], sh[2][8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const f32x4v v = tq[q];
                    sc[q >> 2][2 * (q & 3)] = v[0]; sc[q >> 2][2 * (q & 3) + 1] = v[1]; sh[q >> 2][2 * (q & 3)] = v[2]; sh[q >> 2][2 * (q & 3) + 1] = v[3];
                }
                float scs[2][8], shs[2][8];
#pragma unroll
                for (int c = 0; c < 16; ++c) { scs[c >> 3][c & 7] = sc[c >> 3][c & 7] * slope_tr; shs[c >> 3][c & 7] = sh[c >> 3][c & 7] * slope_tr; }
                cl_u32x4 raw[NSLOT][2];
#pragma unroll
                for (int k = 0; k < NSLOT; ++k)
                    if (k < NSLOT - 1 || last_slot) {
#pragma unroll
                        for (int h = 0; h < 2; ++h) raw[k][h] = *reinterpret_cast<const cl_u32x4*>(buf + (h * NVP + k * LTH) * 16);
                    }
#pragma unroll
                for (int k = 0; k < NSLOT; ++k)
                    if (k < NSLOT - 1 || last_slot) {
                        const unsigned keep = ((keep_tr >> k) & 1u) ? 0xFFFFFFFFu : 0u;
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const cl_u32x4 w = cl_transform8<true>(raw[k][h], sc[h], sh[h], slope_tr);
                            *reinterpret_cast<cl_u32x4*>(buf + (h * NVP + k * LTH) * 16) = cl_u32x4{w[0] & keep, w[1] & keep, w[2] & keep, w[3] & keep};
                        }
                    }
            } else {                                          // nothing pending on this source: only the padding needs writing
#pragma unroll
                for (int k = 0; k < NSLOT; ++k)
                    if ((k < NSLOT - 1 || last_slot) && !((keep_tr >> k) & 1u)) {
#pragma unroll
                        for (int h = 0; h < 2; ++h) *reinterpret_cast<cl_u32x4*>(buf + (h * NVP + k * LTH) * 16) = cl_u32x4{0u, 0u, 0u, 0u};
                    }
            }
        };
        // advance a stream by one chunk; false when it already stands on the last chunk (the streams then repeat it into
        // buffers nobody reads any more: the loop body stays branch-free)
        auto advance = [&](int& k, int& c) __attribute__((always_inline)) -> bool {
            if (k * nch + c + 1 >= total) return false;
            if (++c == nch) { c = 0; ++k; return true; }
            return false;
        };

        // prologue: chunk 0 landed, normalised and its weights in LDS; chunk 1 in flight
        setup_dma();
        CL_BARRIER();                                        // the first sample's scale / shift table is complete
        setup_tr();
        load_ss();
        dma(0);
        if (advance(kd, cd)) setup_dma();
        dma(1);
        asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" : CL_TQ_OPS :: "memory");
        transform(0);
        CL_BARRIER();
        int b_tr = 1, b_dma = 2;                             // buffers of chunk it+1 / it+2
#ifdef WMH_CL_DIAG
        unsigned s_issue = 0, s_wait = 0, s_tr = 0, s_b1 = 0, s_w = 0, s_b2 = 0;
#endif
        for (int it = 0; it < total; ++it) {
            CL_STAMP(t0);
            if (advance(kt, ct)) setup_tr();
            load_ss();
            if (advance(kd, cd)) setup_dma();
            if (!CL_ABL(1)) dma(b_dma);
            CL_STAMP(t1);
            // everything older than this iteration's DMAs has landed, i.e. chunk it+1's DMA (issued one iteration ago; this wave
            // has nothing but LDS-DMA in flight, which completes in issue order); a lane transforms exactly the pieces it fetched
#ifdef WMH_CL_DRAIN
            asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" : CL_TQ_OPS :: "memory");
#else
            if (last_slot) asm volatile("s_waitcnt vmcnt(%8)\n\ts_waitcnt lgkmcnt(0)" : CL_TQ_OPS : "n"(2 * NSLOT) : "memory");
            else           asm volatile("s_waitcnt vmcnt(%8)\n\ts_waitcnt lgkmcnt(0)" : CL_TQ_OPS : "n"(2 * NSLOT - 2) : "memory");
#endif
            CL_STAMP(t2);
            if (!CL_ABL(3)) transform(b_tr);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            CL_STAMP(t3);
            if (!CL_ONE_BARRIER) CL_BARRIER();               // (two-barrier form: consumers are done with chunk `it`, its last weight slice may go)
            CL_STAMP(t4);
            CL_STAMP(t5);
            CL_BARRIER();
            CL_STAMP(t6);
            CL_STAMP_ADD(s_issue, t0, t1); CL_STAMP_ADD(s_wait, t1, t2); CL_STAMP_ADD(s_tr, t2, t3); CL_STAMP_ADD(s_b1, t3, t4);
            CL_STAMP_ADD(s_w, t4, t5); CL_STAMP_ADD(s_b2, t5, t6);
            b_tr = b_tr == 2 ? 0 : b_tr + 1;
            b_dma = b_dma == 2 ? 0 : b_dma + 1;
        }
#ifdef WMH_CL_DIAG
        if (a.stamps && tid == 256) {
            unsigned long long* o = a.stamps + (size_t)blockIdx.x * 16;
            o[0] = s_issue; o[1] = s_wait; o[2] = s_tr; o[3] = s_b1; o[4] = s_w; o[5] = s_b2; o[6] = (unsigned long long)total;
        }
#endif
    } else {
        // ================================ consumer waves ===================================================
        const int j = lane & 31, kh = lane >> 5;
        // B operand of (kd, input row iy, kw): voxel ((wave + kd) * IY + iy) * IX + j + kw of plane kh
        const unsigned char* bbase = lds + (kh * NVP + (wave * IY) * IX + j) * 16;
        const cl_bf16x8* wl = reinterpret_cast<const cl_bf16x8*>(wlds) + lane;
        // InstanceNorm moments: every lane keeps running sums about a fixed shift over ALL the tiles of one sample this
        // workgroup computes (a persistent workgroup stays inside a sample for many tiles); the cross-lane reduction and the
        // record (count, sum, M2) happen once per (workgroup, sample), not once per tile.  Record index = workgroup * 4 + wave;
        // the host zero-fills the record buffer, so samples a workgroup never touches read as empty records.
        // (bias and the shift live in a wave-private LDS scratch, [bias | shift][lane half][16 registers]: registers are short)
        float* scr = reinterpret_cast<float*>(lds + 3 * BUF + WQ * 16) + wave * 64;
        const unsigned cnt_addr = (unsigned)(G::LDS + 2 * (a.C0 + a.C1) * 8);
        // "this wave has every fragment of weight slice sl": one lane adds 1 to the slice's counter (LDS operations of a wave
        // complete in order, so the add lands after the wave's reads of the slice)
        auto release_slice = [&](int sl) __attribute__((always_inline)) {
            unsigned long long saved;
            asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tds_add_u32 %1, %2\n\ts_mov_b64 exec, %0"
                         : "=&s"(saved) : "v"(cnt_addr + 4u * sl), "v"(1u) : "memory");
        };
        float m_sd[16], m_sq[16];
        int m_cnt = 0, m_n = -1, m_cob = 0;
        auto flush = [&]() __attribute__((always_inline)) {
            if (a.part == nullptr || m_n < 0) return;
            float vals[32];
#pragma unroll
            for (int r = 0; r < 16; ++r) { vals[2 * r] = m_sd[r]; vals[2 * r + 1] = m_sq[r]; }
            halve_step<16>(vals, j); halve_step<8>(vals, j); halve_step<4>(vals, j); halve_step<2>(vals, j); halve_step<1>(vals, j);
            const float mine = vals[0];                       // lane j: (j&1) ? sum d^2 : sum d, of register j>>1
            const float other = __shfl_xor(mine, 1);
            const float Kr = scr[32 + kh * 16 + (j >> 1)];
            if ((j & 1) == 0) {
                const float fc = (float)m_cnt;
                const float m2 = m_cnt > 0 ? fmaxf(other - mine * mine / fc, 0.f) : 0.f;
                reinterpret_cast<float4*>(a.part)[((size_t)m_n * a.Cout + m_cob + cl_moment_channel(j, kh)) * a.npart + blockIdx.x * 4 + wave] =
                    make_float4(fc, fmaf(fc, Kr, mine), m2, 0.f);
            }
        };
        int bias_g = -1;
        CL_BARRIER();                                        // (the loaders' table barrier)
        CL_BARRIER();
        int b_cur = 0;
#if defined(WMH_CL_DIAG) && WMH_CL_DIAG == 2
        unsigned c_mfma = 0, c_b1 = 0, c_b2 = 0;             // (32-bit sums; the epilogue by difference)
        CL_CSTAMP(c_start);
        const unsigned long long r_start = __builtin_amdgcn_s_memrealtime();
#endif
        for (int k = 0; k < ntiles_mine; ++k) {
            int g, tx, ty, tz, n0;
            tile_of(k, g, tx, ty, tz, n0);
            if (g != bias_g) {                                // uniform; a workgroup's tiles share one cout group (see the launcher)
                bias_g = g;
                if (j == 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) scr[kh * 16 + r] = a.bias != nullptr ? a.bias[g * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh] : 0.f;
                }
            }
            f32x16 acc[MT];
            {
                const f32x16 b16 = *reinterpret_cast<const f32x16*>(scr + kh * 16);
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m] = b16;
            }
            for (int ch = 0; ch < nch; ++ch) {
                const unsigned char* cur = bbase + b_cur * BUF;
                CL_CSTAMP(v0);
                if (!CL_ABL(4)) {
                    // 18 row steps (kd, input row iy); a step's MFMAs are grouped by kw, and as soon as a group has been issued
                    // its B fragment is reloaded for the NEXT step -- it then has the other groups' MFMAs to land under, with no
                    // second register set.  A fragments: the three kr sets of a kd rotate the same way (a set is loaded one
                    // step before its first use, into registers whose previous contents were last used two steps earlier).
                    // sched_barrier pins the order (left alone, the compiler waits for every read right in front of its MFMA).
                    cl_bf16x8 av[3][3], bv[3];
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        av[0][kw] = wl[kw * 64];
                        bv[kw] = *reinterpret_cast<const cl_bf16x8*>(cur + kw * 16);
                    }
#pragma unroll
                    for (int kd = 0; kd < 3; ++kd)
#pragma unroll
                        for (int iy = 0; iy < IY; ++iy)
#pragma unroll
                            for (int kw = 0; kw < 3; ++kw) {
#pragma unroll
                                for (int kr = 0; kr < 3; ++kr) {
                                    const int m = iy - kr;
                                    if (m >= 0 && m < MT)
                                        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[kr][kw], bv[kw], acc[m], 0, 0, 0);
                                }
                                const int nkd = iy + 1 < IY ? kd : kd + 1, niy = iy + 1 < IY ? iy + 1 : 0;      // next step
                                if (!CL_ABL(7)) {
                                if (nkd < 3) bv[kw] = *reinterpret_cast<const cl_bf16x8*>(cur + ((nkd * IY + niy) * IX + kw) * 16);
                                if (iy + 1 <= 2) av[iy + 1][kw] = wl[(kd * 9 + (iy + 1) * 3 + kw) * 64];
                                else if (iy == IY - 1 && kd < 2) av[0][kw] = wl[((kd + 1) * 9 + kw) * 64];
                                } else if (iy + 1 <= 2) av[iy + 1][kw] = av[0][kw];
                                // all nine fragments of this kd were issued by row step 1 and have been consumed by now
                                if (iy == CL_REL_IY && kw == 2 && (kd < 2 || CL_ONE_BARRIER) && !CL_ABL(8)) release_slice(kd);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                }
                CL_CSTAMP(v1);
                if (!CL_ONE_BARRIER) CL_BARRIER();
                CL_CSTAMP(v2);
                CL_BARRIER();                                // chunk it+1 is normalised, its weights are in place, nobody reads chunk it any more
                CL_CSTAMP(v3);
                CL_CSTAMP_ADD(c_mfma, v0, v1); CL_CSTAMP_ADD(c_b1, v1, v2); CL_CSTAMP_ADD(c_b2, v2, v3);
                b_cur = b_cur == 2 ? 0 : b_cur + 1;
            }
            __builtin_amdgcn_s_setprio(3);                    // the epilogue is vector work: not behind the loaders (priority 2)
            // ---- epilogue: this wave's z-slice, MT rows of 32 voxels x 32 output channels
            const int oz = tz * G::TZ + wave, oy0 = ty * G::TY, ox = tx * G::TX + j;
            const int cob = g * 32;
            if (n0 != m_n || cob != m_cob) {                  // uniform: a new sample (or cout group): close the running record
                flush();
                m_n = n0; m_cob = cob; m_cnt = 0;
                if (j == 0) *reinterpret_cast<f32x16*>(scr + 32 + kh * 16) = acc[0];        // shift = the run's first voxel
#pragma unroll
                for (int r = 0; r < 16; ++r) { m_sd[r] = 0.f; m_sq[r] = 0.f; }
            }
            const f32x16 m_k = *reinterpret_cast<const f32x16*>(scr + 32 + kh * 16);
#pragma unroll
            for (int m = 0; m < MT; ++m)
                if (oz < a.D && oy0 + m < a.H && !CL_ABL(5)) {   // uniform
                    m_cnt += 32;
                    if (!CL_ABL(10))
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {         // two registers per packed instruction
                        const cl_f32x2 d = cl_f32x2{acc[m][r], acc[m][r + 1]} - cl_f32x2{m_k[r], m_k[r + 1]};
                        const cl_f32x2 sd = cl_f32x2{m_sd[r], m_sd[r + 1]} + d;
                        const cl_f32x2 sq = __builtin_elementwise_fma(d, d, cl_f32x2{m_sq[r], m_sq[r + 1]});
                        m_sd[r] = sd[0]; m_sd[r + 1] = sd[1]; m_sq[r] = sq[0]; m_sq[r + 1] = sq[1];
                    }
                    unsigned short* yv = a.y + cl_index((size_t)n0, a.Cout, DHW, cob >> 3, ((size_t)oz * a.H + (oy0 + m)) * a.W + ox);
                    if (!CL_ABL(9) || a.slope < -1.f) cl_store_mtile(acc[m], yv, kh, DHW * 8);      // (ablation: a predicate that is never true)
                }
            __builtin_amdgcn_s_setprio(0);
        }
        flush();
#if defined(WMH_CL_DIAG) && WMH_CL_DIAG == 2
        if (a.stamps && lane == 0) {                          // every consumer wave: [grid][4 waves][8] behind the staging waves' [grid][16]
            unsigned long long* o = a.stamps + (size_t)gridDim.x * 16 + ((size_t)blockIdx.x * 4 + wave) * 8;
            CL_CSTAMP(c_end);
            o[0] = c_mfma; o[1] = c_b1; o[2] = c_b2; o[3] = (unsigned)(c_end - c_start) - c_mfma - c_b1 - c_b2;
            o[4] = (unsigned long long)total;
            o[5] = c_end - c_start; o[6] = __builtin_amdgcn_s_memrealtime() - r_start;   // shader-clock ticks / 100 MHz ticks of the whole run
        }
#endif
    }
}

typedef ClGeo<4, 4> ClGeo44;

static int g_cl_ncu = 0;
static int cl_num_cus() {
    if (!g_cl_ncu) {
        int dev = 0, n = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        if (n < 8) n = 8;
        g_cl_ncu = n & ~7;
    }
    return g_cl_ncu;
}
static inline int cl_cdiv(int a, int b) { return (a + b - 1) / b; }
static inline unsigned cl_magic(int d) { return (unsigned)((((unsigned long long)1 << 32) + (unsigned)d - 1) / (unsigned)d); }

// InstanceNorm partial records per (n, c) row this kernel writes: one per consumer wave of every (persistent) workgroup.
// The caller ZERO-FILLS the record buffer before every launch (a workgroup writes only the samples it computed tiles of).
extern "C" int wmh_cl_conv3_num_partials(int D, int H, int W) {
    if (D < 1 || H < 1 || W < 32 || W % 32) return wmh_fail(WMH_EINVAL, "cl_conv3: rows must be a multiple of 32 voxels");
    return cl_num_cus() * 4;
}

// y[n][z][y][x][Cout] (bf16) = conv3x3x3(lrelu(x * scale + shift)) + bias over one or two CL16 sources (the concatenation
// torch.cat((x0, x1), 1) of nnU-Net's decoder is never formed).  wf: the bf16 fragment image of
// wmh_conv3d_k3_pack_weights_bf16(cfg 0).  partials: [N][Cout][wmh_cl_conv3_num_partials][4] fp32 moment records,
// zero-filled here (stream ordered) before the launch.
extern "C" int wmh_cl_conv3_fwd(const void* x0, const float* ss0, int C0, const void* x1, const float* ss1, int C1,
                                const uint16_t* wf, const float* bias, void* y, float* partials,
                                int N, int D, int H, int W, int Cout, float lrelu_slope, void* stream) {
    if (!x0 || !wf || !y || N < 1 || D < 1 || H < 1) return wmh_fail(WMH_EINVAL, "cl_conv3_fwd: null/empty argument");
    if (x1 == nullptr) C1 = 0;
    if (C0 < 16 || C0 % 16 || C1 % 16 || Cout % 32 || W % 32)
        return wmh_fail(WMH_EINVAL, "cl_conv3_fwd: channels must be multiples of 16 (in) / 32 (out), rows multiples of 32 voxels");
    if (!(lrelu_slope >= 0.f && lrelu_slope <= 1.f)) return wmh_fail(WMH_EINVAL, "cl_conv3_fwd: LeakyReLU slope must be in [0, 1]");
    const size_t dhw = (size_t)D * H * W;
    if (dhw * 16 >= ((size_t)1 << 32)) return wmh_fail(WMH_EINVAL, "cl_conv3_fwd: an 8-channel plane exceeds 4 GB");
    if ((reinterpret_cast<uintptr_t>(x0) | reinterpret_cast<uintptr_t>(x1) | reinterpret_cast<uintptr_t>(y)) & 15)
        return wmh_fail(WMH_EINVAL, "cl_conv3_fwd: tensors must be 16-byte aligned");
    ClConvArgs a;
    a.x0 = static_cast<const uint16_t*>(x0); a.ss0 = ss0; a.C0 = C0;
    a.x1 = static_cast<const uint16_t*>(x1); a.ss1 = ss1; a.C1 = C1;
    a.wf = wf; a.bias = bias; a.y = static_cast<uint16_t*>(y); a.part = partials;
    a.N = N; a.D = D; a.H = H; a.W = W; a.Cout = Cout;
    a.nchunks = (C0 + C1) / 16;
    a.ntx = W / 32; a.nty = cl_cdiv(H, ClGeo44::TY); a.ntz = cl_cdiv(D, ClGeo44::TZ);
    a.G = Cout / 32;
    a.xcd_swizzle = 1;
    a.slope = lrelu_slope;
#ifdef WMH_CL_DIAG
    a.stamps = g_cl_stamps;
#endif
    const long long T = (long long)a.ntx * a.nty * a.ntz * N * a.G;
    int dmax = a.G > a.ntx ? a.G : a.ntx;
    dmax = dmax > a.nty ? dmax : a.nty; dmax = dmax > a.ntz ? dmax : a.ntz;
    if (T * dmax >= ((long long)1 << 32)) return wmh_fail(WMH_EINVAL, "cl_conv3_fwd: too many tiles");     // range of the reciprocal division
    a.mG = cl_magic(a.G); a.mtx = cl_magic(a.ntx); a.mty = cl_magic(a.nty); a.mtz = cl_magic(a.ntz);
    const int ncu = cl_num_cus();
    a.npart = ncu * 4;
    // A workgroup's tiles are blockIdx.x + k * grid, i.e. (XCD-aware order) tile ids a multiple of grid/8 apart: with grid/8 a
    // multiple of G every workgroup stays in ONE cout group, and since the sample index never decreases along its tiles, the
    // (sample, cout group) runs its running moments are kept over are contiguous -- a record is written exactly once.
    int per_xcd = ncu / 8;
    if (per_xcd >= a.G) per_xcd -= per_xcd % a.G;
    else if (T > ncu) return wmh_fail(WMH_EINVAL, "cl_conv3_fwd: more cout groups than workgroups per XCD");
    const int grid = T <= (long long)per_xcd * 8 ? (int)T : per_xcd * 8;
    hipStream_t st = (hipStream_t)stream;
    if (partials != nullptr && hipMemsetAsync(partials, 0, (size_t)N * Cout * a.npart * 16, st) != hipSuccess)
        return wmh_fail(WMH_ELAUNCH, "cl_conv3_fwd: clearing the moment records failed");
    const int lds_bytes = ClGeo44::LDS + 2 * (C0 + C1) * 8 + 16;           // + the weight-slice counters
    if (lds_bytes > 160 * 1024) return wmh_fail(WMH_EINVAL, "cl_conv3_fwd: too many input channels for the LDS scale/shift tables");
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cl_conv3_kernel<ClGeo44>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipLaunchKernelGGL((cl_conv3_kernel<ClGeo44>), dim3(grid), dim3(768), lds_bytes, st, a, (int)T);
    return wmh_check_launch("cl_conv3_kernel");
}

// ----------------------------------------------------------------------------------------------------------------
// Stride 2 (the first convolution of an encoder stage).  Eight input voxels per output voxel: the layer streams its input
// (it is bandwidth / staging bound, the matrix work is 1/8 of a stride-1 layer's), so the structure is the simple one:
// a 256-thread workgroup per tile of 2 x 2 x 32 outputs, two workgroups per CU hiding each other's DMA latency; per
// 16-channel chunk: LDS-DMA of the 5 x 5 x 65 input box and of the weight fragments, in-place InstanceNorm + LeakyReLU,
// 27 MFMAs per wave (one M-tile each).  The box's rows are stored parity-split (even x positions, then odd ones), so the
// stride-2 operand reads of consecutive lanes are consecutive 16-byte slots again (conflict-free).
// ----------------------------------------------------------------------------------------------------------------
struct ClConvS2Args {
    const uint16_t* x; const float* ss; int Cin;
    const uint16_t* wf; const float* bias; uint16_t* y; float* part;
    int N, D, H, W, Do, Ho, Wo, Cout;
    int nchunks, ntx, nty, ntz, npart, G;
    float slope;
};
struct ClS2Geo {
    static constexpr int IZ = 5, IY = 5, IX = 65, NE = 33;        // NE: even positions of a row (0, 2, .., 64)
    static constexpr int NV = IZ * IY * IX, NVP = 1664, NSLOT = 7;  // NVP: multiple of 128 >= NV; slot 6 exists for waves 0-1
    static constexpr int ACT = 2 * NVP * 16, WQ = 27 * 64;
    static constexpr int LDS = ACT;                               // the weights do not pass through LDS here (see the kernel)
    static_assert(3 * LDS <= 160 * 1024, "three workgroups per CU");
};

// GW: groups of 32 output channels a workgroup computes from ONE staged box (1, 2 or 4).  Wave w owns output-channel group
// w % GW for the MT = GW output rows [w / GW * MT, +MT) of the 2 x 2 x 32 tile: the box is staged and transformed once for all
// GW groups (not once per group), and a weight fragment is loaded by 4 / GW waves instead of by all four -- the kernel is bound
// by the L2 -> CU traffic of exactly these two streams.
template <int GW>
__global__ __launch_bounds__(256, GW == 4 ? 2 : 3) void cl_conv3s2_kernel(const ClConvS2Args a) {
    using G = ClS2Geo;
    constexpr int MT = GW;
    constexpr int IY = G::IY, IX = G::IX, NE = G::NE, NV = G::NV, NVP = G::NVP, NSLOT = G::NSLOT, WQ = G::WQ;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, kh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int t = blockIdx.x;
    const int g = (t % a.G) * GW + wave % GW; t /= a.G;              // a.G: workgroups per tile = Cout / (32 GW)
    const int m0 = wave / GW * MT;                                   // first of this wave's output rows (row m: z = m >> 1, y = m & 1)
    const int tx = t % a.ntx; t /= a.ntx;
    const int ty = t % a.nty; t /= a.nty;
    const int tz = t % a.ntz; t /= a.ntz;
    const int n = t;
    const int HW = a.H * a.W;
    const size_t DHW = (size_t)a.D * HW;
    const int z0 = 4 * tz - 1, y0 = 4 * ty - 1, x0 = 64 * tx - 1;       // first voxel of the input box
    const int center = ((z0 + 1) * a.H + y0 + 1) * a.W + x0 + 1;         // always inside the volume
    int gv[NSLOT]; unsigned keep = 0;
#pragma unroll
    for (int k = 0; k < NSLOT; ++k) {
        const int e = k * 256 + tid;
        const int ec = min(e, NV - 1);
        const int iz = ec / (IY * IX), rem = ec - iz * (IY * IX), iy = rem / IX, pos = rem - iy * IX;
        const int ix = pos;      // rows in natural order: consecutive DMA lanes fetch consecutive voxels (half the cache lines per
                                 // instruction of an even/odd split); the stride-2 operand reads then are 2-way bank conflicts, and
                                 // there are only 27 of them per chunk
        const int gz = z0 + iz, gy = y0 + iy, gx = x0 + ix;
        const int ok = (int)(e < NV) & (int)((unsigned)gz < (unsigned)a.D) & (int)((unsigned)gy < (unsigned)a.H) & (int)((unsigned)gx < (unsigned)a.W);
        gv[k] = ok ? (gz * a.H + gy) * a.W + gx : center;
        keep |= ok ? (1u << k) : 0u;
    }
    f32x16 acc[MT];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float b = a.bias != nullptr ? a.bias[g * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh] : 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[i][r] = b;
    }
    const char* xbase = reinterpret_cast<const char*>(a.x) + cl_index((size_t)n, a.Cin, DHW, 0, 0) * 2;
    const size_t plane = DHW * 16;
    const unsigned char* bbase = lds + (kh * NVP + 2 * j) * 16;
    static __device__ const float ident2s[2] = {1.f, 0.f};
    const bool has_ss = a.ss != nullptr;
    const float slope = has_ss ? a.slope : 1.f;
    for (int ch = 0; ch < a.nchunks; ++ch) {
        if (ch > 0) CL_BARRIER();                                        // the previous chunk's operand reads are done
        const char* src = xbase + (size_t)(2 * ch) * plane;
        unsigned char* dst = lds + wave * (64 * 16);
#pragma unroll
        for (int k = 0; k < NSLOT; ++k)
            if (k < NSLOT - 1 || wave < 2) {                             // uniform
                const unsigned off = (unsigned)gv[k] * 16u;
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    __builtin_amdgcn_global_load_lds(CL_GLOBAL_PTR(src + h * plane + off), CL_LDS_PTR(dst + (h * NVP + k * 256) * 16), 16, 0, 0);
            }
        // A operands straight from global memory (the layer's whole weight image is a few hundred KB and every workgroup
        // reads it: L2 hits): with 27 MFMAs per chunk their latency is cheap, and without a 27 KB weight buffer three
        // workgroups fit a CU and hide each other's DMA latency.  kd = 0, 1 now (they land under the transform), kd = 2 later.
        const cl_bf16x8* wg = reinterpret_cast<const cl_bf16x8*>(a.wf) + ((size_t)g * a.nchunks + ch) * WQ + lane;
        cl_bf16x8 av[18];
#pragma unroll
        for (int t = 0; t < 18; ++t) av[t] = wg[t * 64];
        const float* tab = has_ss ? a.ss + ((size_t)n * a.Cin + 16 * ch) * 2 : ident2s;
        const int mul = has_ss ? 2 : 0;
        float sc[2][8], sh[2][8];
#pragma unroll
        for (int c = 0; c < 16; ++c) { sc[c >> 3][c & 7] = cl_ld_uniform(tab + mul * c); sh[c >> 3][c & 7] = cl_ld_uniform(tab + mul * c + 1); }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned char* buf = lds + tid * 16;
#pragma unroll
        for (int k = 0; k < NSLOT; ++k)
            if (k < NSLOT - 1 || wave < 2) {
                const unsigned kp = ((keep >> k) & 1u) ? 0xFFFFFFFFu : 0u;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    cl_u32x4* p = reinterpret_cast<cl_u32x4*>(buf + (h * NVP + k * 256) * 16);
                    const cl_u32x4 w = cl_transform8(*p, sc[h], sh[h], slope);
                    *p = cl_u32x4{w[0] & kp, w[1] & kp, w[2] & kp, w[3] & kp};
                }
            }
        CL_BARRIER();
        cl_bf16x8 av2[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) av2[t] = wg[(18 + t) * 64];
#pragma unroll
        for (int tap = 0; tap < 27; ++tap) {
            const int kd = tap / 9, kr = (tap / 3) % 3, kw = tap % 3;
            const cl_bf16x8 aw = tap < 18 ? av[tap < 18 ? tap : 0] : av2[tap >= 18 ? tap - 18 : 0];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int zz = (m0 + i) >> 1, yy = (m0 + i) & 1;
                const cl_bf16x8 bv = *reinterpret_cast<const cl_bf16x8*>(bbase + (((2 * zz + kd) * IY + 2 * yy + kr) * IX + kw) * 16);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aw, bv, acc[i], 0, 0, 0);
            }
        }
    }
    // ---- epilogue
    const int ox = 32 * tx + j, cob = g * 32;
    unsigned vmask = 0;
#pragma unroll
    for (int i = 0; i < MT; ++i) vmask |= (2 * tz + ((m0 + i) >> 1) < a.Do && 2 * ty + ((m0 + i) & 1) < a.Ho) ? (1u << i) : 0u;
    if (a.part != nullptr) {
        const float4 rec = cl_wave_moments<MT>(acc, vmask, 32 * __builtin_popcount(vmask), j, kh);
        CL_BARRIER();                                                    // operand reads done: the activation area is free
        float4* scr = reinterpret_cast<float4*>(lds);
        if ((j & 1) == 0) scr[wave * 32 + cl_moment_channel(j, kh)] = rec;
        CL_BARRIER();
        if (wave < GW && (j & 1) == 0) {                                 // the waves of one channel group, in fixed order
            const int c = cl_moment_channel(j, kh);
            float4 m = scr[wave * 32 + c];
#pragma unroll
            for (int w = wave + GW; w < 4; w += GW) m = cl_merge_moments(m, scr[w * 32 + c]);
            reinterpret_cast<float4*>(a.part)[((size_t)n * a.Cout + cob + c) * a.npart + (tz * a.nty + ty) * a.ntx + tx] = m;
        }
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
        if ((vmask >> i) & 1u) {
            const int oz = 2 * tz + ((m0 + i) >> 1), oy = 2 * ty + ((m0 + i) & 1);
            cl_store_mtile(acc[i], a.y + cl_index((size_t)n, a.Cout, (size_t)a.Do * a.Ho * a.Wo, cob >> 3, ((size_t)oz * a.Ho + oy) * a.Wo + ox), kh,
                           (size_t)a.Do * a.Ho * a.Wo * 8);
        }
}

extern "C" int wmh_cl_conv3s2_num_partials(int Do, int Ho, int Wo) {
    if (Do < 1 || Ho < 1 || Wo < 32 || Wo % 32) return wmh_fail(WMH_EINVAL, "cl_conv3s2: output rows must be a multiple of 32 voxels");
    return (Wo / 32) * cl_cdiv(Ho, 2) * cl_cdiv(Do, 2);
}
extern "C" int wmh_cl_conv3s2_fwd(const void* x, const float* ss, int Cin, const uint16_t* wf, const float* bias, void* y,
                                  float* partials, int N, int D, int H, int W, int Cout, float lrelu_slope, void* stream) {
    if (!x || !wf || !y || N < 1 || D < 1 || H < 1) return wmh_fail(WMH_EINVAL, "cl_conv3s2_fwd: null/empty argument");
    if (Cin < 16 || Cin % 16 || Cout % 32 || W % 64)
        return wmh_fail(WMH_EINVAL, "cl_conv3s2_fwd: Cin % 16, Cout % 32 and input rows that are multiples of 64 voxels");
    if (!(lrelu_slope >= 0.f && lrelu_slope <= 1.f)) return wmh_fail(WMH_EINVAL, "cl_conv3s2_fwd: LeakyReLU slope must be in [0, 1]");
    if ((size_t)D * H * W * 16 >= ((size_t)1 << 32)) return wmh_fail(WMH_EINVAL, "cl_conv3s2_fwd: an 8-channel plane exceeds 4 GB");
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(wf)) & 15)
        return wmh_fail(WMH_EINVAL, "cl_conv3s2_fwd: tensors must be 16-byte aligned");
    ClConvS2Args a;
    a.x = static_cast<const uint16_t*>(x); a.ss = ss; a.Cin = Cin; a.wf = wf; a.bias = bias;
    a.y = static_cast<uint16_t*>(y); a.part = partials;
    a.N = N; a.D = D; a.H = H; a.W = W; a.Do = (D + 1) / 2; a.Ho = (H + 1) / 2; a.Wo = W / 2; a.Cout = Cout;
    a.nchunks = Cin / 16;
    a.ntx = a.Wo / 32; a.nty = cl_cdiv(a.Ho, 2); a.ntz = cl_cdiv(a.Do, 2);
    a.npart = a.ntx * a.nty * a.ntz;
    const int G = Cout / 32, GW = G % 4 == 0 ? 4 : G % 2 == 0 ? 2 : 1;   // channel groups per workgroup
    a.G = G / GW;
    a.slope = lrelu_slope;
    const long long T = (long long)a.ntx * a.nty * a.ntz * N * a.G;
    if (T > 0x7fffffffLL) return wmh_fail(WMH_EINVAL, "cl_conv3s2_fwd: too many tiles");
    auto launch = [&](auto kern) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, ClS2Geo::LDS);
        hipLaunchKernelGGL(kern, dim3((unsigned)T), dim3(256), ClS2Geo::LDS, (hipStream_t)stream, a);
    };
    if (GW == 4) launch(&cl_conv3s2_kernel<4>); else if (GW == 2) launch(&cl_conv3s2_kernel<2>); else launch(&cl_conv3s2_kernel<1>);
    return wmh_check_launch("cl_conv3s2_kernel");
}
