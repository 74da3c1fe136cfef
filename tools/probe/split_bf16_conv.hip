// Prototype for VERDICT r2 item 5 (tuning aid, NOT product code, not part of libkanconv): the forward of ONE real layer shape -- KAN-VGG11's
// 256 -> 256 @ 8x8, 3x3 / pad 1, batch 256, default B-spline basis (grid 5, order 3, SiLU base branch: P = 9 planes, GEMM depth 20736) -- as a
// SPLIT-PRECISION contraction: every fp32 operand (expanded plane value, weight) is cut into three bf16 pieces hi + mid + lo (24 mantissa bits),
// and each 16-deep k-block runs six v_mfma_f32_32x32x16_bf16 products (lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi) into the fp32 accumulator.
// A/B in one process against the shipped exact-fp32 kernel (libkanconv's kan_conv_fwd through the C ABI, k_conv_fwd_halo on this shape):
//   * error of both against an fp64 evaluation of the layer (host, 6 of the 256 images, all outputs),
//   * time per launch of both (HIP events, 20 launches after 3 warm-up).
//
// Kernel design (k_split_fwd), following the halo forward of the library:
//   tile    128 outputs x 128 pixels (two whole 8x8 images) per workgroup, 256 workgroups (one per CU); 512 threads = 4 MFMA waves (wave tile 64 x 64,
//           2 x 2 blocks of 32 x 32) + 4 PRODUCER waves, one of each per SIMD;
//   B side  per group of 8 input channels the producers expand the 2 x 64 input values once (SiLU + 8 B-spline planes, fp32 vector ALU), cut each
//           plane into 3 bf16 pieces (v_cvt_pk_bf16_f32 on channel pairs) and write a zero-bordered halo tile per image into LDS, plane-major inside a
//           cell: the 8 channels of one plane are 16 contiguous bytes = one lane's share of a 16-deep MFMA operand (k = 8 (lane >> 5) + j), so every tap
//           reads the SAME tile through a shifted address with one ds_read_b128 per piece and block.  Cell = 9 chunks of 16 B (odd: 8 consecutive pixels
//           hit 8 different 16-byte slots), rows of 88 chunks (= 8 mod 16: the four lane groups of ds_read_b128 each cover all 16 slots; the right border
//           cell of a row overlaps the left border cell of the next -- both zero, never written).  The next group's pieces are computed into registers
//           WHILE the current group is contracted and written in the last step of the group (the MFMA waves hold that step's operands in registers);
//   depth   per channel group 81 (tap, plane) k-groups of 8 (+ 3 zero groups = 42 steps of 16); lanes 0-31 and 32-63 of a step read DIFFERENT
//           k-groups (own shift each, a compile-time constant per step: the 21 step pairs of a group are unrolled);
//   A side  weights pre-cut on the device into [step][piece][k-half][output][8 bf16] and streamed by 16-byte LDS-DMA into a ring of six one-step
//           buffers (12 KB per step) by the PRODUCER waves, six steps ahead; one barrier per TWO steps (48 MFMAs per wave), entered by the producers
//           under a counted s_waitcnt vmcnt(6) and issued as a bare s_barrier (__syncthreads() drains vmcnt to 0: the L2 latency of the newest copies
//           then sits on the critical path at any ring depth);
//   loop    explicit ISA: the 24 MFMAs of a step with the 12 ds_read_b128 of the NEXT step placed one per MFMA gap (a burst of 12 reads before the
//           MFMAs lets the matrix pipe drain; reads-then-MFMAs without overlap doubled the step time with one MFMA wave per SIMD).
// Measured steps of the design on MI355X (cycles per 16-deep step from s_memtime stamps, 768 = 24 MFMAs back to back): serial expansion by the MFMA
// waves 1600; producer waves 1270; reads interleaved 1079; counted vmcnt + bare barrier 992; DMA issued by the producers 903 (776 without the expansion
// work, i.e. the loop itself is at the matrix-pipe floor; the rest is the producers' vector work next to the MFMAs, and the clock falls from 2.2 to
// 1.8 GHz as the loop gets denser).
//
//   hipcc -O3 --offload-arch=gfx950 -I include -I convolutional-kan-for-image-classification_amd/csrc tools/probe/split_bf16_conv.hip \
//         -L convolutional-kan-for-image-classification_amd -lkanconv -Wl,-rpath,'$ORIGIN/../../convolutional-kan-for-image-classification_amd' -o tools/probe/split_bf16_conv
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include "kan_device.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define BF(v) __builtin_bit_cast(bf16x8, v)
#define BAR_PLAIN() asm volatile("s_barrier" ::: "memory")
#define BAR_LDS() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e)); exit(2); } } while (0)

constexpr int NB = 256, NC = 256, NO = 256, HH = 8, HW = 64, NP = 9, CG = 8, NCG = NC / CG, NGRP = 84, NSTEP = NGRP / 2;
#ifndef X_ROWC
#define X_ROWC 88
#endif
constexpr int CELLC = 9, ROWC = X_ROWC, IMGC = 10 * ROWC;              // chunks (16 B) per cell / halo row / image.  A row holds 9 cells + 7 pad chunks: the right
// border cell of a row IS (overlaps) the left border cell of the next -- both are zero and never written -- which brings the row to 88 chunks = 8 mod 16
constexpr int SPLITB = (2 * IMGC + 2) * 16, HALOB = 3 * SPLITB;      // bytes per piece (two images + the overhang of the last border cell), per halo tile
constexpr int WSTEP_G = 3 * 2 * NO * 16;                             // bytes of one step in the cut weights (all 256 outputs)
constexpr int WSLOT = 3 * 2 * 128 * 16, NBUF = 6;                    // one step of one 128-output tile in LDS
constexpr int LDS_BYTES = HALOB + NBUF * WSLOT + NGRP * 4 + 64;
constexpr int TOTAL_STEPS = NCG * NSTEP;

__device__ constexpr int off_of(int gi) { return gi < 81 ? (((gi / 9) / 3 - 1) * ROWC + ((gi / 9) % 3 - 1) * CELLC + gi % 9) * 16 : 0; }
__device__ __host__ inline int mfma_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

__device__ inline void split3(float v, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)v; const float r1 = v - (float)h;
    m = (__bf16)r1; const float r2 = r1 - (float)m;
    l = (__bf16)r2;
}

// ---- weights: reference layout (base [O][C][3][3], spline [O][C*8][3][3], channel c*8+k) -> wc[cg][step][piece][k-half][o][8 bf16]
__global__ void k_cut_weights(const float* __restrict__ wb, const float* __restrict__ ws, __bf16* __restrict__ wc) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;             // (cg, step, kh, o)
    if (idx >= NCG * NSTEP * 2 * NO) return;
    const int o = idx % NO, kh = (idx / NO) & 1, st = (idx / (2 * NO)) % NSTEP, cg = idx / (2 * NO * NSTEP);
    const int gi = 2 * st + kh;
    bf16x8 h, m, l;
    for (int j = 0; j < 8; ++j) {
        float v = 0.f;
        if (gi < 81) {
            const int tap = gi / 9, p = gi % 9, c = cg * CG + j;
            v = p == 0 ? wb[((size_t)o * NC + c) * 9 + tap] : ws[((size_t)o * NC * 8 + c * 8 + (p - 1)) * 9 + tap];
        }
        __bf16 a, b, d; split3(v, a, b, d); h[j] = a; m[j] = b; l[j] = d;
    }
    const size_t step = (size_t)cg * NSTEP + st;
    bf16x8* dst = (bf16x8*)wc;
    dst[((step * 3 + 0) * 2 + kh) * NO + o] = h;
    dst[((step * 3 + 1) * 2 + kh) * NO + o] = m;
    dst[((step * 3 + 2) * 2 + kh) * NO + o] = l;
}

// ---- the forward
__global__ __launch_bounds__(512, 1) void k_split_fwd(const float* __restrict__ x, const __bf16* __restrict__ wc, float* __restrict__ z, DevBasis bs, unsigned long long* __restrict__ stamps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sH = smem;
    unsigned char* sW = smem + HALOB;
    int* sOff = (int*)(smem + HALOB + NBUF * WSLOT);
    float* sTab = (float*)(sOff + NGRP);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w_o = wave & 1, w_p = wave >> 1, kh = lane >> 5, m = lane & 31;
    const int ot = blockIdx.x & 1, b0 = (blockIdx.x >> 1) * 2;

    for (int i = tid; i < HALOB / 16; i += 512) ((uint4*)sH)[i] = uint4{0u, 0u, 0u, 0u};
    if (tid < NGRP) {
        int off = 0;
        if (tid < 81) { const int tap = tid / 9, p = tid % 9, dr = tap / 3 - 1, dc = tap % 3 - 1; off = (dr * ROWC + dc * CELLC + p) * 16; }
        sOff[tid] = off;
    }
    if (tid < 16) sTab[tid] = bs.tab[tid];

    // weight stream: 3 chunks of 16 B per thread and step
    const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(wc), 0, (int)((size_t)TOTAL_STEPS * WSTEP_G), 0x00020000);
    unsigned voff[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) { const int q = (tid & 255) + 256 * j, sk = q >> 7, o = q & 127; voff[j] = (unsigned)((sk * NO + ot * 128 + o) * 16); }
    auto issue2 = [&](int t, int slot) {
        unsigned char* dst = sW + (slot % NBUF) * WSLOT + (wave & 3) * 1024;
        const int so = __builtin_amdgcn_readfirstlane(t * WSTEP_G);
#pragma unroll
        for (int j = 0; j < 3; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, (__attribute__((address_space(3))) void*)(dst + j * 4096), 16, (int)voff[j], so, 0, 0);
    };

    // operand addresses (bytes from smem)
    unsigned bBase[2];
#pragma unroll
    for (int bj = 0; bj < 2; ++bj) {
        const int lp = bj * 32 + m, r = lp >> 3, c = lp & 7;
        bBase[bj] = (unsigned)(((w_p * 10 + r + 1) * ROWC + (c + 1) * CELLC) * 16);
    }
    const unsigned aLane = (unsigned)(HALOB + (kh * 128 + w_o * 64 + m) * 16);

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // expansion: thread = (pixel q of the 128, half hh of the channel group)
    const int q = tid & 127, hh = (tid >> 7) & 1, qi = q >> 6, lp = q & 63;
    const unsigned cellB = (unsigned)(((qi * 10 + (lp >> 3) + 1) * ROWC + ((lp & 7) + 1) * CELLC) * 16 + hh * 8);
    const float* xq = x + ((size_t)(b0 + qi) * NC + hh * 4) * HW + lp;

    // PRODUCER waves (4 .. 7, one per SIMD next to an MFMA wave): the hardware issues their vector work in the MFMA waves' gaps.  They compute the
    // next channel group's pieces into registers while the current group is contracted, and write them once the halo tile is free.
    struct Pieces { uint2 h[NP], m[NP], l[NP]; };          // per plane: the 4 channels' pieces, packed bf16 pairs (ch0 ch1 | ch2 ch3)
    // two values -> three packed bf16 pairs (v_cvt_pk_bf16_f32 rounds to nearest even; a bf16 widens to fp32 by a shift / mask)
    auto split_pair = [&](float v0, float v1, unsigned& h, unsigned& m, unsigned& l) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        auto pk = [](float a, float b) { f32x2 f = {a, b}; return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2)); };
        h = pk(v0, v1);
        const float r0 = v0 - __builtin_bit_cast(float, h << 16), r1 = v1 - __builtin_bit_cast(float, h & 0xffff0000u);
        m = pk(r0, r1);
        l = pk(r0 - __builtin_bit_cast(float, m << 16), r1 - __builtin_bit_cast(float, m & 0xffff0000u));
    };
    auto compute = [&](int cg, Pieces& pc) {
#ifdef X_NO_EXPAND
        return;
#endif
        float v[4][NP];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float xv = xq[(size_t)(cg * CG + j) * HW];
            v[j][0] = xv * kan_rcp(1.0f + kan_exp2k(xv, -1.44269504088896340736f));      // SiLU through hardware exp2 / rcp, as the library's fast specs
            int j0 = 0; float N[4];
            const bool ok = bspline_uniform<false>(3, xv, sTab, 12, bs.inv_h, j0, N);
            const int e = ok ? -j0 : 64;                     // plane p holds basis p - 1: N[p - 1 + e] where that index is 0..3, else zero
            bool mk[11];
#pragma unroll
            for (int k = 0; k < 11; ++k) mk[k] = e == k - 7;
#pragma unroll
            for (int p = 1; p < NP; ++p) {
                float val = 0.f;
                val = mk[8 - p] ? N[0] : val; val = mk[9 - p] ? N[1] : val; val = mk[10 - p] ? N[2] : val;
                if (11 - p <= 10) val = mk[11 - p] ? N[3] : val;
                v[j][p] = val;
            }
        }
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            split_pair(v[0][p], v[1][p], pc.h[p].x, pc.m[p].x, pc.l[p].x);
            split_pair(v[2][p], v[3][p], pc.h[p].y, pc.m[p].y, pc.l[p].y);
        }
    };
    auto write = [&](const Pieces& pc) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            *(uint2*)(sH + cellB + p * 16) = pc.h[p];
            *(uint2*)(sH + SPLITB + cellB + p * 16) = pc.m[p];
            *(uint2*)(sH + 2 * SPLITB + cellB + p * 16) = pc.l[p];
        }
    };
    if (tid >= 256) {
        // producers also stream the weights (the MFMA waves issue no vector-memory instruction at all in the main loop): steps t+6, t+7 go out at the
        // barrier of pair t, and that barrier is entered only when all but the six newest copies (steps t+4, t+5) have landed
        Pieces pc;
        for (int i = 0; i < NBUF; ++i) issue2(i, i);
        __syncthreads();                                   // B1: zero fill, tables
        compute(0, pc); write(pc);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                   // B2
        int t = 0;
#pragma unroll 1
        for (int cg = 0; cg < NCG; ++cg) {
            const bool more = cg + 1 < NCG;
            if (more) compute(cg + 1, pc);
#pragma unroll
            for (int pr = 0; pr < NSTEP / 2; ++pr) {
                if (t + 4 < TOTAL_STEPS) asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
#ifndef X_NO_DMA
                if (t + 6 < TOTAL_STEPS) { issue2(t + 6, 2 * pr); issue2(t + 7, 2 * pr + 1); }
#endif
                t += 2;
            }
            if (more) { write(pc); BAR_LDS(); }
        }
        return;
    }
    // Operand fetch in explicit ISA (as the library's kernels): the 12 ds_read_b128 of step t+1 are issued BEFORE the 24 MFMAs of step t, so the
    // LDS phase of the four waves (one per SIMD) hides behind the matrix phase instead of alternating with it.
    struct Frag { f32x4 a[3][2], b[3][2]; };
#define DSR(d, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(d) : "v"(addr), "n"(imm) : "memory")
#define FRAG_REGS(f) "+v"(f.a[0][0]), "+v"(f.a[0][1]), "+v"(f.a[1][0]), "+v"(f.a[1][1]), "+v"(f.a[2][0]), "+v"(f.a[2][1]), \
                     "+v"(f.b[0][0]), "+v"(f.b[0][1]), "+v"(f.b[1][0]), "+v"(f.b[1][1]), "+v"(f.b[2][0]), "+v"(f.b[2][1])
    auto loadF = [&](Frag& f, int tt, int st) {
#ifdef X_NO_LOADS
        if (tt > 1) return;
#endif
        const unsigned ao = aLane + (unsigned)((tt % NBUF) * WSLOT);
        const unsigned off = (unsigned)(kh ? off_of(2 * st + 1) : off_of(2 * st));
        const unsigned p0 = bBase[0] + off, p1 = bBase[1] + off, q0 = p0 + 2 * SPLITB, q1 = p1 + 2 * SPLITB;
        DSR(f.a[0][0], ao, 0);    DSR(f.a[0][1], ao, 512);
        DSR(f.b[0][0], p0, 0);    DSR(f.b[0][1], p1, 0);
        DSR(f.a[1][0], ao, 4096); DSR(f.a[1][1], ao, 4608);
        DSR(f.b[1][0], p0, SPLITB); DSR(f.b[1][1], p1, SPLITB);
        DSR(f.a[2][0], ao, 8192); DSR(f.a[2][1], ao, 8704);
        DSR(f.b[2][0], q0, 0);    DSR(f.b[2][1], q1, 0);
    };
    // One 16-deep step in explicit ISA: 24 MFMAs (product-major, smallest products first: every accumulator sees lo*hi, hi*lo, mid*mid, mid*hi, hi*mid,
    // hi*hi in that order), with the 12 operand reads of the NEXT step and the weight DMA of a later pair placed one per MFMA gap.
#define MF(i, j, A, Bv) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(A), "v"(Bv) : "memory")
#define MF4(A0, A1, B0, B1) MF(0, 0, A0, B0); G(); MF(0, 1, A0, B1); G(); MF(1, 0, A1, B0); G(); MF(1, 1, A1, B1); G()
    auto step = [&](Frag& c, Frag& n, bool load_next, int tt_next, int st_next, int dma_t, int dma_slot) {
        unsigned ao = 0, p0 = 0, p1 = 0, q0 = 0, q1 = 0;
        if (load_next) {
            ao = aLane + (unsigned)((tt_next % NBUF) * WSLOT);
            const unsigned off = (unsigned)(kh ? off_of(2 * st_next + 1) : off_of(2 * st_next));
            p0 = bBase[0] + off; p1 = bBase[1] + off; q0 = p0 + 2 * SPLITB; q1 = p1 + 2 * SPLITB;
        }
        int gap = 0;
        auto G = [&]() {
            if (load_next) {
                switch (gap) {
                    case 0: DSR(n.a[0][0], ao, 0); break;        case 1: DSR(n.a[0][1], ao, 512); break;
                    case 2: DSR(n.b[0][0], p0, 0); break;        case 3: DSR(n.b[0][1], p1, 0); break;
                    case 4: DSR(n.a[1][0], ao, 4096); break;     case 5: DSR(n.a[1][1], ao, 4608); break;
                    case 6: DSR(n.b[1][0], p0, SPLITB); break;   case 7: DSR(n.b[1][1], p1, SPLITB); break;
                    case 8: DSR(n.a[2][0], ao, 8192); break;     case 9: DSR(n.a[2][1], ao, 8704); break;
                    case 10: DSR(n.b[2][0], q0, 0); break;       case 11: DSR(n.b[2][1], q1, 0); break;
                    default: break;
                }
            }
#ifndef X_NO_DMA
            if (dma_t >= 0 && dma_t < TOTAL_STEPS) { if (gap == 13) issue2(dma_t, dma_slot); if (gap == 17) issue2(dma_t + 1, dma_slot + 1); }
#endif
            ++gap;
        };
        asm volatile("s_waitcnt lgkmcnt(0)" : FRAG_REGS(c) :: "memory");
        MF4(c.a[2][0], c.a[2][1], c.b[0][0], c.b[0][1]);
        MF4(c.a[0][0], c.a[0][1], c.b[2][0], c.b[2][1]);
        MF4(c.a[1][0], c.a[1][1], c.b[1][0], c.b[1][1]);
        MF4(c.a[1][0], c.a[1][1], c.b[0][0], c.b[0][1]);
        MF4(c.a[0][0], c.a[0][1], c.b[1][0], c.b[1][1]);
        MF4(c.a[0][0], c.a[0][1], c.b[0][0], c.b[0][1]);
    };
    __builtin_amdgcn_s_setprio(3);                         // the MFMA wave wins the issue arbitration of its SIMD; the producer fills what is left
    Frag F0, F1;
#ifdef X_STAMP
    const unsigned long long st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    static_assert(NSTEP % NBUF == 0, "buffer slots repeat per channel group");
    __syncthreads();                                       // B1
    __syncthreads();                                       // B2: halo tile of group 0 written, steps 0..5 landed
    loadF(F0, 0, 0);
    int t = 0;
#pragma unroll 1
    for (int cg = 0; cg < NCG; ++cg) {
#pragma unroll
        for (int pr = 0; pr < NSTEP / 2; ++pr) {
            const bool last = pr == NSTEP / 2 - 1;
            step(F0, F1, true, 2 * pr + 1, 2 * pr + 1, -1, 0);
            asm volatile("s_waitcnt lgkmcnt(0)" : FRAG_REGS(F1) :: "memory");
            // all but the six newest copies (steps t+4, t+5) done: steps t+2, t+3 have landed ... for every thread after the barrier; every MFMA wave
            // holds steps t, t+1 in registers: their buffers are free.  (A bare s_barrier: __syncthreads() makes the compiler drain vmcnt to 0,
            // which puts the whole L2 latency of the newest copies on the critical path whatever the ring depth.)
            BAR_PLAIN();
            step(F1, F0, !last, 2 * pr + 2, 2 * pr + 2, -1, 0);  // (last pair: the producers write the next group's halo tile meanwhile)
            if (last && cg + 1 < NCG) { BAR_PLAIN(); loadF(F0, 0, 0); }
            t += 2;
        }
    }
#ifdef X_STAMP
    if (tid == 0) { stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - st_c0; stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - st_r0; }
#endif
    // ---- store: column (lane) = pixel
    float* zi = z + ((size_t)(b0 + w_p) * NO + ot * 128 + w_o * 64) * HW;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) zi[(size_t)(i * 32 + mfma_row(r, lane)) * HW + j * 32 + m] = acc[i][j][r];
}

// ------------------------------------------------------------------------------------------------ host
static double silu64(double v) { return v / (1.0 + exp(-v)); }
// cubic cardinal B-spline on the fp32 knots, evaluated in double (kan_layers.py:209-233 on uniform knots)
static void planes64(const float* kn, float xf, double* out /*9*/) {
    const double xv = xf;
    out[0] = silu64(xv);
    for (int k = 0; k < 8; ++k) out[1 + k] = 0.0;
    if (!(xf >= kn[0] && xf < kn[11])) return;
    int i = 0;
    while (i < 10 && !(xf >= kn[i] && xf < kn[i + 1])) ++i;
    // Cox-de Boor on the actual knots, double
    double N[12];
    for (int j = 0; j < 11; ++j) N[j] = (xf >= kn[j] && xf < kn[j + 1]) ? 1.0 : 0.0;
    for (int k = 1; k <= 3; ++k)
        for (int j = 0; j + k < 11; ++j)
            N[j] = (xv - kn[j]) / ((double)kn[j + k] - kn[j]) * N[j] + ((double)kn[j + k + 1] - xv) / ((double)kn[j + k + 1] - kn[j + 1]) * N[j + 1];
    for (int k = 0; k < 8; ++k) out[1 + k] = N[k];
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20;
    // knots of torch.linspace(-1 - 3h, 1 + 3h, 12), h = 2/5 (kan_layers.py:184-190)
    KanBasis kb; memset(&kb, 0, sizeof kb);
    kb.kind = KAN_BASIS_BSPLINE; kb.n_basis = 8; kb.order = 3; kb.act = KAN_ACT_SILU;
    {
        const float h = 2.0f / 5.0f, start = -1.0f - 3 * h, end = 1.0f + 3 * h, step = (end - start) / 11.0f;
        for (int i = 0; i < 12; ++i) kb.table[i] = i < 6 ? start + step * i : end - step * (11 - i);
    }
    DevBasis db; memset(&db, 0, sizeof db);
    db.kind = kb.kind; db.nb = 8; db.order = 3; db.act = kb.act; db.hb = 1; db.P = 9;
    for (int i = 0; i < KAN_MAX_TABLE; ++i) db.tab[i] = kb.table[i];
    db.inv_h = 11.0f / (kb.table[11] - kb.table[0]); db.g0 = kb.table[0]; db.gN = kb.table[11]; db.ctab = nullptr;

    const size_t nx = (size_t)NB * NC * HW, nwb = (size_t)NO * NC * 9, nws = nwb * 8, nz = (size_t)NB * NO * HW;
    std::vector<float> hx(nx), hwb(nwb), hws(nws);
    unsigned long long s = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0; };
    for (auto& v : hx) { const double u1 = rnd() + 1e-12, u2 = rnd(); v = (float)(sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2)); }
    const double bb = 1.0 / sqrt((double)NC * 9), bsb = 1.0 / sqrt((double)NC * 8 * 9);      // kaiming-uniform-like bounds (fan-in of the two convs)
    for (auto& v : hwb) v = (float)((2 * rnd() - 1) * bb * 1.7320508);
    for (auto& v : hws) v = (float)((2 * rnd() - 1) * bsb * 1.7320508);

    float *dx, *dwb, *dws, *dz; __bf16* dwc;
    CK(hipMalloc(&dx, nx * 4)); CK(hipMalloc(&dwb, nwb * 4)); CK(hipMalloc(&dws, nws * 4)); CK(hipMalloc(&dz, nz * 4));
    CK(hipMalloc(&dwc, (size_t)TOTAL_STEPS * WSTEP_G));
    CK(hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dwb, hwb.data(), nwb * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dws, hws.data(), nws * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

    // ---- split-precision kernel
    CK(hipFuncSetAttribute((const void*)k_split_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    {
        const int n = NCG * NSTEP * 2 * NO;
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_cut_weights, dim3((n + 255) / 256), dim3(256), 0, 0, dwb, dws, dwc);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("cut weights (one-off per weight update, 3 x bf16 pieces, %.1f MB): %.3f ms\n", (double)TOTAL_STEPS * WSTEP_G / 1e6, ms);
    }
    unsigned long long* dstamps; CK(hipMalloc(&dstamps, 256 * 16)); CK(hipMemset(dstamps, 0, 256 * 16));
    CK(hipMemset(dz, 0xff, nz * 4));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_split_fwd, dim3(256), dim3(512), LDS_BYTES, 0, dx, dwc, dz, db, dstamps);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k_split_fwd, dim3(256), dim3(512), LDS_BYTES, 0, dx, dwc, dz, db, dstamps);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize()); CK(hipGetLastError());
    float ms_split; CK(hipEventElapsedTime(&ms_split, e0, e1)); ms_split /= iters;
#ifdef X_STAMP
    {
        std::vector<unsigned long long> hs(512);
        CK(hipMemcpy(hs.data(), dstamps, 256 * 16, hipMemcpyDeviceToHost));
        std::vector<double> clk, cyc;
        for (int i = 0; i < 256; ++i) if (hs[2 * i + 1]) { clk.push_back((double)hs[2 * i] / (double)hs[2 * i + 1] * 0.1); cyc.push_back((double)hs[2 * i]); }
        std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
        printf("in-kernel clock (median over workgroups) %.3f GHz; main loop %.0f cycles = %.1f per 16-deep step (24 MFMAs = 768)\n", clk[clk.size() / 2], cyc[cyc.size() / 2], cyc[cyc.size() / 2] / TOTAL_STEPS);
    }
#endif
    std::vector<float> z_split(nz);
    CK(hipMemcpy(z_split.data(), dz, nz * 4, hipMemcpyDeviceToHost));

    // ---- the shipped exact-fp32 path through the C ABI
    KanGeom g; memset(&g, 0, sizeof g);
    g.B = NB; g.C = NC; g.H = HH; g.W = HH; g.O = NO; g.Ho = HH; g.Wo = HH; g.kh = g.kw = 3; g.sh = g.sw = 1; g.ph = g.pw = 1; g.dh = g.dw = 1;
    g.groups = 1; g.x_bstride = (long long)NC * HW; g.y_bstride = (long long)NO * HW;
    KanPlan plan;
    if (kan_plan(&g, &kb, &plan)) { fprintf(stderr, "kan_plan: %s\n", kan_last_error()); return 2; }
    float *dwp, *dzs;
    CK(hipMalloc(&dwp, plan.packed_weight_bytes)); CK(hipMalloc(&dzs, (size_t)plan.fwd_splits * plan.fwd_slab_elems * 4));
    if (kan_pack_weights(dwb, dws, dwp, nullptr, &g, &kb, nullptr)) { fprintf(stderr, "kan_pack_weights: %s\n", kan_last_error()); return 2; }
    for (int i = 0; i < 3; ++i) if (kan_conv_fwd(dx, dx, dwp, dzs, &g, &kb, nullptr, nullptr)) { fprintf(stderr, "kan_conv_fwd: %s\n", kan_last_error()); return 2; }
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) kan_conv_fwd(dx, dx, dwp, dzs, &g, &kb, nullptr, nullptr);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms_f32; CK(hipEventElapsedTime(&ms_f32, e0, e1)); ms_f32 /= iters;
    std::vector<float> z_f32(nz, 0.f), slab(nz);
    for (int sidx = 0; sidx < plan.fwd_splits; ++sidx) {
        CK(hipMemcpy(slab.data(), dzs + (size_t)sidx * plan.fwd_slab_elems, nz * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < nz; ++i) z_f32[i] += slab[i];
    }

    // ---- fp64 evaluation of the layer on six images
    const int imgs[6] = {0, 1, 130, 131, 254, 255};
    double emax[2] = {0, 0}, esq[2] = {0, 0}, rmax = 0, rsq = 0;
    std::vector<double> E((size_t)NC * NP * 100), W64((size_t)NO * NC * NP * 9), zr((size_t)NO * HW);
    for (int o = 0; o < NO; ++o)
        for (int c = 0; c < NC; ++c)
            for (int p = 0; p < NP; ++p)
                for (int tp = 0; tp < 9; ++tp)
                    W64[(((size_t)o * NC + c) * NP + p) * 9 + tp] = p == 0 ? hwb[((size_t)o * NC + c) * 9 + tp] : hws[((size_t)o * NC * 8 + c * 8 + p - 1) * 9 + tp];
    for (int ii = 0; ii < 6; ++ii) {
        const int b = imgs[ii];
        std::fill(E.begin(), E.end(), 0.0);
        for (int c = 0; c < NC; ++c)
            for (int px = 0; px < HW; ++px) {
                double pl[9]; planes64(kb.table, hx[((size_t)b * NC + c) * HW + px], pl);
                for (int p = 0; p < NP; ++p) E[((size_t)c * NP + p) * 100 + (px / 8 + 1) * 10 + (px % 8) + 1] = pl[p];
            }
        std::fill(zr.begin(), zr.end(), 0.0);
        for (int o = 0; o < NO; ++o)
            for (int c = 0; c < NC; ++c)
                for (int p = 0; p < NP; ++p) {
                    const double* e = &E[((size_t)c * NP + p) * 100];
                    const double* w = &W64[(((size_t)o * NC + c) * NP + p) * 9];
                    double* zo = &zr[(size_t)o * HW];
                    for (int tp = 0; tp < 9; ++tp) {
                        const double wv = w[tp]; const int sh = (tp / 3) * 10 + tp % 3;
                        for (int r = 0; r < 8; ++r)
                            for (int cc = 0; cc < 8; ++cc) zo[r * 8 + cc] += wv * e[r * 10 + cc + sh];
                    }
                }
        for (int o = 0; o < NO; ++o)
            for (int px = 0; px < HW; ++px) {
                const double ref = zr[(size_t)o * HW + px];
                const size_t zi = ((size_t)b * NO + o) * HW + px;
                const double d0 = z_f32[zi] - ref, d1 = z_split[zi] - ref;
                emax[0] = fmax(emax[0], fabs(d0)); emax[1] = fmax(emax[1], fabs(d1));
                esq[0] += d0 * d0; esq[1] += d1 * d1; rmax = fmax(rmax, fabs(ref)); rsq += ref * ref;
            }
    }
    const double gf = 2.0 * NB * HW * NO * (double)NC * 81;
    printf("shape 256 -> 256 @ 8x8 k3 p1, batch 256, B-spline P = 9 (K = 20736): %.1f GFLOP dense\n", gf / 1e9);
    printf("exact fp32 MFMA (libkanconv kan_conv_fwd, %d slab%s)   %8.4f ms  %6.1f TFLOP/s   error vs fp64: max-normalised %.3e  L2-relative %.3e\n",
           plan.fwd_splits, plan.fwd_splits > 1 ? "s" : "", ms_f32, gf / ms_f32 / 1e9, emax[0] / rmax, sqrt(esq[0] / rsq));
    printf("3 x bf16 pieces, 6 products (k_split_fwd)              %8.4f ms  %6.1f TFLOP/s (fp32-equivalent)   error vs fp64: max-normalised %.3e  L2-relative %.3e\n",
           ms_split, gf / ms_split / 1e9, emax[1] / rmax, sqrt(esq[1] / rsq));
    printf("speed-up %.2fx\n", ms_f32 / ms_split);
    return 0;
}
