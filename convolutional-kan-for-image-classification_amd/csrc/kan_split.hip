// libkanconv, OPT-IN split-precision forward (round 3).  NOT on any default path: `dtype` f32 stays exact; this mode is reached only
// through its own entry points (kan_split_*), carries its own tolerance in the tests and is reported under `other_workloads` by bench.py.
//
// Every fp32 operand (expanded plane value, weight) is cut into three bf16 pieces hi + mid + lo (24 mantissa bits); each 16-deep k-block runs six
// v_mfma_f32_32x32x16_bf16 products (lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi) into the fp32 accumulator.  Scope of this first kernel: the default
// B-spline spec (grid 5, order 3, SiLU base branch: P = 9 planes) on 8x8 or 16x16 planes, 3x3 / stride 1 / pad 1, one group, C % 8 == 0, O % 128 == 0 (even
// batch on 8x8) -- KAN-VGG11's 64 -> 128 @ 16x16, 128 -> 256 and 256 -> 256 @ 8x8 layers.  Replaces, for such a layer, kan_layers.py:199-200, 203-239 (as kan_conv_fwd does).
//
// Kernel design (k_split_fwd), following the halo forward of the library:
//   tile    128 outputs x 128 pixels (two whole 8x8 images, or eight rows of one 16x16 image) per workgroup, one workgroup per CU at the VGG shapes; 512 threads = 4 MFMA waves (wave tile
//           64 x 64, 2 x 2 blocks of 32 x 32) + 4 PRODUCER waves, one of each per SIMD;
//   B side  per group of 8 input channels the producers expand the 2 x 64 input values once (SiLU + 8 B-spline planes, fp32 vector ALU), cut each
//           plane into 3 bf16 pieces (v_cvt_pk_bf16_f32 on channel pairs) and write a zero-bordered halo tile per image into LDS, plane-major inside a
//           cell: the 8 channels of one plane are 16 contiguous bytes = one lane's share of a 16-deep MFMA operand (k = 8 (lane >> 5) + j), so every tap
//           reads the SAME tile through a shifted address with one ds_read_b128 per piece and block.  Cell = 9 chunks of 16 B (odd: 8 consecutive pixels
//           hit 8 different 16-byte slots), rows of 88 chunks (= 8 mod 16: the four lane groups of ds_read_b128 each cover all 16 slots; the right border
//           cell of a row overlaps the left border cell of the next -- both zero, never written).  The next group's pieces are computed into registers
//           WHILE the current group is contracted and written in the last step of the group (the MFMA waves hold that step's operands in registers);
//   depth   per channel group 81 (tap, plane) k-groups of 8 (+ 3 zero groups = 42 steps of 16); lanes 0-31 and 32-63 of a step read DIFFERENT
//           k-groups (own shift each, a compile-time constant per step: the 21 step pairs of a group are unrolled);
//   A side  weights pre-cut (kan_split_pack_weights) into [step][piece][k-half][output][8 bf16] and streamed by 16-byte LDS-DMA into a ring of six
//           one-step buffers (12 KB per step) by the PRODUCER waves, six steps ahead; one barrier per TWO steps (48 MFMAs per wave), entered by the
//           producers under a counted s_waitcnt vmcnt(6) and issued as a bare s_barrier (__syncthreads() drains vmcnt to 0: the L2 latency of the
//           newest copies then sits on the critical path at any ring depth);
//   loop    explicit ISA: the 24 MFMAs of a step with the 12 ds_read_b128 of the NEXT step placed one per MFMA gap.
// tools/probe/split_bf16_conv.hip is the stand-alone A/B harness this kernel was developed in (cycles per step of every design step: its header).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <cstdint>
#include "kanconv.h"
#include "kan_device.h"
#include "kan_common.h"
#include "kan_internal.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16s __attribute__((ext_vector_type(16)));
typedef float f32x4s __attribute__((ext_vector_type(4)));
#define f32x16 f32x16s
#define f32x4 f32x4s
#define BF(v) __builtin_bit_cast(bf16x8, v)
#define BAR_PLAIN() asm volatile("s_barrier" ::: "memory")
#define BAR_LDS() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

constexpr int NP = 9, CG = 8, NGRP = 84, NSTEP = NGRP / 2;
constexpr int CELLC = 9;                                             // chunks (16 B) per cell
// Pixel tile = 128 pixels: two whole 8x8 images (PW = 8: 2 x 10 halo rows of 10 cells) or eight rows of one 16x16 image (PW = 16: 10 halo rows of 18
// cells, the outer two real data or image border depending on the half).  Row pitch in chunks: the right border cell of a row overlaps the left border
// cell of the next (both zero, never written); 88 = 8 mod 16 for 8-pixel rows and 160 = 0 mod 16 for 16-pixel rows make the four lane groups of a
// ds_read_b128 (two / one image rows of a 32-pixel block each) cover all sixteen 16-byte slots.
template <int PW> struct SplitGeo {
    static constexpr int ROWC = PW == 8 ? 88 : 160, ROWS = PW == 8 ? 20 : 10, HWP = PW * PW;
    static constexpr int SPLITB = (ROWS * ROWC + 2) * 16, HALOB = 3 * SPLITB;      // bytes per piece (+ the overhang of the last border cell), per halo tile
};
constexpr int WSLOT = 3 * 2 * 128 * 16, NBUF = 6;                    // one step of one 128-output tile in LDS
template <int PW> constexpr int lds_bytes() { return SplitGeo<PW>::HALOB + NBUF * WSLOT + NGRP * 4 + 64; }

template <int ROWC> __device__ constexpr int off_of(int gi) { return gi < 81 ? (((gi / 9) / 3 - 1) * ROWC + ((gi / 9) % 3 - 1) * CELLC + gi % 9) * 16 : 0; }
__device__ inline int split_mfma_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }
#define mfma_row split_mfma_row

__device__ inline void split3(float v, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)v; const float r1 = v - (float)h;
    m = (__bf16)r1; const float r2 = r1 - (float)m;
    l = (__bf16)r2;
}

// ---- weights: reference layout (base [O][C][3][3], spline [O][C*8][3][3], channel c*8+k) -> wc[cg][step][piece][k-half][o][8 bf16]
__global__ void k_cut_weights(const float* __restrict__ wb, const float* __restrict__ ws, __bf16* __restrict__ wc, int NC, int NO) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;             // (cg, step, kh, o)
    if (idx >= (NC / CG) * NSTEP * 2 * NO) return;
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
template <int PW>
__global__ __launch_bounds__(512, 1) void k_split_fwd(const float* __restrict__ x, const __bf16* __restrict__ wc, float* __restrict__ z, DevBasis bs, int NC, int NO, int o_tiles) {
    const int NCG = NC / CG, TOTAL_STEPS = NCG * NSTEP, WSTEP_G = 3 * 2 * NO * 16;        // channel groups, 16-deep steps, bytes of one step of the cut weights
    constexpr int ROWC = SplitGeo<PW>::ROWC, SPLITB = SplitGeo<PW>::SPLITB, HALOB = SplitGeo<PW>::HALOB, HW = SplitGeo<PW>::HWP;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sH = smem;
    unsigned char* sW = smem + HALOB;
    int* sOff = (int*)(smem + HALOB + NBUF * WSLOT);
    float* sTab = (float*)(sOff + NGRP);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w_o = wave & 1, w_p = wave >> 1, kh = lane >> 5, m = lane & 31;
    const int ot = blockIdx.x % o_tiles, ptile = blockIdx.x / o_tiles;
    const int b0 = PW == 8 ? ptile * 2 : ptile >> 1, half_img = PW == 8 ? 0 : (ptile & 1);      // PW = 16: image b0, rows [8 half_img, 8 half_img + 8)

    for (int i = tid; i < HALOB / 16; i += 512) ((uint4*)sH)[i] = uint4{0u, 0u, 0u, 0u};
    if (tid < NGRP) {
        int off = 0;
        if (tid < 81) { const int tap = tid / 9, p = tid % 9, dr = tap / 3 - 1, dc = tap % 3 - 1; off = (dr * ROWC + dc * CELLC + p) * 16; }
        sOff[tid] = off;
    }
    if (tid < 16) sTab[tid] = bs.tab[tid];

    // weight stream: 3 chunks of 16 B per thread and step
    const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(wc), 0, (int)((long long)TOTAL_STEPS * WSTEP_G), 0x00020000);
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
        const int lp = bj * 32 + m;
        const int hrow = PW == 8 ? w_p * 10 + (lp >> 3) + 1 : w_p * 4 + (lp >> 4) + 1, c = PW == 8 ? (lp & 7) : (lp & 15);      // halo row, column
        bBase[bj] = (unsigned)((hrow * ROWC + (c + 1) * CELLC) * 16);
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
    const unsigned cellB = PW == 8 ? (unsigned)(((qi * 10 + (lp >> 3) + 1) * ROWC + ((lp & 7) + 1) * CELLC) * 16 + hh * 8)
                                   : (unsigned)((((q >> 4) + 1) * ROWC + ((q & 15) + 1) * CELLC) * 16 + hh * 8);
    const float* xq = PW == 8 ? x + ((size_t)(b0 + qi) * NC + hh * 4) * HW + lp : x + ((size_t)b0 * NC + hh * 4) * HW + half_img * 128 + q;
    // PW = 16: the ninth real row of the tile (image row 8 below the upper half, row 7 above the lower half) -- 16 pixels x 2 channel halves, lanes q < 16
    const bool extra = PW == 16 && q < 16;
    const unsigned cellX = (unsigned)(((half_img ? 0 : 9) * ROWC + (q + 1) * CELLC) * 16 + hh * 8);
    const float* xqx = x + ((size_t)b0 * NC + hh * 4) * HW + (half_img ? 7 : 8) * 16 + (q & 15);

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
    auto compute = [&](int cg, Pieces& pc, const float* xsrc) {
        float v[4][NP];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float xv = xsrc[(size_t)(cg * CG + j) * HW];
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
    auto write = [&](const Pieces& pc, unsigned cellB) {
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
        Pieces pc, px;                                     // px: the extra halo row of a 16x16 tile (lanes q < 16)
        for (int i = 0; i < NBUF; ++i) issue2(i, i);
        __syncthreads();                                   // B1: zero fill, tables
        compute(0, pc, xq); write(pc, cellB);
        if (extra) { compute(0, px, xqx); write(px, cellX); }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                   // B2
        int t = 0;
#pragma unroll 1
        for (int cg = 0; cg < NCG; ++cg) {
            const bool more = cg + 1 < NCG;
            if (more) { compute(cg + 1, pc, xq); if (extra) compute(cg + 1, px, xqx); }
#pragma unroll
            for (int pr = 0; pr < NSTEP / 2; ++pr) {
                if (t + 4 < TOTAL_STEPS) asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                if (t + 6 < TOTAL_STEPS) { issue2(t + 6, 2 * pr); issue2(t + 7, 2 * pr + 1); }
                t += 2;
            }
            if (more) { write(pc, cellB); if (extra) write(px, cellX); BAR_LDS(); }
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
        const unsigned ao = aLane + (unsigned)((tt % NBUF) * WSLOT);
        const unsigned off = (unsigned)(kh ? off_of<ROWC>(2 * st + 1) : off_of<ROWC>(2 * st));
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
            const unsigned off = (unsigned)(kh ? off_of<ROWC>(2 * st_next + 1) : off_of<ROWC>(2 * st_next));
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
            if (dma_t >= 0 && dma_t < TOTAL_STEPS) { if (gap == 13) issue2(dma_t, dma_slot); if (gap == 17) issue2(dma_t + 1, dma_slot + 1); }
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
    // ---- store: column (lane) = pixel
    float* zi = PW == 8 ? z + ((size_t)(b0 + w_p) * NO + ot * 128 + w_o * 64) * HW
                        : z + ((size_t)b0 * NO + ot * 128 + w_o * 64) * HW + half_img * 128 + w_p * 64;      // (16-pixel rows: a block's 32 pixels are contiguous)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) zi[(size_t)(i * 32 + mfma_row(r, lane)) * HW + j * 32 + m] = acc[i][j][r];
}


const char* split_reject(const KanGeom* g, const KanBasis* b) {
    if (!g || !b) return "null geometry / basis";
    if (b->kind != KAN_BASIS_BSPLINE || b->n_basis != 8 || b->order != 3 || b->act != KAN_ACT_SILU) return "split-precision forward: default B-spline spec only (grid 5, order 3, SiLU)";
    if (!((g->H == 8 && g->W == 8) || (g->H == 16 && g->W == 16)) || g->Ho != g->H || g->Wo != g->W || g->kh != 3 || g->kw != 3 || g->sh != 1 || g->sw != 1 || g->ph != 1 || g->pw != 1 || g->dh != 1 || g->dw != 1)
        return "split-precision forward: 8x8 or 16x16 planes, 3x3 / stride 1 / pad 1 only";
    if (g->groups > 1 || g->C % CG || g->O % 128 || (g->H == 8 && g->B % 2) || g->C < CG) return "split-precision forward: one group, C % 8 == 0, O % 128 == 0, even batch on 8x8 planes";
    if (g->x_bstride != (long long)g->C * g->H * g->W || g->y_bstride != (long long)g->O * g->H * g->W) return "split-precision forward: dense NCHW tensors";
    if ((long long)(g->C / CG) * NSTEP * 3 * 2 * g->O * 16 >= (1ll << 31)) return "split-precision forward: cut weights must stay under 2 GiB";
    return nullptr;
}

}  // namespace

extern "C" {

int kan_split_supported(const KanGeom* g, const KanBasis* b) { return split_reject(g, b) == nullptr ? 1 : 0; }

long long kan_split_weight_bytes(const KanGeom* g, const KanBasis* b) {
    if (split_reject(g, b)) return 0;
    return (long long)(g->C / CG) * NSTEP * 3 * 2 * g->O * 16;
}

int kan_split_pack_weights(const float* w_base, const float* w_basis, void* wc, const KanGeom* g, const KanBasis* b, void* stream) {
    const char* why = split_reject(g, b);
    if (why) return kan_fail_msg("%s", why);
    if (!w_base || !w_basis || !wc) return kan_fail_msg("kan_split_pack_weights: null pointer%s", "");
    const int n = (g->C / CG) * NSTEP * 2 * g->O;
    hipLaunchKernelGGL(k_cut_weights, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_base, w_basis, (__bf16*)wc, g->C, g->O);
    return hipGetLastError() == hipSuccess ? 0 : kan_fail_msg("kan_split_pack_weights: launch failed%s", "");
}

int kan_conv_fwd_split(const float* x, const void* wc, float* z, const KanGeom* g, const KanBasis* b, void* stream) {
    const char* why = split_reject(g, b);
    if (why) return kan_fail_msg("%s", why);
    if (!x || !wc || !z) return kan_fail_msg("kan_conv_fwd_split: null pointer%s", "");
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)k_split_fwd<8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes<8>()) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_split_fwd<16>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes<16>()) != hipSuccess)
            return kan_fail_msg("kan_conv_fwd_split: cannot reserve %s of LDS", "158 KB");
        attr_set = true;
    }
    const DevBasis db = dev_basis(b);
    const int o_tiles = g->O / 128;
    if (g->H == 8)
        hipLaunchKernelGGL(k_split_fwd<8>, dim3((unsigned)((g->B / 2) * o_tiles)), dim3(512), lds_bytes<8>(), (hipStream_t)stream, x, (const __bf16*)wc, z, db, g->C, g->O, o_tiles);
    else
        hipLaunchKernelGGL(k_split_fwd<16>, dim3((unsigned)(g->B * 2 * o_tiles)), dim3(512), lds_bytes<16>(), (hipStream_t)stream, x, (const __bf16*)wc, z, db, g->C, g->O, o_tiles);
    return hipGetLastError() == hipSuccess ? 0 : kan_fail_msg("kan_conv_fwd_split: launch failed%s", "");
}

}  // extern "C"
