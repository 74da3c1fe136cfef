// Device-side building blocks shared by the translation units of libkanconv (gfx950 only): MFMA step macros, scalar fast division,
// geometry structs, XCD-aware block order, structural-zero helpers, the plane-staging functor (stage_unit) and the raw-buffer / LDS-DMA
// load helpers.  Everything lives in an anonymous namespace: each translation unit gets its own copy, nothing is exported.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <type_traits>
#include "kanconv.h"
#include "kan_device.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
// ds_read_b32 x4 with immediate byte offsets (asm: the compiler cannot see these loads, so LDS_WAIT4 both waits and
// "touches" the destination registers to order their consumers after the wait)
#define LDS_READ4(r0, r1, r2, r3, addrA, addrB, o0, o1, o2, o3)                                                        \
    asm volatile("ds_read_b32 %0, %4 offset:%6\n\tds_read_b32 %1, %4 offset:%7\n\tds_read_b32 %2, %5 offset:%8\n\tds_read_b32 %3, %5 offset:%9" \
                 : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(addrA), "v"(addrB), "n"(o0), "n"(o1), "n"(o2), "n"(o3) : "memory")
#define LDS_WAIT4(r0, r1, r2, r3, N) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) :: "memory")
__device__ __forceinline__ unsigned lds_addr(const float* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const float*)p; }
// One LDS step of 2x2 MFMA tiles per wave: NK k-pairs, operand A rows LDA floats apart at LDS byte address addrA (k-row
// kh2 already folded in), operand B likewise; reads software-pipelined one k-pair ahead (see the forward kernel).
#define KAN_MFMA_STEP(NK, addrA, LDA, addrB, LDB)                                                                                   \
    do {                                                                                                                           \
        float fa_[2][2], fb_[2][2];                                                                                                \
        LDS_READ4(fa_[0][0], fa_[0][1], fb_[0][0], fb_[0][1], addrA, addrB, 0, 32 * 4, 0, 32 * 4);                                 \
        _Pragma("unroll") for (int kk = 0; kk < (NK); ++kk) {                                                                       \
            const int c_ = kk & 1, n_ = c_ ^ 1;                                                                                    \
            if (kk + 1 < (NK)) {                                                                                                   \
                LDS_READ4(fa_[n_][0], fa_[n_][1], fb_[n_][0], fb_[n_][1], addrA, addrB, (2 * (kk + 1)) * (LDA) * 4,                \
                          (2 * (kk + 1)) * (LDA) * 4 + 128, (2 * (kk + 1)) * (LDB) * 4, (2 * (kk + 1)) * (LDB) * 4 + 128);         \
                LDS_WAIT4(fa_[c_][0], fa_[c_][1], fb_[c_][0], fb_[c_][1], 4);                                                      \
            } else {                                                                                                               \
                LDS_WAIT4(fa_[c_][0], fa_[c_][1], fb_[c_][0], fb_[c_][1], 0);                                                      \
            }                                                                                                                      \
            acc[0][0] = MFMA32(fa_[c_][0], fb_[c_][0], acc[0][0]);                                                                 \
            acc[0][1] = MFMA32(fa_[c_][0], fb_[c_][1], acc[0][1]);                                                                 \
            acc[1][0] = MFMA32(fa_[c_][1], fb_[c_][0], acc[1][0]);                                                                 \
            acc[1][1] = MFMA32(fa_[c_][1], fb_[c_][1], acc[1][1]);                                                                 \
        }                                                                                                                          \
    } while (0)

// The same step with the two pixel blocks of the wave individually switchable in each half of the k-pairs (wave-uniform flags:
// scalar branches around two MFMAs): l0a / l0b = pixel block 0 alive in the first / second NK/2 k-pairs, l1a / l1b likewise block 1
#define KAN_MFMA_STEP_LIVE(NK, addrA, LDA, addrB, LDB, l0a, l0b, l1a, l1b)                                                          \
    do {                                                                                                                           \
        float fa_[2][2], fb_[2][2];                                                                                                \
        LDS_READ4(fa_[0][0], fa_[0][1], fb_[0][0], fb_[0][1], addrA, addrB, 0, 32 * 4, 0, 32 * 4);                                 \
        _Pragma("unroll") for (int kk = 0; kk < (NK); ++kk) {                                                                       \
            const int c_ = kk & 1, n_ = c_ ^ 1;                                                                                    \
            if (kk + 1 < (NK)) {                                                                                                   \
                LDS_READ4(fa_[n_][0], fa_[n_][1], fb_[n_][0], fb_[n_][1], addrA, addrB, (2 * (kk + 1)) * (LDA) * 4,                \
                          (2 * (kk + 1)) * (LDA) * 4 + 128, (2 * (kk + 1)) * (LDB) * 4, (2 * (kk + 1)) * (LDB) * 4 + 128);         \
                LDS_WAIT4(fa_[c_][0], fa_[c_][1], fb_[c_][0], fb_[c_][1], 4);                                                      \
            } else {                                                                                                               \
                LDS_WAIT4(fa_[c_][0], fa_[c_][1], fb_[c_][0], fb_[c_][1], 0);                                                      \
            }                                                                                                                      \
            if (kk < (NK) / 2 ? (l0a) : (l0b)) {                                                                                   \
                acc[0][0] = MFMA32(fa_[c_][0], fb_[c_][0], acc[0][0]);                                                             \
                acc[1][0] = MFMA32(fa_[c_][1], fb_[c_][0], acc[1][0]);                                                             \
            }                                                                                                                      \
            if (kk < (NK) / 2 ? (l1a) : (l1b)) {                                                                                   \
                acc[0][1] = MFMA32(fa_[c_][0], fb_[c_][1], acc[0][1]);                                                             \
                acc[1][1] = MFMA32(fa_[c_][1], fb_[c_][1], acc[1][1]);                                                             \
            }                                                                                                                      \
        }                                                                                                                          \
    } while (0)

namespace {

// Division of a WAVE-UNIFORM non-negative int by a launch constant without the vector ALU.  hipcc expands `a / d` with
// runtime d into ~20 VALU instructions (float reciprocal) even when a is uniform, and on gfx950 VALU issue slots are
// matrix time (measured: each non-MFMA VALU per MFMA costs ~9 cycles of the 64-cycle MFMA).  q = (mulhi(a, m) + a) >> s
// with the round-up magic m (Granlund-Montgomery; exact for 0 <= a < 2^31) is four scalar instructions.
struct FastDiv { unsigned m; int s; };
__host__ __device__ inline FastDiv make_fastdiv(int d) {
    FastDiv f; f.s = 0;
    while ((1ll << f.s) < d) ++f.s;                                   // s = ceil(log2 d)
    f.m = (unsigned)((((1ull << f.s) - (unsigned long long)d) << 32) / (unsigned long long)d + 1ull);
    return f;
}
__device__ __forceinline__ int fastdiv(int a, FastDiv f) {
    if (f.s == 0) return a;                                           // d == 1 (uniform branch)
    const unsigned t = __umulhi((unsigned)a, f.m);
    return (int)((t + (((unsigned)a - t) >> 1)) >> (f.s - 1));
}

struct DevGeom {
    int B, C, H, W, O, Ho, Wo, kh, kw, sh, sw, ph, pw, dh, dw;
    int howo_shift, wo_shift;    // log2(Ho*Wo), log2(Wo) when both are powers of two, else -1 (pixel decode by shifts)
    int pix_major;               // 1: GEMM pixel index = position*B + image (small planes: lets whole taps be skipped)
    int b_shift;                 // log2(B) or -1
    long long xbs, ybs;
    FastDiv divC, divKw;         // by C and by kw (item -> tap, channel; tap -> r, t)
};

constexpr int PERM_MAX = 64;
struct TilePerm { int n; unsigned short idx[PERM_MAX]; };          // pixel-tile dispatch order (n == 0: identity), see balance_tiles

// XCD-aware block order.  Workgroups go to the 8 XCDs (each with its own L2) round-robin by linear block id, so with the
// plain (x = pixel/row tile, y, z = split) order every XCD sees every split and pulls every byte of the streamed operand
// (weights; or x / dz for the weight gradient) into its own L2: 8 copies from HBM / Infinity Cache.  Re-numbering the
// blocks so that XCD j works through the j-th contiguous eighth of the (z, y, x) order keeps the blocks that stream the
// SAME split on the SAME XCD, in step with each other.  `on` is false for position-major launches (own order).
struct BlockId { int x, y, z; };
__device__ __forceinline__ BlockId xcd_block_order(bool on) {
    BlockId b{(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z};
    const unsigned gx = gridDim.x, gy = gridDim.y, T = gx * gy * gridDim.z;
    if (!on || (T & 7u)) return b;
    const unsigned lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const unsigned logical = (lin >> 3) + (lin & 7u) * (T >> 3);
    const unsigned r = logical / gx;
    b.x = (int)(logical - r * gx); b.z = (int)(r / gy); b.y = (int)(r - (unsigned)b.z * gy);
    return b;
}

// Structural zeros.  With zero padding, tap (r,t) of output position (ho,wo) reads outside the image for a fixed set
// of positions; on 4x4 / 2x2 planes that is 31 % / 56 % of all (position, tap) products.  When the pixel axis is
// ordered position-major, a 128-pixel tile holds one or two positions, so a tap is dead or alive for the WHOLE tile and
// its LDS steps (gather, expansion, MFMAs) are skipped outright -- exact, the skipped products are exact zeros.
__device__ __forceinline__ bool tap_alive_out(const DevGeom& g, int hw, int tap) {      // output position hw, forward tap
    const int ho = hw / g.Wo, wo = hw - ho * g.Wo, r = tap / g.kw, t = tap - r * g.kw;
    const int hi = ho * g.sh - g.ph + r * g.dh, wi = wo * g.sw - g.pw + t * g.dw;
    return (unsigned)hi < (unsigned)g.H && (unsigned)wi < (unsigned)g.W;
}
// Split-K ranges must be cut over LIVE steps, or the splits that land on dead taps exit at once while the others
// run full length.  The step axis is a sequence of `nseg` segments (taps, or pixel positions), segment i starting at
// step seg_start(i); live_step_pos returns the step at which `target` live steps have gone by (n_steps if past the end).
template <typename F>
__device__ __forceinline__ int live_step_pos(unsigned mask, int nseg, int n_steps, int target, F seg_start) {
    int acc = 0;
    for (int i = 0; i < nseg; ++i) {
        if (!((mask >> i) & 1u)) continue;
        const int s0 = seg_start(i), n = min(seg_start(i + 1), n_steps) - s0;
        if (target < acc + n) return s0 + (target - acc);
        acc += n;
    }
    return n_steps;
}
template <typename F>
__device__ __forceinline__ int live_step_count(unsigned mask, int nseg, int n_steps, F seg_start) {
    int acc = 0;
    for (int i = 0; i < nseg; ++i)
        if ((mask >> i) & 1u) acc += min(seg_start(i + 1), n_steps) - seg_start(i);
    return acc;
}

__device__ __forceinline__ bool tap_alive_in(const DevGeom& g, int hw, int tap) {       // input position hw, transposed tap
    const int h = hw / g.W, w = hw - h * g.W, r = tap / g.kw, t = tap - r * g.kw;
    const int hn = h + g.ph - r * g.dh, wn = w + g.pw - t * g.dw;
    if (hn < 0 || wn < 0) return false;
    const int ho = hn / g.sh, wo = wn / g.sw;
    return ho * g.sh == hn && wo * g.sw == wn && ho < g.Ho && wo < g.Wo;
}

// C/D register -> row inside a 32x32 MFMA tile (cdna guide section 3)
__device__ __forceinline__ int mfma_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// ============================================================================ staging helpers
// Write the P planes of one (pixel, item) unit into an LDS column, branch-free.  `col` points at row 0 of the unit,
// `ld` is the row stride; the caller guarantees that rows [0, P) of the column may be written (the weight-gradient
// tile keeps a P-row margin on both sides for units that straddle the tile edge) and provides `dump`, an LDS word
// nobody reads.  B-spline planes are sparse (<= S+1 of n_basis non-zero): zero the column, then overwrite the live
// rows (same lane, in-order LDS => correct); rows of bases outside [0, n_basis) go to `dump` instead of a branch.
// FAST != 0 fixes one of the configurations BASELINE.json names at compile time: straight-line code, no runtime loop
// bounds, fewer live scalars.  That matters twice on gfx950: VALU instructions steal fp32-MFMA issue time, and SGPR
// spills are VALU (v_readlane).
//   1 / 2 : B-spline grid 5, order 3 (8 bases) + base branch SiLU / GELU          P = 9   (KANConv2DLayer defaults)
//   3     : RBF, 8 centres + base branch SiLU                                     P = 9   (FastKANConv2DLayer defaults)
//   4 / 5 : Chebyshev degree 4 / 3, no base branch                                P = 5 / 4
//   9     : ReLU-KAN g = 5, k = 3 (8 planes) + base branch SiLU, phases in device memory  P = 9   (halo kernels only)
//   10    : GRAM-KAN degree 3 (4 planes) + base branch, SiLU, coefficients in device memory  P = 5   (halo kernels only)
//   11    : a base branch + ONE constant polynomial plane (degree-0 recurrence)  P = 2   -- with zero weights on the constant plane this is a plain
//           convolution of act(x): Wav-KAN's base conv and 1x1 `wavelet_out` conv (layers/wav_layers.py), Bessel/Taylor/... at degree 0
__host__ __device__ constexpr int fast_planes(int fast) { return (fast == 4 || fast == 6 || fast == 10) ? 5 : (fast == 5 || fast == 7) ? 4 : fast == 8 ? 6 : fast == 11 ? 2 : 9; }
__device__ __forceinline__ float silu_fast(float x) { return x * kan_rcp(1.0f + kan_exp2k(x, -1.44269504088896340736f)); }

template <int KIND, int FAST>
__device__ __forceinline__ void stage_unit(const DevBasis& bs, const float* sTab, bool inb, float xa, float xb,
                                           float* col, int ld, float* dump, int c = 0) {
    if (KIND == KAN_BASIS_RBF && (FAST == 3 || FAST == 8)) {          // 8 (FastKAN default) or 5 (grid_size 5, as kan_vgg.py builds it) centres
        // utils/utils.py:33 with hardware exp2: exp(-u^2) = exp2(-u^2 log2 e), u = (x - c_g) / d
        col[0] = inb ? silu_fast(xa) : 0.f;
        const float inv_d = 1.0f / bs.p0;
#pragma unroll
        for (int j = 0; j < fast_planes(FAST) - 1; ++j) {
            const float u = (xb - bs.tab[j]) * inv_d;
            col[(1 + j) * ld] = inb ? kan_exp2k(u * u, -1.44269504088896340736f) : 0.f;
        }
        return;
    }
    if (KIND == KAN_BASIS_CHEBY && (FAST == 4 || FAST == 5)) {
        // cheby_kan_layers.py:93-96 by recurrence; tanh through hardware exp2/rcp: (e-1)/(e+1), e = exp(2x), |err| ~1e-7 absolute
        const float t = fminf(fmaxf(kan_tanh_fast(xb), bs.p0), bs.p1);
        float Tm = 1.f, Tc = t;
        col[0] = inb ? 1.f : 0.f;
#pragma unroll
        for (int k = 1; k < fast_planes(FAST); ++k) {
            col[k * ld] = inb ? Tc : 0.f;
            const float Tn = 2.f * t * Tc - Tm; Tm = Tc; Tc = Tn;
        }
        return;
    }
    if (KIND == KAN_BASIS_POLY && (FAST == 6 || FAST == 7)) {
        // Recurrence families with a base branch and 4 (FAST 6) or 3 (FAST 7) polynomial planes -- degree 3, the
        // reference's default -- so P is a compile-time 5 / 4.  Activation and squash stay runtime (uniform branches),
        // coefficients come from the kernel argument (scalar registers).  tanh through hardware exp2/rcp as above.
        constexpr int NB = FAST == 6 ? 4 : 3;
        const float base = bs.act == KAN_ACT_SILU ? silu_fast(xa) : kan_act(bs.act, xa);
        float t = xb;
        if (bs.order) {
            t = kan_tanh_fast(xb);
        }
        float Tm = bs.tab[0], Tc = bs.tab[1] * t + bs.tab[2];
        col[0] = inb ? base : 0.f;
        col[ld] = inb ? Tm : 0.f;
#pragma unroll
        for (int k = 1; k < NB; ++k) {
            col[(1 + k) * ld] = inb ? Tc : 0.f;
            if (k + 1 < NB) { const float Tn = (bs.tab[3 * k] * t + bs.tab[3 * k + 1]) * Tc + bs.tab[3 * k + 2] * Tm; Tm = Tc; Tc = Tn; }
        }
        return;
    }
    if (KIND == KAN_BASIS_POLY && FAST == 11) {
        col[0] = inb ? (bs.act == KAN_ACT_SILU ? silu_fast(xa) : kan_act(bs.act, xa)) : 0.f;
        col[ld] = inb ? bs.tab[0] : 0.f;
        return;
    }
    if (KIND == KAN_BASIS_RELU && FAST == 9) {
        // ReLU-KAN defaults (relu_kan_layers.py:118-136: g = 5, k = 3 => 8 planes, SiLU base branch, P = 9): the per-channel phases come
        // from device memory (16 loads; the lanes of a wave mostly share the channel).  bs.order selects value / d phase_low / d phase_high
        // as in kan_planes<KAN_BASIS_RELU> (uniform branch).  Used by the halo kernels, where expansions are rare.
        const float* lo = bs.ctab + (size_t)(inb ? c : 0) * 16;
        const float r = bs.p0;
        const int mode = bs.order;
        col[0] = (inb && mode == 0) ? silu_fast(xa) : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float x1 = fmaxf(xb - lo[j], 0.f), x2 = fmaxf(lo[8 + j] - xb, 0.f);
            const float q = x1 * x2 * r, q2 = 2.0f * q * r;
            const float v = mode == 0 ? q * q : mode == 1 ? -(q2 * x2) : q2 * x1;
            col[(1 + j) * ld] = inb ? v : 0.f;
        }
        return;
    }
    if (KIND == KAN_BASIS_GRAM && FAST == 10) {
        // GRAM-KAN degree 3 with SiLU (gram_kan_layers.py:150-182): planes act(P_k(tanh x)), P_0 = 1, P_1 = t, P_k = t P_{k-1} - c_k P_{k-2},
        // c_k from device memory (layer-global); bs.order = m >= 1 selects the derivative w.r.t. c_{m+1} (act'(P_k) dP_k/dc), base plane
        // zero, as kan_planes<KAN_BASIS_GRAM>.  tanh and SiLU through hardware exp2 / rcp.  P = 5.  Halo kernels only.
        const float* cf = bs.ctab;
        const int mode = bs.order;
        const float t = kan_tanh_fast(xb);
        col[0] = (inb && mode == 0) ? silu_fast(xa) : 0.f;
        float Pm = 1.f, Pc = t, Qm = 0.f, Qc = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float Pk = k == 0 ? 1.f : Pc, Qk = k == 0 ? 0.f : Qc;
            const float sg = kan_rcp(1.0f + kan_exp2k(Pk, -1.44269504088896340736f));
            const float v = mode == 0 ? Pk * sg : sg * (1.0f + Pk * (1.0f - sg)) * Qk;
            col[(1 + k) * ld] = inb ? v : 0.f;
            if (k >= 1 && k + 1 < 4) {
                const float ck = cf[k + 1];
                const float Pn = t * Pc - ck * Pm, Qn = t * Qc - ck * Qm - (k + 1 == mode + 1 ? Pm : 0.f);
                Pm = Pc; Pc = Pn; Qm = Qc; Qc = Qn;
            }
        }
        return;
    }
    if (KIND == KAN_BASIS_BSPLINE && FAST != 0) {
        float base = 0.f, N0 = 0.f, N1 = 0.f, N2 = 0.f, N3 = 0.f; int j0 = -8;
        const bool live = inb && xa >= bs.g0 && xa < bs.gN;             // NaN fails both, as the reference's indicator
        if (inb) base = FAST == 1 ? silu_fast(xa) : kan_act(KAN_ACT_GELU, xa);
        if (live) {
            const int i = min((int)((xa - bs.g0) * bs.inv_h), 10);       // 11 knot intervals
#ifdef KAN_EXACT_TRANSCENDENTALS
            // measurement build: u and the four pieces in double on the fp32 knots, rounded once
            const double ud = fmin(fmax(((double)xa - (double)sTab[i]) / ((double)sTab[i + 1] - (double)sTab[i]), 0.0), 1.0), vd = 1.0 - ud;
            N0 = (float)(vd * vd * vd / 6.0); N3 = (float)(ud * ud * ud / 6.0);
            N1 = (float)((((-3.0 * vd + 3.0) * vd + 3.0) * vd + 1.0) / 6.0); N2 = (float)((((-3.0 * ud + 3.0) * ud + 3.0) * ud + 1.0) / 6.0);
#else
            const float u = fminf(fmaxf((xa - sTab[i]) * bs.inv_h, 0.f), 1.f), v = 1.f - u;
            // two-wide (v_pk_*_f32): (N0, N3) = (v^3, u^3)/6 and, by the cubic's symmetry N1(u) = N2(1 - u),
            // (N1, N2) = q(v), q(u) with q(t) = (-3t^3 + 3t^2 + 3t + 1)/6 in Horner form
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            const f32x2 vu = {v, u};
            const f32x2 n03 = vu * vu * vu * (1.f / 6.f);
            const f32x2 n12 = (((vu * -3.f + 3.f) * vu + 3.f) * vu + 1.f) * (1.f / 6.f);
            N0 = n03.x; N3 = n03.y; N1 = n12.x; N2 = n12.y;
#endif
            j0 = i - 3;
        }
        col[0] = base;
#pragma unroll
        for (int p = 1; p < 9; ++p) col[p * ld] = 0.f;
        *(((unsigned)j0 < 8u) ? col + (1 + j0) * ld : dump) = N0;
        *(((unsigned)(j0 + 1) < 8u) ? col + (2 + j0) * ld : dump) = N1;
        *(((unsigned)(j0 + 2) < 8u) ? col + (3 + j0) * ld : dump) = N2;
        *(((unsigned)(j0 + 3) < 8u) ? col + (4 + j0) * ld : dump) = N3;
        return;
    }
    const int P = bs.P, hb = bs.hb;
    if (KIND == KAN_BASIS_BSPLINE) {
        float base = 0.f, N[4] = {0.f, 0.f, 0.f, 0.f}; int j0 = -8;
        if (inb) {
            if (hb) base = kan_act_fast(bs.act, xa);
            if (!bspline_uniform<false>(bs.order, xb, sTab, bs.nb + bs.order + 1, bs.inv_h, j0, N)) j0 = -8;
        }
#pragma unroll 1
        for (int p = 0; p < P; ++p) col[p * ld] = (p < hb) ? base : 0.f;      // (run-time loop: see kan_planes_each)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (r <= bs.order) {                                     // uniform
                const int j = j0 + r;
                float* dst = ((unsigned)j < (unsigned)bs.nb) ? col + (hb + j) * ld : dump;
                *dst = N[r];
            }
        }
    } else {
        // generic (run-time P) specs: planes streamed into the LDS column by run-time loops, no register array (kan_device.h: kan_planes_each)
        kan_planes_each<KIND, false>(bs, sTab, xa, xb, inb ? c : 0, [&](int p, float v) { col[p * ld] = inb ? v : 0.f; });
    }
}

constexpr int KCM = 36;                        // (legacy constant kept for plan arithmetic)

// Occupancy is the lever on this chip for an exact-fp32 MFMA GEMM (measured: 2 -> 4 workgroups per CU took the
// bwd-data kernel from 79 to 120 TFLOP/s): every kernel below is sized for FOUR 256-thread workgroups per CU,
// i.e. <= 128 VGPRs and <= 40 KB of LDS, with 16-18 deep LDS steps, two LDS buffers and one barrier per step.
// Masked gathers go through a raw buffer descriptor: an offset >= num_records returns 0 from the hardware bounds
// check, so "out of image / out of tensor" costs neither a branch nor a second load (hipcc otherwise serialises
// such loads behind s_waitcnt vmcnt(0)).  Offsets are 32-bit bytes: the host rejects tensors >= 2 GiB.
#define KAN_OOB 0x80000000u
typedef __amdgpu_buffer_rsrc_t kan_rsrc;
__device__ __forceinline__ kan_rsrc make_rsrc(const float* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float buf_load(kan_rsrc r, unsigned byte_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, 0));
}
// Async masked 4-byte-per-lane gather global -> LDS through the buffer descriptor: lane i lands at lds_base + 4*i
// (lds_base WAVE-UNIFORM), an out-of-range offset lands a zero.  No VGPR destination, no ds_write.
__device__ __forceinline__ void buf_load_lds4(kan_rsrc r, unsigned byte_off, float* lds_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_base, 4, (int)byte_off, 0, 0, 0);
}
// Async 16-byte-per-lane copy global -> LDS (no VGPRs): lds_base is the WAVE-UNIFORM destination, lane i lands at
// lds_base + 16*i; the source address is per lane.
__device__ __forceinline__ void glds16(const float* gsrc, float* lds_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_base, 16, 0, 0);
}

// ---------------------------------------------------------------- small host helpers (plan arithmetic, kernel-argument structs)
inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }
inline int round_up(int a, int b) { return (a + b - 1) / b * b; }
inline int ngroups(const KanGeom* g) { return g->groups > 0 ? g->groups : 1; }
inline int log2_exact(int v) { int s = 0; while ((1 << s) < v) ++s; return (1 << s) == v ? s : -1; }

inline DevGeom dev_geom(const KanGeom* g) {
    DevGeom d{g->B, g->C, g->H, g->W, g->O, g->Ho, g->Wo, g->kh, g->kw, g->sh, g->sw, g->ph, g->pw, g->dh, g->dw, -1, -1, 0, -1,
              g->x_bstride, g->y_bstride};
    const int a = log2_exact(g->Ho * g->Wo), b = log2_exact(g->Wo);
    if (a >= 0 && b >= 0) { d.howo_shift = a; d.wo_shift = b; }
    d.b_shift = log2_exact(g->B);
    d.divC = make_fastdiv(g->C); d.divKw = make_fastdiv(g->kw);
    return d;
}
inline DevBasis dev_basis(const KanBasis* b) {
    DevBasis d;
    d.kind = b->kind; d.nb = b->n_basis; d.order = b->order; d.act = b->act;
    d.hb = b->act != KAN_ACT_NONE ? 1 : 0; d.P = b->n_basis + d.hb;
    d.p0 = b->p0; d.p1 = b->p1; d.inv_h = 0.f; d.g0 = 0.f; d.gN = 0.f;
    for (int i = 0; i < KAN_MAX_TABLE; ++i) d.tab[i] = b->table[i];
    d.ctab = (b->kind == KAN_BASIS_RELU || b->kind == KAN_BASIS_GRAM) ? b->chan_table : nullptr;
    if (b->kind == KAN_BASIS_BSPLINE) {
        int nk = b->n_basis + b->order + 1;
        float span = b->table[nk - 1] - b->table[0];
        d.inv_h = span > 0.f ? (float)(nk - 1) / span : 0.f;
        d.g0 = b->table[0]; d.gN = b->table[nk - 1];
    }
    return d;
}

}  // namespace
