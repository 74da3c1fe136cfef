// Device-side basis / activation functors shared by the three conv kernels (gfx950 only).
//
// Every "plane" p of an input value is one row of the implicit GEMM:
//   p = 0            : base branch  act(x)                     (absent when act == KAN_ACT_NONE)
//   p = hb + k       : basis k evaluated on xn                 (hb = 1 if a base branch exists)
// The same functors produce the planes (forward, weight gradient) and their derivatives
// (input gradient), so forward and backward cannot drift apart.
#pragma once
#include <hip/hip_runtime.h>
#include "kanconv.h"

#define KAN_PMAX KAN_MAX_PLANES

struct DevBasis {            // by-value kernel argument
    int kind, nb, order, act, P, hb;
    float p0, p1, inv_h;
    float g0, gN;            // B-spline: first / last knot (the span outside which every basis is zero)
    float tab[KAN_MAX_TABLE];
    const float* ctab;       // device-memory parameters: ReLU-KAN per-channel phases [C][2][nb]; Gram coefficients [nb]
};

// ---------------------------------------------------------------- transcendentals of the hot loops
// The staging code and the bwd-data epilogue evaluate exp / reciprocal / tanh through the hardware's v_exp_f32 / v_rcp_f32 (1 ulp each):
// on gfx950 every vector instruction next to the fp32 MFMAs is matrix time, and libm's expf / IEEE division are ~10 / ~8 instructions.
// -DKAN_EXACT_TRANSCENDENTALS builds the same kernels on libm-accurate expf / tanhf and IEEE division instead -- a measurement build
// (tools/exact_ab.py: what the fast forms cost in accuracy, what the exact ones cost in time), never the shipped library.
//   kan_exp2k(x, K) = 2^(x K) with K a compile-time multiple of log2(e): i.e. exp(x * (K ln 2)), K ln 2 in {-1, 2, -1/2} exactly
#ifdef KAN_EXACT_TRANSCENDENTALS
__device__ __forceinline__ float kan_exp2k(float x, float K) { return expf(x * (float)((double)K * 0.69314718055994530942)); }
__device__ __forceinline__ float kan_rcp(float x) { return 1.0f / x; }
__device__ __forceinline__ float kan_tanh_fast(float x) { return tanhf(x); }
#else
__device__ __forceinline__ float kan_exp2k(float x, float K) { return __builtin_amdgcn_exp2f(x * K); }
__device__ __forceinline__ float kan_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// tanh through hardware exp2 / rcp: (e - 1) / (e + 1), e = exp(2x) (argument clamped: e stays finite), |err| ~ 1e-7 absolute
__device__ __forceinline__ float kan_tanh_fast(float x) {
    const float e = __builtin_amdgcn_exp2f(fminf(x, 40.f) * 2.88539008177792681472f);
    return (e - 1.0f) * __builtin_amdgcn_rcpf(e + 1.0f);
}
#endif

// ---------------------------------------------------------------- activations
// kan_layers.py:199 / fast_kan_layers.py:103: base_activation(x); torch CPU formulas.
__device__ __forceinline__ float kan_act(int act, float x) {
    switch (act) {
        case KAN_ACT_GELU:      return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
        case KAN_ACT_SILU:      return x / (1.0f + expf(-x));
        case KAN_ACT_RELU:      return x > 0.f ? x : 0.f;
        case KAN_ACT_TANH:      return tanhf(x);
        case KAN_ACT_SIGMOID:   return 1.0f / (1.0f + expf(-x));
        case KAN_ACT_GELU_TANH: {
            float u = 0.79788456080286535588f * (x + 0.044715f * x * x * x);
            return 0.5f * x * (1.0f + tanhf(u));
        }
        default:                return x;
    }
}

__device__ __forceinline__ float kan_act_grad(int act, float x) {
    switch (act) {
        case KAN_ACT_GELU: {
            float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
            float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
            return cdf + x * pdf;
        }
        case KAN_ACT_SILU: {
            float s = 1.0f / (1.0f + expf(-x));
            return s * (1.0f + x * (1.0f - s));
        }
        case KAN_ACT_RELU:    return x > 0.f ? 1.f : 0.f;
        case KAN_ACT_TANH:    { float t = tanhf(x); return 1.0f - t * t; }
        case KAN_ACT_SIGMOID: { float s = 1.0f / (1.0f + expf(-x)); return s * (1.0f - s); }
        case KAN_ACT_GELU_TANH: {
            float x2 = x * x;
            float u = 0.79788456080286535588f * (x + 0.044715f * x * x2);
            float t = tanhf(u);
            float du = 0.79788456080286535588f * (1.0f + 3.0f * 0.044715f * x2);
            return 0.5f * (1.0f + t) + 0.5f * x * (1.0f - t * t) * du;
        }
        default:              return 1.f;
    }
}

// SiLU for the staging hot loop.  On gfx950 the fp32 MFMA shares the vector ALU, so every VALU instruction here costs
// matrix time; x * rcp(1 + exp2(-x*log2e)) is 5 instructions instead of ~25 for x / (1 + expf(-x)), at <= 4e-7
// relative error for |x| <= 6 (hardware exp2/rcp are 1 ulp).
__device__ __forceinline__ float kan_act_fast(int act, float x) {
    if (act == KAN_ACT_SILU) return x * kan_rcp(1.0f + kan_exp2k(x, -1.44269504088896340736f));
    return kan_act(act, x);
}

// ---------------------------------------------------------------- B-spline (uniform knots, closed form)
// Restates kan_layers.py:209-233 for the <= S+1 bases that are non-zero at x.  The knot interval
// [g_i, g_{i+1}) holding x is found with the SAME fp32 knots and half-open comparisons as the
// reference's order-0 indicator.  The reference's knot vector is always torch.linspace (uniform,
// kan_layers.py:184-190; the host side rejects anything else), on which every Cox-de Boor
// denominator is k*h, so the recursion collapses to the cardinal B-spline pieces in
// u = (x - g_i)/h:   S=1: [1-u, u]   S=2: [(1-u)^2, -2u^2+2u+1, u^2]/2
//                    S=3: [(1-u)^3, 3u^3-6u^2+4, -3u^3+3u^2+3u+1, u^3]/6
// (max |delta| vs the reference recursion 1.8e-7).  N[r] is basis j0 + r,
// j0 = i - S.  With DERIV, N holds d/dx instead (the indicator has zero gradient, so this is
// what autograd of the reference recursion evaluates to).  `kn`: knot table in LDS.
template <bool DERIV>
__device__ __forceinline__ bool bspline_uniform(int S, float x, const float* kn, int nkn, float inv_h,
                                                int& j0, float (&N)[4]) {
    const int NI = nkn - 1;                          // number of knot intervals
    if (!(x >= kn[0] && x < kn[NI])) return false;   // also rejects NaN, as the indicator does
    int i = (int)floorf((x - kn[0]) * inv_h);
    i = min(max(i, 0), NI - 1);
    float gi = kn[i];
    if (S == 0) {                                    // piecewise constant: the cell itself is the value, settle it exactly
        if (x < gi) { i = max(i - 1, 0); gi = kn[i]; }
        else if (x >= kn[i + 1]) { i = min(i + 1, NI - 1); gi = kn[i]; }
    }
    // S >= 1: a cell picked one off at a knot (x within an ulp of g_i) only shifts u to ~0 or ~1 of the neighbour,
    // where the pieces agree to O(ulp) (C^0 for S=1 ... C^2 for S=3); u is clamped below.
    j0 = i - S;
#ifdef KAN_EXACT_TRANSCENDENTALS
    // measurement build: u and the pieces in double on the fp32 knots themselves, rounded once (<= 0.5 ulp per plane)
    typedef double real;
    const real ih = 1.0 / ((double)kn[i + 1] - (double)gi);
    const real u = fmin(fmax(((double)x - (double)gi) * ih, 0.0), 1.0);
#else
    typedef float real;
    const real ih = inv_h;
    const real u = fminf(fmaxf((x - gi) * inv_h, 0.f), 1.f);
#endif
    const real v = (real)1 - u;
    N[0] = N[1] = N[2] = N[3] = 0.f;
    if (!DERIV) {
        if (S == 0) { N[0] = 1.f; }
        else if (S == 1) { N[0] = (float)v; N[1] = (float)u; }
        else if (S == 2) { N[0] = (float)((real)0.5 * v * v); N[1] = (float)((real)0.5 * ((real)1 + (real)2 * u * v)); N[2] = (float)((real)0.5 * u * u); }
        else {
            const real u2 = u * u, u3 = u2 * u;
            const real k6 = (real)1 / (real)6;
            N[0] = (float)(k6 * v * v * v);
            N[1] = (float)(k6 * ((real)3 * u3 - (real)6 * u2 + (real)4));
            N[2] = (float)(k6 * ((real)-3 * u3 + (real)3 * u2 + (real)3 * u + (real)1));
            N[3] = (float)(k6 * u3);
        }
    } else {
        if (S == 1) { N[0] = (float)-ih; N[1] = (float)ih; }
        else if (S == 2) { N[0] = (float)(-v * ih); N[1] = (float)(((real)1 - (real)2 * u) * ih); N[2] = (float)(u * ih); }
        else if (S == 3) {
            const real u2 = u * u, hh = (real)0.5 * ih;
            N[0] = (float)(-hh * v * v);
            N[1] = (float)(hh * ((real)3 * u2 - (real)4 * u));
            N[2] = (float)(hh * ((real)-3 * u2 + (real)2 * u + (real)1));
            N[3] = (float)(hh * u2);
        }
    }
    return true;
}

// ---------------------------------------------------------------- plane expansion
// DERIV == false: v[p] = plane p of (xa, xb);  DERIV == true: v[p] = d plane_p / d input.
// xa feeds the base branch, xb feeds the basis (xa == xb except for FastKAN).
// KIND is a template parameter so that each kernel instantiation carries ONE basis family's code
// (all families inlined at every staging site made the kernels thrash the instruction cache).
// `c` is the channel inside its group; only families with per-channel parameters (ReLU-KAN) look at it.
template <int KIND, bool DERIV>
__device__ __forceinline__ void kan_planes(const DevBasis& bs, const float* tabs, float xa, float xb,
                                           float (&v)[KAN_PMAX], int c = 0) {
#pragma unroll
    for (int p = 0; p < KAN_PMAX; ++p) v[p] = 0.f;
    const int hb = bs.hb;
    if (hb) v[0] = DERIV ? kan_act_grad(bs.act, xa) : kan_act(bs.act, xa);

    if (KIND == KAN_BASIS_BSPLINE) {
        int j0; float N[4];
        if (bspline_uniform<DERIV>(bs.order, xb, tabs, bs.nb + bs.order + 1, bs.inv_h, j0, N)) {
#pragma unroll
            for (int p = 0; p < KAN_PMAX; ++p) {
                const int j = p - hb;                 // basis index of this plane
                const int d = j - j0;                 // position inside the local support
                float val = 0.f;
                val = d == 0 ? N[0] : val; val = d == 1 ? N[1] : val; val = d == 2 ? N[2] : val; val = d == 3 ? N[3] : val;
                if (j >= 0 && j < bs.nb && d <= bs.order) v[p] = val;
            }
        }
    } else if (KIND == KAN_BASIS_RBF) {
        // utils/utils.py:33  exp(-((x - c)/d)^2)
        const float dn = bs.p0;
#pragma unroll
        for (int p = 0; p < KAN_PMAX; ++p) {
            const int j = p - hb;
            if (j >= 0 && j < bs.nb) {
                float u = (xb - tabs[j]) / dn;
                float e = expf(-(u * u));
                v[p] = DERIV ? e * (-2.0f * u) / dn : e;
            }
        }
    } else if (KIND == KAN_BASIS_FOURIER) {
        // fourier_kan_layers.py:163-187: planes cos(k x), k = 1..G, then sin(k x), k = 1..G (nb = 2G).  ONE sincos(x); the harmonics by the
        // angle-addition recurrence (c_{k+1}, s_{k+1}) = (c_k c_1 - s_k s_1, s_k c_1 + c_k s_1), restarted at the head of the sine block -- a few ulp
        // per step against the reference's cos(fl(k x)), whose own argument rounding is of the same size (|k x| ulp); 2G sincosf calls per value
        // were most of this family's expansion time.
        const int G = bs.nb >> 1;
        float s1, c1;
        sincosf(xb, &s1, &c1);
        float ck = c1, sk = s1, kf = 1.f;
#pragma unroll
        for (int p = 0; p < KAN_PMAX; ++p) {
            const int j = p - hb;
            if (j >= 0 && j < bs.nb) {
                if (j == G) { ck = c1; sk = s1; kf = 1.f; }
                const bool is_cos = j < G;
                v[p] = is_cos ? (DERIV ? -kf * sk : ck) : (DERIV ? kf * ck : sk);
                const float cn = ck * c1 - sk * s1, sn = sk * c1 + ck * s1;
                ck = cn; sk = sn; kf += 1.f;
            }
        }
    } else if (KIND == KAN_BASIS_RELU) {
        // relu_kan_layers.py:127-131: x1 = relu(x - lo), x2 = relu(hi - x), q = x1 * x2 * r, plane = q * q, in that order.
        // q' terms: dq/dx = r (x2 - x1) wherever q != 0 (and 2q kills the rest), dq/dlo = -r x2, dq/dhi = r x1.
        // Non-derivative modes (bs.order): 0 value, 1 d/dlo, 2 d/dhi (base plane zero) -- the weight-gradient kernel run
        // on mode 1 / 2 gives the phase gradients their data-dependent factor.
        const float* lo = bs.ctab + (size_t)c * 2 * bs.nb;
        const float* hi = lo + bs.nb;
        const float r = bs.p0;
        const int mode = DERIV ? 3 : bs.order;
        if (hb && mode != 0 && !DERIV) v[0] = 0.f;
#pragma unroll
        for (int p = 0; p < KAN_PMAX; ++p) {
            const int j = p - hb;
            if (j >= 0 && j < bs.nb) {
                const float x1 = fmaxf(xb - lo[j], 0.f), x2 = fmaxf(hi[j] - xb, 0.f);
                const float q = x1 * x2 * r;
                const float q2 = 2.0f * q * r;
                v[p] = mode == 0 ? q * q : mode == 1 ? -(q2 * x2) : mode == 2 ? q2 * x1 : q2 * (x2 - x1);
            }
        }
    } else if (KIND == KAN_BASIS_GRAM) {
        // gram_kan_layers.py:156-182: planes act(P_k(t)), t = tanh x, P_0 = 1, P_1 = t, P_k = t P_{k-1} - c_k P_{k-2} with the
        // trainable c_k in device memory.  Alongside: D_k = dP_k/dt and, for mode m >= 1, Q_k = dP_k/dc_{m+1}
        // (Q_k = t Q_{k-1} - c_k Q_{k-2} - [k == m+1] P_{k-2}).
        const float* cf = bs.ctab;
        const float t = tanhf(xb), chain = 1.0f - t * t;
        const int mode = DERIV ? 0 : bs.order;
        if (hb && mode != 0) v[0] = 0.f;
        float Pm = 1.f, Pc = t, Dm = 0.f, Dc = 1.f, Qm = 0.f, Qc = 0.f;
#pragma unroll
        for (int p = 0; p < KAN_PMAX; ++p) {
            const int k = p - hb;
            if (k >= 0 && k < bs.nb) {
                const float P = k == 0 ? 1.f : Pc, D = k == 0 ? 0.f : Dc, Q = k == 0 ? 0.f : Qc;
                v[p] = DERIV ? kan_act_grad(bs.act, P) * D * chain : mode == 0 ? kan_act(bs.act, P) : kan_act_grad(bs.act, P) * Q;
                if (k >= 1 && k + 1 < bs.nb) {
                    const float c = cf[k + 1];
                    const float Pn = t * Pc - c * Pm, Dn = Pc + t * Dc - c * Dm;
                    const float Qn = t * Qc - c * Qm - (k + 1 == mode + 1 ? Pm : 0.f);
                    Pm = Pc; Pc = Pn; Dm = Dc; Dc = Dn; Qm = Qc; Qc = Qn;
                }
            }
        }
    } else if (KIND == KAN_BASIS_POLY) {
        // Three-term-recurrence families (bessel / fibonacci / gegenbauer / hermite / laguerre / lucas / taylor / jacobi
        // _kan_layers.py, compute_*_basis): on t = tanh(x) (order = 1) or t = x (order = 0),
        //   T_0 = c0,  T_1 = a1 t + b1,  T_k = (A_k t + B_k) T_{k-1} + C_k T_{k-2}   (k >= 2)
        // with the per-family coefficients precomputed on the host: tab = [c0, a1, b1, A_2, B_2, C_2, A_3, ...].
        // The derivative runs the differentiated recurrence alongside: T_k' = A_k T_{k-1} + (A_k t + B_k) T_{k-1}' + C_k T_{k-2}'.
        const float t = bs.order ? tanhf(xb) : xb;
        const float chain = bs.order ? (1.0f - t * t) : 1.0f;
        float Tm = tabs[0], Tc = tabs[1] * t + tabs[2];
        float Dm = 0.f, Dc = tabs[1];
#pragma unroll
        for (int p = 0; p < KAN_PMAX; ++p) {
            const int k = p - hb;
            if (k >= 0 && k < bs.nb) {
                if (k == 0) v[p] = DERIV ? 0.f : Tm;
                else {
                    v[p] = DERIV ? Dc * chain : Tc;
                    if (k + 1 < bs.nb) {
                        const float A = tabs[3 * k], B = tabs[3 * k + 1], Cc = tabs[3 * k + 2];      // coefficients of T_{k+1}
                        const float s = A * t + B;
                        const float Tn = s * Tc + Cc * Tm, Dn = A * Tc + s * Dc + Cc * Dm;
                        Tm = Tc; Tc = Tn; Dm = Dc; Dc = Dn;
                    }
                }
            }
        }
    } else {
        // cheby_kan_layers.py:93-96  T_k = cos(k * acos(t)), t = clamp(tanh x, lo, hi), evaluated by the three-term
        // recurrence T_k = 2 t T_{k-1} - T_{k-2} (max |delta| 9.8e-7 vs the reference for
        // degree 4).  Gradient as autograd forms it: dT_k/dx = k sin(k th)/sin(th) * (1 - tanh^2 x) inside the clamp
        // = k U_{k-1}(t) (1 - t0^2), zero where the clamp is active (clamp passes gradient on [lo, hi]).
        const float t0 = tanhf(xb);
        const float t = fminf(fmaxf(t0, bs.p0), bs.p1);
        const bool inside = (t0 >= bs.p0) && (t0 <= bs.p1);
        const float chain = inside ? (1.0f - t0 * t0) : 0.f;
        float Tm = 1.f, Tc = t;              // T_{k-1}, T_k   at k = 1
        float Um = 0.f, Uc = 1.f;            // U_{k-2}, U_{k-1} at k = 1  (U_{-1} = 0, U_0 = 1)
#pragma unroll
        for (int p = 0; p < KAN_PMAX; ++p) {
            const int k = p - hb;
            if (k >= 0 && k < bs.nb) {
                if (k == 0) v[p] = DERIV ? 0.f : 1.f;
                else {
                    v[p] = DERIV ? (float)k * Uc * chain : Tc;
                    const float Tn = 2.f * t * Tc - Tm; Tm = Tc; Tc = Tn;
                    const float Un = 2.f * t * Uc - Um; Um = Uc; Uc = Un;
                }
            }
        }
    }
}

// ---------------------------------------------------------------- plane expansion, streamed (round 3)
// The same planes as kan_planes, handed to `emit(p, value)` one at a time in increasing p by RUN-TIME loops.  The array form above is
// unrolled over all KAN_PMAX planes with run-time plane counts, which is what made the generic (run-time P) conv kernels 8 000 lines of ISA
// with 150 - 280 spilled scalar registers (each spill a v_readlane / v_writelane, i.e. a vector instruction next to the fp32 MFMAs).  The
// generic staging (stage_unit) and the generic bwd-data epilogue use this form; formulas and their order are kan_planes' own.
template <int KIND, bool DERIV, typename F>
__device__ __forceinline__ void kan_planes_each(const DevBasis& bs, const float* tabs, float xa, float xb, int c, F&& emit) {
    const int hb = bs.hb, nb = bs.nb;
    if (hb) {
        float base = DERIV ? kan_act_grad(bs.act, xa) : kan_act(bs.act, xa);
        if ((KIND == KAN_BASIS_RELU || KIND == KAN_BASIS_GRAM) && !DERIV && bs.order != 0) base = 0.f;      // parameter-derivative modes: base plane zero
        emit(0, base);
    }
    if (KIND == KAN_BASIS_BSPLINE) {
        int j0 = 0; float N[4];
        const bool ok = bspline_uniform<DERIV>(bs.order, xb, tabs, nb + bs.order + 1, bs.inv_h, j0, N);
#pragma unroll 1
        for (int j = 0; j < nb; ++j) {
            const int d = j - j0;
            float val = 0.f;
            val = d == 0 ? N[0] : val; val = d == 1 ? N[1] : val; val = d == 2 ? N[2] : val; val = d == 3 ? N[3] : val;
            emit(hb + j, (ok && d >= 0 && d <= bs.order) ? val : 0.f);
        }
    } else if (KIND == KAN_BASIS_RBF) {
        const float dn = bs.p0;
#pragma unroll 1
        for (int j = 0; j < nb; ++j) {
            const float u = (xb - tabs[j]) / dn;
            const float e = expf(-(u * u));
            emit(hb + j, DERIV ? e * (-2.0f * u) / dn : e);
        }
    } else if (KIND == KAN_BASIS_FOURIER) {
        const int G = nb >> 1;
        float s1, c1;
        sincosf(xb, &s1, &c1);
        float ck = c1, sk = s1, kf = 1.f;
#pragma unroll 1
        for (int j = 0; j < nb; ++j) {
            if (j == G) { ck = c1; sk = s1; kf = 1.f; }
            const bool is_cos = j < G;
            emit(hb + j, is_cos ? (DERIV ? -kf * sk : ck) : (DERIV ? kf * ck : sk));
            const float cn = ck * c1 - sk * s1, sn = sk * c1 + ck * s1;
            ck = cn; sk = sn; kf += 1.f;
        }
    } else if (KIND == KAN_BASIS_RELU) {
        const float* lo = bs.ctab + (size_t)c * 2 * nb;
        const float* hi = lo + nb;
        const float r = bs.p0;
        const int mode = DERIV ? 3 : bs.order;
#pragma unroll 1
        for (int j = 0; j < nb; ++j) {
            const float x1 = fmaxf(xb - lo[j], 0.f), x2 = fmaxf(hi[j] - xb, 0.f);
            const float q = x1 * x2 * r;
            const float q2 = 2.0f * q * r;
            emit(hb + j, mode == 0 ? q * q : mode == 1 ? -(q2 * x2) : mode == 2 ? q2 * x1 : q2 * (x2 - x1));
        }
    } else if (KIND == KAN_BASIS_GRAM) {
        const float* cf = bs.ctab;
        const float t = tanhf(xb), chain = 1.0f - t * t;
        const int mode = DERIV ? 0 : bs.order;
        float Pm = 1.f, Pc = t, Dm = 0.f, Dc = 1.f, Qm = 0.f, Qc = 0.f;
#pragma unroll 1
        for (int k = 0; k < nb; ++k) {
            const float P = k == 0 ? 1.f : Pc, D = k == 0 ? 0.f : Dc, Q = k == 0 ? 0.f : Qc;
            emit(hb + k, DERIV ? kan_act_grad(bs.act, P) * D * chain : mode == 0 ? kan_act(bs.act, P) : kan_act_grad(bs.act, P) * Q);
            if (k >= 1 && k + 1 < nb) {
                const float cc = cf[k + 1];
                const float Pn = t * Pc - cc * Pm, Dn = Pc + t * Dc - cc * Dm;
                const float Qn = t * Qc - cc * Qm - (k + 1 == mode + 1 ? Pm : 0.f);
                Pm = Pc; Pc = Pn; Dm = Dc; Dc = Dn; Qm = Qc; Qc = Qn;
            }
        }
    } else if (KIND == KAN_BASIS_POLY) {
        const float t = bs.order ? tanhf(xb) : xb;
        const float chain = bs.order ? (1.0f - t * t) : 1.0f;
        float Tm = tabs[0], Tc = tabs[1] * t + tabs[2];
        float Dm = 0.f, Dc = tabs[1];
        emit(hb, DERIV ? 0.f : Tm);
#pragma unroll 1
        for (int k = 1; k < nb; ++k) {
            emit(hb + k, DERIV ? Dc * chain : Tc);
            if (k + 1 < nb) {
                const float A = tabs[3 * k], B = tabs[3 * k + 1], Cc = tabs[3 * k + 2];      // coefficients of T_{k+1}
                const float sA = A * t + B;
                const float Tn = sA * Tc + Cc * Tm, Dn = A * Tc + sA * Dc + Cc * Dm;
                Tm = Tc; Tc = Tn; Dm = Dc; Dc = Dn;
            }
        }
    } else {      // Chebyshev
        const float t0 = tanhf(xb);
        const float t = fminf(fmaxf(t0, bs.p0), bs.p1);
        const bool inside = (t0 >= bs.p0) && (t0 <= bs.p1);
        const float chain = inside ? (1.0f - t0 * t0) : 0.f;
        float Tm = 1.f, Tc = t, Um = 0.f, Uc = 1.f;
        emit(hb, DERIV ? 0.f : 1.f);
#pragma unroll 1
        for (int k = 1; k < nb; ++k) {
            emit(hb + k, DERIV ? (float)k * Uc * chain : Tc);
            const float Tn = 2.f * t * Tc - Tm; Tm = Tc; Tc = Tn;
            const float Un = 2.f * t * Uc - Um; Um = Uc; Uc = Un;
        }
    }
}
