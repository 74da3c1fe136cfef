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
    float tab[KAN_MAX_TABLE];
};

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

// ---------------------------------------------------------------- B-spline (local de Boor)
// Restates kan_layers.py:209-233 for the <= S+1 bases that are non-zero at x: find the knot
// interval [g_i, g_{i+1}) holding x with the SAME fp32 knots and half-open comparisons as
// the reference's order-0 indicator, then run the Cox-de Boor recurrence on that interval
// only.  N[r] is basis j0 + r, j0 = i - S.  With DERIV, D[r] is its derivative
//   S * ( N_{j,S-1}/(g_{j+S}-g_j) - N_{j+1,S-1}/(g_{j+S+1}-g_{j+1}) ),
// which is what autograd of the reference recursion evaluates to (the indicator has zero
// gradient).  `kn` is the knot table in LDS, nkn = n_basis + S + 1 knots.
template <int S, bool DERIV>
__device__ __forceinline__ bool bspline_local(float x, const float* kn, int nkn, float inv_h,
                                              int& j0, float (&N)[4], float (&D)[4]) {
    const int NI = nkn - 1;                      // number of knot intervals
    if (!(x >= kn[0] && x < kn[NI])) return false;   // also rejects NaN, as the indicator does
    int i = (int)floorf((x - kn[0]) * inv_h);
    i = min(max(i, 0), NI - 1);
#pragma unroll
    for (int it = 0; it < 2; ++it) {             // settle on the reference's own comparisons
        if (x < kn[i]) i = max(i - 1, 0);
        else if (x >= kn[i + 1]) i = min(i + 1, NI - 1);
    }
    j0 = i - S;
    float left[5], right[5];
#pragma unroll
    for (int j = 1; j <= S + 1; ++j) {
        left[j]  = x - kn[min(max(i + 1 - j, 0), NI)];
        right[j] = kn[min(max(i + j, 0), NI)] - x;
    }
    N[0] = 1.f; N[1] = 0.f; N[2] = 0.f; N[3] = 0.f;
    float Nm[4] = {1.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 1; j <= S; ++j) {
        if (DERIV && j == S) {
#pragma unroll
            for (int q = 0; q < 4; ++q) Nm[q] = N[q];
        }
        float saved = 0.f;
#pragma unroll
        for (int r = 0; r < j; ++r) {
            float temp = N[r] / (right[r + 1] + left[j - r]);
            N[r] = saved + right[r + 1] * temp;
            saved = left[j - r] * temp;
        }
        N[j] = saved;
    }
    if (DERIV) {
        D[0] = D[1] = D[2] = D[3] = 0.f;
        if (S > 0) {
#pragma unroll
            for (int r = 0; r <= S; ++r) {
                float a = 0.f, b = 0.f;
                if (r >= 1) a = Nm[r - 1] / (right[r] + left[S - r + 1]);
                if (r <= S - 1) b = Nm[r] / (right[r + 1] + left[S - r]);
                D[r] = (float)S * (a - b);
            }
        }
    }
    return true;
}

template <bool DERIV>
__device__ __forceinline__ bool bspline_dispatch(int S, float x, const float* kn, int nkn, float inv_h,
                                                 int& j0, float (&N)[4], float (&D)[4]) {
    switch (S) {
        case 0:  return bspline_local<0, DERIV>(x, kn, nkn, inv_h, j0, N, D);
        case 1:  return bspline_local<1, DERIV>(x, kn, nkn, inv_h, j0, N, D);
        case 2:  return bspline_local<2, DERIV>(x, kn, nkn, inv_h, j0, N, D);
        default: return bspline_local<3, DERIV>(x, kn, nkn, inv_h, j0, N, D);
    }
}

// ---------------------------------------------------------------- plane expansion
// DERIV == false: v[p] = plane p of (xa, xb);  DERIV == true: v[p] = d plane_p / d input.
// xa feeds the base branch, xb feeds the basis (xa == xb except for FastKAN).
template <bool DERIV>
__device__ __forceinline__ void kan_planes(const DevBasis& bs, const float* tabs, float xa, float xb,
                                           float (&v)[KAN_PMAX]) {
#pragma unroll
    for (int p = 0; p < KAN_PMAX; ++p) v[p] = 0.f;
    const int hb = bs.hb;
    if (hb) v[0] = DERIV ? kan_act_grad(bs.act, xa) : kan_act(bs.act, xa);

    if (bs.kind == KAN_BASIS_BSPLINE) {
        int j0; float N[4], D[4];
        if (bspline_dispatch<DERIV>(bs.order, xb, tabs, bs.nb + bs.order + 1, bs.inv_h, j0, N, D)) {
#pragma unroll
            for (int p = 0; p < KAN_PMAX; ++p) {
                const int j = p - hb;                 // basis index of this plane
                const int d = j - j0;                 // position inside the local support
                float val = 0.f;
                if (DERIV) { val = d == 0 ? D[0] : val; val = d == 1 ? D[1] : val; val = d == 2 ? D[2] : val; val = d == 3 ? D[3] : val; }
                else       { val = d == 0 ? N[0] : val; val = d == 1 ? N[1] : val; val = d == 2 ? N[2] : val; val = d == 3 ? N[3] : val; }
                if (j >= 0 && j < bs.nb && d <= bs.order) v[p] = val;
            }
        }
    } else if (bs.kind == KAN_BASIS_RBF) {
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
    } else {
        // cheby_kan_layers.py:93-96  cos(k * acos(clamp(tanh x)))
        const float t0 = tanhf(xb);
        const float t = fminf(fmaxf(t0, bs.p0), bs.p1);
        const float th = acosf(t);
        float chain = 0.f;                            // d theta / d x  (clamp passes grad on [lo, hi])
        if (DERIV) {
            const bool inside = (t0 >= bs.p0) && (t0 <= bs.p1);
            chain = inside ? -rsqrtf(1.0f - t * t) * (1.0f - t0 * t0) : 0.f;
        }
#pragma unroll
        for (int p = 0; p < KAN_PMAX; ++p) {
            const int j = p - hb;
            if (j >= 0 && j < bs.nb) {
                const float kf = (float)j;
                v[p] = DERIV ? (-sinf(kf * th) * kf) * chain : cosf(kf * th);
            }
        }
    }
}
