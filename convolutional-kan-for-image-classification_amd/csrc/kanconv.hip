// libkanconv: conv-KAN forward / backward for MI355X (gfx950, CDNA4).  C ABI in include/kanconv.h.
//
// One implicit GEMM per direction, all on the exact-fp32 matrix pipe (v_mfma_f32_32x32x2_f32):
//
//   forward      z [o][pixel]   = sum_k  Wp[k][o]        * E[k][pixel]      k = (c, tap, plane)
//   bwd-data     G [(c,p)][pix] = sum_kd Wp[(c,tap,p)][o] * dz[o][pix(+)tap] kd = (tap, o)
//                dx[c][pix]     = sum_p  plane_p'(x) * G[(c,p)][pix]        (LDS epilogue)
//   bwd-weight   dWp[k][o]      = sum_pix E[k][pixel]    * dz[o][pixel]
//
// E is never materialised in HBM: workgroups gather x, expand it to its P planes in registers
// (kan_device.h) and write the expanded tile straight into LDS in GEMM order; out-of-image taps
// write zeros (the reference zero-pads the EXPANDED tensor, kan_layers.py:239).
//
// Accumulator orientation: MFMA C/D puts the column index on the lane, so the dimension that is
// contiguous in HBM (pixels for z / dx, o for dWp) is always the column => 128-B coalesced stores.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include "kanconv.h"
#include "kan_device.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

namespace {

constexpr int KC = 32;                      // GEMM-depth rows staged in LDS per step

thread_local char g_err[512] = "";
int fail(const char* fmt, const char* a = "") {
    snprintf(g_err, sizeof(g_err), fmt, a);
    return -1;
}

struct DevGeom {
    int B, C, H, W, O, Ho, Wo, kh, kw, sh, sw, ph, pw, dh, dw;
    long long xbs, ybs;
};

// C/D register -> row inside a 32x32 MFMA tile (cdna guide section 3)
__device__ __forceinline__ int mfma_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// ============================================================================ pack / unpack
// Reference weight layouts -> Wp[((c*T+tap)*P+p)][o].  A 32x32 tile goes through LDS so that both
// the read (along the source's contiguous (channel,tap) axis) and the write (along o) coalesce.
// src_kind 0: base weights [O][C][T]        -> plane 0
// src_kind 1: basis weights [O][C*nb][T]    -> plane hb + q
template <bool UNPACK>
__global__ __launch_bounds__(256) void k_pack(const float* __restrict__ src_c, float* __restrict__ dst_c,
                                              float* __restrict__ src_m, const float* __restrict__ wp_in, float* __restrict__ wp_out,
                                              int O, int C, int T, int P, int hb, int nb, int src_kind, int Opad,
                                              int n_slabs, long long slab_elems) {
    __shared__ float tile[32][33];
    const int J = src_kind == 0 ? C * T : C * nb * T;       // source row length
    const int j0 = blockIdx.x * 32, o0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    auto kof = [&](int j) -> int {
        int cq = j / T, tap = j - cq * T;
        int c = src_kind == 0 ? cq : cq / nb;
        int q = src_kind == 0 ? 0 : cq - c * nb;
        int p = src_kind == 0 ? 0 : hb + q;
        return (c * T + tap) * P + p;
    };
    if (!UNPACK) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int o = o0 + ty + 8 * i, j = j0 + tx;
            tile[ty + 8 * i][tx] = (o < O && j < J) ? src_c[(size_t)o * J + j] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int j = j0 + ty + 8 * i, o = o0 + tx;
            if (j < J && o < Opad) wp_out[(size_t)kof(j) * Opad + o] = tile[tx][ty + 8 * i];
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int j = j0 + ty + 8 * i, o = o0 + tx;
            float s = 0.f;
            if (j < J && o < O) {
                size_t a = (size_t)kof(j) * Opad + o;
                for (int sl = 0; sl < n_slabs; ++sl) s += wp_in[a + (size_t)sl * slab_elems];
            }
            tile[ty + 8 * i][tx] = s;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int o = o0 + ty + 8 * i, j = j0 + tx;
            if (o < O && j < J) src_m[(size_t)o * J + j] = tile[tx][ty + 8 * i];
        }
    }
}

// ============================================================================ forward
// Workgroup tile: TO = WO*64 outputs x TP = WP*64 output pixels, one 64x64 wave tile per wave
// (2x2 MFMA 32x32x2).  Per step: KC rows of Wp -> sW, the matching KC rows of E -> sE.
template <int WO, int WP>
__global__ __launch_bounds__(WO * WP * 64, 2) void k_conv_fwd(
    const float* __restrict__ x, const float* __restrict__ xn, const float* __restrict__ wp, float* __restrict__ z,
    DevGeom g, DevBasis bs, int Opad, int n_chunks, int chunks_per_split, long long slab_elems) {
    constexpr int TO = WO * 64, TP = WP * 64, NT = WO * WP * 64;
    __shared__ float sW[KC * TO];
    __shared__ float sE[KC * TP];
    __shared__ float sTab[KAN_MAX_TABLE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int w_o = wave / WP, w_p = wave % WP;
    if (tid < KAN_MAX_TABLE) sTab[tid] = bs.tab[tid];

    const int HoWo = g.Ho * g.Wo, HW = g.H * g.W, T = g.kh * g.kw, P = bs.P;
    const int Mtot = g.B * HoWo;
    const int px_tile0 = blockIdx.x * TP, o_tile0 = blockIdx.y * TO;

    // the pixel this thread expands for (fixed for the whole kernel: NT % TP == 0)
    const int my_px = px_tile0 + (tid % TP);
    const bool pv = my_px < Mtot;
    int hi0, wi0; long long xoff;
    {
        int b = my_px / HoWo, hw = my_px - b * HoWo;
        int ho = hw / g.Wo, wo = hw - ho * g.Wo;
        hi0 = ho * g.sh - g.ph; wi0 = wo * g.sw - g.pw;
        xoff = (long long)b * g.xbs;
    }
    const bool same_in = (x == xn);

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int ch0 = blockIdx.z * chunks_per_split;
    const int ch1 = min(n_chunks, ch0 + chunks_per_split);
    __syncthreads();

    for (int ch = ch0; ch < ch1; ++ch) {
        const int k0 = ch * KC;
        // ---- weights: KC x TO floats, 16-B loads along o
        for (int i = tid; i < KC * TO / 4; i += NT) {
            int row = i / (TO / 4), c4 = i - row * (TO / 4);
            float4 v = *reinterpret_cast<const float4*>(wp + (size_t)(k0 + row) * Opad + o_tile0 + c4 * 4);
            *reinterpret_cast<float4*>(&sW[row * TO + c4 * 4]) = v;
        }
        // ---- expanded operand: items (c, tap) overlapping rows [k0, k0+KC)
        const int item_first = k0 / P;
        const int n_items = (k0 + KC - 1) / P - item_first + 1;
        for (int il = tid / TP; il < n_items; il += NT / TP) {
            const int item = __builtin_amdgcn_readfirstlane(item_first + il);   // wave-uniform
            const int c = item / T, tap = item - c * T;
            const int r = tap / g.kw, t = tap - r * g.kw;
            const int hi = hi0 + r * g.dh, wi = wi0 + t * g.dw;
            const bool inb = pv && c < g.C && (unsigned)hi < (unsigned)g.H && (unsigned)wi < (unsigned)g.W;
            float v[KAN_PMAX];
            if (inb) {
                long long idx = xoff + (long long)c * HW + hi * g.W + wi;
                float xa = x[idx];
                float xb = same_in ? xa : xn[idx];
                kan_planes<false>(bs, sTab, xa, xb, v);
            } else {
#pragma unroll
                for (int p = 0; p < KAN_PMAX; ++p) v[p] = 0.f;
            }
            const int rbase = item * P - k0;
#pragma unroll
            for (int p = 0; p < KAN_PMAX; ++p) {
                if (p < P) {
                    int row = rbase + p;
                    if ((unsigned)row < (unsigned)KC) sE[row * TP + (tid % TP)] = v[p];
                }
            }
        }
        __syncthreads();
        // ---- MFMA: D[o][pixel] += W^T[o][k] * E[k][pixel]
        const int ao = w_o * 64 + (lane & 31), bp = w_p * 64 + (lane & 31), kh2 = lane >> 5;
#pragma unroll
        for (int kk = 0; kk < KC / 2; ++kk) {
            const int krow = 2 * kk + kh2;
            float a0 = sW[krow * TO + ao], a1 = sW[krow * TO + ao + 32];
            float b0 = sE[krow * TP + bp], b1 = sE[krow * TP + bp + 32];
            acc[0][0] = MFMA32(a0, b0, acc[0][0]);
            acc[0][1] = MFMA32(a0, b1, acc[0][1]);
            acc[1][0] = MFMA32(a1, b0, acc[1][0]);
            acc[1][1] = MFMA32(a1, b1, acc[1][1]);
        }
        __syncthreads();
    }

    // ---- store: column (lane) = pixel => coalesced along the plane
    float* zs = z + (size_t)blockIdx.z * slab_elems;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int px = px_tile0 + w_p * 64 + ni * 32 + (lane & 31);
        if (px >= Mtot) continue;
        const int b = px / HoWo, hw = px - b * HoWo;
        float* zb = zs + (size_t)b * g.ybs + hw;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = o_tile0 + w_o * 64 + mi * 32 + mfma_row(r, lane);
                if (o < g.O) zb[(size_t)o * HoWo] = acc[mi][ni][r];
            }
        }
    }
}

// ============================================================================ backward data
// Tile: 128 rows = CT channels x P planes (flat c_local*P + p, CT = 128 / P) x 128 input pixels.
// Depth chunks: (tap, 32 outputs).  Epilogue: G tile -> LDS, then dx = sum_p plane_p'(x) * G_p.
__global__ __launch_bounds__(256, 2) void k_conv_bwd_data(
    const float* __restrict__ dz, const float* __restrict__ x, const float* __restrict__ xn, const float* __restrict__ wp,
    float* __restrict__ dx, float* __restrict__ dxn, DevGeom g, DevBasis bs, int Opad, int CT, int n_ob,
    int n_chunks, int chunks_per_split, long long slab_elems) {
    constexpr int TR = 128, TP = 128, LDW = TR + 1;
    __shared__ float smem[TR * TP];                 // staging (sW: KC x LDW, sG: KC x TP) / epilogue (TR x TP)
    __shared__ int sRowK[TR];
    __shared__ float sTab[KAN_MAX_TABLE];
    float* sW = smem;
    float* sG = smem + KC * LDW;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int w_r = wave >> 1, w_p = wave & 1;
    const int HoWo = g.Ho * g.Wo, HW = g.H * g.W, T = g.kh * g.kw, P = bs.P;
    const int Min = g.B * HW;
    const int px_tile0 = blockIdx.x * TP, c_tile0 = blockIdx.y * CT;

    if (tid < KAN_MAX_TABLE) sTab[tid] = bs.tab[tid];
    if (tid < TR) {
        int cl = tid / P, p = tid - cl * P, c = c_tile0 + cl;
        sRowK[tid] = (cl < CT && c < g.C) ? (c * T) * P + p : -1;
    }
    const int my_px = px_tile0 + (tid & 127);
    const bool pv = my_px < Min;
    int pb, ph_, pw_;
    {
        pb = my_px / HW; int hw = my_px - pb * HW;
        ph_ = hw / g.W; pw_ = hw - ph_ * g.W;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int ch0 = blockIdx.z * chunks_per_split;
    const int ch1 = min(n_chunks, ch0 + chunks_per_split);
    __syncthreads();

    for (int ch = ch0; ch < ch1; ++ch) {
        const int tap = ch / n_ob, o0 = (ch - tap * n_ob) * KC;
        const int r = tap / g.kw, t = tap - r * g.kw;
        // ---- weights: sW[o_local][row], 32 consecutive o per (c,p) row = one 128-B segment
        {
            const int ol = tid & 31;
            for (int row = tid >> 5; row < TR; row += 8) {
                const int kb = sRowK[row];
                float v = 0.f;
                if (kb >= 0 && o0 + ol < g.O) v = wp[(size_t)(kb + tap * P) * Opad + o0 + ol];
                sW[ol * LDW + row] = v;
            }
        }
        // ---- dz gathered at the output position this (input pixel, tap) pair feeds
        {
            int hn = ph_ + g.ph - r * g.dh, wn = pw_ + g.pw - t * g.dw;
            int ho = hn / g.sh, wo = wn / g.sw;
            bool ok = pv && hn >= 0 && wn >= 0 && ho * g.sh == hn && wo * g.sw == wn && ho < g.Ho && wo < g.Wo;
            const float* src = dz + (size_t)pb * g.ybs + (size_t)ho * g.Wo + wo;
            for (int ol = tid >> 7; ol < KC; ol += 2) {
                float v = 0.f;
                if (ok && o0 + ol < g.O) v = src[(size_t)(o0 + ol) * HoWo];
                sG[ol * TP + (tid & 127)] = v;
            }
        }
        __syncthreads();
        const int ar = w_r * 64 + (lane & 31), bp = w_p * 64 + (lane & 31), kh2 = lane >> 5;
#pragma unroll
        for (int kk = 0; kk < KC / 2; ++kk) {
            const int krow = 2 * kk + kh2;
            float a0 = sW[krow * LDW + ar], a1 = sW[krow * LDW + ar + 32];
            float b0 = sG[krow * TP + bp], b1 = sG[krow * TP + bp + 32];
            acc[0][0] = MFMA32(a0, b0, acc[0][0]);
            acc[0][1] = MFMA32(a0, b1, acc[0][1]);
            acc[1][0] = MFMA32(a1, b0, acc[1][0]);
            acc[1][1] = MFMA32(a1, b1, acc[1][1]);
        }
        __syncthreads();
    }

    // ---- epilogue: G tile to LDS, contract the P planes of each channel with plane'(x)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = w_r * 64 + mi * 32 + mfma_row(r, lane);
                int col = w_p * 64 + ni * 32 + (lane & 31);
                smem[row * TP + col] = acc[mi][ni][r];
            }
    __syncthreads();
    const bool same_in = (x == xn);
    const bool split_out = (dxn != nullptr);
    float* dxs = dx + (size_t)blockIdx.z * slab_elems;
    float* dxns = split_out ? dxn + (size_t)blockIdx.z * slab_elems : nullptr;
    for (int cl = tid >> 7; cl < CT; cl += 2) {
        const int c = c_tile0 + cl;
        if (c >= g.C || !pv) continue;
        const size_t idx = (size_t)pb * g.xbs + (size_t)c * HW + (size_t)(ph_ * g.W + pw_);
        const float xa = x[idx];
        const float xb = same_in ? xa : xn[idx];
        float d[KAN_PMAX];
        kan_planes<true>(bs, sTab, xa, xb, d);
        float s_base = 0.f, s_bas = 0.f;
#pragma unroll
        for (int p = 0; p < KAN_PMAX; ++p) {
            if (p < P) {
                float gv = smem[(cl * P + p) * TP + (tid & 127)];
                if (p < bs.hb) s_base += d[p] * gv; else s_bas += d[p] * gv;
            }
        }
        if (split_out) { dxs[idx] = s_base; dxns[idx] = s_bas; }
        else dxs[idx] = s_base + s_bas;
    }
}

// ============================================================================ backward weight
// Tile: TR = WR*64 rows of the packed K axis x TO = WC*64 outputs; depth chunks of 32 output pixels.
template <int WR, int WC>
__global__ __launch_bounds__(WR * WC * 64, 2) void k_conv_bwd_weight(
    const float* __restrict__ dz, const float* __restrict__ x, const float* __restrict__ xn, float* __restrict__ dwp,
    DevGeom g, DevBasis bs, int Kpad, int Opad, int n_chunks, int chunks_per_split, long long slab_elems) {
    constexpr int TR = WR * 64, TO = WC * 64, NT = WR * WC * 64, LDE = TR + 1, LDZ = TO + 1;
    constexpr int MAXI = TR + 2;                    // items overlapping a row tile
    __shared__ float sE[KC * LDE];
    __shared__ float sZ[KC * LDZ];
    __shared__ int sItem[MAXI];                     // c | r<<16 | t<<24, or -1
    __shared__ float sTab[KAN_MAX_TABLE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int w_r = wave / WC, w_c = wave % WC;
    const int HoWo = g.Ho * g.Wo, HW = g.H * g.W, T = g.kh * g.kw, P = bs.P;
    const int Mtot = g.B * HoWo, NI = g.C * T;
    const int k0 = blockIdx.x * TR, o_tile0 = blockIdx.y * TO;
    const int item_first = k0 / P;
    const int n_items = (k0 + TR - 1) / P - item_first + 1;

    if (tid < KAN_MAX_TABLE) sTab[tid] = bs.tab[tid];
    for (int i = tid; i < n_items; i += NT) {
        int item = item_first + i, v = -1;
        if (item < NI) {
            int c = item / T, tap = item - c * T, r = tap / g.kw, t = tap - r * g.kw;
            v = c | (r << 16) | (t << 24);
        }
        sItem[i] = v;
    }
    const bool same_in = (x == xn);

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int ch0 = blockIdx.z * chunks_per_split;
    const int ch1 = min(n_chunks, ch0 + chunks_per_split);
    __syncthreads();

    for (int ch = ch0; ch < ch1; ++ch) {
        const int pl = tid & 31;
        const int px = ch * KC + pl;
        const bool pv = px < Mtot;
        const int b = px / HoWo, hw = px - b * HoWo;
        const int ho = hw / g.Wo, wo = hw - ho * g.Wo;
        const int hi0 = ho * g.sh - g.ph, wi0 = wo * g.sw - g.pw;
        // ---- expanded operand sE[pixel][row]
        for (int il = tid >> 5; il < n_items; il += NT / 32) {
            const int it = sItem[il];
            const int c = it & 0xffff, r = (it >> 16) & 0xff, t = (it >> 24) & 0xff;
            const int hi = hi0 + r * g.dh, wi = wi0 + t * g.dw;
            const bool inb = pv && it >= 0 && (unsigned)hi < (unsigned)g.H && (unsigned)wi < (unsigned)g.W;
            float v[KAN_PMAX];
            if (inb) {
                size_t idx = (size_t)b * g.xbs + (size_t)c * HW + (size_t)(hi * g.W + wi);
                float xa = x[idx];
                float xb = same_in ? xa : xn[idx];
                kan_planes<false>(bs, sTab, xa, xb, v);
            } else {
#pragma unroll
                for (int p = 0; p < KAN_PMAX; ++p) v[p] = 0.f;
            }
            const int rbase = (item_first + il) * P - k0;
#pragma unroll
            for (int p = 0; p < KAN_PMAX; ++p) {
                if (p < P) {
                    int row = rbase + p;
                    if ((unsigned)row < (unsigned)TR) sE[pl * LDE + row] = v[p];
                }
            }
        }
        // ---- dz tile sZ[pixel][o]
        {
            const float* src = dz + (size_t)b * g.ybs + hw;
            for (int ol = tid >> 5; ol < TO; ol += NT / 32) {
                int o = o_tile0 + ol;
                sZ[pl * LDZ + ol] = (pv && o < g.O) ? src[(size_t)o * HoWo] : 0.f;
            }
        }
        __syncthreads();
        const int ar = w_r * 64 + (lane & 31), bo = w_c * 64 + (lane & 31), kh2 = lane >> 5;
#pragma unroll
        for (int kk = 0; kk < KC / 2; ++kk) {
            const int krow = 2 * kk + kh2;
            float a0 = sE[krow * LDE + ar], a1 = sE[krow * LDE + ar + 32];
            float b0 = sZ[krow * LDZ + bo], b1 = sZ[krow * LDZ + bo + 32];
            acc[0][0] = MFMA32(a0, b0, acc[0][0]);
            acc[0][1] = MFMA32(a0, b1, acc[0][1]);
            acc[1][0] = MFMA32(a1, b0, acc[1][0]);
            acc[1][1] = MFMA32(a1, b1, acc[1][1]);
        }
        __syncthreads();
    }

    float* out = dwp + (size_t)blockIdx.z * slab_elems;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = k0 + w_r * 64 + mi * 32 + mfma_row(r, lane);
            if (row >= Kpad) continue;
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int o = o_tile0 + w_c * 64 + ni * 32 + (lane & 31);
                out[(size_t)row * Opad + o] = acc[mi][ni][r];
            }
        }
}

// ============================================================================ slab reduce
__global__ __launch_bounds__(256) void k_slab_reduce(const float* __restrict__ slabs, int n_slabs, long long slab_elems,
                                                     float* __restrict__ out, int Cn, int HW, long long bstride, long long total) {
    const long long plane = (long long)Cn * HW;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        long long b = i / plane, rem = i - b * plane;
        size_t a = (size_t)b * bstride + rem;
        float s = 0.f;
        for (int sl = 0; sl < n_slabs; ++sl) s += slabs[a + (size_t)sl * slab_elems];
        out[a] = s;
    }
}

// ============================================================================ InstanceNorm + PReLU
// G lanes cooperate on one (b, channel) plane; 256/G planes per workgroup.
template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int off = G / 2; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <int G>
__global__ __launch_bounds__(256) void k_in_prelu_fwd(const float* __restrict__ z, int n_slabs, long long slab_elems,
                                                      float* __restrict__ z_out, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, const float* __restrict__ prelu_a,
                                                      float* __restrict__ y, float* __restrict__ mean_o, float* __restrict__ rstd_o,
                                                      int n_planes, int Cn, int HW, long long bstride, float eps) {
    const int tid = threadIdx.x, sub = tid % G;
    const int plane = blockIdx.x * (256 / G) + tid / G;
    const bool act = plane < n_planes;
    const int b = act ? plane / Cn : 0, c = act ? plane - b * Cn : 0;
    const size_t base = (size_t)b * bstride + (size_t)c * HW;
    const bool need_sum = (n_slabs > 1) || (z_out != z);
    float s = 0.f;
    if (act) for (int i = sub; i < HW; i += G) {
        float v = z[base + i];
        for (int sl = 1; sl < n_slabs; ++sl) v += z[base + i + (size_t)sl * slab_elems];
        if (need_sum) z_out[base + i] = v;
        s += v;
    }
    const float mu = group_sum<G>(s) / (float)HW;
    float q = 0.f;
    if (act) for (int i = sub; i < HW; i += G) { float d = z_out[base + i] - mu; q += d * d; }
    const float var = group_sum<G>(q) / (float)HW;
    const float rs = 1.0f / sqrtf(var + eps);
    if (!act) return;
    const float ga = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
    const bool has_p = prelu_a != nullptr;
    const float a = has_p ? prelu_a[0] : 1.f;
    for (int i = sub; i < HW; i += G) {
        float n = (z_out[base + i] - mu) * rs * ga + be;
        y[base + i] = (has_p && !(n > 0.f)) ? a * n : n;
    }
    if (sub == 0) { mean_o[plane] = mu; rstd_o[plane] = rs; }
}

template <int G>
__global__ __launch_bounds__(256) void k_in_prelu_bwd(const float* __restrict__ dy, const float* __restrict__ z,
                                                      const float* __restrict__ mean_i, const float* __restrict__ rstd_i,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ prelu_a, float* __restrict__ dz,
                                                      float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dprelu,
                                                      int n_planes, int Cn, int HW, long long bstride) {
    __shared__ float s_da[256 / G];
    const int tid = threadIdx.x, sub = tid % G;
    const int plane = blockIdx.x * (256 / G) + tid / G;
    const bool act = plane < n_planes;
    const int b = act ? plane / Cn : 0, c = act ? plane - b * Cn : 0;
    const size_t base = (size_t)b * bstride + (size_t)c * HW;
    const float mu = act ? mean_i[plane] : 0.f, rs = act ? rstd_i[plane] : 0.f;
    const float ga = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
    const bool has_p = prelu_a != nullptr;
    const float a = has_p ? prelu_a[0] : 1.f;
    float s1 = 0.f, s2 = 0.f, sa = 0.f, sg = 0.f, sb = 0.f;
    if (act) for (int i = sub; i < HW; i += G) {
        float nh = (z[base + i] - mu) * rs;
        float n = nh * ga + be;
        float g = dy[base + i];
        bool neg = has_p && !(n > 0.f);
        float dn = neg ? a * g : g;
        if (neg) sa += n * g;
        sb += dn; sg += dn * nh;
        float dnh = dn * ga;
        s1 += dnh; s2 += dnh * nh;
    }
    s1 = group_sum<G>(s1); s2 = group_sum<G>(s2);
    sa = group_sum<G>(sa); sg = group_sum<G>(sg); sb = group_sum<G>(sb);
    const float m1 = s1 / (float)HW, m2 = s2 / (float)HW;
    if (act) for (int i = sub; i < HW; i += G) {
        float nh = (z[base + i] - mu) * rs;
        float n = nh * ga + be;
        float g = dy[base + i];
        bool neg = has_p && !(n > 0.f);
        float dnh = (neg ? a * g : g) * ga;
        dz[base + i] = rs * (dnh - m1 - nh * m2);
    }
    if (act && sub == 0) {
        if (dgamma) atomicAdd(&dgamma[c], sg);
        if (dbeta) atomicAdd(&dbeta[c], sb);
    }
    if (dprelu) {                                   // one atomic per workgroup
        if (sub == 0) s_da[tid / G] = act ? sa : 0.f;
        __syncthreads();
        if (tid == 0) {
            float t = 0.f;
            for (int i = 0; i < 256 / G; ++i) t += s_da[i];
            atomicAdd(dprelu, t);
        }
    }
}

// ============================================================================ host side
int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }
int round_up(int a, int b) { return (a + b - 1) / b * b; }

int check(const KanGeom* g, const KanBasis* b) {
    if (!g || !b) return fail("null geometry/basis");
    if (g->B <= 0 || g->C <= 0 || g->O <= 0 || g->H <= 0 || g->W <= 0 || g->Ho <= 0 || g->Wo <= 0) return fail("non-positive dimension");
    if (g->kh <= 0 || g->kw <= 0 || g->sh <= 0 || g->sw <= 0 || g->dh <= 0 || g->dw <= 0 || g->ph < 0 || g->pw < 0) return fail("bad conv parameters");
    if (g->kh > 255 || g->kw > 255 || g->C > 65535) return fail("kernel size / channel count out of supported range");
    if ((g->H + 2 * g->ph - g->dh * (g->kh - 1) - 1) / g->sh + 1 != g->Ho || (g->W + 2 * g->pw - g->dw * (g->kw - 1) - 1) / g->sw + 1 != g->Wo)
        return fail("Ho/Wo inconsistent with H/W, kernel, stride, padding, dilation");
    if ((long long)g->B * g->Ho * g->Wo >= (1ll << 31) || (long long)g->B * g->H * g->W >= (1ll << 31)) return fail("pixel count exceeds int32");
    if (b->kind < 0 || b->kind > 2) return fail("unknown basis kind");
    if (b->act < KAN_ACT_NONE || b->act > KAN_ACT_GELU_TANH) return fail("unknown activation");
    int P = b->n_basis + (b->act != KAN_ACT_NONE);
    if (b->n_basis < 1 || P > KAN_MAX_PLANES) return fail("planes per channel exceed KAN_MAX_PLANES");
    if ((long long)g->C * g->kh * g->kw * P >= (1ll << 30)) return fail("GEMM depth too large");
    if (b->kind == KAN_BASIS_BSPLINE) {
        if (b->order < 0 || b->order > 3) return fail("spline_order must be in 0..3");
        if (b->n_basis + b->order + 1 > KAN_MAX_TABLE) return fail("too many knots");
        if (b->n_basis - b->order < 1) return fail("grid_size must be >= 1");
    }
    if (b->kind == KAN_BASIS_RBF && (b->n_basis > KAN_MAX_TABLE || !(b->p0 != 0.f))) return fail("bad RBF parameters");
    return 0;
}

DevGeom dev_geom(const KanGeom* g) {
    DevGeom d{g->B, g->C, g->H, g->W, g->O, g->Ho, g->Wo, g->kh, g->kw, g->sh, g->sw, g->ph, g->pw, g->dh, g->dw, g->x_bstride, g->y_bstride};
    return d;
}

DevBasis dev_basis(const KanBasis* b) {
    DevBasis d;
    d.kind = b->kind; d.nb = b->n_basis; d.order = b->order; d.act = b->act;
    d.hb = b->act != KAN_ACT_NONE ? 1 : 0; d.P = b->n_basis + d.hb;
    d.p0 = b->p0; d.p1 = b->p1; d.inv_h = 0.f;
    for (int i = 0; i < KAN_MAX_TABLE; ++i) d.tab[i] = b->table[i];
    if (b->kind == KAN_BASIS_BSPLINE) {
        int nk = b->n_basis + b->order + 1;
        float span = b->table[nk - 1] - b->table[0];
        d.inv_h = span > 0.f ? (float)(nk - 1) / span : 0.f;
    }
    return d;
}

int pick_splits(long long tiles, int chunks, int min_chunks) {
    if (tiles >= 384) return 1;
    int want = ceil_div(768, tiles);
    int cap = chunks / min_chunks; if (cap < 1) cap = 1;
    int s = want < cap ? want : cap;
    if (s < 1) s = 1;
    int cps = ceil_div(chunks, s);
    return ceil_div(chunks, cps);                 // no empty splits
}

int launch_ok(const char* what) {
    hipError_t e = hipGetLastError();
    (void)what;
    if (e != hipSuccess) { fail("launch failed: %s", hipGetErrorString(e)); return -2; }
    return 0;
}

struct FwdCfg { int TO, TP, tiles_o, tiles_p, chunks, splits; };
FwdCfg fwd_cfg(const KanGeom* g, const KanPlan& pl) {
    FwdCfg c;
    c.TO = (pl.Opad % 128 == 0) ? 128 : 64;
    c.TP = c.TO == 128 ? 128 : 256;
    c.tiles_o = pl.Opad / c.TO;
    c.tiles_p = ceil_div((long long)g->B * g->Ho * g->Wo, c.TP);
    c.chunks = pl.Kpad / KC;
    c.splits = pick_splits((long long)c.tiles_o * c.tiles_p, c.chunks, 8);
    return c;
}
struct BdCfg { int CT, tiles_c, tiles_p, n_ob, chunks, splits; };
BdCfg bd_cfg(const KanGeom* g, const KanPlan& pl) {
    BdCfg c;
    c.CT = 128 / pl.P;
    c.tiles_c = ceil_div(g->C, c.CT);
    c.tiles_p = ceil_div((long long)g->B * g->H * g->W, 128);
    c.n_ob = ceil_div(g->O, KC);
    c.chunks = g->kh * g->kw * c.n_ob;
    c.splits = pick_splits((long long)c.tiles_c * c.tiles_p, c.chunks, 8);
    return c;
}
struct BwCfg { int TR, TO, tiles_r, tiles_o, chunks, splits; };
BwCfg bw_cfg(const KanGeom* g, const KanPlan& pl) {
    BwCfg c;
    c.TO = (pl.Opad % 128 == 0) ? 128 : 64;
    c.TR = c.TO == 128 ? 128 : 256;
    c.tiles_r = ceil_div(pl.Kpad, c.TR);
    c.tiles_o = pl.Opad / c.TO;
    c.chunks = ceil_div((long long)g->B * g->Ho * g->Wo, KC);
    c.splits = pick_splits((long long)c.tiles_r * c.tiles_o, c.chunks, 16);
    return c;
}

int make_plan(const KanGeom* g, const KanBasis* b, KanPlan* pl) {
    if (int rc = check(g, b)) return rc;
    pl->P = b->n_basis + (b->act != KAN_ACT_NONE);
    pl->K = g->C * g->kh * g->kw * pl->P;
    pl->Kpad = round_up(pl->K, KC);
    pl->Opad = round_up(g->O, 64);
    pl->packed_weight_bytes = (long long)pl->Kpad * pl->Opad * 4;
    pl->fwd_slab_elems = (long long)g->B * g->y_bstride;
    pl->bwd_data_slab_elems = (long long)g->B * g->x_bstride;
    pl->bwd_weight_slab_elems = (long long)pl->Kpad * pl->Opad;
    pl->fwd_splits = fwd_cfg(g, *pl).splits;
    pl->bwd_data_splits = bd_cfg(g, *pl).splits;
    pl->bwd_weight_splits = bw_cfg(g, *pl).splits;
    return 0;
}

template <int G>
void launch_in_fwd(hipStream_t st, int planes, const float* z, int n_slabs, long long slab_elems, float* z_out, const float* gamma,
                   const float* beta, const float* a, float* y, float* mean, float* rstd, int Cn, int HW, long long bs, float eps) {
    int ppb = 256 / G;
    hipLaunchKernelGGL((k_in_prelu_fwd<G>), dim3(ceil_div(planes, ppb)), dim3(256), 0, st, z, n_slabs, slab_elems, z_out, gamma, beta, a, y,
                       mean, rstd, planes, Cn, HW, bs, eps);
}
template <int G>
void launch_in_bwd(hipStream_t st, int planes, const float* dy, const float* z, const float* mean, const float* rstd, const float* gamma,
                   const float* beta, const float* a, float* dz, float* dgamma, float* dbeta, float* dprelu, int Cn, int HW, long long bs) {
    int ppb = 256 / G;
    hipLaunchKernelGGL((k_in_prelu_bwd<G>), dim3(ceil_div(planes, ppb)), dim3(256), 0, st, dy, z, mean, rstd, gamma, beta, a, dz, dgamma,
                       dbeta, dprelu, planes, Cn, HW, bs);
}
int group_lanes(int HW) { int g = 4; while (g < 64 && g < HW) g <<= 1; return g; }

}  // namespace

// ============================================================================ C ABI
extern "C" {

const char* kan_version(void) { return "kanconv 0.1 (gfx950, fp32 MFMA 32x32x2)"; }
const char* kan_last_error(void) { return g_err; }

int kan_plan(const KanGeom* geom, const KanBasis* basis, KanPlan* plan) {
    if (!plan) return fail("null plan");
    return make_plan(geom, basis, plan);
}

int kan_pack_weights(const float* w_base, const float* w_basis, float* wp, const KanGeom* g, const KanBasis* b, void* stream) {
    KanPlan pl;
    if (int rc = make_plan(g, b, &pl)) return rc;
    const int hb = b->act != KAN_ACT_NONE;
    if ((hb && !w_base) || !w_basis || !wp) return fail("null weight pointer");
    hipStream_t st = (hipStream_t)stream;
    const int T = g->kh * g->kw;
    if (pl.Kpad != pl.K) {   // zero the padding rows (columns >= O are written as zeros by the tiles)
        if (hipMemsetAsync(wp + (size_t)pl.K * pl.Opad, 0, (size_t)(pl.Kpad - pl.K) * pl.Opad * 4, st) != hipSuccess) return fail("memset failed");
    }
    if (hb) {
        dim3 grid(ceil_div(g->C * T, 32), pl.Opad / 32);
        hipLaunchKernelGGL((k_pack<false>), grid, dim3(256), 0, st, w_base, (float*)nullptr, (float*)nullptr, (const float*)nullptr, wp,
                           g->O, g->C, T, pl.P, hb, b->n_basis, 0, pl.Opad, 0, 0ll);
    }
    dim3 grid(ceil_div(g->C * b->n_basis * T, 32), pl.Opad / 32);
    hipLaunchKernelGGL((k_pack<false>), grid, dim3(256), 0, st, w_basis, (float*)nullptr, (float*)nullptr, (const float*)nullptr, wp,
                       g->O, g->C, T, pl.P, hb, b->n_basis, 1, pl.Opad, 0, 0ll);
    return launch_ok("pack");
}

int kan_unpack_wgrad(const float* dwp, float* dw_base, float* dw_basis, const KanGeom* g, const KanBasis* b, void* stream) {
    KanPlan pl;
    if (int rc = make_plan(g, b, &pl)) return rc;
    const int hb = b->act != KAN_ACT_NONE;
    if ((hb && !dw_base) || !dw_basis || !dwp) return fail("null weight-gradient pointer");
    hipStream_t st = (hipStream_t)stream;
    const int T = g->kh * g->kw;
    if (hb) {
        dim3 grid(ceil_div(g->C * T, 32), ceil_div(g->O, 32));
        hipLaunchKernelGGL((k_pack<true>), grid, dim3(256), 0, st, (const float*)nullptr, (float*)nullptr, dw_base, dwp, (float*)nullptr,
                           g->O, g->C, T, pl.P, hb, b->n_basis, 0, pl.Opad, pl.bwd_weight_splits, pl.bwd_weight_slab_elems);
    }
    dim3 grid(ceil_div(g->C * b->n_basis * T, 32), ceil_div(g->O, 32));
    hipLaunchKernelGGL((k_pack<true>), grid, dim3(256), 0, st, (const float*)nullptr, (float*)nullptr, dw_basis, dwp, (float*)nullptr,
                       g->O, g->C, T, pl.P, hb, b->n_basis, 1, pl.Opad, pl.bwd_weight_splits, pl.bwd_weight_slab_elems);
    return launch_ok("unpack");
}

int kan_conv_fwd(const float* x, const float* xn, const float* wp, float* z, const KanGeom* g, const KanBasis* b, void* stream) {
    KanPlan pl;
    if (int rc = make_plan(g, b, &pl)) return rc;
    if (!x || !xn || !wp || !z) return fail("null tensor pointer");
    FwdCfg c = fwd_cfg(g, pl);
    DevGeom dg = dev_geom(g);
    DevBasis db = dev_basis(b);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(c.tiles_p, c.tiles_o, c.splits);
    int cps = ceil_div(c.chunks, c.splits);
    if (c.TO == 128)
        hipLaunchKernelGGL((k_conv_fwd<2, 2>), grid, dim3(256), 0, st, x, xn, wp, z, dg, db, pl.Opad, c.chunks, cps, pl.fwd_slab_elems);
    else
        hipLaunchKernelGGL((k_conv_fwd<1, 4>), grid, dim3(256), 0, st, x, xn, wp, z, dg, db, pl.Opad, c.chunks, cps, pl.fwd_slab_elems);
    return launch_ok("conv_fwd");
}

int kan_conv_bwd_data(const float* dz, const float* x, const float* xn, const float* wp, float* dx, float* dxn,
                      const KanGeom* g, const KanBasis* b, void* stream) {
    KanPlan pl;
    if (int rc = make_plan(g, b, &pl)) return rc;
    if (!dz || !x || !xn || !wp || !dx) return fail("null tensor pointer");
    if (!dxn && x != xn) return fail("dxn is required when xn != x");
    if (pl.P > 128) return fail("too many planes");
    BdCfg c = bd_cfg(g, pl);
    DevGeom dg = dev_geom(g);
    DevBasis db = dev_basis(b);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(c.tiles_p, c.tiles_c, c.splits);
    int cps = ceil_div(c.chunks, c.splits);
    hipLaunchKernelGGL(k_conv_bwd_data, grid, dim3(256), 0, st, dz, x, xn, wp, dx, dxn, dg, db, pl.Opad, c.CT, c.n_ob, c.chunks, cps,
                       pl.bwd_data_slab_elems);
    return launch_ok("conv_bwd_data");
}

int kan_conv_bwd_weight(const float* dz, const float* x, const float* xn, float* dwp, const KanGeom* g, const KanBasis* b, void* stream) {
    KanPlan pl;
    if (int rc = make_plan(g, b, &pl)) return rc;
    if (!dz || !x || !xn || !dwp) return fail("null tensor pointer");
    BwCfg c = bw_cfg(g, pl);
    DevGeom dg = dev_geom(g);
    DevBasis db = dev_basis(b);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(c.tiles_r, c.tiles_o, c.splits);
    int cps = ceil_div(c.chunks, c.splits);
    if (c.TO == 128)
        hipLaunchKernelGGL((k_conv_bwd_weight<2, 2>), grid, dim3(256), 0, st, dz, x, xn, dwp, dg, db, pl.Kpad, pl.Opad, c.chunks, cps,
                           pl.bwd_weight_slab_elems);
    else
        hipLaunchKernelGGL((k_conv_bwd_weight<4, 1>), grid, dim3(256), 0, st, dz, x, xn, dwp, dg, db, pl.Kpad, pl.Opad, c.chunks, cps,
                           pl.bwd_weight_slab_elems);
    return launch_ok("conv_bwd_weight");
}

int kan_slab_reduce(const float* slabs, int n_slabs, long long slab_elems, float* out, int B, int Cn, int HW, long long bstride, void* stream) {
    if (!slabs || !out || n_slabs < 1) return fail("bad slab_reduce arguments");
    long long total = (long long)B * Cn * HW;
    int blocks = (int)((total + 255) / 256); if (blocks > 4096) blocks = 4096; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_slab_reduce, dim3(blocks), dim3(256), 0, (hipStream_t)stream, slabs, n_slabs, slab_elems, out, Cn, HW, bstride, total);
    return launch_ok("slab_reduce");
}

int kan_instnorm_prelu_fwd(const float* z, int n_slabs, long long slab_elems, float* z_out, const float* gamma, const float* beta,
                           const float* prelu_a, float* y, float* mean, float* rstd, int B, int Cn, int HW, long long bstride, float eps,
                           void* stream) {
    if (!z || !z_out || !y || !mean || !rstd || n_slabs < 1 || B < 1 || Cn < 1 || HW < 1) return fail("bad instnorm_fwd arguments");
    hipStream_t st = (hipStream_t)stream;
    int planes = B * Cn;
    switch (group_lanes(HW)) {
        case 4:  launch_in_fwd<4>(st, planes, z, n_slabs, slab_elems, z_out, gamma, beta, prelu_a, y, mean, rstd, Cn, HW, bstride, eps); break;
        case 8:  launch_in_fwd<8>(st, planes, z, n_slabs, slab_elems, z_out, gamma, beta, prelu_a, y, mean, rstd, Cn, HW, bstride, eps); break;
        case 16: launch_in_fwd<16>(st, planes, z, n_slabs, slab_elems, z_out, gamma, beta, prelu_a, y, mean, rstd, Cn, HW, bstride, eps); break;
        case 32: launch_in_fwd<32>(st, planes, z, n_slabs, slab_elems, z_out, gamma, beta, prelu_a, y, mean, rstd, Cn, HW, bstride, eps); break;
        default: launch_in_fwd<64>(st, planes, z, n_slabs, slab_elems, z_out, gamma, beta, prelu_a, y, mean, rstd, Cn, HW, bstride, eps); break;
    }
    return launch_ok("instnorm_fwd");
}

int kan_instnorm_prelu_bwd(const float* dy, const float* z, const float* mean, const float* rstd, const float* gamma, const float* beta,
                           const float* prelu_a, float* dz, float* dgamma, float* dbeta, float* dprelu, int B, int Cn, int HW,
                           long long bstride, void* stream) {
    if (!dy || !z || !mean || !rstd || !dz || B < 1 || Cn < 1 || HW < 1) return fail("bad instnorm_bwd arguments");
    hipStream_t st = (hipStream_t)stream;
    int planes = B * Cn;
    switch (group_lanes(HW)) {
        case 4:  launch_in_bwd<4>(st, planes, dy, z, mean, rstd, gamma, beta, prelu_a, dz, dgamma, dbeta, dprelu, Cn, HW, bstride); break;
        case 8:  launch_in_bwd<8>(st, planes, dy, z, mean, rstd, gamma, beta, prelu_a, dz, dgamma, dbeta, dprelu, Cn, HW, bstride); break;
        case 16: launch_in_bwd<16>(st, planes, dy, z, mean, rstd, gamma, beta, prelu_a, dz, dgamma, dbeta, dprelu, Cn, HW, bstride); break;
        case 32: launch_in_bwd<32>(st, planes, dy, z, mean, rstd, gamma, beta, prelu_a, dz, dgamma, dbeta, dprelu, Cn, HW, bstride); break;
        default: launch_in_bwd<64>(st, planes, dy, z, mean, rstd, gamma, beta, prelu_a, dz, dgamma, dbeta, dprelu, Cn, HW, bstride); break;
    }
    return launch_ok("instnorm_bwd");
}

}  // extern "C"
