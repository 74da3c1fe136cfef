// Prototype for VERDICT r2 item 5 (tuning aid, NOT product code, not part of libkanconv): the forward of ONE real layer shape -- KAN-VGG11's
// 256 -> 256 @ 8x8, 3x3 / pad 1, batch 256, default B-spline basis (grid 5, order 3, SiLU base branch: P = 9 planes, GEMM depth 20736) -- as a
// SPLIT-PRECISION contraction: every fp32 operand (expanded plane value, weight) is cut into three bf16 pieces hi + mid + lo (24 mantissa bits),
// and each 16-deep k-block runs six v_mfma_f32_32x32x16_bf16 products (lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi) into the fp32 accumulator.
// A/B in one process against the shipped exact-fp32 kernel (libkanconv's kan_conv_fwd through the C ABI, k_conv_fwd_halo on this shape):
//   * error of both against an fp64 evaluation of the layer (host, 6 of the 256 images, all outputs),
//   * time per launch of both (HIP events, 20 launches after 3 warm-up).
//
// Kernel design (k_split_fwd), following the halo forward of the library:
//   tile    128 outputs x 128 pixels (two whole 8x8 images) per 256-thread workgroup, wave tile 64 x 64 (2 x 2 MFMA blocks), 256 workgroups;
//   B side  per group of 8 input channels the workgroup expands the 2 x 64 input values once (SiLU + 8 B-spline planes, fp32 vector ALU), cuts
//           each plane into 3 bf16 pieces and writes a zero-bordered 10x10 halo tile per image into LDS, plane-major inside a cell: the 8 channels
//           of one plane are 16 contiguous bytes = one lane's share of a 16-deep MFMA operand (k = 8 (lane >> 5) + j), so every tap reads the SAME
//           tile through a shifted address with one ds_read_b128 per piece and block.  Cell = 9 chunks of 16 B (odd: 8 consecutive pixels hit 8
//           different 4-bank groups), rows padded to 104 chunks so that the second image row of a 16-lane quarter lands 8 chunks further mod 16;
//   depth   per channel group 81 (tap, plane) k-groups of 8 (+ 3 zero groups = 42 steps of 16); lanes 0-31 and 32-63 of a step read DIFFERENT
//           k-groups (own shift each, from a small LDS table);
//   A side  weights pre-cut on the device into [step][piece][k-half][output][8 bf16] and streamed by 16-byte LDS-DMA into a ring of four
//           one-step buffers (12 KB per step), one barrier per TWO steps (48 MFMAs per wave).
//
//   hipcc -O3 --offload-arch=gfx950 -I include -I convolutional-kan-for-image-classification_amd/csrc tools/probe/split_bf16_conv.hip \
//         -L convolutional-kan-for-image-classification_amd -lkanconv -Wl,-rpath,'$ORIGIN/../../convolutional-kan-for-image-classification_amd' -o tools/probe/split_bf16_conv
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "kan_device.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e)); exit(2); } } while (0)

constexpr int NB = 256, NC = 256, NO = 256, HH = 8, HW = 64, NP = 9, CG = 8, NCG = NC / CG, NGRP = 84, NSTEP = NGRP / 2;
#ifndef X_ROWC
#define X_ROWC 104
#endif
constexpr int CELLC = 9, ROWC = X_ROWC, IMGC = 10 * ROWC;              // chunks (16 B) per cell / halo row / image
constexpr int SPLITB = 2 * IMGC * 16, HALOB = 3 * SPLITB;            // bytes per piece (two images), per halo tile
constexpr int WSTEP_G = 3 * 2 * NO * 16;                             // bytes of one step in the cut weights (all 256 outputs)
constexpr int WSLOT = 3 * 2 * 128 * 16, NBUF = 4;                    // one step of one 128-output tile in LDS
constexpr int LDS_BYTES = HALOB + NBUF * WSLOT + NGRP * 4 + 64;
constexpr int TOTAL_STEPS = NCG * NSTEP;

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
__global__ __launch_bounds__(256, 1) void k_split_fwd(const float* __restrict__ x, const __bf16* __restrict__ wc, float* __restrict__ z, DevBasis bs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sH = smem;
    unsigned char* sW = smem + HALOB;
    int* sOff = (int*)(smem + HALOB + NBUF * WSLOT);
    float* sTab = (float*)(sOff + NGRP);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w_o = wave & 1, w_p = wave >> 1, kh = lane >> 5, m = lane & 31;
    const int ot = blockIdx.x & 1, b0 = (blockIdx.x >> 1) * 2;

    for (int i = tid; i < HALOB / 16; i += 256) ((uint4*)sH)[i] = uint4{0u, 0u, 0u, 0u};
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
    for (int j = 0; j < 3; ++j) { const int q = tid + 256 * j, sk = q >> 7, o = q & 127; voff[j] = (unsigned)((sk * NO + ot * 128 + o) * 16); }
    auto issue = [&](int t) {
        unsigned char* dst = sW + (t & (NBUF - 1)) * WSLOT + wave * 1024;
        const int so = __builtin_amdgcn_readfirstlane(t * WSTEP_G);
#pragma unroll
        for (int j = 0; j < 3; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, (__attribute__((address_space(3))) void*)(dst + j * 4096), 16, (int)voff[j], so, 0, 0);
    };
    issue(0); issue(1);

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
    const int q = tid & 127, hh = tid >> 7, qi = q >> 6, lp = q & 63;
    const unsigned cellB = (unsigned)(((qi * 10 + (lp >> 3) + 1) * ROWC + ((lp & 7) + 1) * CELLC) * 16 + hh * 8);
    const float* xq = x + ((size_t)(b0 + qi) * NC + hh * 4) * HW + lp;

    int t = 0;
    for (int cg = 0; cg < NCG; ++cg) {
        __syncthreads();                                   // every wave is done with the previous group's halo tile (first pass: zero fill, tables)
        {
            float v[4][KAN_PMAX];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float xv = xq[(size_t)(cg * CG + j) * HW];
                kan_planes<KAN_BASIS_BSPLINE, false>(bs, sTab, xv, xv, v[j]);
            }
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                bf16x4 h, md, l;
#pragma unroll
                for (int j = 0; j < 4; ++j) { __bf16 a, b, d; split3(v[j][p], a, b, d); h[j] = a; md[j] = b; l[j] = d; }
                *(bf16x4*)(sH + cellB + p * 16) = h;
                *(bf16x4*)(sH + SPLITB + cellB + p * 16) = md;
                *(bf16x4*)(sH + 2 * SPLITB + cellB + p * 16) = l;
            }
        }
#pragma unroll 1
        for (int pr = 0; pr < NSTEP / 2; ++pr, t += 2) {
#ifndef X_NO_WAIT
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
#ifndef X_NO_BARRIER
            __syncthreads();                               // steps t, t+1 landed for every thread; everyone is past steps t-2, t-1
#endif
#ifndef X_NO_DMA
            if (t + 2 < TOTAL_STEPS) { issue(t + 2); issue(t + 3); }
#endif
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int st = pr * 2 + u;
                const unsigned ao = aLane + (unsigned)(((t + u) & (NBUF - 1)) * WSLOT);
                const int off = sOff[2 * st + kh];
                bf16x8 a[3][2], b[3][2];
#pragma unroll
                for (int s = 0; s < 3; ++s)
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        a[s][k] = *(const bf16x8*)(smem + ao + s * 4096 + k * 512);
                        b[s][k] = *(const bf16x8*)(smem + (bBase[k] + off) + s * SPLITB);
                    }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][i], b[0][j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[2][j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[1][j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[0][j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[1][j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[0][j], acc[i][j], 0, 0, 0);
                    }
            }
        }
    }
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
    CK(hipMemset(dz, 0xff, nz * 4));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_split_fwd, dim3(256), dim3(256), LDS_BYTES, 0, dx, dwc, dz, db);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k_split_fwd, dim3(256), dim3(256), LDS_BYTES, 0, dx, dwc, dz, db);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize()); CK(hipGetLastError());
    float ms_split; CK(hipEventElapsedTime(&ms_split, e0, e1)); ms_split /= iters;
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
