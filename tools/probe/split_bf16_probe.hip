// Probe for VERDICT r2 item 5 (tuning aid, not product code): can a split-precision contraction -- every fp32 operand cut into three bf16
// pieces (hi + mid + lo = 24 mantissa bits), six bf16 MFMA products per k-block accumulated in fp32 -- replace the exact-fp32 MFMA of the conv
// kernels?  Two questions, one binary (gfx950):
//
//  (1) ACCURACY on the operands of the 256 -> 256 @ 8x8 forward (K = 256 * 9 * 9 = 20736): W ~ kaiming-uniform, E = B-spline planes in [0, 1]
//      plus SiLU values; a 128 x 128 output tile computed
//        a) with v_mfma_f32_32x32x2_f32 (what the library runs: a k-ordered fp32 fma chain),
//        b) with v_mfma_f32_32x32x16_bf16 on 3-way split operands, 6 products (hh, hm, mh, mm, hl, lh),
//        c) the same with 3 products (hh, hm, mh: ~16 bits),
//      each against the fp64 result computed on the host: max-normalised and L2-relative error.
//  (2) RATE of the inner loop with both operands resident in LDS, as the halo kernels hold them: per 16-deep k-block and wave 12 ds_read_b128
//      (3 pieces x (2 A + 2 B) fragments) feeding 24 (6-product) or 12 (3-product) bf16 MFMAs on a 2 x 2 register tile, next to the fp32 MFMA
//      loop of tools/probe/mfma_probe.hip (4 ds_read_b32 + 4 MFMA per k-pair).  fp32-EQUIVALENT TFLOP/s = 2 M N K / time.
//
//   hipcc -O3 --offload-arch=gfx950 tools/probe/split_bf16_probe.hip -o /tmp/split_probe && /tmp/split_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __host__ inline int mfma_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// ---------------------------------------------------------------------------------------------- (1) accuracy
// One wave per 32 x 32 output block; A [M][K] and B [N][K] row-major fp32 in HBM (K contiguous).
__device__ inline void split3(float v, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)v; const float r1 = v - (float)h;
    m = (__bf16)r1; const float r2 = r1 - (float)m;
    l = (__bf16)r2;
}

template <int MODE>   // 0: fp32 MFMA chain, 1: 6-product split, 2: 3-product split
__global__ __launch_bounds__(64) void k_acc(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K) {
    const int lane = threadIdx.x, bi = blockIdx.x * 32, bj = blockIdx.y * 32;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (MODE == 0) {
        const float* a = A + (size_t)(bi + (lane & 31)) * K + (lane >> 5);
        const float* b = B + (size_t)(bj + (lane & 31)) * K + (lane >> 5);
        for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], b[k], acc, 0, 0, 0);
    } else {
        // lane l holds A[row l & 31][k = 8 (l >> 5) + j], j = 0..7 of a 16-deep block (cdna guide section 3)
        const float* a = A + (size_t)(bi + (lane & 31)) * K + 8 * (lane >> 5);
        const float* b = B + (size_t)(bj + (lane & 31)) * K + 8 * (lane >> 5);
        for (int k = 0; k < K; k += 16) {
            bf16x8 ah, am, al, bh, bm, bl;
            for (int j = 0; j < 8; ++j) {
                __bf16 h, m, l;
                split3(a[k + j], h, m, l); ah[j] = h; am[j] = m; al[j] = l;
                split3(b[k + j], h, m, l); bh[j] = h; bm[j] = m; bl[j] = l;
            }
            // smallest terms first (they are added into the same fp32 accumulator either way)
            if (MODE == 1) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
        }
    }
    for (int r = 0; r < 16; ++r) C[(size_t)(bi + mfma_row(r, lane)) * N + bj + (lane & 31)] = acc[r];
}

// ---------------------------------------------------------------------------------------------- (2) rate
// 256 threads, tile 128 x 128, wave tile 64 x 64 (2 x 2 blocks of 32 x 32), operands in LDS:
//   MODE 0: fp32 [k 16][128] per operand, 4 ds_read_b32 + 4 MFMA 32x32x2 per k-pair (the library's step)
//   MODE 1 / 2: three bf16 pieces [piece][row 128][k 16] per operand (rows 32 bytes apart, XOR-swizzled by 16-byte half), per 16-deep block
//   12 ds_read_b128 + 24 / 12 MFMA 32x32x16
template <int MODE>
__global__ __launch_bounds__(256, 2) void k_rate(const float* __restrict__ src, float* __restrict__ out, int iters) {
    __shared__ __attribute__((aligned(16))) float smem[2 * 3 * 128 * 8];           // 24 KB: [operand][piece][row][8 floats = 16 bf16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * 3 * 128 * 8; i += 256) smem[i] = src[i];
    __syncthreads();
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int ro = (wave >> 1) * 64 + (lane & 31), co = (wave & 1) * 64 + (lane & 31), kh2 = lane >> 5;
    if (MODE == 0) {
        const float* sA = smem; const float* sB = smem + 16 * 128;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const int krow = 2 * kk + kh2;
                const float a0 = sA[krow * 128 + ro], a1 = sA[krow * 128 + ro + 32], b0 = sB[krow * 128 + co], b1 = sB[krow * 128 + co + 32];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
        }
    } else {
        const bf16x8* sA = reinterpret_cast<const bf16x8*>(smem);                     // [piece][row][half]: 16 bytes per (row, half)
        const bf16x8* sB = sA + 3 * 128 * 2;
        for (int it = 0; it < iters; ++it) {
            bf16x8 fa[2][3], fb[2][3];
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int r = ro + 32 * m, c = co + 32 * m;
                    fa[m][p] = sA[(p * 128 + r) * 2 + (kh2 ^ ((r >> 2) & 1))];
                    fb[m][p] = sB[(p * 128 + c) * 2 + (kh2 ^ ((c >> 2) & 1))];
                }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    if (MODE == 1) {
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[m][2], fb[n][0], acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[m][0], fb[n][2], acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[m][1], fb[n][1], acc[m][n], 0, 0, 0);
                    }
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[m][1], fb[n][0], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[m][0], fb[n][1], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[m][0], fb[n][0], acc[m][n], 0, 0, 0);
                }
        }
    }
    float s = 0.f;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
    out[blockIdx.x * 256 + tid] = s;
}

static double rnd(unsigned& st) { st = st * 1664525u + 1013904223u; return (double)(st >> 8) / 16777216.0; }

template <int MODE>
static void rate(const float* src, float* out, int blocks, int iters, const char* what) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_rate<MODE>), dim3(blocks), dim3(256), 0, 0, src, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_rate<MODE>), dim3(blocks), dim3(256), 0, 0, src, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * iters * 2.0 * 128 * 128 * 16;                // one 16-deep block of a 128 x 128 tile, counted ONCE (fp32-equivalent)
    printf("rate  %-44s blocks %5d: %8.3f ms  %7.1f fp32-equivalent TFLOP/s\n", what, blocks, ms, flops / ms / 1e9);
}

int main() {
    // ---- (1) accuracy
    const int M = 128, N = 128, K = 20736;
    std::vector<float> A((size_t)M * K), B((size_t)N * K);
    unsigned st = 12345u;
    const double bound = sqrt(3.0 / (256.0 * 9.0 * 9.0)) * sqrt(3.0);                    // ~ kaiming_uniform(linear) bound of a [O, 2304, 3, 3] weight
    for (auto& v : A) v = (float)((2.0 * rnd(st) - 1.0) * bound);
    for (size_t i = 0; i < B.size(); ++i) {                                              // expanded operand: plane 0 = SiLU(x), planes 1..8 = cubic B-spline values (<= 4 non-zero)
        const int p = (int)(i % 9);
        const double x = 2.2 * (2.0 * rnd(st) - 1.0);
        if (p == 0) B[i] = (float)(x / (1.0 + exp(-x)));
        else { const double u = rnd(st); B[i] = rnd(st) < 0.45 ? (float)(u * u * u / 6.0 + 0.3 * rnd(st)) : 0.f; }
    }
    std::vector<double> ref((size_t)M * N);
    double refmax = 0, refl2 = 0;
    for (int i = 0; i < M; ++i)
        for (int j = 0; j < N; ++j) {
            double s = 0;
            for (int k = 0; k < K; ++k) s += (double)A[(size_t)i * K + k] * (double)B[(size_t)j * K + k];
            ref[(size_t)i * N + j] = s; refmax = fmax(refmax, fabs(s)); refl2 += s * s;
        }
    float *dA, *dB, *dC;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, (size_t)M * N * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> C((size_t)M * N);
    const char* names[3] = {"fp32 MFMA 32x32x2 (k-ordered fma chain)", "3 x bf16 pieces, 6 products (hh hm mh mm hl lh)", "3 x bf16 pieces, 3 products (hh hm mh)"};
    for (int mode = 0; mode < 3; ++mode) {
        if (mode == 0) hipLaunchKernelGGL((k_acc<0>), dim3(M / 32, N / 32), dim3(64), 0, 0, dA, dB, dC, M, N, K);
        if (mode == 1) hipLaunchKernelGGL((k_acc<1>), dim3(M / 32, N / 32), dim3(64), 0, 0, dA, dB, dC, M, N, K);
        if (mode == 2) hipLaunchKernelGGL((k_acc<2>), dim3(M / 32, N / 32), dim3(64), 0, 0, dA, dB, dC, M, N, K);
        hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
        double emax = 0, el2 = 0;
        for (size_t i = 0; i < C.size(); ++i) { const double d = (double)C[i] - ref[i]; emax = fmax(emax, fabs(d)); el2 += d * d; }
        printf("error %-52s K %d: max-normalised %.3e  L2-relative %.3e\n", names[mode], K, emax / refmax, sqrt(el2 / refl2));
    }
    // ---- (2) rate
    float *src, *out;
    hipMalloc(&src, 65536 * 4); hipMalloc(&out, 8192 * 256 * 4);
    std::vector<float> h(65536);
    for (auto& v : h) v = (float)(rnd(st) - 0.5);                                        // random data so that clocks behave as on real operands
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int blocks : {512, 2048}) {
        rate<0>(src, out, blocks, 4000, "fp32 MFMA, 4 ds_read_b32 per k-pair");
        rate<1>(src, out, blocks, 4000, "3 x bf16, 6 products, 12 ds_read_b128 / block");
        rate<2>(src, out, blocks, 4000, "3 x bf16, 3 products, 12 ds_read_b128 / block");
    }
    return 0;
}
