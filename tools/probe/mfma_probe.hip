// Microbenchmark: what limits a v_mfma_f32_32x32x2_f32 loop on gfx950?  (tuning aid, not product code)
//   variant 0: MFMA only (4 independent accumulators)
//   variant 1: + per k-step 4 ds_read_b32 feeding the MFMAs (as the conv kernels do)
//   variant 2: variant 1 + one __syncthreads() per 32 MFMAs
//   variant 3: variant 2 + 12 ds_write_b32 and 10 global loads per 32 MFMAs (a stand-in for staging)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

template <int V, int NV>
__global__ __launch_bounds__(256, 4) void probe(const float* __restrict__ src, float* __restrict__ out, int iters) {
    __shared__ float sA[2 * 16 * 128], sB[2 * 16 * 128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * 16 * 128; i += 256) { sA[i] = src[i]; sB[i] = src[i + 4096]; }
    __syncthreads();
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int ao = (wave >> 1) * 64 + (lane & 31), bp = (wave & 1) * 64 + (lane & 31), kh2 = lane >> 5;
    float a0 = src[tid], a1 = src[tid + 256], b0 = src[tid + 512], b1 = src[tid + 768];
    float g[10];
    float vv = src[tid + 1024]; int si = iters;
    for (int it = 0; it < iters; ++it) {
        const int cur = it & 1;
        if (V >= 3) {
#pragma unroll
            for (int n = 0; n < 10; ++n) g[n] = src[((it + 1) & 63) * 4096 + n * 256 + tid];
        }
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            if (V >= 1) {
                const int krow = 2 * kk + kh2;
                a0 = sA[cur * 2048 + krow * 128 + ao]; a1 = sA[cur * 2048 + krow * 128 + ao + 32];
                b0 = sB[cur * 2048 + krow * 128 + bp]; b1 = sB[cur * 2048 + krow * 128 + bp + 32];
            }
            acc[0][0] = MFMA32(a0, b0, acc[0][0]);
            acc[0][1] = MFMA32(a0, b1, acc[0][1]);
            acc[1][0] = MFMA32(a1, b0, acc[1][0]);
            acc[1][1] = MFMA32(a1, b1, acc[1][1]);
            if (V >= 4) {                                   // NV dependent VALU ops per 4 MFMAs (address-math stand-in)
#pragma unroll
                for (int q = 0; q < NV; ++q) vv = vv * 1.0001f + 0.5f;
            }
            if (V >= 5) {                                   // NV scalar ops per 4 MFMAs
#pragma unroll
                for (int q = 0; q < NV; ++q) si = __builtin_amdgcn_readfirstlane(si * 3 + 1);
            }
        }
        if (V >= 3) {
#pragma unroll
            for (int n = 0; n < 10; ++n) sA[(cur ^ 1) * 2048 + ((n * 256 + tid) & 2047)] = g[n];
            sB[(cur ^ 1) * 2048 + tid] = g[0]; sB[(cur ^ 1) * 2048 + tid + 256] = g[1];
        }
        if (V >= 2) __syncthreads();
    }
    float s = vv + (float)si;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int V, int NV = 0>
void run(const float* src, float* out, int blocks, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    (void)hipGetLastError(); hipLaunchKernelGGL((probe<V, NV>), dim3(blocks), dim3(256), 0, 0, src, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    (void)hipGetLastError(); hipLaunchKernelGGL((probe<V, NV>), dim3(blocks), dim3(256), 0, 0, src, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 * iters * 32 * 4096.0;
    printf("variant %d NV %2d blocks %5d (%.1f per CU): %.3f ms  %.1f TFLOP/s\n", V, NV, blocks, blocks / 256.0, ms, flops / ms / 1e9);
}

int main() {
    float *src, *out;
    hipMalloc(&src, 64 * 4096 * 4 + 65536); hipMalloc(&out, 8192 * 256 * 4);
    hipMemset(src, 0, 64 * 4096 * 4 + 65536);
    // random-ish data so that clocks behave as on real operands
    float* h = (float*)malloc(64 * 4096 * 4);
    for (int i = 0; i < 64 * 4096; ++i) h[i] = (float)((i * 2654435761u >> 8) & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(src, h, 64 * 4096 * 4, hipMemcpyHostToDevice);
    for (int blocks : {1024, 4096}) {
        run<0>(src, out, blocks, 2000);
        run<2>(src, out, blocks, 2000);
        run<3>(src, out, blocks, 2000);
        run<4, 8>(src, out, blocks, 2000);
        run<4, 16>(src, out, blocks, 2000);
        run<4, 32>(src, out, blocks, 2000);
        run<4, 64>(src, out, blocks, 2000);
        run<5, 8>(src, out, blocks, 2000);
        run<5, 16>(src, out, blocks, 2000);
        run<5, 32>(src, out, blocks, 2000);
    }
    return 0;
}
