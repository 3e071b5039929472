// Micro-benchmark: sustained v_mfma_f32_32x32x2_f32 rate per wave configuration (no memory traffic).
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int NACC, int VALU_PER_MFMA>
__global__ __launch_bounds__(256) void mfma_loop(float *out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x16{0};
    float a = a0 + threadIdx.x * 1e-3f, b = b0;
    float extra = a0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < VALU_PER_MFMA * NACC; ++v) extra = fmaf(extra, 1.0001f, 0.5f);
        }
    }
    float s = extra;
    for (int i = 0; i < NACC; ++i)
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, int VALU>
void run(const char *name, int blocks, int threads, float *d) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((mfma_loop<NACC, VALU>), dim3(blocks), dim3(threads), 0, 0, d, 10, 1.0f, 2.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((mfma_loop<NACC, VALU>), dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0f, 2.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)blocks * threads / 64;
    const double mfmas = waves * iters * 16.0 * NACC;
    printf("%-34s blocks=%5d thr=%4d  %.3f ms  %.1f TFLOP/s  (%.1f ns per MFMA per SIMD-slot)\n", name, blocks, threads, ms,
           mfmas * 4096 / ms / 1e9, ms * 1e6 / (iters * 16.0 * NACC) / ((waves / 1024.0)));
}

int main() {
    float *d;
    hipMalloc(&d, 4 * 1024 * 1024 * 4);
    run<1, 0>("1 acc, 1 wave/SIMD", 256, 256, d);
    run<2, 0>("2 acc, 1 wave/SIMD", 256, 256, d);
    run<4, 0>("4 acc, 1 wave/SIMD", 256, 256, d);
    run<1, 0>("1 acc, 2 waves/SIMD", 512, 256, d);
    run<1, 0>("1 acc, 3 waves/SIMD", 768, 256, d);
    run<1, 1>("1 acc + 1 VALU/MFMA, 1 wave/SIMD", 256, 256, d);
    run<1, 4>("1 acc + 4 VALU/MFMA, 1 wave/SIMD", 256, 256, d);
    run<1, 4>("1 acc + 4 VALU/MFMA, 3 waves/SIMD", 768, 256, d);
    run<1, 8>("1 acc + 8 VALU/MFMA, 3 waves/SIMD", 768, 256, d);
    return 0;
}
