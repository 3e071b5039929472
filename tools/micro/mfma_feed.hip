// Micro-benchmark: v_mfma_f32_32x32x2_f32 fed like k_chain.hip's GEMM loop:
//   B operand: one global_load_dwordx4 per 4 MFMAs (fragment-packed, L2-resident), register ring PF k-blocks ahead
//   A operand: one ds_read_b128 per 4 MFMAs (lane-linear)  -- or registers
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int NT, bool LDS_A, bool GLOBAL_B>
__global__ __launch_bounds__(256, 3) void feed(const float *__restrict__ w, float *out, int tiles) {
    __shared__ __attribute__((aligned(16))) float xb[4 * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += 256) xb[i] = 0.001f * i;
    __syncthreads();
    f32x16 acc[NT];
    for (int t = 0; t < NT; ++t) acc[t] = f32x16{0};
    f32x4 areg[4] = {f32x4{1, 2, 3, 4}, f32x4{2, 3, 4, 5}, f32x4{3, 4, 5, 6}, f32x4{4, 5, 6, 7}};
    for (int tile = 0; tile < tiles; ++tile) {
        unsigned lo = 0;
        asm volatile("" : "+v"(lo));
        const float *wl = w + (size_t)wave * NT * 4 * 1024 + lane * 4 + lo;
        f32x4 b[2][NT][4];
        if (GLOBAL_B) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int s = 0; s < 4; ++s) b[0][t][s] = *reinterpret_cast<const f32x4 *>(wl + (t * 4 + 0) * 1024 + s * 256);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            if (GLOBAL_B && kb + 1 < 4) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        b[(kb + 1) & 1][t][s] = *reinterpret_cast<const f32x4 *>(wl + (t * 4 + kb + 1) * 1024 + s * 256);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const f32x4 a = LDS_A ? *reinterpret_cast<const f32x4 *>(xb + (kb * 4 + s) * 256 + lane * 4) : areg[s];
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(GLOBAL_B ? b[kb & 1][t][s][j] : areg[s][j], a[j], acc[t], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int t = 0; t < NT; ++t)
        for (int e = 0; e < 16; ++e) s += acc[t][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NT, bool LDS_A, bool GLOBAL_B>
void run(const char *name, int blocks, const float *w, float *d) {
    const int tiles = 400;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((feed<NT, LDS_A, GLOBAL_B>), dim3(blocks), dim3(256), 0, 0, w, d, 4);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((feed<NT, LDS_A, GLOBAL_B>), dim3(blocks), dim3(256), 0, 0, w, d, tiles);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)blocks * 4 * tiles * 64.0 * NT;
    printf("%-44s blocks=%4d  %.3f ms  %.1f TFLOP/s\n", name, blocks, ms, mfmas * 4096 / ms / 1e9);
}

int main() {
    float *d, *w;
    hipMalloc(&d, 4 * 1024 * 1024 * 4);
    hipMalloc(&w, 4 * 1024 * 1024);   // 4 waves x NT<=3 x 4 blocks x 4 KB = 192 KB used
    hipMemset(w, 0, 4 * 1024 * 1024);
    run<1, false, false>("NT=1 regs only, 1 WG/CU", 256, w, d);
    run<1, false, false>("NT=1 regs only, 3 WG/CU", 768, w, d);
    run<1, true, false>("NT=1 A from LDS, 1 WG/CU", 256, w, d);
    run<1, true, false>("NT=1 A from LDS, 3 WG/CU", 768, w, d);
    run<1, false, true>("NT=1 B from global ring, 1 WG/CU", 256, w, d);
    run<1, false, true>("NT=1 B from global ring, 3 WG/CU", 768, w, d);
    run<1, true, true>("NT=1 LDS A + global B, 1 WG/CU", 256, w, d);
    run<1, true, true>("NT=1 LDS A + global B, 3 WG/CU", 768, w, d);
    run<3, true, true>("NT=3 LDS A + global B, 1 WG/CU", 256, w, d);
    run<3, true, true>("NT=3 LDS A + global B, 3 WG/CU", 768, w, d);
    return 0;
}
