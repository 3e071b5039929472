// Micro-benchmark: the instruction pattern of the training attention kernels without any memory -- chains of 16 dependent
// v_mfma_f32_32x32x2_f32 whose result feeds a block of vector instructions (sub, exp2, mul, fma per element) whose result is
// the B operand of the next chain -- at 1, 2, 3 and 4 waves per SIMD.  Question: is ~0.62 of the fp32 MFMA roof inherent to
// this dependency pattern, or do the real kernels lose it to LDS / global memory / barriers?
//   make -C tools/micro attn_mix && gpurun -- tools/micro/attn_mix
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int VALU_PER_ELEM>
__global__ __launch_bounds__(256) void loop(float *out, int iters, float a0) {
    f32x16 acc_a = {0}, acc_b = {0};
    float bop[16], aop[16];
    for (int i = 0; i < 16; ++i) { bop[i] = a0 + i * 1e-3f + threadIdx.x * 1e-5f; aop[i] = 0.5f + i * 1e-3f; }
    for (int it = 0; it < iters; ++it) {
        // chain 1: s = A . B  (fresh accumulator)
        f32x16 s = __builtin_amdgcn_mfma_f32_32x32x2f32(aop[0], bop[0], f32x16{0}, 0, 0, 0);
#pragma unroll
        for (int k = 1; k < 16; ++k) s = __builtin_amdgcn_mfma_f32_32x32x2f32(aop[k], bop[k], s, 0, 0, 0);
        // vector block on the result
        float p[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = __builtin_amdgcn_exp2f(s[r] - a0);
            if (VALU_PER_ELEM >= 4) v = v * aop[r];
            if (VALU_PER_ELEM >= 6) v = fmaf(v, a0, -bop[r] * 1e-3f);
            p[r] = v;
        }
        // chain 2: accumulate with the vector block's output as B operand
#pragma unroll
        for (int k = 0; k < 16; ++k) acc_a = __builtin_amdgcn_mfma_f32_32x32x2f32(aop[k], p[k], acc_a, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < 16; ++k) acc_b = __builtin_amdgcn_mfma_f32_32x32x2f32(bop[k], p[k], acc_b, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) bop[r] = bop[r] * 0.999f + 1e-6f * p[r];   // keep the loop from being hoisted
    }
    float v = 0.f;
    for (int e = 0; e < 16; ++e) v += acc_a[e] + acc_b[e] + bop[e];
    out[blockIdx.x * 256 + threadIdx.x] = v;
}

template <int V>
void run(const char *name, int wgs_per_cu, float *out) {
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount, iters = 4000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(loop<V>, dim3(cus * wgs_per_cu), dim3(256), 0, 0, out, 100, 0.25f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(loop<V>, dim3(cus * wgs_per_cu), dim3(256), 0, 0, out, iters, 0.25f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)cus * wgs_per_cu * 4 * iters * 48;          // per wave: 48 MFMAs per iteration
    const double tf = mfma * 4096 / (ms * 1e-3) / 1e12;
    printf("%-28s %d waves/SIMD: %7.3f ms  %6.1f TF  (%.3f of 157.3)\n", name, wgs_per_cu, ms, tf, tf / 157.3);
}

int main() {
    float *out;
    (void)hipMalloc(&out, sizeof(float) * 256 * 1024 * 8);
    for (int w = 1; w <= 4; ++w) run<2>("exp+sub per element", w, out);
    for (int w = 1; w <= 4; ++w) run<6>("6 vector ops per element", w, out);
    return 0;
}
