// Micro-benchmark: does the WORKGROUP SHAPE decide how well dependent fp32-MFMA chains share a CU?  The training attention kernels
// run three-wave workgroups (192 threads) at four workgroups per CU = 12 waves = "three per SIMD" on paper, and sit at ~106 cycles
// per MFMA whatever is removed from them (round 4: LDS reads, vector math, barriers, global traffic -- each worth <= 6 %).  Same
// instruction stream as attn_mix.hip (chains of 16 dependent v_mfma_f32_32x32x2_f32 + a vector block), registers only:
//   T threads per workgroup x G workgroups per CU, 12 waves per CU in every row, LDS padded so that exactly G workgroups are resident.
//   make -C tools/micro wg_shape && gpurun -- tools/micro/wg_shape
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int T, int FAT = 0>   // FAT: that many more live accumulators (16 registers each, one MFMA per iteration): register pressure
__global__ __launch_bounds__(T) void loop(float *out, int iters, float a0) {
    extern __shared__ float pad[];
    f32x16 acc_a = {0}, acc_b = {0};
    f32x16 fat[FAT ? FAT : 1] = {};
    float bop[16], aop[16];
    for (int i = 0; i < 16; ++i) { bop[i] = a0 + i * 1e-3f + threadIdx.x * 1e-5f; aop[i] = 0.5f + i * 1e-3f; }
    if (iters < 0) pad[threadIdx.x] = a0;
    for (int it = 0; it < iters; ++it) {
        f32x16 s = __builtin_amdgcn_mfma_f32_32x32x2f32(aop[0], bop[0], f32x16{0}, 0, 0, 0);
#pragma unroll
        for (int k = 1; k < 16; ++k) s = __builtin_amdgcn_mfma_f32_32x32x2f32(aop[k], bop[k], s, 0, 0, 0);
        float p[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) p[r] = __builtin_amdgcn_exp2f(s[r] - a0);
#pragma unroll
        for (int k = 0; k < 16; ++k) acc_a = __builtin_amdgcn_mfma_f32_32x32x2f32(aop[k], p[k], acc_a, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < 16; ++k) acc_b = __builtin_amdgcn_mfma_f32_32x32x2f32(bop[k], p[k], acc_b, 0, 0, 0);
#pragma unroll
        for (int f = 0; f < FAT; ++f) fat[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(aop[f], p[f], fat[f], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) bop[r] = bop[r] * 0.999f + 1e-6f * p[r];
    }
    for (int f = 0; f < FAT; ++f) acc_a += fat[f];
    float v = 0.f;
    for (int e = 0; e < 16; ++e) v += acc_a[e] + acc_b[e] + bop[e];
    out[blockIdx.x * T + threadIdx.x] = v;
}

template <int T, int FAT = 0>
void run(int wgs_per_cu, int grid_mult_num, int grid_mult_den, float *out) {
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount, iters = 2000;
    const size_t lds = (size_t)(160 * 1024 / wgs_per_cu / 1280) * 1280;   // exactly wgs_per_cu resident by LDS
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(loop<T, FAT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int grid = cus * wgs_per_cu * grid_mult_num / grid_mult_den;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((loop<T, FAT>), dim3(grid), dim3(T), lds, 0, out, 100, 0.25f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((loop<T, FAT>), dim3(grid), dim3(T), lds, 0, out, iters, 0.25f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)grid * (T / 64) * iters * (48 + FAT);
    const double tf = mfma * 4096 / (ms * 1e-3) / 1e12;
    hipFuncAttributes fa;
    (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(loop<T, FAT>));
    printf("%3d threads x %d workgroups per CU (grid %5d, %3d VGPRs): %7.3f ms  %6.1f TF  (%.3f of 157.3)\n", T, wgs_per_cu, grid, fa.numRegs, ms, tf, tf / 157.3);
}

int main() {
    float *out;
    (void)hipMalloc(&out, sizeof(float) * 256 * 4096 * 8);
    run<256>(3, 1, 1, out);    // 4-wave workgroups, 3 per CU: one wave of each on every SIMD
    run<192>(4, 1, 1, out);    // 3-wave workgroups, 4 per CU: the training attention kernels' shape
    run<192>(3, 1, 1, out);    // 9 waves per CU
    run<192>(2, 1, 1, out);
    run<192>(1, 1, 1, out);
    run<64>(12, 1, 1, out);    // single-wave workgroups
    run<128>(6, 1, 1, out);
    run<384>(2, 1, 1, out);    // 6-wave workgroups
    run<768>(1, 1, 1, out);    // 12-wave workgroup
    printf("-- with register pressure (three waves per SIMD at most) --\n");
    run<256, 2>(3, 1, 1, out);
    run<192, 2>(4, 1, 1, out);
    run<64, 2>(12, 1, 1, out);
    run<256, 3>(3, 1, 1, out);
    run<192, 3>(4, 1, 1, out);
    run<192, 3>(3, 1, 1, out);
    run<64, 3>(12, 1, 1, out);
    run<768, 3>(1, 1, 1, out);
    printf("-- four waves per SIMD at most (the training attention forward's regime) --\n");
    run<192, 1>(5, 1, 1, out);
    run<256, 1>(4, 1, 1, out);
    run<64, 1>(16, 1, 1, out);
    run<960, 1>(1, 1, 1, out);
    run<768, 1>(1, 1, 1, out);
    return 0;
}
