// How many workgroups of 256 threads are REALLY resident per CU for a given dynamic LDS size?  Every workgroup spins ~40 us;
// 3 x CUs workgroups are launched: one round (~40 us) if three fit, two rounds (~80 us) if only two do.
//   make -C tools/micro lds_residency && gpurun -- tools/micro/lds_residency
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void spin(float *o, long long cycles) {
    extern __shared__ float s[];
    s[threadIdx.x] = 1.f;
    __syncthreads();
    const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
    while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) o[blockIdx.x] = s[1];
}
int main() {
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    float *o;
    (void)hipMalloc(&o, sizeof(float) * cus * 4);
    (void)hipFuncSetAttribute((const void *)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    auto sweep = [&](int per_cu, int threads, size_t lo, size_t hi, size_t step) {
        for (size_t lds = lo; lds <= hi; lds += step) {
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            hipLaunchKernelGGL(spin, dim3(cus * per_cu), dim3(threads), lds, 0, o, 4000LL);   // 4000 ticks of 10 ns = 40 us
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(spin, dim3(cus * per_cu), dim3(threads), lds, 0, o, 4000LL);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
            int nb = 0;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, spin, threads, lds);
            printf("LDS %6zu B: %6.1f us for %d workgroups of %d threads per CU (occupancy API says %d)\n", lds, ms * 1e3, per_cu, threads, nb);
        }
    };
    sweep(3, 256, 50 * 1024, 55 * 1024, 512);      // the row-local chain kernels: three workgroups per CU
    sweep(4, 192, 39 * 1024, 42 * 1024, 512);      // attn_bwd_kernel: four workgroups of three waves, 40 320 B each
    sweep(4, 192, 40320, 40320, 512);
    return 0;
}
