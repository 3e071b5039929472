// Variants of the weight-feed loop (NT=3, A from LDS, B from global) at 1 and 3 workgroups per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int NT = 3;

// MODE 0: loads before MFMAs of the block (PF=1)   1: PF=2   2: loads after the MFMAs (end of block)
//      3: no sched_barrier (compiler schedule)     4: buffer_load via SRD   5: loads spread: 3 loads after every 4-MFMA group
template <int MODE>
__global__ __launch_bounds__(256, 3) void feed(const float *__restrict__ w, float *out, int tiles) {
    __shared__ __attribute__((aligned(16))) float xb[4 * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += 256) xb[i] = 0.001f * i;
    __syncthreads();
    f32x16 acc[NT];
    for (int t = 0; t < NT; ++t) acc[t] = f32x16{0};
    constexpr int PF = MODE == 1 ? 2 : 1;
    constexpr int RING = PF + 1;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(w), 0, 4 << 20, 0x00020000);
    for (int tile = 0; tile < tiles; ++tile) {
        unsigned lo = 0;
        asm volatile("" : "+v"(lo));
        const unsigned base = (unsigned)wave * NT * 4 * 1024 + lane * 4 + lo;
        const float *wl = w + base;
        f32x4 b[RING][NT][4];
        auto load = [&](int kb, int t, int s) {
            const int off = (t * 4 + kb) * 1024 + s * 256;
            if constexpr (MODE == 4)
                b[kb % RING][t][s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (base + off) * 4, 0, 0));
            else
                b[kb % RING][t][s] = *reinterpret_cast<const f32x4 *>(wl + off);
        };
#pragma unroll
        for (int p = 0; p < PF; ++p)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int s = 0; s < 4; ++s) load(p, t, s);
        if (MODE != 3) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            if (MODE != 2 && MODE != 5 && kb + PF < 4) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int s = 0; s < 4; ++s) load(kb + PF, t, s);
            }
            if (MODE != 3) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const f32x4 a = *reinterpret_cast<const f32x4 *>(xb + (kb * 4 + s) * 256 + lane * 4);
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[kb % RING][t][s][j], a[j], acc[t], 0, 0, 0);
                if (MODE == 5 && kb + PF < 4) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < NT; ++t) load(kb + PF, t, s);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (MODE == 2 && kb + PF < 4) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int s = 0; s < 4; ++s) load(kb + PF, t, s);
            }
            if (MODE != 3) __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int t = 0; t < NT; ++t)
        for (int e = 0; e < 16; ++e) s += acc[t][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char *name, const float *w, float *d) {
    for (int blocks : {256, 512, 768}) {
        const int tiles = 300;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL((feed<MODE>), dim3(blocks), dim3(256), 0, 0, w, d, 4);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL((feed<MODE>), dim3(blocks), dim3(256), 0, 0, w, d, tiles);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double mfmas = (double)blocks * 4 * tiles * 64.0 * NT;
        printf("%-40s WG/CU=%d  %.1f TFLOP/s\n", name, blocks / 256, mfmas * 4096 / ms / 1e9);
    }
}

int main() {
    float *d, *w;
    hipMalloc(&d, 4 * 1024 * 1024 * 4);
    hipMalloc(&w, 4 << 20);
    hipMemset(w, 0, 4 << 20);
    run<0>("0 loads before block MFMAs, PF=1", w, d);
    run<1>("1 PF=2", w, d);
    run<2>("2 loads after block MFMAs", w, d);
    run<3>("3 compiler-scheduled", w, d);
    run<4>("4 buffer_load (SRD)", w, d);
    run<5>("5 loads spread per 4-MFMA group", w, d);
    return 0;
}
