// Micro-benchmark: what ONE extra instruction of each kind costs next to a v_mfma_f32_32x32x2_f32 /
// v_mfma_f32_16x16x4_f32 stream (registers only, no memory), at 1 and 3 waves per SIMD.
// Everything is inline asm so the instruction mix is exactly what the table says.
//   make -C tools/micro valu_cost && gpurun -- tools/micro/valu_cost
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

enum Kind { NONE, FMA, PKFMA, EXP, MOV, ADDU, CNDMASK, DSREAD128, DSREAD32, SNOP, PKMUL, MAXF, DPP, SNOP3, SNOP7, SMOV };

template <int KIND>
__device__ __forceinline__ void filler(float (&r)[8], f32x2 (&p)[4], f32x4 &ld, unsigned lds_addr, int i) {
    float &x = r[i & 7];
    if constexpr (KIND == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(r[(i + 3) & 7]), "v"(r[(i + 5) & 7]));
    if constexpr (KIND == PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i & 3]) : "v"(p[(i + 1) & 3]), "v"(p[(i + 2) & 3]));
    if constexpr (KIND == PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i & 3]) : "v"(p[(i + 1) & 3]));
    if constexpr (KIND == EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
    if constexpr (KIND == MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(r[(i + 3) & 7]));
    if constexpr (KIND == ADDU) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(r[(i + 3) & 7]));
    if constexpr (KIND == MAXF) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x) : "v"(r[(i + 3) & 7]));
    if constexpr (KIND == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(r[(i + 3) & 7]));
    if constexpr (KIND == DPP) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(r[(i + 3) & 7]));
    if constexpr (KIND == DSREAD128) asm volatile("ds_read_b128 %0, %1" : "=v"(ld) : "v"(lds_addr));
    if constexpr (KIND == DSREAD32) asm volatile("ds_read_b32 %0, %1" : "=v"(x) : "v"(lds_addr));
    if constexpr (KIND == SNOP) asm volatile("s_nop 0");
    if constexpr (KIND == SNOP3) asm volatile("s_nop 3");
    if constexpr (KIND == SNOP7) asm volatile("s_nop 7");
    if constexpr (KIND == SMOV) asm volatile("s_mov_b32 s40, 0" ::: "s40");
}

// SHAPE 32: one 32x32x2 accumulator chain.  SHAPE 16: four 16x16x4 accumulators round-robin (equal FLOPs per "slot":
// 2 x 16x16x4 = 1 x 32x32x2 / 2 ... we count FLOPs).
template <int SHAPE, int KIND, int N>
__global__ __launch_bounds__(256) void loop(float *out, int iters, float a0) {
    __shared__ float lds[4096];
    f32x16 acc = {0}, accb = {0};
    f32x4 acc4[4] = {{0}, {0}, {0}, {0}};
    float r[8];
    f32x2 p[4];
    for (int i = 0; i < 8; ++i) r[i] = a0 + i * 0.001f;
    for (int i = 0; i < 4; ++i) p[i] = f32x2{a0, 0.5f};
    f32x4 ld = {0};
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = a0;
    __syncthreads();
    const unsigned lds_addr = (threadIdx.x & 63) * 16;
    float a = a0 + threadIdx.x * 1e-3f, b = 1.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if constexpr (SHAPE == 32) {
                asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
            } else if constexpr (SHAPE == 322) {   // two independent 32x32x2 chains, alternating
                if (u & 1) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(accb) : "v"(a), "v"(b));
                else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
            } else {
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc4[(2 * u) & 3]) : "v"(a), "v"(b));
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc4[(2 * u + 1) & 3]) : "v"(a), "v"(b));
            }
#pragma unroll
            for (int v = 0; v < N; ++v) filler<KIND>(r, p, ld, lds_addr, u * N + v);
        }
        if constexpr (KIND == DSREAD128 || KIND == DSREAD32) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float s = ld[0] + ld[3];
    for (int i = 0; i < 8; ++i) s += r[i];
    for (int i = 0; i < 4; ++i) s += p[i][0] + p[i][1];
    for (int e = 0; e < 16; ++e) s += acc[e] + accb[e];
    for (int i = 0; i < 4; ++i) s += acc4[i][0] + acc4[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static float *dbuf;
template <int SHAPE, int KIND, int N>
double run(int wps) {
    const int iters = 1500, blocks = 256 * wps;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((loop<SHAPE, KIND, N>), dim3(blocks), dim3(256), 0, 0, dbuf, 20, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((loop<SHAPE, KIND, N>), dim3(blocks), dim3(256), 0, 0, dbuf, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)blocks * 4;
    const double flop = waves * iters * 16.0 * 4096;   // both shapes: 4096 FLOP per slot
    return flop / ms / 1e9;
}

template <int SHAPE, int KIND, int N>
void row(const char *name, double base1, double base3) {
    const double t1 = run<SHAPE, KIND, N>(1), t3 = run<SHAPE, KIND, N>(3);
    // cost per filler in "64-cycle slots": (base/t - 1) * 64 / N cycles of matrix time each
    printf("%-3d %-22s x%-2d | 1 w/SIMD %6.1f TF (%5.1f cyc each) | 3 w/SIMD %6.1f TF (%5.1f cyc each)\n", SHAPE, name, N, t1,
           N ? (base1 / t1 - 1) * 64 / N : 0.0, t3, N ? (base3 / t3 - 1) * 64 / N : 0.0);
}

template <int SHAPE>
void table() {
    const double b1 = run<SHAPE, NONE, 0>(1), b3 = run<SHAPE, NONE, 0>(3);
    printf("%-3d %-22s     | 1 w/SIMD %6.1f TF               | 3 w/SIMD %6.1f TF\n", SHAPE, "bare MFMA stream", b1, b3);
    row<SHAPE, FMA, 2>("v_fma_f32", b1, b3);
    row<SHAPE, FMA, 4>("v_fma_f32", b1, b3);
    row<SHAPE, FMA, 8>("v_fma_f32", b1, b3);
    row<SHAPE, PKFMA, 1>("v_pk_fma_f32", b1, b3);
    row<SHAPE, PKFMA, 2>("v_pk_fma_f32", b1, b3);
    row<SHAPE, PKFMA, 4>("v_pk_fma_f32", b1, b3);
    row<SHAPE, PKMUL, 4>("v_pk_mul_f32", b1, b3);
    row<SHAPE, EXP, 1>("v_exp_f32", b1, b3);
    row<SHAPE, EXP, 2>("v_exp_f32", b1, b3);
    row<SHAPE, EXP, 4>("v_exp_f32", b1, b3);
    row<SHAPE, MOV, 4>("v_mov_b32", b1, b3);
    row<SHAPE, ADDU, 4>("v_add_u32", b1, b3);
    row<SHAPE, MAXF, 4>("v_max_f32", b1, b3);
    row<SHAPE, CNDMASK, 4>("v_cndmask_b32", b1, b3);
    row<SHAPE, DPP, 4>("v_mov_b32_dpp", b1, b3);
    row<SHAPE, DSREAD128, 1>("ds_read_b128", b1, b3);
    row<SHAPE, DSREAD128, 2>("ds_read_b128", b1, b3);
    row<SHAPE, DSREAD32, 2>("ds_read_b32", b1, b3);
    row<SHAPE, SNOP, 4>("s_nop 0", b1, b3);
}

int main(int argc, char **argv) {
    hipMalloc(&dbuf, 4 * 1024 * 1024 * 4);
    if (argc > 1) {   // short form: the two-chain question only
        const double a1 = run<32, NONE, 0>(1), a3 = run<32, NONE, 0>(3), b1 = run<322, NONE, 0>(1), b3 = run<322, NONE, 0>(3);
        printf("bare 32x32x2: one chain %.1f / %.1f TF (1 / 3 waves per SIMD); two alternating chains %.1f / %.1f TF\n", a1, a3, b1, b3);
        row<322, FMA, 2>("v_fma_f32 (2 chains)", b1, b3);
        row<322, FMA, 4>("v_fma_f32 (2 chains)", b1, b3);
        row<322, EXP, 2>("v_exp_f32 (2 chains)", b1, b3);
        row<322, DSREAD128, 1>("ds_read_b128 (2 chains)", b1, b3);
        row<32, SNOP, 1>("s_nop 0", a1, a3);
        row<32, SNOP, 2>("s_nop 0", a1, a3);
        row<32, SNOP3, 1>("s_nop 3", a1, a3);
        row<32, SNOP7, 1>("s_nop 7", a1, a3);
        row<32, SNOP7, 2>("s_nop 7", a1, a3);
        row<32, SMOV, 1>("s_mov_b32", a1, a3);
        row<32, SMOV, 4>("s_mov_b32", a1, a3);
        row<322, SNOP, 1>("s_nop 0 (2 chains)", b1, b3);
        row<322, SNOP, 4>("s_nop 0 (2 chains)", b1, b3);
        return 0;
    }
    table<32>();
    table<16>();
    return 0;
}
