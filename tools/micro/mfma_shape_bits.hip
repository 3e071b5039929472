// Micro-experiment (round 6): do v_mfma_f32_32x32x2_f32 and v_mfma_f32_16x16x4_f32 give the SAME BITS for the same dot products when the
// k values pass in the same order?  D[i][j] = sum_k A[i][k] B[k][j], K = 128, random operands of mixed magnitude.
//   path 32: 64 MFMAs 32x32x2, step s takes k = 2 s + h from lane half h          (the chain / attention kernels' shape)
//   path 16: 32 MFMAs 16x16x4, step s takes k = 4 s + perm[g] from lane group g    (what a 16-ROW tile kernel would use)
// If some perm reproduces path 32's bits, a chain kernel on 16-row tiles could serve small batches (1 120 tiles of 32 rows on 768
// resident workgroups at 64 frames) without batch-dependent bits.  Prints the number of differing elements of the 16 x 16 corner for
// every permutation of the four k's of a step, and against an fp32 FMA chain in k order on the vector ALU.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int K = 128;

// A [32][K], B [K][32] row-major; out32 [32][32]
__global__ void path32(const float *A, const float *B, float *out) {
    const int lane = threadIdx.x, j = lane & 31, h = lane >> 5;
    f32x16 acc = {0};
    for (int s = 0; s < K / 2; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[j * K + 2 * s + h], B[(2 * s + h) * 32 + j], acc, 0, 0, 0);
    // D[i = 8 (e >> 2) + 4 h + (e & 3)][j]
    for (int e = 0; e < 16; ++e) out[(8 * (e >> 2) + 4 * h + (e & 3)) * 32 + j] = acc[e];
}
// the 16 x 16 corner on 16x16x4: lane (g, i): A[i][k = 4 s + perm[g]], B[k][j = i]; D[4 g + v][j = lane % 16]
__global__ void path16(const float *A, const float *B, float *out, int p0, int p1, int p2, int p3) {
    const int lane = threadIdx.x, i = lane & 15, g = lane >> 4;
    const int perm = g == 0 ? p0 : g == 1 ? p1 : g == 2 ? p2 : p3;
    f32x4 acc = {0};
    for (int s = 0; s < K / 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i * K + 4 * s + perm], B[(4 * s + perm) * 32 + i], acc, 0, 0, 0);
    for (int v = 0; v < 4; ++v) out[(4 * g + v) * 16 + i] = acc[v];
}
__global__ void path_fma(const float *A, const float *B, float *out) {
    const int i = threadIdx.x >> 4, j = threadIdx.x & 15;
    float acc = 0.f;
    for (int k = 0; k < K; ++k) acc = fmaf(A[i * K + k], B[k * 32 + j], acc);
    out[i * 16 + j] = acc;
}

int main() {
    std::vector<float> A(32 * K), B(K * 32);
    srand(7);
    auto rnd = [] { return ((rand() % 20001) - 10000) * 1e-4f * (1.f + (rand() % 7)) * ((rand() & 3) ? 1.f : 1e-3f); };
    for (auto &x : A) x = rnd();
    for (auto &x : B) x = rnd();
    float *dA, *dB, *d32, *d16, *dF;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&d32, 32 * 32 * 4); hipMalloc(&d16, 16 * 16 * 4); hipMalloc(&dF, 16 * 16 * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(path32, dim3(1), dim3(64), 0, 0, dA, dB, d32);
    hipLaunchKernelGGL(path_fma, dim3(1), dim3(256), 0, 0, dA, dB, dF);
    std::vector<float> r32(32 * 32), r16(16 * 16), rF(16 * 16);
    hipMemcpy(r32.data(), d32, r32.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(rF.data(), dF, rF.size() * 4, hipMemcpyDeviceToHost);
    int dfma = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) dfma += memcmp(&r32[i * 32 + j], &rF[i * 16 + j], 4) != 0;
    printf("32x32x2 chain vs fp32 FMA chain in k order: %d of 256 elements differ\n", dfma);
    int p[4] = {0, 1, 2, 3};
    do {
        hipLaunchKernelGGL(path16, dim3(1), dim3(64), 0, 0, dA, dB, d16, p[0], p[1], p[2], p[3]);
        hipMemcpy(r16.data(), d16, r16.size() * 4, hipMemcpyDeviceToHost);
        int d = 0, df = 0;
        double worst = 0;
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                d += memcmp(&r32[i * 32 + j], &r16[i * 16 + j], 4) != 0;
                df += memcmp(&rF[i * 16 + j], &r16[i * 16 + j], 4) != 0;
                worst = std::max(worst, (double)fabsf(r32[i * 32 + j] - r16[i * 16 + j]));
            }
        printf("16x16x4, lane groups take k = 4s + {%d,%d,%d,%d}: %3d of 256 differ from 32x32x2 (max |diff| %.3g), %3d from the FMA chain\n", p[0],
               p[1], p[2], p[3], d, worst, df);
    } while (std::next_permutation(p, p + 4));
    return 0;
}
