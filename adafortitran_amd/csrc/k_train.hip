// k_train.hip -- row-wise pieces of the encoder layer's training forward / backward:
// residual + dropout + LayerNorm (forward keeps the pre-norm sum and the row statistics), LayerNorm
// backward, activation (+ dropout) forward / backward, dropout scaling of a gradient.
//
// Reference semantics: torch.nn.TransformerEncoderLayer, norm_first = False
// (reference src/models/blocks/encoders.py:44-55):
//     x1 = LN1(x + drop(attn_proj));   x2 = LN2(x1 + drop(W2 drop(act(W1 x1 + b1)) + b2))
// LayerNorm: biased variance, eps inside the square root.  All tensors row-major [rows][n], fp32.
// HBM-bound elementwise work: one wave per row (n = 128 or 256 -> 2 or 4 floats per lane), row
// reductions by DPP/shuffle butterflies; the column reductions of the LayerNorm weight gradients go
// through per-workgroup partial slices and the shared deterministic slice reduction.
#include "aft_internal.h"

namespace aft {

// dropout factor of element (row, col) of a [rows][n] activation: the factored mask of aft_internal.h
__device__ __forceinline__ float tdrop(uint32_t row_word, uint32_t col_word, uint32_t threshold, float keep_scale) {
    return dropmask_keep(row_word, col_word, threshold) ? keep_scale : 0.f;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// s = res + drop(y);  out = LN(s) * gamma + beta;  keeps s, mean, rstd.   NPL = ceil(n / 64) floats per lane; HALF: n = 64 NPL - 32
// (model dims 32, 96, 160, 224: the last float of a lane exists in lanes 0..31 only)
template <int NPL, bool HALF = false>
__global__ __launch_bounds__(256) void add_ln_fwd_kernel(const float *__restrict__ res, const float *__restrict__ y,
                                                         const float *__restrict__ gamma, const float *__restrict__ beta,
                                                         float *__restrict__ s_out, float *__restrict__ stats,
                                                         float *__restrict__ out, int rows, float eps, uint32_t seed,
                                                         uint32_t threshold, float keep_scale) {
    constexpr int N = NPL * 64 - (HALF ? 32 : 0);
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    float v[NPL];
    float sum = 0.f;
    const uint32_t rw = dropmask_row_word(seed, (uint32_t)row);
#pragma unroll
    for (int q = 0; q < NPL; ++q) {
        const int col = lane + 64 * q;
        const bool ok = !HALF || q < NPL - 1 || lane < 32;
        float t = ok ? y[(size_t)row * N + col] : 0.f;
        if (threshold) t *= tdrop(rw, dropmask_col_word(seed, (uint32_t)col), threshold, keep_scale);
        v[q] = ok ? res[(size_t)row * N + col] + t : 0.f;
        sum += v[q];
    }
    const float mean = wave_sum(sum) * (1.f / N);
    float var = 0.f;
#pragma unroll
    for (int q = 0; q < NPL; ++q) {
        const bool ok = !HALF || q < NPL - 1 || lane < 32;
        var += ok ? (v[q] - mean) * (v[q] - mean) : 0.f;
    }
    const float rstd = rsqrtf(wave_sum(var) * (1.f / N) + eps);
#pragma unroll
    for (int q = 0; q < NPL; ++q) {
        const int col = lane + 64 * q;
        if (HALF && q == NPL - 1 && lane >= 32) continue;
        s_out[(size_t)row * N + col] = v[q];
        out[(size_t)row * N + col] = (v[q] - mean) * rstd * gamma[col] + beta[col];
    }
    if (lane == 0) {
        stats[2 * (size_t)row] = mean;
        stats[2 * (size_t)row + 1] = rstd;
    }
}

// ds = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma;  per-workgroup partial sums of
// dgamma = sum dy * xhat and dbeta = sum dy into slices[block][3n].  ds times the dropout factor of the
// forward's `y` branch is the gradient of y: written to `dbranch`, its column sums (the bias gradient of
// the linear layer that produced y) into the third n of the slice.
template <int NPL, bool HALF = false>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ s,
                                                     const float *__restrict__ stats, const float *__restrict__ gamma,
                                                     float *__restrict__ ds, float *__restrict__ dbranch,
                                                     float *__restrict__ slices, int rows, int rows_per_block, uint32_t seed,
                                                     uint32_t threshold, float keep_scale) {
    constexpr int N = NPL * 64 - (HALF ? 32 : 0);
    __shared__ float part[4][3 * N];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    float dg[NPL], db[NPL], dbr[NPL], gm[NPL];
    uint32_t cw[NPL];
    bool okq[NPL];      // HALF: the last column group exists in lanes 0..31 only
#pragma unroll
    for (int q = 0; q < NPL; ++q) {
        okq[q] = !HALF || q < NPL - 1 || lane < 32;
        dg[q] = 0.f;
        db[q] = 0.f;
        dbr[q] = 0.f;
        gm[q] = okq[q] ? gamma[lane + 64 * q] : 0.f;
        cw[q] = dropmask_col_word(seed, (uint32_t)(lane + 64 * q));
    }
    for (int row = r0 + wave; row < r1; row += 4) {
        const float mean = stats[2 * (size_t)row], rstd = stats[2 * (size_t)row + 1];
        const uint32_t rw = dropmask_row_word(seed, (uint32_t)row);
        float g[NPL], xh[NPL], a = 0.f, b = 0.f;
#pragma unroll
        for (int q = 0; q < NPL; ++q) {
            const int col = lane + 64 * q;
            const float d = okq[q] ? dy[(size_t)row * N + col] : 0.f;
            xh[q] = okq[q] ? (s[(size_t)row * N + col] - mean) * rstd : 0.f;
            g[q] = d * gm[q];
            a += g[q];
            b += g[q] * xh[q];
            dg[q] += d * xh[q];
            db[q] += d;
        }
        a = wave_sum(a) * (1.f / N);
        b = wave_sum(b) * (1.f / N);
#pragma unroll
        for (int q = 0; q < NPL; ++q) {
            const int col = lane + 64 * q;
            if (!okq[q]) continue;
            const float v = rstd * (g[q] - a - xh[q] * b);
            ds[(size_t)row * N + col] = v;
            const float vb = threshold ? v * tdrop(rw, cw[q], threshold, keep_scale) : v;
            dbranch[(size_t)row * N + col] = vb;
            dbr[q] += vb;
        }
    }
#pragma unroll
    for (int q = 0; q < NPL; ++q) {
        if (!okq[q]) continue;
        part[wave][lane + 64 * q] = dg[q];
        part[wave][N + lane + 64 * q] = db[q];
        part[wave][2 * N + lane + 64 * q] = dbr[q];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * N; i += 256)
        slices[(size_t)blockIdx.x * 3 * N + i] = part[0][i] + part[1][i] + part[2][i] + part[3][i];
}

// The same two kernels for ANY n <= 512 (the general engine, DESIGN.md 4.6: model dims that are not multiples of 32, and 288 .. 512):
// column lane + 64 q exists when it is < n; everything else as above.
constexpr int kLnAnyNpl = 8;
__global__ __launch_bounds__(256) void add_ln_fwd_any_kernel(const float *__restrict__ res, const float *__restrict__ y,
                                                             const float *__restrict__ gamma, const float *__restrict__ beta,
                                                             float *__restrict__ s_out, float *__restrict__ stats,
                                                             float *__restrict__ out, int rows, int n, float eps, uint32_t seed,
                                                             uint32_t threshold, float keep_scale) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    float v[kLnAnyNpl];
    float sum = 0.f;
    const uint32_t rw = dropmask_row_word(seed, (uint32_t)row);
    const float inv_n = 1.f / (float)n;
#pragma unroll
    for (int q = 0; q < kLnAnyNpl; ++q) {
        const int col = lane + 64 * q;
        const bool ok = col < n;
        float t = ok ? y[(size_t)row * n + col] : 0.f;
        if (threshold) t *= tdrop(rw, dropmask_col_word(seed, (uint32_t)col), threshold, keep_scale);
        v[q] = ok ? res[(size_t)row * n + col] + t : 0.f;
        sum += v[q];
    }
    const float mean = wave_sum(sum) * inv_n;
    float var = 0.f;
#pragma unroll
    for (int q = 0; q < kLnAnyNpl; ++q) var += lane + 64 * q < n ? (v[q] - mean) * (v[q] - mean) : 0.f;
    const float rstd = rsqrtf(wave_sum(var) * inv_n + eps);
#pragma unroll
    for (int q = 0; q < kLnAnyNpl; ++q) {
        const int col = lane + 64 * q;
        if (col >= n) continue;
        s_out[(size_t)row * n + col] = v[q];
        out[(size_t)row * n + col] = (v[q] - mean) * rstd * gamma[col] + beta[col];
    }
    if (lane == 0) {
        stats[2 * (size_t)row] = mean;
        stats[2 * (size_t)row + 1] = rstd;
    }
}

__global__ __launch_bounds__(256) void ln_bwd_any_kernel(const float *__restrict__ dy, const float *__restrict__ s,
                                                         const float *__restrict__ stats, const float *__restrict__ gamma,
                                                         float *__restrict__ ds, float *__restrict__ dbranch,
                                                         float *__restrict__ slices, int rows, int n, int rows_per_block, uint32_t seed,
                                                         uint32_t threshold, float keep_scale) {
    extern __shared__ float part_any[];     // [4][3 n]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    const float inv_n = 1.f / (float)n;
    float dg[kLnAnyNpl], db[kLnAnyNpl], dbr[kLnAnyNpl], gm[kLnAnyNpl];
    uint32_t cw[kLnAnyNpl];
#pragma unroll
    for (int q = 0; q < kLnAnyNpl; ++q) {
        dg[q] = 0.f;
        db[q] = 0.f;
        dbr[q] = 0.f;
        gm[q] = lane + 64 * q < n ? gamma[lane + 64 * q] : 0.f;
        cw[q] = dropmask_col_word(seed, (uint32_t)(lane + 64 * q));
    }
    for (int row = r0 + wave; row < r1; row += 4) {
        const float mean = stats[2 * (size_t)row], rstd = stats[2 * (size_t)row + 1];
        const uint32_t rw = dropmask_row_word(seed, (uint32_t)row);
        float g[kLnAnyNpl], xh[kLnAnyNpl], a = 0.f, b = 0.f;
#pragma unroll
        for (int q = 0; q < kLnAnyNpl; ++q) {
            const int col = lane + 64 * q;
            const bool ok = col < n;
            const float d = ok ? dy[(size_t)row * n + col] : 0.f;
            xh[q] = ok ? (s[(size_t)row * n + col] - mean) * rstd : 0.f;
            g[q] = d * gm[q];
            a += g[q];
            b += g[q] * xh[q];
            dg[q] += d * xh[q];
            db[q] += d;
        }
        a = wave_sum(a) * inv_n;
        b = wave_sum(b) * inv_n;
#pragma unroll
        for (int q = 0; q < kLnAnyNpl; ++q) {
            const int col = lane + 64 * q;
            if (col >= n) continue;
            const float v = rstd * (g[q] - a - xh[q] * b);
            ds[(size_t)row * n + col] = v;
            const float vb = threshold ? v * tdrop(rw, cw[q], threshold, keep_scale) : v;
            dbranch[(size_t)row * n + col] = vb;
            dbr[q] += vb;
        }
    }
#pragma unroll
    for (int q = 0; q < kLnAnyNpl; ++q) {
        const int col = lane + 64 * q;
        if (col >= n) continue;
        part_any[wave * 3 * n + col] = dg[q];
        part_any[wave * 3 * n + n + col] = db[q];
        part_any[wave * 3 * n + 2 * n + col] = dbr[q];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * n; i += 256)
        slices[(size_t)blockIdx.x * 3 * n + i] = part_any[i] + part_any[3 * n + i] + part_any[6 * n + i] + part_any[9 * n + i];
}

__device__ __forceinline__ float gelu_exact(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad(float x) {
    return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * expf(-0.5f * x * x);
}

// h = drop(act(a))       ACT: 0 = relu, 1 = gelu (exact erf form, torch default)
template <int ACT>
__global__ __launch_bounds__(256) void act_fwd_kernel(const float *__restrict__ a, float *__restrict__ hout, size_t n4, int n,
                                                      uint32_t seed, uint32_t threshold, float keep_scale) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const f32x4 x = reinterpret_cast<const f32x4 *>(a)[i];
    const uint32_t row = (uint32_t)((4 * i) / (unsigned)n), col = (uint32_t)((4 * i) % (unsigned)n);
    const uint32_t rw = dropmask_row_word(seed, row);
    f32x4 r;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float v = ACT ? gelu_exact(x[c]) : fmaxf(x[c], 0.f);
        if (threshold) v *= tdrop(rw, dropmask_col_word(seed, col + c), threshold, keep_scale);
        r[c] = v;
    }
    reinterpret_cast<f32x4 *>(hout)[i] = r;
}

// da = dh * dropfactor * act'(a)   (in place on dh), [rows][n]; column sums of da (the bias gradient of
// linear1) per workgroup into slices[block][n].  A thread owns 4 adjacent columns and every
// (256 / (n/4))-th row of the workgroup's row chunk.
template <int ACT>
__global__ __launch_bounds__(256) void act_bwd_kernel(const float *__restrict__ a, float *__restrict__ dh,
                                                      float *__restrict__ slices, int rows, int n, int rows_per_block,
                                                      uint32_t seed, uint32_t threshold, float keep_scale) {
    __shared__ f32x4 part[256];
    const int groups = n >> 2, cg = threadIdx.x % groups, rsub = threadIdx.x / groups, rstep = 256 / groups;   // rsub >= rstep: idle
    const int r0 = blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    uint32_t cw[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) cw[c] = dropmask_col_word(seed, (uint32_t)(4 * cg + c));
    for (int row = r0 + rsub; row < r1 && rsub < rstep; row += rstep) {
        const size_t i = ((size_t)row * n >> 2) + cg;
        const uint32_t rw = dropmask_row_word(seed, (uint32_t)row);
        const f32x4 x = reinterpret_cast<const f32x4 *>(a)[i];
        f32x4 g = reinterpret_cast<f32x4 *>(dh)[i];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float v = g[c] * (ACT ? gelu_grad(x[c]) : (x[c] > 0.f ? 1.f : 0.f));
            if (threshold) v *= tdrop(rw, cw[c], threshold, keep_scale);
            g[c] = v;
        }
        reinterpret_cast<f32x4 *>(dh)[i] = g;
        sum += g;
    }
    part[threadIdx.x] = sum;
    __syncthreads();
    if (rsub == 0) {
        for (int q = 1; q < rstep; ++q) sum += part[q * groups + cg];
        reinterpret_cast<f32x4 *>(slices + (size_t)blockIdx.x * n)[cg] = sum;
    }
}

// out = a + b   (gradient joins of the two residual branches)
__global__ __launch_bounds__(256) void add_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                  float *__restrict__ out, size_t n4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const f32x4 x = reinterpret_cast<const f32x4 *>(a)[i], y = reinterpret_cast<const f32x4 *>(b)[i];
    reinterpret_cast<f32x4 *>(out)[i] = x + y;
}

// torch.optim.Adam (reference src/main/trainer.py:407-413, no amsgrad), one pass over a flat parameter shard:
//   g = grad * grad_scale + wd * p;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2
//   p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
__global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                                   float *__restrict__ v, size_t n, float lr, float b1, float b2, float eps,
                                                   float wd, float grad_scale, float inv_bc1, float inv_sqrt_bc2) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float pw = p[i];
    const float gr = g[i] * grad_scale + wd * pw;
    const float mn = b1 * m[i] + (1.f - b1) * gr;
    const float vn = b2 * v[i] + (1.f - b2) * gr * gr;
    m[i] = mn;
    v[i] = vn;
    p[i] = pw - lr * inv_bc1 * mn / (sqrtf(vn) * inv_sqrt_bc2 + eps);
}

hipError_t launch_adam(float *p, const float *g, float *m, float *v, size_t n, float lr, float b1, float b2, float eps,
                       float wd, float grad_scale, int step, hipStream_t st) {
    const double bc1 = 1.0 - pow((double)b1, step), bc2 = 1.0 - pow((double)b2, step);
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, g, m, v, n, lr, b1, b2, eps, wd,
                       grad_scale, (float)(1.0 / bc1), (float)(1.0 / sqrt(bc2)));
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// ChannelAdapter in the training path (reference src/models/blocks/channel_adaptivity.py:24-40,59-63): three
// MLPs Linear(1,h0)-ReLU-Linear(h0,h1)-ReLU-Linear(h1,h2) on raw scalars, output element i of encoder e ->
// token i/2, feature 2e + i%2.  0.07 MFLOP per frame: as separate dense layers its forward + backward is
// ~45 launches of pure latency, so it is three kernels here: forward (keeps the hidden activations), data
// gradients per (frame, encoder), and one kernel for all 18 parameter gradients (a thread per element loops
// over the frames in a fixed order: deterministic, no atomics).
// ---------------------------------------------------------------------------------------------
struct AdapterTrainArgs {
    const float *cond[3];
    const float *w[3][3], *b[3][3];
    float *tokens6;            // forward out [frames][tokens][6]
    float *a0, *a1;            // hidden activations (post-ReLU) [frames][3][h0], [frames][3][h1]
    const float *dtok;         // backward in [frames][tokens][6]
    float *da0, *da1;          // [frames][3][h0], [frames][3][h1]
    float *dw[3][3], *db[3][3];
    int h0, h1, h2, tokens, frames, accumulate;
};

__global__ __launch_bounds__(256) void adapter_train_fwd_kernel(const AdapterTrainArgs a) {
    extern __shared__ float sm[];
    float *a0 = sm, *a1 = sm + a.h0;
    const int b = blockIdx.x, e = blockIdx.y, tid = threadIdx.x;
    const float x = a.cond[e][b];
    for (int i = tid; i < a.h0; i += 256) {
        const float v = fmaxf(fmaf(a.w[e][0][i], x, a.b[e][0][i]), 0.f);
        a0[i] = v;
        a.a0[((size_t)b * 3 + e) * a.h0 + i] = v;
    }
    __syncthreads();
    for (int i = tid; i < a.h1; i += 256) {
        float acc = a.b[e][1][i];
        for (int k = 0; k < a.h0; ++k) acc = fmaf(a.w[e][1][i * a.h0 + k], a0[k], acc);
        const float v = fmaxf(acc, 0.f);
        a1[i] = v;
        a.a1[((size_t)b * 3 + e) * a.h1 + i] = v;
    }
    __syncthreads();
    for (int i = tid; i < a.h2; i += 256) {
        float acc = a.b[e][2][i];
        const float *wr = a.w[e][2] + (size_t)i * a.h1;
        for (int k = 0; k < a.h1; ++k) acc = fmaf(wr[k], a1[k], acc);
        a.tokens6[((size_t)b * a.tokens + (i >> 1)) * 6 + 2 * e + (i & 1)] = acc;
    }
}

// da1[k] = relu'(a1[k]) sum_i dout[i] W3[i][k];  da0[m] = relu'(a0[m]) sum_k da1[k] W2[k][m]   per (frame, encoder).
// The first sum runs over h2 = 560 rows of W3 [h2][h1]: thread (g, k) of 256 / h1 row groups walks the rows i = g, g + G, ...
// of column k (a wave reads consecutive k: coalesced), the groups meet in LDS.  (Round 1 gave every thread all h1
// columns of a few rows and paid h1 x 6 cross-lane shuffles per wave: 78 us of pure latency.)
constexpr int kAdaH1Max = 64;
__global__ __launch_bounds__(256) void adapter_train_dgrad_kernel(const AdapterTrainArgs a) {
    extern __shared__ float dout[];   // [h2]
    __shared__ float part[256];
    __shared__ float d1[kAdaH1Max];
    const int b = blockIdx.x, e = blockIdx.y, tid = threadIdx.x;
    for (int i = tid; i < a.h2; i += 256) dout[i] = a.dtok[((size_t)b * a.tokens + (i >> 1)) * 6 + 2 * e + (i & 1)];
    __syncthreads();
    const int groups = 256 / a.h1, g = tid / a.h1, k = tid - g * a.h1;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (g < groups) {
        const float *wc = a.w[e][2] + k;
        int i = g;
        for (; i + 3 * groups < a.h2; i += 4 * groups) {
            s0 = fmaf(dout[i], wc[(size_t)i * a.h1], s0);
            s1 = fmaf(dout[i + groups], wc[(size_t)(i + groups) * a.h1], s1);
            s2 = fmaf(dout[i + 2 * groups], wc[(size_t)(i + 2 * groups) * a.h1], s2);
            s3 = fmaf(dout[i + 3 * groups], wc[(size_t)(i + 3 * groups) * a.h1], s3);
        }
        for (; i < a.h2; i += groups) s0 = fmaf(dout[i], wc[(size_t)i * a.h1], s0);
    }
    part[tid] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (tid < a.h1) {
        float v = 0.f;
        for (int q = 0; q < groups; ++q) v += part[q * a.h1 + tid];
        const float r = a.a1[((size_t)b * 3 + e) * a.h1 + tid] > 0.f ? v : 0.f;
        d1[tid] = r;
        a.da1[((size_t)b * 3 + e) * a.h1 + tid] = r;
    }
    __syncthreads();
    if (tid < a.h0) {
        float v = 0.f;
        for (int kk = 0; kk < a.h1; ++kk) v = fmaf(d1[kk], a.w[e][1][kk * a.h0 + tid], v);
        a.da0[((size_t)b * 3 + e) * a.h0 + tid] = a.a0[((size_t)b * 3 + e) * a.h0 + tid] > 0.f ? v : 0.f;
    }
}

__device__ const float kAdapterOne = 1.f;

// all parameter gradients: element index -> (encoder, tensor, position); sum over the frames in order
__global__ __launch_bounds__(256) void adapter_train_wgrad_kernel(const AdapterTrainArgs a) {
    const int per = a.h2 * a.h1 + a.h2 + a.h1 * a.h0 + a.h1 + a.h0 + a.h0;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= 3 * per) return;
    const int e = idx / per;
    int r = idx - e * per;
    // every gradient is sum_b u[b] * v[b] (or sum_b u[b]) with strided u, v: describe the two streams, then one
    // loop with four independent float64 partial sums (the raw conditions, up to 1400, make these reductions
    // ill-conditioned in fp32; four chains keep enough loads in flight)
    const float *u, *v = nullptr;
    size_t us, vs = 0;
    float *out;
    const size_t tstride = (size_t)a.tokens * 6;
    if (r < a.h2 * a.h1) {                       // dW3[i][k] = sum_b dout[b][i] a1[b][k]
        const int i = r / a.h1, k = r - i * a.h1;
        u = a.dtok + (size_t)(i >> 1) * 6 + 2 * e + (i & 1); us = tstride;
        v = a.a1 + (size_t)e * a.h1 + k; vs = (size_t)3 * a.h1;
        out = a.dw[e][2] + r;
    } else if ((r -= a.h2 * a.h1) < a.h2) {      // db3[i]
        u = a.dtok + (size_t)(r >> 1) * 6 + 2 * e + (r & 1); us = tstride;
        out = a.db[e][2] + r;
    } else if ((r -= a.h2) < a.h1 * a.h0) {      // dW2[k][m] = sum_b da1[b][k] a0[b][m]
        const int k = r / a.h0, m = r - k * a.h0;
        u = a.da1 + (size_t)e * a.h1 + k; us = (size_t)3 * a.h1;
        v = a.a0 + (size_t)e * a.h0 + m; vs = (size_t)3 * a.h0;
        out = a.dw[e][1] + r;
    } else if ((r -= a.h1 * a.h0) < a.h1) {      // db2[k]
        u = a.da1 + (size_t)e * a.h1 + r; us = (size_t)3 * a.h1;
        out = a.db[e][1] + r;
    } else if ((r -= a.h1) < a.h0) {             // dW1[m] = sum_b da0[b][m] x[b]
        u = a.da0 + (size_t)e * a.h0 + r; us = (size_t)3 * a.h0;
        v = a.cond[e]; vs = 1;
        out = a.dw[e][0] + r;
    } else {                                     // db1[m]
        r -= a.h0;
        u = a.da0 + (size_t)e * a.h0 + r; us = (size_t)3 * a.h0;
        out = a.db[e][0] + r;
    }
    if (!v) {   // plain sums: multiply by a constant one at stride 0, so that the loop below has no branch around its loads
        v = &kAdapterOne;
        vs = 0;
    }
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int b = 0;
    for (; b + 15 < a.frames; b += 16) {   // 32 loads in flight per pass: the loop is a chain of L2 round trips otherwise
        float uu[16], vv[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            uu[q] = u[(size_t)(b + q) * us];
            vv[q] = v[(size_t)(b + q) * vs];
        }
#pragma unroll
        for (int q = 0; q < 16; q += 4) {
            s0 += (double)uu[q] * (double)vv[q];
            s1 += (double)uu[q + 1] * (double)vv[q + 1];
            s2 += (double)uu[q + 2] * (double)vv[q + 2];
            s3 += (double)uu[q + 3] * (double)vv[q + 3];
        }
    }
    for (; b + 3 < a.frames; b += 4) {
        const float u0 = u[(size_t)b * us], u1 = u[(size_t)(b + 1) * us], u2 = u[(size_t)(b + 2) * us], u3 = u[(size_t)(b + 3) * us];
        const float v0 = v[(size_t)b * vs], v1 = v[(size_t)(b + 1) * vs], v2 = v[(size_t)(b + 2) * vs], v3 = v[(size_t)(b + 3) * vs];
        s0 += (double)u0 * (double)v0;
        s1 += (double)u1 * (double)v1;
        s2 += (double)u2 * (double)v2;
        s3 += (double)u3 * (double)v3;
    }
    for (; b < a.frames; ++b) s0 += (double)u[(size_t)b * us] * (double)v[(size_t)b * vs];
    const float t = (float)((s0 + s1) + (s2 + s3));
    *out = a.accumulate ? *out + t : t;
}

static void fill_adapter(AdapterTrainArgs &a, const float *const cond[3], const float *const w[9], const float *const b[9],
                         const int hidden[3], int tokens, int frames) {
    for (int e = 0; e < 3; ++e) {
        a.cond[e] = cond[e];
        for (int j = 0; j < 3; ++j) {
            a.w[e][j] = w[3 * e + j];
            a.b[e][j] = b[3 * e + j];
        }
    }
    a.h0 = hidden[0]; a.h1 = hidden[1]; a.h2 = hidden[2];
    a.tokens = tokens; a.frames = frames;
}

hipError_t launch_adapter_train_fwd(const float *const cond[3], const float *const w[9], const float *const b[9],
                                    const int hidden[3], int tokens, int frames, float *tokens6, float *a0, float *a1,
                                    hipStream_t st) {
    AdapterTrainArgs a{};
    fill_adapter(a, cond, w, b, hidden, tokens, frames);
    a.tokens6 = tokens6; a.a0 = a0; a.a1 = a1;
    hipLaunchKernelGGL(adapter_train_fwd_kernel, dim3(frames, 3), dim3(256), sizeof(float) * (a.h0 + a.h1), st, a);
    return hipGetLastError();
}

hipError_t launch_adapter_train_bwd(const float *const cond[3], const float *const w[9], const float *const b[9],
                                    const int hidden[3], int tokens, int frames, const float *a0, const float *a1,
                                    const float *dtok, float *da0, float *da1, float *const dw[9], float *const db[9],
                                    bool accumulate, hipStream_t st) {
    if (hidden[1] > kAdaH1Max || hidden[0] > 256 || hidden[2] > 12 * 1024) return hipErrorInvalidValue;
    AdapterTrainArgs a{};
    fill_adapter(a, cond, w, b, hidden, tokens, frames);
    a.a0 = const_cast<float *>(a0); a.a1 = const_cast<float *>(a1);
    a.dtok = dtok; a.da0 = da0; a.da1 = da1; a.accumulate = accumulate;
    for (int e = 0; e < 3; ++e)
        for (int j = 0; j < 3; ++j) { a.dw[e][j] = dw[3 * e + j]; a.db[e][j] = db[3 * e + j]; }
    hipLaunchKernelGGL(adapter_train_dgrad_kernel, dim3(frames, 3), dim3(256), sizeof(float) * a.h2, st, a);
    const int per = a.h2 * a.h1 + a.h2 + a.h1 * a.h0 + a.h1 + a.h0 + a.h0;
    hipLaunchKernelGGL(adapter_train_wgrad_kernel, dim3((3 * per + 255) / 256), dim3(256), 0, st, a);
    return hipGetLastError();
}

static inline uint32_t drop_threshold(float p) { return p > 0.f ? (uint32_t)((double)p * 4294967296.0) : 0u; }
static inline float drop_keep(float p) { return p > 0.f ? 1.f / (1.f - p) : 1.f; }

hipError_t launch_add_ln_fwd(const float *res, const float *y, const float *gamma, const float *beta, float *s_out,
                             float *stats, float *out, int rows, int n, float eps, float dropout_p, uint32_t seed,
                             hipStream_t st) {
    const dim3 grid((rows + 3) / 4), block(256);
    const uint32_t th = drop_threshold(dropout_p);
    const float ks = drop_keep(dropout_p);
    if (n == 64)
        hipLaunchKernelGGL(add_ln_fwd_kernel<1>, grid, block, 0, st, res, y, gamma, beta, s_out, stats, out, rows, eps, seed, th, ks);
    else if (n == 128)
        hipLaunchKernelGGL(add_ln_fwd_kernel<2>, grid, block, 0, st, res, y, gamma, beta, s_out, stats, out, rows, eps, seed, th, ks);
    else if (n == 192)
        hipLaunchKernelGGL(add_ln_fwd_kernel<3>, grid, block, 0, st, res, y, gamma, beta, s_out, stats, out, rows, eps, seed, th, ks);
    else if (n == 256)
        hipLaunchKernelGGL(add_ln_fwd_kernel<4>, grid, block, 0, st, res, y, gamma, beta, s_out, stats, out, rows, eps, seed, th, ks);
    else if (n == 32)
        hipLaunchKernelGGL((add_ln_fwd_kernel<1, true>), grid, block, 0, st, res, y, gamma, beta, s_out, stats, out, rows, eps, seed, th, ks);
    else if (n == 96)
        hipLaunchKernelGGL((add_ln_fwd_kernel<2, true>), grid, block, 0, st, res, y, gamma, beta, s_out, stats, out, rows, eps, seed, th, ks);
    else if (n == 160)
        hipLaunchKernelGGL((add_ln_fwd_kernel<3, true>), grid, block, 0, st, res, y, gamma, beta, s_out, stats, out, rows, eps, seed, th, ks);
    else if (n == 224)
        hipLaunchKernelGGL((add_ln_fwd_kernel<4, true>), grid, block, 0, st, res, y, gamma, beta, s_out, stats, out, rows, eps, seed, th, ks);
    else if (n >= 1 && n <= 64 * kLnAnyNpl)
        hipLaunchKernelGGL(add_ln_fwd_any_kernel, grid, block, 0, st, res, y, gamma, beta, s_out, stats, out, rows, n, eps, seed, th, ks);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

int ln_bwd_blocks(int rows) { return std::max(1, std::min(kColsumMaxSlices, rows / 32)); }

// dgamma / dbeta / dbias (+)= column sums; `slices` holds ln_bwd_blocks(rows) * 3n floats
hipError_t launch_ln_bwd(const float *dy, const float *s, const float *stats, const float *gamma, float *ds, float *dbranch,
                         float *dgamma, float *dbeta, float *dbias, float *slices, int rows, int n, float dropout_p,
                         uint32_t seed, bool accumulate, hipStream_t st) {
    const int nb = ln_bwd_blocks(rows), rpb = (rows + nb - 1) / nb;
    const uint32_t th = drop_threshold(dropout_p);
    const float ks = drop_keep(dropout_p);
    if (n == 64)
        hipLaunchKernelGGL(ln_bwd_kernel<1>, dim3(nb), dim3(256), 0, st, dy, s, stats, gamma, ds, dbranch, slices, rows, rpb, seed, th, ks);
    else if (n == 128)
        hipLaunchKernelGGL(ln_bwd_kernel<2>, dim3(nb), dim3(256), 0, st, dy, s, stats, gamma, ds, dbranch, slices, rows, rpb, seed, th, ks);
    else if (n == 192)
        hipLaunchKernelGGL(ln_bwd_kernel<3>, dim3(nb), dim3(256), 0, st, dy, s, stats, gamma, ds, dbranch, slices, rows, rpb, seed, th, ks);
    else if (n == 256)
        hipLaunchKernelGGL(ln_bwd_kernel<4>, dim3(nb), dim3(256), 0, st, dy, s, stats, gamma, ds, dbranch, slices, rows, rpb, seed, th, ks);
    else if (n == 32)
        hipLaunchKernelGGL((ln_bwd_kernel<1, true>), dim3(nb), dim3(256), 0, st, dy, s, stats, gamma, ds, dbranch, slices, rows, rpb, seed, th, ks);
    else if (n == 96)
        hipLaunchKernelGGL((ln_bwd_kernel<2, true>), dim3(nb), dim3(256), 0, st, dy, s, stats, gamma, ds, dbranch, slices, rows, rpb, seed, th, ks);
    else if (n == 160)
        hipLaunchKernelGGL((ln_bwd_kernel<3, true>), dim3(nb), dim3(256), 0, st, dy, s, stats, gamma, ds, dbranch, slices, rows, rpb, seed, th, ks);
    else if (n == 224)
        hipLaunchKernelGGL((ln_bwd_kernel<4, true>), dim3(nb), dim3(256), 0, st, dy, s, stats, gamma, ds, dbranch, slices, rows, rpb, seed, th, ks);
    else if (n >= 1 && n <= 64 * kLnAnyNpl)
        hipLaunchKernelGGL(ln_bwd_any_kernel, dim3(nb), dim3(256), sizeof(float) * 12 * n, st, dy, s, stats, gamma, ds, dbranch, slices, rows, n,
                           rpb, seed, th, ks);
    else
        return hipErrorInvalidValue;
    // slices[b][0..n) = dgamma partials, [n..2n) = dbeta partials, [2n..3n) = branch bias gradient
    return launch_reduce_slices3(slices, dgamma, dbeta, dbias, n, 3, nb, (size_t)3 * n, accumulate, st);
}

hipError_t launch_act_fwd(int act, const float *a, float *h, int rows, int n, float dropout_p, uint32_t seed, hipStream_t st) {
    if (n & 3) return hipErrorInvalidValue;
    const size_t n4 = (size_t)rows * n / 4;
    const dim3 grid((unsigned)((n4 + 255) / 256)), block(256);
    if (act)
        hipLaunchKernelGGL(act_fwd_kernel<1>, grid, block, 0, st, a, h, n4, n, seed, drop_threshold(dropout_p), drop_keep(dropout_p));
    else
        hipLaunchKernelGGL(act_fwd_kernel<0>, grid, block, 0, st, a, h, n4, n, seed, drop_threshold(dropout_p), drop_keep(dropout_p));
    return hipGetLastError();
}

// dbias (+)= column sums of da; `slices` holds ln_bwd_blocks(rows) * n floats; n = 256 or 512
hipError_t launch_act_bwd(int act, const float *a, float *dh, float *dbias, float *slices, int rows, int n, float dropout_p,
                          uint32_t seed, bool accumulate, hipStream_t st) {
    if (n % 4 || n / 4 > 256) return hipErrorInvalidValue;
    const int nb = ln_bwd_blocks(rows), rpb = (rows + nb - 1) / nb;
    if (act)
        hipLaunchKernelGGL(act_bwd_kernel<1>, dim3(nb), dim3(256), 0, st, a, dh, slices, rows, n, rpb, seed, drop_threshold(dropout_p), drop_keep(dropout_p));
    else
        hipLaunchKernelGGL(act_bwd_kernel<0>, dim3(nb), dim3(256), 0, st, a, dh, slices, rows, n, rpb, seed, drop_threshold(dropout_p), drop_keep(dropout_p));
    return launch_reduce_slices(slices, dbias, n, nb, (size_t)n, accumulate, st);
}

hipError_t launch_add(const float *a, const float *b, float *out, size_t n, hipStream_t st) {
    const size_t n4 = n / 4;
    hipLaunchKernelGGL(add_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, a, b, out, n4);
    return hipGetLastError();
}

}  // namespace aft
