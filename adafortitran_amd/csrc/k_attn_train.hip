// k_attn_train.hip -- multi-head self-attention for the training path: forward that keeps the
// log-sum-exp of every row, and the backward pass.
//
// Reference semantics: torch.nn.MultiheadAttention inside nn.TransformerEncoderLayer
// (reference src/models/blocks/encoders.py:44-55): P = softmax(Q K^T / sqrt(dh)), dropout on P,
// O = P_drop V, per (plane, head); backward
//     D_i  = dO_i . O_i                      dV = P_drop^T dO
//     dP   = (dO V^T) o mask/(1-p)           dS = P o (dP - D)
//     dQ   = dS K / sqrt(dh)                 dK = dS^T Q / sqrt(dh)
// Operands are PyTorch row-major: qkv [rows][3d] (q | k | v column blocks), o / dO [rows][d].
//
// MI355X mapping: one wave per 32-row tile of one (plane, head); no LDS, no barriers.  Everything is
// v_mfma_f32_32x32x2_f32 (exact fp32).  An MFMA accumulator has lane = column, register = row, so it
// can be fed back as the B operand of a product that contracts over its ROW index.  Each pass picks
// the orientation of S that makes that true:
//   forward, dQ pass: S^T = K Q^T  (rows = keys)    -> O^T = V^T P^T, dQ^T = K^T dS^T contract over keys
//   dK/dV pass:       S   = Q K^T  (rows = queries) -> dV^T = dO^T P, dK^T = Q^T dS contract over queries
// so the probabilities never leave registers.  Row statistics are base-2 (scores carry log2 e/sqrt(dh)).
// The backward recomputes P from the saved LSE twice (once per pass) instead of using atomics, which
// keeps it deterministic.  Dropout masks are a counter-based hash of (seed, site, plane, head, q, k),
// recomputed in the backward, never stored.
#include "aft_internal.h"

namespace aft {

struct AttnTrainArgs {
    const float *qkv;     // [rows][3d]
    const float *o;       // [rows][d]  forward output (backward input)
    const float *d_o;     // [rows][d]
    float *out;           // forward: o ; backward: dqkv [rows][3d]
    float *lse;           // [planes][H][tokens] base-2 log-sum-exp of the scaled scores
    float *dsum;          // [planes][H][tokens] D_i
    int planes, tokens, heads, d, ntiles;
    float scale2, scale;  // log2(e)/sqrt(dh), 1/sqrt(dh)
    float keep_scale;     // 1/(1-p)
    uint32_t threshold;   // drop when hash < threshold (0 = no dropout)
    uint32_t seed;
};

__device__ __forceinline__ uint32_t mix32(uint32_t x) {   // murmur3 finaliser
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ float drop_factor(uint32_t seed, uint32_t idx, uint32_t threshold, float keep_scale) {
    return mix32(idx * 0x9E3779B1u ^ seed) >= threshold ? keep_scale : 0.f;
}

__device__ __forceinline__ int rowmap(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// lane <-> row (token), 16 of the 32 head features: slot kb <-> feature 8*(kb>>2) + 4h + (kb&3)
__device__ __forceinline__ void load_rowfrag(const float *__restrict__ base, int ld, int tok, int tokens, int h, float mul,
                                             float (&f)[16]) {
    const bool ok = tok < tokens;
    const float *p = base + (size_t)min(tok, tokens - 1) * ld + 4 * h;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(p + 8 * q);
#pragma unroll
        for (int c = 0; c < 4; ++c) f[4 * q + c] = ok ? v[c] * mul : 0.f;
    }
}
// lane <-> feature j, slot r <-> token tile*32 + rowmap(r, h)  (the transposed operand)
__device__ __forceinline__ void load_colfrag(const float *__restrict__ base, int ld, int tile, int tokens, int j, int h,
                                             float (&f)[16]) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int tok = tile * 32 + rowmap(r, h);
        f[r] = tok < tokens ? base[(size_t)tok * ld + j] : 0.f;
    }
}
__device__ __forceinline__ f32x16 mma16(const float (&a)[16], const float (&b)[16], f32x16 acc) {
#pragma unroll
    for (int k = 0; k < 16; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], b[k], acc, 0, 0, 0);
    return acc;
}
__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int e = 0; e < 16; ++e) z[e] = 0.f;
    return z;
}
// accumulator [row = feature][col = token lane] -> row-major dst[token][feature], 4 x b128 per lane
__device__ __forceinline__ void store_transposed(float *__restrict__ base, int ld, int tok, int tokens, int h, const f32x16 &acc,
                                                 float mul) {
    if (tok >= tokens) return;
    float *p = base + (size_t)tok * ld + 4 * h;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        *reinterpret_cast<f32x4 *>(p + 8 * q) = f32x4{acc[4 * q] * mul, acc[4 * q + 1] * mul, acc[4 * q + 2] * mul, acc[4 * q + 3] * mul};
}

// ---- forward: one wave per (plane, head, query tile) ----
__global__ __launch_bounds__(256) void attn_train_fwd_kernel(const AttnTrainArgs a) {
    const int task = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (task >= a.planes * a.heads * a.ntiles) return;
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const int qt = task % a.ntiles, ph = task / a.ntiles, head = ph % a.heads, plane = ph / a.heads;
    const int ld = 3 * a.d;
    const float *qb = a.qkv + (size_t)plane * a.tokens * ld + head * 32;
    const float *kb = qb + a.d, *vb = qb + 2 * a.d;
    const int query = qt * 32 + j;

    float qf[16];
    load_rowfrag(qb, ld, query, a.tokens, h, a.scale2, qf);
    float m = -__builtin_inff(), l = 0.f;
    f32x16 o = zero16();
    const uint32_t idx0 = ((uint32_t)ph * a.tokens + min(query, a.tokens - 1)) * a.tokens;
    for (int kt = 0; kt < a.ntiles; ++kt) {
        float kf[16], vt[16];
        load_rowfrag(kb, ld, kt * 32 + j, a.tokens, h, 1.f, kf);
        load_colfrag(vb, ld, kt, a.tokens, j, h, vt);
        f32x16 s = mma16(kf, qf, zero16());               // [row = key][col = query]
        float mx = -__builtin_inff();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (kt * 32 + rowmap(r, h) >= a.tokens) s[r] = -__builtin_inff();
            mx = fmaxf(mx, s[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float mn = fmaxf(m, mx), alpha = exp2f(m - mn);
        float p[16], sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            p[r] = exp2f(s[r] - mn);
            sum += p[r];
            if (a.threshold) p[r] *= drop_factor(a.seed, idx0 + kt * 32 + rowmap(r, h), a.threshold, a.keep_scale);
        }
        l = l * alpha + sum;
        m = mn;
#pragma unroll
        for (int e = 0; e < 16; ++e) o[e] *= alpha;
        o = mma16(vt, p, o);                              // [row = feature][col = query]
    }
    l += __shfl_xor(l, 32);
    store_transposed(a.out + (size_t)plane * a.tokens * a.d + head * 32, a.d, query, a.tokens, h, o, 1.f / l);
    if (h == 0 && query < a.tokens) a.lse[(size_t)ph * a.tokens + query] = m + log2f(l);
}

// ---- D_i = dO_i . O_i per (plane, head, row) ----
__global__ __launch_bounds__(256) void attn_bwd_prep_kernel(const AttnTrainArgs a) {
    const int i = blockIdx.x * 256 + threadIdx.x;   // (plane, token, head)
    if (i >= a.planes * a.tokens * a.heads) return;
    const int head = i % a.heads, row = i / a.heads, plane = row / a.tokens, tok = row % a.tokens;
    const float *po = a.o + (size_t)row * a.d + head * 32, *pd = a.d_o + (size_t)row * a.d + head * 32;
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const f32x4 x = *reinterpret_cast<const f32x4 *>(po + 4 * q), y = *reinterpret_cast<const f32x4 *>(pd + 4 * q);
        s += x[0] * y[0] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3];
    }
    a.dsum[((size_t)plane * a.heads + head) * a.tokens + tok] = s;
}

// ---- dK, dV: one wave per (plane, head, key tile), loop over query tiles ----
__global__ __launch_bounds__(256) void attn_bwd_kv_kernel(const AttnTrainArgs a) {
    const int task = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (task >= a.planes * a.heads * a.ntiles) return;
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const int kt = task % a.ntiles, ph = task / a.ntiles, head = ph % a.heads, plane = ph / a.heads;
    const int ld = 3 * a.d;
    const float *qb = a.qkv + (size_t)plane * a.tokens * ld + head * 32;
    const float *kb = qb + a.d, *vb = qb + 2 * a.d;
    const float *dob = a.d_o + (size_t)plane * a.tokens * a.d + head * 32;
    const float *lse = a.lse + (size_t)ph * a.tokens, *dsum = a.dsum + (size_t)ph * a.tokens;
    const int key = kt * 32 + j;
    const bool key_ok = key < a.tokens;

    float kf[16], vf[16];
    load_rowfrag(kb, ld, key, a.tokens, h, a.scale2, kf);   // B operands: lane <-> key
    load_rowfrag(vb, ld, key, a.tokens, h, 1.f, vf);
    f32x16 dv = zero16(), dk = zero16();
    for (int qt = 0; qt < a.ntiles; ++qt) {
        float qf[16], dof[16], qT[16], doT[16];
        load_rowfrag(qb, ld, qt * 32 + j, a.tokens, h, 1.f, qf);      // A operands: lane <-> query
        load_rowfrag(dob, a.d, qt * 32 + j, a.tokens, h, 1.f, dof);
        load_colfrag(qb, ld, qt, a.tokens, j, h, qT);                  // A operands: lane <-> feature
        load_colfrag(dob, a.d, qt, a.tokens, j, h, doT);
        const f32x16 s = mma16(qf, kf, zero16());                      // [row = query][col = key]
        const f32x16 dp = mma16(dof, vf, zero16());
        float pd[16], ds[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int query = qt * 32 + rowmap(r, h);
            const bool ok = key_ok && query < a.tokens;
            const int qc = min(query, a.tokens - 1);
            const float p = ok ? exp2f(s[r] - lse[qc]) : 0.f;
            const float f = a.threshold ? drop_factor(a.seed, ((uint32_t)ph * a.tokens + qc) * a.tokens + min(key, a.tokens - 1),
                                                      a.threshold, a.keep_scale) : 1.f;
            pd[r] = p * f;
            ds[r] = p * (dp[r] * f - dsum[qc]);
        }
        dv = mma16(doT, pd, dv);                                       // [row = feature][col = key]
        dk = mma16(qT, ds, dk);
    }
    float *dst = a.out + (size_t)plane * a.tokens * ld + head * 32;
    store_transposed(dst + a.d, ld, key, a.tokens, h, dk, a.scale);
    store_transposed(dst + 2 * a.d, ld, key, a.tokens, h, dv, 1.f);
}

// ---- dQ: one wave per (plane, head, query tile), loop over key tiles ----
__global__ __launch_bounds__(256) void attn_bwd_q_kernel(const AttnTrainArgs a) {
    const int task = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (task >= a.planes * a.heads * a.ntiles) return;
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const int qt = task % a.ntiles, ph = task / a.ntiles, head = ph % a.heads, plane = ph / a.heads;
    const int ld = 3 * a.d;
    const float *qb = a.qkv + (size_t)plane * a.tokens * ld + head * 32;
    const float *kb = qb + a.d, *vb = qb + 2 * a.d;
    const float *dob = a.d_o + (size_t)plane * a.tokens * a.d + head * 32;
    const int query = qt * 32 + j, qc = min(query, a.tokens - 1);
    const bool q_ok = query < a.tokens;
    const float lse = a.lse[(size_t)ph * a.tokens + qc], dsum = a.dsum[(size_t)ph * a.tokens + qc];
    const uint32_t idx0 = ((uint32_t)ph * a.tokens + qc) * a.tokens;

    float qf[16], dof[16];
    load_rowfrag(qb, ld, query, a.tokens, h, a.scale2, qf);   // B operands: lane <-> query
    load_rowfrag(dob, a.d, query, a.tokens, h, 1.f, dof);
    f32x16 dq = zero16();
    for (int kt = 0; kt < a.ntiles; ++kt) {
        float kf[16], vf[16], kT[16];
        load_rowfrag(kb, ld, kt * 32 + j, a.tokens, h, 1.f, kf);      // A operands: lane <-> key
        load_rowfrag(vb, ld, kt * 32 + j, a.tokens, h, 1.f, vf);
        load_colfrag(kb, ld, kt, a.tokens, j, h, kT);                  // A operand: lane <-> feature
        const f32x16 s = mma16(kf, qf, zero16());                      // [row = key][col = query]
        const f32x16 dp = mma16(vf, dof, zero16());
        float ds[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kt * 32 + rowmap(r, h);
            const bool ok = q_ok && key < a.tokens;
            const float p = ok ? exp2f(s[r] - lse) : 0.f;
            const float f = a.threshold ? drop_factor(a.seed, idx0 + min(key, a.tokens - 1), a.threshold, a.keep_scale) : 1.f;
            ds[r] = p * (dp[r] * f - dsum);
        }
        dq = mma16(kT, ds, dq);                                        // [row = feature][col = query]
    }
    store_transposed(a.out + (size_t)plane * a.tokens * ld + head * 32, ld, query, a.tokens, h, dq, a.scale);
}

static AttnTrainArgs make_args(const aft_config &c, int planes, int tokens, float dropout_p, uint32_t seed) {
    AttnTrainArgs a{};
    a.planes = planes; a.tokens = tokens; a.heads = c.num_head; a.d = c.model_dim;
    a.ntiles = (tokens + 31) / 32;
    const float inv = 1.f / sqrtf((float)(c.model_dim / c.num_head));
    a.scale = inv;
    a.scale2 = inv * 1.4426950408889634f;
    a.keep_scale = dropout_p > 0.f ? 1.f / (1.f - dropout_p) : 1.f;
    a.threshold = dropout_p > 0.f ? (uint32_t)((double)dropout_p * 4294967296.0) : 0u;
    a.seed = seed;
    return a;
}

hipError_t launch_attn_train_fwd(const aft_config &c, const float *qkv, float *o, float *lse, int planes, int tokens,
                                 float dropout_p, uint32_t seed, hipStream_t st) {
    AttnTrainArgs a = make_args(c, planes, tokens, dropout_p, seed);
    a.qkv = qkv; a.out = o; a.lse = lse;
    const int tasks = planes * a.heads * a.ntiles;
    hipLaunchKernelGGL(attn_train_fwd_kernel, dim3((tasks + 3) / 4), dim3(256), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_attn_train_bwd(const aft_config &c, const float *qkv, const float *o, const float *d_o, const float *lse,
                                 float *dsum, float *dqkv, int planes, int tokens, float dropout_p, uint32_t seed,
                                 hipStream_t st) {
    AttnTrainArgs a = make_args(c, planes, tokens, dropout_p, seed);
    a.qkv = qkv; a.o = o; a.d_o = d_o; a.lse = const_cast<float *>(lse); a.dsum = dsum; a.out = dqkv;
    const int tasks = planes * a.heads * a.ntiles;
    hipLaunchKernelGGL(attn_bwd_prep_kernel, dim3((planes * tokens * a.heads + 255) / 256), dim3(256), 0, st, a);
    hipLaunchKernelGGL(attn_bwd_kv_kernel, dim3((tasks + 3) / 4), dim3(256), 0, st, a);
    hipLaunchKernelGGL(attn_bwd_q_kernel, dim3((tasks + 3) / 4), dim3(256), 0, st, a);
    return hipGetLastError();
}

}  // namespace aft
