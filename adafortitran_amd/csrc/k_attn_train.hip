// k_attn_train.hip -- multi-head self-attention for the training path: forward that keeps the
// log-sum-exp of every row, and the backward pass.
//
// Reference semantics: torch.nn.MultiheadAttention inside nn.TransformerEncoderLayer
// (reference src/models/blocks/encoders.py:44-55): P = softmax(Q K^T / sqrt(dh)), dropout on P,
// O = P_drop V, per (plane, head); backward
//     D_i  = dO_i . O_i (in the dQ pass)     dV = P_drop^T dO
//     dP   = (dO V^T) o mask/(1-p)           dS = P o (dP - D)
//     dQ   = dS K / sqrt(dh)                 dK = dS^T Q / sqrt(dh)
// Operands are PyTorch row-major: qkv [rows][3d] (q | k | v column blocks), o / dO [rows][d].
//
// MI355X mapping: one wave per 32-row tile of one (plane, head), three such waves per workgroup sharing
// the tiles they all walk through LDS.  Everything is
// v_mfma_f32_32x32x2_f32 (exact fp32).  An MFMA accumulator has lane = column, register = row, so it
// can be fed back as the B operand of a product that contracts over its ROW index.  Each pass picks
// the orientation of S that makes that true:
//   forward, dQ pass: S^T = K Q^T  (rows = keys)    -> O^T = V^T P^T, dQ^T = K^T dS^T contract over keys
//   dK/dV pass:       S   = Q K^T  (rows = queries) -> dV^T = dO^T P, dK^T = Q^T dS contract over queries
// so the probabilities never leave registers.  Row statistics are base-2 (scores carry log2 e/sqrt(dh)).
// The backward recomputes P from the saved LSE.  attn_bwd_kernel (the default) produces dQ, dK and dV in ONE pass over the
// (query tile, key tile) pairs: dK / dV in the second orientation, the pair's dS tile through LDS for dQ, the per-tile dQ
// contributions added in a fixed order -- deterministic, no atomics.  attn_bwd_q_kernel + attn_bwd_kv_kernel are the two-pass
// form it replaced (S and dP recomputed once per pass), kept for shapes whose LDS tables do not fit and as the A/B reference
// (AFT_TRAIN_ATTN_BWD_SPLIT).  Dropout masks are a counter-based function of (seed, site, plane, head, q, k) -- a hashed
// word per query row times a hashed word per key column, see drop_keep below -- recomputed in the backward, never stored.
#include <cstdlib>

#include "aft_internal.h"

namespace aft {

using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

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
    float dq_scale;       // scale / scale2: attn_bwd_kernel stages K x scale2
    uint32_t threshold;   // drop when hash < threshold (0 = no dropout)
    uint32_t seed;
};

// Dropout mask of P: the factored mask of aft_internal.h (dropmask_*), rows = queries, columns = keys of one (plane, head):
// the per-key words are hashed once per workgroup into LDS, the per-query word once per lane (or the other way round in
// the dK/dV pass), three VALU instructions per element.
__device__ __forceinline__ uint32_t drop_row_word(uint32_t seed, uint32_t idx) { return dropmask_row_word(seed, idx); }
__device__ __forceinline__ uint32_t drop_col_word(uint32_t seed, uint32_t idx) { return dropmask_col_word(seed, idx); }
__device__ __forceinline__ bool drop_keep(uint32_t row_word, uint32_t col_word, uint32_t threshold) {
    return dropmask_keep(row_word, col_word, threshold);
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
// product into a fresh accumulator: the first MFMA takes the inline constant 0 as C (no 16 v_mov to clear it)
__device__ __forceinline__ f32x16 mma16z(const float (&a)[16], const float (&b)[16]) {
    f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], f32x16{0}, 0, 0, 0);
#pragma unroll
    for (int k = 1; k < 16; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], b[k], acc, 0, 0, 0);
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

// ---- LDS staging shared by the three kernels ----
// A workgroup is kAtWaves waves working on consecutive "own" tiles of one (plane, head); all of them
// walk the same sequence of "other" tiles, so each other-tile (32 rows x 32 head features of up to two
// matrices, plus two per-row scalars) is fetched from HBM/L2 once per workgroup, coalesced, into LDS
// (row stride 36 floats), double-buffered: the loads of tile t+1 are in flight during the MFMAs of t.
constexpr int kAtWaves = 3, kAtThreads = 64 * kAtWaves, kAtLd = 36;
constexpr int kAtTileFloats = 32 * kAtLd;

struct StageRegs { f32x4 a[2], b[2]; float s0, s1; };

// The fetch side of the staging: buffer resources of the two matrices ([tokens][ld], already offset to the head's
// columns) and of the two per-row scalars (NULL -> a zero-length resource: reads return 0), and the lane's byte
// offsets, computed ONCE per workgroup; a tile only adds a scalar offset.  (Round 2: per tile this was two 64-bit
// multiply-adds, four selects and six compares per lane -- ~40 VALU instructions beside 32 MFMAs.)  The hardware range
// check does not cover the scalar offset, so rows of the ragged last tile beyond `tokens` get an out-of-range lane
// offset explicitly: they read as zero.
using AtSrd = __amdgpu_buffer_rsrc_t;
constexpr unsigned kAtOutOfRange = 0x7ffffff0u;
struct Stager {
    AtSrd a, b, s0, s1;
    unsigned va[2], vb[2], vs;   // lane byte offsets into a / b (two 16-byte elements per lane) and into the scalars
    unsigned ta, tb;             // bytes per 32-row tile
    int last_row[2];             // tokens - 1 - (row of element u): the element exists in tile t iff 32 t <= last_row
    int last_s;
};
__device__ __forceinline__ AtSrd at_srd(const float *p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ Stager make_stager(const float *A, int lda, const float *B, int ldb, const float *s0, const float *s1,
                                              int tokens, int tid) {
    Stager g;
    g.a = at_srd(A, (unsigned)((tokens - 1) * lda + 32) * 4u);
    g.b = at_srd(B, (unsigned)((tokens - 1) * ldb + 32) * 4u);
    g.s0 = at_srd(s0, s0 ? (unsigned)tokens * 4u : 0u);
    g.s1 = at_srd(s1, s1 ? (unsigned)tokens * 4u : 0u);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int e = tid + kAtThreads * u, row = e >> 3, q4 = e & 7;   // vec4 index 0..255: row = e >> 3, quad = e & 7
        g.va[u] = e < 256 ? (unsigned)(row * lda + 4 * q4) * 4u : kAtOutOfRange;
        g.vb[u] = e < 256 ? (unsigned)(row * ldb + 4 * q4) * 4u : kAtOutOfRange;
        g.last_row[u] = tokens - 1 - row;
    }
    g.vs = tid < 32 ? (unsigned)tid * 4u : kAtOutOfRange;
    g.last_s = tokens - 1 - (tid & 31);
    g.ta = 32u * (unsigned)lda * 4u;
    g.tb = 32u * (unsigned)ldb * 4u;
    return g;
}
__device__ __forceinline__ f32x4 at_ld4(AtSrd r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ float at_ld1(AtSrd r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
// fetch tile `tile` into registers; rows >= tokens read as zero
__device__ __forceinline__ void stage_fetch(StageRegs &r, const Stager &g, int tile) {
    const int t32 = tile * 32;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const bool in = t32 <= g.last_row[u];
        r.a[u] = at_ld4(g.a, in ? g.va[u] : kAtOutOfRange, (unsigned)tile * g.ta);
        r.b[u] = at_ld4(g.b, in ? g.vb[u] : kAtOutOfRange, (unsigned)tile * g.tb);
    }
    const unsigned vs = t32 <= g.last_s ? g.vs : kAtOutOfRange;
    r.s0 = at_ld1(g.s0, vs, (unsigned)tile * 128u);
    r.s1 = at_ld1(g.s1, vs, (unsigned)tile * 128u);
}
// buf: [A tile][B tile][s0 32][s1 32]
__device__ __forceinline__ void stage_store(const StageRegs &r, float *__restrict__ buf, int tid) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int e = tid + kAtThreads * u, row = e >> 3, q4 = e & 7;
        if (e < 256) {
            *reinterpret_cast<f32x4 *>(buf + row * kAtLd + 4 * q4) = r.a[u];
            *reinterpret_cast<f32x4 *>(buf + kAtTileFloats + row * kAtLd + 4 * q4) = r.b[u];
        }
    }
    if (tid < 32) {
        buf[2 * kAtTileFloats + tid] = r.s0;
        buf[2 * kAtTileFloats + 32 + tid] = r.s1;
    }
}
constexpr int kAtBufFloats = 2 * kAtTileFloats + 64;

// operand fragments out of a staged tile: lane <-> row (rowfrag) or lane <-> feature (colfrag)
__device__ __forceinline__ void lds_rowfrag(const float *__restrict__ t, int j, int h, float (&f)[16]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(t + j * kAtLd + 8 * q + 4 * h);
#pragma unroll
        for (int c = 0; c < 4; ++c) f[4 * q + c] = v[c];
    }
}
__device__ __forceinline__ void lds_colfrag(const float *__restrict__ t, int j, int h, float (&f)[16]) {
#pragma unroll
    for (int r = 0; r < 16; ++r) f[r] = t[rowmap(r, h) * kAtLd + j];
}
// per-row scalars of the accumulator rows this lane half holds: rows rowmap(r, h), r = 0..15
__device__ __forceinline__ void lds_rowscalars(const float *__restrict__ s, int h, float (&f)[16]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(s + 8 * q + 4 * h);
#pragma unroll
        for (int c = 0; c < 4; ++c) f[4 * q + c] = v[c];
    }
}

// the 16 mask words of the accumulator rows this lane half holds
__device__ __forceinline__ void lds_rowwords(const uint32_t *__restrict__ s, int h, uint32_t (&f)[16]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const u32x4 v = *reinterpret_cast<const u32x4 *>(s + 8 * q + 4 * h);
#pragma unroll
        for (int c = 0; c < 4; ++c) f[4 * q + c] = v[c];
    }
}

// ---- forward: own = query tile, walks the key tiles (staged: K, V) ----
// NB = 32-feature blocks per head: 1 (head dim 32, the tuned shape; head dim 16 runs here as zero-padded heads) or 2 (head dim 64,
// round 5: S sums both blocks' products, O is one accumulator per block; covered, not tuned).
template <int NB>
__device__ __forceinline__ void attn_train_fwd_body(const AttnTrainArgs &a, float (*lds)[NB * kAtBufFloats], uint32_t *words) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int groups = (a.ntiles + kAtWaves - 1) / kAtWaves;
    const int ph = blockIdx.x / groups, qt = (blockIdx.x % groups) * kAtWaves + wave;
    const int head = ph % a.heads, plane = ph / a.heads;
    const bool active = qt < a.ntiles;
    const int ld = 3 * a.d;
    const float *qb = a.qkv + (size_t)plane * a.tokens * ld + head * (32 * NB);
    const float *kb = qb + a.d, *vb = qb + 2 * a.d;
    const int query = qt * 32 + j;

    float qf[NB][16];
#pragma unroll
    for (int b = 0; b < NB; ++b) load_rowfrag(qb + 32 * b, ld, active ? query : 0, a.tokens, h, a.scale2, qf[b]);
    float m = -__builtin_inff(), l = 0.f;
    f32x16 o[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) o[b] = zero16();
    const uint32_t row_word = drop_row_word(a.seed, (uint32_t)ph * a.tokens + min(query, a.tokens - 1));
    if (a.threshold)
        for (int i = tid; i < a.ntiles * 32; i += kAtThreads) words[i] = drop_col_word(a.seed, (uint32_t)ph * a.tokens + i);
    StageRegs sr[NB];
    Stager sg[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        sg[b] = make_stager(kb + 32 * b, ld, vb + 32 * b, ld, nullptr, nullptr, a.tokens, tid);
        stage_fetch(sr[b], sg[b], 0);
        stage_store(sr[b], lds[0] + b * kAtBufFloats, tid);
    }
    __syncthreads();
    for (int kt = 0; kt < a.ntiles; ++kt) {
        const float *buf = lds[kt & 1];
        if (kt + 1 < a.ntiles) {
#pragma unroll
            for (int b = 0; b < NB; ++b) stage_fetch(sr[b], sg[b], kt + 1);
        }
        if (active) {
            f32x16 s;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                float kf[16];
                lds_rowfrag(buf + b * kAtBufFloats, j, h, kf);
                s = b == 0 ? mma16z(kf, qf[0]) : mma16(kf, qf[b], s);   // [row = key][col = query]
            }
            if (kt == a.ntiles - 1) {                         // only the last key tile can be ragged
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kt * 32 + rowmap(r, h) >= a.tokens) s[r] = -__builtin_inff();
            }
            float mx = s[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float mn = fmaxf(m, mx), alpha = __builtin_amdgcn_exp2f(m - mn);
            float p[16], sum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                p[r] = __builtin_amdgcn_exp2f(s[r] - mn);
                sum += p[r];
            }
            if (a.threshold) {
                uint32_t cw[16];
                lds_rowwords(words + kt * 32, h, cw);
#pragma unroll
                for (int r = 0; r < 16; ++r) p[r] = drop_keep(row_word, cw[r], a.threshold) ? p[r] : 0.f;
            }
            l = l * alpha + sum;
            m = mn;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                float vt[16];
                lds_colfrag(buf + b * kAtBufFloats + kAtTileFloats, j, h, vt);
#pragma unroll
                for (int e = 0; e < 16; ++e) o[b][e] *= alpha;
                o[b] = mma16(vt, p, o[b]);                    // [row = feature][col = query]
            }
        }
        if (kt + 1 < a.ntiles) {
#pragma unroll
            for (int b = 0; b < NB; ++b) stage_store(sr[b], lds[(kt + 1) & 1] + b * kAtBufFloats, tid);
        }
        __syncthreads();
    }
    if (!active) return;
    l += __shfl_xor(l, 32);
#pragma unroll
    for (int b = 0; b < NB; ++b)
        store_transposed(a.out + (size_t)plane * a.tokens * a.d + head * (32 * NB) + 32 * b, a.d, query, a.tokens, h, o[b], a.keep_scale / l);
    if (h == 0 && query < a.tokens) a.lse[(size_t)ph * a.tokens + query] = m + log2f(l);
}
__global__ __launch_bounds__(kAtThreads) __attribute__((amdgpu_waves_per_eu(4, 4))) void attn_train_fwd_kernel(const AttnTrainArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[2][kAtBufFloats];
    extern __shared__ __attribute__((aligned(16))) uint32_t words[];   // [ntiles * 32] column (key) words of the dropout mask
    attn_train_fwd_body<1>(a, lds, words);
}
__global__ __launch_bounds__(kAtThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_train_fwd64_kernel(const AttnTrainArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[2][2 * kAtBufFloats];
    extern __shared__ __attribute__((aligned(16))) uint32_t words[];
    attn_train_fwd_body<2>(a, lds, words);
}

// ---- dK, dV: own = key tile, walks the query tiles (staged: Q, dO, lse, D) ----
template <int NB>
__device__ __forceinline__ void attn_bwd_kv_body(const AttnTrainArgs &a, float (*lds)[NB * kAtBufFloats], uint32_t *words) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int groups = (a.ntiles + kAtWaves - 1) / kAtWaves;
    const int ph = blockIdx.x / groups, kt = (blockIdx.x % groups) * kAtWaves + wave;
    const int head = ph % a.heads, plane = ph / a.heads;
    const bool active = kt < a.ntiles;
    const int ld = 3 * a.d;
    const float *qb = a.qkv + (size_t)plane * a.tokens * ld + head * (32 * NB);
    const float *kb = qb + a.d, *vb = qb + 2 * a.d;
    const float *dob = a.d_o + (size_t)plane * a.tokens * a.d + head * (32 * NB);
    const float *lse = a.lse + (size_t)ph * a.tokens, *dsum = a.dsum + (size_t)ph * a.tokens;
    const int key = kt * 32 + j;
    const int keyc = min(key, a.tokens - 1);

    float kf[NB][16], vf[NB][16];
    f32x16 dv[NB], dk[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        load_rowfrag(kb + 32 * b, ld, active ? key : 0, a.tokens, h, a.scale2, kf[b]);   // B operands: lane <-> key
        load_rowfrag(vb + 32 * b, ld, active ? key : 0, a.tokens, h, 1.f, vf[b]);
        dv[b] = zero16();
        dk[b] = zero16();
    }
    const uint32_t col_word = drop_col_word(a.seed, (uint32_t)ph * a.tokens + keyc);
    if (a.threshold)
        for (int i = tid; i < a.ntiles * 32; i += kAtThreads) words[i] = drop_row_word(a.seed, (uint32_t)ph * a.tokens + min(i, a.tokens - 1));
    StageRegs sr;
    Stager sg[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        sg[b] = make_stager(qb + 32 * b, ld, dob + 32 * b, a.d, lse, dsum, a.tokens, tid);
        stage_fetch(sr, sg[b], 0);
        stage_store(sr, lds[0] + b * kAtBufFloats, tid);
    }
    __syncthreads();
    for (int qt = 0; qt < a.ntiles; ++qt) {
        const float *buf = lds[qt & 1];
        if (active) {
            // operands are read from LDS right before their product so that at most two 16-register
            // fragments are live next to the accumulators (3 waves per SIMD need <= 168 VGPRs)
            f32x16 s, dp;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                float qf[16];
                lds_rowfrag(buf + b * kAtBufFloats, j, h, qf);                       // A operand: lane <-> query
                s = b == 0 ? mma16z(qf, kf[0]) : mma16(qf, kf[b], s);                // [row = query][col = key]
            }
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                float dof[16];
                lds_rowfrag(buf + b * kAtBufFloats + kAtTileFloats, j, h, dof);
                dp = b == 0 ? mma16z(dof, vf[0]) : mma16(dof, vf[b], dp);
            }
            // No range masks: query rows beyond the plane were staged as zeros (Q, dO rows = 0, lse = D = 0), so
            // whatever finite p / ds they get meets a zero A-operand column; key lanes beyond it are never stored.
            // pd carries P o mask, the 1/(1-p) factor is applied to dV once at the end.
            float pd[16], ds[16];
            {
                float ls[16], dsm[16];
                lds_rowscalars(buf + 2 * kAtTileFloats, h, ls);
                lds_rowscalars(buf + 2 * kAtTileFloats + 32, h, dsm);
#pragma unroll
                for (int r = 0; r < 16; ++r) pd[r] = __builtin_amdgcn_exp2f(s[r] - ls[r]);
#pragma unroll
                for (int r = 0; r < 16; ++r) ds[r] = pd[r] * dsm[r];
                if (a.threshold) {
                    uint32_t rw[16];
                    lds_rowwords(words + qt * 32, h, rw);
#pragma unroll
                    for (int r = 0; r < 16; ++r) pd[r] = drop_keep(rw[r], col_word, a.threshold) ? pd[r] : 0.f;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) ds[r] = fmaf(pd[r] * dp[r], a.keep_scale, -ds[r]);   // P o (dP o mask/(1-p) - D)
            }
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                float qT[16], doT[16];
                lds_colfrag(buf + b * kAtBufFloats + kAtTileFloats, j, h, doT);          // A operands: lane <-> feature
                dv[b] = mma16(doT, pd, dv[b]);                                       // [row = feature][col = key]
                lds_colfrag(buf + b * kAtBufFloats, j, h, qT);
                dk[b] = mma16(qT, ds, dk[b]);
            }
        }
        if (qt + 1 < a.ntiles) {   // fetched after the products: the 18 staging registers are not live across them
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                stage_fetch(sr, sg[b], qt + 1);
                stage_store(sr, lds[(qt + 1) & 1] + b * kAtBufFloats, tid);
            }
        }
        __syncthreads();
    }
    if (!active) return;
    float *dst = a.out + (size_t)plane * a.tokens * ld + head * (32 * NB);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        store_transposed(dst + a.d + 32 * b, ld, key, a.tokens, h, dk[b], a.scale);
        store_transposed(dst + 2 * a.d + 32 * b, ld, key, a.tokens, h, dv[b], a.keep_scale);
    }
}
__global__ __launch_bounds__(kAtThreads) __attribute__((amdgpu_waves_per_eu(3, 3))) void attn_bwd_kv_kernel(const AttnTrainArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[2][kAtBufFloats];
    extern __shared__ __attribute__((aligned(16))) uint32_t words[];   // [ntiles * 32] row (query) words of the dropout mask
    attn_bwd_kv_body<1>(a, lds, words);
}
__global__ __launch_bounds__(kAtThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_bwd_kv64_kernel(const AttnTrainArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[2][2 * kAtBufFloats];
    extern __shared__ __attribute__((aligned(16))) uint32_t words[];
    attn_bwd_kv_body<2>(a, lds, words);
}

// ---- dQ: own = query tile, walks the key tiles (staged: K, V) ----
template <int NB>
__device__ __forceinline__ void attn_bwd_q_body(const AttnTrainArgs &a, float (*lds)[NB * kAtBufFloats], uint32_t *words) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, h = lane >> 5;
    const int groups = (a.ntiles + kAtWaves - 1) / kAtWaves;
    const int ph = blockIdx.x / groups, qt = (blockIdx.x % groups) * kAtWaves + wave;
    const int head = ph % a.heads, plane = ph / a.heads;
    const bool active = qt < a.ntiles;
    const int ld = 3 * a.d;
    const float *qb = a.qkv + (size_t)plane * a.tokens * ld + head * (32 * NB);
    const float *kb = qb + a.d, *vb = qb + 2 * a.d;
    const float *dob = a.d_o + (size_t)plane * a.tokens * a.d + head * (32 * NB);
    const int query = qt * 32 + j, qc = min(query, a.tokens - 1);
    const float lse = a.lse[(size_t)ph * a.tokens + qc];
    const uint32_t row_word = drop_row_word(a.seed, (uint32_t)ph * a.tokens + qc);
    if (a.threshold)
        for (int i = tid; i < a.ntiles * 32; i += kAtThreads) words[i] = drop_col_word(a.seed, (uint32_t)ph * a.tokens + i);

    float qf[NB][16], dof[NB][16];
    // D_i = dO_i . O_i of this lane's query (the two lane halves hold 16 features of each block); also left in a.dsum for the
    // dK / dV pass, which runs after this one (this used to be a kernel of its own: 18 us of pure traffic per layer)
    float dsum;
    {
        float part = 0.f;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            load_rowfrag(qb + 32 * b, ld, active ? query : 0, a.tokens, h, a.scale2, qf[b]);   // B operands: lane <-> query
            load_rowfrag(dob + 32 * b, a.d, active ? query : 0, a.tokens, h, 1.f, dof[b]);
            float of[16];
            load_rowfrag(a.o + (size_t)plane * a.tokens * a.d + head * (32 * NB) + 32 * b, a.d, active ? query : 0, a.tokens, h, 1.f, of);
#pragma unroll
            for (int e = 0; e < 16; ++e) part = fmaf(dof[b][e], of[e], part);
        }
        dsum = part + __shfl_xor(part, 32);
        if (active && h == 0 && query < a.tokens) a.dsum[(size_t)ph * a.tokens + query] = dsum;
    }
    f32x16 dq[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) dq[b] = zero16();
    StageRegs sr[NB];
    Stager sg[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        sg[b] = make_stager(kb + 32 * b, ld, vb + 32 * b, ld, nullptr, nullptr, a.tokens, tid);
        stage_fetch(sr[b], sg[b], 0);
        stage_store(sr[b], lds[0] + b * kAtBufFloats, tid);
    }
    __syncthreads();
    for (int kt = 0; kt < a.ntiles; ++kt) {
        const float *buf = lds[kt & 1];
        if (kt + 1 < a.ntiles) {
#pragma unroll
            for (int b = 0; b < NB; ++b) stage_fetch(sr[b], sg[b], kt + 1);
        }
        if (active) {
            f32x16 s, dp;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                float kf[16], vf[16];
                lds_rowfrag(buf + b * kAtBufFloats, j, h, kf);                           // A operands: lane <-> key
                lds_rowfrag(buf + b * kAtBufFloats + kAtTileFloats, j, h, vf);
                s = b == 0 ? mma16z(kf, qf[0]) : mma16(kf, qf[b], s);                // [row = key][col = query]
                dp = b == 0 ? mma16z(vf, dof[0]) : mma16(vf, dof[b], dp);
            }
            // Range masks only in the (possibly ragged) last key tile, where exp2(0 - lse) of a padded key could
            // overflow and meet the zero row of K as 0 * inf; query lanes beyond the plane are never stored.
            if (kt == a.ntiles - 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kt * 32 + rowmap(r, h) >= a.tokens) s[r] = -__builtin_inff();
            }
            float ds[16], dpm[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) dpm[r] = dp[r];
            if (a.threshold) {
                uint32_t cw[16];
                lds_rowwords(words + kt * 32, h, cw);
#pragma unroll
                for (int r = 0; r < 16; ++r) dpm[r] = drop_keep(row_word, cw[r], a.threshold) ? dpm[r] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) ds[r] = __builtin_amdgcn_exp2f(s[r] - lse) * fmaf(dpm[r], a.keep_scale, -dsum);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                float kT[16];
                lds_colfrag(buf + b * kAtBufFloats, j, h, kT);                           // A operand: lane <-> feature
                dq[b] = mma16(kT, ds, dq[b]);                                        // [row = feature][col = query]
            }
        }
        if (kt + 1 < a.ntiles) {
#pragma unroll
            for (int b = 0; b < NB; ++b) stage_store(sr[b], lds[(kt + 1) & 1] + b * kAtBufFloats, tid);
        }
        __syncthreads();
    }
    if (!active) return;
#pragma unroll
    for (int b = 0; b < NB; ++b)
        store_transposed(a.out + (size_t)plane * a.tokens * ld + head * (32 * NB) + 32 * b, ld, query, a.tokens, h, dq[b], a.scale);
}
__global__ __launch_bounds__(kAtThreads) __attribute__((amdgpu_waves_per_eu(3, 3))) void attn_bwd_q_kernel(const AttnTrainArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[2][kAtBufFloats];
    extern __shared__ __attribute__((aligned(16))) uint32_t words[];   // [ntiles * 32] column (key) words of the dropout mask
    attn_bwd_q_body<1>(a, lds, words);
}
__global__ __launch_bounds__(kAtThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_bwd_q64_kernel(const AttnTrainArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[2][2 * kAtBufFloats];
    extern __shared__ __attribute__((aligned(16))) uint32_t words[];
    attn_bwd_q_body<2>(a, lds, words);
}

// ---- heads of up to 96 / 128 features (round 6, the general engine): three / four 32-feature blocks per head, the same bodies.  The
// staging buffers no longer fit the 64 KB of static LDS (2 x NB x 9.5 KB), so they lie in the dynamic region in front of the mask
// words; registers are whatever the bodies need (two or one wave per SIMD): covered, not tuned.
template <int NB>
__global__ __launch_bounds__(kAtThreads) void attn_train_fwd_wide_kernel(const AttnTrainArgs a) {
    extern __shared__ __attribute__((aligned(16))) float dyn_lds[];
    attn_train_fwd_body<NB>(a, reinterpret_cast<float (*)[NB * kAtBufFloats]>(dyn_lds), reinterpret_cast<uint32_t *>(dyn_lds + 2 * NB * kAtBufFloats));
}
template <int NB>
__global__ __launch_bounds__(kAtThreads) void attn_bwd_q_wide_kernel(const AttnTrainArgs a) {
    extern __shared__ __attribute__((aligned(16))) float dyn_lds[];
    attn_bwd_q_body<NB>(a, reinterpret_cast<float (*)[NB * kAtBufFloats]>(dyn_lds), reinterpret_cast<uint32_t *>(dyn_lds + 2 * NB * kAtBufFloats));
}
template <int NB>
__global__ __launch_bounds__(kAtThreads) void attn_bwd_kv_wide_kernel(const AttnTrainArgs a) {
    extern __shared__ __attribute__((aligned(16))) float dyn_lds[];
    attn_bwd_kv_body<NB>(a, reinterpret_cast<float (*)[NB * kAtBufFloats]>(dyn_lds), reinterpret_cast<uint32_t *>(dyn_lds + 2 * NB * kAtBufFloats));
}
static size_t wide_lds_bytes(int nb, int ntiles) { return sizeof(float) * 2 * nb * kAtBufFloats + (size_t)ntiles * 32 * sizeof(uint32_t); }

// ---- dQ, dK and dV in ONE pass (round 3): own = key tile, walks the query tiles (staged: Q, dO) ----
// The two-kernel backward recomputes S and dP in both passes: 7 products of 16 MFMAs per (query tile, key tile) pair for 5
// useful ones.  Here a workgroup owns one whole (plane, head): its three waves take the key tiles three at a time (pass p:
// tiles 3p .. 3p+2), keep dK / dV in registers exactly as attn_bwd_kv_kernel does, and ALSO produce the pair's dQ
// contribution dS K: that product contracts over the keys, i.e. over the COLUMNS of the dS accumulator, so the tile goes
// through LDS once (16 writes, 4 wide reads -- a fraction of the 32 MFMAs + exp2 + mask of recomputing it in the other
// orientation).  The three waves' contributions to the same query tile meet in LDS behind the step's barrier and are added
// in a fixed order (wave 0 + 1 + 2) to what the previous passes left in dqkv's q block -- a read-modify-write by the SAME
// thread in every pass, so the result is deterministic; the 1/sqrt(dh) factor goes on with the last pass.
// D_i = dO_i . O_i and the LSE of the whole (plane, head) sit in LDS tables filled by the prologue (no dsum round trip).
// Registers: 3 waves per SIMD leave 168; the wave's own K tile therefore lives in LDS (scaled by log2 e / sqrt(dh), which the
// dQ product takes back out with its final factor) and is read in both operand forms, only V's stays in registers.
// LDS per workgroup at 9 tiles: staging 9.2 KB (single: the step has two barriers anyway) + exchange 13.8 + K tiles 13.8 +
// tables 3.4 = 40.3 KB, four workgroups per CU.
// TOK > 0: the token count (and with it the tile and pass counts) as a compile-time constant -- the default grid's 280, as in
// attn_kernel<32, 280> (k_attn.hip): 327.6 -> 320.4 us, same bits; any other count runs the generic instantiation.
// GROUPS: (plane, head) problems per workgroup, each worked by its own three waves exactly as described above (own staging, own
// exchange blocks, own tables; only the barriers are shared).  GROUPS = 4 is a TWELVE-wave workgroup, one per CU, and exists because
// of where the hardware puts waves (tools/micro/wg_shape.hip, round 4): with registers for three waves per SIMD, four resident
// THREE-wave workgroups per CU sustain 0.64 of the fp32 MFMA roof on a registers-only chain benchmark where four-wave, single-wave
// and twelve-wave workgroups sustain 0.83-0.87 -- the dispatcher does not spread a three-wave workgroup's waves so that every SIMD
// ends up with three (late-starting workgroups and an idle SIMD showed in in-kernel timestamps), and this kernel sat at 106 cycles
// per MFMA whatever was removed from it (LDS operand reads, the vector math, the barriers, all global traffic: each <= 6 %).
// GROUPS = 1 is the old shape, kept for token counts whose tables do not fit four times into the CU's LDS.
constexpr int kAtBwdStatic = (2 + 2 * kAtWaves) * kAtTileFloats;   // floats per group: stage (Q | dO) + exchange blocks + K tiles
template <int TOK = 0, int GROUPS = 1>
__global__ __launch_bounds__(kAtThreads * GROUPS) __attribute__((amdgpu_waves_per_eu(3, 3))) void attn_bwd_kernel(const AttnTrainArgs a_rt) {
    AttnTrainArgs a = a_rt;
    if constexpr (TOK > 0) {
        a.tokens = TOK;
        a.ntiles = (TOK + 31) / 32;
    }
    extern __shared__ __attribute__((aligned(16))) float smem_bwd[];
    const int nrow = a.ntiles * 32;
    const int grp = GROUPS > 1 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x / kAtThreads) : 0;
    float *gbase = smem_bwd + (size_t)grp * (kAtBwdStatic + 3 * nrow);
    float *stage = gbase;                                        // Q tile | dO tile of the step
    float *xch_base = stage + 2 * kAtTileFloats;                 // per wave: dS transposed, then its dQ contribution
    float *ktile_base = xch_base + kAtWaves * kAtTileFloats;     // per wave: its own K tile x scale2
    uint32_t *words = reinterpret_cast<uint32_t *>(ktile_base + kAtWaves * kAtTileFloats);   // [ntiles * 32] row (query) words | lse | D
    // the two float tables behind the mask words, as LDS-address-space float pointers: float stores, float4 loads (reading them
    // back as uint32 vectors and bit-casting the elements made hipcc use element 0 for all four)
    using LdsF = __attribute__((address_space(3))) float;
    using LdsF4 = __attribute__((address_space(3))) f32x4;
    LdsF *lse_t = (LdsF *)(words + nrow), *dsum_t = lse_t + nrow;
    const int tid = (int)threadIdx.x - grp * kAtThreads, wave = tid >> 6, lane = tid & 63, j = lane & 31, h = lane >> 5;   // within the group
    const int ph = blockIdx.x * GROUPS + grp, head = ph % a.heads, plane = ph / a.heads;
    const int ld = 3 * a.d;
    const float *qb = a.qkv + (size_t)plane * a.tokens * ld + head * 32;
    const float *kb = qb + a.d, *vb = qb + 2 * a.d;
    const float *dob = a.d_o + (size_t)plane * a.tokens * a.d + head * 32;
    const float *ob = a.o + (size_t)plane * a.tokens * a.d + head * 32;
    float *dst = a.out + (size_t)plane * a.tokens * ld + head * 32;

    for (int i = tid; i < nrow; i += kAtThreads) {
        words[i] = drop_row_word(a.seed, (uint32_t)ph * a.tokens + min(i, a.tokens - 1));
        lse_t[i] = i < a.tokens ? a.lse[(size_t)ph * a.tokens + i] : 0.f;
    }
    // D: eight lanes per row, one float4 of dO and of O each; six row groups' loads in flight at a time (all four workgroups of a
    // CU run this prologue at the same moment: a load round trip per row group, one after the other, was 20 us of idle CU)
    for (int r0 = 0; r0 < nrow; r0 += 6 * (kAtThreads / 8)) {
        const int q4 = tid & 7;
        f32x4 x[6], y[6];
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int row = min(r0 + u * (kAtThreads / 8) + (tid >> 3), a.tokens - 1);
            x[u] = *reinterpret_cast<const f32x4 *>(dob + (size_t)row * a.d + 4 * q4);
            y[u] = *reinterpret_cast<const f32x4 *>(ob + (size_t)row * a.d + 4 * q4);
        }
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int row = r0 + u * (kAtThreads / 8) + (tid >> 3);
            float part = fmaf(x[u][3], y[u][3], fmaf(x[u][2], y[u][2], fmaf(x[u][1], y[u][1], x[u][0] * y[u][0])));
            part += __shfl_xor(part, 1);
            part += __shfl_xor(part, 2);
            part += __shfl_xor(part, 4);
            if (q4 == 0 && row < nrow) dsum_t[row] = row < a.tokens ? part : 0.f;
        }
    }

    // Staging and reduce share one element map: float4 e = tid (and tid + 192 for wave 0) <-> row e >> 3, quad e & 7 of a 32 x 32
    // tile.  The tile's offset goes into the LANE offset of the buffer access, so the hardware range check zeroes the rows of the
    // ragged last tile on load and drops them on store (a scalar offset is not range-checked).
    const AtSrd q_rs = at_srd(qb, (unsigned)((a.tokens - 1) * ld + 32) * 4u);
    const AtSrd do_rs = at_srd(dob, (unsigned)((a.tokens - 1) * a.d + 32) * 4u);
    const AtSrd dq_rs = at_srd(dst, (unsigned)((a.tokens - 1) * ld + 32) * 4u);
    const unsigned tile_q = 32u * (unsigned)ld * 4u, tile_do = 32u * (unsigned)a.d * 4u;
    const int e1 = tid + kAtThreads;
    // (element e1 exists for wave 0 only; the other waves give it an out-of-range offset -- an unconditional load of zeros --
    // rather than a branch: a conditionally written register is live around the whole loop)
    const unsigned vq0 = (unsigned)((tid >> 3) * ld + 4 * (tid & 7)) * 4u, vd0 = (unsigned)((tid >> 3) * a.d + 4 * (tid & 7)) * 4u;
    const unsigned vq1 = wave == 0 ? (unsigned)((e1 >> 3) * ld + 4 * (e1 & 7)) * 4u : kAtOutOfRange;
    const unsigned vd1 = wave == 0 ? (unsigned)((e1 >> 3) * a.d + 4 * (e1 & 7)) * 4u : kAtOutOfRange;
    const int l0 = (tid >> 3) * kAtLd + 4 * (tid & 7), l1 = (e1 >> 3) * kAtLd + 4 * (e1 & 7);   // the element's place in an LDS tile
    f32x4 sq0, sq1, sd0, sd1;
    auto stage_load = [&](int tile) {
        sq0 = at_ld4(q_rs, vq0 + (unsigned)tile * tile_q, 0);
        sd0 = at_ld4(do_rs, vd0 + (unsigned)tile * tile_do, 0);
        sq1 = at_ld4(q_rs, vq1 + (unsigned)tile * tile_q, 0);
        sd1 = at_ld4(do_rs, vd1 + (unsigned)tile * tile_do, 0);
    };
    auto stage_put = [&]() {
        *reinterpret_cast<f32x4 *>(stage + l0) = sq0;
        *reinterpret_cast<f32x4 *>(stage + kAtTileFloats + l0) = sd0;
        if (wave == 0) {
            *reinterpret_cast<f32x4 *>(stage + l1) = sq1;
            *reinterpret_cast<f32x4 *>(stage + kAtTileFloats + l1) = sd1;
        }
    };
    const int npass = (a.ntiles + kAtWaves - 1) / kAtWaves;
    float *T = xch_base + wave * kAtTileFloats;
    const float *KT = ktile_base + wave * kAtTileFloats;
    for (int pass = 0; pass < npass; ++pass) {
        const int kt = pass * kAtWaves + wave;
        const bool active = kt < a.ntiles;
        const int nact = min(kAtWaves, a.ntiles - pass * kAtWaves);
        const int key = kt * 32 + j;
        const bool ragged = active && kt * 32 + 32 > a.tokens;
        const float dq_mul = pass == npass - 1 ? a.dq_scale : 1.f;
        // pass prologue: every global read first (first query tile, own V fragment, own K tile), then the LDS writes
        stage_load(0);
        float vf[16];
        load_rowfrag(vb, ld, active ? key : 0, a.tokens, h, 1.f, vf);   // B operand: lane <-> key
        {   // own K tile x scale2 -> LDS (wave-private; rows beyond the plane are zero)
            f32x4 kv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = lane + 64 * u, row = e >> 3, q4 = e & 7, tok = (active ? kt : 0) * 32 + row;
                kv[u] = *reinterpret_cast<const f32x4 *>(kb + (size_t)min(tok, a.tokens - 1) * ld + 4 * q4);
                if (tok >= a.tokens) kv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = lane + 64 * u, row = e >> 3, q4 = e & 7;
                *reinterpret_cast<f32x4 *>(ktile_base + wave * kAtTileFloats + row * kAtLd + 4 * q4) = kv[u] * a.scale2;
            }
        }
        f32x16 dv = zero16(), dk = zero16();
        const uint32_t col_word = drop_col_word(a.seed, (uint32_t)ph * a.tokens + min(key, a.tokens - 1));
        stage_put();
        __syncthreads();
        for (int qt = 0; qt < a.ntiles; ++qt) {
            // The step's global reads -- what the earlier passes left of this query tile's dQ, and the next tile's Q / dO --
            // are issued in the middle of the products, two MFMA chains and the barrier before their first use: all four
            // workgroups of a CU run in lock-step, so a latency a wave does not cover itself is covered by nobody.
            const unsigned t_off = (unsigned)qt * tile_q;
            f32x4 old0 = f32x4{0.f, 0.f, 0.f, 0.f}, old1 = old0;
            auto issue_loads = [&]() {
                if (pass > 0) {
                    old0 = at_ld4(dq_rs, vq0 + t_off, 0);
                    old1 = at_ld4(dq_rs, vq1 + t_off, 0);
                }
                if (qt + 1 < a.ntiles) stage_load(qt + 1);
            };
            if (!active) issue_loads();
            if (active) {
                f32x16 s, dp;
                {
                    float qf[16], kf[16];
                    lds_rowfrag(stage, j, h, qf);                              // A operand: lane <-> query
                    lds_rowfrag(KT, j, h, kf);                                 // B operand: lane <-> key
                    s = mma16z(qf, kf);                                        // [row = query][col = key]
                }
                __builtin_amdgcn_sched_barrier(0);
                {
                    float dof[16];
                    lds_rowfrag(stage + kAtTileFloats, j, h, dof);
                    dp = mma16z(dof, vf);
                }
                __builtin_amdgcn_sched_barrier(0);
                float pd[16], ds[16];
                // a quarter of the tile at a time: the three table reads of a quarter are 12 registers, not 48
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 ls = *(const LdsF4 *)(lse_t + qt * 32 + 8 * q + 4 * h);
                    const f32x4 dsm = *(const LdsF4 *)(dsum_t + qt * 32 + 8 * q + 4 * h);
                    const u32x4 rw = *reinterpret_cast<const u32x4 *>(words + qt * 32 + 8 * q + 4 * h);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int r = 4 * q + c;
                        const float p = __builtin_amdgcn_exp2f(s[r] - ls[c]);
                        const float pm = (!a.threshold || drop_keep(rw[c], col_word, a.threshold)) ? p : 0.f;
                        pd[r] = pm;
                        ds[r] = fmaf(pm * dp[r], a.keep_scale, -(p * dsm[c]));   // P o (dP o mask/(1-p) - D)
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (ragged && key >= a.tokens) {   // key lanes beyond the plane: never stored as dK / dV, but dS K sums over them
#pragma unroll
                    for (int r = 0; r < 16; ++r) ds[r] = 0.f;
                }
                __builtin_amdgcn_sched_barrier(0);
                {
                    float doT[16];
                    lds_colfrag(stage + kAtTileFloats, j, h, doT);             // A operands: lane <-> feature
                    dv = mma16(doT, pd, dv);                                   // [row = feature][col = key]
                }
                __builtin_amdgcn_sched_barrier(0);
                issue_loads();
                __builtin_amdgcn_sched_barrier(0);
                {
                    float qT[16];
                    lds_colfrag(stage, j, h, qT);
                    dk = mma16(qT, ds, dk);
                }
                __builtin_amdgcn_sched_barrier(0);
                // dS -> LDS [query][key] -> B operand with lane <-> query (wave-private region: LDS ops of a wave complete in order)
#pragma unroll
                for (int r = 0; r < 16; ++r) T[rowmap(r, h) * kAtLd + j] = ds[r];
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // orders the wave's LDS writes before its reads of other lanes' elements
                __builtin_amdgcn_wave_barrier();
                f32x16 dq;
                {
                    float dsT[16], kT[16];
                    lds_rowfrag(T, j, h, dsT);
                    lds_colfrag(KT, j, h, kT);                                 // A operand: lane <-> feature, slot <-> key
                    dq = mma16z(kT, dsT);                                      // [row = feature][col = query]
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // orders the wave's LDS writes before its reads of other lanes' elements
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<f32x4 *>(T + j * kAtLd + 8 * q + 4 * h) = f32x4{dq[4 * q], dq[4 * q + 1], dq[4 * q + 2], dq[4 * q + 3]};
            }
            __syncthreads();
            // the waves' contributions to query tile qt, added in wave order to what the earlier passes left
            {
                f32x4 v0 = *reinterpret_cast<const f32x4 *>(xch_base + l0);
                if (nact > 1) v0 += *reinterpret_cast<const f32x4 *>(xch_base + kAtTileFloats + l0);
                if (nact > 2) v0 += *reinterpret_cast<const f32x4 *>(xch_base + 2 * kAtTileFloats + l0);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, (v0 + old0) * dq_mul), dq_rs, vq0 + t_off, 0, 0);
                if (wave == 0) {
                    f32x4 v1 = *reinterpret_cast<const f32x4 *>(xch_base + l1);
                    if (nact > 1) v1 += *reinterpret_cast<const f32x4 *>(xch_base + kAtTileFloats + l1);
                    if (nact > 2) v1 += *reinterpret_cast<const f32x4 *>(xch_base + 2 * kAtTileFloats + l1);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, (v1 + old1) * dq_mul), dq_rs, vq1 + t_off, 0, 0);
                }
            }
            if (qt + 1 < a.ntiles) stage_put();
            __syncthreads();
        }
        if (active) {
            store_transposed(dst + a.d, ld, key, a.tokens, h, dk, a.scale);
            store_transposed(dst + 2 * a.d, ld, key, a.tokens, h, dv, a.keep_scale);
        }
    }
}

// ---- head dimensions other than 32 and 64 (round 5): the kernels above contract 32 (or, in their two-block form, 64) features per
// head.  A head of 8 / 16 / 24 features is run as a 32-feature head, one of 40 / 48 as a 64-feature head, whose upper features are
// zero: q | k | v are re-laid with every head padded (the zeros add nothing to Q K^T, the padded part of O / dV / dQ / dK comes out
// zero and is dropped), the softmax scale stays 1 / sqrt(true head dim).  More matrix work than a tuned kernel would do and four
// re-lay passes per layer -- covered, not tuned -- but every gradient comes from this library's kernels
// (reference blocks/encoders.py:44-51 builds whatever num_head the YAML says; trainer.py:195-233 trains it).
// pad: dst [rows][nseg][heads][hp] <- src [rows][nseg][heads][hd]; unpad: the reverse.  One 16-byte piece per thread.
static int padded_head(int hd) { return (hd + 31) / 32 * 32; }   // 32 | 64 | 96 | 128 features
// any head dim (round 6: not a multiple of 4 -- model_dim 120 with 8 heads): one float per thread
__global__ __launch_bounds__(256) void pad_heads1_kernel(const float *__restrict__ src, float *__restrict__ dst, size_t n, int hd, int hp) {
    const size_t v = (size_t)blockIdx.x * 256 + threadIdx.x;      // element of dst: (row, seg * heads + head, j in 0 .. hp - 1)
    if (v >= n) return;
    const size_t rh = v / hp;
    const int j = (int)(v - rh * hp);
    dst[v] = j < hd ? src[rh * hd + j] : 0.f;
}
__global__ __launch_bounds__(256) void unpad_heads1_kernel(const float *__restrict__ src, float *__restrict__ dst, size_t n, int hd, int hp) {
    const size_t v = (size_t)blockIdx.x * 256 + threadIdx.x;      // element of dst
    if (v >= n) return;
    const size_t rh = v / hd;
    dst[v] = src[rh * hp + (v - rh * hd)];
}
__global__ __launch_bounds__(256) void pad_heads_kernel(const float *__restrict__ src, float *__restrict__ dst, size_t pieces, int hd4, int hp4) {
    const size_t v = (size_t)blockIdx.x * 256 + threadIdx.x;      // piece of dst: (row, seg * heads + head, j4 in 0 .. hp4 - 1)
    if (v >= pieces) return;
    const size_t rh = v / hp4;                                     // row * nseg * heads + (seg, head)
    const int j4 = (int)(v - rh * hp4);
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    if (j4 < hd4) o = *reinterpret_cast<const f32x4 *>(src + (rh * hd4 + j4) * 4);
    *reinterpret_cast<f32x4 *>(dst + v * 4) = o;
}
__global__ __launch_bounds__(256) void unpad_heads_kernel(const float *__restrict__ src, float *__restrict__ dst, size_t pieces, int hd4, int hp4) {
    const size_t v = (size_t)blockIdx.x * 256 + threadIdx.x;      // piece of dst: (row * nseg * heads + (seg, head), j4 in 0 .. hd4 - 1)
    if (v >= pieces) return;
    const size_t rh = v / hd4;
    *reinterpret_cast<f32x4 *>(dst + v * 4) = *reinterpret_cast<const f32x4 *>(src + (rh * hp4 + (v - rh * hd4)) * 4);
}
static hipError_t pad_heads(const float *src, float *dst, size_t rows, int nseg, int heads, int hd, hipStream_t st) {
    const int hp = padded_head(hd);
    if (hd % 4 != 0) {
        const size_t n = rows * nseg * heads * hp;
        hipLaunchKernelGGL(pad_heads1_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, dst, n, hd, hp);
        return hipGetLastError();
    }
    const size_t pieces = rows * nseg * heads * (hp / 4);
    hipLaunchKernelGGL(pad_heads_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, st, src, dst, pieces, hd / 4, hp / 4);
    return hipGetLastError();
}
static hipError_t unpad_heads(const float *src, float *dst, size_t rows, int nseg, int heads, int hd, hipStream_t st) {
    const int hp = padded_head(hd);
    if (hd % 4 != 0) {
        const size_t n = rows * nseg * heads * hd;
        hipLaunchKernelGGL(unpad_heads1_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, dst, n, hd, hp);
        return hipGetLastError();
    }
    const size_t pieces = rows * nseg * heads * (hd / 4);
    hipLaunchKernelGGL(unpad_heads_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, st, src, dst, pieces, hd / 4, hp / 4);
    return hipGetLastError();
}
size_t attn_train_pad_floats(const aft_config &c, size_t rows) {   // scratch the padded-head path needs: qkv, o, d_o, dqkv padded
    if (c.num_head <= 0) return 0;
    const int hd = c.model_dim / c.num_head, hp = padded_head(hd);
    return hp != hd ? rows * (size_t)c.num_head * hp * 8 : 0;
}

static AttnTrainArgs make_args(const aft_config &c, int planes, int tokens, float dropout_p, uint32_t seed) {
    AttnTrainArgs a{};
    a.planes = planes; a.tokens = tokens; a.heads = c.num_head; a.d = c.model_dim;
    a.d = padded_head(c.model_dim / c.num_head) * c.num_head;     // padded heads (the launchers re-lay the operands): > model_dim
    a.ntiles = (tokens + 31) / 32;
    const float inv = 1.f / sqrtf((float)(c.model_dim / c.num_head));
    a.scale = inv;
    a.scale2 = inv * 1.4426950408889634f;
    a.dq_scale = (float)(1.0 / 1.4426950408889634);
    a.keep_scale = dropout_p > 0.f ? 1.f / (1.f - dropout_p) : 1.f;
    a.threshold = dropout_p > 0.f ? (uint32_t)((double)dropout_p * 4294967296.0) : 0u;
    a.seed = seed;
    return a;
}

hipError_t launch_attn_train_fwd(const aft_config &c, const float *qkv, float *o, float *lse, int planes, int tokens,
                                 float dropout_p, uint32_t seed, hipStream_t st, float *pad) {
    AttnTrainArgs a = make_args(c, planes, tokens, dropout_p, seed);
    a.qkv = qkv; a.out = o; a.lse = lse;
    const int hd = c.model_dim / c.num_head, hp = padded_head(hd);
    const bool padded = hp != hd;
    const size_t rows = (size_t)planes * tokens;
    if (padded) {
        if (pad == nullptr) return hipErrorInvalidValue;
        float *qkv_p = pad, *o_p = pad + rows * 3 * a.d;
        hipError_t e = pad_heads(qkv, qkv_p, rows, 3, c.num_head, hd, st);
        if (e != hipSuccess) return e;
        a.qkv = qkv_p; a.out = o_p;
    }
    const int wgs = planes * a.heads * ((a.ntiles + kAtWaves - 1) / kAtWaves);
    if (hp == 96 || hp == 128) {   // three / four blocks per head (round 6; the general engine)
        static PerDeviceOnce attr3, attr4;
        const size_t lds = wide_lds_bytes(hp / 32, a.ntiles);
        if (lds > 160 * 1024) return hipErrorInvalidValue;
        hipError_t ea = hp == 96 ? ensure_dynamic_lds(attr3, reinterpret_cast<const void *>(attn_train_fwd_wide_kernel<3>), 160 * 1024)
                                 : ensure_dynamic_lds(attr4, reinterpret_cast<const void *>(attn_train_fwd_wide_kernel<4>), 160 * 1024);
        if (ea != hipSuccess) return ea;
        if (hp == 96) hipLaunchKernelGGL(attn_train_fwd_wide_kernel<3>, dim3(wgs), dim3(kAtThreads), lds, st, a);
        else hipLaunchKernelGGL(attn_train_fwd_wide_kernel<4>, dim3(wgs), dim3(kAtThreads), lds, st, a);
    } else if (hp == 64)      // a head = two 32-feature blocks (round 5; covered, not tuned)
        hipLaunchKernelGGL(attn_train_fwd64_kernel, dim3(wgs), dim3(kAtThreads), (size_t)a.ntiles * 32 * sizeof(uint32_t), st, a);
    else
        hipLaunchKernelGGL(attn_train_fwd_kernel, dim3(wgs), dim3(kAtThreads), (size_t)a.ntiles * 32 * sizeof(uint32_t), st, a);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && padded) e = unpad_heads(a.out, o, rows, 1, c.num_head, hd, st);
    return e;
}

static hipError_t launch_attn_train_bwd_padded(const aft_config &c, AttnTrainArgs a, const float *qkv, const float *o, const float *d_o,
                                               float *dqkv, int planes, int tokens, hipStream_t st);

hipError_t launch_attn_train_bwd(const aft_config &c, const float *qkv, const float *o, const float *d_o, const float *lse,
                                 float *dsum, float *dqkv, int planes, int tokens, float dropout_p, uint32_t seed,
                                 hipStream_t st, float *pad) {
    AttnTrainArgs a = make_args(c, planes, tokens, dropout_p, seed);
    a.qkv = qkv; a.o = o; a.d_o = d_o; a.lse = const_cast<float *>(lse); a.dsum = dsum; a.out = dqkv;
    const int hd = c.model_dim / c.num_head;
    if (padded_head(hd) != hd) {   // padded heads: re-lay the three operands, run on the padded image, drop the padding of dqkv
        if (pad == nullptr) return hipErrorInvalidValue;
        const size_t rows = (size_t)planes * tokens;
        float *qkv_p = pad, *o_p = qkv_p + rows * 3 * a.d, *do_p = o_p + rows * a.d, *dqkv_p = do_p + rows * a.d;
        hipError_t e = pad_heads(qkv, qkv_p, rows, 3, c.num_head, hd, st);
        if (e == hipSuccess) e = pad_heads(o, o_p, rows, 1, c.num_head, hd, st);
        if (e == hipSuccess) e = pad_heads(d_o, do_p, rows, 1, c.num_head, hd, st);
        if (e != hipSuccess) return e;
        aft_config cp = c;
        cp.model_dim = a.d;                  // 32 or 64 features per head: the plain path below, scales already set from the true head dim
        AttnTrainArgs keep = a;
        e = launch_attn_train_bwd_padded(cp, keep, qkv_p, o_p, do_p, dqkv_p, planes, tokens, st);
        if (e == hipSuccess) e = unpad_heads(dqkv_p, dqkv, rows, 3, c.num_head, hd, st);
        return e;
    }
    return launch_attn_train_bwd_padded(c, a, qkv, o, d_o, dqkv, planes, tokens, st);
}

static hipError_t launch_attn_train_bwd_padded(const aft_config &c, AttnTrainArgs a, const float *qkv, const float *o, const float *d_o,
                                               float *dqkv, int planes, int tokens, hipStream_t st) {
    a.qkv = qkv; a.o = o; a.d_o = d_o; a.out = dqkv;
    if (const int hp = c.model_dim / c.num_head; hp == 96 || hp == 128) {   // three / four blocks per head: the two-pass form (round 6)
        static PerDeviceOnce attr[4];
        const size_t lds = wide_lds_bytes(hp / 32, a.ntiles);
        if (lds > 160 * 1024) return hipErrorInvalidValue;
        const void *fq = hp == 96 ? reinterpret_cast<const void *>(attn_bwd_q_wide_kernel<3>) : reinterpret_cast<const void *>(attn_bwd_q_wide_kernel<4>);
        const void *fk = hp == 96 ? reinterpret_cast<const void *>(attn_bwd_kv_wide_kernel<3>) : reinterpret_cast<const void *>(attn_bwd_kv_wide_kernel<4>);
        hipError_t ea = ensure_dynamic_lds(attr[hp == 96 ? 0 : 1], fq, 160 * 1024);
        if (ea == hipSuccess) ea = ensure_dynamic_lds(attr[hp == 96 ? 2 : 3], fk, 160 * 1024);
        if (ea != hipSuccess) return ea;
        const int wgsw = planes * a.heads * ((a.ntiles + kAtWaves - 1) / kAtWaves);
        if (hp == 96) {
            hipLaunchKernelGGL(attn_bwd_q_wide_kernel<3>, dim3(wgsw), dim3(kAtThreads), lds, st, a);     // also writes D_i = dO_i . O_i
            hipLaunchKernelGGL(attn_bwd_kv_wide_kernel<3>, dim3(wgsw), dim3(kAtThreads), lds, st, a);    // reads it
        } else {
            hipLaunchKernelGGL(attn_bwd_q_wide_kernel<4>, dim3(wgsw), dim3(kAtThreads), lds, st, a);
            hipLaunchKernelGGL(attn_bwd_kv_wide_kernel<4>, dim3(wgsw), dim3(kAtThreads), lds, st, a);
        }
        return hipGetLastError();
    }
    if (c.model_dim / c.num_head == 64) {   // head dim 64: the two-pass form with two 32-feature blocks per head (round 5)
        const int wgs64 = planes * a.heads * ((a.ntiles + kAtWaves - 1) / kAtWaves);
        const size_t wb = (size_t)a.ntiles * 32 * sizeof(uint32_t);
        hipLaunchKernelGGL(attn_bwd_q64_kernel, dim3(wgs64), dim3(kAtThreads), wb, st, a);     // also writes D_i = dO_i . O_i
        hipLaunchKernelGGL(attn_bwd_kv64_kernel, dim3(wgs64), dim3(kAtThreads), wb, st, a);    // reads it
        return hipGetLastError();
    }
    const int wgs = planes * a.heads * ((a.ntiles + kAtWaves - 1) / kAtWaves);
    const size_t words_bytes = (size_t)a.ntiles * 32 * sizeof(uint32_t);
    // one pass (attn_bwd_kernel) unless its three LDS tables do not fit beside the static staging, or the two-pass form is asked for (A/B)
    const bool two_pass = switch_on("AFT_TRAIN_ATTN_BWD_SPLIT");   // read per call: tools/debug/attn_bwd_check.py flips it
    const size_t group_lds = sizeof(float) * kAtBwdStatic + 3 * words_bytes;
    if (!two_pass && group_lds <= 64 * 1024) {
        // four problems per workgroup (twelve waves, one workgroup per CU) when their LDS fits four times; else the three-wave shape
        // Problems per workgroup: 4 (twelve waves, ONE workgroup per CU, all in lock-step) is the shape of fact 10 and wins exactly when
        // the problems fill whole rounds of it -- 128 frames: 1 024 problems = 256 twelve-wave workgroups (6.94 ms per step against 7.24
        // with three-wave workgroups), 256 frames likewise.  Anywhere else the hardware's dispatcher does better with three-wave
        // workgroups (measured, ms per training step, three- vs twelve-wave: 32 frames 2.93 vs 3.80, 64 frames -- the reference's
        // default batch -- 4.30 vs 4.81: its 128 twelve-wave workgroups left half the CUs idle and the kernel took as long as at 128
        // frames; 96: 5.71 vs 5.83, 160: 8.97 vs 9.65, 192: 10.33 vs 10.76).  (round 5)
        // Round 6, measured and NOT adopted: TWO problems per workgroup (six waves) when the problems are exactly two per CU -- 64 frames
        // of the default model, the reference's default batch: 512 problems = 256 six-wave workgroups, one per CU, instead of 512
        // three-wave ones -- 4.18-4.20 ms per 64-frame training step against 4.14-4.16 (same box, interleaved): a six-wave workgroup's
        // waves are placed as unevenly as a three-wave one's (fact 10).  The instantiation stays behind AFT_ATTN_BWD_GROUPS=2 (A/B, tests).
        const int problems = planes * a.heads, cus = current_device_cus();
        int groups = (4 * group_lds <= 160 * 1024 && problems % (4 * cus) == 0) ? 4 : 1;
        if (switch_on("AFT_ATTN_BWD_3WAVE")) groups = 1;
        if (switch_on("AFT_ATTN_BWD_GROUPS")) {     // A/B: force 1, 2 or 4 where the shape allows it
            const int want = switch_int("AFT_ATTN_BWD_GROUPS", 0);
            if ((want == 1 || want == 2 || want == 4) && problems % want == 0 && (size_t)want * group_lds <= 160 * 1024) groups = want;
        }
        const bool tok280 = tokens == 280 && !switch_on("AFT_ATTN_GENERIC");
        const int gi = groups == 4 ? 2 : groups == 2 ? 1 : 0;
        const void *fns[3][2] = {{reinterpret_cast<const void *>(attn_bwd_kernel<0, 1>), reinterpret_cast<const void *>(attn_bwd_kernel<280, 1>)},
                                 {reinterpret_cast<const void *>(attn_bwd_kernel<0, 2>), reinterpret_cast<const void *>(attn_bwd_kernel<280, 2>)},
                                 {reinterpret_cast<const void *>(attn_bwd_kernel<0, 4>), reinterpret_cast<const void *>(attn_bwd_kernel<280, 4>)}};
        static PerDeviceOnce lds_attr[3][2];
        hipError_t ea = ensure_dynamic_lds(lds_attr[gi][tok280 ? 1 : 0], fns[gi][tok280 ? 1 : 0], groups == 1 ? 64 * 1024 : 160 * 1024);
        if (ea != hipSuccess) return ea;
        const dim3 grid(problems / groups), block(kAtThreads * groups);
        const size_t lds = group_lds * groups;
        if (groups == 4 && tok280) hipLaunchKernelGGL((attn_bwd_kernel<280, 4>), grid, block, lds, st, a);
        else if (groups == 4) hipLaunchKernelGGL((attn_bwd_kernel<0, 4>), grid, block, lds, st, a);
        else if (groups == 2 && tok280) hipLaunchKernelGGL((attn_bwd_kernel<280, 2>), grid, block, lds, st, a);
        else if (groups == 2) hipLaunchKernelGGL((attn_bwd_kernel<0, 2>), grid, block, lds, st, a);
        else if (tok280) hipLaunchKernelGGL((attn_bwd_kernel<280, 1>), grid, block, lds, st, a);
        else hipLaunchKernelGGL((attn_bwd_kernel<0, 1>), grid, block, lds, st, a);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(attn_bwd_q_kernel, dim3(wgs), dim3(kAtThreads), words_bytes, st, a);    // also writes D_i = dO_i . O_i
    hipLaunchKernelGGL(attn_bwd_kv_kernel, dim3(wgs), dim3(kAtThreads), words_bytes, st, a);   // reads it
    return hipGetLastError();
}

}  // namespace aft
