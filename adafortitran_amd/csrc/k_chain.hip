// k_chain.hip -- the row-local part of an encoder layer as ONE gfx950 kernel.
//
// Reference semantics (nn.TransformerEncoderLayer, post-LN, eval; constructed at
// reference src/models/blocks/encoders.py:44-55, dim_feedforward = 2*model_dim :47):
//     x1 = LN1(x + attn @ Wo^T + bo)
//     x2 = LN2(x1 + act(x1 @ W1^T + b1) @ W2^T + b2)
// and, fused behind it, the NEXT layer's packed in-projection (in_proj_weight [3d,d] =
// [Wq;Wk;Wv]):   q,k,v = split(x2 @ Wqkv^T + bqkv)  written per head for k_attn.hip.
// Everything between two attention calls is row-local, so one workgroup owns a tile of
// ROWS token rows through all four GEMMs; activations never leave LDS in between.
//
// MI355X mapping
//   * v_mfma_f32_32x32x2_f32 (exact fp32, 64 cyc/SIMD, = fp32 vector peak 157 TF): fp32 parity
//     is the contract, there is no xf32 on gfx950.
//   * 8 waves / workgroup (2 per SIMD).  Wave w owns row tile  w / CG  and column group
//     w % CG; A operand (activations) comes from LDS with ds_read_b128 (row stride K+4 floats
//     -> conflict-free), B operand (weights, torch [out,in] layout, K contiguous) streams
//     straight from L2 into registers with global_load_dwordx4, double-buffered per 32-deep
//     k-block: every weight byte is read once per workgroup, nothing is re-staged.
//   * k pairing: lane (r, h) feeds k = 8s + 4h + j to MFMA (s, j) for both operands, so one
//     16-byte load serves four MFMAs.
//   * LayerNorm: the pre-LN tile goes through LDS once; each wave then owns ROWS/8 rows, a
//     row is D/64 values per lane, mean/var by wave-wide butterfly -- no atomics, no HBM.
//   * HBM traffic per row: read attn + x (2*D*4 B), write x + q,k,v (4*D*4 B); weights
//     (512 KB/layer at d=128) stay L2-resident.
#include "aft_internal.h"

namespace aft {

template <int D, int RT>
struct ChainShape {
    static constexpr int ROWS = 32 * RT;
    static constexpr int WAVES = 8;
    static constexpr int CG = WAVES / RT;       // column groups
    static constexpr int LDA = D + 4;           // padded row strides (floats)
    static constexpr int LDH = 2 * D + 4;
    static constexpr int NT_D = (D / 32) / CG;      // column tiles per wave for N = D
    static constexpr int NT_FF = (2 * D / 32) / CG; // N = 2D
    static constexpr int NT_QKV = (3 * D / 32) / CG;
    static constexpr size_t LDS_BYTES = sizeof(float) * (size_t)(2 * ROWS * LDA + ROWS * LDH);
    static_assert(NT_D >= 1 && (D / 32) % CG == 0, "column split must be whole tiles");
};

struct ChainArgs {
    // MLP part (may be disabled)
    const float *attn;  // [rows, D]
    float *x;           // [rows, D] residual in, layer output out
    const float *wo, *bo, *w1, *b1, *w2, *b2, *g1, *be1, *g2, *be2;
    // QKV part (may be disabled)
    const float *wqkv, *bqkv;
    float *q, *k, *vt;
    int rows, tokens, tokpad, heads, activation, do_mlp, do_qkv;
};

__device__ __forceinline__ int acc_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

// acc[t] += A_tile[32 x K] * W[col0 + 32t .. +31][K]^T for this wave.
//   a_lane = &A[(tile row r)][4h]   (LDS),  w_lane = &W[col0 + r][4h]   (global)
template <int K, int NT>
__device__ __forceinline__ void wave_gemm(const float *a_lane, const float *__restrict__ w_lane,
                                          f32x16 (&acc)[NT]) {
    constexpr int NKB = K / 32;
    f32x4 bcur[NT][4], bnxt[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int s = 0; s < 4; ++s) bcur[t][s] = *reinterpret_cast<const f32x4 *>(w_lane + (size_t)t * 32 * K + s * 8);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        if (kb + 1 < NKB) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    bnxt[t][s] = *reinterpret_cast<const f32x4 *>(w_lane + (size_t)t * 32 * K + (kb + 1) * 32 + s * 8);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const f32x4 a = *reinterpret_cast<const f32x4 *>(a_lane + kb * 32 + s * 8);
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], bcur[t][s][j], acc[t], 0, 0, 0);
        }
        if (kb + 1 < NKB) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int s = 0; s < 4; ++s) bcur[t][s] = bnxt[t][s];
        }
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__device__ __forceinline__ float activate(float v, int activation) {
    // exact erf GELU (F.gelu default) or ReLU -- schemas.py:128-131 allows both
    return activation == AFT_ACT_GELU ? 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)) : fmaxf(v, 0.0f);
}

// LayerNorm(eps=1e-5, biased variance) of ROWS rows held in `src` (stride lds), result to `dst`
// (stride ldd) and optionally to global `gout` rows (coalesced 4*D bytes per row).
template <int D, int ROWS>
__device__ __forceinline__ void layernorm_tile(const float *src, int lds, float *dst, int ldd,
                                               const float *__restrict__ gamma, const float *__restrict__ beta,
                                               float *gout, long row0, int rows, int wave, int lane) {
    constexpr int PER = D / 64;
    float g[PER], b[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        g[i] = gamma[lane + 64 * i];
        b[i] = beta[lane + 64 * i];
    }
#pragma unroll 4
    for (int rr = 0; rr < ROWS / 8; ++rr) {
        const int row = wave * (ROWS / 8) + rr;
        float v[PER], s = 0.f;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            v[i] = src[row * lds + lane + 64 * i];
            s += v[i];
        }
        const float mean = wave_sum(s) * (1.0f / D);
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            v[i] -= mean;
            sq += v[i] * v[i];
        }
        const float rstd = rsqrtf(wave_sum(sq) * (1.0f / D) + 1e-5f);
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const float y = v[i] * rstd * g[i] + b[i];
            dst[row * ldd + lane + 64 * i] = y;
            if (gout != nullptr && row0 + row < rows) gout[(row0 + row) * (long)D + lane + 64 * i] = y;
        }
    }
}

template <int D, int RT>
__global__ __launch_bounds__(512) void chain_kernel(const ChainArgs a) {
    using S = ChainShape<D, RT>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *bufA = smem;                       // attn tile, later pre-LN2 scratch
    float *bufB = bufA + S::ROWS * S::LDA;    // x1 (LN1 out), later x2 (LN2 out) = QKV operand
    float *bufH = bufB + S::ROWS * S::LDA;    // pre-LN1 scratch, then FFN hidden [ROWS][2D]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int rt = wave / S::CG, cg = wave % S::CG;
    const long row0 = (long)blockIdx.x * S::ROWS;

    if (a.do_mlp) {
        // ---- stage the attention-output tile (A operand of the out-projection) ----
        for (int i = tid; i < S::ROWS * (D / 4); i += 512) {
            const int row = i / (D / 4), c4 = i % (D / 4);
            const long grow = min(row0 + row, (long)a.rows - 1);
            *reinterpret_cast<f32x4 *>(bufA + row * S::LDA + c4 * 4) =
                *reinterpret_cast<const f32x4 *>(a.attn + grow * D + c4 * 4);
        }
        __syncthreads();
        // ---- out-projection + bias + residual -> pre-LN1 scratch (bufH, stride LDA) ----
        {
            f32x16 acc[S::NT_D];
#pragma unroll
            for (int t = 0; t < S::NT_D; ++t) acc[t] = f32x16{0};
            const int col0 = cg * S::NT_D * 32;
            wave_gemm<D, S::NT_D>(bufA + (rt * 32 + r) * S::LDA + 4 * h, a.wo + (size_t)(col0 + r) * D + 4 * h, acc);
#pragma unroll
            for (int t = 0; t < S::NT_D; ++t) {
                const int col = col0 + t * 32 + r;
                const float bias = a.bo[col];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = rt * 32 + acc_row(e, h);
                    const long grow = row0 + row;
                    const float res = grow < a.rows ? a.x[grow * D + col] : 0.f;
                    bufH[row * S::LDA + col] = acc[t][e] + bias + res;
                }
            }
        }
        __syncthreads();
        layernorm_tile<D, S::ROWS>(bufH, S::LDA, bufB, S::LDA, a.g1, a.be1, nullptr, row0, a.rows, wave, lane);
        __syncthreads();
        // ---- FFN up-projection + activation -> hidden tile in LDS ----
        {
            f32x16 acc[S::NT_FF];
#pragma unroll
            for (int t = 0; t < S::NT_FF; ++t) acc[t] = f32x16{0};
            const int col0 = cg * S::NT_FF * 32;
            wave_gemm<D, S::NT_FF>(bufB + (rt * 32 + r) * S::LDA + 4 * h, a.w1 + (size_t)(col0 + r) * D + 4 * h, acc);
#pragma unroll
            for (int t = 0; t < S::NT_FF; ++t) {
                const int col = col0 + t * 32 + r;
                const float bias = a.b1[col];
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    bufH[(rt * 32 + acc_row(e, h)) * S::LDH + col] = activate(acc[t][e] + bias, a.activation);
            }
        }
        __syncthreads();
        // ---- FFN down-projection + bias + residual(x1) -> pre-LN2 scratch (bufA) ----
        {
            f32x16 acc[S::NT_D];
#pragma unroll
            for (int t = 0; t < S::NT_D; ++t) acc[t] = f32x16{0};
            const int col0 = cg * S::NT_D * 32;
            wave_gemm<2 * D, S::NT_D>(bufH + (rt * 32 + r) * S::LDH + 4 * h, a.w2 + (size_t)(col0 + r) * (2 * D) + 4 * h, acc);
#pragma unroll
            for (int t = 0; t < S::NT_D; ++t) {
                const int col = col0 + t * 32 + r;
                const float bias = a.b2[col];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = rt * 32 + acc_row(e, h);
                    bufA[row * S::LDA + col] = acc[t][e] + bias + bufB[row * S::LDA + col];
                }
            }
        }
        __syncthreads();
        layernorm_tile<D, S::ROWS>(bufA, S::LDA, bufB, S::LDA, a.g2, a.be2, a.x, row0, a.rows, wave, lane);
        __syncthreads();
    } else {
        for (int i = tid; i < S::ROWS * (D / 4); i += 512) {
            const int row = i / (D / 4), c4 = i % (D / 4);
            const long grow = min(row0 + row, (long)a.rows - 1);
            *reinterpret_cast<f32x4 *>(bufB + row * S::LDA + c4 * 4) =
                *reinterpret_cast<const f32x4 *>(a.x + grow * D + c4 * 4);
        }
        __syncthreads();
    }

    if (a.do_qkv) {
        // ---- packed in-projection of the next attention: q,k row-major per head, v transposed ----
        f32x16 acc[S::NT_QKV];
#pragma unroll
        for (int t = 0; t < S::NT_QKV; ++t) acc[t] = f32x16{0};
        const int col0 = cg * S::NT_QKV * 32;
        wave_gemm<D, S::NT_QKV>(bufB + (rt * 32 + r) * S::LDA + 4 * h, a.wqkv + (size_t)(col0 + r) * D + 4 * h, acc);
        const bool vec_ok = (a.tokens & 3) == 0;
#pragma unroll
        for (int t = 0; t < S::NT_QKV; ++t) {
            const int col = col0 + t * 32 + r;   // in [0, 3D)
            const float bias = a.bqkv[col];
            const int which = col / D;           // 0 q, 1 k, 2 v  (wave-uniform: tiles are 32-aligned)
            const int head = (col % D) / kHeadDim;
            const int e = col % kHeadDim;        // == r
            if (which < 2) {
                float *dst = which == 0 ? a.q : a.k;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const long grow = row0 + rt * 32 + acc_row(i, h);
                    if (grow < a.rows) {
                        const long plane = grow / a.tokens, tok = grow % a.tokens;
                        dst[((plane * a.heads + head) * a.tokpad + tok) * kHeadDim + e] = acc[t][i] + bias;
                    }
                }
            } else {
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {  // registers 4gq..4gq+3 = four consecutive rows
                    const long grow = row0 + rt * 32 + acc_row(4 * gq, h);
                    if (vec_ok && grow + 3 < a.rows) {
                        const long plane = grow / a.tokens, tok = grow % a.tokens;
                        f32x4 v = {acc[t][4 * gq] + bias, acc[t][4 * gq + 1] + bias, acc[t][4 * gq + 2] + bias,
                                   acc[t][4 * gq + 3] + bias};
                        *reinterpret_cast<f32x4 *>(a.vt + ((plane * a.heads + head) * kHeadDim + e) * a.tokpad + tok) = v;
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const long gr = grow + i;
                            if (gr < a.rows) {
                                const long plane = gr / a.tokens, tok = gr % a.tokens;
                                a.vt[((plane * a.heads + head) * kHeadDim + e) * a.tokpad + tok] = acc[t][4 * gq + i] + bias;
                            }
                        }
                    }
                }
            }
        }
    }
}

template <int D, int RT>
static hipError_t launch_chain_t(const ChainArgs &args, hipStream_t st) {
    using S = ChainShape<D, RT>;
    static bool attr_set = false;  // idempotent; races only repeat the same call
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(chain_kernel<D, RT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int blocks = (args.rows + S::ROWS - 1) / S::ROWS;
    hipLaunchKernelGGL((chain_kernel<D, RT>), dim3(blocks), dim3(512), S::LDS_BYTES, st, args);
    return hipGetLastError();
}

hipError_t launch_chain(const aft_config &c, const aft_layer_weights *m, const aft_layer_weights *qw,
                        const float *attn, float *x, float *q, float *k, float *vt, int rows, int tokens,
                        int tokpad, hipStream_t st) {
    ChainArgs a{};
    a.attn = attn;
    a.x = x;
    if (m != nullptr) {
        a.wo = m->out_proj_w; a.bo = m->out_proj_b;
        a.w1 = m->lin1_w; a.b1 = m->lin1_b;
        a.w2 = m->lin2_w; a.b2 = m->lin2_b;
        a.g1 = m->norm1_w; a.be1 = m->norm1_b;
        a.g2 = m->norm2_w; a.be2 = m->norm2_b;
    }
    if (qw != nullptr) {
        a.wqkv = qw->in_proj_w;
        a.bqkv = qw->in_proj_b;
    }
    a.q = q; a.k = k; a.vt = vt;
    a.rows = rows; a.tokens = tokens; a.tokpad = tokpad;
    a.heads = c.num_head; a.activation = c.activation;
    a.do_mlp = m != nullptr; a.do_qkv = qw != nullptr;
    if (c.model_dim == 128) return launch_chain_t<128, 2>(a, st);
    if (c.model_dim == 256) return launch_chain_t<256, 1>(a, st);
    return hipErrorInvalidValue;
}

}  // namespace aft
