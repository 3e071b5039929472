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
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <type_traits>

#include "aft_internal.h"

namespace aft {

template <int D, int RT, int NWAVES>
struct ChainShape {
    static constexpr int ROWS = 32 * RT;
    static constexpr int WAVES = NWAVES;
    static constexpr int THREADS = 64 * NWAVES;
    static constexpr int CG = WAVES / RT;       // column groups
    static constexpr int LDA = D + 4;           // padded row strides (floats)
    static constexpr int LDH = 2 * D + 4;
    static constexpr int NT_D = (D / 32) / CG;      // column tiles per wave for N = D
    static constexpr int NT_FF = (2 * D / 32) / CG; // N = 2D
    static constexpr int NT_QKV = (3 * D / 32) / CG;
    // LDS plan (floats): [B: ROWS x LDA][H: ROWS x LDH]; the attention tile A aliases the upper half
    // of H and both pre-LayerNorm scratch tiles alias its lower half (lifetimes in chain_kernel).
    static constexpr int H_HALF = ROWS * LDA;                 // A starts right above the scratch tile
    static constexpr int H_FLOATS = 2 * ROWS * LDA > ROWS * LDH ? 2 * ROWS * LDA : ROWS * LDH;
    static constexpr int LN_FLOATS = 6 * D;                    // gamma1, beta1, gamma2, beta2, q bias, k bias
    static constexpr size_t LDS_BYTES = sizeof(float) * (size_t)(ROWS * LDA + H_FLOATS + LN_FLOATS);
    static_assert(NT_D >= 1 && (D / 32) % CG == 0, "column split must be whole tiles");
    static_assert(CG == D / kHeadDim && NT_QKV == 3, "QKV epilogue: one head per wave (q, k, v tiles)");
};

struct ChainArgs {
    // MLP part (may be disabled)
    const float *attn;  // [rows, D]
    float *x;           // [rows, D] residual in, layer output out
    const float *wo, *bo, *w1, *b1, *w2, *b2, *g1, *be1, *g2, *be2;   // wo/w1/w2/wqkv: PACKED copies
    // QKV part (may be disabled)
    const float *wqkv, *bqkv;
    float *q, *k, *vt;
    int rows, tokens, tokpad, heads, activation, do_mlp, do_qkv;
    unsigned long long *stamps;  // diagnostic build only (AFT_DIAG_STAMPS), else NULL
};

__device__ __forceinline__ int acc_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

// Weight-fragment ring of one wave: RING k-blocks (32 deep) x NT column tiles x 4 k-steps.
//   w_lane = &W[col0 + r][4h] (global, torch [out,in] layout): lane (r,h) loads W[col][8s+4h..+3].
template <int NT, int PF>
struct WRing {
    f32x4 b[PF + 1][NT][4];
};

// Packed weight layout (see pack_weights_kernel): [col tile][k-block][k-step s][lane][4 floats] so
// that ONE global_load_dwordx4 of a wave reads 1 KB contiguous (16 x 64-B accesses in the TA
// instead of 64 scattered ones: with the torch [out,in] layout the texture addresser, not the
// matrix pipe, was the bottleneck -- GRBM_TA_BUSY 92 %, 61 cache accesses per load instruction).
//   w_lane = packed + (first col tile) * (K/32) * 1024 + lane * 4
template <int K, int NT, int PF, int TS = 1>
__device__ __forceinline__ void ring_load(WRing<NT, PF> &ring, const float *__restrict__ w_lane, int kb) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int s = 0; s < 4; ++s)
            ring.b[kb % (PF + 1)][t][s] =
                *reinterpret_cast<const f32x4 *>(w_lane + (size_t)(t * TS * (K / 32) + kb) * 1024 + s * 256);
}

// Issue the first PF k-blocks of a GEMM's weights.  Called BEFORE the previous phase's epilogue /
// barrier / LayerNorm so the L2 latency of a phase's first fragments hides under that work.
template <int K, int NT, int PF, int TS = 1>
__device__ __forceinline__ void gemm_preload(WRing<NT, PF> &ring, const float *__restrict__ w_lane) {
#pragma unroll
    for (int p = 0; p < PF && p < K / 32; ++p) ring_load<K, NT, PF, TS>(ring, w_lane, p);
    __builtin_amdgcn_sched_barrier(0);   // keep the loads here, ahead of the epilogue that follows
}

// acc[t] += A_tile[32 x K] * W[tile t][K]^T for this wave; a_lane = &A[row r][4h] (LDS).
// Fragments run PF k-blocks ahead of the MFMAs that consume them; sched_barrier pins the issue
// order so the compiler cannot sink the loads back next to their use.  Activation and weight
// fragments have the same (row, k) lane map, so passing them to the MFMA in the other order yields
// the TRANSPOSED product: bit t of SWAP makes tile t come out as acc[feature][row] (lane = row).
template <int K, int NT, int PF, int TS = 1, unsigned SWAP = 0>
__device__ __forceinline__ void gemm_run(WRing<NT, PF> &ring, const float *a_lane, const float *__restrict__ w_lane,
                                         f32x16 (&acc)[NT]) {
    constexpr int NKB = K / 32;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        if (kb + PF < NKB) ring_load<K, NT, PF, TS>(ring, w_lane, kb + PF);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const f32x4 a = *reinterpret_cast<const f32x4 *>(a_lane + kb * 32 + s * 8);
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float wv = ring.b[kb % (PF + 1)][t][s][j];
                    acc[t] = (SWAP >> t) & 1 ? __builtin_amdgcn_mfma_f32_32x32x2f32(wv, a[j], acc[t], 0, 0, 0)
                                             : __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], wv, acc[t], 0, 0, 0);
                }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// erf(x) = sign(x) * (1 - 2^p(|x|)), p = degree-8 fit of log2(erfc(t)) on [0,4] with p(0) = 0
// (erfc(4) = 1.5e-8 rounds to 0 against 1 in fp32).  Branch-free, 8 FMA + one v_exp_f32;
// max |error| 1e-7 (<= 1 ulp of erf near 1) measured against scipy.special.erf on 4e5 points --
// the libm erff it replaces cost ~40 VALU ops and a divergent branch per element.
__device__ __forceinline__ float erf_poly(float x) {
    const float t = fminf(fabsf(x), 4.0f);
    float p = -4.535924745e-05f;
    p = fmaf(p, t, 4.455104063e-04f);
    p = fmaf(p, t, -1.489443355e-03f);
    p = fmaf(p, t, -7.746370393e-04f);
    p = fmaf(p, t, 2.825369500e-02f);
    p = fmaf(p, t, -1.484816223e-01f);
    p = fmaf(p, t, -9.184163809e-01f);
    p = fmaf(p, t, -1.627908587e+00f);
    return copysignf(1.0f - __builtin_amdgcn_exp2f(p * t), x);
}

template <int ACT>
__device__ __forceinline__ float activate(float v) {
    // exact-erf GELU (F.gelu default, activation="gelu") or ReLU -- schemas.py:128-131 allows both
    if constexpr (ACT == AFT_ACT_GELU) {
        const float hv = 0.5f * v;
        return fmaf(hv, erf_poly(v * 0.70710678118654752440f), hv);
    } else {
        return fmaxf(v, 0.0f);
    }
}

// Two activations at once: the polynomial runs on v_pk_fma_f32 (two floats per lane per issue).
template <int ACT>
__device__ __forceinline__ f32x2 activate2(f32x2 v) {
    if constexpr (ACT == AFT_ACT_GELU) {
        const f32x2 x = v * 0.70710678118654752440f;
        const f32x2 t = {fminf(fabsf(x[0]), 4.0f), fminf(fabsf(x[1]), 4.0f)};
        f32x2 p = {-4.535924745e-05f, -4.535924745e-05f};
        p = __builtin_elementwise_fma(p, t, f32x2{4.455104063e-04f, 4.455104063e-04f});
        p = __builtin_elementwise_fma(p, t, f32x2{-1.489443355e-03f, -1.489443355e-03f});
        p = __builtin_elementwise_fma(p, t, f32x2{-7.746370393e-04f, -7.746370393e-04f});
        p = __builtin_elementwise_fma(p, t, f32x2{2.825369500e-02f, 2.825369500e-02f});
        p = __builtin_elementwise_fma(p, t, f32x2{-1.484816223e-01f, -1.484816223e-01f});
        p = __builtin_elementwise_fma(p, t, f32x2{-9.184163809e-01f, -9.184163809e-01f});
        p = __builtin_elementwise_fma(p, t, f32x2{-1.627908587e+00f, -1.627908587e+00f});
        p = p * t;
        const f32x2 e = {copysignf(1.0f - __builtin_amdgcn_exp2f(p[0]), x[0]),
                         copysignf(1.0f - __builtin_amdgcn_exp2f(p[1]), x[1])};
        const f32x2 hv = v * 0.5f;
        return __builtin_elementwise_fma(hv, e, hv);
    } else {
        return f32x2{fmaxf(v[0], 0.0f), fmaxf(v[1], 0.0f)};
    }
}

// Sum over groups of LPR consecutive lanes with DPP (no LDS crossbar): quad_perm xor-1, xor-2,
// then row_half_mirror (8 lanes) and row_mirror (16 lanes).  Every lane ends with its group's sum.
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
    static_assert(LPR == 4 || LPR == 8 || LPR == 16, "group size");
    auto dpp = [](float x, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, true));
    };
    v += dpp(v, std::integral_constant<int, 0xB1>{});   // quad_perm [1,0,3,2]
    v += dpp(v, std::integral_constant<int, 0x4E>{});   // quad_perm [2,3,0,1]
    if constexpr (LPR >= 8) v += dpp(v, std::integral_constant<int, 0x141>{});   // row_half_mirror
    if constexpr (LPR >= 16) v += dpp(v, std::integral_constant<int, 0x140>{});  // row_mirror
    return v;
}

// LayerNorm(eps=1e-5, biased variance) of ROWS rows held in `src` (row stride ld floats), result to
// `dst` (same stride) and optionally to global rows.  A wave owns ROWS/WAVES rows and processes
// them in ONE pass: LPR = 64/(ROWS/WAVES) lanes share a row, each lane holds D/LPR contiguous
// values (ds_read_b128), the row statistics are two DPP group reductions.
template <int D, int ROWS, int WAVES>
__device__ __forceinline__ void layernorm_tile(const float *src, float *dst, int ld,
                                               const float *gamma, const float *beta,
                                               float *gout, long row0, int rows, int wave, int lane) {
    constexpr int RPW = ROWS / WAVES, LPR = 64 / RPW, VPL = D / LPR;
    static_assert(VPL % 4 == 0, "per-lane slice must be float4-able");
    const int row = wave * RPW + lane / LPR, c0 = (lane % LPR) * VPL;
    float v[VPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; i += 4) {
        const f32x4 t = *reinterpret_cast<const f32x4 *>(src + row * ld + c0 + i);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[i + j] = t[j];
            s += t[j];
        }
    }
    const float mean = group_sum<LPR>(s) * (1.0f / D);
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        v[i] -= mean;
        sq = fmaf(v[i], v[i], sq);
    }
    const float rstd = rsqrtf(group_sum<LPR>(sq) * (1.0f / D) + 1e-5f);
    const bool store = gout != nullptr && row0 + row < rows;
#pragma unroll
    for (int i = 0; i < VPL; i += 4) {
        const f32x4 g = *reinterpret_cast<const f32x4 *>(gamma + c0 + i);   // LDS copies
        const f32x4 b = *reinterpret_cast<const f32x4 *>(beta + c0 + i);
        f32x4 y;
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] = v[i + j] * rstd * g[j] + b[j];
        *reinterpret_cast<f32x4 *>(dst + row * ld + c0 + i) = y;
        if (store) *reinterpret_cast<f32x4 *>(gout + (row0 + row) * (long)D + c0 + i) = y;
    }
}

// MLP / QKV select the three launch variants at compile time (distinct symbols in a profile):
//   <true,true>  layer l's out-proj+LN1+FFN+LN2 and layer l+1's in-projection   (5 of 7 launches at L=6)
//   <false,true> in-projection only (first layer)      <true,false> last layer, no in-projection
template <int D, int RT, int NWAVES, int ACT, bool MLP, bool QKV>
__global__ __launch_bounds__(64 * NWAVES, NWAVES == 4 ? 3 : 2) void chain_kernel(const ChainArgs a) {
    using S = ChainShape<D, RT, NWAVES>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *bufB = smem;                       // x (residual) -> x1 (LN1 out) -> x2 (LN2 out) = QKV operand
    float *bufH = bufB + S::ROWS * S::LDA;    // FFN hidden [ROWS][LDH]
    float *bufA = bufH + S::H_HALF;           // attention tile (dead after the out-projection) = top of H
    float *bufS = bufH;                       // pre-LayerNorm scratch [ROWS][LDA] = bottom of H
    float *lnp = bufH + S::H_FLOATS;          // LayerNorm affine parameters (only with do_mlp)
    float *qkb = lnp + 4 * D;                 // q and k in-proj bias [2][D] (per-feature = per-register in the swapped tiles)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int rt = wave / S::CG, cg = wave % S::CG;
    const long row0 = (long)blockIdx.x * S::ROWS;

    WRing<S::NT_D, 2> ring_d;       // out-proj / FFN-down fragments
    WRing<S::NT_FF, 1> ring_ff;     // FFN-up fragments
    WRing<S::NT_QKV, 1> ring_qkv;   // in-projection fragments
    const int col0_d = cg * S::NT_D * 32, col0_ff = cg * S::NT_FF * 32;
    const int col0_qkv = cg * 32;   // wave cg owns head cg: column tiles cg (q), CG+cg (k), 2CG+cg (v)
    // fragment-packed weights: [col tile][k-block][s][lane][4]
    const float *wo_lane = a.wo + (size_t)(col0_d / 32) * (D / 32) * 1024 + lane * 4;
    const float *w1_lane = a.w1 + (size_t)(col0_ff / 32) * (D / 32) * 1024 + lane * 4;
    const float *w2_lane = a.w2 + (size_t)(col0_d / 32) * (2 * D / 32) * 1024 + lane * 4;
    const float *wq_lane = a.wqkv + (size_t)(col0_qkv / 32) * (D / 32) * 1024 + lane * 4;

#ifdef AFT_DIAG_STAMPS   // diagnostic build only (-DAFT_DIAG_STAMPS): per-phase s_memtime stamps of wave 0
#define STAMP(i)                                                                              \
    do {                                                                                      \
        if (a.stamps && tid == 0) a.stamps[(size_t)blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define STAMP(i) do { } while (0)
#endif
    STAMP(0);
    // per-column vectors are fetched ONCE, up front: a global load inside an epilogue would sit
    // behind an in-order vmcnt wait together with the weight prefetch and every earlier store
    float bias_o[S::NT_D], bias_1[S::NT_FF], bias_2[S::NT_D], bias_q[S::NT_QKV];
    if constexpr (MLP) {
#pragma unroll
        for (int t = 0; t < S::NT_D; ++t) {
            bias_o[t] = a.bo[col0_d + t * 32 + r];
            bias_2[t] = a.b2[col0_d + t * 32 + r];
        }
#pragma unroll
        for (int t = 0; t < S::NT_FF; ++t) bias_1[t] = a.b1[col0_ff + t * 32 + r];
        for (int i = tid; i < D; i += S::THREADS) {
            lnp[i] = a.g1[i];
            lnp[D + i] = a.be1[i];
            lnp[2 * D + i] = a.g2[i];
            lnp[3 * D + i] = a.be2[i];
        }
    }
    if constexpr (QKV) {
        for (int i = tid; i < 2 * D; i += S::THREADS) qkb[i] = a.bqkv[i];
#pragma unroll
        for (int t = 0; t < S::NT_QKV; ++t) bias_q[t] = a.bqkv[t * D + col0_qkv + r];   // v tile: lane = feature
    }
    if constexpr (MLP) {
        gemm_preload<D, S::NT_D, 2>(ring_d, wo_lane);   // in flight while the tiles are staged
        // ---- stage the attention-output tile (A operand of the out-projection) and the residual x ----
        // all loads first, then the LDS writes: one memory round trip for the whole tile
        {
            constexpr int ITER = S::ROWS * (D / 4) / S::THREADS;
            f32x4 va[ITER], vx[ITER];
#pragma unroll
            for (int it = 0; it < ITER; ++it) {
                const int i = tid + it * S::THREADS, row = i / (D / 4), c4 = i % (D / 4);
                const long grow = min(row0 + row, (long)a.rows - 1);
                va[it] = *reinterpret_cast<const f32x4 *>(a.attn + grow * D + c4 * 4);
                vx[it] = *reinterpret_cast<const f32x4 *>(a.x + grow * D + c4 * 4);
            }
#pragma unroll
            for (int it = 0; it < ITER; ++it) {
                const int i = tid + it * S::THREADS, row = i / (D / 4), c4 = i % (D / 4);
                *reinterpret_cast<f32x4 *>(bufA + row * S::LDA + c4 * 4) = va[it];
                *reinterpret_cast<f32x4 *>(bufB + row * S::LDA + c4 * 4) = vx[it];
            }
        }
        __syncthreads();
        STAMP(1);
        // ---- out-projection + bias + residual -> pre-LN1 scratch (bufH, stride LDA) ----
        {
            f32x16 acc[S::NT_D];
#pragma unroll
            for (int t = 0; t < S::NT_D; ++t) acc[t] = f32x16{0};
            const int col0 = col0_d;
            gemm_run<D, S::NT_D, 2>(ring_d, bufA + (rt * 32 + r) * S::LDA + 4 * h, wo_lane, acc);
            STAMP(2);
            gemm_preload<D, S::NT_FF, 1>(ring_ff, w1_lane);   // next phase's first fragments
#pragma unroll
            for (int t = 0; t < S::NT_D; ++t) {
                const int col = col0 + t * 32 + r;
                const float bias = bias_o[t];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = rt * 32 + acc_row(e, h);
                    bufS[row * S::LDA + col] = acc[t][e] + bias + bufB[row * S::LDA + col];
                }
            }
        }
        __syncthreads();
        STAMP(3);
        layernorm_tile<D, S::ROWS, S::WAVES>(bufS, bufB, S::LDA, lnp, lnp + D, nullptr, row0, a.rows, wave, lane);
        __syncthreads();
        STAMP(4);
        // ---- FFN up-projection + activation -> hidden tile in LDS ----
        {
            f32x16 acc[S::NT_FF];
#pragma unroll
            for (int t = 0; t < S::NT_FF; ++t) acc[t] = f32x16{0};
            const int col0 = col0_ff;
            gemm_run<D, S::NT_FF, 1>(ring_ff, bufB + (rt * 32 + r) * S::LDA + 4 * h, w1_lane, acc);
            STAMP(5);
            gemm_preload<2 * D, S::NT_D, 2>(ring_d, w2_lane);
#pragma unroll
            for (int t = 0; t < S::NT_FF; ++t) {
                const int col = col0 + t * 32 + r;
                const float bias = bias_1[t];
#pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    const f32x2 g = activate2<ACT>(f32x2{acc[t][e] + bias, acc[t][e + 1] + bias});
                    bufH[(rt * 32 + acc_row(e, h)) * S::LDH + col] = g[0];
                    bufH[(rt * 32 + acc_row(e + 1, h)) * S::LDH + col] = g[1];
                }
            }
        }
        __syncthreads();
        STAMP(6);
        // ---- FFN down-projection + bias + residual(x1) -> pre-LN2 scratch (bufA) ----
        {
            f32x16 acc[S::NT_D];
#pragma unroll
            for (int t = 0; t < S::NT_D; ++t) acc[t] = f32x16{0};
            const int col0 = col0_d;
            gemm_run<2 * D, S::NT_D, 2>(ring_d, bufH + (rt * 32 + r) * S::LDH + 4 * h, w2_lane, acc);
            if constexpr (QKV) gemm_preload<D, S::NT_QKV, 1, S::CG>(ring_qkv, wq_lane);
            STAMP(7);
            __syncthreads();   // every wave is done reading H before its bottom half becomes LN scratch
#pragma unroll
            for (int t = 0; t < S::NT_D; ++t) {
                const int col = col0 + t * 32 + r;
                const float bias = bias_2[t];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = rt * 32 + acc_row(e, h);
                    bufS[row * S::LDA + col] = acc[t][e] + bias + bufB[row * S::LDA + col];
                }
            }
        }
        __syncthreads();
        STAMP(8);
        layernorm_tile<D, S::ROWS, S::WAVES>(bufS, bufB, S::LDA, lnp + 2 * D, lnp + 3 * D, a.x, row0, a.rows, wave, lane);
        __syncthreads();
        STAMP(9);
    } else {
        gemm_preload<D, S::NT_QKV, 1, S::CG>(ring_qkv, wq_lane);
        {
            constexpr int ITER = S::ROWS * (D / 4) / S::THREADS;
            f32x4 vx[ITER];
#pragma unroll
            for (int it = 0; it < ITER; ++it) {
                const int i = tid + it * S::THREADS, row = i / (D / 4), c4 = i % (D / 4);
                const long grow = min(row0 + row, (long)a.rows - 1);
                vx[it] = *reinterpret_cast<const f32x4 *>(a.x + grow * D + c4 * 4);
            }
#pragma unroll
            for (int it = 0; it < ITER; ++it) {
                const int i = tid + it * S::THREADS, row = i / (D / 4), c4 = i % (D / 4);
                *reinterpret_cast<f32x4 *>(bufB + row * S::LDA + c4 * 4) = vx[it];
            }
        }
        __syncthreads();
    }

    if constexpr (QKV) {
        // ---- packed in-projection of the next attention: q,k row-major per head, v transposed ----
        f32x16 acc[S::NT_QKV];
#pragma unroll
        for (int t = 0; t < S::NT_QKV; ++t) acc[t] = f32x16{0};
        gemm_run<D, S::NT_QKV, 1, S::CG, 0x3>(ring_qkv, bufB + (rt * 32 + r) * S::LDA + 4 * h, wq_lane, acc);
        STAMP(10);
        // ---- epilogue: q, k, v of head `cg` for 32 token rows, written in MFMA-FRAGMENT order so that
        // k_attn.hip reads every operand with fully coalesced 1-KB loads:
        //   q, k : [plane*H + head][key tile][s][lane = key%32 + 32*hh][4]   value (key, d = 8s + 4hh + j)
        //   vt   : [plane*H + head][key tile][g][lane = d + 32*hh][4]        value (d, key = 32kt + 8g + 4hh + j)
        // q/k tiles were computed transposed (lane = token row, registers = features): registers
        // 4s..4s+3 of half hh are exactly one 16-byte fragment element; the v tile (lane = feature,
        // registers = tokens) likewise.  32 lanes x 16 B = 512 B contiguous per store instruction.
        const int base_row = (int)row0 + rt * 32;
        const int plane0 = base_row / a.tokens, tok0 = base_row - plane0 * a.tokens;   // tok0 % 8 == 0
        const int nkt = a.tokpad / kTile;
        const unsigned head_stride = (unsigned)a.tokpad * kHeadDim;            // floats per (plane, head)
        const unsigned ph0 = (unsigned)(plane0 * a.heads + cg);
        const bool full = base_row + 32 <= a.rows;
        // q / k : this lane's token
        {
            int tok = tok0 + r;
            unsigned ph = ph0;
            if (tok >= a.tokens) { tok -= a.tokens; ph += a.heads; }
            const unsigned lane_off = ph * head_stride + (unsigned)(tok >> 5) * 1024 + ((tok & 31) + 32 * h) * 4;
            const bool ok = full || base_row + r < a.rows;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float *dst = (t == 0 ? a.q : a.k) + lane_off;
                // bias is per FEATURE = per register here
#pragma unroll
                for (int sgrp = 0; sgrp < 4; ++sgrp) {
                    const f32x4 bv = *reinterpret_cast<const f32x4 *>(qkb + t * D + cg * kHeadDim + 8 * sgrp + 4 * h);
                    const f32x4 v = {acc[t][4 * sgrp] + bv[0], acc[t][4 * sgrp + 1] + bv[1], acc[t][4 * sgrp + 2] + bv[2],
                                     acc[t][4 * sgrp + 3] + bv[3]};
                    if (ok) *reinterpret_cast<f32x4 *>(dst + sgrp * 256) = v;
                }
            }
        }
        // v : this lane's feature d = r, registers 4g..4g+3 = tokens tok0 + 8g + 4h + {0..3}
        {
            const float bias = bias_q[2];
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                int tok = tok0 + 8 * gq + 4 * h;
                unsigned ph = ph0;
                if (tok >= a.tokens) { tok -= a.tokens; ph += a.heads; }
                const unsigned off = ph * head_stride + (unsigned)(tok >> 5) * 1024 + ((tok >> 3) & 3) * 256 + (r + 32 * h) * 4;
                const f32x4 v = {acc[2][4 * gq] + bias, acc[2][4 * gq + 1] + bias, acc[2][4 * gq + 2] + bias,
                                 acc[2][4 * gq + 3] + bias};
                if (full || base_row + 8 * gq + 4 * h + 3 < a.rows) *reinterpret_cast<f32x4 *>(a.vt + off) = v;
                else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (base_row + 8 * gq + 4 * h + j < a.rows) a.vt[off + j] = v[j];
                }
            }
        }
        (void)nkt;
    }
    STAMP(11);
}

template <int D, int RT, int NWAVES, int ACT, bool MLP, bool QKV>
static hipError_t launch_chain_v(const ChainArgs &args, hipStream_t st) {
    using S = ChainShape<D, RT, NWAVES>;
    static bool attr_set = false;  // idempotent; races only repeat the same call
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(chain_kernel<D, RT, NWAVES, ACT, MLP, QKV>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int blocks = (args.rows + S::ROWS - 1) / S::ROWS;
    const size_t lds = S::LDS_BYTES;
#ifdef AFT_DIAG_STAMPS
    if (getenv("AFT_STAMPS")) {   // phase stamps, printed on the host (never in the product build)
        static unsigned long long *dbuf = nullptr;
        if (!dbuf) (void)hipMalloc(&dbuf, sizeof(unsigned long long) * 16 * 4096);
        (void)hipMemset(dbuf, 0, sizeof(unsigned long long) * 16 * 4096);
        ChainArgs a2 = args;
        a2.stamps = dbuf;
        hipLaunchKernelGGL((chain_kernel<D, RT, NWAVES, ACT, MLP, QKV>), dim3(blocks), dim3(S::THREADS), lds, st, a2);
        (void)hipDeviceSynchronize();
        static int printed = 0;
        if (printed == 0) {
            int nb = 0;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, chain_kernel<D, RT, NWAVES, ACT, MLP, QKV>, S::THREADS, lds);
            printf("occupancy API: %d blocks/CU at %zu B LDS, %d threads\n", nb, lds, S::THREADS);
        }
        if (printed++ < 2) {
            std::vector<unsigned long long> h(16 * 4096);
            (void)hipMemcpy(h.data(), dbuf, h.size() * 8, hipMemcpyDeviceToHost);
            unsigned long long t0 = ~0ull, t1 = 0;
            double sum[12] = {0};
            for (int b = 0; b < blocks; ++b) {
                t0 = std::min(t0, h[b * 16]);
                t1 = std::max(t1, h[b * 16 + 11]);
                for (int i = 1; i < 12; ++i) {
                    unsigned long long prev = h[b * 16 + i - 1], cur = h[b * 16 + i];
                    if (cur == 0 || prev == 0) { int k = i - 1; while (k > 0 && h[b * 16 + k] == 0) --k; prev = h[b * 16 + k]; }
                    if (cur) sum[i] += (double)(cur - prev);
                }
            }
            printf("chain stamps (100MHz ticks?): total span %llu ; mean per-phase:", (unsigned long long)(t1 - t0));
            for (int i = 1; i < 12; ++i) printf(" [%d]=%.0f", i, sum[i] / blocks);
            printf("\n  first 6 WG start offsets:");
            for (int b = 0; b < 6; ++b) printf(" %llu", h[b * 16] - t0);
            printf(" ... WG 768: %llu, WG 1536: %llu\n", h[768 * 16] - t0, h[1536 * 16] - t0);
        }
        return hipGetLastError();
    }
#endif
    hipLaunchKernelGGL((chain_kernel<D, RT, NWAVES, ACT, MLP, QKV>), dim3(blocks), dim3(S::THREADS), lds, st, args);
    return hipGetLastError();
}

// Packed-weight block of one layer inside the workspace (floats): [in_proj 3D*D][out_proj D*D][lin1 2D*D][lin2 D*2D]
size_t packed_layer_floats(int d) { return (size_t)8 * d * d; }

__global__ __launch_bounds__(256) void pack_weights_kernel(const aft_weights w, float *__restrict__ packed, int d,
                                                           int num_layers) {
    const size_t per_layer = (size_t)8 * d * d;
    const size_t v = (size_t)blockIdx.x * 256 + threadIdx.x;   // one float4 of the packed image
    if (v * 4 >= per_layer * num_layers) return;
    const int layer = (int)(v * 4 / per_layer);
    size_t off = v * 4 - (size_t)layer * per_layer;
    const aft_layer_weights &lw = w.layers[layer];
    const float *src;
    int K;
    if (off < (size_t)3 * d * d) { src = lw.in_proj_w; K = d; }
    else if ((off -= (size_t)3 * d * d) < (size_t)d * d) { src = lw.out_proj_w; K = d; }
    else if ((off -= (size_t)d * d) < (size_t)2 * d * d) { src = lw.lin1_w; K = d; }
    else { off -= (size_t)2 * d * d; src = lw.lin2_w; K = 2 * d; }
    // off = (((ct*NKB + kb)*4 + s)*64 + lane)*4
    const int lane = (int)(off / 4) % 64, s = (int)(off / 256) % 4;
    const int blk = (int)(off / 1024), nkb = K / 32, kb = blk % nkb, ct = blk / nkb;
    const int col = ct * 32 + (lane & 31), k = kb * 32 + s * 8 + (lane >> 5) * 4;
    *reinterpret_cast<f32x4 *>(packed + v * 4) = *reinterpret_cast<const f32x4 *>(src + (size_t)col * K + k);
}

hipError_t launch_pack_weights(const aft_config &c, const aft_weights &w, float *packed, int first_layer, int count,
                               hipStream_t st) {
    aft_weights shifted = w;   // kernel indexes layers from 0: shift the window
    for (int i = 0; i < count; ++i) shifted.layers[i] = w.layers[first_layer + i];
    const size_t vecs = packed_layer_floats(c.model_dim) * count / 4;
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)((vecs + 255) / 256)), dim3(256), 0, st, shifted, packed,
                       c.model_dim, count);
    return hipGetLastError();
}

template <int D, int RT, int NWAVES, int ACT>
static hipError_t launch_chain_t(const ChainArgs &args, hipStream_t st) {
    if (args.do_mlp && args.do_qkv) return launch_chain_v<D, RT, NWAVES, ACT, true, true>(args, st);
    if (args.do_mlp) return launch_chain_v<D, RT, NWAVES, ACT, true, false>(args, st);
    return launch_chain_v<D, RT, NWAVES, ACT, false, true>(args, st);
}

hipError_t launch_chain(const aft_config &c, const aft_layer_weights *m, const float *m_packed,
                        const aft_layer_weights *qw, const float *q_packed,
                        const float *attn, float *x, float *q, float *k, float *vt, int rows, int tokens,
                        int tokpad, hipStream_t st) {
    const size_t dd = (size_t)c.model_dim * c.model_dim;
    ChainArgs a{};
    a.attn = attn;
    a.x = x;
    if (m != nullptr) {
        a.wo = m_packed + 3 * dd; a.bo = m->out_proj_b;
        a.w1 = m_packed + 4 * dd; a.b1 = m->lin1_b;
        a.w2 = m_packed + 6 * dd; a.b2 = m->lin2_b;
        a.g1 = m->norm1_w; a.be1 = m->norm1_b;
        a.g2 = m->norm2_w; a.be2 = m->norm2_b;
    }
    if (qw != nullptr) {
        a.wqkv = q_packed;
        a.bqkv = qw->in_proj_b;
    }
    a.q = q; a.k = k; a.vt = vt;
    a.rows = rows; a.tokens = tokens; a.tokpad = tokpad;
    a.heads = c.num_head; a.activation = c.activation;
    a.do_mlp = m != nullptr; a.do_qkv = qw != nullptr;
    // d=128: 32-row tiles, 4 waves, 67 KB LDS -> two workgroups per CU overlap each other's
    // barrier/epilogue/LayerNorm phases; d=256: 32-row tiles, 8 waves, 133 KB LDS.
    const bool gelu = c.activation == AFT_ACT_GELU;
    if (c.model_dim == 128)
        return gelu ? launch_chain_t<128, 1, 4, AFT_ACT_GELU>(a, st) : launch_chain_t<128, 1, 4, AFT_ACT_RELU>(a, st);
    if (c.model_dim == 256)
        return gelu ? launch_chain_t<256, 1, 8, AFT_ACT_GELU>(a, st) : launch_chain_t<256, 1, 8, AFT_ACT_RELU>(a, st);
    return hipErrorInvalidValue;
}

}  // namespace aft
