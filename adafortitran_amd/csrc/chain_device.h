// chain_device.h (device code of k_chain.hip, shared with k_encoder.hip) -- the row-local part of an encoder layer as ONE gfx950 kernel.
//
// Reference semantics (nn.TransformerEncoderLayer, post-LN, eval; constructed at
// reference src/models/blocks/encoders.py:44-55, dim_feedforward = 2*model_dim :47):
//     x1 = LN1(x + attn @ Wo^T + bo)
//     x2 = LN2(x1 + act(x1 @ W1^T + b1) @ W2^T + b2)
// and, fused behind it, the NEXT layer's packed in-projection (in_proj_weight [3d,d] =
// [Wq;Wk;Wv]):   q,k,v = split(x2 @ Wqkv^T + bqkv)  written per head for k_attn.hip.
// Everything between two attention calls is row-local, so one workgroup owns a tile of 32 token
// rows through all four GEMMs.
//
// MI355X mapping (round-1 design, see DESIGN.md 4.1 for the measurements behind each choice)
//   * v_mfma_f32_32x32x2_f32: exact fp32 (the parity contract; gfx950 has no xf32), 64 cycles per
//     SIMD.  It shares the SIMD's issue with ordinary VALU work, so every VALU / LDS instruction
//     removed from the epilogues is matrix time won.
//   * W = d/32 waves per workgroup; wave w owns the 32-feature block w of EVERY activation tensor
//     (out-proj tile w, hidden tiles 2w and 2w+1, q/k/v tile of head w).
//   * every GEMM is computed TRANSPOSED: A operand = weight fragment (lane = output feature),
//     B operand = activation fragment (lane = token row).  The accumulator then has lane = row and
//     registers = features, and register e of lane half h is feature (e&3) + 8(e>>2) + 4h -- which
//     is exactly the operand-fragment element (s = e>>2, j = e&3).  So a GEMM's output IS the next
//     GEMM's operand for that feature block: bias, residual, GELU and LayerNorm all run on
//     registers, and the only data that crosses waves is one fragment-ordered 4-KB block per wave
//     per exchange (ds_write_b128 / ds_read_b128, lane-linear => conflict-free, no padding).
//   * LayerNorm: each lane reduces its 16 features, pairs with the other half-wave (one
//     cross-half exchange), the W per-wave partial (mean, M2) pairs meet in a 1-KB LDS table and are
//     merged with Chan's formula -- two-pass accuracy without re-reading the tile.
//   * weights: fragment-packed per call (pack_weights_kernel), streamed from L2 into a register
//     ring PF k-blocks ahead of their MFMAs, the first fragments of the NEXT GEMM issued before the
//     current epilogue; sched_barrier pins that order.
//   * 5 workgroup barriers per tile (2 LN tables, x1 / hidden / x2 exchanges); biases enter as the
//     accumulators' initial values; LDS = 51 KB at d=128 => 3 workgroups per CU (measured: the LDS
//     allocator rounds up, 53 KB admitted only 2 although the occupancy API said 3).
//   * HBM per row: read attn + x, write x + q,k,v; weights (512 KB/layer at d=128) stay in L2.
//   * K's in-projection bias is dropped: softmax_j(q.(k_j + b)) = softmax_j(q.k_j + q.b) and the
//     row-constant q.b cancels (it only adds rounding error to the logits); Q's bias is applied by
//     k_attn.hip when it loads the query fragment.
#pragma once
#include "aft_internal.h"
#include "srd.h"

namespace aft {

template <int D>
struct ChainShape {
    static constexpr int WAVES = D / 32;
    static constexpr int THREADS = 64 * WAVES;
    static constexpr int XB = WAVES * 1024;          // x1 / x2 exchange   [feature block][s][lane][4]
    static constexpr int HB = 2 * WAVES * 1024;      // FFN hidden exchange [hidden block][s][lane][4]
    static constexpr int ST = 32 * WAVES * 2;        // LayerNorm partials [row][wave]{mean, M2}
    static constexpr int PAR = 4 * D;                // g1, be1, g2, be2 (biases ride in as accumulator initial values)
    static constexpr size_t LDS_BYTES = sizeof(float) * (size_t)(XB + HB + ST + PAR);
};

struct ChainArgs {
    const float *attn;  // fragment-packed attention output [row tile][head][s][lane][4]
    float *x;           // [rows, D] row-major: residual in, layer output out
    const float *wo, *w1, *w2, *wqkv;                  // fragment-PACKED weights
    const float *bo, *b1, *b2, *g1, *be1, *g2, *be2;   // torch vectors
    const float *bv;                                   // in_proj_bias + 2D (value bias)
    float *q, *k, *vt;
    int rows, tokens, tokpad, heads;
    // x (the residual stream between two launches) in TILE-BLOCKED order: tile t's 32 rows as [feature block][fragment s][lane][4],
    // i.e. every 16-byte load / store of a wave is 1 KB contiguous instead of 64 pieces of 32 rows (round 4: -0.7 % on the whole
    // forward).  Only the whole-forward launch sequence sets it -- nobody else reads x there; the stage entry points, the
    // plane-resident kernel and the training path keep x row-major.  The x region must hold whole tiles.
    int x_blocked;
    // <MLP,!QKV> only, optional: transformer_encoder.linear_2 (reference blocks/encoders.py:56,70) fused behind LN2.
    // out6 [rows][out6_stride] receives x2 W2^T + b2 and x is NOT stored (the conv tail reads out6); NULL = store x.
    const float *lin2_w, *lin2_b;   // torch [P][D], [P]
    float *out6;
    int out6_features, out6_stride;
    // <!MLP,QKV> only, optional: patch embedding + adapter concat + linear_1 + positional table (reference
    // blocks/patch_processors.py:22,34-35, fortitran.py:217, encoders.py:67-68) fused in front of the in-projection:
    // x0 is computed from conv_enhanced / tokens6 instead of being read, and written to x.  NULL conv = read x.
    const float *emb_conv, *emb_tok6, *emb_w1, *emb_b1, *emb_pos;   // [planes][S][T], [frames][tokens][6] or NULL, [D][K], [D], [>=tokens][D]
    int emb_S, emb_T, emb_p0, emb_p1, emb_K;                          // K = p0*p1 (+6 with adapter tokens)
    unsigned long long *stamps;  // diagnostic build only (AFT_DIAG_STAMPS), else NULL
};

// Weight-fragment ring of one wave: PF+1 k-blocks (32 deep) x NT tiles x 4 k-steps.
// Packed layout (pack_weights_kernel): [tile][k-block][s][lane][4] so ONE global_load_dwordx4 of a
// wave reads 1 KB contiguous (16 x 64-B TA accesses instead of 64 scattered ones; with the torch
// [out,in] layout GRBM_TA_BUSY was 92 % and the matrix pipe starved).
template <int NT, int PFS>
struct WRing {   // PFS + 1 k-STEPS (8 deep: one 16-byte fragment element per tile) of NT tiles
    f32x4 b[PFS + 1][NT];
};


// Weight fragments are streamed with k-STEP granularity (one 16-byte element of each of the NT tiles = NT x 4 MFMAs):
// the ring holds the PFS steps in flight plus the one being consumed.  Round 1 prefetched whole 32-deep k-blocks
// (96 registers for the in-projection's ring); steps keep the same lead in cycles (PFS x NT x 256) at a third to a half
// of the registers -- the kernel no longer spills (a scratch reload is a VMEM load: its wait drains vmcnt, i.e. it waited
// for the stores and loads in flight at the tile seams).
template <int NKB, int NT, int PFS, int TS>
__device__ __forceinline__ void ring_load(WRing<NT, PFS> &ring, Srd w, unsigned w_lane, int step) {
    const int kb = step >> 2, s = step & 3;
#pragma unroll
    for (int t = 0; t < NT; ++t)
        ring.b[step % (PFS + 1)][t] = srd_load_c(w, w_lane, (unsigned)((t * TS * NKB + kb) * 1024 + s * 256) * 4);
}

// Issue the first PFS k-steps of a GEMM's weights -- called BEFORE the previous phase's epilogue /
// barrier so their L2 latency hides under that work.
template <int NKB, int NT, int PFS, int TS>
__device__ __forceinline__ void gemm_preload(WRing<NT, PFS> &ring, Srd w, unsigned w_lane) {
#pragma unroll
    for (int p = 0; p < PFS && p < 4 * NKB; ++p) ring_load<NKB, NT, PFS, TS>(ring, w, w_lane, p);
    __builtin_amdgcn_sched_barrier(0);
}

// acc[t] += W_tile[t] (32 features x 32*NKB) . act (32*NKB x 32 rows): transposed product, lane =
// token row.  `act(kb, s)` yields this lane's activation fragment (registers or LDS).  Bit t of
// NORMAL swaps the operands of tile t back (lane = feature), used for the V tile.
// Bit t of ZERO: tile t's accumulator starts from 0 -- its first MFMA then takes the inline constant as C and acc[t]
// needs no initialisation at all (16 v_mov per tile otherwise).
template <int NKB, int NT, int PFS, int TS, unsigned NORMAL, class Act, unsigned ZERO = 0>
__device__ __forceinline__ void gemm_run(WRing<NT, PFS> &ring, Srd w, unsigned w_lane, f32x16 (&acc)[NT], Act act) {
#pragma unroll
    for (int step = 0; step < 4 * NKB; ++step) {
        if (step + PFS < 4 * NKB) ring_load<NKB, NT, PFS, TS>(ring, w, w_lane, step + PFS);
        __builtin_amdgcn_sched_barrier(0);
        const f32x4 a = act(step >> 2, step & 3);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float wv = ring.b[step % (PFS + 1)][t][j];
                // (the s_nop spacer of aft_internal.h::mfma_f32 measured 1.5 % slower here, unlike in k_attn.hip)
                if (((ZERO >> t) & 1) && step == 0 && j == 0)
                    acc[t] = (NORMAL >> t) & 1 ? __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], wv, f32x16{0}, 0, 0, 0)
                                               : __builtin_amdgcn_mfma_f32_32x32x2f32(wv, a[j], f32x16{0}, 0, 0, 0);
                else
                    acc[t] = (NORMAL >> t) & 1 ? __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], wv, acc[t], 0, 0, 0)
                                               : __builtin_amdgcn_mfma_f32_32x32x2f32(wv, a[j], acc[t], 0, 0, 0);
            }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---- split-precision tier (aft_config.precision = AFT_PRECISION_BF16X3; never the default) ----
// Every fp32 operand x of the row-local GEMMs is split into two bf16 terms, hi = bf16(x), lo = bf16(x - hi), and a
// product is accumulated as hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_bf16 (fp32 accumulate; the dropped lo.lo term and
// the split residuals are ~2^-16 relative).  Per 32-deep k-block that is 6 MFMAs of 32 cycles instead of 16 fp32 MFMAs of
// 64: 5.3x less matrix time, and the bf16 pipe does not take the vector ALU's cycles as the fp32 MFMA does.  The data
// flow of the fp32 kernel carries over unchanged because the 32x32x16 accumulator has the same layout as the 32x32x2 one
// (lane = token row, register e = feature (e&3) + 8(e>>2) + 4h): registers 8m .. 8m+7 of a lane ARE its 8 k-values of
// MFMA m of the next product, k = 16m + 8(j>>2) + 4h + (j&3) for element j (guide: "An accumulator tile as the next
// MFMA's operand") -- the packed weights use the same k order (pack_weights_kernel, split image).
// exchange buffers in the split tier: [feature block][m][hi | lo][lane][8 bf16] -- 4 KB per block, as the fp32 fragments
__device__ __forceinline__ void bs_publish(float *buf, int block, int lane, const f32x16 &v) {
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const BsFrag f = bs_split(f32x4{v[8 * m], v[8 * m + 1], v[8 * m + 2], v[8 * m + 3]},
                                  f32x4{v[8 * m + 4], v[8 * m + 5], v[8 * m + 6], v[8 * m + 7]});
        *reinterpret_cast<f32x4 *>(buf + (block * 4 + 2 * m) * 256 + lane * 4) = f.hi;
        *reinterpret_cast<f32x4 *>(buf + (block * 4 + 2 * m + 1) * 256 + lane * 4) = f.lo;
    }
}
__device__ __forceinline__ BsFrag bs_fetch(const float *buf, int kb, int m, int lane) {
    return BsFrag{*reinterpret_cast<const f32x4 *>(buf + (kb * 4 + 2 * m) * 256 + lane * 4),
                  *reinterpret_cast<const f32x4 *>(buf + (kb * 4 + 2 * m + 1) * 256 + lane * 4)};
}

// gemm_run of the split tier.  The weight ring is the fp32 kernel's (16 bytes per step and tile): step (kb, s) now
// carries the A operand of MFMA m = s >> 1 as bf16 hi (s even) or lo (s odd).  `act(kb, m)` yields the activation operand.
template <int NKB, int NT, int PFS, int TS, unsigned NORMAL, class Act, unsigned ZERO = 0>
__device__ __forceinline__ void gemm_run_bs(WRing<NT, PFS> &ring, Srd w, unsigned w_lane, f32x16 (&acc)[NT], Act act) {
    BsFrag bfrag{};
#pragma unroll
    for (int step = 0; step < 4 * NKB; ++step) {
        if (step + PFS < 4 * NKB) ring_load<NKB, NT, PFS, TS>(ring, w, w_lane, step + PFS);
        __builtin_amdgcn_sched_barrier(0);
        const bool lo_step = step & 1;
        if (!lo_step) bfrag = act(step >> 2, (step & 3) >> 1);
        const bf16x8 bh = __builtin_bit_cast(bf16x8, bfrag.hi), bl = __builtin_bit_cast(bf16x8, bfrag.lo);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const bf16x8 wv = __builtin_bit_cast(bf16x8, ring.b[step % (PFS + 1)][t]);
            const bool normal = (NORMAL >> t) & 1;
            if (((ZERO >> t) & 1) && step == 0)
                acc[t] = normal ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, wv, f32x16{0}, 0, 0, 0)
                                : __builtin_amdgcn_mfma_f32_32x32x16_bf16(wv, bh, f32x16{0}, 0, 0, 0);
            else
                acc[t] = normal ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, wv, acc[t], 0, 0, 0)
                                : __builtin_amdgcn_mfma_f32_32x32x16_bf16(wv, bh, acc[t], 0, 0, 0);
            if (!lo_step)      // hi weights also meet the activation's lo term
                acc[t] = normal ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl, wv, acc[t], 0, 0, 0)
                                : __builtin_amdgcn_mfma_f32_32x32x16_bf16(wv, bl, acc[t], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// LayerNorm(eps = 1e-5, biased variance) over D features of the row this lane belongs to, given the
// lane's 16 pre-norm values v (features fb + 8s + 4h + j).  Partial (mean, M2) of the wave's 32
// features goes to `stats[row][wave]`; after the barrier every lane merges the W partials (Chan).
template <int D>
__device__ __forceinline__ void layernorm_rows(f32x16 &v, float *stats, const float *gamma, const float *beta,
                                               int wave, int r, int h) {
    constexpr int W = D / 32;
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) s += v[e];
    const float mp = s * (1.0f / 16.0f);
    float m2 = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) m2 = fmaf(v[e] - mp, v[e] - mp, m2);
    // other half-wave holds the row's other 16 features of this block
    const float mo = __shfl_xor(mp, 32), m2o = __shfl_xor(m2, 32);
    const float dlt = mp - mo;
    if (h == 0) *reinterpret_cast<float2 *>(stats + (r * W + wave) * 2) = make_float2(0.5f * (mp + mo), m2 + m2o + 8.0f * dlt * dlt);
    __syncthreads();
    float mean = 0.f, msum = 0.f;
    float pm[W], pM[W];
#pragma unroll
    for (int u = 0; u < W; ++u) {
        const float2 p = *reinterpret_cast<const float2 *>(stats + (r * W + u) * 2);
        pm[u] = p.x;
        pM[u] = p.y;
        mean += p.x;
    }
    mean *= (1.0f / W);
#pragma unroll
    for (int u = 0; u < W; ++u) msum += pM[u] + 32.0f * (pm[u] - mean) * (pm[u] - mean);
    const float rstd = rsqrtf(msum * (1.0f / D) + 1e-5f);
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        const f32x4 g = *reinterpret_cast<const f32x4 *>(gamma + 8 * s4 + 4 * h);
        const f32x4 b = *reinterpret_cast<const f32x4 *>(beta + 8 * s4 + 4 * h);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[4 * s4 + j] = (v[4 * s4 + j] - mean) * rstd * g[j] + b[j];
    }
}

// A GEMM's bias enters as the accumulator's initial value: in the transposed product register e of
// lane half h is feature f0 + (e&3) + 8(e>>2) + 4h, so four 16-byte loads fill the accumulator
// (no LDS copy of the bias vectors, no VALU add in the epilogue, issued long before the first MFMA).
__device__ __forceinline__ f32x16 bias_acc(Srd bias, int f0, int h) {
    f32x16 acc;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const f32x4 b = srd_load(bias, (unsigned)(f0 + 8 * s + 4 * h) * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[4 * s + j] = b[j];
    }
    return acc;
}

#ifdef AFT_DIAG_STAMPS  // diagnostic build only: per-phase s_memtime stamps of wave 0
#define STAMP(i)                                                                                               \
    do {                                                                                                       \
        if (a.stamps && tid == 0) a.stamps[(size_t)tile * 16 + (i)] = __builtin_amdgcn_s_memtime();    \
    } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

// MLP / QKV select the three launch variants at compile time (distinct symbols in a profile):
//   <true,true>  layer l's out-proj+LN1+FFN+LN2 and layer l+1's in-projection   (5 of 7 launches at L=6)
//   <false,true> in-projection only (first layer)      <true,false> last layer, no in-projection
// The body works on the row tiles first_tile, first_tile + tile_stride, ... < tile_end with the S::THREADS threads
// `tid` = 0 .. THREADS-1 of one wave GROUP whose LDS block is `smem`: a whole workgroup in chain_kernel (k_chain.hip), one
// of three groups of a 12-wave workgroup in the plane-resident encoder kernel (k_encoder.hip).  __syncthreads() is the
// only cross-wave synchronisation, so every group of a workgroup must walk the same NUMBER of tiles (tiles past the
// last row are computed on clamped rows and never stored).
template <int D, int ACT, bool MLP, bool QKV, bool BS = false>
__device__ __forceinline__ void chain_body(const ChainArgs &a, float *smem, const int tid_in, const int first_tile,
                                           const int tile_stride, const int tile_end) {
    using S = ChainShape<D>;
    constexpr int W = S::WAVES;
    // laundered: everything derived from the thread index is (re)computed inside this body.  In k_encoder.hip the body
    // sits inside a plane loop and a layer loop; LICM hoisted a dozen lane-dependent offsets to the top of the kernel,
    // where they were spilled at once and reloaded from scratch at every tile start (a scratch reload drains vmcnt).
    int tid = tid_in;
    asm volatile("" : "+v"(tid));
    float *xb = smem;              // x1 then x2, fragment order
    float *hb = xb + S::XB;        // FFN hidden, fragment order
    float *stats = hb + S::HB;     // LayerNorm partials
    float *par = stats + S::ST;    // g1 | be1 | g2 | be2

    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave = feature block = head
    const int r = lane & 31, h = lane >> 5;
    const int fb = 32 * w;                                    // first feature of this wave's block

    const Srd srd_wo = make_srd(a.wo), srd_w1 = make_srd(a.w1), srd_w2 = make_srd(a.w2), srd_wq = make_srd(a.wqkv);
    const Srd srd_attn = make_srd(a.attn), srd_x = make_srd(a.x);
    const Srd srd_bo = make_srd(a.bo), srd_b1 = make_srd(a.b1), srd_b2 = make_srd(a.b2);
    const Srd srd_q = make_srd(a.q), srd_k = make_srd(a.k), srd_vt = make_srd(a.vt);
    const unsigned wo_off = (unsigned)w * W * 1024 + lane * 4, w1_off = (unsigned)(2 * w) * W * 1024 + lane * 4;
    const unsigned w2_off = (unsigned)w * (2 * W) * 1024 + lane * 4;
    const unsigned wq_off = (unsigned)w * W * 1024 + lane * 4;   // tiles w, W+w, 2W+w (stride W tiles)
    const Srd srd_bv = make_srd(a.bv);

    if constexpr (MLP) {
        for (int i = tid; i < D; i += S::THREADS) {   // LayerNorm affine vectors -> LDS, once per workgroup
            par[i] = a.g1[i];
            par[D + i] = a.be1[i];
            par[2 * D + i] = a.g2[i];
            par[3 * D + i] = a.be2[i];
        }
        __syncthreads();
    }
    // fused embedding (<!MLP,QKV> with emb_conv set): linear_1's weights [D][K] and the K input-feature offsets are
    // staged once per workgroup in the upper half of the hidden buffer (idle in this variant: `xq` below alternates
    // between xb and hb's lower half), so nothing tile-invariant is held in registers across the in-projection GEMM.
    constexpr int kEmbSteps = (kMaxPatchFeatures + 6 + 1) / 2;
    float *emb_w = hb + S::XB;                                   // [D][K]
    int *emb_offs = reinterpret_cast<int *>(emb_w + D * (kMaxPatchFeatures + 6));   // [K]: offset into the row's patch block, or ~index into tokens6
    if constexpr (!MLP) {
        if (a.emb_conv != nullptr) {
            const int pk = a.emb_p0 * a.emb_p1;
            for (int i = tid; i < D * a.emb_K; i += S::THREADS) emb_w[i] = a.emb_w1[i];
            for (int k = tid; k < a.emb_K; k += S::THREADS)
                emb_offs[k] = k < pk ? (k / a.emb_p1) * a.emb_T + k % a.emb_p1 : ~(k - pk);
            __syncthreads();
        }
    }
    // persistent workgroups: the grid is sized to the co-resident count and each workgroup walks the
    // row tiles with its stride (no dispatch gaps, no launch tail; LDS buffers need no extra
    // barrier between tiles: every re-write sits >= 1 barrier after the last read of the old data)
    const int ntiles = (a.rows + 31) / 32;
    // Wave priority by ROUNDS LEFT (3, 2, 1, 0 for the last round).  Measured (AFT_STAMPS build): at equal priority the
    // workgroups of one round finish up to 2x apart and the late ones run alone at the end of the launch; a workgroup
    // that gets a round ahead drops its priority so the laggards catch up (chain_last 103 -> 100 us, forward -1.2 %).
    // Tried and rejected: priority bands by quarter tiles (slower), per-CU arrival tickets + rotating priority (the
    // ticket atomic costs 4 us per launch and the finish-time spread did not shrink).
    const int rounds = (tile_end - first_tile + tile_stride - 1) / tile_stride;
    int round = 0;
    // Operands a tile starts from -- the first AFT_CHAIN_PFK feature blocks of the attention output of its 32 rows
    // (straight into operand registers) and the residual rows of x -- are requested one tile AHEAD, just before the
    // previous tile's store epilogue: their latency hides behind those stores, and (vmcnt counts loads and stores in
    // issue order) the first out-projection MFMA does not wait for the previous tile's stores to drain.
#ifndef AFT_CHAIN_PFK
#define AFT_CHAIN_PFK 0
#endif
    constexpr int PFK = MLP ? (AFT_CHAIN_PFK < W ? AFT_CHAIN_PFK : W) : 0;
    constexpr bool AHEAD = AFT_CHAIN_PFK > 0;
    f32x4 of[MLP ? W : 1][4], xres[4];
    auto request_attn = [&](int t, int kb0, int kb1) {
        const unsigned ap = ((unsigned)t * W * 1024 + lane * 4) * 4;
#pragma unroll
        for (int kb = 0; kb < W; ++kb)
#pragma unroll
            for (int s = 0; s < 4; ++s)
                if (kb >= kb0 && kb < kb1) of[kb][s] = srd_load_c(srd_attn, ap, (unsigned)(kb * 1024 + s * 256) * 4);
    };
    auto request_x = [&](int t) {
        if (a.x_blocked) {
            const unsigned xr = ((unsigned)t * 32 * D + w * 1024 + lane * 4) * 4;
#pragma unroll
            for (int s = 0; s < 4; ++s) xres[s] = srd_load(srd_x, xr + 1024 * s);
        } else {
            const unsigned xr = ((unsigned)min(t * 32 + r, a.rows - 1) * D + fb + 4 * h) * 4;
#pragma unroll
            for (int s = 0; s < 4; ++s) xres[s] = srd_load(srd_x, xr + 32 * s);
        }
    };
    auto request_tile = [&](int t) {
        if constexpr (MLP) request_attn(t, 0, PFK);
        if (MLP || a.emb_conv == nullptr) request_x(t);
    };
    if (AHEAD && first_tile < ntiles) request_tile(first_tile);
    // Fused embedding (<!MLP,QKV> with emb_conv set): the operands a tile starts from -- b1 + pos[token] (the accumulator's
    // initial value) and the row's K input features -- are requested ONE TILE AHEAD, before the previous tile's q / k / v^T
    // stores: round 2 issued them at the tile start, where 19 dependent global loads per lane sat in front of the first MFMA
    // (the launch ran 22 us over its 50-us MFMA time).  27 registers carried across the in-projection (111 -> 138 VGPRs).
    f32x16 emb_acc0;
    float emb_bv[kEmbSteps];
    auto emb_request = [&](int t) {
        const int grow_n = min(t * 32 + r, a.rows - 1);
        const int plane = grow_n / a.tokens, tok = grow_n - plane * a.tokens;
        const int tpr = a.emb_T / a.emb_p1, g = tok / tpr, tc = tok - g * tpr;
        const float *cplane = a.emb_conv + ((size_t)plane * a.emb_S + g * a.emb_p0) * a.emb_T + tc * a.emb_p1;
        const float *t6 = a.emb_tok6 ? a.emb_tok6 + ((size_t)(plane >> 1) * a.tokens + tok) * 6 : nullptr;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const f32x4 b = *reinterpret_cast<const f32x4 *>(a.emb_b1 + fb + 8 * s + 4 * h);
            const f32x4 ps = *reinterpret_cast<const f32x4 *>(a.emb_pos + (size_t)tok * D + fb + 8 * s + 4 * h);
#pragma unroll
            for (int j = 0; j < 4; ++j) emb_acc0[4 * s + j] = b[j] + ps[j];
        }
#pragma unroll
        for (int st = 0; st < kEmbSteps; ++st) {
            const int k = 2 * st + h;      // this lane half's k index of step st
            emb_bv[st] = 0.f;
            if (k < a.emb_K) {
                const int off = emb_offs[k];
                emb_bv[st] = off >= 0 ? cplane[off] : t6[~off];
            }
        }
    };
    if constexpr (!MLP) {
        if (a.emb_conv != nullptr && first_tile < tile_end) emb_request(first_tile);
    }
    // linear_2 partials of the previous tile (fused variant of <MLP,!QKV>): sum the W per-wave partials in wave order, add
    // the bias, store [row][out6_stride].  Called by ONE wave per tile, after the barrier that ended that tile.
    int pending_row0 = -1;
    auto reduce_out6 = [&]() {
        const int P = a.out6_features;
        if (pending_row0 + r < a.rows) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {      // features 4h + {0..3} (half 0), 8 + 4h + {0..3} (half 1)
                const int f0 = 8 * half + 4 * h;
                if (f0 < P) {
                    f32x4 sum = *reinterpret_cast<const f32x4 *>(hb + lane * 8 + 4 * half);
#pragma unroll
                    for (int u = 1; u < W; ++u) {
                        const f32x4 t = *reinterpret_cast<const f32x4 *>(hb + (u * 64 + lane) * 8 + 4 * half);
                        sum += t;
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) sum[j] += f0 + j < P ? a.lin2_b[f0 + j] : 0.f;   // padding columns hold 0
                    *reinterpret_cast<f32x4 *>(a.out6 + (size_t)(pending_row0 + r) * a.out6_stride + f0) = sum;
                }
            }
        }
    };
#pragma unroll 1
  for (int tile = first_tile; tile < tile_end; tile += tile_stride, ++round) {
    if constexpr (MLP && !QKV) {
        if (pending_row0 >= 0 && w == (round & (W - 1 < 3 ? W - 1 : 3))) reduce_out6();
    }
#ifndef AFT_NO_PROGRESS_PRIORITY
    {
        const int left = rounds - 1 - round;
        if (left >= 3) __builtin_amdgcn_s_setprio(3);
        else if (left == 2) __builtin_amdgcn_s_setprio(2);
        else if (left == 1) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
    }
#endif
    const int row0 = tile * 32;
    const int grow = min(row0 + r, a.rows - 1);               // clamped: ragged last tile computes, never stores
    const bool row_ok = row0 + r < a.rows;
    const unsigned xrow = ((unsigned)grow * D + fb + 4 * h) * 4;   // byte offset of this lane's 16 features: + 32s + 4j
    // the weight addresses do not depend on the tile: launder the pointers so LICM cannot hoist all
    // 128 KB of this wave's fragment loads out of the tile loop (241 spilled VGPRs when it did)
    // (launder an OFFSET, never a pointer: a laundered pointer loses its address space and the loads
    //  become flat_load, whose out-of-order return forces vmcnt(0)+lgkmcnt(0) waits)
    unsigned lo = 0;
    asm volatile("" : "+v"(lo));
    const unsigned wo_lane = (wo_off + lo) * 4, w1_lane = (w1_off + lo) * 4, w2_lane = (w2_off + lo) * 4,
                   wq_lane = (wq_off + lo) * 4;   // byte offsets into the packed weight blocks
    STAMP(0);
#ifdef AFT_DIAG_STAMPS
    if (a.stamps && tid == 0) {
        a.stamps[(size_t)tile * 16 + 12] = __builtin_amdgcn_s_memrealtime();
        a.stamps[(size_t)tile * 16 + 14] = ((unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) << 32) |
                                           __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));   // HW_ID, XCC_ID
    }
#endif
#ifndef AFT_CHAIN_PFS
#define AFT_CHAIN_PFS 4   // lead of the weight stream in units of 4 MFMAs (256 cycles); 3..8 measured within 1.5 %
#endif
    constexpr int PFD = AFT_CHAIN_PFS, PFF = (AFT_CHAIN_PFS + 1) / 2, PFQ = (AFT_CHAIN_PFS + 2) / 3;
    WRing<1, PFD> ring_d;    // out-proj / FFN-down fragments   (declared per tile: nothing is live across tiles)
    WRing<2, PFF> ring_ff;   // FFN-up fragments
    WRing<3, PFQ> ring_qkv;  // in-projection fragments (q, k, v tiles of head w)
    f32x16 cur;   // this lane's 16 features of the current activation (operand layout)
    if constexpr (MLP) {
        f32x16 acc_o[1] = {bias_acc(srd_bo, fb, h)};
        gemm_preload<W, 1, PFD, 1>(ring_d, srd_wo, wo_lane);
        // attention output of this row tile, all W feature blocks, straight into operand registers
        request_attn(tile, AHEAD ? PFK : 0, W);
        if (!AHEAD) request_x(tile);
        STAMP(1);
        // ---- out-projection (transposed) + bias + residual ----
        if constexpr (BS)
            gemm_run_bs<W, 1, PFD, 1, 0>(ring_d, srd_wo, wo_lane, acc_o, [&](int kb, int m) { return bs_split(of[kb][2 * m], of[kb][2 * m + 1]); });
        else
            gemm_run<W, 1, PFD, 1, 0>(ring_d, srd_wo, wo_lane, acc_o, [&](int kb, int s) { return of[kb][s]; });
        f32x16 acc_h[2] = {bias_acc(srd_b1, 2 * fb, h), bias_acc(srd_b1, 2 * fb + 32, h)};
        gemm_preload<W, 2, PFF, 1>(ring_ff, srd_w1, w1_lane);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) cur[4 * s + j] = acc_o[0][4 * s + j] + xres[s][j];
        STAMP(2);
        layernorm_rows<D>(cur, stats, par + fb, par + D + fb, w, r, h);   // -> x1 (kept: FFN residual)
        STAMP(3);
        if constexpr (BS) {
            bs_publish(xb, w, lane, cur);
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s)
                *reinterpret_cast<f32x4 *>(xb + (w * 4 + s) * 256 + lane * 4) = f32x4{cur[4 * s], cur[4 * s + 1], cur[4 * s + 2], cur[4 * s + 3]};
        }
        __syncthreads();
        STAMP(4);
        // ---- FFN up-projection + activation -> hidden blocks 2w, 2w+1 ----
        if constexpr (BS)
            gemm_run_bs<W, 2, PFF, 1, 0>(ring_ff, srd_w1, w1_lane, acc_h, [&](int kb, int m) { return bs_fetch(xb, kb, m, lane); });
        else
            gemm_run<W, 2, PFF, 1, 0>(ring_ff, srd_w1, w1_lane, acc_h, [&](int kb, int s) {
                return *reinterpret_cast<const f32x4 *>(xb + (kb * 4 + s) * 256 + lane * 4);
            });
        f32x16 acc_d[1] = {bias_acc(srd_b2, fb, h)};
        gemm_preload<2 * W, 1, PFD, 1>(ring_d, srd_w2, w2_lane);
        STAMP(5);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x16 hid;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const f32x2 g0 = activate2<ACT>(f32x2{acc_h[t][4 * s], acc_h[t][4 * s + 1]});
                const f32x2 g1 = activate2<ACT>(f32x2{acc_h[t][4 * s + 2], acc_h[t][4 * s + 3]});
                if constexpr (BS) {
                    hid[4 * s] = g0[0]; hid[4 * s + 1] = g0[1]; hid[4 * s + 2] = g1[0]; hid[4 * s + 3] = g1[1];
                } else {
                    *reinterpret_cast<f32x4 *>(hb + ((2 * w + t) * 4 + s) * 256 + lane * 4) = f32x4{g0[0], g0[1], g1[0], g1[1]};
                }
            }
            if constexpr (BS) bs_publish(hb, 2 * w + t, lane, hid);
        }
        __syncthreads();
        STAMP(6);
        // ---- FFN down-projection (transposed) + bias + residual(x1, registers) ----
        if constexpr (BS)
            gemm_run_bs<2 * W, 1, PFD, 1, 0>(ring_d, srd_w2, w2_lane, acc_d, [&](int kb, int m) { return bs_fetch(hb, kb, m, lane); });
        else
            gemm_run<2 * W, 1, PFD, 1, 0>(ring_d, srd_w2, w2_lane, acc_d, [&](int kb, int s) {
                return *reinterpret_cast<const f32x4 *>(hb + (kb * 4 + s) * 256 + lane * 4);
            });
        if constexpr (QKV) gemm_preload<W, 3, PFQ, W>(ring_qkv, srd_wq, wq_lane);
#pragma unroll
        for (int e = 0; e < 16; ++e) cur[e] += acc_d[0][e];
        STAMP(7);
        layernorm_rows<D>(cur, stats, par + 2 * D + fb, par + 3 * D + fb, w, r, h);   // -> x2
        STAMP(8);
        if constexpr (!QKV) {
            if (AHEAD && tile + tile_stride < ntiles) request_tile(tile + tile_stride);
        }
        bool store_x = true;
        if constexpr (!QKV) {
            if (a.out6 != nullptr) {
                // linear_2 as one more transposed product: A = W2 rows (lane = output feature p < P, zero above),
                // B = x2 of this wave's 32-feature block (registers, operand layout) -> this wave's partial of
                // out6^T [p][row]; the W partials meet in the idle hidden buffer and are summed in wave order.
                store_x = false;
                const int P = a.out6_features;
                f32x16 acc6 = f32x16{0};
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    f32x4 wv = {0.f, 0.f, 0.f, 0.f};
                    if (r < P) wv = *reinterpret_cast<const f32x4 *>(a.lin2_w + (size_t)r * D + fb + 8 * s + 4 * h);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc6 = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[j], cur[4 * s + j], acc6, 0, 0, 0);
                }
                // accumulator: lane = row r, register e = output feature (e&3) + 8(e>>2) + 4h; P <= 16 -> registers 0..7.
                // The partials are summed one tile LATER (reduce_out6 at the top of the loop / after it), behind the
                // barrier that ends this tile anyway: no extra barrier, and the waves take turns as the reducer.
                float *part = hb + (w * 64 + lane) * 8;
                *reinterpret_cast<f32x4 *>(part) = f32x4{acc6[0], acc6[1], acc6[2], acc6[3]};
                *reinterpret_cast<f32x4 *>(part + 4) = f32x4{acc6[4], acc6[5], acc6[6], acc6[7]};
                pending_row0 = row0;
            }
        }
        if (store_x && row_ok) {
            if (a.x_blocked) {
                const unsigned xblk = ((unsigned)row0 * D + w * 1024 + lane * 4) * 4;
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    srd_store(srd_x, xblk + 1024 * s, f32x4{cur[4 * s], cur[4 * s + 1], cur[4 * s + 2], cur[4 * s + 3]});
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    srd_store(srd_x, xrow + 32 * s, f32x4{cur[4 * s], cur[4 * s + 1], cur[4 * s + 2], cur[4 * s + 3]});
            }
        }
    } else {
        gemm_preload<W, 3, PFQ, W>(ring_qkv, srd_wq, wq_lane);
        if (a.emb_conv == nullptr) {
            if (!AHEAD) request_x(tile);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int j = 0; j < 4; ++j) cur[4 * s + j] = xres[s][j];
        } else {
            // x0 = [patch features | adapter features] W1^T + b1 + pos[token]: ceil(K/2) MFMAs of the same transposed
            // form (A = W1 rows of this wave's feature block, B = the row's input features), the accumulator starting
            // from b1 + pos -- it comes out in operand layout like any other `cur`.
            f32x16 acc0 = emb_acc0;
            float av[kEmbSteps];
#pragma unroll
            for (int st = 0; st < kEmbSteps; ++st) {
                const int k = 2 * st + h;
                av[st] = k < a.emb_K ? emb_w[(fb + r) * a.emb_K + k] : 0.f;
            }
#pragma unroll
            for (int st = 0; st < kEmbSteps; ++st)
                if (2 * st < a.emb_K) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[st], emb_bv[st], acc0, 0, 0, 0);
            cur = acc0;
            if (row_ok) {
                if (a.x_blocked) {
                    const unsigned xblk = ((unsigned)row0 * D + w * 1024 + lane * 4) * 4;
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        srd_store(srd_x, xblk + 1024 * s, f32x4{cur[4 * s], cur[4 * s + 1], cur[4 * s + 2], cur[4 * s + 3]});
                } else {
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        srd_store(srd_x, xrow + 32 * s, f32x4{cur[4 * s], cur[4 * s + 1], cur[4 * s + 2], cur[4 * s + 3]});
                }
            }
        }
    }

    if constexpr (QKV) {
        // V tile: lane = feature (fetched before the barrier wait; through an SRD: a 64-bit per-lane pointer here was one
        // of the values the register allocator spilled, and a scratch reload drains vmcnt -- i.e. waited for the x stores)
        const float bias_v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd_bv, (unsigned)(fb + r) * 4, 0, 0));
        // publish x2 (or x0) for the in-projection.  x1 readers are all past the hidden-exchange barrier.
        // The QKV-only variant has one barrier per tile, so it alternates between two exchange buffers
        // (xb / the idle hidden buffer): a re-write then sits two barriers behind the last read.
        float *xq = xb;
        if constexpr (!MLP) xq = (round & 1) ? hb : xb;
        if constexpr (BS) {
            bs_publish(xq, w, lane, cur);
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s)
                *reinterpret_cast<f32x4 *>(xq + (w * 4 + s) * 256 + lane * 4) = f32x4{cur[4 * s], cur[4 * s + 1], cur[4 * s + 2], cur[4 * s + 3]};
        }
        __syncthreads();
        STAMP(9);
        f32x16 vinit;
#pragma unroll
        for (int e = 0; e < 16; ++e) vinit[e] = bias_v;
        f32x16 acc[3];
        acc[2] = vinit;
        if constexpr (BS) {
            // split tier: the attention kernel receives q as bf16 pairs, so the query bias (the accumulator's initial value)
            // and the softmax scale are applied here, before the split
            acc[0] = bias_acc(make_srd(a.bv - 2 * D), fb, h);
            auto xq_bs = [&](int kb, int m) { return bs_fetch(xq, kb, m, lane); };
            gemm_run_bs<W, 3, PFQ, W, 0x4, decltype(xq_bs), 0x2>(ring_qkv, srd_wq, wq_lane, acc, xq_bs);
        } else {
            auto xq_frag = [&](int kb, int s) { return *reinterpret_cast<const f32x4 *>(xq + (kb * 4 + s) * 256 + lane * 4); };
            gemm_run<W, 3, PFQ, W, 0x4, decltype(xq_frag), 0x3>(ring_qkv, srd_wq, wq_lane, acc, xq_frag);   // q, k start from 0
        }
        STAMP(10);
        if (AHEAD && tile + tile_stride < ntiles) request_tile(tile + tile_stride);   // before this tile's stores
        if constexpr (!MLP) {
            if (a.emb_conv != nullptr && tile + tile_stride < tile_end) emb_request(tile + tile_stride);
        }
        // ---- epilogue: q, k, v of head w for 32 token rows, written in MFMA-FRAGMENT order so that
        // k_attn.hip reads every operand with fully coalesced 1-KB loads:
        //   q, k : [plane*H + head][key tile][s][lane = key%32 + 32*hh][4]   value (key, d = 8s + 4hh + j)
        //   vt   : [plane*H + head][key tile][g][lane = d + 32*hh][4]        value (d, key = 32kt + 8g + 4hh + j)
        // q/k tiles (lane = token row, registers = features): registers 4s..4s+3 of half hh are one
        // 16-byte fragment element; the v tile (lane = feature, registers = tokens) likewise.
        const int plane0 = row0 / a.tokens, tok0 = row0 - plane0 * a.tokens;
        const unsigned head_stride = (unsigned)a.tokpad * kHeadDim;             // floats per (plane, head)
        // token index past the end of its plane -> the next plane's (plane, head) block.  A 32-row tile crosses at most one
        // plane boundary when tokens >= 32; grids with fewer tokens (a whole plane inside a tile) walk on
        auto wrap_plane = [&](int &tok, unsigned &ph) {
            if (tok >= a.tokens) { tok -= a.tokens; ph += a.heads; }
            if (a.tokens < kTile)
                while (tok >= a.tokens) { tok -= a.tokens; ph += a.heads; }
        };
        const unsigned ph0 = (unsigned)(plane0 * a.heads + w);
        const bool full = row0 + 32 <= a.rows;
        if constexpr (BS) {
            // split tier: q / k / v^T leave as bf16 hi / lo fragments.  Slot 2m + term of a tile replaces slot s: the lane's
            // registers 8m .. 8m+7 are the 8 k-values of the attention kernel's MFMA m (k = head feature for q / k, key for
            // v^T); same bytes, same addresses as the fp32 fragments.
            const float qscale = 1.4426950408889634f * 0.17677669529663687f;   // log2(e) / sqrt(32)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[0][e] *= qscale;
            {
                int tok = tok0 + r;
                unsigned ph = ph0;
                wrap_plane(tok, ph);
                const unsigned lane_off = (ph * head_stride + (unsigned)(tok >> 5) * 1024 + ((tok & 31) + 32 * h) * 4) * 4;
                if (full || row_ok) {
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int m = 0; m < 2; ++m) {
                            const BsFrag f = bs_split(f32x4{acc[t][8 * m], acc[t][8 * m + 1], acc[t][8 * m + 2], acc[t][8 * m + 3]},
                                                      f32x4{acc[t][8 * m + 4], acc[t][8 * m + 5], acc[t][8 * m + 6], acc[t][8 * m + 7]});
                            srd_store(t == 0 ? srd_q : srd_k, lane_off + (2 * m) * 1024, f.hi);
                            srd_store(t == 0 ? srd_q : srd_k, lane_off + (2 * m + 1) * 1024, f.lo);
                        }
                }
            }
            // v^T: lane = head feature, registers 4gq .. 4gq+3 = tokens tok0 + 8gq + 4h + {0..3} = one half (8 bytes) of the
            // 16-byte element (key tile, slot 2m + term) with m = (8-key group within the tile) >> 1
            __bf16 *vt16 = reinterpret_cast<__bf16 *>(a.vt);
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                int tok = tok0 + 8 * gq + 4 * h;
                unsigned ph = ph0;
                wrap_plane(tok, ph);
                __bf16 vh[4], vl[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    vh[j] = (__bf16)acc[2][4 * gq + j];
                    vl[j] = (__bf16)(acc[2][4 * gq + j] - (float)vh[j]);
                }
                const bool in_rows = full || row0 + 8 * gq + 4 * h + 3 < a.rows;
                if ((tok & 3) == 0 && tok + 3 < a.tokens && in_rows) {
                    // the four tokens are keys 32kt + 8g + 4hh + {0..3} of one plane: element (kt, slot 2(g>>1) + term), lane
                    // (feature, hh), bf16 positions 4(g&1) .. 4(g&1)+3
                    const unsigned g = (tok >> 3) & 3, hh = (tok >> 2) & 1;
                    const size_t e16 = ((size_t)ph * head_stride + (size_t)(tok >> 5) * 1024 + (2 * (g >> 1)) * 256 + (r + 32 * hh) * 4) * 2 + 4 * (g & 1);
                    using bf16x4 = __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16;
                    *reinterpret_cast<bf16x4 *>(vt16 + e16) = bf16x4{vh[0], vh[1], vh[2], vh[3]};
                    *reinterpret_cast<bf16x4 *>(vt16 + e16 + 512) = bf16x4{vl[0], vl[1], vl[2], vl[3]};   // next slot: + 256 floats
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        int tj = tok + j;
                        unsigned pj = ph;
                        wrap_plane(tj, pj);
                        if (row0 + 8 * gq + 4 * h + j < a.rows) {
                            const unsigned g = (tj >> 3) & 3, hh = (tj >> 2) & 1;
                            const size_t e16 = ((size_t)pj * head_stride + (size_t)(tj >> 5) * 1024 + (2 * (g >> 1)) * 256 + (r + 32 * hh) * 4) * 2 +
                                               4 * (g & 1) + (tj & 3);
                            vt16[e16] = vh[j];
                            vt16[e16 + 512] = vl[j];
                        }
                    }
                }
            }
        } else
        if (full && tok0 + 32 <= a.tokens && (a.tokens & 7) == 0) {
            // the common case -- the tile lies inside one plane and 8 divides the token count (tok0 is then a multiple of
            // 8): every V^T group of four tokens is one whole fragment element at lane-linear offset, and all the block
            // indices are wave-uniform (scalar offset operand); only the q / k row index needs per-lane arithmetic
            const int tok = tok0 + r;
            const unsigned ph_base = ph0 * head_stride;
            const unsigned lane_off = (ph_base + (unsigned)(tok >> 5) * 1024 + ((tok & 31) + 32 * h) * 4) * 4;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                srd_store(srd_q, lane_off + s * 1024, f32x4{acc[0][4 * s], acc[0][4 * s + 1], acc[0][4 * s + 2], acc[0][4 * s + 3]});
                srd_store(srd_k, lane_off + s * 1024, f32x4{acc[1][4 * s], acc[1][4 * s + 1], acc[1][4 * s + 2], acc[1][4 * s + 3]});
            }
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const unsigned kb8 = (unsigned)(tok0 >> 3) + gq;     // 8-key group of tokens tok0 + 8 gq + 4h + {0..3}
                const f32x4 v = {acc[2][4 * gq], acc[2][4 * gq + 1], acc[2][4 * gq + 2], acc[2][4 * gq + 3]};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(decltype(__builtin_amdgcn_raw_buffer_load_b128(srd_vt, 0, 0, 0)), v),
                                                       srd_vt, lane * 16, (ph_base + (kb8 >> 2) * 1024 + (kb8 & 3) * 256) * 4, 0);
            }
        } else {
        {
            int tok = tok0 + r;
            unsigned ph = ph0;
            wrap_plane(tok, ph);
            const unsigned lane_off = ph * head_stride + (unsigned)(tok >> 5) * 1024 + ((tok & 31) + 32 * h) * 4;
            if (full || row_ok) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        srd_store(t == 0 ? srd_q : srd_k, (lane_off + s * 256) * 4,
                                  f32x4{acc[t][4 * s], acc[t][4 * s + 1], acc[t][4 * s + 2], acc[t][4 * s + 3]});
                }
            }
        }
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {   // registers 4gq..4gq+3 = tokens tok0 + 8gq + 4h + {0..3}
            int tok = tok0 + 8 * gq + 4 * h;
            unsigned ph = ph0;
            wrap_plane(tok, ph);
            const f32x4 v = {acc[2][4 * gq], acc[2][4 * gq + 1], acc[2][4 * gq + 2], acc[2][4 * gq + 3]};
            const bool in_rows = full || row0 + 8 * gq + 4 * h + 3 < a.rows;
            if ((tok & 3) == 0 && tok + 3 < a.tokens && in_rows) {
                // the four tokens are one fragment element (key = 32kt + 8g + 4hh + j) of one plane
                const unsigned off = ph * head_stride + (unsigned)(tok >> 5) * 1024 + ((tok >> 3) & 3) * 256 +
                                     (r + 32 * ((tok >> 2) & 1)) * 4;
                srd_store(srd_vt, off * 4, v);
            } else {   // token counts that are not a multiple of 4, plane boundaries, the last partial row tile
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int tj = tok + j;
                    unsigned pj = ph;
                    wrap_plane(tj, pj);
                    if (row0 + 8 * gq + 4 * h + j < a.rows)
                        a.vt[pj * head_stride + (unsigned)(tj >> 5) * 1024 + ((tj >> 3) & 3) * 256 +
                             (r + 32 * ((tj >> 2) & 1)) * 4 + (tj & 3)] = v[j];
                }
            }
        }
        }   // general epilogue
    }
    STAMP(11);
#ifdef AFT_DIAG_STAMPS
    if (a.stamps && tid == 0) a.stamps[(size_t)tile * 16 + 13] = __builtin_amdgcn_s_memrealtime();
#endif
    // cross-tile LDS hazards of the persistent loop.  <MLP,QKV>: every re-write sits behind a barrier
    // that follows the last read (xb: LN1 barrier of the next tile; stats: the x2-exchange barrier; hb:
    // two barriers).  <!MLP,QKV> double-buffers its exchange (above).  <MLP,!QKV> has no x2-exchange
    // barrier in front of the next tile's LN1 partials: one more barrier.
    if constexpr (MLP && !QKV) __syncthreads();
  }
    if constexpr (MLP && !QKV) {
        if (pending_row0 >= 0 && w == 0) reduce_out6();   // the last tile's partials (behind the loop's closing barrier)
    }
}

}  // namespace aft
