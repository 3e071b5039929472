// attn_device.h (device code of k_attn.hip, shared with k_encoder.hip) -- multi-head self-attention core on gfx950 fp32 MFMA.
//
// Reference semantics: the scaled-dot-product part of nn.MultiheadAttention inside
// nn.TransformerEncoderLayer (constructed at reference src/models/blocks/encoders.py:44-55):
// per (plane, head): P = softmax(Q K^T / sqrt(dh)) over keys, O = P V, no mask, eval mode
// (attention dropout = identity); heads are contiguous 32-wide column slices of the packed
// in-projection.  q/k/vt come from k_chain.hip's QKV epilogue.
//
// MI355X mapping
//   * one WAVE owns one (plane, head, 32-query tile) task end to end; no LDS, no barriers:
//     at 64 cycles per v_mfma_f32_32x32x2_f32 the matrix pipe needs only 16 operand bytes per
//     lane per 256 cycles, which L1/L2 deliver directly (K and V^T of one head = 72 KB, hot in
//     the XCD's L2 for all 9 query tiles of that head).  q/k/vt are stored by the producer in
//     fragment order, so each operand load is 1 KB contiguous per wave (16 TA accesses, not 64).
//   * "swapped" products so the softmax row lives in ONE lane:  S^T = K Q^T  (A = K tile,
//     B = Q^T) leaves lane (q, h) holding 16 of the 32 keys of query q per tile -> row max /
//     row sum are register reductions + one cross-half exchange; then O^T = V^T P^T takes the
//     probability registers *as they are* for the B operand (register r of half h is key
//     8*(r>>2) + (r&3) + 4h, which is exactly the k index lane half h must supply when the
//     A operand is loaded as V^T[d][8g + 4h + j]) -- P never moves between lanes or to LDS.
//   * online softmax over chunks of CH key tiles (exact running max / sum in fp32); logits are
//     pre-scaled by log2(e)/sqrt(dh) through Q so the exponential is a bare v_exp_f32.
//   * tokens = 280 is 8.75 tiles: keys 280..287 are masked to -inf / V^T columns zeroed in
//     registers, query rows >= tokens are computed and dropped at the store.
#pragma once
#include <math.h>

#include <type_traits>

#include "aft_internal.h"
#include "srd.h"

namespace aft {

#ifndef AFT_ATTN_WAVES
#define AFT_ATTN_WAVES 3   // waves per SIMD the register budget is capped for (and the persistent grid sized to)
#endif

// Growth of a row's maximum (in log2 units) over the reference that is tolerated before the accumulators are rescaled:
// the probabilities are exp2(s - m_ref) with a STALE reference m_ref, so they may reach 2^kRescaleThreshold instead
// of 1.  In fp32 that costs no precision (only the exponent moves); 2^64 x 1120 keys x |v| is far from overflow.
constexpr float kRescaleThreshold = 64.0f;
// |row maximum| of the first key tile below which a wave keeps the reference at ZERO for all its rows (no subtraction
// at all); trained encoders live here (|logit| of a few units)
constexpr float kZeroRefThreshold = 32.0f;

__device__ __forceinline__ float other_half(float x) {   // value held by lane (l ^ 32): v_permlane32_swap
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, (threadIdx.x & 32) ? r[0] : r[1]);
}

__device__ __forceinline__ float max16(const f32x16 &v) {   // 8 x v_max3_f32
    float m = __builtin_fmaxf(v[0], v[1]);
#pragma unroll
    for (int e = 2; e < 16; e += 2) m = __builtin_fmaxf(__builtin_fmaxf(m, v[e]), v[e + 1]);
    return m;
}

// S^T tile = K tile . Q^T - m_ref: 16 MFMAs on one accumulator that starts from the inline constant 0.  A non-zero
// reference enters as ONE more MFMA of the same chain (A = 1 on the k = 0 half, B = -m_ref of the lane's query), so
// no register tuple of initial values and no copies are needed; `zero_ref` (wave-uniform) skips it.
// NMF = MFMAs of the chain: 16 per 32-feature block; 8 for a 16-feature head, whose operands arrive in slots 0, 1 (round 5)
template <int NB, int NMF = 16 * NB>
__device__ __forceinline__ f32x16 qk_tile(const f32x4 (&kreg)[NB][4], const f32x4 (&qreg)[NB][4], bool zero_ref, float a_one,
                                          float neg_m) {
    f32x16 c;
    if (zero_ref) {
        c = mfma_f32(kreg[0][0][0], qreg[0][0][0], f32x16{0});
    } else {
        c = mfma_f32(a_one, neg_m, f32x16{0});
        c = mfma_f32(kreg[0][0][0], qreg[0][0][0], c);
    }
#pragma unroll
    for (int i = 1; i < NMF; ++i) c = mfma_f32(kreg[i >> 4][(i >> 2) & 3][i & 3], qreg[i >> 4][(i >> 2) & 3][i & 3], c);
    return c;
}

template <int NB>
struct AttnRow {          // per-lane softmax state of the lane's query row
    float m_ref;          // reference maximum (log2 units); logits are produced as s - m_ref
    f32x2 lsum2;          // partial row sums of this lane's 16 keys per tile (two interleaved chains)
    f32x16 oacc[NB];      // O^T accumulator, one per 32-feature block of the head
    bool zero_ref;        // wave-uniform: m_ref == 0 in every lane
};

// launch bound (256, 3): <= 168 registers keeps accumulators in VGPRs (the MFMA's VGPR form) --
// with the default 512-register budget hipcc parks them in AGPRs and pays ~2.5 v_accvgpr moves
// per MFMA around the softmax, which on the fp32 matrix path comes straight out of MFMA time.
//
// Softmax with a stale reference maximum (round 2).  A VALU instruction costs 2.6-4.3 cycles of fp32-MFMA time
// and v_exp_f32 8 (tools/micro/valu_cost.hip) and both kernels of the encoder are ALU-bound (MFMA + VALU cycles
// fill ~95 % of the SIMD time), so the per-tile VALU work is cut to what the arithmetic needs:
//   * m_ref = row maximum of the FIRST key tile (or 0 for the whole wave when all those maxima are small).  Every
//     later tile comes out of its MFMA chain as S - m_ref, is exponentiated with a bare v_exp_f32 and summed in
//     per-lane partial sums; no running-max update, no alpha, no O rescale, no register copies per tile.
//   * one wave-uniform test per tile (8 x v_max3 + compare): only when some row's tile maximum exceeds m_ref by
//     more than kRescaleThreshold are m_ref, O, the partial sums and the pending logits rescaled (exact: every
//     factor is exp2 of the change of reference).  Rows that did not grow get a factor of 1.
//   * QK^T of tile t+1 is issued before the exponentials of tile t; two named accumulators alternate roles
//     (the loop is unrolled by two) so neither is ever copied.
//   * the next task's Q / K tile 0 / V tile 0 are requested during the last key tile of the current one.
// `tail_task` (round 5): the launcher may hand the tasks beyond the last WHOLE round (ntasks then counts only the whole rounds) out
// itself, one per wave at most, so that a partial last round is spread over all XCDs and CUs instead of filling half the chip for a
// full round (k_attn.hip); -1 = none.  Which wave works a task never changes the task's arithmetic: same bits.
// One WAVE walks the tasks first_task, first_task + total_waves, ... < ntasks (task = (plane * heads + head) * nkt + query
// tile): attn_kernel (k_attn.hip) deals them over a persistent grid, the plane-resident encoder kernel (k_encoder.hip)
// over the 12 waves of the workgroup that owns the plane.  No LDS, no barriers.
// S^T tile of the split-precision tier: K and Q^T arrive as bf16 hi / lo fragments (slot 2m + term of a tile = MFMA m's
// 8 k-values, written by the chain kernel's in-projection epilogue with the query bias and the softmax scale already
// applied): hi.hi + hi.lo + lo.hi per MFMA m on v_mfma_f32_32x32x16_bf16.  The reference enters as one more MFMA whose
// A operand is 1 at k = 0 and whose B operand is -m_ref there: m_ref is kept bf16-representable so that this is exact.
__device__ __forceinline__ f32x16 qk_tile_bs(const f32x4 (&kreg)[4], const f32x4 (&qreg)[4], bool zero_ref, float neg_m, int h) {
    f32x16 c = f32x16{0};
    if (!zero_ref) {
        bf16x8 one = {}, mref = {};
        one[0] = (__bf16)(h == 0 ? 1.0f : 0.0f);
        mref[0] = (__bf16)(h == 0 ? neg_m : 0.0f);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(one, mref, c, 0, 0, 0);
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const bf16x8 kh = __builtin_bit_cast(bf16x8, kreg[2 * m]), kl = __builtin_bit_cast(bf16x8, kreg[2 * m + 1]);
        const bf16x8 qh = __builtin_bit_cast(bf16x8, qreg[2 * m]), ql = __builtin_bit_cast(bf16x8, qreg[2 * m + 1]);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qh, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, ql, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qh, c, 0, 0, 0);
    }
    return c;
}
__device__ __forceinline__ float round_to_bf16(float x) { return (float)(__bf16)x; }

// HD = head dimension.  q / k / v^T and the output are laid out per 32-FEATURE BLOCK of the model dimension (`nblk` = model_dim / 32
// blocks per plane), whatever the head count: for HD = 32 a block is a head.  HD = 64: a head is two adjacent blocks -- S^T sums
// both blocks' products in one 32-MFMA chain, O^T is one accumulator per block.  HD = 16: a block holds two heads -- a task takes
// ONE of them (`sub`): S^T is an 8-MFMA chain over this head's two 8-feature slots of Q and K (round 5; round 4 zeroed the other head's
// query features and ran the whole 16-MFMA chain), O^T is computed for the whole block and only this head's 16 rows are stored (the
// value product still does twice the work it needs: a 32x32 MFMA cannot take half its rows).  nn.MultiheadAttention as built at reference blocks/encoders.py:44-51 accepts any
// num_head that divides model_dim (schemas.py:124-127).
// Every other multiple of 8 (8, 24, 40, 48: what 4 or 8 heads give at model_dim 96 / 160 / 192, or 16 heads at 128; late round 5): a
// head is a run of HD / 8 of the plane's 8-FEATURE GROUPS (the slots of the fragment layout) that starts wherever the heads before
// it end -- anywhere in a block.  S^T contracts exactly the head's features: one Q / K slot per group, fetched from whichever block
// holds it, 4 MFMAs each.  O^T is computed for the one or two whole blocks the head touches (V^T's rows cannot be picked apart in a
// 32-row MFMA operand) and only the head's groups are stored: exact logits work, 1.3-4 x the value work -- covered, not tuned.
// HD = 56 could straddle three blocks (three accumulators do not fit beside the operands) and head dims that are not multiples of 8
// would split a fragment slot: refused by aft_check_config.
// TOK > 0: the token count as a compile-time constant (the default grid's 280: nine key tiles, the last one ragged) -- the tile loop's
// trip count, the "last two or three tiles" logic and the padding masks resolve at compile time.  TOK = 0: any count at run time.
template <bool BS = false, int HD = 32, int TOK = 0>
__device__ __forceinline__ void attn_body(const float *__restrict__ q, const float *__restrict__ k,
                                          const float *__restrict__ vt, const float *__restrict__ qbias,
                                          float *__restrict__ out, int nblk, int tokens_rt, int tokpad_rt, int model_dim,
                                          float scale_log2e, const int first_task, const int total_waves, int ntasks,
                                          unsigned long long *stamps, const int tail_task = -1) {
    int lane_l = threadIdx.x & 63;
    asm volatile("" : "+v"(lane_l));   // laundered: lane-dependent offsets are recomputed per call, not hoisted out of the caller's loops
    const int lane = lane_l;
    const Srd qs = make_srd(q), ks = make_srd(k), vs = make_srd(vt), os = make_srd(out);
    static_assert(HD % 8 == 0 && HD >= 8 && HD <= 64 && HD != 56, "head dimension");
    static_assert(!BS || HD == 32, "the split-precision tier is instantiated for head dimension 32");
    constexpr bool GEN = HD != 16 && HD != 32 && HD != 64;   // a head = HD / 8 feature groups starting anywhere in a block
    constexpr int NS = HD / 8;                          // 8-feature groups (fragment slots) per head
    constexpr int NB = GEN ? (32 % HD == 0 ? 1 : 2) : HD == 64 ? 2 : 1;   // 32-feature blocks a head touches (V^T tiles, O^T accumulators)
    constexpr int NQ = GEN ? (NS + 3) / 4 : NB;         // Q / K operand registers, in units of four slots
    const int heads = model_dim / HD;
    const int tokens = TOK > 0 ? TOK : tokens_rt;
    const int tokpad = TOK > 0 ? (TOK + kTile - 1) / kTile * kTile : tokpad_rt;
    const int nkt = tokpad / kTile;
    const int r = lane & 31, h = lane >> 5;
    const bool ragged = (tokens & (kTile - 1)) != 0;   // last key tile holds padded keys
    const float a_one = h == 0 ? 1.0f : 0.0f;           // A operand of the reference MFMA: 1 on the k = 0 half
#ifdef AFT_DIAG_STAMPS
#define ASTAMP(i) do { if (stamps && lane == 0) stamps[(size_t)task * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define ASTAMP(i) do { } while (0)
#endif
    // q, k, vt arrive in MFMA-fragment order from k_chain.hip: [ph][tile][s or g][lane][4] -> every
    // operand load below is one fully coalesced 1-KB buffer_load_dwordx4 per wave
    // (plane, block) index of a task's first block: task / nkt is (plane, head)
    // GEN: first group of the task's head, counted over the whole launch (plane * nblk * 4 + head * NS): group G lies in (plane,
    // block) G >> 2, slot G & 3
    auto first_group = [&](int task) { const int ph = task / nkt, plane = ph / heads; return plane * nblk * 4 + (ph - plane * heads) * NS; };
    auto first_block = [&](int task) {
        if constexpr (GEN) return first_group(task) >> 2;
        const int ph = task / nkt;
        return HD == 16 ? ph >> 1 : ph * NB;
    };
    auto head_base = [&](int task) { return ((unsigned)first_block(task) * tokpad * kHeadDim + lane * 4) * 4; };
    const unsigned blk_bytes = (unsigned)tokpad * kHeadDim * 4;     // one (plane, block) of q / k / v^T
    // GEN: a head that ends inside its first block, in the plane's last block, has no second block: the second V^T tile re-reads the
    // first (its product is computed and dropped at the store) rather than running past the plane
    auto second_block_bytes = [&](int task) {
        if constexpr (GEN && NB == 2) return ((first_group(task) + NS - 1) >> 2) != first_block(task) ? blk_bytes : 0u;
        return blk_bytes;
    };
    auto load_tile = [&](Srd src, int kt, f32x4 (&dst)[NB][4], unsigned base, unsigned b1_bytes) {
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int s = 0; s < 4; ++s) dst[b][s] = srd_load(src, base + b * b1_bytes + (unsigned)(kt * 1024 + s * 256) * 4);
    };
    // HD = 16: only the head's two slots (8-feature groups 2 sub, 2 sub + 1) of a Q / K tile, into slots 0, 1
    constexpr int NMF = GEN ? 4 * NS : HD == 16 ? 8 : 16 * NB;
    // sub_: HD = 16: which of the block's two heads.  GEN: the head's first slot within its first block (first_group & 3); `base` is
    // that block's, and slot i of the head is slot (sub_ + i) & 3 of block (sub_ + i) >> 2 from there
    auto load_qk = [&](Srd src, int kt, f32x4 (&dst)[NQ][4], unsigned base, int sub_) {
        if constexpr (GEN) {
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const int sl = sub_ + i;
                dst[i >> 2][i & 3] = srd_load(src, base + (unsigned)(sl >> 2) * blk_bytes + (unsigned)(kt * 1024 + (sl & 3) * 256) * 4);
            }
        } else if constexpr (HD == 16) {
#pragma unroll
            for (int s = 0; s < 2; ++s) dst[0][s] = srd_load(src, base + (unsigned)(kt * 1024 + (2 * sub_ + s) * 256) * 4);
        } else {
            load_tile(src, kt, dst, base, blk_bytes);
        }
    };
    auto sub_of = [&](int task) {
        if constexpr (GEN) return first_group(task) & 3;
        return HD == 16 ? (task / nkt) & 1 : 0;
    };
    // padded keys of the ragged last tile: logits -> -inf (probability 0), V^T columns -> 0 (the workspace pad is
    // never trusted: 0 x NaN would poison the row)
    // (8 | tokens, the usual case: whole 8-key groups are padding, the same registers in every lane -- a wave-uniform test per
    // group instead of an add, a compare and a select per register and lane: 100 vector instructions per task less)
#ifdef AFT_ATTN_OLD_MASKS
    const bool pad8 = false;
#else
    const bool pad8 = (tokens & 7) == 0;
#endif
    auto mask_logits = [&](f32x16 &sv, int kt) {
        if (pad8) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                if (kt * kTile + 8 * g >= tokens) sv[4 * g] = sv[4 * g + 1] = sv[4 * g + 2] = sv[4 * g + 3] = -INFINITY;
            return;
        }
#pragma unroll
        for (int e = 0; e < 16; ++e)
            if (kt * kTile + (e & 3) + 8 * (e >> 2) + 4 * h >= tokens) sv[e] = -INFINITY;
    };
    auto mask_values = [&](f32x4 (&vvb)[NB][4], int kt) {
      if (!BS && pad8) return;    // padded 8-key groups are skipped as a whole by the O^T loop: their V^T columns are never read
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        f32x4 (&vv)[4] = vvb[b];
        if constexpr (BS) {   // slot 2m + term, element j: key 16 m + 8 (j >> 2) + 4 h + (j & 3) of the tile
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) {
                bf16x8 e = __builtin_bit_cast(bf16x8, vv[sl]);
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (kt * kTile + 16 * (sl >> 1) + 8 * (j >> 2) + 4 * h + (j & 3) >= tokens) e[j] = (__bf16)0.0f;
                vv[sl] = __builtin_bit_cast(f32x4, e);
            }
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (kt * kTile + 8 * g + 4 * h + j >= tokens) vv[g][j] = 0.f;
        }
      }
    };

  // wave priority by work left (set_progress_priority, aft_internal.h): keeps the waves of a SIMD abreast
  const int rounds = (ntasks + total_waves - 1) / total_waves + (tail_task >= 0 ? 1 : 0);
  int round = 0;
  int task = first_task < ntasks ? first_task : tail_task;     // -1: nothing to do
  f32x4 qreg[NQ][4], kcur[NQ][4], vcur[NB][4];
  if (task >= 0) {           // operands of the first task; later ones are requested during the previous task's last tile
      const unsigned hb0 = head_base(task);
      load_qk(qs, task % nkt, qreg, hb0, sub_of(task));
      load_qk(ks, 0, kcur, hb0, sub_of(task));
      load_tile(vs, 0, vcur, hb0, second_block_bytes(task));
  }
  for (; task >= 0; ++round) {
    ASTAMP(0);
#ifdef AFT_DIAG_STAMPS
    if (stamps && lane == 0) stamps[(size_t)task * 8 + 6] = __builtin_amdgcn_s_memrealtime();
#endif
    const int qt = task % nkt;
    const int pb = first_block(task);       // plane * nblk + the head's first block
    const int sub = sub_of(task);           // HD = 16: which of the block's two heads; GEN: the head's first slot in its first block
    const unsigned hb = head_base(task);   // byte offset of this (plane, block)
    const unsigned vb1 = second_block_bytes(task);
    // the strided tasks of the whole rounds, then (at most) one task of the partial round
    const int next_task = task >= ntasks ? -1 : (task + total_waves < ntasks ? task + total_waves : tail_task);
    const bool has_next = next_task >= 0;

    // B operand of S^T = K Q^T : lane (q = r, h) holds Q[q][8s + 4h + j]; the query bias of the
    // packed in-projection is added here (k_chain.hip stores q and k without bias: K's bias only adds
    // a row constant to the logits, which softmax cancels), then everything is pre-scaled
    if constexpr (!BS) {   // (split tier: bias and scale were applied before the bf16 split, in the chain kernel's epilogue)
        // HD = 16: slots 0, 1 hold features 16 sub ..; GEN: slot i holds the features 8 (sub + i) .. of the first block
        const float *bq = qbias + (pb % nblk) * kHeadDim + 4 * h + (HD == 16 ? 16 * sub : GEN ? 8 * sub : 0);
#pragma unroll
        for (int b = 0; b < NQ; ++b)
#pragma unroll
            for (int s = 0; s < 4; ++s)
                if (4 * b + s < (GEN ? NS : HD == 16 ? 2 : 4 * NB))
                    qreg[b][s] = (qreg[b][s] + *reinterpret_cast<const f32x4 *>(bq + 32 * b + 8 * s)) * scale_log2e;
    }
    // Padded query lanes of the ragged last query tile read workspace nobody wrote: their results are never stored, but
    // the reference tests below are wave-wide (__any), so a large stale value there would switch the VALID lanes of the
    // wave onto the rescale path -- same mathematics, different rounding, i.e. output bits that depend on what the
    // allocator handed out (found by a NaN / 1e30-poisoned pool, tools/debug/poison_repro.py).  Zero queries never trigger.
    if (ragged && qt == nkt - 1 && qt * kTile + r >= tokens) {
#pragma unroll
        for (int b = 0; b < NQ; ++b)
#pragma unroll
            for (int s = 0; s < 4; ++s)
                if (4 * b + s < (GEN ? NS : HD == 16 ? 2 : 4 * NB)) qreg[b][s] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // ---- key tile 0: plain logits, reference maximum ----
    AttnRow<NB> st;
    f32x16 sA, sB;
    if constexpr (BS) sA = qk_tile_bs(kcur[0], qreg[0], true, 0.f, h);
    else sA = qk_tile<NQ, NMF>(kcur, qreg, true, a_one, 0.f);
    if (nkt > 1) load_qk(ks, 1, kcur, hb, sub);
    if (ragged && nkt == 1) { mask_logits(sA, 0); mask_values(vcur, 0); }
    {
        float m0 = max16(sA);
        m0 = fmaxf(m0, other_half(m0));                  // finite: tile 0 holds >= 1 real key
        if constexpr (BS) m0 = round_to_bf16(m0);          // the reference must be exact in the reference MFMA (any value is a valid reference)
        st.zero_ref = !__any(fabsf(m0) > kZeroRefThreshold);
        st.m_ref = st.zero_ref ? 0.f : m0;
        if (!st.zero_ref) {
#pragma unroll
            for (int e = 0; e < 16; ++e) sA[e] -= m0;
        }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) st.oacc[b] = f32x16{0};
    st.lsum2 = f32x2{0.f, 0.f};
    ASTAMP(1);

    // one key tile: `cur` holds S_kt - m_ref (its chain finished an iteration ago), `nxt` receives S_{kt+1} - m_ref.
    // FAST (compile time) = the steady state: tiles kt+1 and kt+2 exist and are full, so there is no mask, no bounds
    // test and no task hand-over in the body; the last two or three tiles of a task run the general form.
    auto step = [&](auto fast, f32x16 &cur, f32x16 &nxt, int kt) {
        constexpr bool FAST = decltype(fast)::value;
        set_progress_priority((rounds - 1 - round) * nkt + (nkt - 1 - kt), rounds * nkt);
        const bool more = FAST || kt + 1 < nkt;
        if (more) {
            if constexpr (BS) nxt = qk_tile_bs(kcur[0], qreg[0], st.zero_ref, -st.m_ref, h);
            else nxt = qk_tile<NQ, NMF>(kcur, qreg, st.zero_ref, a_one, -st.m_ref);   // independent of everything below
            if (FAST || kt + 2 < nkt) load_qk(ks, kt + 2, kcur, hb, sub);
        } else if (has_next) {      // last tile: Q and K are idle -> request the next task's
            const unsigned hbn = head_base(next_task);
            load_qk(qs, next_task % nkt, qreg, hbn, sub_of(next_task));
            load_qk(ks, 0, kcur, hbn, sub_of(next_task));
        }
        // stale-reference test for THIS tile
        const float tmax = max16(cur);
        if (__builtin_expect(__any(tmax > kRescaleThreshold), 0)) {
            const float tm = fmaxf(tmax, other_half(tmax));
            float grow = fmaxf(tm, 0.f);                 // new reference = m_ref + grow  (0 for rows that stay)
            if constexpr (BS) grow = round_to_bf16(st.m_ref + grow) - st.m_ref;   // keep the reference bf16-representable (the difference of two such values is exact)
            const float f = __builtin_amdgcn_exp2f(-grow);
            st.m_ref += grow;
            st.zero_ref = false;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                cur[e] -= grow;
                if (more) nxt[e] -= grow;                 // the pending tile was started from the old reference
#pragma unroll
                for (int b = 0; b < NB; ++b) st.oacc[b][e] *= f;
            }
            st.lsum2 *= f;
        }
        if (!FAST && ragged && more && kt + 2 == nkt) mask_logits(nxt, kt + 1);
        // probabilities (bare v_exp_f32: logits are pre-scaled by log2 e) and per-lane partial row sums
        f32x16 p;
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
            p[e] = __builtin_amdgcn_exp2f(cur[e]);
            p[e + 1] = __builtin_amdgcn_exp2f(cur[e + 1]);
            st.lsum2 += f32x2{p[e], p[e + 1]};
        }
        // O^T += V^T P^T  (k-groups of 8 keys that are all padding -- only in the ragged last tile -- are skipped)
        if constexpr (BS) {   // V^T slot 2m + term; P registers 8m .. 8m+7 are MFMA m's k-values (keys) as they stand
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                if (!FAST && ragged && kt * kTile + 16 * m >= tokens) continue;
                const BsFrag pf = bs_split(f32x4{p[8 * m], p[8 * m + 1], p[8 * m + 2], p[8 * m + 3]},
                                           f32x4{p[8 * m + 4], p[8 * m + 5], p[8 * m + 6], p[8 * m + 7]});
                const bf16x8 ph8 = __builtin_bit_cast(bf16x8, pf.hi), pl8 = __builtin_bit_cast(bf16x8, pf.lo);
                const bf16x8 vh = __builtin_bit_cast(bf16x8, vcur[0][2 * m]), vl = __builtin_bit_cast(bf16x8, vcur[0][2 * m + 1]);
                st.oacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, ph8, st.oacc[0], 0, 0, 0);
                st.oacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, pl8, st.oacc[0], 0, 0, 0);
                st.oacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, ph8, st.oacc[0], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (!FAST && ragged && kt * kTile + 8 * g >= tokens) continue;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        st.oacc[b] = mfma_f32(vcur[b][g][j], p[4 * g + j], st.oacc[b]);
                }
        }
        if (more) {
            load_tile(vs, kt + 1, vcur, hb, vb1);
            if (!FAST && ragged && kt + 2 == nkt) mask_values(vcur, kt + 1);
        } else if (has_next) {
            load_tile(vs, 0, vcur, head_base(next_task), second_block_bytes(next_task));
        }
    };
    using Fast = std::integral_constant<bool, true>;
    using General = std::integral_constant<bool, false>;
    int kt = 0;
#pragma unroll 1
    for (; kt + 3 < nkt; kt += 2) {       // both steps see tiles kt+1 .. kt+3 in range
        step(Fast{}, sA, sB, kt);
        step(Fast{}, sB, sA, kt + 1);
    }
    step(General{}, sA, sB, kt);           // the last two or three tiles (nkt - kt is 2 or 3; 1 when nkt == 1)
    if (kt + 1 < nkt) step(General{}, sB, sA, kt + 1);
    if (kt + 2 < nkt) step(General{}, sA, sB, kt + 2);
    ASTAMP(2);
    ASTAMP(3);

    // O^T accumulator: lane = query r, register e = feature d = (e&3) + 8*(e>>2) + 4h -- i.e. registers
    // 4s..4s+3 are the operand-fragment element (s, h) of this head's feature block.  Stored in the
    // fragment order k_chain.hip consumes: [global 32-row tile][head][s][lane = row%32 + 32h][4]
    // (global rows = plane*tokens + q; a query tile straddles two row tiles when 32 does not divide tokens).
    float l_run = st.lsum2[0] + st.lsum2[1];
    l_run += other_half(l_run);
    const int qrow = qt * kTile + r;
    if (qrow < tokens) {
        const float inv = 1.0f / l_run;
        const int plane = pb / nblk, blk = pb % nblk;
        const unsigned grow = (unsigned)plane * tokens + qrow;
        const unsigned dst = (((grow >> 5) * (unsigned)nblk + blk) * 1024 + ((grow & 31) + 32 * h) * 4) * 4;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (HD == 16 && (g >> 1) != sub) continue;   // rows 8g .. 8g+7 of the block belong to the other head
                if (GEN && (4 * b + g < sub || 4 * b + g >= sub + NS)) continue;   // group 4b + g of the first block onwards: not this head's
                f32x4 o = {st.oacc[b][4 * g] * inv, st.oacc[b][4 * g + 1] * inv, st.oacc[b][4 * g + 2] * inv, st.oacc[b][4 * g + 3] * inv};
                srd_store(os, dst + (b * 4 + g) * 1024, o);
            }
    }
    ASTAMP(4);
#ifdef AFT_DIAG_STAMPS
    if (stamps && lane == 0) stamps[(size_t)task * 8 + 5] = __builtin_amdgcn_s_memrealtime();
#endif
    task = next_task;
  }
}

}  // namespace aft
