// attn16_device.h (device code of k_attn.hip) -- the attention core for HEAD DIMENSION 16 on v_mfma_f32_16x16x4_f32 (round 6).
//
// Reference semantics: as attn_device.h (nn.MultiheadAttention inside nn.TransformerEncoderLayer, reference
// src/models/blocks/encoders.py:44-55), `num_head: 8` at `model_dim: 128` and the like: heads of 16 features, two per 32-feature block of
// the q / k / v^T layout the chain kernel writes.
//
// Why a second body.  attn_body<HD = 16> runs the head on 32x32x2 MFMAs: S^T is an 8-MFMA chain over the head's 16 features, but the
// value product O^T = V^T P^T has to take a whole 32-feature block as its A operand (a 32-row MFMA cannot take half its rows): 16 MFMAs
// of 64 cycles of which half the rows are the OTHER head's -- 1 536 matrix cycles per (head, 32 x 32 key-query tile) for 1 024 useful
// (other_shapes.d128_h8_hd16: attention at 0.49 of the fp32 roof).  A block-diagonal P does not help there (an MFMA has ONE B operand:
// masking V^T's rows per head doubles the k steps).  The 16x16x4 shape has 16-row operands -- exactly a head:
//   S^T[16 keys x 16 queries]  = sum_t  K[keys][4t .. 4t+3] . Q^T[4t .. 4t+3][queries]        4 MFMAs, 4 tiles per 32 x 32: 16 x 32 cycles
//   O^T[16 feats x 16 queries] += V^T[feats][4 keys] . P^T[4 keys][queries]                    4 MFMAs per (key half, query half): 16 x 32
// = 1 024 cycles.  As in the 32x32 formulation the probabilities never move: the S^T accumulator of (key half kb, query half qb) has
// lane (g, j) = query 16 qb + j, register v = key 16 kb + 4 g + v -- as B operand of the value product's MFMA v it supplies k index g =
// key 16 kb + 4 g + v, and the A operand is loaded to match: lane (g, i) = V^T[feature i][key 16 kb + 4 g + v], which in the chain
// kernel's v^T fragment order (lane = feature + 32 hh, 16 bytes = keys 8 g4 + 4 hh + 0..3) is ONE 16-byte load per lane and key half
// (piece g4 = 2 kb + g / 2 of lane (feature 16 sub + i) + 32 (g & 1)): registers v = 0..3 as they arrive.  K and Q^T: the contraction
// over the head's 16 features may visit them in any order as long as both operands agree, so MFMA t takes feature 4 g + t from lane
// group g: lane (g, i) then needs features 4 g .. 4 g + 3 of key / query i -- ONE 16-byte piece of the q / k fragment order (slot
// 2 sub + g / 2, lane 32 (g & 1) + 16 kb + i), registers t = 0..3 as they arrive (the first version took feature 4 t + g: eight dword
// loads per tile, 109 us at 8 heads of 16; this one: see DESIGN.md 4.2).
// Softmax: the stale-reference scheme of attn_device.h; a query's row is spread over the four lane groups g (lanes j, j + 16, j + 32,
// j + 48), so row maxima / sums end with two cross-lane steps -- once per task and on the (rare) rescale path only.  The reference
// enters as the C operand of each tile's first MFMA (a 4-register splat per query half), no extra MFMA.
#pragma once
#include <math.h>

#include <type_traits>

#include "attn_device.h"

namespace aft {

__device__ __forceinline__ f32x4 mfma16x4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float srd_load1(Srd r, unsigned byte_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
}
// maximum / sum over the four lane groups that share a query (lanes j, j + 16, j + 32, j + 48)
__device__ __forceinline__ float groups_max(float x) {
    x = fmaxf(x, __shfl_xor(x, 16));
    return fmaxf(x, other_half(x));
}
__device__ __forceinline__ float groups_sum(float x) {
    x += __shfl_xor(x, 16);
    return x + other_half(x);
}

// HD = 8 (late round 6; `num_head: 16` at `model_dim: 128`): a head is ONE 8-feature slot of the layouts, four heads per block.  The same
// body with half of every 16-row / 16-deep operand unused: lane groups 2, 3 of the Q^T operand are zero (the logits contract the head's
// 8 features; K's lane groups 2, 3 re-read groups 0, 1 -- finite, multiplied by zero), V^T's rows 8 .. 15 re-read rows 0 .. 7 and the
// matching half of O^T is not stored.  Per task the matrix work of a 16-feature head -- half of it idle -- against the 32x32x2 body's
// 32-feature block per 8-feature head (attn_kernel<8>: three quarters idle).
// One WAVE walks the tasks first_task, first_task + total_waves, ... < ntasks, then (at most) `tail_task` -- as attn_body.
// task = (plane * heads + head) * nkt + query tile; head = (32 / HD) block + sub.
template <int TOK = 0, int HD = 16>
__device__ __forceinline__ void attn16_body(const float *__restrict__ q, const float *__restrict__ k, const float *__restrict__ vt,
                                            const float *__restrict__ qbias, float *__restrict__ out, int nblk, int tokens_rt,
                                            int tokpad_rt, int model_dim, float scale_log2e, const int first_task, const int total_waves,
                                            int ntasks, const int tail_task = -1) {
    int lane_l = threadIdx.x & 63;
    asm volatile("" : "+v"(lane_l));
    const int lane = lane_l, i = lane & 15, g = lane >> 4;
    const Srd qs = make_srd(q), ks = make_srd(k), vs = make_srd(vt), os = make_srd(out);
    const int tokens = TOK > 0 ? TOK : tokens_rt;
    const int tokpad = TOK > 0 ? (TOK + kTile - 1) / kTile * kTile : tokpad_rt;
    const int nkt = tokpad / kTile;
    const bool ragged = (tokens & (kTile - 1)) != 0;
    const unsigned blk_bytes = (unsigned)tokpad * kHeadDim * 4;
    static_assert(HD == 16 || HD == 8, "heads of 16 or 8 features");
    constexpr int NSUB = 32 / HD;                                   // heads per 32-feature block
    auto first_block = [&](int task) { return (task / nkt) / NSUB; };
    auto sub_of = [&](int task) { return (task / nkt) & (NSUB - 1); };
    // the 8-feature slot of a block and the lane's feature inside the head: HD 16: slot 2 sub + g / 2; HD 8: slot sub (groups 2, 3 = 0, 1)
    auto slot_of = [&](int sub) { return HD == 16 ? 2 * sub + (g >> 1) : sub; };
    const int vfeat = HD == 16 ? i : (i & 7);                      // V^T row of lane i (HD 8: rows 8 .. 15 repeat 0 .. 7, never stored)
    // K or Q^T of 32-token tile `tile`: [half][t] = feature HD sub + 4 g + t of token 16 half + i: one 16-byte piece per half
    auto load_kq = [&](Srd src, int tile, f32x4 (&dst)[2], unsigned base, int sub) {
#pragma unroll
        for (int half = 0; half < 2; ++half)
            dst[half] = srd_load(src, base + (unsigned)(tile * 1024 + slot_of(sub) * 256 + (32 * (g & 1) + 16 * half + i) * 4) * 4u);
    };
    // V^T of key tile kt: [kb] = keys 16 kb + 4 g + 0..3 of feature HD sub + i
    auto load_v = [&](int kt, f32x4 (&dst)[2], unsigned base, int sub) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
            dst[kb] = srd_load(vs, base + (unsigned)(kt * 1024 + (2 * kb + (g >> 1)) * 256 + (32 * (g & 1) + HD * sub + vfeat) * 4) * 4u);
    };
    auto mask_logits = [&](f32x4 (&sv)[2][2], int kt) {      // padded keys of the ragged last tile
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int v = 0; v < 4; ++v)
                if (kt * kTile + 16 * kb + 4 * g + v >= tokens) sv[kb][0][v] = sv[kb][1][v] = -INFINITY;
    };
    auto mask_values = [&](f32x4 (&vv)[2], int kt) {          // the workspace pad is never trusted: 0 x NaN would poison the row
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int v = 0; v < 4; ++v)
                if (kt * kTile + 16 * kb + 4 * g + v >= tokens) vv[kb][v] = 0.f;
    };

    const int rounds = (ntasks + total_waves - 1) / total_waves + (tail_task >= 0 ? 1 : 0);
    int round = 0;
    int task = first_task < ntasks ? first_task : tail_task;
    f32x4 qreg[2], kcur[2];
    f32x4 vcur[2];
    if (task >= 0) {
        const unsigned hb0 = (unsigned)first_block(task) * blk_bytes;
        load_kq(qs, task % nkt, qreg, hb0, sub_of(task));
        load_kq(ks, 0, kcur, hb0, sub_of(task));
        load_v(0, vcur, hb0, sub_of(task));
    }
    for (; task >= 0; ++round) {
        const int qt = task % nkt;
        const int pb = first_block(task), sub = sub_of(task);
        const unsigned hb = (unsigned)pb * blk_bytes;
        const int next_task = task >= ntasks ? -1 : (task + total_waves < ntasks ? task + total_waves : tail_task);
        const bool has_next = next_task >= 0;

        // Q^T operand: (q + query bias) x log2 e / sqrt(16); padded query lanes of the ragged last query tile are zeroed (their results
        // are never stored, but the reference tests are wave-wide: attn_device.h)
        {
            const f32x4 bq = *reinterpret_cast<const f32x4 *>(qbias + (pb % nblk) * kHeadDim + HD * sub + 4 * (HD == 16 ? g : (g & 1)));   // features 4 g .. 4 g + 3 of the head
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                const bool pad = (ragged && qt == nkt - 1 && qt * kTile + 16 * qb + i >= tokens) || (HD == 8 && g >= 2);
                qreg[qb] = pad ? f32x4{0.f, 0.f, 0.f, 0.f} : (qreg[qb] + bq) * scale_log2e;
            }
        }
        // S^T of one key tile minus the reference: four independent 4-MFMA chains, [kb][qb]; `neg` = -m_ref as the chains' initial value
        float m_ref[2] = {0.f, 0.f};
        f32x4 neg[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        bool zero_ref = true;
        auto qk_tile = [&](f32x4 (&s)[2][2]) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int qb = 0; qb < 2; ++qb)
                        s[kb][qb] = mfma16x4(kcur[kb][t], qreg[qb][t], t == 0 ? neg[qb] : s[kb][qb]);   // (neg is 0 while the reference is: no select)
        };
        auto tile_max = [&](const f32x4 (&s)[2][2], int qb) {      // v_max3 chain over the lane's 8 logits of query half qb
            float m = __builtin_fmaxf(s[0][qb][0], s[0][qb][1]);
            m = __builtin_fmaxf(__builtin_fmaxf(m, s[0][qb][2]), s[0][qb][3]);
            m = __builtin_fmaxf(__builtin_fmaxf(m, s[1][qb][0]), s[1][qb][1]);
            return __builtin_fmaxf(__builtin_fmaxf(m, s[1][qb][2]), s[1][qb][3]);
        };

        // ---- key tile 0: plain logits, reference maximum ----
        f32x4 sA[2][2], sB[2][2];
        qk_tile(sA);
        if (nkt > 1) load_kq(ks, 1, kcur, hb, sub);
        if (ragged && nkt == 1) { mask_logits(sA, 0); mask_values(vcur, 0); }
        {
            float m0[2];
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) m0[qb] = groups_max(tile_max(sA, qb));      // finite: tile 0 holds >= 1 real key
            zero_ref = !__any(fmaxf(fabsf(m0[0]), fabsf(m0[1])) > kZeroRefThreshold);
            if (!zero_ref) {
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
                    m_ref[qb] = m0[qb];
                    neg[qb] = f32x4{-m0[qb], -m0[qb], -m0[qb], -m0[qb]};
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) sA[kb][qb] += neg[qb];
                }
            }
        }
        f32x4 oacc[2][2];      // [kb][qb]: the two key halves accumulate separately (four independent chains), summed at the end
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) oacc[kb][qb] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x2 lsum[2] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};

        auto step = [&](auto fast, f32x4 (&cur)[2][2], f32x4 (&nxt)[2][2], int kt) {
            constexpr bool FAST = decltype(fast)::value;
            set_progress_priority((rounds - 1 - round) * nkt + (nkt - 1 - kt), rounds * nkt);
            const bool more = FAST || kt + 1 < nkt;
            if (more) {
                qk_tile(nxt);
                if (FAST || kt + 2 < nkt) load_kq(ks, kt + 2, kcur, hb, sub);
            } else if (has_next) {      // last tile: Q and K are idle -> request the next task's
                const unsigned hbn = (unsigned)first_block(next_task) * blk_bytes;
                load_kq(qs, next_task % nkt, qreg, hbn, sub_of(next_task));
                load_kq(ks, 0, kcur, hbn, sub_of(next_task));
            }
            // stale-reference test for THIS tile (lane-local maxima suffice for the wave-wide test)
            const float t0 = tile_max(cur, 0), t1 = tile_max(cur, 1);
            if (__builtin_expect(__any(fmaxf(t0, t1) > kRescaleThreshold), 0)) {
                const float tm[2] = {groups_max(t0), groups_max(t1)};
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
                    const float grow = fmaxf(tm[qb], 0.f);           // new reference = m_ref + grow (0 for rows that stay)
                    const float f = __builtin_amdgcn_exp2f(-grow);
                    m_ref[qb] += grow;
                    neg[qb] = f32x4{-m_ref[qb], -m_ref[qb], -m_ref[qb], -m_ref[qb]};
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) {
                        cur[kb][qb] -= f32x4{grow, grow, grow, grow};
                        if (more) nxt[kb][qb] -= f32x4{grow, grow, grow, grow};     // the pending tile was started from the old reference
                        oacc[kb][qb] *= f;
                    }
                    lsum[qb] *= f;
                }
                zero_ref = false;
            }
            if (!FAST && ragged && more && kt + 2 == nkt) mask_logits(nxt, kt + 1);
            // probabilities and per-lane partial row sums
            f32x4 p[2][2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) p[kb][qb][v] = __builtin_amdgcn_exp2f(cur[kb][qb][v]);
                    lsum[qb] += f32x2{p[kb][qb][0], p[kb][qb][1]} + f32x2{p[kb][qb][2], p[kb][qb][3]};
                }
            // O^T += V^T P^T: MFMA v of (kb, qb) contracts keys 16 kb + 4 g + v
#pragma unroll
            for (int v = 0; v < 4; ++v)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int qb = 0; qb < 2; ++qb) oacc[kb][qb] = mfma16x4(vcur[kb][v], p[kb][qb][v], oacc[kb][qb]);
            if (more) {
                load_v(kt + 1, vcur, hb, sub);
                if (!FAST && ragged && kt + 2 == nkt) mask_values(vcur, kt + 1);
            } else if (has_next) {
                load_v(0, vcur, (unsigned)first_block(next_task) * blk_bytes, sub_of(next_task));
            }
        };
        using Fast = std::integral_constant<bool, true>;
        using General = std::integral_constant<bool, false>;
        int kt = 0;
#pragma unroll 1
        for (; kt + 3 < nkt; kt += 2) {
            step(Fast{}, sA, sB, kt);
            step(Fast{}, sB, sA, kt + 1);
        }
        step(General{}, sA, sB, kt);
        if (kt + 1 < nkt) step(General{}, sB, sA, kt + 1);
        if (kt + 2 < nkt) step(General{}, sA, sB, kt + 2);

        // O^T accumulator of query half qb: lane (g, j) = query 16 qb + j, register v = feature 16 sub + 4 g + v: the 16-byte piece
        // (slot 2 sub + g / 2, lane (row % 32) + 32 (g & 1)) of the attention tile the chain kernel consumes
        const int plane = pb / nblk, blk = pb % nblk;
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const float l_run = groups_sum(lsum[qb][0] + lsum[qb][1]);
            const int qrow = qt * kTile + 16 * qb + i;
            if (qrow < tokens && (HD == 16 || g < 2)) {
                const float inv = 1.0f / l_run;
                const unsigned grow = (unsigned)plane * tokens + qrow;
                const unsigned dst = (((grow >> 5) * (unsigned)nblk + blk) * 1024 + slot_of(sub) * 256 + ((grow & 31) + 32 * (g & 1)) * 4) * 4;
                srd_store(os, dst, (oacc[0][qb] + oacc[1][qb]) * inv);
            }
        }
        task = next_task;
    }
}

}  // namespace aft
