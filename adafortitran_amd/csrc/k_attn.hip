// k_attn.hip -- multi-head self-attention core on gfx950 fp32 MFMA.
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
#include <math.h>

#include "aft_internal.h"

namespace aft {

template <int CH>
__global__ __launch_bounds__(256) void attn_kernel(const float *__restrict__ q, const float *__restrict__ k,
                                                   const float *__restrict__ vt, float *__restrict__ out,
                                                   int heads, int tokens, int tokpad, int model_dim,
                                                   float scale_log2e, int ntasks) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int task = blockIdx.x * 4 + wave;
    if (task >= ntasks) return;
    const int nkt = tokpad / kTile;
    const int qt = task % nkt;
    const int ph = task / nkt;  // plane * heads + head
    const int r = lane & 31, h = lane >> 5;

    // q, k, vt arrive in MFMA-fragment order from k_chain.hip: [ph][tile][s or g][lane][4] -> every
    // operand load below is one fully coalesced 1-KB global_load_dwordx4 per wave
    const float *qb = q + (size_t)ph * tokpad * kHeadDim + lane * 4;
    const float *kb = k + (size_t)ph * tokpad * kHeadDim + lane * 4;
    const float *vb = vt + (size_t)ph * kHeadDim * tokpad + lane * 4;

    // B operand of S^T = K Q^T : lane (q = r, h) holds Q[q][8s + 4h + j], pre-scaled
    f32x4 qreg[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        qreg[s] = *reinterpret_cast<const f32x4 *>(qb + qt * 1024 + s * 256);
        qreg[s] *= scale_log2e;
    }

    float m_run = -INFINITY, l_run = 0.f;
    f32x16 oacc = f32x16{0};

    for (int c0 = 0; c0 < nkt; c0 += CH) {
        f32x16 sacc[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int kt = c0 + c;
            sacc[c] = f32x16{0};
            if (kt < nkt) {
                f32x4 kreg[4];
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    kreg[s] = *reinterpret_cast<const f32x4 *>(kb + kt * 1024 + s * 256);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        sacc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(kreg[s][j], qreg[s][j], sacc[c], 0, 0, 0);
                if (kt * kTile + kTile > tokens) {  // ragged last tile: pad keys -> -inf
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int key = kt * kTile + (e & 3) + 8 * (e >> 2) + 4 * h;
                        if (key >= tokens) sacc[c][e] = -INFINITY;
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) sacc[c][e] = -INFINITY;
            }
        }
        // running max over this chunk (lane-local over registers, then the other key half)
        float cmax = sacc[0][0];
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) cmax = fmaxf(cmax, sacc[c][e]);
        cmax = fmaxf(cmax, __shfl_xor(cmax, 32));
        const float m_new = fmaxf(m_run, cmax);  // finite: every chunk holds >= 1 real key
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float p = __builtin_amdgcn_exp2f(sacc[c][e] - m_new);
                sacc[c][e] = p;
                psum += p;
            }
        psum += __shfl_xor(psum, 32);
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[e] *= alpha;
        // O^T += V^T P^T
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int kt = c0 + c;
            if (kt < nkt) {
                const bool ragged = kt * kTile + kTile > tokens;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int key0 = kt * kTile + 8 * g + 4 * h;
                    f32x4 v = *reinterpret_cast<const f32x4 *>(vb + kt * 1024 + g * 256);
                    if (ragged) {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (key0 + j >= tokens) v[j] = 0.f;  // workspace pad is never trusted
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        oacc = __builtin_amdgcn_mfma_f32_32x32x2f32(v[j], sacc[c][4 * g + j], oacc, 0, 0, 0);
                }
            }
        }
    }

    // O^T accumulator: lane = query r, register e = feature d = (e&3) + 8*(e>>2) + 4h
    const int qrow = qt * kTile + r;
    if (qrow < tokens) {
        const float inv = 1.0f / l_run;
        const int plane = ph / heads, head = ph % heads;
        float *dst = out + ((size_t)plane * tokens + qrow) * model_dim + head * kHeadDim + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 o = {oacc[4 * g] * inv, oacc[4 * g + 1] * inv, oacc[4 * g + 2] * inv, oacc[4 * g + 3] * inv};
            *reinterpret_cast<f32x4 *>(dst + 8 * g) = o;
        }
    }
}

hipError_t launch_attention(const aft_config &c, const float *q, const float *k, const float *vt, float *attn,
                            int planes, int tokens, int tokpad, hipStream_t st) {
    const int ntasks = planes * c.num_head * (tokpad / kTile);
    const int blocks = (ntasks + 3) / 4;
    const float scale_log2e = 1.4426950408889634f / sqrtf((float)kHeadDim);
    hipLaunchKernelGGL((attn_kernel<3>), dim3(blocks), dim3(256), 0, st, q, k, vt, attn, c.num_head, tokens,
                       tokpad, c.model_dim, scale_log2e, ntasks);
    return hipGetLastError();
}

}  // namespace aft
