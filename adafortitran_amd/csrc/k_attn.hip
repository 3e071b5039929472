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

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "aft_internal.h"

namespace aft {

// Buffer-resource loads (SRD + 32-bit byte offset): with several waves per SIMD streaming operands,
// global_load's 64-bit per-lane addressing throttles the fp32 MFMA stream (tools/micro/mfma_feed2.hip:
// 104 vs 149 TFLOP/s at 3 workgroups/CU); buffer_load does not.
using Srd = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ Srd make_srd(const float *p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ f32x4 srd_load(Srd r, unsigned byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
}
__device__ __forceinline__ void srd_store(Srd r, unsigned byte_off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(decltype(__builtin_amdgcn_raw_buffer_load_b128(r, 0, 0, 0)), v), r, byte_off, 0, 0);
}

#ifndef AFT_ATTN_WAVES
#define AFT_ATTN_WAVES 3   // waves per SIMD the register budget is capped for (and the persistent grid sized to)
#endif

struct AttnState {
    float m_run, l_run;
    f32x16 oacc;
};

// One chunk of CH key tiles: S^T tiles -> online-softmax update -> O^T += V^T P^T.
// TAIL = false: all CH tiles exist and are full (no masks, no bounds checks: the steady state);
// TAIL = true : the final chunk -- tiles >= nkt are skipped, the ragged last tile is masked.
template <int CH, bool TAIL>
__device__ __forceinline__ void attn_chunk(AttnState &st, const f32x4 (&qreg)[4], Srd ks, Srd vs, unsigned kb,
                                           unsigned vb, int c0, int nkt, int tokens, int h) {
    f32x16 sacc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int kt = c0 + c;
        sacc[c] = f32x16{0};
        if (!TAIL || kt < nkt) {
            f32x4 kreg[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) kreg[s] = srd_load(ks, kb + (unsigned)(kt * 1024 + s * 256) * 4);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    sacc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(kreg[s][j], qreg[s][j], sacc[c], 0, 0, 0);
            if (TAIL && kt * kTile + kTile > tokens) {  // ragged last tile: pad keys -> -inf
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
    const float m_new = fmaxf(st.m_run, cmax);  // finite: every chunk holds >= 1 real key
    const float alpha = __builtin_amdgcn_exp2f(st.m_run - m_new);
    // exponentials: the shift and the row sum run on register PAIRS (v_pk_add_f32: two floats per
    // lane per issue) -- VALU issue comes straight out of fp32-MFMA time on this path
    const f32x2 m2 = {m_new, m_new};
    f32x2 psum2 = {0.f, 0.f};
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
            f32x2 t = f32x2{sacc[c][e], sacc[c][e + 1]} - m2;
            t[0] = __builtin_amdgcn_exp2f(t[0]);
            t[1] = __builtin_amdgcn_exp2f(t[1]);
            sacc[c][e] = t[0];
            sacc[c][e + 1] = t[1];
            psum2 += t;
        }
    float psum = psum2[0] + psum2[1];
    psum += __shfl_xor(psum, 32);
    st.l_run = st.l_run * alpha + psum;
    st.m_run = m_new;
#pragma unroll
    for (int e = 0; e < 16; ++e) st.oacc[e] *= alpha;
    // O^T += V^T P^T
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int kt = c0 + c;
        if (!TAIL || kt < nkt) {
            const bool ragged = TAIL && kt * kTile + kTile > tokens;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (ragged && kt * kTile + 8 * g >= tokens) continue;   // both halves of this k group are padding
                const int key0 = kt * kTile + 8 * g + 4 * h;
                f32x4 v = srd_load(vs, vb + (unsigned)(kt * 1024 + g * 256) * 4);
                if (ragged) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (key0 + j >= tokens) v[j] = 0.f;  // workspace pad is never trusted
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    st.oacc = __builtin_amdgcn_mfma_f32_32x32x2f32(v[j], sacc[c][4 * g + j], st.oacc, 0, 0, 0);
            }
        }
    }
}

// launch bound (256, 3): <= 168 registers keeps accumulators in VGPRs (the MFMA's VGPR form) --
// with the default 512-register budget hipcc parks them in AGPRs and pays ~2.5 v_accvgpr moves
// per MFMA around the softmax, which on the fp32 matrix path comes straight out of MFMA time.
template <int CH>
__global__ __launch_bounds__(256, AFT_ATTN_WAVES) void attn_kernel(const float *__restrict__ q, const float *__restrict__ k,
                                                      const float *__restrict__ vt, const float *__restrict__ qbias,
                                                      float *__restrict__ out, int heads, int tokens, int tokpad,
                                                      int model_dim,
                                                      float scale_log2e, int ntasks, unsigned long long *stamps) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // persistent waves with a static, balanced schedule: the grid is sized to the co-resident wave
    // count and wave w takes tasks w, w + W, w + 2W ... (B=128: 9216 tasks over 3072 waves = exactly
    // 3 each), so every SIMD finishes together -- a 40-us task has no tail to wait for.
    const Srd qs = make_srd(q), ks = make_srd(k), vs = make_srd(vt), os = make_srd(out);
    const int total_waves = gridDim.x * 4;
    const int nkt = tokpad / kTile;
    const int r = lane & 31, h = lane >> 5;
#ifdef AFT_DIAG_STAMPS
#define ASTAMP(i) do { if (stamps && lane == 0) stamps[(size_t)task * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define ASTAMP(i) do { } while (0)
#endif
  for (int task = blockIdx.x * 4 + wave; task < ntasks; task += total_waves) {
    ASTAMP(0);
#ifdef AFT_DIAG_STAMPS
    if (stamps && lane == 0) stamps[(size_t)task * 8 + 6] = __builtin_amdgcn_s_memrealtime();
#endif
    const int qt = task % nkt;
    const int ph = task / nkt;  // plane * heads + head

    // q, k, vt arrive in MFMA-fragment order from k_chain.hip: [ph][tile][s or g][lane][4] -> every
    // operand load below is one fully coalesced 1-KB global_load_dwordx4 per wave
    const unsigned hb = ((unsigned)ph * tokpad * kHeadDim + lane * 4) * 4;   // byte offset of this (plane, head)
    const unsigned kb = hb, vb = hb;

    // B operand of S^T = K Q^T : lane (q = r, h) holds Q[q][8s + 4h + j]; the query bias of the
    // packed in-projection is added here (k_chain.hip stores q and k without bias: K's bias only adds
    // a row constant to the logits, which softmax cancels), then everything is pre-scaled
    f32x4 qreg[4];
    const float *bq = qbias + (ph % heads) * kHeadDim + 4 * h;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        qreg[s] = srd_load(qs, hb + (unsigned)(qt * 1024 + s * 256) * 4);
        qreg[s] = (qreg[s] + *reinterpret_cast<const f32x4 *>(bq + 8 * s)) * scale_log2e;
    }

    AttnState st;
    st.m_run = -INFINITY;
    st.l_run = 0.f;
    st.oacc = f32x16{0};

    int c0 = 0;
    const int full_tiles = tokens / kTile;   // tiles with no padded key
    ASTAMP(1);
    for (; c0 + CH <= full_tiles; c0 += CH) attn_chunk<CH, false>(st, qreg, ks, vs, kb, vb, c0, nkt, tokens, h);
    ASTAMP(2);
    for (; c0 < nkt; c0 += CH) attn_chunk<CH, true>(st, qreg, ks, vs, kb, vb, c0, nkt, tokens, h);
    ASTAMP(3);

    // O^T accumulator: lane = query r, register e = feature d = (e&3) + 8*(e>>2) + 4h -- i.e. registers
    // 4s..4s+3 are the operand-fragment element (s, h) of this head's feature block.  Stored in the
    // fragment order k_chain.hip consumes: [global 32-row tile][head][s][lane = row%32 + 32h][4]
    // (global rows = plane*tokens + q; a query tile straddles two row tiles when 32 does not divide tokens).
    const int qrow = qt * kTile + r;
    if (qrow < tokens) {
        const float inv = 1.0f / st.l_run;
        const int plane = ph / heads, head = ph % heads;
        const unsigned grow = (unsigned)plane * tokens + qrow;
        const unsigned dst = (((grow >> 5) * (unsigned)(model_dim / kHeadDim) + head) * 1024 + ((grow & 31) + 32 * h) * 4) * 4;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 o = {st.oacc[4 * g] * inv, st.oacc[4 * g + 1] * inv, st.oacc[4 * g + 2] * inv, st.oacc[4 * g + 3] * inv};
            srd_store(os, dst + g * 1024, o);
        }
    }
    ASTAMP(4);
#ifdef AFT_DIAG_STAMPS
    if (stamps && lane == 0) stamps[(size_t)task * 8 + 5] = __builtin_amdgcn_s_memrealtime();
#endif
  }
}

hipError_t launch_attention(const aft_config &c, const float *q, const float *k, const float *vt, const float *qbias,
                            float *attn, int planes, int tokens, int tokpad, hipStream_t st) {
    const int ntasks = planes * c.num_head * (tokpad / kTile);
    const int resident_blocks = AFT_ATTN_WAVES * current_device_cus();   // CUs x 3 workgroups (launch bound: 3 waves / SIMD)
    const int blocks = std::min((ntasks + 3) / 4, resident_blocks);
    const float scale_log2e = 1.4426950408889634f / sqrtf((float)kHeadDim);
#ifdef AFT_DIAG_STAMPS
    if (getenv("AFT_STAMPS")) {   // per-task stamps + in-kernel clock (diagnostic build only)
        static unsigned long long *dbuf = nullptr;
        if (!dbuf) (void)hipMalloc(&dbuf, sizeof(unsigned long long) * 8 * 16384);
        (void)hipMemset(dbuf, 0, sizeof(unsigned long long) * 8 * 16384);
        hipLaunchKernelGGL((attn_kernel<3>), dim3(blocks), dim3(256), 0, st, q, k, vt, qbias, attn, c.num_head, tokens,
                           tokpad, c.model_dim, scale_log2e, ntasks, dbuf);
        (void)hipDeviceSynchronize();
        static int printed = 0;
        if (printed++ < 2) {
            std::vector<unsigned long long> hbuf(8 * 16384);
            (void)hipMemcpy(hbuf.data(), dbuf, hbuf.size() * 8, hipMemcpyDeviceToHost);
            double sum[5] = {0};
            int n = ntasks < 16384 ? ntasks : 16384;
            for (int t = 0; t < n; ++t)
                for (int i = 1; i < 5; ++i) sum[i] += (double)(hbuf[t * 8 + i] - hbuf[t * 8 + i - 1]);
            double clk = 0;
            for (int t = 0; t < n; ++t)
                clk += (double)(hbuf[t * 8 + 4] - hbuf[t * 8 + 0]) / (double)(hbuf[t * 8 + 5] - hbuf[t * 8 + 6]) * 100e6;
            printf("in-kernel shader clock: %.3f GHz\n", clk / n / 1e9);
            printf("attn stamps: mean per task: prologue=%.0f steady=%.0f tail=%.0f store=%.0f (cycles); MFMA ideal %d\n",
                   sum[1] / n, sum[2] / n, sum[3] / n, sum[4] / n, 288 * 64);
        }
        return hipGetLastError();
    }
#endif
    hipLaunchKernelGGL((attn_kernel<3>), dim3(blocks), dim3(256), 0, st, q, k, vt, qbias, attn, c.num_head, tokens,
                       tokpad, c.model_dim, scale_log2e, ntasks, (unsigned long long *)nullptr);
    return hipGetLastError();
}

}  // namespace aft
