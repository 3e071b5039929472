// k_conv.hip -- the two ConvEnhancer stacks, each fused with the op that feeds it.
//
// Reference semantics
//   head mode  (S1+S2, reference src/models/fortitran.py:203-209): split complex pilots into
//              Re/Im planes, pilot_upsampler Linear(Ps*Pt -> S*T), view(S,T), initial_enhancer.
//   tail mode  (linear_2 + S6 + S7 + S8 + torch.complex, encoders.py:70, fortitran.py:225-231,180):
//              linear_2 (d -> p) on the encoder output, inverse patch map (patch_processors.py:
//              53-57,69-71: token t=(sc/p0)*(T/p1)+sym/p1, feature f=(sc%p0)*p1+sym%p1), residual
//              with conv_enhanced, final_refiner, interleave Re/Im planes into complex64.
//   ConvEnhancer (blocks/enhancers.py:12-20): 3x3 cross-correlations, zero padding 1,
//              channels 1->8->32->8->1, ReLU after the first three.
//
// MI355X mapping: one workgroup (8 waves) per (plane, row band); the whole receptive field lives in
// LDS as [channel][symbol column + 1][subcarrier row], rows contiguous, so a wave's 32 pixel lanes
// always read consecutive banks.  conv1 (1->8) and conv4 (8->1) are 4 % of the FLOPs and run on the
// VALU.  conv2 (8->32) and conv3 (32->8) are implicit GEMMs on v_mfma_f32_32x32x2_f32 (exact fp32):
//   conv2  A = weights [co 32][k = (kx, ky, ci)]  B = conv1 output read straight from LDS at the
//          tap's offset (one ds_read_b32 per MFMA, pixel = lane)      K = 72  -> 36 MFMAs / column
//   conv3  the accumulator of conv2 has lane = pixel, register = channel, i.e. it already IS the
//          B operand of conv3 for the centre row tap; the ky = -1/+1 taps are the same registers
//          moved one lane (DPP wave shift), so the 32-channel tensor never leaves registers.
//          The kx taps are folded into M: row (kx, co) of the product is the contribution of input
//          column t' to output column t'-kx+1, so the three row groups are three output columns in
//          flight; rotating the accumulator registers by 4 between columns sums them for free.
//          M = 24 of 32 rows used, K = 96 -> 48 MFMAs / column.
// A wave owns a tile of 32 rows (30 of them valid conv3 outputs: one halo lane per side replaces
// any cross-wave exchange) over half of the symbol columns; weights fragments stay in registers for
// the whole kernel.  HBM traffic per plane = compulsory only (pilots / x rows in, one plane out).
// Bands carry a 4-row halo (4 stacked 3x3 convs); the default 120x14 grid is one band, 4 tiles.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "conv_device.h"

namespace aft {

// TRAIN = false is the inference kernel.  TRAIN = true adds what the training path needs:
// a plain-plane input mode, the stage outputs written to HBM (forward: the activations the backward
// needs; backward: the pre-activation gradients the weight-gradient kernels need), and a masked
// activation (backward: the ReLU derivative taken from the saved forward activation).  The backward of
// the stack IS this kernel run on the gradient with transposed, flipped weights: conv4^T is 1->8,
// conv3^T 8->32, conv2^T 32->8, conv1^T 8->1.
// FIXED = the default 120 x 14 grid as ONE band (S = 120, T = 14, SP = 128, 4 row tiles x 2 column segments) with the
// geometry as compile-time constants: every LDS offset of the matrix phase folds into the instruction's immediate (the
// generic kernel spent ~60 of its ~175 vector instructions per column sweep on address arithmetic with the runtime plane /
// column strides -- vector instructions cost fp32-MFMA time) and the divisions by T / band_rows of the
// input, conv1 and conv4 loops become shifts and multiplies.  Other grids run the generic instantiation.
template <bool TRAIN, bool FIXED>
__global__ __launch_bounds__(kConvThreads) void conv_stack_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int S = FIXED ? 120 : a.S, T = FIXED ? 14 : a.T, SP = FIXED ? 128 : a.SP;
    const int BR = FIXED ? 120 : a.band_rows, NB = FIXED ? 1 : a.nbands, NT = FIXED ? 4 : a.ntiles, NSEG = FIXED ? 2 : a.nseg;
    const int LR = BR + 8;
    const int col_stride = SP, plane = (T + 2) * SP;
    float *in0 = smem;                 // [T+2][SP]      column index = symbol + 1
    float *c1 = in0 + plane;           // [8][T+2][SP]   conv1 output
    float *c3 = c1 + 8 * plane;        // [8][T+2][SP]   conv3 output

    const int tid = threadIdx.x;
#ifdef AFT_DIAG_STAMPS
#define CSTAMP(i) do { if (a.stamps && tid == 0) a.stamps[(size_t)blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define CSTAMP(i) do { } while (0)
#endif
    CSTAMP(0);
    const int n = blockIdx.x / NB, band = blockIdx.x % NB;
    const int frame = n >> 1, part = n & 1;
    const int gr0 = band * BR - 4;  // global row of local row 0

    // ---- stage the conv2/conv3 weights through LDS (coalesced), gather the MFMA A fragments ----
    // conv2.weight [32][8][3][3] and conv3.weight [8][32][3][3] are 2304 floats each.
    float *small = smem + a.arena;     // head: pilot plane [pf]; tail: lin2 weights [p][d] + bias [p]
    // Staged TRANSPOSED -- conv2 as [k = (ci, ky, kx)][co], conv3 as [(ci, ky)][(kx, co)], row stride 33 -- so that the
    // gather below reads consecutive floats across the lanes.  In torch order the lanes of a gather sit 72 (conv2) or 288
    // (conv3) floats apart: 8-way bank conflicts on each of the 84 reads of every wave, 4 us of the 59-us kernel.
    for (int i = tid; i < 2304; i += kConvThreads) {
        smem[(i % 72) * 33 + i / 72] = a.cw[1][i];
        const int co = i / 288, rem = i % 288;   // conv3.weight [co 8][ci 32][ky 3][kx 3]
        smem[kW3Off + (rem / 3) * 33 + (rem % 3) * 8 + co] = a.cw[2][i];
    }
    if (TRAIN) {
    } else if (a.mode == 0) {
        for (int i = tid; i < a.pf; i += kConvThreads) small[i] = a.pilots[((size_t)frame * a.pf + i) * 2 + part];
    } else if (a.lin2_out == nullptr) {
        const int p = a.p0 * a.p1;
        for (int i = tid; i < p * a.d; i += kConvThreads) small[i] = a.lin2_w[i];
        for (int i = tid; i < p; i += kConvThreads) small[p * a.d + i] = a.lin2_b[i];
    }
    const int lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    float bias3[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) bias3[e] = (TRAIN && !a.cb[2]) ? 0.f : a.cb[2][e + 4 * h];
    __syncthreads();
    float wa2[36], wa3[48];
#pragma unroll
    for (int kb = 0; kb < 36; ++kb) {        // k slot (kb, h): tap = kb>>2 = kx*3+ky, ci = 4h + (kb&3); row = co = j
        const int tap = kb >> 2, kx = tap / 3, ky = tap % 3, ci = 4 * h + (kb & 3);
        wa2[kb] = smem[(ci * 9 + ky * 3 + kx) * 33 + j];
    }
    {
        const int kx = min(j >> 3, 2), co = j & 7;   // row j = (kx, co); rows 24..31 are padding
        const float keep = j < 24 ? 1.f : 0.f;
#pragma unroll
        for (int kb = 0; kb < 48; ++kb) {    // k slot (kb, h): ky = kb>>4, ci = C-layout row of register kb&15
            const int ky = kb >> 4, e = kb & 15, ci = (e & 3) + 8 * (e >> 2) + 4 * h;
            wa3[kb] = keep * smem[kW3Off + (ci * 3 + ky) * 33 + kx * 8 + co];
        }
    }
    __syncthreads();
    CSTAMP(1);
    // ---- zero LDS (border columns, rows outside the plane, halo rows nobody writes); the conv2
    //      bias in accumulator-register order sits behind `small` ----
    {
        f32x4 *z = reinterpret_cast<f32x4 *>(smem);
        const int n4 = (17 * plane + 3) >> 2;   // arena is a multiple of 4 floats
        for (int i = tid; i < n4; i += kConvThreads) z[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float *bias2 = small + a.extra;    // [2 halves][16]
    float *w1s = bias2 + 32;           // conv1: 72 weights + 8 biases
    if (tid >= 64 && tid < 144) w1s[tid - 64] = tid < 136 ? a.cw[0][tid - 64] : ((TRAIN && !a.cb[0]) ? 0.f : a.cb[0][tid - 136]);
    if (tid < 32) bias2[tid] = (TRAIN && !a.cb[1]) ? 0.f : a.cb[1][(tid & 3) + 8 * ((tid & 15) >> 2) + 4 * (tid >> 4)];
    __syncthreads();
    CSTAMP(2);

    // ---- input plane ----
    if (TRAIN && a.mode == 2) {
        for (int i = tid; i < LR * T; i += kConvThreads) {
            const int lr = i / T, t = i - lr * T, gr = gr0 + lr;
            if (gr >= 0 && gr < S) in0[(t + 1) * col_stride + lr] = a.in_plane[((size_t)n * S + gr) * T + t];
        }
    } else if (a.mode == 0 && a.in_plane != nullptr) {
        // the upsampled planes were computed by upsample_planes_kernel (one product over all planes): copy the band's rows
        for (int i = tid; i < LR * T; i += kConvThreads) {
            const int lr = i / T, t = i - lr * T, gr = gr0 + lr;
            if (gr >= 0 && gr < S) in0[(t + 1) * col_stride + lr] = a.in_plane[((size_t)n * S + gr) * T + t];
        }
    } else if (a.mode == 0) {   // pilot_upsampler row (gr*T + t): idx = sc*T + sym (view(B,1,S,T))
        if (FIXED && a.pf == 24) {
            // Default grid and pilots (round 3): up_w [1680][24] is 161 KB that EVERY workgroup streams through its CU's L1.
            // Round 2 gave each lane whole 96-byte rows of pixels 14 rows apart: a wave's load instruction touched 64
            // different cache lines for 16 bytes each (8 100 cycles for this phase, TA-bound).  Now a wave reads the matrix
            // as it lies in memory -- three 1-KB loads = 192 consecutive float4 = the rows of 32 consecutive pixels --,
            // each lane multiplies its float4 with the matching quarter of the pilot vector, the 192 partial sums meet in
            // LDS and lanes 0..31 add the six partials of "their" pixel.  Scratch: interior columns of conv1's channel-0
            // plane (conv1 rewrites every entry of those columns in the next phase, behind the barrier).
            constexpr int kPix = 120 * 14, kChunks = (kPix + 31) / 32;
            const int lane_i = tid & 63, wave_i = tid >> 6;
            float *scr = c1 + col_stride + wave_i * 192;              // 8 waves x 192 floats = 12 columns of 128
            const int l6 = lane_i % 6;
            f32x4 pq[3];                                              // pilot quarter of float4 (64 i + lane) % 6
#pragma unroll
            for (int i = 0; i < 3; ++i) pq[i] = *reinterpret_cast<const f32x4 *>(small + 4 * ((l6 + 4 * i) % 6));
            const f32x4 *w4 = reinterpret_cast<const f32x4 *>(a.up_w);
            // the stream is latency-bound per wave (each round waits for its own three loads: 7 rounds x an L2 round trip
            // was 7 200 cycles): the loads of the next kAhead rounds are in flight while a round is reduced
            constexpr int kRounds = (kChunks + kConvWaves - 1) / kConvWaves, kAhead = 3;
            f32x4 wv[kAhead + 1][3];
            float bias[kAhead + 1];
            auto request = [&](int rnd) {
                const int ch = wave_i + rnd * kConvWaves;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int f = ch * 192 + 64 * i + lane_i;         // chunk ch = float4 [192 ch, 192 ch + 192) = pixels [32 ch, 32 ch + 32)
                    wv[rnd % (kAhead + 1)][i] = f < kPix * 6 ? w4[f] : f32x4{0.f, 0.f, 0.f, 0.f};
                }
                const int pix = 32 * ch + (lane_i & 31);
                bias[rnd % (kAhead + 1)] = (lane_i < 32 && pix < kPix) ? a.up_b[pix] : 0.f;
            };
#pragma unroll
            for (int rnd = 0; rnd < kAhead && rnd < kRounds; ++rnd) request(rnd);
#pragma unroll
            for (int rnd = 0; rnd < kRounds; ++rnd) {
                if (rnd + kAhead < kRounds) request(rnd + kAhead);
                const int ch = wave_i + rnd * kConvWaves;
                const int pix = 32 * ch + (lane_i & 31);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    // four-term quarter sums, added in q order below (bias first)
                    const f32x4 w = wv[rnd % (kAhead + 1)][i];
                    float v = w[0] * pq[i][0];
                    v = fmaf(w[1], pq[i][1], v);
                    v = fmaf(w[2], pq[i][2], v);
                    v = fmaf(w[3], pq[i][3], v);
                    scr[64 * i + lane_i] = v;
                }
                __builtin_amdgcn_wave_barrier();                      // wave-private scratch: LDS ops of one wave complete in order
                if (lane_i < 32 && pix < kPix) {
                    const float *pp = scr + 6 * lane_i;
                    float v = bias[rnd % (kAhead + 1)];
#pragma unroll
                    for (int q = 0; q < 6; ++q) v += pp[q];
                    const int gr = pix / 14, t = pix - gr * 14;       // view(B, 1, S, T): pixel = sc * T + sym
                    in0[(t + 1) * col_stride + (gr - gr0)] = v;
                }
                __builtin_amdgcn_wave_barrier();
            }
        } else if (a.pf == 24) {   // default pilot grid: four pixels per pass, all 24 row loads in flight before the FMAs
            for (int i0 = tid; i0 < LR * T; i0 += 4 * kConvThreads) {
                f32x4 wv[4][6];
                float bv[4];
                int dsti[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = i0 + u * kConvThreads, t = i / LR, lr = i - t * LR, gr = gr0 + lr;
                    const bool ok = i < LR * T && gr >= 0 && gr < S;
                    const int pix = ok ? gr * T + t : 0;
                    dsti[u] = ok ? (t + 1) * col_stride + lr : -1;
                    const f32x4 *wr = reinterpret_cast<const f32x4 *>(a.up_w + (size_t)pix * 24);
#pragma unroll
                    for (int q = 0; q < 6; ++q) wv[u][q] = wr[q];
                    bv[u] = a.up_b[pix];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float v = bv[u];
#pragma unroll
                    for (int q = 0; q < 6; ++q) {
                        const f32x4 pv = *reinterpret_cast<const f32x4 *>(small + 4 * q);
                        v = fmaf(wv[u][q][0], pv[0], v);
                        v = fmaf(wv[u][q][1], pv[1], v);
                        v = fmaf(wv[u][q][2], pv[2], v);
                        v = fmaf(wv[u][q][3], pv[3], v);
                    }
                    if (dsti[u] >= 0) in0[dsti[u]] = v;
                }
            }
        } else if (a.stream_ok) {
            // Any pilot count that is a multiple of 8 (round 3; config 5 has 96): the band's rows of up_w are one contiguous range,
            // streamed as it lies in memory like the default grid's -- a wave takes 32 pixels = 32 nq float4 (nq = pf / 4), every
            // lane multiplies its float4 with the matching quarter of the pilot vector, the partial sums meet in the wave's scratch
            // (row stride nq + 1: conflict-free) and lanes 0..31 add the nq partials of "their" pixel in k order.  Before, each
            // lane walked whole rows 4 pf bytes apart with scalar loads: 44 600 of config 5's 158 000 cycles per band.
            // Scratch: the conv1 / conv3 planes from conv1's column 1 on; it is zeroed again below (it covers border columns).
            const int nq = a.pf >> 2, nld = nq >> 1;                  // float4 per pixel; 64-lane loads per 32-pixel chunk
            const int lane_i = tid & 63, wave_i = tid >> 6;
            const int g_lo = max(gr0, 0), g_hi = min(gr0 + LR, S);    // band rows inside the plane
            const int pix0 = g_lo * T, npix = (g_hi - g_lo) * T, nchunk = (npix + 31) >> 5;
            const int sstride = 32 * (nq + 1);
            float *scr = c1 + col_stride + wave_i * sstride;
            const f32x4 *w4 = reinterpret_cast<const f32x4 *>(a.up_w) + (size_t)pix0 * nq;
            const int total4 = npix * nq;
            constexpr int kGroup = 6;                                 // loads in flight per wave (each waits an L2 round trip)
            for (int ch = wave_i; ch < nchunk; ch += kConvWaves) {
                const int base4 = ch * 32 * nq;
                for (int i0 = 0; i0 < nld; i0 += kGroup) {
                    f32x4 wv[kGroup];
#pragma unroll
                    for (int u = 0; u < kGroup; ++u) {
                        const int f = 64 * (i0 + u) + lane_i;         // float4 index inside the chunk: pixel f / nq, quarter f % nq
                        wv[u] = (i0 + u < nld && base4 + f < total4) ? w4[base4 + f] : f32x4{0.f, 0.f, 0.f, 0.f};
                    }
#pragma unroll
                    for (int u = 0; u < kGroup; ++u) {
                        if (i0 + u >= nld) break;
                        const int f = 64 * (i0 + u) + lane_i;
                        const int pq = f / nq, q = f - pq * nq;
                        const f32x4 pv = *reinterpret_cast<const f32x4 *>(small + 4 * q);
                        float v = wv[u][0] * pv[0];
                        v = fmaf(wv[u][1], pv[1], v);
                        v = fmaf(wv[u][2], pv[2], v);
                        v = fmaf(wv[u][3], pv[3], v);
                        scr[pq * (nq + 1) + q] = v;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                __builtin_amdgcn_wave_barrier();                      // wave-private scratch: LDS ops of one wave complete in order
                const int pl = 32 * ch + (lane_i & 31);
                if (lane_i < 32 && pl < npix) {
                    const float *pp = scr + lane_i * (nq + 1);
                    float v = a.up_b[pix0 + pl];
                    for (int q = 0; q < nq; ++q) v += pp[q];
                    const int gr = g_lo + pl / T, t = pl - (pl / T) * T;
                    in0[(t + 1) * col_stride + (gr - gr0)] = v;
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            __syncthreads();
            for (int i = tid; i < kConvWaves * sstride; i += kConvThreads) c1[col_stride + i] = 0.f;
        } else {
            for (int i = tid; i < LR * T; i += kConvThreads) {
                const int t = i / LR, lr = i - t * LR, gr = gr0 + lr;
                if (gr < 0 || gr >= S) continue;
                const int pix = gr * T + t;
                const float *wr = a.up_w + (size_t)pix * a.pf;
                float v = a.up_b[pix];
                for (int kk = 0; kk < a.pf; ++kk) v = fmaf(wr[kk], small[kk], v);
                in0[(t + 1) * col_stride + lr] = v;
            }
        }
    } else if (!TRAIN && a.lin2_out != nullptr) {
        // inverse patch map + conv_enhanced residual on the linear_2 output of the last chain launch: feature f of
        // token (g, tc) is pixel (g*p0 + f/p1, tc*p1 + f%p1); one thread per pixel of the band
        const int p0 = a.p0, p1 = a.p1, tpr = T / p1;
        for (int i = tid; i < LR * T; i += kConvThreads) {
            const int lr = i / T, t = i - lr * T, gr = gr0 + lr;
            if (gr < 0 || gr >= S) continue;
            const int g = gr / p0, tc = t / p1, f = (gr - g * p0) * p1 + (t - tc * p1);
            in0[(t + 1) * col_stride + lr] = a.lin2_out[((size_t)n * a.tokens + g * tpr + tc) * a.lin2_stride + f] +
                                             a.resid[((size_t)n * S + gr) * T + t];
        }
    } else if (!TRAIN) {
        // linear_2 + inverse patch map + conv_enhanced residual.  One thread per token: its x row is read
        // once (d floats) and dotted with the p weight rows (LDS, wave-uniform address -> broadcast), in
        // groups of up to 8 features.  Feature f of token (g, tc) is pixel (g*p0 + f/p1, tc*p1 + f%p1).
        const int p0 = a.p0, p1 = a.p1, p = p0 * p1, tpr = T / p1;
        const int g_lo = max(gr0, 0) / p0, g_hi = (min(gr0 + LR, S) - 1) / p0;
        const int items = (g_hi - g_lo + 1) * tpr;
        for (int i = tid; i < items; i += kConvThreads) {
            const int g = g_lo + i / tpr, tc = i % tpr;
            const float *xr = a.x + ((size_t)n * a.tokens + g * tpr + tc) * a.d;
            for (int f0 = 0; f0 < p; f0 += 8) {
                float acc[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) acc[q] = f0 + q < p ? small[p * a.d + f0 + q] : 0.f;
                for (int e0 = 0; e0 < a.d; e0 += 16) {
                    f32x4 xv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) xv[u] = *reinterpret_cast<const f32x4 *>(xr + e0 + 4 * u);
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        if (f0 + q >= p) break;
                        const float *wr = small + (f0 + q) * a.d + e0;
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const f32x4 wv = *reinterpret_cast<const f32x4 *>(wr + 4 * u);
                            acc[q] = fmaf(xv[u][0], wv[0], acc[q]);
                            acc[q] = fmaf(xv[u][1], wv[1], acc[q]);
                            acc[q] = fmaf(xv[u][2], wv[2], acc[q]);
                            acc[q] = fmaf(xv[u][3], wv[3], acc[q]);
                        }
                    }
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int f = f0 + q;
                    if (f >= p) break;
                    const int gr = g * p0 + f / p1, t = tc * p1 + f % p1, lr = gr - gr0;
                    if (lr >= 0 && lr < LR && gr < S)
                        in0[(t + 1) * col_stride + lr] = acc[q] + a.resid[((size_t)n * S + gr) * T + t];
                }
            }
        }
    }
    __syncthreads();
    CSTAMP(3);

    // ---- conv1: 1 -> 8, ReLU, rows [1, LR-1); one thread = 4 consecutive rows of one column (a 6 x 3 window read once).
    //      The 80 weights / biases come from LDS into VECTOR registers: as scalar operands they overflowed the SGPR file
    //      (205 v_readlane per thread), and the store addresses advance by one plane per channel (no multiplies). ----
    {
        float w1[80];
#pragma unroll
        for (int q = 0; q < 20; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(w1s + 4 * q);
#pragma unroll
            for (int c = 0; c < 4; ++c) w1[4 * q + c] = v[c];
        }
        const int nrg = (LR - 1 + 3) >> 2;   // row groups [4 rg, 4 rg + 4)
        for (int i = tid; i < nrg * T; i += kConvThreads) {
            const int t = i / nrg, lr0 = 4 * (i - t * nrg);
            float win[6][3];  // [row lr0 - 1 + r][kx]
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float *col = in0 + (t + kx) * col_stride;
#pragma unroll
                for (int r = 0; r < 6; ++r) win[r][kx] = col[min(max(lr0 - 1 + r, 0), LR - 1)];
            }
            bool okq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int lr = lr0 + q, gr = gr0 + lr;
                okq[q] = lr >= 1 && lr < LR - 1 && gr >= 0 && gr < S;
            }
            const bool whole = lr0 + 3 < LR;   // all four rows inside the column: unconditional stores (zeros where not valid, as zeroed)
            float *dstp = c1 + (t + 1) * col_stride + lr0;
#pragma unroll
            for (int o = 0; o < 8; ++o) {
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float acc = w1[72 + o];
#pragma unroll
                    for (int k9 = 0; k9 < 9; ++k9) acc = fmaf(win[q + k9 / 3][k9 % 3], w1[o * 9 + k9], acc);
                    v[q] = fmaxf(acc, 0.f);
                    if constexpr (TRAIN) {
                        if (okq[q]) {
                            const size_t gi = ((size_t)(n * 8 + o) * T + t) * S + gr0 + lr0 + q;
                            if (a.mask[0]) v[q] = a.mask[0][gi] > 0.f ? acc : 0.f;
                            if (a.save[0] && lr0 + q >= 4 && lr0 + q < 4 + BR) a.save[0][gi] = v[q];
                        }
                    }
                    v[q] = okq[q] ? v[q] : 0.f;
                }
                if (whole) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) dstp[q] = v[q];
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (okq[q]) dstp[q] = v[q];
                }
                dstp += plane;
            }
        }
    }
    __syncthreads();
    CSTAMP(4);

    // ---- conv2 + conv3 on the matrix cores ----
    const int r3lo = band == 0 ? 4 : 3;   // first local row whose conv3 output is needed and inside the plane
    for (int task = wave; task < NT * NSEG; task += kConvWaves) {
        const int tile = task / NSEG, seg = task - tile * NSEG;
        const int ta = seg * T / NSEG, tb = (seg + 1) * T / NSEG;   // output columns [ta, tb)
        // Inference: every conv2 column is computed ONCE.  A wave sweeps exactly its own columns [ta, tb); the two
        // output columns next to a segment seam then lack one kx tap each: the owner stores its raw partial sum in place
        // (c3), the neighbour stores the missing tap's contribution in the exchange planes (the dead input plane), and a
        // small fix-up pass behind the barrier adds them and applies bias + ReLU.  14 column sweeps for 14 columns
        // instead of 16 (the training instantiation keeps the overlapping sweep: its stage outputs are masked / saved
        // in store_col).
        // Seam s between segments s and s + 1 has two exchange slots of 8 channel rows: side 0 = column tb_s - 1 (owner: the left
        // segment), side 1 = column tb_s (owner: the right one).  They live in the input plane (dead after conv1) when its T + 2
        // row vectors hold them, else behind the arena (plan_bands).  Grids whose bands have two row tiles run FOUR column
        // segments this way (round 3: all eight waves in the matrix phase, every column still swept once).
        const bool EXACT = !TRAIN && NSEG >= 2 && (FIXED || a.xoff >= 0);
        float *xch = smem + (FIXED ? 0 : a.xoff);
        const int tlo = EXACT ? ta : max(ta - 1, 0), thi = EXACT ? tb - 1 : min(tb, T - 1);   // conv2 columns swept
        const int r = r3lo + kTileRows * tile - 1 + j;                   // this lane's local row
        const int gr = gr0 + r;
        const bool ok2 = gr >= 0 && gr < S;                              // conv2 output inside the plane (else zero padding)
        const float relu_hi = ok2 ? __builtin_inff() : 0.f;
        const bool ok3 = ok2 && j >= 1 && j <= kTileRows && r < LR - 3;
        const float *bsrc = c1 + (4 * h) * plane + r - 1;                // + c*plane + (t'+kx)*col_stride + ky
        float *dst = c3 + (4 * h) * plane + r;

        f32x16 acc3;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc3[e] = 0.f;

        const bool own_row = r >= 4 && r < 4 + BR;   // rows this band is responsible for (no halo)
        auto store_col = [&](int tout, float v0, float v1, float v2, float v3) {
            if (EXACT) {
                if (!ok3 || tout < 0 || tout >= T) return;
                const bool mine = tout >= ta && tout < tb;
                const bool seam = (tout == ta && ta > 0) || (tout == tb - 1 && tb < T);   // own column next to a seam
                if (!mine || seam) {
                    if (!mine && !(tout == ta - 1 || tout == tb)) return;
                    // raw sums: own seam column -> c3 in place, neighbour's seam column -> its exchange slot (seam, side)
                    const int slot = tout == ta - 1 ? (seg - 1) * 2 : seg * 2 + 1;
                    float *p = mine ? dst + (tout + 1) * col_stride : xch + (slot * 8 + 4 * h) * SP + r;
                    const int cs = mine ? plane : SP;
                    p[0] = v0; p[cs] = v1; p[2 * cs] = v2; p[3 * cs] = v3;
                    return;
                }
            }
            if (tout >= ta && tout < tb && ok3) {
                float *p = dst + (tout + 1) * col_stride;
                float v[4] = {v0, v1, v2, v3};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float y = fmaxf(v[k] + bias3[k], 0.f);
                    if constexpr (TRAIN) {
                        const unsigned gi = ((unsigned)(n * 8 + 4 * h + k) * T + tout) * S + gr;
                        if (a.mask[2]) y = conv_ld(conv_srd(a.mask[2]), gi) > 0.f ? v[k] : 0.f;
                        if (a.save[2] && own_row) conv_st(conv_srd(a.save[2]), gi, y);
                    }
                    p[k * plane] = y;
                }
            }
        };
        // B operand slot kb of conv2 column tcol: channel 4h + (kb&3), tap (kx, ky) = ((kb>>2)/3, (kb>>2)%3)
        auto b_at = [&](int kb, int tcol) {
            const int tap = kb >> 2, kx = tap / 3, ky = tap % 3;
            return bsrc[(kb & 3) * plane + (tcol + kx) * col_stride + ky];
        };
        float b[36];
#pragma unroll
        for (int kb = 0; kb < 36; ++kb) b[kb] = b_at(kb, tlo);
        auto bias2_acc = [&]() {
            f32x16 acc;
            const f32x4 *bp = reinterpret_cast<const f32x4 *>(bias2 + 16 * h);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = bp[q];
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[4 * q + u] = v[u];
            }
            return acc;
        };
        // conv2's activation for column tcol: ReLU (or 0 outside the plane); training: masked / saved
        auto activate2 = [&](const f32x16 &acc2, int tcol, float (&x2)[16]) {
#pragma unroll
            for (int e = 0; e < 16; ++e) x2[e] = __builtin_amdgcn_fmed3f(acc2[e], 0.f, relu_hi);   // ReLU, or 0 outside the plane
            if constexpr (TRAIN) {
                if (a.mask[1] || a.save[1]) {
                    const unsigned g0 = ((unsigned)(n * 32 + 4 * h) * T + tcol) * S + gr;   // channel (e&3) + 8(e>>2) + 4h
                    const unsigned cstride = (unsigned)T * S;
                    if (a.mask[1]) {
                        const ConvSrd m = conv_srd(a.mask[1]);
#pragma unroll
                        for (int e = 0; e < 16; ++e)
                            x2[e] = (ok2 && conv_ld(m, g0 + ((e & 3) + 8 * (e >> 2)) * cstride) > 0.f) ? acc2[e] : 0.f;
                    }
                    if (a.save[1] && ok2 && own_row) {
                        const ConvSrd sv = conv_srd(a.save[1]);
#pragma unroll
                        for (int e = 0; e < 16; ++e) conv_st(sv, g0 + ((e & 3) + 8 * (e >> 2)) * cstride, x2[e]);
                    }
                }
            }
        };
        if constexpr (!TRAIN && AFT_CONV_PIPE) {
            // Software pipeline over the columns (round 3): conv3 of column tcol and conv2 of column tcol + 1 are two
            // INDEPENDENT accumulation chains issued alternately, so neither waits for its own previous MFMA: before, a wave
            // ran conv2's 36 dependent MFMAs, waited for the result, applied the ReLU, ran conv3's 48, waited again -- with the
            // SIMD's other wave in the same rhythm the matrix pipe was 81 % busy during this phase (DESIGN.md 4.3 stamps).
            float x2[16];
            {   // prologue: conv2 of the first column
                f32x16 acc2 = bias2_acc();
                const int tnext = min(tlo + 1, thi);
#pragma unroll
                for (int kb = 0; kb < 36; ++kb) {
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa2[kb], b[kb], acc2, 0, 0, 0);
                    b[kb] = b_at(kb, tnext);
                }
                activate2(acc2, tlo, x2);
            }
            // conv3 MFMA number i of a column: ky = centre (16..31), below (0..15), above (32..47) as before
            auto conv3_step = [&](int i) {
                const int e = i & 15;
                const float xv = i < 16 ? x2[e] : (i < 32 ? lane_from_below(x2[e]) : lane_from_above(x2[e]));
                const int wi = i < 16 ? 16 + e : (i < 32 ? e : 32 + e);
                acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa3[wi], xv, acc3, 0, 0, 0);
            };
#pragma unroll 1
            for (int tcol = tlo; tcol < thi; ++tcol) {
                // registers 8..11 hold output column tcol-2, complete since the previous column's MFMAs
                store_col(tcol - 2, acc3[8], acc3[9], acc3[10], acc3[11]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc3[8 + e] = acc3[4 + e];
                    acc3[4 + e] = acc3[e];
                    acc3[e] = 0.f;
                }
                f32x16 acc2 = bias2_acc();
                const int tnext = min(tcol + 2, thi);
                // 48 conv3 MFMAs of column tcol interleaved with the 36 conv2 MFMAs of column tcol + 1 (4 : 3); each conv2 operand
                // register is refilled for the column after as soon as its MFMA has issued
#pragma unroll
                for (int g = 0; g < 12; ++g) {
                    conv3_step(4 * g);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa2[3 * g], b[3 * g], acc2, 0, 0, 0);
                    b[3 * g] = b_at(3 * g, tnext);
                    conv3_step(4 * g + 1);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa2[3 * g + 1], b[3 * g + 1], acc2, 0, 0, 0);
                    b[3 * g + 1] = b_at(3 * g + 1, tnext);
                    conv3_step(4 * g + 2);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa2[3 * g + 2], b[3 * g + 2], acc2, 0, 0, 0);
                    b[3 * g + 2] = b_at(3 * g + 2, tnext);
                    conv3_step(4 * g + 3);
                }
                activate2(acc2, tcol + 1, x2);
            }
            {   // epilogue: conv3 of the last column
                store_col(thi - 2, acc3[8], acc3[9], acc3[10], acc3[11]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc3[8 + e] = acc3[4 + e];
                    acc3[4 + e] = acc3[e];
                    acc3[e] = 0.f;
                }
#pragma unroll
                for (int i = 0; i < 48; ++i) conv3_step(i);
            }
        } else {
            // training instantiation: the plain column loop, as in round 2 (the pipelined form needs 16 more registers)
#pragma unroll 1
            for (int tcol = tlo; tcol <= thi; ++tcol) {
                f32x16 acc2;
                {
                    const f32x4 *bp = reinterpret_cast<const f32x4 *>(bias2 + 16 * h);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 v = bp[q];
#pragma unroll
                        for (int u = 0; u < 4; ++u) acc2[4 * q + u] = v[u];
                    }
                }
                // conv2: each operand register is refilled for the next column as soon as its MFMA has issued
                const int tnext = min(tcol + 1, thi);
#pragma unroll
                for (int kb = 0; kb < 36; ++kb) {
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa2[kb], b[kb], acc2, 0, 0, 0);
                    b[kb] = b_at(kb, tnext);
                }
                // registers 8..11 hold output column tcol-2, complete since the previous column's MFMAs
                store_col(tcol - 2, acc3[8], acc3[9], acc3[10], acc3[11]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc3[8 + e] = acc3[4 + e];
                    acc3[4 + e] = acc3[e];
                    acc3[e] = 0.f;
                }
                float x2[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) x2[e] = __builtin_amdgcn_fmed3f(acc2[e], 0.f, relu_hi);   // ReLU, or 0 outside the plane
                if constexpr (TRAIN) {
                    if (a.mask[1] || a.save[1]) {
                        const unsigned g0 = ((unsigned)(n * 32 + 4 * h) * T + tcol) * S + gr;   // channel (e&3) + 8(e>>2) + 4h
                        const unsigned cstride = (unsigned)T * S;
                        if (a.mask[1]) {
                            const ConvSrd m = conv_srd(a.mask[1]);
#pragma unroll
                            for (int e = 0; e < 16; ++e)
                                x2[e] = (ok2 && conv_ld(m, g0 + ((e & 3) + 8 * (e >> 2)) * cstride) > 0.f) ? acc2[e] : 0.f;
                        }
                        if (a.save[1] && ok2 && own_row) {
                            const ConvSrd sv = conv_srd(a.save[1]);
#pragma unroll
                            for (int e = 0; e < 16; ++e) conv_st(sv, g0 + ((e & 3) + 8 * (e >> 2)) * cstride, x2[e]);
                        }
                    }
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa3[16 + e], x2[e], acc3, 0, 0, 0);
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa3[e], lane_from_below(x2[e]), acc3, 0, 0, 0);
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa3[32 + e], lane_from_above(x2[e]), acc3, 0, 0, 0);
            }
        }
        store_col(thi - 1, acc3[8], acc3[9], acc3[10], acc3[11]);
        store_col(thi, acc3[4], acc3[5], acc3[6], acc3[7]);   // overlapping sweep: only stored when thi == T-1 (column T is zero padding)
        if (EXACT) store_col(thi + 1, acc3[0], acc3[1], acc3[2], acc3[3]);   // the kx = 0 tap of column thi for the right neighbour
    }
    CSTAMP(5);
    __syncthreads();
    CSTAMP(6);
    if constexpr (!TRAIN) {
        // seam fix-up: c3[ch][t][row] = ReLU(own partial + neighbour's tap + bias) for the two columns at each seam
        const int nseam = (NSEG >= 2 && (FIXED || a.xoff >= 0)) ? NSEG - 1 : 0;
        if (nseam > 0) {
            const float *xch = smem + (FIXED ? 0 : a.xoff);
            for (int i = tid; i < nseam * 2 * 8 * (LR - 6); i += kConvThreads) {
                const int lr = 3 + i % (LR - 6), q = i / (LR - 6), ch = q & 7, side = (q >> 3) & 1, sm = q >> 4;
                const int t = (sm + 1) * T / NSEG - 1 + side;   // tb - 1 of the left segment, ta of the right one
                float *pc = c3 + ch * plane + (t + 1) * col_stride + lr;
                const int gr = gr0 + lr;
                if (gr >= 0 && gr < S) *pc = fmaxf(*pc + xch[((sm * 2 + side) * 8 + ch) * SP + lr] + a.cb[2][ch], 0.f);
            }
            __syncthreads();
        }
    }

    CSTAMP(7);
    // ---- conv4: 8 -> 1, no activation, rows of this band only; one thread = one row x 4 columns ----
    const int tstrips = (T + 3) >> 2;
    for (int i = tid; i < BR * tstrips; i += kConvThreads) {
        const int ts = i / BR, lr = 4 + i - ts * BR, gr = gr0 + lr, t0 = 4 * ts;
        if (gr >= S) continue;
        float acc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = (TRAIN && !a.cb[3]) ? 0.f : a.cb[3][0];
#pragma unroll 2
        for (int ci = 0; ci < 8; ++ci) {
            float win[3][6];   // [ky][column t0-1 .. t0+4]; columns past T+1 are never used by a stored output
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int c = 0; c < 6; ++c) win[ky][c] = c3[ci * plane + min(t0 + c, T + 1) * col_stride + lr - 1 + ky];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float w = a.cw[3][ci * 9 + ky * 3 + kx];
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[q] = fmaf(win[ky][q + kx], w, acc[q]);
                }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int t = t0 + q;
            if (t >= T) break;
            if (TRAIN || a.mode == 0)
                a.out_plane[((size_t)n * S + gr) * T + t] = acc[q];
            else
                a.out_complex[(((size_t)frame * S + gr) * T + t) * 2 + part] = acc[q];
        }
    }
    CSTAMP(8);
#undef CSTAMP
}

// Largest row band (divides S) whose 17 channel planes fit 160 KB of LDS.  A band needs conv3 rows
// [-1, band_rows+1) around it, covered by 30-row tiles; SP spans the last tile's halo lane.
static bool plan_bands(int S, int T, int extra_floats, ConvArgs *a, size_t *lds_bytes, bool train = false) {
    const int extra = (extra_floats + 3) & ~3;
    for (int nb = 1; nb <= S; ++nb) {
        if (S % nb) continue;
        const int br = S / nb;
        const int rows3 = nb == 1 ? br : br + 2;
        const int ntiles = (rows3 + kTileRows - 1) / kTileRows;
        const int sp = std::max(br + 8, 4 + kTileRows * ntiles + 2);
        const int arena = (std::max(17 * (T + 2) * sp, kWStage) + 3) & ~3;   // also stages the conv2/conv3 weights
        const size_t bytes = sizeof(float) * ((size_t)arena + extra + 32 + 80);   // + conv2 bias in register order (32) + conv1 weights and biases (80)
        if (bytes <= 160 * 1024) {
            a->band_rows = br;
            a->nbands = nb;
            a->ntiles = ntiles;
            a->nseg = T >= 8 ? 2 : 1;
            a->SP = sp;
            a->arena = arena;
            a->extra = extra;
            a->xoff = (!train && a->nseg == 2 && T + 2 >= 16) ? 0 : -1;   // one seam: its 16 exchange rows are the dead input plane
            *lds_bytes = bytes;
            // Bands of one or two row tiles leave half of the eight waves without a task: four column segments instead (inference
            // only: the three seams' 48 exchange rows go behind the arena when the input plane has fewer row vectors)
            if (!train && ntiles * 2 <= kConvWaves / 2 && T >= 16) {
                const int xrows = 3 * 16;
                const int xoff = xrows <= T + 2 ? 0 : ((arena + extra + 32 + 80 + 3) & ~3);
                const size_t b4 = xoff == 0 ? bytes : sizeof(float) * ((size_t)xoff + (size_t)xrows * sp);
                if (b4 <= 160 * 1024) {
                    a->nseg = 4;
                    a->xoff = xoff;
                    *lds_bytes = b4;
                }
            }
            return true;
        }
    }
    return false;
}

bool conv_plan_ok(int S, int T, int extra_floats) {
    ConvArgs a{};
    size_t lds = 0;
    return S > 0 && T > 0 && plan_bands(S, T, extra_floats, &a, &lds);
}

template <bool TRAIN, bool FIXED>
static hipError_t launch_conv_geo(ConvArgs &a, int planes, size_t lds, hipStream_t st);

template <bool TRAIN>
static hipError_t launch_conv(ConvArgs &a, int planes, int extra_floats, hipStream_t st) {
    size_t lds = 0;
    if (!plan_bands(a.S, a.T, extra_floats, &a, &lds, TRAIN)) return hipErrorInvalidValue;
    if (a.stream_ok < 0)   // streaming upsampler: 8 waves x 32 pixels x (pf / 4 + 1) partial sums inside the 16 conv1 / conv3 planes
        a.stream_ok = (a.pf % 8 == 0 && a.up_w && (reinterpret_cast<uintptr_t>(a.up_w) & 15) == 0 &&
                       (size_t)kConvWaves * 32 * (a.pf / 4 + 1) + a.SP <= (size_t)16 * (a.T + 2) * a.SP) ? 1 : 0;
    const bool fixed = a.S == 120 && a.T == 14 && a.SP == 128 && a.band_rows == 120 && a.nbands == 1 && a.ntiles == 4 && a.nseg == 2;
    // default grid, inference, inputs as the whole forward provides them: the column-streaming pipeline (k_conv_stream.hip);
    // AFT_CONV_BANDED=1 keeps the banded kernel (A/B runs)
    if (fixed && conv_stream_ok(a) && !switch_on("AFT_CONV_BANDED")) return launch_conv_stream(a, planes, st);   // (training: mode 2)
    if (fixed) return launch_conv_geo<TRAIN, true>(a, planes, lds, st);
    // planes that need several bands here (config 5: five bands of two row tiles): the whole-height column-streaming kernel
    // (k_conv_rows.hip) when its shape conditions hold; AFT_CONV_BANDED=1 keeps the banded kernel (A/B runs)
    if (!TRAIN && a.nbands > 1 && conv_rows_ok(a, planes) && !switch_on("AFT_CONV_BANDED")) return launch_conv_rows(a, planes, st);
    return launch_conv_geo<TRAIN, false>(a, planes, lds, st);
}

template <bool TRAIN, bool FIXED>
static hipError_t launch_conv_geo(ConvArgs &a, int planes, size_t lds, hipStream_t st) {
    static PerDeviceOnce lds_attr;   // per instantiation x device
    hipError_t ea = ensure_dynamic_lds(lds_attr, reinterpret_cast<const void *>(conv_stack_kernel<TRAIN, FIXED>), 160 * 1024);
    if (ea != hipSuccess) return ea;
#ifdef AFT_DIAG_STAMPS
    if (!TRAIN && switch_on("AFT_STAMPS")) {   // diagnostic build only: mean cycles per phase (thread 0 of every workgroup)
        static unsigned long long *dbuf = nullptr;
        const int nb = planes * a.nbands;
        if (!dbuf) (void)hipMalloc(&dbuf, sizeof(unsigned long long) * 16 * 4096);
        (void)hipMemset(dbuf, 0, sizeof(unsigned long long) * 16 * 4096);
        a.stamps = dbuf;
        hipLaunchKernelGGL((conv_stack_kernel<TRAIN, FIXED>), dim3(nb), dim3(kConvThreads), lds, st, a);
        (void)hipDeviceSynchronize();
        static int printed = 0;
        if (printed++ < 4 && nb <= 4096) {
            std::vector<unsigned long long> hb(16 * (size_t)nb);
            (void)hipMemcpy(hb.data(), dbuf, hb.size() * 8, hipMemcpyDeviceToHost);
            double sum[9] = {0};
            for (int b = 0; b < nb; ++b)
                for (int i = 1; i < 9; ++i) sum[i] += (double)(hb[(size_t)b * 16 + i] - hb[(size_t)b * 16 + i - 1]);
            printf("conv mode %d stamps (mean cycles, thread 0): weights=%.0f zero=%.0f input=%.0f conv1=%.0f mfma(wave0)=%.0f "
                   "wait=%.0f fixup=%.0f conv4=%.0f total=%.0f\n", a.mode, sum[1] / nb, sum[2] / nb, sum[3] / nb, sum[4] / nb, sum[5] / nb,
                   sum[6] / nb, sum[7] / nb, sum[8] / nb, (sum[1] + sum[2] + sum[3] + sum[4] + sum[5] + sum[6] + sum[7] + sum[8]) / nb);
        }
        a.stamps = nullptr;
        return hipGetLastError();
    }
#endif
    hipLaunchKernelGGL((conv_stack_kernel<TRAIN, FIXED>), dim3(planes * a.nbands), dim3(kConvThreads), lds, st, a);
    return hipGetLastError();
}

// Training path: ConvEnhancer on plain planes.  Forward: w/b = the module's tensors, save = {c1, c2, c3},
// mask = {}.  Backward (dgrad): x = dL/dy, w = transposed + flipped weights of conv4..conv1, b = {},
// mask = {c3, c2, c1}, save = the pre-activation gradients {g3, g2, g1}; y = dL/dx.
hipError_t launch_conv_train(const float *const w[4], const float *const b[4], const float *x, float *y, float *const save[3],
                             const float *const mask[3], int planes, int S, int T, hipStream_t st, float *frag) {
    ConvArgs a{};
    a.mode = 2;
    a.S = S; a.T = T;
    a.in_plane = x; a.out_plane = y;
    for (int i = 0; i < 4; ++i) { a.cw[i] = w[i]; a.cb[i] = b ? b[i] : nullptr; }
    for (int i = 0; i < 3; ++i) { a.save[i] = save ? save[i] : nullptr; a.mask[i] = mask ? mask[i] : nullptr; }
    if (frag != nullptr && S == 120 && T == 14) {   // the default grid's 16x16x4 training kernel reads the weights as operand fragments
        hipError_t e = launch_conv_frag_pack(w, b, frag, st);
        if (e != hipSuccess) return e;
        a.wfrag = frag;
    }
    return launch_conv<true>(a, planes, 0, st);
}

// pilot_upsampler (reference fortitran.py:86,203-206: Linear(Ps*Pt -> S*T) on the Re and the Im plane) as ONE product over all
// planes: planes[n][pix] = up_b[pix] + sum_k up_w[pix][k] * pilot[n][k], n = 2 frame + part.  Used for grids other than the
// default one when the caller can lend a scratch buffer: inside the conv head every (plane, band) workgroup streams its rows of
// up_w again -- config 5: 602 KB per workgroup, 385 MB per launch out of the L2s, 34 500 of its 147 000 cycles.
// One workgroup = 64 pixels x 64 planes: the 64 rows of up_w (coalesced) and the 64 pilot vectors sit in LDS, thread
// (pixel, plane group g) accumulates 16 planes; k runs 0 .. pf-1 in order with the bias first (the per-pixel loop's order).
__global__ __launch_bounds__(256) void upsample_planes_kernel(const float *__restrict__ up_w, const float *__restrict__ up_b,
                                                              const float *__restrict__ pilots, float *__restrict__ planes_out,
                                                              int npix, int pf, int nplanes) {
    extern __shared__ __attribute__((aligned(16))) float ups[];
    upsample_planes_body(ups, up_w, up_b, pilots, planes_out, npix, pf, nplanes, blockIdx.x, blockIdx.y);
}

hipError_t launch_upsample(const aft_config &c, const WeightsDev &w, const float *pilots, float *conv_enhanced,
                           int batch, hipStream_t st, float *scratch_planes, bool planes_ready, const float *conv_frag) {
    static_assert(kConvFragFloats == (size_t)kFragFloats, "fragment image size");
    ConvArgs a{};
    a.mode = 0;
    a.wfrag = conv_frag;
    a.S = c.num_scs; a.T = c.num_symbols;
    a.pilots = pilots; a.up_w = w.up_w; a.up_b = w.up_b;
    a.pf = c.pilot_scs * c.pilot_symbols;
    for (int i = 0; i < 4; ++i) { a.cw[i] = w.enh_w[i]; a.cb[i] = w.enh_b[i]; }
    a.out_plane = conv_enhanced;
    a.stream_ok = -1;   // decided by launch_conv once the band plan is known
    // With a scratch buffer the pilot_upsampler is ONE product over all planes (up_w read once per launch) and the conv head reads
    // the planes: in the forward that product rides in the prologue launch (planes_ready); here it is its own launch.  Without
    // scratch (the per-stage entry point) the head streams up_w itself.
    if (scratch_planes && upsample_planes_ok(w.up_w, a.pf)) {
        if (!planes_ready) {
            const int npix = c.num_scs * c.num_symbols, nplanes = 2 * batch;
            hipLaunchKernelGGL(upsample_planes_kernel, dim3((npix + kUpPix - 1) / kUpPix, (nplanes + kUpPlanes - 1) / kUpPlanes), dim3(256),
                               upsample_planes_lds(a.pf), st, w.up_w, w.up_b, pilots, scratch_planes, npix, a.pf, nplanes);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return e;
        }
        a.in_plane = scratch_planes;
    }
    return launch_conv<false>(a, 2 * batch, a.pf, st);
}

hipError_t launch_tail(const aft_config &c, const WeightsDev &w, const float *x, const float *conv_enhanced,
                       float *out, int batch, hipStream_t st, const float *out6, const float *conv_frag) {
    ConvArgs a{};
    a.mode = 1;
    a.wfrag = conv_frag;
    a.lin2_out = out6;
    a.lin2_stride = out6_stride(c);
    a.S = c.num_scs; a.T = c.num_symbols;
    a.x = x; a.lin2_w = w.lin2_w; a.lin2_b = w.lin2_b; a.resid = conv_enhanced;
    a.d = c.model_dim; a.p0 = c.patch_scs; a.p1 = c.patch_symbols;
    a.tokens = (c.num_scs / c.patch_scs) * (c.num_symbols / c.patch_symbols);
    for (int i = 0; i < 4; ++i) { a.cw[i] = w.ref_w[i]; a.cb[i] = w.ref_b[i]; }
    a.out_complex = out;
    const int p = a.p0 * a.p1;
    // LDS beside the plane: linear_2's weights when the kernel applies it (x given).  The packed engine keeps that reservation with
    // out6 as well (its band plans, hence its bits, are those of rounds 2-5); the general engine's out6 tail reserves nothing.
    return launch_conv<false>(a, 2 * batch, out6 != nullptr && !packed_engine_ok(c) ? 0 : p * a.d + p, st);
}

}  // namespace aft
