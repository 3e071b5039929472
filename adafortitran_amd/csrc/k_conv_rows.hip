// k_conv_rows.hip -- the two ConvEnhancer stacks for grids whose planes do not fit the LDS (config 5: 240 x 28), as WHOLE-HEIGHT
// workgroups that stream over the symbol columns through ring buffers (late round 4).
//
// Reference semantics: exactly k_conv.hip's / k_conv_stream.hip's (head: reference src/models/fortitran.py:203-209 behind the
// pilot_upsampler product; tail: fortitran.py:225-231,180 on linear_2's output; ConvEnhancer blocks/enhancers.py:12-20) -- the same
// arithmetic per output element: conv1 / conv4 fp32 FMA chains in tap order, conv2 / conv3 the same MFMA chains.
//
// Why a third kernel.  The banded kernel (k_conv.hip) holds all 17 channel planes of a row band in LDS; at 240 x 28 that allows
// bands of two 30-row tiles only: five bands per plane (300 tile rows for 240), 640 workgroups on 256 CUs = three rounds, every
// band with its own serial phases -- 0.34 of the fp32 roof.  Here a workgroup takes ALL rows of the plane (up to
// eight row tiles of 30 rows, one per wave, two waves per SIMD: no bands, no halo rows) and a range of columns, and keeps only what
// the column pipeline needs: the input columns of its range, FOUR columns of conv1's output, FOUR columns of conv3's output (ring
// buffers indexed by symbol & 3) and its own output columns -- 100 KB for config 5.  Planes are split into column ranges when there are
// fewer planes than CUs (config 5 at 64 frames per GPU: 128 planes x 2 halves = 256 workgroups, ONE round); a range recomputes the
// two conv2 and the one conv3 column on either side that its outputs need (16 + 15 column sweeps for 14 columns).
//
// Every wave does the same thing (no helper waves as in k_conv_stream.hip: all eight are needed as matrix waves), one column per
// iteration, one workgroup barrier per iteration:
//     conv1 of symbol t + 4            (VALU; lane = (local row 32 w + j, channel half h))          -> c1 ring
//     conv2 of symbol t + 1 || conv3's share of conv2 column t, as two interleaved MFMA chains exactly as in k_conv_stream.hip
//       (conv2's B operands from the c1 ring, conv3's from registers / DPP); conv3's output column t - 2 -> c3 ring
//     conv4 of symbol t - 4            (VALU)                                                       -> the output columns in LDS
// The stages of one iteration read what earlier iterations wrote and write ring slots nobody reads in this iteration, so the
// barrier at its end is the only synchronisation.
#include <cstdint>
#include <cstdlib>

#include "conv_device.h"

namespace aft {

namespace {

constexpr int kRowsSP = 256;        // local rows per column vector: local row lr <-> plane row lr - 4; eight waves x 32 rows
constexpr int kRing = 4;            // columns held of conv1's / conv3's outputs
constexpr int kRingPlane = kRing * kRowsSP;   // floats per channel of a ring

__device__ __forceinline__ float rows_other_half32(float x) {   // value held by lane (l ^ 32): v_permlane32_swap
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, (threadIdx.x & 32) ? r[0] : r[1]);
}

__device__ __forceinline__ f32x4 rows_mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float rows_from_below16(float v) {   // lane i <- lane i-1 inside each 16-lane row, 0 into lane 0 (DPP row_shr:1)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
}
__device__ __forceinline__ float rows_from_above16(float v) {   // lane i <- lane i+1 inside each 16-lane row, 0 into lane 15 (DPP row_shl:1)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x101, 0xf, 0xf, true));
}

}  // namespace

// LDS (floats): in0 [wmax][SP] | c1 [8][4][SP] | c3 [8][4][SP] (its start stages the conv2 / conv3 weights first) | obuf [S][wcols] |
// bias2 [32] | w1s [80] | w4s [80]
// (checked build: + 16 words behind the tables -- which symbol each slot of the two rings holds, see conv_rows16_kernel)
#ifdef AFT_CHECKED
constexpr int kRingTagFloats = 16;
#else
constexpr int kRingTagFloats = 0;
#endif
__host__ __device__ inline size_t conv_rows_lds_floats(int S, int wcols) {
    return (size_t)(wcols + 8) * kRowsSP + 2 * (size_t)8 * kRingPlane + (size_t)S * wcols + 32 + 80 + 80 + kRingTagFloats;
}

// MODE 0 = head (a.in_plane = upsampled planes [planes][S][T] -> a.out_plane), 1 = tail (a.lin2_out + a.resid -> a.out_complex)
template <int MODE>
__global__ __launch_bounds__(kConvThreads) void conv_rows_kernel(const ConvArgs a, int nsplit) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int SP = kRowsSP;
    const int S = a.S, T = a.T, LR = S + 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const int n = blockIdx.x / nsplit, part_c = blockIdx.x - n * nsplit, frame = n >> 1, part = n & 1;
    const int wcols = (T + nsplit - 1) / nsplit;
    const int ta = part_c * wcols, tb = min(T, ta + wcols);          // output columns of this workgroup
    const int cb = max(ta - 2, 0), ce = min(tb + 2, T);                // conv2 columns swept: [cb, ce)
    const int ci0 = cb - 2, win = ce + 2 - ci0;                        // input columns held: symbols ci0 .. ce + 1
    const int ntiles = (S + kTileRows - 1) / kTileRows;                // row tiles = matrix waves (<= 8)
    float *in0 = smem;
    float *c1 = in0 + (size_t)(wcols + 8) * SP;
    float *c3 = c1 + 8 * kRingPlane;
    float *obuf = c3 + 8 * kRingPlane;
    float *bias2 = obuf + (size_t)S * wcols, *w1s = bias2 + 32, *w4s = w1s + 80;

    // ---- phase 0: conv2 / conv3 weights transposed into c3 (conflict-free gathers, as k_conv_stream.hip), tables, the input columns ----
    {
        float *stage = c3;
        for (int i = tid; i < 2304; i += kConvThreads) {
            stage[(i % 72) * 33 + i / 72] = a.cw[1][i];
            const int rem = i % 288;   // conv3.weight [co 8][ci 32][ky 3][kx 3]
            stage[kW3Off + (rem / 3) * 33 + (rem % 3) * 8 + i / 288] = a.cw[2][i];
        }
        if (tid < 32) bias2[tid] = a.cb[1][(tid & 3) + 8 * ((tid & 15) >> 2) + 4 * (tid >> 4)];
        if (tid >= 64 && tid < 144) w1s[tid - 64] = tid < 136 ? a.cw[0][tid - 64] : a.cb[0][tid - 136];
        if (tid >= 192 && tid < 265) w4s[tid - 192] = tid < 264 ? a.cw[3][tid - 192] : a.cb[3][0];
        // input: element (local row lr, held column c) <- plane pixel (lr - 4, ci0 + c), zero outside the plane.  Thread = (column
        // c = tid & 31, row tid >> 5 + 16 pass): the column runs fastest so that the global reads of a row are contiguous, and what
        // depends on the column alone (the patch column of the tail's inverse patch map) is formed once (the flat-index form with its
        // divisions per element was 14 000 of the workgroup's 257 000 cycles)
        for (int c0 = 0; c0 < win; c0 += 32) {
            const int c = c0 + (tid & 31), t = ci0 + c;
            const bool cok = c < win && t >= 0 && t < T;
            const int p0 = MODE == 1 ? a.p0 : 1, p1 = MODE == 1 ? a.p1 : 1, tpr = T / p1;
            const int tc = cok ? t / p1 : 0, ft = t - tc * p1;
            constexpr int kPasses = SP / (kConvThreads / 32);      // 16 rows per pass: all of a thread's requests in flight together
            float v[kPasses], v2[kPasses];
#pragma unroll
            for (int u = 0; u < kPasses; ++u) {
                const int lr = (tid >> 5) + (kConvThreads / 32) * u, gr = lr - 4;
                v[u] = 0.f;
                v2[u] = 0.f;
                if (cok && gr >= 0 && gr < S) {
                    const int pix = gr * T + t;
                    if (MODE == 0) {
                        v[u] = a.in_plane[(size_t)n * (S * T) + pix];
                    } else {   // inverse patch map + conv_enhanced residual: feature f of token (g, tc) is pixel (g*p0 + f/p1, tc*p1 + f%p1)
                        const int g = gr / p0, f = (gr - g * p0) * p1 + ft;
                        v[u] = a.lin2_out[((size_t)n * a.tokens + g * tpr + tc) * a.lin2_stride + f];
                        v2[u] = a.resid[(size_t)n * (S * T) + pix];
                    }
                }
            }
            if (c < win) {
#pragma unroll
                for (int u = 0; u < kPasses; ++u) in0[c * SP + (tid >> 5) + (kConvThreads / 32) * u] = v[u] + v2[u];
            }
        }
    }
    __syncthreads();

    // ---- phase 1: the MFMA A fragments (84 registers, kept for the whole kernel) ----
    const bool matrix = wave < ntiles;
    float wa2[36], wa3[48], bias3[4];
    {
        const float *stage = c3;
#pragma unroll
        for (int kb = 0; kb < 36; ++kb) {        // k slot (kb, h): tap = kb>>2 = kx*3+ky, ci = 4h + (kb&3); row = co = j
            const int tap = kb >> 2, kx = tap / 3, ky = tap % 3, ci = 4 * h + (kb & 3);
            wa2[kb] = stage[(ci * 9 + ky * 3 + kx) * 33 + j];
        }
        const int kx3 = min(j >> 3, 2), co3 = j & 7;   // row j = (kx, co); rows 24..31 are padding
        const float keep = j < 24 ? 1.f : 0.f;
#pragma unroll
        for (int kb = 0; kb < 48; ++kb) {        // k slot (kb, h): ky = kb>>4, ci = C-layout row of register kb&15
            const int ky = kb >> 4, e = kb & 15, ci = (e & 3) + 8 * (e >> 2) + 4 * h;
            wa3[kb] = keep * stage[kW3Off + (ci * 3 + ky) * 33 + kx3 * 8 + co3];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) bias3[e] = a.cb[2][e + 4 * h];
    }
    __syncthreads();   // the staging area is dead: c3 becomes the ring

    // conv1 of symbol `sym` (1 -> 8, ReLU; zero outside the plane): lane = (local row lr = 32 wave + j, channel half h)
    auto conv1_col = [&](int sym) {
        const int lr = 32 * wave + j, gr = lr - 4;
        const bool ok = gr >= 0 && gr < S && sym >= 0 && sym < T;
        const int r0 = max(lr - 1, 0), r2 = min(lr + 1, SP - 1);
        const float *src = in0 + (sym - ci0 - 1) * SP;
        float win9[3][3];   // [ky][kx]
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            win9[0][kx] = src[kx * SP + r0];
            win9[1][kx] = src[kx * SP + lr];
            win9[2][kx] = src[kx * SP + r2];
        }
        float *dst = c1 + ((4 * h) * kRing + (sym & 3)) * SP + lr;
#pragma unroll
        for (int k = 0; k < 2; ++k) {      // two channels per v_pk_fma_f32, the same fma chain per channel in tap order
            f32x2 acc2 = f32x2{w1s[72 + 4 * h + 2 * k], w1s[72 + 4 * h + 2 * k + 1]};
#pragma unroll
            for (int k9 = 0; k9 < 9; ++k9) {
                const float x = win9[k9 / 3][k9 % 3];
                acc2 = __builtin_elementwise_fma(f32x2{x, x}, f32x2{w1s[(4 * h + 2 * k) * 9 + k9], w1s[(4 * h + 2 * k + 1) * 9 + k9]}, acc2);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) dst[(2 * k + q) * kRingPlane] = ok ? fmaxf(acc2[q], 0.f) : 0.f;
        }
    };
    // conv4 of symbol `sym` (8 -> 1): lane = (local row, input-channel half), the halves meet through one lane swap
    auto conv4_col = [&](int sym) {
        const int lr = 32 * wave + j;
        const bool okrow = lr >= 4 && lr < 4 + S;
        const int r0 = max(lr - 1, 0), r2 = min(lr + 1, SP - 1);
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float *p = c3 + ((4 * h + c) * kRing) * SP;
#pragma unroll
            for (int k9 = 0; k9 < 9; ++k9) {
                const int ky = k9 / 3, kx = k9 % 3;
                const float x = p[((sym + kx - 1) & 3) * SP + (ky == 0 ? r0 : ky == 1 ? lr : r2)];
                acc = fmaf(x, w4s[(4 * h + c) * 9 + k9], acc);
            }
        }
        acc += rows_other_half32(acc);
        if (h == 0 && okrow) obuf[(lr - 4) * wcols + (sym - ta)] = acc + w4s[72];
    };

    // ---- matrix-wave state (row tile `wave`): as k_conv_stream.hip, the LDS columns replaced by ring slots ----
    const int r = 4 + kTileRows * wave - 1 + j;                       // this lane's local row
    const int gr = r - 4;
    const bool ok2 = matrix && gr >= 0 && gr < S;                     // conv2 output inside the plane (else zero padding)
    const float relu_hi = ok2 ? __builtin_inff() : 0.f;
    const bool ok3 = ok2 && j >= 1 && j <= kTileRows && r < LR - 3;
    const float *bsrc = c1 + (4 * h) * kRingPlane + min(r, SP - 1) - 1;     // + c*kRingPlane + slot*SP + ky
    float *dst3 = c3 + (4 * h) * kRingPlane + min(r, SP - 1);
    f32x16 acc3;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc3[e] = 0.f;
    auto store_col = [&](int tout, float v0, float v1, float v2, float v3) {   // conv3's output column `tout` (zeros outside the plane)
        if (!ok3) return;
        float *p = dst3 + (tout & 3) * SP;
        const bool inr = tout >= 0 && tout < T;
        const float v[4] = {v0, v1, v2, v3};
#pragma unroll
        for (int k = 0; k < 4; ++k) p[k * kRingPlane] = inr ? fmaxf(v[k] + bias3[k], 0.f) : 0.f;
    };
    auto b_at = [&](int kb, int tcol) {      // B operand slot kb of conv2 column tcol: channel 4h + (kb&3), tap (kx, ky)
        const int tap = kb >> 2, kx = tap / 3, ky = tap % 3;
        return bsrc[(kb & 3) * kRingPlane + ((tcol + kx - 1) & 3) * SP + ky];
    };
    float b[36];
    f32x16 bias2v;
    {
        const f32x4 *bp = reinterpret_cast<const f32x4 *>(bias2 + 16 * h);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = bp[q];
#pragma unroll
            for (int u = 0; u < 4; ++u) bias2v[4 * q + u] = v[u];
        }
    }
    float x2[16];
    auto activate2 = [&](const f32x16 &acc2) {
#pragma unroll
        for (int e = 0; e < 16; ++e) x2[e] = __builtin_amdgcn_fmed3f(acc2[e], 0.f, relu_hi);
    };
    auto conv3_step = [&](int i) {   // ky = centre (16..31), below (0..15), above (32..47)
        const int e = i & 15;
        const float xv = i < 16 ? x2[e] : (i < 32 ? lane_from_below(x2[e]) : lane_from_above(x2[e]));
        const int wi = i < 16 ? 16 + e : (i < 32 ? e : 32 + e);
        acc3 = mfma_f32(wa3[wi], xv, acc3);
    };
    auto rotate = [&]() {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc3[8 + e] = acc3[4 + e];
            acc3[4 + e] = acc3[e];
            acc3[e] = 0.f;
        }
    };

    // ---- prologue: c3's rows outside the tiles' stores stay zero for the whole kernel; conv1 of the first four symbols ----
    for (int i = tid; i < 8 * kRingPlane; i += kConvThreads) {
        const int row = i & (SP - 1);
        if (row < 4 || row >= 4 + S) c3[i] = 0.f;      // (rows 4 .. S + 3 of a slot are stored before they are read)
    }
#pragma unroll 1
    for (int sym = cb - 1; sym <= cb + 2; ++sym) conv1_col(sym);
    __syncthreads();
    if (matrix) {      // conv2 of column cb; the operands of column cb + 1 requested behind it
#pragma unroll
        for (int kb = 0; kb < 36; ++kb) b[kb] = b_at(kb, cb);
        f32x16 acc2 = mfma_f32(wa2[0], b[0], bias2v);
        b[0] = b_at(0, cb + 1);
#pragma unroll
        for (int kb = 1; kb < 36; ++kb) {
            acc2 = mfma_f32(wa2[kb], b[kb], acc2);
            b[kb] = b_at(kb, cb + 1);
        }
        activate2(acc2);
    }
    __syncthreads();
    conv1_col(cb + 3);
    __syncthreads();

    constexpr int kLead = 4;   // conv2 MFMAs of the next column issued BEFORE the column hand-over (store + rotate wait for conv3's last MFMA)
#pragma unroll 1
    for (int tcol = cb; tcol <= tb + 3; ++tcol) {
        if (tcol + 4 <= ce) conv1_col(tcol + 4);
        if (matrix) {
            if (tcol + 1 < ce) {           // conv3's share of conv2 column tcol || conv2 of column tcol + 1
                const int tnext = tcol + 2;
                f32x16 acc2 = mfma_f32(wa2[0], b[0], bias2v);
                b[0] = b_at(0, tnext);
#pragma unroll
                for (int kb = 1; kb < kLead; ++kb) {
                    acc2 = mfma_f32(wa2[kb], b[kb], acc2);
                    b[kb] = b_at(kb, tnext);
                }
                __builtin_amdgcn_sched_barrier(0);
                store_col(tcol - 2, acc3[8], acc3[9], acc3[10], acc3[11]);   // complete since the previous column's MFMAs
                rotate();
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < 16; ++g) {      // 48 conv3 MFMAs interleaved with the remaining 32 conv2 MFMAs (3 : 2)
                    conv3_step(3 * g);
                    acc2 = mfma_f32(wa2[kLead + 2 * g], b[kLead + 2 * g], acc2);
                    b[kLead + 2 * g] = b_at(kLead + 2 * g, tnext);
                    conv3_step(3 * g + 1);
                    acc2 = mfma_f32(wa2[kLead + 2 * g + 1], b[kLead + 2 * g + 1], acc2);
                    b[kLead + 2 * g + 1] = b_at(kLead + 2 * g + 1, tnext);
                    conv3_step(3 * g + 2);
                }
                activate2(acc2);
            } else {
                store_col(tcol - 2, acc3[8], acc3[9], acc3[10], acc3[11]);
                rotate();
                if (tcol < ce) {           // conv3's share of the last conv2 column
#pragma unroll
                    for (int i = 0; i < 48; ++i) conv3_step(i);
                }
            }
        }
        if (tcol - 4 >= ta && tcol - 4 < tb) conv4_col(tcol - 4);
        __syncthreads();
    }

    // ---- the output columns leave: rows of (tb - ta) contiguous floats ----
    {
        const int wout = tb - ta;
        if (MODE == 0) {
            float *dst = a.out_plane + (size_t)n * (S * T) + ta;
            for (int i = tid; i < S * wout; i += kConvThreads) {
                const int row = i / wout, c = i - row * wout;
                dst[(size_t)row * T + c] = obuf[row * wcols + c];
            }
        } else {   // interleave this plane into the complex64 output (the frame's other plane is another workgroup's)
            float *dst = a.out_complex + ((size_t)frame * (S * T) + ta) * 2 + part;
            for (int i = tid; i < S * wout; i += kConvThreads) {
                const int row = i / wout, c = i - row * wout;
                dst[((size_t)row * T + c) * 2] = obuf[row * wcols + c];
            }
        }
    }
}

// conv_rows16_kernel (round 5): the same column pipeline with conv2 / conv3 as v_mfma_f32_16x16x4_f32 chains -- k_conv_stream.hip's
// conv_stream16_kernel formulation (all 72 (ky, kx, co) products of a pixel as rows of one operand: 80 + 72 MFMAs of 32 cycles per
// column instead of 48 + 36 of 64; pixel tiles by row parity, so the ky sums are adds across the two tiles' registers and conv2's
// accumulators are conv3's B operands unshifted; the kx rotation as the C operand of each tile's first MFMA of a column) and the
// operand fragments of the forward's prologue launch (a.wfrag) instead of a transposed LDS staging + gather.
// MODE 0 = head (a.in_plane = upsampled planes [planes][S][T] -> a.out_plane), 1 = tail (a.lin2_out + a.resid -> a.out_complex)
template <int MODE>
__global__ __launch_bounds__(kConvThreads) void conv_rows16_kernel(const ConvArgs a, int nsplit) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int SP = kRowsSP;
    const int S = a.S, T = a.T, LR = S + 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const int n = blockIdx.x / nsplit, part_c = blockIdx.x - n * nsplit, frame = n >> 1, part = n & 1;
    const int wcols = (T + nsplit - 1) / nsplit;
    const int ta = part_c * wcols, tb = min(T, ta + wcols);          // output columns of this workgroup
    const int cb = max(ta - 2, 0), ce = min(tb + 2, T);                // conv2 columns swept: [cb, ce)
    const int ci0 = cb - 2, win = ce + 2 - ci0;                        // input columns held: symbols ci0 .. ce + 1
    const int ntiles = (S + kTileRows - 1) / kTileRows;                // row tiles = matrix waves (<= 8)
    float *in0 = smem;
    float *c1 = in0 + (size_t)(wcols + 8) * SP;
    float *c3 = c1 + 8 * kRingPlane;
    float *obuf = c3 + 8 * kRingPlane;
    float *bias2 = obuf + (size_t)S * wcols, *w1s = bias2 + 32, *w4s = w1s + 80;
#ifdef AFT_CHECKED
    // The ring protocol, checked: tag[slot] = the symbol a slot of the conv1 ring ([0..3]) / conv3 ring ([4..7]) holds, written with the
    // column; every reader asserts that the slot it is about to read holds the symbol it wants ("an iteration only reads what earlier
    // iterations wrote and only writes slots nobody reads in the same iteration", the file header).
    volatile int *ring_tag = reinterpret_cast<volatile int *>(w4s + 80);
    if (tid < 8) ring_tag[tid] = -(1 << 30);
#endif

    // ---- phase 0: the operand fragments (22 lane-linear 16-byte loads per wave), the small tables, the input columns ----
    f32x4 fq[kFragQuads];
    {
        const f32x4 *fp = reinterpret_cast<const f32x4 *>(a.wfrag) + lane;
#pragma unroll
        for (int q = 0; q < kFragQuads; ++q) fq[q] = fp[q * 64];
    }
    {
        if (tid >= 64 && tid < 144) w1s[tid - 64] = tid < 136 ? a.cw[0][tid - 64] : a.cb[0][tid - 136];
        if (tid >= 192 && tid < 265) w4s[tid - 192] = tid < 264 ? a.cw[3][tid - 192] : a.cb[3][0];
        // input: element (local row lr, held column c) <- plane pixel (lr - 4, ci0 + c), zero outside the plane.  Thread = (column
        // c = tid & 31, row tid >> 5 + 16 pass): the column runs fastest so that the global reads of a row are contiguous, and what
        // depends on the column alone (the patch column of the tail's inverse patch map) is formed once (the flat-index form with its
        // divisions per element was 14 000 of the workgroup's 257 000 cycles)
        for (int c0 = 0; c0 < win; c0 += 32) {
            const int c = c0 + (tid & 31), t = ci0 + c;
            const bool cok = c < win && t >= 0 && t < T;
            const int p0 = MODE == 1 ? a.p0 : 1, p1 = MODE == 1 ? a.p1 : 1, tpr = T / p1;
            const int tc = cok ? t / p1 : 0, ft = t - tc * p1;
            constexpr int kPasses = SP / (kConvThreads / 32);      // 16 rows per pass: all of a thread's requests in flight together
            float v[kPasses], v2[kPasses];
#pragma unroll
            for (int u = 0; u < kPasses; ++u) {
                const int lr = (tid >> 5) + (kConvThreads / 32) * u, gr = lr - 4;
                v[u] = 0.f;
                v2[u] = 0.f;
                if (cok && gr >= 0 && gr < S) {
                    const int pix = gr * T + t;
                    if (MODE == 0) {
                        v[u] = a.in_plane[(size_t)n * (S * T) + pix];
                    } else {   // inverse patch map + conv_enhanced residual: feature f of token (g, tc) is pixel (g*p0 + f/p1, tc*p1 + f%p1)
                        const int g = gr / p0, f = (gr - g * p0) * p1 + ft;
                        v[u] = a.lin2_out[((size_t)n * a.tokens + g * tpr + tc) * a.lin2_stride + f];
                        v2[u] = a.resid[(size_t)n * (S * T) + pix];
                    }
                }
            }
            if (c < win) {
#pragma unroll
                for (int u = 0; u < kPasses; ++u) in0[c * SP + (tid >> 5) + (kConvThreads / 32) * u] = v[u] + v2[u];
            }
        }
    }
    __syncthreads();

    // ---- phase 1: the MFMA A fragments (76 registers, kept for the whole kernel; conv_device.h: conv_frag16_entry) ----
    const bool matrix = wave < ntiles;
    float wa2[36], wa3[40], bias3[2];
    f32x4 bias2v[2];
#pragma unroll
    for (int f = 0; f < 36; ++f) wa2[f] = fq[f >> 2][f & 3];
#pragma unroll
    for (int f = 0; f < 40; ++f) wa3[f] = fq[9 + (f >> 2)][f & 3];
    bias3[0] = fq[19][0]; bias3[1] = fq[19][1];
    bias2v[0] = fq[20]; bias2v[1] = fq[21];

    // conv1 of symbol `sym` (1 -> 8, ReLU; zero outside the plane): lane = (local row lr = 32 wave + j, channel half h)
    auto conv1_col = [&](int sym) {
        const int lr = 32 * wave + j, gr = lr - 4;
        const bool ok = gr >= 0 && gr < S && sym >= 0 && sym < T;
        const int r0 = max(lr - 1, 0), r2 = min(lr + 1, SP - 1);
        const float *src = in0 + (sym - ci0 - 1) * SP;
        float win9[3][3];   // [ky][kx]
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            win9[0][kx] = src[kx * SP + r0];
            win9[1][kx] = src[kx * SP + lr];
            win9[2][kx] = src[kx * SP + r2];
        }
        float *dst = c1 + ((4 * h) * kRing + (sym & 3)) * SP + lr;
#ifdef AFT_CHECKED
        AFT_DEV_ASSERT(lr >= 0 && lr < SP && dst >= c1 && (dst + 3 * kRingPlane) < c3);      // channels 4 h .. 4 h + 3 of the conv1 ring
        if (tid == 0) ring_tag[sym & 3] = sym;
#endif
#pragma unroll
        for (int k = 0; k < 2; ++k) {      // two channels per v_pk_fma_f32, the same fma chain per channel in tap order
            f32x2 acc2 = f32x2{w1s[72 + 4 * h + 2 * k], w1s[72 + 4 * h + 2 * k + 1]};
#pragma unroll
            for (int k9 = 0; k9 < 9; ++k9) {
                const float x = win9[k9 / 3][k9 % 3];
                acc2 = __builtin_elementwise_fma(f32x2{x, x}, f32x2{w1s[(4 * h + 2 * k) * 9 + k9], w1s[(4 * h + 2 * k + 1) * 9 + k9]}, acc2);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) dst[(2 * k + q) * kRingPlane] = ok ? fmaxf(acc2[q], 0.f) : 0.f;
        }
    };
    // conv4 of symbol `sym` (8 -> 1): lane = (local row, input-channel half), the halves meet through one lane swap
    auto conv4_col = [&](int sym) {
        const int lr = 32 * wave + j;
        const bool okrow = lr >= 4 && lr < 4 + S;
        const int r0 = max(lr - 1, 0), r2 = min(lr + 1, SP - 1);
        // two input channels per v_pk_fma_f32 (as conv_stream16_kernel's helpers: the vector stages cost the matrix pipe ALU time)
        f32x2 acc2[2] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};
#pragma unroll
        for (int cp = 0; cp < 2; ++cp) {
            const float *p = c3 + ((4 * h + 2 * cp) * kRing) * SP;
#pragma unroll
            for (int k9 = 0; k9 < 9; ++k9) {
                const int ky = k9 / 3, kx = k9 % 3;
                const int off = ((sym + kx - 1) & 3) * SP + (ky == 0 ? r0 : ky == 1 ? lr : r2);
#ifdef AFT_CHECKED
                AFT_DEV_ASSERT(ring_tag[4 + ((sym + kx - 1) & 3)] == sym + kx - 1);     // conv4 reads conv3's symbols sym - 1 .. sym + 1
#endif
                acc2[cp] = __builtin_elementwise_fma(f32x2{p[off], p[kRingPlane + off]},
                                                     f32x2{w4s[(4 * h + 2 * cp) * 9 + k9], w4s[(4 * h + 2 * cp + 1) * 9 + k9]}, acc2[cp]);
            }
        }
        const f32x2 s2 = acc2[0] + acc2[1];
        float acc = s2[0] + s2[1];
        acc += rows_other_half32(acc);
        if (h == 0 && okrow) obuf[(lr - 4) * wcols + (sym - ta)] = acc + w4s[72];
    };

    // ---- matrix-wave state (row tile `wave`): conv_stream16_kernel's, the LDS columns replaced by ring slots ----
    const int p = lane & 15, g = lane >> 4;
    const int r0 = 4 + kTileRows * wave - 1 + 2 * p;                 // local row of this lane's pixel in tile pt = 0 (pt = 1: r0 + 1)
    bool ok3[2];
    float relu_hi[2];
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
        const int jj = 2 * p + pt, r = r0 + pt, gr = r - 4;
        const bool ok2 = matrix && gr >= 0 && gr < S;                // conv2 output inside the plane (else zero padding)
        relu_hi[pt] = ok2 ? __builtin_inff() : 0.f;
        ok3[pt] = ok2 && jj >= 1 && jj <= kTileRows && r < LR - 3;
    }
    const float *bsrc = c1 + g * kRingPlane + min(r0, SP - 4) - 1;   // + 4 cih channels + slot * SP: rows r0 - 1 .. r0 + 2
    float *dst3[2];
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) dst3[pt] = c3 + g * kRingPlane + (ok3[pt] ? r0 + pt : 0);   // halo / outside lanes: row 0, which nobody reads into a stored result
    const int c3first = cb == 0 ? 0 : cb + 1;                        // first conv3 column with all three kx taps inside the sweep (= ta - 1)
    float bv[3][2][4];
    auto load_b = [&](int kx, int cih, int tcol) {                   // conv2 column tcol: conv1 symbol tcol + kx - 1
        const float *q = bsrc + 4 * cih * kRingPlane + ((tcol + kx - 1) & 3) * SP;
#ifdef AFT_CHECKED
        // conv2 column tcol reads conv1's symbol tcol + kx - 1.  (A halo lane of tile 0 reads one float in front of the ring -- the last
        // element of the input columns, inside the allocation; its products never reach a stored result.)
        // (The sweep requests the operands of column `ce` behind its last column: a prefetch nobody consumes, of whatever the slots hold.)
        AFT_DEV_ASSERT((tcol >= ce || ring_tag[(tcol + kx - 1) & 3] == tcol + kx - 1) && q + 1 >= c1 && q + 3 < c3);
#endif
        const f32x2 lo = *reinterpret_cast<const f32x2 *>(q), hi = *reinterpret_cast<const f32x2 *>(q + 2);
        bv[kx][cih][0] = lo[0]; bv[kx][cih][1] = lo[1]; bv[kx][cih][2] = hi[0]; bv[kx][cih][3] = hi[1];
    };
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const f32x4 bias3v = {0.f, 0.f, bias3[0], bias3[1]};
    f32x4 a3[2][5], n3[2][5], x2[2][2], acc2[2][2];
#pragma unroll
    for (int pt = 0; pt < 2; ++pt)
#pragma unroll
        for (int rt = 0; rt < 5; ++rt) a3[pt][rt] = n3[pt][rt] = rt == 0 ? bias3v : zero4;
    auto relu2 = [&]() {
#pragma unroll
        for (int pt = 0; pt < 2; ++pt)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int v = 0; v < 4; ++v) x2[pt][mt][v] = __builtin_amdgcn_fmed3f(acc2[pt][mt][v], 0.f, relu_hi[pt]);
    };
    auto conv2_step = [&](int gi, int u) {
        const int kx = gi >> 1, cih = gi & 1, ky = u >> 2, pt = (u >> 1) & 1, mt = u & 1, ks2 = 2 * (kx * 3 + ky) + cih;
        acc2[pt][mt] = rows_mfma16(wa2[mt * 18 + ks2], bv[kx][cih][ky + pt], (gi == 0 && ky == 0) ? bias2v[mt] : acc2[pt][mt]);
    };
    auto rotated = [&](int pt, int rt) -> f32x4 {
        if (rt == 0) return bias3v;
        if (rt < 3) return a3[pt][rt - 1];
        if (rt == 3) return zero4;
        return f32x4{a3[pt][3][2], a3[pt][3][3], a3[pt][4][0], a3[pt][4][1]};
    };
    auto conv3_step = [&](int m) {
        constexpr int kOrder[5] = {2, 4, 1, 0, 3};
        const int ks = m / 10, i = m % 10, rt = kOrder[i >> 1], pt = i & 1;
        n3[pt][rt] = rows_mfma16(wa3[rt * 8 + ks], x2[pt][ks >> 2][ks & 3], ks == 0 ? rotated(pt, rt) : n3[pt][rt]);
    };
    // conv3's output column `tout` into its ring slot: the ky sums of the six finished registers Y[pt][2 ky + co half]; zeros for a
    // column outside the grid (the ring has no zero border), nothing for a column before the sweep's first complete one
    auto store_col = [&](int tout, const float (&Y)[2][6]) {
        const bool inr = tout >= 0 && tout < T;
        if (inr && tout < c3first) return;
#ifdef AFT_CHECKED
        if (tid == 0) ring_tag[4 + (tout & 3)] = tout;
#endif
#pragma unroll
        for (int cohi = 0; cohi < 2; ++cohi) {
            float o0 = Y[0][2 + cohi] + rows_from_below16(Y[1][cohi]) + Y[1][4 + cohi];
            float o1 = Y[1][2 + cohi] + Y[0][cohi] + rows_from_above16(Y[0][4 + cohi]);
            o0 = inr ? fmaxf(o0, 0.f) : 0.f;
            o1 = inr ? fmaxf(o1, 0.f) : 0.f;
            dst3[0][4 * cohi * kRingPlane + (tout & 3) * SP] = o0;
            dst3[1][4 * cohi * kRingPlane + (tout & 3) * SP] = o1;
        }
    };
    auto kx2_set = [&](float (&Y)[2][6]) {       // a3's registers of output column (last swept column - 1)
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
#pragma unroll
            for (int v = 0; v < 4; ++v) Y[pt][v] = a3[pt][2][v];
            Y[pt][4] = a3[pt][4][2]; Y[pt][5] = a3[pt][4][3];
        }
    };
    auto kx1_set = [&](float (&Y)[2][6]) {       // ... of output column (last swept column): complete only at the grid's right edge
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
#pragma unroll
            for (int v = 0; v < 4; ++v) Y[pt][v] = a3[pt][1][v];
            Y[pt][4] = a3[pt][4][0]; Y[pt][5] = a3[pt][4][1];
        }
    };
    const float zeros6[2][6] = {{0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}};

    // ---- prologue: c3's rows outside the tiles' stores stay zero for the whole kernel; conv1 of the first four symbols ----
    for (int i = tid; i < 8 * kRingPlane; i += kConvThreads) {
        const int row = i & (SP - 1);
        if (row < 4 || row >= 4 + S) c3[i] = 0.f;      // (rows 4 .. S + 3 of a slot are stored before they are read)
    }
#pragma unroll 1
    for (int sym = cb - 1; sym <= cb + 2; ++sym) conv1_col(sym);
    __syncthreads();
    if (matrix) {      // conv2 of column cb; the operands of column cb + 1 requested behind each group
#pragma unroll
        for (int gi = 0; gi < 6; ++gi) load_b(gi >> 1, gi & 1, cb);
#pragma unroll
        for (int gi = 0; gi < 6; ++gi) {
#pragma unroll
            for (int u = 0; u < 12; ++u) conv2_step(gi, u);
            load_b(gi >> 1, gi & 1, cb + 1);
        }
        relu2();
    }
    __syncthreads();
    conv1_col(cb + 3);
    __syncthreads();

#pragma unroll 1
    for (int tcol = cb; tcol <= tb + 3; ++tcol) {
        if (tcol + 4 <= ce) conv1_col(tcol + 4);
        if (matrix) {
            // output column tcol - 2 from the state the previous iterations left (a3): the kx = 2 registers while columns are being
            // swept and one iteration beyond, the kx = 1 registers of the last swept column after that (complete only when that
            // column is the grid's last: column T is zero padding), zeros further out
            auto store_pending = [&]() {
                float Y[2][6];
                if (tcol <= ce) {
                    kx2_set(Y);
                    store_col(tcol - 2, Y);
                } else if (tcol == ce + 1 && ce == T) {
                    kx1_set(Y);
                    store_col(tcol - 2, Y);
                } else if (tcol - 2 >= T) {
                    store_col(tcol - 2, zeros6);
                }
            };
            if (tcol + 1 < ce) {           // conv3's share of conv2 column tcol || conv2 of column tcol + 1
                const int tnext = tcol + 2;
#pragma unroll
                for (int gi = 0; gi < 6; ++gi) {
                    const int m0 = 12 * gi, m1 = gi == 5 ? 80 : 12 * (gi + 1);
#pragma unroll
                    for (int u = 0; u < 12; ++u) {
                        conv3_step(m0 + u);
                        conv2_step(gi, u);
                    }
                    load_b(gi >> 1, gi & 1, tnext);     // symbols tcol + 1 .. tcol + 3: not the slot conv1_col(tcol + 4) writes in this iteration
                    if (m1 - m0 > 12) {
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int u = 12; u < 20; ++u) conv3_step(m0 + u);
                    }
                    if (gi == 0) store_pending();
                    __builtin_amdgcn_sched_barrier(0);
                }
                relu2();
#pragma unroll
                for (int pt = 0; pt < 2; ++pt)
#pragma unroll
                    for (int rt = 0; rt < 5; ++rt) a3[pt][rt] = n3[pt][rt];
            } else {
                store_pending();
                if (tcol < ce) {           // conv3's share of the last conv2 column
#pragma unroll
                    for (int m = 0; m < 80; ++m) conv3_step(m);
#pragma unroll
                    for (int pt = 0; pt < 2; ++pt)
#pragma unroll
                        for (int rt = 0; rt < 5; ++rt) a3[pt][rt] = n3[pt][rt];
                }
            }
        }
        if (tcol - 4 >= ta && tcol - 4 < tb) conv4_col(tcol - 4);
        __syncthreads();
    }

    // ---- the output columns leave: rows of (tb - ta) contiguous floats ----
    {
        const int wout = tb - ta;
        if (MODE == 0) {
            float *dst = a.out_plane + (size_t)n * (S * T) + ta;
            for (int i = tid; i < S * wout; i += kConvThreads) {
                const int row = i / wout, c = i - row * wout;
                dst[(size_t)row * T + c] = obuf[row * wcols + c];
            }
        } else {   // interleave this plane into the complex64 output (the frame's other plane is another workgroup's)
            float *dst = a.out_complex + ((size_t)frame * (S * T) + ta) * 2 + part;
            for (int i = tid; i < S * wout; i += kConvThreads) {
                const int row = i / wout, c = i - row * wout;
                dst[((size_t)row * T + c) * 2] = obuf[row * wcols + c];
            }
        }
    }
}

// The shapes this kernel takes: inference, the plane's rows as at most eight 30-row tiles, inputs as the whole forward provides them
// (head: upsampled planes; tail: linear_2's output), and an LDS footprint that fits -- everything else stays with k_conv.hip.
static int conv_rows_split(const ConvArgs &a, int planes) {
    const int cus = current_device_cus();
    int nsplit = 1;
    while (planes * nsplit < cus && a.T / (nsplit * 2) >= 8) nsplit *= 2;     // column ranges of at least 8 columns
    // ranges are ceil(T / nsplit) columns wide: drop the ranges that would start behind the last column (T = 129, 16 ranges of 9
    // columns: the 16th would start at column 135 -- ADVICE r4), so that every workgroup of the launch owns at least one column
    const int wcols = (a.T + nsplit - 1) / nsplit;
    return (a.T + wcols - 1) / wcols;
}
bool conv_rows_ok(const ConvArgs &a, int planes) {
    if (a.S + 8 > kRowsSP || (a.S + kTileRows - 1) / kTileRows > kConvWaves || a.T < 8) return false;
    if (a.mode == 0 && a.in_plane == nullptr) return false;
    if (a.mode == 1 && (a.lin2_out == nullptr || a.resid == nullptr || a.T % a.p1 != 0 || a.S % a.p0 != 0)) return false;
    if (a.mode != 0 && a.mode != 1) return false;
    const int nsplit = conv_rows_split(a, planes), wcols = (a.T + nsplit - 1) / nsplit;
    return sizeof(float) * conv_rows_lds_floats(a.S, wcols) <= 160 * 1024;
}

hipError_t launch_conv_rows(ConvArgs &a, int planes, hipStream_t st) {
    if (!conv_rows_ok(a, planes)) return hipErrorNotSupported;
    const int nsplit = conv_rows_split(a, planes), wcols = (a.T + nsplit - 1) / nsplit;
    const size_t lds = sizeof(float) * conv_rows_lds_floats(a.S, wcols);
    static PerDeviceOnce lds_head, lds_tail, lds_head16, lds_tail16;
    // the 16x16x4 matrix phase needs the operand fragments of the forward's prologue launch; AFT_CONV_MFMA32=1 keeps the 32x32x2 kernel (A/B)
    const bool m16 = a.wfrag != nullptr && !switch_on("AFT_CONV_MFMA32");
    hipError_t e;
    if (m16)
        e = a.mode == 0 ? ensure_dynamic_lds(lds_head16, reinterpret_cast<const void *>(conv_rows16_kernel<0>), 160 * 1024)
                        : ensure_dynamic_lds(lds_tail16, reinterpret_cast<const void *>(conv_rows16_kernel<1>), 160 * 1024);
    else
        e = a.mode == 0 ? ensure_dynamic_lds(lds_head, reinterpret_cast<const void *>(conv_rows_kernel<0>), 160 * 1024)
                        : ensure_dynamic_lds(lds_tail, reinterpret_cast<const void *>(conv_rows_kernel<1>), 160 * 1024);
    if (e != hipSuccess) return e;
    const dim3 grid(planes * nsplit), block(kConvThreads);
    if (m16) {
        if (a.mode == 0) hipLaunchKernelGGL((conv_rows16_kernel<0>), grid, block, lds, st, a, nsplit);
        else hipLaunchKernelGGL((conv_rows16_kernel<1>), grid, block, lds, st, a, nsplit);
    } else if (a.mode == 0) hipLaunchKernelGGL((conv_rows_kernel<0>), grid, block, lds, st, a, nsplit);
    else hipLaunchKernelGGL((conv_rows_kernel<1>), grid, block, lds, st, a, nsplit);
    return hipGetLastError();
}

}  // namespace aft
