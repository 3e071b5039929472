// k_conv_stream.hip -- the two ConvEnhancer stacks of the default 120 x 14 grid as a COLUMN-STREAMING PIPELINE (round 4).
//
// Reference semantics: exactly k_conv.hip's (head: reference src/models/fortitran.py:203-209 behind the pilot_upsampler product;
// tail: fortitran.py:225-231,180 on linear_2's output; ConvEnhancer blocks/enhancers.py:12-20) -- same arithmetic per output
// element (conv1 / conv4 fp32 FMA chains in tap order, conv2 / conv3 the same MFMA chains), so the two kernels agree to rounding
// of a different summation order only where the banded kernel exchanges seam taps.
//
// Why a second kernel.  The banded kernel runs its phases one after the other -- weights 3 600, zero 2 400, input 4 700,
// conv1 3 300, matrix phase 88 400, seam fix-up 1 650, conv4 4 100 cycles (DESIGN.md 4.3 stamps) -- and at the benchmark's batch
// there is exactly ONE workgroup per CU (256 planes, 256 CUs, 139 KB of LDS), so nothing runs under those 19 800 cycles.  Here the
// eight waves of a workgroup SPECIALISE:
//   waves 0..3  (one per SIMD)  matrix waves: row tile w, ALL 14 columns -- conv3(t) || conv2(t+1) as two interleaved MFMA chains
//                               exactly as in the banded kernel, but no column segments, hence no seams and no fix-up pass;
//   waves 4..7  (one per SIMD)  helper waves: border zeroing, the input plane, conv1 for all columns while the matrix waves
//                               gather their weight fragments; then conv4 of column t as soon as the matrix waves have published
//                               conv3's columns t-1 .. t+1 (one LDS flag per matrix wave), into an LDS output plane.
// The matrix pipe of every SIMD is fed from the first sweep to the last; the vector work of conv1 / conv4 / the input plane rides
// in the shadow of the MFMA chains (it still costs issue slots -- fp32 MFMA and VALU share the SIMD's ALU -- but no longer its own
// serial phases), and the output leaves as whole 6.7-KB planes in one coalesced pass.
// The head takes the upsampled planes from the forward's prologue launch (one product over all planes, up_w read once per launch
// instead of 161 KB streamed through every CU's L1); callers without that scratch (the per-stage entry points) run the banded kernel.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "conv_device.h"

// The wave-to-wave hand-overs of both kernels in this file publish a column count with a plain LDS store behind the column's data
// stores and NO release fence (a fence would drain the publishing wave's pending operand reads and stall its MFMA chain once per
// column): correct because gfx950's LDS serves one wave's requests in issue order, so a reader that sees the count finds the data.
// That is a property of this target, not of the memory model -- refuse to build for anything else; the every-output
// comparison that guards it is tests/test_hip_parity.py::test_conv_stream_hand_over_soak.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "k_conv_stream.hip relies on gfx950's in-order LDS service for its flag hand-overs"
#endif

namespace aft {

namespace {

constexpr int S = 120, T = 14, SP = 128, LR = 128;   // LR = S + 8: local row lr <-> plane row lr - 4
constexpr int kPlane = (T + 2) * SP;                 // floats per channel plane [T + 2][SP]: LDS column = symbol + 1
constexpr int kIn0 = 0, kC1 = kPlane, kC3 = 9 * kPlane, kArena = 17 * kPlane;
constexpr int kStage = kArena - kWStage;             // conv2 / conv3 weights staged in the END of c3 (dead until the second barrier)
constexpr int kBias2 = kArena, kW1 = kBias2 + 32, kW4 = kW1 + 80, kFlags = kW4 + 80, kStreamFloats = kFlags + 16;
constexpr size_t kStreamLds = sizeof(float) * kStreamFloats;   // 140 352 B: one workgroup per CU, as before

using lds_int = __attribute__((address_space(3))) int;
using i32x4 = __attribute__((ext_vector_type(4))) int;
using lds_i32x4 = __attribute__((address_space(3))) i32x4;

__device__ __forceinline__ float other_half32(float x) {   // value held by lane (l ^ 32): v_permlane32_swap
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, (threadIdx.x & 32) ? r[0] : r[1]);
}

// v_mfma_f32_16x16x4_f32: A[i = lane % 16][k = lane / 16], B[k = lane / 16][j = lane % 16], D[i = 4 (lane / 16) + v][j = lane % 16]
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
#if defined(__HIP_DEVICE_COMPILE__) && defined(AFT_M16_SPACER)
    asm volatile("s_nop 0" : "+v"(c));
#endif
    return c;
}
__device__ __forceinline__ float row16_from_below(float v) {   // lane i <- lane i-1 inside each 16-lane row, 0 into lane 0 (DPP row_shr:1)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_from_above(float v) {   // lane i <- lane i+1 inside each 16-lane row, 0 into lane 15 (DPP row_shl:1)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x101, 0xf, 0xf, true));
}
// conv3's 80 MFMAs of a column over the six conv2 groups (12 MFMAs each): the last group takes more of conv3, so conv2's last MFMA --
// whose result the next column's first MFMAs wait for -- issues well before the column ends
#ifndef AFT_POLL_SLEEP
#define AFT_POLL_SLEEP 24
#endif
#ifndef AFT_HELPER_PRIO
#define AFT_HELPER_PRIO 3
#endif
#ifndef AFT_C3_SPLIT
#define AFT_C3_SPLIT {0, 12, 24, 36, 48, 60, 80}
#endif
__device__ constexpr int kC3Split[7] = AFT_C3_SPLIT;

}  // namespace

// MODE 0 = head (a.in_plane = upsampled planes [planes][S][T], a.out_plane), 1 = tail (a.lin2_out + a.resid, a.out_complex).
// TRAIN (MODE 0 with a.mode == 2 semantics: plain plane in, plain plane out) adds what k_conv.hip's training instantiation adds
// (SURVEY 8f-1): the three stage outputs written to HBM in [plane][C][T][S] order (forward: the activations the backward needs;
// backward: the pre-activation gradients the weight-gradient kernels need), a masked activation (backward: the ReLU derivative from
// the saved forward activation instead of bias + ReLU), and NULL biases.  The backward of the stack IS this kernel on dL/dy with
// transposed, flipped weights (conv4^T 1->8, conv3^T 8->32, conv2^T 32->8, conv1^T 8->1: the forward's stage shapes).
template <int MODE, bool TRAIN = false>
__global__ __launch_bounds__(kConvThreads) void conv_stream_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *in0 = smem + kIn0, *c1 = smem + kC1, *c3 = smem + kC3;
    float *bias2 = smem + kBias2, *w1s = smem + kW1, *w4s = smem + kW4;
    // [0..3]: conv3 columns published by matrix wave w.  An address_space(3) pointer: volatile accesses through a GENERIC pointer are
    // not rewritten to LDS instructions -- they were flat_load / flat_store with sc0 sc1 and an s_waitcnt vmcnt(0) behind every
    // publication (round 5, found in the disassembly)
    volatile lds_int *flags = (volatile lds_int *)(smem + kFlags);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    // Column ranges (training instantiation only, late round 5): with fewer planes than CUs a plane is worked by `ranges` workgroups,
    // each producing the output columns [ta, tb) and recomputing what it needs of its neighbours' columns -- conv3 columns
    // [ta - 1, tb + 1), conv2 [ta - 2, tb + 2), conv1 [ta - 3, tb + 3), clipped to the plane.  Every column goes through the same
    // instruction sequence as in the whole-plane sweep (same bits; columns two ranges both compute are stored twice with the same value).
    const int ranges = TRAIN ? max(a.ranges, 1) : 1;
    const int n = blockIdx.x / ranges, range = blockIdx.x - n * ranges, frame = n >> 1, part = n & 1;
    const int wcols = (T + ranges - 1) / ranges, ta = range * wcols, tb = min(T, ta + wcols);
    const int c3lo = max(ta - 1, 0), c3hi = min(tb + 1, T), c2lo = max(ta - 2, 0), c2hi = min(tb + 2, T);
    const int c1lo = max(ta - 3, 0), c1hi = min(tb + 3, T), c1mid = c1lo + (c1hi - c1lo + 1) / 2;
    const bool matrix = wave < 4;
#ifdef AFT_DIAG_STAMPS
    // slots 0..7: matrix wave 0, 8..15: helper wave 4 (thread 256)
#define SSTAMP(i) do { if (a.stamps && (tid == 0 || tid == 256)) a.stamps[(size_t)blockIdx.x * 16 + (tid ? 8 : 0) + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SSTAMP(i) do { } while (0)
#endif
    SSTAMP(0);

    // ---- phase 0 (all waves): stage the conv2 / conv3 weights TRANSPOSED in the end of c3 (as k_conv.hip does in the arena:
    //      conflict-free gathers), the small tables, the flags ----
    {
        float *stage = smem + kStage;
        for (int i = tid; i < 2304; i += kConvThreads) {
            stage[(i % 72) * 33 + i / 72] = a.cw[1][i];
            const int rem = i % 288;   // conv3.weight [co 8][ci 32][ky 3][kx 3]
            stage[kW3Off + (rem / 3) * 33 + (rem % 3) * 8 + i / 288] = a.cw[2][i];
        }
        if (tid < 32) bias2[tid] = (TRAIN && !a.cb[1]) ? 0.f : a.cb[1][(tid & 3) + 8 * ((tid & 15) >> 2) + 4 * (tid >> 4)];
        if (tid >= 64 && tid < 144) w1s[tid - 64] = tid < 136 ? a.cw[0][tid - 64] : ((TRAIN && !a.cb[0]) ? 0.f : a.cb[0][tid - 136]);
        if (tid >= 192 && tid < 265) w4s[tid - 192] = tid < 264 ? a.cw[3][tid - 192] : ((TRAIN && !a.cb[3]) ? 0.f : a.cb[3][0]);
        if (tid >= 320 && tid < 336) flags[tid - 320] = 0;
    }
    __syncthreads();
    SSTAMP(1);

    // conv1: 1 -> 8, ReLU.  Thread = (local row lr = 32 rw + j, channel half h): 4 channels x 9 taps per column, the window slides
    // over the columns [t0, t1).  The helper waves take columns 0..6 right behind their input pass, the matrix waves 7..13 once
    // their fragments are gathered and the helpers have signalled the input plane.
    auto conv1_columns = [&](int rw, int t0, int t1) {
        const int lr = 32 * rw + j, gr = lr - 4;
        const bool ok = lr >= 1 && lr < LR - 1 && gr >= 0 && gr < S;
        // two channels per v_pk_fma_f32: the phase is bound by vector-instruction issue (36 FMAs per thread and column otherwise)
        f32x2 w[2][9], b[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
#pragma unroll
            for (int k9 = 0; k9 < 9; ++k9) w[k][k9] = f32x2{w1s[(4 * h + 2 * k) * 9 + k9], w1s[(4 * h + 2 * k + 1) * 9 + k9]};
            b[k] = f32x2{w1s[72 + 4 * h + 2 * k], w1s[72 + 4 * h + 2 * k + 1]};
        }
        const int r0 = max(lr - 1, 0), r2 = min(lr + 1, LR - 1);
        float win[3][3];   // [ky][kx]
#pragma unroll
        for (int kx = 1; kx < 3; ++kx) {
            win[0][kx] = in0[(t0 + kx - 1) * SP + r0];
            win[1][kx] = in0[(t0 + kx - 1) * SP + lr];
            win[2][kx] = in0[(t0 + kx - 1) * SP + r2];
        }
        float *dst = c1 + (4 * h) * kPlane + SP + lr;
#pragma unroll
        for (int u = 0; u < T / 2; ++u) {     // 7 columns per call, fully unrolled: the window slides by renaming
            const int t = t0 + u;
            if (t >= t1) break;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) { win[ky][0] = win[ky][1]; win[ky][1] = win[ky][2]; }
            win[0][2] = in0[(t + 2) * SP + r0];
            win[1][2] = in0[(t + 2) * SP + lr];
            win[2][2] = in0[(t + 2) * SP + r2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                f32x2 acc2 = b[k];
#pragma unroll
                for (int k9 = 0; k9 < 9; ++k9) {
                    const float x = win[k9 / 3][k9 % 3];
                    acc2 = __builtin_elementwise_fma(f32x2{x, x}, w[k][k9], acc2);     // per channel: the same fma chain in tap order
                }
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const float acc = acc2[q];
                    float v = fmaxf(acc, 0.f);
                    if constexpr (TRAIN) {
                        if (ok) {
                            const unsigned gi = ((unsigned)(n * 8 + 4 * h + 2 * k + q) * T + t) * S + gr;
                            if (a.mask[0]) v = conv_ld(conv_srd(a.mask[0]), gi) > 0.f ? acc : 0.f;
                            if (a.save[0]) conv_st(conv_srd(a.save[0]), gi, v);
                        }
                    }
                    dst[(2 * k + q) * kPlane + t * SP] = ok ? v : 0.f;
                }
            }
        }
    };
    auto wait_count = [&](int slot, int count) {   // workgroup-scope hand-over through an LDS counter
        AFT_DEV_ASSERT(slot >= 0 && slot < 16 && count >= 0);
        AFT_SPIN_GUARD_INIT();
        while (flags[slot] < count) { __builtin_amdgcn_s_sleep(2); AFT_SPIN_GUARD(); }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    auto signal_count = [&](int slot) {
        AFT_DEV_ASSERT(slot >= 0 && slot < 16);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_fetch_add(const_cast<lds_int *>(flags + slot), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };

    float wa2[36], wa3[48], bias3[4];
    if (matrix) {
        // ---- matrix waves: gather the MFMA A fragments (84 registers, kept for the whole kernel) ----
        const float *stage = smem + kStage;
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
        for (int e = 0; e < 4; ++e) bias3[e] = (TRAIN && !a.cb[2]) ? 0.f : a.cb[2][e + 4 * h];
        wait_count(4, 4);                        // the helpers' input plane
        conv1_columns(wave, c1mid, c1hi);
    } else {
        // ---- helper waves: borders, input plane, conv1 of the first columns ----
        const int ht = tid - 256;   // 0..255
        // zero padding of in0 and c1: LDS columns 0 and T + 1 (the symbol borders), and in0's rows outside the plane
        for (int i = ht; i < 9 * 2 * SP; i += 256) {
            const int pl = i / (2 * SP), rem = i - pl * 2 * SP, col = rem < SP ? 0 : T + 1, row = rem & (SP - 1);
            smem[pl * kPlane + col * SP + row] = 0.f;   // planes 0..8 = in0, c1[0..7]
        }
        for (int i = ht; i < T * 8; i += 256) {
            const int t = i >> 3, q = i & 7;
            in0[(t + 1) * SP + (q < 4 ? q : 120 + q)] = 0.f;   // rows 0..3 and 124..127
        }
        // input plane (one pixel per thread and pass, global reads coalesced; all 7 requests of a thread in flight together)
        {
            float v[7];
#pragma unroll
            for (int u = 0; u < 7; ++u) {
                const int i = ht + 256 * u;
                v[u] = 0.f;
                if (i < S * T) {
                    if (MODE == 0) {
                        v[u] = a.in_plane[(size_t)n * (S * T) + i];
                    } else {   // inverse patch map + conv_enhanced residual: feature f of token (g, tc) is pixel (g*p0 + f/p1, tc*p1 + f%p1)
                        const int gr = i / T, t = i - gr * T, p0 = a.p0, p1 = a.p1, tpr = T / p1;
                        const int g = gr / p0, tc = t / p1, f = (gr - g * p0) * p1 + (t - tc * p1);
                        v[u] = a.lin2_out[((size_t)n * a.tokens + g * tpr + tc) * a.lin2_stride + f] + a.resid[(size_t)n * (S * T) + i];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 7; ++u) {
                const int i = ht + 256 * u, gr = i / T, t = i - gr * T;
                if (i < S * T) in0[(t + 1) * SP + gr + 4] = v[u];
            }
        }
        signal_count(4);
        wait_count(4, 4);
        conv1_columns(wave - 4, c1lo, c1mid);
    }
    SSTAMP(2);
    __syncthreads();   // fragments gathered (the staging area is dead), c1 complete (in0 is dead: it becomes the output plane)
    SSTAMP(3);

    if (matrix) {
        // ---- conv2 + conv3 on the matrix cores: row tile `wave`, columns 0 .. T-1 (k_conv.hip's pipelined sweep, one segment) ----
        const int r = 4 + kTileRows * wave - 1 + j;                       // this lane's local row
        const int gr = r - 4;
        const bool ok2 = gr >= 0 && gr < S;                              // conv2 output inside the plane (else zero padding)
        const float relu_hi = ok2 ? __builtin_inff() : 0.f;
        const bool ok3 = ok2 && j >= 1 && j <= kTileRows && r < LR - 3;
        const float *bsrc = c1 + (4 * h) * kPlane + r - 1;                // + c*plane + (t'+kx)*SP + ky
        float *dst = c3 + (4 * h) * kPlane + r;
        f32x16 acc3;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc3[e] = 0.f;
        // conv3 output columns [0, columns_done) of this tile are in LDS.  No s_waitcnt here: the LDS serves one wave's requests in
        // issue order, so a reader that sees the flag finds the data stores before it already done; the compiler barrier keeps the
        // flag store behind them in the instruction stream (a release fence would also drain this wave's pending operand reads:
        // a stall of the MFMA chain at every column)
        auto publish = [&](int columns_done) {
            AFT_DEV_ASSERT(wave >= 0 && wave < 4 && columns_done >= 0 && columns_done <= T);
            asm volatile("" ::: "memory");
            AFT_CHECKED_FENCE();
            if (lane == 0) flags[wave] = columns_done;
            asm volatile("" ::: "memory");
        };
        auto store_col = [&](int tout, float v0, float v1, float v2, float v3) {
            if (tout < c3lo || tout >= c3hi) return;      // (a range's first / last conv3 sums lack a conv2 column of the neighbour range)
            if (ok3) {
                float *p = dst + (tout + 1) * SP;
                const float v[4] = {v0, v1, v2, v3};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float y = fmaxf(v[k] + bias3[k], 0.f);
                    if constexpr (TRAIN) {
                        const unsigned gi = ((unsigned)(n * 8 + 4 * h + k) * T + tout) * S + gr;
                        if (a.mask[2]) y = conv_ld(conv_srd(a.mask[2]), gi) > 0.f ? v[k] : 0.f;
                        if (a.save[2]) conv_st(conv_srd(a.save[2]), gi, y);
                    }
                    p[k * kPlane] = y;
                }
            }
            publish(tout + 1);
        };
        auto b_at = [&](int kb, int tcol) {
            const int tap = kb >> 2, kx = tap / 3, ky = tap % 3;
            return bsrc[(kb & 3) * kPlane + (tcol + kx) * SP + ky];
        };
        float b[36];
#pragma unroll
        for (int kb = 0; kb < 36; ++kb) b[kb] = b_at(kb, c2lo);
        // conv2's bias in accumulator-register order, held in registers: every column's chain STARTS from it as the C operand of
        // its first MFMA (no copy, no LDS wait at the column boundary)
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
        // conv2's activation for column tcol: ReLU (or 0 outside the plane); training: masked by / saved as the stage tensor
        // (the mask values of column tcol were requested a column earlier: m1)
        float m1[TRAIN ? 16 : 1];
        const bool own_row = r >= 4 && r < 4 + S;
        auto request_mask = [&](int tcol) {
            if constexpr (TRAIN) {
                if (a.mask[1] && ok2) {
                    const ConvSrd m = conv_srd(a.mask[1]);
                    const unsigned g0 = ((unsigned)(n * 32 + 4 * h) * T + tcol) * S + gr, cstride = (unsigned)T * S;
#pragma unroll
                    for (int e = 0; e < 16; ++e) m1[e] = conv_ld(m, g0 + ((e & 3) + 8 * (e >> 2)) * cstride);
                }
            }
        };
        auto activate2 = [&](const f32x16 &acc2, int tcol) {
#pragma unroll
            for (int e = 0; e < 16; ++e) x2[e] = __builtin_amdgcn_fmed3f(acc2[e], 0.f, relu_hi);
            if constexpr (TRAIN) {
                if (a.mask[1]) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) x2[e] = (ok2 && m1[e] > 0.f) ? acc2[e] : 0.f;
                }
                if (a.save[1] && ok2 && own_row) {
                    const ConvSrd sv = conv_srd(a.save[1]);
                    const unsigned g0 = ((unsigned)(n * 32 + 4 * h) * T + tcol) * S + gr, cstride = (unsigned)T * S;
#pragma unroll
                    for (int e = 0; e < 16; ++e) conv_st(sv, g0 + ((e & 3) + 8 * (e >> 2)) * cstride, x2[e]);
                }
            }
        };
        request_mask(c2lo);
        {   // prologue: conv2 of the first column
            f32x16 acc2 = mfma_f32(wa2[0], b[0], bias2v);
            b[0] = b_at(0, c2lo + 1);
#pragma unroll
            for (int kb = 1; kb < 36; ++kb) {
                acc2 = mfma_f32(wa2[kb], b[kb], acc2);
                b[kb] = b_at(kb, c2lo + 1);
            }
            activate2(acc2, c2lo);
        }
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
        constexpr int kLead = 4;   // conv2 MFMAs of the next column issued BEFORE the column hand-over (store + rotate wait for conv3's last MFMA)
#pragma unroll 1
        for (int tcol = c2lo; tcol < c2hi - 1; ++tcol) {
            const int tnext = min(tcol + 2, c2hi - 1);
            request_mask(tcol + 1);
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
            // 48 conv3 MFMAs of column tcol interleaved with the remaining 32 conv2 MFMAs of column tcol + 1 (3 : 2)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                conv3_step(3 * g);
                acc2 = mfma_f32(wa2[kLead + 2 * g], b[kLead + 2 * g], acc2);
                b[kLead + 2 * g] = b_at(kLead + 2 * g, tnext);
                conv3_step(3 * g + 1);
                acc2 = mfma_f32(wa2[kLead + 2 * g + 1], b[kLead + 2 * g + 1], acc2);
                b[kLead + 2 * g + 1] = b_at(kLead + 2 * g + 1, tnext);
                conv3_step(3 * g + 2);
            }
            activate2(acc2, tcol + 1);
        }
        store_col(c2hi - 3, acc3[8], acc3[9], acc3[10], acc3[11]);
        rotate();
#pragma unroll
        for (int i = 0; i < 48; ++i) conv3_step(i);      // conv3 of the last column
        store_col(c2hi - 2, acc3[8], acc3[9], acc3[10], acc3[11]);
        store_col(c2hi - 1, acc3[4], acc3[5], acc3[6], acc3[7]);   // whole plane: registers 0..3 would be column T (zero padding); a range that ends inside the plane drops it
    } else {
        // ---- helper waves: conv3's zero borders, then conv4 (8 -> 1) column by column behind the matrix waves ----
        const int ht = tid - 256;
        for (int i = ht; i < 8 * 2 * SP; i += 256) {
            const int pl = i / (2 * SP), rem = i - pl * 2 * SP, col = rem < SP ? 0 : T + 1, row = rem & (SP - 1);
            c3[pl * kPlane + col * SP + row] = 0.f;
        }
        for (int i = ht; i < 8 * T * 2; i += 256) {   // rows 3 and 124: plane rows -1 and S (never stored by conv3)
            const int pl = i / (2 * T), rem = i - pl * 2 * T, t = rem >> 1;
            c3[pl * kPlane + (t + 1) * SP + ((rem & 1) ? 124 : 3)] = 0.f;
        }
        signal_count(5);
        wait_count(5, 4);
        // thread = (local row lr, input-channel half): 4 channels x 9 taps, the halves meet through one lane swap
        const int lr = 32 * (wave - 4) + j;
        const bool okrow = lr >= 4 && lr < 4 + S;
        const int r0 = max(lr - 1, 0), r2 = min(lr + 1, LR - 1);
        float w[4][9];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int k9 = 0; k9 < 9; ++k9) w[c][k9] = w4s[(4 * h + c) * 9 + k9];
        const float b4 = w4s[72];
        const float *src = c3 + (4 * h) * kPlane;
        float win[4][3][3];   // [channel][ky][kx]
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) win[c][ky][1] = win[c][ky][2] = 0.f;
        float *obuf = in0;   // output plane [S][T] row-major (in0 is dead)
        int have = 0;        // conv3 columns known to be published by all four matrix waves
        auto need = [&](int columns) {
            if (have >= columns) return;
            AFT_DEV_ASSERT(columns >= 0 && columns <= T);
            AFT_SPIN_GUARD_INIT();
            for (;;) {   // one 16-byte LDS read per poll, a long sleep between polls: a polling wave costs its SIMD's matrix wave issue slots
                const i32x4 f = *(const volatile lds_i32x4 *)(flags);
                const int m = min(min(f[0], f[1]), min(f[2], f[3]));
                if (m >= columns) { have = m; break; }
                AFT_SPIN_GUARD();
                __builtin_amdgcn_s_sleep(8);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        };
        auto load_col = [&](int ldscol) {   // window column kx = 2 <- LDS column `ldscol`
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float *p = src + c * kPlane + ldscol * SP;
                win[c][0][2] = p[r0];
                win[c][1][2] = p[lr];
                win[c][2][2] = p[r2];
            }
        };
        // window before the loop: kx = 1 <- LDS column ta (symbol ta - 1, or the zero border), kx = 2 <- LDS column ta + 1 (symbol ta)
        need(ta + 1);
        load_col(ta);
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) win[c][ky][1] = win[c][ky][2];
        load_col(ta + 1);
#pragma unroll
        for (int u = 0; u < T; ++u) {        // fully unrolled: the window slides by renaming, no register moves
            const int t = ta + u;
            if (t >= tb) break;
            need(min(t + 2, c3hi));          // symbol t + 1 (LDS column t + 2; beyond the plane it is the zero border)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) { win[c][ky][0] = win[c][ky][1]; win[c][ky][1] = win[c][ky][2]; }
            load_col(t + 2);
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int k9 = 0; k9 < 9; ++k9) acc = fmaf(win[c][k9 / 3][k9 % 3], w[c][k9], acc);
            acc += other_half32(acc);
            if (h == 0 && okrow) obuf[(lr - 4) * T + t] = acc + b4;
        }
    }
    SSTAMP(4);
    __syncthreads();
    SSTAMP(5);
    // ---- the output plane leaves in one coalesced pass ----
    {
        const float *obuf = in0;
        if (MODE == 0 && ranges > 1) {      // this range's columns only
            float *dst = a.out_plane + (size_t)n * (S * T);
            for (int i = tid; i < S * T; i += kConvThreads) {
                const int t = i % T;
                if (t >= ta && t < tb) dst[i] = obuf[i];
            }
        } else if (MODE == 0) {
            f32x4 *dst = reinterpret_cast<f32x4 *>(a.out_plane + (size_t)n * (S * T));
            const f32x4 *src4 = reinterpret_cast<const f32x4 *>(obuf);
            for (int i = tid; i < S * T / 4; i += kConvThreads) dst[i] = src4[i];
        } else {   // interleave this plane into the complex64 output (the frame's other plane is another workgroup's)
            float *dst = a.out_complex + (size_t)frame * (S * T) * 2 + part;
            for (int i = tid; i < S * T; i += kConvThreads) dst[2 * i] = obuf[i];
        }
    }
    SSTAMP(6);
#undef SSTAMP
}


// =====================================================================================================================================
// conv_stream16_kernel (round 5): the same pipeline with conv2 / conv3 as v_mfma_f32_16x16x4_f32 chains and no staging phase.
//
// Matrix phase.  The 32x32x2 form above spends 48 MFMAs of 64 cycles per column on conv3 with 24 of 32 product rows used (row =
// (kx, co)).  Here ALL 72 (ky, kx, co) products of an input pixel are rows of ONE operand -- 4.5 tiles of 16 rows -> 5 tiles x 2 pixel
// tiles x 8 k-steps = 80 MFMAs of 32 cycles (2 560 instead of 3 072 cycles per column); conv2 is 2 x 2 tiles x 18 k-steps = 72 MFMAs
// (2 304 cycles, as before).  What makes it work:
//  * pixel tile pt of a wave's 32 rows = the rows of parity pt (row j = 2 p + pt, p = lane % 16).  The ky = -1 / +1 products are
//    needed one row further down / up, i.e. in the OTHER pixel tile's register at the same lane or one lane along the 16-lane DPP row
//    -- no carries between tiles, and conv2's accumulators are conv3's B operands unshifted (the 32 DPP moves per column of the 32x32
//    form are gone; the ky sums are 8 adds per output column);
//  * the kx products of input column t belong to output column t + 1 - kx, so the six (ky, co half) registers of tap column kx
//    become those of kx + 1 one column later: the rotation that sums the kx taps is the C operand of each tile's FIRST MFMA of a
//    column.  Tiles 0, 1, 2 hold (ky 0 / 1, co half) of kx = 0, 1, 2 -- their rotation is a whole tuple, no copies -- and the ky = 2
//    pairs share tiles 3 = [idle, idle, kx 0] and 4 = [kx 1, kx 2]: 4 register copies per pixel tile and column (a layout with six
//    consecutive registers per kx needed 12); conv3's bias is the initial value of the fresh kx = 0 centre-row registers.
// Staging.  The forward's prologue launch leaves both stacks' weights as operand fragments in the workspace (conv_frag16_entry): a
// matrix wave fetches its 76 fragment registers with 22 lane-linear 16-byte loads and starts sweeping as soon as the helper waves
// have published conv1's first two columns -- no transposed LDS staging, no gather, no workgroup barrier before the sweeps (the
// 32x32 kernel: 1 950 + 5 300 + 1 050 cycles before its first MFMA).  The helper waves run conv1 of ALL columns (one LDS flag per
// helper wave and column), in the shadow of the first sweeps.
// Column ranges (NSPLIT = 2, 4).  With fewer planes than half the CUs (batches of 64 frames and fewer -- the reference's default
// batch is 64, parser.py:81) a plane is worked by NSPLIT workgroups, each producing the output columns [ta, tb) of its range.  What a
// range needs of the earlier stages reaches one column further per stage -- conv3 [ta-1, tb+1), conv2 [ta-2, tb+2), conv1
// [ta-3, tb+3), input [ta-4, tb+4), clipped to the grid -- and is RECOMPUTED, not exchanged (7 output columns cost 9 conv2 and 8
// conv3 sweeps: 0.6 of a whole plane).  The LDS planes keep their whole-grid layout and absolute column numbers; the loops run over
// the range, the hand-over counters stay absolute.  Every output element goes through the same instruction sequence whatever the
// split, so a batch of 16 frames (four ranges per plane) reproduces the same frames inside a batch of 128 bit for bit.
// =====================================================================================================================================
// TRAIN (round 6; MODE 0 = plain plane in, plain plane out): what conv_stream_kernel<0, true> adds -- the three hidden activations saved
// as [plane][C][T][S] (backward: the pre-activation gradients the weight-gradient kernels need) and a masked activation (backward: the
// ReLU derivative from the forward's saved activation instead of bias + ReLU) -- in this kernel's layouts: conv2's 32 channels of a
// pixel pair leave from the accumulators as they lie (channel 16 mt + 4 (lane / 16) + v, rows r0 / r0 + 1 of pixel tiles 0 / 1), the
// mask values of a column are requested one column ahead (conv2) / one iteration ahead (conv3), so no sweep waits for them.
template <int MODE, int NSPLIT = 1, bool TRAIN = false>
__global__ __launch_bounds__(kConvThreads) void conv_stream16_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *in0 = smem + kIn0, *c1 = smem + kC1, *c3 = smem + kC3;
    // flags [0..3]: conv3 columns published by matrix wave w; [4], [5]: helper-only counters; [8..11]: conv1 columns published by helper wave
    volatile lds_int *flags = (volatile lds_int *)(smem + kFlags);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const int n = a.plane0 + (NSPLIT == 1 ? blockIdx.x : blockIdx.x / NSPLIT), frame = n >> 1, part = n & 1;
    // this workgroup's output columns [ta, tb) and what it computes of each earlier stage (compile-time constants for NSPLIT = 1)
    const int range = NSPLIT == 1 ? 0 : blockIdx.x % NSPLIT;
    const int ta = NSPLIT == 1 ? 0 : (NSPLIT == 2 ? 7 * range : (range < 2 ? 4 * range : 8 + 3 * (range - 2)));
    const int tb = NSPLIT == 1 ? T : (NSPLIT == 2 ? ta + 7 : (range < 2 ? ta + 4 : ta + 3));
    const int c2lo = max(ta - 2, 0), c2hi = min(tb + 2, T);      // conv2 columns swept
    const int c1lo = max(ta - 3, 0), c1hi = min(tb + 3, T);      // conv1 columns computed
    const int inlo = max(ta - 4, 0), inhi = min(tb + 4, T);      // input columns fetched
    const int c3first = c2lo == 0 ? 0 : c2lo + 1;                // first conv3 column with all three kx taps inside the sweep (= ta - 1)
    const bool matrix = wave < 4;
#ifdef AFT_DIAG_STAMPS
#define SSTAMP(i) do { if (a.stamps && (tid == 0 || tid == 256)) a.stamps[(size_t)blockIdx.x * 16 + (tid ? 8 : 0) + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SSTAMP(i) do { } while (0)
#endif
    SSTAMP(0);
    // the kernel's first instructions are its global requests: the matrix waves' operand fragments, the helpers' input plane (one pixel
    // per thread and pass, coalesced; all 7 requests of a thread in flight together) -- the input's trip from the previous launch's
    // output (another XCD's L2 or the memory side) is the one latency nothing in this kernel can hide
    f32x4 fq[kFragQuads];
    float vin[7];
    // training, backward: the forward's conv3 activation decides the conv1 stage's mask -- every column's four values of a helper thread
    // (local row 32 hw + j, channel half h) are requested here, with the input plane (column by column inside the conv1 loop the
    // helpers, which the sweeps wait for, paid one round trip per column)
    float mk[TRAIN ? T : 1][TRAIN ? 4 : 1];
    if constexpr (TRAIN) {
        if (!matrix && a.mask[0]) {
            const int mlr = 32 * (wave - 4) + j, mgr = mlr - 4;
            if (mlr >= 1 && mlr < LR - 1 && mgr >= 0 && mgr < S) {
                const ConvSrd m = conv_srd(a.mask[0]);
#pragma unroll
                for (int t = 0; t < T; ++t)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        mk[t][c] = (NSPLIT == 1 || (t >= c1lo && t < c1hi)) ? conv_ld(m, ((unsigned)(n * 8 + 4 * h + c) * T + t) * S + mgr) : 0.f;
            }
        }
    }
    if (!matrix) {
#pragma unroll
        for (int u = 0; u < 7; ++u) {
            const int i = tid - 256 + 256 * u;
            vin[u] = 0.f;
            if (i < S * T && (NSPLIT == 1 || (i % T >= inlo && i % T < inhi))) {
                if (MODE == 0) {
                    vin[u] = a.in_plane[(size_t)n * (S * T) + i];
                } else {   // inverse patch map + conv_enhanced residual: feature f of token (g, tc) is pixel (g*p0 + f/p1, tc*p1 + f%p1)
                    const int gr = i / T, t = i - gr * T, p0 = a.p0, p1 = a.p1, tpr = T / p1;
                    const int g = gr / p0, tc = t / p1, f = (gr - g * p0) * p1 + (t - tc * p1);
                    vin[u] = a.lin2_out[((size_t)n * a.tokens + g * tpr + tc) * a.lin2_stride + f] + a.resid[(size_t)n * (S * T) + i];
                }
            }
        }
    }
    if (tid < 16) flags[tid] = 0;
    // a bare barrier behind the LDS stores: __syncthreads() would also wait for the global requests just issued (vmcnt(0)), i.e. every
    // wave would sit here until the slowest wave's data had arrived (2 000 cycles, stamps)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    SSTAMP(1);
    if (matrix) {   // behind the helpers' requests: 88 KB of fragments per workgroup would otherwise queue in front of the input plane
        __builtin_amdgcn_s_sleep(3);
        const f32x4 *fp = reinterpret_cast<const f32x4 *>(a.wfrag) + lane;
#pragma unroll
        for (int q = 0; q < kFragQuads; ++q) fq[q] = fp[q * 64];
    }
    auto wait_count = [&](int slot, int count) {   // workgroup-scope hand-over through an LDS counter
        AFT_DEV_ASSERT(slot >= 0 && slot < 16 && count >= 0);
        AFT_SPIN_GUARD_INIT();
        while (flags[slot] < count) { __builtin_amdgcn_s_sleep(2); AFT_SPIN_GUARD(); }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    auto signal_count = [&](int slot) {
        AFT_DEV_ASSERT(slot >= 0 && slot < 16);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_fetch_add(const_cast<lds_int *>(flags + slot), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };

    if (matrix) {
        // ---- operand fragments straight from the packed image ----
        float wa2[36], wa3[40], bias3[2];
        f32x4 bias2v[2];
        {
#pragma unroll
            for (int f = 0; f < 36; ++f) wa2[f] = fq[f >> 2][f & 3];
#pragma unroll
            for (int f = 0; f < 40; ++f) wa3[f] = fq[9 + (f >> 2)][f & 3];
            bias3[0] = fq[19][0]; bias3[1] = fq[19][1];
            bias2v[0] = fq[20]; bias2v[1] = fq[21];
        }
        const int p = lane & 15, g = lane >> 4;
        const int r0 = 4 + kTileRows * wave - 1 + 2 * p;                 // local row of this lane's pixel in tile pt = 0 (pt = 1: r0 + 1)
        bool ok3[2], ok2v[2];
        float relu_hi[2];
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            const int jj = 2 * p + pt, r = r0 + pt, gr = r - 4;
            const bool ok2 = gr >= 0 && gr < S;                          // conv2 output inside the plane (else zero padding)
            ok2v[pt] = ok2;
            relu_hi[pt] = ok2 ? __builtin_inff() : 0.f;
            ok3[pt] = ok2 && jj >= 1 && jj <= kTileRows && r < LR - 3;
        }
        // training: element index of (stage channel c, column t, this lane's row of pixel tile pt) in a [plane][C][T][S] stage tensor
        const unsigned grow0 = (unsigned)(r0 - 4);                       // plane row of pixel tile 0 (tile 1: + 1); valid where ok2v / ok3 say so
        auto stage_index = [&](int nch, int c, int t, int pt) { return ((unsigned)(n * nch + c) * T + (unsigned)t) * S + grow0 + (unsigned)pt; };
        // conv2's B operands: for (kx, ci half) the four rows r0 - 1 .. r0 + 2 of conv1's column t' + kx serve both pixel tiles and
        // the three ky (row r0 + pt + ky - 1): 24 registers per column, two 8-byte LDS reads per (kx, ci half)
        const float *bsrc = c1 + g * kPlane + r0 - 1;                    // + 4 cih planes + (t' + kx) SP
        // stores of finished conv3 columns: halo lanes (row j = 0 / 31 of the tile; with 4 x 30 = 120 rows these are the only
        // invalid ones) write to row 0 of the plane instead, which nobody reads into a stored result
        float *dst[2];
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) dst[pt] = c3 + g * kPlane + (ok3[pt] ? r0 + pt : 0);   // + 4 cohi planes + (t + 1) SP
        // conv3 output columns [0, columns_done) of this tile are in LDS.  No s_waitcnt: the LDS serves one wave's requests in issue
        // order, so a reader that sees the flag finds the data stores before it already done; the compiler barriers keep the flag
        // store behind them in the instruction stream (a release fence would drain this wave's pending operand reads)
        auto publish = [&](int columns_done) {
            AFT_DEV_ASSERT(wave >= 0 && wave < 4 && columns_done >= 0 && columns_done <= T);
            asm volatile("" ::: "memory");
            AFT_CHECKED_FENCE();
            if (lane == 0) flags[wave] = columns_done;
            asm volatile("" ::: "memory");
        };
        int have1 = 0;   // conv1 columns known to be published by all four helper waves (wave-uniform)
        auto need_c1 = [&](int columns) {
            if (have1 >= columns) return;
            AFT_DEV_ASSERT(columns >= 0 && columns <= T);
            AFT_SPIN_GUARD_INIT();
            for (;;) {
                const i32x4 f = *(const volatile lds_i32x4 *)(flags + 8);
                const int m = __builtin_amdgcn_readfirstlane(min(min(f[0], f[1]), min(f[2], f[3])));
                if (m >= columns) { have1 = m; break; }
                AFT_SPIN_GUARD();
                __builtin_amdgcn_s_sleep(1);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        };
        float bv[3][2][4];
        auto load_b = [&](int kx, int cih, int tcol) {
            const float *q = bsrc + 4 * cih * kPlane + (tcol + kx) * SP;
            const f32x2 lo = *reinterpret_cast<const f32x2 *>(q), hi = *reinterpret_cast<const f32x2 *>(q + 2);
            bv[kx][cih][0] = lo[0]; bv[kx][cih][1] = lo[1]; bv[kx][cih][2] = hi[0]; bv[kx][cih][3] = hi[1];
        };
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        const f32x4 bias3v = {0.f, 0.f, bias3[0], bias3[1]};        // tile 0 = fresh kx 0 registers [(ky 0, c), (ky 1 = centre row, c)]
        f32x4 a3[2][5];
#pragma unroll
        for (int pt = 0; pt < 2; ++pt)
#pragma unroll
            for (int rt = 0; rt < 5; ++rt) a3[pt][rt] = rt == 0 ? bias3v : zero4;   // as if the zero column -1 had been swept: output column 0's bias
        f32x4 x2[2][2];   // [pt][mt]: conv2's activation of the current column = conv3's B operands
        f32x4 acc2[2][2];
        f32x4 m2[TRAIN ? 2 : 1][TRAIN ? 2 : 1];   // training, backward: the forward's conv2 activation of the column being swept (the mask)
        float m3[TRAIN ? 2 : 1][TRAIN ? 2 : 1];   // ... conv3's, of the next column to be stored: [pt][co half]
        auto request_mask2 = [&](int tcol) {       // one column ahead of relu2(tcol)
            if constexpr (TRAIN) {
                if (a.mask[1]) {
                    const ConvSrd m = conv_srd(a.mask[1]);
                    if (ok2v[0] && ok2v[1]) {      // the pixel pair of a channel: one 8-byte load
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                            for (int v = 0; v < 4; ++v) {
                                const f32x2 pr = conv_ld2(m, stage_index(32, 16 * mt + 4 * g + v, tcol, 0));
                                m2[0][mt][v] = pr[0];
                                m2[1][mt][v] = pr[1];
                            }
                    } else {
#pragma unroll
                        for (int pt = 0; pt < 2; ++pt)
#pragma unroll
                            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                                for (int v = 0; v < 4; ++v)
                                    m2[pt][mt][v] = ok2v[pt] ? conv_ld(m, stage_index(32, 16 * mt + 4 * g + v, tcol, pt)) : 0.f;
                    }
                }
            }
        };
        auto request_mask3 = [&](int tout) {       // ahead of store_col(tout)
            if constexpr (TRAIN) {
                if (a.mask[2] && tout >= 0 && tout < T) {
                    const ConvSrd m = conv_srd(a.mask[2]);
#pragma unroll
                    for (int pt = 0; pt < 2; ++pt)
#pragma unroll
                        for (int cohi = 0; cohi < 2; ++cohi) m3[pt][cohi] = ok3[pt] ? conv_ld(m, stage_index(8, 4 * cohi + g, tout, pt)) : 0.f;
                }
            }
        };
        auto relu2 = [&](int tcol) {
            (void)tcol;
#pragma unroll
            for (int pt = 0; pt < 2; ++pt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int v = 0; v < 4; ++v) x2[pt][mt][v] = __builtin_amdgcn_fmed3f(acc2[pt][mt][v], 0.f, relu_hi[pt]);
            if constexpr (TRAIN) {
                if (a.mask[1]) {
#pragma unroll
                    for (int pt = 0; pt < 2; ++pt)
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                            for (int v = 0; v < 4; ++v) x2[pt][mt][v] = (ok2v[pt] && m2[pt][mt][v] > 0.f) ? acc2[pt][mt][v] : 0.f;
                }
                if (a.save[1]) {       // the pixel pair (rows r0, r0 + 1 = tiles 0, 1) of a channel is 8 contiguous bytes of the stage tensor
                    const ConvSrd sv = conv_srd(a.save[1]);
                    if (ok2v[0] && ok2v[1]) {
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                            for (int v = 0; v < 4; ++v) conv_st2(sv, stage_index(32, 16 * mt + 4 * g + v, tcol, 0), x2[0][mt][v], x2[1][mt][v]);
                    } else {
#pragma unroll
                        for (int pt = 0; pt < 2; ++pt)
#pragma unroll
                            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                                for (int v = 0; v < 4; ++v)
                                    if (ok2v[pt]) conv_st(sv, stage_index(32, 16 * mt + 4 * g + v, tcol, pt), x2[pt][mt][v]);
                    }
                }
            }
        };
        // conv2 MFMA u (0..11) of group gi = (kx, ci half): ky = u / 4, pixel tile (u / 2) % 2, channel tile u % 2
        auto conv2_step = [&](int gi, int u) {
            const int kx = gi >> 1, cih = gi & 1, ky = u >> 2, pt = (u >> 1) & 1, mt = u & 1, ks2 = 2 * (kx * 3 + ky) + cih;
            acc2[pt][mt] = mfma16(wa2[mt * 18 + ks2], bv[kx][cih][ky + pt], (gi == 0 && ky == 0) ? bias2v[mt] : acc2[pt][mt]);
        };
        // the C operand of tile rt's first MFMA of a column: the previous column's registers one kx further (header comment)
        auto rotated = [&](int pt, int rt) -> f32x4 {
            if (rt == 0) return bias3v;
            if (rt < 3) return a3[pt][rt - 1];
            if (rt == 3) return zero4;
            return f32x4{a3[pt][3][2], a3[pt][3][3], a3[pt][4][0], a3[pt][4][1]};
        };
        f32x4 n3[2][5];
        // conv3 MFMA m (0..79): k-step m / 10; the tiles whose registers finish a column (2 and 4) first
        auto conv3_step = [&](int m) {
            constexpr int kOrder[5] = {2, 4, 1, 0, 3};
            const int ks = m / 10, i = m % 10, rt = kOrder[i >> 1], pt = i & 1;
            n3[pt][rt] = mfma16(wa3[rt * 8 + ks], x2[pt][ks >> 2][ks & 3], ks == 0 ? rotated(pt, rt) : n3[pt][rt]);
        };
        // conv3's output column `tout` from six finished registers (ky, co half) of both pixel tiles
        auto store_col = [&](int tout, const float (&Y)[2][6]) {
            if (tout < c3first) return;    // (wave-uniform: a scalar branch) columns before the sweep's first complete one
            float *p0 = dst[0], *p1 = dst[1];
#pragma unroll
            for (int cohi = 0; cohi < 2; ++cohi) {
                // row j = 2p (pt 0): ky = 0 from row j - 1 = (pt 1, lane p - 1), ky = 2 from row j + 1 = (pt 1, lane p)
                // row j = 2p + 1 (pt 1): ky = 0 from (pt 0, lane p), ky = 2 from (pt 0, lane p + 1)
                float o0 = Y[0][2 + cohi] + row16_from_below(Y[1][cohi]) + Y[1][4 + cohi];
                float o1 = Y[1][2 + cohi] + Y[0][cohi] + row16_from_above(Y[0][4 + cohi]);
                if (TRAIN && a.mask[2]) {  // backward: the forward's saved activation decides (no bias, no ReLU)
                    o0 = m3[0][cohi] > 0.f ? o0 : 0.f;
                    o1 = m3[1][cohi] > 0.f ? o1 : 0.f;
                } else {
                    o0 = fmaxf(o0, 0.f);   // (the bias rode in as the accumulator's initial value)
                    o1 = fmaxf(o1, 0.f);
                }
                if constexpr (TRAIN) {
                    if (a.save[2]) {
                        const ConvSrd sv = conv_srd(a.save[2]);
                        if (ok3[0] && ok3[1]) conv_st2(sv, stage_index(8, 4 * cohi + g, tout, 0), o0, o1);
                        else {
                            if (ok3[0]) conv_st(sv, stage_index(8, 4 * cohi + g, tout, 0), o0);
                            if (ok3[1]) conv_st(sv, stage_index(8, 4 * cohi + g, tout, 1), o1);
                        }
                    }
                }
                p0[4 * cohi * kPlane + (tout + 1) * SP] = o0;
                p1[4 * cohi * kPlane + (tout + 1) * SP] = o1;
            }
            publish(tout + 1);
        };
        // the registers of `acc` that hold output column (input column - 1): the kx = 2 set, Y[pt][2 ky + co half]
        auto finished = [&](const f32x4 (&acc)[2][5], float (&Y)[2][6]) {
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
#pragma unroll
                for (int v = 0; v < 4; ++v) Y[pt][v] = acc[pt][2][v];
                Y[pt][4] = acc[pt][4][2]; Y[pt][5] = acc[pt][4][3];
            }
        };
        SSTAMP(2);
        request_mask2(c2lo);
        request_mask3(c3first);
        need_c1(min(c2lo + 2, T));
#pragma unroll
        for (int gi = 0; gi < 6; ++gi) load_b(gi >> 1, gi & 1, c2lo);
        SSTAMP(3);
        // prologue: conv2 of the first column (conv1's third column is only needed behind the first group: the sweeps start a column earlier)
#pragma unroll
        for (int gi = 0; gi < 6; ++gi) {
#pragma unroll
            for (int u = 0; u < 12; ++u) conv2_step(gi, u);
            if (gi == 0) need_c1(min(c2lo + 3, T));
            load_b(gi >> 1, gi & 1, c2lo + 1);
        }
        relu2(c2lo);
#pragma unroll 1
        for (int tcol = c2lo; tcol < c2hi - 1; ++tcol) {
            const int tnext = min(tcol + 2, c2hi - 1);
            request_mask2(tcol + 1);       // (training, backward) the column whose conv2 this iteration runs
            need_c1(min(tnext + 2, T));
            // 80 conv3 MFMAs of column tcol merged with the 72 conv2 MFMAs of column tcol + 1; the B operands of a conv2 group are
            // re-requested for column tcol + 2 right behind their last use (pinned: the compiler otherwise sinks all reads to the end
            // of the column and the next column starts with an LDS round trip).  Output column tcol - 2 -- finished by the PREVIOUS
            // iteration -- is combined and stored behind the first group, when its registers have long left the matrix pipe.
#pragma unroll
            for (int gi = 0; gi < 6; ++gi) {
                const int m0 = kC3Split[gi], m1 = kC3Split[gi + 1];
#pragma unroll
                for (int u = 0; u < 12; ++u) {
                    conv3_step(m0 + u);
                    conv2_step(gi, u);
                }
                load_b(gi >> 1, gi & 1, tnext);
                if (m1 - m0 > 12) {   // the tail of conv3 behind conv2's last MFMA (pinned: the compiler otherwise moves conv2's MFMAs to the end)
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 12; u < 20; ++u)
                        if (m0 + u < m1) conv3_step(m0 + u);
                }
                if (gi == 0) {
                    float Y[2][6];
                    finished(a3, Y);
                    store_col(tcol - 2, Y);
                    request_mask3(max(tcol - 1, c3first));     // the next iteration's column (the first complete one until the sweep reaches it)
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            relu2(tcol + 1);
#pragma unroll
            for (int pt = 0; pt < 2; ++pt)
#pragma unroll
                for (int rt = 0; rt < 5; ++rt) a3[pt][rt] = n3[pt][rt];
        }
        {   // output column c2hi - 3, then conv3 of the last swept column: its kx = 2 registers are column c2hi - 2, its kx = 1 registers
            // column c2hi - 1 -- complete only at the grid's edge (column T is zero padding), not needed inside it
            float Y[2][6], Z[2][6];
            finished(a3, Y);
            store_col(c2hi - 3, Y);
            request_mask3(c2hi - 2);
#pragma unroll
            for (int m = 0; m < 80; ++m)
                if (m % 10 < 6) conv3_step(m);     // tiles 2, 4, 1 only: tiles 0, 3 hold the kx = 0 registers = output column T (padding)
            finished(n3, Y);
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
#pragma unroll
                for (int v = 0; v < 4; ++v) Z[pt][v] = n3[pt][1][v];
                Z[pt][4] = n3[pt][4][0]; Z[pt][5] = n3[pt][4][1];
            }
            store_col(c2hi - 2, Y);
            if (NSPLIT == 1 || c2hi == T) {
                request_mask3(T - 1);
                store_col(T - 1, Z);
            }
        }
    } else {
        // ---- helper waves: input plane, borders, conv1 of all columns (published column by column), conv4 behind the matrix waves ----
        // Priority: a SIMD's issue arbiter serves the OLDER wave first at equal priority, and the matrix wave always has an independent
        // MFMA ready -- at equal priority the helper got one issue slot per ~70 cycles (conv1 of 14 columns: 45 700 cycles of wall time
        // for ~3 000 cycles of vector work, conv4 then finished 5 500 cycles behind the last sweep; stamps, round 5).  The helpers'
        // vector work is on the matrix waves' critical path either way (they consume conv1's columns), so it goes first.
        __builtin_amdgcn_s_setprio(AFT_HELPER_PRIO);
        const int ht = tid - 256, hw = wave - 4;
        // conv1's weights: thread = (local row lr = 32 hw + j, channel half h), two channels per v_pk_fma_f32
        f32x2 w[2][9], b[2];
        f32x4 hq[20];   // this channel half's tables (conv_helper_entry): conv1 in quads 0..9, conv4 in 10..19
        {
            const f32x4 *hp = reinterpret_cast<const f32x4 *>(a.wfrag + kFragQuads * 64 * 4 + 80 * h);
#pragma unroll
            for (int q = 0; q < 20; ++q) hq[q] = hp[q];
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
#pragma unroll
            for (int k9 = 0; k9 < 9; ++k9) {
                const int i = 2 * (9 * k + k9);
                w[k][k9] = f32x2{hq[i >> 2][i & 3], hq[i >> 2][(i & 3) + 1]};
            }
            b[k] = f32x2{hq[9][2 * k], hq[9][2 * k + 1]};
        }
        // zero padding of in0 and c1: LDS columns 0 and T + 1 (the symbol borders), and in0's rows outside the plane
        for (int i = ht; i < 9 * 2 * SP; i += 256) {
            const int pl = i / (2 * SP), rem = i - pl * 2 * SP, col = rem < SP ? 0 : T + 1, row = rem & (SP - 1);
            smem[pl * kPlane + col * SP + row] = 0.f;   // planes 0..8 = in0, c1[0..7]
        }
        for (int i = ht; i < T * 8; i += 256) {
            const int t = i >> 3, q = i & 7;
            in0[(t + 1) * SP + (q < 4 ? q : 120 + q)] = 0.f;   // rows 0..3 and 124..127
        }
#pragma unroll
        for (int u = 0; u < 7; ++u) {
            const int i = ht + 256 * u, gr = i / T, t = i - gr * T;
            if (i < S * T) in0[(t + 1) * SP + gr + 4] = vin[u];
        }
        SSTAMP(7);    // (helper slot 7: the input plane has arrived and is in LDS)
        signal_count(4);
        wait_count(4, 4);
        SSTAMP(2);
        {   // conv1: 1 -> 8, ReLU, all T columns; the window slides by renaming (fully unrolled)
            const int lr = 32 * hw + j, gr = lr - 4;
            const bool ok = lr >= 1 && lr < LR - 1 && gr >= 0 && gr < S;
            const int r0 = max(lr - 1, 0), r2 = min(lr + 1, LR - 1);
            float win[3][3];   // [ky][kx]
#pragma unroll
            for (int kx = 1; kx < 3; ++kx) {
                win[0][kx] = in0[(kx - 1) * SP + r0];
                win[1][kx] = in0[(kx - 1) * SP + lr];
                win[2][kx] = in0[(kx - 1) * SP + r2];
            }
            float *dst = c1 + (4 * h) * kPlane + SP + lr;
#pragma unroll
            for (int t = 0; t < T; ++t) {
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) { win[ky][0] = win[ky][1]; win[ky][1] = win[ky][2]; }
                win[0][2] = in0[(t + 2) * SP + r0];
                win[1][2] = in0[(t + 2) * SP + lr];
                win[2][2] = in0[(t + 2) * SP + r2];
                if (NSPLIT == 1 || (t >= c1lo && t < c1hi)) {      // (wave-uniform) the window slides over every column, the work is the range's
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        f32x2 acc2 = b[k];
#pragma unroll
                        for (int k9 = 0; k9 < 9; ++k9) {
                            const float x = win[k9 / 3][k9 % 3];
                            acc2 = __builtin_elementwise_fma(f32x2{x, x}, w[k][k9], acc2);     // per channel: the same fma chain in tap order
                        }
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            float v = fmaxf(acc2[q], 0.f);
                            if constexpr (TRAIN) {
                                if (ok) {
                                    const unsigned gi = ((unsigned)(n * 8 + 4 * h + 2 * k + q) * T + t) * S + gr;
                                    if (a.mask[0]) v = mk[t][2 * k + q] > 0.f ? acc2[q] : 0.f;
                                    if (a.save[0]) conv_st(conv_srd(a.save[0]), gi, v);
                                }
                            }
                            dst[(2 * k + q) * kPlane + t * SP] = ok ? v : 0.f;
                        }
                    }
                }
                asm volatile("" ::: "memory");
                AFT_DEV_ASSERT(hw >= 0 && hw < 4 && t >= 0 && t < T);
                AFT_CHECKED_FENCE();
                if (lane == 0) flags[8 + hw] = t + 1;     // LDS order: the column's stores are done when a reader sees the count
                asm volatile("" ::: "memory");
            }
        }
        SSTAMP(3);
        // conv3's zero borders, then conv4 (8 -> 1) column by column behind the matrix waves
        for (int i = ht; i < 8 * 2 * SP; i += 256) {
            const int pl = i / (2 * SP), rem = i - pl * 2 * SP, col = rem < SP ? 0 : T + 1, row = rem & (SP - 1);
            c3[pl * kPlane + col * SP + row] = 0.f;
        }
        for (int i = ht; i < 8 * T * 2; i += 256) {   // rows 3 and 124: plane rows -1 and S (never stored by conv3)
            const int pl = i / (2 * T), rem = i - pl * 2 * T, t = rem >> 1;
            c3[pl * kPlane + (t + 1) * SP + ((rem & 1) ? 124 : 3)] = 0.f;
        }
        signal_count(5);        // every helper is through conv1 (in0 is dead: it becomes the output plane) and the borders
        wait_count(5, 4);
        // thread = (local row lr, input-channel half): 4 channels x 9 taps, the halves meet through one lane swap
        const int lr = 32 * hw + j;
        const bool okrow = lr >= 4 && lr < 4 + S;
        const int r0 = max(lr - 1, 0), r2 = min(lr + 1, LR - 1);
        f32x2 w4[2][9];   // [input-channel pair][tap]
#pragma unroll
        for (int cp = 0; cp < 2; ++cp)
#pragma unroll
            for (int k9 = 0; k9 < 9; ++k9) {
                const int i = 40 + 2 * (9 * cp + k9);
                w4[cp][k9] = f32x2{hq[i >> 2][i & 3], hq[i >> 2][(i & 3) + 1]};
            }
        const float b4 = hq[19][0];
        const float *src = c3 + (4 * h) * kPlane;
        float win[4][3][3];   // [channel][ky][kx]
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) win[c][ky][1] = win[c][ky][2] = 0.f;
        float *obuf = in0;   // output plane [S][T] row-major
        int have = 0;        // conv3 columns known to be published by all four matrix waves
        auto need = [&](int columns) {
            if (have >= columns) return;
            AFT_DEV_ASSERT(columns >= 0 && columns <= T);
            AFT_SPIN_GUARD_INIT();
            for (;;) {   // one 16-byte LDS read per poll, a long sleep between polls: a polling wave costs its SIMD's matrix wave issue slots
                const i32x4 f = *(const volatile lds_i32x4 *)(flags);
                const int m = min(min(f[0], f[1]), min(f[2], f[3]));
                if (m >= columns) { have = m; break; }
                AFT_SPIN_GUARD();
                // a poll is ~6 instructions on the SIMD the matrix wave needs: a column takes ~5 000 cycles, so sleep long while
                // columns are far apart and short only for the last ones, whose conv4 is the kernel's tail
                if (columns >= T - 1) __builtin_amdgcn_s_sleep(2);
                else __builtin_amdgcn_s_sleep(AFT_POLL_SLEEP);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        };
        auto load_col = [&](int ldscol) {   // window column kx = 2 <- LDS column `ldscol`
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float *pc = src + c * kPlane + ldscol * SP;
                win[c][0][2] = pc[r0];
                win[c][1][2] = pc[lr];
                win[c][2][2] = pc[r2];
            }
        };
        // window before the first output column ta: kx = 1 <- LDS column ta (symbol ta - 1; the zero border for ta = 0), kx = 2 <- symbol ta
        auto start_window = [&](int t0) {
            need(t0 + 1);
            load_col(t0);
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) win[c][ky][1] = win[c][ky][2];
            load_col(t0 + 1);
        };
        if (NSPLIT == 1) start_window(0);
#pragma unroll
        for (int t = 0; t < T; ++t) {        // fully unrolled: the window slides by renaming, no register moves
            if (NSPLIT != 1) {               // (wave-uniform) this range's columns only
                if (t < ta || t >= tb) continue;
                if (t == ta) start_window(t);
            }
            need(t + 2 < T ? t + 2 : T);     // symbol t + 1 (LDS column t + 2; the last one is the zero border)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) { win[c][ky][0] = win[c][ky][1]; win[c][ky][1] = win[c][ky][2]; }
            load_col(t + 2);
            // two channels per v_pk_fma_f32 (the helpers' vector work costs the SIMD's matrix wave ALU time: 36 -> 18 issue slots)
            f32x2 acc2[2] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};
#pragma unroll
            for (int cp = 0; cp < 2; ++cp)
#pragma unroll
                for (int k9 = 0; k9 < 9; ++k9)
                    acc2[cp] = __builtin_elementwise_fma(f32x2{win[2 * cp][k9 / 3][k9 % 3], win[2 * cp + 1][k9 / 3][k9 % 3]}, w4[cp][k9], acc2[cp]);
            const f32x2 s2 = acc2[0] + acc2[1];
            float acc = s2[0] + s2[1];
            acc += other_half32(acc);
            if (h == 0 && okrow) obuf[(lr - 4) * T + t] = acc + b4;
        }
    }
    SSTAMP(4);
    __syncthreads();
    SSTAMP(5);
    // ---- the output plane leaves in one coalesced pass ----
    {
        const float *obuf = in0;
        if (NSPLIT != 1) {   // this range's columns of the plane
            const int wc = tb - ta;
            for (int i = tid; i < S * wc; i += kConvThreads) {
                const int e = (i / wc) * T + ta + i % wc;
                if (MODE == 0) a.out_plane[(size_t)n * (S * T) + e] = obuf[e];
                else a.out_complex[((size_t)frame * (S * T) + e) * 2 + part] = obuf[e];
            }
        } else if (MODE == 0) {
            f32x4 *dstp = reinterpret_cast<f32x4 *>(a.out_plane + (size_t)n * (S * T));
            const f32x4 *src4 = reinterpret_cast<const f32x4 *>(obuf);
            for (int i = tid; i < S * T / 4; i += kConvThreads) dstp[i] = src4[i];
        } else {   // interleave this plane into the complex64 output (the frame's other plane is another workgroup's)
            float *dstp = a.out_complex + (size_t)frame * (S * T) * 2 + part;
            for (int i = tid; i < S * T; i += kConvThreads) dstp[2 * i] = obuf[i];
        }
    }
    SSTAMP(6);
#undef SSTAMP
}

// The fragment image of ONE ConvEnhancer for the training instantiation (the forward's prologue launch packs the inference images):
// [22 x 64 x 4 operand quads | 2 x 80 helper tables] from conv_block.{0,2,4,6} as the caller holds them.  cb == NULL (the backward's
// data-gradient pass runs the stack on transposed, flipped weights WITHOUT biases): the bias entries are zero.
struct FragPackArgs { const float *cw[4], *cb[4]; float *dst; };
__global__ __launch_bounds__(256) void conv_frag_pack_kernel(const FragPackArgs q) {
    const int v = blockIdx.x * 256 + threadIdx.x;      // one 16-byte piece of the image
    constexpr int kPieces = kFragFloats / 4;
    if (v >= kPieces) return;
    const bool nb = q.cb[0] == nullptr;
    f32x4 o;
    if (v < kFragQuads * 64) {
        const int quad = v >> 6, lane = v & 63;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int e = 4 * quad + jj;
            o[jj] = (nb && e >= 76) ? 0.f : conv_frag16_entry(q.cw[1], nb ? q.cw[1] : q.cb[1], q.cw[2], nb ? q.cw[2] : q.cb[2], e, lane);
        }
    } else {
        const int i0 = 4 * (v - kFragQuads * 64), h = i0 / 80;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int i = i0 - 80 * h + jj;
            const bool is_bias = (i >= 36 && i < 40) || i >= 76;
            o[jj] = (nb && is_bias) ? 0.f : conv_helper_entry(q.cw[0], nb ? q.cw[0] : q.cb[0], q.cw[3], nb ? q.cw[3] : q.cb[3], h, i);
        }
    }
    *reinterpret_cast<f32x4 *>(q.dst + (size_t)v * 4) = o;
}
hipError_t launch_conv_frag_pack(const float *const cw[4], const float *const cb[4], float *dst, hipStream_t st) {
    FragPackArgs q{};
    for (int i = 0; i < 4; ++i) { q.cw[i] = cw[i]; q.cb[i] = cb ? cb[i] : nullptr; }
    q.dst = dst;
    hipLaunchKernelGGL(conv_frag_pack_kernel, dim3((kFragFloats / 4 + 255) / 256), dim3(256), 0, st, q);
    return hipGetLastError();
}

bool conv_stream_ok(const ConvArgs &a) {
    if (a.S != S || a.T != T) return false;
    if (a.mode == 0 || a.mode == 2) return a.in_plane != nullptr && (reinterpret_cast<uintptr_t>(a.out_plane) & 15) == 0;
    if (a.mode == 1) return a.lin2_out != nullptr && a.resid != nullptr && T % a.p1 == 0 && S % a.p0 == 0;
    return false;
}

hipError_t launch_conv_stream(ConvArgs &a, int planes, hipStream_t st) {
    if (!conv_stream_ok(a)) return hipErrorNotSupported;
    static PerDeviceOnce lds_head, lds_tail, lds_train, lds_train16[3];
    if (a.mode == 2) {   // training path: plain plane in / out, stage tensors saved, masked activation
        // column ranges when the planes would leave half (three quarters) of the CUs idle: 64 frames -- the reference's default batch --
        // are 128 planes (AFT_CONV_NSPLIT=1|2|4 forces a split: tests, A/B)
        const int cus = current_device_cus();
        a.ranges = 4 * planes <= cus ? 4 : (2 * planes <= cus ? 2 : 1);
        if (switch_on("AFT_CONV_NSPLIT")) { const int f = switch_int("AFT_CONV_NSPLIT", 1); a.ranges = f == 4 ? 4 : (f == 2 ? 2 : 1); }
        if (a.wfrag != nullptr && !switch_on("AFT_CONV_MFMA32")) {
            // round 6: the 16x16x4 formulation (conv_stream16_kernel<0, NSPLIT, true>) on the fragment image the caller's scratch holds
            // (launch_conv_train packs it in front of this launch); AFT_CONV_MFMA32=1 keeps the 32x32x2 kernel (A/B, tests)
            a.plane0 = 0;
            const dim3 grid(planes * a.ranges), block(kConvThreads);
            hipError_t e16;
            if (a.ranges == 4) {
                e16 = ensure_dynamic_lds(lds_train16[2], reinterpret_cast<const void *>(conv_stream16_kernel<0, 4, true>), kStreamLds);
                if (e16 == hipSuccess) hipLaunchKernelGGL((conv_stream16_kernel<0, 4, true>), grid, block, kStreamLds, st, a);
            } else if (a.ranges == 2) {
                e16 = ensure_dynamic_lds(lds_train16[1], reinterpret_cast<const void *>(conv_stream16_kernel<0, 2, true>), kStreamLds);
                if (e16 == hipSuccess) hipLaunchKernelGGL((conv_stream16_kernel<0, 2, true>), grid, block, kStreamLds, st, a);
            } else {
                e16 = ensure_dynamic_lds(lds_train16[0], reinterpret_cast<const void *>(conv_stream16_kernel<0, 1, true>), kStreamLds);
                if (e16 == hipSuccess) hipLaunchKernelGGL((conv_stream16_kernel<0, 1, true>), grid, block, kStreamLds, st, a);
            }
            return e16 != hipSuccess ? e16 : hipGetLastError();
        }
        hipError_t et = ensure_dynamic_lds(lds_train, reinterpret_cast<const void *>(conv_stream_kernel<0, true>), kStreamLds);
        if (et != hipSuccess) return et;
        hipLaunchKernelGGL((conv_stream_kernel<0, true>), dim3(planes * a.ranges), dim3(kConvThreads), kStreamLds, st, a);
        return hipGetLastError();
    }
    // the 16x16x4 kernel needs the fragment image of the forward's prologue launch; AFT_CONV_MFMA32=1 keeps the 32x32x2 kernel (A/B runs)
    const bool m16 = a.wfrag != nullptr && !switch_on("AFT_CONV_MFMA32");
    hipError_t e;
    // column ranges when the planes would leave half (three quarters) of the CUs idle; AFT_CONV_NSPLIT=1|2|4 forces a split (tests, A/B)
    int nsplit = 1;
    if (m16) {
        const int cus = current_device_cus();
        nsplit = 4 * planes <= cus ? 4 : (2 * planes <= cus ? 2 : 1);
        if (switch_on("AFT_CONV_NSPLIT")) { const int f = switch_int("AFT_CONV_NSPLIT", 1); nsplit = f == 4 ? 4 : (f == 2 ? 2 : 1); }
    }
    const void *fn16 = a.mode == 0 ? (nsplit == 4 ? reinterpret_cast<const void *>(conv_stream16_kernel<0, 4>)
                                                  : nsplit == 2 ? reinterpret_cast<const void *>(conv_stream16_kernel<0, 2>)
                                                                : reinterpret_cast<const void *>(conv_stream16_kernel<0, 1>))
                                   : (nsplit == 4 ? reinterpret_cast<const void *>(conv_stream16_kernel<1, 4>)
                                                  : nsplit == 2 ? reinterpret_cast<const void *>(conv_stream16_kernel<1, 2>)
                                                                : reinterpret_cast<const void *>(conv_stream16_kernel<1, 1>));
    static PerDeviceOnce lds16[6];
    if (m16)
        e = ensure_dynamic_lds(lds16[(a.mode == 0 ? 0 : 3) + (nsplit == 4 ? 2 : nsplit == 2 ? 1 : 0)], fn16, kStreamLds);
    else
        e = a.mode == 0 ? ensure_dynamic_lds(lds_head, reinterpret_cast<const void *>(conv_stream_kernel<0>), kStreamLds)
                        : ensure_dynamic_lds(lds_tail, reinterpret_cast<const void *>(conv_stream_kernel<1>), kStreamLds);
    if (e != hipSuccess) return e;
#ifdef AFT_DIAG_STAMPS
    static unsigned long long *dbuf = nullptr;
    const bool stamp = switch_on("AFT_STAMPS") && planes <= 4096;
    if (stamp) {
        if (!dbuf) (void)hipMalloc(&dbuf, sizeof(unsigned long long) * 16 * 4096);
        (void)hipMemset(dbuf, 0, sizeof(unsigned long long) * 16 * 4096);
        a.stamps = dbuf;
    }
#endif
    if (m16) {
        auto go = [&](int ns, int first, int count) {
            a.plane0 = first;
            const dim3 grid(count * ns), block(kConvThreads);
            if (a.mode == 0) {
                if (ns == 4) hipLaunchKernelGGL((conv_stream16_kernel<0, 4>), grid, block, kStreamLds, st, a);
                else if (ns == 2) hipLaunchKernelGGL((conv_stream16_kernel<0, 2>), grid, block, kStreamLds, st, a);
                else hipLaunchKernelGGL((conv_stream16_kernel<0, 1>), grid, block, kStreamLds, st, a);
            } else {
                if (ns == 4) hipLaunchKernelGGL((conv_stream16_kernel<1, 4>), grid, block, kStreamLds, st, a);
                else if (ns == 2) hipLaunchKernelGGL((conv_stream16_kernel<1, 2>), grid, block, kStreamLds, st, a);
                else hipLaunchKernelGGL((conv_stream16_kernel<1, 1>), grid, block, kStreamLds, st, a);
            }
        };
        // One plane per CU (the LDS): planes beyond a whole number of rounds would cost a full round for a few workgroups (258 planes
        // on 256 CUs: the 128 -> 129 frames step).  When that remainder fits the chip as column ranges it is a second launch of
        // ranges -- 0.42 (four ranges) or 0.62 (two) of a round instead of 1.0.  Same bits (ranges reproduce whole planes).
        const int cus = current_device_cus();
        const int rem = planes % cus, whole = planes - rem;
        const bool forced = switch_on("AFT_CONV_NSPLIT");
        if (!forced && nsplit == 1 && whole > 0 && rem > 0 && 2 * rem <= cus) {
            const int ns2 = 4 * rem <= cus ? 4 : 2;
            hipError_t ea = ensure_dynamic_lds(lds16[(a.mode == 0 ? 0 : 3) + (ns2 == 4 ? 2 : 1)],
                                               a.mode == 0 ? (ns2 == 4 ? reinterpret_cast<const void *>(conv_stream16_kernel<0, 4>) : reinterpret_cast<const void *>(conv_stream16_kernel<0, 2>))
                                                           : (ns2 == 4 ? reinterpret_cast<const void *>(conv_stream16_kernel<1, 4>) : reinterpret_cast<const void *>(conv_stream16_kernel<1, 2>)),
                                               kStreamLds);
            if (ea != hipSuccess) return ea;
            go(1, 0, whole);
            go(ns2, whole, rem);
        } else {
            go(nsplit, 0, planes);
        }
        a.plane0 = 0;
    } else if (a.mode == 0) hipLaunchKernelGGL((conv_stream_kernel<0>), dim3(planes), dim3(kConvThreads), kStreamLds, st, a);
    else hipLaunchKernelGGL((conv_stream_kernel<1>), dim3(planes), dim3(kConvThreads), kStreamLds, st, a);
#ifdef AFT_DIAG_STAMPS
    if (stamp) {
        (void)hipDeviceSynchronize();
        static int printed = 0;
        if (printed++ < 4) {
            std::vector<unsigned long long> hb(16 * (size_t)planes);
            (void)hipMemcpy(hb.data(), dbuf, hb.size() * 8, hipMemcpyDeviceToHost);
            double m[7] = {0}, hsum[7] = {0};
            for (int b = 0; b < planes; ++b)
                for (int i = 1; i < 7; ++i) {
                    m[i] += (double)(hb[(size_t)b * 16 + i] - hb[(size_t)b * 16 + i - 1]);
                    hsum[i] += (double)(hb[(size_t)b * 16 + 8 + i] - hb[(size_t)b * 16 + 8 + i - 1]);
                }
            double in_lds = 0;
            for (int b = 0; b < planes; ++b) in_lds += (double)(hb[(size_t)b * 16 + 15] - hb[(size_t)b * 16 + 8]);
            if (m16) printf("conv stream16: input plane in LDS %.0f cycles after the kernel's start (helper wave)\n", in_lds / planes);
            printf("conv stream mode %d (mean cycles): matrix wave: stage=%.0f gather=%.0f wait=%.0f sweeps=%.0f wait=%.0f store=%.0f | "
                   "helper wave: stage=%.0f zero+input+conv1=%.0f wait=%.0f conv4=%.0f wait=%.0f store=%.0f\n", a.mode, m[1] / planes,
                   m[2] / planes, m[3] / planes, m[4] / planes, m[5] / planes, m[6] / planes, hsum[1] / planes, hsum[2] / planes,
                   hsum[3] / planes, hsum[4] / planes, hsum[5] / planes, hsum[6] / planes);
        }
        a.stamps = nullptr;
    }
#endif
    return hipGetLastError();
}

}  // namespace aft
