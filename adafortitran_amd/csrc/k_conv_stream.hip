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

#include "conv_device.h"

namespace aft {

namespace {

constexpr int S = 120, T = 14, SP = 128, LR = 128;   // LR = S + 8: local row lr <-> plane row lr - 4
constexpr int kPlane = (T + 2) * SP;                 // floats per channel plane [T + 2][SP]: LDS column = symbol + 1
constexpr int kIn0 = 0, kC1 = kPlane, kC3 = 9 * kPlane, kArena = 17 * kPlane;
constexpr int kStage = kArena - kWStage;             // conv2 / conv3 weights staged in the END of c3 (dead until the second barrier)
constexpr int kBias2 = kArena, kW1 = kBias2 + 32, kW4 = kW1 + 80, kFlags = kW4 + 80, kStreamFloats = kFlags + 16;
constexpr size_t kStreamLds = sizeof(float) * kStreamFloats;   // 140 352 B: one workgroup per CU, as before

__device__ __forceinline__ float other_half32(float x) {   // value held by lane (l ^ 32): v_permlane32_swap
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, (threadIdx.x & 32) ? r[0] : r[1]);
}

}  // namespace

// MODE 0 = head (a.in_plane = upsampled planes [planes][S][T], a.out_plane), 1 = tail (a.lin2_out + a.resid, a.out_complex)
template <int MODE>
__global__ __launch_bounds__(kConvThreads) void conv_stream_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *in0 = smem + kIn0, *c1 = smem + kC1, *c3 = smem + kC3;
    float *bias2 = smem + kBias2, *w1s = smem + kW1, *w4s = smem + kW4;
    volatile int *flags = reinterpret_cast<volatile int *>(smem + kFlags);   // [0..3]: conv3 columns published by matrix wave w
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const int n = blockIdx.x, frame = n >> 1, part = n & 1;
    const bool matrix = wave < 4;

    // ---- phase 0 (all waves): stage the conv2 / conv3 weights TRANSPOSED in the end of c3 (as k_conv.hip does in the arena:
    //      conflict-free gathers), the small tables, the flags ----
    {
        float *stage = smem + kStage;
        for (int i = tid; i < 2304; i += kConvThreads) {
            stage[(i % 72) * 33 + i / 72] = a.cw[1][i];
            const int rem = i % 288;   // conv3.weight [co 8][ci 32][ky 3][kx 3]
            stage[kW3Off + (rem / 3) * 33 + (rem % 3) * 8 + i / 288] = a.cw[2][i];
        }
        if (tid < 32) bias2[tid] = a.cb[1][(tid & 3) + 8 * ((tid & 15) >> 2) + 4 * (tid >> 4)];
        if (tid >= 64 && tid < 144) w1s[tid - 64] = tid < 136 ? a.cw[0][tid - 64] : a.cb[0][tid - 136];
        if (tid >= 192 && tid < 265) w4s[tid - 192] = tid < 264 ? a.cw[3][tid - 192] : a.cb[3][0];
        if (tid >= 320 && tid < 336) flags[tid - 320] = 0;
    }
    __syncthreads();

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
        for (int e = 0; e < 4; ++e) bias3[e] = a.cb[2][e + 4 * h];
    } else {
        // ---- helper waves: borders, input plane, conv1 of every column ----
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
        // input plane (one pixel per thread and pass, global reads coalesced)
        for (int i = ht; i < S * T; i += 256) {
            const int gr = i / T, t = i - gr * T;
            float v;
            if (MODE == 0) {
                v = a.in_plane[(size_t)n * (S * T) + i];
            } else {   // inverse patch map + conv_enhanced residual: feature f of token (g, tc) is pixel (g*p0 + f/p1, tc*p1 + f%p1)
                const int p0 = a.p0, p1 = a.p1, tpr = T / p1;
                const int g = gr / p0, tc = t / p1, f = (gr - g * p0) * p1 + (t - tc * p1);
                v = a.lin2_out[((size_t)n * a.tokens + g * tpr + tc) * a.lin2_stride + f] + a.resid[(size_t)n * (S * T) + i];
            }
            in0[(t + 1) * SP + gr + 4] = v;
        }
        // the four helper waves only: a named barrier would do; LDS ops of a workgroup are visible to it once complete, and the
        // conv1 reads below need ALL helpers' input writes -> cross-wave hand-over through a counter flag
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_fetch_add(const_cast<int *>(flags + 4), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        while (flags[4] < 4) __builtin_amdgcn_s_sleep(2);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        // conv1: 1 -> 8, ReLU.  Thread = (local row lr, channel half): 4 channels x 9 taps per column, window slides over the columns
        {
            const int lr = 32 * (wave - 4) + j, gr = lr - 4;
            const bool ok = lr >= 1 && lr < LR - 1 && gr >= 0 && gr < S;
            float w[4][9], b[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
#pragma unroll
                for (int k9 = 0; k9 < 9; ++k9) w[k][k9] = w1s[(4 * h + k) * 9 + k9];
                b[k] = w1s[72 + 4 * h + k];
            }
            const int r0 = max(lr - 1, 0), r2 = min(lr + 1, LR - 1);
            float win[3][3];   // [ky][kx]
#pragma unroll
            for (int kx = 1; kx < 3; ++kx) {
                win[0][kx] = in0[(kx - 1) * SP + r0];
                win[1][kx] = in0[(kx - 1) * SP + lr];
                win[2][kx] = in0[(kx - 1) * SP + r2];
            }
            float *dst = c1 + (4 * h) * kPlane + SP + lr;
#pragma unroll 2
            for (int t = 0; t < T; ++t) {
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) { win[ky][0] = win[ky][1]; win[ky][1] = win[ky][2]; }
                win[0][2] = in0[(t + 2) * SP + r0];
                win[1][2] = in0[(t + 2) * SP + lr];
                win[2][2] = in0[(t + 2) * SP + r2];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float acc = b[k];
#pragma unroll
                    for (int k9 = 0; k9 < 9; ++k9) acc = fmaf(win[k9 / 3][k9 % 3], w[k][k9], acc);
                    dst[k * kPlane + t * SP] = ok ? fmaxf(acc, 0.f) : 0.f;
                }
            }
        }
    }
    __syncthreads();   // fragments gathered (the staging area is dead), c1 complete (in0 is dead: it becomes the output plane)

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
        auto publish = [&](int columns_done) {   // conv3 output columns [0, columns_done) of this tile are in LDS
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) flags[wave] = columns_done;
        };
        auto store_col = [&](int tout, float v0, float v1, float v2, float v3) {
            if (tout < 0 || tout >= T) return;
            if (ok3) {
                float *p = dst + (tout + 1) * SP;
                p[0] = fmaxf(v0 + bias3[0], 0.f);
                p[kPlane] = fmaxf(v1 + bias3[1], 0.f);
                p[2 * kPlane] = fmaxf(v2 + bias3[2], 0.f);
                p[3 * kPlane] = fmaxf(v3 + bias3[3], 0.f);
            }
            publish(tout + 1);
        };
        auto b_at = [&](int kb, int tcol) {
            const int tap = kb >> 2, kx = tap / 3, ky = tap % 3;
            return bsrc[(kb & 3) * kPlane + (tcol + kx) * SP + ky];
        };
        float b[36];
#pragma unroll
        for (int kb = 0; kb < 36; ++kb) b[kb] = b_at(kb, 0);
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
        float x2[16];
        {   // prologue: conv2 of column 0
            f32x16 acc2 = bias2_acc();
#pragma unroll
            for (int kb = 0; kb < 36; ++kb) {
                acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa2[kb], b[kb], acc2, 0, 0, 0);
                b[kb] = b_at(kb, 1);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) x2[e] = __builtin_amdgcn_fmed3f(acc2[e], 0.f, relu_hi);
        }
        auto conv3_step = [&](int i) {   // ky = centre (16..31), below (0..15), above (32..47)
            const int e = i & 15;
            const float xv = i < 16 ? x2[e] : (i < 32 ? lane_from_below(x2[e]) : lane_from_above(x2[e]));
            const int wi = i < 16 ? 16 + e : (i < 32 ? e : 32 + e);
            acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa3[wi], xv, acc3, 0, 0, 0);
        };
        auto rotate = [&]() {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc3[8 + e] = acc3[4 + e];
                acc3[4 + e] = acc3[e];
                acc3[e] = 0.f;
            }
        };
#pragma unroll 1
        for (int tcol = 0; tcol < T - 1; ++tcol) {
            store_col(tcol - 2, acc3[8], acc3[9], acc3[10], acc3[11]);   // complete since the previous column's MFMAs
            rotate();
            f32x16 acc2 = bias2_acc();
            const int tnext = min(tcol + 2, T - 1);
            // 48 conv3 MFMAs of column tcol interleaved with the 36 conv2 MFMAs of column tcol + 1 (4 : 3)
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
#pragma unroll
            for (int e = 0; e < 16; ++e) x2[e] = __builtin_amdgcn_fmed3f(acc2[e], 0.f, relu_hi);
        }
        store_col(T - 3, acc3[8], acc3[9], acc3[10], acc3[11]);
        rotate();
#pragma unroll
        for (int i = 0; i < 48; ++i) conv3_step(i);      // conv3 of the last column
        store_col(T - 2, acc3[8], acc3[9], acc3[10], acc3[11]);
        store_col(T - 1, acc3[4], acc3[5], acc3[6], acc3[7]);   // (registers 0..3 would be column T: zero padding)
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
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_fetch_add(const_cast<int *>(flags + 5), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        while (flags[5] < 4) __builtin_amdgcn_s_sleep(2);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
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
            for (;;) {
                const int m = min(min(flags[0], flags[1]), min(flags[2], flags[3]));
                if (m >= columns) { have = m; break; }
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
        // window before the loop: kx = 1 <- LDS column 0 (zero border), kx = 2 <- LDS column 1 (symbol 0)
        need(1);
        load_col(0);
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) win[c][ky][1] = win[c][ky][2];
        load_col(1);
#pragma unroll 1
        for (int t = 0; t < T; ++t) {
            need(min(t + 2, T));             // symbol t + 1 (LDS column t + 2; the last one is the zero border)
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
    __syncthreads();
    // ---- the output plane leaves in one coalesced pass ----
    {
        const float *obuf = in0;
        if (MODE == 0) {
            f32x4 *dst = reinterpret_cast<f32x4 *>(a.out_plane + (size_t)n * (S * T));
            const f32x4 *src4 = reinterpret_cast<const f32x4 *>(obuf);
            for (int i = tid; i < S * T / 4; i += kConvThreads) dst[i] = src4[i];
        } else {   // interleave this plane into the complex64 output (the frame's other plane is another workgroup's)
            float *dst = a.out_complex + (size_t)frame * (S * T) * 2 + part;
            for (int i = tid; i < S * T; i += kConvThreads) dst[2 * i] = obuf[i];
        }
    }
}

bool conv_stream_ok(const ConvArgs &a) {
    if (a.S != S || a.T != T) return false;
    if (a.mode == 0) return a.in_plane != nullptr && (reinterpret_cast<uintptr_t>(a.out_plane) & 15) == 0;
    if (a.mode == 1) return a.lin2_out != nullptr && a.resid != nullptr && T % a.p1 == 0 && S % a.p0 == 0;
    return false;
}

hipError_t launch_conv_stream(ConvArgs &a, int planes, hipStream_t st) {
    if (!conv_stream_ok(a)) return hipErrorNotSupported;
    static PerDeviceOnce lds_head, lds_tail;
    if (a.mode == 0) {
        hipError_t e = ensure_dynamic_lds(lds_head, reinterpret_cast<const void *>(conv_stream_kernel<0>), kStreamLds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((conv_stream_kernel<0>), dim3(planes), dim3(kConvThreads), kStreamLds, st, a);
    } else {
        hipError_t e = ensure_dynamic_lds(lds_tail, reinterpret_cast<const void *>(conv_stream_kernel<1>), kStreamLds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((conv_stream_kernel<1>), dim3(planes), dim3(kConvThreads), kStreamLds, st, a);
    }
    return hipGetLastError();
}

}  // namespace aft
