// k_attn.hip -- launcher of the attention kernel (device code: attn_device.h).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "attn_device.h"
#include "attn16_device.h"

namespace aft {

// The task of the partial last round this wave takes (-1: none), see attn_kernel.  vblock = the XCD-contiguous workgroup index.
__device__ __forceinline__ int partial_round_task(int vblock, int wave, int ntasks) {
    const int total_waves = gridDim.x * 4, nfull = ntasks / total_waves * total_waves, rem = ntasks - nfull;
    if (rem <= 0) return -1;
    if ((gridDim.x & 7) == 0) {
        const int per = gridDim.x >> 3, xcd = vblock / per, jj = (vblock - xcd * per) * 4 + wave;
        const int s0 = (xcd * rem) >> 3, s1 = ((xcd + 1) * rem) >> 3;
        return jj < s1 - s0 ? nfull + s0 + jj : -1;
    }
    return vblock * 4 + wave < rem ? nfull + vblock * 4 + wave : -1;
}

// launch bound (256, 3): <= 168 registers keeps accumulators in VGPRs, see attn_device.h
// HD = 32: three waves per SIMD (the tuned shape).  HD = 64 holds two blocks of q / k / v^T and two accumulators (~210 VGPRs): two; so do
// the head dims whose heads straddle two blocks (24, 40, 48).
constexpr int attn_waves_per_simd(int hd) { return hd == 64 || hd == 24 || hd == 40 || hd == 48 ? 2 : AFT_ATTN_WAVES; }
template <int HD, int TOK = 0>
__global__ __launch_bounds__(256, attn_waves_per_simd(HD)) void attn_kernel(const float *__restrict__ q, const float *__restrict__ k,
                                                      const float *__restrict__ vt, const float *__restrict__ qbias,
                                                      float *__restrict__ out, int nblk, int tokens, int tokpad,
                                                      int model_dim,
                                                      float scale_log2e, int ntasks, unsigned long long *stamps) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // XCD-aware task order: workgroups are dealt round-robin over the 8 XCDs, so workgroup b and b+8 share an
    // L2.  Give each XCD a CONTIGUOUS range of tasks: the 9 query tiles of one (plane, head) then read their K / V^T
    // (72 KB) through ONE L2 instead of two or three.  A pure speed choice; any mapping is correct.
    int vblock = blockIdx.x;
    if ((gridDim.x & 7) == 0) vblock = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    // A partial last round (tasks beyond the last whole multiple of the wave count; none at the benchmark's batch: 9 216 tasks = 3 x
    // 3 072 waves) is NOT left to the stride: with the XCD-contiguous order above its tasks would all land on the first XCDs -- at 64
    // frames (4 608 tasks) XCDs 0-3 ran a second full 29-us round with three waves per SIMD while XCDs 4-7 idled: 59.6 us for 1.5
    // rounds of work (profiles/r05_b64_kernel_trace_summary.txt).  Each XCD takes an eighth of the partial round (still a contiguous
    // task range per XCD) on its earliest-dispatched workgroups, i.e. spread over its CUs.  Same arithmetic per task: same bits.
    const int total_waves = gridDim.x * 4, nfull = ntasks / total_waves * total_waves;
    attn_body<false, HD, TOK>(q, k, vt, qbias, out, nblk, tokens, tokpad, model_dim, scale_log2e, vblock * 4 + wave, total_waves, nfull,
                              stamps, partial_round_task(vblock, wave, ntasks));
}

// head dimension 16 on 16x16x4 MFMAs (attn16_device.h, round 6): the value product takes 16-row operands -- a head -- instead of a
// whole 32-feature block; same task order and hand-out as attn_kernel
template <int TOK = 0, int HD = 16>
__global__ __launch_bounds__(256, AFT_ATTN_WAVES) void attn16_kernel(const float *__restrict__ q, const float *__restrict__ k,
                                                                      const float *__restrict__ vt, const float *__restrict__ qbias,
                                                                      float *__restrict__ out, int nblk, int tokens, int tokpad, int model_dim,
                                                                      float scale_log2e, int ntasks) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int vblock = blockIdx.x;
    if ((gridDim.x & 7) == 0) vblock = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int total_waves = gridDim.x * 4, nfull = ntasks / total_waves * total_waves;
    attn16_body<TOK, HD>(q, k, vt, qbias, out, nblk, tokens, tokpad, model_dim, scale_log2e, vblock * 4 + wave, total_waves, nfull,
                         partial_round_task(vblock, wave, ntasks));
}

// split-precision tier: K / Q^T / V^T arrive as bf16 hi / lo fragments from chain_split_kernel (attn_device.h)
__global__ __launch_bounds__(256, AFT_ATTN_WAVES) void attn_split_kernel(const float *__restrict__ q, const float *__restrict__ k,
                                                      const float *__restrict__ vt, float *__restrict__ out, int heads, int tokens,
                                                      int tokpad, int model_dim, int ntasks) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int vblock = blockIdx.x;
    if ((gridDim.x & 7) == 0) vblock = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int total_waves = gridDim.x * 4, nfull = ntasks / total_waves * total_waves;
    attn_body<true>(q, k, vt, nullptr, out, heads, tokens, tokpad, model_dim, 0.f, vblock * 4 + wave, total_waves, nfull, nullptr,
                    partial_round_task(vblock, wave, ntasks));
}

hipError_t launch_attention(const aft_config &c, const float *q, const float *k, const float *vt, const float *qbias,
                            float *attn, int planes, int tokens, int tokpad, hipStream_t st) {
    const int hd = c.model_dim / c.num_head, nblk = c.model_dim / kHeadDim;   // head dimension: a multiple of 8 up to 64, not 56 (check_config); 32-feature blocks
    const int ntasks = planes * c.num_head * (tokpad / kTile);
    const int resident_blocks = attn_waves_per_simd(hd) * current_device_cus();   // CUs x workgroups (launch bound: waves / SIMD)
    const int blocks = std::min((ntasks + 3) / 4, resident_blocks);
    const float scale_log2e = 1.4426950408889634f / sqrtf((float)hd);
    if (c.precision == AFT_PRECISION_BF16X3) {
        hipLaunchKernelGGL(attn_split_kernel, dim3(blocks), dim3(256), 0, st, q, k, vt, attn, nblk, tokens, tokpad, c.model_dim,
                           ntasks);
        return hipGetLastError();
    }
#ifdef AFT_DIAG_STAMPS
    if (switch_on("AFT_STAMPS")) {   // per-task stamps + in-kernel clock (diagnostic build only)
        static unsigned long long *dbuf = nullptr;
        if (!dbuf) (void)hipMalloc(&dbuf, sizeof(unsigned long long) * 8 * 16384);
        (void)hipMemset(dbuf, 0, sizeof(unsigned long long) * 8 * 16384);
        hipLaunchKernelGGL((attn_kernel<32>), dim3(blocks), dim3(256), 0, st, q, k, vt, qbias, attn, nblk, tokens,
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
            printf("attn stamps: mean per task: prologue=%.0f loop=%.0f store=%.0f (cycles); MFMA ideal %d\n",
                   sum[1] / n, sum[2] / n, sum[4] / n, 32 * (tokpad / kTile) * 64);
            // schedule: when do the tasks of each persistent round start / end (memrealtime = 100 MHz ticks)
            unsigned long long t0 = ~0ull, t1 = 0;
            for (int t = 0; t < n; ++t) { t0 = std::min(t0, hbuf[t * 8 + 6]); t1 = std::max(t1, hbuf[t * 8 + 5]); }
            const int per_round = blocks * 4;
            double busy = 0;
            for (int t = 0; t < n; ++t) busy += (double)(hbuf[t * 8 + 5] - hbuf[t * 8 + 6]);
            printf("  span %.1f us, %d tasks in rounds of %d; mean concurrent waves per SIMD %.2f\n", (t1 - t0) / 100.0, n, per_round,
                   busy / (double)(t1 - t0) / 1024.0);
            for (int r0 = 0; r0 < n; r0 += per_round) {
                double smin = 1e30, smax = 0, ssum = 0, emin = 1e30, emax = 0, esum = 0;
                const int r1 = std::min(n, r0 + per_round);
                for (int t = r0; t < r1; ++t) {
                    const double st = (hbuf[t * 8 + 6] - t0) / 100.0, en = (hbuf[t * 8 + 5] - t0) / 100.0;
                    smin = std::min(smin, st); smax = std::max(smax, st); ssum += st;
                    emin = std::min(emin, en); emax = std::max(emax, en); esum += en;
                }
                printf("  round %d: start %.1f / %.1f / %.1f us (min/mean/max), end %.1f / %.1f / %.1f us\n", r0 / per_round, smin,
                       ssum / (r1 - r0), smax, emin, esum / (r1 - r0), emax);
            }
        }
        return hipGetLastError();
    }
#endif
    unsigned long long *no_stamps = nullptr;
#define AFT_ATTN_HD(HD_)                                                                                                              \
    if (hd == HD_) {                                                                                                                  \
        hipLaunchKernelGGL((attn_kernel<HD_>), dim3(blocks), dim3(256), 0, st, q, k, vt, qbias, attn, nblk, tokens, tokpad, c.model_dim, \
                           scale_log2e, ntasks, no_stamps);                                                                           \
        return hipGetLastError();                                                                                                     \
    }
    if (hd == 8 && !switch_on("AFT_ATTN_HD8_MFMA32")) {       // late round 6: 16x16x4 MFMAs with half of every operand idle (attn16_device.h)
        if (tokens == 280 && !switch_on("AFT_ATTN_GENERIC"))
            hipLaunchKernelGGL((attn16_kernel<280, 8>), dim3(blocks), dim3(256), 0, st, q, k, vt, qbias, attn, nblk, tokens, tokpad, c.model_dim,
                               scale_log2e, ntasks);
        else
            hipLaunchKernelGGL((attn16_kernel<0, 8>), dim3(blocks), dim3(256), 0, st, q, k, vt, qbias, attn, nblk, tokens, tokpad, c.model_dim,
                               scale_log2e, ntasks);
        return hipGetLastError();
    }
    AFT_ATTN_HD(8) AFT_ATTN_HD(24) AFT_ATTN_HD(40) AFT_ATTN_HD(48)   // heads that start anywhere in a block (attn_device.h); covered, not tuned
#undef AFT_ATTN_HD
    if (hd != 16 && hd != 32 && hd != 64) return hipErrorInvalidValue;   // check_config refuses these
    if (hd == 16 && !switch_on("AFT_ATTN_HD16_MFMA32")) {     // 16x16x4 MFMAs (attn16_device.h); the switch keeps round 5's 32x32x2 form (A/B)
        if (tokens == 280 && !switch_on("AFT_ATTN_GENERIC"))
            hipLaunchKernelGGL((attn16_kernel<280>), dim3(blocks), dim3(256), 0, st, q, k, vt, qbias, attn, nblk, tokens, tokpad, c.model_dim,
                               scale_log2e, ntasks);
        else
            hipLaunchKernelGGL((attn16_kernel<0>), dim3(blocks), dim3(256), 0, st, q, k, vt, qbias, attn, nblk, tokens, tokpad, c.model_dim,
                               scale_log2e, ntasks);
    } else if (hd == 16)
        hipLaunchKernelGGL((attn_kernel<16>), dim3(blocks), dim3(256), 0, st, q, k, vt, qbias, attn, nblk, tokens, tokpad, c.model_dim,
                           scale_log2e, ntasks, no_stamps);
    else if (hd == 64)
        hipLaunchKernelGGL((attn_kernel<64>), dim3(blocks), dim3(256), 0, st, q, k, vt, qbias, attn, nblk, tokens, tokpad, c.model_dim,
                           scale_log2e, ntasks, no_stamps);
    // the two benchmark grids with the token count at compile time (120 x 14 -> 280 tokens, 240 x 28 -> 1120): the tile loop's trip
    // count, the last-tiles logic and the padding masks resolve in the compiler -- 125 VGPRs and no scalar spills instead of 168 and 40,
    // -2.7 % time at 280 tokens (A/B knob: AFT_ATTN_GENERIC=1 runs the generic instantiation; same bits)
    else if (tokens == 280 && !switch_on("AFT_ATTN_GENERIC"))
        hipLaunchKernelGGL((attn_kernel<32, 280>), dim3(blocks), dim3(256), 0, st, q, k, vt, qbias, attn, nblk, tokens, tokpad, c.model_dim,
                           scale_log2e, ntasks, no_stamps);
    else if (tokens == 1120 && !switch_on("AFT_ATTN_GENERIC"))
        hipLaunchKernelGGL((attn_kernel<32, 1120>), dim3(blocks), dim3(256), 0, st, q, k, vt, qbias, attn, nblk, tokens, tokpad, c.model_dim,
                           scale_log2e, ntasks, no_stamps);
    else
        hipLaunchKernelGGL((attn_kernel<32>), dim3(blocks), dim3(256), 0, st, q, k, vt, qbias, attn, nblk, tokens, tokpad, c.model_dim,
                           scale_log2e, ntasks, no_stamps);
    return hipGetLastError();
}

}  // namespace aft
