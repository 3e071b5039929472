// k_conv_train.hip -- weight and bias gradients of a ConvEnhancer.
//
// Reference semantics: four 3x3 cross-correlations with zero padding 1, channels 1->8->32->8->1, ReLU
// after the first three (reference src/models/blocks/enhancers.py:12-20); PyTorch's backward gives, for
// conv k with input a_{k-1}, pre-activation gradient g_k and weight [co][ci][ky][kx]:
//     dW_k[co][ci][ky][kx] = sum_{n,s,t} g_k[n][co][s][t] * a_{k-1}[n][ci][s+ky-1][t+kx-1]
//     db_k[co]             = sum_{n,s,t} g_k[n][co][s][t]
// The forward (k_conv.hip, training variant) saves a_1..a_3 and its backward run saves g_1..g_3, all as
// [plane][channel][symbol column][subcarrier row] (rows contiguous); a_0 (the stack's input) and g_4 (the
// incoming gradient) are PyTorch planes [plane][row][column].
//
// conv2 and conv3 hold 2 x 2304 of the 2 x 2376 weights and nearly all the FLOPs.  Both are the same
// product "32-channel tensor x 8-channel tensor shifted by the nine taps":
//     conv2: dW2[a][b][tap] = sum A32=g2[a][p] * B8=a1[b][p + (tap-1)]
//     conv3: dW3[b][a][tap] = sum A32=a2[a][p] * B8=g3[b][p - (tap-1)]
// computed on v_mfma_f32_32x32x2_f32 with K = pixels: M = the 32 channels, N = (tap, b) = 72 of 96.
// A wave takes chunks of 64 rows of one column of one plane, stages the A rows and the three B columns
// (one halo row each side) coalesced into wave-private LDS, reads fragments from there (A: row stride
// 65 floats, conflict-free), and keeps its 32 x 96 partial in registers across all its chunks; the four
// waves of a workgroup are summed through LDS into one slice, slices by the shared fixed-order reduction.
#include "aft_internal.h"

namespace aft {

constexpr int kWgA = 32 * 65, kWgBRow = 68, kWgB = 25 * kWgBRow;   // 24 (channel, column) rows + one zero row
constexpr int kWgWave = kWgA + kWgB;                                // floats of LDS per wave (3780 >= 48*64 for the final sum)
constexpr int kWgradSlices = 512;

struct Wgrad32x8Args {
    const float *A, *B;   // [planes][32][T][S], [planes][8][T][S]
    float *slices;        // [gridDim.x][2304]
    int planes, S, T, sign, sa, sb;   // out index = a * sa + b * sb + tap
};

__global__ __launch_bounds__(256) void wgrad32x8_kernel(const Wgrad32x8Args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 31, kh = lane >> 5;
    float *As = lds + wave * kWgWave, *Bs = As + kWgA;
    const int S = a.S, T = a.T, schunks = (S + 63) / 64;
    const int nchunks = a.planes * T * schunks;

    // per-lane B fragment row for the three N tiles: column jn = 32 nt + i -> (tap = jn >> 3, b = jn & 7)
    int boff[3];
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) {
        const int jn = 32 * nt + i, tap = jn >> 3, b = jn & 7, ky = tap / 3, kx = tap % 3;
        boff[nt] = tap < 9 ? (b * 3 + kx) * kWgBRow + 1 + a.sign * (ky - 1) : 24 * kWgBRow + 1;
    }
    for (int q = lane; q < kWgBRow; q += 64) Bs[24 * kWgBRow + q] = 0.f;   // the zero row (padding taps)

    f32x16 acc[3];
#pragma unroll
    for (int nt = 0; nt < 3; ++nt)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;

    // chunk c = (plane n, column t, 64-row block): the 32 A rows, the 24 B rows (8 channels x 3 columns, 64
    // values each) and their 2-value tails (lanes 0..47: row = lane / 2, one instruction for all 24) are
    // fetched into registers one chunk ahead, so their latency hides under the previous chunk's MFMAs
    float ra[32], rb[24], rt;
    auto fetch = [&](int c) {
        const int sc = c % schunks, t = (c / schunks) % T, n = c / (schunks * T), s0 = sc * 64;
        const float *ap = a.A + ((size_t)n * 32 * T + t) * S + s0;
#pragma unroll
        for (int ch = 0; ch < 32; ++ch) ra[ch] = s0 + lane < S ? ap[(size_t)ch * T * S + lane] : 0.f;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int tb = t + a.sign * (kx - 1);
            const bool col_ok = tb >= 0 && tb < T;
            const float *bp = a.B + ((size_t)n * 8 * T + (col_ok ? tb : 0)) * S;
            const int s = s0 - 1 + lane;
#pragma unroll
            for (int b = 0; b < 8; ++b) rb[b * 3 + kx] = col_ok && s >= 0 && s < S ? bp[(size_t)b * T * S + s] : 0.f;
        }
        {
            const int row = lane >> 1, b = row / 3, kx = row % 3, tb = t + a.sign * (kx - 1), s2 = s0 + 63 + (lane & 1);
            rt = lane < 48 && tb >= 0 && tb < T && s2 < S ? a.B[((size_t)(n * 8 + b) * T + tb) * S + s2] : 0.f;
        }
    };
    const int c0 = blockIdx.x * 4 + wave, cstep = gridDim.x * 4;
    if (c0 < nchunks) fetch(c0);
    for (int c = c0; c < nchunks; c += cstep) {
        // ---- stage registers -> LDS: A rows [32][64], B rows [24][66] ----
#pragma unroll
        for (int ch = 0; ch < 32; ++ch) As[ch * 65 + lane] = ra[ch];
#pragma unroll
        for (int row = 0; row < 24; ++row) Bs[row * kWgBRow + lane] = rb[row];
        if (lane < 48) Bs[(lane >> 1) * kWgBRow + 64 + (lane & 1)] = rt;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the staging writes of all lanes precede the fragment reads
        __builtin_amdgcn_wave_barrier();
        if (c + cstep < nchunks) fetch(c + cstep);
        // ---- K = 64 pixels: 32 k-pairs x 3 N tiles ----
        const float *af = As + i * 65 + kh;
#pragma unroll 8
        for (int kb = 0; kb < 32; ++kb) {
            const float av = af[2 * kb];
#pragma unroll
            for (int nt = 0; nt < 3; ++nt)
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, Bs[boff[nt] + 2 * kb + kh], acc[nt], 0, 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // ... and the reads precede the next chunk's writes
        __builtin_amdgcn_wave_barrier();
    }
    // ---- sum the four waves, write this workgroup's slice ----
    __syncthreads();
    float *mine = lds + wave * kWgWave;
#pragma unroll
    for (int nt = 0; nt < 3; ++nt)
#pragma unroll
        for (int e = 0; e < 16; ++e) mine[(nt * 16 + e) * 64 + lane] = acc[nt][e];
    __syncthreads();
    if (wave == 0) {
        float *out = a.slices + (size_t)blockIdx.x * 2304;
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) {
            const int jn = 32 * nt + i, tap = jn >> 3, b = jn & 7;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int idx = (nt * 16 + e) * 64 + lane;
                const float v = (lds[idx] + lds[kWgWave + idx]) + (lds[2 * kWgWave + idx] + lds[3 * kWgWave + idx]);
                const int ach = (e & 3) + 8 * (e >> 2) + 4 * kh;
                if (tap < 9) out[ach * a.sa + b * a.sb + tap] = v;
            }
        }
    }
}

// result[a][tap] = sum_p A8[a][p] * B1[p + sign * (tap - 1)],  A8 [planes][8][T][S], B1 [planes][S][T]
// One thread per pixel position (row fastest) of kWg81Planes consecutive planes: the 72 accumulators run over the
// planes in registers and meet in ONE wave reduction per workgroup (round 1 reduced after every single plane: 72 x 6
// cross-lane steps per pixel were the whole 40 us of this kernel); one slice per workgroup.
// Both products of a stack (conv1: g1 x input, shift +; conv4: a3 x dy, shift -) in ONE launch: blockIdx.z picks the product (round 6).
constexpr int kWg81Planes = 8;
struct Wgrad8x1Args {
    const float *A8[2], *B1[2];
    float *slices[2];
    int sign[2];
    int planes, S, T;
};
__global__ __launch_bounds__(256) void wgrad8x1_kernel(const Wgrad8x1Args a) {
    const float *__restrict__ A8 = a.A8[blockIdx.z], *__restrict__ B1 = a.B1[blockIdx.z];
    float *__restrict__ slices = a.slices[blockIdx.z];
    const int planes = a.planes, S = a.S, T = a.T, sign = a.sign[blockIdx.z];
    __shared__ float red[4 * 36 * 65];
    const int p = blockIdx.x * 256 + threadIdx.x;   // p = t * S + s
    const int n0 = blockIdx.y * kWg81Planes, n1 = min(planes, n0 + kWg81Planes);
    float acc[72];
#pragma unroll
    for (int q = 0; q < 72; ++q) acc[q] = 0.f;
    if (p < S * T) {
        const int t = p / S, s = p - t * S;
        int woff[9];   // offset of the tap's pixel inside a B1 plane, -1 = outside (zero padding)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ss = s + sign * (tap / 3 - 1), tt = t + sign * (tap % 3 - 1);
            woff[tap] = ss >= 0 && ss < S && tt >= 0 && tt < T ? ss * T + tt : -1;
        }
#pragma unroll 4
        for (int n = n0; n < n1; ++n) {
            const float *bp = B1 + (size_t)n * S * T;
            float win[9], av[8];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) win[tap] = woff[tap] >= 0 ? bp[woff[tap]] : 0.f;
#pragma unroll
            for (int ch = 0; ch < 8; ++ch) av[ch] = A8[((size_t)(n * 8 + ch) * T + t) * S + s];
#pragma unroll
            for (int ch = 0; ch < 8; ++ch)
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) acc[ch * 9 + tap] = fmaf(av[ch], win[tap], acc[ch * 9 + tap]);
        }
    }
    // the 256 threads' partials meet through LDS, 36 outputs at a time: thread q < 36 adds its row of 4 x 64 values
    // (72 x 6 cross-lane steps per wave were as slow as everything else in this kernel together)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int q = 0; q < 36; ++q) red[(wave * 36 + q) * 65 + lane] = acc[36 * half + q];
        __syncthreads();
        if (threadIdx.x < 36) {
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const float *r = red + (w * 36 + threadIdx.x) * 65;
#pragma unroll 4
                for (int l = 0; l < 64; l += 4) {
                    s0 += r[l];
                    s1 += r[l + 1];
                    s2 += r[l + 2];
                    s3 += r[l + 3];
                }
            }
            slices[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 72 + 36 * half + threadIdx.x] = (s0 + s1) + (s2 + s3);
        }
        __syncthreads();
    }
}

// per-channel sums of X [planes][C][npix] (bias gradients): one workgroup per (plane, channel), slices[plane][C]; the four tensors of a
// stack (g1, g2, g3, dy: 8 + 32 + 8 + 1 channels) in ONE launch (round 6: four launches of 5-13 us each before)
struct ChsumArgs {
    const float *X[4];
    float *slices[4];
    int first[5];   // first[k] = blocks before tensor k = planes * (channels of tensors < k)
    int npix;
};
__global__ __launch_bounds__(256) void chsum_kernel(const ChsumArgs a) {
    __shared__ float part[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int k = (int)blockIdx.x >= a.first[3] ? 3 : (int)blockIdx.x >= a.first[2] ? 2 : (int)blockIdx.x >= a.first[1] ? 1 : 0;
    const int local = (int)blockIdx.x - a.first[k], npix = a.npix;   // local = plane * C + channel
    float *__restrict__ slices = a.slices[k] + local;
    const float *x = a.X[k] + (size_t)local * npix;
    float v = 0.f;
    for (int p = threadIdx.x; p < npix; p += 256) v += x[p];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    if (lane == 0) part[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) *slices = (part[0] + part[1]) + (part[2] + part[3]);
}

// The data gradient of the stack is the stack itself run on dy with the transposed, flipped weights
// (conv4^T 1->8, conv3^T 8->32, conv2^T 32->8, conv1^T 8->1):  dst_k[ci][co][ky][kx] = w_{3-k}[co][ci][2-ky][2-kx].
// All four in one launch (4 801 weights; PyTorch's transpose + flip + contiguous were sixteen launches per stack).
struct FlipArgs {
    const float *w[4];   // conv1 .. conv4 weights [Co][Ci][3][3]
    float *dst;          // kConvFlipFloats floats: conv4^T | conv3^T | conv2^T | conv1^T
};
__global__ __launch_bounds__(256) void conv_flip_weights_kernel(const FlipArgs a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= kConvFlipFloats) return;
    // destination block, source tensor and its (Co, Ci)
    int k, off;
    if (i < 72) { k = 3; off = 0; }
    else if (i < 72 + 2304) { k = 2; off = 72; }
    else if (i < 72 + 4608) { k = 1; off = 72 + 2304; }
    else { k = 0; off = 72 + 4608; }
    const int co_n = k == 0 ? 8 : k == 1 ? 32 : k == 2 ? 8 : 1, ci_n = k == 0 ? 1 : k == 1 ? 8 : k == 2 ? 32 : 8;
    const int r = i - off, tap = r % 9, co = (r / 9) % co_n, ci = r / (9 * co_n);   // dst index = (ci * Co + co) * 9 + tap
    a.dst[i] = a.w[k][(co * ci_n + ci) * 9 + (8 - tap)];                            // (2-ky)*3 + (2-kx) = 8 - tap
}
hipError_t launch_conv_flip_weights(const float *const w[4], float *dst, hipStream_t st) {
    FlipArgs a{{w[0], w[1], w[2], w[3]}, dst};
    hipLaunchKernelGGL(conv_flip_weights_kernel, dim3((kConvFlipFloats + 255) / 256), dim3(256), 0, st, a);
    return hipGetLastError();
}

size_t conv_wgrad_slice_floats(int planes, int S, int T) {   // all eight reductions keep their slices until one launch
    return 2 * (size_t)kWgradSlices * 2304 + 2 * (size_t)planes * ((S * T + 255) / 256) * 72 + (size_t)planes * 64;
}

hipError_t launch_conv_wgrad(const float *x, const float *c1, const float *c2, const float *c3, const float *g1,
                             const float *g2, const float *g3, const float *dy, float *const dw[4], float *const db[4],
                             float *slices, int planes, int S, int T, bool accumulate, hipStream_t st) {
    const int npix = S * T, pblocks = (npix + 255) / 256;
    static PerDeviceOnce lds_attr;
    const size_t lds = sizeof(float) * 4 * kWgWave;
    hipError_t ea = ensure_dynamic_lds(lds_attr, reinterpret_cast<const void *>(wgrad32x8_kernel), lds);
    if (ea != hipSuccess) return ea;
    const int chunks = planes * T * ((S + 63) / 64);
    const int grid = std::max(1, std::min(kWgradSlices, (chunks + 3) / 4));
    hipError_t e;
    ReduceBatchScope reductions;
    float *sl2 = slices, *sl3 = sl2 + (size_t)kWgradSlices * 2304, *sl1 = sl3 + (size_t)kWgradSlices * 2304;
    float *sl4 = sl1 + (size_t)planes * pblocks * 72, *slb = sl4 + (size_t)planes * pblocks * 72;
    // conv2: dW2[co = a][ci = b][tap]          A32 = g2, B8 = a1, shift +
    Wgrad32x8Args w2{g2, c1, sl2, planes, S, T, +1, 72, 9};
    hipLaunchKernelGGL(wgrad32x8_kernel, dim3(grid), dim3(256), lds, st, w2);
    if ((e = launch_reduce_slices(sl2, dw[1], 2304, grid, 2304, accumulate, st)) != hipSuccess) return e;
    // conv3: dW3[co = b][ci = a][tap]          A32 = a2, B8 = g3, shift -
    Wgrad32x8Args w3{c2, g3, sl3, planes, S, T, -1, 9, 288};
    hipLaunchKernelGGL(wgrad32x8_kernel, dim3(grid), dim3(256), lds, st, w3);
    if ((e = launch_reduce_slices(sl3, dw[2], 2304, grid, 2304, accumulate, st)) != hipSuccess) return e;
    // conv1: dW1[co][tap] = sum g1[co][p] x[p + (tap-1)];  conv4: dW4[ci][tap] = sum a3[ci][p'] dy[p' - (tap-1)]
    const int pchunks = (planes + kWg81Planes - 1) / kWg81Planes;
    const Wgrad8x1Args w81{{g1, c3}, {x, dy}, {sl1, sl4}, {+1, -1}, planes, S, T};
    hipLaunchKernelGGL(wgrad8x1_kernel, dim3(pblocks, pchunks, 2), dim3(256), 0, st, w81);
    if ((e = launch_reduce_slices(sl1, dw[0], 72, pchunks * pblocks, 72, accumulate, st)) != hipSuccess) return e;
    if ((e = launch_reduce_slices(sl4, dw[3], 72, pchunks * pblocks, 72, accumulate, st)) != hipSuccess) return e;
    // biases: channel sums of the pre-activation gradients
    const float *gs[4] = {g1, g2, g3, dy};
    const int cs[4] = {8, 32, 8, 1};
    ChsumArgs ca{};
    ca.npix = npix;
    for (int k = 0; k < 4; ++k) {
        ca.X[k] = gs[k];
        ca.slices[k] = slb;
        ca.first[k + 1] = ca.first[k] + planes * cs[k];
        slb += (size_t)planes * cs[k];
    }
    hipLaunchKernelGGL(chsum_kernel, dim3(ca.first[4]), dim3(256), 0, st, ca);
    for (int k = 0; k < 4; ++k)
        if ((e = launch_reduce_slices(ca.slices[k], db[k], cs[k], planes, cs[k], accumulate, st)) != hipSuccess) return e;
    if ((e = reductions.flush(st)) != hipSuccess) return e;
    return hipGetLastError();
}

}  // namespace aft
