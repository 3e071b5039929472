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
// MI355X mapping: one workgroup (8 waves) per (plane, row band).  The whole receptive field
// lives in LDS: the 1-channel input, one 8-channel buffer (conv1 out, later conv3 out) and one
// 8-channel buffer holding ONE group of 8 of conv2's 32 output channels at a time; conv3's
// 8 outputs x 4 rows per thread accumulate in registers across the four groups, so the 32-channel
// tensor (250 KB/plane) never exists.  Weights are wave-uniform -> scalar loads feeding v_fma
// from SGPRs.  Each thread owns a 4-row x 1-column strip and slides a 6x3 window, 18 LDS reads
// per 288 FMAs.  HBM traffic per plane = compulsory only (pilots / x rows in, one plane out).
// Bands carry a 4-row halo (4 stacked 3x3 convs); the default 120x14 grid is one band.
#include "aft_internal.h"

namespace aft {

struct ConvArgs {
    int mode;  // 0 = head (pilots -> conv_enhanced), 1 = tail (x, conv_enhanced -> complex out)
    int S, T, TP, band_rows, nbands;   // TP = padded LDS row stride (>= T+2)
    // head
    const float *pilots, *up_w, *up_b;
    int pf;
    // tail
    const float *x, *lin2_w, *lin2_b, *resid;
    int d, tokens, p0, p1;
    const float *cw[4], *cb[4];
    float *out_plane;    // head: [planes][S][T]
    float *out_complex;  // tail: [B][S][T][2]
};

constexpr int kConvThreads = 1024;  // 16 waves: 4 per SIMD hide the LDS / scalar-load latency of the tap loops
constexpr int kStrip = 2;         // rows per thread (one v_pk_fma row pair)

// one 8-in -> 8-out 3x3 group on a 4-row strip: acc[row][o] += sum_ci sum_tap win * w
// wbase points at w[o = 0][ci = 0][0][0] of the group; strides in floats.
// The FMAs run on ROW PAIRS (v_pk_fma_f32: rows (0,1) and (2,3) of the strip share the weight), which
// halves the VALU issue of the 288 FMAs per input channel; the 6-row window is kept twice, as
// even-aligned pairs (0,1),(2,3),(4,5) and odd-aligned pairs (1,2),(3,4), so every tap's two rows are
// one register pair.
__device__ __forceinline__ void conv8x8_strip(const float *__restrict__ src, int plane_stride, int TP, int lr0,
                                              int LR, int col, const float *__restrict__ wbase, int w_o_stride,
                                              int w_ci_stride, f32x2 (&acc)[kStrip / 2][8]) {
    int rows[kStrip + 2];
#pragma unroll
    for (int i = 0; i < kStrip + 2; ++i) rows[i] = min(max(lr0 - 1 + i, 0), LR - 1) * TP + col;
#pragma unroll 1
    for (int ci = 0; ci < 8; ++ci) {
        const float *sp = src + ci * plane_stride;
        float win[kStrip + 2][3];
#pragma unroll
        for (int i = 0; i < kStrip + 2; ++i)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) win[i][dx] = sp[rows[i] + dx];  // col is the padded index of x-1
        // pair[i][dx] = (win[i], win[i+1]) for i = 0..kStrip
        f32x2 pr[kStrip + 1][3];
#pragma unroll
        for (int i = 0; i < kStrip + 1; ++i)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) pr[i][dx] = f32x2{win[i][dx], win[i + 1][dx]};
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            const float *wp = wbase + o * w_o_stride + ci * w_ci_stride;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const float wv = wp[dy * 3 + dx];
                    const f32x2 w2 = {wv, wv};
#pragma unroll
                    for (int pp = 0; pp < kStrip / 2; ++pp)   // rows 2pp, 2pp+1 use window rows 2pp+dy, 2pp+dy+1
                        acc[pp][o] = __builtin_elementwise_fma(pr[2 * pp + dy][dx], w2, acc[pp][o]);
                }
        }
    }
}

__global__ __launch_bounds__(kConvThreads) void conv_stack_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int S = a.S, T = a.T, TP = a.TP, LR = a.band_rows + 8;
    const int plane_stride = LR * TP;
    float *in0 = smem;                       // [LR][TP]
    float *bufA = in0 + plane_stride;        // [8][LR][TP]
    float *bufB = bufA + 8 * plane_stride;   // [8][LR][TP]
    float *small = bufB + 8 * plane_stride;  // head: pilot plane [pf]; tail: lin2 weights [p][d] + bias [p]

    const int tid = threadIdx.x;
    const int n = blockIdx.x / a.nbands, band = blockIdx.x % a.nbands;
    const int frame = n >> 1, part = n & 1;
    const int gr0 = band * a.band_rows - 4;  // global row of local row 0

    // ---- zero LDS (padding columns, rows outside the plane, unwritten halo rows) ----
    for (int i = tid; i < 17 * plane_stride; i += kConvThreads) smem[i] = 0.f;
    if (a.mode == 0) {
        for (int i = tid; i < a.pf; i += kConvThreads) small[i] = a.pilots[((size_t)frame * a.pf + i) * 2 + part];
    } else {
        const int p = a.p0 * a.p1;
        for (int i = tid; i < p * a.d; i += kConvThreads) small[i] = a.lin2_w[i];
        for (int i = tid; i < p; i += kConvThreads) small[p * a.d + i] = a.lin2_b[i];
    }
    __syncthreads();

    // ---- input plane ----
    for (int i = tid; i < LR * T; i += kConvThreads) {
        const int lr = i / T, t = i % T, gr = gr0 + lr;
        if (gr < 0 || gr >= S) continue;
        float v;
        if (a.mode == 0) {  // pilot_upsampler row (gr*T + t): idx = sc*T + sym (view(B,1,S,T))
            const int pix = gr * T + t;
            const float *wr = a.up_w + (size_t)pix * a.pf;
            v = a.up_b[pix];
            for (int kk = 0; kk < a.pf; ++kk) v = fmaf(wr[kk], small[kk], v);
        } else {            // linear_2 feature f of token tk + conv_enhanced residual
            const int tk = (gr / a.p0) * (T / a.p1) + t / a.p1, f = (gr % a.p0) * a.p1 + t % a.p1;
            const float *xr = a.x + ((size_t)n * a.tokens + tk) * a.d;
            const float *wr = small + f * a.d;
            float acc = small[a.p0 * a.p1 * a.d + f];
            for (int e = 0; e < a.d; e += 4) {
                const f32x4 xv = *reinterpret_cast<const f32x4 *>(xr + e);
                acc = fmaf(xv[0], wr[e], acc);
                acc = fmaf(xv[1], wr[e + 1], acc);
                acc = fmaf(xv[2], wr[e + 2], acc);
                acc = fmaf(xv[3], wr[e + 3], acc);
            }
            v = acc + a.resid[((size_t)n * S + gr) * T + t];
        }
        in0[lr * TP + t + 1] = v;
    }
    __syncthreads();

    // ---- conv1: 1 -> 8, ReLU, rows [1, LR-1) ----
    for (int i = tid; i < LR * T; i += kConvThreads) {
        const int lr = i / T, t = i % T, gr = gr0 + lr;
        if (lr < 1 || lr >= LR - 1 || gr < 0 || gr >= S) continue;
        float win[3][3];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) win[dy][dx] = in0[(lr - 1 + dy) * TP + t + dx];
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            float acc = a.cb[0][o];
#pragma unroll
            for (int k9 = 0; k9 < 9; ++k9) acc = fmaf(win[k9 / 3][k9 % 3], a.cw[0][o * 9 + k9], acc);
            bufA[o * plane_stride + lr * TP + t + 1] = fmaxf(acc, 0.f);
        }
    }
    __syncthreads();

    // ---- conv2 (8 -> 32, ReLU) in four 8-channel groups, conv3 (32 -> 8) accumulated in registers ----
    const int ngroups = (LR / kStrip) * T;
    const bool active = tid < ngroups;
    const int lr0 = (tid / T) * kStrip, col = tid % T;  // padded column index of x-1 is `col`
    bool any_valid = false;
#pragma unroll
    for (int rr = 0; rr < kStrip; ++rr) any_valid |= (gr0 + lr0 + rr >= 0 && gr0 + lr0 + rr < S);
    const bool work = active && any_valid;

    f32x2 acc3[kStrip / 2][8];   // [row pair][out channel]
#pragma unroll
    for (int pp = 0; pp < kStrip / 2; ++pp)
#pragma unroll
        for (int o = 0; o < 8; ++o) acc3[pp][o] = f32x2{0.f, 0.f};

#pragma unroll 1
    for (int g = 0; g < 4; ++g) {
        if (work) {
            f32x2 acc2[kStrip / 2][8];
#pragma unroll
            for (int o = 0; o < 8; ++o) {
                const float bias = a.cb[1][8 * g + o];
#pragma unroll
                for (int pp = 0; pp < kStrip / 2; ++pp) acc2[pp][o] = f32x2{bias, bias};
            }
            conv8x8_strip(bufA, plane_stride, TP, lr0, LR, col, a.cw[1] + (size_t)(8 * g) * 72, 72, 9, acc2);
#pragma unroll
            for (int rr = 0; rr < kStrip; ++rr) {
                const int lr = lr0 + rr, gr = gr0 + lr;
                if (lr >= 2 && lr < LR - 2 && gr >= 0 && gr < S) {
#pragma unroll
                    for (int o = 0; o < 8; ++o) bufB[o * plane_stride + lr * TP + col + 1] = fmaxf(acc2[rr >> 1][o][rr & 1], 0.f);
                }
            }
        }
        __syncthreads();
        if (work)  // conv3 weights [8][32][3][3]: o stride 288, ci stride 9, channel offset 8g
            conv8x8_strip(bufB, plane_stride, TP, lr0, LR, col, a.cw[2] + (size_t)(8 * g) * 9, 288, 9, acc3);
        __syncthreads();
    }

    // ---- conv3 epilogue: bias + ReLU -> bufA (conv1 output is dead), rows [3, LR-3) ----
    if (work) {
#pragma unroll
        for (int rr = 0; rr < kStrip; ++rr) {
            const int lr = lr0 + rr, gr = gr0 + lr;
            const bool ok = lr >= 3 && lr < LR - 3 && gr >= 0 && gr < S;
#pragma unroll
            for (int o = 0; o < 8; ++o)
                bufA[o * plane_stride + lr * TP + col + 1] = ok ? fmaxf(acc3[rr >> 1][o][rr & 1] + a.cb[2][o], 0.f) : 0.f;
        }
    }
    __syncthreads();

    // ---- conv4: 8 -> 1, no activation, rows of this band only ----
    for (int i = tid; i < a.band_rows * T; i += kConvThreads) {
        const int lr = 4 + i / T, t = i % T, gr = gr0 + lr;
        if (gr >= S) continue;
        float acc = a.cb[3][0];
#pragma unroll
        for (int ci = 0; ci < 8; ++ci)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
                    acc = fmaf(bufA[ci * plane_stride + (lr - 1 + dy) * TP + t + dx], a.cw[3][ci * 9 + dy * 3 + dx], acc);
        if (a.mode == 0)
            a.out_plane[((size_t)n * S + gr) * T + t] = acc;
        else
            a.out_complex[(((size_t)frame * S + gr) * T + t) * 2 + part] = acc;
    }
}

// pick the largest row band (multiple of 4, divides S) whose strips fit 512 threads and 160 KB LDS
// Row stride: a wave's lanes are (column fastest, then 4-row strip), so strips sit 4*TP floats apart;
// TP = T+2 = 16 puts every strip on the same banks (3-way conflicts measured: SQ_LDS_BANK_CONFLICT =
// 66 % of LDS cycles).  T+4 staggers strips by 8 banks; fall back to T+2 when LDS would overflow.
static bool plan_bands(int S, int T, int extra_floats, int *band_rows, int *tp, size_t *lds_bytes) {
    if (S % kStrip) return false;
    for (int nb = 1; nb <= S / kStrip; ++nb) {
        if (S % nb) continue;
        const int br = S / nb;
        if (br % kStrip) continue;
        const int LR = br + 8;
        if ((LR / kStrip) * T > kConvThreads) continue;
        for (int pad = 4; pad >= 2; pad -= 2) {
            const size_t bytes = sizeof(float) * ((size_t)17 * LR * (T + pad) + extra_floats);
            if (bytes <= 160 * 1024) {
                *band_rows = br;
                *tp = T + pad;
                *lds_bytes = bytes;
                return true;
            }
        }
    }
    return false;
}

static hipError_t launch_conv(ConvArgs &a, int planes, int extra_floats, hipStream_t st) {
    size_t lds = 0;
    if (!plan_bands(a.S, a.T, extra_floats, &a.band_rows, &a.TP, &lds)) return hipErrorInvalidValue;
    a.nbands = a.S / a.band_rows;
    static bool attr_set = false;  // idempotent; a race only repeats the same call
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_stack_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(conv_stack_kernel, dim3(planes * a.nbands), dim3(kConvThreads), lds, st, a);
    return hipGetLastError();
}

hipError_t launch_upsample(const aft_config &c, const aft_weights &w, const float *pilots, float *conv_enhanced,
                           int batch, hipStream_t st) {
    ConvArgs a{};
    a.mode = 0;
    a.S = c.num_scs; a.T = c.num_symbols;
    a.pilots = pilots; a.up_w = w.up_w; a.up_b = w.up_b;
    a.pf = c.pilot_scs * c.pilot_symbols;
    for (int i = 0; i < 4; ++i) { a.cw[i] = w.enh_w[i]; a.cb[i] = w.enh_b[i]; }
    a.out_plane = conv_enhanced;
    return launch_conv(a, 2 * batch, a.pf, st);
}

hipError_t launch_tail(const aft_config &c, const aft_weights &w, const float *x, const float *conv_enhanced,
                       float *out, int batch, hipStream_t st) {
    ConvArgs a{};
    a.mode = 1;
    a.S = c.num_scs; a.T = c.num_symbols;
    a.x = x; a.lin2_w = w.lin2_w; a.lin2_b = w.lin2_b; a.resid = conv_enhanced;
    a.d = c.model_dim; a.p0 = c.patch_scs; a.p1 = c.patch_symbols;
    a.tokens = (c.num_scs / c.patch_scs) * (c.num_symbols / c.patch_symbols);
    for (int i = 0; i < 4; ++i) { a.cw[i] = w.ref_w[i]; a.cb[i] = w.ref_b[i]; }
    a.out_complex = out;
    const int p = a.p0 * a.p1;
    return launch_conv(a, 2 * batch, p * a.d + p, st);
}

}  // namespace aft
