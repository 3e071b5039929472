// k_ends_train.hip -- the two THIN ends of the encoder in the training path (round 6).
//
// Reference semantics (src/models/fortitran.py:212-231, blocks/patch_processors.py:22,34-35, blocks/encoders.py:67-70):
//     tokens = cat(PatchEmbedding(conv_enhanced), channel_adapter(...))     [planes, tokens, p (+6)]
//     x0     = linear_1(tokens) + position_embeddings[:, :tokens]            [planes, tokens, d]         -- "embed"
//     ...encoder layers...
//     out    = conv_enhanced + InversePatchEmbedding(linear_2(x_L))          [planes, S, T]              -- "tail"
// Both ends are a product with 6 .. 38 columns on one side of a [planes*tokens, d] tensor: every one of them streams that tensor
// (37 MB at 128 frames of the default model) through HBM once and does next to no arithmetic.  Until round 6 the training path ran
// them as PyTorch's unfold / cat / broadcast add around this library's general GEMMs -- 64-row x 128-column tiles for 6- and 12-column
// products: 250 us of a 6.8-ms step in ~25 launches.  Here each end is ONE streaming launch forward and ONE backward (+ the fixed-order
// slice reductions of the parameter gradients):
//     embed_fwd   x0[row][c]   = b1[c] + pos[t][c] + sum_f W1[c][f] in[row][f]         in[row] = patch elements | adapter features
//     embed_bwd   d_in[row][f] = sum_c dx0[row][c] W1[c][f]   -> scattered to d_conv_enhanced / d_adapter_tokens
//                 dW1[c][f]    = sum_row dx0[row][c] in[row][f],  db1[c] = sum_row dx0[row][c],  dpos[t][c] = sum_planes dx0[n, t, c]
//     tail_fwd    out[n][s][t] = resid[n][s][t] + b2[f] + sum_c x[row][c] W2[f][c]      (row, f) <-> (n, s, t) by the patch map
//     tail_bwd    dx[row][c]   = sum_f d6[row][f] W2[f][c],  dW2[f][c] = sum_row d6[row][f] x[row][c],  db2[f] = sum_row d6[row][f]
//                 with d6[row][f] = d_out[n][s][t] (and d_resid = d_out: the caller passes the same tensor on)
// Every element-wise result (x0, out, d_conv_enhanced, d_tokens6, dx) goes through the SAME rounding sequence as in the launches it
// replaces -- products summed from zero in ascending feature order (an fp32 MFMA chain is that FMA chain: DESIGN 4.0 fact 11), the
// bias added behind the sum, the table / residual last -- so the activations and the gradients that flow on are bit-identical to the
// unfused path's (checked: the hashes of the encoder's and the conv stacks' gradients, tools/debug/full_grad_errs.py); only the five
// parameter gradients summed here (dW1, db1, dpos, dW2, db2) are added up in another order.
// Plain vector-ALU kernels: the bound is HBM, the products are staged through LDS so that every global access is a whole row piece.
// Parameter-gradient partials leave as one slice per workgroup and meet in reduce_jobs_kernel's fixed order (deterministic run to run).
// Planes are the training composite's [real planes | imaginary planes] stack; adapter features come PER PLANE ([planes][tokens][6]).
#include "aft_internal.h"

namespace aft {

namespace {

struct EndsDims {
    int planes, S, T, p0, p1, tokens, tw, d, p, din;   // tw = tokens per grid row (T / p1); din = p (+6 with adapter features)
    long rows;
};
size_t al64(size_t floats) { return (floats + 63) / 64 * 64; }   // slice regions start on 256-byte boundaries
constexpr int kEndsTile = 32;   // token rows per tile of the forward kernels and the tail's backward
constexpr int kEmbTok = 8;      // tokens per workgroup of the embed backward

__device__ __forceinline__ int patch_pixel(const EndsDims &g, int t, int f) {   // offset of patch element f of token t inside a plane
    const int tq = t / g.tw, tr = t - tq * g.tw;
    return (tq * g.p0 + f / g.p1) * g.T + tr * g.p1 + f % g.p1;
}

// ---- embed forward: persistent workgroups, W1 transposed in LDS once, 32-row tiles ----
struct EmbedFwdArgs {
    EndsDims g;
    const float *conv, *tok6, *w1, *b1, *pos;
    float *x;
    int tok_shift;   // adapter features of plane n: row (n >> tok_shift) of tok6 -- 0: per plane (training), 1: per frame (the general engine)
    int pos_last;    // 1: ((sum_f ..) + b1) + pos -- the rounding sequence of the launches this kernel replaced in the training path (the
                     // GEMM adds its bias in the epilogue, PyTorch adds the table): the SAME BITS in x0, so every gradient fixture sees
                     // the activations it saw before; 0: (b1 + pos) + sum_f .. -- embed_any_kernel's sequence (the general engine)
};
__global__ __launch_bounds__(256) void embed_rows_fwd_kernel(const EmbedFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const EndsDims &g = a.g;
    float *w1t = lds;                      // [din][d]
    float *ins = lds + g.din * g.d;        // [32][din]
    const int tid = threadIdx.x, qn = g.d / 4, rg_n = 256 / qn, rg = tid / qn, q = tid - rg * qn;
    for (int i = tid; i < g.d * g.din; i += 256) {
        const int c = i / g.din, f = i - c * g.din;
        w1t[f * g.d + c] = a.w1[i];
    }
    const long ntiles = (g.rows + kEndsTile - 1) / kEndsTile;
    f32x4 bias = {0.f, 0.f, 0.f, 0.f};
    if (rg < rg_n) bias = *reinterpret_cast<const f32x4 *>(a.b1 + 4 * q);
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long row0 = tile * kEndsTile;
        __syncthreads();   // the previous tile's readers are done with `ins` (and, first time, w1t is complete after the next barrier)
        for (int i = tid; i < kEndsTile * g.din; i += 256) {
            const int r = i / g.din, f = i - r * g.din;
            const long row = row0 + r;
            float v = 0.f;
            if (row < g.rows) {
                const int n = (int)(row / g.tokens), t = (int)(row - (long)n * g.tokens);
                v = f < g.p ? a.conv[(size_t)n * g.S * g.T + patch_pixel(g, t, f)]
                            : a.tok6[((size_t)(n >> a.tok_shift) * g.tokens + t) * 6 + (f - g.p)];
            }
            ins[i] = v;
        }
        __syncthreads();
        if (rg < rg_n) {
            for (int r = rg; r < kEndsTile; r += rg_n) {
                const long row = row0 + r;
                if (row >= g.rows) break;
                const int t = (int)(row % g.tokens);
                const f32x4 pv = *reinterpret_cast<const f32x4 *>(a.pos + (size_t)t * g.d + 4 * q);
                f32x4 acc = a.pos_last ? f32x4{0.f, 0.f, 0.f, 0.f} : bias + pv;
                const float *in = ins + r * g.din;
                for (int f = 0; f < g.din; ++f) {
                    const f32x4 w = *reinterpret_cast<const f32x4 *>(w1t + f * g.d + 4 * q);
                    const float v = in[f];
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[c] = fmaf(w[c], v, acc[c]);
                }
                if (a.pos_last) acc = (acc + bias) + pv;
                *reinterpret_cast<f32x4 *>(a.x + (size_t)row * g.d + 4 * q) = acc;
            }
        }
    }
}

// ---- embed backward: workgroup = (8 tokens, a chunk of planes), four planes per pass; dpos / dW1 partials live in registers ----
struct EmbedBwdArgs {
    EndsDims g;
    const float *conv, *tok6, *w1, *dx;
    float *d_conv, *d_tok6;
    float *sl_w, *sl_b, *sl_pos;   // slices: [wg][d * din], [wg][d], [plane chunk][tokens * d] (sl_pos NULL: no table gradient wanted)
    int planes_per_chunk;
};
constexpr int kEmbBwdMaxOut = 80;   // d * din / 256 accumulators per thread (512 x 38 -> 76)
constexpr int kEmbBwdMaxPieces = 4; // 8 rows x d / 4 sixteen-byte pieces over 256 threads (d = 512 -> 4)
constexpr int kEmbPass = 4;         // planes staged per pass: 32 rows between two barriers, four 16-byte loads in flight per thread
// NOUT / NPIECE: the register arrays' sizes (the loops over them are unrolled: an instantiation per size class keeps the default
// model's 6 + 1 from being allocated as 80 + 4)
template <int NOUT, int NPIECE>
__global__ __launch_bounds__(256) void embed_rows_bwd_kernel(const EmbedBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const EndsDims &g = a.g;
    const int ldw = g.d + 4;                       // W1^T rows on different banks, 16-byte aligned
    float *w1t = lds;                              // [din][ldw]
    float *dhs = w1t + g.din * ldw;                // [4 planes][8][d]
    float *ins = dhs + kEmbPass * kEmbTok * g.d;   // [4 planes][8][din]
    const int tid = threadIdx.x, qn = g.d / 4, npieces = kEmbTok * qn, nout = g.d * g.din, tile = kEmbTok * g.d, nin = kEmbTok * g.din;
    const int t0 = blockIdx.x * kEmbTok, nrow = min(kEmbTok, g.tokens - t0);
    const int n0 = blockIdx.y * a.planes_per_chunk, n1 = min(g.planes, n0 + a.planes_per_chunk);
    const size_t plane_px = (size_t)g.S * g.T;
    for (int i = tid; i < nout; i += 256) {
        const int c = i / g.din, f = i - c * g.din;
        w1t[f * ldw + c] = a.w1[i];
    }
    f32x4 dpos[NPIECE];
    float dw[NOUT];
#pragma unroll
    for (int k = 0; k < NPIECE; ++k) dpos[k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < NOUT; ++k) dw[k] = 0.f;
    for (int nb = n0; nb < n1; nb += kEmbPass) {
        const int np = min(kEmbPass, n1 - nb);
        __syncthreads();   // the previous pass's readers are done with dhs / ins
        for (int pl = 0; pl < np; ++pl) {
            const size_t rowbase = (size_t)(nb + pl) * g.tokens + t0;
#pragma unroll
            for (int k = 0; k < NPIECE; ++k) {
                const int i = tid + 256 * k;
                if (i < npieces) {
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (i / qn < nrow) v = *reinterpret_cast<const f32x4 *>(a.dx + rowbase * g.d + (size_t)i * 4);   // the 8 rows are contiguous
                    dpos[k] += v;
                    *reinterpret_cast<f32x4 *>(dhs + pl * tile + i * 4) = v;
                }
            }
        }
        for (int i = tid; i < np * nin; i += 256) {
            const int pl = i / nin, j = i - pl * nin, r = j / g.din, f = j - r * g.din;
            const int n = nb + pl;
            float v = 0.f;
            if (r < nrow)
                v = f < g.p ? a.conv[n * plane_px + patch_pixel(g, t0 + r, f)] : a.tok6[((size_t)n * g.tokens + t0 + r) * 6 + (f - g.p)];
            ins[i] = v;
        }
        __syncthreads();
        // d_in[plane][r][f] = sum_c dh[r][c] W1[c][f]: one (plane, row, feature) per thread, sixteen bytes of both operands per step
        for (int i = tid; i < np * nin; i += 256) {
            const int pl = i / nin, j = i - pl * nin, r = j / g.din, f = j - r * g.din;
            if (r >= nrow) continue;
            const float *dh = dhs + pl * tile + r * g.d, *wt = w1t + f * ldw;
            float acc = 0.f;
            for (int q = 0; q < qn; ++q) {
                const f32x4 x = *reinterpret_cast<const f32x4 *>(dh + 4 * q), w = *reinterpret_cast<const f32x4 *>(wt + 4 * q);
#pragma unroll
                for (int c = 0; c < 4; ++c) acc = fmaf(x[c], w[c], acc);
            }
            const int n = nb + pl;
            if (f < g.p) a.d_conv[n * plane_px + patch_pixel(g, t0 + r, f)] = acc;
            else a.d_tok6[((size_t)n * g.tokens + t0 + r) * 6 + (f - g.p)] = acc;
        }
        // dW1[c][f] += sum_(plane, r) dh[r][c] in[r][f]: output o = tid + 256 k <-> (f = o / d, c = o % d), lanes along c
#pragma unroll
        for (int k = 0; k < NOUT; ++k) {
            const int o = tid + 256 * k;
            if (o < nout) {
                const int f = o / g.d, c = o - f * g.d;
                float acc = dw[k];
                for (int j = 0; j < np * kEmbTok; ++j) acc = fmaf(dhs[j * g.d + c], ins[j * g.din + f], acc);
                dw[k] = acc;
            }
        }
    }
    const size_t wg = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
#pragma unroll
    for (int k = 0; k < NOUT; ++k) {
        const int o = tid + 256 * k;
        if (o < nout) {
            const int f = o / g.d, c = o - f * g.d;
            a.sl_w[wg * nout + (size_t)c * g.din + f] = dw[k];
        }
    }
    // the table's gradient rows of this chunk, and their column sums (= the bias gradient's share)
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NPIECE; ++k) {
        const int i = tid + 256 * k;
        if (i < npieces) {
            *reinterpret_cast<f32x4 *>(dhs + i * 4) = dpos[k];
            if (a.sl_pos != nullptr && i / qn < nrow)
                *reinterpret_cast<f32x4 *>(a.sl_pos + ((size_t)blockIdx.y * g.tokens + t0) * g.d + (size_t)i * 4) = dpos[k];
        }
    }
    __syncthreads();
    for (int c = tid; c < g.d; c += 256) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < kEmbTok; ++r) s += dhs[r * g.d + c];
        a.sl_b[wg * g.d + c] = s;
    }
}

// ---- embed backward of the default model (d = 128, at most 16 input features): both products of a pass on 16x16x4 MFMAs ----
// The generic kernel above is bound by its vector ALU and LDS reads (PMC: 3 500 vector instructions per wave, 57 % of the kernel's
// cycles; as many LDS bank-conflict cycles as LDS instructions): per pass a thread issues ~450 FMAs with two LDS reads each.  An fp32
// MFMA is that same FMA chain, fed 16 x 4 operands per instruction:
//     d_in[32 rows][16 f] = dh[32][128] . W1[128][16]      two row blocks (waves 0, 1) x 32 steps over c, ascending
//     dW1[128 c][16 f]   += dh^T[128][32] . in[32][16]      eight c blocks (two per wave) x 8 steps over the pass's rows, ascending
// -- 128 matrix instructions per pass and workgroup instead of ~4 700 vector ones per wave, operands read from LDS once per 16 x 4
// tile (dh rows padded to 132 floats: the d_in reads are conflict-free), the same summation order as the generic kernel: same bits.
__global__ __launch_bounds__(256) void embed_rows_bwd128_kernel(const EmbedBwdArgs a) {
    constexpr int D = 128, LDH = D + 4, FP = 16, ROWS = kEmbPass * kEmbTok;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const EndsDims &g = a.g;
    float *w1t = lds;                // [16][LDH]: W1^T, rows >= din zero
    float *dhs = w1t + FP * LDH;     // [32][LDH]: the pass's gradient rows (plane-major, 8 tokens each)
    float *ins = dhs + ROWS * LDH;   // [32][16]: the pass's input features, columns >= din zero
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, g4 = lane >> 4;
    const int t0 = blockIdx.x * kEmbTok, nrow = min(kEmbTok, g.tokens - t0), nout = D * g.din;
    const int n0 = blockIdx.y * a.planes_per_chunk, n1 = min(g.planes, n0 + a.planes_per_chunk);
    const size_t plane_px = (size_t)g.S * g.T;
    for (int i = tid; i < FP * D; i += 256) {
        const int f = i >> 7, c = i & (D - 1);
        w1t[f * LDH + c] = f < g.din ? a.w1[c * g.din + f] : 0.f;
    }
    // what does not change from pass to pass: the staging piece (row, quad) of this thread, its gather item (row r, feature f; planes
    // pl and pl + 2 of the pass) and -- waves 0, 1 -- the four d_in results it stores (rows 4 (g4 & 1) + v of plane 2 wave + g4 / 2)
    const int srow = tid >> 5, sq = tid & 31;
    const int gr = (tid >> 4) & 7, gf = tid & 15, gpl = tid >> 7;
    const bool gvalid = gf < g.din && gr < nrow;
    const int goff = gf < g.p ? patch_pixel(g, t0 + min(gr, nrow - 1), gf) : (t0 + gr) * 6 + (gf - g.p);
    const int opl = 2 * wave + (g4 >> 1);
    int ooff[4];
    bool ovalid[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int r = 4 * (g4 & 1) + v;
        ovalid[v] = wave < 2 && i16 < g.din && r < nrow;
        ooff[v] = i16 < g.p ? patch_pixel(g, t0 + min(r, nrow - 1), i16) : (t0 + r) * 6 + (i16 - g.p);
    }
    f32x4 dpos = {0.f, 0.f, 0.f, 0.f};
    f32x4 dw[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    for (int nb = n0; nb < n1; nb += kEmbPass) {
        const int np = min(kEmbPass, n1 - nb);
        __syncthreads();   // the previous pass's readers are done with dhs / ins
#pragma unroll
        for (int pl = 0; pl < kEmbPass; ++pl) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (pl < np && srow < nrow) v = *reinterpret_cast<const f32x4 *>(a.dx + ((size_t)(nb + pl) * g.tokens + t0) * D + (size_t)tid * 4);
            dpos += v;
            *reinterpret_cast<f32x4 *>(dhs + (pl * kEmbTok + srow) * LDH + 4 * sq) = v;   // (planes beyond the chunk: zero rows)
        }
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int pl = gpl + 2 * m;
            float v = 0.f;
            if (gvalid && pl < np) {
                const size_t n = (size_t)(nb + pl);
                v = gf < g.p ? a.conv[n * plane_px + goff] : a.tok6[n * g.tokens * 6 + goff];
            }
            ins[(pl * kEmbTok + gr) * FP + gf] = v;
        }
        __syncthreads();
        if (wave < 2) {   // d_in rows 16 wave .. 16 wave + 15: A[i][k] = dh[16 wave + i][4 s + k], B[k][j] = W1[4 s + k][j]
            const float *ap = dhs + (16 * wave + i16) * LDH + g4, *bp = w1t + i16 * LDH + g4;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
            for (int s = 0; s < D / 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[4 * s], bp[4 * s], acc, 0, 0, 0);
            if (opl < np) {
                const size_t n = (size_t)(nb + opl);
#pragma unroll
                for (int v = 0; v < 4; ++v) {   // D[i = 4 g4 + v][j = i16]
                    if (!ovalid[v]) continue;
                    if (i16 < g.p) a.d_conv[n * plane_px + ooff[v]] = acc[v];
                    else a.d_tok6[n * g.tokens * 6 + ooff[v]] = acc[v];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {   // dW1 rows c = 16 (2 wave + u) + ..: A[i][k] = dh[4 s + k][16 cb + i], B[k][j] = in[4 s + k][j]
            const float *ap = dhs + g4 * LDH + 16 * (2 * wave + u) + i16, *bp = ins + g4 * FP + i16;
#pragma unroll
            for (int s = 0; s < ROWS / 4; ++s) dw[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[4 * s * LDH], bp[4 * s * FP], dw[u], 0, 0, 0);
        }
    }
    const size_t wg = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    if (i16 < g.din) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int v = 0; v < 4; ++v) a.sl_w[wg * nout + (size_t)(16 * (2 * wave + u) + 4 * g4 + v) * g.din + i16] = dw[u][v];
    }
    __syncthreads();
    float *tmp = dhs;   // [8][D]: the table's gradient rows of this chunk, for their column sums (= the bias gradient's share)
    *reinterpret_cast<f32x4 *>(tmp + tid * 4) = dpos;
    if (a.sl_pos != nullptr && srow < nrow)
        *reinterpret_cast<f32x4 *>(a.sl_pos + ((size_t)blockIdx.y * g.tokens + t0) * D + (size_t)tid * 4) = dpos;
    __syncthreads();
    if (tid < D) {
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < kEmbTok; ++r) sum += tmp[r * D + tid];
        a.sl_b[wg * D + tid] = sum;
    }
}

// ---- tail forward: persistent workgroups, W2 in LDS once, 32-row tiles of x through LDS ----
struct TailFwdArgs {
    EndsDims g;
    const float *x, *w2, *b2, *resid;
    float *out;
};
__global__ __launch_bounds__(256) void tail_rows_fwd_kernel(const TailFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const EndsDims &g = a.g;
    const int ldx = g.d + 4;               // row stride of both LDS blocks: 16-byte aligned, rows on different banks
    float *w2s = lds;                      // [p][ldx]
    float *xs = lds + g.p * ldx;           // [32][ldx]
    const int tid = threadIdx.x, qn = g.d / 4;
    for (int i = tid; i < g.p * qn; i += 256) {
        const int f = i / qn, q = i - f * qn;
        *reinterpret_cast<f32x4 *>(w2s + f * ldx + 4 * q) = *reinterpret_cast<const f32x4 *>(a.w2 + (size_t)f * g.d + 4 * q);
    }
    const long ntiles = (g.rows + kEndsTile - 1) / kEndsTile;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long row0 = tile * kEndsTile;
        __syncthreads();
        for (int i = tid; i < kEndsTile * qn; i += 256) {
            const int r = i / qn, q = i - r * qn;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (row0 + r < g.rows) v = *reinterpret_cast<const f32x4 *>(a.x + (size_t)(row0 + r) * g.d + 4 * q);
            *reinterpret_cast<f32x4 *>(xs + r * ldx + 4 * q) = v;
        }
        __syncthreads();
        for (int i = tid; i < kEndsTile * g.p; i += 256) {
            const int r = i / g.p, f = i - r * g.p;
            const long row = row0 + r;
            if (row >= g.rows) continue;
            const float *xr = xs + r * ldx, *wr = w2s + f * ldx;
            float acc = 0.f;   // (sum_c ..) + b2, then the residual: the rounding sequence of the GEMM + PyTorch add this kernel replaced
            for (int q = 0; q < qn; ++q) {
                const f32x4 xv = *reinterpret_cast<const f32x4 *>(xr + 4 * q), wv = *reinterpret_cast<const f32x4 *>(wr + 4 * q);
#pragma unroll
                for (int c = 0; c < 4; ++c) acc = fmaf(xv[c], wv[c], acc);
            }
            const int n = (int)(row / g.tokens), t = (int)(row - (long)n * g.tokens);
            const size_t px = (size_t)n * g.S * g.T + patch_pixel(g, t, f);
            a.out[px] = a.resid[px] + (acc + a.b2[f]);
        }
    }
}

// ---- tail backward: persistent workgroups; dx per tile, dW2 / db2 partials in registers across the workgroup's tiles ----
struct TailBwdArgs {
    EndsDims g;
    const float *x, *w2, *d_out;
    float *dx;
    float *sl_w, *sl_b;   // [wg][p * d], [wg][p]
};
constexpr int kTailMaxF = 32;   // patch elements (kMaxPatchGeneral)
// NF: patch elements per thread in the weight gradient (ceil(p / (256 / (d / 4)))): 1 for the default model
template <int NF>
__global__ __launch_bounds__(256) void tail_rows_bwd_kernel(const TailBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const EndsDims &g = a.g;
    const int tid = threadIdx.x, qn = g.d / 4, fg_n = 256 / qn, fg = tid / qn, q = tid - fg * qn;
    float *w2s = lds;                      // [p][d]
    float *xs = w2s + g.p * g.d;           // [32][d]
    float *d6s = xs + kEndsTile * g.d;     // [32][p]
    for (int i = tid; i < g.p * qn; i += 256) *reinterpret_cast<f32x4 *>(w2s + 4 * i) = *reinterpret_cast<const f32x4 *>(a.w2 + 4 * (size_t)i);
    const int nf = fg < fg_n ? (g.p - fg + fg_n - 1) / fg_n : 0;   // this thread's patch elements f = fg + fg_n k (weight gradient)
    f32x4 dw[NF];
#pragma unroll
    for (int k = 0; k < NF; ++k) dw[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    float db = 0.f;
    const long ntiles = (g.rows + kEndsTile - 1) / kEndsTile;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long row0 = tile * kEndsTile;
        __syncthreads();
        for (int i = tid; i < kEndsTile * qn; i += 256) {
            const int r = i / qn;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (row0 + r < g.rows) v = *reinterpret_cast<const f32x4 *>(a.x + (size_t)row0 * g.d + 4 * (size_t)i);
            *reinterpret_cast<f32x4 *>(xs + 4 * i) = v;
        }
        for (int i = tid; i < kEndsTile * g.p; i += 256) {
            const int r = i / g.p, f = i - r * g.p;
            const long row = row0 + r;
            float v = 0.f;
            if (row < g.rows) {
                const int n = (int)(row / g.tokens), t = (int)(row - (long)n * g.tokens);
                v = a.d_out[(size_t)n * g.S * g.T + patch_pixel(g, t, f)];
            }
            d6s[i] = v;
        }
        __syncthreads();
        if (fg < fg_n) {
            // dx[r][4q ..] = sum_f d6[r][f] W2[f][4q ..] for the rows r = fg + fg_n k
            for (int r = fg; r < kEndsTile; r += fg_n) {
                if (row0 + r >= g.rows) break;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                for (int f = 0; f < g.p; ++f) {
                    const f32x4 w = *reinterpret_cast<const f32x4 *>(w2s + f * g.d + 4 * q);
                    const float v = d6s[r * g.p + f];
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[c] = fmaf(v, w[c], acc[c]);
                }
                *reinterpret_cast<f32x4 *>(a.dx + (size_t)(row0 + r) * g.d + 4 * q) = acc;
            }
            // dW2[f][4q ..] += sum_r d6[r][f] x[r][4q ..]
#pragma unroll
            for (int k = 0; k < NF; ++k) {
                if (k < nf) {
                    const int f = fg + fg_n * k;
                    f32x4 acc = dw[k];
                    for (int r = 0; r < kEndsTile; ++r) {
                        const f32x4 xv = *reinterpret_cast<const f32x4 *>(xs + r * g.d + 4 * q);
                        const float v = d6s[r * g.p + f];
#pragma unroll
                        for (int c = 0; c < 4; ++c) acc[c] = fmaf(v, xv[c], acc[c]);
                    }
                    dw[k] = acc;
                }
            }
        }
        if (tid < g.p) {
#pragma unroll 8
            for (int r = 0; r < kEndsTile; ++r) db += d6s[r * g.p + tid];
        }
    }
    const size_t wg = blockIdx.x;
    if (fg < fg_n) {
#pragma unroll
        for (int k = 0; k < NF; ++k)
            if (k < nf) *reinterpret_cast<f32x4 *>(a.sl_w + wg * g.p * g.d + (size_t)(fg + fg_n * k) * g.d + 4 * q) = dw[k];
    }
    if (tid < g.p) a.sl_b[wg * g.p + tid] = db;
}

bool make_dims(EndsDims &g, int planes, int S, int T, int p0, int p1, int d, bool adapter) {
    if (planes <= 0 || S <= 0 || T <= 0 || p0 <= 0 || p1 <= 0 || S % p0 || T % p1 || d < 4 || d % 4 || d > 1024) return false;
    g.planes = planes; g.S = S; g.T = T; g.p0 = p0; g.p1 = p1; g.d = d;
    g.tw = T / p1;
    g.tokens = (S / p0) * g.tw;
    g.p = p0 * p1;
    g.din = g.p + (adapter ? 6 : 0);
    g.rows = (long)planes * g.tokens;
    return d <= 512 && g.p <= kTailMaxF && (long)d * g.din <= 256L * kEmbBwdMaxOut && kEmbTok * (d / 4) <= 256 * kEmbBwdMaxPieces;
}
int embed_bwd_chunks(const EndsDims &g) {   // plane chunks: about four workgroups per CU over (token blocks x chunks)
    const int tb = (g.tokens + kEmbTok - 1) / kEmbTok;
    return std::max(1, std::min(g.planes, (4 * current_device_cus() + tb - 1) / tb));
}
int ends_grid(const EndsDims &g) { return (int)std::min<long>((g.rows + kEndsTile - 1) / kEndsTile, 4L * current_device_cus()); }

}  // namespace

bool ends_train_ok(int planes, int S, int T, int p0, int p1, int d, bool adapter) {
    EndsDims g;
    return make_dims(g, planes, S, T, p0, p1, d, adapter);
}

size_t embed_bwd_slice_floats(int planes, int S, int T, int p0, int p1, int d, bool adapter) {
    EndsDims g;
    if (!make_dims(g, planes, S, T, p0, p1, d, adapter)) return 0;
    const size_t tb = (g.tokens + kEmbTok - 1) / kEmbTok, ch = embed_bwd_chunks(g);
    return al64(tb * ch * (size_t)g.d * g.din) + al64(tb * ch * (size_t)g.d) + al64(ch * (size_t)g.tokens * g.d);
}
size_t tail_bwd_slice_floats(int planes, int S, int T, int p0, int p1, int d) {
    EndsDims g;
    if (!make_dims(g, planes, S, T, p0, p1, d, false)) return 0;
    const size_t wgs = ends_grid(g);
    return al64(wgs * (size_t)g.p * g.d) + al64(wgs * (size_t)g.p);
}

hipError_t launch_embed_train_fwd(const float *conv, const float *tok6, const float *w1, const float *b1, const float *pos, float *x,
                                  int planes, int S, int T, int p0, int p1, int d, hipStream_t st, bool tok6_per_frame) {
    EmbedFwdArgs a{};
    if (!make_dims(a.g, planes, S, T, p0, p1, d, tok6 != nullptr)) return hipErrorInvalidValue;
    a.conv = conv; a.tok6 = tok6; a.w1 = w1; a.b1 = b1; a.pos = pos; a.x = x;
    a.tok_shift = tok6_per_frame ? 1 : 0;
    a.pos_last = tok6_per_frame ? 0 : 1;
    const size_t lds = sizeof(float) * ((size_t)a.g.din * d + (size_t)kEndsTile * a.g.din);
    static PerDeviceOnce attr;
    hipError_t e = ensure_dynamic_lds(attr, reinterpret_cast<const void *>(embed_rows_fwd_kernel), 160 * 1024);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(embed_rows_fwd_kernel, dim3(ends_grid(a.g)), dim3(256), lds, st, a);
    return hipGetLastError();
}

hipError_t launch_embed_train_bwd(const float *conv, const float *tok6, const float *w1, const float *dx, float *d_conv, float *d_tok6,
                                  float *dw1, float *db1, float *dpos, bool accumulate, float *slices, int planes, int S, int T, int p0,
                                  int p1, int d, hipStream_t st) {
    EmbedBwdArgs a{};
    if (!make_dims(a.g, planes, S, T, p0, p1, d, tok6 != nullptr) || (tok6 != nullptr && d_tok6 == nullptr)) return hipErrorInvalidValue;
    const EndsDims &g = a.g;
    const int tb = (g.tokens + kEmbTok - 1) / kEmbTok, ch = embed_bwd_chunks(g);
    a.conv = conv; a.tok6 = tok6; a.w1 = w1; a.dx = dx; a.d_conv = d_conv; a.d_tok6 = d_tok6;
    a.planes_per_chunk = (planes + ch - 1) / ch;
    const int chunks = (planes + a.planes_per_chunk - 1) / a.planes_per_chunk;
    a.sl_w = slices;
    a.sl_b = a.sl_w + al64((size_t)tb * ch * g.d * g.din);
    a.sl_pos = dpos != nullptr ? a.sl_b + al64((size_t)tb * ch * g.d) : nullptr;
    const size_t lds = sizeof(float) * ((size_t)g.din * (g.d + 4) + (size_t)kEmbPass * kEmbTok * (g.d + g.din));
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    static PerDeviceOnce attr[4];
    const int nout = (g.d * g.din + 255) / 256, npiece = (kEmbTok * (g.d / 4) + 255) / 256;
    hipError_t e;
    if (g.d == 128 && g.din <= 16 && !switch_on("AFT_EMBED_BWD_GENERIC")) {   // the default model: both products on MFMAs
        const size_t lds128 = sizeof(float) * ((16 + kEmbPass * kEmbTok) * (128 + 4) + kEmbPass * kEmbTok * 16);
        if ((e = ensure_dynamic_lds(attr[3], reinterpret_cast<const void *>(embed_rows_bwd128_kernel), 160 * 1024)) != hipSuccess) return e;
        hipLaunchKernelGGL(embed_rows_bwd128_kernel, dim3(tb, chunks), dim3(256), lds128, st, a);
    } else if (nout <= 8 && npiece <= 1) {
        if ((e = ensure_dynamic_lds(attr[0], reinterpret_cast<const void *>(embed_rows_bwd_kernel<8, 1>), 160 * 1024)) != hipSuccess) return e;
        hipLaunchKernelGGL((embed_rows_bwd_kernel<8, 1>), dim3(tb, chunks), dim3(256), lds, st, a);
    } else if (nout <= 24) {
        if ((e = ensure_dynamic_lds(attr[1], reinterpret_cast<const void *>(embed_rows_bwd_kernel<24, 4>), 160 * 1024)) != hipSuccess) return e;
        hipLaunchKernelGGL((embed_rows_bwd_kernel<24, 4>), dim3(tb, chunks), dim3(256), lds, st, a);
    } else {
        if ((e = ensure_dynamic_lds(attr[2], reinterpret_cast<const void *>(embed_rows_bwd_kernel<kEmbBwdMaxOut, kEmbBwdMaxPieces>), 160 * 1024)) != hipSuccess) return e;
        hipLaunchKernelGGL((embed_rows_bwd_kernel<kEmbBwdMaxOut, kEmbBwdMaxPieces>), dim3(tb, chunks), dim3(256), lds, st, a);
    }
    if ((e = hipGetLastError()) != hipSuccess) return e;
    ReduceBatchScope reductions;
    const int nz = tb * chunks;
    if ((e = launch_reduce_slices(a.sl_w, dw1, g.d * g.din, nz, (size_t)g.d * g.din, accumulate, st)) != hipSuccess) return e;
    if (db1 != nullptr && (e = launch_reduce_slices(a.sl_b, db1, g.d, nz, (size_t)g.d, accumulate, st)) != hipSuccess) return e;
    if (dpos != nullptr &&
        (e = launch_reduce_slices(a.sl_pos, dpos, g.tokens * g.d, chunks, (size_t)g.tokens * g.d, accumulate, st)) != hipSuccess)
        return e;
    return reductions.flush(st);
}

hipError_t launch_tail_train_fwd(const float *x, const float *w2, const float *b2, const float *resid, float *out, int planes, int S, int T,
                                 int p0, int p1, int d, hipStream_t st) {
    TailFwdArgs a{};
    if (!make_dims(a.g, planes, S, T, p0, p1, d, false)) return hipErrorInvalidValue;
    a.x = x; a.w2 = w2; a.b2 = b2; a.resid = resid; a.out = out;
    const size_t lds = sizeof(float) * (size_t)(a.g.p + kEndsTile) * (d + 4);
    static PerDeviceOnce attr;
    hipError_t e = ensure_dynamic_lds(attr, reinterpret_cast<const void *>(tail_rows_fwd_kernel), 160 * 1024);
    if (e != hipSuccess) return e;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(tail_rows_fwd_kernel, dim3(ends_grid(a.g)), dim3(256), lds, st, a);
    return hipGetLastError();
}

hipError_t launch_tail_train_bwd(const float *x, const float *w2, const float *d_out, float *dx, float *dw2, float *db2, bool accumulate,
                                 float *slices, int planes, int S, int T, int p0, int p1, int d, hipStream_t st) {
    TailBwdArgs a{};
    if (!make_dims(a.g, planes, S, T, p0, p1, d, false)) return hipErrorInvalidValue;
    const EndsDims &g = a.g;
    const int wgs = ends_grid(g);
    a.x = x; a.w2 = w2; a.d_out = d_out; a.dx = dx;
    a.sl_w = slices;
    a.sl_b = slices + al64((size_t)wgs * g.p * g.d);
    const size_t lds = sizeof(float) * ((size_t)g.p * g.d + (size_t)kEndsTile * g.d + (size_t)kEndsTile * g.p);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    static PerDeviceOnce attr[3];
    const int fg_n = 256 / (g.d / 4), nf = (g.p + fg_n - 1) / fg_n;
    hipError_t e;
    if (nf <= 1) {
        if ((e = ensure_dynamic_lds(attr[0], reinterpret_cast<const void *>(tail_rows_bwd_kernel<1>), 160 * 1024)) != hipSuccess) return e;
        hipLaunchKernelGGL(tail_rows_bwd_kernel<1>, dim3(wgs), dim3(256), lds, st, a);
    } else if (nf <= 4) {
        if ((e = ensure_dynamic_lds(attr[1], reinterpret_cast<const void *>(tail_rows_bwd_kernel<4>), 160 * 1024)) != hipSuccess) return e;
        hipLaunchKernelGGL(tail_rows_bwd_kernel<4>, dim3(wgs), dim3(256), lds, st, a);
    } else {
        if ((e = ensure_dynamic_lds(attr[2], reinterpret_cast<const void *>(tail_rows_bwd_kernel<16>), 160 * 1024)) != hipSuccess) return e;
        hipLaunchKernelGGL(tail_rows_bwd_kernel<16>, dim3(wgs), dim3(256), lds, st, a);
    }
    if ((e = hipGetLastError()) != hipSuccess) return e;
    ReduceBatchScope reductions;
    if ((e = launch_reduce_slices(a.sl_w, dw2, g.p * g.d, wgs, (size_t)g.p * g.d, accumulate, st)) != hipSuccess) return e;
    if (db2 != nullptr && (e = launch_reduce_slices(a.sl_b, db2, g.p, wgs, (size_t)g.p, accumulate, st)) != hipSuccess) return e;
    return reductions.flush(st);
}

}  // namespace aft
