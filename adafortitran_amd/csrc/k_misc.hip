// k_misc.hip -- the small VALU stages: channel adapter, patch embedding + linear_1 + positional
// table, the LinearEstimator, and the MSE metric.  None of them is MFMA-shaped (K = 1..42, 12, 24).
#include <algorithm>

#include "aft_internal.h"
#include "conv_device.h"
#include "pack_device.h"

namespace aft {

// ---------------------------------------------------------------------------------------------
// ChannelAdapter (reference src/models/blocks/channel_adaptivity.py:24-40,59-63): three MLPs
// Linear(1,h0)-ReLU-Linear(h0,h1)-ReLU-Linear(h1,h2) on the RAW snr / delay-spread / doppler
// scalars; output element k of encoder e goes to token k/2, feature 2e + k%2.  Computed once per
// frame (the reference recomputes it for the Re and Im pass with identical results).
// grid = (B, 3 encoders), block = 256.
// ---------------------------------------------------------------------------------------------
struct AdapterArgs {
    const float *cond[3];
    const float *w[3][3], *b[3][3];
    float *tokens6;
    int h0, h1, h2, tokens;
};

// one (frame b, encoder e) of the adapter; sm = h0 + h1 floats of LDS.
// The three layers depend on each other, but their WEIGHTS depend on nothing: every thread requests the weights of all its outputs
// (layer 1 rows, layer 2 rows for the default h1 = 42) before the first barrier, so the kernel waits for ONE round trip to memory
// instead of three -- it is latency-bound (a few hundred MACs per thread), and it sits in front of the whole forward.
__device__ __forceinline__ void adapter_body(const AdapterArgs &a, int b, int e, float *sm) {
    float *a0 = sm, *a1 = sm + a.h0;
    const int tid = threadIdx.x;
    const float x = a.cond[e][b];
    constexpr int kH0Max = 8, kPer2 = 3;          // register-resident fast path: h0 <= 8, h1 == 42, h2 <= 3 x 256
    const bool fast = a.h0 <= kH0Max && a.h1 == 42 && a.h2 <= kPer2 * 256 && a.h1 <= 256;   // (prologue launch: 12.5 -> 11.1 us)
    if (fast) {
        // requests: layer 0 (thread i < h0), layer 1 (thread i < h1: h0 weights), layer 2 (outputs tid, tid + 256, tid + 512: 21 float2 each)
        float w0 = 0.f, b0 = 0.f, w1[kH0Max], b1 = 0.f, b2[kPer2];
        f32x2 w2[kPer2][21];
        if (tid < a.h0) { w0 = a.w[e][0][tid]; b0 = a.b[e][0][tid]; }
#pragma unroll
        for (int k = 0; k < kH0Max; ++k) w1[k] = (tid < a.h1 && k < a.h0) ? a.w[e][1][tid * a.h0 + k] : 0.f;
        if (tid < a.h1) b1 = a.b[e][1][tid];
#pragma unroll
        for (int u = 0; u < kPer2; ++u) {
            const int i = tid + 256 * u;
            const bool ok = i < a.h2;
            const float *wr = a.w[e][2] + (size_t)(ok ? i : 0) * 42;
#pragma unroll
            for (int k = 0; k < 21; ++k) w2[u][k] = *reinterpret_cast<const f32x2 *>(wr + 2 * k);
            b2[u] = ok ? a.b[e][2][i] : 0.f;
        }
        if (tid < a.h0) a0[tid] = fmaxf(fmaf(w0, x, b0), 0.f);
        __syncthreads();
        if (tid < a.h1) {
            float acc = b1;
#pragma unroll
            for (int k = 0; k < kH0Max; ++k)
                if (k < a.h0) acc = fmaf(w1[k], a0[k], acc);
            a1[tid] = fmaxf(acc, 0.f);
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < kPer2; ++u) {
            const int i = tid + 256 * u;
            if (i >= a.h2) break;
            float acc = b2[u];
#pragma unroll
            for (int k = 0; k < 21; ++k) {
                acc = fmaf(w2[u][k][0], a1[2 * k], acc);
                acc = fmaf(w2[u][k][1], a1[2 * k + 1], acc);
            }
            a.tokens6[((size_t)b * a.tokens + (i >> 1)) * 6 + 2 * e + (i & 1)] = acc;
        }
        return;
    }
    for (int i = tid; i < a.h0; i += 256) a0[i] = fmaxf(fmaf(a.w[e][0][i], x, a.b[e][0][i]), 0.f);
    __syncthreads();
    for (int i = tid; i < a.h1; i += 256) {
        float acc = a.b[e][1][i];
        for (int k = 0; k < a.h0; ++k) acc = fmaf(a.w[e][1][i * a.h0 + k], a0[k], acc);
        a1[i] = fmaxf(acc, 0.f);
    }
    __syncthreads();
    for (int i = tid; i < a.h2; i += 256) {
        float acc = a.b[e][2][i];
        const float *wr = a.w[e][2] + (size_t)i * a.h1;
        for (int k = 0; k < a.h1; ++k) acc = fmaf(wr[k], a1[k], acc);
        a.tokens6[((size_t)b * a.tokens + (i >> 1)) * 6 + 2 * e + (i & 1)] = acc;
    }
}

__global__ __launch_bounds__(256) void adapter_kernel(const AdapterArgs a) {
    extern __shared__ float sm[];
    adapter_body(a, blockIdx.x, blockIdx.y, sm);
}

static AdapterArgs adapter_args(const aft_config &c, const WeightsDev &w, const float *snr, const float *ds,
                                const float *dop, float *tokens6) {
    AdapterArgs a{};
    a.cond[0] = snr; a.cond[1] = ds; a.cond[2] = dop;
    for (int e = 0; e < 3; ++e)
        for (int j = 0; j < 3; ++j) { a.w[e][j] = w.ada_w[e][j]; a.b[e][j] = w.ada_b[e][j]; }
    a.tokens6 = tokens6;
    a.h0 = c.hidden[0]; a.h1 = c.hidden[1]; a.h2 = c.hidden[2];
    a.tokens = (c.num_scs / c.patch_scs) * (c.num_symbols / c.patch_symbols);
    return a;
}

hipError_t launch_adapter(const aft_config &c, const WeightsDev &w, const float *snr, const float *ds,
                          const float *dop, float *tokens6, int batch, hipStream_t st) {
    const AdapterArgs a = adapter_args(c, w, snr, ds, dop, tokens6);
    hipLaunchKernelGGL(adapter_kernel, dim3(batch, 3), dim3(256), sizeof(float) * (a.h0 + a.h1), st, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Prologue of a whole forward: the two small jobs that depend on nothing the forward computes -- the channel adapter
// (3 B workgroups) and the re-lay of the encoder's GEMM weights into fragment order (8 d^2 L / 1024 workgroups) -- as ONE
// launch.  As two launches they cost 7.6 + 5.1 us of a 1.57-ms forward; together they take as long as the longer one,
// which makes the stateless entry point (weights re-packed on every call: the image can never be stale) as fast as a
// cached image.  Either part may be absent (FortiTran has no adapter; a caller-owned image needs no pack).
// ---------------------------------------------------------------------------------------------
struct PrologueArgs {
    AdapterArgs ad;
    float *packed;
    const float *pilots;       // upsampler product: planes[n][pix] = up_b[pix] + sum_k up_w[pix][k] pilot[n][k]
    float *up_planes;
    int adapter_blocks, pack_blocks, d, num_layers, split;
    int up_bx, npix, pf, nplanes;
    float *conv_frag;          // [2 stacks][22 quads][64 lanes][4]: conv2 / conv3 weights as 16x16x4 operand fragments (conv_device.h)
    int frag_blocks;
};

__global__ __launch_bounds__(256) void prologue_kernel(const WeightsDev w, const PrologueArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    int blk = blockIdx.x;
    if (blk < a.adapter_blocks) {            // workgroup-uniform: the barriers inside are safe
        adapter_body(a.ad, blk / 3, blk % 3, sm);
        return;
    }
    blk -= a.adapter_blocks;
    if (blk < a.pack_blocks) {
        pack_weights_vec(w, a.packed, a.d, 0, a.num_layers, a.split, (size_t)blk * 256 + threadIdx.x);
        return;
    }
    blk -= a.pack_blocks;
    if (blk < a.frag_blocks) {               // thread = (stack, quad, lane): one 16-byte piece of the image
        const int v = blk * 256 + threadIdx.x;
        constexpr int kPerStack = kFragFloats / 4;      // 16-byte pieces per stack: 22 x 64 operand quads + 40 of helper tables
        if (v < 2 * kPerStack) {
            const int stack = v / kPerStack, rem = v - stack * kPerStack;
            const float *const *cw = stack ? w.ref_w : w.enh_w;
            const float *const *cb = stack ? w.ref_b : w.enh_b;
            f32x4 o;
            if (rem < kFragQuads * 64) {
                const int quad = rem >> 6, lane = rem & 63;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) o[jj] = conv_frag16_entry(cw[1], cb[1], cw[2], cb[2], 4 * quad + jj, lane);
            } else {
                const int i0 = 4 * (rem - kFragQuads * 64), h = i0 / 80;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) o[jj] = conv_helper_entry(cw[0], cb[0], cw[3], cb[3], h, i0 - 80 * h + jj);
            }
            *reinterpret_cast<f32x4 *>(a.conv_frag + (size_t)v * 4) = o;
        }
        return;
    }
    blk -= a.frag_blocks;
    upsample_planes_body(sm, w.up_w, w.up_b, a.pilots, a.up_planes, a.npix, a.pf, a.nplanes, blk % a.up_bx, blk / a.up_bx);
}

bool prologue_upsample_ok(const aft_config &c, const WeightsDev &w) {
    return upsample_planes_ok(w.up_w, c.pilot_scs * c.pilot_symbols);
}

hipError_t launch_prologue(const aft_config &c, const WeightsDev &w, const float *snr, const float *ds, const float *dop,
                           float *tokens6, int batch, float *packed, const float *pilots, float *up_planes, hipStream_t st,
                           float *conv_frag) {
    PrologueArgs a{};
    if (conv_frag != nullptr) {
        a.conv_frag = conv_frag;
        a.frag_blocks = (2 * (kFragFloats / 4) + 255) / 256;
    }
    size_t lds = 0;
    if (c.adaptive) {
        a.ad = adapter_args(c, w, snr, ds, dop, tokens6);
        a.adapter_blocks = 3 * batch;
        lds = sizeof(float) * (a.ad.h0 + a.ad.h1);
    }
    if (packed != nullptr) {
        a.packed = packed; a.d = c.model_dim; a.num_layers = std::min(c.num_layers, kLayerWindow);   // the window `w` holds; the caller packs the rest
        a.split = c.precision == AFT_PRECISION_BF16X3 ? 1 : 0;
        a.pack_blocks = (int)((packed_layer_floats(c.model_dim) * a.num_layers / 4 + 255) / 256);
    }
    int up_blocks = 0;
    if (up_planes != nullptr && prologue_upsample_ok(c, w)) {
        a.pilots = pilots; a.up_planes = up_planes;
        a.npix = c.num_scs * c.num_symbols; a.pf = c.pilot_scs * c.pilot_symbols; a.nplanes = 2 * batch;
        a.up_bx = (a.npix + kUpPix - 1) / kUpPix;
        up_blocks = a.up_bx * ((a.nplanes + kUpPlanes - 1) / kUpPlanes);
        lds = std::max(lds, upsample_planes_lds(a.pf));
    }
    const int blocks = a.adapter_blocks + a.pack_blocks + a.frag_blocks + up_blocks;
    if (blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(prologue_kernel, dim3(blocks), dim3(256), lds, st, w, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// S3 + S4-concat + linear_1 + positional table (reference fortitran.py:212-217, encoders.py:67-68,
// patch_processors.py:22,34-35, positional_encodings.py:38,64):
//   x[n, t, :] = W1 [patch(n,t) ; tokens6(frame,t)] + b1 + pos[t, :]
// The patch gather is pure addressing: feature f of token t is grid element
// (sc, sym) = ((t / tw)*p0 + f / p1, (t % tw)*p1 + f % p1).
// block = 256 threads = 8 tokens x 32 lanes; each lane produces d/32 consecutive outputs.
// ---------------------------------------------------------------------------------------------
struct EmbedArgs {
    const float *conv_enhanced, *tokens6, *w1, *b1, *pos;
    float *x;
    int S, T, p0, p1, tokens, d, din, planes;
};

// PMAX = unroll bound of the patch features: 6 (the default 3x2 patch and smaller) or 16.
// Half a wave (32 lanes) owns one token at a time and walks tokens with a grid stride; lane l produces the
// D/32 consecutive outputs l*D/32.. of its token.  The lane's slice of W1 ((p + 6) x D/32 values) and of
// the bias stays in registers for all its tokens, so per token there are only the 6..12 broadcast input
// loads, one 16-byte positional load and one 16-byte store: the kernel is bound by writing x.
template <int D, bool ADAPTIVE, int PMAX>
__global__ __launch_bounds__(256) void embed_kernel(const EmbedArgs a) {
    constexpr int PER = D / 32, NF = ADAPTIVE ? PMAX + 6 : PMAX;
    const int tid = threadIdx.x, hw = tid >> 5, lane32 = tid & 31;
    const int tw = a.T / a.p1, p = a.p0 * a.p1;
    extern __shared__ float sm[];   // W1 transposed to [din][D]: staged once per workgroup with coalesced loads
    for (int i = tid; i < a.din * D; i += 256) sm[(i % a.din) * D + i / a.din] = a.w1[i];
    __syncthreads();
    float w[NF][PER], b1[PER];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const int frow = f < PMAX ? f : p + (f - PMAX);   // column of W1 [D][din]: patch features then adapter features
#pragma unroll
        for (int i = 0; i < PER; ++i) w[f][i] = (f < PMAX && f >= p) ? 0.f : sm[frow * D + lane32 * PER + i];
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) b1[i] = a.b1[lane32 * PER + i];
    // 32-bit token bookkeeping (the forward's size guard keeps rows * D below 2^31); the patch-element
    // offsets are fixed per thread, (plane, token) advance incrementally: no division by a run-time value
    // per element -- those were most of this kernel's instructions
    const unsigned rows = (unsigned)a.planes * a.tokens, rstep = gridDim.x * 8u, tokens = a.tokens;
    const unsigned dn = rstep / tokens, dt = rstep % tokens;
    int poff[PMAX];
#pragma unroll
    for (int f = 0; f < PMAX; ++f) poff[f] = (f / a.p1) * a.T + f % a.p1;
    struct Tok { unsigned row, n, t; };
    auto advance = [&](Tok k) {
        k.row += rstep; k.n += dn; k.t += dt;
        if (k.t >= tokens) { k.t -= tokens; ++k.n; }
        return k;
    };
    // inputs of a token (its patch elements, adapter features and positional slice); fetched one token
    // ahead so that their latency hides under the previous token's FMAs and store
    auto fetch = [&](const Tok &k, float (&vin)[NF], float (&ps)[PER]) {
        const unsigned tq = k.t / (unsigned)tw, tr = k.t - tq * tw;
        const float *ce = a.conv_enhanced + ((size_t)k.n * a.S + tq * a.p0) * a.T + tr * a.p1;
#pragma unroll
        for (int f = 0; f < PMAX; ++f) vin[f] = f < p ? ce[poff[f]] : 0.f;
        if constexpr (ADAPTIVE) {
            const float *tk = a.tokens6 + ((size_t)(k.n >> 1) * tokens + k.t) * 6;
#pragma unroll
            for (int f = 0; f < 6; ++f) vin[PMAX + f] = tk[f];
        }
#pragma unroll
        for (int i = 0; i < PER; ++i) ps[i] = a.pos[k.t * D + lane32 * PER + i];
    };
    auto emit = [&](const Tok &k, const float (&vin)[NF], const float (&ps)[PER]) {
        float acc[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) acc[i] = b1[i] + ps[i];
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int i = 0; i < PER; ++i) acc[i] = fmaf(w[f][i], vin[f], acc[i]);
#pragma unroll
        for (int i = 0; i < PER; ++i) a.x[(size_t)k.row * D + lane32 * PER + i] = acc[i];
    };
    float va[NF], pa[PER], vb[NF], pb[PER];
    Tok k0;
    k0.row = blockIdx.x * 8u + hw;
    k0.n = k0.row / tokens;
    k0.t = k0.row - k0.n * tokens;
    if (k0.row < rows) fetch(k0, va, pa);
    while (k0.row < rows) {
        const Tok k1 = advance(k0), k2 = advance(k1);
        if (k1.row < rows) fetch(k1, vb, pb);
        emit(k0, va, pa);
        if (k1.row < rows) {
            if (k2.row < rows) fetch(k2, va, pa);
            emit(k1, vb, pb);
        }
        k0 = k2;
    }
}

hipError_t launch_embed(const aft_config &c, const WeightsDev &w, const float *conv_enhanced, const float *tokens6,
                        float *x, int batch, hipStream_t st) {
    EmbedArgs a{};
    a.conv_enhanced = conv_enhanced; a.tokens6 = tokens6;
    a.w1 = w.lin1_w; a.b1 = w.lin1_b; a.pos = w.pos; a.x = x;
    a.S = c.num_scs; a.T = c.num_symbols; a.p0 = c.patch_scs; a.p1 = c.patch_symbols;
    a.tokens = (a.S / a.p0) * (a.T / a.p1);
    a.d = c.model_dim;
    a.din = a.p0 * a.p1 + (c.adaptive ? 6 : 0);
    a.planes = 2 * batch;
    const long rows = (long)a.planes * a.tokens;
    const int blocks = (int)std::min<long>((rows + 7) / 8, 1024);   // swept 256..4096 at B = 128: 4 waves per SIMD, ~9 tokens per half wave
    const size_t lds = sizeof(float) * a.din * a.d;
    const bool small = a.p0 * a.p1 <= 6;
    if (a.p0 * a.p1 > kMaxPatchFeatures) return hipErrorInvalidValue;   // check_config rejects these
#define AFT_EMBED(D_, A_)                                                                              \
    do {                                                                                               \
        if (small) hipLaunchKernelGGL((embed_kernel<D_, A_, 6>), dim3(blocks), dim3(256), lds, st, a); \
        else hipLaunchKernelGGL((embed_kernel<D_, A_, kMaxPatchFeatures>), dim3(blocks), dim3(256), lds, st, a); \
    } while (0)
    if (c.model_dim == 32 && c.adaptive) AFT_EMBED(32, true);
    else if (c.model_dim == 32) AFT_EMBED(32, false);
    else if (c.model_dim == 96 && c.adaptive) AFT_EMBED(96, true);
    else if (c.model_dim == 96) AFT_EMBED(96, false);
    else if (c.model_dim == 160 && c.adaptive) AFT_EMBED(160, true);
    else if (c.model_dim == 160) AFT_EMBED(160, false);
    else if (c.model_dim == 224 && c.adaptive) AFT_EMBED(224, true);
    else if (c.model_dim == 224) AFT_EMBED(224, false);
    else if (c.model_dim == 64 && c.adaptive) AFT_EMBED(64, true);
    else if (c.model_dim == 64) AFT_EMBED(64, false);
    else if (c.model_dim == 128 && c.adaptive) AFT_EMBED(128, true);
    else if (c.model_dim == 128) AFT_EMBED(128, false);
    else if (c.model_dim == 192 && c.adaptive) AFT_EMBED(192, true);
    else if (c.model_dim == 192) AFT_EMBED(192, false);
    else if (c.model_dim == 256 && c.adaptive) AFT_EMBED(256, true);
    else if (c.model_dim == 256) AFT_EMBED(256, false);
    else return hipErrorInvalidValue;
#undef AFT_EMBED
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// LinearEstimator (reference src/models/linear.py:65-97) applied plane-wise to complex pilots:
// out[b, o, c] = sum_k W[o, k] x[b, k, c] + bias[o]  for c in {Re, Im}.
// one thread per complex output element; x row staged in LDS.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void linear_kernel(const float *__restrict__ w, const float *__restrict__ bias,
                                                     const float *__restrict__ pilots, float *__restrict__ out,
                                                     int in_f, int out_f) {
    extern __shared__ float sm[];  // [in_f][2]
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < 2 * in_f; i += 256) sm[i] = pilots[(size_t)b * 2 * in_f + i];
    __syncthreads();
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= out_f) return;
    const float bv = bias != nullptr ? bias[o] : 0.f;
    float re = bv, im = bv;
    const float *wr = w + (size_t)o * in_f;
    for (int k = 0; k < in_f; ++k) {
        re = fmaf(wr[k], sm[2 * k], re);
        im = fmaf(wr[k], sm[2 * k + 1], im);
    }
    *reinterpret_cast<float2 *>(out + ((size_t)b * out_f + o) * 2) = make_float2(re, im);
}

hipError_t launch_linear(const float *weight, const float *bias, const float *pilots, float *out, int batch,
                         int in_features, int out_features, hipStream_t st) {
    hipLaunchKernelGGL(linear_kernel, dim3((out_features + 255) / 256, batch), dim3(256),
                       sizeof(float) * 2 * in_features, st, weight, bias, pilots, out, in_features, out_features);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Metric (reference src/utils.py:164-180 + src/main/trainer.py:338-347): 2*MSELoss(cat(Re,Im))
// == mean over complex elements of |est - ref|^2.  This kernel accumulates the SUM in float64
// (one atomic per workgroup); the caller divides by the element count (and all-gathers across
// ranks).  HBM-bound: 16 B per complex element, float4 loads.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mse_kernel(const float *__restrict__ est, const float *__restrict__ ref,
                                                  double *sum_sq, long long n_floats) {
    double acc = 0.0;
    const long long n4 = n_floats / 4;
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const f32x4 e = reinterpret_cast<const f32x4 *>(est)[i];
        const f32x4 r = reinterpret_cast<const f32x4 *>(ref)[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float dlt = e[j] - r[j];
            acc += (double)dlt * (double)dlt;
        }
    }
    if (blockIdx.x == 0)
        for (long long i = n4 * 4 + threadIdx.x; i < n_floats; i += 256) {
            const float dlt = est[i] - ref[i];
            acc += (double)dlt * (double)dlt;
        }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
    __shared__ double part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(sum_sq, part[0] + part[1] + part[2] + part[3]);
}

hipError_t launch_mse(const float *est, const float *ref, double *sum_sq, long long n_complex, hipStream_t st) {
    const long long n_floats = 2 * n_complex;
    long long blocks = (n_floats / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 128) blocks = 128;   // one float64 atomic per workgroup on ONE address: few, fat workgroups
    hipLaunchKernelGGL(mse_kernel, dim3((int)blocks), dim3(256), 0, st, est, ref, sum_sq, n_floats);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Pilot extraction (reference src/data/dataset.py:116-139): per frame, compact the non-zero complex
// entries of the sparse LS grid in row-major order.  One wave per frame; order is kept with a
// ballot + prefix popcount per 64-element step (no atomics, deterministic).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pilot_gather_kernel(const float2 *__restrict__ grid, float2 *__restrict__ pilots,
                                                           int *__restrict__ counts, int batch, int n, int expected) {
    const int lane = threadIdx.x & 63;
    const int frame = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (frame >= batch) return;
    const float2 *src = grid + (size_t)frame * n;
    int base = 0;
    for (int i0 = 0; i0 < n; i0 += 8 * 64) {   // eight 512-byte loads in flight per wave, then the ordered compaction
        float2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + 64 * u + lane;
            v[u] = i < n ? src[i] : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool nz = v[u].x != 0.f || v[u].y != 0.f;   // complex != 0, as torch compares it (padding reads as 0)
            const unsigned long long m = __ballot(nz);
            const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
            if (nz && pos < expected) pilots[(size_t)frame * expected + pos] = v[u];
            base += __popcll(m);
        }
    }
    for (int pos = base + lane; pos < expected; pos += 64)            // fewer non-zero entries than expected: the rest reads as zero
        pilots[(size_t)frame * expected + pos] = make_float2(0.f, 0.f);   // (the caller need not zero-fill the output: one launch less)
    if (lane == 0) counts[frame] = base;
}

hipError_t launch_pilot_gather(const float *hzero_ls, float *pilots, int *counts, int batch, int grid_elems,
                               int expected, hipStream_t st) {
    hipLaunchKernelGGL(pilot_gather_kernel, dim3((batch + 3) / 4), dim3(256), 0, st,
                       reinterpret_cast<const float2 *>(hzero_ls), reinterpret_cast<float2 *>(pilots), counts, batch,
                       grid_elems, expected);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// The same stage (S3 + S4-concat + linear_1 + positions) for the general engine: ANY model_dim, patches of up to 32 elements.
// A workgroup takes 8 token rows: their <= 38 input features go to LDS once, thread c' walks the output columns c = c', c' + 256, ..
// with row c of W1 ([d][din], as PyTorch holds it) in registers across the 8 rows.  x[row][c] = b1[c] + pos[t][c] + sum_f W1[c][f] in[f],
// summed in feature order.  Write-bound like embed_kernel; a few microseconds of a general-engine forward.
// ---------------------------------------------------------------------------------------------
constexpr int kEmbAnyRows = 8, kEmbAnyDin = kMaxPatchGeneral + 6;
__global__ __launch_bounds__(256) void embed_any_kernel(const EmbedArgs a) {
    __shared__ float in[kEmbAnyRows][kEmbAnyDin];
    const int tid = threadIdx.x, tw = a.T / a.p1, p = a.p0 * a.p1;
    const long rows = (long)a.planes * a.tokens, row0 = (long)blockIdx.x * kEmbAnyRows;
    // every slot of `in` is written: the products below run over all kEmbAnyDin slots with zero weights beyond din, and 0 x whatever
    // the LDS held (a NaN bit pattern left by another kernel) is not 0 (caught by the module-surface tests, where the preceding kernels differ)
    for (int i = tid; i < kEmbAnyRows * kEmbAnyDin; i += 256) {
        const int r = i / kEmbAnyDin, f = i - r * kEmbAnyDin;
        const long row = row0 + r;
        float v = 0.f;
#ifdef AFT_TEST_LDS_BUG     // round 6's bug behind a flag (slots beyond din left unwritten): tools/debug/lds_fill_check.py proves that the
        if (f >= a.din) continue;   // LDS fill of aft_debug_fill_lds_f32 reaches it -- never defined in the product or the checked build
#endif
        if (row < rows && f < a.din) {
            const int n = (int)(row / a.tokens), t = (int)(row - (long)n * a.tokens);
            if (f < p) {
                const int tq = t / tw, tr = t - tq * tw;
                v = a.conv_enhanced[((size_t)n * a.S + tq * a.p0 + f / a.p1) * a.T + tr * a.p1 + f % a.p1];
            } else {
                v = a.tokens6[((size_t)(n >> 1) * a.tokens + t) * 6 + (f - p)];
            }
        }
        in[r][f] = v;
    }
    __syncthreads();
    for (int c = tid; c < a.d; c += 256) {
        float w[kEmbAnyDin];
#pragma unroll
        for (int f = 0; f < kEmbAnyDin; ++f) w[f] = f < a.din ? a.w1[(size_t)c * a.din + f] : 0.f;
        const float b = a.b1[c];
        for (int r = 0; r < kEmbAnyRows; ++r) {
            const long row = row0 + r;
            if (row >= rows) break;
            const int t = (int)(row % a.tokens);
            float acc = b + a.pos[(size_t)t * a.d + c];
#pragma unroll
            for (int f = 0; f < kEmbAnyDin; ++f) acc = fmaf(w[f], in[r][f], acc);
            a.x[(size_t)row * a.d + c] = acc;
        }
    }
}

hipError_t launch_embed_any(const aft_config &c, const WeightsDev &w, const float *conv_enhanced, const float *tokens6,
                            float *x, int batch, hipStream_t st) {
    EmbedArgs a{};
    a.conv_enhanced = conv_enhanced; a.tokens6 = tokens6;
    a.w1 = w.lin1_w; a.b1 = w.lin1_b; a.pos = w.pos; a.x = x;
    a.S = c.num_scs; a.T = c.num_symbols; a.p0 = c.patch_scs; a.p1 = c.patch_symbols;
    a.tokens = (a.S / a.p0) * (a.T / a.p1);
    a.d = c.model_dim;
    a.din = a.p0 * a.p1 + (c.adaptive ? 6 : 0);
    a.planes = 2 * batch;
    if (a.p0 * a.p1 > kMaxPatchGeneral || (c.adaptive && tokens6 == nullptr)) return hipErrorInvalidValue;
    // Round 6 (late): the training path's embedding kernel (k_ends_train.hip: persistent workgroups, W1^T staged once per workgroup,
    // 32-row tiles) computes the same sums in the same order -- bias + position first, then the features ascending -- and does not
    // re-read W1's rows per 8 token rows (embed_any_kernel: 326 us at d = 512 / 128 frames, 0.45 TB/s of a write-bound stage).
    // AFT_EMBED_ANY_OLD keeps the kernel below (A/B, tests).
    if (!switch_on("AFT_EMBED_ANY_OLD") && ends_train_ok(a.planes, a.S, a.T, a.p0, a.p1, a.d, c.adaptive != 0))
        return launch_embed_train_fwd(conv_enhanced, c.adaptive ? tokens6 : nullptr, w.lin1_w, w.lin1_b, w.pos, x, a.planes, a.S, a.T, a.p0,
                                      a.p1, a.d, st, true);
    const long rows = (long)a.planes * a.tokens;
    hipLaunchKernelGGL(embed_any_kernel, dim3((unsigned)((rows + kEmbAnyRows - 1) / kEmbAnyRows)), dim3(256), 0, st, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Test hook: every CU's LDS filled with one value (aft_debug_fill_lds_f32).  A workgroup allocates the whole 160 KB, so one is resident
// per CU at a time; each spins a few microseconds after its fill so that the dispatcher hands the later workgroups to the other CUs.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void fill_lds_kernel(float value, float *sink) {
    extern __shared__ __attribute__((aligned(16))) float lds_all[];
    constexpr int kFloats = 160 * 1024 / 4;
    for (int i = threadIdx.x; i < kFloats; i += 1024) lds_all[i] = value;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < 20000) __builtin_amdgcn_s_sleep(16);
    if (sink != nullptr && lds_all[(threadIdx.x * 977) % kFloats] == 12345.678f) *sink = 1.f;   // keeps the stores alive
}
__global__ __launch_bounds__(256) void peek_lds_kernel(float *__restrict__ out, int n) {
    extern __shared__ __attribute__((aligned(16))) float lds_peek[];
    for (int i = threadIdx.x; i < n; i += 256) out[(size_t)blockIdx.x * n + i] = lds_peek[i];
}
hipError_t launch_peek_lds(float *out, int workgroups, int n, hipStream_t st) {
    hipLaunchKernelGGL(peek_lds_kernel, dim3(workgroups), dim3(256), 40 * 1024, st, out, n);
    return hipGetLastError();
}
hipError_t launch_fill_lds(float value, hipStream_t st) {
    static PerDeviceOnce attr;
    hipError_t e = ensure_dynamic_lds(attr, reinterpret_cast<const void *>(fill_lds_kernel), 160 * 1024);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fill_lds_kernel, dim3(4 * current_device_cus()), dim3(1024), 160 * 1024, st, value, (float *)nullptr);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// LS-baseline metric (reference src/utils.py:248-261, applied per file by get_ls_mse_per_folder
// :264-303): db[b] = 10 log10(mean |ls - ideal|^2).  One wave per frame, float64 accumulation.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ls_mse_db_kernel(const float2 *__restrict__ ls, const float2 *__restrict__ ideal,
                                                        float *__restrict__ db, int batch, int n) {
    const int lane = threadIdx.x & 63;
    const int frame = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (frame >= batch) return;
    double acc = 0.0;
    for (int i = lane; i < n; i += 64) {
        const float2 a = ls[(size_t)frame * n + i], b = ideal[(size_t)frame * n + i];
        const double dr = (double)a.x - (double)b.x, di = (double)a.y - (double)b.y;
        acc += dr * dr + di * di;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) db[frame] = (float)(10.0 * log10(acc / (double)n));
}

hipError_t launch_ls_mse_db(const float *ls, const float *ideal, float *db, int batch, int grid_elems, hipStream_t st) {
    hipLaunchKernelGGL(ls_mse_db_kernel, dim3((batch + 3) / 4), dim3(256), 0, st, reinterpret_cast<const float2 *>(ls),
                       reinterpret_cast<const float2 *>(ideal), db, batch, grid_elems);
    return hipGetLastError();
}

}  // namespace aft
