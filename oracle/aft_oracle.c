/*
 * aft_oracle.c -- CPU restatement of the AdaFortiTran / FortiTran forward path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle for the HIP path: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The
 * product (adafortitran_amd/) never links, imports or falls back to it.
 *
 * Pinning: the reference ships no tests or golden vectors, and
 * the arithmetic lives in third-party torch (unpinned in requirements.txt; the pin
 * used here is torch 2.10.0+rocm7.0 CPU, oneDNN 3.7.1 / MKL 2024.2).  The oracle is
 * pinned against OUTPUTS OF THE REFERENCE ITSELF, imported in the build container by
 * tests/golden/make_golden.py, and committed as the .npz files under tests/golden/
 * (tests/test_oracle_golden.py checks every stage of the tiny set and the end-to-end
 * default sets).
 *
 * Numerics: tensors are stored in IEEE float32 between ops exactly as the reference
 * does; dot products accumulate in double and round once, so the oracle sits inside
 * the reference's own fp32 noise floor (torch fused vs unfused encoder paths differ
 * by 4e-7) while being independent of summation order.
 *
 * Every function cites the reference file:line it restates (paths relative to the
 * reference root).  Plane n = 2*frame + (0 = Re, 1 = Im): the reference runs the
 * real-valued network once on .real and once on .imag with shared weights
 * (src/models/fortitran.py:176-177).
 */
#include "../include/adafortitran_amd.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct aft_oracle_dump {
    float *upsampled;     /* [2B,S,T]       after pilot_upsampler (fortitran.py:203)         */
    float *conv_enhanced; /* [2B,S,T]       after initial_enhancer (fortitran.py:209)        */
    float *tokens6;       /* [B,tokens,6]   ChannelAdapter output (channel_adaptivity.py:63) */
    float *embed_in;      /* [2B,tokens,din] transformer_input (fortitran.py:217-219)        */
    float *x0;            /* [2B,tokens,d]  after linear_1 + pos (encoders.py:67-68)         */
    float *layer_out;     /* [L,2B,tokens,d] after each encoder layer (encoders.py:69)       */
    float *enc_out;       /* [2B,tokens,p]  after linear_2 (encoders.py:70)                  */
    float *residual;      /* [2B,S,T]       conv_enhanced + reconstructed (fortitran.py:228) */
} aft_oracle_dump;

static int tokens_of(const aft_config *c) {
    return (c->num_scs / c->patch_scs) * (c->num_symbols / c->patch_symbols);
}

/* y[m,n] = sum_k x[m,k] W[n,k] + b[n]   -- nn.Linear (torch layout W[out,in]);
 * call sites: fortitran.py:86,203; encoders.py:36,56,67,70; channel_adaptivity.py:35-39;
 * linear.py:62,90. */
static void linear_f32(const float *x, const float *W, const float *b, int M, int K, int N,
                       float *y) {
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            double acc = b ? (double)b[n] : 0.0;
            const float *xr = x + (size_t)m * K, *wr = W + (size_t)n * K;
            for (int k = 0; k < K; ++k) acc += (double)xr[k] * (double)wr[k];
            y[(size_t)m * N + n] = (float)acc;
        }
}

/* nn.Conv2d(cin,cout,3,padding=1) (+ optional ReLU): cross-correlation, zero padding,
 * weight [cout,cin,3,3] -- blocks/enhancers.py:13-19. */
static void conv3x3_f32(const float *in, int cin, int cout, int H, int W, const float *w,
                        const float *b, int relu, float *out) {
    for (int co = 0; co < cout; ++co)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                double acc = (double)b[co];
                for (int ci = 0; ci < cin; ++ci)
                    for (int dy = -1; dy <= 1; ++dy) {
                        int yy = y + dy;
                        if (yy < 0 || yy >= H) continue;
                        for (int dx = -1; dx <= 1; ++dx) {
                            int xx = x + dx;
                            if (xx < 0 || xx >= W) continue;
                            acc += (double)in[((size_t)ci * H + yy) * W + xx] *
                                   (double)w[(((size_t)co * cin + ci) * 3 + (dy + 1)) * 3 + (dx + 1)];
                        }
                    }
                float v = (float)acc;
                out[((size_t)co * H + y) * W + x] = (relu && v < 0.f) ? 0.f : v;
            }
}

/* ConvEnhancer: 1->8->32->8->1, ReLU after the first three -- blocks/enhancers.py:5-31. */
static void conv_enhancer_f32(const float *in, int H, int W, const float *const w[4],
                              const float *const b[4], float *out, float *tmpA, float *tmpB) {
    conv3x3_f32(in, 1, 8, H, W, w[0], b[0], 1, tmpA);
    conv3x3_f32(tmpA, 8, 32, H, W, w[1], b[1], 1, tmpB);
    conv3x3_f32(tmpB, 32, 8, H, W, w[2], b[2], 1, tmpA);
    conv3x3_f32(tmpA, 8, 1, H, W, w[3], b[3], 0, out);
}

/* PatchEmbedding = nn.Unfold(kernel=stride=patch) + permute -- blocks/patch_processors.py:22,34-35:
 * token t = (sc/p0)*(T/p1) + sym/p1 , feature f = (sc%p0)*p1 + sym%p1. */
static void patchify_f32(const float *plane, int S, int T, int p0, int p1, float *tok, int stride) {
    int tw = T / p1;
    for (int sc = 0; sc < S; ++sc)
        for (int sym = 0; sym < T; ++sym) {
            int t = (sc / p0) * tw + sym / p1, f = (sc % p0) * p1 + sym % p1;
            tok[(size_t)t * stride + f] = plane[(size_t)sc * T + sym];
        }
}

/* InversePatchEmbedding = permute + nn.Fold (non-overlapping, nothing is summed)
 * -- blocks/patch_processors.py:53-57,69-71. */
static void unpatchify_f32(const float *tok, int S, int T, int p0, int p1, float *plane) {
    int tw = T / p1, p = p0 * p1;
    for (int sc = 0; sc < S; ++sc)
        for (int sym = 0; sym < T; ++sym) {
            int t = (sc / p0) * tw + sym / p1, f = (sc % p0) * p1 + sym % p1;
            plane[(size_t)sc * T + sym] = tok[(size_t)t * p + f];
        }
}

/* ChannelAdapter: three MLPs Linear(1,h0)-ReLU-Linear(h0,h1)-ReLU-Linear(h1,h2); each output
 * reshaped (-1,2) so token t takes elements [2t,2t+1]; concatenated in the order snr, ds, dop
 * -- blocks/channel_adaptivity.py:24-40,59-63.  cond = raw un-normalised scalars (B14). */
static void adapter_one_frame(const aft_config *c, const aft_weights *w, const float cond[3],
                              float *tok6) {
    int h0 = c->hidden[0], h1 = c->hidden[1], h2 = c->hidden[2];
    float *a = (float *)malloc(sizeof(float) * (size_t)(h0 + h1 + h2));
    float *bb = a + h0, *cc = bb + h1;
    for (int e = 0; e < 3; ++e) {
        linear_f32(&cond[e], w->ada_w[e][0], w->ada_b[e][0], 1, 1, h0, a);
        for (int i = 0; i < h0; ++i) a[i] = a[i] < 0.f ? 0.f : a[i];
        linear_f32(a, w->ada_w[e][1], w->ada_b[e][1], 1, h0, h1, bb);
        for (int i = 0; i < h1; ++i) bb[i] = bb[i] < 0.f ? 0.f : bb[i];
        linear_f32(bb, w->ada_w[e][2], w->ada_b[e][2], 1, h1, h2, cc);
        for (int k = 0; k < h2; ++k) tok6[(size_t)(k / 2) * 6 + 2 * e + (k % 2)] = cc[k];
    }
    free(a);
}

/* nn.LayerNorm(d, eps=1e-5): biased variance, affine -- used by nn.TransformerEncoderLayer. */
static void layernorm_row(const float *x, const float *g, const float *b, int d, float *y) {
    double mean = 0.0, var = 0.0;
    for (int i = 0; i < d; ++i) mean += (double)x[i];
    mean /= d;
    for (int i = 0; i < d; ++i) {
        double t = (double)x[i] - mean;
        var += t * t;
    }
    var /= d;
    double rstd = 1.0 / sqrt(var + 1e-5);
    for (int i = 0; i < d; ++i)
        y[i] = (float)((((double)x[i] - mean) * rstd) * (double)g[i] + (double)b[i]);
}

static float act_f32(float v, int activation) {
    if (activation == AFT_ACT_GELU) /* exact erf GELU: F.gelu(approximate="none") */
        return (float)(0.5 * (double)v * (1.0 + erf((double)v * 0.70710678118654752440)));
    return v < 0.f ? 0.f : v;
}

/* One nn.TransformerEncoderLayer, norm_first=False, eval mode (dropout = identity), no mask:
 *   x = LN1(x + out_proj(MHA(x)));  x = LN2(x + W2 act(W1 x + b1) + b2)
 * MHA: packed in_proj [Wq;Wk;Wv], heads = contiguous dh-wide column slices, logits scaled by
 * 1/sqrt(dh), softmax over keys.  Constructed at blocks/encoders.py:44-55 (dim_feedforward =
 * 2*model_dim :47); semantics per torch/nn/modules/transformer.py, probed in SURVEY.md 3.3. */
static void encoder_layer_plane(const aft_config *c, const aft_layer_weights *lw, float *x,
                                int tokens, float *qkv, float *att, float *hid, float *prob) {
    int d = c->model_dim, H = c->num_head, dh = d / H, ff = 2 * d;
    double scale = 1.0 / sqrt((double)dh);
    linear_f32(x, lw->in_proj_w, lw->in_proj_b, tokens, d, 3 * d, qkv);
    for (int h = 0; h < H; ++h)
        for (int i = 0; i < tokens; ++i) {
            const float *q = qkv + (size_t)i * 3 * d + h * dh;
            double mx = -INFINITY;
            for (int j = 0; j < tokens; ++j) {
                const float *k = qkv + (size_t)j * 3 * d + d + h * dh;
                double s = 0.0;
                for (int e = 0; e < dh; ++e) s += (double)q[e] * (double)k[e];
                float sf = (float)(s * scale);
                prob[j] = sf;
                if (sf > mx) mx = sf;
            }
            double den = 0.0;
            for (int j = 0; j < tokens; ++j) {
                double ex = exp((double)prob[j] - mx);
                prob[j] = (float)ex;
                den += ex;
            }
            for (int e = 0; e < dh; ++e) {
                double o = 0.0;
                for (int j = 0; j < tokens; ++j)
                    o += (double)prob[j] * (double)qkv[(size_t)j * 3 * d + 2 * d + h * dh + e];
                att[(size_t)i * d + h * dh + e] = (float)(o / den);
            }
        }
    for (int i = 0; i < tokens; ++i) {
        float *xr = x + (size_t)i * d;
        float *tmp = hid; /* first d entries reused as the pre-LN row */
        linear_f32(att + (size_t)i * d, lw->out_proj_w, lw->out_proj_b, 1, d, d, tmp);
        for (int e = 0; e < d; ++e) tmp[e] = tmp[e] + xr[e];
        layernorm_row(tmp, lw->norm1_w, lw->norm1_b, d, xr);
        float *h1 = hid, *h2 = hid + ff;
        linear_f32(xr, lw->lin1_w, lw->lin1_b, 1, d, ff, h1);
        for (int e = 0; e < ff; ++e) h1[e] = act_f32(h1[e], c->activation);
        linear_f32(h1, lw->lin2_w, lw->lin2_b, 1, ff, d, h2);
        for (int e = 0; e < d; ++e) h2[e] = h2[e] + xr[e];
        layernorm_row(h2, lw->norm2_w, lw->norm2_b, d, xr);
    }
}

static int check_cfg(const aft_config *c) {
    if (!c) return AFT_ERR_ARG;
    if (c->num_scs <= 0 || c->num_symbols <= 0 || c->patch_scs <= 0 || c->patch_symbols <= 0 ||
        c->num_scs % c->patch_scs || c->num_symbols % c->patch_symbols)
        return AFT_ERR_SHAPE;
    if (c->num_layers <= 0 || c->model_dim <= 0 ||
        c->num_head <= 0 || c->model_dim % c->num_head)
        return AFT_ERR_SHAPE;
    if (c->adaptive && (c->hidden[2] != 2 * tokens_of(c) || c->hidden[0] <= 0 || c->hidden[1] <= 0))
        return AFT_ERR_SHAPE;
    return AFT_OK;
}

/* ChannelAdapter for a batch -- fortitran.py:216 (the reference evaluates it in both the Re
 * and the Im pass with identical results; once per frame is equivalent). */
int aft_oracle_adapter_f32(const aft_config *c, const aft_weights *w, const float *snr,
                           const float *ds, const float *dop, float *tokens6, int batch) {
    int rc = check_cfg(c);
    if (rc) return rc;
    if (!c->adaptive || !snr || !ds || !dop || !tokens6) return AFT_ERR_ARG;
    int tokens = tokens_of(c);
#pragma omp parallel for schedule(static)
    for (int b = 0; b < batch; ++b) {
        float cond[3] = {snr[b], ds[b], dop[b]};
        adapter_one_frame(c, w, cond, tokens6 + (size_t)b * tokens * 6);
    }
    return AFT_OK;
}

/* BaseFortiTranEstimator.forward + _forward_real_valued -- fortitran.py:145-233.
 * pilots: complex64 [B,Ps,Pt] interleaved; out: complex64 [B,S,T] interleaved. */
int aft_oracle_forward_f32(const aft_config *c, const aft_weights *w, const float *pilots,
                           const float *snr, const float *ds, const float *dop, float *out,
                           int batch, const aft_oracle_dump *dump) {
    int rc = check_cfg(c);
    if (rc) return rc;
    if (!w || !pilots || !out || batch <= 0) return AFT_ERR_ARG;
    if (c->adaptive && (!snr || !ds || !dop)) return AFT_ERR_ARG; /* fortitran.py:157-158 */
    const int S = c->num_scs, T = c->num_symbols, ST = S * T;
    const int PF = c->pilot_scs * c->pilot_symbols;
    const int p0 = c->patch_scs, p1 = c->patch_symbols, p = p0 * p1;
    const int tokens = tokens_of(c), d = c->model_dim, L = c->num_layers;
    const int din = p + (c->adaptive ? 6 : 0);
    const int planes = 2 * batch;

    float *tok6_all = NULL;
    if (c->adaptive) {
        tok6_all = (float *)malloc(sizeof(float) * (size_t)batch * tokens * 6);
        rc = aft_oracle_adapter_f32(c, w, snr, ds, dop, tok6_all, batch);
        if (rc) {
            free(tok6_all);
            return rc;
        }
        if (dump && dump->tokens6) memcpy(dump->tokens6, tok6_all, sizeof(float) * (size_t)batch * tokens * 6);
    }

#pragma omp parallel for schedule(dynamic, 1)
    for (int n = 0; n < planes; ++n) {
        const int b = n >> 1, part = n & 1;
        float *xin = (float *)malloc(sizeof(float) * PF);
        float *up = (float *)malloc(sizeof(float) * ST);
        float *ce = (float *)malloc(sizeof(float) * ST);
        float *tA = (float *)malloc(sizeof(float) * 32 * ST);
        float *tB = (float *)malloc(sizeof(float) * 32 * ST);
        float *ein = (float *)calloc((size_t)tokens * din, sizeof(float));
        float *x = (float *)malloc(sizeof(float) * (size_t)tokens * d);
        float *qkv = (float *)malloc(sizeof(float) * (size_t)tokens * 3 * d);
        float *att = (float *)malloc(sizeof(float) * (size_t)tokens * d);
        float *hid = (float *)malloc(sizeof(float) * (size_t)3 * d);
        float *prob = (float *)malloc(sizeof(float) * tokens);
        float *eo = (float *)malloc(sizeof(float) * (size_t)tokens * p);
        float *rec = (float *)malloc(sizeof(float) * ST);
        float *fin = (float *)malloc(sizeof(float) * ST);

        /* .real / .imag views, view(B, Ps*Pt) row-major idx = sc*Pt + sym -- fortitran.py:176-177,199-200 */
        for (int k = 0; k < PF; ++k) xin[k] = pilots[((size_t)b * PF + k) * 2 + part];
        /* S1 pilot_upsampler, .view(B,1,S,T) -- fortitran.py:203-206 */
        linear_f32(xin, w->up_w, w->up_b, 1, PF, ST, up);
        if (dump && dump->upsampled) memcpy(dump->upsampled + (size_t)n * ST, up, sizeof(float) * ST);
        /* S2 initial_enhancer -- fortitran.py:209 */
        conv_enhancer_f32(up, S, T, w->enh_w, w->enh_b, ce, tA, tB);
        if (dump && dump->conv_enhanced) memcpy(dump->conv_enhanced + (size_t)n * ST, ce, sizeof(float) * ST);
        /* S3 patch_embedder -- fortitran.py:212 ; S4 cat(patch, adapter; dim=2) -- :215-217 */
        patchify_f32(ce, S, T, p0, p1, ein, din);
        if (c->adaptive)
            for (int t = 0; t < tokens; ++t)
                for (int f = 0; f < 6; ++f)
                    ein[(size_t)t * din + p + f] = tok6_all[((size_t)b * tokens + t) * 6 + f];
        if (dump && dump->embed_in)
            memcpy(dump->embed_in + (size_t)n * tokens * din, ein, sizeof(float) * (size_t)tokens * din);
        /* S5 linear_1 + positional table rows [:tokens] -- encoders.py:67-68, positional_encodings.py:38,64 */
        linear_f32(ein, w->lin1_w, w->lin1_b, tokens, din, d, x);
        for (int t = 0; t < tokens; ++t)
            for (int e = 0; e < d; ++e) x[(size_t)t * d + e] = x[(size_t)t * d + e] + w->pos[(size_t)t * d + e];
        if (dump && dump->x0) memcpy(dump->x0 + (size_t)n * tokens * d, x, sizeof(float) * (size_t)tokens * d);
        /* encoder stack -- encoders.py:69 */
        for (int l = 0; l < L; ++l) {
            encoder_layer_plane(c, &w->layers[l], x, tokens, qkv, att, hid, prob);
            if (dump && dump->layer_out)
                memcpy(dump->layer_out + ((size_t)l * planes + n) * tokens * d, x,
                       sizeof(float) * (size_t)tokens * d);
        }
        /* linear_2 -- encoders.py:70 */
        linear_f32(x, w->lin2_w, w->lin2_b, tokens, d, p, eo);
        if (dump && dump->enc_out) memcpy(dump->enc_out + (size_t)n * tokens * p, eo, sizeof(float) * (size_t)tokens * p);
        /* S6 patch_reconstructor, S7 residual -- fortitran.py:225-228 */
        unpatchify_f32(eo, S, T, p0, p1, rec);
        for (int i = 0; i < ST; ++i) rec[i] = ce[i] + rec[i];
        if (dump && dump->residual) memcpy(dump->residual + (size_t)n * ST, rec, sizeof(float) * ST);
        /* S8 final_refiner -- fortitran.py:231 ; torch.complex(real, imag) -- :180 */
        conv_enhancer_f32(rec, S, T, w->ref_w, w->ref_b, fin, tA, tB);
        for (int i = 0; i < ST; ++i) out[((size_t)b * ST + i) * 2 + part] = fin[i];

        free(xin); free(up); free(ce); free(tA); free(tB); free(ein); free(x); free(qkv);
        free(att); free(hid); free(prob); free(eo); free(rec); free(fin);
    }
    free(tok6_all);
    return AFT_OK;
}

/* One encoder layer on x [planes*tokens, d] in place (for per-stage parity tests). */
int aft_oracle_encoder_layer_f32(const aft_config *c, const aft_weights *w, int layer, float *x,
                                 int batch) {
    int rc = check_cfg(c);
    if (rc) return rc;
    if (layer < 0 || layer >= c->num_layers || !x) return AFT_ERR_ARG;
    const int tokens = tokens_of(c), d = c->model_dim;
#pragma omp parallel for schedule(dynamic, 1)
    for (int n = 0; n < 2 * batch; ++n) {
        float *qkv = (float *)malloc(sizeof(float) * (size_t)tokens * 3 * d);
        float *att = (float *)malloc(sizeof(float) * (size_t)tokens * d);
        float *hid = (float *)malloc(sizeof(float) * (size_t)3 * d);
        float *prob = (float *)malloc(sizeof(float) * tokens);
        encoder_layer_plane(c, &w->layers[layer], x + (size_t)n * tokens * d, tokens, qkv, att, hid, prob);
        free(qkv); free(att); free(hid); free(prob);
    }
    return AFT_OK;
}

/* LinearEstimator.forward (src/models/linear.py:65-97) applied plane-wise: the reference's real
 * nn.Linear raises on the complex64 input its own dataset produces; the only
 * reading consistent with fortitran.py:176-180 is W applied to Re and Im separately. */
int aft_oracle_linear_forward_f32(const float *weight, const float *bias, const float *pilots,
                                  float *out, int batch, int in_features, int out_features) {
    if (!weight || !pilots || !out || batch <= 0) return AFT_ERR_ARG;
#pragma omp parallel for schedule(static)
    for (int n = 0; n < 2 * batch; ++n) {
        int b = n >> 1, part = n & 1;
        for (int o = 0; o < out_features; ++o) {
            double acc = bias ? (double)bias[o] : 0.0;
            for (int k = 0; k < in_features; ++k)
                acc += (double)weight[(size_t)o * in_features + k] *
                       (double)pilots[((size_t)b * in_features + k) * 2 + part];
            out[((size_t)b * out_features + o) * 2 + part] = (float)acc;
        }
    }
    return AFT_OK;
}

/* Metric: 2*MSELoss(cat(Re,Im; dim=1)) == mean over complex elements of |est-ref|^2
 * -- src/utils.py:164-180, src/main/trainer.py:338-347.  Returns the SUM; caller divides. */
int aft_oracle_mse_partial_f32(const float *est, const float *ref, double *sum_sq, long long n_complex) {
    if (!est || !ref || !sum_sq) return AFT_ERR_ARG;
    double acc = 0.0;
#pragma omp parallel for reduction(+ : acc) schedule(static)
    for (long long i = 0; i < 2 * n_complex; ++i) {
        double dlt = (double)est[i] - (double)ref[i];
        acc += dlt * dlt;
    }
    *sum_sq += acc;
    return AFT_OK;
}

int aft_oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void aft_oracle_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
