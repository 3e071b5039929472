// aft_train.hip -- C-ABI of the encoder layer's training forward / backward.
//
// One nn.TransformerEncoderLayer (post-LN, reference src/models/blocks/encoders.py:44-55) in
// train() mode as two calls: the forward keeps what the backward needs in a caller-owned "tape",
// the backward turns d(loss)/d(x_out) into d(loss)/d(x_in) and the gradients of the layer's twelve
// parameter tensors.  Everything row-major fp32, rows = 2B * tokens.  The caller (PyTorch autograd,
// adafortitran_amd/training.py) owns x_in, the tape and the gradient tensors.
#include <algorithm>
#include <cstdarg>
#include <cstdlib>

#include "aft_internal.h"

namespace aft {

namespace {

size_t al64(size_t floats) { return (floats + 63) / 64 * 64; }

struct Tape {   // offsets in floats
    size_t qkv, attn, lse, s1, st1, x1, a, hd, s2, st2, total;
};
struct Scratch {
    size_t g1, g2, g2b, gff, dqkv, dsum, slices, packed_t, lnp, attn_pad, total;
};

int tokens_of_cfg(const aft_config &c) { return (c.num_scs / c.patch_scs) * (c.num_symbols / c.patch_symbols); }

Tape plan_tape(const aft_config &c, int batch) {
    const size_t rows = (size_t)2 * batch * tokens_of_cfg(c), d = c.model_dim, ff = 2 * d;
    Tape t{};
    size_t off = 0;
    t.qkv = off;  off += al64(rows * 3 * d);
    t.attn = off; off += al64(rows * d);
    t.lse = off;  off += al64(rows * c.num_head);
    t.s1 = off;   off += al64(rows * d);
    t.st1 = off;  off += al64(rows * 2);
    t.x1 = off;   off += al64(rows * d);
    t.a = off;    off += al64(rows * ff);
    t.hd = off;   off += al64(rows * ff);   // drop(act(a)): kept so the backward does not recompute it (24 us per layer)
    t.s2 = off;   off += al64(rows * d);
    t.st2 = off;  off += al64(rows * 2);
    t.total = off;
    return t;
}

Scratch plan_scratch(const aft_config &c, int batch) {
    const size_t rows = (size_t)2 * batch * tokens_of_cfg(c), d = c.model_dim, ff = 2 * d;
    Scratch s{};
    size_t off = 0;
    s.g1 = off;     off += al64(rows * d);
    s.g2 = off;     off += al64(rows * d);
    s.g2b = off;    off += al64(rows * d);   // d(out_proj output): g2 stays alive for linear2's weight gradient (batched at the end)
    s.gff = off;    off += al64(rows * ff);
    s.dqkv = off;   off += al64(rows * 3 * d);
    s.dsum = off;   off += al64(rows * c.num_head);
    // every gradient kernel of the layer keeps its partial slices until the single reduction launch at the end
    const int r = (int)rows;
    s.slices = off;
    off += al64(gemm_tn_slice_floats(d, ff, r)) + al64(gemm_tn_slice_floats(ff, d, r)) + al64(gemm_tn_slice_floats(d, d, r)) +
           al64(gemm_tn_slice_floats(3 * d, d, r)) + 2 * al64((size_t)ln_bwd_blocks(r) * 3 * d) +
           al64((size_t)ln_bwd_blocks(r) * ff) + al64((size_t)colsum_slices(r) * 3 * d);
    // fused row-local backward (k_chain_bwd.hip): transposed fragment-packed weights, per-tile LayerNorm parameter sums
    s.packed_t = off; off += al64(std::max(chain_bwd_packed_floats((int)d), packed_layer_floats((int)d)));   // also the forward chain's fp32 image
    s.lnp = off;      off += al64(chain_bwd_lnp_floats(r, (int)d));
    s.attn_pad = off; off += al64(attn_train_pad_floats(c, rows));    // head dim 16: padded-head images of qkv, o, d_o, dqkv
    s.total = off;
    return s;
}

uint32_t site_seed(uint64_t seed, uint32_t site) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (site + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (uint32_t)(z ^ (z >> 31));
}

thread_local const char *g_step = "";
int fail(hipError_t e) {
    set_error("%s: %s", g_step, hipGetErrorString(e));
    return AFT_ERR_HIP;
}
#define STEP(name, call)                     \
    do {                                     \
        g_step = name;                       \
        hipError_t e_ = (call);              \
        if (e_ != hipSuccess) return fail(e_); \
    } while (0)

int check_train(const aft_config *cfg, int batch, float dropout_p) {
    int rc = check_config(cfg);
    if (rc != AFT_OK) return rc;
    if (batch <= 0 || !(dropout_p >= 0.f && dropout_p < 1.f)) {
        set_error("bad batch %d or dropout %g", batch, (double)dropout_p);
        return AFT_ERR_ARG;
    }
    // head dim 32 is the kernels' own shape, 64 runs as two 32-feature blocks under one softmax (forward + the two-pass backward
    // instantiated for two blocks), every other multiple of 8 as zero-padded 32- or 64-feature heads (k_attn_train.hip); check_config
    // above has already refused what the inference engine does not take (head dims off the multiples of 8, 56, > 64)
    if ((size_t)2 * batch * tokens_of_cfg(*cfg) * 3 * cfg->model_dim >= ((size_t)1 << 32)) {
        set_error("batch %d: dropout counters are 32-bit", batch);
        return AFT_ERR_ARG;
    }
    return AFT_OK;
}

}  // namespace

}  // namespace aft

using namespace aft;

extern "C" {

size_t aft_encoder_tape_bytes(const aft_config *cfg, int batch) {
    if (check_config(cfg) != AFT_OK || batch <= 0) return 0;
    return plan_tape(*cfg, batch).total * sizeof(float);
}

size_t aft_encoder_train_scratch_bytes(const aft_config *cfg, int batch) {
    if (check_config(cfg) != AFT_OK || batch <= 0) return 0;
    return plan_scratch(*cfg, batch).total * sizeof(float);
}

int aft_encoder_layer_fwd_train_chained_f32(const aft_config *cfg, const aft_layer_weights *w, const float *x_in, float *x_out,
                                            void *tape, size_t tape_bytes, void *scratch, size_t scratch_bytes, int batch,
                                            float dropout_p, uint64_t seed, int qkv_ready, const aft_layer_weights *next_w,
                                            void *next_tape, size_t next_tape_bytes, int *next_qkv_written, void *stream) {
    if (next_qkv_written) *next_qkv_written = 0;
    int rc = check_train(cfg, batch, dropout_p);
    if (rc != AFT_OK) return rc;
    if (!w || !x_in || !x_out || !tape || !scratch) { set_error("NULL pointer argument"); return AFT_ERR_ARG; }
    const Tape t = plan_tape(*cfg, batch);
    const Scratch s = plan_scratch(*cfg, batch);
    if (tape_bytes < t.total * sizeof(float) || scratch_bytes < s.total * sizeof(float)) {
        set_error("tape or scratch too small");
        return AFT_ERR_ARG;
    }
    if (next_w && next_tape && next_tape_bytes < t.total * sizeof(float)) {   // this call writes rows x 3d floats into next_tape's qkv block
        set_error("next_tape too small: %zu < %zu bytes (the linked layers share cfg and batch)", next_tape_bytes, t.total * sizeof(float));
        return AFT_ERR_ARG;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    float *tp = static_cast<float *>(tape), *sc = static_cast<float *>(scratch);
    const int tokens = tokens_of_cfg(*cfg), planes = 2 * batch, rows = planes * tokens, d = cfg->model_dim, ff = 2 * d;
    float *o = sc + s.g1;   // projection outputs before the residual joins

    if (!qkv_ready)   // (else the previous layer's row-local kernel already left this layer's in-projection in the tape)
        STEP("qkv", launch_gemm(0, x_in, w->in_proj_w, tp + t.qkv, w->in_proj_b, rows, 3 * d, d, d, d, 3 * d, false, st));
    STEP("attention", launch_attn_train_fwd(*cfg, tp + t.qkv, tp + t.attn, tp + t.lse, planes, tokens, dropout_p,
                                            site_seed(seed, 0), st, sc + s.attn_pad));
    const uint32_t drop_th = dropout_p > 0.f ? (uint32_t)((double)dropout_p * 4294967296.0) : 0u;
    const float drop_ks = dropout_p > 0.f ? 1.f / (1.f - dropout_p) : 1.f;
    // the NEXT layer's in-projection as the tail of this layer's row-local kernel (same tape layout: the layers share cfg and batch)
    const bool chain_next = next_w && next_w->in_proj_w && next_w->in_proj_b && next_tape && next_qkv_written;
    if (chain_fwd_train_ok(*cfg, rows) && !switch_on("AFT_TRAIN_UNFUSED_FWD")) {
        // everything row-local behind the attention in ONE launch, the tape written from its epilogues (k_chain_bwd.hip);
        // the packed weight image lives in the scratch's packed_t region (8 d^2 floats needed, see plan_scratch)
        STEP("row-local forward chain", launch_chain_fwd_train(*cfg, *w, tp + t.attn, x_in, sc + s.packed_t, tp + t.s1, tp + t.st1, tp + t.x1,
                                                              tp + t.a, tp + t.hd, tp + t.s2, tp + t.st2, x_out, rows, site_seed(seed, 1),
                                                              site_seed(seed, 2), site_seed(seed, 3), drop_th, drop_ks, st,
                                                              chain_next ? next_w->in_proj_w : nullptr, chain_next ? next_w->in_proj_b : nullptr,
                                                              chain_next ? static_cast<float *>(next_tape) + t.qkv : nullptr));
        if (chain_next && next_qkv_written) *next_qkv_written = 1;
        return AFT_OK;
    }
    // projection + residual + dropout + LayerNorm: one launch when the fused epilogue covers the shape (d = 128)
    if (gemm_add_ln_ok(rows, d, d, d, d)) {
        STEP("out_proj + norm1", launch_gemm_add_ln(tp + t.attn, w->out_proj_w, w->out_proj_b, x_in, w->norm1_w, w->norm1_b, tp + t.s1,
                                                    tp + t.st1, tp + t.x1, rows, d, d, d, d, 1e-5f, drop_ks, drop_th, site_seed(seed, 1), st));
    } else {
        STEP("out_proj", launch_gemm(0, tp + t.attn, w->out_proj_w, o, w->out_proj_b, rows, d, d, d, d, d, false, st));
        STEP("norm1", launch_add_ln_fwd(x_in, o, w->norm1_w, w->norm1_b, tp + t.s1, tp + t.st1, tp + t.x1, rows, d, 1e-5f,
                                        dropout_p, site_seed(seed, 1), st));
    }
    if (gemm_act_ok(rows, ff, d, d, d, ff)) {   // linear1 + activation + dropout: one launch
        STEP("linear1 + activation", launch_gemm_act(tp + t.x1, w->lin1_w, w->lin1_b, tp + t.a, tp + t.hd, rows, ff, d, d, d, ff,
                                                     cfg->activation, drop_ks, drop_th, site_seed(seed, 2), st));
    } else {
        STEP("linear1", launch_gemm(0, tp + t.x1, w->lin1_w, tp + t.a, w->lin1_b, rows, ff, d, d, d, ff, false, st));
        STEP("activation", launch_act_fwd(cfg->activation, tp + t.a, tp + t.hd, rows, ff, dropout_p, site_seed(seed, 2), st));
    }
    if (gemm_add_ln_ok(rows, d, ff, ff, ff)) {
        STEP("linear2 + norm2", launch_gemm_add_ln(tp + t.hd, w->lin2_w, w->lin2_b, tp + t.x1, w->norm2_w, w->norm2_b, tp + t.s2, tp + t.st2,
                                                   x_out, rows, d, ff, ff, ff, 1e-5f, drop_ks, drop_th, site_seed(seed, 3), st));
    } else {
        STEP("linear2", launch_gemm(0, tp + t.hd, w->lin2_w, o, w->lin2_b, rows, d, ff, ff, ff, d, false, st));
        STEP("norm2", launch_add_ln_fwd(tp + t.x1, o, w->norm2_w, w->norm2_b, tp + t.s2, tp + t.st2, x_out, rows, d, 1e-5f,
                                        dropout_p, site_seed(seed, 3), st));
    }
    return AFT_OK;
}

int aft_encoder_layer_fwd_train_f32(const aft_config *cfg, const aft_layer_weights *w, const float *x_in, float *x_out,
                                    void *tape, size_t tape_bytes, void *scratch, size_t scratch_bytes, int batch,
                                    float dropout_p, uint64_t seed, void *stream) {
    return aft_encoder_layer_fwd_train_chained_f32(cfg, w, x_in, x_out, tape, tape_bytes, scratch, scratch_bytes, batch, dropout_p, seed, 0,
                                                   nullptr, nullptr, 0, nullptr, stream);
}

int aft_encoder_layer_bwd_f32(const aft_config *cfg, const aft_layer_weights *w, const float *x_in, const void *tape,
                              size_t tape_bytes, const float *dx_out, float *dx_in, const aft_layer_grads *g,
                              int accumulate, void *scratch, size_t scratch_bytes, int batch, float dropout_p,
                              uint64_t seed, void *stream) {
    int rc = check_train(cfg, batch, dropout_p);
    if (rc != AFT_OK) return rc;
    if (!w || !x_in || !tape || !dx_out || !dx_in || !g || !scratch) { set_error("NULL pointer argument"); return AFT_ERR_ARG; }
    const Tape t = plan_tape(*cfg, batch);
    const Scratch s = plan_scratch(*cfg, batch);
    if (tape_bytes < t.total * sizeof(float) || scratch_bytes < s.total * sizeof(float)) {
        set_error("tape or scratch too small");
        return AFT_ERR_ARG;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float *tp = static_cast<const float *>(tape);
    float *sc = static_cast<float *>(scratch);
    const int tokens = tokens_of_cfg(*cfg), planes = 2 * batch, rows = planes * tokens, d = cfg->model_dim, ff = 2 * d;
    const bool acc = accumulate != 0;
    float *g1 = sc + s.g1, *g2 = sc + s.g2, *g2b = sc + s.g2b, *gff = sc + s.gff, *dqkv = sc + s.dqkv, *sl = sc + s.slices;
    const float *hd = tp + t.hd;

    // slice storage per producer (bump-allocated from the scratch's slice region), one reduction launch at the end
    float *sl_ln2 = sl;
    float *sl_w2 = sl_ln2 + al64((size_t)ln_bwd_blocks(rows) * 3 * d);
    float *sl_act = sl_w2 + al64(gemm_tn_slice_floats(d, ff, rows));
    float *sl_w1 = sl_act + al64((size_t)ln_bwd_blocks(rows) * ff);
    float *sl_ln1 = sl_w1 + al64(gemm_tn_slice_floats(ff, d, rows));
    float *sl_wo = sl_ln1 + al64((size_t)ln_bwd_blocks(rows) * 3 * d);
    float *sl_wq = sl_wo + al64(gemm_tn_slice_floats(d, d, rows));
    float *sl_bq = sl_wq + al64(gemm_tn_slice_floats(3 * d, d, rows));
    ReduceBatchScope reductions;
    const uint32_t drop_th = dropout_p > 0.f ? (uint32_t)((double)dropout_p * 4294967296.0) : 0u;
    const float drop_ks = dropout_p > 0.f ? 1.f / (1.f - dropout_p) : 1.f;
    if (chain_bwd_ok(*cfg, rows) && !switch_on("AFT_TRAIN_UNFUSED_BWD")) {
        // everything row-local in ONE launch: dx_out -> g2, gff, g2b (operands of the weight gradients), g1 = d(attention
        // output), dx_in = the residual branch of dL/dx, per-tile LayerNorm parameter sums
        float *lnp = sc + s.lnp;
        STEP("row-local backward chain", launch_chain_bwd(*cfg, *w, dx_out, tp + t.s2, tp + t.st2, tp + t.a, tp + t.s1, tp + t.st1,
                                                          sc + s.packed_t, g2, gff, g2b, g1, dx_in, lnp, rows, site_seed(seed, 1),
                                                          site_seed(seed, 2), site_seed(seed, 3), drop_th, drop_ks, st));
        const int ntl = chain_bwd_blocks(rows);
        STEP("norm2 parameter gradients", launch_reduce_slices3(lnp, g->norm2_w, g->norm2_b, nullptr, d, 2, ntl, (size_t)4 * d, acc, st));
        STEP("norm1 parameter gradients", launch_reduce_slices3(lnp + 2 * d, g->norm1_w, g->norm1_b, nullptr, d, 2, ntl, (size_t)4 * d, acc, st));
        STEP("attention bwd", launch_attn_train_bwd(*cfg, tp + t.qkv, tp + t.attn, g1, tp + t.lse, sc + s.dsum, dqkv, planes, tokens,
                                                    dropout_p, site_seed(seed, 0), st, sc + s.attn_pad));
        STEP("in_proj dgrad", launch_gemm(1, dqkv, w->in_proj_w, dx_in, nullptr, rows, d, 3 * d, 3 * d, d, d, true, st));
        {   // the four weight gradients and all four bias gradients (column sums of the A operands) in one launch
            const float *A[4] = {g2, gff, g2b, dqkv}, *B[4] = {hd, tp + t.x1, tp + t.attn, x_in};
            float *C[4] = {g->lin2_w, g->lin1_w, g->out_proj_w, g->in_proj_w}, *S[4] = {sl_w2, sl_w1, sl_wo, sl_wq};
            const int M[4] = {d, ff, d, 3 * d}, N[4] = {ff, d, d, d}, lda[4] = {d, ff, d, 3 * d}, ldb[4] = {ff, d, d, d};
            float *bias_out[4] = {g->lin2_b, g->lin1_b, g->out_proj_b, g->in_proj_b};
            float *bias_sl[4] = {sl_ln2, sl_act, sl_ln1, sl_bq};   // the LayerNorm slice regions are free on this path
            STEP("weight gradients", launch_gemm_tn_batch(A, B, C, S, M, N, lda, ldb, 4, rows, acc, st, bias_out, bias_sl));
        }
        STEP("gradient reductions", reductions.flush(st));
        return AFT_OK;
    }
    // LN2: g1 = d(x1) through the residual, g2 = d(linear2 output) (dropout 3 applied)
    STEP("norm2 bwd", launch_ln_bwd(dx_out, tp + t.s2, tp + t.st2, w->norm2_w, g1, g2, g->norm2_w, g->norm2_b, g->lin2_b, sl_ln2,
                                    rows, d, dropout_p, site_seed(seed, 3), acc, st));
    // linear2's data gradient with the activation backward as its epilogue (gff = d(linear1 output)); linear1's bias gradient
    // then rides on the batched weight-gradient launch below (column sums of gff)
    const bool fused_act = gemm_actbwd_ok(rows, ff, d, d, ff, ff);
    if (fused_act) {
        STEP("linear2 dgrad + activation bwd", launch_gemm_actbwd(g2, w->lin2_w, tp + t.a, gff, rows, ff, d, d, ff, ff, cfg->activation,
                                                                  drop_ks, drop_th, site_seed(seed, 2), st));
    } else {
        STEP("linear2 dgrad", launch_gemm(1, g2, w->lin2_w, gff, nullptr, rows, ff, d, d, ff, ff, false, st));
        STEP("activation bwd", launch_act_bwd(cfg->activation, tp + t.a, gff, g->lin1_b, sl_act, rows, ff, dropout_p, site_seed(seed, 2),
                                              acc, st));
    }
    STEP("linear1 dgrad", launch_gemm(1, gff, w->lin1_w, g1, nullptr, rows, d, ff, ff, d, d, true, st));
    // LN1: dx_in = d(x_in) through the residual, g2b = d(out_proj output) (dropout 1 applied)
    STEP("norm1 bwd", launch_ln_bwd(g1, tp + t.s1, tp + t.st1, w->norm1_w, dx_in, g2b, g->norm1_w, g->norm1_b, g->out_proj_b, sl_ln1,
                                    rows, d, dropout_p, site_seed(seed, 1), acc, st));
    STEP("out_proj dgrad", launch_gemm(1, g2b, w->out_proj_w, g1, nullptr, rows, d, d, d, d, d, false, st));
    STEP("attention bwd", launch_attn_train_bwd(*cfg, tp + t.qkv, tp + t.attn, g1, tp + t.lse, sc + s.dsum, dqkv, planes, tokens,
                                                dropout_p, site_seed(seed, 0), st, sc + s.attn_pad));
    STEP("in_proj dgrad", launch_gemm(1, dqkv, w->in_proj_w, dx_in, nullptr, rows, d, 3 * d, 3 * d, d, d, true, st));
    {   // the layer's four weight gradients in one launch: dW2 = g2^T hd, dW1 = gff^T x1, dWo = g2b^T attn, dWqkv = dqkv^T x_in
        const float *A[4] = {g2, gff, g2b, dqkv}, *B[4] = {hd, tp + t.x1, tp + t.attn, x_in};
        float *C[4] = {g->lin2_w, g->lin1_w, g->out_proj_w, g->in_proj_w}, *S[4] = {sl_w2, sl_w1, sl_wo, sl_wq};
        const int M[4] = {d, ff, d, 3 * d}, N[4] = {ff, d, d, d}, lda[4] = {d, ff, d, 3 * d}, ldb[4] = {ff, d, d, d};
        float *bias_out[4] = {nullptr, fused_act ? g->lin1_b : nullptr, nullptr, g->in_proj_b};   // db1 = gff^T 1, db_qkv = dqkv^T 1
        float *bias_sl[4] = {nullptr, sl_act, nullptr, sl_bq};
        STEP("weight gradients", launch_gemm_tn_batch(A, B, C, S, M, N, lda, ldb, 4, rows, acc, st, bias_out, bias_sl));
    }
    STEP("gradient reductions", reductions.flush(st));
    return AFT_OK;
}

static size_t dense_slice_floats(int rows, int in_f, int out_f) {
    const int tiles = ((in_f + 127) / 128) * ((out_f + 127) / 128);
    return al64((size_t)gemm_split_slices(rows, tiles) * in_f * out_f) + al64((size_t)colsum_slices(rows) * out_f);
}

size_t aft_dense_bwd_scratch_bytes(int rows, int in_features, int out_features) {
    if (rows <= 0 || in_features <= 0 || out_features <= 0) return 0;
    return sizeof(float) * dense_slice_floats(rows, in_features, out_features);
}

int aft_dense_fwd_f32(const float *x, const float *weight, const float *bias, float *y, int rows, int in_features,
                      int out_features, void *stream) {
    if (!x || !weight || !y || rows <= 0 || in_features <= 0 || out_features <= 0) { set_error("bad dense argument"); return AFT_ERR_ARG; }
    STEP("dense forward", launch_gemm(0, x, weight, y, bias, rows, out_features, in_features, in_features, in_features,
                                      out_features, false, static_cast<hipStream_t>(stream)));
    return AFT_OK;
}

int aft_dense_bwd_f32(const float *x, const float *weight, const float *dy, float *dx, float *dweight, float *dbias,
                      int accumulate, void *scratch, size_t scratch_bytes, int rows, int in_features, int out_features,
                      void *stream) {
    if (!x || !weight || !dy || !dweight || !scratch || rows <= 0 || in_features <= 0 || out_features <= 0) {
        set_error("bad dense argument");
        return AFT_ERR_ARG;
    }
    if (scratch_bytes < aft_dense_bwd_scratch_bytes(rows, in_features, out_features)) { set_error("dense scratch too small"); return AFT_ERR_ARG; }
    hipStream_t st = static_cast<hipStream_t>(stream);
    float *sl = static_cast<float *>(scratch);
    const int tiles = ((in_features + 127) / 128) * ((out_features + 127) / 128);
    float *sl2 = sl + al64((size_t)gemm_split_slices(rows, tiles) * in_features * out_features);
    ReduceBatchScope reductions;
    if (dx) STEP("dense dgrad", launch_gemm(1, dy, weight, dx, nullptr, rows, in_features, out_features, out_features, in_features,
                                            in_features, false, st));
    STEP("dense wgrad", launch_gemm_tn(dy, x, dweight, sl, out_features, in_features, rows, out_features, in_features,
                                       accumulate != 0, st));
    if (dbias) STEP("dense bgrad", launch_colsum(dy, dbias, sl2, rows, out_features, out_features, accumulate != 0, st));
    STEP("gradient reductions", reductions.flush(st));
    return AFT_OK;
}

size_t aft_embed_bwd_scratch_bytes(int planes, int num_scs, int num_symbols, int patch_scs, int patch_symbols, int model_dim,
                                   int with_tokens6) {
    return sizeof(float) * embed_bwd_slice_floats(planes, num_scs, num_symbols, patch_scs, patch_symbols, model_dim, with_tokens6 != 0);
}

int aft_embed_fwd_train_f32(const float *conv_enhanced, const float *tokens6, const float *w1, const float *b1, const float *pos,
                            float *x0, int planes, int num_scs, int num_symbols, int patch_scs, int patch_symbols, int model_dim,
                            void *stream) {
    if (!conv_enhanced || !w1 || !b1 || !pos || !x0 ||
        !ends_train_ok(planes, num_scs, num_symbols, patch_scs, patch_symbols, model_dim, tokens6 != nullptr)) {
        set_error("bad embed argument (patch of at most 32 elements, model_dim a multiple of 4 up to 512)");
        return AFT_ERR_ARG;
    }
    STEP("embed forward", launch_embed_train_fwd(conv_enhanced, tokens6, w1, b1, pos, x0, planes, num_scs, num_symbols, patch_scs,
                                                 patch_symbols, model_dim, static_cast<hipStream_t>(stream)));
    return AFT_OK;
}

int aft_embed_bwd_f32(const float *conv_enhanced, const float *tokens6, const float *w1, const float *dx0, float *d_conv_enhanced,
                      float *d_tokens6, float *dw1, float *db1, float *dpos, int accumulate, void *scratch, size_t scratch_bytes,
                      int planes, int num_scs, int num_symbols, int patch_scs, int patch_symbols, int model_dim, void *stream) {
    if (!conv_enhanced || !w1 || !dx0 || !d_conv_enhanced || !dw1 || !scratch || (tokens6 != nullptr) != (d_tokens6 != nullptr) ||
        !ends_train_ok(planes, num_scs, num_symbols, patch_scs, patch_symbols, model_dim, tokens6 != nullptr)) {
        set_error("bad embed argument (patch of at most 32 elements, model_dim a multiple of 4 up to 512)");
        return AFT_ERR_ARG;
    }
    if (scratch_bytes < aft_embed_bwd_scratch_bytes(planes, num_scs, num_symbols, patch_scs, patch_symbols, model_dim, tokens6 != nullptr)) {
        set_error("embed scratch too small");
        return AFT_ERR_ARG;
    }
    STEP("embed backward", launch_embed_train_bwd(conv_enhanced, tokens6, w1, dx0, d_conv_enhanced, d_tokens6, dw1, db1, dpos,
                                                  accumulate != 0, static_cast<float *>(scratch), planes, num_scs, num_symbols, patch_scs,
                                                  patch_symbols, model_dim, static_cast<hipStream_t>(stream)));
    return AFT_OK;
}

size_t aft_tail_bwd_scratch_bytes(int planes, int num_scs, int num_symbols, int patch_scs, int patch_symbols, int model_dim) {
    return sizeof(float) * tail_bwd_slice_floats(planes, num_scs, num_symbols, patch_scs, patch_symbols, model_dim);
}

int aft_tail_fwd_train_f32(const float *x, const float *w2, const float *b2, const float *resid, float *out, int planes, int num_scs,
                           int num_symbols, int patch_scs, int patch_symbols, int model_dim, void *stream) {
    if (!x || !w2 || !b2 || !resid || !out || !ends_train_ok(planes, num_scs, num_symbols, patch_scs, patch_symbols, model_dim, false)) {
        set_error("bad tail argument (patch of at most 32 elements, model_dim a multiple of 4 up to 512)");
        return AFT_ERR_ARG;
    }
    STEP("tail forward", launch_tail_train_fwd(x, w2, b2, resid, out, planes, num_scs, num_symbols, patch_scs, patch_symbols, model_dim,
                                               static_cast<hipStream_t>(stream)));
    return AFT_OK;
}

int aft_tail_bwd_f32(const float *x, const float *w2, const float *d_out, float *dx, float *dw2, float *db2, int accumulate,
                     void *scratch, size_t scratch_bytes, int planes, int num_scs, int num_symbols, int patch_scs, int patch_symbols,
                     int model_dim, void *stream) {
    if (!x || !w2 || !d_out || !dx || !dw2 || !scratch ||
        !ends_train_ok(planes, num_scs, num_symbols, patch_scs, patch_symbols, model_dim, false)) {
        set_error("bad tail argument (patch of at most 32 elements, model_dim a multiple of 4 up to 512)");
        return AFT_ERR_ARG;
    }
    if (scratch_bytes < aft_tail_bwd_scratch_bytes(planes, num_scs, num_symbols, patch_scs, patch_symbols, model_dim)) {
        set_error("tail scratch too small");
        return AFT_ERR_ARG;
    }
    STEP("tail backward", launch_tail_train_bwd(x, w2, d_out, dx, dw2, db2, accumulate != 0, static_cast<float *>(scratch), planes, num_scs,
                                                num_symbols, patch_scs, patch_symbols, model_dim, static_cast<hipStream_t>(stream)));
    return AFT_OK;
}

size_t aft_conv_enhancer_scratch_bytes(int planes, int num_scs, int num_symbols) {
    if (planes <= 0 || num_scs <= 0 || num_symbols <= 0 || !conv_plan_ok(num_scs, num_symbols, 0)) return 0;
    return sizeof(float) * (al64((size_t)planes * 48 * num_scs * num_symbols) + al64(conv_wgrad_slice_floats(planes, num_scs, num_symbols)) +
                            al64(kConvFlipFloats) + al64(kConvFragFloats));
}

size_t aft_conv_enhancer_fwd_scratch_bytes(int planes, int num_scs, int num_symbols) {
    if (planes <= 0 || num_scs <= 0 || num_symbols <= 0 || !conv_plan_ok(num_scs, num_symbols, 0)) return 0;
    return sizeof(float) * al64(kConvFragFloats);
}

int aft_conv_enhancer_fwd_train_f32(const float *const weights[4], const float *const biases[4], const float *x, float *y,
                                    float *c1, float *c2, float *c3, void *scratch, size_t scratch_bytes, int planes, int num_scs,
                                    int num_symbols, void *stream) {
    if (!weights || !biases || !x || !y || !c1 || !c2 || !c3 || planes <= 0 || num_scs <= 0 || num_symbols <= 0) {
        set_error("bad ConvEnhancer argument");
        return AFT_ERR_ARG;
    }
    if (scratch != nullptr && scratch_bytes < aft_conv_enhancer_fwd_scratch_bytes(planes, num_scs, num_symbols)) {
        set_error("ConvEnhancer forward scratch too small");
        return AFT_ERR_ARG;
    }
    float *const save[3] = {c1, c2, c3};
    STEP("conv forward", launch_conv_train(weights, biases, x, y, save, nullptr, planes, num_scs, num_symbols,
                                           static_cast<hipStream_t>(stream), static_cast<float *>(scratch)));
    return AFT_OK;
}

int aft_conv_enhancer_bwd_f32(const float *const weights[4], const float *x, const float *c1, const float *c2,
                              const float *c3, const float *dy, float *dx, float *const dweights[4], float *const dbiases[4],
                              int accumulate, void *scratch, size_t scratch_bytes, int planes, int num_scs, int num_symbols,
                              void *stream) {
    if (!weights || !x || !c1 || !c2 || !c3 || !dy || !dx || !dweights || !dbiases || !scratch || planes <= 0) {
        set_error("bad ConvEnhancer argument");
        return AFT_ERR_ARG;
    }
    if (scratch_bytes < aft_conv_enhancer_scratch_bytes(planes, num_scs, num_symbols)) {
        set_error("ConvEnhancer scratch too small");
        return AFT_ERR_ARG;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t plane8 = (size_t)planes * 8 * num_scs * num_symbols;
    float *g3 = static_cast<float *>(scratch), *g2 = g3 + plane8, *g1 = g2 + 4 * plane8;
    float *slices = static_cast<float *>(scratch) + al64(6 * plane8);
    float *flip = slices + al64(conv_wgrad_slice_floats(planes, num_scs, num_symbols));
    float *frag = flip + al64(kConvFlipFloats);      // the flipped weights as 16x16x4 operand fragments (default grid)
    // dgrad: the stack run on dy with conv4^T (1->8), conv3^T (8->32), conv2^T (32->8), conv1^T (8->1);
    // stage outputs masked by the saved activations = g3, g2, g1
    STEP("conv weight transposition", launch_conv_flip_weights(weights, flip, st));
    const float *const flipped_weights[4] = {flip, flip + 72, flip + 72 + 2304, flip + 72 + 4608};
    float *const save[3] = {g3, g2, g1};
    const float *const mask[3] = {c3, c2, c1};
    STEP("conv dgrad", launch_conv_train(flipped_weights, nullptr, dy, dx, save, mask, planes, num_scs, num_symbols, st, frag));
    STEP("conv wgrad", launch_conv_wgrad(x, c1, c2, c3, g1, g2, g3, dy, dweights, dbiases, slices, planes, num_scs, num_symbols,
                                         accumulate != 0, st));
    return AFT_OK;
}

int aft_adapter_fwd_train_f32(const float *const conditions[3], const float *const weights[9], const float *const biases[9],
                              const int32_t hidden[3], int tokens, int frames, float *tokens6, float *hidden0, float *hidden1,
                              void *stream) {
    if (!conditions || !weights || !biases || !hidden || !tokens6 || !hidden0 || !hidden1 || frames <= 0 || tokens <= 0 ||
        hidden[2] != 2 * tokens) {
        set_error("bad ChannelAdapter argument");
        return AFT_ERR_ARG;
    }
    const int h[3] = {hidden[0], hidden[1], hidden[2]};
    STEP("adapter forward", launch_adapter_train_fwd(conditions, weights, biases, h, tokens, frames, tokens6, hidden0, hidden1,
                                                     static_cast<hipStream_t>(stream)));
    return AFT_OK;
}

int aft_adapter_bwd_f32(const float *const conditions[3], const float *const weights[9], const float *const biases[9],
                        const int32_t hidden[3], int tokens, int frames, const float *hidden0, const float *hidden1,
                        const float *dtokens6, float *dhidden0, float *dhidden1, float *const dweights[9],
                        float *const dbiases[9], int accumulate, void *stream) {
    if (!conditions || !weights || !biases || !hidden || !hidden0 || !hidden1 || !dtokens6 || !dhidden0 || !dhidden1 ||
        !dweights || !dbiases || frames <= 0 || tokens <= 0 || hidden[2] != 2 * tokens) {
        set_error("bad ChannelAdapter argument");
        return AFT_ERR_ARG;
    }
    if (hidden[1] > 64 || hidden[0] > 256) {
        set_error("ChannelAdapter hidden sizes [%d, %d] not covered by the fused backward (<= 256, <= 64)", hidden[0], hidden[1]);
        return AFT_ERR_SHAPE;
    }
    const int h[3] = {hidden[0], hidden[1], hidden[2]};
    STEP("adapter backward", launch_adapter_train_bwd(conditions, weights, biases, h, tokens, frames, hidden0, hidden1, dtokens6,
                                                      dhidden0, dhidden1, dweights, dbiases, accumulate != 0,
                                                      static_cast<hipStream_t>(stream)));
    return AFT_OK;
}

int aft_adam_step_f32(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, size_t n, float lr, float beta1,
                      float beta2, float eps, float weight_decay, float grad_scale, int step, void *stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || step < 1) { set_error("bad Adam argument"); return AFT_ERR_ARG; }
    if (n == 0) return AFT_OK;
    STEP("adam", launch_adam(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, grad_scale, step,
                             static_cast<hipStream_t>(stream)));
    return AFT_OK;
}

}  // extern "C"
