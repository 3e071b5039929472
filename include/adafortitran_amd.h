/*
 * adafortitran_amd.h -- C ABI of the MI355X (gfx950) AdaFortiTran forward path.
 *
 * The reference (BerkIGuler/AdaFortiTran) has no native layer: its boundary is the
 * Python nn.Module surface in src/models/ (SURVEY.md 8b).  This header is the ABI
 * that sits directly *below* that surface; each entry point names the reference
 * interface it replaces.  The Python-side binding a maintainer adds is a ctypes
 * stub (adafortitran_amd/_lib.py is that stub).
 *
 * Conventions
 *   - plain C symbols, no C++/torch types; all tensors are raw DEVICE pointers owned
 *     by the caller (PyTorch allocates inputs, outputs, weights and the workspace);
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*, NULL =
 *     default stream), never synchronises, allocates nothing and keeps no state that
 *     describes a call (hipGraph-capturable, also as the very first call of a process);
 *     what the library caches is per DEVICE and idempotent -- the CU count and the
 *     "kernel may use N bytes of dynamic LDS" attribute, in tables indexed by the
 *     current device ordinal -- so one process may drive several GPUs through this ABI;
 *     the caller's loss.item() is the sync point, as in reference
 *     src/main/trainer.py:229,253,344;
 *   - complex64 tensors are passed as float* to interleaved (re,im) pairs, i.e. the
 *     memory of torch.view_as_real(x);
 *   - return 0 on success; AFT_ERR_ARG / AFT_ERR_SHAPE -> the Python side raises
 *     ValueError, AFT_ERR_HIP -> RuntimeError; aft_last_error() gives the text
 *     (thread-local).
 *   - inside the library a "plane" is one real-valued pass of the reference's
 *     _forward_real_valued (fortitran.py:176-177): plane n = 2*frame + (0:Re | 1:Im).
 */
#ifndef ADAFORTITRAN_AMD_H
#define ADAFORTITRAN_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AFT_ABI_VERSION 8   /* bump whenever an entry point's meaning, a struct or a scratch size changes */

#define AFT_OK 0
#define AFT_ERR_ARG 1   /* NULL pointer, bad batch, workspace too small ...          */
#define AFT_ERR_SHAPE 2 /* configuration the kernels do not cover                    */
#define AFT_ERR_HIP 3   /* a HIP runtime call or kernel launch failed                */

/* aft_config.encoder_path.  LAUNCHES: embedding+QKV, then [attention, chain] per layer (13 launches at 6 layers).
 * PLANE: the plane-resident kernel -- embedding + all layers + linear_2 in ONE launch, one 12-wave workgroup per plane
 * (model_dim 128 only; other shapes run the launches).  Identical output bits.  AUTO = LAUNCHES: on the MI355X the
 * plane kernel measured 1.5 % slower even at its best case, 256 planes on 256 CUs (profiles/r03_ab_encoder.json). */
#define AFT_ENCODER_AUTO 0
#define AFT_ENCODER_LAUNCHES 1
#define AFT_ENCODER_PLANE 2

/* aft_config.precision.  F32: every product on exact-fp32 MFMAs (v_mfma_f32_32x32x2_f32) -- the default, the parity
 * contract (5e-5 |y|max, |dMSE|/MSE <= 1e-4) and the only mode the headline benchmark runs.  BF16X3 (opt-in, reported
 * separately, SURVEY.md 8d "bf16-MFMA tier"): the encoder's GEMMs (in-projection, out-projection, FFN) and attention products split
 * each fp32 operand into bf16 hi + lo and accumulate hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_bf16 in fp32 (~2^-16
 * relative per product; stated tolerance max|d| <= 1e-3 |y|max, |dMSE|/MSE <= 1e-2, observed ~1e-5 / ~1e-6: tests/
 * test_hip_parity.py).  The softmax, LayerNorm, GELU, the conv stacks and every accumulator stay fp32.  model_dim 128 or 256;
 * inference only. */
#define AFT_PRECISION_F32 0
#define AFT_PRECISION_BF16X3 1

#define AFT_ACT_RELU 0
#define AFT_ACT_GELU 1 /* exact erf form, as activation="gelu" in encoders.py:44-51 */

/* Shape of one estimator: system_config.yaml + model yaml
 * (reference src/config/schemas.py:20-45,113-146; fortitran.py:52-81). */
typedef struct aft_config {
    int32_t num_scs, num_symbols;     /* OFDM grid S x T (120 x 14)                  */
    int32_t pilot_scs, pilot_symbols; /* pilot grid (12 x 2)                         */
    int32_t patch_scs, patch_symbols; /* patch (3 x 2): tokens = (S/p0)*(T/p1)       */
    int32_t num_layers, model_dim, num_head; /* covered (aft_check_config says why not otherwise): any layer count, model_dim a multiple
                                       * of 8 up to 512, model_dim / num_head up to 128, patches of <= 32 elements.  Two engines behind one
                                       * call (aft_engine_of): the fragment-packed launch sequence -- model_dim a multiple of 32 up to 256,
                                       * head dim a multiple of 8 up to 64 except 56, patches of <= 16 elements -- and the row-major
                                       * general sequence for everything else (slower) */
    int32_t activation;               /* AFT_ACT_*                                   */
    int32_t adaptive;                 /* 1 = AdaFortiTran (adapter tokens), 0 = FortiTran */
    int32_t hidden[3];                /* channel_adaptivity_hidden_sizes (adaptive only) */
    int32_t encoder_path;             /* AFT_ENCODER_*: how aft_forward_f32 runs the encoder (same bits either way) */
    int32_t precision;                /* AFT_PRECISION_*: 0 = exact fp32 everywhere (the default and the parity contract) */
} aft_config;

/* One nn.TransformerEncoderLayer (post-LN), PyTorch layouts
 * (reference blocks/encoders.py:44-55). */
typedef struct aft_layer_weights {
    const float *in_proj_w, *in_proj_b;   /* [3d,d] = [Wq;Wk;Wv], [3d]              */
    const float *out_proj_w, *out_proj_b; /* [d,d], [d]                              */
    const float *lin1_w, *lin1_b;         /* [2d,d], [2d]                            */
    const float *lin2_w, *lin2_b;         /* [d,2d], [d]                             */
    const float *norm1_w, *norm1_b, *norm2_w, *norm2_b; /* [d] each                  */
} aft_layer_weights;

/* Every tensor of the reference state_dict, as device pointers to the tensors
 * PyTorch already holds (no repacking, no ownership transfer). */
typedef struct aft_weights {
    const float *up_w, *up_b;        /* pilot_upsampler [S*T, Ps*Pt], [S*T]  (fortitran.py:86) */
    const float *enh_w[4], *enh_b[4];/* initial_enhancer.conv_block.{0,2,4,6} (enhancers.py:12-20) */
    const float *ref_w[4], *ref_b[4];/* final_refiner.conv_block.{0,2,4,6}                   */
    const float *ada_w[3][3], *ada_b[3][3]; /* channel_adapter.{snr,ds,dop}_encoder.{0,2,4}
                                               (channel_adaptivity.py:35-39); NULL if !adaptive */
    const float *lin1_w, *lin1_b;    /* transformer_encoder.linear_1 [d, p(+6)], [d]          */
    const float *pos;                /* positional table rows [>=tokens, d] (learnable or sinusoid) */
    const float *lin2_w, *lin2_b;    /* transformer_encoder.linear_2 [p, d], [p]              */
    const aft_layer_weights *layers; /* HOST array of cfg->num_layers entries (any layer count: nn.TransformerEncoder stacks
                                        whatever num_layers says, encoders.py:52-55); read during the call only            */
} aft_weights;

int aft_version(void);
const char *aft_last_error(void);

/* AFT_OK when the gfx950 kernels cover `cfg`; else AFT_ERR_SHAPE / AFT_ERR_ARG with the reason in
 * aft_last_error().  The estimator calls it at construction (the reference validates its config in
 * __init__, fortitran.py:52-81), so an uncovered shape is refused before any training starts. */
int aft_check_config(const aft_config *cfg);

/* Which launch sequence aft_forward_f32 runs for `cfg` (a property of the configuration alone -- never of the batch, so a frame's
 * output bits do not depend on the batch it travels in): AFT_ENGINE_PACKED = the tuned fragment-packed kernels (chain + attention,
 * DESIGN.md 4.1 / 4.2), AFT_ENGINE_GENERAL = the row-major sequence that covers every shape the reference builds within the bounds
 * above (GEMM / attention / LayerNorm kernels of the training path, dropout off); < 0 = not covered (aft_check_config says why). */
#define AFT_ENGINE_PACKED 0
#define AFT_ENGINE_GENERAL 1
int aft_engine_of(const aft_config *cfg);

/* Measurement / A-B switches (no reference counterpart).  The library reads every environment variable that starts with "AFT_" ONCE,
 * when it is loaded; afterwards a switch changes only through aft_set_switch (value NULL = unset) -- no getenv() on any call path, so
 * forwards on several host threads never race a setenv() elsewhere in the process.  aft_get_switch copies the current value into
 * buf (at most n bytes including the terminator) and returns 1, or returns 0 when the switch is unset.  The switches in use are
 * listed in DESIGN.md section 5; none of them changes results beyond summation order, and the product never sets one. */
int aft_set_switch(const char *name, const char *value);
int aft_get_switch(const char *name, char *buf, size_t n);

/* Largest `batch` one aft_forward_f32 call accepts for `cfg` (0 on a bad config): the kernels use 32-bit byte offsets
 * into each workspace region, so no region (q / k / v^T: 2*batch*tokpad*model_dim floats) may reach 2 GiB.  The reference
 * accepts any batch (fortitran.py:145-182); the module surface above this ABI splits larger batches into chunks. */
int aft_max_batch(const aft_config *cfg);

/* Bytes of scratch aft_forward_f32 needs for `batch` frames (0 on a bad config). */
size_t aft_workspace_bytes(const aft_config *cfg, int batch);

/* Lanes (no reference counterpart).  A forward whose launches do not fill whole rounds of the kernels' persistent grids -- more than
 * one row tile per CU, fewer than ~15, and not as well aligned as the default model's 127 / 128 frames (DESIGN.md section 5 has the
 * rule and the measurements) -- is run as TWO complete forwards over contiguous shares of the batch: share 0 on the
 * caller's stream, share 1 on a library-owned side stream that is forked from the caller's stream by an event when the call starts
 * and joined back into it by an event before the call returns -- the hardware fills the idle tail of one share's launch with the
 * other share's next launch.  To the caller the call is still asynchronous on ONE stream (everything the call enqueues is ordered
 * after the stream's earlier work and before its later work; capturable in a hipGraph: the side stream joins the capture).  Frames
 * are independent, so the output bits are those of the unsplit forward.  The side stream and its two events are created on the
 * first such call per (device, caller stream) and kept for the 16 most recently used caller streams of the process (the least
 * recently used idle entry is destroyed when a seventeenth stream arrives).  Should the side stream be unavailable (creation failed,
 * or all 16 entries are in use by calls in flight on other host threads) the shares run one after the other on the caller's stream
 * in the SAME workspace layout -- slower, same bits, and aft_workspace_lanes stays true.  The switch AFT_LANES=1 (aft_set_switch)
 * turns the split off.
 * aft_workspace_lanes reports the split a forward of `batch` frames uses: lanes (1 .. AFT_MAX_LANES), and per lane its
 * frame count and the byte offset of its slice of `workspace` (a lane's slice is laid out as the workspace of a forward of that many
 * frames).  `frames` and `offset_bytes` are arrays of AFT_MAX_LANES entries. */
#define AFT_MAX_LANES 4
int aft_workspace_lanes(const aft_config *cfg, int batch, int *lanes, int *frames, size_t *offset_bytes);

/* Where aft_forward_f32 LEAVES its intermediates in `workspace` (no reference counterpart; known-answer tests use it to pin
 * the kernels the forward itself launches -- the per-stage entry points further down do not always run the same kernels):
 * byte offset and byte size of one region of a lane of `batch` frames, relative to that lane's slice of the workspace
 * (aft_workspace_lanes; a forward that is not split has one lane at offset 0).  Regions stay valid until the next call on that workspace.
 *   AFT_REGION_CONV_ENHANCED  f32 [2B,S,T]            output of S1+S2 (fortitran.py:203-209), kept for the S7 residual
 *   AFT_REGION_TOKENS6        f32 [B,tokens,6]        ChannelAdapter output (channel_adaptivity.py:59-63); adaptive configs
 *   AFT_REGION_ENC_OUT        f32 [2B*tokens,stride]  linear_2's output (encoders.py:70), `stride` = the patch element count rounded
 *                                                     up to 8 / 16 / 24 / 32; columns [0, patch elements) are valid
 * Returns AFT_OK, or AFT_ERR_ARG / AFT_ERR_SHAPE (bad region, batch or config). */
#define AFT_REGION_CONV_ENHANCED 0
#define AFT_REGION_TOKENS6 1
#define AFT_REGION_ENC_OUT 2
int aft_workspace_region(const aft_config *cfg, int batch, int region, size_t *offset_bytes, size_t *size_bytes);

/* Replaces BaseFortiTranEstimator.forward (reference src/models/fortitran.py:145-182):
 * pilots complex64 [B,Ps,Pt] -> out complex64 [B,S,T].  snr/ds/dop are float32 [B]
 * raw (un-normalised) channel conditions, NULL for FortiTran (meta_data[1..3],
 * fortitran.py:166-170).
 * `pilots`, `snr`, `ds`, `dop` may be ANY device-addressable memory, pinned host memory (hipHostMalloc / torch's
 * pin_memory()) included: the conv head and the adapter read them directly, so the model-owned H2D transfer of
 * fortitran.py:167-173 needs no copy engine hop in front of the first launch -- the caller keeps the buffer unchanged until
 * the call's kernels have run (record an event behind the call).  `out`, `workspace` and the weights are device memory.
 * Launches: [adapter + weight re-lay] (one prologue launch), conv head, embedding + in-projection, L x (attention, row-local
 * chain), conv tail -- once, or once per lane when a short forward is split ("Lanes" above). */
int aft_forward_f32(const aft_config *cfg, const aft_weights *w, const float *pilots,
                    const float *snr, const float *ds, const float *dop, float *out,
                    void *workspace, size_t workspace_bytes, int batch, void *stream);

/* The same forward for a caller that knows when its parameters change (an nn.Module in eval mode does: reference
 * trainer.py:332 `model.eval()` ... the weights are constant across a whole evaluation sweep).  aft_forward_f32 re-lays
 * the encoder's GEMM weights into MFMA-fragment order on EVERY call (5 us, 3 MB: it is stateless and always reflects the
 * caller's current parameters); here the caller owns that image -- aft_packed_weights_bytes() of device memory, filled by
 * aft_pack_weights_f32 whenever in_proj / out_proj / linear1 / linear2 weights of any layer changed -- and every forward
 * reads it.  `w` is still needed (biases, LayerNorm vectors, conv / adapter / dense weights are read in place).
 * THE CALLER'S DUTY: the image must have been packed with the SAME cfg (model_dim, num_layers AND precision: the fp32 and
 * the split-precision layouts have the same size) from the CURRENT weights; the library cannot check either without a
 * device-to-host synchronisation, which this ABI never performs.  Since ABI 4 aft_forward_f32 packs in the same launch as
 * the channel adapter, so the stateless entry point costs no more than this one: prefer it unless the 3 MB of weight reads
 * per call matter. */
size_t aft_packed_weights_bytes(const aft_config *cfg);
int aft_pack_weights_f32(const aft_config *cfg, const aft_weights *w, void *packed, size_t packed_bytes, void *stream);
int aft_forward_prepacked_f32(const aft_config *cfg, const aft_weights *w, const void *packed, const float *pilots,
                              const float *snr, const float *ds, const float *dop, float *out, void *workspace,
                              size_t workspace_bytes, int batch, void *stream);

/* Replaces LinearEstimator.forward (reference src/models/linear.py:65-97), applied to
 * the Re and Im planes separately: out[b,:,c] = W x[b,:,c] + bias. */
int aft_linear_forward_f32(const float *weight, const float *bias, const float *pilots,
                           float *out, int batch, int in_features, int out_features,
                           void *stream);

/* Replaces the metric of reference src/utils.py:164-180 + trainer.py:338-347:
 * *sum_sq (device, float64) += sum |est-ref|^2 over n_complex complex elements.
 * MSE = sum_sq / n_complex; dB = 10 log10 (utils.py:233-245). */
int aft_mse_partial_f32(const float *est, const float *ref, double *sum_sq,
                        long long n_complex, void *stream);

/* ---- the data formats either side of the path ---- */

/* Replaces the boolean-mask pilot extraction of MatDataset._process_channel_data (reference
 * src/data/dataset.py:116-139): hzero_ls complex64 [B, grid_elems] is the LS estimate with zeros at
 * non-pilot positions; the non-zero entries (re != 0 or im != 0) of each frame are compacted in
 * row-major order into pilots complex64 [B, expected].  counts[b] (device int32) receives the number
 * of non-zero entries found, so the caller can raise the reference's ValueError when it differs from
 * `expected` (dataset.py:128-132); exactly `expected` values are written per frame (zeros behind the
 * entries found when there are fewer: the output needs no initialisation). */
int aft_pilot_gather_f32(const float *hzero_ls, float *pilots, int *counts, int batch, int grid_elems,
                         int expected, void *stream);

/* Replaces utils.mse (reference src/utils.py:248-261) as get_ls_mse_per_folder applies it per file
 * (:264-303): db[b] = 10 log10( mean_i |ls[b,i] - ideal[b,i]|^2 ), complex64 [B, grid_elems] inputs. */
int aft_ls_mse_db_f32(const float *ls, const float *ideal, float *db, int batch, int grid_elems, void *stream);

/* ---- training path of the encoder ---- */

/* Gradients of one nn.TransformerEncoderLayer: same fields and shapes as aft_layer_weights,
 * writable.  These are the .grad tensors autograd hands to the optimizer
 * (reference src/main/trainer.py:195-233, loss.backward()). */
typedef struct aft_layer_grads {
    float *in_proj_w, *in_proj_b, *out_proj_w, *out_proj_b;
    float *lin1_w, *lin1_b, *lin2_w, *lin2_b;
    float *norm1_w, *norm1_b, *norm2_w, *norm2_b;
} aft_layer_grads;

/* Bytes of the per-layer activation tape the training forward fills and the backward reads, and
 * of the scratch either call needs (both 0 on a bad config). */
size_t aft_encoder_tape_bytes(const aft_config *cfg, int batch);
size_t aft_encoder_train_scratch_bytes(const aft_config *cfg, int batch);

/* Replaces nn.TransformerEncoderLayer.forward in train() mode (reference blocks/encoders.py:44-55,
 * called from encoders.py:69): x_in f32 [2B*tokens, d] -> x_out (may not alias x_in, which the
 * backward reads again).  Dropout with probability `dropout_p` at the layer's four sites (attention
 * probabilities, after out_proj, after the activation, after linear2) from a counter-based
 * generator keyed by `seed`; the backward must be given the same seed.  dropout_p = 0 reproduces
 * the eval-mode layer. */
int aft_encoder_layer_fwd_train_f32(const aft_config *cfg, const aft_layer_weights *w, const float *x_in, float *x_out,
                                    void *tape, size_t tape_bytes, void *scratch, size_t scratch_bytes, int batch,
                                    float dropout_p, uint64_t seed, void *stream);

/* The same layer inside a stack (reference blocks/encoders.py:69 loops over the layers): what changes is only WHERE the
 * in-projection of a layer runs.  `qkv_ready` != 0: the previous call already left this layer's q | k | v in `tape`
 * (skip the in-projection GEMM).  `next_w` + `next_tape` (the NEXT layer's parameters and its tape, same cfg and batch) +
 * `next_qkv_written`: when the fused row-local kernel covers the shape, the next layer's in-projection is computed as its
 * tail while the output tile is still in registers and written into `next_tape`; *next_qkv_written says whether that
 * happened (pass it as the next call's qkv_ready).  The backward is unchanged: each layer's aft_encoder_layer_bwd_f32
 * still produces its own in-projection gradients from its own tape. */
int aft_encoder_layer_fwd_train_chained_f32(const aft_config *cfg, const aft_layer_weights *w, const float *x_in, float *x_out,
                                            void *tape, size_t tape_bytes, void *scratch, size_t scratch_bytes, int batch,
                                            float dropout_p, uint64_t seed, int qkv_ready, const aft_layer_weights *next_w,
                                            void *next_tape, size_t next_tape_bytes, int *next_qkv_written, void *stream);

/* Replaces autograd's backward through that layer: dx_out = dL/dx_out [2B*tokens, d] ->
 * dx_in = dL/dx_in (may alias dx_out) and the twelve parameter gradients in `grads`
 * (overwritten, or added to when accumulate != 0). */
int aft_encoder_layer_bwd_f32(const aft_config *cfg, const aft_layer_weights *w, const float *x_in, const void *tape,
                              size_t tape_bytes, const float *dx_out, float *dx_in, const aft_layer_grads *grads,
                              int accumulate, void *scratch, size_t scratch_bytes, int batch, float dropout_p,
                              uint64_t seed, void *stream);

/* nn.Linear in the training path (pilot_upsampler fortitran.py:86, linear_1 / linear_2 encoders.py:33,56,
 * the adapter MLPs channel_adaptivity.py:35-39): y[rows,out] = x[rows,in] W[out,in]^T + b, and its
 * backward dx = dy W (skipped when dx is NULL), dW (+)= dy^T x, db (+)= column sums of dy (NULL: none).
 * Any sizes; the weight gradient's reduction over rows is split and summed in a fixed order. */
int aft_dense_fwd_f32(const float *x, const float *weight, const float *bias, float *y, int rows, int in_features,
                      int out_features, void *stream);
size_t aft_dense_bwd_scratch_bytes(int rows, int in_features, int out_features);
int aft_dense_bwd_f32(const float *x, const float *weight, const float *dy, float *dx, float *dweight, float *dbias,
                      int accumulate, void *scratch, size_t scratch_bytes, int rows, int in_features, int out_features,
                      void *stream);

/* ConvEnhancer (reference src/models/blocks/enhancers.py:5-31: conv 1->8->32->8->1, ReLU after the
 * first three) in train() mode on `planes` real planes x, y: f32 [planes, S, T].  weights[k] / biases[k]
 * are conv_block.{0,2,4,6}.{weight,bias} in PyTorch layout.  c1, c2, c3 receive the activations the
 * backward needs: f32 [planes, C, T, S] with C = 8, 32, 8 (an internal layout; treat as opaque).
 * `scratch`: aft_conv_enhancer_fwd_scratch_bytes() bytes the call may overwrite (the weights re-laid as MFMA operand fragments for the
 * default grid's kernel; ABI 7), or NULL = the kernels that read the weights in place. */
size_t aft_conv_enhancer_fwd_scratch_bytes(int planes, int num_scs, int num_symbols);
int aft_conv_enhancer_fwd_train_f32(const float *const weights[4], const float *const biases[4], const float *x, float *y,
                                    float *c1, float *c2, float *c3, void *scratch, size_t scratch_bytes, int planes, int num_scs,
                                    int num_symbols, void *stream);

/* Backward of that call: dy = dL/dy -> dx = dL/dx and the eight parameter gradients (PyTorch layouts;
 * overwritten, or added to when accumulate != 0).  weights[k] are the module's conv weights as in the forward call:
 * the data gradient of the stack is the stack itself run on dy with weights[3-k].transpose(0,1).flip(2,3), which the
 * call lays out in its scratch (one small kernel).
 * aft_conv_enhancer_scratch_bytes returns 0 for a grid the fused kernel has no LDS band plan for. */
size_t aft_conv_enhancer_scratch_bytes(int planes, int num_scs, int num_symbols);
int aft_conv_enhancer_bwd_f32(const float *const weights[4], const float *x, const float *c1, const float *c2,
                              const float *c3, const float *dy, float *dx, float *const dweights[4], float *const dbiases[4],
                              int accumulate, void *scratch, size_t scratch_bytes, int planes, int num_scs, int num_symbols,
                              void *stream);

/* ChannelAdapter (reference src/models/blocks/channel_adaptivity.py:24-63) in train() mode: conditions[e] f32
 * [frames] (snr, delay spread, doppler), weights / biases in the order {snr,ds,dop}_encoder.{0,2,4}; tokens6 f32
 * [frames, tokens, 6]; hidden0 / hidden1 receive the post-ReLU activations [frames,3,h0] / [frames,3,h1] the
 * backward needs.  The backward turns dL/dtokens6 into the 18 parameter gradients (inputs carry no gradient);
 * dhidden0 / dhidden1 are scratch of the same shapes. */
int aft_adapter_fwd_train_f32(const float *const conditions[3], const float *const weights[9], const float *const biases[9],
                              const int32_t hidden[3], int tokens, int frames, float *tokens6, float *hidden0, float *hidden1,
                              void *stream);
int aft_adapter_bwd_f32(const float *const conditions[3], const float *const weights[9], const float *const biases[9],
                        const int32_t hidden[3], int tokens, int frames, const float *hidden0, const float *hidden1,
                        const float *dtokens6, float *dhidden0, float *dhidden1, float *const dweights[9],
                        float *const dbiases[9], int accumulate, void *stream);

/* The two thin ends of the encoder in the training path (ABI 8), each ONE streaming launch per direction instead of PyTorch's
 * unfold / cat / broadcast add around a 6- or 12-column GEMM.  `planes` = 2 x frames in the training composite's order
 * [real planes | imaginary planes]; grid = num_scs x num_symbols, tokens = (num_scs / patch_scs) (num_symbols / patch_symbols),
 * p = patch_scs patch_symbols <= 32, model_dim a multiple of 4 up to 512.
 *   embed  (fortitran.py:212-217, blocks/patch_processors.py:22,34-35, blocks/encoders.py:67-68):
 *     x0[planes*tokens, d] = cat(PatchEmbedding(conv_enhanced), tokens6) W1^T + b1 + pos[:tokens]
 *     conv_enhanced f32 [planes,S,T]; tokens6 f32 [planes,tokens,6] PER PLANE or NULL (then W1 is [d,p], else [d,p+6]); pos f32
 *     [>= tokens, d] (the learnable table's or the sinusoid buffer's first rows).
 *     backward: dx0 -> d_conv_enhanced [planes,S,T] (every element written), d_tokens6 (NULL with tokens6 NULL), dW1, db1, dpos
 *     [tokens,d] (NULL: no table gradient, e.g. the sinusoid); the parameter gradients are overwritten, or added to when
 *     accumulate != 0.
 *   tail   (blocks/encoders.py:70, blocks/patch_processors.py InversePatchEmbedding, fortitran.py:225-227):
 *     out[planes,S,T] = resid + InversePatchEmbedding(x W2^T + b2),  x f32 [planes*tokens, d], W2 [p,d]
 *     backward: d_out [planes,S,T] -> dx [planes*tokens,d], dW2, db2 (the residual's gradient IS d_out: nothing to compute). */
size_t aft_embed_bwd_scratch_bytes(int planes, int num_scs, int num_symbols, int patch_scs, int patch_symbols, int model_dim,
                                   int with_tokens6);
int aft_embed_fwd_train_f32(const float *conv_enhanced, const float *tokens6, const float *w1, const float *b1, const float *pos,
                            float *x0, int planes, int num_scs, int num_symbols, int patch_scs, int patch_symbols, int model_dim,
                            void *stream);
int aft_embed_bwd_f32(const float *conv_enhanced, const float *tokens6, const float *w1, const float *dx0, float *d_conv_enhanced,
                      float *d_tokens6, float *dw1, float *db1, float *dpos, int accumulate, void *scratch, size_t scratch_bytes,
                      int planes, int num_scs, int num_symbols, int patch_scs, int patch_symbols, int model_dim, void *stream);
size_t aft_tail_bwd_scratch_bytes(int planes, int num_scs, int num_symbols, int patch_scs, int patch_symbols, int model_dim);
int aft_tail_fwd_train_f32(const float *x, const float *w2, const float *b2, const float *resid, float *out, int planes, int num_scs,
                           int num_symbols, int patch_scs, int patch_symbols, int model_dim, void *stream);
int aft_tail_bwd_f32(const float *x, const float *w2, const float *d_out, float *dx, float *dw2, float *db2, int accumulate,
                     void *scratch, size_t scratch_bytes, int planes, int num_scs, int num_symbols, int patch_scs, int patch_symbols,
                     int model_dim, void *stream);

/* Replaces torch.optim.Adam.step (reference src/main/trainer.py:407-413; amsgrad off) on one flat
 * float32 shard of n elements: grad is first multiplied by grad_scale (1/world_size after a
 * sum-reduce-scatter), weight_decay is the L2 form Adam uses; `step` is the 1-based step count. */
int aft_adam_step_f32(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, size_t n, float lr, float beta1,
                      float beta2, float eps, float weight_decay, float grad_scale, int step, void *stream);

/* ---- per-stage entry points (known-answer tests) ----
 * Same arithmetic as aft_forward_f32 stage by stage, but NOT always the same kernels: these calls own no scratch, so on the default
 * 120 x 14 grid S1+S2 and the tail run the banded conv kernel with the pilot_upsampler product inside it (k_conv.hip), while the
 * forward runs the product in its prologue launch and the column-streaming conv kernel (k_conv_stream.hip); the embedding stage is
 * its own kernel here and part of the first chain launch there.  The kernels the forward launches are pinned through
 * aft_workspace_region (tests/test_hip_parity.py::test_forward_intermediates_match_golden). */

/* S1+S2 (fortitran.py:203-209): pilots complex64 [B,Ps,Pt] -> conv_enhanced f32 [2B,S,T]. */
int aft_stage_upsample_f32(const aft_config *cfg, const aft_weights *w, const float *pilots,
                           float *conv_enhanced, int batch, void *stream);
/* S4 (channel_adaptivity.py:59-63): snr/ds/dop [B] -> adapter tokens f32 [B,tokens,6]. */
int aft_stage_adapter_f32(const aft_config *cfg, const aft_weights *w, const float *snr,
                          const float *ds, const float *dop, float *tokens6, int batch,
                          void *stream);
/* S3+S4-concat+linear_1+pos (fortitran.py:212-217, encoders.py:67-68):
 * conv_enhanced [2B,S,T] (+ tokens6 [B,tokens,6] or NULL) -> x f32 [2B*tokens, d]. */
int aft_stage_embed_f32(const aft_config *cfg, const aft_weights *w, const float *conv_enhanced,
                        const float *tokens6, float *x, int batch, void *stream);
/* One nn.TransformerEncoderLayer in eval mode, in place on x [2B*tokens, d]
 * (encoders.py:69).  `scratch` needs aft_workspace_bytes(cfg,batch) bytes.  Runs the engine aft_engine_of(cfg) names. */
int aft_stage_encoder_layer_f32(const aft_config *cfg, const aft_weights *w, int layer, float *x,
                                void *scratch, size_t scratch_bytes, int batch, void *stream);
/* linear_2 + S6 + S7 + S8 + complex recombination (encoders.py:70, fortitran.py:225-231,180):
 * x [2B*tokens, d], conv_enhanced [2B,S,T] -> out complex64 [B,S,T].  This call owns no scratch, so linear_2's weights must fit the
 * conv kernel's LDS beside the plane (AFT_ERR_SHAPE otherwise, e.g. 32-element patches at model_dim 512 on the default grid: only
 * aft_forward_f32 serves those). */
int aft_stage_tail_f32(const aft_config *cfg, const aft_weights *w, const float *x,
                       const float *conv_enhanced, float *out, int batch, void *stream);

/* ---- measurement hook (bench.py roofline leg; no reference counterpart) ----
 * Launch ONE kernel class `reps` times back to back on `stream`, on the activations a previous
 * aft_forward_f32 of the same (cfg, batch) left in `workspace`, so the caller can bracket it with
 * events on that stream.  Same kernels, grids and arguments as inside aft_forward_f32. */
#define AFT_KERNEL_UPSAMPLE 0   /* initial ConvEnhancer as the forward launches it (on the planes of the prologue's
                                   pilot_upsampler product where the forward does that; else product + ConvEnhancer fused) */
#define AFT_KERNEL_EMBED 1      /* patch gather + adapter concat + linear_1 + pos as its own kernel (the stage entry
                                   point's; aft_forward_f32 runs it inside AFT_KERNEL_QKV)                     */
#define AFT_KERNEL_QKV 2        /* chain kernel: embedding + in-projection of layer 0        */
#define AFT_KERNEL_ATTENTION 3  /* MFMA attention                                            */
#define AFT_KERNEL_CHAIN 4      /* chain kernel: out-proj+LN1+FFN+LN2 (layer 0) + QKV (layer 1) */
#define AFT_KERNEL_TAIL 5       /* fold + residual + final ConvEnhancer (linear_2 done by AFT_KERNEL_CHAIN_LAST) */
#define AFT_KERNEL_CHAIN_LAST 6 /* chain kernel of the last layer: out-proj+LN1+FFN+LN2 + linear_2             */
#define AFT_KERNEL_ENCODER_PLANE 7 /* plane-resident encoder: embedding + all layers + linear_2 in one launch (k_encoder.hip) */
#define AFT_KERNEL_PROLOGUE 8   /* the forward's first launch: channel adapter + weight re-lay + pilot_upsampler product; `out` = one
                                   buffer [pilots: batch x Ps x Pt x 2 floats | snr: batch | ds: batch | dop: batch]        */
int aft_profile_kernel_f32(const aft_config *cfg, const aft_weights *w, int which, float *out,
                           void *workspace, size_t workspace_bytes, int batch, int reps, void *stream);

/* ---- test hook (no reference counterpart) ----
 * Fill the LDS of every CU of the current device with `value` (a NaN, 1e30 ...): what a kernel finds in its LDS at start is whatever
 * the previous kernel on that CU left there, and a kernel that multiplies a zero weight with an LDS word it never wrote computes
 * 0 x NaN (round 6: embed_any_kernel did; caught only because another test's kernels had run before).  The allocator-poisoning
 * tests cannot reach LDS; this can: workgroups of 160 KB each, enough of them that every CU runs at least one.  Asynchronous on
 * `stream`. */
int aft_debug_fill_lds_f32(float value, void *stream);
/* The other half of the hook: what a kernel that writes nothing finds in its LDS.  `workgroups` workgroups (256 threads, 40 KB of
 * LDS each) copy the first `n` (<= 10240) floats of their LDS to out[workgroup][n] (device memory). */
int aft_debug_peek_lds_f32(float *out, int workgroups, int n, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ADAFORTITRAN_AMD_H */
