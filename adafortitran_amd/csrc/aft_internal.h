// Internal declarations shared by the gfx950 kernels and the C-ABI (not installed).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "../../include/adafortitran_amd.h"

namespace aft {

using f32x2 = __attribute__((ext_vector_type(2))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kWave = 64;       // CDNA4 wavefront
constexpr int kHeadDim = 32;    // model_dim / num_head for every config the kernels cover
constexpr int kTile = 32;       // v_mfma_f32_32x32x2_f32 tile edge
constexpr int kMaxPatchFeatures = 16;
constexpr int kSchedSlots = 2048;       // (xcc id, se id, sh id, cu id) -> one ticket counter per CU   // patch_scs * patch_symbols the embedding kernel unrolls

inline __host__ __device__ int round_up(int v, int m) { return (v + m - 1) / m * m; }

// ---- the CHECKED build (python -m adafortitran_amd.build --variant check -DAFT_CHECKED=1; SURVEY.md section 5 "LDS-bounds asserts in
// debug builds"; GPU AddressSanitizer is not available on this pool).  Never the product: it exists to be run against it.
//   * AFT_DEV_ASSERT: index / slot / ring-offset checks in the wave-specialised conv pipelines; a violation is __builtin_trap -> the
//     launch faults and the next HIP call reports it, instead of a silent wrong LDS word.
//   * AFT_CHECKED_FENCE: the LDS-flag hand-overs of k_conv_stream.hip publish WITHOUT a release fence in the product (gfx950 serves one
//     wave's LDS requests in issue order); the checked build puts the fence back, so "same bits as the product build" on the soak
//     test is a test of that assumption.
//   * AFT_SPIN_GUARD: a poll that does not end within ~2^22 sleeps traps (a lost hand-over becomes a fault, not a hung box).
//   * AFT_HOST_ASSERT: workspace-plan invariants in aft_api.hip (regions inside the workspace, lanes disjoint).
#ifdef AFT_CHECKED
#define AFT_DEV_ASSERT(cond) do { if (!(cond)) __builtin_trap(); } while (0)
#define AFT_CHECKED_FENCE() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup")
#define AFT_SPIN_GUARD_INIT() unsigned aft_spin_ = 0
#define AFT_SPIN_GUARD() do { if (++aft_spin_ > (1u << 22)) __builtin_trap(); } while (0)
#define AFT_HOST_ASSERT(cond, what) do { if (!(cond)) { set_error("checked build: %s", what); return AFT_ERR_ARG; } } while (0)
#else
#define AFT_DEV_ASSERT(cond) ((void)0)
#define AFT_CHECKED_FENCE() ((void)0)
#define AFT_SPIN_GUARD_INIT() ((void)0)
#define AFT_SPIN_GUARD() ((void)0)
#define AFT_HOST_ASSERT(cond, what) ((void)0)
#endif

// fp32 MFMA with one scalar instruction behind it.  Measured (tools/micro/valu_cost.hip short): a v_mfma_f32_32x32x2_f32
// that directly follows another one in the instruction stream issues ~6 cycles late (143 TFLOP/s for back-to-back
// chains, one or two accumulators, 1 or 3 waves per SIMD alike); ONE scalar instruction between them (s_nop 0, s_mov)
// removes the bubble: 156 TFLOP/s = 99 % of the 157.3 TFLOP/s roof.  VALU or LDS instructions in the gap do the same,
// so the s_nop only matters where MFMAs would otherwise be adjacent -- which is every GEMM inner loop here.
__device__ __forceinline__ f32x16 mfma_f32(float a, float b, f32x16 c) {
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
#if defined(__HIP_DEVICE_COMPILE__) && !defined(AFT_NO_MFMA_SPACER)
    asm volatile("s_nop 0" : "+v"(c));   // tied to the accumulator so that it stays between this MFMA and the next of the chain
#endif
    return c;
}

// Wave priority from the work a persistent wave has LEFT (`left` of `total` equal steps; both wave-uniform).
// Why: at equal priority the oldest wave of a SIMD wins every fp32-MFMA issue conflict, so co-resident waves (or
// workgroups) that run the same program advance almost one after the other and the youngest ones finish alone --
// measured with the AFT_STAMPS build: identical attention tasks of one round ended after 20 .. 65 us, 2.4 of 3
// wave slots busy on average.  A wave that gets ahead drops its priority, so the laggards catch up at every band
// edge; the bands shrink geometrically towards the end (1/2, 1/4, 1/8 of the work), which bounds the stagger of the
// finish times to the last eighth.  Priority only arbitrates issue; results do not depend on it.
__device__ __forceinline__ void set_progress_priority(int left, int total) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(AFT_NO_PROGRESS_PRIORITY)
    if (2 * left > total) __builtin_amdgcn_s_setprio(3);
    else if (4 * left > total) __builtin_amdgcn_s_setprio(2);
    else if (8 * left > total) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
#endif
}

// erf(x) = sign(x) * (1 - 2^p(|x|)), p = degree-8 fit of log2(erfc(t)) on [0,4] with p(0) = 0
// (erfc(4) = 1.5e-8 rounds to 0 against 1 in fp32).  Branch-free; max |error| 1e-7 (<= 1 ulp of erf
// near 1) against scipy.special.erf on 4e5 points -- libm erff cost ~40 VALU ops and a divergent
// branch per element.  Two values at once: the polynomial runs on v_pk_fma_f32.
template <int ACT>
__device__ __forceinline__ f32x2 activate2(f32x2 v) {
    if constexpr (ACT == AFT_ACT_GELU) {  // exact-erf GELU (F.gelu default, activation="gelu")
        const f32x2 x = v * 0.70710678118654752440f;
        const f32x2 t = {fminf(fabsf(x[0]), 4.0f), fminf(fabsf(x[1]), 4.0f)};
        f32x2 p = {-4.535924745e-05f, -4.535924745e-05f};
        p = __builtin_elementwise_fma(p, t, f32x2{4.455104063e-04f, 4.455104063e-04f});
        p = __builtin_elementwise_fma(p, t, f32x2{-1.489443355e-03f, -1.489443355e-03f});
        p = __builtin_elementwise_fma(p, t, f32x2{-7.746370393e-04f, -7.746370393e-04f});
        p = __builtin_elementwise_fma(p, t, f32x2{2.825369500e-02f, 2.825369500e-02f});
        p = __builtin_elementwise_fma(p, t, f32x2{-1.484816223e-01f, -1.484816223e-01f});
        p = __builtin_elementwise_fma(p, t, f32x2{-9.184163809e-01f, -9.184163809e-01f});
        p = __builtin_elementwise_fma(p, t, f32x2{-1.627908587e+00f, -1.627908587e+00f});
        p = p * t;
        const f32x2 e = {copysignf(1.0f - __builtin_amdgcn_exp2f(p[0]), x[0]),
                         copysignf(1.0f - __builtin_amdgcn_exp2f(p[1]), x[1])};
        const f32x2 hv = v * 0.5f;
        return __builtin_elementwise_fma(hv, e, hv);
    } else {  // ReLU (schemas.py:128-131 allows both)
        return f32x2{fmaxf(v[0], 0.0f), fmaxf(v[1], 0.0f)};
    }
}

// d act(v) / dv for the same two activations (GELU: Phi(v) + v phi(v), Phi from the erf polynomial above)
template <int ACT>
__device__ __forceinline__ f32x2 activate2_grad(f32x2 v) {
    if constexpr (ACT == AFT_ACT_GELU) {
        const f32x2 x = v * 0.70710678118654752440f;
        const f32x2 t = {fminf(fabsf(x[0]), 4.0f), fminf(fabsf(x[1]), 4.0f)};
        f32x2 p = {-4.535924745e-05f, -4.535924745e-05f};
        p = __builtin_elementwise_fma(p, t, f32x2{4.455104063e-04f, 4.455104063e-04f});
        p = __builtin_elementwise_fma(p, t, f32x2{-1.489443355e-03f, -1.489443355e-03f});
        p = __builtin_elementwise_fma(p, t, f32x2{-7.746370393e-04f, -7.746370393e-04f});
        p = __builtin_elementwise_fma(p, t, f32x2{2.825369500e-02f, 2.825369500e-02f});
        p = __builtin_elementwise_fma(p, t, f32x2{-1.484816223e-01f, -1.484816223e-01f});
        p = __builtin_elementwise_fma(p, t, f32x2{-9.184163809e-01f, -9.184163809e-01f});
        p = __builtin_elementwise_fma(p, t, f32x2{-1.627908587e+00f, -1.627908587e+00f});
        p = p * t;
        const f32x2 e = {copysignf(1.0f - __builtin_amdgcn_exp2f(p[0]), x[0]),
                         copysignf(1.0f - __builtin_amdgcn_exp2f(p[1]), x[1])};
        const f32x2 q = v * v * -0.72134752044448170368f;   // -v^2 / 2 in base 2
        const f32x2 phi = {0.3989422804014327f * __builtin_amdgcn_exp2f(q[0]), 0.3989422804014327f * __builtin_amdgcn_exp2f(q[1])};
        return __builtin_elementwise_fma(v, phi, __builtin_elementwise_fma(e, f32x2{0.5f, 0.5f}, f32x2{0.5f, 0.5f}));
    } else {
        return f32x2{v[0] > 0.f ? 1.f : 0.f, v[1] > 0.f ? 1.f : 0.f};
    }
}

// Dropout masks of the training path (every site: attention probabilities, the two residual branches, the FFN hidden
// layer).  A murmur-quality hash per element costs three quarter-rate 32-bit multiplies and ~10 more VALU instructions;
// inside an MFMA kernel that is matrix time (the attention forward spent 70 % of a tile's MFMA time on it).  The mask is
// therefore FACTORED: one strong 24-bit word per row and one per column of the masked matrix (hashed once per row / lane
// / workgroup), and element (r, c) is KEPT when the low 32 bits of their product (v_mul_u32_u24, full rate) reach
// p * 2^32: three VALU instructions per element.  The middle bits of a product of two random odd 24-bit words are
// uniform and pairwise uncorrelated across rows, columns and 2x2 rectangles to within sampling noise
// (tests/test_dropout_mask.py).  Forward and backward kernels of a site evaluate the same function of
// (seed, row, column), so nothing is stored.
__device__ __forceinline__ uint32_t dropmask_mix32(uint32_t x) {   // murmur3 finaliser
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t dropmask_row_word(uint32_t seed, uint32_t row) {
    return (dropmask_mix32(row * 0x9E3779B1u ^ seed) >> 8) | 1u;
}
__device__ __forceinline__ uint32_t dropmask_col_word(uint32_t seed, uint32_t col) {
    return (dropmask_mix32(col * 0x9E3779B1u ^ (~seed * 0x632BE5ABu + 0x7F4A7C15u)) >> 8) | 1u;
}
__device__ __forceinline__ bool dropmask_keep(uint32_t row_word, uint32_t col_word, uint32_t threshold) {
    return __umul24(row_word, col_word) >= threshold;
}

// Device-resident scratch of one forward call; all offsets in floats, 256-B aligned.
// Layout in HBM (DESIGN.md "data layout"):
//   conv_enhanced [2B][S][T]            f32  kept for the S7 residual
//   tokens6       [B][tokens][6]        f32  adapter features (adaptive only)
//   x             [2B*tokens][d]        f32  token activations: row-major (stage entry points, plane-resident path) or, in the
//                                            whole-forward launch sequence, tile-blocked [tile][feature block][s][lane][4] (ChainArgs::x_blocked)
//   attn          [ceil(2B*tokens/32)][H][4 s][64 lanes][4]  f32  attention output in operand-fragment order
//   q, k          [2B][H][tokpad/32][4 s][64 lanes][4]  f32  MFMA-fragment order: (key%32 + 32hh, d = 8s+4hh+j)
//   vt            [2B][H][tokpad/32][4 g][64 lanes][4]  f32  fragment order: (d + 32hh, key = 32kt+8g+4hh+j)
//   wpack         [L][8*d*d]            f32  encoder GEMM weights in MFMA-fragment order (rebuilt per call)
//   out6          [2B*tokens][8 | 16]   f32  linear_2 output of the last chain launch (input of the conv tail)
//   convfrag      [2][22*64*4 + 160]    f32  conv2 / conv3 weights of the initial enhancer | the final refiner as 16x16x4 MFMA
//                                            operand fragments (conv_device.h: conv_frag16_entry; rebuilt per call by the prologue)
struct Workspace {
    size_t conv_enhanced, tokens6, x, attn, q, k, vt, wpack, out6, convfrag, total_floats;
    // general engine only (row-major tensors of ONE layer, re-used layer after layer): x1 = LN1 output, y = projection outputs
    // before the residual joins, s / stats = pre-norm sum and (mean, rstd) that launch_add_ln_fwd also writes, qkv [rows][3d],
    // lse, a / hd = FFN pre-activation and activation [rows][2d], pad = padded-head images of qkv and o
    size_t g_x1, g_y, g_s, g_stats, g_qkv, g_lse, g_a, g_hd, g_pad;
    int tokens, tokpad, planes;
};

Workspace plan_workspace(const aft_config &c, int batch);

// error plumbing (thread-local message, see aft_api.hip)
void set_error(const char *fmt, ...);
int check_config(const aft_config *c);
// The launch sequence a configuration runs (aft_engine_of): decided by the configuration alone.
bool packed_engine_ok(const aft_config &c);

// Measurement / A-B switches (aft_set_switch in the header): a table filled ONCE from the "AFT_*" environment variables when the
// library is loaded and changed only through the ABI afterwards -- nothing on a call path calls getenv().
bool switch_on(const char *name);                 // set (to anything)
int switch_int(const char *name, int dflt);       // atoi of the value, dflt when unset

// The pointer table kernels receive BY VALUE: the public aft_weights with a window of at most kLayerWindow layers inline (the public
// struct holds a host pointer to any number of layers; kernel arguments cannot follow it).
constexpr int kLayerWindow = 32;
struct WeightsDev {
    const float *up_w, *up_b;
    const float *enh_w[4], *enh_b[4];
    const float *ref_w[4], *ref_b[4];
    const float *ada_w[3][3], *ada_b[3][3];
    const float *lin1_w, *lin1_b;
    const float *pos;
    const float *lin2_w, *lin2_b;
    aft_layer_weights layers[kLayerWindow];
};
// everything but the layers + layers [first, first + count) of w.layers at layers[0 ..)
WeightsDev weights_window(const aft_weights &w, int first, int count);

// Per-DEVICE facts (aft_api.hip).  The library keeps no state that describes a call, but two things
// are properties of a device, not of a call: its CU count (persistent grids are sized to it) and the
// "this kernel may use N bytes of dynamic LDS" function attribute, which HIP keeps per device.  Both
// live in tables indexed by the current device ordinal, so a process that drives several GPUs through
// this ABI gets each of them configured; entries are idempotent (a race repeats the same call).
constexpr int kMaxDevices = 64;
int current_device();        // hipGetDevice, clamped to [0, kMaxDevices)
int current_device_cus();    // multiprocessor count of the current device (cached per device)
struct PerDeviceOnce {       // one flag per device; `static PerDeviceOnce x;` inside a launcher
    unsigned long long done = 0;
    bool needed(int dev) const { return !((__atomic_load_n(&done, __ATOMIC_RELAXED) >> dev) & 1ull); }
    void mark(int dev) { __atomic_fetch_or(&done, 1ull << dev, __ATOMIC_RELAXED); }
};
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device)
hipError_t ensure_dynamic_lds(PerDeviceOnce &once, const void *kernel, size_t bytes);

// ---- kernel launchers (each enqueues on `st`, returns hipError_t of the launch) ----
// scratch_planes (optional, 2 B * S * T floats, free to overwrite): grids other than the default one compute the upsampler as one
// product over all planes into it and the conv head reads the planes; NULL = inside the conv head
// planes_ready: scratch_planes already holds the upsampled planes (the forward's prologue launch computed them)
// conv_frag: this stack's fragment image (kConvFragFloats floats, written by launch_prologue) or NULL
constexpr size_t kConvFragFloats = 22 * 64 * 4 + 160;   // = conv_device.h kFragFloats, per ConvEnhancer (16-byte multiple)
hipError_t launch_upsample(const aft_config &c, const WeightsDev &w, const float *pilots,
                           float *conv_enhanced, int batch, hipStream_t st, float *scratch_planes = nullptr, bool planes_ready = false,
                           const float *conv_frag = nullptr);
hipError_t launch_adapter(const aft_config &c, const WeightsDev &w, const float *snr, const float *ds,
                          const float *dop, float *tokens6, int batch, hipStream_t st);
// Whole forward: adapter (if c.adaptive), the weight re-pack (if packed != NULL, all layers) and the pilot_upsampler product over all
// planes (if up_planes != NULL and prologue_upsample_ok: [2 batch][S*T] floats) as ONE launch (k_misc.hip)
bool prologue_upsample_ok(const aft_config &c, const WeightsDev &w);
// conv_frag != NULL: + both ConvEnhancers' conv2 / conv3 weights as 16x16x4 operand fragments (2 x kConvFragFloats floats)
hipError_t launch_prologue(const aft_config &c, const WeightsDev &w, const float *snr, const float *ds, const float *dop,
                           float *tokens6, int batch, float *packed, const float *pilots, float *up_planes, hipStream_t st,
                           float *conv_frag = nullptr);
hipError_t launch_embed(const aft_config &c, const WeightsDev &w, const float *conv_enhanced,
                        const float *tokens6, float *x, int batch, hipStream_t st);
// the same stage for ANY model_dim and patches of up to kMaxPatchGeneral elements (general engine, k_misc.hip)
constexpr int kMaxPatchGeneral = 32;
hipError_t launch_embed_any(const aft_config &c, const WeightsDev &w, const float *conv_enhanced,
                            const float *tokens6, float *x, int batch, hipStream_t st);
// Row-local chain on [rows, d]: (mlp) x <- LN2(x1 + FFN(x1)), x1 = LN1(x + attn Wo^T + bo);
// (qkv) q,k,vt <- split(x Wqkv^T + b).  `mlp_w` may be NULL (QKV only), `qkv_w` may be NULL.
// `*_packed` = that layer's block of the fragment-packed weight image (launch_pack_weights).
// Optional work fused into the first / last chain launch of a forward (k_chain.hip, ChainArgs):
//   first (QKV only)  : x0 = embed(conv_enhanced, tokens6) computed in the kernel and written to x   (set conv_enhanced)
//   last (no QKV)     : out6 = linear_2(x2) written instead of x                                     (set out6)
struct ChainFusion {
    const float *conv_enhanced = nullptr, *tokens6 = nullptr, *lin1_w = nullptr, *lin1_b = nullptr, *pos = nullptr;
    const float *lin2_w = nullptr, *lin2_b = nullptr;
    float *out6 = nullptr;
    bool x_blocked = false;   // x between the launches in tile-blocked order (ChainArgs::x_blocked): the whole-forward sequence only
};
inline int out6_stride(const aft_config &c) { return round_up(c.patch_scs * c.patch_symbols, 8); }   // 8 | 16 (packed engine); 24 | 32 too (general)
hipError_t launch_chain(const aft_config &c, const aft_layer_weights *mlp_w, const float *mlp_packed,
                        const aft_layer_weights *qkv_w, const float *qkv_packed, const float *attn, float *x,
                        float *q, float *k, float *vt, int rows, int tokens, int tokpad, hipStream_t st,
                        const ChainFusion *fuse = nullptr);
size_t packed_layer_floats(int d);
// Plane-resident encoder (k_encoder.hip): embedding + all layers + linear_2 of every plane in ONE launch, one 12-wave
// workgroup per plane.  encoder_plane_ok: the shape is instantiated (d = 128).
bool encoder_plane_ok(const aft_config &c);
hipError_t launch_encoder_plane(const aft_config &c, const WeightsDev &w, const float *wpack, const float *conv_enhanced,
                                const float *tokens6, float *x, float *attn, float *q, float *k, float *vt, float *out6,
                                int planes, int tokens, int tokpad, hipStream_t st);
// Re-lay the encoder GEMM weights of layers [first, first+count) into MFMA-fragment order.
// `layers`: HOST array of `count` layers; the image starts at layers[0]'s block (any count: windows of kLayerWindow per launch)
hipError_t launch_pack_weights(const aft_config &c, const aft_layer_weights *layers, float *packed, int count, hipStream_t st);
// qbias = the layer's in_proj_bias (first d entries are the query bias, applied at fragment load).
hipError_t launch_attention(const aft_config &c, const float *q, const float *k, const float *vt, const float *qbias,
                            float *attn, int planes, int tokens, int tokpad, hipStream_t st);
// true when the fused conv-stack kernel has an LDS band plan for an S x T grid with `extra_floats` of side data
bool conv_plan_ok(int S, int T, int extra_floats);
// x = encoder output [rows][d] (linear_2 applied here), or NULL with out6 = linear_2 output [rows][out6_stride(c)]
hipError_t launch_tail(const aft_config &c, const WeightsDev &w, const float *x, const float *conv_enhanced,
                       float *out, int batch, hipStream_t st, const float *out6 = nullptr, const float *conv_frag = nullptr);
hipError_t launch_linear(const float *weight, const float *bias, const float *pilots, float *out, int batch,
                         int in_features, int out_features, hipStream_t st);
hipError_t launch_pilot_gather(const float *hzero_ls, float *pilots, int *counts, int batch, int grid_elems,
                               int expected, hipStream_t st);
hipError_t launch_ls_mse_db(const float *ls, const float *ideal, float *db, int batch, int grid_elems, hipStream_t st);
hipError_t launch_mse(const float *est, const float *ref, double *sum_sq, long long n_complex, hipStream_t st);
hipError_t launch_fill_lds(float value, hipStream_t st);   // test hook: every CU's LDS filled with `value`
hipError_t launch_peek_lds(float *out, int workgroups, int n, hipStream_t st);   // ... and what a kernel finds in its LDS at start


// ---- training path: row-major GEMMs, attention with saved LSE, row-wise pieces ----
constexpr int kGemmMaxSlices = 256;   // split of the token-row reduction in weight gradients
constexpr int kColsumMaxSlices = 1024; // ... in bias / LayerNorm-parameter gradients (one slice per workgroup)
int colsum_slices(int rows);
int gemm_split_slices(int rows, int tiles);
size_t gemm_tn_slice_floats(int M, int N, int R);   // slice storage launch_gemm_tn needs
// While a ReduceBatchScope is alive on this thread, launch_reduce_slices* only queue their job (each producer must
// then own its slice storage until flush()); flush() runs all queued reductions in one launch.
constexpr int kMaxReduceJobs = 12;
struct ReduceJobs;
struct ReduceBatchScope {
    ReduceJobs *jobs;
    ReduceBatchScope();
    ~ReduceBatchScope();
    hipError_t flush(hipStream_t st);
    ReduceBatchScope(const ReduceBatchScope &) = delete;
    ReduceBatchScope &operator=(const ReduceBatchScope &) = delete;
};
int ln_bwd_blocks(int rows);
// op 0: C = A[M][K] B[N][K]^T + bias;  op 1: C = A[M][K] B[K][N]   (accumulate: C += ...)
hipError_t launch_gemm(int op, const float *A, const float *B, float *C, const float *bias, int M, int N, int K, int lda,
                       int ldb, int ldc, bool accumulate, hipStream_t st);
hipError_t launch_gemm_tn(const float *A, const float *B, float *C, float *slices, int M, int N, int R, int lda, int ldb,
                          bool accumulate, hipStream_t st);
// out = LayerNorm(res + drop(A W^T + bias)) * gamma + beta with s = res + drop(..) and (mean, rstd) kept, in ONE launch
// (N = 128 and aligned operands: gemm_add_ln_ok); `threshold` / `keep_scale` as in launch_add_ln_fwd (0 / 1 = no dropout)
bool gemm_add_ln_ok(int M, int N, int K, int lda, int ldb);
hipError_t launch_gemm_add_ln(const float *A, const float *W, const float *bias, const float *res, const float *gamma,
                              const float *beta, float *s_out, float *stats, float *out, int M, int N, int K, int lda, int ldw,
                              float eps, float keep_scale, uint32_t threshold, uint32_t seed, hipStream_t st);
// a_out = A W^T + bias and hd = drop(act(a_out)) (both [M][N], leading dimension ldc) in ONE launch (gemm_act_ok: aligned
// operands, N <= 512)
bool gemm_act_ok(int M, int N, int K, int lda, int ldb, int ldc);
hipError_t launch_gemm_act(const float *A, const float *W, const float *bias, float *a_out, float *hd, int M, int N, int K, int lda,
                           int ldw, int ldc, int activation, float keep_scale, uint32_t threshold, uint32_t seed, hipStream_t st);
// da = (dy W) o dropout mask o act'(a): the data gradient of linear2 with the activation backward as its epilogue
// (W [K][N] row-major as launch_gemm op 1; a, da [M][N], leading dimension ldc; gemm_act_ok-style shapes, N <= 512)
bool gemm_actbwd_ok(int M, int N, int K, int lda, int ldb, int ldc);
hipError_t launch_gemm_actbwd(const float *dy, const float *W, const float *a, float *da, int M, int N, int K, int lda, int ldw,
                              int ldc, int activation, float keep_scale, uint32_t threshold, uint32_t seed, hipStream_t st);
// up to four weight gradients over the same R token rows in ONE GEMM launch; slices[j] holds gemm_tn_slice_floats(M[j], N[j], R)
// floats (the batch never uses more slices than the single launches)
// colsum_out[j] != NULL: also db_j[M_j] (+)= column sums of A_j (the bias gradient beside dW_j), slices in colsum_slices[j]
// (colsum_slices(R) * M_j floats: enough for the batched path and for the fallback's launch_colsum)
hipError_t launch_gemm_tn_batch(const float *const *A, const float *const *B, float *const *C, float *const *slices, const int *M,
                                const int *N, const int *lda, const int *ldb, int n, int R, bool accumulate, hipStream_t st,
                                float *const *colsum_out = nullptr, float *const *colsum_slices = nullptr);
hipError_t launch_colsum(const float *x, float *out, float *slices, int rows, int n, int ld, bool accumulate, hipStream_t st);
hipError_t launch_reduce_slices(const float *slices, float *out, int n, int nz, size_t stride, bool accumulate, hipStream_t st);
hipError_t launch_reduce_slices3(const float *slices, float *out0, float *out1, float *out2, int n, int nout, int nz,
                                 size_t stride, bool accumulate, hipStream_t st);
// Fused row-local backward of an encoder layer (k_chain_bwd.hip): LayerNorm-2 backward, linear2 / linear1 data gradients with
// the activation backward between them, LayerNorm-1 backward, out_proj data gradient -- one launch (+ the transposed weight
// pack).  packed_t: chain_bwd_packed_floats(d) floats of scratch; lnp: chain_bwd_lnp_floats(rows, d) floats that receive the
// per-workgroup column sums [chain_bwd_blocks(rows)][4][d] = dgamma2 | dbeta2 | dgamma1 | dbeta1 (reduce with stride 4 d).
bool chain_bwd_ok(const aft_config &c, int rows);
size_t chain_bwd_packed_floats(int d);
size_t chain_bwd_lnp_floats(int rows, int d);
int chain_bwd_blocks(int rows);   // workgroups of the launch = slices in lnp
hipError_t launch_chain_bwd(const aft_config &c, const aft_layer_weights &w, const float *g, const float *s2, const float *st2,
                            const float *a_pre, const float *s1, const float *st1, float *packed_t, float *g2, float *gff,
                            float *g2b, float *d_o, float *dx, float *lnp, int rows, uint32_t seed1, uint32_t seed2,
                            uint32_t seed3, uint32_t threshold, float keep_scale, hipStream_t st);
// Fused row-local training forward of a layer (k_chain_bwd.hip): out_proj + LN1 + FFN + LN2 with the tape written from the epilogues
bool chain_fwd_train_ok(const aft_config &c, int rows);
hipError_t launch_chain_fwd_train(const aft_config &c, const aft_layer_weights &w, const float *attn, const float *x, float *packed,
                                  float *s1, float *st1, float *x1, float *a_pre, float *hd, float *s2, float *st2, float *x_out,
                                  int rows, uint32_t seed1, uint32_t seed2, uint32_t seed3, uint32_t threshold, float keep_scale,
                                  hipStream_t st, const float *next_in_proj_w = nullptr, const float *next_in_proj_b = nullptr,
                                  float *next_qkv = nullptr);   // all three set: + the next layer's in-projection, row-major [rows][3 d]
// pad: attn_train_pad_floats(c, rows) floats of scratch (head dim 16: the operands re-laid with every head padded to 32 features), else NULL
size_t attn_train_pad_floats(const aft_config &c, size_t rows);
hipError_t launch_attn_train_fwd(const aft_config &c, const float *qkv, float *o, float *lse, int planes, int tokens,
                                 float dropout_p, uint32_t seed, hipStream_t st, float *pad = nullptr);
hipError_t launch_attn_train_bwd(const aft_config &c, const float *qkv, const float *o, const float *d_o, const float *lse,
                                 float *dsum, float *dqkv, int planes, int tokens, float dropout_p, uint32_t seed,
                                 hipStream_t st, float *pad = nullptr);
hipError_t launch_add_ln_fwd(const float *res, const float *y, const float *gamma, const float *beta, float *s_out,
                             float *stats, float *out, int rows, int n, float eps, float dropout_p, uint32_t seed,
                             hipStream_t st);
hipError_t launch_ln_bwd(const float *dy, const float *s, const float *stats, const float *gamma, float *ds, float *dbranch,
                         float *dgamma, float *dbeta, float *dbias, float *slices, int rows, int n, float dropout_p,
                         uint32_t seed, bool accumulate, hipStream_t st);
hipError_t launch_act_fwd(int act, const float *a, float *h, int rows, int n, float dropout_p, uint32_t seed, hipStream_t st);
hipError_t launch_act_bwd(int act, const float *a, float *dh, float *dbias, float *slices, int rows, int n, float dropout_p,
                          uint32_t seed, bool accumulate, hipStream_t st);
hipError_t launch_add(const float *a, const float *b, float *out, size_t n, hipStream_t st);
// ConvEnhancer training forward / backward-dgrad on plain planes (k_conv.hip) and its weight gradients (k_conv_train.hip)
// frag: kConvFragFloats floats of scratch for the default grid's 16x16x4 training kernel (NULL: the 32x32x2 kernels)
hipError_t launch_conv_train(const float *const w[4], const float *const b[4], const float *x, float *y, float *const save[3],
                             const float *const mask[3], int planes, int S, int T, hipStream_t st, float *frag = nullptr);
constexpr int kConvFlipFloats = 72 + 2304 + 2304 + 72;   // conv4^T | conv3^T | conv2^T | conv1^T
hipError_t launch_conv_flip_weights(const float *const w[4], float *dst, hipStream_t st);
hipError_t launch_conv_wgrad(const float *x, const float *c1, const float *c2, const float *c3, const float *g1,
                             const float *g2, const float *g3, const float *dy, float *const dw[4], float *const db[4],
                             float *slices, int planes, int S, int T, bool accumulate, hipStream_t st);
size_t conv_wgrad_slice_floats(int planes, int S, int T);
hipError_t launch_adapter_train_fwd(const float *const cond[3], const float *const w[9], const float *const b[9],
                                    const int hidden[3], int tokens, int frames, float *tokens6, float *a0, float *a1,
                                    hipStream_t st);
hipError_t launch_adapter_train_bwd(const float *const cond[3], const float *const w[9], const float *const b[9],
                                    const int hidden[3], int tokens, int frames, const float *a0, const float *a1,
                                    const float *dtok, float *da0, float *da1, float *const dw[9], float *const db[9],
                                    bool accumulate, hipStream_t st);
// k_ends_train.hip: patch embedding + linear_1 + positions, and linear_2 + inverse patch embedding + residual, forward and backward
bool ends_train_ok(int planes, int S, int T, int p0, int p1, int d, bool adapter);
size_t embed_bwd_slice_floats(int planes, int S, int T, int p0, int p1, int d, bool adapter);
size_t tail_bwd_slice_floats(int planes, int S, int T, int p0, int p1, int d);
// (tok6_per_frame: adapter features [frames][tokens][6] shared by a frame's two INTERLEAVED planes -- the general engine's forward)
hipError_t launch_embed_train_fwd(const float *conv, const float *tok6, const float *w1, const float *b1, const float *pos, float *x,
                                  int planes, int S, int T, int p0, int p1, int d, hipStream_t st, bool tok6_per_frame = false);
hipError_t launch_embed_train_bwd(const float *conv, const float *tok6, const float *w1, const float *dx, float *d_conv, float *d_tok6,
                                  float *dw1, float *db1, float *dpos, bool accumulate, float *slices, int planes, int S, int T, int p0,
                                  int p1, int d, hipStream_t st);
hipError_t launch_tail_train_fwd(const float *x, const float *w2, const float *b2, const float *resid, float *out, int planes, int S, int T,
                                 int p0, int p1, int d, hipStream_t st);
hipError_t launch_tail_train_bwd(const float *x, const float *w2, const float *d_out, float *dx, float *dw2, float *db2, bool accumulate,
                                 float *slices, int planes, int S, int T, int p0, int p1, int d, hipStream_t st);
hipError_t launch_adam(float *p, const float *g, float *m, float *v, size_t n, float lr, float b1, float b2, float eps,
                       float wd, float grad_scale, int step, hipStream_t st);

}  // namespace aft
