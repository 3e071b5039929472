// k_encoder.hip -- the whole transformer encoder of ONE plane inside one workgroup (plane-resident encoder).
//
// Reference semantics: TransformerEncoderForChannels.forward (reference src/models/blocks/encoders.py:58-70):
// linear_1 + positional table, L x nn.TransformerEncoderLayer (:44-55), linear_2 -- for every real plane of the batch
// (fortitran.py:176-177 runs the Re and Im planes as two passes; here they are planes 2f and 2f+1).
//
// Planes never exchange data, so a plane's six layers need no kernel boundary at all: the 13 launches of the
// layer-by-layer path (k_chain.hip / k_attn.hip; launch ramps and tails, row tiles that do not divide the resident
// grid, every x / q / k / v^T / attention tile crossing HBM between launches) become ONE launch in which a 12-wave
// workgroup owns a plane from the patch embedding to linear_2:
//   * the device code IS the layer-by-layer path's (chain_device.h, attn_device.h): three wave groups of W = d/32
//     waves each walk the plane's row tiles through chain_body with their own LDS block (3 x 51 KB of the CU's
//     160 KB), then the 12 waves deal the plane's (head, query tile) attention tasks among themselves; a workgroup
//     barrier separates the phases (workgroup scope is enough: producer and consumer share the CU's L1);
//   * x, q, k, v^T and the attention tiles of a plane are written and re-read by the same CU, i.e. they live in its
//     XCD's L2 (0.7 MB per plane); only conv_enhanced / tokens6 come in and out6 goes out;
//   * row tiles are PER PLANE here (9 tiles for 280 tokens, the last one ragged) while the launches walk global row
//     tiles; the arithmetic per row is the same instruction sequence either way, so the two paths give identical bits
//     (tests/test_hip_parity.py::test_plane_resident_encoder_matches_launch_path).
// One plane per CU, so the path suits plane counts that fill the CUs evenly (B = 128 -> 256 planes on 256 CUs).
// MEASURED (tools/ab_encoder.py, profiles/r03_ab_encoder.json): even there it is 1.5 % slower than the launches
// (encoder 1496 vs 1474 us) -- 9 row tiles per plane instead of 8.75, and workgroup-wide barriers keep the three wave
// groups in lock-step, which costs what the 12 removed launch boundaries gave.  It is therefore selected only on
// request (aft_config.encoder_path = AFT_ENCODER_PLANE); AUTO runs the launches.
#include <algorithm>

#include "attn_device.h"
#include "chain_device.h"

namespace aft {

struct EncoderArgs {
    WeightsDev w;                          // torch vectors per layer, linear_1 / linear_2 / positions (3.4 KB of kernarg)
    const float *wpack;                     // fragment-packed GEMM weights [L][8 d^2]
    const float *conv_enhanced, *tokens6;   // [planes][S][T], [frames][tokens][6] or NULL
    float *x, *attn, *q, *k, *vt, *out6;
    int planes, tokens, tokpad, heads, layers;
    int S, T, p0, p1, emb_K, out6_features, out6_stride;
    float scale_log2e;
};

template <int D>
struct EncoderShape {
    static constexpr int GROUPS = 3;
    static constexpr int THREADS = GROUPS * ChainShape<D>::THREADS;
    static constexpr size_t GROUP_FLOATS = ChainShape<D>::LDS_BYTES / sizeof(float);
    static constexpr size_t LDS_BYTES = GROUPS * ChainShape<D>::LDS_BYTES;
};

// Register budget: the chain bodies need 152-155 VGPRs and the attention body 144 of the 168 a wave may have at three
// waves per SIMD.  Two things kept the four bodies inside one kernel at 166 VGPRs / 0 spills: the bodies launder the
// thread index they start from (LICM otherwise hoists a dozen lane-dependent offsets to the top of the kernel, where they
// are spilled at once and reloaded at every tile start -- a scratch reload drains vmcnt), and the
// phases read the kernel arguments through the kernarg-segment pointer instead of holding ~280 SGPRs of pointers live.
// (Non-inlined phase functions were tried: the calling convention's reserved and callee-saved registers pushed 55-75
// scratch accesses into each chain tile.)
// The phases read the kernel arguments where they are -- the kernarg segment, constant address space, scalar loads --
// through this pointer type (a reference to the by-value argument would make the compiler copy all 3.5 KB to scratch).
using EncoderArgsPtr = const __attribute__((address_space(4))) EncoderArgs *;

template <int D>
__device__ __forceinline__ ChainArgs plane_args(EncoderArgsPtr ep, int plane) {
    const auto &e = *ep;
    const int ntp = (e.tokens + 31) / 32;
    const size_t hoff = (size_t)plane * e.heads * e.tokpad * kHeadDim;
    ChainArgs a{};
    a.x = e.x + (size_t)plane * e.tokens * D;
    a.attn = e.attn + (size_t)plane * ntp * 32 * D;
    a.q = e.q + hoff; a.k = e.k + hoff; a.vt = e.vt + hoff;
    a.rows = e.tokens; a.tokens = e.tokens; a.tokpad = e.tokpad; a.heads = e.heads;
    return a;
}

// patch embedding + adapter features + linear_1 + positions + layer 0's in-projection
template <int D, int ACT>
__device__ __forceinline__ void phase_embed_qkv(EncoderArgsPtr ep, float *gs, int gtid, int plane, int g, int tile_end) {
    const auto &e = *ep;
    ChainArgs f = plane_args<D>(ep, plane);
    f.wqkv = e.wpack;
    f.bv = e.w.layers[0].in_proj_b + 2 * D;
    f.emb_conv = e.conv_enhanced + (size_t)plane * e.S * e.T;
    f.emb_tok6 = e.tokens6 ? e.tokens6 + (size_t)(plane >> 1) * e.tokens * 6 : nullptr;
    f.emb_w1 = e.w.lin1_w; f.emb_b1 = e.w.lin1_b; f.emb_pos = e.w.pos;
    f.emb_S = e.S; f.emb_T = e.T; f.emb_p0 = e.p0; f.emb_p1 = e.p1; f.emb_K = e.emb_K;
    chain_body<D, ACT, false, true>(f, gs, gtid, g, EncoderShape<D>::GROUPS, tile_end);
}

template <int D>
__device__ __forceinline__ void phase_attention(EncoderArgsPtr ep, int plane, int l, int wave) {
    const auto &e = *ep;
    const ChainArgs a = plane_args<D>(ep, plane);
    attn_body(a.q, a.k, a.vt, e.w.layers[l].in_proj_b, const_cast<float *>(a.attn), e.heads, e.tokens, e.tokpad, D, e.scale_log2e,
              wave, EncoderShape<D>::GROUPS * ChainShape<D>::WAVES, e.heads * (e.tokpad / kTile), nullptr);
}

template <int D, int ACT, bool LAST>
__device__ __forceinline__ void phase_chain(EncoderArgsPtr ep, float *gs, int gtid, int plane, int l, int g, int tile_end) {
    const auto &e = *ep;
    const size_t dd = (size_t)D * D;
    const auto &lw = e.w.layers[l];
    const float *wp = e.wpack + (size_t)l * 8 * dd;
    ChainArgs m = plane_args<D>(ep, plane);
    m.wo = wp + 3 * dd; m.bo = lw.out_proj_b;
    m.w1 = wp + 4 * dd; m.b1 = lw.lin1_b;
    m.w2 = wp + 6 * dd; m.b2 = lw.lin2_b;
    m.g1 = lw.norm1_w; m.be1 = lw.norm1_b; m.g2 = lw.norm2_w; m.be2 = lw.norm2_b;
    if constexpr (!LAST) {
        m.wqkv = wp + 8 * dd;
        m.bv = e.w.layers[l + 1].in_proj_b + 2 * D;
        chain_body<D, ACT, true, true>(m, gs, gtid, g, EncoderShape<D>::GROUPS, tile_end);
    } else {
        m.lin2_w = e.w.lin2_w; m.lin2_b = e.w.lin2_b;
        m.out6 = e.out6 + (size_t)plane * e.tokens * e.out6_stride;
        m.out6_features = e.out6_features; m.out6_stride = e.out6_stride;
        chain_body<D, ACT, true, false>(m, gs, gtid, g, EncoderShape<D>::GROUPS, tile_end);
    }
}

template <int D, int ACT>
__global__ __launch_bounds__(EncoderShape<D>::THREADS, EncoderShape<D>::THREADS / 256) void encoder_plane_kernel(const EncoderArgs e_by_value) {
    const EncoderArgsPtr e = (EncoderArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();   // = &e_by_value (explicit arguments start at offset 0)
    using S = ChainShape<D>;
    using E = EncoderShape<D>;
    constexpr int W = S::WAVES, NG = E::GROUPS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = wave / W;                                  // wave group = chain workgroup of the layer-by-layer path
    const int gtid = (int)threadIdx.x - g * S::THREADS;
    float *gs = smem + (size_t)g * E::GROUP_FLOATS;
    const int ntp = (e->tokens + 31) / 32;                   // row tiles per plane
    const int tile_end = g + (ntp + NG - 1) / NG * NG;       // every group walks the same number of tiles (barriers)
#pragma unroll 1
    for (int plane = blockIdx.x; plane < e->planes; plane += gridDim.x) {
        phase_embed_qkv<D, ACT>(e, gs, gtid, plane, g, tile_end);
#pragma unroll 1
        for (int l = 0; l < e->layers; ++l) {
            __syncthreads();    // q / k / v^T of every row tile of the plane are stored (vmcnt drained by the fence)
            phase_attention<D>(e, plane, l, wave);
            __syncthreads();    // all attention tiles of the plane are stored
            if (l + 1 < e->layers) phase_chain<D, ACT, false>(e, gs, gtid, plane, l, g, tile_end);
            else phase_chain<D, ACT, true>(e, gs, gtid, plane, l, g, tile_end);
        }
        __syncthreads();        // the next plane re-uses the LDS blocks
    }
}

bool encoder_plane_ok(const aft_config &c) {
    // 3 groups x 51 KB of LDS, 12 waves: the d = 128 / head dim 32 shape is the one instantiated (>= 32 tokens: a row tile inside one plane)
    return c.model_dim == 128 && c.num_head == 4 && c.num_layers <= kLayerWindow && (c.num_scs / c.patch_scs) * (c.num_symbols / c.patch_symbols) >= kTile;
}

template <int ACT>
static hipError_t launch_encoder_plane_t(const EncoderArgs &args, hipStream_t st) {
    constexpr int D = 128;
    using E = EncoderShape<D>;
    static PerDeviceOnce lds_attr;
    hipError_t ea = ensure_dynamic_lds(lds_attr, reinterpret_cast<const void *>(encoder_plane_kernel<D, ACT>), E::LDS_BYTES);
    if (ea != hipSuccess) return ea;
    const int blocks = std::min(args.planes, current_device_cus());
    hipLaunchKernelGGL((encoder_plane_kernel<D, ACT>), dim3(blocks), dim3(E::THREADS), E::LDS_BYTES, st, args);
    return hipGetLastError();
}

hipError_t launch_encoder_plane(const aft_config &c, const WeightsDev &w, const float *wpack, const float *conv_enhanced,
                                const float *tokens6, float *x, float *attn, float *q, float *k, float *vt, float *out6,
                                int planes, int tokens, int tokpad, hipStream_t st) {
    EncoderArgs a{};
    a.w = w;
    a.wpack = wpack;
    a.conv_enhanced = conv_enhanced; a.tokens6 = tokens6;
    a.x = x; a.attn = attn; a.q = q; a.k = k; a.vt = vt; a.out6 = out6;
    a.planes = planes; a.tokens = tokens; a.tokpad = tokpad; a.heads = c.num_head; a.layers = c.num_layers;
    a.S = c.num_scs; a.T = c.num_symbols; a.p0 = c.patch_scs; a.p1 = c.patch_symbols;
    a.emb_K = c.patch_scs * c.patch_symbols + (tokens6 ? 6 : 0);
    a.out6_features = c.patch_scs * c.patch_symbols;
    a.out6_stride = out6_stride(c);
    a.scale_log2e = 1.4426950408889634f / sqrtf((float)kHeadDim);
    return c.activation == AFT_ACT_GELU ? launch_encoder_plane_t<AFT_ACT_GELU>(a, st) : launch_encoder_plane_t<AFT_ACT_RELU>(a, st);
}

}  // namespace aft
