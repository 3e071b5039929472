// pack_device.h -- one float4 of the fragment-packed encoder weight image (shared by pack_weights_kernel, k_chain.hip, and the
// forward's prologue launch, k_misc.hip).  Layout: chain_device.h / DESIGN.md section 3.
#pragma once

#include "aft_internal.h"

namespace aft {

using bf16x8_pack = __attribute__((ext_vector_type(8))) __bf16;

// v = index of the float4 inside the image [layer][in_proj 3dd | out_proj dd | lin1 2dd | lin2 2dd]; layers are taken from
// w.layers[first_layer + i], the image starts at layer first_layer's block
__device__ __forceinline__ void pack_weights_vec(const WeightsDev &w, float *__restrict__ packed, int d, int first_layer,
                                                 int num_layers, int split, size_t v) {
    const size_t per_layer = (size_t)8 * d * d;
    if (v * 4 >= per_layer * num_layers) return;
    const int layer = (int)(v * 4 / per_layer);
    size_t off = v * 4 - (size_t)layer * per_layer;
    const aft_layer_weights &lw = w.layers[first_layer + layer];
    const float *src;
    int K;
    if (off < (size_t)3 * d * d) { src = lw.in_proj_w; K = d; }
    else if ((off -= (size_t)3 * d * d) < (size_t)d * d) { src = lw.out_proj_w; K = d; }
    else if ((off -= (size_t)d * d) < (size_t)2 * d * d) { src = lw.lin1_w; K = d; }
    else { off -= (size_t)2 * d * d; src = lw.lin2_w; K = 2 * d; }
    // off = (((tile*NKB + kb)*4 + s)*64 + lane)*4 ; element (row = 32 tile + lane%32, k = 32kb + 8s + 4(lane/32) + j)
    const int lane = (int)(off / 4) % 64, s = (int)(off / 256) % 4;
    const int blk = (int)(off / 1024), nkb = K / 32, kb = blk % nkb, ct = blk / nkb;
    const int col = ct * 32 + (lane & 31), k = kb * 32 + s * 8 + (lane >> 5) * 4;
    if (!split) {
        *reinterpret_cast<f32x4 *>(packed + v * 4) = *reinterpret_cast<const f32x4 *>(src + (size_t)col * K + k);
        return;
    }
    // split-precision image: slot s of a block = MFMA m = s >> 1, term hi (s even) / lo (s odd); the lane's 8 bf16 values
    // are k = 32 kb + 16 m + 8 (j >> 2) + 4 h + (j & 3), the k order of an accumulator used as operand (chain_device.h)
    const int m = s >> 1, hh = lane >> 5;
    bf16x8_pack o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float x = src[(size_t)col * K + kb * 32 + 16 * m + 8 * (j >> 2) + 4 * hh + (j & 3)];
        const __bf16 hi = (__bf16)x;
        o[j] = (s & 1) ? (__bf16)(x - (float)hi) : hi;
    }
    *reinterpret_cast<f32x4 *>(packed + v * 4) = __builtin_bit_cast(f32x4, o);
}

}  // namespace aft
