// srd.h -- buffer-resource (SRD) loads / stores shared by the gfx950 kernels (device code only).
#pragma once
#include "aft_internal.h"

namespace aft {

// All hot loads go through a buffer resource (SRD + 32-bit byte offset), not global_load with 64-bit
// per-lane addresses: micro-benchmarked on MI355X (tools/micro/mfma_feed2.hip, NT=3 feed loop) the
// global_load form drops from 136 to 104 TFLOP/s as 1 -> 3 workgroups per CU stream weights, the
// buffer_load form goes 137 -> 149 TFLOP/s.
using Srd = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ Srd make_srd(const float *p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ f32x4 srd_load(Srd r, unsigned byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
}
// byte_off + CONST with a compile-time CONST: the part above the 12-bit instruction offset rides in the scalar
// offset operand (a literal), so no v_add_u32 per load is needed to form the address (128 weight loads per tile)
__device__ __forceinline__ f32x4 srd_load_c(Srd r, unsigned byte_off, unsigned const_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off + (const_off & 0xfffu), const_off & ~0xfffu, 0));
}
__device__ __forceinline__ void srd_store(Srd r, unsigned byte_off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(decltype(__builtin_amdgcn_raw_buffer_load_b128(r, 0, 0, 0)), v), r, byte_off, 0, 0);
}

// split-precision tier: bf16 hi / lo terms of fp32 values (chain_device.h explains the scheme)
using bf16x8 = __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16;
struct BsFrag {     // one MFMA's worth of an activation operand: 8 k-values as bf16 hi / lo (16 bytes each)
    f32x4 hi, lo;
};
__device__ __forceinline__ BsFrag bs_split(f32x4 x0, f32x4 x1) {
    bf16x8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float x = j < 4 ? x0[j] : x1[j - 4];
        hi[j] = (__bf16)x;
        lo[j] = (__bf16)(x - (float)hi[j]);
    }
    return BsFrag{__builtin_bit_cast(f32x4, hi), __builtin_bit_cast(f32x4, lo)};
}

}  // namespace aft
