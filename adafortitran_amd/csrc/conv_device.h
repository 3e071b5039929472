// conv_device.h -- declarations shared by the two conv-stack kernels (k_conv.hip: banded, every grid, training;
// k_conv_stream.hip: the default 120 x 14 grid as a column-streaming pipeline).
#pragma once
#include <cstdint>

#include "aft_internal.h"

namespace aft {

struct ConvArgs {
    int mode;  // 0 = head (pilots -> conv_enhanced), 1 = tail (x, conv_enhanced -> complex out), 2 = plain (training)
    int S, T, SP, band_rows, nbands, ntiles, nseg, arena, extra;   // SP = LDS row-vector length (>= band_rows + 8)
    int stream_ok;   // head: pf is a multiple of 8 and the streaming scratch fits the conv1 / conv3 planes
    int xoff;   // inference: the seam exchange rows [(seam, side)][8 channels][SP] start here (floats; 0 = the dead input plane), -1 = overlapping sweeps
    // head
    const float *pilots, *up_w, *up_b;
    int pf;
    // tail
    const float *x, *lin2_w, *lin2_b, *resid;
    const float *lin2_out;   // tail: linear_2 already applied by the last chain launch, [rows][lin2_stride]; NULL = apply it here
    int lin2_stride;
    int d, tokens, p0, p1;
    const float *cw[4], *cb[4];
    float *out_plane;    // head / plain: [planes][S][T]
    float *out_complex;  // tail: [B][S][T][2]
    // training variant only (mode 2 = plain plane in, plain plane out):
    const float *in_plane;   // [planes][S][T]
    float *save[3];          // outputs of conv1 / conv2 / conv3 after their activation, [planes][C][T][S] (C = 8, 32, 8)
    const float *mask[3];    // backward: activation of stage k = acc where mask[k] > 0 else 0 (instead of bias + ReLU)
    unsigned long long *stamps;   // diagnostic build only (AFT_DIAG_STAMPS): per-workgroup s_memtime at the phase boundaries
    int plane0;                   // conv_stream16_kernel: first plane of this launch (a forward may launch its planes in two parts)
    const float *wfrag;           // conv_stream16_kernel: this stack's conv2 / conv3 weights as 16x16x4 operand fragments (conv_frag16_entry)
    int ranges;                   // conv_stream_kernel<., true> (training): column ranges a plane is split into (1 | 2 | 4; 0 = 1)
};

// ---- conv2 / conv3 weights of one ConvEnhancer as v_mfma_f32_16x16x4_f32 A-operand fragments, packed by the forward's prologue launch
// (k_misc.hip) and read by conv_stream16_kernel's matrix waves with 22 lane-linear 16-byte loads: [22 quads][64 lanes][4] floats.
// Entry e = 4 quad + j of lane l (li = l % 16: the operand's row, gk = l / 16: its k index):
//   0 .. 35   conv2, e = 18 mt + ks2: output channel 16 mt + li, k-step ks2 = 2 (kx*3 + ky) + ci half, input channel 4 (ci half) + gk
//   36 .. 75  conv3, e - 36 = 8 rt + ks: product row 16 rt + li = accumulator register v = li % 4 of tile rt, lane group li / 4 = output
//             channel low bits (co = 4 (co half) + li / 4).  Tiles 0, 1, 2 = kx 0, 1, 2 with v = 2 ky + (co half), ky = 0, 1; tile 3 =
//             [idle, idle, (kx 0, ky 2, co half 0 / 1)]; tile 4 = [(kx 1, ky 2, co half 0 / 1), (kx 2, ky 2, co half 0 / 1)]
//             (conv3_row16).  Input channel 16 (ks / 4) + 4 gk + ks % 4.
//   76, 77    conv3.bias[4 (co half) + gk]        78, 79  unused
//   80 .. 87  conv2.bias[16 mt + 4 gk + v], e - 80 = 4 mt + v
// Behind the 22 x 64 quads: the helper waves' tables, [channel half h][80]: conv1 as v_pk_fma operands -- pair k (channels 4h + 2k,
// 4h + 2k + 1), tap k9: floats 2 (9 k + k9) + {0, 1}; their biases at 36 + 2k + {0, 1} -- and conv4 at 40 ..: input-channel pair cp
// (channels 4h + 2cp, + 1), tap k9: 40 + 2 (9 cp + k9) + {0, 1}; conv4's bias at 76 (conv_helper_entry).  Lanes of one half read the
// same addresses: 20 broadcast 16-byte loads per helper wave instead of 77 dword loads with two addresses each.
constexpr int kFragQuads = 22, kHelperFloats = 2 * 80, kFragFloats = kFragQuads * 64 * 4 + kHelperFloats;
__device__ __forceinline__ float conv_helper_entry(const float *__restrict__ w1, const float *__restrict__ b1, const float *__restrict__ w4,
                                                   const float *__restrict__ b4, int h, int i) {
    if (i < 36) { const int pr = i >> 1, k = pr / 9, k9 = pr - 9 * k; return w1[(4 * h + 2 * k + (i & 1)) * 9 + k9]; }
    if (i < 40) return b1[4 * h + (i - 36)];
    if (i < 76) { const int pr = (i - 40) >> 1, cp = pr / 9, k9 = pr - 9 * cp; return w4[(4 * h + 2 * cp + (i & 1)) * 9 + k9]; }
    return i == 76 ? b4[0] : 0.f;
}
__device__ __forceinline__ float conv_frag16_entry(const float *__restrict__ w2, const float *__restrict__ b2, const float *__restrict__ w3,
                                                   const float *__restrict__ b3, int e, int lane) {
    const int li = lane & 15, gk = lane >> 4;
    if (e < 36) {
        const int mt = e / 18, ks2 = e - 18 * mt, tap = ks2 >> 1, cih = ks2 & 1, kx = tap / 3, ky = tap - 3 * kx;
        return w2[(16 * mt + li) * 72 + (4 * cih + gk) * 9 + ky * 3 + kx];
    }
    if (e < 76) {
        const int i = e - 36, rt = i >> 3, ks = i & 7, v = li & 3, cohi = v & 1;
        int kx, ky;
        if (rt < 3) { kx = rt; ky = v >> 1; }
        else if (rt == 3) { if (v < 2) return 0.f; kx = 0; ky = 2; }
        else { kx = 1 + (v >> 1); ky = 2; }
        return w3[(4 * cohi + (li >> 2)) * 288 + (16 * (ks >> 2) + 4 * gk + (ks & 3)) * 9 + ky * 3 + kx];
    }
    if (e < 78) return b3[4 * (e - 76) + gk];
    if (e < 80) return 0.f;
    return b2[16 * ((e - 80) >> 2) + 4 * gk + ((e - 80) & 3)];
}

#ifndef AFT_CONV_PIPE
#define AFT_CONV_PIPE 1        // A/B knob: conv3(t) and conv2(t+1) as two interleaved MFMA chains
#endif
constexpr int kConvThreads = 512;
constexpr int kConvWaves = kConvThreads / 64;
constexpr int kTileRows = 30;   // valid conv3 rows per 32-lane tile
constexpr int kW3Off = 72 * 33, kWStage = kW3Off + 96 * 33;   // LDS staging of the conv2 / conv3 weights (floats)

// 32-bit-offset buffer accesses for the training variant's saved tensors (each < 2 GB)
using ConvSrd = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ ConvSrd conv_srd(const float *p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ float conv_ld(ConvSrd r, unsigned idx) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, idx * 4u, 0, 0));
}
__device__ __forceinline__ void conv_st(ConvSrd r, unsigned idx, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, idx * 4u, 0, 0);
}
__device__ __forceinline__ f32x2 conv_ld2(ConvSrd r, unsigned idx) {
    return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, idx * 4u, 0, 0));
}
// two adjacent elements (idx, idx + 1) as one 8-byte store; gfx950 only needs dword alignment of the address
__device__ __forceinline__ void conv_st2(ConvSrd r, unsigned idx, float v0, float v1) {
    using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
    __builtin_amdgcn_raw_buffer_store_b64(u32x2{__builtin_bit_cast(unsigned, v0), __builtin_bit_cast(unsigned, v1)}, r, idx * 4u, 0, 0);
}

__device__ __forceinline__ float lane_from_below(float v) {   // lane i <- lane i-1 (DPP wave_shr:1)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_from_above(float v) {   // lane i <- lane i+1 (DPP wave_shl:1)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}


// A block = 64 pixels x kUpPlanes planes.  16 planes (late round 5; 64 before): four times the blocks, a quarter of the serial work of
// each -- at config 5 (6 720 pixels x 96 pilots) the product's 210 blocks of 1 536 fused multiply-adds per thread were the prologue
// launch's critical path (13 of its 37 us); per element the sum runs in the same order whatever the block shape (same bits).
constexpr int kUpPix = 64, kUpPlanes = 16, kUpPerThread = kUpPlanes / 4;
__device__ __forceinline__ void upsample_planes_body(float *ups, const float *__restrict__ up_w, const float *__restrict__ up_b,
                                                     const float *__restrict__ pilots, float *__restrict__ planes_out, int npix,
                                                     int pf, int nplanes, int bx, int by) {
    const int wld = pf + 4;                       // row stride of the weight tile: 16-byte rows, conflict-free float4 reads
    float *Ws = ups, *Ps = ups + kUpPix * wld;    // [64][pf + 4] | [kUpPlanes][pf]
    const int tid = threadIdx.x, pix0 = bx * kUpPix, plane0 = by * kUpPlanes;
    const int nq = pf >> 2;
    for (int i = tid; i < kUpPix * nq; i += 256) {
        const int r = i / nq, q = i - r * nq;
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (pix0 + r < npix) v = *reinterpret_cast<const f32x4 *>(up_w + (size_t)(pix0 + r) * pf + 4 * q);
        *reinterpret_cast<f32x4 *>(Ws + r * wld + 4 * q) = v;
    }
    for (int i = tid; i < kUpPlanes * pf; i += 256) {
        const int pl = i / pf, k = i - pl * pf, n = plane0 + pl;
        Ps[i] = n < nplanes ? pilots[((size_t)(n >> 1) * pf + k) * 2 + (n & 1)] : 0.f;
    }
    __syncthreads();
    const int px = tid & 63, g = tid >> 6;
    const float b = pix0 + px < npix ? up_b[pix0 + px] : 0.f;
    float acc[kUpPerThread];
#pragma unroll
    for (int i = 0; i < kUpPerThread; ++i) acc[i] = b;
    for (int q = 0; q < nq; ++q) {
        const f32x4 wv = *reinterpret_cast<const f32x4 *>(Ws + px * wld + 4 * q);
#pragma unroll
        for (int i = 0; i < kUpPerThread; ++i) {
            const f32x4 pv = *reinterpret_cast<const f32x4 *>(Ps + (g * kUpPerThread + i) * pf + 4 * q);   // wave-uniform address: broadcast
            float v = acc[i];
            v = fmaf(wv[0], pv[0], v);
            v = fmaf(wv[1], pv[1], v);
            v = fmaf(wv[2], pv[2], v);
            v = fmaf(wv[3], pv[3], v);
            acc[i] = v;
        }
    }
    if (pix0 + px < npix) {
#pragma unroll
        for (int i = 0; i < kUpPerThread; ++i) {
            const int n = plane0 + g * kUpPerThread + i;
            if (n < nplanes) planes_out[(size_t)n * npix + pix0 + px] = acc[i];
        }
    }
}
inline size_t upsample_planes_lds(int pf) { return sizeof(float) * ((size_t)kUpPix * (pf + 4) + (size_t)kUpPlanes * pf); }
inline bool upsample_planes_ok(const float *up_w, int pf) {
    return pf % 4 == 0 && upsample_planes_lds(pf) <= 64 * 1024 && (reinterpret_cast<uintptr_t>(up_w) & 15) == 0;   // (config 5: 96 pilots = 50 176 B)
}

// k_conv_stream.hip: default grid, inference, head with pre-computed upsampled planes (a.in_plane) or tail on linear_2's output
// (a.lin2_out); returns hipErrorNotSupported when the arguments need the banded kernel
// the fragment image (kFragFloats floats) of one ConvEnhancer from its four conv weights / biases (cb NULL: no biases), k_conv_stream.hip
hipError_t launch_conv_frag_pack(const float *const cw[4], const float *const cb[4], float *dst, hipStream_t st);
bool conv_stream_ok(const ConvArgs &a);
hipError_t launch_conv_stream(ConvArgs &a, int planes, hipStream_t st);
// k_conv_rows.hip: inference on grids whose planes need more than one band in k_conv.hip (config 5): whole-height workgroups that
// stream over the columns through ring buffers; returns hipErrorNotSupported when the arguments need the banded kernel
bool conv_rows_ok(const ConvArgs &a, int planes);
hipError_t launch_conv_rows(ConvArgs &a, int planes, hipStream_t st);

}  // namespace aft
