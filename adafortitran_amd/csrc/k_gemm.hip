// k_gemm.hip -- row-major fp32 GEMMs on v_mfma_f32_32x32x2_f32 for the training path.
//
// The inference path keeps activations in MFMA-fragment order and never needs a general GEMM.  The
// backward pass does: every nn.Linear of the encoder (reference src/models/blocks/encoders.py:44-55,
// torch.nn.TransformerEncoderLayer) needs, besides y = x W^T + b,
//     dgrad  dx = dy W          (NN)
//     wgrad  dW = dy^T x        (TN, reduction over all token rows)
// with every operand in PyTorch's row-major layout.  One kernel, three operand modes:
//     NT  C[M][N] = A[M][K] . B[N][K]^T (+ bias[N])
//     NN  C[M][N] = A[M][K] . B[K][N]
//     TN  C[M][N] = A[K][M]^T . B[K][N]      K = token rows, split over blockIdx.z into partial
//                                            slices that reduce_slices_kernel sums (deterministic)
// Two kernels share the tiling below: gemm_kernel takes any shape (guarded, optionally element-wise loads);
// gemm_fast_body takes the aligned shapes -- every GEMM of the encoder's training step -- with buffer-resource
// loads, no address arithmetic in the k loop and a transposed accumulator whose epilogue is 16-byte row stores.
// On that accumulator (a lane owns whole row pieces) ride the row-wise operations of the layer as epilogues:
// residual + dropout + LayerNorm (gemm_add_ln_kernel), activation + dropout (gemm_act_kernel), activation backward
// (gemm_actbwd_kernel); the four weight gradients of a layer run as one batched launch (gemm_tn_batch_kernel) that
// also produces the bias gradients db = A^T 1 from the A fragments it loads.
// Tiling: 128x128 (TN) or 64x128 (NT / NN) output tile per workgroup (4 waves, each 2x2 or 1x2 MFMA tiles), K step 16.  Both
// operands are staged in LDS k-major ([k][m], row stride 132 floats) so that a fragment read is
// 32 consecutive floats per half-wave -- conflict-free for A and B alike -- and the global->LDS
// transposition of k-contiguous sources lands on 64 distinct banks.  Global loads of step i+1 are in
// flight during the 32 MFMAs per wave of step i (register staging, two LDS buffers, one barrier).
#include "aft_internal.h"

namespace aft {

constexpr int GBM = 128, GBN = 128, GBK = 16, GLD = 132;

// tile element (x, k), source contiguous in k:  src[(x0 + x) * ld + k0 + k]
template <int NU, bool VEC>
__device__ __forceinline__ void load_kc(const float *__restrict__ src, int ld, int x0, int xlim, int k0, int klim,
                                        int tid, f32x4 (&v)[NU]) {
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int e = tid + 256 * u, x = e >> 2, kq = e & 3;
        const bool ok = x0 + x < xlim && k0 + 4 * kq < klim;
        const float *p = src + (size_t)(x0 + x) * ld + k0 + 4 * kq;
        if constexpr (VEC) {
            v[u] = ok ? *reinterpret_cast<const f32x4 *>(p) : f32x4{0.f, 0.f, 0.f, 0.f};
        } else {   // row length or leading dimension not a multiple of 4: element-wise, guarded
#pragma unroll
            for (int c = 0; c < 4; ++c) v[u][c] = ok && k0 + 4 * kq + c < klim ? p[c] : 0.f;
        }
    }
}
template <int NU>
__device__ __forceinline__ void store_kc(float *__restrict__ lds, int tid, const f32x4 (&v)[NU]) {
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int e = tid + 256 * u, x = e >> 2, kq = e & 3;
#pragma unroll
        for (int c = 0; c < 4; ++c) lds[(4 * kq + c) * GLD + x] = v[u][c];
    }
}
// tile element (x, k), source contiguous in x:  src[(k0 + k) * ld + x0 + x]
template <int NU, bool VEC>   // NU = 2: 128-wide tile (32 vec4 per k row); NU = 1: 64-wide tile (16 vec4 per k row)
__device__ __forceinline__ void load_xc(const float *__restrict__ src, int ld, int x0, int xlim, int k0, int klim,
                                        int tid, f32x4 (&v)[NU]) {
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int e = tid + 256 * u, k = e >> (3 + NU), xq = e & (16 * NU - 1);
        const bool ok = k0 + k < klim && x0 + 4 * xq < xlim;
        const float *p = src + (size_t)(k0 + k) * ld + x0 + 4 * xq;
        if constexpr (VEC) {
            v[u] = ok ? *reinterpret_cast<const f32x4 *>(p) : f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) v[u][c] = ok && x0 + 4 * xq + c < xlim ? p[c] : 0.f;
        }
    }
}
template <int NU>
__device__ __forceinline__ void store_xc(float *__restrict__ lds, int tid, const f32x4 (&v)[NU]) {
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int e = tid + 256 * u, k = e >> (3 + NU), xq = e & (16 * NU - 1);
        *reinterpret_cast<f32x4 *>(lds + k * GLD + 4 * xq) = v[u];
    }
}

// OP: 0 = NT, 1 = NN, 2 = TN (see header).  Any M, N, K (16-byte loads where the operand allows them).
// BM = rows of C per workgroup: 128 (each wave 64x64; the split-K TN product) or 64 (each wave 32x64; NT / NN).  VEC = both operands allow
// 16-byte loads (4-aligned leading dimensions, extents and base pointers); the element-wise variant is a
// separate instantiation so that the fast path keeps its loads branch-free and back to back.  (16-byte
// loads need 4-element-aligned leading dimensions and extents; the base address only dword alignment.)
template <int OP, int BM, bool VEC>
__global__ __launch_bounds__(256) void gemm_kernel(const float *__restrict__ A, const float *__restrict__ B,
                                                   float *__restrict__ C, const float *__restrict__ bias, int M, int N,
                                                   int K, int lda, int ldb, int ldc, int k_chunk, size_t c_slice,
                                                   int accumulate) {
    __shared__ __attribute__((aligned(16))) float As[2][GBK * GLD];
    __shared__ __attribute__((aligned(16))) float Bs[2][GBK * GLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    constexpr int TI = BM / 64, NUA = BM / 64;   // MFMA row tiles per wave; A-tile vec4 per thread
    // Work items.  TN: one (tile, row slice) per workgroup (grid = n tiles x m tiles x slices).  NT / NN: the
    // grid is a few workgroups per CU and each WALKS tiles (n tile fastest: neighbours share their A rows in
    // L2), fetching the first k-step of its next tile during the last step of the current one.  Without
    // that, all workgroups of a launch run their (8-step) K loops in lock-step rounds and the first fetch and
    // the epilogue of every round are exposed chip-wide.
    const int ntn = (N + GBN - 1) / GBN, ntm = (M + BM - 1) / BM;
    const int ntiles = OP == 2 ? 1 : ntn * ntm, tstep = OP == 2 ? 1 : (int)gridDim.x;
    const int kbeg = OP == 2 ? blockIdx.z * k_chunk : 0, kend = OP == 2 ? min(K, kbeg + k_chunk) : K;
    if constexpr (OP == 2) C += (size_t)blockIdx.z * c_slice;
    const int nsteps = (kend - kbeg + GBK - 1) / GBK;

    f32x4 ra[NUA], rb[2];
    auto fetch = [&](int m0, int n0, int k0) {
        if constexpr (OP == 2) load_xc<NUA, VEC>(A, lda, m0, M, k0, kend, tid, ra); else load_kc<NUA, VEC>(A, lda, m0, M, k0, kend, tid, ra);
        if constexpr (OP == 0) load_kc<2, VEC>(B, ldb, n0, N, k0, kend, tid, rb); else load_xc<2, VEC>(B, ldb, n0, N, k0, kend, tid, rb);
    };
    auto stage = [&](int buf) {
        if constexpr (OP == 2) store_xc(As[buf], tid, ra); else store_kc(As[buf], tid, ra);
        if constexpr (OP == 0) store_kc(Bs[buf], tid, rb); else store_xc(Bs[buf], tid, rb);
    };
    auto origin = [&](int tile, int &m0, int &n0) {
        if constexpr (OP == 2) {
            m0 = blockIdx.y * BM;
            n0 = blockIdx.x * GBN;
        } else {
            m0 = (tile / ntn) * BM;
            n0 = (tile % ntn) * GBN;
        }
    };

    int tile = OP == 2 ? 0 : (int)blockIdx.x, m0 = 0, n0 = 0;
    if (tile < ntiles) {
        origin(tile, m0, n0);
        if (nsteps > 0) fetch(m0, n0, kbeg);
    }
    while (tile < ntiles) {
        f32x16 acc[TI][2];
#pragma unroll
        for (int a = 0; a < TI; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
        const int next = tile + tstep;
        int m1 = 0, n1 = 0;
        if (next < ntiles) origin(next, m1, n1);
        if (nsteps > 0) stage(0);
        __syncthreads();
        for (int it = 0; it < nsteps; ++it) {
            const int buf = it & 1;
            if (it + 1 < nsteps) fetch(m0, n0, kbeg + (it + 1) * GBK);
            else if (next < ntiles) fetch(m1, n1, kbeg);   // the next tile's first k-step rides under this one's last
            const float *as = As[buf] + h * GLD + wm * (BM / 2) + j;
            const float *bs = Bs[buf] + h * GLD + wn * 64 + j;
#pragma unroll
            for (int kb = 0; kb < GBK / 2; ++kb) {
                const float b0 = bs[2 * kb * GLD], b1 = bs[2 * kb * GLD + 32];
#pragma unroll
                for (int ti = 0; ti < TI; ++ti) {
                    const float av = as[2 * kb * GLD + 32 * ti];
                    acc[ti][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, acc[ti][0], 0, 0, 0);
                    acc[ti][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1, acc[ti][1], 0, 0, 0);
                }
            }
            if (it + 1 < nsteps) {
                stage(buf ^ 1);
                __syncthreads();
            }
        }
#pragma unroll
        for (int ti = 0; ti < TI; ++ti)
#pragma unroll
            for (int tj = 0; tj < 2; ++tj) {
                const int col = n0 + wn * 64 + tj * 32 + j;
                if (col >= N) continue;
                const float bv = bias ? bias[col] : 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = m0 + wm * (BM / 2) + ti * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (row >= M) continue;
                    float *p = C + (size_t)row * ldc + col;
                    float v = acc[ti][tj][e] + bv;
                    if (accumulate) v += *p;
                    *p = v;
                }
            }
        __syncthreads();   // every wave is done with the LDS buffers before the next tile restages buffer 0
        tile = next;
        m0 = m1;
        n0 = n1;
    }
}

// ---- the aligned fast path -------------------------------------------------------------------------------------
// Same tiling, LDS layout and tile walk as gemm_kernel, for shapes whose leading dimensions, N and k extent allow
// 16-byte accesses everywhere (every GEMM of the encoder's training step).  What differs is everything beside the
// MFMAs -- a PMC pass over the training step showed 4.8-5.1 VALU instructions per MFMA in gemm_kernel, and VALU
// issued beside fp32 MFMAs costs matrix time:
//   * operand loads go through buffer resources: the per-lane byte offset is computed ONCE per kernel, the tile
//     origin and the k position ride in the scalar offset, rows outside the matrix get an out-of-range offset
//     (the hardware returns zeros), so the k loop carries no address arithmetic and no guards;
//   * the k loop is unrolled over the two LDS buffers, the buffer choice folds into the LDS immediates;
//   * the product is accumulated TRANSPOSED (the MFMA's A operand is the B-matrix fragment): a lane then owns one
//     row of C and its registers run along the columns, so the epilogue is eight 16-byte stores per 32x64 wave tile
//     straight from the accumulators, and the bias is the accumulators' initial value (16-byte loads).
using GemmSrd = __amdgpu_buffer_rsrc_t;
using u32x4g = __attribute__((ext_vector_type(4))) uint32_t;
constexpr unsigned kGemmOutOfRange = 0x7ffffff0u;   // >= any num_records used here: the load returns zeros
__device__ __forceinline__ GemmSrd gemm_srd(const float *p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 gemm_ld(GemmSrd r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

// Fused epilogue of the two projections that feed a LayerNorm (out-proj -> norm1, linear2 -> norm2; reference
// nn.TransformerEncoderLayer, norm_first = False):  s = res + drop(A W^T + b);  C = LN(s) * gamma + beta;  s and the row
// statistics (mean, rstd) are kept for the backward pass.  Needs N = 128 (one column tile = whole rows in a workgroup);
// replaces add_ln_fwd_kernel and its 147 MB of traffic per call.
struct LnArgs {
    const float *res, *gamma, *beta;
    float *s_out, *stats;
    float eps, keep_scale;
    uint32_t seed, threshold;
};

// (bx, by, bz, gx) stand for the block indices and the grid width: the batched weight-gradient launch maps its flat
// block index onto several products
// Fused epilogue of linear1 (reference nn.TransformerEncoderLayer: linear2(dropout(activation(linear1(x))))):
// C = A W^T + b is stored as the pre-activation the backward needs, and hd = drop(act(C)) goes out beside it --
// replaces act_fwd_kernel and its re-read of C.  GELU is the exact-erf polynomial of the inference chain kernel.
struct ActArgs {
    float *hd;            // [M][N], leading dimension ldc
    float keep_scale;
    uint32_t seed, threshold;
    int act;              // AFT_ACT_RELU / AFT_ACT_GELU
};
constexpr int kActMaxN = 512;   // columns whose dropout words fit the epilogue's LDS table

// ... and of linear2's data gradient: da = (dy W2) o mask/(1-p) o act'(a), the backward of that same site
struct ActBwdArgs {
    const float *a;       // pre-activations [M][N], leading dimension ldc
    float keep_scale;
    uint32_t seed, threshold;
    int act;
};
constexpr int EPI_NONE = 0, EPI_LN = 1, EPI_ACT = 2, EPI_ACTBWD = 3;
// CSUM (TN only): also the column sums of A over this block's rows (db = A^T 1, the bias gradient that goes with
// dW = A^T B): each lane adds up the A fragments it feeds to the matrix cores anyway, one VALU add per two MFMAs.
// WN = waves along the columns of the 128-column tile: 2 (each wave 32 TI x 64, the 2 x 2 wave grid of the 64- and 128-row tiles) or 4
// (each wave BM x 32: the 96-ROW tile of round 6 -- three row tiles per wave, three workgroups per CU.  71 680 token rows are 1 120
// tiles of 64 rows on 1 024 resident workgroups -- a second round of 96 lone tiles -- but 747 tiles of 96 rows on 768: one round, 0.97
// of a plane per CU.  Same k order per output element: the same bits whichever tile height runs.)
template <int OP, int BM, int EPI = EPI_NONE, bool CSUM = false, int WN = 2>
__device__ __forceinline__ void gemm_fast_body(float (*As)[GBK * GLD], float (*Bs)[GBK * GLD], const float *__restrict__ A,
                                               const float *__restrict__ B, float *__restrict__ C, const float *__restrict__ bias,
                                               int M, int N, int K, int lda, int ldb, int ldc, int k_chunk, size_t c_slice,
                                               int accumulate, int bx, int by, int bz, int gx, const LnArgs *ln = nullptr,
                                               const ActArgs *act = nullptr, float *csum_out = nullptr,
                                               const ActBwdArgs *actb = nullptr) {
    static_assert(!CSUM || OP == 2, "column sums ride on the TN product");
    constexpr bool LN = EPI == EPI_LN, ACT = EPI == EPI_ACT, ACTB = EPI == EPI_ACTBWD;
    __shared__ float ln_red[LN ? 2 * 64 * 2 : 1];
    __shared__ __attribute__((aligned(16))) uint32_t ln_colw[LN ? GBN : (ACT || ACTB) ? kActMaxN : 4];
    if constexpr (LN) {   // column words of the dropout mask, once per workgroup (published by the first tile's barrier)
        if (threadIdx.x < GBN) ln_colw[threadIdx.x] = dropmask_col_word(ln->seed, threadIdx.x);
    }
    if constexpr (ACT) {
        for (int i = threadIdx.x; i < N; i += 256) ln_colw[i] = dropmask_col_word(act->seed, (uint32_t)i);
    }
    if constexpr (ACTB) {
        for (int i = threadIdx.x; i < N; i += 256) ln_colw[i] = dropmask_col_word(actb->seed, (uint32_t)i);
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    static_assert(WN == 2 || (WN == 4 && OP != 2 && EPI == EPI_NONE && !CSUM), "the 1 x 4 wave grid serves the plain NT / NN products");
    const int wm = WN == 2 ? wave >> 1 : 0, wn = WN == 2 ? wave & 1 : wave;
    constexpr int WM = 4 / WN, TI = BM / (32 * WM), TJ = GBN / (32 * WN);   // MFMA tiles per wave: TI along m, TJ along n
    constexpr int NUA = (BM + 63) / 64;                                      // A-tile vec4 per thread (a 96-row tile: the upper quarter idle)
    constexpr int WROWS = 32 * TI, WCOLS = 32 * TJ;                          // a wave's piece of the output tile
    constexpr bool A_KC = OP != 2, B_KC = OP == 0;   // operand rows contiguous in k (else contiguous in m / n)
    const int ntn = (N + GBN - 1) / GBN, ntm = (M + BM - 1) / BM;
    const int ntiles = OP == 2 ? 1 : ntn * ntm, tstep = OP == 2 ? 1 : gx;
    const int kbeg = OP == 2 ? bz * k_chunk : 0, kend = OP == 2 ? min(K, kbeg + k_chunk) : K;
    if constexpr (OP == 2) C += (size_t)bz * c_slice;
    const int nsteps = (kend - kbeg) / GBK;   // the k extent is a multiple of GBK (launch-side condition)

    const GemmSrd sa = gemm_srd(A, (unsigned)((OP == 2 ? K : M) * lda) * 4u);
    const GemmSrd sb = gemm_srd(B, (unsigned)((OP == 0 ? N : K) * ldb) * 4u);
    const GemmSrd sbias = gemm_srd(bias, bias ? (unsigned)N * 4u : 0u);
    // tile-invariant lane offsets (bytes) and the lane's extent coordinate for the per-tile range check
    unsigned va[NUA], vb[2];
    int xa[NUA], xb[2];
#pragma unroll
    for (int u = 0; u < NUA; ++u) {
        const int e = tid + 256 * u;
        if constexpr (A_KC) { xa[u] = e >> 2; va[u] = (unsigned)((e >> 2) * lda + 4 * (e & 3)) * 4u; }
        else { xa[u] = 4 * (e & (16 * NUA - 1)); va[u] = (unsigned)((e >> (3 + NUA)) * lda + xa[u]) * 4u; }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int e = tid + 256 * u;
        if constexpr (B_KC) { xb[u] = e >> 2; vb[u] = (unsigned)((e >> 2) * ldb + 4 * (e & 3)) * 4u; }
        else { xb[u] = 4 * (e & 31); vb[u] = (unsigned)((e >> 5) * ldb + xb[u]) * 4u; }
    }
    const unsigned ka = (A_KC ? GBK : GBK * lda) * 4u, kb_ = (B_KC ? GBK : GBK * ldb) * 4u;   // scalar offset per k-step

    f32x4 ra[NUA], rb[2];
    unsigned oa[NUA], ob[2], sa0 = 0, sb0 = 0;   // this tile's lane offsets (range-checked) and scalar origins
    auto locate = [&](int m0, int n0) {
#pragma unroll
        for (int u = 0; u < NUA; ++u) oa[u] = (BM % 64 == 0 || xa[u] < BM) && m0 + xa[u] < M ? va[u] : kGemmOutOfRange;
#pragma unroll
        for (int u = 0; u < 2; ++u) ob[u] = n0 + xb[u] < N ? vb[u] : kGemmOutOfRange;
        sa0 = (unsigned)(A_KC ? m0 * lda + kbeg : kbeg * lda + m0) * 4u;
        sb0 = (unsigned)(B_KC ? n0 * ldb + kbeg : kbeg * ldb + n0) * 4u;
    };
    auto fetch = [&](int step) {
#pragma unroll
        for (int u = 0; u < NUA; ++u) ra[u] = gemm_ld(sa, oa[u], __builtin_amdgcn_readfirstlane(sa0 + step * ka));
#pragma unroll
        for (int u = 0; u < 2; ++u) rb[u] = gemm_ld(sb, ob[u], __builtin_amdgcn_readfirstlane(sb0 + step * kb_));
    };
    auto stage = [&](int buf) {
        if constexpr (A_KC) store_kc(As[buf], tid, ra); else store_xc(As[buf], tid, ra);
        if constexpr (B_KC) store_kc(Bs[buf], tid, rb); else store_xc(Bs[buf], tid, rb);
    };
    auto origin = [&](int tile, int &m0, int &n0) {
        if constexpr (OP == 2) {
            m0 = by * BM;
            n0 = bx * GBN;
        } else {
            m0 = (tile / ntn) * BM;
            n0 = (tile % ntn) * GBN;
        }
    };

    int tile = OP == 2 ? 0 : bx, m0 = 0, n0 = 0;
    if (tile < ntiles) {
        origin(tile, m0, n0);
        locate(m0, n0);
        if (nsteps > 0) fetch(0);
    }
    while (tile < ntiles) {
        // accumulators start from the bias: register e of column tile tj is column 32*tj + 8*(e>>2) + 4h + (e&3)
        f32x16 acc[TI][TJ];
        const unsigned bcol = (unsigned)(n0 + wn * WCOLS + 4 * h) * 4u;
#pragma unroll
        for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const f32x4 bv = gemm_ld(sbias, bcol + (32 * tj + 8 * s) * 4u, 0);   // zeros without a bias or beyond N
#pragma unroll
                for (int ti = 0; ti < TI; ++ti)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[ti][tj][4 * s + c] = bv[c];
            }
        // The bias loads land in the accumulators themselves.  Left pending, the compiler's wait for them sits in front of the k loop's
        // first MFMAs -- INSIDE the loop, behind the step's fetch, as vmcnt(0): every k step of the 64-row kernels then waited for the
        // operand loads it had just issued (found in the ISA in round 6; the other three waves of the SIMD covered most of it: 20 us of
        // a 6.8-ms training step).  Waited for here, once per tile, beside the wait for the tile's first operands that stage(0) needs.
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
        const int next = tile + tstep;
        int m1 = 0, n1 = 0;
        if (next < ntiles) origin(next, m1, n1);
        float asum[CSUM ? TI : 1];
        if constexpr (CSUM) {
#pragma unroll
            for (int ti = 0; ti < TI; ++ti) asum[ti] = 0.f;
        }
        // LayerNorm epilogue: the residual rows are requested now, a whole k loop ahead of their use
        f32x4 ln_res[LN ? 2 : 1][LN ? 4 : 1];
        if constexpr (LN) {
            const int row = m0 + wm * 32 + j;
            const float *rp = ln->res + (size_t)min(row, M - 1) * N + wn * 64 + 4 * h;
#pragma unroll
            for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                for (int s = 0; s < 4; ++s) ln_res[tj][s] = *reinterpret_cast<const f32x4 *>(rp + 32 * tj + 8 * s);
        }
        f32x4 a_pre[ACTB ? 2 : 1][ACTB ? 4 : 1];   // activation backward: the pre-activations, requested a k loop ahead
        if constexpr (ACTB) {
            const int row = min(m0 + wm * 32 + j, M - 1);
            const float *ap = actb->a + (size_t)row * ldc + min(n0 + wn * 64 + 4 * h, N - 4);
#pragma unroll
            for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    a_pre[tj][s] = n0 + wn * 64 + 4 * h + 32 * tj + 8 * s < N ? *reinterpret_cast<const f32x4 *>(ap + 32 * tj + 8 * s)
                                                                               : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (nsteps > 0) stage(0);
        __syncthreads();
        auto step = [&](int it, auto bufc) {
            constexpr int buf = decltype(bufc)::value;
            if (it + 1 < nsteps) {
                fetch(it + 1);
            } else if (next < ntiles) {   // the next tile's first k-step rides under this one's last
                locate(m1, n1);
                fetch(0);
            }
            const float *as = As[buf] + h * GLD + wm * WROWS + j;
            const float *bs = Bs[buf] + h * GLD + wn * WCOLS + j;
#pragma unroll
            for (int kb = 0; kb < GBK / 2; ++kb) {
                float bv[TJ];
#pragma unroll
                for (int tj = 0; tj < TJ; ++tj) bv[tj] = bs[2 * kb * GLD + 32 * tj];
#pragma unroll
                for (int ti = 0; ti < TI; ++ti) {
                    const float av = as[2 * kb * GLD + 32 * ti];
#pragma unroll
                    for (int tj = 0; tj < TJ; ++tj) acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[tj], av, acc[ti][tj], 0, 0, 0);
                    if constexpr (CSUM) asum[ti] += av;
                }
            }
            if (it + 1 < nsteps) {
                stage(buf ^ 1);
                __syncthreads();
            }
        };
        for (int it = 0; it < nsteps; it += 2) {
            step(it, std::integral_constant<int, 0>{});
            if (it + 1 < nsteps) step(it + 1, std::integral_constant<int, 1>{});
        }
        if constexpr (LN) {
            // lane (j, h) of wave (wm, wn) holds 32 of the 128 values of row m0 + 32 wm + j: columns 64 wn + 32 tj + 8 s + 4 h + c
            static_assert(BM == 64, "the LayerNorm epilogue is written for 64-row tiles");
            const int row = m0 + wm * 32 + j, lr = wm * 32 + j;
            const bool rowok = row < M;
            const uint32_t rw = dropmask_row_word(ln->seed, (uint32_t)row);
            const size_t rbase = (size_t)row * N + wn * 64 + 4 * h;
            f32x4 v[2][4];
            float sum = 0.f;
#pragma unroll
            for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int col = wn * 64 + 32 * tj + 8 * s + 4 * h;
                    f32x4 y = {acc[0][tj][4 * s], acc[0][tj][4 * s + 1], acc[0][tj][4 * s + 2], acc[0][tj][4 * s + 3]};
                    if (ln->threshold) {
                        const u32x4g cw = *reinterpret_cast<const u32x4g *>(ln_colw + col);
#pragma unroll
                        for (int c = 0; c < 4; ++c) y[c] = dropmask_keep(rw, cw[c], ln->threshold) ? y[c] * ln->keep_scale : 0.f;
                    }
                    v[tj][s] = ln_res[tj][s] + y;
                    if (rowok) *reinterpret_cast<f32x4 *>(ln->s_out + rbase + 32 * tj + 8 * s) = v[tj][s];
                    sum += (v[tj][s][0] + v[tj][s][1]) + (v[tj][s][2] + v[tj][s][3]);
                }
            sum += __shfl_xor(sum, 32);
            if (h == 0) ln_red[lr * 2 + wn] = sum;
            __syncthreads();
            const float mean = (ln_red[lr * 2] + ln_red[lr * 2 + 1]) * (1.f / GBN);
            float var = 0.f;
#pragma unroll
            for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int c = 0; c < 4; ++c) var = fmaf(v[tj][s][c] - mean, v[tj][s][c] - mean, var);
            var += __shfl_xor(var, 32);
            if (h == 0) ln_red[128 + lr * 2 + wn] = var;
            __syncthreads();
            const float rstd = rsqrtf((ln_red[128 + lr * 2] + ln_red[128 + lr * 2 + 1]) * (1.f / GBN) + ln->eps);
            if (rowok) {
#pragma unroll
                for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const int col = wn * 64 + 32 * tj + 8 * s + 4 * h;
                        const f32x4 g4 = *reinterpret_cast<const f32x4 *>(ln->gamma + col), b4 = *reinterpret_cast<const f32x4 *>(ln->beta + col);
                        f32x4 o;
#pragma unroll
                        for (int c = 0; c < 4; ++c) o[c] = fmaf((v[tj][s][c] - mean) * rstd, g4[c], b4[c]);
                        *reinterpret_cast<f32x4 *>(C + rbase + 32 * tj + 8 * s) = o;
                    }
                if (wn == 0 && h == 0) {
                    ln->stats[2 * (size_t)row] = mean;
                    ln->stats[2 * (size_t)row + 1] = rstd;
                }
            }
        } else if constexpr (ACT) {
            static_assert(BM == 64, "the activation epilogue is written for 64-row tiles");
            const int row = m0 + wm * 32 + j;
            const uint32_t rw = dropmask_row_word(act->seed, (uint32_t)row);
            const size_t rbase = (size_t)row * ldc + n0 + wn * 64 + 4 * h;
#pragma unroll
            for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int col = n0 + wn * 64 + 32 * tj + 8 * s + 4 * h;
                    if (row >= M || col >= N) continue;
                    const f32x4 v = {acc[0][tj][4 * s], acc[0][tj][4 * s + 1], acc[0][tj][4 * s + 2], acc[0][tj][4 * s + 3]};
                    *reinterpret_cast<f32x4 *>(C + rbase + 32 * tj + 8 * s) = v;
                    f32x2 g0 = {v[0], v[1]}, g1 = {v[2], v[3]};
                    if (act->act == AFT_ACT_GELU) {
                        g0 = activate2<AFT_ACT_GELU>(g0);
                        g1 = activate2<AFT_ACT_GELU>(g1);
                    } else {
                        g0 = activate2<AFT_ACT_RELU>(g0);
                        g1 = activate2<AFT_ACT_RELU>(g1);
                    }
                    f32x4 y = {g0[0], g0[1], g1[0], g1[1]};
                    if (act->threshold) {
                        const u32x4g cw = *reinterpret_cast<const u32x4g *>(ln_colw + col);
#pragma unroll
                        for (int c = 0; c < 4; ++c) y[c] = dropmask_keep(rw, cw[c], act->threshold) ? y[c] * act->keep_scale : 0.f;
                    }
                    *reinterpret_cast<f32x4 *>(act->hd + rbase + 32 * tj + 8 * s) = y;
                }
        } else if constexpr (ACTB) {
            static_assert(BM == 64, "the activation-backward epilogue is written for 64-row tiles");
            const int row = m0 + wm * 32 + j;
            const uint32_t rw = dropmask_row_word(actb->seed, (uint32_t)row);
            const size_t rbase = (size_t)row * ldc + n0 + wn * 64 + 4 * h;
#pragma unroll
            for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int col = n0 + wn * 64 + 32 * tj + 8 * s + 4 * h;
                    if (row >= M || col >= N) continue;
                    const f32x4 av = a_pre[tj][s];
                    f32x2 d0 = {av[0], av[1]}, d1 = {av[2], av[3]};
                    if (actb->act == AFT_ACT_GELU) {
                        d0 = activate2_grad<AFT_ACT_GELU>(d0);
                        d1 = activate2_grad<AFT_ACT_GELU>(d1);
                    } else {
                        d0 = activate2_grad<AFT_ACT_RELU>(d0);
                        d1 = activate2_grad<AFT_ACT_RELU>(d1);
                    }
                    f32x4 y = {acc[0][tj][4 * s] * d0[0], acc[0][tj][4 * s + 1] * d0[1], acc[0][tj][4 * s + 2] * d1[0],
                               acc[0][tj][4 * s + 3] * d1[1]};
                    if (actb->threshold) {
                        const u32x4g cw = *reinterpret_cast<const u32x4g *>(ln_colw + col);
#pragma unroll
                        for (int c = 0; c < 4; ++c) y[c] = dropmask_keep(rw, cw[c], actb->threshold) ? y[c] * actb->keep_scale : 0.f;
                    }
                    *reinterpret_cast<f32x4 *>(C + rbase + 32 * tj + 8 * s) = y;
                }
        } else {
        // epilogue: lane j owns row j of each 32-row tile; 16-byte stores along the row
        const bool full = m0 + BM <= M && n0 + GBN <= N;
#pragma unroll
        for (int ti = 0; ti < TI; ++ti) {
            const int row = m0 + wm * WROWS + ti * 32 + j;
            float *crow = C + (size_t)row * ldc + n0 + wn * WCOLS + 4 * h;
#pragma unroll
            for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    if (!full && (row >= M || n0 + wn * WCOLS + 4 * h + 32 * tj + 8 * s >= N)) continue;
                    f32x4 *p = reinterpret_cast<f32x4 *>(crow + 32 * tj + 8 * s);
                    f32x4 v = {acc[ti][tj][4 * s], acc[ti][tj][4 * s + 1], acc[ti][tj][4 * s + 2], acc[ti][tj][4 * s + 3]};
                    if (accumulate) v += *p;
                    *p = v;
                }
        }
        }
        if constexpr (CSUM) {   // the n-tile-0 workgroup's left column of waves holds every A column of its m tile once
            if (bx == 0 && wn == 0 && csum_out != nullptr) {
#pragma unroll
                for (int ti = 0; ti < TI; ++ti) {
                    const float v = asum[ti] + __shfl_xor(asum[ti], 32);   // the two k parities
                    const int m = m0 + wm * WROWS + ti * 32 + j;
                    if (h == 0 && m < M) csum_out[m] = v;
                }
            }
        }
        __syncthreads();   // every wave is done with the LDS buffers before the next tile restages buffer 0
        tile = next;
        m0 = m1;
        n0 = n1;
    }
}

template <int OP, int BM>
__global__ __launch_bounds__(256) void gemm_fast_kernel(const float *__restrict__ A, const float *__restrict__ B,
                                                        float *__restrict__ C, const float *__restrict__ bias, int M, int N,
                                                        int K, int lda, int ldb, int ldc, int k_chunk, size_t c_slice,
                                                        int accumulate) {
    __shared__ __attribute__((aligned(16))) float As[2][GBK * GLD];
    __shared__ __attribute__((aligned(16))) float Bs[2][GBK * GLD];
    gemm_fast_body<OP, BM, EPI_NONE, false, BM == 96 ? 4 : 2>(As, Bs, A, B, C, bias, M, N, K, lda, ldb, ldc, k_chunk, c_slice, accumulate, (int)blockIdx.x,
                           (int)blockIdx.y, (int)blockIdx.z, (int)gridDim.x);
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void gemm_add_ln_kernel(const float *__restrict__ A, const float *__restrict__ B,
                                                          float *__restrict__ C, const float *__restrict__ bias, int M, int N, int K,
                                                          int lda, int ldb, const LnArgs ln) {
    __shared__ __attribute__((aligned(16))) float As[2][GBK * GLD];
    __shared__ __attribute__((aligned(16))) float Bs[2][GBK * GLD];
    gemm_fast_body<0, 64, EPI_LN>(As, Bs, A, B, C, bias, M, N, K, lda, ldb, N, K, (size_t)0, 0, (int)blockIdx.x, 0, 0, (int)gridDim.x, &ln);
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void gemm_act_kernel(
    const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ C, const float *__restrict__ bias, int M, int N, int K,
    int lda, int ldb, int ldc, const ActArgs act) {
    __shared__ __attribute__((aligned(16))) float As[2][GBK * GLD];
    __shared__ __attribute__((aligned(16))) float Bs[2][GBK * GLD];
    gemm_fast_body<0, 64, EPI_ACT>(As, Bs, A, B, C, bias, M, N, K, lda, ldb, ldc, K, (size_t)0, 0, (int)blockIdx.x, 0, 0, (int)gridDim.x,
                                   nullptr, &act);
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void gemm_actbwd_kernel(
    const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ C, int M, int N, int K, int lda, int ldb, int ldc,
    const ActBwdArgs actb) {
    __shared__ __attribute__((aligned(16))) float As[2][GBK * GLD];
    __shared__ __attribute__((aligned(16))) float Bs[2][GBK * GLD];
    gemm_fast_body<1, 64, EPI_ACTBWD>(As, Bs, A, B, C, nullptr, M, N, K, lda, ldb, ldc, K, (size_t)0, 0, (int)blockIdx.x, 0, 0,
                                      (int)gridDim.x, nullptr, nullptr, nullptr, &actb);
}

// Several weight gradients dW_j = A_j^T B_j over the SAME token rows in one launch (the four of an encoder layer): with
// all their tiles in flight together each product needs far fewer row slices to fill the chip (8 tiles x 64 slices instead
// of 1-3 tiles x 170-256 each), so the partial slices written here and read back by the reduction shrink 3.5x, and three
// launch ramps per layer go away.
constexpr int kTnBatchMax = 4;
struct TnBatchJob {
    const float *A, *B;
    float *slices;
    float *csum_slices;   // [nz][M] column sums of A per row slice (NULL: not wanted)
    int M, N, lda, ldb, tiles_n, blocks_per_slice, first_block;
};
struct TnBatch {
    TnBatchJob job[kTnBatchMax];
    int njobs, R, nz, chunk;
};
__global__ __launch_bounds__(256) void gemm_tn_batch_kernel(const TnBatch q) {
    __shared__ __attribute__((aligned(16))) float As[2][GBK * GLD];
    __shared__ __attribute__((aligned(16))) float Bs[2][GBK * GLD];
    const int block = (int)blockIdx.x;
    int jb = 0;
    while (jb + 1 < q.njobs && block >= q.job[jb + 1].first_block) ++jb;
    const TnBatchJob &t = q.job[jb];
    const int local = block - t.first_block, z = local / t.blocks_per_slice, tile = local % t.blocks_per_slice;
    if (t.csum_slices != nullptr)
        gemm_fast_body<2, 128, EPI_NONE, true>(As, Bs, t.A, t.B, t.slices, nullptr, t.M, t.N, q.R, t.lda, t.ldb, t.N, q.chunk,
                                               (size_t)t.M * t.N, 0, tile % t.tiles_n, tile / t.tiles_n, z, 1, nullptr, nullptr,
                                               t.csum_slices + (size_t)z * t.M);
    else
        gemm_fast_body<2, 128>(As, Bs, t.A, t.B, t.slices, nullptr, t.M, t.N, q.R, t.lda, t.ldb, t.N, q.chunk, (size_t)t.M * t.N, 0,
                               tile % t.tiles_n, tile / t.tiles_n, z, 1);
}

// out[i] = (accumulate ? out[i] : 0) + sum_z slices[z * stride + i].  64 columns per workgroup, the
// slice index split over the 4 waves, eight loads in flight per thread; fixed summation order
// (deterministic run to run).  Up to three outputs of n columns each, laid side by side in a slice
// (LayerNorm backward: dgamma | dbeta | bias gradient), in one launch.
struct ReduceArgs {
    const float *slices;
    float *out[3];
    int n, nout, nz, accumulate;
    size_t stride;
};
__global__ __launch_bounds__(256) void reduce_slices_kernel(const ReduceArgs a) {
    __shared__ float part[4][64];
    const int c = threadIdx.x & 63, zq = threadIdx.x >> 6, i = blockIdx.x * 64 + c;   // i over nout * n columns
    float s[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) s[u] = 0.f;
    if (i < a.n * a.nout) {
        int z = zq;
        for (; z + 28 < a.nz; z += 32) {
#pragma unroll
            for (int u = 0; u < 8; ++u) s[u] += a.slices[(size_t)(z + 4 * u) * a.stride + i];
        }
        for (; z < a.nz; z += 4) s[0] += a.slices[(size_t)z * a.stride + i];
    }
    part[zq][c] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    __syncthreads();
    if (zq == 0 && i < a.n * a.nout) {
        const float t = (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]);
        float *o = a.out[i / a.n] + i % a.n;
        *o = a.accumulate ? *o + t : t;
    }
}

// Several reductions in ONE launch: the gradient kernels of a layer each leave their partial slices in their
// own scratch region and queue a job here; the layer's backward ends with a single flush (the reductions are
// a few microseconds each, i.e. mostly launch latency when issued one by one).
struct ReduceJobs {
    ReduceArgs job[kMaxReduceJobs];
    int first_block[kMaxReduceJobs + 1];
    int njobs;
};
__global__ __launch_bounds__(256) void reduce_jobs_kernel(const ReduceJobs q) {
    __shared__ float part[8][64];
    int jb = 0;
    while (jb + 1 < q.njobs && (int)blockIdx.x >= q.first_block[jb + 1]) ++jb;
    const ReduceArgs &a = q.job[jb];
    // round 6: when the columns pair up (even stride and count -- every weight-gradient job) a thread takes TWO adjacent columns with
    // 8-byte loads and the slice index is split over eight thread groups instead of four: half the load instructions for the same
    // bytes in flight and the same workgroup count (16-byte loads with a quarter of the workgroups measured 4 x slower in round 5).
    // Fixed summation order as before (deterministic run to run); the order differs from the 4-byte form's.
    if (((a.stride | (size_t)(a.n * a.nout)) & 1) == 0 && (reinterpret_cast<uintptr_t>(a.slices) & 7) == 0) {
        const int c2 = threadIdx.x & 31, zg = threadIdx.x >> 5, i2 = (blockIdx.x - q.first_block[jb]) * 64 + 2 * c2;
        f32x2 t2[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t2[u] = f32x2{0.f, 0.f};
        if (i2 < a.n * a.nout) {
            int z = zg;
            for (; z + 56 < a.nz; z += 64) {
#pragma unroll
                for (int u = 0; u < 8; ++u) t2[u] += *reinterpret_cast<const f32x2 *>(a.slices + (size_t)(z + 8 * u) * a.stride + i2);
            }
            for (; z < a.nz; z += 8) t2[0] += *reinterpret_cast<const f32x2 *>(a.slices + (size_t)z * a.stride + i2);
        }
        const f32x2 tsum = ((t2[0] + t2[1]) + (t2[2] + t2[3])) + ((t2[4] + t2[5]) + (t2[6] + t2[7]));
        part[zg][2 * c2] = tsum[0];
        part[zg][2 * c2 + 1] = tsum[1];
        __syncthreads();
        const int c = threadIdx.x & 63, i = (blockIdx.x - q.first_block[jb]) * 64 + c;
        if (threadIdx.x < 64 && i < a.n * a.nout) {
            const float t = ((part[0][c] + part[1][c]) + (part[2][c] + part[3][c])) + ((part[4][c] + part[5][c]) + (part[6][c] + part[7][c]));
            float *o = a.out[i / a.n] + i % a.n;
            *o = a.accumulate ? *o + t : t;
        }
        return;
    }
    const int c = threadIdx.x & 63, zq = threadIdx.x >> 6, i = (blockIdx.x - q.first_block[jb]) * 64 + c;
    float s[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) s[u] = 0.f;
    if (i < a.n * a.nout) {
        int z = zq;
        for (; z + 28 < a.nz; z += 32) {
#pragma unroll
            for (int u = 0; u < 8; ++u) s[u] += a.slices[(size_t)(z + 4 * u) * a.stride + i];
        }
        for (; z < a.nz; z += 4) s[0] += a.slices[(size_t)z * a.stride + i];
    }
    part[zq][c] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    __syncthreads();
    if (zq == 0 && i < a.n * a.nout) {
        const float t = (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]);
        float *o = a.out[i / a.n] + i % a.n;
        *o = a.accumulate ? *o + t : t;
    }
}

static thread_local ReduceJobs *g_batch = nullptr;

ReduceBatchScope::ReduceBatchScope() : jobs(new ReduceJobs()) {
    jobs->njobs = 0;
    jobs->first_block[0] = 0;
    g_batch = jobs;
}
ReduceBatchScope::~ReduceBatchScope() {
    g_batch = nullptr;
    delete jobs;
}
hipError_t ReduceBatchScope::flush(hipStream_t st) {
    g_batch = nullptr;   // whatever follows launches directly again
    if (jobs->njobs == 0) return hipSuccess;
    hipLaunchKernelGGL(reduce_jobs_kernel, dim3(jobs->first_block[jobs->njobs]), dim3(256), 0, st, *jobs);
    jobs->njobs = 0;
    return hipGetLastError();
}

hipError_t launch_reduce_slices3(const float *slices, float *out0, float *out1, float *out2, int n, int nout, int nz,
                                 size_t stride, bool accumulate, hipStream_t st) {
    ReduceArgs a{slices, {out0, out1, out2}, n, nout, nz, (int)accumulate, stride};
    if (g_batch && g_batch->njobs < kMaxReduceJobs) {   // deferred: the caller keeps `slices` untouched until the flush
        ReduceJobs &q = *g_batch;
        q.job[q.njobs] = a;
        q.first_block[q.njobs + 1] = q.first_block[q.njobs] + (n * nout + 63) / 64;
        ++q.njobs;
        return hipSuccess;
    }
    hipLaunchKernelGGL(reduce_slices_kernel, dim3((n * nout + 63) / 64), dim3(256), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_reduce_slices(const float *slices, float *out, int n, int nz, size_t stride, bool accumulate,
                                hipStream_t st) {
    return launch_reduce_slices3(slices, out, nullptr, nullptr, n, 1, nz, stride, accumulate, st);
}

// column sums of a row-major [rows][n] matrix (bias gradients): one slice per row chunk
__global__ __launch_bounds__(256) void colsum_kernel(const float *__restrict__ x, float *__restrict__ slices, int rows,
                                                     int n, int ld, int chunk) {
    const int r0 = blockIdx.y * chunk, r1 = min(rows, r0 + chunk);
    const int c = threadIdx.x & 63, col = blockIdx.x * 64 + c, rq = threadIdx.x >> 6;
    __shared__ float part[4][64];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (col < n) {
        int r = r0 + rq;
        for (; r + 12 < r1; r += 16) {
            s0 += x[(size_t)r * ld + col];
            s1 += x[(size_t)(r + 4) * ld + col];
            s2 += x[(size_t)(r + 8) * ld + col];
            s3 += x[(size_t)(r + 12) * ld + col];
        }
        for (; r < r1; r += 4) s0 += x[(size_t)r * ld + col];
    }
    part[rq][c] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rq == 0 && col < n) slices[(size_t)blockIdx.y * n + col] = (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]);
}

// enough slices to put two workgroups on every CU whatever the number of output tiles, never fewer than 128 rows each
int gemm_split_slices(int rows, int tiles) {
    return std::max(1, std::min(std::min(kGemmMaxSlices, (512 + tiles - 1) / tiles), rows / 128));   // swept 192..1024: flat above 512
}
size_t gemm_tn_slice_floats(int M, int N, int R) {
    const int tiles = ((N + GBN - 1) / GBN) * ((M + GBM - 1) / GBM);
    return (size_t)gemm_split_slices(R, tiles) * M * N;
}
int colsum_slices(int rows) { return std::max(1, std::min(kColsumMaxSlices, rows / 64)); }

static bool gemm_vec_ok(int op, const float *A, const float *B, int M, int N, int K, int lda, int ldb) {
    // gfx950 global loads of 16 bytes only need dword alignment of the address; what matters is that the
    // four elements are consecutive in-bounds elements of one row
    (void)A; (void)B;
    const bool a = !(lda & 3) && !((op == 2 ? M : K) & 3);
    const bool b = !(ldb & 3) && !((op == 0 ? K : N) & 3);
    return a && b;
}

// conditions of gemm_fast_kernel: 16-byte accesses everywhere, whole k-steps, 31-bit buffer sizes
static bool gemm_fast_ok(int op, int M, int N, int K, int lda, int ldb, int ldc) {
    const long long a_bytes = 4ll * (op == 2 ? K : M) * lda, b_bytes = 4ll * (op == 0 ? N : K) * ldb;
    return !(lda & 3) && !(ldb & 3) && !(ldc & 3) && !(N & 3) && !(K % GBK) && (op != 2 || !(M & 3)) &&
           a_bytes < 0x7ffffff0ll && b_bytes < 0x7ffffff0ll;
}

template <int OP, int BM>
static void gemm_go(const float *A, const float *B, float *C, const float *bias, int M, int N, int K, int lda, int ldb,
                    int ldc, bool accumulate, hipStream_t st) {
    const int ntiles = ((N + GBN - 1) / GBN) * ((M + BM - 1) / BM);
    const int cus = current_device_cus();
    const dim3 grid(std::min(ntiles, cus * (BM == 128 ? 2 : BM == 96 ? 3 : 4)), 1, 1);   // resident workgroups per CU (VGPR bound); swept 2..8
    if (gemm_fast_ok(OP, M, N, K, lda, ldb, ldc))
        hipLaunchKernelGGL((gemm_fast_kernel<OP, BM>), grid, dim3(256), 0, st, A, B, C, bias, M, N, K, lda, ldb, ldc, K, (size_t)0,
                           (int)accumulate);
    else if constexpr (BM == 96)
        return;   // (not reached: launch_gemm asks for 96 rows on the aligned path only)
    else if (gemm_vec_ok(OP, A, B, M, N, K, lda, ldb))
        hipLaunchKernelGGL((gemm_kernel<OP, BM, true>), grid, dim3(256), 0, st, A, B, C, bias, M, N, K, lda, ldb, ldc, K, (size_t)0,
                           (int)accumulate);
    else
        hipLaunchKernelGGL((gemm_kernel<OP, BM, false>), grid, dim3(256), 0, st, A, B, C, bias, M, N, K, lda, ldb, ldc, K, (size_t)0,
                           (int)accumulate);
}

hipError_t launch_gemm(int op, const float *A, const float *B, float *C, const float *bias, int M, int N, int K, int lda,
                       int ldb, int ldc, bool accumulate, hipStream_t st) {
    // 64-row tiles: the tile-walking kernel then needs 120 VGPRs (4 workgroups per CU) instead of 192 (2), which
    // measures faster on every shape of the training step than 128-row tiles.  96-row tiles (3 workgroups per CU) where they save a
    // ROUND of tiles over the resident workgroups (round 6; the model below counts rounds x rows a CU works through) -- the same bits.
    // Only for one column tile (N <= 128): the 96-row kernel itself is 5-10 % slower per flop (three waves per SIMD, 24 MFMAs between
    // barriers) and wins only where the 64-row form leaves a lone last round -- 71 680 rows x (384 -> 128): 70.5 -> 67.9 us; with
    // several column tiles it measured slower at every row count (tools/debug/gemm_tile_ab.py).
    int rows = 64;
    if (gemm_fast_ok(op, M, N, K, lda, ldb, ldc) && (N <= GBN || switch_int("AFT_GEMM_BM", 0) == 96)) {
        const int forced = switch_int("AFT_GEMM_BM", 0);
        const long long cus = current_device_cus(), ntn = (N + GBN - 1) / GBN;
        auto cost = [&](int bm, int per_cu) {
            const long long tiles = (M + bm - 1) / bm * ntn, slots = cus * per_cu;
            return (tiles + slots - 1) / slots * bm * per_cu;
        };
        rows = forced == 64 || forced == 96 ? forced : cost(96, 3) < cost(64, 4) ? 96 : 64;
    }
    if (op == 0 && rows == 96) {
        gemm_go<0, 96>(A, B, C, bias, M, N, K, lda, ldb, ldc, accumulate, st);
    } else if (op == 1 && rows == 96) {
        gemm_go<1, 96>(A, B, C, bias, M, N, K, lda, ldb, ldc, accumulate, st);
    } else if (op == 0) {
        gemm_go<0, 64>(A, B, C, bias, M, N, K, lda, ldb, ldc, accumulate, st);
    } else if (op == 1) {
        gemm_go<1, 64>(A, B, C, bias, M, N, K, lda, ldb, ldc, accumulate, st);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// dW[M][N] (+)= A[R][M]^T . B[R][N]; `slices` holds up to kGemmMaxSlices * 128*128 * tiles floats (<= 512 tiles*slices)
hipError_t launch_gemm_tn(const float *A, const float *B, float *C, float *slices, int M, int N, int R, int lda, int ldb,
                          bool accumulate, hipStream_t st) {
    const int tiles = ((N + GBN - 1) / GBN) * ((M + GBM - 1) / GBM);
    const int nz = gemm_split_slices(R, tiles);
    const int chunk = ((R + nz - 1) / nz + GBK - 1) / GBK * GBK;
    const dim3 grid((N + GBN - 1) / GBN, (M + GBM - 1) / GBM, nz);
    if (gemm_fast_ok(2, M, N, R, lda, ldb, N))
        hipLaunchKernelGGL((gemm_fast_kernel<2, 128>), grid, dim3(256), 0, st, A, B, slices, (const float *)nullptr, M, N, R, lda, ldb, N,
                           chunk, (size_t)M * N, 0);
    else if (gemm_vec_ok(2, A, B, M, N, R, lda, ldb))
        hipLaunchKernelGGL((gemm_kernel<2, 128, true>), grid, dim3(256), 0, st, A, B, slices, (const float *)nullptr, M, N, R, lda, ldb, N,
                           chunk, (size_t)M * N, 0);
    else
        hipLaunchKernelGGL((gemm_kernel<2, 128, false>), grid, dim3(256), 0, st, A, B, slices, (const float *)nullptr, M, N, R, lda, ldb,
                           N, chunk, (size_t)M * N, 0);
    return launch_reduce_slices(slices, C, M * N, nz, (size_t)M * N, accumulate, st);
}

// out = LayerNorm(res + drop(A W^T + bias)) * gamma + beta in one launch when the fused epilogue covers the shape
// (N = 128, aligned operands); returns false without launching otherwise (the caller runs GEMM + add_ln_fwd).
bool gemm_add_ln_ok(int M, int N, int K, int lda, int ldb) { return N == GBN && gemm_fast_ok(0, M, N, K, lda, ldb, N); }
hipError_t launch_gemm_add_ln(const float *A, const float *W, const float *bias, const float *res, const float *gamma,
                              const float *beta, float *s_out, float *stats, float *out, int M, int N, int K, int lda, int ldw,
                              float eps, float keep_scale, uint32_t threshold, uint32_t seed, hipStream_t st) {
    if (!gemm_add_ln_ok(M, N, K, lda, ldw)) return hipErrorInvalidValue;
    const LnArgs ln{res, gamma, beta, s_out, stats, eps, keep_scale, seed, threshold};
    const int ntiles = (M + 63) / 64;
    hipLaunchKernelGGL(gemm_add_ln_kernel, dim3(std::min(ntiles, current_device_cus() * 4)), dim3(256), 0, st, A, W, out, bias, M, N, K,
                       lda, ldw, ln);
    return hipGetLastError();
}

// a = A W^T + bias (stored) and hd = drop(act(a)) in one launch when the fused epilogue covers the shape
bool gemm_act_ok(int M, int N, int K, int lda, int ldb, int ldc) { return N <= kActMaxN && gemm_fast_ok(0, M, N, K, lda, ldb, ldc); }
hipError_t launch_gemm_act(const float *A, const float *W, const float *bias, float *a_out, float *hd, int M, int N, int K, int lda,
                           int ldw, int ldc, int activation, float keep_scale, uint32_t threshold, uint32_t seed, hipStream_t st) {
    if (!gemm_act_ok(M, N, K, lda, ldw, ldc)) return hipErrorInvalidValue;
    const ActArgs act{hd, keep_scale, seed, threshold, activation};
    const int ntiles = ((N + GBN - 1) / GBN) * ((M + 63) / 64);
    hipLaunchKernelGGL(gemm_act_kernel, dim3(std::min(ntiles, current_device_cus() * 4)), dim3(256), 0, st, A, W, a_out, bias, M, N, K, lda,
                       ldw, ldc, act);
    return hipGetLastError();
}

bool gemm_actbwd_ok(int M, int N, int K, int lda, int ldb, int ldc) { return N <= kActMaxN && gemm_fast_ok(1, M, N, K, lda, ldb, ldc); }
hipError_t launch_gemm_actbwd(const float *dy, const float *W, const float *a, float *da, int M, int N, int K, int lda, int ldw,
                              int ldc, int activation, float keep_scale, uint32_t threshold, uint32_t seed, hipStream_t st) {
    if (!gemm_actbwd_ok(M, N, K, lda, ldw, ldc)) return hipErrorInvalidValue;
    const ActBwdArgs actb{a, keep_scale, seed, threshold, activation};
    const int ntiles = ((N + GBN - 1) / GBN) * ((M + 63) / 64);
    hipLaunchKernelGGL(gemm_actbwd_kernel, dim3(std::min(ntiles, current_device_cus() * 4)), dim3(256), 0, st, dy, W, da, M, N, K, lda, ldw,
                       ldc, actb);
    return hipGetLastError();
}

// Batched weight gradients: dW_j[M_j][N_j] (+)= A_j[R][M_j]^T . B_j[R][N_j], j < n <= 4, one GEMM launch; the slice reductions go
// through launch_reduce_slices as usual (queued when a ReduceBatchScope is alive).  Falls back to one launch per product
// when a shape is not on the aligned path.
hipError_t launch_gemm_tn_batch(const float *const *A, const float *const *B, float *const *C, float *const *slices, const int *M,
                                const int *N, const int *lda, const int *ldb, int n, int R, bool accumulate, hipStream_t st,
                                float *const *colsum_out, float *const *colsum_slices) {
    bool fast = n >= 1 && n <= kTnBatchMax;
    for (int j = 0; j < n && fast; ++j) fast = gemm_fast_ok(2, M[j], N[j], R, lda[j], ldb[j], N[j]);
    if (!fast) {
        for (int j = 0; j < n; ++j) {
            hipError_t e = launch_gemm_tn(A[j], B[j], C[j], slices[j], M[j], N[j], R, lda[j], ldb[j], accumulate, st);
            if (e == hipSuccess && colsum_out && colsum_out[j])
                e = launch_colsum(A[j], colsum_out[j], colsum_slices[j], R, M[j], lda[j], accumulate, st);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    }
    TnBatch q{};
    int tiles = 0;
    for (int j = 0; j < n; ++j) tiles += ((N[j] + GBN - 1) / GBN) * ((M[j] + GBM - 1) / GBM);
    q.njobs = n;
    q.R = R;
    q.nz = gemm_split_slices(R, tiles);   // <= the slice count of any single product: slices[j] is large enough
    q.chunk = ((R + q.nz - 1) / q.nz + GBK - 1) / GBK * GBK;
    int blocks = 0;
    for (int j = 0; j < n; ++j) {
        const int tn = (N[j] + GBN - 1) / GBN, tm = (M[j] + GBM - 1) / GBM;
        q.job[j] = TnBatchJob{A[j], B[j], slices[j], colsum_out && colsum_out[j] ? colsum_slices[j] : nullptr, M[j], N[j], lda[j], ldb[j], tn,
                              tn * tm, blocks};
        blocks += tn * tm * q.nz;
    }
    hipLaunchKernelGGL(gemm_tn_batch_kernel, dim3(blocks), dim3(256), 0, st, q);
    hipError_t e = hipGetLastError();
    for (int j = 0; j < n && e == hipSuccess; ++j) {
        e = launch_reduce_slices(slices[j], C[j], M[j] * N[j], q.nz, (size_t)M[j] * N[j], accumulate, st);
        if (e == hipSuccess && colsum_out && colsum_out[j])
            e = launch_reduce_slices(colsum_slices[j], colsum_out[j], M[j], q.nz, (size_t)M[j], accumulate, st);
    }
    return e;
}

// db[n] (+)= sum_r x[r][n]; `slices` holds colsum_slices(rows) * n floats
hipError_t launch_colsum(const float *x, float *out, float *slices, int rows, int n, int ld, bool accumulate, hipStream_t st) {
    const int nz = colsum_slices(rows);
    const int chunk = (rows + nz - 1) / nz;
    hipLaunchKernelGGL(colsum_kernel, dim3((n + 63) / 64, nz), dim3(256), 0, st, x, slices, rows, n, ld, chunk);
    return launch_reduce_slices(slices, out, n, nz, (size_t)n, accumulate, st);
}

}  // namespace aft
