// k_chain_bwd.hip -- the row-local part of an encoder layer's BACKWARD pass as one gfx950 kernel.
//
// Reference semantics: autograd through nn.TransformerEncoderLayer (post-LN, constructed at reference
// src/models/blocks/encoders.py:44-55; called from TrainingLoop.train_epoch, src/main/trainer.py:195-233).  With the
// forward of k_train.hip / k_gemm.hip
//     s1 = x + drop1(attn Wo^T + bo)      x1 = LN1(s1)
//     a  = x1 W1^T + b1                   hd = drop2(act(a))
//     s2 = x1 + drop3(hd W2^T + b2)       x2 = LN2(s2)
// everything between the gradient of x2 and the gradient of the attention output is row-local:
//     ds2 = LN2'(g; s2)         g2  = drop3 (.) ds2                (gradient of linear2's output)
//     gff = (g2 W2) (.) drop2 (.) act'(a)                           (gradient of linear1's output)
//     dx1 = gff W1 + ds2        ds1 = LN1'(dx1; s1)                (ds1 = the residual branch of dL/dx)
//     g2b = drop1 (.) ds1       d_o = g2b Wo                        (gradient of the attention output)
// Round 2 ran this as five launches (LayerNorm backward x 2, linear2's data gradient with the activation backward as
// its epilogue, two plain data-gradient GEMMs: 294 us per layer at B = 128, each tensor crossing HBM between them).
// Here one workgroup owns 32 token rows through all of it -- the mirror image of chain_device.h's forward chain: the
// same transposed MFMA products (A = weight fragment, B = activation fragment, accumulator lane = token row), with the
// TRANSPOSED weights packed in fragment order (pack_weights_t_kernel), LayerNorm backward on registers with the two row
// sums merged across the W waves through a 1-KB LDS table, the dropout masks of the three sites evaluated from the
// factored hash (aft_internal.h::dropmask_*: one row word per lane and site, column words from an LDS table), the
// activation derivative from the shared erf polynomial.  What leaves the kernel is what the remaining launches need:
// g2 / gff / g2b row-major (operands of the batched weight-gradient GEMM, which also sums their columns into the bias
// gradients), d_o for the attention backward, ds1 as the residual part of dx, and per-tile column sums of
// (g (.) xhat, g) for the LayerNorm parameter gradients (cross-lane DPP reduction, fixed-order slice reduction later).
#include <algorithm>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "chain_device.h"

namespace aft {

struct ChainBwdArgs {
    const float *g;                     // dL/dx2 [rows][D]
    const float *s2, *st2, *a, *s1, *st1;   // tape: pre-norm sums, (mean, rstd) per row, linear1's pre-activation [rows][2D]
    const float *w2t, *w1t, *wot;       // fragment-packed W2^T (2D x D), W1^T (D x 2D), Wo^T (D x D)
    const float *gam2, *gam1;           // LayerNorm weights
    float *g2, *gff, *g2b, *d_o, *dx;   // outputs, row-major
    float *lnp;                         // [workgroups][4][D]: column sums of g (.) xhat2, g, dx1 (.) xhat1, dx1 over the workgroup's tiles
    int rows;
    uint32_t seed1, seed2, seed3, threshold;   // dropout sites: after out_proj, after the activation, after linear2
    float keep_scale;
};

template <int D>
struct ChainBwdShape {
    using S = ChainShape<D>;
    static constexpr int PAR = 2 * D;    // gamma2 | gamma1
    static constexpr int COLW = 4 * D;   // column words of site 1 (D), site 2 (2D), site 3 (D)
    static constexpr size_t LDS_BYTES = sizeof(float) * (size_t)(S::XB + S::HB + S::ST + PAR + COLW);
};

// ---- row-major tensors leave and enter the two kernels below as WHOLE CACHE LINES (round 4) ----
// A lane of the transposed products holds ONE token row: 16 bytes at (row r, feature 8 s + 4 h) per fragment s, so a wave's
// buffer_store_b128 of fragment s touched 64 different 16-byte pieces of 32 rows (a quad of lanes = four rows), and the tape's
// 1 408 floats per row left the forward as 44 such instructions per wave and tile.  Measured with the same bytes stored
// lane-linear (wrong layout, timing only): 207 -> 187 us for the forward chain with the in-projection tail, 148 -> 132 without.
// Now every 32 x 32 block takes one turn through LDS: written as the fragments it is (one 1-KB block per s, exactly the exchange
// buffers' blocks -- x1, hd, x2, g2, gff, g2b are published there anyway), read back with eight lanes per 128-byte row piece and
// stored as 8 rows x 128 bytes per instruction.  The block is SWIZZLED so that both accesses are conflict-free: lane l = r + 32 h
// of fragment s sits at 16-byte position l ^ (2 s + h) of block s (the operand reads of the products -- one block, all lanes --
// stay a permutation inside every 16-lane group), and a read-out instruction takes rows R, R + 8 in each 16-lane group (the eight
// pieces of a row land on eight different 16-byte bank groups, those of row R + 8 on the other eight).
// Row-major INPUTS take the same road the other way (fetch_lines / stage_lines inside the kernels: 8 rows x 128 bytes per load, staged
// into a block with the read-out's offsets, fragments read with pw[]), a tile ahead where the tile's first product needs them.
struct LineIo {
    int pw[4];     // float offset of this lane's fragment s inside block s (the products' operand reads and the publishing writes)
};
__device__ __forceinline__ LineIo make_line_io(int lane) {
    LineIo io;
#pragma unroll
    for (int s = 0; s < 4; ++s) io.pw[s] = (lane ^ (2 * s + (lane >> 5))) * 4;
    return io;
}
__device__ __forceinline__ void srd_store_c(Srd r, unsigned byte_off, unsigned const_off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(decltype(__builtin_amdgcn_raw_buffer_load_b128(r, 0, 0, 0)), v), r,
                                           byte_off + (const_off & 0xfffu), const_off & ~0xfffu, 0);
}
// this lane's fragment s of a 32 x 32 block -> its place in the four 1-KB blocks at `blk`
__device__ __forceinline__ void line_put(float *blk, const LineIo &io, int s, f32x4 v) {
    *reinterpret_cast<f32x4 *>(blk + s * 256 + io.pw[s]) = v;
}
// the 32 x 32 block at `blk` -> rows row0 .. row0 + 31, columns col .. col + 31 of a row-major [rows][LD] tensor: lane = (rho, c),
// c = the 16-byte piece of the 128-byte row piece, instruction i stores row 8 (rho & 1) + (rho >> 1) + 4 (i & 1) + 16 (i >> 1); rows
// beyond the tensor's end (ragged last tile) are not stored.  The lane's offsets are formed here, from a laundered lane index, so
// that they are not held in registers across the products (both kernels sit at the 168 registers three waves per SIMD allow).
// Same wave wrote the block: LDS serves a wave's requests in order, the wave barriers only pin the compiler's order.
template <int LD>
__device__ __forceinline__ void line_store(Srd dst, const float *blk, int row0, int col, int rows) {
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));
    const int rho = lane >> 3, c = lane & 7, brow = 8 * (rho & 1) + (rho >> 1);
    const int rd0 = (c >> 1) * 256 + ((32 * (c & 1) + brow) ^ c) * 4;
    const unsigned gbase = ((unsigned)(row0 + brow) * LD + col + 4 * c) * 4;
    const int rows_left = rows - row0 - brow;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(blk + (rd0 ^ (16 * (i & 1))) + 64 * (i >> 1));
        if (4 * (i & 1) + 16 * (i >> 1) < rows_left) srd_store_c(dst, gbase, (unsigned)((4 * (i & 1) + 16 * (i >> 1)) * LD * 4), v);
    }
    __builtin_amdgcn_wave_barrier();
}

// The same with the lane's part of the offsets held in three registers for the whole tile (chain_bwd_kernel has them to spare, the
// forward kernels do not): the per-call address arithmetic of line_store / fetch_lines / stage_lines was ~35 vector instructions, 13
// calls per tile = 30 % of the backward kernel's vector instructions.  (The tile's origin is ADDED to the lane offset, one instruction
// per call: with it in the scalar-offset operand the gff stores of tiles beyond a workgroup's first came out wrong at B = 128 --
// dW1 / db1 5e-2 off, everything else exact -- for a reason not found; only compile-time constants ride in the scalar offset.)
struct LineLane {
    int rd0;            // read-out / staging float offset of instruction 0 (instruction i: (rd0 ^ 16 (i & 1)) + 64 (i >> 1))
    unsigned p128;      // byte offset of (this lane's first row, its 16-byte piece) in a 128-wide tensor: brow x 512 + c x 16
};
__device__ __forceinline__ LineLane make_line_lane(int lane) {
    const int rho = lane >> 3, c = lane & 7, brow = 8 * (rho & 1) + (rho >> 1);
    return LineLane{(c >> 1) * 256 + ((32 * (c & 1) + brow) ^ c) * 4, (unsigned)brow * 512u + (unsigned)c * 16u};
}
template <int LD>
__device__ __forceinline__ unsigned line_lane_off(const LineLane &ll) { return ll.p128 + (ll.p128 & ~127u) * (LD / 128 - 1); }
// RAGGED = the row count is not a multiple of 32 (the last tile is partial): per-row guards; else only "no such tile" (row0 >= rows)
template <int LD, bool RAGGED>
__device__ __forceinline__ void line_store_c(Srd dst, const float *blk, const LineLane &ll, int row0, int col, int rows) {
    const unsigned vo = line_lane_off<LD>(ll), so = ((unsigned)row0 * LD + col) * 4;
    const int rows_left = RAGGED ? rows - row0 - (int)(ll.p128 >> 9) : 32;      // rows beyond the end: lane offset beyond the resource (dropped)
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int dr = 4 * (i & 1) + 16 * (i >> 1);
        const f32x4 v = *reinterpret_cast<const f32x4 *>(blk + (ll.rd0 ^ (16 * (i & 1))) + 64 * (i >> 1));
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(decltype(__builtin_amdgcn_raw_buffer_load_b128(dst, 0, 0, 0)), v), dst,
                                               (!RAGGED || dr < rows_left) ? vo + so : 0x80000000u, (unsigned)(dr * LD * 4), 0);
    }
    __builtin_amdgcn_wave_barrier();
}
// rows of [row0n, row0n + 32) beyond `rows` read as zero (lane offset beyond the resource: no memory access)
template <int LD, bool RAGGED>
__device__ __forceinline__ void fetch_lines_c(Srd src, const LineLane &ll, int row0n, int col, int rows, f32x4 (&v)[4]) {
    const unsigned so = ((unsigned)min(row0n, rows) * LD + col) * 4;
    const unsigned vo = row0n < rows ? line_lane_off<LD>(ll) + so : 0x80000000u;       // (wave-uniform) no such tile
    const int left = RAGGED ? rows - row0n - (int)(ll.p128 >> 9) : 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int dr = 4 * (i & 1) + 16 * (i >> 1);
        v[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(src, (!RAGGED || dr < left) ? vo : 0x80000000u, (unsigned)(dr * LD * 4), 0));
    }
}
__device__ __forceinline__ void stage_lines_c(float *blk, const LineLane &ll, const f32x4 (&v)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4 *>(blk + (ll.rd0 ^ (16 * (i & 1))) + 64 * (i >> 1)) = v[i];
}

// LayerNorm backward on the lane's 16 features: v = upstream gradient (in: dL/dy, out: dL/ds), sfrag = the pre-norm
// sum of the forward, (mean, rstd) of the row.  Also returns the lane's contributions to the parameter gradients.
template <int D>
__device__ __forceinline__ void layernorm_bwd_rows(f32x16 &v, const f32x4 (&sfrag)[4], float mean, float rstd, float *stats,
                                                   const float *gamma, int wave, int r, int h, f32x16 &dgam) {
    constexpr int W = D / 32;
    f32x16 xh;
    float p1 = 0.f, p2 = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const f32x4 g4 = *reinterpret_cast<const f32x4 *>(gamma + 8 * s + 4 * h);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = 4 * s + j;
            xh[e] = (sfrag[s][j] - mean) * rstd;
            dgam[e] = v[e] * xh[e];          // dgamma contribution (dbeta's is v itself)
            v[e] *= g4[j];
            p1 += v[e];
            p2 = fmaf(v[e], xh[e], p2);
        }
    }
    p1 += __shfl_xor(p1, 32);
    p2 += __shfl_xor(p2, 32);
    if (h == 0) *reinterpret_cast<float2 *>(stats + (r * W + wave) * 2) = make_float2(p1, p2);
    __syncthreads();
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int u = 0; u < W; ++u) {
        const float2 p = *reinterpret_cast<const float2 *>(stats + (r * W + u) * 2);
        a += p.x;
        b += p.y;
    }
    a *= (1.0f / D);
    b *= (1.0f / D);
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = rstd * (v[e] - a - xh[e] * b);
}

template <int D, int ACT, bool RAGGED>
__global__ __launch_bounds__(D * 2, D <= 128 ? 3 : 2) void chain_bwd_kernel(const ChainBwdArgs a) {
    using S = ChainShape<D>;
    using B = ChainBwdShape<D>;
    constexpr int W = S::WAVES;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *xb = smem;              // g2 then g2b, fragment order
    float *hb = xb + S::XB;        // gff, fragment order
    float *stats = hb + S::HB;     // LayerNorm-backward row sums
    float *par = stats + S::ST;    // gamma2 | gamma1
    uint32_t *colw = reinterpret_cast<uint32_t *>(par + B::PAR);   // site 1 [D] | site 2 [2D] | site 3 [D]

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int fb = 32 * w;
    const bool drop = a.threshold != 0;

    const Srd srd_g = make_srd(a.g), srd_s2 = make_srd(a.s2), srd_s1 = make_srd(a.s1), srd_a = make_srd(a.a);
    const Srd srd_w2t = make_srd(a.w2t), srd_w1t = make_srd(a.w1t), srd_wot = make_srd(a.wot);
    const Srd srd_g2 = make_srd(a.g2), srd_gff = make_srd(a.gff), srd_g2b = make_srd(a.g2b), srd_do = make_srd(a.d_o),
              srd_dx = make_srd(a.dx);
    // (W2^T: 2W tiles x W k-blocks like W1 in the forward, W1^T: W tiles x 2W k-blocks like W2: offsets formed inside the tile loop)

    for (int i = tid; i < D; i += S::THREADS) {
        par[i] = a.gam2[i];
        par[D + i] = a.gam1[i];
    }
    if (drop) {
        for (int i = tid; i < D; i += S::THREADS) {
            colw[i] = dropmask_col_word(a.seed1, (uint32_t)i);
            colw[3 * D + i] = dropmask_col_word(a.seed3, (uint32_t)i);
        }
        for (int i = tid; i < 2 * D; i += S::THREADS) colw[D + i] = dropmask_col_word(a.seed2, (uint32_t)i);
    }
    __syncthreads();

    // element e of a site's mask for this lane: keep_scale where kept, else 0 (cw4 = four consecutive column words)
    auto mask4 = [&](f32x4 v, uint32_t rw, const uint32_t *cw4) {
        using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
        const u32x4 cw = *reinterpret_cast<const u32x4 *>(cw4);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = dropmask_keep(rw, cw[j], a.threshold) ? v[j] * a.keep_scale : 0.f;
        return o;
    };

    const int ntiles = (a.rows + 31) / 32;
    float psum[2] = {0.f, 0.f};     // lane (r, h): column sum of feature fb + r, quantity h, for LayerNorm 2 and LayerNorm 1
    // The tile's row-major inputs arrive as whole cache lines (LineIo above), one 32 x 32 block per instruction group, and reach the
    // lanes' fragments through the wave's own blocks of the exchange buffers: g and s2 are fetched a tile AHEAD (issued before the
    // last product of the previous tile, staged into the wave's two hidden blocks at its end), a's two blocks are requested at the
    // tile start and staged into the hidden blocks behind the first product (where gff then replaces them), s1 is requested behind the
    // hidden barrier and staged into the wave's x block behind the second product (g2's readers are past that barrier).
    const Srd srd_st2 = make_srd(a.st2), srd_st1 = make_srd(a.st1);
    {
        f32x4 ng[4], ns[4];
        const int first = (int)blockIdx.x < ntiles ? (int)blockIdx.x * 32 : a.rows;
        const LineLane ll0 = make_line_lane(threadIdx.x & 63);
        fetch_lines_c<D, RAGGED>(srd_g, ll0, first, fb, a.rows, ng);
        fetch_lines_c<D, RAGGED>(srd_s2, ll0, first, fb, a.rows, ns);
        stage_lines_c(hb + w * 2048, ll0, ng);
        stage_lines_c(hb + w * 2048 + 1024, ll0, ns);
    }
#pragma unroll 1
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        // everything derived from the lane index is recomputed per tile from a laundered copy: as loop invariants those values
        // (row / half / fragment offsets) were two registers over the 168 the launch bound allows, spilled before the loop and
        // reloaded from scratch inside it (a scratch reload is a VMEM load whose wait drains vmcnt; here the
        // layer backward measured 801-805 us either way, so this is hygiene: no vector spills in any hot kernel)
        int tid_l = threadIdx.x;
        asm volatile("" : "+v"(tid_l));
        const int lane = tid_l & 63, r = lane & 31, h = lane >> 5;
        const unsigned w2t_off = (unsigned)(2 * w) * W * 1024 + lane * 4, w1t_off = (unsigned)w * (2 * W) * 1024 + lane * 4,
                       wot_off = (unsigned)w * W * 1024 + lane * 4;
        const int row0 = tile * 32;
        const int grow = min(row0 + r, a.rows - 1);
        const bool row_ok = row0 + r < a.rows;
        // swizzled fragment positions inside the exchange buffers' blocks (LineIo above)
        const LineIo io = make_line_io(lane);
        const LineLane ll_tile = make_line_lane(lane);
        // the partial-last-tile instantiation needs its registers for the per-row guards: it re-forms the two offsets at each use
        auto line_lane = [&]() -> LineLane {
            if constexpr (RAGGED) {
                int l = threadIdx.x & 63;
                asm volatile("" : "+v"(l));
                return make_line_lane(l);
            } else {
                return ll_tile;
            }
        };
        float *xw = xb + w * 1024, *hw = hb + w * 2048;     // this wave's blocks of the two exchange buffers
        unsigned lo = 0;
        asm volatile("" : "+v"(lo));    // keep the tile-invariant weight loads inside the loop (see chain_device.h)
        const unsigned w2t_lane = (w2t_off + lo) * 4, w1t_lane = (w1t_off + lo) * 4, wot_lane = (wot_off + lo) * 4;
        constexpr int PFD = 4, PFF = 2;
        WRing<1, PFD> ring_d;
        WRing<2, PFF> ring_ff;

        // ---- LayerNorm-2 backward ----
        f32x4 gq[4], sq[4];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            gq[s] = *reinterpret_cast<const f32x4 *>(hw + s * 256 + io.pw[s]);
            sq[s] = *reinterpret_cast<const f32x4 *>(hw + 1024 + s * 256 + io.pw[s]);
        }
        const f32x2 st2 = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(srd_st2, (unsigned)grow * 8, 0, 0));
        // linear1's pre-activations of this wave's two hidden blocks: requested now, staged behind the first product
        f32x4 al[2][4];
        fetch_lines_c<2 * D, RAGGED>(srd_a, line_lane(), row0, 2 * fb, a.rows, al[0]);
        fetch_lines_c<2 * D, RAGGED>(srd_a, line_lane(), row0, 2 * fb + 32, a.rows, al[1]);
        gemm_preload<W, 2, PFF, 1>(ring_ff, srd_w2t, w2t_lane);
        f32x16 cur, dgam;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) cur[4 * s + j] = row_ok ? gq[s][j] : 0.f;   // rows past the end contribute nothing
        // per-tile column sums for the LayerNorm parameter gradients.  Lanes are rows here, so the sum over the tile's 32 rows
        // is a cross-lane reduction: instead of 5 DPP steps per value (320 vector instructions per LayerNorm, and vector
        // instructions cost fp32-MFMA time) the wave transposes its [32 rows][32 features] x 2 blocks through the idle hidden
        // buffer (column index XOR row: conflict-free both ways) and lane (feature, quantity) adds its column -- 32 LDS reads
        // (free beside MFMAs) + 32 adds.  hb is idle at both call sites: behind the LayerNorm barrier every wave has finished
        // the product that read it.  The second wave barrier keeps the scratch from being rewritten before all lanes read it.
        float *tsc = hb + w * 2048;               // this wave's scratch: [2 quantities][32 rows][32], column index XOR row
        auto store_param_sums = [&](const f32x16 &dg, const f32x16 &db, int which) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = (8 * s + 4 * h + j) ^ r;
                    tsc[r * 32 + c] = dg[4 * s + j];
                    tsc[1024 + r * 32 + c] = db[4 * s + j];
                }
            __builtin_amdgcn_wave_barrier();
            const float *col = tsc + h * 1024;               // lane (r, h): feature fb + r of quantity h (0 = dgamma, 1 = dbeta)
            float sum = 0.f;
#pragma unroll 8
            for (int q = 0; q < 32; ++q) sum += col[q * 32 + (r ^ q)];
            psum[which] += sum;                                   // accumulated over this workgroup's tiles, stored once
            __builtin_amdgcn_wave_barrier();
        };
        {
            const f32x16 dbeta = cur;
            layernorm_bwd_rows<D>(cur, sq, st2[0], st2[1], stats, par + fb, w, r, h, dgam);
            store_param_sums(dgam, dbeta, 0);
        }
        const f32x16 ds2 = cur;     // joins dx1 behind the second product
        uint32_t rw1 = 0, rw2 = 0, rw3 = 0;
        if (drop) {
            rw1 = dropmask_row_word(a.seed1, (uint32_t)grow);
            rw2 = dropmask_row_word(a.seed2, (uint32_t)grow);
            rw3 = dropmask_row_word(a.seed3, (uint32_t)grow);
        }
        // g2 = drop3 (.) ds2: stored row-major for the weight-gradient launch, published in fragment order
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            f32x4 v = {cur[4 * s], cur[4 * s + 1], cur[4 * s + 2], cur[4 * s + 3]};
            if (drop) v = mask4(v, rw3, colw + 3 * D + fb + 8 * s + 4 * h);
            line_put(xw, io, s, v);
        }
        line_store_c<D, RAGGED>(srd_g2, xw, line_lane(), row0, fb, a.rows);
        __syncthreads();

        // ---- (g2 W2): hidden blocks 2w, 2w+1; epilogue = dropout-2 mask and activation derivative ----
        f32x16 acc_h[2];
        auto xb_frag = [&](int kb, int s) { return *reinterpret_cast<const f32x4 *>(xb + (kb * 4 + s) * 256 + io.pw[s]); };
        gemm_run<W, 2, PFF, 1, 0, decltype(xb_frag), 0x3>(ring_ff, srd_w2t, w2t_lane, acc_h, xb_frag);
        gemm_preload<2 * W, 1, PFD, 1>(ring_d, srd_w1t, w1t_lane);
        stage_lines_c(hw, line_lane(), al[0]);
        stage_lines_c(hw + 1024, line_lane(), al[1]);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const f32x4 aq = *reinterpret_cast<const f32x4 *>(hw + 1024 * t + s * 256 + io.pw[s]);   // gff takes its place below
                const f32x2 d0 = activate2_grad<ACT>(f32x2{aq[0], aq[1]});
                const f32x2 d1 = activate2_grad<ACT>(f32x2{aq[2], aq[3]});
                f32x4 v = {acc_h[t][4 * s] * d0[0], acc_h[t][4 * s + 1] * d0[1], acc_h[t][4 * s + 2] * d1[0], acc_h[t][4 * s + 3] * d1[1]};
                if (drop) v = mask4(v, rw2, colw + D + 2 * fb + 32 * t + 8 * s + 4 * h);
                line_put(hw + 1024 * t, io, s, v);
            }
            line_store_c<2 * D, RAGGED>(srd_gff, hw + 1024 * t, line_lane(), row0, 2 * fb + 32 * t, a.rows);
        }
        __syncthreads();
        // the pre-norm sum of LayerNorm 1: requested here, staged behind the second product
        f32x4 sl[4];
        fetch_lines_c<D, RAGGED>(srd_s1, line_lane(), row0, fb, a.rows, sl);
        const f32x2 st1 = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(srd_st1, (unsigned)grow * 8, 0, 0));

        // ---- dx1 = gff W1 + ds2 (the accumulator starts from ds2) ----
        f32x16 acc_d[1] = {ds2};
        gemm_run<2 * W, 1, PFD, 1, 0>(ring_d, srd_w1t, w1t_lane, acc_d, [&](int kb, int s) {
            return *reinterpret_cast<const f32x4 *>(hb + (kb * 4 + s) * 256 + io.pw[s]);
        });
        gemm_preload<W, 1, PFD, 1>(ring_d, srd_wot, wot_lane);
        f32x4 s1q[4];
        stage_lines_c(xw, line_lane(), sl);        // g2's readers are past the hidden barrier
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s = 0; s < 4; ++s) s1q[s] = *reinterpret_cast<const f32x4 *>(xw + s * 256 + io.pw[s]);
        cur = acc_d[0];
        if (!row_ok) {
#pragma unroll
            for (int e = 0; e < 16; ++e) cur[e] = 0.f;
        }
        {
            const f32x16 dbeta = cur;
            layernorm_bwd_rows<D>(cur, s1q, st1[0], st1[1], stats, par + D + fb, w, r, h, dgam);
            store_param_sums(dgam, dbeta, 1);
        }
        // ds1: the residual branch of dL/dx (the in-projection's data gradient is added by the next launch);
        // g2b = drop1 (.) ds1: stored row-major, published for the last product
        // (ds1 through the wave's hidden blocks: behind the LayerNorm barrier every wave has finished the product that read them;
        //  g2 was consumed by the first product, so g2b takes its place in the x block)
#pragma unroll
        for (int s = 0; s < 4; ++s) line_put(hw, io, s, f32x4{cur[4 * s], cur[4 * s + 1], cur[4 * s + 2], cur[4 * s + 3]});
        line_store_c<D, RAGGED>(srd_dx, hw, line_lane(), row0, fb, a.rows);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            f32x4 v = {cur[4 * s], cur[4 * s + 1], cur[4 * s + 2], cur[4 * s + 3]};
            if (drop) v = mask4(v, rw1, colw + fb + 8 * s + 4 * h);
            line_put(xw, io, s, v);
        }
        line_store_c<D, RAGGED>(srd_g2b, xw, line_lane(), row0, fb, a.rows);
        __syncthreads();
        // the next tile's g and s2 blocks: requested here, staged at the end of this tile
        f32x4 ng[4], ns[4];
        {
            const int ntile = tile + gridDim.x, row0n = ntile < ntiles ? ntile * 32 : a.rows;
            fetch_lines_c<D, RAGGED>(srd_g, line_lane(), row0n, fb, a.rows, ng);
            fetch_lines_c<D, RAGGED>(srd_s2, line_lane(), row0n, fb, a.rows, ns);
        }

        // ---- d_o = g2b Wo ----
        f32x16 acc_o[1];
        gemm_run<W, 1, PFD, 1, 0, decltype(xb_frag), 0x1>(ring_d, srd_wot, wot_lane, acc_o, xb_frag);
#pragma unroll
        for (int s = 0; s < 4; ++s) line_put(hw, io, s, f32x4{acc_o[0][4 * s], acc_o[0][4 * s + 1], acc_o[0][4 * s + 2], acc_o[0][4 * s + 3]});
        line_store_c<D, RAGGED>(srd_do, hw, line_lane(), row0, fb, a.rows);
        stage_lines_c(hw, line_lane(), ng);
        stage_lines_c(hw + 1024, line_lane(), ns);
        // LDS hazards across tiles: xb is rewritten behind the next tile's LayerNorm-2 barrier, hb behind two barriers,
        // the row-sum table of LayerNorm 2 behind the xb barrier above.
    }
    // one slice per workgroup: [blocks][4][D] = dgamma2 | dbeta2 | dgamma1 | dbeta1
#pragma unroll
    for (int which = 0; which < 2; ++which) a.lnp[((size_t)blockIdx.x * 4 + 2 * which + h) * D + fb + r] = psum[which];
}

// ---------------------------------------------------------------------------------------------------------------------------
// The row-local part of the training FORWARD of a layer as one launch (round 3): out_proj + dropout + residual + LayerNorm 1,
// linear1 + activation + dropout, linear2 + dropout + residual + LayerNorm 2 -- what gemm_add_ln_kernel, gemm_act_kernel and
// gemm_add_ln_kernel did in three launches (166 us per layer, x1 and the hidden activations re-read from HBM) -- with the
// activation tape the backward needs written from the epilogues: s1, (mean, rstd) 1, x1, a, hd, s2, (mean, rstd) 2 and the
// layer's output.  Same structure as the inference chain (chain_device.h, <MLP, !QKV>): weights from the fp32 fragment-packed
// image, bias as the accumulators' initial value; the attention output arrives row-major from attn_train_fwd_kernel.
struct ChainTrainArgs {
    const float *attn, *x;              // attention output [rows][D] (row-major), layer input (residual)
    const float *wo, *w1, *w2;          // fragment-packed weights (pack_weights_kernel, fp32 image)
    const float *bo, *b1, *b2, *g1, *be1, *g2, *be2;
    float *s1, *st1, *x1, *a, *hd, *s2, *st2, *x_out;   // the tape + the layer output, row-major
    // QKV instantiation: the NEXT layer's in-projection on this layer's output, row-major [rows][3 D] into the next layer's tape
    const float *wqkv, *bqkv;
    float *qkv_next;
    unsigned long long *stamps;   // diagnostic build only (AFT_DIAG_STAMPS): per tile, s_memtime of wave 0 at the phase boundaries [0..8], s_memrealtime at the start [9]
    int rows;
    uint32_t seed1, seed2, seed3, threshold;
    float keep_scale;
};

template <int D>
struct ChainTrainShape {
    using S = ChainShape<D>;
    static constexpr int COLW = 4 * D;
    // The two LayerNorm BETAS are read from global memory (below), which takes 2 D floats off the inference chain's parameter block:
    // 53 248 B.  At 54 272 B (betas in LDS too) only TWO workgroups are resident per CU although hipOccupancyMaxActiveBlocksPerMultiprocessor
    // says three -- LDS is handed out in 1 280-byte granules (tools/micro/lds_residency: 53 760 B is the last size of which three
    // fit) -- and the launch ran a full round of 512 workgroups and a second one of 256 at one workgroup per CU.
    static constexpr int PAR = 2 * D;    // g1 | g2
    static constexpr size_t LDS_BYTES = sizeof(float) * (size_t)(S::XB + S::HB + S::ST + PAR + COLW);
    static_assert(3 * ((LDS_BYTES + 1279) / 1280 * 1280) <= 160 * 1024 || D != 128, "three workgroups per CU need <= 53 760 B of LDS each");
};

// LayerNorm over D features on the lane's 16 values (chain_device.h::layernorm_rows) that also hands back (mean, rstd)
template <int D>
__device__ __forceinline__ void layernorm_rows_stats(f32x16 &v, float *stats, const float *gamma, Srd beta, int beta_f0, int wave, int r,
                                                     int h, float &mean_out, float &rstd_out) {
    constexpr int W = D / 32;
    f32x4 bfrag[4];                       // beta of the lane's 16 features, from global memory: in flight across the barrier below
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) bfrag[s4] = srd_load(beta, (unsigned)(beta_f0 + 8 * s4 + 4 * h) * 4);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) s += v[e];
    const float mp = s * (1.0f / 16.0f);
    float m2 = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) m2 = fmaf(v[e] - mp, v[e] - mp, m2);
    const float mo = __shfl_xor(mp, 32), m2o = __shfl_xor(m2, 32);
    const float dlt = mp - mo;
    if (h == 0) *reinterpret_cast<float2 *>(stats + (r * W + wave) * 2) = make_float2(0.5f * (mp + mo), m2 + m2o + 8.0f * dlt * dlt);
    __syncthreads();
    float mean = 0.f, msum = 0.f;
    float pm[W], pM[W];
#pragma unroll
    for (int u = 0; u < W; ++u) {
        const float2 p = *reinterpret_cast<const float2 *>(stats + (r * W + u) * 2);
        pm[u] = p.x;
        pM[u] = p.y;
        mean += p.x;
    }
    mean *= (1.0f / W);
#pragma unroll
    for (int u = 0; u < W; ++u) msum += pM[u] + 32.0f * (pm[u] - mean) * (pm[u] - mean);
    const float rstd = rsqrtf(msum * (1.0f / D) + 1e-5f);
    mean_out = mean;
    rstd_out = rstd;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        const f32x4 g = *reinterpret_cast<const f32x4 *>(gamma + 8 * s4 + 4 * h);
        const f32x4 b = bfrag[s4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[4 * s4 + j] = (v[4 * s4 + j] - mean) * rstd * g[j] + b[j];
    }
}

template <int D, int ACT, bool QKV>
__global__ __launch_bounds__(D * 2, D <= 128 ? 3 : 2) void chain_fwd_train_kernel(const ChainTrainArgs a) {
    using S = ChainShape<D>;
    constexpr int W = S::WAVES;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *xb = smem;
    float *hb = xb + S::XB;
    float *stats = hb + S::HB;
    float *par = stats + S::ST;                                     // g1 | g2
    uint32_t *colw = reinterpret_cast<uint32_t *>(par + ChainTrainShape<D>::PAR);   // site 1 [D] | site 2 [2D] | site 3 [D]

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int fb = 32 * w;
    const bool drop = a.threshold != 0;
    // (timing experiment, round 3: without the five tape stores below the launch takes 106 us instead of 145 -- the inference chain's
    //  time for the same products; the 39 us are store waits)
    constexpr bool TAPE = true;
#ifdef AFT_DIAG_STAMPS
#define TSTAMP(i) do { if (a.stamps && tid == 0) a.stamps[(size_t)tile * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TSTAMP(i) do { } while (0)
#endif

    const Srd srd_wo = make_srd(a.wo), srd_w1 = make_srd(a.w1), srd_w2 = make_srd(a.w2);
    const Srd srd_attn = make_srd(a.attn), srd_x = make_srd(a.x);
    const Srd srd_bo = make_srd(a.bo), srd_b1 = make_srd(a.b1), srd_b2 = make_srd(a.b2);
    const Srd srd_be1 = make_srd(a.be1), srd_be2 = make_srd(a.be2);
    const Srd srd_s1 = make_srd(a.s1), srd_x1 = make_srd(a.x1), srd_a = make_srd(a.a), srd_hd = make_srd(a.hd), srd_s2 = make_srd(a.s2),
              srd_out = make_srd(a.x_out);
    const unsigned wo_off = (unsigned)w * W * 1024 + lane * 4, w1_off = (unsigned)(2 * w) * W * 1024 + lane * 4;
    const unsigned w2_off = (unsigned)w * (2 * W) * 1024 + lane * 4;
    const Srd srd_wq = make_srd(a.wqkv), srd_bq = make_srd(a.bqkv), srd_qkv = make_srd(a.qkv_next);
    const unsigned wq_off = (unsigned)w * W * 1024 + lane * 4;   // tiles w, W + w, 2 W + w: the q, k, v features of block w

    for (int i = tid; i < D; i += S::THREADS) {
        par[i] = a.g1[i];
        par[D + i] = a.g2[i];
    }
    if (drop) {
        for (int i = tid; i < D; i += S::THREADS) {
            colw[i] = dropmask_col_word(a.seed1, (uint32_t)i);
            colw[3 * D + i] = dropmask_col_word(a.seed3, (uint32_t)i);
        }
        for (int i = tid; i < 2 * D; i += S::THREADS) colw[D + i] = dropmask_col_word(a.seed2, (uint32_t)i);
    }
    __syncthreads();
    auto mask4 = [&](f32x4 v, uint32_t rw, const uint32_t *cw4) {
        using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
        const u32x4 cw = *reinterpret_cast<const u32x4 *>(cw4);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = dropmask_keep(rw, cw[j], a.threshold) ? v[j] * a.keep_scale : 0.f;
        return o;
    };

    const int ntiles = (a.rows + 31) / 32;
    // The tile's two row-major inputs -- the attention output and the layer input (residual) -- arrive as whole cache lines too, a
    // tile AHEAD: every wave fetches only ITS feature block of both (eight line loads; until round 4 every wave fetched all W blocks
    // of the attention tile itself as 16 scattered loads into 64 registers), issued before the previous tile's last product and
    // staged at its end -- the attention block into the wave's EVEN hidden block (the out-projection reads its operand from there,
    // like every other product reads the exchange buffers), the residual block into the odd one.
    float *xw = xb + w * 1024, *hw = hb + w * 2048;     // this wave's blocks of the two exchange buffers
    f32x4 na[4], nx[4];
    auto fetch_tile = [&](int row0n) {      // row0n >= rows: every load is out of range (zeros, no memory access)
        int l = threadIdx.x & 63;
        asm volatile("" : "+v"(l));
        const int rho = l >> 3, c = l & 7, brow = 8 * (rho & 1) + (rho >> 1);
        const unsigned gb = ((unsigned)(row0n + brow) * D + fb + 4 * c) * 4;
        const int left = a.rows - row0n - brow;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int dr = 4 * (i & 1) + 16 * (i >> 1);
            const unsigned vo = dr < left ? gb : 0x80000000u;   // beyond the resource's 2^31 - 1 bytes: reads as zero
            na[i] = srd_load_c(srd_attn, vo, (unsigned)(dr * D * 4));
            nx[i] = srd_load_c(srd_x, vo, (unsigned)(dr * D * 4));
        }
    };
    auto stage_lines = [&](float *blk, const f32x4 (&v)[4]) {
        int l = threadIdx.x & 63;
        asm volatile("" : "+v"(l));
        const int rho = l >> 3, c = l & 7, brow = 8 * (rho & 1) + (rho >> 1);
        const int rd0 = (c >> 1) * 256 + ((32 * (c & 1) + brow) ^ c) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4 *>(blk + (rd0 ^ (16 * (i & 1))) + 64 * (i >> 1)) = v[i];
    };
    if ((int)blockIdx.x < ntiles) {
        fetch_tile(blockIdx.x * 32);
        stage_lines(hw, na);
        stage_lines(hw + 1024, nx);
    }
    __syncthreads();
#pragma unroll 1
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * 32;
        const int grow = min(row0 + r, a.rows - 1);
        const bool row_ok = row0 + r < a.rows;
        // swizzled fragment positions inside the exchange buffers' blocks (LineIo above)
        const LineIo io = make_line_io(lane);
        unsigned lo = 0;
        asm volatile("" : "+v"(lo));
        const unsigned wo_lane = (wo_off + lo) * 4, w1_lane = (w1_off + lo) * 4, w2_lane = (w2_off + lo) * 4;
        constexpr int PFD = 4, PFF = 2;
        WRing<1, PFD> ring_d;
        WRing<2, PFF> ring_ff;

        TSTAMP(0);
#ifdef AFT_DIAG_STAMPS
        if (a.stamps && tid == 0) a.stamps[(size_t)tile * 16 + 9] = __builtin_amdgcn_s_memrealtime();   // 100 MHz, one clock for the chip
#endif
        // ---- out-projection + bias, dropout 1, residual ----
        f32x16 acc_o[1] = {bias_acc(srd_bo, fb, h)};
        gemm_preload<W, 1, PFD, 1>(ring_d, srd_wo, wo_lane);
        f32x4 xres[4];      // this lane's fragments of the residual block (staged by this wave at the end of the previous tile)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s = 0; s < 4; ++s) xres[s] = *reinterpret_cast<const f32x4 *>(hw + 1024 + s * 256 + io.pw[s]);
        gemm_run<W, 1, PFD, 1, 0>(ring_d, srd_wo, wo_lane, acc_o, [&](int kb, int s) {
            return *reinterpret_cast<const f32x4 *>(hb + (2 * kb * 4 + s) * 256 + io.pw[s]);   // attention tile: the waves' even hidden blocks
        });
        TSTAMP(1);
        f32x16 acc_h[2] = {bias_acc(srd_b1, 2 * fb, h), bias_acc(srd_b1, 2 * fb + 32, h)};
        gemm_preload<W, 2, PFF, 1>(ring_ff, srd_w1, w1_lane);
        uint32_t rw1 = 0, rw2 = 0, rw3 = 0;
        if (drop) {
            rw1 = dropmask_row_word(a.seed1, (uint32_t)grow);
            rw2 = dropmask_row_word(a.seed2, (uint32_t)grow);
            rw3 = dropmask_row_word(a.seed3, (uint32_t)grow);
        }
        f32x16 cur;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            f32x4 y = {acc_o[0][4 * s], acc_o[0][4 * s + 1], acc_o[0][4 * s + 2], acc_o[0][4 * s + 3]};
            if (drop) y = mask4(y, rw1, colw + fb + 8 * s + 4 * h);
            const f32x4 v = xres[s] + y;
            if (TAPE) line_put(hw + 1024, io, s, v);    // the wave's odd hidden block: the residual fragments are out, the even one is still being read
#pragma unroll
            for (int j = 0; j < 4; ++j) cur[4 * s + j] = v[j];
        }
        if (TAPE) line_store<D>(srd_s1, hw + 1024, row0, fb, a.rows);
        float mean, rstd;
        layernorm_rows_stats<D>(cur, stats, par + fb, srd_be1, fb, w, r, h, mean, rstd);   // -> x1
        if (row_ok && w == 0 && h == 0) *reinterpret_cast<float2 *>(a.st1 + 2 * (size_t)grow) = make_float2(mean, rstd);
#pragma unroll
        for (int s = 0; s < 4; ++s)
            line_put(xw, io, s, f32x4{cur[4 * s], cur[4 * s + 1], cur[4 * s + 2], cur[4 * s + 3]});   // published, and stored from there
        if (TAPE) line_store<D>(srd_x1, xw, row0, fb, a.rows);
        __syncthreads();
        TSTAMP(2);
        // ---- linear1 + bias -> a (kept), activation, dropout 2 -> hd (kept, published) ----
        gemm_run<W, 2, PFF, 1, 0>(ring_ff, srd_w1, w1_lane, acc_h, [&](int kb, int s) {
            return *reinterpret_cast<const f32x4 *>(xb + (kb * 4 + s) * 256 + io.pw[s]);
        });
        TSTAMP(3);
        f32x16 acc_d[1] = {bias_acc(srd_b2, fb, h)};
        gemm_preload<2 * W, 1, PFD, 1>(ring_d, srd_w2, w2_lane);
        // a (block t) takes its turn through the wave's OTHER hidden block, hd is stored from where it is published; the LDS serves
        // the wave's requests in order, so a block is re-written right behind its read-out
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (TAPE) {
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    line_put(hw + 1024, io, s, f32x4{acc_h[t][4 * s], acc_h[t][4 * s + 1], acc_h[t][4 * s + 2], acc_h[t][4 * s + 3]});
                line_store<2 * D>(srd_a, hw + 1024, row0, 2 * fb + 32 * t, a.rows);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const f32x2 g0 = activate2<ACT>(f32x2{acc_h[t][4 * s], acc_h[t][4 * s + 1]});
                const f32x2 g1 = activate2<ACT>(f32x2{acc_h[t][4 * s + 2], acc_h[t][4 * s + 3]});
                f32x4 v = {g0[0], g0[1], g1[0], g1[1]};
                if (drop) v = mask4(v, rw2, colw + D + 2 * fb + 32 * t + 8 * s + 4 * h);
                line_put(hw + 1024 * t, io, s, v);
            }
            if (TAPE) line_store<2 * D>(srd_hd, hw + 1024 * t, row0, 2 * fb + 32 * t, a.rows);
        }
        __syncthreads();
        TSTAMP(4);
        {   // the next tile's attention / residual blocks: requested here, staged at the end of this tile
            const int ntile = tile + gridDim.x;
            fetch_tile(ntile < ntiles ? ntile * 32 : a.rows);
        }
        // ---- linear2 + bias, dropout 3, residual (x1, registers) ----
        gemm_run<2 * W, 1, PFD, 1, 0>(ring_d, srd_w2, w2_lane, acc_d, [&](int kb, int s) {
            return *reinterpret_cast<const f32x4 *>(hb + (kb * 4 + s) * 256 + io.pw[s]);
        });
        TSTAMP(5);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            f32x4 y = {acc_d[0][4 * s], acc_d[0][4 * s + 1], acc_d[0][4 * s + 2], acc_d[0][4 * s + 3]};
            if (drop) y = mask4(y, rw3, colw + 3 * D + fb + 8 * s + 4 * h);
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = cur[4 * s + j] + y[j];
            if (TAPE) line_put(xw, io, s, v);    // x1's readers are past the hidden barrier: the wave's x block is free
#pragma unroll
            for (int j = 0; j < 4; ++j) cur[4 * s + j] = v[j];
        }
        if (TAPE) line_store<D>(srd_s2, xw, row0, fb, a.rows);
        layernorm_rows_stats<D>(cur, stats, par + D + fb, srd_be2, fb, w, r, h, mean, rstd);   // -> x2
        if (row_ok && w == 0 && h == 0) *reinterpret_cast<float2 *>(a.st2 + 2 * (size_t)grow) = make_float2(mean, rstd);
        // x2: into the wave's x block (the in-projection tail's operand), stored from there
#pragma unroll
        for (int s = 0; s < 4; ++s) line_put(xw, io, s, f32x4{cur[4 * s], cur[4 * s + 1], cur[4 * s + 2], cur[4 * s + 3]});
        line_store<D>(srd_out, xw, row0, fb, a.rows);
        TSTAMP(6);
        if constexpr (QKV) {
            // ---- the next layer's in-projection on x2 (the tile is still in registers): q | k | v of feature block w, bias as the
            //      accumulators' initial value, row-major into the next layer's tape.  x1's readers are past the hidden barrier. ----
            constexpr int PFQ = 2;
            const unsigned wq_lane = (wq_off + lo) * 4;
            WRing<3, PFQ> ring_q;
            gemm_preload<W, 3, PFQ, W>(ring_q, srd_wq, wq_lane);
            f32x16 acc_q[3] = {bias_acc(srd_bq, fb, h), bias_acc(srd_bq, D + fb, h), bias_acc(srd_bq, 2 * D + fb, h)};
            __syncthreads();
            gemm_run<W, 3, PFQ, W, 0>(ring_q, srd_wq, wq_lane, acc_q, [&](int kb, int s) {
                return *reinterpret_cast<const f32x4 *>(xb + (kb * 4 + s) * 256 + io.pw[s]);
            });
            // q, k, v of feature block w: each through the wave's odd hidden block (free behind the barrier above)
            stage_lines(hw, na);      // (the hidden blocks' readers are past the barrier above)
#pragma unroll
            for (int t = 0; t < 3; ++t) {
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    line_put(hw + 1024, io, s, f32x4{acc_q[t][4 * s], acc_q[t][4 * s + 1], acc_q[t][4 * s + 2], acc_q[t][4 * s + 3]});
                line_store<3 * D>(srd_qkv, hw + 1024, row0, t * D + fb, a.rows);
            }
        } else {
            stage_lines(hw, na);      // (the hidden blocks' readers are past the LayerNorm-2 barrier)
        }
        stage_lines(hw + 1024, nx);
        TSTAMP(7);
        __syncthreads();    // the next tile's LayerNorm-1 partials must not overtake this tile's LayerNorm-2 readers
        TSTAMP(8);
    }
}

bool chain_fwd_train_ok(const aft_config &c, int rows) {
    return c.model_dim == 128 && (size_t)rows * 3 * c.model_dim * sizeof(float) < ((size_t)1 << 31);
}

template <int ACT, bool QKV>
static hipError_t launch_chain_fwd_train_t(const ChainTrainArgs &args, hipStream_t st) {
    constexpr int D = 128;
    using T = ChainTrainShape<D>;
    static PerDeviceOnce lds_attr;
    hipError_t ea = ensure_dynamic_lds(lds_attr, reinterpret_cast<const void *>(chain_fwd_train_kernel<D, ACT, QKV>), T::LDS_BYTES);
    if (ea != hipSuccess) return ea;
    const int blocks = std::min((args.rows + 31) / 32, current_device_cus() * 3);
#ifdef AFT_DIAG_STAMPS
    if (switch_on("AFT_STAMPS")) {   // diagnostic build only: mean cycles per phase of a tile (wave 0), and when the workgroups started
        const int ntiles = (args.rows + 31) / 32;
        static unsigned long long *dbuf = nullptr;
        if (!dbuf) (void)hipMalloc(&dbuf, sizeof(unsigned long long) * 16 * 8192);
        (void)hipMemsetAsync(dbuf, 0, sizeof(unsigned long long) * 16 * 8192, st);
        ChainTrainArgs a2 = args;
        a2.stamps = ntiles <= 8192 ? dbuf : nullptr;
        hipLaunchKernelGGL((chain_fwd_train_kernel<D, ACT, QKV>), dim3(blocks), dim3(ChainShape<D>::THREADS), T::LDS_BYTES, st, a2);
        (void)hipStreamSynchronize(st);
        static int printed = 0;
        if (a2.stamps && printed++ < 3) {
            std::vector<unsigned long long> hb(16 * (size_t)ntiles);
            (void)hipMemcpy(hb.data(), dbuf, hb.size() * 8, hipMemcpyDeviceToHost);
            const int nb = std::min(blocks, ntiles);
            unsigned long long t0 = ~0ull, tend = 0;
            for (int t = 0; t < nb; ++t) t0 = std::min(t0, hb[(size_t)t * 16 + 9]);
            for (int t = 0; t < ntiles; ++t) tend = std::max(tend, hb[(size_t)t * 16 + 9]);
            int late = 0;
            double mean = 0, mx = 0;
            for (int t = 0; t < nb; ++t) {
                const double us = (double)(hb[(size_t)t * 16 + 9] - t0) * 0.01;
                mean += us; mx = std::max(mx, us); late += us > 20.0;
            }
            printf("chain_fwd_train<QKV=%d>: %d workgroups, first tile starts %.1f us (mean) / %.1f us (max) after the earliest, %d later than 20 us; "
                   "the launch's last tile starts at %.1f us\n", (int)QKV, nb, mean / nb, mx, late, (double)(tend - t0) * 0.01);
            double sum[9] = {0};
            for (int t = 0; t < ntiles; ++t)
                for (int i = 1; i < 9; ++i) sum[i] += (double)(hb[(size_t)t * 16 + i] - hb[(size_t)t * 16 + i - 1]);
            printf("  mean cycles per tile: outproj=%.0f s1+LN1+x1+barrier=%.0f ffn_up=%.0f act+a+hd+barrier=%.0f ffn_down=%.0f s2+LN2+out=%.0f "
                   "qkv=%.0f end_barrier=%.0f total=%.0f\n", sum[1] / ntiles, sum[2] / ntiles, sum[3] / ntiles, sum[4] / ntiles, sum[5] / ntiles,
                   sum[6] / ntiles, sum[7] / ntiles, sum[8] / ntiles,
                   (sum[1] + sum[2] + sum[3] + sum[4] + sum[5] + sum[6] + sum[7] + sum[8]) / ntiles);
        }
        return hipGetLastError();
    }
#endif
    hipLaunchKernelGGL((chain_fwd_train_kernel<D, ACT, QKV>), dim3(blocks), dim3(ChainShape<D>::THREADS), T::LDS_BYTES, st, args);
    return hipGetLastError();
}

// packed: one layer's fp32 fragment image (packed_layer_floats(d) floats of scratch, filled here)
hipError_t launch_chain_fwd_train(const aft_config &c, const aft_layer_weights &w, const float *attn, const float *x, float *packed,
                                  float *s1, float *st1, float *x1, float *a_pre, float *hd, float *s2, float *st2, float *x_out,
                                  int rows, uint32_t seed1, uint32_t seed2, uint32_t seed3, uint32_t threshold, float keep_scale,
                                  hipStream_t st, const float *next_in_proj_w, const float *next_in_proj_b, float *next_qkv) {
    aft_layer_weights one = w;
    const bool qkv = next_in_proj_w && next_in_proj_b && next_qkv;
    if (qkv) one.in_proj_w = next_in_proj_w;   // the image's in-projection slot carries the NEXT layer's matrix
    aft_config c1 = c;
    c1.precision = AFT_PRECISION_F32;
    hipError_t e = launch_pack_weights(c1, &one, packed, 1, st);
    if (e != hipSuccess) return e;
    const size_t dd = (size_t)c.model_dim * c.model_dim;
    ChainTrainArgs a{};
    a.attn = attn; a.x = x;
    a.wo = packed + 3 * dd; a.w1 = packed + 4 * dd; a.w2 = packed + 6 * dd;
    a.bo = w.out_proj_b; a.b1 = w.lin1_b; a.b2 = w.lin2_b;
    a.g1 = w.norm1_w; a.be1 = w.norm1_b; a.g2 = w.norm2_w; a.be2 = w.norm2_b;
    a.s1 = s1; a.st1 = st1; a.x1 = x1; a.a = a_pre; a.hd = hd; a.s2 = s2; a.st2 = st2; a.x_out = x_out;
    a.rows = rows;
    a.seed1 = seed1; a.seed2 = seed2; a.seed3 = seed3; a.threshold = threshold; a.keep_scale = keep_scale;
    if (qkv) {
        a.wqkv = packed; a.bqkv = next_in_proj_b; a.qkv_next = next_qkv;
        return c.activation == AFT_ACT_GELU ? launch_chain_fwd_train_t<AFT_ACT_GELU, true>(a, st) : launch_chain_fwd_train_t<AFT_ACT_RELU, true>(a, st);
    }
    return c.activation == AFT_ACT_GELU ? launch_chain_fwd_train_t<AFT_ACT_GELU, false>(a, st) : launch_chain_fwd_train_t<AFT_ACT_RELU, false>(a, st);
}

// Fragment-packed TRANSPOSES of one layer's linear2 / linear1 / out_proj weights (the data-gradient products use W, i.e.
// the transposed-product form needs W^T as its "weight" matrix): packed[(((tile * NKB + kb) * 4 + s) * 64 + lane) * 4 + j]
// = M[32 tile + lane % 32][32 kb + 8 s + 4 (lane / 32) + j] with M = W^T, i.e. M[row][k] = W[k][row].
__global__ __launch_bounds__(256) void pack_weights_t_kernel(const float *__restrict__ w2, const float *__restrict__ w1,
                                                             const float *__restrict__ wo, float *__restrict__ packed, int d) {
    const size_t v = (size_t)blockIdx.x * 256 + threadIdx.x;   // one float4 of the packed image
    const size_t dd = (size_t)d * d;
    if (v * 4 >= 5 * dd) return;
    size_t off = v * 4;
    const float *src;
    int K, ld;      // M is [rows_out][K]; W is [K][rows_out] row-major with leading dimension ld = rows_out
    if (off < 2 * dd) { src = w2; K = d; ld = 2 * d; }                       // W2 [d][2d]  -> M = W2^T [2d][d]
    else if ((off -= 2 * dd) < 2 * dd) { src = w1; K = 2 * d; ld = d; }      // W1 [2d][d]  -> M = W1^T [d][2d]
    else { off -= 2 * dd; src = wo; K = d; ld = d; }                         // Wo [d][d]   -> M = Wo^T
    const int lane = (int)(off / 4) % 64, s = (int)(off / 256) % 4;
    const int blk = (int)(off / 1024), nkb = K / 32, kb = blk % nkb, ct = blk / nkb;
    const int row = ct * 32 + (lane & 31), k = kb * 32 + s * 8 + (lane >> 5) * 4;
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = src[(size_t)(k + j) * ld + row];
    *reinterpret_cast<f32x4 *>(packed + v * 4) = o;
}

bool chain_bwd_ok(const aft_config &c, int rows) {
    // d = 128 is instantiated; the five row-major tensors are addressed with 32-bit byte offsets
    return c.model_dim == 128 && (size_t)rows * 2 * c.model_dim * sizeof(float) < ((size_t)1 << 31);
}
size_t chain_bwd_packed_floats(int d) { return (size_t)5 * d * d; }
int chain_bwd_blocks(int rows) { return std::min((rows + 31) / 32, current_device_cus() * 3); }
size_t chain_bwd_lnp_floats(int rows, int d) { return (size_t)((rows + 31) / 32) * 4 * d; }   // upper bound: one slice per tile

template <int ACT>
static hipError_t launch_chain_bwd_t(const ChainBwdArgs &args, hipStream_t st) {
    constexpr int D = 128;
    using B = ChainBwdShape<D>;
    static PerDeviceOnce lds_attr[2];
    const bool ragged = (args.rows & 31) != 0 || switch_on("AFT_CHAIN_BWD_RAGGED");      // a partial last tile: the instantiation with per-row guards (the variable forces it: A/B)
    hipError_t ea = ragged ? ensure_dynamic_lds(lds_attr[1], reinterpret_cast<const void *>(chain_bwd_kernel<D, ACT, true>), B::LDS_BYTES)
                           : ensure_dynamic_lds(lds_attr[0], reinterpret_cast<const void *>(chain_bwd_kernel<D, ACT, false>), B::LDS_BYTES);
    if (ea != hipSuccess) return ea;
    const int blocks = chain_bwd_blocks(args.rows);
    if (ragged) hipLaunchKernelGGL((chain_bwd_kernel<D, ACT, true>), dim3(blocks), dim3(ChainShape<D>::THREADS), B::LDS_BYTES, st, args);
    else hipLaunchKernelGGL((chain_bwd_kernel<D, ACT, false>), dim3(blocks), dim3(ChainShape<D>::THREADS), B::LDS_BYTES, st, args);
    return hipGetLastError();
}

hipError_t launch_chain_bwd(const aft_config &c, const aft_layer_weights &w, const float *g, const float *s2, const float *st2,
                            const float *a_pre, const float *s1, const float *st1, float *packed_t, float *g2, float *gff,
                            float *g2b, float *d_o, float *dx, float *lnp, int rows, uint32_t seed1, uint32_t seed2,
                            uint32_t seed3, uint32_t threshold, float keep_scale, hipStream_t st) {
    const int d = c.model_dim;
    const size_t dd = (size_t)d * d;
    hipLaunchKernelGGL(pack_weights_t_kernel, dim3((unsigned)((5 * dd / 4 + 255) / 256)), dim3(256), 0, st, w.lin2_w, w.lin1_w,
                       w.out_proj_w, packed_t, d);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    ChainBwdArgs a{};
    a.g = g; a.s2 = s2; a.st2 = st2; a.a = a_pre; a.s1 = s1; a.st1 = st1;
    a.w2t = packed_t; a.w1t = packed_t + 2 * dd; a.wot = packed_t + 4 * dd;
    a.gam2 = w.norm2_w; a.gam1 = w.norm1_w;
    a.g2 = g2; a.gff = gff; a.g2b = g2b; a.d_o = d_o; a.dx = dx; a.lnp = lnp;
    a.rows = rows;
    a.seed1 = seed1; a.seed2 = seed2; a.seed3 = seed3; a.threshold = threshold; a.keep_scale = keep_scale;
    return c.activation == AFT_ACT_GELU ? launch_chain_bwd_t<AFT_ACT_GELU>(a, st) : launch_chain_bwd_t<AFT_ACT_RELU>(a, st);
}

}  // namespace aft
