// k_chain.hip -- launchers of the row-local chain kernel (device code: chain_device.h) and the weight re-packing kernel.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "chain_device.h"
#include "pack_device.h"

#ifndef AFT_CHAIN_MANY_MAX_D
#define AFT_CHAIN_MANY_MAX_D 96
#endif

namespace aft {

// MLP / QKV select the three launch variants at compile time (distinct symbols in a profile), see chain_device.h
template <int D, int ACT, bool MLP, bool QKV>
__global__ __launch_bounds__(D * 2, D <= 128 ? 3 : 2) void chain_kernel(const ChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // persistent workgroups: the grid is sized to the co-resident count and each workgroup walks the row tiles with
    // stride gridDim.x (no dispatch gaps, no launch tail)
    chain_body<D, ACT, MLP, QKV>(a, smem, threadIdx.x, blockIdx.x, gridDim.x, (a.rows + 31) / 32);
}


// split-precision tier (aft_config.precision = AFT_PRECISION_BF16X3): the same body with its GEMMs on bf16 hi/lo terms
template <int D, int ACT, bool MLP, bool QKV>
__global__ __launch_bounds__(D * 2, D <= 128 ? 3 : 2) void chain_split_kernel(const ChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    chain_body<D, ACT, MLP, QKV, true>(a, smem, threadIdx.x, blockIdx.x, gridDim.x, (a.rows + 31) / 32);
}

template <int D, int ACT, bool MLP, bool QKV>
static hipError_t launch_chain_split_v(const ChainArgs &args, hipStream_t st) {
    using S = ChainShape<D>;
    static PerDeviceOnce lds_attr;
    hipError_t ea = ensure_dynamic_lds(lds_attr, reinterpret_cast<const void *>(chain_split_kernel<D, ACT, MLP, QKV>), S::LDS_BYTES);
    if (ea != hipSuccess) return ea;
    const int blocks = std::min((args.rows + 31) / 32, current_device_cus() * (D <= 128 ? 3 : 1));
    hipLaunchKernelGGL((chain_split_kernel<D, ACT, MLP, QKV>), dim3(blocks), dim3(S::THREADS), S::LDS_BYTES, st, args);
    return hipGetLastError();
}

template <int D, int ACT>
static hipError_t launch_chain_split_t(const ChainArgs &args, bool mlp, bool qkv, hipStream_t st) {
    if (mlp && qkv) return launch_chain_split_v<D, ACT, true, true>(args, st);
    if (mlp) return launch_chain_split_v<D, ACT, true, false>(args, st);
    return launch_chain_split_v<D, ACT, false, true>(args, st);
}

template <int D, int ACT, bool MLP, bool QKV>
static hipError_t launch_chain_v(const ChainArgs &args, hipStream_t st) {
    using S = ChainShape<D>;
    static PerDeviceOnce lds_attr;   // per instantiation x device
    hipError_t ea = ensure_dynamic_lds(lds_attr, reinterpret_cast<const void *>(chain_kernel<D, ACT, MLP, QKV>), S::LDS_BYTES);
    if (ea != hipSuccess) return ea;
    // co-resident workgroups: CUs x (3 at d = 128 | 1 above), see __launch_bounds__ / LDS.  The one-, two- and three-wave workgroups
    // of d = 32 / 64 / 96 would leave the SIMDs at 0.75 / 1.5 / 2.25 waves with three per CU: as many as give twelve waves per CU and
    // fit the LDS (1 280-byte granules, section 4.0 fact 8) -- round 5: d = 64 154 k -> 166 k frames/s, d = 96 chain 0.60 -> 0.67 of
    // the roof (bench `other_shapes`)
    constexpr int kGranules = (int)((S::LDS_BYTES + 1279) / 1280);
    constexpr int kPerCu = D <= AFT_CHAIN_MANY_MAX_D ? std::min(12 / S::WAVES, 128 / kGranules) : (D <= 128 ? 3 : 1);
    const int resident = current_device_cus() * kPerCu;
    const int blocks = std::min((args.rows + 31) / 32, resident);
    const size_t lds = S::LDS_BYTES;
#ifdef AFT_DIAG_STAMPS
    if (switch_on("AFT_STAMPS")) {   // phase stamps, printed on the host (never in the product build)
        static unsigned long long *dbuf = nullptr;
        if (!dbuf) (void)hipMalloc(&dbuf, sizeof(unsigned long long) * 16 * 4096);
        (void)hipMemset(dbuf, 0, sizeof(unsigned long long) * 16 * 4096);
        ChainArgs a2 = args;
        a2.stamps = dbuf;
        hipLaunchKernelGGL((chain_kernel<D, ACT, MLP, QKV>), dim3(blocks), dim3(S::THREADS), lds, st, a2);
        (void)hipDeviceSynchronize();
        static int printed = 0;
        if (printed++ < 2) {
            int nb = 0;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, chain_kernel<D, ACT, MLP, QKV>, S::THREADS, lds);
            std::vector<unsigned long long> h(16 * 4096);
            (void)hipMemcpy(h.data(), dbuf, h.size() * 8, hipMemcpyDeviceToHost);
            double sum[12] = {0};
            const int nblk = std::min((args.rows + 31) / 32, 4096);
            for (int b = 0; b < nblk; ++b)
                for (int i = 1; i < 12; ++i) {
                    int k = i - 1;
                    while (k > 0 && h[b * 16 + k] == 0) --k;
                    if (h[b * 16 + i]) sum[i] += (double)(h[b * 16 + i] - h[b * 16 + k]);
                }
            printf("chain<%d,mlp=%d,qkv=%d> %d blocks/CU at %zu B LDS; mean cycles per phase:", D, (int)MLP, (int)QKV, nb, lds);
            for (int i = 1; i < 12; ++i) printf(" [%d]=%.0f", i, sum[i] / nblk);
            printf("\n");
            // residency: busy tile-time per (xcc, hw cu id) / kernel span (memrealtime = 100 MHz)
            unsigned long long t0 = ~0ull, t1 = 0;
            std::vector<double> busy(8 * 4096, 0.0);
            std::vector<int> cnt(8 * 4096, 0);
            for (int b = 0; b < nblk; ++b) {
                const unsigned long long s0 = h[b * 16 + 12], s1 = h[b * 16 + 13], id = h[b * 16 + 14];
                t0 = std::min(t0, s0); t1 = std::max(t1, s1);
                const unsigned hw = (unsigned)(id >> 32), xcc = (unsigned)id & 7;
                const unsigned cu = ((hw >> 8) & 0xf) | (((hw >> 13) & 0x7) << 4) | (((hw >> 12) & 1) << 7);   // cu_id, se_id, sh_id
                busy[xcc * 4096 + cu] += (double)(s1 - s0);
                cnt[xcc * 4096 + cu]++;
            }
            int ncu = 0, mn = 1 << 30, mx = 0; double tot = 0;
            for (size_t i = 0; i < busy.size(); ++i) if (cnt[i]) { ++ncu; tot += busy[i]; mn = std::min(mn, cnt[i]); mx = std::max(mx, cnt[i]); }
            printf("  span %.1f us; %d distinct CUs; tiles/CU min %d max %d; mean concurrent workgroups per CU %.2f; mean tile time %.1f us\n",
                   (t1 - t0) / 100.0, ncu, mn, mx, tot / ncu / (double)(t1 - t0), tot / nblk / 100.0);
            {   // is the spread of finish times WITHIN a CU (arbitration between its workgroups) or ACROSS CUs / XCDs?
                std::vector<double> first(8 * 4096, 1e30), last(8 * 4096, 0.0);
                double xlast[8] = {0}; int xn[8] = {0};
                for (int b = 0; b < nblk; ++b) {
                    const unsigned long long id = h[b * 16 + 14];
                    const unsigned hw = (unsigned)(id >> 32), xcc = (unsigned)id & 7;
                    const unsigned cu = ((hw >> 8) & 0xf) | (((hw >> 13) & 0x7) << 4) | (((hw >> 12) & 1) << 7);
                    const double en = (h[b * 16 + 13] - t0) / 100.0;
                    if (b + blocks >= nblk || b + 2 * blocks >= nblk) {   // a workgroup's last tile
                        if (b + blocks >= nblk) { first[xcc * 4096 + cu] = std::min(first[xcc * 4096 + cu], en); last[xcc * 4096 + cu] = std::max(last[xcc * 4096 + cu], en); }
                    }
                }
                double fmin = 1e30, fmax = 0, fsum = 0, lmin = 1e30, lmax = 0, lsum = 0; int n = 0;
                for (size_t i = 0; i < last.size(); ++i) if (last[i] > 0) {
                    ++n; fmin = std::min(fmin, first[i]); fmax = std::max(fmax, first[i]); fsum += first[i];
                    lmin = std::min(lmin, last[i]); lmax = std::max(lmax, last[i]); lsum += last[i];
                    xlast[i / 4096] += last[i]; xn[i / 4096]++;
                }
                printf("  per CU, last-round tiles: first workgroup done %.1f / %.1f / %.1f us, last workgroup done %.1f / %.1f / %.1f us (min/mean/max over %d CUs)\n",
                       fmin, fsum / n, fmax, lmin, lsum / n, lmax, n);
                printf("  mean finish per XCD:");
                for (int x = 0; x < 8; ++x) printf(" %.1f", xn[x] ? xlast[x] / xn[x] : 0.0);
                printf("\n");
            }
            for (int r0 = 0; r0 < nblk; r0 += blocks) {   // persistent rounds: when do their tiles start / end
                double smin = 1e30, smax = 0, ssum = 0, emin = 1e30, emax = 0, esum = 0;
                const int r1 = std::min(nblk, r0 + blocks);
                for (int b = r0; b < r1; ++b) {
                    const double st = (h[b * 16 + 12] - t0) / 100.0, en = (h[b * 16 + 13] - t0) / 100.0;
                    smin = std::min(smin, st); smax = std::max(smax, st); ssum += st;
                    emin = std::min(emin, en); emax = std::max(emax, en); esum += en;
                }
                printf("  round %d (%d tiles): start %.1f / %.1f / %.1f us (min/mean/max), end %.1f / %.1f / %.1f us\n", r0 / blocks, r1 - r0,
                       smin, ssum / (r1 - r0), smax, emin, esum / (r1 - r0), emax);
            }
        }
        return hipGetLastError();
    }
#endif
    hipLaunchKernelGGL((chain_kernel<D, ACT, MLP, QKV>), dim3(blocks), dim3(S::THREADS), lds, st, args);
    return hipGetLastError();
}

// Packed-weight block of one layer inside the workspace (floats): [in_proj 3D*D][out_proj D*D][lin1 2D*D][lin2 D*2D]
size_t packed_layer_floats(int d) { return (size_t)8 * d * d; }

__global__ __launch_bounds__(256) void pack_weights_kernel(const WeightsDev w, float *__restrict__ packed, int d,
                                                           int num_layers, int split) {
    pack_weights_vec(w, packed, d, 0, num_layers, split, (size_t)blockIdx.x * 256 + threadIdx.x);   // one float4 of the image
}

hipError_t launch_pack_weights(const aft_config &c, const aft_layer_weights *layers, float *packed, int count, hipStream_t st) {
    for (int first = 0; first < count; first += kLayerWindow) {   // the kernel takes a window of the layer table by value
        const int n = std::min(kLayerWindow, count - first);
        WeightsDev win{};
        for (int i = 0; i < n; ++i) win.layers[i] = layers[first + i];
        const size_t vecs = packed_layer_floats(c.model_dim) * n / 4;
        hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)((vecs + 255) / 256)), dim3(256), 0, st, win,
                           packed + packed_layer_floats(c.model_dim) * first, c.model_dim, n, c.precision == AFT_PRECISION_BF16X3 ? 1 : 0);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

template <int D, int ACT>
static hipError_t launch_chain_t(const ChainArgs &args, bool mlp, bool qkv, hipStream_t st) {
    if (mlp && qkv) return launch_chain_v<D, ACT, true, true>(args, st);
    if (mlp) return launch_chain_v<D, ACT, true, false>(args, st);
    return launch_chain_v<D, ACT, false, true>(args, st);
}

hipError_t launch_chain(const aft_config &c, const aft_layer_weights *m, const float *m_packed,
                        const aft_layer_weights *qw, const float *q_packed,
                        const float *attn, float *x, float *q, float *k, float *vt, int rows, int tokens,
                        int tokpad, hipStream_t st, const ChainFusion *fuse) {
    const size_t dd = (size_t)c.model_dim * c.model_dim;
    ChainArgs a{};
    if (fuse != nullptr && fuse->out6 != nullptr && m != nullptr && qw == nullptr) {
        a.lin2_w = fuse->lin2_w; a.lin2_b = fuse->lin2_b; a.out6 = fuse->out6;
        a.out6_features = c.patch_scs * c.patch_symbols;
        a.out6_stride = out6_stride(c);
    }
    if (fuse != nullptr && fuse->conv_enhanced != nullptr && m == nullptr) {
        a.emb_conv = fuse->conv_enhanced; a.emb_tok6 = fuse->tokens6; a.emb_w1 = fuse->lin1_w; a.emb_b1 = fuse->lin1_b;
        a.emb_pos = fuse->pos;
        a.emb_S = c.num_scs; a.emb_T = c.num_symbols; a.emb_p0 = c.patch_scs; a.emb_p1 = c.patch_symbols;
        a.emb_K = c.patch_scs * c.patch_symbols + (fuse->tokens6 ? 6 : 0);
    }
    a.attn = attn;
    a.x = x;
    a.x_blocked = fuse != nullptr && fuse->x_blocked ? 1 : 0;
    if (m != nullptr) {
        a.wo = m_packed + 3 * dd; a.bo = m->out_proj_b;
        a.w1 = m_packed + 4 * dd; a.b1 = m->lin1_b;
        a.w2 = m_packed + 6 * dd; a.b2 = m->lin2_b;
        a.g1 = m->norm1_w; a.be1 = m->norm1_b;
        a.g2 = m->norm2_w; a.be2 = m->norm2_b;
    }
    if (qw != nullptr) {
        a.wqkv = q_packed;
        a.bv = qw->in_proj_b + 2 * c.model_dim;
    }
    a.q = q; a.k = k; a.vt = vt;
    a.rows = rows; a.tokens = tokens; a.tokpad = tokpad;
    a.heads = c.model_dim / kHeadDim;   // q / k / v^T are laid out per 32-feature block, whatever the head count (attn_device.h)
    const bool gelu = c.activation == AFT_ACT_GELU;
    const bool mlp = m != nullptr, qkv = qw != nullptr;
    if (c.precision == AFT_PRECISION_BF16X3) {
        if (c.model_dim == 128)
            return gelu ? launch_chain_split_t<128, AFT_ACT_GELU>(a, mlp, qkv, st) : launch_chain_split_t<128, AFT_ACT_RELU>(a, mlp, qkv, st);
        if (c.model_dim == 256)
            return gelu ? launch_chain_split_t<256, AFT_ACT_GELU>(a, mlp, qkv, st) : launch_chain_split_t<256, AFT_ACT_RELU>(a, mlp, qkv, st);
        return hipErrorInvalidValue;    // refused earlier by check_config
    }
    // every multiple of 32 up to 256 (round 5: the body is generic in D = 32 x waves; 128 and 256 are the tuned shapes, the others are
    // covered -- odd wave counts leave SIMDs unevenly filled)
#define AFT_CHAIN_DIM(DD) \
    if (c.model_dim == DD) return gelu ? launch_chain_t<DD, AFT_ACT_GELU>(a, mlp, qkv, st) : launch_chain_t<DD, AFT_ACT_RELU>(a, mlp, qkv, st);
    AFT_CHAIN_DIM(32) AFT_CHAIN_DIM(96) AFT_CHAIN_DIM(160) AFT_CHAIN_DIM(224)
#undef AFT_CHAIN_DIM
    if (c.model_dim == 64)
        return gelu ? launch_chain_t<64, AFT_ACT_GELU>(a, mlp, qkv, st) : launch_chain_t<64, AFT_ACT_RELU>(a, mlp, qkv, st);
    if (c.model_dim == 192)
        return gelu ? launch_chain_t<192, AFT_ACT_GELU>(a, mlp, qkv, st) : launch_chain_t<192, AFT_ACT_RELU>(a, mlp, qkv, st);
    if (c.model_dim == 128)
        return gelu ? launch_chain_t<128, AFT_ACT_GELU>(a, mlp, qkv, st) : launch_chain_t<128, AFT_ACT_RELU>(a, mlp, qkv, st);
    if (c.model_dim == 256)
        return gelu ? launch_chain_t<256, AFT_ACT_GELU>(a, mlp, qkv, st) : launch_chain_t<256, AFT_ACT_RELU>(a, mlp, qkv, st);
    return hipErrorInvalidValue;
}

}  // namespace aft
