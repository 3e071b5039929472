// aft_api.hip -- the C ABI of include/adafortitran_amd.h: argument checking, workspace plan and
// the launch sequence of one forward.  No allocation, no synchronisation, no per-call state: besides a
// thread-local error string the only things cached are per-device facts (CU count, LDS function
// attribute) in tables indexed by device ordinal (hipGraph-capturable, re-entrant per stream).
#include <algorithm>
#include <cstdlib>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

extern char **environ;

#include "aft_internal.h"

namespace aft {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = 0;
    return dev;
}

int current_device_cus() {
    static int cus[kMaxDevices];   // 0 = not asked yet; per device, idempotent
    const int dev = current_device();
    int c = __atomic_load_n(&cus[dev], __ATOMIC_RELAXED);
    if (c == 0) {
        if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) c = 256;
        __atomic_store_n(&cus[dev], c, __ATOMIC_RELAXED);
    }
    return c;
}

hipError_t ensure_dynamic_lds(PerDeviceOnce &once, const void *kernel, size_t bytes) {
    const int dev = current_device();
    if (!once.needed(dev)) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) once.mark(dev);
    return e;
}

// ---- switches: the "AFT_*" environment, read once at load; aft_set_switch afterwards (header) ----
namespace {
struct Switches {
    std::mutex mu;
    std::map<std::string, std::string> table;
    Switches() {      // static initialisation of the shared object: before any call, on the loading thread
        for (char **e = environ; e != nullptr && *e != nullptr; ++e) {
            if (strncmp(*e, "AFT_", 4) != 0) continue;
            const char *eq = strchr(*e, '=');
            if (eq != nullptr) table[std::string(*e, eq - *e)] = std::string(eq + 1);
        }
    }
};
Switches &switches() {
    static Switches s;
    return s;
}
struct SwitchesAtLoad { SwitchesAtLoad() { switches(); } } g_switches_at_load;
}  // namespace
bool switch_on(const char *name) {
    Switches &s = switches();
    std::lock_guard<std::mutex> lock(s.mu);
    return s.table.find(name) != s.table.end();
}
int switch_int(const char *name, int dflt) {
    Switches &s = switches();
    std::lock_guard<std::mutex> lock(s.mu);
    auto it = s.table.find(name);
    return it == s.table.end() ? dflt : atoi(it->second.c_str());
}

WeightsDev weights_window(const aft_weights &w, int first, int count) {
    WeightsDev d{};
    d.up_w = w.up_w; d.up_b = w.up_b;
    for (int i = 0; i < 4; ++i) { d.enh_w[i] = w.enh_w[i]; d.enh_b[i] = w.enh_b[i]; d.ref_w[i] = w.ref_w[i]; d.ref_b[i] = w.ref_b[i]; }
    for (int e = 0; e < 3; ++e)
        for (int j = 0; j < 3; ++j) { d.ada_w[e][j] = w.ada_w[e][j]; d.ada_b[e][j] = w.ada_b[e][j]; }
    d.lin1_w = w.lin1_w; d.lin1_b = w.lin1_b; d.pos = w.pos; d.lin2_w = w.lin2_w; d.lin2_b = w.lin2_b;
    for (int i = 0; i < count && i < kLayerWindow && w.layers != nullptr; ++i) d.layers[i] = w.layers[first + i];
    return d;
}

static int tokens_of(const aft_config &c) { return (c.num_scs / c.patch_scs) * (c.num_symbols / c.patch_symbols); }

bool packed_engine_ok(const aft_config &c) {
    if (c.num_head <= 0 || c.model_dim <= 0 || c.model_dim % c.num_head != 0) return false;
    const int hd = c.model_dim / c.num_head;
    return c.model_dim >= 32 && c.model_dim <= 256 && c.model_dim % 32 == 0 && hd >= 8 && hd <= 64 && hd % 8 == 0 && hd != 56 &&
           c.patch_scs * c.patch_symbols <= kMaxPatchFeatures;
}

int check_config(const aft_config *c) {
    if (c == nullptr) {
        set_error("aft_config is NULL");
        return AFT_ERR_ARG;
    }
    if (c->num_scs <= 0 || c->num_symbols <= 0 || c->patch_scs <= 0 || c->patch_symbols <= 0 ||
        c->num_scs % c->patch_scs || c->num_symbols % c->patch_symbols) {
        set_error("OFDM grid %dx%d is not divisible by patch %dx%d", c->num_scs, c->num_symbols, c->patch_scs,
                  c->patch_symbols);
        return AFT_ERR_SHAPE;
    }
    if (c->pilot_scs <= 0 || c->pilot_symbols <= 0 || c->num_layers <= 0) {     // any layer count (encoders.py:52-55)
        set_error("bad pilot grid %dx%d or layer count %d", c->pilot_scs, c->pilot_symbols, c->num_layers);
        return AFT_ERR_SHAPE;
    }
    // Two engines (aft_engine_of).  The PACKED one: the row-local chain kernel gives every wave one 32-feature block of every
    // activation and keeps a 32-row tile's whole hidden layer (2 x model_dim wide) in the registers of its model_dim / 32 waves
    // (251 VGPRs at 256); q / k / v^T live in 32-feature blocks whose fragment slots hold 8 features, a head may straddle at most
    // two blocks; the fused embedding stages linear_1's (patch + 6) x model_dim weights in the chain kernel's idle LDS and
    // linear_2's outputs ride in one 16-column MFMA tile: model_dim a multiple of 32 up to 256, head dims that are multiples of 8 up
    // to 64 except 56, patches of up to 16 elements.  The GENERAL one (round 6): row-major GEMM / attention / LayerNorm launches for
    // everything else the reference's schema (> 0, schemas.py:124-127) and nn.TransformerEncoderLayer (encoders.py:44-55) build
    // within: model_dim a multiple of 8 up to 512, any num_head that divides it with heads of up to 128 features, patches of up to
    // 32 elements.  What is still refused, and why:
    if (c->model_dim < 8 || c->model_dim > 512 || c->model_dim % 8 != 0) {
        set_error("model_dim %d not covered by the gfx950 kernels (multiples of 8 up to 512: the row-wise LayerNorm / activation kernels "
                  "hold a row in at most 8 floats per lane and move it in 16-byte pieces)", c->model_dim);
        return AFT_ERR_SHAPE;
    }
    // nn.MultiheadAttention takes any num_head that divides model_dim (reference blocks/encoders.py:44-51, schemas.py:124-127)
    const int hd = c->num_head > 0 && c->model_dim % c->num_head == 0 ? c->model_dim / c->num_head : 0;
    if (hd < 1 || hd > 128) {
        set_error("head dim not covered (model_dim=%d, num_head=%d): num_head must divide model_dim (nn.MultiheadAttention demands it) "
                  "and a head may have at most 128 features (four 32-feature accumulator blocks per query tile)", c->model_dim, c->num_head);
        return AFT_ERR_SHAPE;
    }
    if (tokens_of(*c) < 1) {
        set_error("no tokens");
        return AFT_ERR_SHAPE;
    }
    if (c->patch_scs * c->patch_symbols > kMaxPatchGeneral) {
        set_error("patch %dx%d has more than %d elements (the embedding kernels hold a token's patch in registers; the reference's "
                  "default is 3x2)", c->patch_scs, c->patch_symbols, kMaxPatchGeneral);
        return AFT_ERR_SHAPE;
    }
    if (c->precision != AFT_PRECISION_F32 && (c->precision != AFT_PRECISION_BF16X3 || (c->model_dim != 128 && c->model_dim != 256) ||
                                              hd != kHeadDim || tokens_of(*c) < kTile || !packed_engine_ok(*c))) {
        set_error("precision %d: the split-precision tier (AFT_PRECISION_BF16X3) is instantiated for model_dim 128 and 256, head dim 32, "
                  ">= 32 tokens", c->precision);
        return AFT_ERR_SHAPE;
    }
    if (c->activation != AFT_ACT_RELU && c->activation != AFT_ACT_GELU) {
        set_error("unknown activation %d", c->activation);
        return AFT_ERR_ARG;
    }
    {
        const int p = c->patch_scs * c->patch_symbols;      // (the general engine's tail reads linear_2's output: nothing beside the plane)
        if (!conv_plan_ok(c->num_scs, c->num_symbols, c->pilot_scs * c->pilot_symbols) ||
            !conv_plan_ok(c->num_scs, c->num_symbols, packed_engine_ok(*c) ? p * c->model_dim + p : 0)) {
            set_error("OFDM grid %dx%d: no LDS band plan for the fused conv stack (too many symbols per row band)",
                      c->num_scs, c->num_symbols);
            return AFT_ERR_SHAPE;
        }
    }
    if (c->adaptive && (c->hidden[0] <= 0 || c->hidden[1] <= 0 || c->hidden[2] != 2 * tokens_of(*c))) {
        set_error("channel_adaptivity_hidden_sizes [%d,%d,%d]: last must be 2 x tokens (%d)", c->hidden[0],
                  c->hidden[1], c->hidden[2], 2 * tokens_of(*c));
        return AFT_ERR_SHAPE;
    }
    return AFT_OK;
}

static size_t align64(size_t floats) { return (floats + 63) / 64 * 64; }

Workspace plan_workspace(const aft_config &c, int batch) {
    Workspace ws{};
    ws.tokens = tokens_of(c);
    ws.tokpad = round_up(ws.tokens, kTile);
    ws.planes = 2 * batch;
    const size_t rows = (size_t)ws.planes * ws.tokens;
    size_t off = 0;
    ws.conv_enhanced = off; off += align64((size_t)ws.planes * c.num_scs * c.num_symbols);
    ws.tokens6 = off;       off += align64((size_t)batch * ws.tokens * 6);
    if (!packed_engine_ok(c)) {   // general engine: row-major tensors of one layer at a time (run_encoder_general)
        const size_t d = c.model_dim, ff = 2 * d;
        ws.x = off;        off += align64(std::max(rows * d, (size_t)ws.planes * c.num_scs * c.num_symbols));   // also the upsampler's plane scratch
        ws.attn = off;     off += align64(rows * d);
        ws.q = ws.k = ws.vt = ws.wpack = off;
        ws.out6 = off;     off += align64(rows * out6_stride(c));
        ws.convfrag = off; off += align64(2 * kConvFragFloats);
        ws.g_x1 = off;     off += align64(rows * d);
        ws.g_y = off;      off += align64(rows * d);
        ws.g_s = off;      off += align64(rows * d);
        ws.g_stats = off;  off += align64(rows * 2);
        ws.g_qkv = off;    off += align64(rows * 3 * d);
        ws.g_lse = off;    off += align64(rows * c.num_head);
        ws.g_a = off;      off += align64(rows * ff);
        ws.g_hd = off;     off += align64(rows * ff);
        ws.g_pad = off;    off += align64(attn_train_pad_floats(c, rows));
        ws.total_floats = off;
        return ws;
    }
    ws.x = off;             off += align64((rows + 31) / 32 * 32 * c.model_dim);   // whole 32-row tiles (tile-blocked x)
    // attention tiles: global 32-row tiles (layer-by-layer path) or ceil(tokens/32) tiles per plane (plane-resident path)
    ws.attn = off;          off += align64(std::max((size_t)round_up((int)rows, kTile), (size_t)ws.planes * ws.tokpad) * c.model_dim);
    const size_t per_head = (size_t)ws.planes * ws.tokpad * c.model_dim;   // planes x (model_dim / 32) blocks x tokpad x 32
    ws.q = off;             off += align64(per_head);
    ws.k = off;             off += align64(per_head);
    ws.vt = off;            off += align64(per_head);
    ws.wpack = off;         off += align64(packed_layer_floats(c.model_dim) * c.num_layers);
    ws.out6 = off;          off += align64(rows * out6_stride(c));
    ws.convfrag = off;      off += align64(2 * kConvFragFloats);
    ws.total_floats = off;
    return ws;
}

// The kernels address every workspace region through a buffer resource with 32-bit byte offsets (srd.h: num_records
// 0x7fffffff; a load past it silently returns 0).  The largest region of a forward is a q / k / v^T block or the
// attention tiles: planes x tokpad x model_dim floats (tokpad >= tokens, so x is never larger).
static size_t largest_region_bytes(const aft_config &c, int batch) {
    const Workspace ws = plan_workspace(c, batch);
    const size_t rows = (size_t)ws.planes * ws.tokens;
    const size_t per_head = (size_t)ws.planes * ws.tokpad * c.model_dim;
    size_t biggest = std::max({rows * (size_t)c.model_dim, per_head, (size_t)ws.planes * c.num_scs * c.num_symbols,
                               rows * (size_t)out6_stride(c)});
    if (!packed_engine_ok(c)) {   // general engine: q | k | v with padded heads is its largest tensor
        const size_t hp = (size_t)(c.model_dim / c.num_head + 31) / 32 * 32;
        biggest = std::max(biggest, rows * 3 * std::max((size_t)c.model_dim, hp * c.num_head));
    }
    return biggest * sizeof(float);
}

static int max_batch_of(const aft_config &c) {
    const size_t per_frame = largest_region_bytes(c, 1);   // every region is linear in the batch
    return (int)std::min<size_t>((((size_t)1 << 31) - 1) / per_frame, (size_t)1 << 24);
}

static int hip_fail(const char *what, hipError_t e) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return AFT_ERR_HIP;
}

// ---- lanes (late round 5) ----
// A forward whose launches are fewer than ~2.5 rounds of the persistent grids leaves CUs idle at the end of every launch (whole row
// tiles / attention tasks over 256 CUs, section 5 of DESIGN.md).  Frames are independent, so such a forward is run as TWO (or more)
// forwards over contiguous shares of the batch: share 0 on the caller's stream, the others on library-owned side streams forked from
// and joined back into it with events -- the hardware then fills the tail of one share's launch with the other share's next one.
// Each share is a complete forward with its own slice of the workspace (laid end to end, share 0 first); per-frame arithmetic does
// not depend on the batch a frame travels in, so the bits are those of the single forward (tests/test_hip_parity.py).
constexpr int kMaxLanes = 4;
struct LanePlan {
    int lanes;
    int frames[kMaxLanes], first[kMaxLanes];
    size_t ws_off[kMaxLanes];     // floats
    size_t total_floats;
};
// When it pays (tools/debug/lanes_threshold.py and the tables of DESIGN.md section 5: one lane against two over eight configurations
// x batches, four boxes):
//  * at most one row tile per CU: every launch already sits at its latency floor and a second lane only adds launches (0.73-0.83 x);
//    model_dim below 128 (two- / three-wave chain workgroups, short launches): 0.93-1.02 x.  One lane.
//  * above ~15 row tiles per CU (4 000 at 256 CUs: 256 frames of the default model, 64 of config 5) the tails are a small share of
//    every launch and two lanes cost 0-3 %.  One lane.
//  * in between two lanes run at a flat ~0.98 of the best per-frame rate whatever the batch, while one lane swings between 0.91 and
//    1.00 with how well the batch fills whole rounds of the three persistent grids (default model: 128 frames 1.00, 129 0.94,
//    132 0.91, 120 0.95, 192 0.96).  So: one lane where the launches of ONE lane are nearly whole rounds -- the round efficiency
//    n / (ceil(n / slots) slots) of the chain's row tiles, the attention tasks and the conv planes, weighted by their share of a
//    forward (0.55 / 0.35 / 0.10), at least 0.97: 127 / 128 frames of the default model (0.985 at 128: the benchmark's batch keeps
//    one launch sequence and its per-kernel accounting) -- else two (16 frames 1.12 x, 32 1.10, 64 1.03-1.06, 96 1.02-1.04,
//    129 1.04, 132 1.07, 160 1.01, 192 1.03; model_dim 256: 16 frames 1.32 x, 64 1.05-1.08; config 5: 4 / 8 / 16 / 48 frames
//    1.27 / 1.15 / 1.05 / 1.03; other head counts and model dims at 64 / 128 frames 0.99-1.10).
static double round_efficiency(long n, long slots) {
    const long rounds = (n + slots - 1) / std::max<long>(slots, 1);
    return rounds > 0 ? (double)n / (double)(rounds * slots) : 1.0;
}
static int lanes_wanted(const aft_config &c, int batch) {
    if (!packed_engine_ok(c)) return 1;         // the general engine's launches are dispatcher-scheduled grids, not persistent rounds
    if (switch_on("AFT_LANES")) {               // A/B switch: 1 = never split, 2 .. 4 = always that many shares
        const int v = switch_int("AFT_LANES", 0);
        if (v >= 1 && v <= kMaxLanes) return std::min(v, batch);
    }
    const long tokens = tokens_of(c), planes = 2L * batch, cus = current_device_cus();
    const long tiles = (planes * tokens + 31) / 32;
    if (batch < 2 || c.model_dim < 128 || tiles <= cus || tiles * 256 > 4000 * cus) return 1;
    const int hd = c.model_dim / c.num_head;
    const long chain_slots = cus * (c.model_dim <= 128 ? 3 : 1);                                  // k_chain.hip: workgroups per CU
    const long attn_slots = cus * 4 * ((hd == 64 || hd == 24 || hd == 40 || hd == 48) ? 2 : 3);   // k_attn.hip: waves per SIMD
    const long attn_tasks = planes * c.num_head * ((tokens + 31) / 32);
    const double eff = 0.55 * round_efficiency(tiles, chain_slots) + 0.35 * round_efficiency(attn_tasks, attn_slots) +
                       0.10 * round_efficiency(planes, cus);
    return eff >= 0.97 ? 1 : 2;
}
static LanePlan plan_lanes(const aft_config &c, int batch, int lanes) {
    LanePlan p{};
    p.lanes = std::max(1, std::min(lanes, std::min(batch, kMaxLanes)));
    size_t off = 0;
    for (int i = 0; i < p.lanes; ++i) {
        p.first[i] = (int)((long)batch * i / p.lanes);
        p.frames[i] = (int)((long)batch * (i + 1) / p.lanes) - p.first[i];
        p.ws_off[i] = off;
        off += plan_workspace(c, p.frames[i]).total_floats;
    }
    p.total_floats = off;
    return p;
}
static size_t workspace_floats_any_lanes(const aft_config &c, int batch) {   // whatever lanes_wanted() answers at call time fits
    size_t m = 0;
    for (int l = 1; l <= kMaxLanes; ++l) m = std::max(m, plan_lanes(c, batch, l).total_floats);
    return m;
}
// Side streams + fork / join events of one caller stream (created on first use, up to the lane count a call asks for; nothing here
// describes a call).  The kLaneCacheMax most recently used (device, caller stream) pairs keep theirs; a further stream evicts the
// least recently used entry that no call is using right now (its streams and events are destroyed: the runtime releases them once
// their pending work has drained).  When no entry can be had -- a create call failed (what was created is destroyed again) or every
// entry is in use by calls in flight on other host threads -- the caller runs the shares one after the other on the caller's stream.
struct LaneStreams {
    hipStream_t side[kMaxLanes - 1];
    hipEvent_t fork, join[kMaxLanes - 1];
    int n_side;
};
namespace {
constexpr int kLaneCacheMax = 16;
struct LaneEntry { bool used; int dev; hipStream_t user; LaneStreams ls; int in_use; unsigned long long tick; };
std::mutex g_lane_mu;
LaneEntry g_lane_cache[kLaneCacheMax];      // fixed storage: a leased entry never moves
unsigned long long g_lane_tick = 0;
void destroy_lane_streams(LaneStreams &ls) {
    for (int i = 0; i < ls.n_side; ++i) {
        (void)hipEventDestroy(ls.join[i]);
        (void)hipStreamDestroy(ls.side[i]);
    }
    if (ls.fork) (void)hipEventDestroy(ls.fork);
    ls = LaneStreams{};
}
bool grow_lane_streams(LaneStreams &ls, int sides) {     // create side streams / join events up to `sides`; false: nothing changed
    if (ls.fork == nullptr && hipEventCreateWithFlags(&ls.fork, hipEventDisableTiming) != hipSuccess) { ls.fork = nullptr; return false; }
    while (ls.n_side < sides) {
        hipStream_t s = nullptr;
        hipEvent_t e = nullptr;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return false;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { (void)hipStreamDestroy(s); return false; }
        ls.side[ls.n_side] = s;
        ls.join[ls.n_side] = e;
        ++ls.n_side;
    }
    return true;
}
}  // namespace
static LaneStreams *acquire_lane_streams(hipStream_t user, int lanes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(g_lane_mu);
    LaneEntry *hit = nullptr, *victim = nullptr;
    for (auto &en : g_lane_cache) {
        if (en.used && en.dev == dev && en.user == user) { hit = &en; break; }
        if (!en.used) { if (victim == nullptr || victim->used) victim = &en; }
        else if (en.in_use == 0 && (victim == nullptr || (victim->used && en.tick < victim->tick))) victim = &en;
    }
    if (hit == nullptr) {
        if (victim == nullptr) return nullptr;                 // sixteen entries, all leased to calls in flight
        if (victim->used) destroy_lane_streams(victim->ls);    // least recently used idle entry
        *victim = LaneEntry{true, dev, user, LaneStreams{}, 0, 0};
        hit = victim;
    }
    if (!grow_lane_streams(hit->ls, lanes - 1)) {
        if (hit->in_use == 0) {                                 // leave nothing half-built behind (a later call tries again)
            destroy_lane_streams(hit->ls);
            hit->used = false;
        }
        return nullptr;
    }
    ++hit->in_use;
    hit->tick = ++g_lane_tick;
    return &hit->ls;
}
static void release_lane_streams(LaneStreams *ls) {
    if (ls == nullptr) return;
    std::lock_guard<std::mutex> lock(g_lane_mu);
    for (auto &en : g_lane_cache)
        if (&en.ls == ls) { --en.in_use; return; }
}

// `fused` (whole forward only): the first launch computes x0 from conv_enhanced / tokens6 itself (no embed kernel) and
// the last one leaves linear_2's output in the q buffer instead of storing x (the conv tail reads it from there).
// `prepacked`: the caller's fragment-packed image of ALL layers (aft_pack_weights_f32), or NULL = pack into the workspace now.
// `w`: the non-layer pointers (+ a window of the first layers for the kernels that take the table by value); `layers`: the HOST array
// of all num_layers layers.
static int run_encoder(const aft_config &c, const WeightsDev &w, const aft_layer_weights *layers, const Workspace &ws, float *base,
                       int first_layer, int last_layer, hipStream_t st, bool fused = false, const float *prepacked = nullptr) {
    float *x = base + ws.x, *attn = base + ws.attn, *q = base + ws.q, *k = base + ws.k, *vt = base + ws.vt;
    const int rows = ws.planes * ws.tokens;
    const size_t pl = packed_layer_floats(c.model_dim);
    const float *wp = prepacked != nullptr ? prepacked : base + ws.wpack;
    hipError_t e;
    if (prepacked == nullptr) {
        // weights arrive in torch layout on every call (stateless ABI): re-lay them into fragment order
        e = launch_pack_weights(c, layers + first_layer, base + ws.wpack + first_layer * pl, last_layer - first_layer + 1, st);
        if (e != hipSuccess) return hip_fail("pack_weights", e);
    }
    // whole forward and the caller asks for it: ONE launch for the encoder (k_encoder.hip).  AUTO means the launches:
    // measured on the MI355X at B = 128 (256 planes on 256 CUs, its best case) the plane-resident kernel is 1.5 % slower
    // (profiles/r03_ab_encoder.json), so nothing selects it by itself.
    if (fused && c.encoder_path == AFT_ENCODER_PLANE && c.precision == AFT_PRECISION_F32 && first_layer == 0 &&
        last_layer == c.num_layers - 1 && encoder_plane_ok(c)) {
        e = launch_encoder_plane(c, w, wp, base + ws.conv_enhanced, c.adaptive ? base + ws.tokens6 : nullptr, x, attn, q, k, vt,
                                 base + ws.out6, ws.planes, ws.tokens, ws.tokpad, st);
        return e == hipSuccess ? AFT_OK : hip_fail("encoder(plane-resident)", e);
    }
    // in-projection of the first layer (QKV-only pass of the chain kernel)
    ChainFusion first{}, last{}, middle{};
    // whole forward: nobody but these launches reads x, so it crosses HBM in tile-blocked order (every access 1 KB contiguous)
    first.x_blocked = last.x_blocked = middle.x_blocked = fused;
    if (fused) {
        first.conv_enhanced = base + ws.conv_enhanced;
        first.tokens6 = c.adaptive ? base + ws.tokens6 : nullptr;
        first.lin1_w = w.lin1_w; first.lin1_b = w.lin1_b; first.pos = w.pos;
        last.lin2_w = w.lin2_w; last.lin2_b = w.lin2_b; last.out6 = base + ws.out6;
    }
    e = launch_chain(c, nullptr, nullptr, &layers[first_layer], wp + first_layer * pl, nullptr, x, q, k, vt, rows,
                     ws.tokens, ws.tokpad, st, fused ? &first : nullptr);
    if (e != hipSuccess) return hip_fail("chain(qkv)", e);
    for (int l = first_layer; l <= last_layer; ++l) {
        e = launch_attention(c, q, k, vt, layers[l].in_proj_b, attn, ws.planes, ws.tokens, ws.tokpad, st);
        if (e != hipSuccess) return hip_fail("attention", e);
        const bool more = l < last_layer;
        e = launch_chain(c, &layers[l], wp + l * pl, more ? &layers[l + 1] : nullptr,
                         more ? wp + (l + 1) * pl : nullptr, attn, x, q, k, vt, rows, ws.tokens, ws.tokpad, st,
                         fused ? (more ? &middle : &last) : nullptr);
        if (e != hipSuccess) return hip_fail("chain(mlp)", e);
    }
    return AFT_OK;
}

// ---- the general engine (round 6) ----
// One nn.TransformerEncoderLayer in eval mode on row-major x [rows][d], in place: the launch sequence of the training forward
// (aft_train.hip) with dropout off -- in-projection GEMM, attention (heads padded to 32-feature blocks where needed), out-projection GEMM,
// residual + LayerNorm, linear1 GEMM, activation, linear2 GEMM, residual + LayerNorm -- for every shape the packed engine does not
// take.  The tensors between the launches are one layer's worth of workspace, re-used layer after layer.
#define GSTEP(name, call)                                   \
    do {                                                    \
        hipError_t e_ = (call);                             \
        if (e_ != hipSuccess) return hip_fail(name, e_);    \
    } while (0)
static int general_layer(const aft_config &c, const aft_layer_weights &lw, const Workspace &ws, float *base, float *x, hipStream_t st) {
    const int rows = ws.planes * ws.tokens, d = c.model_dim, ff = 2 * d;
    float *qkv = base + ws.g_qkv, *attn = base + ws.attn, *y = base + ws.g_y, *x1 = base + ws.g_x1, *s = base + ws.g_s,
          *stats = base + ws.g_stats, *a = base + ws.g_a, *hd = base + ws.g_hd;
    GSTEP("general: in-projection", launch_gemm(0, x, lw.in_proj_w, qkv, lw.in_proj_b, rows, 3 * d, d, d, d, 3 * d, false, st));
    GSTEP("general: attention", launch_attn_train_fwd(c, qkv, attn, base + ws.g_lse, ws.planes, ws.tokens, 0.f, 0u, st, base + ws.g_pad));
    GSTEP("general: out-projection", launch_gemm(0, attn, lw.out_proj_w, y, lw.out_proj_b, rows, d, d, d, d, d, false, st));
    GSTEP("general: norm1", launch_add_ln_fwd(x, y, lw.norm1_w, lw.norm1_b, s, stats, x1, rows, d, 1e-5f, 0.f, 0u, st));
    GSTEP("general: linear1", launch_gemm(0, x1, lw.lin1_w, a, lw.lin1_b, rows, ff, d, d, d, ff, false, st));
    GSTEP("general: activation", launch_act_fwd(c.activation, a, hd, rows, ff, 0.f, 0u, st));
    GSTEP("general: linear2", launch_gemm(0, hd, lw.lin2_w, y, lw.lin2_b, rows, d, ff, ff, ff, d, false, st));
    GSTEP("general: norm2", launch_add_ln_fwd(x1, y, lw.norm2_w, lw.norm2_b, s, stats, x, rows, d, 1e-5f, 0.f, 0u, st));
    return AFT_OK;
}
// embedding -> layers -> linear_2 into out6 [rows][out6_stride] (what the conv tail reads)
static int run_encoder_general(const aft_config &c, const WeightsDev &w, const aft_layer_weights *layers, const Workspace &ws, float *base,
                               int batch, hipStream_t st) {
    float *x = base + ws.x;
    const int rows = ws.planes * ws.tokens, d = c.model_dim, p = c.patch_scs * c.patch_symbols;
    GSTEP("general: embedding", launch_embed_any(c, w, base + ws.conv_enhanced, c.adaptive ? base + ws.tokens6 : nullptr, x, batch, st));
    for (int l = 0; l < c.num_layers; ++l) {
        const int rc = general_layer(c, layers[l], ws, base, x, st);
        if (rc != AFT_OK) return rc;
    }
    GSTEP("general: linear_2", launch_gemm(0, x, w.lin2_w, base + ws.out6, w.lin2_b, rows, p, d, d, d, out6_stride(c), false, st));
    return AFT_OK;
}

}  // namespace aft

using namespace aft;

extern "C" {

int aft_version(void) { return AFT_ABI_VERSION; }

const char *aft_last_error(void) { return g_err; }

int aft_check_config(const aft_config *cfg) { return check_config(cfg); }

int aft_engine_of(const aft_config *cfg) {
    if (check_config(cfg) != AFT_OK) return -1;
    return packed_engine_ok(*cfg) ? AFT_ENGINE_PACKED : AFT_ENGINE_GENERAL;
}

int aft_set_switch(const char *name, const char *value) {
    if (name == nullptr || strncmp(name, "AFT_", 4) != 0) {
        set_error("aft_set_switch: switch names start with AFT_");
        return AFT_ERR_ARG;
    }
    Switches &s = switches();
    std::lock_guard<std::mutex> lock(s.mu);
    if (value == nullptr) s.table.erase(name);
    else s.table[name] = value;
    return AFT_OK;
}

int aft_get_switch(const char *name, char *buf, size_t n) {
    if (name == nullptr) return 0;
    Switches &s = switches();
    std::lock_guard<std::mutex> lock(s.mu);
    auto it = s.table.find(name);
    if (it == s.table.end()) return 0;
    if (buf != nullptr && n > 0) {
        strncpy(buf, it->second.c_str(), n - 1);
        buf[n - 1] = 0;
    }
    return 1;
}

int aft_max_batch(const aft_config *cfg) {
    if (check_config(cfg) != AFT_OK) return 0;
    return max_batch_of(*cfg);
}

size_t aft_workspace_bytes(const aft_config *cfg, int batch) {
    if (check_config(cfg) != AFT_OK || batch <= 0) return 0;
    return workspace_floats_any_lanes(*cfg, batch) * sizeof(float);
}

int aft_workspace_region(const aft_config *cfg, int batch, int region, size_t *offset_bytes, size_t *size_bytes) {
    int rc = check_config(cfg);
    if (rc != AFT_OK) return rc;
    if (batch <= 0 || !offset_bytes || !size_bytes) {
        set_error("aft_workspace_region: bad batch or NULL result pointer");
        return AFT_ERR_ARG;
    }
    const Workspace ws = plan_workspace(*cfg, batch);
    size_t off = 0, n = 0;
    switch (region) {
        case AFT_REGION_CONV_ENHANCED: off = ws.conv_enhanced; n = (size_t)ws.planes * cfg->num_scs * cfg->num_symbols; break;
        case AFT_REGION_TOKENS6: off = ws.tokens6; n = cfg->adaptive ? (size_t)batch * ws.tokens * 6 : 0; break;
        case AFT_REGION_ENC_OUT: off = ws.out6; n = (size_t)ws.planes * ws.tokens * out6_stride(*cfg); break;
        default: set_error("aft_workspace_region: unknown region %d", region); return AFT_ERR_ARG;
    }
    *offset_bytes = off * sizeof(float);
    *size_bytes = n * sizeof(float);
    return AFT_OK;
}

int aft_workspace_lanes(const aft_config *cfg, int batch, int *lanes, int *frames, size_t *offset_bytes) {
    int rc = check_config(cfg);
    if (rc != AFT_OK) return rc;
    if (batch <= 0 || !lanes || !frames || !offset_bytes) {
        set_error("aft_workspace_lanes: bad batch or NULL result pointer");
        return AFT_ERR_ARG;
    }
    static_assert(kMaxLanes == AFT_MAX_LANES, "header and implementation disagree");
    const LanePlan lp = plan_lanes(*cfg, batch, lanes_wanted(*cfg, batch));
    *lanes = lp.lanes;
    for (int i = 0; i < kMaxLanes; ++i) {
        frames[i] = i < lp.lanes ? lp.frames[i] : 0;
        offset_bytes[i] = i < lp.lanes ? lp.ws_off[i] * sizeof(float) : 0;
    }
    return AFT_OK;
}

#define AFT_REQUIRE(cond, ...)        \
    do {                              \
        if (!(cond)) {                \
            set_error(__VA_ARGS__);   \
            return AFT_ERR_ARG;       \
        }                             \
    } while (0)

static int forward_lane(const aft_config *cfg, const aft_weights *w, const float *prepacked, const float *pilots, const float *snr,
                        const float *ds, const float *dop, float *out, float *base, int batch, hipStream_t st);

static int forward_impl(const aft_config *cfg, const aft_weights *w, const float *prepacked, const float *pilots,
                        const float *snr, const float *ds, const float *dop, float *out, void *workspace,
                        size_t workspace_bytes, int batch, void *stream) {
    int rc = check_config(cfg);
    if (rc != AFT_OK) return rc;
    AFT_REQUIRE(w && pilots && out && workspace, "NULL pointer argument");
    AFT_REQUIRE(batch > 0, "batch must be positive (got %d)", batch);
    AFT_REQUIRE(batch <= max_batch_of(*cfg), "batch %d exceeds aft_max_batch = %d for this configuration (32-bit buffer "
                "offsets: no workspace region may reach 2 GiB); split the batch", batch, max_batch_of(*cfg));
    // reference fortitran.py:157-158: meta_data is required when channel adaptation is enabled
    AFT_REQUIRE(!cfg->adaptive || (snr && ds && dop), "meta_data is required when channel adaptation is enabled");
    hipStream_t user = static_cast<hipStream_t>(stream);
    AFT_REQUIRE(w->layers != nullptr, "aft_weights.layers is NULL (host array of num_layers entries)");
    const LanePlan lp = plan_lanes(*cfg, batch, lanes_wanted(*cfg, batch));
    AFT_REQUIRE(workspace_bytes >= lp.total_floats * sizeof(float), "workspace too small: %zu < %zu bytes (aft_workspace_bytes)",
                workspace_bytes, lp.total_floats * sizeof(float));
    // side streams for the shares beyond the first; none to be had (header, "Lanes") = the shares run one after the other on the
    // caller's stream in the same workspace layout: same bits, and aft_workspace_lanes stays true
    LaneStreams *ls = lp.lanes > 1 ? acquire_lane_streams(user, lp.lanes) : nullptr;
    struct Lease { LaneStreams *p; ~Lease() { release_lane_streams(p); } } lease{ls};
    if (ls != nullptr) {
        hipError_t ef = hipEventRecord(ls->fork, user);
        for (int i = 1; i < lp.lanes && ef == hipSuccess; ++i) ef = hipStreamWaitEvent(ls->side[i - 1], ls->fork, 0);
        if (ef != hipSuccess) return hip_fail("lanes(fork)", ef);
    }
    const int pil_floats = 2 * cfg->pilot_scs * cfg->pilot_symbols, out_floats = 2 * cfg->num_scs * cfg->num_symbols;   // complex64 per frame
#ifdef AFT_CHECKED
    {   // workspace-plan invariants: shares cover the batch exactly once, slices are disjoint, in order and inside the caller's buffer,
        // every region of a slice lies inside it and starts on a 256-byte boundary
        int covered = 0;
        for (int i = 0; i < lp.lanes; ++i) {
            const Workspace wl = plan_workspace(*cfg, lp.frames[i]);
            AFT_HOST_ASSERT(lp.first[i] == covered && lp.frames[i] > 0, "lane shares are not a partition of the batch");
            covered += lp.frames[i];
            AFT_HOST_ASSERT(lp.ws_off[i] % 64 == 0 && (lp.ws_off[i] + wl.total_floats) * sizeof(float) <= workspace_bytes, "lane slice outside the workspace");
            AFT_HOST_ASSERT(i + 1 == lp.lanes || lp.ws_off[i] + wl.total_floats <= lp.ws_off[i + 1], "lane slices overlap");
            const size_t regions[] = {wl.conv_enhanced, wl.tokens6, wl.x, wl.attn, wl.q, wl.k, wl.vt, wl.wpack, wl.out6, wl.convfrag,
                                      wl.g_x1, wl.g_y, wl.g_s, wl.g_stats, wl.g_qkv, wl.g_lse, wl.g_a, wl.g_hd, wl.g_pad};
            for (size_t r : regions) AFT_HOST_ASSERT(r % 64 == 0 && r <= wl.total_floats, "workspace region outside its slice");
        }
        AFT_HOST_ASSERT(covered == batch, "lane shares do not cover the batch");
    }
#endif
    int result = AFT_OK;
    for (int i = 0; i < lp.lanes; ++i) {
        const int f0 = lp.first[i];
        const int rc_lane = forward_lane(cfg, w, prepacked, pilots + (size_t)f0 * pil_floats, snr ? snr + f0 : nullptr, ds ? ds + f0 : nullptr,
                                         dop ? dop + f0 : nullptr, out + (size_t)f0 * out_floats, static_cast<float *>(workspace) + lp.ws_off[i],
                                         lp.frames[i], i == 0 || ls == nullptr ? user : ls->side[i - 1]);
        if (rc_lane != AFT_OK && result == AFT_OK) result = rc_lane;     // keep going: the join below must still happen
    }
    if (ls != nullptr) {
        hipError_t ej = hipSuccess;
        for (int i = 1; i < lp.lanes && ej == hipSuccess; ++i) {
            ej = hipEventRecord(ls->join[i - 1], ls->side[i - 1]);
            if (ej == hipSuccess) ej = hipStreamWaitEvent(user, ls->join[i - 1], 0);
        }
        if (ej != hipSuccess && result == AFT_OK) return hip_fail("lanes(join)", ej);
    }
    return result;
}

// one share of the batch: a complete forward on `st` in its own workspace slice
static int forward_lane(const aft_config *cfg, const aft_weights *w, const float *prepacked, const float *pilots, const float *snr,
                        const float *ds, const float *dop, float *out, float *base, int batch, hipStream_t st) {
    const Workspace ws = plan_workspace(*cfg, batch);
    const bool general = !packed_engine_ok(*cfg);
    const WeightsDev wd = weights_window(*w, 0, cfg->num_layers);
    int rc = AFT_OK;
    // prologue: the channel adapter, -- unless the caller owns a packed image -- the re-lay of the encoder weights into fragment
    // order, and the pilot_upsampler product over all planes, in ONE launch (none depends on anything the forward computes).
    // The activation buffer x is idle until the first encoder launch: it lends the upsampler its plane scratch when it is large enough
    float *wpack = prepacked != nullptr || general ? nullptr : base + ws.wpack;      // (the general engine reads the weights as PyTorch holds them)
    const size_t x_floats = general ? ((size_t)1 << 62) : (size_t)ws.planes * ws.tokens * cfg->model_dim;   // general: x is sized for the planes
    const size_t up_floats = (size_t)ws.planes * cfg->num_scs * cfg->num_symbols;
    float *up_planes = x_floats >= up_floats && prologue_upsample_ok(*cfg, wd) ? base + ws.x : nullptr;
    hipError_t e = launch_prologue(*cfg, wd, snr, ds, dop, base + ws.tokens6, batch, wpack, pilots, up_planes, st, base + ws.convfrag);
    if (e != hipSuccess) return hip_fail("prologue(adapter + pack_weights + upsampler product)", e);
    if (wpack != nullptr && cfg->num_layers > kLayerWindow) {     // the prologue launch packs the window of layers its argument holds
        e = launch_pack_weights(*cfg, w->layers + kLayerWindow, wpack + packed_layer_floats(cfg->model_dim) * kLayerWindow,
                                cfg->num_layers - kLayerWindow, st);
        if (e != hipSuccess) return hip_fail("pack_weights(layers beyond the prologue's window)", e);
    }
    e = launch_upsample(*cfg, wd, pilots, base + ws.conv_enhanced, batch, st, x_floats >= up_floats ? base + ws.x : nullptr, up_planes != nullptr,
                        base + ws.convfrag);
    if (e != hipSuccess) return hip_fail("upsample", e);
    if (general) {
        rc = run_encoder_general(*cfg, wd, w->layers, ws, base, batch, st);
    } else {
        // patch embedding + linear_1 + positions run inside the first chain launch, linear_2 inside the last one
        rc = run_encoder(*cfg, wd, w->layers, ws, base, 0, cfg->num_layers - 1, st, true, prepacked != nullptr ? prepacked : wpack);
    }
    if (rc != AFT_OK) return rc;
    e = launch_tail(*cfg, wd, nullptr, base + ws.conv_enhanced, out, batch, st, base + ws.out6, base + ws.convfrag + kConvFragFloats);
    if (e != hipSuccess) return hip_fail("tail", e);
    return AFT_OK;
}

int aft_forward_f32(const aft_config *cfg, const aft_weights *w, const float *pilots, const float *snr,
                    const float *ds, const float *dop, float *out, void *workspace, size_t workspace_bytes,
                    int batch, void *stream) {
    return forward_impl(cfg, w, nullptr, pilots, snr, ds, dop, out, workspace, workspace_bytes, batch, stream);
}

size_t aft_packed_weights_bytes(const aft_config *cfg) {
    if (check_config(cfg) != AFT_OK) return 0;
    if (!packed_engine_ok(*cfg)) return 64;     // the general engine reads the weights in place: a token image keeps the prepacked calls valid
    return packed_layer_floats(cfg->model_dim) * cfg->num_layers * sizeof(float);
}

int aft_pack_weights_f32(const aft_config *cfg, const aft_weights *w, void *packed, size_t packed_bytes, void *stream) {
    int rc = check_config(cfg);
    if (rc != AFT_OK) return rc;
    AFT_REQUIRE(w && packed, "NULL pointer argument");
    AFT_REQUIRE(packed_bytes >= aft_packed_weights_bytes(cfg), "packed-weight buffer too small: %zu < %zu bytes", packed_bytes,
                aft_packed_weights_bytes(cfg));
    AFT_REQUIRE(w->layers != nullptr, "aft_weights.layers is NULL (host array of num_layers entries)");
    if (!packed_engine_ok(*cfg)) return AFT_OK;
    hipError_t e = launch_pack_weights(*cfg, w->layers, static_cast<float *>(packed), cfg->num_layers, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? AFT_OK : hip_fail("pack_weights", e);
}

int aft_forward_prepacked_f32(const aft_config *cfg, const aft_weights *w, const void *packed, const float *pilots,
                              const float *snr, const float *ds, const float *dop, float *out, void *workspace,
                              size_t workspace_bytes, int batch, void *stream) {
    AFT_REQUIRE(packed != nullptr, "NULL packed-weight image");
    return forward_impl(cfg, w, static_cast<const float *>(packed), pilots, snr, ds, dop, out, workspace, workspace_bytes, batch,
                        stream);
}

int aft_linear_forward_f32(const float *weight, const float *bias, const float *pilots, float *out, int batch,
                           int in_features, int out_features, void *stream) {
    AFT_REQUIRE(weight && pilots && out, "NULL pointer argument");
    AFT_REQUIRE(batch > 0 && in_features > 0 && out_features > 0, "bad sizes");
    AFT_REQUIRE(in_features <= 8192, "in_features too large for the LDS-staged kernel");
    hipError_t e = launch_linear(weight, bias, pilots, out, batch, in_features, out_features,
                                 static_cast<hipStream_t>(stream));
    return e == hipSuccess ? AFT_OK : hip_fail("linear", e);
}

int aft_mse_partial_f32(const float *est, const float *ref, double *sum_sq, long long n_complex, void *stream) {
    AFT_REQUIRE(est && ref && sum_sq, "NULL pointer argument");
    AFT_REQUIRE(n_complex >= 0, "negative element count");
    if (n_complex == 0) return AFT_OK;
    hipError_t e = launch_mse(est, ref, sum_sq, n_complex, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? AFT_OK : hip_fail("mse", e);
}

int aft_pilot_gather_f32(const float *hzero_ls, float *pilots, int *counts, int batch, int grid_elems, int expected,
                         void *stream) {
    AFT_REQUIRE(hzero_ls && pilots && counts, "NULL pointer argument");
    AFT_REQUIRE(batch > 0 && grid_elems > 0 && expected > 0 && expected <= grid_elems, "bad sizes");
    hipError_t e = launch_pilot_gather(hzero_ls, pilots, counts, batch, grid_elems, expected,
                                       static_cast<hipStream_t>(stream));
    return e == hipSuccess ? AFT_OK : hip_fail("pilot_gather", e);
}

int aft_ls_mse_db_f32(const float *ls, const float *ideal, float *db, int batch, int grid_elems, void *stream) {
    AFT_REQUIRE(ls && ideal && db, "NULL pointer argument");
    AFT_REQUIRE(batch > 0 && grid_elems > 0, "bad sizes");
    hipError_t e = launch_ls_mse_db(ls, ideal, db, batch, grid_elems, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? AFT_OK : hip_fail("ls_mse_db", e);
}

int aft_debug_fill_lds_f32(float value, void *stream) {
    hipError_t e = launch_fill_lds(value, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? AFT_OK : hip_fail("fill_lds", e);
}

int aft_debug_peek_lds_f32(float *out, int workgroups, int n, void *stream) {
    AFT_REQUIRE(out && workgroups > 0 && n > 0 && n <= 10240, "bad argument");
    hipError_t e = launch_peek_lds(out, workgroups, n, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? AFT_OK : hip_fail("peek_lds", e);
}

int aft_stage_upsample_f32(const aft_config *cfg, const aft_weights *w, const float *pilots, float *conv_enhanced,
                           int batch, void *stream) {
    int rc = check_config(cfg);
    if (rc != AFT_OK) return rc;
    AFT_REQUIRE(w && pilots && conv_enhanced && batch > 0, "bad argument");
    hipError_t e = launch_upsample(*cfg, weights_window(*w, 0, 0), pilots, conv_enhanced, batch, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? AFT_OK : hip_fail("upsample", e);
}

int aft_stage_adapter_f32(const aft_config *cfg, const aft_weights *w, const float *snr, const float *ds,
                          const float *dop, float *tokens6, int batch, void *stream) {
    int rc = check_config(cfg);
    if (rc != AFT_OK) return rc;
    AFT_REQUIRE(cfg->adaptive, "adapter stage needs an adaptive config");
    AFT_REQUIRE(w && snr && ds && dop && tokens6 && batch > 0, "bad argument");
    hipError_t e = launch_adapter(*cfg, weights_window(*w, 0, 0), snr, ds, dop, tokens6, batch, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? AFT_OK : hip_fail("adapter", e);
}

int aft_stage_embed_f32(const aft_config *cfg, const aft_weights *w, const float *conv_enhanced,
                        const float *tokens6, float *x, int batch, void *stream) {
    int rc = check_config(cfg);
    if (rc != AFT_OK) return rc;
    AFT_REQUIRE(w && conv_enhanced && x && batch > 0, "bad argument");
    AFT_REQUIRE(!cfg->adaptive || tokens6, "adaptive config needs tokens6");
    hipError_t e = packed_engine_ok(*cfg) ? launch_embed(*cfg, weights_window(*w, 0, 0), conv_enhanced, tokens6, x, batch, static_cast<hipStream_t>(stream))
                                          : launch_embed_any(*cfg, weights_window(*w, 0, 0), conv_enhanced, tokens6, x, batch, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? AFT_OK : hip_fail("embed", e);
}

int aft_stage_encoder_layer_f32(const aft_config *cfg, const aft_weights *w, int layer, float *x, void *scratch,
                                size_t scratch_bytes, int batch, void *stream) {
    int rc = check_config(cfg);
    if (rc != AFT_OK) return rc;
    AFT_REQUIRE(w && w->layers && x && scratch && batch > 0, "bad argument");
    AFT_REQUIRE(layer >= 0 && layer < cfg->num_layers, "layer %d out of range", layer);
    AFT_REQUIRE(batch <= max_batch_of(*cfg), "batch %d exceeds aft_max_batch = %d", batch, max_batch_of(*cfg));
    Workspace ws = plan_workspace(*cfg, batch);
    AFT_REQUIRE(scratch_bytes >= ws.total_floats * sizeof(float), "scratch too small");
    // run on the caller's x: point the plan's x slot at it (offsets are relative to scratch)
    float *base = static_cast<float *>(scratch);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (!packed_engine_ok(*cfg)) return general_layer(*cfg, w->layers[layer], ws, base, x, st);
    const int rows = ws.planes * ws.tokens;
    float *wp = base + ws.wpack + layer * packed_layer_floats(cfg->model_dim);
    hipError_t e = launch_pack_weights(*cfg, w->layers + layer, wp, 1, st);
    if (e != hipSuccess) return hip_fail("pack_weights", e);
    e = launch_chain(*cfg, nullptr, nullptr, &w->layers[layer], wp, nullptr, x, base + ws.q, base + ws.k, base + ws.vt,
                     rows, ws.tokens, ws.tokpad, st);
    if (e != hipSuccess) return hip_fail("chain(qkv)", e);
    e = launch_attention(*cfg, base + ws.q, base + ws.k, base + ws.vt, w->layers[layer].in_proj_b, base + ws.attn,
                         ws.planes, ws.tokens, ws.tokpad, st);
    if (e != hipSuccess) return hip_fail("attention", e);
    e = launch_chain(*cfg, &w->layers[layer], wp, nullptr, nullptr, base + ws.attn, x, nullptr, nullptr, nullptr, rows,
                     ws.tokens, ws.tokpad, st);
    return e == hipSuccess ? AFT_OK : hip_fail("chain(mlp)", e);
}

int aft_stage_tail_f32(const aft_config *cfg, const aft_weights *w, const float *x, const float *conv_enhanced,
                       float *out, int batch, void *stream) {
    int rc = check_config(cfg);
    if (rc != AFT_OK) return rc;
    AFT_REQUIRE(w && x && conv_enhanced && out && batch > 0, "bad argument");
    {   // this call owns no scratch: the conv kernel applies linear_2 itself, from weights staged in its LDS beside the plane
        const int p = cfg->patch_scs * cfg->patch_symbols;
        if (cfg->model_dim % 16 != 0 || !conv_plan_ok(cfg->num_scs, cfg->num_symbols, p * cfg->model_dim + p)) {
            set_error("aft_stage_tail_f32: linear_2 (%d x %d) does not fit the conv kernel's LDS beside the plane, or model_dim is not a "
                      "multiple of 16 -- only aft_forward_f32 serves this configuration", p, cfg->model_dim);
            return AFT_ERR_SHAPE;
        }
    }
    hipError_t e = launch_tail(*cfg, weights_window(*w, 0, 0), x, conv_enhanced, out, batch, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? AFT_OK : hip_fail("tail", e);
}

int aft_profile_kernel_f32(const aft_config *cfg, const aft_weights *w, int which, float *out, void *workspace,
                           size_t workspace_bytes, int batch, int reps, void *stream) {
    int rc = check_config(cfg);
    if (rc != AFT_OK) return rc;
    AFT_REQUIRE(w && w->layers && workspace && batch > 0 && reps > 0, "bad argument");
    AFT_REQUIRE(batch <= max_batch_of(*cfg), "batch %d exceeds aft_max_batch = %d", batch, max_batch_of(*cfg));
    if (!packed_engine_ok(*cfg)) {
        set_error("aft_profile_kernel_f32 replays the packed engine's kernel classes; this configuration runs the general engine");
        return AFT_ERR_SHAPE;
    }
    const Workspace ws = plan_workspace(*cfg, batch);
    AFT_REQUIRE(workspace_bytes >= ws.total_floats * sizeof(float), "workspace too small");
    const WeightsDev wd = weights_window(*w, 0, cfg->num_layers);
    AFT_REQUIRE(cfg->num_layers >= 2 || which != AFT_KERNEL_CHAIN, "chain profile needs >= 2 layers");
    hipStream_t st = static_cast<hipStream_t>(stream);
    float *base = static_cast<float *>(workspace);
    float *x = base + ws.x, *attn = base + ws.attn, *q = base + ws.q, *k = base + ws.k, *vt = base + ws.vt;
    const int rows = ws.planes * ws.tokens;
    hipError_t e = hipSuccess;
    for (int i = 0; i < reps && e == hipSuccess; ++i) {
        switch (which) {
            case AFT_KERNEL_UPSAMPLE:
                AFT_REQUIRE(out != nullptr, "upsample profile needs the pilots pointer in `out`");
                // as the forward: the planes of the pilot_upsampler product lie in x (computed by the prologue launch in the forward;
                // here whatever x holds -- the conv stack's time does not depend on its data)
                e = launch_upsample(*cfg, wd, out, base + ws.conv_enhanced, batch, st,
                                    (size_t)ws.planes * ws.tokens * cfg->model_dim >= (size_t)ws.planes * cfg->num_scs * cfg->num_symbols ? x : nullptr,
                                    prologue_upsample_ok(*cfg, wd), base + ws.convfrag);
                break;
            case AFT_KERNEL_EMBED:
                e = launch_embed(*cfg, wd, base + ws.conv_enhanced, cfg->adaptive ? base + ws.tokens6 : nullptr, x, batch, st);
                break;
            case AFT_KERNEL_QKV: {   // as in the forward: embedding fused in front of the in-projection
                ChainFusion f{};
                f.conv_enhanced = base + ws.conv_enhanced;
                f.tokens6 = cfg->adaptive ? base + ws.tokens6 : nullptr;
                f.lin1_w = w->lin1_w; f.lin1_b = w->lin1_b; f.pos = w->pos;
                f.x_blocked = true;
                e = launch_chain(*cfg, nullptr, nullptr, &w->layers[0], base + ws.wpack, nullptr, x, q, k, vt, rows,
                                 ws.tokens, ws.tokpad, st, &f);
                break;
            }
            case AFT_KERNEL_ATTENTION:
                e = launch_attention(*cfg, q, k, vt, w->layers[0].in_proj_b, attn, ws.planes, ws.tokens, ws.tokpad, st);
                break;
            case AFT_KERNEL_CHAIN: {   // as in the forward: x in tile-blocked order
                ChainFusion f{};
                f.x_blocked = true;
                e = launch_chain(*cfg, &w->layers[0], base + ws.wpack, &w->layers[1],
                                 base + ws.wpack + packed_layer_floats(cfg->model_dim), attn, x, q, k, vt, rows,
                                 ws.tokens, ws.tokpad, st, &f);
                break;
            }
            case AFT_KERNEL_CHAIN_LAST: {   // as in the forward: linear_2 fused behind LN2, x not stored
                ChainFusion f{};
                f.lin2_w = w->lin2_w; f.lin2_b = w->lin2_b; f.out6 = base + ws.out6;
                f.x_blocked = true;
                e = launch_chain(*cfg, &w->layers[0], base + ws.wpack, nullptr, nullptr, attn, x, q, k, vt, rows,
                                 ws.tokens, ws.tokpad, st, &f);
                break;
            }
            case AFT_KERNEL_ENCODER_PLANE:
                AFT_REQUIRE(cfg->model_dim == 128, "the plane-resident encoder is instantiated for model_dim 128");
                e = launch_encoder_plane(*cfg, wd, base + ws.wpack, base + ws.conv_enhanced,
                                         cfg->adaptive ? base + ws.tokens6 : nullptr, x, attn, q, k, vt, base + ws.out6,
                                         ws.planes, ws.tokens, ws.tokpad, st);
                break;
            case AFT_KERNEL_PROLOGUE: {
                AFT_REQUIRE(out != nullptr, "prologue profile needs the pilots pointer in `out`");
                const bool lend = (size_t)ws.planes * ws.tokens * cfg->model_dim >= (size_t)ws.planes * cfg->num_scs * cfg->num_symbols;
                const float *cond = out + (size_t)batch * cfg->pilot_scs * cfg->pilot_symbols * 2;   // [snr | ds | dop] behind the pilots
                // AFT_PROLOGUE_NO_UP=1 (measurement only): the launch without the pilot_upsampler product -- bench.py charges the
                // difference to the upsampler stage (the stage is K0 + K1 + K2)
                e = launch_prologue(*cfg, wd, cond, cond + batch, cond + 2 * batch, base + ws.tokens6, batch, base + ws.wpack, out,
                                    lend && prologue_upsample_ok(*cfg, wd) && !switch_on("AFT_PROLOGUE_NO_UP") ? x : nullptr, st,
                                    base + ws.convfrag);
                break;
            }
            case AFT_KERNEL_TAIL:
                AFT_REQUIRE(out != nullptr, "tail profile needs an output buffer");
                e = launch_tail(*cfg, wd, nullptr, base + ws.conv_enhanced, out, batch, st, base + ws.out6, base + ws.convfrag + kConvFragFloats);
                break;
            default:
                set_error("unknown kernel id %d", which);
                return AFT_ERR_ARG;
        }
    }
    return e == hipSuccess ? AFT_OK : hip_fail("profile", e);
}

}  // extern "C"
