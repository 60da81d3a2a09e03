// cart_engine.hip -- C-ABI implementation (include/cart_engine.h): workspace pool, stage
// sequencing and the host-side peak finder.  No exceptions cross the ABI and nothing exits.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "engine_internal.h"

using namespace cart_amd;

namespace {

thread_local std::string g_last_error;
thread_local int g_last_slot = 0;  // first slot of this thread's most recent compute lease

int fail(const std::string &msg) {
    g_last_error = msg;
    return -1;
}

#define HIP_TRY(expr)                                                                               \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess)                                                                       \
            return fail(std::string(#expr) + ": " + hipGetErrorString(_e) + " (" + __FILE__ + ":" + \
                        std::to_string(__LINE__) + ")");                                           \
    } while (0)

struct Slot {
    bool busy = false;
    hipEvent_t done = nullptr;          // recorded only on the FIRST slot of a lease ...
    int owner = -1;                     // ... every slot of the lease points at that slot
    hipStream_t last_stream = nullptr;
    bool used = false;
    unsigned long long released_seq = 0;  // order of the last release (guarded by mu)
};

constexpr int kMaxTimings = 8;
constexpr int kTimingRing = 256;

struct TimingRec {
    hipEvent_t ev[kMaxTimings + 1] = {};
    const char *names[kMaxTimings] = {};
    int n = 0;
};

}  // namespace

// The slab workspace: one plain hipMalloc per GROUP of workspace slots, no group larger than kSlabChunkBytes (see slab_pool_alloc).
// Slot s lives at base[s / group_slots] + (s % group_slots) * slot_bytes; the kernels of a launch get the slab pointers of their
// frames as a table (SlabTable), so a launch may span groups.
static_assert(kMaxLaunchFrames == 64, "cart_engine_set_option(CART_OPT_CHUNK_FRAMES) documents 1..64");
struct SlabPool {
    std::vector<uint8_t *> base;   // one device allocation per group
    int group_slots = 0;           // slots per group (the last group may hold fewer)
    int slots = 0;
    size_t slot_bytes = 0;         // P path slabs of one frame
    int groups() const { return (int)base.size(); }
    int slots_of(int gi) const { return std::min(group_slots, slots - gi * group_slots); }
    size_t bytes_of(int gi) const { return (size_t)slots_of(gi) * slot_bytes; }
    uint8_t *slot_ptr(int s) const { return base[s / group_slots] + (size_t)(s % group_slots) * slot_bytes; }
};

struct cart_engine {
    cart_engine_params params;
    Geometry g;
    float uniq;
    uint16_t *uniq_thr = nullptr;   // device: integer uniqueness threshold of every best cost 0..2047 for this engine's ratio (WTA kernels)
    // workspaces, each [max_inflight][...]
    uint8_t *gray_l = nullptr, *gray_r = nullptr;
    uint32_t *cen_l = nullptr, *cen_r = nullptr;      // point `cen_slack` elements into their allocations
    uint32_t *cen_l_alloc = nullptr, *cen_r_alloc = nullptr;
    size_t cen_slack = 0;
    uint16_t *wta_l = nullptr;
    uint32_t *right_pk = nullptr;
    int16_t *tmp_a = nullptr, *tmp_b = nullptr;  // tight s16 planes (interpolate ping-pong)
    int32_t *ccl_work = nullptr;
    uint32_t *rv_partial = nullptr; // [max_inflight][wta_fused_partial_elems], allocated by the first fused batch
    uint8_t *flow_ws = nullptr;     // [max_inflight][flow_ws_bytes]: gray x2, census x2, scratch; first cart_optical_flow allocates
    int32_t *ccl_stats_ws = nullptr; // component-table workspace (ensure_ccl_stats_ws): [max_inflight][npx][5] scratch + [max_inflight][h][tile columns]; first table call allocates
    unsigned *sp_votes = nullptr;   // [max_inflight][kSpMaxLabels*3], allocated by the first cart_superpixel_plane_classify
    AggArgs agg;
    AggArgs agg_fused;              // the same launch without the "up" direction (computed inside wta_fused_kernel)
    SlabPool slab_pool;             // the cost slabs of every slot (slab_pool_alloc / slab_pool_free / cart_engine_tune_placement)
    int auto_fused_min_frames = 1 << 30; // CART_OPT_PLAN = auto: launches of at least this many frames take the fused WTA
    int opt_plan = CART_PLAN_AUTO;       // cart_engine_set_option
    int opt_plan_min_frames = 1;         // with a forced plan: launches of fewer frames still take CART_PLAN_SLABS
    int opt_spec = 0;                    // CART_OPT_SPEC_* bits: upstream variants of S8 / S7 (default: the oracle's spec)
    std::mutex mu;
    std::condition_variable cv;
    std::vector<Slot> slots;
    unsigned long long release_counter = 0;    // guarded by mu
    int chunk_frames = kLaunchFrames;          // frames per launch sequence inside one batched call
    bool post_only = false;         // no SGM workspaces (num_disparities == 0)
    bool timing = false;
    int timing_every = 1;           // stage events on every timing_every-th compute call (cart_engine_set_timing)
    unsigned long long timing_calls = 0;   // compute calls seen while timing is on (guarded by mu)
    std::vector<TimingRec> ring;  // stage events of the last kTimingRing compute calls (guarded by mu)
    size_t ring_calls = 0;
};

namespace {

struct Lease {
    cart_engine *e;
    int s0, n;
    hipStream_t stream;
};

// Leases `n` contiguous slots; makes `stream` wait for earlier work on them from other streams.
int acquire(cart_engine *e, int n, hipStream_t stream, Lease *out) {
    if (n <= 0 || n > (int)e->slots.size()) return fail("n_frames must be in [1, max_inflight]");
    std::unique_lock<std::mutex> lk(e->mu);
    int s0 = -1;
    for (;;) {
        // Preference: a free range last used on THIS stream (or never): no event wait at all; otherwise the free range that
        // was released longest ago -- a first fit would hand a pipelined caller the slots of its previous batch, whose tail
        // may still be queued on another stream, and serialise the two streams.
        const int total = (int)e->slots.size();
        int oldest = -1;
        unsigned long long oldest_seq = ~0ull;
        for (int i = 0; i + n <= total && s0 < 0; ++i) {
            bool ok = true, same = true;
            unsigned long long seq = 0;
            for (int k = 0; k < n; ++k) {
                const Slot &sl = e->slots[i + k];
                if (sl.busy) { ok = false; i += k; break; }
                same &= !sl.used || sl.last_stream == stream;
                seq = std::max(seq, sl.released_seq);
            }
            if (!ok) continue;
            if (same) s0 = i;
            else if (seq < oldest_seq) { oldest_seq = seq; oldest = i; }
        }
        if (s0 < 0) s0 = oldest;
        if (s0 >= 0) break;
        e->cv.wait(lk);
    }
    for (int k = 0; k < n; ++k) e->slots[s0 + k].busy = true;
    lk.unlock();
    // Cross-stream reuse of a slot waits for the event of the lease that used it last.  That event may have been
    // re-recorded by a later lease of its owner slot: waiting for more than necessary is harmless (an event wait only
    // ever refers to work enqueued before the wait).
    int waited = -1;
    for (int k = 0; k < n; ++k) {
        Slot &s = e->slots[s0 + k];
        if (s.used && s.last_stream != stream && s.owner != waited) {
            hipError_t err = hipStreamWaitEvent(stream, e->slots[s.owner].done, 0);
            if (err != hipSuccess) {   // hand the slots back: later callers must not wait for a lease that never existed
                {
                    std::lock_guard<std::mutex> relk(e->mu);
                    for (int j = 0; j < n; ++j) e->slots[s0 + j].busy = false;
                }
                e->cv.notify_all();
                return fail(std::string("hipStreamWaitEvent: ") + hipGetErrorString(err));
            }
            waited = s.owner;
        }
    }
    out->e = e; out->s0 = s0; out->n = n; out->stream = stream;
    return 0;
}

void release(const Lease &l) {
    cart_engine *e = l.e;
    (void)hipEventRecord(e->slots[l.s0].done, l.stream);  // ONE in-queue marker per lease (16 of them cost ~80 us of GPU idle)
    for (int k = 0; k < l.n; ++k) {
        Slot &s = e->slots[l.s0 + k];
        s.owner = l.s0;
        s.last_stream = l.stream;
        s.used = true;
    }
    {
        std::lock_guard<std::mutex> lk(e->mu);
        const unsigned long long seq = ++e->release_counter;
        for (int k = 0; k < l.n; ++k) { e->slots[l.s0 + k].busy = false; e->slots[l.s0 + k].released_seq = seq; }
    }
    e->cv.notify_all();
}

void build_agg_args(cart_engine *e, AggArgs &a, unsigned keep) {
    const Geometry &g = e->g;
    a.g = g;
    // launch order: the long serial scans (horizontal, W steps) get the lowest block ids so they
    // start first; slab index `path` keeps the oracle's order {down, up, right, left, diagonals}.
    struct D { int dx, dy, path; };
    static const D order8[8] = {{1, 0, 2}, {-1, 0, 3}, {0, 1, 0}, {0, -1, 1}, {1, 1, 4}, {-1, 1, 5}, {-1, -1, 6}, {1, -1, 7}};
    const unsigned mask = keep;
    int blk = 0, nd = 0;
    const int lpb = agg_lines_per_block(g.D);
    for (int i = 0; i < g.P; ++i) {
        if (!((mask >> i) & 1u)) continue;
        DirDesc &d = a.dirs[nd++];
        d.dx = order8[i].dx; d.dy = order8[i].dy; d.path = order8[i].path;
        if (d.dy == 0) { d.nlines = g.h; d.jmin = 0; }
        else if (d.dx == 0) { d.nlines = g.w; d.jmin = 0; }
        else { d.nlines = g.w + g.h - 1; d.jmin = d.dx > 0 ? -(g.h - 1) : 0; }
        d.blk0 = blk;
        blk += (d.nlines + lpb - 1) / lpb;
    }
    a.ndirs = nd;
    a.blocks_per_frame = blk;
    a.cen_l = e->cen_l; a.cen_r = e->cen_r; a.slabs = SlabTable{};   // filled per launch
}

// The options a call works with: read once under the engine's mutex, so that a concurrent cart_engine_set_option
// cannot change the plan between the workspace allocation and the launches of one call.
struct Options {
    int plan, plan_min_frames, chunk_frames;
    bool timing;
    int spec;
};
Options snapshot_options(const cart_engine *e) { return Options{e->opt_plan, e->opt_plan_min_frames, e->chunk_frames, e->timing, e->opt_spec}; }   // caller holds e->mu

// The launch plan of `n` frames handed to one launch sequence (include/cart_engine.h, CART_PLAN_*).
int plan_for(const cart_engine *e, const Options &o, int n) {
    if (o.spec & 4) return CART_PLAN_SLABS;   // the S5 variant exists in the two-kernel WTA only
    if (o.plan == CART_PLAN_AUTO) return n >= e->auto_fused_min_frames ? CART_PLAN_FUSED_UP : CART_PLAN_SLABS;
    if (n < o.plan_min_frames) return CART_PLAN_SLABS;
    return o.plan;
}

int validate(const cart_engine_params *p) {
    if (!p) return fail("params is NULL");
    if (p->width < 16 || p->height < 8 || p->width > 16384 || p->height > 16384) return fail("unsupported image size");
    const bool post_only = p->num_disparities == 0 && p->paths == 0;
    if (!post_only && !(p->num_disparities == 64 || p->num_disparities == 128 || p->num_disparities == 256))
        return fail("num_disparities must be 64, 128 or 256");
    if (!post_only && !(p->paths == 4 || p->paths == 8)) return fail("paths must be 4 or 8");
    if (p->min_disparity < 0 || p->min_disparity > 64) return fail("min_disparity must be in [0, 64]");
    if (p->p1 < 0 || p->p2 < p->p1 || p->p2 + 31 > 255) return fail("need 0 <= p1 <= p2 and 31 + p2 <= 255");
    if (p->uniqueness_ratio < 0 || p->uniqueness_ratio > 100) return fail("uniqueness_ratio must be in [0, 100]");
    if (p->smoothing_radius > 8) return fail("smoothing_radius must be <= 8");
    if (p->max_inflight < 1 || p->max_inflight > 4096) return fail("max_inflight must be in [1, 4096]");
    return 0;
}

template <typename T>
int dev_alloc(T **p, size_t count) {
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(p), count * sizeof(T)));
    return 0;
}

// The slab workspace.  Measured on MI355X (profiles/r03_alloc.txt): the aggregation launch writes its slabs 8-9 % faster into a device
// allocation of at most 8 GiB than into a larger one (1.41-1.43 against 1.53-1.55 ms per 16 pairs at 1242x375 D=128 P=8; the L2's write
// requests to the fabric stall 20-30x as often in the larger one; TLB counters and clock are the same) -- whatever the physical layout
// rule behind it, it goes by the size of the PHYSICAL allocation.  So the workspace is cut into groups of slots, each group one plain
// hipMalloc of at most kSlabChunkBytes (a single slot larger than that gets an allocation of its own).  Round 3 kept one address range
// and backed it with several hipMemCreate / hipMemMap allocations instead; that path met two behaviours of ROCm 7.2's virtual-memory
// management that end in GPU memory access faults (profiles/r04_vmm_faults.txt: a hipMemSetAccess per mapping returns success and leaves
// the range inaccessible; a range re-reserved while other ranges are live did the same) and is gone: nothing in the engine calls
// hipMemAddressReserve / hipMemMap any more.
constexpr size_t kSlabChunkBytes = ((size_t)8 << 30) - ((size_t)64 << 20);
// cart_engine_tune_placement.  Each of the two launches has a fast and a slow level per placement: probed on a warmed-up GPU with real census planes the
// launch pair times at ~2.52-2.57 ms (both fast), ~2.63-2.65 (one slow) or ~2.75-2.78 (both slow) at the headline, and a set re-timed twelve times
// scatters by 0.4 % (profiles/r05_placement.txt sections 2, 5, 6; on an idle GPU straight after engine creation the extremes lie 12-13 % apart,
// which is where round 4's 0.87 came from).  A candidate replaces the kept set when it is 1.5 % faster (both timed back to back); the search stops at
// the first kept set 7 % under the slowest seen (both launches fast against both slow); the kept set is called fast when it is 5.5 % under the
// slowest seen; after kUniformAfter timed placements that are all within 1.5 % of each other the pool offers one kind only.
constexpr float kStopRatio = 0.930f, kFastRatio = 0.945f, kUniformRatio = 0.985f, kSwitchRatio = 0.985f;
constexpr int kUniformAfter = 6;

void slab_pool_free(SlabPool &sp) {
    for (uint8_t *b : sp.base)
        if (b) (void)hipFree(b);
    sp = SlabPool{};
}

// slots per group: as many as fit kSlabChunkBytes; from 16 up a multiple of kLaunchFrames, so that the default launch sequences of a
// call whose lease starts on a multiple of 16 stay inside one group
int slab_group_slots(size_t slot_bytes, int slots) {
    size_t g = std::max<size_t>(1, kSlabChunkBytes / std::max<size_t>(1, slot_bytes));
    if (g >= (size_t)kLaunchFrames) g = g / kLaunchFrames * kLaunchFrames;
    return (int)std::min<size_t>(g, (size_t)slots);
}

int slab_pool_alloc(SlabPool &sp, size_t slot_bytes, int slots) {
    sp = SlabPool{};
    sp.slot_bytes = slot_bytes;
    sp.slots = slots;
    sp.group_slots = slab_group_slots(slot_bytes, slots);
    const int ng = (slots + sp.group_slots - 1) / sp.group_slots;
    for (int gi = 0; gi < ng; ++gi) {
        uint8_t *b = nullptr;
        if (dev_alloc(&b, sp.bytes_of(gi))) { slab_pool_free(sp); return -1; }
        sp.base.push_back(b);
    }
    return 0;
}

// slab pointers of the n frames of one launch at slots [s0, s0 + n); `subst` (may be null) replaces the base of some groups (placement probes)
// A range that does not lie inside the pool (a caller passing a wrong s0 / n) yields an EMPTY table (frame[0] == nullptr) and the callers fail
// on the host: a launch must never see a pointer computed from a slot the pool does not hold (profiles/r04_vmm_faults.txt, fault 3).
SlabTable slab_table(const SlabPool &sp, int s0, int n, const std::vector<uint8_t *> *subst = nullptr) {
    SlabTable t{};
    if (s0 < 0 || n < 1 || n > kMaxLaunchFrames || s0 + n > sp.slots || sp.group_slots < 1) return t;
    for (int f = 0; f < n; ++f) {
        const int s = s0 + f, gi = s / sp.group_slots;
        uint8_t *b = subst && (*subst)[gi] ? (*subst)[gi] : sp.base[gi];
        t.frame[f] = b + (size_t)(s % sp.group_slots) * sp.slot_bytes;
    }
    return t;
}

}  // namespace

extern "C" {

void cart_engine_default_params(cart_engine_params *p) {
    if (!p) return;
    std::memset(p, 0, sizeof(*p));
    p->device_id = 0;
    p->min_disparity = 4;       // cartconfig.cpp:147
    p->num_disparities = 256;   // cartconfig.cpp:148
    p->paths = 4;               // cv::cuda::createStereoSGM default mode MODE_HH4
    p->p1 = 10; p->p2 = 120;    // cv::cuda::createStereoSGM defaults
    p->uniqueness_ratio = 12;   // disparity.hpp:32
    p->smoothing_radius = -1;   // cartconfig.cpp:150
    p->smoothing_iterations = 5;  // cartconfig.cpp:151
    p->max_inflight = 12;       // CARTSLAM_CONCURRENT_RUN_LIMIT, cartslam.hpp:4
}

int cart_engine_create(const cart_engine_params *params, cart_engine **out) {
    if (!out) return fail("out is NULL");
    *out = nullptr;
    if (validate(params)) return -1;
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (params->device_id < 0 || params->device_id >= ndev) return fail("device_id out of range (no such GPU)");
    HIP_TRY(hipSetDevice(params->device_id));
    cart_engine *e = new (std::nothrow) cart_engine();
    if (!e) return fail("out of host memory");
    e->params = *params;
    Geometry &g = e->g;
    g.w = params->width; g.h = params->height; g.D = params->num_disparities; g.P = params->paths;
    g.min_disp = params->min_disparity; g.p1 = params->p1; g.p2 = params->p2;
    g.cpadl = ((g.min_disp + g.D + 15) / 16) * 16;
    g.cpitch = ((g.cpadl + g.w + 16 + 15) / 16) * 16;
    g.npx = (size_t)g.w * g.h;
    g.census_elems = (size_t)g.h * g.cpitch;
    g.slab_bytes = g.npx * g.D;
    e->uniq = (float)(100 - params->uniqueness_ratio) / 100.0f;  // oracle S5
    const size_t n = (size_t)params->max_inflight;
    int rc = 0;
    e->post_only = g.D == 0;
    if (e->post_only) {  // geometry-only engine: interpolate ping-pong and CCL links are all the post stages need
        rc |= dev_alloc(&e->tmp_a, n * g.npx);
        rc |= dev_alloc(&e->tmp_b, n * g.npx);
        rc |= dev_alloc(&e->ccl_work, n * g.npx);
        if (rc) { cart_engine_destroy(e); return -1; }
        e->slots.resize(n);
        for (auto &s : e->slots)
            if (hipEventCreateWithFlags(&s.done, hipEventDisableTiming) != hipSuccess) { cart_engine_destroy(e); return fail("hipEventCreate failed"); }
        *out = e;
        return 0;
    }
    rc |= dev_alloc(&e->gray_l, n * g.npx);
    rc |= dev_alloc(&e->gray_r, n * g.npx);
    // the cooperative window loads of waves whose leading scan lines are still outside the image touch
    // addresses up to (h + D + min_disp + 64) features before / after a frame's census plane
    // and the software-pipelined prefetches run up to 3 rows past the first / last step
    e->cen_slack = (size_t)4 * g.cpitch + g.h + 1024;
    rc |= dev_alloc(&e->cen_l_alloc, n * g.census_elems + 2 * e->cen_slack);
    rc |= dev_alloc(&e->cen_r_alloc, n * g.census_elems + 2 * e->cen_slack);
    rc |= slab_pool_alloc(e->slab_pool, (size_t)g.P * g.slab_bytes, (int)n);
    rc |= dev_alloc(&e->wta_l, n * g.npx);
    rc |= dev_alloc(&e->right_pk, n * g.npx);
    rc |= dev_alloc(&e->tmp_a, n * g.npx);
    rc |= dev_alloc(&e->tmp_b, n * g.npx);
    rc |= dev_alloc(&e->ccl_work, n * g.npx);
    if (rc) { cart_engine_destroy(e); return -1; }
    e->cen_l = e->cen_l_alloc + e->cen_slack;
    e->cen_r = e->cen_r_alloc + e->cen_slack;
    // the census padding columns are never written again: out-of-image right features read as 0 (oracle S3)
    if (hipMemset(e->cen_l_alloc, 0, (n * g.census_elems + 2 * e->cen_slack) * 4) != hipSuccess ||
        hipMemset(e->cen_r_alloc, 0, (n * g.census_elems + 2 * e->cen_slack) * 4) != hipSuccess) {
        cart_engine_destroy(e);
        return fail("hipMemset of census workspace failed");
    }
    e->slots.resize(n);
    for (auto &s : e->slots)
        if (hipEventCreateWithFlags(&s.done, hipEventDisableTiming) != hipSuccess) {
            cart_engine_destroy(e);
            return fail("hipEventCreate failed");
        }
    // the WTA kernels' uniqueness test as a table: uniq_threshold() evaluated once per cost by the device function itself
    if (dev_alloc(&e->uniq_thr, 2048)) { cart_engine_destroy(e); return -1; }
    launch_uniq_table(e->uniq, e->uniq_thr, nullptr);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) { cart_engine_destroy(e); return fail("building the uniqueness table failed"); }
    build_agg_args(e, e->agg, 0xffu);
    build_agg_args(e, e->agg_fused, 0xffu & ~(1u << 3));  // launch-order slot 3 = {0,-1} = "up" (slab kFusedUpPath)
    // Fused WTA (the "up" direction computed inside the WTA sweep, 1/P less slab traffic): measured on MI355X at
    // 1242x375, batch 16 (profiles/tools/disparity_only.py): D=256 -9 % (4 paths) / -15 % (8 paths) per batch, D=128
    // even, D=64 +4..6 %; D=256 batches of 4 frames: +5 % at 1242x375, -11 % at 1920x1080 -- so it is the default for
    // D=256 batches that give the sweep enough workgroups.  Every plan gives the same bits; cart_engine_set_option
    // overrides the choice (tests and measurements), nothing is read from the environment.
    {   // from ~450 workgroups (16 columns each at D=256) the sweep fills the chip: 6 frames at 1242 px, 4 at 1920 px
        const int nblk = (g.w + 15) / 16;
        e->auto_fused_min_frames = g.D >= 256 ? std::max(2, (448 + nblk - 1) / nblk) : 1 << 30;
    }
    *out = e;
    return 0;
}

void cart_engine_destroy(cart_engine *e) {
    if (!e) return;
    (void)hipSetDevice(e->params.device_id);   // the caller's current device may be another one
    (void)hipDeviceSynchronize();
    slab_pool_free(e->slab_pool);
    void *bufs[] = {e->gray_l, e->gray_r, e->cen_l_alloc, e->cen_r_alloc, e->wta_l, e->right_pk, e->tmp_a, e->tmp_b, e->ccl_work, e->sp_votes, e->rv_partial, e->flow_ws, e->ccl_stats_ws, e->uniq_thr};
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    for (auto &s : e->slots) {
        if (s.done) (void)hipEventDestroy(s.done);
    }
    for (auto &r : e->ring)
        for (auto &ev : r.ev)
            if (ev) (void)hipEventDestroy(ev);
    delete e;
}

const char *cart_last_error(const cart_engine *) { return g_last_error.c_str(); }

const char *cart_engine_version(void) {
    static char buf[64];
    std::snprintf(buf, sizeof(buf), "cart_engine gfx950 %d kernels", kernel_count());
    return buf;
}

int cart_engine_set_option(cart_engine *e, int option, int value) {
    if (!e) return fail("engine is NULL");
    std::lock_guard<std::mutex> lk(e->mu);
    switch (option) {
        case CART_OPT_PLAN:
            if (value < CART_PLAN_AUTO || value > CART_PLAN_FUSED_UP) return fail("unknown plan");
            e->opt_plan = value;
            return 0;
        case CART_OPT_PLAN_MIN_FRAMES:
            if (value < 1) return fail("min frames must be >= 1");
            e->opt_plan_min_frames = value;
            return 0;
        case CART_OPT_CHUNK_FRAMES:
            if (value < 1 || value > kMaxLaunchFrames) return fail("chunk frames must be in [1, 64]");   // 64 = entries of the per-launch slab table (SlabTable); pointer-table (multi) calls stay at kLaunchFrames
            e->chunk_frames = value;
            return 0;
        case CART_OPT_SPEC_S8_ZERO_INVALID:
        case CART_OPT_SPEC_S7_REPLICATE_BORDER:
        case CART_OPT_SPEC_S5_TOP2: {
            if (value != 0 && value != 1) return fail("spec variants are 0 or 1");
            const int bit = option == CART_OPT_SPEC_S8_ZERO_INVALID ? 1 : option == CART_OPT_SPEC_S7_REPLICATE_BORDER ? 2 : 4;
            e->opt_spec = value ? (e->opt_spec | bit) : (e->opt_spec & ~bit);
            return 0;
        }
        default: return fail("unknown option");
    }
}

int cart_engine_get_option(cart_engine *e, int option, int *value) {
    if (!e || !value) return fail("bad arguments");
    std::lock_guard<std::mutex> lk(e->mu);
    switch (option) {
        case CART_OPT_PLAN: *value = e->opt_plan; return 0;
        case CART_OPT_PLAN_MIN_FRAMES: *value = e->opt_plan_min_frames; return 0;
        case CART_OPT_CHUNK_FRAMES: *value = e->chunk_frames; return 0;
        case CART_OPT_SPEC_S8_ZERO_INVALID: *value = (e->opt_spec & 1) ? 1 : 0; return 0;
        case CART_OPT_SPEC_S7_REPLICATE_BORDER: *value = (e->opt_spec & 2) ? 1 : 0; return 0;
        case CART_OPT_SPEC_S5_TOP2: *value = (e->opt_spec & 4) ? 1 : 0; return 0;
        default: return fail("unknown option");
    }
}

int cart_engine_describe_plan(cart_engine *e, int n_frames, cart_launch_plan *out) {
    if (!e || !out) return fail("bad arguments");
    if (e->post_only) return fail("this engine was created without SGM workspaces (num_disparities = 0)");
    if (n_frames < 1) return fail("n_frames must be positive");
    std::lock_guard<std::mutex> lk(e->mu);
    const Options o = snapshot_options(e);
    out->frames_per_launch = std::min(n_frames, o.chunk_frames);
    out->plan = plan_for(e, o, out->frames_per_launch);
    out->slabs_written = out->plan == CART_PLAN_FUSED_UP ? e->g.P - 1 : e->g.P;
    return 0;
}

namespace {
// Time of the aggregation + WTA launches of `n` frames at slots [s0, s0 + n) with their slabs at `slabs` (ms, best of three after one
// warm-up; the census planes hold whatever they hold: the cost of these launches does not depend on the data).  < 0 on error.
float probe_placement(cart_engine *e, const Options &opt, const SlabTable &slabs, size_t s0, int n, hipEvent_t ev0, hipEvent_t ev1) {
    const Geometry &g = e->g;
    if (!slabs.frame[0]) return -1.f;   // slot range outside the pool (slab_table)
    uint32_t *cl = e->cen_l + s0 * g.census_elems, *cr = e->cen_r + s0 * g.census_elems;
    uint16_t *wl = e->wta_l + s0 * g.npx;
    uint32_t *rpk = e->right_pk + s0 * g.npx;
    const bool fused = plan_for(e, opt, n) == CART_PLAN_FUSED_UP && e->rv_partial;
    float best = -1.f;
    for (int rep = 0; rep < 4; ++rep) {   // one warm-up (first touch of a fresh allocation), three timed: the fastest counts
        if (hipMemsetAsync(rpk, 0xff, (size_t)n * g.npx * sizeof(uint32_t), nullptr) != hipSuccess) return -1.f;   // what launch_census leaves there
        if (hipEventRecord(ev0, nullptr) != hipSuccess) return -1.f;
        AggArgs a = fused ? e->agg_fused : e->agg;
        a.cen_l = cl; a.cen_r = cr; a.slabs = slabs;
        launch_aggregate(a, n, nullptr);
        if (fused) launch_wta_fused(cl, cr, slabs, wl, rpk, e->rv_partial + s0 * wta_fused_partial_elems(g), g, e->uniq_thr, n, nullptr);
        else launch_wta(slabs, wl, rpk, g, e->uniq_thr, n, nullptr, false);
        if (hipEventRecord(ev1, nullptr) != hipSuccess || hipEventSynchronize(ev1) != hipSuccess || hipGetLastError() != hipSuccess) return -1.f;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ev0, ev1) != hipSuccess) return -1.f;
        if (rep && (best < 0.f || ms < best)) best = ms;
    }
    return best;
}
}  // namespace

int cart_engine_tune_placement(cart_engine *e, int n_frames, int max_tries, size_t max_extra_bytes, cart_placement_report *report) {
    if (report) std::memset(report, 0, sizeof(*report));
    if (!e) return fail("engine is NULL");
    if (e->post_only) return fail("this engine was created without SGM workspaces (num_disparities = 0)");
    if (n_frames < 1 || n_frames > (int)e->slots.size()) return fail("n_frames must be in [1, max_inflight]");
    HIP_TRY(hipSetDevice(e->params.device_id));
    std::unique_lock<std::mutex> lk(e->mu);
    for (const auto &sl : e->slots)
        if (sl.busy) return fail("cart_engine_tune_placement needs an idle engine");
    const Options opt = snapshot_options(e);
    HIP_TRY(hipDeviceSynchronize());
    const int n = std::min(n_frames, opt.chunk_frames);
    if (plan_for(e, opt, n) == CART_PLAN_FUSED_UP && !e->rv_partial)
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&e->rv_partial), e->slots.size() * wta_fused_partial_elems(e->g) * sizeof(uint32_t)));
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    HIP_TRY(hipEventCreate(&ev0));
    if (hipEventCreate(&ev1) != hipSuccess) { (void)hipEventDestroy(ev0); return fail("hipEventCreate failed"); }
    SlabPool &sp = e->slab_pool;
    // The search runs per UNIT = the groups behind the slots [k n, (k + 1) n) of one n-frame call, for the first (at most four) such
    // ranges: a lease takes the lowest free range its stream used last (acquire), so a caller with one call in flight lives in unit 0
    // and one with several walks up the units.  A group already settled by an earlier unit is not touched again.  A group that holds
    // more slots than the launch has frames (32-slot groups, 16-frame launches) is scored on the slots of its first launch only.
    const int units = std::max(1, std::min(4, (int)e->slots.size() / n));
    std::vector<char> settled((size_t)sp.groups(), 0);
    // Candidates that lost stay allocated while the search goes on (freed at once, their pages would come straight back from the
    // allocator), oldest first out when the byte cap is reached; everything is plain hipMalloc / hipFree.
    struct Held { uint8_t *p; size_t bytes; };
    std::vector<Held> held;
    size_t extra = 0;
    const auto t_begin = std::chrono::steady_clock::now();
    auto seconds_since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); };
    // Allocating tens of GB takes 0.1-0.6 s per candidate: no more than 0.25 s per allowed try + 1 s per 20 GB of workspace in all (the caller
    // buys search time with max_tries), and every unit gets its own share of that, so that unit 0 cannot spend what units 1-3 were promised.
    const double unit_budget = (0.25 * max_tries + (double)sp.slots * (double)sp.slot_bytes / 20e9) / units;
    double sum_first = 0.0, sum_kept = 0.0;
    int rc = 0, probed = 0, total_candidates = 0;
    for (int u = 0; u < units && rc == 0; ++u) {
        const auto t_unit = std::chrono::steady_clock::now();
        const int s0 = u * n, g0 = s0 / sp.group_slots, g1 = (s0 + n - 1) / sp.group_slots;
        std::vector<int> mine;
        size_t unit_bytes = 0;
        for (int gi = g0; gi <= g1; ++gi)
            if (!settled[(size_t)gi]) { mine.push_back(gi); unit_bytes += sp.bytes_of(gi); settled[(size_t)gi] = 1; }
        float kept = probe_placement(e, opt, slab_table(sp, s0, n), (size_t)s0, n, ev0, ev1);
        if (kept < 0.f) { rc = fail("placement probe failed"); break; }
        sum_first += kept;
        ++probed;
        // Every candidate is judged against the kept set RE-TIMED right after it (same clock, same temperature: the GPU's clock drifts by a few per
        // cent over a search, which is as much as the modes differ), and replaces it only when it is kSwitchRatio faster -- a search that follows
        // the probe's noise ends on a worse set than it started from as often as not (profiles/r05_placement.txt section 5).  `worst_rel` = the
        // slowest set seen, as a multiple of the kept one.
        float worst_rel = 1.f;
        int seen = 1, stop = mine.empty() ? CART_PLACE_STOP_NOTHING_TO_DO : CART_PLACE_STOP_TRIES;
        // at most this many bytes beyond the workspace at any time (0 = two units' worth); SIZE_MAX = whatever leaves 4 GiB free
        const size_t cap = max_extra_bytes ? max_extra_bytes : 2 * unit_bytes;
        for (int t = 1; t < max_tries && !mine.empty(); ++t) {
            // a kept placement 7 % under the slowest pair seen has both launches in their fast modes -- stop looking
            if (worst_rel * kStopRatio > 1.f) { stop = CART_PLACE_STOP_FAST_FOUND; break; }
            // A process in which no placement is fast (round 4's driver box: 64 candidates between 2.57 and 2.60 ms, 7.7 s of search for 1.4 %): once
            // kUniformAfter placements have been timed and the slowest is within 1.5 % of the kept one, there is nothing to find here -- stop.
            if (seen >= kUniformAfter && worst_rel * kUniformRatio < 1.f) { stop = CART_PLACE_STOP_UNIFORM; break; }
            if (seconds_since(t_unit) > unit_budget) { stop = CART_PLACE_STOP_TIME; break; }
            while (extra + unit_bytes > cap && !held.empty()) {   // make room under the cap: the oldest loser goes
                (void)hipFree(held.front().p);
                extra -= held.front().bytes;
                held.erase(held.begin());
            }
            if (extra + unit_bytes > cap) { stop = CART_PLACE_STOP_MEMORY; break; }
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < unit_bytes + ((size_t)4 << 30)) { stop = CART_PLACE_STOP_MEMORY; break; }   // no room for another candidate
            std::vector<uint8_t *> cand((size_t)sp.groups(), nullptr);
            bool ok = true;
            for (int gi : mine)
                if (dev_alloc(&cand[(size_t)gi], sp.bytes_of(gi))) { ok = false; break; }
            if (!ok) {
                (void)hipGetLastError();
                for (int gi : mine) if (cand[(size_t)gi]) (void)hipFree(cand[(size_t)gi]);
                stop = CART_PLACE_STOP_MEMORY;
                break;
            }
            extra += unit_bytes;
            const float sc = probe_placement(e, opt, slab_table(sp, s0, n, &cand), (size_t)s0, n, ev0, ev1);
            const float again = sc < 0.f ? -1.f : probe_placement(e, opt, slab_table(sp, s0, n), (size_t)s0, n, ev0, ev1);
            if (sc < 0.f || again < 0.f) {
                for (int gi : mine) (void)hipFree(cand[(size_t)gi]);
                extra -= unit_bytes;
                rc = fail("placement probe failed");
                break;
            }
            ++seen;
            const float rel = sc / again;   // the candidate as a multiple of the kept set, both timed now
            const bool better = rel < kSwitchRatio;
            for (int gi : mine) {   // the loser of every group joins the held list
                uint8_t *lose = better ? sp.base[(size_t)gi] : cand[(size_t)gi];
                if (better) sp.base[(size_t)gi] = cand[(size_t)gi];
                held.push_back(Held{lose, sp.bytes_of(gi)});
            }
            if (better) { worst_rel = std::max(worst_rel / rel, 1.f / rel); kept = sc; }   // everything seen so far, the old kept set included, relative to the new one
            else { worst_rel = std::max(worst_rel, rel); kept = again; }
        }
        if (stop == CART_PLACE_STOP_TRIES && worst_rel * kStopRatio > 1.f) stop = CART_PLACE_STOP_FAST_FOUND;   // the last allowed try was the fast one
        const float worst = kept * worst_rel;
        sum_kept += kept;
        total_candidates += seen;
        if (u == 0 && report && rc == 0) {   // the unit a caller with one call in flight lives in
            report->stop_reason = stop;
            report->mode = worst_rel * kFastRatio > 1.f ? CART_PLACE_MODE_FAST
                         : (seen >= kUniformAfter && worst_rel * kUniformRatio < 1.f) ? CART_PLACE_MODE_UNIFORM
                         : seen == 1 ? CART_PLACE_MODE_UNKNOWN : CART_PLACE_MODE_MIXED;
            report->ms_fastest_seen = kept;
            report->ms_slowest_seen = worst;
        }
    }
    (void)hipDeviceSynchronize();
    for (auto &h : held) (void)hipFree(h.p);
    (void)hipEventDestroy(ev0); (void)hipEventDestroy(ev1);
    if (probed && rc == 0 && report) {   // mean over the probed units, before and after
        report->ms_first = (float)(sum_first / probed);
        report->ms_kept = (float)(sum_kept / probed);
        report->units = probed;
        report->candidates = total_candidates;
        report->seconds = (float)seconds_since(t_begin);
    }
    return rc;
}

int cart_engine_set_timing(cart_engine *e, int enabled) {
    if (!e) return fail("engine is NULL");
    std::lock_guard<std::mutex> lk(e->mu);
    if (enabled && e->ring.empty()) {
        e->ring.resize(kTimingRing);
        for (auto &r : e->ring)
            for (auto &ev : r.ev)
                if (hipEventCreate(&ev) != hipSuccess) return fail("hipEventCreate failed");
    }
    for (auto &r : e->ring) r.n = 0;
    e->ring_calls = 0;
    e->timing = enabled != 0;
    e->timing_every = enabled > 1 ? enabled : 1;
    e->timing_calls = 0;
    return 0;
}

int cart_engine_collect_timing(cart_engine *e, const char **names, float *mean_ms, int cap, int *n_calls) {
    if (!e || !names || !mean_ms) return fail("bad arguments");
    HIP_TRY(hipSetDevice(e->params.device_id));
    HIP_TRY(hipDeviceSynchronize());
    std::lock_guard<std::mutex> lk(e->mu);
    int nstages = 0, calls = 0;
    double sum[kMaxTimings] = {};
    for (auto &r : e->ring) {
        if (r.n == 0) continue;
        if (nstages == 0) { nstages = r.n; for (int i = 0; i < r.n; ++i) names[i < cap ? i : 0] = r.names[i]; }
        if (r.n != nstages) continue;
        for (int i = 0; i < r.n; ++i) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, r.ev[i], r.ev[i + 1]) != hipSuccess) return fail("hipEventElapsedTime failed");
            sum[i] += ms;
        }
        ++calls;
    }
    const int n = std::min(cap, nstages);
    for (int i = 0; i < n; ++i) mean_ms[i] = calls ? (float)(sum[i] / calls) : 0.f;
    if (n_calls) *n_calls = calls;
    return n;
}

#define STAGE(name)                                                        \
    do {                                                                   \
        if (rec && nt < kMaxTimings) {                                     \
            (void)hipEventRecord(rec->ev[nt], stream);                     \
            rec->names[nt] = name;                                         \
            ++nt;                                                          \
        }                                                                  \
    } while (0)

namespace {
// Where the frames of one call live: a base + stride per image (the batch entry point) or one pointer per frame (multi).
struct FrameSet {
    const uint8_t *left, *right; size_t left_fs, right_fs;
    int16_t *out; size_t out_fs;
    const uint8_t *const *lefts, *const *rights; int16_t *const *outs;   // non-NULL: scattered frames
    size_t left_step, right_step, out_step;
    ImageBatch images(bool right_side, int f0, int n) const {
        const size_t step = right_side ? right_step : left_step;
        if (!lefts) return strided_images((right_side ? right : left) + (size_t)f0 * (right_side ? right_fs : left_fs), step, right_side ? right_fs : left_fs);
        ImageBatch b{}; b.step = step; b.scattered = 1;
        for (int f = 0; f < n; ++f) b.frames[f] = (right_side ? rights : lefts)[f0 + f];
        return b;
    }
    OutBatch output(int f0, int n) const {
        if (!outs) return strided_out(reinterpret_cast<int16_t *>(reinterpret_cast<uint8_t *>(out) + (size_t)f0 * out_fs), out_step, out_fs);
        OutBatch b{}; b.step = out_step; b.scattered = 1;
        for (int f = 0; f < n; ++f) b.frames[f] = outs[f0 + f];
        return b;
    }
};

int compute_disparity_impl(cart_engine *e, int n_frames, const FrameSet &fr, int channels, void *stream_) {
    if (!e) return fail("engine is NULL");
    if (e->post_only) return fail("this engine was created without SGM workspaces (num_disparities = 0)");
    if (channels != 1 && channels != 3) return fail("channels must be 1 (gray) or 3 (BGR)");
    const Geometry &g = e->g;
    if (fr.left_step < (size_t)g.w * channels || fr.right_step < (size_t)g.w * channels) return fail("input step smaller than a row");
    if (fr.out_step < (size_t)g.w * 2 || (fr.out_step & 1)) return fail("out_step must be even and >= 2*width");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    HIP_TRY(hipSetDevice(e->params.device_id));
    Options opt;
    {
        std::lock_guard<std::mutex> lk(e->mu);
        opt = snapshot_options(e);
        const int launch_plan = plan_for(e, opt, std::min(n_frames, opt.chunk_frames));   // later (shorter) launches of the call never need more
        if (launch_plan == CART_PLAN_FUSED_UP && !e->rv_partial)
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&e->rv_partial), e->slots.size() * wta_fused_partial_elems(g) * sizeof(uint32_t)));
    }
    Lease l;
    if (acquire(e, n_frames, stream, &l)) return -1;
    g_last_slot = l.s0;
    TimingRec *rec = nullptr;
    if (opt.timing) {
        std::lock_guard<std::mutex> lk(e->mu);
        if (!e->ring.empty() && e->timing_calls++ % (unsigned long long)e->timing_every == 0) { rec = &e->ring[e->ring_calls++ % kTimingRing]; rec->n = 0; }
    }
    int nt = 0;
    const int radius = e->params.smoothing_radius, iters = e->params.smoothing_iterations;
    const bool smooth = radius > 0 && iters > 0;  // disparity.cu:73
    const size_t tight_step = (size_t)g.w * 2, tight_fs = g.npx * 2;
    // Enqueues every stage for frames [f0, f0+n) of this call on stream `st`.
    bool bad_slots = false;
    auto enqueue = [&](int f0, int n, hipStream_t st, bool timed) {
        const size_t s0 = (size_t)l.s0 + f0;
        if (!slab_table(e->slab_pool, (int)s0, n).frame[0]) { bad_slots = true; return; }   // nothing is launched on a slot range the pool does not hold
        uint8_t *gl = e->gray_l + s0 * g.npx, *gr = e->gray_r + s0 * g.npx;
        uint32_t *cl = e->cen_l + s0 * g.census_elems, *cr = e->cen_r + s0 * g.census_elems;
        const SlabTable slabs = slab_table(e->slab_pool, (int)s0, n);
        uint16_t *wl = e->wta_l + s0 * g.npx;
        uint32_t *rpk = e->right_pk + s0 * g.npx;
        int16_t *ta = e->tmp_a + s0 * g.npx, *tb = e->tmp_b + s0 * g.npx;
        const OutBatch o = fr.output(f0, n);
        hipStream_t stream = st;  // STAGE records on the stream the kernels are launched on
        TimingRec *rec_save = rec;
        if (!timed) rec = nullptr;
        STAGE("census");
        launch_census(fr.images(false, f0, n), fr.images(true, f0, n), channels, n, gl, gr, cl, cr, rpk, g, st);
        const int launch_plan = plan_for(e, opt, n);
        const bool fused = launch_plan == CART_PLAN_FUSED_UP && e->rv_partial;
        STAGE("aggregate");
        AggArgs a = fused ? e->agg_fused : e->agg;
        a.cen_l = cl; a.cen_r = cr; a.slabs = slabs;
        launch_aggregate(a, n, st);
        STAGE("wta");
        if (fused) launch_wta_fused(cl, cr, slabs, wl, rpk, e->rv_partial + s0 * wta_fused_partial_elems(g), g, e->uniq_thr, n, st);
        else launch_wta(slabs, wl, rpk, g, e->uniq_thr, n, st, (opt.spec & 4) != 0);
        STAGE("post");
        // disparity.hpp:27-28: minDisparity = cfg*16, maxDisparity = image width (not x16)
        const int min16 = e->params.min_disparity * 16, maxd = g.w;
        if (!smooth) {
            launch_post(wl, rpk, gl, o, g, n, st, opt.spec);
        } else {
            int16_t *src = ta, *dst = tb;
            int it = 0;
            if (post_interp_fusable(radius, min16, maxd)) {   // the first pass rides on the post stage: one launch, no intermediate image
                launch_post_interp(wl, rpk, gl, iters == 1 ? o : strided_out(ta, tight_step, tight_fs), g, n, st, opt.spec, min16, maxd);
                it = 1;
            } else {
                launch_post(wl, rpk, gl, strided_out(ta, tight_step, tight_fs), g, n, st, opt.spec);
            }
            if (it < iters) STAGE("interpolate");
            for (; it < iters; ++it) {
                const bool last = it == iters - 1;
                launch_interpolate(src, tight_step, tight_fs, last ? o : strided_out(dst, tight_step, tight_fs), g.w, g.h, radius, min16, maxd, n, st);
                std::swap(src, dst);
            }
        }
        if (rec) {
            (void)hipEventRecord(rec->ev[nt], st);
            std::lock_guard<std::mutex> lk(e->mu);
            rec->n = nt;
        }
        rec = rec_save;
    };
    // Large batches run as cache-sized sub-batches on the caller's stream: the census planes every direction
    // re-reads (4.2 MB per frame) then stay in L2 + Infinity Cache (measured: 64 frames in one launch are 13 %
    // slower per frame than 4 x 16).  Two-stream overlap of sub-batches was measured and buys nothing.
    const int chunk = fr.lefts ? std::min(opt.chunk_frames, kLaunchFrames) : opt.chunk_frames;  // pointer tables hold kLaunchFrames entries
    for (int f0 = 0; f0 < n_frames; f0 += chunk) enqueue(f0, std::min(chunk, n_frames - f0), stream, f0 == 0);
    hipError_t err = hipGetLastError();
    release(l);
    if (bad_slots) return fail("internal error: a launch's slot range lies outside the slab pool");
    if (err != hipSuccess) return fail(std::string("kernel launch failed: ") + hipGetErrorString(err));
    return 0;
}
}  // namespace

int cart_compute_disparity_batch(cart_engine *e, int n_frames, const uint8_t *left, size_t left_step,
                                 size_t left_frame_stride, const uint8_t *right, size_t right_step,
                                 size_t right_frame_stride, int channels, int16_t *out, size_t out_step,
                                 size_t out_frame_stride, void *stream) {
    if (!left || !right || !out) return fail("NULL image pointer");
    if (out_frame_stride & 1) return fail("out_frame_stride must be even");
    FrameSet fr{};
    fr.left = left; fr.right = right; fr.left_fs = left_frame_stride; fr.right_fs = right_frame_stride; fr.out = out; fr.out_fs = out_frame_stride;
    fr.left_step = left_step; fr.right_step = right_step; fr.out_step = out_step;
    return compute_disparity_impl(e, n_frames, fr, channels, stream);
}

int cart_compute_disparity_multi(cart_engine *e, int n_frames, const uint8_t *const *left, size_t left_step,
                                 const uint8_t *const *right, size_t right_step, int channels, int16_t *const *out,
                                 size_t out_step, void *stream) {
    if (!left || !right || !out) return fail("NULL pointer table");
    for (int f = 0; f < n_frames; ++f) {
        if (!left[f] || !right[f] || !out[f]) return fail("NULL image pointer in a pointer table");
        if (reinterpret_cast<uintptr_t>(out[f]) & 1) return fail("output images must be 2-byte aligned");
    }
    FrameSet fr{};
    fr.lefts = left; fr.rights = right; fr.outs = out;
    fr.left_step = left_step; fr.right_step = right_step; fr.out_step = out_step;
    return compute_disparity_impl(e, n_frames, fr, channels, stream);
}

int cart_compute_disparity(cart_engine *e, const uint8_t *left, size_t left_step, const uint8_t *right,
                           size_t right_step, int channels, int16_t *out, size_t out_step, void *stream) {
    return cart_compute_disparity_batch(e, 1, left, left_step, 0, right, right_step, 0, channels, out, out_step, 0, stream);
}

int cart_interpolate(cart_engine *e, int n_frames, int16_t *disp, size_t step, size_t frame_stride, int radius,
                     int iterations, int min_disp16, int max_disp, void *stream_) {
    if (!e) return fail("engine is NULL");
    if (!disp) return fail("NULL image pointer");
    if (radius <= 0 || iterations <= 0) return 0;
    if (radius > 8) return fail("radius must be <= 8");
    const Geometry &g = e->g;
    if (step < (size_t)g.w * 2 || (step & 1) || (frame_stride & 1)) return fail("bad step");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    HIP_TRY(hipSetDevice(e->params.device_id));
    Lease l;
    if (acquire(e, n_frames, stream, &l)) return -1;
    int16_t *ta = e->tmp_a + (size_t)l.s0 * g.npx, *tb = e->tmp_b + (size_t)l.s0 * g.npx;
    const size_t ts = (size_t)g.w * 2, tfs = g.npx * 2;
    // pass 0 reads the caller's buffer, the last pass writes it; an extra tight copy keeps Jacobi semantics
    launch_interpolate(disp, step, frame_stride, strided_out(ta, ts, tfs), g.w, g.h, radius, min_disp16, max_disp, n_frames, stream);
    int16_t *src = ta, *dst = tb;
    for (int it = 1; it < iterations; ++it) {
        launch_interpolate(src, ts, tfs, strided_out(dst, ts, tfs), g.w, g.h, radius, min_disp16, max_disp, n_frames, stream);
        std::swap(src, dst);
    }
    hipError_t err = hipSuccess;
    for (int f = 0; f < n_frames && err == hipSuccess; ++f)
        err = hipMemcpy2DAsync(reinterpret_cast<uint8_t *>(disp) + (size_t)f * frame_stride, step, src + (size_t)f * g.npx, ts, ts,
                               g.h, hipMemcpyDeviceToDevice, stream);
    if (err == hipSuccess) err = hipGetLastError();
    release(l);
    if (err != hipSuccess) return fail(std::string("interpolate failed: ") + hipGetErrorString(err));
    return 0;
}

int cart_disparity_derivative(cart_engine *e, int n_frames, const int16_t *disp, size_t disp_step,
                              size_t disp_frame_stride, int16_t *out, size_t out_step, size_t out_frame_stride,
                              int32_t *hist512, void *stream_) {
    if (!e) return fail("engine is NULL");
    if (!disp || !out || !hist512) return fail("NULL pointer");
    if (n_frames <= 0) return fail("n_frames must be positive");
    const Geometry &g = e->g;
    if (disp_step < (size_t)g.w * 2 || out_step < (size_t)g.w * 4 || (out_step & 3) || (out_frame_stride & 3) || (disp_step & 1) || (disp_frame_stride & 1))
        return fail("bad step (derivative rows must be 4-byte aligned)");
    HIP_TRY(hipSetDevice(e->params.device_id));
    launch_dir_derivative(disp, disp_step, disp_frame_stride, out, out_step, out_frame_stride, hist512, g.w, g.h, n_frames,
                          static_cast<hipStream_t>(stream_));
    HIP_TRY(hipGetLastError());
    return 0;
}

int cart_plane_derivative_hist(cart_engine *e, int n_frames, const int16_t *disp, size_t disp_step,
                               size_t disp_frame_stride, int16_t *out, size_t out_step, size_t out_frame_stride,
                               int32_t *hist256, size_t hist_frame_stride_elems, void *stream_) {
    if (!e) return fail("engine is NULL");
    if (!disp || !out || !hist256) return fail("NULL pointer");
    if (n_frames <= 0) return fail("n_frames must be positive");
    const Geometry &g = e->g;
    if (disp_step < (size_t)g.w * 2 || out_step < (size_t)g.w * 2 || (disp_step & 1) || (out_step & 1) || (disp_frame_stride & 1) || (out_frame_stride & 1))
        return fail("bad step");
    HIP_TRY(hipSetDevice(e->params.device_id));
    launch_plane_derivative(disp, disp_step, disp_frame_stride, out, out_step, out_frame_stride, hist256,
                            hist_frame_stride_elems, g.w, g.h, n_frames, static_cast<hipStream_t>(stream_));
    HIP_TRY(hipGetLastError());
    return 0;
}

int cart_plane_classify(cart_engine *e, int n_frames, const int16_t *deriv, size_t deriv_step, size_t deriv_frame_stride,
                        const cart_plane_params *params, int params_per_frame, uint8_t *planes, size_t planes_step,
                        size_t planes_frame_stride, void *stream_) {
    if (!e) return fail("engine is NULL");
    if (!deriv || !planes || !params) return fail("NULL pointer");
    if (n_frames <= 0) return fail("n_frames must be positive");
    const Geometry &g = e->g;
    if (deriv_step < (size_t)g.w * 2 || planes_step < (size_t)g.w || (deriv_step & 1) || (deriv_frame_stride & 1)) return fail("bad step");
    HIP_TRY(hipSetDevice(e->params.device_id));
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    for (int f0 = 0; f0 < n_frames; f0 += kMaxBatchArgs) {
        const int n = std::min(kMaxBatchArgs, n_frames - f0);
        ClassifyParams cp;
        if (params_per_frame) std::memcpy(cp.p, params + f0, sizeof(cart_plane_params) * (size_t)n);
        else cp.p[0] = params[0];
        launch_classify(reinterpret_cast<const int16_t *>(reinterpret_cast<const uint8_t *>(deriv) + (size_t)f0 * deriv_frame_stride),
                        deriv_step, deriv_frame_stride, cp, params_per_frame, planes + (size_t)f0 * planes_frame_stride,
                        planes_step, planes_frame_stride, g.w, g.h, n, stream);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

int cart_plane_derivative_hist_multi(cart_engine *e, int n_frames, const int16_t *const *disp, size_t disp_step, int16_t *const *out,
                                     size_t out_step, int32_t *hist256, size_t hist_frame_stride_elems, void *stream_) {
    if (!e) return fail("engine is NULL");
    if (!disp || !out || !hist256) return fail("NULL pointer");
    if (n_frames <= 0) return fail("n_frames must be positive");
    const Geometry &g = e->g;
    if (disp_step < (size_t)g.w * 2 || out_step < (size_t)g.w * 2 || (disp_step & 1) || (out_step & 1)) return fail("bad step");
    for (int f = 0; f < n_frames; ++f)
        if (!disp[f] || !out[f]) return fail("NULL image pointer in a pointer table");
    HIP_TRY(hipSetDevice(e->params.device_id));
    for (int f0 = 0; f0 < n_frames; f0 += kLaunchFrames) {
        const int n = std::min(kLaunchFrames, n_frames - f0);
        FrameTable dt{}, ot{};
        dt.scattered = ot.scattered = 1;
        for (int f = 0; f < n; ++f) { dt.p[f] = disp[f0 + f]; ot.p[f] = out[f0 + f]; }
        launch_plane_derivative(nullptr, disp_step, 0, nullptr, out_step, 0, hist256 + (size_t)f0 * hist_frame_stride_elems, hist_frame_stride_elems, g.w, g.h, n,
                                static_cast<hipStream_t>(stream_), &dt, &ot);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

int cart_plane_classify_multi(cart_engine *e, int n_frames, const int16_t *const *deriv, size_t deriv_step, const cart_plane_params *params,
                              int params_per_frame, uint8_t *const *planes, size_t planes_step, void *stream_) {
    if (!e) return fail("engine is NULL");
    if (!deriv || !planes || !params) return fail("NULL pointer");
    if (n_frames <= 0) return fail("n_frames must be positive");
    const Geometry &g = e->g;
    if (deriv_step < (size_t)g.w * 2 || planes_step < (size_t)g.w || (deriv_step & 1)) return fail("bad step");
    for (int f = 0; f < n_frames; ++f)
        if (!deriv[f] || !planes[f]) return fail("NULL image pointer in a pointer table");
    HIP_TRY(hipSetDevice(e->params.device_id));
    static_assert(kMaxBatchArgs >= kLaunchFrames, "one ClassifyParams covers a launch");
    for (int f0 = 0; f0 < n_frames; f0 += kLaunchFrames) {
        const int n = std::min(kLaunchFrames, n_frames - f0);
        ClassifyParams cp;
        if (params_per_frame) std::memcpy(cp.p, params + f0, sizeof(cart_plane_params) * (size_t)n);
        else cp.p[0] = params[0];
        FrameTable dt{}, pt{};
        dt.scattered = pt.scattered = 1;
        for (int f = 0; f < n; ++f) { dt.p[f] = deriv[f0 + f]; pt.p[f] = planes[f0 + f]; }
        launch_classify(nullptr, deriv_step, 0, cp, params_per_frame, nullptr, planes_step, 0, g.w, g.h, n, static_cast<hipStream_t>(stream_), &dt, &pt);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

int cart_plane_ccl(cart_engine *e, int n_frames, const uint8_t *planes, size_t planes_step, size_t planes_frame_stride,
                   int32_t *ids, size_t ids_step, size_t ids_frame_stride, int32_t *n_components, void *stream_) {
    if (!e) return fail("engine is NULL");
    if (!planes || !ids) return fail("NULL pointer");
    const Geometry &g = e->g;
    if (planes_step < (size_t)g.w || ids_step < (size_t)g.w * 4 || (ids_step & 3) || (ids_frame_stride & 3)) return fail("bad step");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    HIP_TRY(hipSetDevice(e->params.device_id));
    Lease l;
    if (acquire(e, n_frames, stream, &l)) return -1;
    launch_ccl(planes, planes_step, planes_frame_stride, e->ccl_work + (size_t)l.s0 * g.npx, ids, ids_step, ids_frame_stride,
               n_components, g.w, g.h, n_frames, stream);
    hipError_t err = hipGetLastError();
    release(l);
    if (err != hipSuccess) return fail(std::string("ccl failed: ") + hipGetErrorString(err));
    return 0;
}

namespace {
// The component-table workspace: [slots][npx][5] statistics scratch, then [slots][h][tile columns] root counts (post_kernels.hip).  The
// scratch is zeroed ONCE, here: every call returns it to zero (ccl_table_kernel collects and clears exactly the entries the call grew).
int ensure_ccl_stats_ws(cart_engine *e) {
    std::lock_guard<std::mutex> lk(e->mu);
    if (e->ccl_stats_ws) return 0;
    const size_t bytes = e->slots.size() * ccl_stats_ws_ints(e->g.w, e->g.h) * sizeof(int32_t);
    int32_t *ws = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&ws), bytes));
    if (hipMemset(ws, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { (void)hipFree(ws); return fail("hipMemset of the component-table workspace failed"); }
    e->ccl_stats_ws = ws;
    return 0;
}
int32_t *ccl_stat_of(cart_engine *e, int slot) { return e->ccl_stats_ws + (size_t)slot * e->g.npx * 5; }
int32_t *ccl_seg_of(cart_engine *e, int slot) {
    return e->ccl_stats_ws + e->slots.size() * e->g.npx * 5 + (size_t)slot * (ccl_stats_ws_ints(e->g.w, e->g.h) - e->g.npx * 5);
}
}  // namespace

int cart_plane_ccl_stats(cart_engine *e, int n_frames, const uint8_t *planes, size_t planes_step, size_t planes_frame_stride,
                         const int32_t *ids, size_t ids_step, size_t ids_frame_stride, cart_component *table, int max_components,
                         int32_t *n_components, void *stream_) {
    if (!e) return fail("engine is NULL");
    if (!planes || !ids || !table) return fail("NULL pointer");
    if (max_components < 1) return fail("max_components must be positive");
    const Geometry &g = e->g;
    if (planes_step < (size_t)g.w || ids_step < (size_t)g.w * 4 || (ids_step & 3) || (ids_frame_stride & 3)) return fail("bad step");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    HIP_TRY(hipSetDevice(e->params.device_id));
    if (ensure_ccl_stats_ws(e)) return -1;
    Lease l;
    if (acquire(e, n_frames, stream, &l)) return -1;
    launch_ccl_stats(planes, planes_step, planes_frame_stride, ids, ids_step, ids_frame_stride, ccl_stat_of(e, l.s0), ccl_seg_of(e, l.s0), table, max_components,
                     n_components, g.w, g.h, n_frames, stream);
    hipError_t err = hipGetLastError();
    release(l);
    if (err != hipSuccess) return fail(std::string("ccl stats failed: ") + hipGetErrorString(err));
    return 0;
}

int cart_plane_ccl_table(cart_engine *e, int n_frames, const uint8_t *planes, size_t planes_step, size_t planes_frame_stride, int32_t *ids, size_t ids_step,
                         size_t ids_frame_stride, cart_component *table, int max_components, int32_t *n_components, void *stream_) {
    if (!e) return fail("engine is NULL");
    if (!planes || !ids || !table) return fail("NULL pointer");
    if (max_components < 1) return fail("max_components must be positive");
    const Geometry &g = e->g;
    if (planes_step < (size_t)g.w || ids_step < (size_t)g.w * 4 || (ids_step & 3) || (ids_frame_stride & 3)) return fail("bad step");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    HIP_TRY(hipSetDevice(e->params.device_id));
    if (ensure_ccl_stats_ws(e)) return -1;
    Lease l;
    if (acquire(e, n_frames, stream, &l)) return -1;
    launch_ccl(planes, planes_step, planes_frame_stride, e->ccl_work + (size_t)l.s0 * g.npx, ids, ids_step, ids_frame_stride, n_components, g.w, g.h, n_frames, stream,
               ccl_stat_of(e, l.s0), ccl_seg_of(e, l.s0), table, max_components);
    hipError_t err = hipGetLastError();
    release(l);
    if (err != hipSuccess) return fail(std::string("ccl failed: ") + hipGetErrorString(err));
    return 0;
}

struct cart_plane_schedule {
    int device_id;         // the engine's; kept here so that the schedule can outlive the engine it was created on
    ScheduleState *state;  // device
    int provider, update_interval, reset_interval;
};

int cart_plane_schedule_create(cart_engine *e, int provider, const cart_plane_params *initial, int update_interval,
                               int reset_interval, cart_plane_schedule **out) {
    if (!e || !out) return fail("bad arguments");
    if (provider != 0 && provider != 1) return fail("Unknown parameter provider type.");
    if (update_interval < 1 || reset_interval < 1) return fail("intervals must be >= 1");
    HIP_TRY(hipSetDevice(e->params.device_id));
    cart_plane_schedule *s = new (std::nothrow) cart_plane_schedule{e->params.device_id, nullptr, provider, update_interval, reset_interval};
    if (!s) return fail("out of host memory");
    ScheduleState init;
    std::memset(&init, 0, sizeof(init));
    if (initial) init.params = *initial;
    if (hipMalloc(reinterpret_cast<void **>(&s->state), sizeof(ScheduleState)) != hipSuccess ||
        hipMemcpy(s->state, &init, sizeof(init), hipMemcpyHostToDevice) != hipSuccess) {
        cart_plane_schedule_destroy(s);
        return fail("hipMalloc/hipMemcpy of the schedule state failed");
    }
    *out = s;
    return 0;
}

void cart_plane_schedule_destroy(cart_plane_schedule *s) {
    if (!s) return;
    (void)hipSetDevice(s->device_id);   // the caller's current device may be another one
    if (s->state) (void)hipFree(s->state);
    delete s;
}

int cart_plane_schedule_advance(cart_plane_schedule *s, int first_id, int n_frames, const int32_t *hists,
                                cart_plane_params *params_out, void *stream) {
    if (!s || !hists || !params_out) return fail("bad arguments");
    if (n_frames <= 0 || first_id < 1) return fail("n_frames must be positive and ids are 1-based");
    HIP_TRY(hipSetDevice(s->device_id));
    launch_plane_schedule(s->state, s->provider, first_id, n_frames, s->update_interval, s->reset_interval, hists, params_out,
                          static_cast<hipStream_t>(stream));
    HIP_TRY(hipGetLastError());
    return 0;
}

int cart_plane_schedule_read(cart_plane_schedule *s, cart_plane_params *params_host, int32_t cum_hist_host[256]) {
    if (!s) return fail("bad arguments");
    HIP_TRY(hipSetDevice(s->device_id));
    HIP_TRY(hipDeviceSynchronize());
    ScheduleState st;
    HIP_TRY(hipMemcpy(&st, s->state, sizeof(st), hipMemcpyDeviceToHost));
    if (params_host) *params_host = st.params;
    if (cum_hist_host) std::memcpy(cum_hist_host, st.cum, sizeof(st.cum));
    return 0;
}

int cart_plane_classify_dev(cart_engine *e, int n_frames, const int16_t *deriv, size_t deriv_step, size_t deriv_frame_stride,
                            const cart_plane_params *params_dev, int params_stride, uint8_t *planes, size_t planes_step,
                            size_t planes_frame_stride, void *stream_) {
    if (!e) return fail("engine is NULL");
    if (!deriv || !planes || !params_dev) return fail("NULL pointer");
    if (n_frames <= 0) return fail("n_frames must be positive");
    const Geometry &g = e->g;
    if (deriv_step < (size_t)g.w * 2 || planes_step < (size_t)g.w || (deriv_step & 1) || (deriv_frame_stride & 1)) return fail("bad step");
    HIP_TRY(hipSetDevice(e->params.device_id));
    launch_classify_dev(deriv, deriv_step, deriv_frame_stride, params_dev, params_stride ? 1 : 0, planes, planes_step, planes_frame_stride,
                        g.w, g.h, n_frames, static_cast<hipStream_t>(stream_));
    HIP_TRY(hipGetLastError());
    return 0;
}

int cart_plane_temporal_vote(cart_engine *e, const uint8_t *planes, size_t planes_step, int n_prev, const uint8_t *const *prev_planes,
                             const size_t *prev_steps, const int16_t *const *flows, const size_t *flow_steps, uint8_t *smoothed,
                             size_t smoothed_step, void *stream_) {
    if (!e) return fail("engine is NULL");
    if (!planes || !smoothed) return fail("NULL pointer");
    if (n_prev < 0 || n_prev > CART_MAX_TEMPORAL) return fail("n_prev must be in [0, CART_MAX_TEMPORAL]");
    if (n_prev > 0 && (!prev_planes || !prev_steps || !flows || !flow_steps)) return fail("NULL table");
    const Geometry &g = e->g;
    if (planes_step < (size_t)g.w || smoothed_step < (size_t)g.w) return fail("bad step");
    TemporalArgs t;
    std::memset(&t, 0, sizeof(t));
    t.n_prev = n_prev;
    for (int k = 0; k < n_prev; ++k) {
        if (!prev_planes[k] || !flows[k]) return fail("NULL entry in the temporal tables");
        if (prev_steps[k] < (size_t)g.w || flow_steps[k] < (size_t)g.w * 4 || (flow_steps[k] & 3)) return fail("bad step in the temporal tables");
        t.prev[k] = prev_planes[k]; t.prev_step[k] = prev_steps[k]; t.flow[k] = flows[k]; t.flow_step[k] = flow_steps[k];
    }
    HIP_TRY(hipSetDevice(e->params.device_id));
    launch_temporal_vote(planes, planes_step, t, smoothed, smoothed_step, g.w, g.h, static_cast<hipStream_t>(stream_));
    HIP_TRY(hipGetLastError());
    return 0;
}

int cart_reproject_depth(cart_engine *e, int n_frames, const int16_t *disp, size_t disp_step, size_t disp_frame_stride, const float Q[16],
                         float *xyz, size_t xyz_step, size_t xyz_frame_stride, void *stream_) {
    if (!e) return fail("engine is NULL");
    if (!disp || !xyz || !Q) return fail("NULL pointer");
    if (n_frames <= 0) return fail("n_frames must be positive");
    const Geometry &g = e->g;
    if (disp_step < (size_t)g.w * 2 || (disp_step & 1) || (disp_frame_stride & 1) || xyz_step < (size_t)g.w * 12 || (xyz_step & 3) || (xyz_frame_stride & 3))
        return fail("bad step");
    HIP_TRY(hipSetDevice(e->params.device_id));
    QMatrix q;
    std::memcpy(q.q, Q, sizeof(q.q));
    launch_reproject(disp, disp_step, disp_frame_stride, q, xyz, xyz_step, xyz_frame_stride, g.w, g.h, n_frames, static_cast<hipStream_t>(stream_));
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---- superpixels (replaces ContourRelaxation + SuperPixelModule's device work; oracle S13/S14) ----
struct cart_superpixels {
    cart_engine *engine = nullptr;
    cart_superpixel_params params;
    int block_w = 0, block_h = 0;
    int max_label_id = 0;        // number of initial blocks (labels are < max_label_id)
    uint16_t *labels[2] = {nullptr, nullptr};  // tight [h][w]; labels[cur] is the state
    int cur = 0;
    uint32_t *ycc = nullptr;
    long long *stats = nullptr;  // [kSpStatRows][ld] statistics followed by [kSpStatRows][ld] of per-sweep delta (capacity 2 x kSpStatRows x kSpMaxLabels)
    double *costs = nullptr;
    int *max_seen = nullptr;
    std::mutex mu;               // serialises calls (superpixels.cu:97-99)
    hipEvent_t done = nullptr;   // orders successive calls that arrive on different streams
    hipStream_t last_stream = nullptr;
    bool used = false;
    unsigned long long released_seq = 0;  // order of the last release (guarded by mu)
};

namespace {
int sp_enter(cart_superpixels *sp, hipStream_t stream) {
    if (sp->used && sp->last_stream != stream) HIP_TRY(hipStreamWaitEvent(stream, sp->done, 0));
    return 0;
}
void sp_leave(cart_superpixels *sp, hipStream_t stream) {
    (void)hipEventRecord(sp->done, stream);
    sp->last_stream = stream;
    sp->used = true;
}
// records the completion event on every way out of a call, early error returns included
struct SpScope {
    cart_superpixels *sp; hipStream_t stream;
    ~SpScope() { sp_leave(sp, stream); }
};
}  // namespace

void cart_superpixel_default_params(cart_superpixel_params *p) {
    if (!p) return;
    p->direct_clique_cost = 0.5;                       // cartconfig.cpp:128
    p->diagonal_clique_cost = 0.5 / std::sqrt(2.0);    // cartconfig.cpp:129
    p->compactness_weight = 0.1;                       // cartconfig.cpp:130
    p->progressive_compactness_cost = 0.0;             // cartconfig.cpp:131
    p->image_weight = 1.5;                             // cartconfig.cpp:132
    p->disparity_weight = 1.0;                         // cartconfig.cpp:133
}

int cart_superpixels_create(cart_engine *e, const cart_superpixel_params *params, int block_w, int block_h, cart_superpixels **out) {
    if (!e || !params || !out) return fail("bad arguments");
    if (block_w < 1 || block_h < 1) return fail("blockSize must be more than 1");                      // superpixels.cu:37-39
    if (params->direct_clique_cost < 0) return fail("directCliqueCost must be non-negative");          // superpixels.cu:41-43
    if (params->compactness_weight < 0 || params->image_weight < 0 || params->disparity_weight < 0)
        return fail("weight must be non-negative");                                                    // superpixels.cu:45-47
    const Geometry &g = e->g;
    if (g.w < block_w || g.h < block_h) return fail("image smaller than one block");                   // initialization.cu:42
    const long blocks = (long)((g.w + block_w - 1) / block_w) * ((g.h + block_h - 1) / block_h);
    if (blocks >= kSpMaxLabels) return fail("too many superpixels: number of blocks must be < 16384 (increase block size)");
    HIP_TRY(hipSetDevice(e->params.device_id));
    cart_superpixels *sp = new (std::nothrow) cart_superpixels;
    if (!sp) return fail("out of host memory");
    sp->engine = e; sp->params = *params; sp->block_w = block_w; sp->block_h = block_h; sp->max_label_id = (int)blocks;
    const size_t stat_elems = (size_t)kSpStatRows * kSpMaxLabels;
    bool ok = hipMalloc(reinterpret_cast<void **>(&sp->labels[0]), g.npx * 2) == hipSuccess &&
              hipMalloc(reinterpret_cast<void **>(&sp->labels[1]), g.npx * 2) == hipSuccess &&
              hipMalloc(reinterpret_cast<void **>(&sp->ycc), g.npx * 4) == hipSuccess &&
              hipMalloc(reinterpret_cast<void **>(&sp->stats), 2 * stat_elems * 8) == hipSuccess &&   // statistics, then their per-sweep delta (one memset per call)
              hipMalloc(reinterpret_cast<void **>(&sp->costs), (size_t)kSpChannels * kSpMaxLabels * 8) == hipSuccess &&
              hipMalloc(reinterpret_cast<void **>(&sp->max_seen), sizeof(int)) == hipSuccess &&
              hipEventCreateWithFlags(&sp->done, hipEventDisableTiming) == hipSuccess;
    if (ok) {
        launch_sp_block_init(sp->labels[0], g.w, g.h, block_w, block_h, nullptr);
        ok = hipGetLastError() == hipSuccess && hipDeviceSynchronize() == hipSuccess;
    }
    if (!ok) {
        cart_superpixels_destroy(sp);
        return fail("allocating the superpixel state failed");
    }
    *out = sp;
    return 0;
}

void cart_superpixels_destroy(cart_superpixels *sp) {
    if (!sp) return;
    if (sp->engine) (void)hipSetDevice(sp->engine->params.device_id);
    (void)hipDeviceSynchronize();
    void *bufs[] = {sp->labels[0], sp->labels[1], sp->ycc, sp->stats, sp->costs, sp->max_seen};
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    if (sp->done) (void)hipEventDestroy(sp->done);
    delete sp;
}

int cart_superpixels_max_label(const cart_superpixels *sp) { return sp ? sp->max_label_id : -1; }

int cart_superpixels_reset(cart_superpixels *sp, void *stream_) {
    if (!sp) return fail("superpixels is NULL");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const Geometry &g = sp->engine->g;
    HIP_TRY(hipSetDevice(sp->engine->params.device_id));
    std::lock_guard<std::mutex> lk(sp->mu);
    if (sp_enter(sp, stream)) return -1;
    SpScope scope{sp, stream};
    launch_sp_block_init(sp->labels[sp->cur], g.w, g.h, sp->block_w, sp->block_h, stream);
    sp->max_label_id = ((g.w + sp->block_w - 1) / sp->block_w) * ((g.h + sp->block_h - 1) / sp->block_h);
    HIP_TRY(hipGetLastError());
    return 0;
}

int cart_superpixels_set_labels(cart_superpixels *sp, const uint16_t *labels, size_t labels_step, int max_label_id, void *stream_) {
    if (!sp || !labels) return fail("bad arguments");
    if (max_label_id < 1 || max_label_id >= kSpMaxLabels) return fail("max_label_id must be in [1, 16384)");
    const Geometry &g = sp->engine->g;
    if (labels_step < (size_t)g.w * 2 || (labels_step & 1)) return fail("bad step");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    HIP_TRY(hipSetDevice(sp->engine->params.device_id));
    std::lock_guard<std::mutex> lk(sp->mu);
    if (sp_enter(sp, stream)) return -1;
    SpScope scope{sp, stream};
    // the copy goes to the spare buffer and becomes the state only if every label is in range
    uint16_t *spare = sp->labels[sp->cur ^ 1];
    HIP_TRY(hipMemsetAsync(sp->max_seen, 0, sizeof(int), stream));
    launch_sp_copy(labels, labels_step, spare, (size_t)g.w * 2, g.w, g.h, sp->max_seen, stream);
    int seen = 0;
    HIP_TRY(hipMemcpyAsync(&seen, sp->max_seen, sizeof(int), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if (seen >= max_label_id) return fail("label image holds a label >= max_label_id");
    sp->cur ^= 1;
    sp->max_label_id = max_label_id;
    return 0;
}

int cart_superpixels_relax(cart_superpixels *sp, const uint8_t *image, size_t image_step, int channels, const int16_t *deriv2,
                           size_t deriv2_step, int iterations, uint16_t *labels_out, size_t labels_out_step, void *stream_) {
    if (!sp) return fail("superpixels is NULL");
    if (!image) return fail("NULL pointer");
    if (channels != 1 && channels != 3) return fail("channels must be 1 or 3");
    if (iterations < 0) return fail("iterations must be >= 0");
    const Geometry &g = sp->engine->g;
    const cart_superpixel_params &p = sp->params;
    if (image_step < (size_t)g.w * channels) return fail("bad step");
    if (p.disparity_weight > 0) {
        if (!deriv2) return fail("the disparity feature needs the 2-channel disparity derivative image");
        if (deriv2_step < (size_t)g.w * 4 || (deriv2_step & 3) || (reinterpret_cast<uintptr_t>(deriv2) & 3)) return fail("bad step");
    }
    if (labels_out && (labels_out_step < (size_t)g.w * 2 || (labels_out_step & 1))) return fail("bad step");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    HIP_TRY(hipSetDevice(sp->engine->params.device_id));
    std::lock_guard<std::mutex> lk(sp->mu);
    if (sp_enter(sp, stream)) return -1;
    SpScope scope{sp, stream};
    const int ld = sp->max_label_id + 1;
    SpRelaxArgs a;
    std::memset(&a, 0, sizeof(a));
    a.ycc = sp->ycc; a.deriv = p.disparity_weight > 0 ? deriv2 : nullptr; a.deriv_step = deriv2_step;
    a.stats = sp->stats; a.costs = sp->costs; a.delta = sp->stats + (size_t)kSpStatRows * ld; a.ld = ld; a.w = g.w; a.h = g.h;
    a.ch_mask = (p.compactness_weight > 0 ? 0x03u : 0u) | (p.disparity_weight > 0 ? 0x0cu : 0u) | (p.image_weight > 0 ? 0x70u : 0u);
    a.direct = p.direct_clique_cost; a.diagonal = p.diagonal_clique_cost; a.w_comp = p.compactness_weight;
    a.prog = p.progressive_compactness_cost; a.w_img = p.image_weight; a.w_disp = p.disparity_weight;
    if (iterations > 0) {   // the colour planes and the label statistics only serve the sweeps; every call rebuilds them from the image and the labels
        launch_sp_ycrcb(image, image_step, channels, sp->ycc, g.w, g.h, stream);
        HIP_TRY(hipMemsetAsync(sp->stats, 0, 2 * (size_t)kSpStatRows * ld * 8, stream));
        a.cur = sp->labels[sp->cur]; a.next = sp->labels[sp->cur ^ 1];
        launch_sp_stats(a, stream);
        launch_sp_fold(a.stats, a.delta, sp->costs, ld, a.ch_mask, stream);
    }
    for (int it = 0; it < iterations; ++it) {
        a.cur = sp->labels[sp->cur]; a.next = sp->labels[sp->cur ^ 1];
        launch_sp_relax(a, stream);
        if (it + 1 < iterations) launch_sp_fold(a.stats, a.delta, sp->costs, ld, a.ch_mask, stream);   // the last sweep's delta has no reader
        sp->cur ^= 1;
    }
    if (labels_out) launch_sp_copy(sp->labels[sp->cur], (size_t)g.w * 2, labels_out, labels_out_step, g.w, g.h, nullptr, stream);
    HIP_TRY(hipGetLastError());
    return 0;
}

int cart_superpixel_plane_classify(cart_engine *e, const int16_t *deriv2, size_t deriv2_step, const uint16_t *labels, size_t labels_step,
                                   int max_label, const cart_plane_params *params, int n_prev, const uint8_t *const *prev_planes,
                                   const size_t *prev_steps, const int16_t *const *flows, const size_t *flow_steps,
                                   uint8_t *planes_unsmoothed, size_t planes_unsmoothed_step, uint8_t *planes, size_t planes_step,
                                   void *stream_) {
    if (!e) return fail("engine is NULL");
    if (!deriv2 || !labels || !params || !planes_unsmoothed || !planes) return fail("NULL pointer");
    if (max_label < 1 || max_label > kSpMaxLabels) return fail("max_label must be in [1, 16384]");
    if (n_prev < 0 || n_prev > CART_MAX_TEMPORAL) return fail("n_prev must be in [0, CART_MAX_TEMPORAL]");
    if (n_prev > 0 && (!prev_planes || !prev_steps || !flows || !flow_steps)) return fail("NULL table");
    const Geometry &g = e->g;
    if (deriv2_step < (size_t)g.w * 4 || (deriv2_step & 3) || labels_step < (size_t)g.w * 2 || (labels_step & 1) ||
        planes_unsmoothed_step < (size_t)g.w || planes_step < (size_t)g.w)
        return fail("bad step");
    SpClassifyArgs a;
    std::memset(&a, 0, sizeof(a));
    a.t.n_prev = n_prev;
    for (int k = 0; k < n_prev; ++k) {
        if (!prev_planes[k] || !flows[k]) return fail("NULL entry in the temporal tables");
        if (prev_steps[k] < (size_t)g.w || flow_steps[k] < (size_t)g.w * 4 || (flow_steps[k] & 3)) return fail("bad step in the temporal tables");
        a.t.prev[k] = prev_planes[k]; a.t.prev_step[k] = prev_steps[k]; a.t.flow[k] = flows[k]; a.t.flow_step[k] = flow_steps[k];
    }
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    HIP_TRY(hipSetDevice(e->params.device_id));
    {
        std::lock_guard<std::mutex> lk(e->mu);
        if (!e->sp_votes) HIP_TRY(hipMalloc(reinterpret_cast<void **>(&e->sp_votes), e->slots.size() * (size_t)kSpMaxLabels * 3 * sizeof(unsigned)));
    }
    Lease lease;
    if (acquire(e, 1, stream, &lease)) return -1;
    a.deriv = deriv2; a.deriv_step = deriv2_step; a.labels = labels; a.labels_step = labels_step;
    a.w = g.w; a.h = g.h; a.max_label = max_label; a.p = *params;
    a.unsmoothed = planes_unsmoothed; a.unsmoothed_step = planes_unsmoothed_step; a.planes = planes; a.planes_step = planes_step;
    a.votes = e->sp_votes + (size_t)lease.s0 * kSpMaxLabels * 3;
    hipError_t err = hipMemsetAsync(a.votes, 0, (size_t)max_label * 3 * sizeof(unsigned), stream);
    if (err == hipSuccess) launch_sp_classify(a, stream);
    release(lease);
    if (err != hipSuccess) return fail(std::string("hipMemsetAsync: ") + hipGetErrorString(err));
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---- optical flow (oracle S15) ----
int cart_optical_flow(cart_engine *e, const uint8_t *cur, size_t cur_step, const uint8_t *prev, size_t prev_step, int channels,
                      int radius, int block, int16_t *flow, size_t flow_step, void *stream_) {
    if (!e) return fail("engine is NULL");
    if (!cur || !prev || !flow) return fail("NULL image pointer");
    if (channels != 1 && channels != 3) return fail("channels must be 1 (gray) or 3 (BGR)");
    if (radius < 1 || radius > 16) return fail("radius must be in [1, 16]");
    if (block < 1 || block > 3) return fail("block must be in [1, 3]");
    const Geometry &g = e->g;
    if (cur_step < (size_t)g.w * channels || prev_step < (size_t)g.w * channels) return fail("input step smaller than a row");
    if (flow_step < (size_t)g.w * 4 || (flow_step & 3) || (reinterpret_cast<uintptr_t>(flow) & 3)) return fail("bad step");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    HIP_TRY(hipSetDevice(e->params.device_id));
    const size_t cen_bytes = g.census_elems * sizeof(uint32_t);
    const size_t ws_bytes = ((2 * g.npx + 255) & ~(size_t)255) + 2 * cen_bytes + g.npx * sizeof(uint32_t);
    {
        std::lock_guard<std::mutex> lk(e->mu);
        if (!e->flow_ws) HIP_TRY(hipMalloc(reinterpret_cast<void **>(&e->flow_ws), e->slots.size() * ws_bytes));
    }
    Lease l;
    if (acquire(e, 1, stream, &l)) return -1;
    uint8_t *ws = e->flow_ws + (size_t)l.s0 * ws_bytes;
    uint8_t *gray_c = ws, *gray_p = ws + g.npx;
    uint32_t *cen_c = reinterpret_cast<uint32_t *>(ws + ((2 * g.npx + 255) & ~(size_t)255));
    uint32_t *cen_p = cen_c + g.census_elems;
    uint32_t *scratch = cen_p + g.census_elems;   // census_kernel also resets a right-view plane: unused here
    const ImageBatch cb = strided_images(cur, cur_step, 0), pb = strided_images(prev, prev_step, 0);
    launch_census(cb, pb, channels, 1, gray_c, gray_p, cen_c, cen_p, scratch, g, stream);
    launch_block_flow(cen_c, cen_p, g, radius, block, flow, flow_step, stream);
    hipError_t err = hipGetLastError();
    release(l);
    if (err != hipSuccess) return fail(std::string("kernel launch failed: ") + hipGetErrorString(err));
    return 0;
}

int cart_resize_linear(int device_id, const uint8_t *src, size_t src_step, int sw, int sh, int channels, uint8_t *dst, size_t dst_step, int dw,
                       int dh, void *stream_) {
    if (!src || !dst) return fail("NULL image pointer");
    if (channels != 1 && channels != 3) return fail("channels must be 1 or 3");
    if (sw < 1 || sh < 1 || dw < 1 || dh < 1 || sw > 16384 || sh > 16384 || dw > 16384 || dh > 16384) return fail("unsupported image size");
    if (src_step < (size_t)sw * channels || dst_step < (size_t)dw * channels) return fail("step smaller than a row");
    HIP_TRY(hipSetDevice(device_id));
    launch_resize_linear(src, src_step, sw, sh, channels, dst, dst_step, dw, dh, static_cast<hipStream_t>(stream_));
    HIP_TRY(hipGetLastError());
    return 0;
}

int cart_copy_narrow(cart_engine *e, void *dst, const void *src, size_t bytes, int workgroups, void *stream_) {
    if (!e) return fail("engine is NULL");
    if (!dst || !src) return fail("NULL pointer");
    if ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15) return fail("buffers must be 16-byte aligned");
    if (workgroups < 0 || workgroups > 1024) return fail("workgroups must be in [0, 1024]");
    if (bytes == 0) return 0;
    HIP_TRY(hipSetDevice(e->params.device_id));
    launch_narrow_copy(src, dst, bytes, workgroups ? workgroups : 8, static_cast<hipStream_t>(stream_));
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---- host-side peak finder (replaces src/utils/peaks.cpp:12-72 and planeseg.cu:405-458) ----
int cart_find_peaks(const int32_t *data, int n, int *born, int *died, int *left, int *right) {
    if (!data || n <= 0 || !born || !died || !left || !right) return fail("bad arguments");
    try {
    std::vector<int> order(n), owner(n, -1);
    for (int i = 0; i < n; ++i) order[i] = i;
    // descending value, ties by ascending index (oracle S11; the reference's std::sort leaves ties open)
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return data[a] > data[b]; });
    int np = 0;
    for (int idx : order) {
        const int il = (idx > 0) ? owner[idx - 1] : -1;
        const int ir = (idx < n - 1) ? owner[idx + 1] : -1;
        if (il < 0 && ir < 0) {  // a new component is born at a local maximum
            born[np] = left[np] = right[np] = idx; died[np] = -1;
            owner[idx] = np++;
        } else if (il >= 0 && ir < 0) {
            right[il] += 1; owner[idx] = il;
        } else if (il < 0 && ir >= 0) {
            left[ir] -= 1; owner[idx] = ir;
        } else if (data[born[il]] > data[born[ir]]) {  // the younger (lower) peak dies at this saddle
            died[ir] = idx; right[il] = right[ir];
            owner[right[il]] = owner[idx] = il;
        } else {
            died[il] = idx; left[ir] = left[il];
            owner[left[ir]] = owner[idx] = ir;
        }
    }
    std::vector<int> perm(np);
    for (int i = 0; i < np; ++i) perm[i] = i;
    auto persistence = [&](int k) -> long long { return died[k] < 0 ? (long long)INT32_MAX : (long long)data[born[k]] - data[died[k]]; };
    std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) { return persistence(a) > persistence(b); });
    std::vector<int> b2(np), d2(np), l2(np), r2(np);
    for (int i = 0; i < np; ++i) { b2[i] = born[perm[i]]; d2[i] = died[perm[i]]; l2[i] = left[perm[i]]; r2[i] = right[perm[i]]; }
    for (int i = 0; i < np; ++i) { born[i] = b2[i]; died[i] = d2[i]; left[i] = l2[i]; right[i] = r2[i]; }
    return np;
    } catch (const std::bad_alloc &) { return fail("out of host memory"); }   // nothing is thrown across the C ABI
}

int cart_find_plane_params(const int32_t hist[256], cart_plane_params *io) {
    if (!hist || !io) return fail("bad arguments");
    int born[256], died[256], left[256], right[256];
    const int np = cart_find_peaks(hist, 256, born, died, left, right);
    if (np < 2) return 0;  // planeseg.cu:408-411
    int pv = born[0], ph = born[1];
    if (std::abs(pv - 128) > std::abs(ph - 128)) std::swap(pv, ph);  // vertical = nearer to zero derivative (:414-416)
    io->vertical_center = pv - 128;
    io->horizontal_center = ph - 128;
    int valley = std::min(pv, ph);
    for (int i = valley; i < std::max(pv, ph); ++i)
        if (hist[i] < hist[valley]) valley = i;  // :422-428
    const int vdist = std::abs(valley - pv), hdist = std::abs(valley - ph);
    if (vdist == 0 || hdist == 0) return 0;  // :436-439
    const int vslope = (hist[pv] - hist[valley]) / vdist, hslope = (hist[ph] - hist[valley]) / hdist;
    if (vslope == 0 || hslope == 0) return 0;  // :444-447
    const int vwidth = hist[pv] / vslope, hwidth = hist[ph] / hslope;
    io->vertical_min = pv - vwidth - 128; io->vertical_max = valley - 127;      // :452
    io->horizontal_min = valley - 127; io->horizontal_max = ph + hwidth - 127;  // :453
    return 1;
}

int cart_debug_slab_layout(cart_engine *e, int *group_slots, int *n_groups, size_t *slot_bytes, size_t *group_bytes) {
    if (!e) return fail("engine is NULL");
    if (e->post_only) return fail("this engine was created without SGM workspaces (num_disparities = 0)");
    const SlabPool &sp = e->slab_pool;
    if (group_slots) *group_slots = sp.group_slots;
    if (n_groups) *n_groups = sp.groups();
    if (slot_bytes) *slot_bytes = sp.slot_bytes;
    if (group_bytes) *group_bytes = sp.bytes_of(0);
    return 0;
}

int cart_debug_ccl_scratch_nonzero(cart_engine *e, size_t *nonzero) {
    if (!e || !nonzero) return fail("bad arguments");
    *nonzero = 0;
    if (!e->ccl_stats_ws) return 0;
    HIP_TRY(hipSetDevice(e->params.device_id));
    HIP_TRY(hipDeviceSynchronize());
    try {
        const size_t per_slot = e->g.npx * 5;   // the statistics scratch of one slot (the segment counts behind it are overwritten, not accumulated)
        std::vector<int32_t> host(per_slot);
        for (size_t sl = 0; sl < e->slots.size(); ++sl) {
            HIP_TRY(hipMemcpy(host.data(), e->ccl_stats_ws + sl * per_slot, per_slot * sizeof(int32_t), hipMemcpyDeviceToHost));
            for (int32_t v : host) *nonzero += v != 0;
        }
    } catch (const std::bad_alloc &) { return fail("out of host memory"); }
    return 0;
}

int cart_debug_uniq_table(cart_engine *e, int uniqueness_ratio, uint16_t *out2048) {
    if (!out2048) return fail("bad arguments");
    if (uniqueness_ratio < 0 || uniqueness_ratio > 100) return fail("uniqueness_ratio must be in [0, 100]");
    const float u = (float)(100 - uniqueness_ratio) / 100.0f;   // as cart_engine_create (oracle S5)
    if (!e) { uniq_table_host(u, out2048); return 0; }
    HIP_TRY(hipSetDevice(e->params.device_id));
    uint16_t *dev = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&dev), 2048 * sizeof(uint16_t)));
    launch_uniq_table(u, dev, nullptr);
    hipError_t err = hipMemcpy(out2048, dev, 2048 * sizeof(uint16_t), hipMemcpyDeviceToHost);
    (void)hipFree(dev);
    if (err != hipSuccess) return fail(std::string("uniq table: ") + hipGetErrorString(err));
    return 0;
}

int cart_debug_read(cart_engine *e, int frame_slot, int what, void *host_dst, size_t bytes) {
    if (!e || !host_dst) return fail("bad arguments");
    if (e->post_only) return fail("this engine has no SGM workspaces");
    const Geometry &g = e->g;
    const int slot = g_last_slot + frame_slot;
    if (frame_slot < 0 || slot >= (int)e->slots.size()) return fail("frame_slot out of range");
    HIP_TRY(hipSetDevice(e->params.device_id));
    HIP_TRY(hipDeviceSynchronize());
    const void *src = nullptr;
    size_t need = 0;
    try {
    if (what == CART_DBG_GRAY_L || what == CART_DBG_GRAY_R) {
        src = (what == CART_DBG_GRAY_L ? e->gray_l : e->gray_r) + (size_t)slot * g.npx; need = g.npx;
        if (bytes < need) return fail("buffer too small");
        HIP_TRY(hipMemcpy(host_dst, src, need, hipMemcpyDeviceToHost));
    } else if (what == CART_DBG_CENSUS_L || what == CART_DBG_CENSUS_R) {
        const uint32_t *c = (what == CART_DBG_CENSUS_L ? e->cen_l : e->cen_r) + (size_t)slot * g.census_elems + g.cpadl;
        need = g.npx * 4;
        if (bytes < need) return fail("buffer too small");
        HIP_TRY(hipMemcpy2D(host_dst, (size_t)g.w * 4, c, (size_t)g.cpitch * 4, (size_t)g.w * 4, g.h, hipMemcpyDeviceToHost));
    } else if (what >= CART_DBG_PATH0 && what < CART_DBG_PATH0 + g.P) {
        src = e->slab_pool.slot_ptr(slot) + (size_t)(what - CART_DBG_PATH0) * g.slab_bytes; need = g.npx * g.D;
        if (bytes < need) return fail("buffer too small");
        std::vector<uint8_t> raw(need);
        HIP_TRY(hipMemcpy(raw.data(), src, need, hipMemcpyDeviceToHost));
        uint8_t *d = static_cast<uint8_t *>(host_dst);  // undo the in-slab chunk order -> plain [h][w][D]
        for (size_t c = 0; c < need; c += 16)
            for (int k = 0; k < 16; ++k) d[c + kSlabChunkOrder[k]] = raw[c + k];
    } else if (what == CART_DBG_WTA_L) {
        src = e->wta_l + (size_t)slot * g.npx; need = g.npx * 2;
        if (bytes < need) return fail("buffer too small");
        HIP_TRY(hipMemcpy(host_dst, src, need, hipMemcpyDeviceToHost));
    } else if (what == CART_DBG_WTA_R) {  // packed (cost<<16 | disparity) -> u16 disparity
        std::vector<uint32_t> tmp(g.npx);
        need = g.npx * 2;
        if (bytes < need) return fail("buffer too small");
        HIP_TRY(hipMemcpy(tmp.data(), e->right_pk + (size_t)slot * g.npx, g.npx * 4, hipMemcpyDeviceToHost));
        uint16_t *d = static_cast<uint16_t *>(host_dst);
        for (size_t i = 0; i < g.npx; ++i) d[i] = (uint16_t)(tmp[i] & 0xffffu);
    } else {
        return fail("unknown debug selector");
    }
    } catch (const std::bad_alloc &) { return fail("out of host memory"); }   // the staging vectors (a slab is up to 530 MB)
    return 0;
}

}  // extern "C"
