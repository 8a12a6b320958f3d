// bvc_api.hip -- host side of libbvc: the C ABI declared in include/bvc.h.
// One context = one gfx950 device + one HIP stream + device scratch.  No CPU fallback exists: every
// compute entry point launches the HIP kernels or fails with an error code.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <mutex>
#include <vector>

#include "bvc_internal.h"

using namespace bvc;

struct bvc_ctx {
    int device = -1;
    hipStream_t stream = nullptr;      // the stream the entry points work on: own_stream until bvc_set_stream names another
    hipStream_t own_stream = nullptr;  // created with the context.  A BLOCKING stream: ordered against the device's default stream as the
                                       // default stream itself is (callers that prepare buffers there need no extra synchronisation), but the
                                       // streams of different contexts run side by side -- sixteen threads of the host program, each with
                                       // its context, were one queue when every context worked on the default stream
    hipEvent_t ev_wait = nullptr;      // blocking-sync event: the long waits of the pileup calls SLEEP on it (wait_stream) where
                                       // hipStreamSynchronize polls -- a host program with a thread per context on a CPU quota cannot afford that
    QualLut *d_lut = nullptr;
    // [sites][512] scratch between the two stages.  Overlap mode cycles through kRing buffers: with three, the
    // histogram pass of call i+1 waits only for the EM of call i-2 (long finished), never for the one running
    // beside it, so both streams run back to back.
    static constexpr int kRing = 4;
    uint32_t *d_cnt[kRing] = {};
    size_t cnt_cap[kRing] = {};
    // overlap mode: stage 2 of call i runs on `side` while stage 1 of call i+1 streams on `stream`
    bool overlap = false;
    hipStream_t side = nullptr;        // stage 2 of even calls
    hipStream_t side_b = nullptr, side_c = nullptr;   // further stage-2 streams (two_em_streams below)
    unsigned side_flip = 0;
    int flip = 0;
    hipEvent_t ev_hist_done[kRing] = {};
    hipEvent_t ev_em_done[kRing] = {};
    bool em_pending[kRing] = {};
    uint32_t *d_grp[kRing] = {};   // [sites][groups + 1][512] in group mode
    size_t grp_cap[kRing] = {};
    // item-engine scratch of stage 2 (em_items.hip), one per ring buffer (the stage 2 of consecutive calls may run
    // side by side); the last one serves bvc_lrt_hist on the context's own stream
    void *d_em[kRing + 1] = {};
    size_t em_cap[kRing + 1] = {};
    void *d_emg[kRing] = {};      // the same for the (site, group) pseudo-sites of group calls
    size_t emg_cap[kRing] = {};
    uint32_t *d_sink = nullptr;        // 256 bytes: sink of the streaming-read measurement kernel; bvc_pack_dense's counter at byte 64
    uint8_t *d_grp_labels = nullptr;   // group mode: the call's group vector clamped to 0..n_groups (hist_kernel.hip)
    size_t grp_labels_cap = 0;
    int64_t *d_grp_scratch = nullptr;  // group mode: "samples ordered by group" flag + column bounds (hist_kernel.hip)
    // staging for BVC_PTR_HOST calls: two sets, so that the upload of chunk i+1 (copy stream) runs under the kernels
    // of chunk i
    char *d_stage[2] = {nullptr, nullptr};
    size_t stage_cap[2] = {0, 0};
    hipStream_t copy = nullptr;
    hipEvent_t ev_upload[2] = {nullptr, nullptr};
    hipEvent_t ev_set_free[2] = {nullptr, nullptr};   // ragged host calls: the kernels that read staging set k have finished
    // bvc_pileup_begin / bvc_pileup_finish: the tile's text, its line tables and counts, its parsed columns and records
    char *d_pl_text = nullptr, *d_pl_meta = nullptr, *d_pl_out = nullptr, *d_pl_called = nullptr;
    size_t pl_text_cap = 0, pl_meta_cap = 0, pl_out_cap = 0, pl_called_cap = 0;
    // tiles inflated on the device (bvc_pileup_begin_bgzf): two text buffers (what a tile leaves of a batch is carried from one to the
    // other), the compressed bytes, what the calls left of every batch
    char *d_pz_text[2] = {nullptr, nullptr}, *d_pz_comp = nullptr;
    size_t pz_text_cap[2] = {0, 0}, pz_comp_cap = 0;
    int pz_cur = 0;                    // the buffer that holds the leftovers
    std::vector<uint32_t> pz_left_src, pz_left_len;
    int64_t pl_text_bytes = 0, pl_indel_bytes = 0;
    bool pl_on_device_text = false;
    // pinned host memory the pileup calls bounce their transfers through: a copy from or to pageable memory makes the calling thread
    // wait inside the runtime -- spinning -- for the whole transfer; from pinned memory it is a DMA the thread sleeps behind (wait_stream)
    char *h_up = nullptr, *h_down = nullptr;
    size_t up_cap = 0, down_cap = 0;
    PileupTile pl;                     // the tile between the two calls
    bool pl_begun = false;
    int64_t pl_entries = 0, pl_obs = 0, pl_indels = 0;
    LaunchState ls;                    // launch policy + one-time kernel setup of this context
    bool profiling = false;
    std::vector<hipEvent_t> ev_pool;   // free events
    struct Triple { hipEvent_t a, b, c, d; int64_t sites; };   // hist = a..b, EM = c..d
    std::vector<Triple> ev_pending;
    bvc_profile prof{};
    std::string err;
};

namespace {

int fail(bvc_ctx *ctx, int code, const char *what, hipError_t e = hipSuccess)
{
    if (ctx) {
        ctx->err = what;
        if (e != hipSuccess) { ctx->err += ": "; ctx->err += hipGetErrorString(e); }
    }
    return code;
}

// Waits for the context's stream without spinning: the thread sleeps until the event behind everything enqueued so far fires.
inline hipError_t wait_stream(bvc_ctx *ctx)
{
    hipError_t e = hipEventRecord(ctx->ev_wait, ctx->stream);
    return e == hipSuccess ? hipEventSynchronize(ctx->ev_wait) : e;
}

#define BVC_HIP(ctx, call)                                                         \
    do {                                                                           \
        hipError_t e__ = (call);                                                   \
        if (e__ != hipSuccess) return fail((ctx), BVC_ERR_DEVICE, #call, e__);     \
    } while (0)

int ensure(bvc_ctx *ctx, void **buf, size_t *cap, size_t need)
{
    if (need <= *cap) return BVC_OK;
    if (*buf) {
        // nothing may still be reading or writing the old buffer: the context's stream, and the copy stream (a staging
        // set may have an upload in flight after a failed host-pointer call)
        BVC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->copy) BVC_HIP(ctx, hipStreamSynchronize(ctx->copy));
        BVC_HIP(ctx, hipFree(*buf));
        *buf = nullptr; *cap = 0;
    }
    size_t want = need + need / 4;
    if (hipMalloc(buf, want) != hipSuccess) {
        (void)hipGetLastError();
        if (hipMalloc(buf, need) != hipSuccess) { (void)hipGetLastError(); return fail(ctx, BVC_ERR_ALLOC, "device scratch allocation failed"); }
        want = need;
    }
    *cap = want;
    // Fresh device memory holds whatever its last owner left.  No kernel of the library is meant to read scratch it has not
    // written, and so that a slip there can never read another call's (or another process's) leftovers the new buffer is
    // cleared before anything can touch it -- to 0xFF bytes in the -DBVC_POISON build, which makes such a slip loud.
    // (Allocation happens once per buffer and size: the synchronize is not on the steady-state path.)
#ifdef BVC_POISON
    constexpr int kFill = 0xFF;
#else
    constexpr int kFill = 0;
#endif
    BVC_HIP(ctx, hipMemsetAsync(*buf, kFill, want, ctx->stream));
    BVC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return BVC_OK;
}

hipEvent_t take_event(bvc_ctx *ctx)
{
    if (!ctx->ev_pool.empty()) { hipEvent_t e = ctx->ev_pool.back(); ctx->ev_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

void give_back(bvc_ctx *ctx, hipEvent_t e) { if (e) ctx->ev_pool.push_back(e); }

// Four timing events for one call, or none at all (never a partial set).
bool take_timing_events(bvc_ctx *ctx, bvc_ctx::Triple &t)
{
    t.a = take_event(ctx); t.b = take_event(ctx); t.c = take_event(ctx); t.d = take_event(ctx);
    if (t.a && t.b && t.c && t.d) return true;
    give_back(ctx, t.a); give_back(ctx, t.b); give_back(ctx, t.c); give_back(ctx, t.d);
    t.a = t.b = t.c = t.d = nullptr;
    return false;
}

// Drains finished timing records into the running totals so that ev_pending stays bounded when the caller never
// asks for the profile.
void reap_timing(bvc_ctx *ctx, bool all);

// Make the context's stream wait for every stage-2 launch still running on the side stream.
int join_side(bvc_ctx *ctx)
{
    for (int b = 0; b < bvc_ctx::kRing; ++b)
        if (ctx->em_pending[b]) {
            BVC_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_em_done[b], 0));
            ctx->em_pending[b] = false;
        }
    return BVC_OK;
}

// Stage 2 of consecutive calls on alternating side streams?  An EM launch ends in a long tail of few busy waves (sites
// differ 4x in EM work and a launch has 1.6-2 sites per wave); on two streams the next launch's workgroups move in as
// the previous one's drain: +20-28 % where the EM is the bound (1e4 x 1e4: 1.46e7 -> 1.76e7 sites/s; ragged sites at
// 10 % coverage: 7.9e6 -> 1.02e7).  Underneath a long histogram pass the EM is not the bound and a second launch only
// adds to the crowd (-0.4 % on the headline), so long rows keep one stream.  (profiles/r02_sweep_em_streams.txt)
// Group mode's stage 2 IS the bound underneath its long histogram pass (k = 5, labels in any order): there two
// streams at half the wave budget each (2 x 4 per CU instead of 1 x 8) give +3.8 % (any order) / +1 % (ordered
// columns) -- the same chip share, the tails covered.  (profiles/r02_sweep_group_em_streams.txt)
int em_stream_count(const bvc_ctx *ctx, int by_default) { return ctx->ls.em_streams > 0 ? ctx->ls.em_streams : by_default; }

hipStream_t em_stream(bvc_ctx *ctx, int by_default)
{
    const int n = em_stream_count(ctx, by_default);
    const int k = (int)(ctx->side_flip++ % (unsigned)n);
    return k == 0 ? ctx->side : (k == 1 ? ctx->side_b : ctx->side_c);
}

// Device scratch of the item engine for a stage 2 of n_sites sites on ring buffer `slot` (null when the call will not
// use the engine: launch_lrt's rule).  Growing it waits for whatever may still be using the old one.
int em_scratch_for(bvc_ctx *ctx, int slot, int64_t n_sites, double min_af, void **out, int n_groups = 0)
{
    *out = nullptr;
    if (ctx->ls.em_engine == 1 || !(min_af > 0.0)) return BVC_OK;
    const size_t need = n_groups > 0 ? em_group_scratch_bytes(n_sites, n_groups) : em_items_scratch_bytes(n_sites);
    void **buf = n_groups > 0 ? &ctx->d_emg[slot] : &ctx->d_em[slot];
    size_t *cap = n_groups > 0 ? &ctx->emg_cap[slot] : &ctx->em_cap[slot];
    if (need > *cap) {
        int rcj = join_side(ctx);
        if (rcj != BVC_OK) return rcj;
    }
    int rc = ensure(ctx, buf, cap, need);
    if (rc != BVC_OK) return rc;
    *out = *buf;
    return BVC_OK;
}

// The two stages on device pointers.  `stage1(counts)` launches the histogram pass of the call (dense, ragged, ...)
// on the context's stream into a [n_sites][512] buffer of the ring; stage 2 (EM/LRT) follows on the same stream, or
// on the side stream behind an event in overlap mode.
template <class Stage1>
int run_two_stages(bvc_ctx *ctx, int64_t n_sites, bool zero_counts, bool long_rows, Stage1 stage1,
                   const int8_t *ref_base, double min_af, const int8_t *comb, const uint8_t *n_comb,
                   bvc_site_result *results, int em_streams_long_rows = 1)
{
    const size_t cbytes = (size_t)n_sites * BVC_NCLASS * sizeof(uint32_t);
    const int buf = ctx->overlap ? ctx->flip : 0;
    if (ctx->overlap) ctx->flip = (ctx->flip + 1) % bvc_ctx::kRing;
    uint32_t **counts_p = &ctx->d_cnt[buf];
    size_t *cap_p = &ctx->cnt_cap[buf];
    if (cbytes > *cap_p) {                      // growing a buffer: nothing may still be reading it
        int rcj = join_side(ctx);
        if (rcj != BVC_OK) return rcj;
    }
    int rc = ensure(ctx, reinterpret_cast<void **>(counts_p), cap_p, cbytes);
    if (rc != BVC_OK) return rc;
    uint32_t *counts = *counts_p;
    void *em_scratch = nullptr;
    rc = em_scratch_for(ctx, buf, n_sites, min_af, &em_scratch);
    if (rc != BVC_OK) return rc;
    bvc_ctx::Triple t{nullptr, nullptr, nullptr, nullptr, n_sites};
    const bool timed = ctx->profiling && take_timing_events(ctx, t);
    auto bail = [&](int code) {                 // an early exit returns the timing events to the pool
        give_back(ctx, t.a); give_back(ctx, t.b); give_back(ctx, t.c); give_back(ctx, t.d);
        return code;
    };
#define BVC_HIP_T(call)                                                                     \
    do {                                                                                    \
        hipError_t e__ = (call);                                                            \
        if (e__ != hipSuccess) return bail(fail(ctx, BVC_ERR_DEVICE, #call, e__));          \
    } while (0)

    // stage 1 on the context's stream; the histogram buffer is free once the EM that read it has finished
    if (ctx->overlap && ctx->em_pending[buf]) {
        BVC_HIP_T(hipStreamWaitEvent(ctx->stream, ctx->ev_em_done[buf], 0));
        ctx->em_pending[buf] = false;
    }
    if (zero_counts) BVC_HIP_T(hipMemsetAsync(counts, 0, cbytes, ctx->stream));
    if (timed) BVC_HIP_T(hipEventRecord(t.a, ctx->stream));
    BVC_HIP_T(stage1(counts));
    if (timed) BVC_HIP_T(hipEventRecord(t.b, ctx->stream));

    // stage 2: same stream, or the side stream behind an event
    hipStream_t s2 = ctx->stream;
    if (ctx->overlap) {
        s2 = em_stream(ctx, long_rows ? em_streams_long_rows : 2);
        BVC_HIP_T(hipEventRecord(ctx->ev_hist_done[buf], ctx->stream));
        BVC_HIP_T(hipStreamWaitEvent(s2, ctx->ev_hist_done[buf], 0));
    }
    if (timed) BVC_HIP_T(hipEventRecord(t.c, s2));
    // underneath a long streaming pass the EM kernel keeps to a few wave slots; with short rows it is the longer
    // kernel and takes the chip
    const bool shared = ctx->overlap && long_rows;
    BVC_HIP_T(launch_lrt(ctx->ls, s2, n_sites, counts, BVC_NCLASS, ref_base, min_af, ctx->d_lut, comb, n_comb, results, shared, 0,
                         em_scratch));
    if (timed) {
        BVC_HIP_T(hipEventRecord(t.d, s2));
        ctx->ev_pending.push_back(t);
        t = bvc_ctx::Triple{nullptr, nullptr, nullptr, nullptr, 0};
        if (ctx->ev_pending.size() > 256) reap_timing(ctx, false);
    }
    if (ctx->overlap) {
        BVC_HIP_T(hipEventRecord(ctx->ev_em_done[buf], s2));
        ctx->em_pending[buf] = true;
    }
#undef BVC_HIP_T
    return BVC_OK;
}

int run_dense_device(bvc_ctx *ctx, int64_t n_sites, int64_t n_samples, int64_t row_stride,
                     const int8_t *bases, const int8_t *quals, const int8_t *ref_base, double min_af,
                     bvc_site_result *results)
{
    const int split = choose_hist_split(ctx->ls, n_sites, n_samples);
    return run_two_stages(ctx, n_sites, split > 1, n_samples >= 200000,
                          [&](uint32_t *counts) {
                              return launch_hist_dense(ctx->ls, ctx->stream, n_sites, n_samples, row_stride, bases, quals,
                                                       nullptr, 0, counts, split);
                          },
                          ref_base, min_af, nullptr, nullptr, results);
}

int run_packed_device(bvc_ctx *ctx, int64_t n_sites, int64_t n_samples, int64_t row_stride, const uint8_t *packed,
                      const int8_t *ref_base, double min_af, bvc_site_result *results)
{
    const int split = choose_hist_split(ctx->ls, n_sites, n_samples);
    return run_two_stages(ctx, n_sites, split > 1, n_samples >= 200000,
                          [&](uint32_t *counts) {
                              return launch_hist_packed(ctx->ls, ctx->stream, n_sites, n_samples, row_stride, packed, counts, split);
                          },
                          // at one byte per sample stage 1 is as short as stage 2 and the call is bound by the VALU the
                          // two share: stage 2 on two streams (4.0e6 -> 4.55e6 sites/s, profiles/r02_packed_sweep.txt)
                          ref_base, min_af, nullptr, nullptr, results, 2);
}

int run_csr_device(bvc_ctx *ctx, int64_t n_sites, const int64_t *offsets, const int8_t *bases, const int8_t *quals,
                   const int8_t *ref_base, double min_af, const int8_t *comb, const uint8_t *n_comb,
                   bvc_site_result *results)
{
    return run_two_stages(ctx, n_sites, false, false,
                          [&](uint32_t *counts) { return launch_hist_csr(ctx->ls, ctx->stream, n_sites, offsets, bases, quals, counts); },
                          ref_base, min_af, comb, n_comb, results);
}

void reap_timing(bvc_ctx *ctx, bool all)
{
    size_t kept = 0;
    for (size_t i = 0; i < ctx->ev_pending.size(); ++i) {
        bvc_ctx::Triple &t = ctx->ev_pending[i];
        if (!all && hipEventQuery(t.d) != hipSuccess) { (void)hipGetLastError(); ctx->ev_pending[kept++] = t; continue; }
        float ms1 = 0.f, ms2 = 0.f;
        if (hipEventElapsedTime(&ms1, t.a, t.b) == hipSuccess && hipEventElapsedTime(&ms2, t.c, t.d) == hipSuccess) {
            ctx->prof.hist_ms += ms1; ctx->prof.em_ms += ms2;
            ctx->prof.hist_launches += 1; ctx->prof.em_launches += 1; ctx->prof.sites += t.sites;
        } else {
            (void)hipGetLastError();
        }
        give_back(ctx, t.a); give_back(ctx, t.b); give_back(ctx, t.c); give_back(ctx, t.d);
    }
    ctx->ev_pending.resize(kept);
}

// Host-pointer calls go through device staging in chunks of sites; `upload(set, s0, ns)` enqueues the H2D copies of
// a chunk on the copy stream into staging set `set`, `compute(set, s0, ns)` enqueues its kernels and the D2H copies
// of its records on the context's stream.  The upload of chunk i+1 is issued before the (blocking) download of chunk
// i, so it runs under chunk i's kernels.
template <class Upload, class Compute>
int run_chunks(bvc_ctx *ctx, int64_t n_sites, int64_t chunk, Upload upload, Compute compute)
{
    // an early exit must not leave an upload or a kernel running on the staging sets: the next call may free or refill them
    auto drained = [&](int code) {
        if (code != BVC_OK) {
            // stage 2 of the chunks already launched may still run on the side streams and it reads / writes the staging sets
            (void)hipStreamSynchronize(ctx->copy);
            (void)hipStreamSynchronize(ctx->stream);
            (void)hipStreamSynchronize(ctx->side);
            (void)hipStreamSynchronize(ctx->side_b);
            (void)hipStreamSynchronize(ctx->side_c);
            for (bool &p : ctx->em_pending) p = false;
            (void)hipGetLastError();
        }
        return code;
    };
#define BVC_HIP_D(call)                                                                   \
    do {                                                                                  \
        hipError_t e__ = (call);                                                          \
        if (e__ != hipSuccess) return drained(fail(ctx, BVC_ERR_DEVICE, #call, e__));     \
    } while (0)
    int set = 0;
    int rc = upload(set, (int64_t)0, n_sites < chunk ? n_sites : chunk);
    if (rc != BVC_OK) return drained(rc);
    BVC_HIP_D(hipEventRecord(ctx->ev_upload[set], ctx->copy));
    for (int64_t s0 = 0; s0 < n_sites; s0 += chunk, set ^= 1) {
        const int64_t ns = n_sites - s0 < chunk ? n_sites - s0 : chunk;
        BVC_HIP_D(hipStreamWaitEvent(ctx->stream, ctx->ev_upload[set], 0));
        rc = compute(set, s0, ns, /*download=*/false);
        if (rc != BVC_OK) return drained(rc);
        const int64_t s1 = s0 + chunk;
        if (s1 < n_sites) {
            // the other set's previous chunk (i-1) has been downloaded synchronously below: it is free
            rc = upload(set ^ 1, s1, n_sites - s1 < chunk ? n_sites - s1 : chunk);
            if (rc != BVC_OK) return drained(rc);
            BVC_HIP_D(hipEventRecord(ctx->ev_upload[set ^ 1], ctx->copy));
        }
        rc = compute(set, s0, ns, /*download=*/true);
        if (rc != BVC_OK) return drained(rc);
        BVC_HIP_D(hipStreamSynchronize(ctx->stream));
    }
#undef BVC_HIP_D
    return BVC_OK;
}

// Group calls: stage 1 (`stage1(grp_counts)`: one pass, n_groups + 1 histograms per site, [site][n_groups + 1][512]) on the
// context's stream; stage 2 (sum, overall LRT, per-group LRT: src/BaseVarC.cpp:612-613, 617-661) on the same stream or, in
// overlap mode, on a side stream.  Dense tiles and ragged columns differ only in their stage 1.
template <class Stage1>
int run_group_stages(bvc_ctx *ctx, int64_t ns, int n_groups, bool long_rows, Stage1 stage1, const int8_t *r, double min_af,
                     bvc_site_result *res, bvc_group_result *gres)
{
    const int n_hist = n_groups + 1;
    const int buf = ctx->overlap ? ctx->flip : 0;
    if (ctx->overlap) ctx->flip = (ctx->flip + 1) % bvc_ctx::kRing;
    uint32_t **cp = &ctx->d_cnt[buf];
    size_t *ccap = &ctx->cnt_cap[buf];
    uint32_t **gp = &ctx->d_grp[buf];
    size_t *gcap = &ctx->grp_cap[buf];
    const size_t cbytes = (size_t)ns * BVC_NCLASS * sizeof(uint32_t), gbytes = cbytes * (size_t)n_hist;
    if (cbytes > *ccap || gbytes > *gcap) { int rj = join_side(ctx); if (rj != BVC_OK) return rj; }
    int rc2 = ensure(ctx, reinterpret_cast<void **>(cp), ccap, cbytes);
    if (rc2 != BVC_OK) return rc2;
    rc2 = ensure(ctx, reinterpret_cast<void **>(gp), gcap, gbytes);
    if (rc2 != BVC_OK) return rc2;
    void *em_scratch = nullptr, *emg_scratch = nullptr;
    rc2 = em_scratch_for(ctx, buf, ns, min_af, &em_scratch);
    if (rc2 != BVC_OK) return rc2;
    rc2 = em_scratch_for(ctx, buf, ns, min_af, &emg_scratch, n_groups);
    if (rc2 != BVC_OK) return rc2;
    if (ctx->overlap && ctx->em_pending[buf]) {
        BVC_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_em_done[buf], 0));
        ctx->em_pending[buf] = false;
    }
    bvc_ctx::Triple t{nullptr, nullptr, nullptr, nullptr, ns};
    const bool timed = ctx->profiling && take_timing_events(ctx, t);
    auto bail = [&](int code) { give_back(ctx, t.a); give_back(ctx, t.b); give_back(ctx, t.c); give_back(ctx, t.d); return code; };
#define BVC_HIP_T(call)                                                                     \
    do {                                                                                    \
        hipError_t e__ = (call);                                                            \
        if (e__ != hipSuccess) return bail(fail(ctx, BVC_ERR_DEVICE, #call, e__));          \
    } while (0)
    if (timed) BVC_HIP_T(hipEventRecord(t.a, ctx->stream));
    BVC_HIP_T(stage1(*gp));
    if (timed) BVC_HIP_T(hipEventRecord(t.b, ctx->stream));
    hipStream_t s2 = ctx->stream;
    if (ctx->overlap) {
        s2 = em_stream(ctx, 2);
        BVC_HIP_T(hipEventRecord(ctx->ev_hist_done[buf], ctx->stream));
        BVC_HIP_T(hipStreamWaitEvent(s2, ctx->ev_hist_done[buf], 0));
    }
    if (timed) BVC_HIP_T(hipEventRecord(t.c, s2));
    BVC_HIP_T(launch_sum_groups(s2, ns, n_hist, *gp, *cp));
    const bool shared = ctx->overlap && long_rows;
    const int per_launch = kGroupSharedWavesPerCu / em_stream_count(ctx, 2) > 2 ? kGroupSharedWavesPerCu / em_stream_count(ctx, 2) : 2;
    BVC_HIP_T(launch_lrt(ctx->ls, s2, ns, *cp, BVC_NCLASS, r, min_af, ctx->d_lut, nullptr, nullptr, res, shared, per_launch, em_scratch));
    BVC_HIP_T(launch_lrt_groups(ctx->ls, s2, ns, n_groups, *gp, r, min_af, ctx->d_lut, res, gres, shared, per_launch, emg_scratch));
    if (timed) {
        BVC_HIP_T(hipEventRecord(t.d, s2));
        ctx->ev_pending.push_back(t);
        t = bvc_ctx::Triple{nullptr, nullptr, nullptr, nullptr, 0};
        if (ctx->ev_pending.size() > 256) reap_timing(ctx, false);
    }
    if (ctx->overlap) {
        BVC_HIP_T(hipEventRecord(ctx->ev_em_done[buf], s2));
        ctx->em_pending[buf] = true;
    }
#undef BVC_HIP_T
    return BVC_OK;
}

// bvc_host_alloc's allocations: a transfer from / to a buffer inside one of them needs no bounce buffer
std::mutex g_pinned_mu;
std::vector<std::pair<const char *, size_t>> g_pinned;
bool in_pinned(const void *p, size_t n)
{
    std::lock_guard<std::mutex> g(g_pinned_mu);
    const char *c = static_cast<const char *>(p);
    for (auto const &r : g_pinned)
        if (c >= r.first && c + n <= r.first + r.second) return true;
    return false;
}

// Transfers of one call through the context's pinned buffers: h2d copies the caller's bytes into pinned memory and enqueues the DMA,
// d2h enqueues a DMA into pinned memory and deliver() -- after the stream has been waited for -- copies the bytes to the caller.
struct PinIO {
    bvc_ctx *ctx;
    size_t up_used = 0, down_used = 0;
    struct Out { void *dst; const char *src; size_t n; };
    std::vector<Out> outs;
    explicit PinIO(bvc_ctx *c) : ctx(c) {}
    static size_t al(size_t n) { return (n + 63) & ~(size_t)63; }
    int reserve(size_t up_bytes, size_t down_bytes)
    {
        auto grow = [&](char **buf, size_t *cap, size_t need) -> int {
            if (need <= *cap) return BVC_OK;
            if (*buf) {
                if (wait_stream(ctx) != hipSuccess) return fail(ctx, BVC_ERR_DEVICE, "wait before growing a pinned buffer");
                (void)hipHostFree(*buf);
                *buf = nullptr; *cap = 0;
            }
            const size_t want = need + need / 4 + 4096;
            if (hipHostMalloc(reinterpret_cast<void **>(buf), want, hipHostMallocDefault) != hipSuccess) {
                (void)hipGetLastError();
                return fail(ctx, BVC_ERR_ALLOC, "pinned host allocation failed");
            }
            *cap = want;
            return BVC_OK;
        };
        int rc = grow(&ctx->h_up, &ctx->up_cap, up_bytes);
        return rc == BVC_OK ? grow(&ctx->h_down, &ctx->down_cap, down_bytes) : rc;
    }
    hipError_t h2d(void *dev, const void *host, size_t n)
    {
        if (n == 0) return hipSuccess;
        if (in_pinned(host, n)) return hipMemcpyAsync(dev, host, n, hipMemcpyHostToDevice, ctx->stream);
        if (up_used + n > ctx->up_cap) return hipErrorOutOfMemory;
        char *p = ctx->h_up + up_used;
        std::memcpy(p, host, n);
        up_used += al(n);
        return hipMemcpyAsync(dev, p, n, hipMemcpyHostToDevice, ctx->stream);
    }
    hipError_t d2h(void *host, const void *dev, size_t n)
    {
        if (n == 0) return hipSuccess;
        if (down_used + n > ctx->down_cap) return hipErrorOutOfMemory;
        char *p = ctx->h_down + down_used;
        down_used += al(n);
        outs.push_back(Out{host, p, n});
        return hipMemcpyAsync(p, dev, n, hipMemcpyDeviceToHost, ctx->stream);
    }
    void deliver() { for (auto const &o : outs) std::memcpy(o.dst, o.src, o.n); outs.clear(); }
};

int check_common(bvc_ctx *ctx, int64_t n_sites, const void *a, const void *b, const void *c, const void *d)
{
    if (!ctx) return BVC_ERR_ARG;
    if (n_sites < 0) return fail(ctx, BVC_ERR_ARG, "n_sites < 0");
    if (n_sites > 0 && (!a || !b || !c || !d)) return fail(ctx, BVC_ERR_ARG, "null data pointer");
    if (n_sites > (int64_t)0x7FFFFFFF / 64) return fail(ctx, BVC_ERR_ARG, "too many sites in one call (split the tile)");
    BVC_HIP(ctx, hipSetDevice(ctx->device));
    return BVC_OK;
}

}  // namespace

extern "C" {

const char *bvc_version(void) { return "libbvc 0.3.0 (gfx950)"; }

int bvc_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    int ok = 0;
    for (int d = 0; d < n; ++d) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, d) == hipSuccess && std::strncmp(p.gcnArchName, "gfx950", 6) == 0) ++ok;
    }
    return ok;
}

static int env_int(const char *name, int lo, int hi, int dflt)
{
    const char *e = getenv(name);
    if (!e) return dflt;
    const int v = atoi(e);
    return (v >= lo && v <= hi) ? v : dflt;
}

int bvc_create(bvc_ctx **out, int device)
{
    if (!out) return BVC_ERR_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); return BVC_ERR_NO_DEVICE; }
    if (device < 0 || device >= n) return BVC_ERR_NO_DEVICE;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, device) != hipSuccess) return BVC_ERR_DEVICE;
    if (std::strncmp(p.gcnArchName, "gfx950", 6) != 0) return BVC_ERR_NO_DEVICE;   // code objects are gfx950 only
    bvc_ctx *ctx = new (std::nothrow) bvc_ctx();
    if (!ctx) return BVC_ERR_ALLOC;
    ctx->device = device;
#ifdef BVC_DIAG_KNOBS
    // experiment (host program on a CPU quota): let host threads SLEEP in hipStreamSynchronize instead of polling.  Process-wide,
    // which is why it is not a tuning of a context.
    if (env_int("BVC_BLOCKING_SYNC", 0, 1, 0)) (void)hipSetDeviceFlags(hipDeviceScheduleBlockingSync);
#endif
    ctx->ls.n_cu = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
    // starting values of the tuning knobs (bvc_set_tuning changes them per context; results never depend on them)
    ctx->ls.em_waves_per_cu = env_int("BVC_EM_WAVES_PER_CU", 1, 32, 0);
    ctx->ls.em_wpb = env_int("BVC_EM_WPB", 1, 4, 4) == 1 ? 1 : 4;
    ctx->ls.hist_split = env_int("BVC_HIST_SPLIT", 1, 64, 0);
    ctx->ls.group_pipe = env_int("BVC_GROUP_PIPE", 0, 1, 1);
    ctx->ls.group_log2c = env_int("BVC_GROUP_LOG2C", 0, 5, -1);
    ctx->ls.group_big_lds = env_int("BVC_GROUP_BIG_LDS", 0, 1, 1);
    ctx->ls.group_h16 = env_int("BVC_GROUP_H16", 0, 1, 0);
    ctx->ls.em_streams = env_int("BVC_EM_STREAMS", 0, 3, 0);
    ctx->ls.em_engine = env_int("BVC_EM_ENGINE", 0, 1, 0);
#ifdef BVC_DIAG_KNOBS
    // timing experiments only (tools/em_stage2.py phase breakdown): cuts region_kernel short, so the records are WRONG.  Not
    // compiled into the product: a stray environment variable must never be able to do that.
    ctx->ls.dbg_levels = env_int("BVC_DBG_LEVELS", 0, 12, 0);
#endif
    ctx->ls.em_tiny_regions = env_int("BVC_EM_TINY_REGIONS", 0, 1, 0);
    ctx->ls.em_prune = env_int("BVC_EM_PRUNE", 0, 1, 1);
    if (hipSetDevice(device) != hipSuccess) { delete ctx; return BVC_ERR_DEVICE; }
    // likelihood table from the host's exp(), as the CPU path computes it (src/BaseType.cpp:13,15)
    QualLut lut;
    for (int q = 0; q < 128; ++q) {
        const double eps = std::exp(-0.23025850929940458 * q);   // MLN10TO10, src/BaseType.h:10
        lut.a[q] = 1.0 - eps;
        lut.e[q] = eps / 3.0;
        lut.log_e[q] = std::log(lut.e[q]);
        lut.log_a[q] = std::log(lut.a[q]);                       // (-inf / NaN below quality 2: such sites never reach the item engine)
    }
    lut.e_empty = 0.25;
    if (hipMalloc(reinterpret_cast<void **>(&ctx->d_lut), sizeof(QualLut)) != hipSuccess ||
        hipMemcpy(ctx->d_lut, &lut, sizeof(QualLut), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipGetLastError();
        if (ctx->d_lut) (void)hipFree(ctx->d_lut);
        delete ctx;
        return BVC_ERR_ALLOC;
    }
    // Stage 2 runs on side streams underneath the histogram pass of the next call (overlap mode).  They get the LOWEST dispatch
    // priority: when wave slots free up, the histogram kernel of the next call -- which the next stage 2 is waiting for -- goes
    // first, instead of queueing behind two stage-2 launches that fill the chip (BVC_SIDE_PRIORITY=0: plain streams, A/B runs).
    int prio_least = 0, prio_greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest) != hipSuccess) { (void)hipGetLastError(); prio_least = 0; }
    const int side_prio = env_int("BVC_SIDE_PRIORITY", 0, 1, 1) ? prio_least : 0;
    bool ok = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamDefault) == hipSuccess &&
              hipEventCreateWithFlags(&ctx->ev_wait, hipEventBlockingSync | hipEventDisableTiming) == hipSuccess &&
              hipStreamCreateWithPriority(&ctx->side, hipStreamNonBlocking, side_prio) == hipSuccess &&
              hipStreamCreateWithPriority(&ctx->side_b, hipStreamNonBlocking, side_prio) == hipSuccess &&
              hipStreamCreateWithPriority(&ctx->side_c, hipStreamNonBlocking, side_prio) == hipSuccess &&
              hipStreamCreateWithFlags(&ctx->copy, hipStreamNonBlocking) == hipSuccess &&
              hipMalloc(reinterpret_cast<void **>(&ctx->d_grp_scratch), kGroupScratchWords * sizeof(int64_t)) == hipSuccess &&
              hipMalloc(reinterpret_cast<void **>(&ctx->d_sink), 256) == hipSuccess;
    for (int b = 0; b < bvc_ctx::kRing && ok; ++b)
        ok = hipEventCreateWithFlags(&ctx->ev_hist_done[b], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&ctx->ev_em_done[b], hipEventDisableTiming) == hipSuccess;
    for (int b = 0; b < 2 && ok; ++b) ok = hipEventCreateWithFlags(&ctx->ev_upload[b], hipEventDisableTiming) == hipSuccess;
    for (int b = 0; b < 2 && ok; ++b) ok = hipEventCreateWithFlags(&ctx->ev_set_free[b], hipEventDisableTiming) == hipSuccess;
    if (!ok) { (void)hipGetLastError(); bvc_destroy(ctx); return BVC_ERR_DEVICE; }
    ctx->stream = ctx->own_stream;
    *out = ctx;
    return BVC_OK;
}

void *bvc_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (bytes == 0) bytes = 1;
    if (hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    std::lock_guard<std::mutex> g(g_pinned_mu);
    g_pinned.push_back({static_cast<const char *>(p), bytes});
    return p;
}

void bvc_host_free(void *p)
{
    if (!p) return;
    {
        std::lock_guard<std::mutex> g(g_pinned_mu);
        for (size_t i = 0; i < g_pinned.size(); ++i)
            if (g_pinned[i].first == p) { g_pinned.erase(g_pinned.begin() + (long)i); break; }
    }
    (void)hipHostFree(p);
}

void bvc_destroy(bvc_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->side) { (void)hipStreamSynchronize(ctx->side); (void)hipStreamDestroy(ctx->side); }
    if (ctx->side_b) { (void)hipStreamSynchronize(ctx->side_b); (void)hipStreamDestroy(ctx->side_b); }
    if (ctx->side_c) { (void)hipStreamSynchronize(ctx->side_c); (void)hipStreamDestroy(ctx->side_c); }
    if (ctx->copy) { (void)hipStreamSynchronize(ctx->copy); (void)hipStreamDestroy(ctx->copy); }
    if (ctx->own_stream) { (void)hipStreamSynchronize(ctx->own_stream); (void)hipStreamDestroy(ctx->own_stream); }
    if (ctx->ev_wait) (void)hipEventDestroy(ctx->ev_wait);
    for (int b = 0; b < bvc_ctx::kRing; ++b) {
        if (ctx->ev_hist_done[b]) (void)hipEventDestroy(ctx->ev_hist_done[b]);
        if (ctx->ev_em_done[b]) (void)hipEventDestroy(ctx->ev_em_done[b]);
        if (ctx->d_cnt[b]) (void)hipFree(ctx->d_cnt[b]);
        if (ctx->d_grp[b]) (void)hipFree(ctx->d_grp[b]);
    }
    for (int b = 0; b <= bvc_ctx::kRing; ++b)
        if (ctx->d_em[b]) (void)hipFree(ctx->d_em[b]);
    for (int b = 0; b < bvc_ctx::kRing; ++b)
        if (ctx->d_emg[b]) (void)hipFree(ctx->d_emg[b]);
    for (int b = 0; b < 2; ++b) {
        if (ctx->ev_upload[b]) (void)hipEventDestroy(ctx->ev_upload[b]);
        if (ctx->ev_set_free[b]) (void)hipEventDestroy(ctx->ev_set_free[b]);
        if (ctx->d_stage[b]) (void)hipFree(ctx->d_stage[b]);
    }
    for (auto &t : ctx->ev_pending) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); (void)hipEventDestroy(t.c); (void)hipEventDestroy(t.d); }
    for (auto e : ctx->ev_pool) (void)hipEventDestroy(e);
    if (ctx->d_lut) (void)hipFree(ctx->d_lut);
    if (ctx->d_grp_scratch) (void)hipFree(ctx->d_grp_scratch);
    if (ctx->d_grp_labels) (void)hipFree(ctx->d_grp_labels);
    if (ctx->d_sink) (void)hipFree(ctx->d_sink);
    if (ctx->d_pl_text) (void)hipFree(ctx->d_pl_text);
    if (ctx->d_pl_meta) (void)hipFree(ctx->d_pl_meta);
    if (ctx->d_pl_out) (void)hipFree(ctx->d_pl_out);
    if (ctx->d_pl_called) (void)hipFree(ctx->d_pl_called);
    if (ctx->h_up) (void)hipHostFree(ctx->h_up);
    if (ctx->h_down) (void)hipHostFree(ctx->h_down);
    for (int k = 0; k < 2; ++k) if (ctx->d_pz_text[k]) (void)hipFree(ctx->d_pz_text[k]);
    if (ctx->d_pz_comp) (void)hipFree(ctx->d_pz_comp);
    delete ctx;
}

const char *bvc_last_error(const bvc_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int bvc_set_stream(bvc_ctx *ctx, void *hip_stream)
{
    if (!ctx) return BVC_ERR_ARG;
    BVC_HIP(ctx, hipSetDevice(ctx->device));
    BVC_HIP(ctx, hipStreamSynchronize(ctx->side));
    BVC_HIP(ctx, hipStreamSynchronize(ctx->side_b));
    BVC_HIP(ctx, hipStreamSynchronize(ctx->side_c));
    for (bool &p : ctx->em_pending) p = false;
    BVC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = reinterpret_cast<hipStream_t>(hip_stream);
    return BVC_OK;
}

int bvc_synchronize(bvc_ctx *ctx)
{
    if (!ctx) return BVC_ERR_ARG;
    BVC_HIP(ctx, hipSetDevice(ctx->device));
    BVC_HIP(ctx, hipStreamSynchronize(ctx->side));
    BVC_HIP(ctx, hipStreamSynchronize(ctx->side_b));
    BVC_HIP(ctx, hipStreamSynchronize(ctx->side_c));
    BVC_HIP(ctx, hipStreamSynchronize(ctx->copy));
    for (bool &p : ctx->em_pending) p = false;
    BVC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return BVC_OK;
}

int bvc_set_overlap(bvc_ctx *ctx, int on)
{
    if (!ctx) return BVC_ERR_ARG;
    BVC_HIP(ctx, hipSetDevice(ctx->device));
    int rc = join_side(ctx);
    if (rc != BVC_OK) return rc;
    ctx->overlap = on != 0;
    return BVC_OK;
}

int bvc_join(bvc_ctx *ctx)
{
    if (!ctx) return BVC_ERR_ARG;
    BVC_HIP(ctx, hipSetDevice(ctx->device));
    return join_side(ctx);
}

int bvc_set_profiling(bvc_ctx *ctx, int on)
{
    if (!ctx) return BVC_ERR_ARG;
    ctx->profiling = on != 0;
    return BVC_OK;
}

int bvc_get_profile(bvc_ctx *ctx, bvc_profile *out, int reset)
{
    if (!ctx || !out) return BVC_ERR_ARG;
    BVC_HIP(ctx, hipSetDevice(ctx->device));
    BVC_HIP(ctx, hipStreamSynchronize(ctx->side));
    BVC_HIP(ctx, hipStreamSynchronize(ctx->side_b));
    BVC_HIP(ctx, hipStreamSynchronize(ctx->side_c));
    BVC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    reap_timing(ctx, true);
    *out = ctx->prof;
    if (reset) ctx->prof = bvc_profile{};
    return BVC_OK;
}

int bvc_lrt_dense(bvc_ctx *ctx, int64_t n_sites, int64_t n_samples, int64_t row_stride,
                  const int8_t *bases, const int8_t *quals, const int8_t *ref_base,
                  double min_af, bvc_site_result *results, uint32_t flags)
{
    // rows of zero samples carry no data: their pointers may be null
    int rc = check_common(ctx, n_sites, n_samples ? bases : ref_base, n_samples ? quals : ref_base, ref_base, results);
    if (rc != BVC_OK) return rc;
    if (n_samples < 0 || row_stride < n_samples) return fail(ctx, BVC_ERR_ARG, "need 0 <= n_samples <= row_stride");
    if (n_sites == 0) return BVC_OK;
    if (flags & BVC_PTR_DEVICE)
        return run_dense_device(ctx, n_sites, n_samples, row_stride, bases, quals, ref_base, min_af, results);

    // host pointers: site chunks of at most ~512 MiB per array through two staging sets
    const int64_t row_bytes = row_stride > 0 ? row_stride : 1;
    int64_t chunk = ctx->ls.host_chunk_bytes / row_bytes;
    if (chunk < 1) chunk = 1;
    if (chunk > n_sites) chunk = n_sites;
    const size_t arr_al = ((size_t)chunk * (size_t)row_stride + 255) & ~(size_t)255;
    const size_t ref_al = ((size_t)chunk + 255) & ~(size_t)255;
    const size_t need = 2 * arr_al + ref_al + (size_t)chunk * sizeof(bvc_site_result) + 256;
    const int n_sets = n_sites > chunk ? 2 : 1;
    for (int k = 0; k < n_sets; ++k) {
        rc = ensure(ctx, reinterpret_cast<void **>(&ctx->d_stage[k]), &ctx->stage_cap[k], need);
        if (rc != BVC_OK) return rc;
    }
    auto d_b = [&](int set) { return reinterpret_cast<int8_t *>(ctx->d_stage[set]); };
    auto d_q = [&](int set) { return d_b(set) + arr_al; };
    auto d_r = [&](int set) { return d_q(set) + arr_al; };
    auto d_res = [&](int set) { return reinterpret_cast<bvc_site_result *>(d_r(set) + ref_al); };
    return run_chunks(ctx, n_sites, chunk,
        [&](int set, int64_t s0, int64_t ns) -> int {
            // the last row may be shorter than row_stride in the caller's allocation: copy exactly what is addressed
            const size_t bytes = n_samples ? (size_t)(ns - 1) * (size_t)row_stride + (size_t)n_samples : 0;
            if (bytes) BVC_HIP(ctx, hipMemcpyAsync(d_b(set), bases + s0 * row_stride, bytes, hipMemcpyHostToDevice, ctx->copy));
            if (bytes) BVC_HIP(ctx, hipMemcpyAsync(d_q(set), quals + s0 * row_stride, bytes, hipMemcpyHostToDevice, ctx->copy));
            BVC_HIP(ctx, hipMemcpyAsync(d_r(set), ref_base + s0, (size_t)ns, hipMemcpyHostToDevice, ctx->copy));
            return BVC_OK;
        },
        [&](int set, int64_t s0, int64_t ns, bool download) -> int {
            if (!download) {
                int rc2 = run_dense_device(ctx, ns, n_samples, row_stride, d_b(set), d_q(set), d_r(set), min_af, d_res(set));
                return rc2 == BVC_OK ? join_side(ctx) : rc2;
            }
            BVC_HIP(ctx, hipMemcpyAsync(results + s0, d_res(set), (size_t)ns * sizeof(bvc_site_result),
                                        hipMemcpyDeviceToHost, ctx->stream));
            return BVC_OK;
        });
}

int bvc_lrt_dense_packed(bvc_ctx *ctx, int64_t n_sites, int64_t n_samples, int64_t row_stride, const uint8_t *packed,
                         const int8_t *ref_base, double min_af, bvc_site_result *results, uint32_t flags)
{
    int rc = check_common(ctx, n_sites, n_samples ? (const void *)packed : (const void *)ref_base, ref_base, ref_base, results);
    if (rc != BVC_OK) return rc;
    if (n_samples < 0 || row_stride < n_samples) return fail(ctx, BVC_ERR_ARG, "need 0 <= n_samples <= row_stride");
    if (n_sites == 0) return BVC_OK;
    if (flags & BVC_PTR_DEVICE) return run_packed_device(ctx, n_sites, n_samples, row_stride, packed, ref_base, min_af, results);

    // host pointers: the chunked staging of bvc_lrt_dense with one array instead of two
    const int64_t row_bytes = row_stride > 0 ? row_stride : 1;
    int64_t chunk = ctx->ls.host_chunk_bytes / row_bytes;
    if (chunk < 1) chunk = 1;
    if (chunk > n_sites) chunk = n_sites;
    const size_t arr_al = ((size_t)chunk * (size_t)row_stride + 255) & ~(size_t)255;
    const size_t ref_al = ((size_t)chunk + 255) & ~(size_t)255;
    const size_t need = arr_al + ref_al + (size_t)chunk * sizeof(bvc_site_result) + 256;
    const int n_sets = n_sites > chunk ? 2 : 1;
    for (int k = 0; k < n_sets; ++k) {
        rc = ensure(ctx, reinterpret_cast<void **>(&ctx->d_stage[k]), &ctx->stage_cap[k], need);
        if (rc != BVC_OK) return rc;
    }
    auto d_p = [&](int set) { return reinterpret_cast<uint8_t *>(ctx->d_stage[set]); };
    auto d_r = [&](int set) { return reinterpret_cast<int8_t *>(d_p(set) + arr_al); };
    auto d_res = [&](int set) { return reinterpret_cast<bvc_site_result *>(d_r(set) + ref_al); };
    return run_chunks(ctx, n_sites, chunk,
        [&](int set, int64_t s0, int64_t ns) -> int {
            const size_t bytes = n_samples ? (size_t)(ns - 1) * (size_t)row_stride + (size_t)n_samples : 0;
            if (bytes) BVC_HIP(ctx, hipMemcpyAsync(d_p(set), packed + s0 * row_stride, bytes, hipMemcpyHostToDevice, ctx->copy));
            BVC_HIP(ctx, hipMemcpyAsync(d_r(set), ref_base + s0, (size_t)ns, hipMemcpyHostToDevice, ctx->copy));
            return BVC_OK;
        },
        [&](int set, int64_t s0, int64_t ns, bool download) -> int {
            if (!download) {
                int rc2 = run_packed_device(ctx, ns, n_samples, row_stride, d_p(set), d_r(set), min_af, d_res(set));
                return rc2 == BVC_OK ? join_side(ctx) : rc2;
            }
            BVC_HIP(ctx, hipMemcpyAsync(results + s0, d_res(set), (size_t)ns * sizeof(bvc_site_result),
                                        hipMemcpyDeviceToHost, ctx->stream));
            return BVC_OK;
        });
}

int bvc_pack_dense(bvc_ctx *ctx, int64_t n_sites, int64_t n_samples, int64_t row_stride, const int8_t *bases,
                   const int8_t *quals, int64_t packed_stride, uint8_t *packed, int64_t *n_unrepresentable, uint32_t flags)
{
    int rc = check_common(ctx, n_sites, n_samples ? (const void *)bases : (const void *)n_unrepresentable,
                          n_samples ? (const void *)quals : (const void *)n_unrepresentable,
                          n_samples ? (const void *)packed : (const void *)n_unrepresentable, n_unrepresentable);
    if (rc != BVC_OK) return rc;
    if (!n_unrepresentable) return fail(ctx, BVC_ERR_ARG, "null data pointer");
    if (n_samples < 0 || row_stride < n_samples || packed_stride < n_samples) return fail(ctx, BVC_ERR_ARG, "need 0 <= n_samples <= row strides");
    if (!(flags & BVC_PTR_DEVICE)) return fail(ctx, BVC_ERR_ARG, "bvc_pack_dense takes device pointers (a host producer writes base << 6 | qual itself)");
    *n_unrepresentable = 0;
    if (n_sites == 0 || n_samples == 0) return BVC_OK;
    unsigned long long *d_bad = reinterpret_cast<unsigned long long *>(ctx->d_sink) + 8;   // bytes 64..71 of the context's 256-byte sink
    BVC_HIP(ctx, hipMemsetAsync(d_bad, 0, sizeof(unsigned long long), ctx->stream));
    BVC_HIP(ctx, launch_pack_dense(ctx->stream, n_sites, n_samples, row_stride, bases, quals, packed_stride, packed, d_bad));
    unsigned long long bad = 0;
    BVC_HIP(ctx, hipMemcpyAsync(&bad, d_bad, sizeof bad, hipMemcpyDeviceToHost, ctx->stream));
    BVC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *n_unrepresentable = (int64_t)bad;
    return BVC_OK;
}

int bvc_hist_dense_packed(bvc_ctx *ctx, int64_t n_sites, int64_t n_samples, int64_t row_stride, const uint8_t *packed,
                          uint32_t *counts, uint32_t flags)
{
    int rc = check_common(ctx, n_sites, n_samples ? (const void *)packed : (const void *)counts, counts, counts, counts);
    if (rc != BVC_OK) return rc;
    if (n_samples < 0 || row_stride < n_samples) return fail(ctx, BVC_ERR_ARG, "need 0 <= n_samples <= row_stride");
    if (!(flags & BVC_PTR_DEVICE)) return fail(ctx, BVC_ERR_ARG, "bvc_hist_dense_packed takes device pointers");
    if (n_sites == 0) return BVC_OK;
    const int split = choose_hist_split(ctx->ls, n_sites, n_samples);
    if (split > 1) BVC_HIP(ctx, hipMemsetAsync(counts, 0, (size_t)n_sites * BVC_NCLASS * sizeof(uint32_t), ctx->stream));
    BVC_HIP(ctx, launch_hist_packed(ctx->ls, ctx->stream, n_sites, n_samples, row_stride, packed, counts, split));
    return BVC_OK;
}

int bvc_hist_dense(bvc_ctx *ctx, int64_t n_sites, int64_t n_samples, int64_t row_stride,
                   const int8_t *bases, const int8_t *quals, uint32_t *counts, uint32_t flags)
{
    int rc = check_common(ctx, n_sites, n_samples ? (const void *)bases : (const void *)counts,
                          n_samples ? (const void *)quals : (const void *)counts, counts, counts);
    if (rc != BVC_OK) return rc;
    if (n_samples < 0 || row_stride < n_samples) return fail(ctx, BVC_ERR_ARG, "need 0 <= n_samples <= row_stride");
    if (n_sites == 0) return BVC_OK;
    const int split = choose_hist_split(ctx->ls, n_sites, n_samples);
    const size_t cbytes = (size_t)n_sites * BVC_NCLASS * sizeof(uint32_t);
    if (flags & BVC_PTR_DEVICE) {
        if (split > 1) BVC_HIP(ctx, hipMemsetAsync(counts, 0, cbytes, ctx->stream));
        BVC_HIP(ctx, launch_hist_dense(ctx->ls, ctx->stream, n_sites, n_samples, row_stride, bases, quals, nullptr, 0, counts, split));
        return BVC_OK;
    }
    const size_t bytes = n_samples ? (size_t)(n_sites - 1) * (size_t)row_stride + (size_t)n_samples : 0;
    const size_t arr_al = (bytes + 255) & ~(size_t)255;
    rc = ensure(ctx, reinterpret_cast<void **>(&ctx->d_stage[0]), &ctx->stage_cap[0], 2 * arr_al + cbytes + 256);
    if (rc != BVC_OK) return rc;
    int8_t *d_b = reinterpret_cast<int8_t *>(ctx->d_stage[0]);
    int8_t *d_q = d_b + arr_al;
    uint32_t *d_c = reinterpret_cast<uint32_t *>(d_q + arr_al);
    if (bytes) BVC_HIP(ctx, hipMemcpyAsync(d_b, bases, bytes, hipMemcpyHostToDevice, ctx->stream));
    if (bytes) BVC_HIP(ctx, hipMemcpyAsync(d_q, quals, bytes, hipMemcpyHostToDevice, ctx->stream));
    if (split > 1) BVC_HIP(ctx, hipMemsetAsync(d_c, 0, cbytes, ctx->stream));
    BVC_HIP(ctx, launch_hist_dense(ctx->ls, ctx->stream, n_sites, n_samples, row_stride, d_b, d_q, nullptr, 0, d_c, split));
    BVC_HIP(ctx, hipMemcpyAsync(counts, d_c, cbytes, hipMemcpyDeviceToHost, ctx->stream));
    BVC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return BVC_OK;
}

static int check_comb_host(bvc_ctx *ctx, int64_t n_sites, const int8_t *base_comb, const uint8_t *n_comb)
{
    for (int64_t s = 0; s < n_sites; ++s) {
        if (n_comb[s] > 4) return fail(ctx, BVC_ERR_ARG, "n_comb > 4");
        for (int c = 0; c < n_comb[s]; ++c)
            if (base_comb[s * 4 + c] < 0 || base_comb[s * 4 + c] > 3) return fail(ctx, BVC_ERR_ARG, "base_comb entry outside 0..3");
    }
    return BVC_OK;
}

int bvc_lrt_hist(bvc_ctx *ctx, int64_t n_sites, const uint32_t *counts, const int8_t *ref_base,
                 double min_af, const int8_t *base_comb, const uint8_t *n_comb,
                 bvc_site_result *results, uint32_t flags)
{
    int rc = check_common(ctx, n_sites, counts, ref_base, results, results);
    if (rc != BVC_OK) return rc;
    if ((base_comb == nullptr) != (n_comb == nullptr)) return fail(ctx, BVC_ERR_ARG, "base_comb and n_comb go together");
    if (n_sites == 0) return BVC_OK;
    void *em_scratch = nullptr;
    rc = em_scratch_for(ctx, bvc_ctx::kRing, n_sites, min_af, &em_scratch);
    if (rc != BVC_OK) return rc;
    if (flags & BVC_PTR_DEVICE) {
        BVC_HIP(ctx, launch_lrt(ctx->ls, ctx->stream, n_sites, counts, BVC_NCLASS, ref_base, min_af, ctx->d_lut, base_comb,
                                n_comb, results, false, 0, em_scratch));
        return BVC_OK;
    }
    if (base_comb && (rc = check_comb_host(ctx, n_sites, base_comb, n_comb)) != BVC_OK) return rc;
    const size_t cbytes = (size_t)n_sites * BVC_NCLASS * sizeof(uint32_t);
    const size_t sa = ((size_t)n_sites + 255) & ~(size_t)255;
    const size_t ca = ((size_t)n_sites * 4 + 255) & ~(size_t)255;
    rc = ensure(ctx, reinterpret_cast<void **>(&ctx->d_stage[0]), &ctx->stage_cap[0],
                cbytes + 2 * sa + ca + (size_t)n_sites * sizeof(bvc_site_result) + 256);
    if (rc != BVC_OK) return rc;
    uint32_t *d_c = reinterpret_cast<uint32_t *>(ctx->d_stage[0]);
    int8_t *d_r = reinterpret_cast<int8_t *>(ctx->d_stage[0] + cbytes);
    uint8_t *d_nc = reinterpret_cast<uint8_t *>(d_r + sa);
    int8_t *d_cb = reinterpret_cast<int8_t *>(d_nc + sa);
    bvc_site_result *d_res = reinterpret_cast<bvc_site_result *>(d_cb + ca);
    BVC_HIP(ctx, hipMemcpyAsync(d_c, counts, cbytes, hipMemcpyHostToDevice, ctx->stream));
    BVC_HIP(ctx, hipMemcpyAsync(d_r, ref_base, (size_t)n_sites, hipMemcpyHostToDevice, ctx->stream));
    if (base_comb) {
        BVC_HIP(ctx, hipMemcpyAsync(d_nc, n_comb, (size_t)n_sites, hipMemcpyHostToDevice, ctx->stream));
        BVC_HIP(ctx, hipMemcpyAsync(d_cb, base_comb, (size_t)n_sites * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    BVC_HIP(ctx, launch_lrt(ctx->ls, ctx->stream, n_sites, d_c, BVC_NCLASS, d_r, min_af, ctx->d_lut,
                            base_comb ? d_cb : nullptr, base_comb ? d_nc : nullptr, d_res, false, 0, em_scratch));
    BVC_HIP(ctx, hipMemcpyAsync(results, d_res, (size_t)n_sites * sizeof(bvc_site_result), hipMemcpyDeviceToHost,
                                ctx->stream));
    BVC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return BVC_OK;
}

// Ragged host-pointer calls (bvc_lrt_csr, bvc_lrt_csr_comb, bvc_lrt_csr_packed).  quals == nullptr: packed observations.
// The per-site arrays (offsets, ref_base, candidate lists) go up once and the records come down once; the observations go
// through two staging sets in chunks of sites, the upload of chunk i + 1 (copy stream) under the kernels of chunk i, the
// sets handed back and forth by events -- the host blocks only in its uploads (pageable memory) and at the end.  A chunk's
// observations keep their element offsets: they are staged `lead` = offsets[s0] mod 256 bytes into the set and the kernels
// get the staging address minus offsets[s0] as their array base (never dereferenced outside the chunk), so every site
// sees the alignment it has in a one-piece call.  (Per-chunk uploads of the small arrays, a per-chunk download and a
// per-chunk synchronize cost 0.12 ms a chunk -- more than the kernels they were to hide; without them a chunk costs 0.06.)
static int run_csr_host(bvc_ctx *ctx, int64_t n_sites, const int64_t *offsets, const int8_t *bases, const int8_t *quals,
                        const int8_t *ref_base, double min_af, const int8_t *base_comb, const uint8_t *n_comb,
                        bvc_site_result *results)
{
    const int64_t total = offsets[n_sites];
    // bytes per chunk and array: host_chunk_kib (512 MiB by default).  Smaller chunks do not pay: a pageable upload of 400 MB
    // takes 7.07 ms (56.6 GB/s), the whole call 7.54 ms in one piece and 7.50 / 7.53 / 7.70 ms in 3 / 6 / 12 chunks -- what
    // the hidden kernels give, the extra uploads take (profiles/r03_host_pointer_ragged_chunks.txt).
    const int64_t target = ctx->ls.host_chunk_bytes;
    const int64_t mean = total / n_sites > 0 ? total / n_sites : 1;
    int64_t chunk = target / mean;
    if (chunk < 1) chunk = 1;
    if (chunk > n_sites) chunk = n_sites;
    int64_t widest = 0;
    for (int64_t s0 = 0; s0 < n_sites; s0 += chunk) {
        const int64_t s1 = s0 + chunk < n_sites ? s0 + chunk : n_sites;
        if (offsets[s1] - offsets[s0] > widest) widest = offsets[s1] - offsets[s0];
    }
    const int arrays = quals ? 2 : 1;
    const size_t arr_al = ((size_t)widest + 256 + 255) & ~(size_t)255;       // + the lead
    const size_t off_al = ((size_t)(n_sites + 1) * 8 + 255) & ~(size_t)255;
    const size_t site_al = ((size_t)n_sites + 255) & ~(size_t)255;
    const size_t comb_al = ((size_t)n_sites * 4 + 255) & ~(size_t)255;
    const size_t res_al = ((size_t)n_sites * sizeof(bvc_site_result) + 255) & ~(size_t)255;
    const size_t head = off_al + 2 * site_al + comb_al + res_al;             // set 0 carries the per-site arrays in front
    const int n_sets = n_sites > chunk ? 2 : 1;
    int rc = ensure(ctx, reinterpret_cast<void **>(&ctx->d_stage[0]), &ctx->stage_cap[0], head + (size_t)arrays * arr_al + 256);
    if (rc == BVC_OK && n_sets > 1) rc = ensure(ctx, reinterpret_cast<void **>(&ctx->d_stage[1]), &ctx->stage_cap[1], (size_t)arrays * arr_al + 256);
    if (rc != BVC_OK) return rc;
    int64_t *d_o = reinterpret_cast<int64_t *>(ctx->d_stage[0]);
    int8_t *d_r = reinterpret_cast<int8_t *>(ctx->d_stage[0] + off_al);
    uint8_t *d_nc = reinterpret_cast<uint8_t *>(d_r + site_al);
    int8_t *d_cb = reinterpret_cast<int8_t *>(d_nc + site_al);
    bvc_site_result *d_res = reinterpret_cast<bvc_site_result *>(d_cb + comb_al);
    auto d_b = [&](int set) { return reinterpret_cast<int8_t *>(ctx->d_stage[set]) + (set == 0 ? head : 0); };
    // an early exit must not leave an upload or a kernel running on the staging sets: the next call may free or refill them
    auto drained = [&](int code) {
        if (code != BVC_OK) {
            // stage 2 of the chunks already launched may still run on the side streams and it reads / writes the staging sets
            (void)hipStreamSynchronize(ctx->copy);
            (void)hipStreamSynchronize(ctx->stream);
            (void)hipStreamSynchronize(ctx->side);
            (void)hipStreamSynchronize(ctx->side_b);
            (void)hipStreamSynchronize(ctx->side_c);
            for (bool &p : ctx->em_pending) p = false;
            (void)hipGetLastError();
        }
        return code;
    };
#define BVC_HIP_D(call)                                                                   \
    do {                                                                                  \
        hipError_t e__ = (call);                                                          \
        if (e__ != hipSuccess) return drained(fail(ctx, BVC_ERR_DEVICE, #call, e__));     \
    } while (0)
    BVC_HIP_D(hipMemcpyAsync(d_o, offsets, (size_t)(n_sites + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    BVC_HIP_D(hipMemcpyAsync(d_r, ref_base, (size_t)n_sites, hipMemcpyHostToDevice, ctx->stream));
    if (base_comb) {
        BVC_HIP_D(hipMemcpyAsync(d_nc, n_comb, (size_t)n_sites, hipMemcpyHostToDevice, ctx->stream));
        BVC_HIP_D(hipMemcpyAsync(d_cb, base_comb, (size_t)n_sites * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    auto upload = [&](int set, int64_t s0, int64_t ns, bool reuse) -> int {
        const int64_t o0 = offsets[s0];
        const size_t bytes = (size_t)(offsets[s0 + ns] - o0), lead = (size_t)(o0 & 255);
        if (reuse) BVC_HIP_D(hipStreamWaitEvent(ctx->copy, ctx->ev_set_free[set], 0));   // the kernels of chunk i - 2 are done with it
        if (bytes) BVC_HIP_D(hipMemcpyAsync(d_b(set) + lead, bases + o0, bytes, hipMemcpyHostToDevice, ctx->copy));
        if (bytes && quals) BVC_HIP_D(hipMemcpyAsync(d_b(set) + arr_al + lead, quals + o0, bytes, hipMemcpyHostToDevice, ctx->copy));
        BVC_HIP_D(hipEventRecord(ctx->ev_upload[set], ctx->copy));
        return BVC_OK;
    };
    int set = 0;
    rc = upload(0, 0, n_sites < chunk ? n_sites : chunk, false);
    if (rc != BVC_OK) return rc;
    int64_t i = 0;
    for (int64_t s0 = 0; s0 < n_sites; s0 += chunk, set ^= 1, ++i) {
        const int64_t ns = n_sites - s0 < chunk ? n_sites - s0 : chunk;
        const int64_t o0 = offsets[s0];
        const uintptr_t shift = (uintptr_t)(o0 - (o0 & 255));                // array base = staging + lead - o0
        const int8_t *pb = reinterpret_cast<const int8_t *>(reinterpret_cast<uintptr_t>(d_b(set)) - shift);
        const int8_t *pq = quals ? reinterpret_cast<const int8_t *>(reinterpret_cast<uintptr_t>(d_b(set) + arr_al) - shift) : nullptr;
        BVC_HIP_D(hipStreamWaitEvent(ctx->stream, ctx->ev_upload[set], 0));
        rc = run_csr_device(ctx, ns, d_o + s0, pb, pq, d_r + s0, min_af, base_comb ? d_cb + s0 * 4 : nullptr,
                            base_comb ? d_nc + s0 : nullptr, d_res + s0);
        if (rc != BVC_OK) return drained(rc);
        // the histogram kernels are the only readers of the set and they run on the context's stream
        BVC_HIP_D(hipEventRecord(ctx->ev_set_free[set], ctx->stream));
        const int64_t s1 = s0 + chunk;
        if (s1 < n_sites) {
            rc = upload(set ^ 1, s1, n_sites - s1 < chunk ? n_sites - s1 : chunk, i >= 1);
            if (rc != BVC_OK) return drained(rc);
        }
    }
    rc = join_side(ctx);
    if (rc != BVC_OK) return drained(rc);
    BVC_HIP_D(hipMemcpyAsync(results, d_res, (size_t)n_sites * sizeof(bvc_site_result), hipMemcpyDeviceToHost, ctx->stream));
    BVC_HIP_D(hipStreamSynchronize(ctx->stream));
#undef BVC_HIP_D
    return BVC_OK;
}

int bvc_lrt_csr_comb(bvc_ctx *ctx, int64_t n_sites, const int64_t *offsets,
                     const int8_t *bases, const int8_t *quals, const int8_t *ref_base,
                     double min_af, const int8_t *base_comb, const uint8_t *n_comb,
                     bvc_site_result *results, uint32_t flags)
{
    int rc = check_common(ctx, n_sites, offsets, ref_base, results, results);
    if (rc != BVC_OK) return rc;
    if ((base_comb == nullptr) != (n_comb == nullptr)) return fail(ctx, BVC_ERR_ARG, "base_comb and n_comb go together");
    if (n_sites == 0) return BVC_OK;
    if (flags & BVC_PTR_DEVICE) {
        if (!bases || !quals) return fail(ctx, BVC_ERR_ARG, "null data pointer");
        return run_csr_device(ctx, n_sites, offsets, bases, quals, ref_base, min_af, base_comb, n_comb, results);
    }
    const int64_t total = offsets[n_sites];
    if (offsets[0] != 0 || total < 0) return fail(ctx, BVC_ERR_ARG, "offsets must start at 0 and be non-decreasing");
    for (int64_t s = 0; s < n_sites; ++s)
        if (offsets[s + 1] < offsets[s]) return fail(ctx, BVC_ERR_ARG, "offsets must start at 0 and be non-decreasing");
    if (total > 0 && (!bases || !quals)) return fail(ctx, BVC_ERR_ARG, "null data pointer");
    if (base_comb && (rc = check_comb_host(ctx, n_sites, base_comb, n_comb)) != BVC_OK) return rc;
    return run_csr_host(ctx, n_sites, offsets, bases, quals, ref_base, min_af, base_comb, n_comb, results);
}

int bvc_lrt_csr_packed(bvc_ctx *ctx, int64_t n_sites, const int64_t *offsets, const uint8_t *packed,
                       const int8_t *ref_base, double min_af, bvc_site_result *results, uint32_t flags)
{
    int rc = check_common(ctx, n_sites, offsets, ref_base, results, results);
    if (rc != BVC_OK) return rc;
    if (n_sites == 0) return BVC_OK;
    const int8_t *obs = reinterpret_cast<const int8_t *>(packed);
    if (flags & BVC_PTR_DEVICE) {
        if (!packed) return fail(ctx, BVC_ERR_ARG, "null data pointer");
        return run_csr_device(ctx, n_sites, offsets, obs, nullptr, ref_base, min_af, nullptr, nullptr, results);
    }
    const int64_t total = offsets[n_sites];
    if (offsets[0] != 0 || total < 0) return fail(ctx, BVC_ERR_ARG, "offsets must start at 0 and be non-decreasing");
    for (int64_t s = 0; s < n_sites; ++s)
        if (offsets[s + 1] < offsets[s]) return fail(ctx, BVC_ERR_ARG, "offsets must start at 0 and be non-decreasing");
    if (total > 0 && !packed) return fail(ctx, BVC_ERR_ARG, "null data pointer");
    // half the bytes of bvc_lrt_csr over the host link
    return run_csr_host(ctx, n_sites, offsets, obs, nullptr, ref_base, min_af, nullptr, nullptr, results);
}

int bvc_lrt_csr(bvc_ctx *ctx, int64_t n_sites, const int64_t *offsets,
                const int8_t *bases, const int8_t *quals, const int8_t *ref_base,
                double min_af, bvc_site_result *results, uint32_t flags)
{
    return bvc_lrt_csr_comb(ctx, n_sites, offsets, bases, quals, ref_base, min_af, nullptr, nullptr, results, flags);
}


// ---- ragged group calls ----------------------------------------------------------------------------------------------------
static int run_csr_groups_device(bvc_ctx *ctx, int64_t n_sites, const int64_t *offsets, const int8_t *bases, const int8_t *quals,
                                 const int32_t *sample_of_obs, const int8_t *ref_base, double min_af, const uint8_t *group_of_sample,
                                 int64_t n_samples, int32_t n_groups, bvc_site_result *results, bvc_group_result *grp_results)
{
    return run_group_stages(ctx, n_sites, n_groups, false,
                            [&](uint32_t *gp) {
                                return launch_hist_csr_groups(ctx->ls, ctx->stream, n_sites, offsets, bases, quals, sample_of_obs,
                                                              group_of_sample, n_samples, n_groups, gp);
                            },
                            ref_base, min_af, results, grp_results);
}

static inline size_t al256(size_t n) { return (n + 255) & ~(size_t)255; }

int bvc_lrt_csr_groups(bvc_ctx *ctx, int64_t n_sites, const int64_t *offsets, const int8_t *bases, const int8_t *quals,
                       const int32_t *sample_of_obs, const int8_t *ref_base, double min_af,
                       const uint8_t *group_of_sample, int64_t n_samples, int32_t n_groups,
                       bvc_site_result *results, bvc_group_result *grp_results, uint32_t flags)
{
    int rc = check_common(ctx, n_sites, offsets, ref_base, results, results);
    if (rc != BVC_OK) return rc;
    if (n_groups < 1 || n_groups > BVC_MAX_GROUPS) return fail(ctx, BVC_ERR_ARG, "n_groups must be 1..32");
    if (n_samples < 0 || (n_samples > 0 && !group_of_sample) || !grp_results) return fail(ctx, BVC_ERR_ARG, "null group pointer");
    if (n_sites == 0) return BVC_OK;
    if (flags & BVC_PTR_DEVICE) {
        if (!bases || !quals || !sample_of_obs) return fail(ctx, BVC_ERR_ARG, "null data pointer");
        return run_csr_groups_device(ctx, n_sites, offsets, bases, quals, sample_of_obs, ref_base, min_af, group_of_sample, n_samples,
                                     n_groups, results, grp_results);
    }
    const int64_t total = offsets[n_sites];
    if (offsets[0] != 0 || total < 0) return fail(ctx, BVC_ERR_ARG, "offsets must start at 0 and be non-decreasing");
    for (int64_t s = 0; s < n_sites; ++s)
        if (offsets[s + 1] < offsets[s]) return fail(ctx, BVC_ERR_ARG, "offsets must start at 0 and be non-decreasing");
    if (total > 0 && (!bases || !quals || !sample_of_obs)) return fail(ctx, BVC_ERR_ARG, "null data pointer");
    // one piece through staging set 0: offsets | ref | labels | bases | quals | samples | records | group records
    const size_t off_al = al256((size_t)(n_sites + 1) * 8), ref_al = al256((size_t)n_sites), g_al = al256((size_t)n_samples + 1);
    const size_t arr_al = al256((size_t)total + 16), smp_al = al256((size_t)total * 4 + 16);
    const size_t res_al = al256((size_t)n_sites * sizeof(bvc_site_result));
    const size_t need = off_al + ref_al + g_al + 2 * arr_al + smp_al + res_al + (size_t)n_sites * n_groups * sizeof(bvc_group_result) + 256;
    rc = ensure(ctx, reinterpret_cast<void **>(&ctx->d_stage[0]), &ctx->stage_cap[0], need);
    if (rc != BVC_OK) return rc;
    char *p = ctx->d_stage[0];
    int64_t *d_o = reinterpret_cast<int64_t *>(p); p += off_al;
    int8_t *d_r = reinterpret_cast<int8_t *>(p); p += ref_al;
    uint8_t *d_g = reinterpret_cast<uint8_t *>(p); p += g_al;
    int8_t *d_b = reinterpret_cast<int8_t *>(p); p += arr_al;
    int8_t *d_q = reinterpret_cast<int8_t *>(p); p += arr_al;
    int32_t *d_s = reinterpret_cast<int32_t *>(p); p += smp_al;
    bvc_site_result *d_res = reinterpret_cast<bvc_site_result *>(p); p += res_al;
    bvc_group_result *d_gres = reinterpret_cast<bvc_group_result *>(p);
    auto drained = [&](int code) {
        if (code != BVC_OK) {
            (void)hipStreamSynchronize(ctx->stream); (void)hipStreamSynchronize(ctx->side); (void)hipStreamSynchronize(ctx->side_b);
            (void)hipStreamSynchronize(ctx->side_c);
            for (bool &pnd : ctx->em_pending) pnd = false;
            (void)hipGetLastError();
        }
        return code;
    };
#define BVC_HIP_D(call)                                                                   \
    do {                                                                                  \
        hipError_t e__ = (call);                                                          \
        if (e__ != hipSuccess) return drained(fail(ctx, BVC_ERR_DEVICE, #call, e__));     \
    } while (0)
    BVC_HIP_D(hipMemcpyAsync(d_o, offsets, (size_t)(n_sites + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    BVC_HIP_D(hipMemcpyAsync(d_r, ref_base, (size_t)n_sites, hipMemcpyHostToDevice, ctx->stream));
    if (n_samples) BVC_HIP_D(hipMemcpyAsync(d_g, group_of_sample, (size_t)n_samples, hipMemcpyHostToDevice, ctx->stream));
    if (total) {
        BVC_HIP_D(hipMemcpyAsync(d_b, bases, (size_t)total, hipMemcpyHostToDevice, ctx->stream));
        BVC_HIP_D(hipMemcpyAsync(d_q, quals, (size_t)total, hipMemcpyHostToDevice, ctx->stream));
        BVC_HIP_D(hipMemcpyAsync(d_s, sample_of_obs, (size_t)total * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    rc = run_csr_groups_device(ctx, n_sites, d_o, d_b, d_q, d_s, d_r, min_af, d_g, n_samples, n_groups, d_res, d_gres);
    if (rc == BVC_OK) rc = join_side(ctx);
    if (rc != BVC_OK) return drained(rc);
    BVC_HIP_D(hipMemcpyAsync(results, d_res, (size_t)n_sites * sizeof(bvc_site_result), hipMemcpyDeviceToHost, ctx->stream));
    BVC_HIP_D(hipMemcpyAsync(grp_results, d_gres, (size_t)n_sites * n_groups * sizeof(bvc_group_result), hipMemcpyDeviceToHost, ctx->stream));
    BVC_HIP_D(hipStreamSynchronize(ctx->stream));
#undef BVC_HIP_D
    return BVC_OK;
}

// ---- BGZF blocks on the device (inflate_kernel.hip) ------------------------------------------------------------------------
int bvc_inflate_blocks(bvc_ctx *ctx, const uint8_t *comp, int64_t comp_bytes, const bvc_bgzf_block *blocks, int64_t n_blocks,
                       uint8_t *out, int64_t out_bytes, uint32_t *status, uint32_t flags)
{
    if (!ctx) return BVC_ERR_ARG;
    if (n_blocks < 0 || comp_bytes < 0 || out_bytes < 0) return fail(ctx, BVC_ERR_ARG, "negative size");
    if (n_blocks == 0) return BVC_OK;
    if (!comp || !blocks || !status || (!out && out_bytes > 0)) return fail(ctx, BVC_ERR_ARG, "null data pointer");
    BVC_HIP(ctx, hipSetDevice(ctx->device));
    if (flags & BVC_PTR_DEVICE) {
        BVC_HIP(ctx, launch_inflate(ctx->stream, comp, blocks, n_blocks, out, status));
        return BVC_OK;
    }
    for (int64_t i = 0; i < n_blocks; ++i) {
        const bvc_bgzf_block &b = blocks[i];
        if (b.comp_off < 0 || b.comp_len < 0 || b.comp_off + b.comp_len > comp_bytes || b.isize < 0 || b.isize > 65536 || b.out_off < 0 ||
            b.out_off + b.isize > out_bytes)
            return fail(ctx, BVC_ERR_ARG, "block outside its buffer");
    }
    const size_t c_al = al256((size_t)comp_bytes + 16), b_al = al256((size_t)n_blocks * sizeof(bvc_bgzf_block)), s_al = al256((size_t)n_blocks * 4);
    int rc = ensure(ctx, reinterpret_cast<void **>(&ctx->d_stage[0]), &ctx->stage_cap[0], c_al + b_al + s_al + (size_t)out_bytes + 256);
    if (rc != BVC_OK) return rc;
    char *p = ctx->d_stage[0];
    uint8_t *d_c = reinterpret_cast<uint8_t *>(p); p += c_al;
    bvc_bgzf_block *d_b = reinterpret_cast<bvc_bgzf_block *>(p); p += b_al;
    uint32_t *d_s = reinterpret_cast<uint32_t *>(p); p += s_al;
    uint8_t *d_o = reinterpret_cast<uint8_t *>(p);
    auto drained = [&](int code) { if (code != BVC_OK) { (void)hipStreamSynchronize(ctx->stream); (void)hipGetLastError(); } return code; };
#define BVC_HIP_D(call)                                                                   \
    do {                                                                                  \
        hipError_t e__ = (call);                                                          \
        if (e__ != hipSuccess) return drained(fail(ctx, BVC_ERR_DEVICE, #call, e__));     \
    } while (0)
    BVC_HIP_D(hipMemcpyAsync(d_c, comp, (size_t)comp_bytes, hipMemcpyHostToDevice, ctx->stream));
    BVC_HIP_D(hipMemcpyAsync(d_b, blocks, (size_t)n_blocks * sizeof(bvc_bgzf_block), hipMemcpyHostToDevice, ctx->stream));
    BVC_HIP_D(launch_inflate(ctx->stream, d_c, d_b, n_blocks, d_o, d_s));
    BVC_HIP_D(hipMemcpyAsync(status, d_s, (size_t)n_blocks * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (out_bytes) BVC_HIP_D(hipMemcpyAsync(out, d_o, (size_t)out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    BVC_HIP_D(hipStreamSynchronize(ctx->stream));
#undef BVC_HIP_D
    return BVC_OK;
}

// ---- temp-batch pileup text -> columns -> records (pileup_kernel.hip) -------------------------------------------------------
int bvc_pileup_begin(bvc_ctx *ctx, const char *text, int64_t text_bytes, const uint32_t *line_start,
                     const int32_t *sample0, const int32_t *n_in_batch, int32_t n_batches, int32_t n_positions,
                     int64_t *n_entries, int64_t *n_indels)
{
    if (!ctx) return BVC_ERR_ARG;
    ctx->pl_begun = false;
    if (n_batches < 0 || n_positions < 0 || text_bytes < 0 || !n_entries || !n_indels) return fail(ctx, BVC_ERR_ARG, "bad argument");
    if (text_bytes > (int64_t)0xFFFFFF00) return fail(ctx, BVC_ERR_ARG, "more than 4 GiB of text in one tile (use fewer positions)");
    const int64_t n_lines = (int64_t)n_batches * n_positions;
    if (n_lines > 0 && (!text || !line_start || !sample0 || !n_in_batch)) return fail(ctx, BVC_ERR_ARG, "null data pointer");
    if (n_positions > (int32_t)(0x7FFFFFFF / 64)) return fail(ctx, BVC_ERR_ARG, "too many positions in one call (split the tile)");
    BVC_HIP(ctx, hipSetDevice(ctx->device));
    *n_entries = 0; *n_indels = 0;
    // the line table is what the kernels index the text with: every line inside the text, the lines of a batch in order
    for (int32_t b = 0; b < n_batches; ++b) {
        const uint32_t *ls = line_start + (int64_t)b * (n_positions + 1);
        if (n_in_batch[b] < 0) return fail(ctx, BVC_ERR_ARG, "negative batch size");
        for (int32_t t = 0; t < n_positions; ++t)
            if (ls[t + 1] <= ls[t]) return fail(ctx, BVC_ERR_ARG, "line_start: every line holds at least its newline, lines of a batch ascend");
        if (n_positions > 0 && (int64_t)ls[n_positions] > text_bytes) return fail(ctx, BVC_ERR_ARG, "line_start points outside the text");
    }
    const size_t T = (size_t)n_positions, nb = (size_t)n_batches;
    const size_t ls_al = al256(nb * (T + 1) * 4), b_al = al256(nb * 4), lw_al = al256((size_t)n_lines * 16), st_al = 256;
    const size_t off_al = al256((T + 1) * 8), tal_al = al256(T * 32 * 4);
    int rc = ensure(ctx, reinterpret_cast<void **>(&ctx->d_pl_text), &ctx->pl_text_cap, (size_t)text_bytes + 64);
    if (rc == BVC_OK) rc = ensure(ctx, reinterpret_cast<void **>(&ctx->d_pl_meta), &ctx->pl_meta_cap, ls_al + 2 * b_al + lw_al + st_al + 2 * off_al + tal_al);
    if (rc != BVC_OK) return rc;
    char *p = ctx->d_pl_meta;
    PileupTile &P = ctx->pl;
    P = PileupTile{};
    P.text = reinterpret_cast<const uint8_t *>(ctx->d_pl_text);
    uint32_t *d_ls = reinterpret_cast<uint32_t *>(p); p += ls_al;
    int32_t *d_s0 = reinterpret_cast<int32_t *>(p); p += b_al;
    int32_t *d_nib = reinterpret_cast<int32_t *>(p); p += b_al;
    P.line_start = d_ls; P.sample0 = d_s0; P.n_in_batch = d_nib;
    P.line_words = reinterpret_cast<uint32_t *>(p); p += lw_al;
    P.status = reinterpret_cast<uint32_t *>(p);
    P.totals = reinterpret_cast<int64_t *>(p + 64); p += st_al;
    P.entry_off = reinterpret_cast<int64_t *>(p); p += off_al;
    P.obs_off = reinterpret_cast<int64_t *>(p); p += off_al;
    P.tally = reinterpret_cast<int32_t *>(p);
    P.n_batches = n_batches; P.n_pos = n_positions; P.n_lines_cap = n_lines; P.line_stride = n_positions + 1;
    auto drained = [&](int code) { if (code != BVC_OK) { (void)hipStreamSynchronize(ctx->stream); (void)hipGetLastError(); } return code; };
#define BVC_HIP_D(call)                                                                   \
    do {                                                                                  \
        hipError_t e__ = (call);                                                          \
        if (e__ != hipSuccess) return drained(fail(ctx, BVC_ERR_DEVICE, #call, e__));     \
    } while (0)
    BVC_HIP_D(hipMemsetAsync(P.status, 0, st_al + 2 * off_al + tal_al, ctx->stream));     // status, totals, offsets of an empty tile, tallies
    if (n_lines > 0) {
        BVC_HIP_D(hipMemcpyAsync(ctx->d_pl_text, text, (size_t)text_bytes, hipMemcpyHostToDevice, ctx->stream));
        BVC_HIP_D(hipMemcpyAsync(d_ls, line_start, nb * (T + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
        BVC_HIP_D(hipMemcpyAsync(d_s0, sample0, nb * 4, hipMemcpyHostToDevice, ctx->stream));
        BVC_HIP_D(hipMemcpyAsync(d_nib, n_in_batch, nb * 4, hipMemcpyHostToDevice, ctx->stream));
        BVC_HIP_D(launch_pileup_count(ctx->stream, P));
    }
    uint32_t st[4] = {0, 0, 0, 0};
    int64_t tot[2] = {0, 0};
    BVC_HIP_D(hipMemcpyAsync(st, P.status, sizeof st, hipMemcpyDeviceToHost, ctx->stream));
    BVC_HIP_D(hipMemcpyAsync(tot, P.totals, sizeof tot, hipMemcpyDeviceToHost, ctx->stream));
    BVC_HIP_D(wait_stream(ctx));
#undef BVC_HIP_D
    if (st[0] != 0) return BVC_PILEUP_IRREGULAR;
    ctx->pl_entries = tot[0]; ctx->pl_obs = tot[1]; ctx->pl_indels = st[1];
    ctx->pl_begun = true;
    ctx->pl_on_device_text = false; ctx->pl_indel_bytes = 0;
    *n_entries = tot[0]; *n_indels = st[1];
    return BVC_OK;
}

int bvc_pileup_begin_bgzf(bvc_ctx *ctx, const uint8_t *comp, int64_t comp_bytes, const bvc_bgzf_block *blocks,
                          const int32_t *blocks_of_batch, const int32_t *skip_bytes, const int32_t *sample0, const int32_t *n_in_batch,
                          int32_t n_batches, int32_t max_positions, int32_t reset, int32_t *n_positions, int32_t *lines_of_batch,
                          int64_t *n_entries, int64_t *n_indels, int64_t *indel_text_bytes)
{
    if (!ctx) return BVC_ERR_ARG;
    ctx->pl_begun = false;
    if (n_batches < 0 || max_positions < 0 || comp_bytes < 0 || !n_positions || !n_entries || !n_indels || !indel_text_bytes)
        return fail(ctx, BVC_ERR_ARG, "bad argument");
    if (n_batches > 0 && (!blocks_of_batch || !sample0 || !n_in_batch || !lines_of_batch)) return fail(ctx, BVC_ERR_ARG, "null data pointer");
    if (max_positions > (int32_t)(0x7FFFFFFF / 64)) return fail(ctx, BVC_ERR_ARG, "too many positions in one call (split the tile)");
    BVC_HIP(ctx, hipSetDevice(ctx->device));
    *n_positions = 0; *n_entries = 0; *n_indels = 0; *indel_text_bytes = 0;
    const size_t nb = (size_t)n_batches;
    if (reset || ctx->pz_left_len.size() != nb) { ctx->pz_left_len.assign(nb, 0u); ctx->pz_left_src.assign(nb, 0u); }
    int64_t n_blocks = 0;
    for (size_t b = 0; b < nb; ++b) { if (blocks_of_batch[b] < 0 || n_in_batch[b] < 0) return fail(ctx, BVC_ERR_ARG, "negative count"); n_blocks += blocks_of_batch[b]; }
    if (n_blocks > 0 && (!comp || !blocks)) return fail(ctx, BVC_ERR_ARG, "null data pointer");
    // layout of the new text buffer: per batch [left over | its new blocks' output], every region from a 16-byte boundary
    std::vector<bvc_bgzf_block> blk((size_t)n_blocks);
    std::vector<bvc_pileup_region> reg(nb);
    std::vector<uint32_t> seg_base(nb + 1);
    std::vector<uint32_t> region_end(nb);
    uint64_t at = 0;
    int64_t bi = 0;
    uint32_t segs = 0;
    for (size_t b = 0; b < nb; ++b) {
        at = (at + 15) & ~(uint64_t)15;
        const uint32_t left = ctx->pz_left_len[b];
        uint64_t fresh = 0;
        for (int32_t k = 0; k < blocks_of_batch[b]; ++k, ++bi) {
            const bvc_bgzf_block &in = blocks[bi];
            if (in.comp_off < 0 || in.comp_len < 0 || in.comp_off + in.comp_len > comp_bytes || in.isize < 0 || in.isize > 65536)
                return fail(ctx, BVC_ERR_ARG, "block outside its buffer");
            blk[(size_t)bi] = in;
            blk[(size_t)bi].out_off = (int64_t)(at + left + fresh);
            fresh += (uint64_t)in.isize;
        }
        uint32_t skip = 0;
        if (skip_bytes && left == 0 && skip_bytes[b] > 0) skip = (uint32_t)skip_bytes[b];
        if (skip > fresh) return fail(ctx, BVC_ERR_ARG, "skip_bytes beyond the batch's first blocks");
        if (at + left + fresh > (uint64_t)0xFFFFFF00u) return fail(ctx, BVC_ERR_ARG, "more than 4 GiB of text in one tile (send fewer blocks)");
        reg[b].start = (uint32_t)at + skip; reg[b].len = left + (uint32_t)fresh - skip; reg[b].left_src = ctx->pz_left_src[b]; reg[b].left_len = left;
        region_end[b] = reg[b].start + reg[b].len;
        seg_base[b] = segs;
        segs += (reg[b].len + 1023u) / 1024u;
        at += left + fresh;
    }
    seg_base[nb] = segs;
    const uint64_t text_bytes = at;
    const int nw = 1 - ctx->pz_cur;
    const size_t T = (size_t)max_positions;
    const int64_t n_lines_cap = (int64_t)max_positions * n_batches;
    const size_t ls_al = al256(nb * (T + 1) * 4), b_al = al256(nb * 4 + 4), lw_al = al256((size_t)n_lines_cap * 16), st_al = 256;
    const size_t off_al = al256((T + 1) * 8), tal_al = al256(T * 32 * 4);
    const size_t blk_al = al256((size_t)n_blocks * sizeof(bvc_bgzf_block)), bst_al = al256((size_t)n_blocks * 4), reg_al = al256(nb * sizeof(bvc_pileup_region));
    const size_t sb_al = al256((nb + 1) * 4), sn_al = al256((size_t)segs * 4 + 4);
    int rc = ensure(ctx, reinterpret_cast<void **>(&ctx->d_pz_text[nw]), &ctx->pz_text_cap[nw], (size_t)text_bytes + 64);
    if (rc == BVC_OK) rc = ensure(ctx, reinterpret_cast<void **>(&ctx->d_pz_comp), &ctx->pz_comp_cap, (size_t)comp_bytes + 64);
    if (rc == BVC_OK)
        rc = ensure(ctx, reinterpret_cast<void **>(&ctx->d_pl_meta), &ctx->pl_meta_cap,
                    ls_al + 2 * b_al + lw_al + st_al + 2 * off_al + tal_al + blk_al + bst_al + reg_al + sb_al + sn_al + 3 * b_al);
    if (rc != BVC_OK) return rc;
    char *p = ctx->d_pl_meta;
    PileupTile &P = ctx->pl;
    P = PileupTile{};
    P.text = reinterpret_cast<const uint8_t *>(ctx->d_pz_text[nw]);
    uint32_t *d_ls = reinterpret_cast<uint32_t *>(p); p += ls_al;
    int32_t *d_s0 = reinterpret_cast<int32_t *>(p); p += b_al;
    int32_t *d_nib = reinterpret_cast<int32_t *>(p); p += b_al;
    P.line_start = d_ls; P.sample0 = d_s0; P.n_in_batch = d_nib;
    P.line_words = reinterpret_cast<uint32_t *>(p); p += lw_al;
    P.status = reinterpret_cast<uint32_t *>(p);
    P.totals = reinterpret_cast<int64_t *>(p + 64);
    int32_t *d_T = reinterpret_cast<int32_t *>(p + 128); p += st_al;
    P.entry_off = reinterpret_cast<int64_t *>(p); p += off_al;
    P.obs_off = reinterpret_cast<int64_t *>(p); p += off_al;
    P.tally = reinterpret_cast<int32_t *>(p); p += tal_al;
    bvc_bgzf_block *d_blk = reinterpret_cast<bvc_bgzf_block *>(p); p += blk_al;
    uint32_t *d_bst = reinterpret_cast<uint32_t *>(p); p += bst_al;
    bvc_pileup_region *d_reg = reinterpret_cast<bvc_pileup_region *>(p); p += reg_al;
    uint32_t *d_sb = reinterpret_cast<uint32_t *>(p); p += sb_al;
    uint32_t *d_sn = reinterpret_cast<uint32_t *>(p); p += sn_al;
    int32_t *d_lines = reinterpret_cast<int32_t *>(p); p += b_al;
    uint32_t *d_ends = reinterpret_cast<uint32_t *>(p); p += b_al;
    P.n_batches = n_batches; P.n_pos = max_positions; P.n_lines_cap = n_lines_cap; P.line_stride = max_positions + 1; P.n_pos_dev = d_T;
    auto drained = [&](int code) { if (code != BVC_OK) { (void)hipStreamSynchronize(ctx->stream); (void)hipGetLastError(); } return code; };
#define BVC_HIP_D(call)                                                                   \
    do {                                                                                  \
        hipError_t e__ = (call);                                                          \
        if (e__ != hipSuccess) return drained(fail(ctx, BVC_ERR_DEVICE, #call, e__));     \
    } while (0)
    PinIO io(ctx);
    rc = io.reserve((n_blocks > 0 && in_pinned(comp, (size_t)comp_bytes) ? 0 : (size_t)comp_bytes) + (size_t)n_blocks * sizeof(bvc_bgzf_block) +
                        nb * (sizeof(bvc_pileup_region) + 12) + 1024,
                    (size_t)n_blocks * 4 + nb * 8 + 1024);
    if (rc != BVC_OK) return rc;
    BVC_HIP_D(hipMemsetAsync(P.status, 0, st_al + 2 * off_al + tal_al, ctx->stream));
    std::vector<uint32_t> bst((size_t)n_blocks);
    std::vector<uint32_t> ends(nb);
    uint32_t st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int64_t tot[2] = {0, 0};
    int32_t Tgot = 0;
    if (nb > 0) {
        if (n_blocks > 0) {
            BVC_HIP_D(io.h2d(ctx->d_pz_comp, comp, (size_t)comp_bytes));
            BVC_HIP_D(io.h2d(d_blk, blk.data(), (size_t)n_blocks * sizeof(bvc_bgzf_block)));
        }
        BVC_HIP_D(io.h2d(d_reg, reg.data(), nb * sizeof(bvc_pileup_region)));
        BVC_HIP_D(io.h2d(d_sb, seg_base.data(), (nb + 1) * 4));
        BVC_HIP_D(io.h2d(d_s0, sample0, nb * 4));
        BVC_HIP_D(io.h2d(d_nib, n_in_batch, nb * 4));
        BVC_HIP_D(launch_region_carry(ctx->stream, reinterpret_cast<const uint8_t *>(ctx->d_pz_text[ctx->pz_cur]),
                                      reinterpret_cast<uint8_t *>(ctx->d_pz_text[nw]), d_reg, n_batches));
        if (n_blocks > 0)
            BVC_HIP_D(launch_inflate(ctx->stream, reinterpret_cast<const uint8_t *>(ctx->d_pz_comp), d_blk, n_blocks,
                                     reinterpret_cast<uint8_t *>(ctx->d_pz_text[nw]), d_bst));
        BVC_HIP_D(launch_region_index(ctx->stream, P, d_reg, d_sb, (int64_t)segs, d_sn, d_lines, max_positions));
        BVC_HIP_D(launch_region_ends(ctx->stream, P, d_ends));
        BVC_HIP_D(launch_pileup_count(ctx->stream, P));
        if (n_blocks > 0) BVC_HIP_D(io.d2h(bst.data(), d_bst, (size_t)n_blocks * 4));
        BVC_HIP_D(io.d2h(lines_of_batch, d_lines, nb * 4));
        BVC_HIP_D(io.d2h(ends.data(), d_ends, nb * 4));
        BVC_HIP_D(io.d2h(&Tgot, d_T, 4));
    }
    BVC_HIP_D(io.d2h(st, P.status, sizeof st));
    BVC_HIP_D(io.d2h(tot, P.totals, sizeof tot));
    BVC_HIP_D(wait_stream(ctx));
    io.deliver();
#undef BVC_HIP_D
    for (int64_t i = 0; i < n_blocks; ++i)
        if (bst[(size_t)i] != 0) {
            ctx->pz_left_len.assign(nb, 0u);                     // the stream of this window is broken: nothing to carry on with
            return fail(ctx, BVC_ERR_DATA, bst[(size_t)i] == 10 ? "a BGZF block of a temp batch fails its CRC32"
                                                                : "a BGZF block of a temp batch is not valid deflate of its ISIZE bytes");
        }
    // what this tile leaves of every batch: from the end of its last line to the end of its region
    for (size_t b = 0; b < nb; ++b) { ctx->pz_left_src[b] = ends[b]; ctx->pz_left_len[b] = region_end[b] - ends[b]; }
    ctx->pz_cur = nw;
    P.n_pos = Tgot; P.n_pos_dev = nullptr;
    ctx->pl_text_bytes = (int64_t)text_bytes;
    ctx->pl_on_device_text = true;
    *n_positions = Tgot;
    if (Tgot == 0) return BVC_OK;
    if (st[0] != 0) return BVC_PILEUP_IRREGULAR;
    ctx->pl_entries = tot[0]; ctx->pl_obs = tot[1]; ctx->pl_indels = st[1]; ctx->pl_indel_bytes = st[4];
    ctx->pl_begun = true;
    *n_entries = tot[0]; *n_indels = st[1]; *indel_text_bytes = st[4];
    return BVC_OK;
}

int bvc_pileup_text(bvc_ctx *ctx, char *text, int64_t text_cap, int64_t *text_bytes_needed, uint32_t *line_start)
{
    if (!ctx || !text_bytes_needed) return BVC_ERR_ARG;
    if (!ctx->pl_on_device_text) return fail(ctx, BVC_ERR_ARG, "bvc_pileup_text without a tile from bvc_pileup_begin_bgzf");
    *text_bytes_needed = ctx->pl_text_bytes;
    if (!text) return BVC_OK;
    if (text_cap < ctx->pl_text_bytes || !line_start) return fail(ctx, BVC_ERR_ARG, "text buffer too small / null line table");
    BVC_HIP(ctx, hipSetDevice(ctx->device));
    const PileupTile &P = ctx->pl;
    if (ctx->pl_text_bytes) BVC_HIP(ctx, hipMemcpyAsync(text, P.text, (size_t)ctx->pl_text_bytes, hipMemcpyDeviceToHost, ctx->stream));
    if (P.n_batches > 0)
        BVC_HIP(ctx, hipMemcpy2DAsync(line_start, (size_t)(P.n_pos + 1) * 4, P.line_start, (size_t)P.line_stride * 4, (size_t)(P.n_pos + 1) * 4,
                                      (size_t)P.n_batches, hipMemcpyDeviceToHost, ctx->stream));
    BVC_HIP(ctx, wait_stream(ctx));
    return BVC_OK;
}

// bvc_pileup_finish (called_off = null: the entries of every position) and bvc_pileup_finish_called (the entries of the called
// positions only, compacted on the device; called_cap = room in entries / samples)
static int pileup_finish_impl(bvc_ctx *ctx, const int8_t *ref_base, double min_af, const uint8_t carry_in[5], uint8_t carry_out[5],
                              const uint8_t *group_of_sample, int64_t n_samples, int32_t n_groups,
                              int64_t *entry_off, int32_t *tally, int64_t *called_off, int64_t called_cap, bvc_pileup_entry *entries,
                              int32_t *samples, bvc_pileup_indel *indels, char *indel_text, bvc_site_result *results,
                              bvc_group_result *grp_results)
{
    if (!ctx) return BVC_ERR_ARG;
    if (!ctx->pl_begun) return fail(ctx, BVC_ERR_ARG, "bvc_pileup_finish without a bvc_pileup_begin that returned BVC_OK");
    if (ctx->pl_on_device_text && ctx->pl_indels > 0 && !indel_text) return fail(ctx, BVC_ERR_ARG, "indel_text is needed: the tile's text is on the device only");
    ctx->pl_begun = false;
    PileupTile &P = ctx->pl;
    const int64_t T = P.n_pos, n_e = ctx->pl_entries, n_o = ctx->pl_obs, n_i = ctx->pl_indels;
    if (!carry_in || !carry_out || !entry_off) return fail(ctx, BVC_ERR_ARG, "null pointer");
    if (T > 0 && (!ref_base || !tally || !results)) return fail(ctx, BVC_ERR_ARG, "null pointer");
    const bool called_only = called_off != nullptr;
    if ((n_e > 0 && !called_only && (!entries || !samples)) || (n_i > 0 && !indels)) return fail(ctx, BVC_ERR_ARG, "null pointer");
    if (called_only && (called_cap < 0 || (called_cap > 0 && (!entries || !samples)))) return fail(ctx, BVC_ERR_ARG, "null pointer");
    if (n_groups < 0 || n_groups > BVC_MAX_GROUPS) return fail(ctx, BVC_ERR_ARG, "n_groups must be 0..32");
    if (n_groups > 0 && (!grp_results || n_samples < 0 || (n_samples > 0 && !group_of_sample))) return fail(ctx, BVC_ERR_ARG, "null group pointer");
    BVC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t e_al = al256((size_t)n_e * sizeof(bvc_pileup_entry) + 16), s_al = al256((size_t)n_e * 4 + 16), o_al = al256((size_t)n_o + 16);
    const size_t os_al = al256((size_t)n_o * 4 + 16), i_al = al256((size_t)n_i * sizeof(bvc_pileup_indel) + 16), r_al = al256((size_t)T + 16);
    const size_t res_al = al256((size_t)T * sizeof(bvc_site_result)), g_al = al256((size_t)(n_groups ? n_samples : 0) + 16);
    const size_t gres_al = al256((size_t)T * (size_t)n_groups * sizeof(bvc_group_result));
    const size_t it_al = al256((size_t)(indel_text ? ctx->pl_indel_bytes : 0) + 16);
    const size_t co_al = al256(called_only ? (size_t)(T + 1) * 8 : 0);
    int rc = ensure(ctx, reinterpret_cast<void **>(&ctx->d_pl_out), &ctx->pl_out_cap,
                    e_al + s_al + 2 * o_al + os_al + i_al + r_al + res_al + g_al + gres_al + it_al + co_al + 256);
    if (rc != BVC_OK) return rc;
    char *p = ctx->d_pl_out;
    P.entries = reinterpret_cast<bvc_pileup_entry *>(p); p += e_al;
    P.samples = reinterpret_cast<int32_t *>(p); p += s_al;
    P.obs_base = reinterpret_cast<int8_t *>(p); p += o_al;
    P.obs_qual = reinterpret_cast<int8_t *>(p); p += o_al;
    P.obs_sample = reinterpret_cast<int32_t *>(p); p += os_al;
    P.indels = reinterpret_cast<bvc_pileup_indel *>(p); p += i_al;
    P.indel_cap = (uint32_t)n_i;
    int8_t *d_ref = reinterpret_cast<int8_t *>(p); p += r_al;
    bvc_site_result *d_res = reinterpret_cast<bvc_site_result *>(p); p += res_al;
    uint8_t *d_g = reinterpret_cast<uint8_t *>(p); p += g_al;
    bvc_group_result *d_gres = reinterpret_cast<bvc_group_result *>(p); p += gres_al;
    uint8_t *d_itext = reinterpret_cast<uint8_t *>(p); p += it_al;
    int64_t *d_called_off = reinterpret_cast<int64_t *>(p);
    const uint32_t cin = (uint32_t)(carry_in[0] & 7u) | ((uint32_t)(carry_in[4] & 1u) << 3) | 0x80u | ((uint32_t)carry_in[1] << 8) |
                         ((uint32_t)carry_in[2] << 16) | ((uint32_t)carry_in[3] << 24);
    auto drained = [&](int code) {
        if (code != BVC_OK) {
            (void)hipStreamSynchronize(ctx->stream); (void)hipStreamSynchronize(ctx->side); (void)hipStreamSynchronize(ctx->side_b);
            (void)hipStreamSynchronize(ctx->side_c);
            for (bool &pnd : ctx->em_pending) pnd = false;
            (void)hipGetLastError();
        }
        return code;
    };
#define BVC_HIP_D(call)                                                                   \
    do {                                                                                  \
        hipError_t e__ = (call);                                                          \
        if (e__ != hipSuccess) return drained(fail(ctx, BVC_ERR_DEVICE, #call, e__));     \
    } while (0)
    uint32_t cout = cin;
    PinIO io(ctx);
    rc = io.reserve((size_t)T + (size_t)(n_groups > 0 ? n_samples : 0) + 1024,
                    (size_t)T * (sizeof(bvc_site_result) + 32 * 4 + 8 + (size_t)n_groups * sizeof(bvc_group_result)) +
                        (size_t)(called_only ? 0 : n_e) * (sizeof(bvc_pileup_entry) + 4) + (called_only ? (size_t)(T + 1) * 8 : 0) +
                        (size_t)n_i * sizeof(bvc_pileup_indel) + (size_t)ctx->pl_indel_bytes + 4096);
    if (rc != BVC_OK) return rc;
    if (T > 0) {
        BVC_HIP_D(io.h2d(d_ref, ref_base, (size_t)T));
        if (n_groups > 0 && n_samples > 0) BVC_HIP_D(io.h2d(d_g, group_of_sample, (size_t)n_samples));
        if ((int64_t)P.n_pos * P.n_batches > 0) BVC_HIP_D(launch_pileup_write(ctx->stream, P, cin));
        if (n_groups > 0)
            rc = run_csr_groups_device(ctx, T, P.obs_off, P.obs_base, P.obs_qual, P.obs_sample, d_ref, min_af, d_g, n_samples, n_groups, d_res, d_gres);
        else
            rc = run_csr_device(ctx, T, P.obs_off, P.obs_base, P.obs_qual, d_ref, min_af, nullptr, nullptr, d_res);
        if (rc == BVC_OK) rc = join_side(ctx);
        if (rc != BVC_OK) return drained(rc);
        BVC_HIP_D(io.d2h(results, d_res, (size_t)T * sizeof(bvc_site_result)));
        if (n_groups > 0)
            BVC_HIP_D(io.d2h(grp_results, d_gres, (size_t)T * (size_t)n_groups * sizeof(bvc_group_result)));
        BVC_HIP_D(io.d2h(tally, P.tally, (size_t)T * 32 * 4));
        if (n_e && !called_only) {
            BVC_HIP_D(io.d2h(entries, P.entries, (size_t)n_e * sizeof(bvc_pileup_entry)));
            BVC_HIP_D(io.d2h(samples, P.samples, (size_t)n_e * 4));
        }
        if (called_only) {
            BVC_HIP_D(launch_called_scan(ctx->stream, P, d_res, d_called_off));
            BVC_HIP_D(io.d2h(called_off, d_called_off, (size_t)(T + 1) * 8));
        }
        if (n_i && indel_text) {
            BVC_HIP_D(launch_indel_text(ctx->stream, P, d_itext, (uint32_t)ctx->pl_indel_bytes, P.status + 5));
            if (ctx->pl_indel_bytes) BVC_HIP_D(io.d2h(indel_text, d_itext, (size_t)ctx->pl_indel_bytes));
        }
        if (n_i) BVC_HIP_D(io.d2h(indels, P.indels, (size_t)n_i * sizeof(bvc_pileup_indel)));
        if ((int64_t)P.n_pos * P.n_batches > 0) BVC_HIP_D(io.d2h(&cout, P.status + 3, 4));
    }
    BVC_HIP_D(io.d2h(entry_off, P.entry_off, (size_t)(T + 1) * 8));
    BVC_HIP_D(wait_stream(ctx));
    io.deliver();
    if (called_only && T == 0) called_off[0] = 0;
    if (called_only && T > 0 && called_off[T] > 0) {
        // the second trip: the called positions' entries, gathered on the device (typically a few per cent of the tile's)
        const int64_t n_c = called_off[T];
        if (n_c > called_cap) return fail(ctx, BVC_ERR_ARG, "called_cap is smaller than the entries of the called positions (n_entries of the begin call always suffices)");
        const size_t ce_al = al256((size_t)n_c * sizeof(bvc_pileup_entry));
        rc = ensure(ctx, reinterpret_cast<void **>(&ctx->d_pl_called), &ctx->pl_called_cap, ce_al + al256((size_t)n_c * 4));
        if (rc != BVC_OK) return rc;
        bvc_pileup_entry *d_ce = reinterpret_cast<bvc_pileup_entry *>(ctx->d_pl_called);
        int32_t *d_cs = reinterpret_cast<int32_t *>(ctx->d_pl_called + ce_al);
        io.down_used = 0;
        rc = io.reserve(0, (size_t)n_c * (sizeof(bvc_pileup_entry) + 4) + 4096);
        if (rc != BVC_OK) return rc;
        BVC_HIP_D(launch_called_gather(ctx->stream, P, d_called_off, d_ce, d_cs));
        BVC_HIP_D(io.d2h(entries, d_ce, (size_t)n_c * sizeof(bvc_pileup_entry)));
        BVC_HIP_D(io.d2h(samples, d_cs, (size_t)n_c * 4));
        BVC_HIP_D(wait_stream(ctx));
        io.deliver();
    }
#undef BVC_HIP_D
    carry_out[0] = (uint8_t)(cout & 7u); carry_out[1] = (uint8_t)(cout >> 8); carry_out[2] = (uint8_t)(cout >> 16);
    carry_out[3] = (uint8_t)(cout >> 24); carry_out[4] = (uint8_t)((cout >> 3) & 1u);
    return BVC_OK;
}

int bvc_pileup_finish(bvc_ctx *ctx, const int8_t *ref_base, double min_af, const uint8_t carry_in[5], uint8_t carry_out[5],
                      const uint8_t *group_of_sample, int64_t n_samples, int32_t n_groups,
                      int64_t *entry_off, int32_t *tally, bvc_pileup_entry *entries, int32_t *samples,
                      bvc_pileup_indel *indels, char *indel_text, bvc_site_result *results, bvc_group_result *grp_results)
{
    return pileup_finish_impl(ctx, ref_base, min_af, carry_in, carry_out, group_of_sample, n_samples, n_groups, entry_off, tally, nullptr, 0,
                              entries, samples, indels, indel_text, results, grp_results);
}

int bvc_pileup_finish_called(bvc_ctx *ctx, const int8_t *ref_base, double min_af, const uint8_t carry_in[5], uint8_t carry_out[5],
                             const uint8_t *group_of_sample, int64_t n_samples, int32_t n_groups,
                             int64_t *entry_off, int32_t *tally, int64_t *called_off, int64_t called_cap, bvc_pileup_entry *entries,
                             int32_t *samples, bvc_pileup_indel *indels, char *indel_text, bvc_site_result *results,
                             bvc_group_result *grp_results)
{
    if (ctx && !called_off) return fail(ctx, BVC_ERR_ARG, "null pointer");
    return pileup_finish_impl(ctx, ref_base, min_af, carry_in, carry_out, group_of_sample, n_samples, n_groups, entry_off, tally, called_off,
                              called_cap, entries, samples, indels, indel_text, results, grp_results);
}

// bvc_lrt_dense_groups and bvc_lrt_dense_groups_packed: `packed` = the tile is one byte per sample in `bases`
// (base << 6 | qual, include/bvc.h) and `quals` is not used.
static int lrt_groups_impl(bvc_ctx *ctx, bool packed, int64_t n_sites, int64_t n_samples, int64_t row_stride,
                           const int8_t *bases, const int8_t *quals, const int8_t *ref_base,
                           double min_af, const uint8_t *group_of_sample, int32_t n_groups,
                           bvc_site_result *results, bvc_group_result *grp_results, uint32_t flags)
{
    if (packed) quals = bases;
    // rows of zero samples carry no data: their pointers may be null
    int rc = check_common(ctx, n_sites, n_samples ? bases : ref_base, n_samples ? quals : ref_base, ref_base, results);
    if (rc != BVC_OK) return rc;
    if (n_samples < 0 || row_stride < n_samples) return fail(ctx, BVC_ERR_ARG, "need 0 <= n_samples <= row_stride");
    if (n_groups < 1 || n_groups > BVC_MAX_GROUPS) return fail(ctx, BVC_ERR_ARG, "n_groups must be 1..32");
    if (!group_of_sample || !grp_results) return fail(ctx, BVC_ERR_ARG, "null group pointer");
    if (n_sites == 0) return BVC_OK;
    auto run_device = [&](int64_t ns, const int8_t *b, const int8_t *q, const int8_t *r, const uint8_t *g,
                          bvc_site_result *res, bvc_group_result *gres) -> int {
        const size_t lbytes = group_labels_bytes(n_samples, ns);       // the call's labels clamped to 0..n_groups + a flag per site
        if (lbytes > ctx->grp_labels_cap) { int rj = join_side(ctx); if (rj != BVC_OK) return rj; }
        int rc2 = ensure(ctx, reinterpret_cast<void **>(&ctx->d_grp_labels), &ctx->grp_labels_cap, lbytes);
        if (rc2 != BVC_OK) return rc2;
        return run_group_stages(ctx, ns, n_groups, n_samples >= 200000,
                                [&](uint32_t *gp) {
                                    if (packed)
                                        return launch_hist_packed_groups(ctx->ls, ctx->stream, ns, n_samples, row_stride,
                                                                         reinterpret_cast<const uint8_t *>(b), g, n_groups, gp,
                                                                         ctx->d_grp_scratch, ctx->d_grp_labels);
                                    return launch_hist_dense(ctx->ls, ctx->stream, ns, n_samples, row_stride, b, q, g, n_groups, gp, 1,
                                                             ctx->d_grp_scratch, ctx->d_grp_labels);
                                },
                                r, min_af, res, gres);
    };

    if (flags & BVC_PTR_DEVICE)
        return run_device(n_sites, bases, quals, ref_base, group_of_sample, results, grp_results);

    const int64_t row_bytes = row_stride > 0 ? row_stride : 1;
    int64_t chunk = ctx->ls.host_chunk_bytes / row_bytes;
    if (chunk < 1) chunk = 1;
    if (chunk > n_sites) chunk = n_sites;
    const size_t arr_al = ((size_t)chunk * (size_t)row_stride + 255) & ~(size_t)255;
    const size_t ref_al = ((size_t)chunk + 255) & ~(size_t)255;
    const size_t g_al = ((size_t)n_samples + 255) & ~(size_t)255;
    const size_t res_al = ((size_t)chunk * sizeof(bvc_site_result) + 255) & ~(size_t)255;
    const size_t need = (packed ? 1 : 2) * arr_al + ref_al + g_al + res_al + (size_t)chunk * n_groups * sizeof(bvc_group_result) + 256;
    const int n_sets = n_sites > chunk ? 2 : 1;
    for (int k = 0; k < n_sets; ++k) {
        rc = ensure(ctx, reinterpret_cast<void **>(&ctx->d_stage[k]), &ctx->stage_cap[k], need);
        if (rc != BVC_OK) return rc;
    }
    auto d_b = [&](int set) { return reinterpret_cast<int8_t *>(ctx->d_stage[set]); };
    auto d_q = [&](int set) { return packed ? d_b(set) : d_b(set) + arr_al; };
    auto d_r = [&](int set) { return d_q(set) + arr_al; };
    auto d_g = [&](int set) { return reinterpret_cast<uint8_t *>(d_r(set) + ref_al); };
    auto d_res = [&](int set) { return reinterpret_cast<bvc_site_result *>(d_g(set) + g_al); };
    auto d_gres = [&](int set) { return reinterpret_cast<bvc_group_result *>(reinterpret_cast<char *>(d_res(set)) + res_al); };
    return run_chunks(ctx, n_sites, chunk,
        [&](int set, int64_t s0, int64_t ns) -> int {
            const size_t bytes = n_samples ? (size_t)(ns - 1) * (size_t)row_stride + (size_t)n_samples : 0;
            // the group vector travels with the first chunk of each staging set
            if (s0 < 2 * chunk && n_samples)
                BVC_HIP(ctx, hipMemcpyAsync(d_g(set), group_of_sample, (size_t)n_samples, hipMemcpyHostToDevice, ctx->copy));
            if (bytes) BVC_HIP(ctx, hipMemcpyAsync(d_b(set), bases + s0 * row_stride, bytes, hipMemcpyHostToDevice, ctx->copy));
            if (bytes && !packed) BVC_HIP(ctx, hipMemcpyAsync(d_q(set), quals + s0 * row_stride, bytes, hipMemcpyHostToDevice, ctx->copy));
            BVC_HIP(ctx, hipMemcpyAsync(d_r(set), ref_base + s0, (size_t)ns, hipMemcpyHostToDevice, ctx->copy));
            return BVC_OK;
        },
        [&](int set, int64_t s0, int64_t ns, bool download) -> int {
            if (!download) {
                int rc2 = run_device(ns, d_b(set), d_q(set), d_r(set), d_g(set), d_res(set), d_gres(set));
                return rc2 == BVC_OK ? join_side(ctx) : rc2;
            }
            BVC_HIP(ctx, hipMemcpyAsync(results + s0, d_res(set), (size_t)ns * sizeof(bvc_site_result), hipMemcpyDeviceToHost,
                                        ctx->stream));
            BVC_HIP(ctx, hipMemcpyAsync(grp_results + s0 * n_groups, d_gres(set), (size_t)ns * n_groups * sizeof(bvc_group_result),
                                        hipMemcpyDeviceToHost, ctx->stream));
            return BVC_OK;
        });
}

int bvc_lrt_dense_groups(bvc_ctx *ctx, int64_t n_sites, int64_t n_samples, int64_t row_stride,
                         const int8_t *bases, const int8_t *quals, const int8_t *ref_base,
                         double min_af, const uint8_t *group_of_sample, int32_t n_groups,
                         bvc_site_result *results, bvc_group_result *grp_results, uint32_t flags)
{
    return lrt_groups_impl(ctx, false, n_sites, n_samples, row_stride, bases, quals, ref_base, min_af, group_of_sample, n_groups,
                           results, grp_results, flags);
}

int bvc_lrt_dense_groups_packed(bvc_ctx *ctx, int64_t n_sites, int64_t n_samples, int64_t row_stride,
                                const uint8_t *packed, const int8_t *ref_base, double min_af,
                                const uint8_t *group_of_sample, int32_t n_groups,
                                bvc_site_result *results, bvc_group_result *grp_results, uint32_t flags)
{
    return lrt_groups_impl(ctx, true, n_sites, n_samples, row_stride, reinterpret_cast<const int8_t *>(packed), nullptr, ref_base,
                           min_af, group_of_sample, n_groups, results, grp_results, flags);
}

#ifdef BVC_CHECK_LDS
// Diagnostic builds only (bvc_device.h): the violations the checked kernels recorded, 8 words per translation unit
// (histogram kernels, wave engine, item engine): [0] count, [1..5] the first one's check id, value, limit, blockIdx.x,
// threadIdx.x.  Synchronises the device.  reset != 0 clears the records.
int bvc_debug_report(bvc_ctx *ctx, uint32_t *out24, int reset)
{
    if (!out24) return BVC_ERR_ARG;
    if (ctx) BVC_HIP(ctx, hipSetDevice(ctx->device));            // null: the calling thread's current device
    BVC_HIP(ctx, hipDeviceSynchronize());
    BVC_HIP(ctx, debug_read_hist(out24, reset != 0));
    BVC_HIP(ctx, debug_read_wave_engine(out24 + 8, reset != 0));
    BVC_HIP(ctx, debug_read_items(out24 + 16, reset != 0));
    uint32_t pl[8];                                              // pileup_kernel.hip: folded into the histogram unit's count
    BVC_HIP(ctx, debug_read_pileup(pl, reset != 0));
    if (pl[0]) { if (out24[0] == 0) for (int i = 1; i < 8; ++i) out24[i] = pl[i]; out24[0] += pl[0]; }
    BVC_HIP(ctx, debug_read_inflate(pl, reset != 0));
    if (pl[0]) { if (out24[0] == 0) for (int i = 1; i < 8; ++i) out24[i] = pl[i]; out24[0] += pl[0]; }
    return BVC_OK;
}
#endif

int bvc_set_tuning(bvc_ctx *ctx, const char *key, int value)
{
    if (!ctx) return BVC_ERR_ARG;
    if (!key) return fail(ctx, BVC_ERR_ARG, "null tuning key");
    BVC_HIP(ctx, hipSetDevice(ctx->device));                     // "em_streams" joins the side streams of THIS device
    if (std::strcmp(key, "em_waves_per_cu") == 0 && value >= 0 && value <= 32) { ctx->ls.em_waves_per_cu = value; return BVC_OK; }
    if (std::strcmp(key, "em_wpb") == 0 && (value == 1 || value == 4)) { ctx->ls.em_wpb = value; return BVC_OK; }
    if (std::strcmp(key, "hist_split") == 0 && value >= 0 && value <= 64) { ctx->ls.hist_split = value; return BVC_OK; }
    if (std::strcmp(key, "group_pipe") == 0 && (value == 0 || value == 1)) { ctx->ls.group_pipe = value; return BVC_OK; }
    if (std::strcmp(key, "group_copies_log2") == 0 && value >= -1 && value <= 5) { ctx->ls.group_log2c = value; return BVC_OK; }
    if (std::strcmp(key, "group_big_lds") == 0 && (value == 0 || value == 1)) { ctx->ls.group_big_lds = value; return BVC_OK; }
    if (std::strcmp(key, "group_h16") == 0 && (value == 0 || value == 1)) { ctx->ls.group_h16 = value; return BVC_OK; }
    if (std::strcmp(key, "em_engine") == 0 && value >= 0 && value <= 1) { ctx->ls.em_engine = value; return BVC_OK; }
    if (std::strcmp(key, "em_tiny_regions") == 0 && value >= 0 && value <= 1) { ctx->ls.em_tiny_regions = value; return BVC_OK; }
    if (std::strcmp(key, "em_prune") == 0 && value >= 0 && value <= 1) { ctx->ls.em_prune = value; return BVC_OK; }
    if (std::strcmp(key, "em_streams") == 0 && value >= 0 && value <= 3) {
        int rcj = join_side(ctx);
        if (rcj != BVC_OK) return rcj;
        ctx->ls.em_streams = value;
        return BVC_OK;
    }
    if (std::strcmp(key, "host_chunk_kib") == 0 && value >= 1 && value <= (1 << 21)) { ctx->ls.host_chunk_bytes = (int64_t)value << 10; return BVC_OK; }
    return fail(ctx, BVC_ERR_ARG, "unknown tuning key or value out of range");
}

int bvc_stream_read_ms(bvc_ctx *ctx, const void *device_ptr, int64_t bytes, int repeats, double *ms_per_pass)
{
    if (!ctx || !device_ptr || !ms_per_pass || bytes < 16 || repeats < 1) return ctx ? fail(ctx, BVC_ERR_ARG, "bad argument") : BVC_ERR_ARG;
    BVC_HIP(ctx, hipSetDevice(ctx->device));
    hipEvent_t a = take_event(ctx), b = take_event(ctx);
    if (!a || !b) { give_back(ctx, a); give_back(ctx, b); return fail(ctx, BVC_ERR_DEVICE, "hipEventCreate failed"); }
    auto bail = [&](int code) { give_back(ctx, a); give_back(ctx, b); return code; };
    hipError_t e = launch_stream_read(ctx->stream, device_ptr, bytes, ctx->d_sink);        // warm-up
    if (e == hipSuccess) e = hipEventRecord(a, ctx->stream);
    for (int i = 0; i < repeats && e == hipSuccess; ++i) e = launch_stream_read(ctx->stream, device_ptr, bytes, ctx->d_sink);
    if (e == hipSuccess) e = hipEventRecord(b, ctx->stream);
    if (e == hipSuccess) e = hipEventSynchronize(b);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, a, b);
    if (e != hipSuccess) return bail(fail(ctx, BVC_ERR_DEVICE, "stream read measurement", e));
    give_back(ctx, a); give_back(ctx, b);
    *ms_per_pass = (double)ms / repeats;
    return BVC_OK;
}

int bvc_synth_dense(bvc_ctx *ctx, uint64_t seed, int64_t site0, int64_t n_sites, int64_t n_samples,
                    int64_t row_stride, uint32_t cov_thr16, int8_t *bases, int8_t *quals, int8_t *ref_base)
{
    int rc = check_common(ctx, n_sites, bases, quals, ref_base, ref_base);
    if (rc != BVC_OK) return rc;
    if (n_samples < 0 || row_stride < n_samples) return fail(ctx, BVC_ERR_ARG, "need 0 <= n_samples <= row_stride");
    BVC_HIP(ctx, launch_synth_dense(ctx->stream, seed, site0, n_sites, n_samples, row_stride, cov_thr16, bases, quals,
                                    ref_base));
    return BVC_OK;
}

}  // extern "C"
