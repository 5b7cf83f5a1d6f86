// capi.cpp -- implementation of include/sketchlib_dist.h on top of kernels.hip.
//
// Host-side responsibilities: device slabs (the MultiSketch bins in the two layouts
// the pair kernel wants), the samebits -> ln(J) / distance tables (computed with the
// host libm, i.e. bit-identical to what the reference's f64::ln produces on this
// machine), banding of large pair spaces through bounded scratch, and mapping the
// reference's panics to status codes.
//
// There is deliberately no CPU compute path here: if HIP cannot run, calls fail.
#include "capi_internal.hpp"
#include "glibc_log.hpp"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <vector>

using namespace skl;

// ---------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------

static thread_local std::string g_last_error;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}



extern "C" const char *skl_last_error(void) { return g_last_error.c_str(); }
extern "C" int skl_abi_version(void) { return SKL_ABI_VERSION; }

static bool is_gfx950(int dev)
{
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return false;
    return strncmp(prop.gcnArchName, "gfx950", 6) == 0;
}

extern "C" int skl_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    int ok = 0;
    for (int d = 0; d < n; ++d) ok += is_gfx950(d) ? 1 : 0;
    return ok;
}

// ---------------------------------------------------------------------------
// handle registry: destroy calls in any order (and twice) must be harmless -- a binding's
// finalisers run in arbitrary order at interpreter shutdown.
// ---------------------------------------------------------------------------

static std::mutex g_registry_mutex;
static std::set<const void *> g_live_ctx, g_live_sketches;

// ---------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------

int ctx_bind(skl_ctx *ctx)
{
    if (!ctx) return fail(SKL_ERR_INVALID_ARG, "null context");
    HIP_TRY(hipSetDevice(ctx->device));
#ifdef SKL_AB
    // A/B build only: one process interleaves variants (scripts/ab_sweep.py), so the switches are read at
    // every API entry -- before any dispatch decision of the call, not in the middle of it
    ctx->knobs = read_knobs();
#endif
    return SKL_OK;
}

int ctx_scratch(skl_ctx *ctx, size_t bytes, void **out, int which)
{
    void *&buf = ctx->scratch[which];
    size_t &cap = ctx->scratch_bytes[which];
    if (bytes > cap) {
        if (buf) {
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            HIP_TRY(hipFree(buf));
            buf = nullptr;
            cap = 0;
        }
        HIP_TRY(hipMalloc(&buf, bytes));
        cap = bytes;
        if (which == 1) ctx->clean_plane1 = nullptr;   // a new counts scratch: nothing is known to be zero in it
    }
    *out = buf;
    return SKL_OK;
}

extern "C" int skl_ctx_create(int device, skl_ctx **out)
{
    if (!out) return fail(SKL_ERR_INVALID_ARG, "out is null");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
        return fail(SKL_ERR_NO_DEVICE, "no HIP device is visible; this library has no CPU path");
    }
    if (device < 0 || device >= n) {
        return fail(SKL_ERR_INVALID_ARG, "device %d out of range (0..%d)", device, n - 1);
    }
    if (!is_gfx950(device)) {
        return fail(SKL_ERR_NO_DEVICE, "device %d is not gfx950 (MI355X); kernels are built for gfx950 only",
                    device);
    }
    HIP_TRY(hipSetDevice(device));
    skl_ctx *ctx = new skl_ctx();
    ctx->device = device;
    ctx->knobs = read_knobs();
    ctx->timing_every = ctx->knobs.timing_every;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ctx->n_cu = prop.multiProcessorCount;
    }
    hipError_t e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete ctx;
        return fail(SKL_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    ctx->stream = ctx->own_stream;
    e = hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking);
    for (int x = 0; x < 2 && e == hipSuccess; ++x) {
        e = hipEventCreateWithFlags(&ctx->knn_pair_done[x], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->knn_topk_done[x], hipEventDisableTiming);
    }
    if (e != hipSuccess) {
        delete ctx;
        return fail(SKL_ERR_HIP, "stream/event creation: %s", hipGetErrorString(e));
    }
    {
        std::lock_guard<std::mutex> lock(g_registry_mutex);
        g_live_ctx.insert(ctx);
    }
    *out = ctx;
    return SKL_OK;
}

static void free_sketches_locked(skl_sketches *s);
void free_plans(skl_ctx *ctx);
uint64_t next_generation();

extern "C" int skl_ctx_destroy(skl_ctx *ctx)
{
    if (!ctx) return SKL_OK;
    std::lock_guard<std::mutex> lock(g_registry_mutex);
    if (!g_live_ctx.erase(ctx)) return SKL_OK;  // already destroyed
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->aux_stream) (void)hipStreamSynchronize(ctx->aux_stream);
    if (ctx->epi_stream) (void)hipStreamSynchronize(ctx->epi_stream);
    // slabs die with their context; their handles become inert
    const std::set<skl_sketches *> owned = ctx->sketches;
    for (skl_sketches *s : owned) free_sketches_locked(s);
    for (auto &ev : ctx->events) {
        (void)hipEventDestroy(ev.first);
        (void)hipEventDestroy(ev.second);
    }
    for (void *buf : ctx->scratch) {
        if (buf) (void)hipFree(buf);
    }
    if (ctx->sampler_stop) {
        *(volatile uint32_t *)ctx->sampler_stop = 1u;   // a sampler still running ends at its next poll
        if (ctx->sampler_stream) (void)hipStreamSynchronize(ctx->sampler_stream);
        (void)hipHostFree(ctx->sampler_stop);
    }
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    free_plans(ctx);
    for (hipEvent_t ev : ctx->eb_events) {
        if (ev) (void)hipEventDestroy(ev);
    }
    if (ctx->eb_counter) (void)hipFree(ctx->eb_counter);
    if (ctx->sampler_buf) (void)hipFree(ctx->sampler_buf);
    if (ctx->sampler_count) (void)hipFree(ctx->sampler_count);
    if (ctx->sampler_stream) (void)hipStreamDestroy(ctx->sampler_stream);
    if (ctx->tile_scratch.d_prefix) (void)hipFree(ctx->tile_scratch.d_prefix);
    if (ctx->tile_scratch.h_staging) (void)hipHostFree(ctx->tile_scratch.h_staging);
    if (ctx->tile_scratch.staged) (void)hipEventDestroy(ctx->tile_scratch.staged);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    if (ctx->aux_stream) (void)hipStreamDestroy(ctx->aux_stream);
    if (ctx->epi_stream) (void)hipStreamDestroy(ctx->epi_stream);
    for (int x = 0; x < 2; ++x) {
        if (ctx->knn_pair_done[x]) (void)hipEventDestroy(ctx->knn_pair_done[x]);
        if (ctx->knn_topk_done[x]) (void)hipEventDestroy(ctx->knn_topk_done[x]);
    }
    delete ctx;
    return SKL_OK;
}

extern "C" int skl_ctx_set_stream(skl_ctx *ctx, void *hip_stream)
{
    SKL_TRY(ctx_bind(ctx));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return SKL_OK;
}

extern "C" int skl_ctx_use_default_stream(skl_ctx *ctx)
{
    SKL_TRY(ctx_bind(ctx));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->stream = nullptr;   // the legacy default stream: ordered with every blocking stream
    return SKL_OK;
}

extern "C" int skl_ctx_synchronize(skl_ctx *ctx)
{
    SKL_TRY(ctx_bind(ctx));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SKL_OK;
}

extern "C" int skl_ctx_reload_env(skl_ctx *ctx)
{
    SKL_TRY(ctx_bind(ctx));
    ctx->knobs = read_knobs();
    ctx->timing_every = ctx->knobs.timing_every;
    return SKL_OK;
}

// Pair-kernel timing is a diagnostic and OFF unless asked for: an event record is a barrier packet on the queue, and two
// per launch cost a sub-millisecond launch ~5 us.  every = 0: off (the default); N >= 1: every N-th launch is bracketed.
extern "C" int skl_ctx_early_break_stats(skl_ctx *ctx, uint64_t *pairs, uint64_t *completed_one_by_one)
{
    SKL_TRY(ctx_bind(ctx));
    uint64_t done = 0;
    if (ctx->eb_counter) {
        std::vector<uint32_t> slots(1024, 0u);
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->aux_stream));
        HIP_TRY(hipMemcpy(slots.data(), ctx->eb_counter, slots.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        for (const uint32_t v : slots) done += v;
    }
    if (pairs) *pairs = ctx->eb_pairs;
    if (completed_one_by_one) *completed_one_by_one = done;
    return SKL_OK;
}

extern "C" int skl_ctx_timing_enable(skl_ctx *ctx, int every)
{
    SKL_TRY(ctx_bind(ctx));
    if (every < 0) return fail(SKL_ERR_INVALID_ARG, "skl_ctx_timing_enable: every = %d must be >= 0", every);
    ctx->timing_every = every;
    ctx->launches_seen = 0;
    return SKL_OK;
}

// Shader clock under load.  start(): launches the one-wave sampler (kernels.hip) on a stream of its own;
// the caller then runs whatever it wants measured and, once THAT work is complete (skl_ctx_synchronize:
// a device-wide synchronisation would wait for the sampler itself), calls stop(), which raises the flag,
// waits for the sampler and reduces its (shader counter, 100 MHz counter) pairs to clock readings, one
// per sampling interval.  The sampler also ends by itself after max_samples intervals.
extern "C" int skl_clock_sampler_start(skl_ctx *ctx, uint32_t interval_us, uint32_t max_samples)
{
    SKL_TRY(ctx_bind(ctx));
    if (ctx->sampler_running) return fail(SKL_ERR_INVALID_ARG, "the clock sampler is already running");
    if (max_samples < 2 || max_samples > (1u << 22)) return fail(SKL_ERR_INVALID_ARG, "max_samples out of range");
    // The sampler ends by itself after interval_us x max_samples: anything that synchronises the whole device while it runs
    // (hipMalloc / hipFree inside a sampled call) waits that long at worst, so the product is capped at 10 s.
    if ((uint64_t)std::max(4u, interval_us) * max_samples > 10000000ull) {
        return fail(SKL_ERR_INVALID_ARG, "clock sampler: interval_us x max_samples = %llu us exceeds the 10 s cap",
                    (unsigned long long)std::max(4u, interval_us) * max_samples);
    }
    if (!ctx->sampler_stream) HIP_TRY(hipStreamCreateWithFlags(&ctx->sampler_stream, hipStreamNonBlocking));
    if (!ctx->sampler_stop) HIP_TRY(hipHostMalloc((void **)&ctx->sampler_stop, sizeof(uint32_t), hipHostMallocMapped));
    if (!ctx->sampler_count) HIP_TRY(hipMalloc((void **)&ctx->sampler_count, sizeof(uint32_t)));
    if (ctx->sampler_max < max_samples) {
        if (ctx->sampler_buf) HIP_TRY(hipFree(ctx->sampler_buf));
        ctx->sampler_buf = nullptr;
        ctx->sampler_max = 0;
        HIP_TRY(hipMalloc((void **)&ctx->sampler_buf, (size_t)max_samples * 2 * sizeof(uint64_t)));
        ctx->sampler_max = max_samples;
    }
    *(volatile uint32_t *)ctx->sampler_stop = 0u;
    uint32_t *stop_dev = nullptr;
    HIP_TRY(hipHostGetDevicePointer((void **)&stop_dev, ctx->sampler_stop, 0));
    HIP_TRY(hipMemsetAsync(ctx->sampler_count, 0, sizeof(uint32_t), ctx->sampler_stream));
    const uint32_t sleeps = std::max(1u, interval_us / 4u);   // s_sleep 127 = 8 128 cycles ~ 4 us
    HIP_TRY(launch_clock_sampler(stop_dev, ctx->sampler_buf, max_samples, sleeps, ctx->sampler_count, ctx->sampler_stream));
    ctx->sampler_running = true;
    return SKL_OK;
}

extern "C" int skl_clock_sampler_stop(skl_ctx *ctx, double *ghz_median, double *ghz_p10, double *ghz_p90, double *ghz_mean,
                                      int *n_intervals)
{
    SKL_TRY(ctx_bind(ctx));
    if (!ctx->sampler_running) return fail(SKL_ERR_INVALID_ARG, "the clock sampler is not running");
    *(volatile uint32_t *)ctx->sampler_stop = 1u;
    ctx->sampler_running = false;
    HIP_TRY(hipStreamSynchronize(ctx->sampler_stream));
    uint32_t count = 0;
    HIP_TRY(hipMemcpy(&count, ctx->sampler_count, sizeof count, hipMemcpyDeviceToHost));
    std::vector<uint64_t> s((size_t)count * 2);
    if (count) HIP_TRY(hipMemcpy(s.data(), ctx->sampler_buf, s.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
    std::vector<double> ghz;
    for (uint32_t i = 1; i < count; ++i) {
        const uint64_t dt = s[2 * i] - s[2 * i - 2], dr = s[2 * i + 1] - s[2 * i - 1];
        if (dr > 0) ghz.push_back((double)dt / (double)dr * 0.1);   // shader cycles per 10 ns tick
    }
    std::sort(ghz.begin(), ghz.end());
    const auto at = [&](double q) { return ghz.empty() ? 0.0 : ghz[std::min(ghz.size() - 1, (size_t)(q * (double)ghz.size()))]; };
    if (ghz_median) *ghz_median = at(0.5);
    if (ghz_p10) *ghz_p10 = at(0.1);
    if (ghz_p90) *ghz_p90 = at(0.9);
    if (ghz_mean) *ghz_mean = count >= 2 && s[2 * count - 1] > s[1] ? (double)(s[2 * count - 2] - s[0]) / (double)(s[2 * count - 1] - s[1]) * 0.1 : 0.0;
    if (n_intervals) *n_intervals = (int)ghz.size();
    return SKL_OK;
}

extern "C" const char *skl_ctx_last_kernel(skl_ctx *ctx) { return ctx ? ctx->last_kernel.c_str() : ""; }

extern "C" int skl_ctx_timing_reset(skl_ctx *ctx)
{
    SKL_TRY(ctx_bind(ctx));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->events_used = 0;
    ctx->launches_seen = 0;   // the every-N-th count starts here: the first launch after a reset is bracketed
    return SKL_OK;
}

extern "C" int skl_ctx_kernel_ms(skl_ctx *ctx, float *total_ms, int *n_launches)
{
    SKL_TRY(ctx_bind(ctx));
    if (!total_ms) return fail(SKL_ERR_INVALID_ARG, "total_ms is null");
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    float total = 0.f;
    for (size_t i = 0; i < ctx->events_used; ++i) {
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, ctx->events[i].first, ctx->events[i].second));
        total += t;
    }
    *total_ms = total;
    if (n_launches) *n_launches = (int)ctx->events_used;
    return SKL_OK;
}

static long long env_int(const char *name, long long dflt)
{
    const char *e = getenv(name);
    return (e && *e) ? atoll(e) : dflt;
}

Knobs read_knobs()
{
    Knobs k;
    // The product library reads EIGHT switches: the timing cadence, the device topology the tile order assumes, and the knobs
    // with which the tests force the banded / sliced / 32-row forms on inputs small enough for the oracle.  Everything that
    // exists only to time one form against another ("A/B only, results identical") is read by the A/B build alone (-DSKL_AB),
    // where scripts/ab_sweep.py, scripts/forced_switch_suites.sh and the tests marked `ab_library` find it.
    k.timing_every = std::max(0ll, env_int("SKL_TIMING_EVERY", 0));
    k.sliced_max_pairs = env_int("SKL_SLICED_MAX_PAIRS", -1);
    k.knn_band_rows = std::max(0ll, env_int("SKL_KNN_BAND_ROWS", 0));
    k.tail_slices = (int)std::min(8ll, std::max(0ll, env_int("SKL_TAIL_SLICES", 4)));
    k.tail_max_pct = env_int("SKL_TAIL_MAX_PCT", 90);
    k.tile32_min = env_int("SKL_TILE32_MIN", 8ll << 20);
    k.group_span = (int)std::min(64ll, std::max(1ll, env_int("SKL_GROUP_SPAN", 2)));
    k.xcds = (int)std::min(8ll, std::max(0ll, env_int("SKL_XCDS", 0)));
#ifdef SKL_AB
    k.k_slices = (int)env_int("SKL_K_SLICES", 0);
    k.round_priority = env_int("SKL_ROUND_PRIORITY", 1) != 0;
    k.half_tiles = env_int("SKL_HALF_TILES", 1) != 0;
    k.mid_band = env_int("SKL_MID_BAND", 1) != 0;
    k.knn_symmetric = env_int("SKL_KNN_SYMMETRIC", 1) != 0;
    k.knn_overlap = env_int("SKL_KNN_OVERLAP", 1) != 0;
    k.knn_prune = env_int("SKL_KNN_PRUNE", 1) != 0;
    k.knn_panel = env_int("SKL_KNN_PANEL", 0);
    k.knn_sparse = env_int("SKL_KNN_SPARSE", 1) != 0;
    k.early_break = (int)std::min(7ll, std::max(0ll, env_int("SKL_EARLY_BREAK", 1)));
    k.epilogue_r5 = env_int("SKL_EPILOGUE_R5", 0) != 0;
    k.eb_pipeline = (int)env_int("SKL_EB_PIPELINE", -1);
    k.eb_pipeline_min = std::max(2ll, env_int("SKL_EB_PIPELINE_MIN", 64ll << 20));
    k.counts_u16 = env_int("SKL_COUNTS_U16", 1) != 0;
    k.eb_lds_rows = env_int("SKL_EB_LDS_ROWS", 1) != 0;
    k.eb_ahead = env_int("SKL_EB_AHEAD", 1) != 0;
    k.eb_lean = env_int("SKL_EB_LEAN", 1) != 0;
    k.knn_epi_blocked = (int)std::min(2ll, std::max(0ll, env_int("SKL_KNN_EPI_BLOCKED", 1)));
    k.eb_blocked = (int)env_int("SKL_EB_BLOCKED", -1);
    k.eb_blk_row_shift = (int)env_int("SKL_EB_BLK_ROW_SHIFT", 10);
    k.fuse_epilogue = env_int("SKL_FUSE_EPILOGUE", 0) != 0;
    k.refheap_wave = env_int("SKL_REFHEAP_WAVE", 1) != 0;
    k.knn_row_flags = env_int("SKL_KNN_ROW_FLAGS", 1) != 0;
    k.topk_stream = env_int("SKL_TOPK_STREAM", 1) != 0;
    k.cand_symmetric = env_int("SKL_CAND_SYMMETRIC", 1) != 0;
    k.cand_row_order = env_int("SKL_CAND_ROW_ORDER", 1) != 0;
    {
        const char *ck = getenv("SKL_CAND_KERNEL");
        k.cand_lanes = ck && strcmp(ck, "lanes") == 0;
    }
    k.inline_prefix = env_int("SKL_INLINE_PREFIX", 1) != 0;
    const char *sk = getenv("SKL_SKETCH_KERNEL");
    k.sketch_global = sk && strcmp(sk, "global") == 0;
    // SKL_KERNEL = ksplit | kslice forces one implementation (0: dispatcher's choice)
    if (const char *e = getenv("SKL_KERNEL")) k.kernel = strcmp(e, "ksplit") == 0 ? 3 : strcmp(e, "kslice") == 0 ? 4 : 0;
    k.kslice_shape = (int)env_int("SKL_KSLICE_SHAPE", 0);
    k.ksplit_rows = (int)env_int("SKL_KSPLIT_ROWS", 0);
    k.kslice_ablate = (int)env_int("SKL_KSLICE_ABLATE", 0);
#endif
    return k;
}

#ifdef SKL_AB
int ab_forced_log_variant() { return (int)std::max(-2ll, std::min(1ll, env_int("SKL_FORCE_LOG_VARIANT", -2))); }   // -2: not forced
#endif

// XCDs the device presents as one: an MI355X XCD has 32 CUs, so an unpartitioned (SPX) part shows 256 CUs = 8 XCDs, a CPX
// partition 32 CUs = 1.  The tile order deals workgroups to XCDs by blockIdx mod that number; SKL_XCDS forces it (tests).
uint32_t ctx_xcd_shift(const skl_ctx *ctx)
{
    int x = ctx->knobs.xcds > 0 ? ctx->knobs.xcds : ctx->n_cu / 32;
    uint32_t shift = 0;
    while (shift < 3u && (2 << shift) <= x) ++shift;
    return shift;
}

int forced_kernel(const skl_ctx *ctx)
{
#ifdef SKL_AB
    // a forced tile shape or ablation counts as a forced kernel (no chunk slices, no mid-band rule)
    if (ctx->knobs.kernel == 0 && (ctx->knobs.kslice_shape || ctx->knobs.kslice_ablate)) return 4;
    return ctx->knobs.kernel;
#else
    (void)ctx;
    return 0;
#endif
}

// One tile computation.  Product library: the chunk-split kernel (pair_kslice.hip: 16 x 128 tiles,
// chunks split over the 4 waves, rows by LDS DMA; one workgroup per (tile, k) for small launches
// and for single-k Jaccard, all k + fused regression otherwise) and, for the shapes it does not
// take, pair_ksplit.hip.  The A/B build adds the round-2/3 forms of the two tile shapes behind
// SKL_KSLICE_SHAPE and SKL_KERNEL=ksplit.
static hipError_t dispatch_pair_kernel(skl_ctx *ctx, const PairArgs &args_in, int mode, hipStream_t stream)
{
    PairArgs args = args_in;
    args.group_span = (uint32_t)ctx->knobs.group_span;
    args.xcd_shift = ctx_xcd_shift(ctx);
    args.inline_prefix_ok = ctx->knobs.inline_prefix ? 1u : 0u;
    const uint32_t n_xcd = 1u << args.xcd_shift;
    static const char *mode_names[] = {"COUNTS", "JACCARD", "COREACC"};
    const std::string m = mode_names[mode];
    std::string *name = &ctx->last_kernel;
    TileScratch &tiles = ctx->tile_scratch;
    const uint64_t rows = args.row_end - args.row_begin;
    const uint64_t pairs = args.self_mode ? rows * args.nB / 2 : rows * (uint64_t)args.nB;
    const bool small = pairs < (8ull << 20);
    // 165 = 16 x 128 tiles in the 128-register form (4 waves per SIMD): +3.5 % at n = 16 000 over the
    // 141-register form 162 (3 waves), equal at n = 1 000 (profiles/r02_ab_tight.jsonl)
    // 325 = 32 x 128 tiles (130-168 registers, 3 waves per SIMD) for launches of at least
    // tile32_min pair x k-mer-length evaluations (~4 096 units of 32 x 128): every column register is
    // used against 32 rows instead of 16, which halves the lane-slab traffic per pair -- HBM bytes per
    // launch at n = 16 000 fall from 59.5 GB to 32.0 GB and the kernel gains 1.5-6 %
    // (profiles/r02_tile32_*.md); smaller launches lose to the coarser tail (n = 1 000: +32 %).
    int shape = 165, ksplit_rows = 8;   // 8 >= 4 rows from n = 1000 up once XCDs are balanced
    {
        const uint64_t k_walked = mode == MODE_JACCARD ? 1u : args.k_count;
        if (ctx->knobs.tile32_min >= 0 && pairs * k_walked >= (uint64_t)ctx->knobs.tile32_min) shape = 325;
        if (args.mid_band) shape = 325;   // dense_band's mid-band rule: 32-row tiles with the last round cut in 2
    }
#ifdef SKL_AB
    const Knobs &kn = ctx->knobs;
    if (kn.kslice_shape) shape = kn.kslice_shape;
    if (kn.ksplit_rows) ksplit_rows = kn.ksplit_rows;
    const bool try_kslice = kn.kernel != 3;
    const int ablate = kn.kslice_ablate;
#else
    const bool try_kslice = true;
    const int ablate = 0;
#endif
    args.no_half_tiles = ctx->knobs.half_tiles ? 0u : 1u;
    // workgroups resident per CU: 4 for every shipped form (the A/B build's 3-wave all-k 32-row form: 3)
    // (sketches beyond 65 535 bins: the k-sliced forms only -- they walk a k-mer length in segments, pair_kslice_walk.inc)
    const bool big_sketch = args.ss64 > (uint32_t)KSLICE_MAX_U16_CHUNKS;
    const bool sliced_launch = mode == MODE_JACCARD || (mode == MODE_COUNTS && (small || args.k_sliced || big_sketch));
    const uint32_t wg_per_cu = ((shape == 3255 && !sliced_launch) || (shape == 3254 && sliced_launch)) ? 3u : 4u;
    args.round_size = ctx->knobs.round_priority ? wg_per_cu * (uint32_t)ctx->n_cu / n_xcd : 0u;
    ctx->last_count_planes = std::max(1u, args.k_slices);
    ctx->last_tail = false;
    if (args.tail_slices > 1u) {
        // tail-sliced one-workgroup-per-unit launch: two planes whatever kernel ends up running (a
        // kernel without the slices leaves plane 1 as it found it: zero)
        args.tail_resident = wg_per_cu * (uint32_t)ctx->n_cu / n_xcd;
        ctx->last_count_planes = 2;
        ctx->last_tail = true;
    }
    if (try_kslice) {
        // single-k Jaccard: the sliced and the all-k form are the same work, the sliced one
        // compiles to fewer registers; core/acc arrives here as MODE_COUNTS from dense_band when sliced
        const bool sliced = sliced_launch;
        if (kslice_supported(args, mode, sliced)) {
            const int jl = (shape == 165 || shape == 325 || shape > 1000) ? 2 : shape % 10;
            const int rr = shape > 1000 ? shape / 100 : shape / 10;
            *name = "skl::pair_kernel_kslice<R=" + std::to_string(rr) + ", JL=" + std::to_string(jl) +
                    ", " + m + (sliced ? ", k-sliced" : ", all k") + ((shape == 165 || shape == 325 || shape > 1000) ? ", tight" : "") + "> (" +
                    std::to_string(rr) + "x" + std::to_string(jl * 64) + " tiles, chunks split over 4 waves" +
                    (big_sketch ? "; segments of " + std::to_string(KSLICE_SEG_CHUNKS) + " chunks" : "") +
                    (sliced && mode == MODE_COUNTS && args.tail_slices > 1u
                         ? "; " + std::to_string(args.tail_slices) + " chunk slices per unit in the last round of workgroups" : "") + ")";
            return launch_pair_kernel_kslice(args, mode, shape, sliced, ablate, tiles, stream);
        }
    }
    *name = "skl::pair_kernel_ksplit<R=" + std::to_string(ksplit_rows) + ", " + m + "> (" + std::to_string(ksplit_rows) +
            "x64 tiles, chunks split over 4 waves)";
    return launch_pair_kernel_ksplit(args, mode, ksplit_rows, tiles, stream);
}

// Launch the pair kernel bracketed by HIP events on the context's stream.
int timed_pair_launch(skl_ctx *ctx, const PairArgs &args, int mode)
{
    constexpr size_t MAX_EVENTS = 4096;
    // skl_ctx_timing_enable(N) / SKL_TIMING_EVERY = N brackets every N-th launch (default 0 = none: an event
    // record is a barrier packet on the queue, and two per launch cost a sub-millisecond launch ~5 us)
    const bool sampled = ctx->timing_every > 0 && (ctx->launches_seen++ % (size_t)ctx->timing_every) == 0;
    if (!sampled || ctx->events_used >= MAX_EVENTS) {
        HIP_TRY(dispatch_pair_kernel(ctx, args, mode, ctx->stream));
        return SKL_OK;
    }
    if (ctx->events_used == ctx->events.size()) {
        hipEvent_t a, b;
        HIP_TRY(hipEventCreate(&a));
        const hipError_t eb = hipEventCreate(&b);
        if (eb != hipSuccess) {
            (void)hipEventDestroy(a);
            return fail(SKL_ERR_HIP, "hipEventCreate: %s", hipGetErrorString(eb));
        }
        ctx->events.emplace_back(a, b);
    }
    auto &ev = ctx->events[ctx->events_used++];
    HIP_TRY(hipEventRecord(ev.first, ctx->stream));
    HIP_TRY(dispatch_pair_kernel(ctx, args, mode, ctx->stream));
    HIP_TRY(hipEventRecord(ev.second, ctx->stream));
    return SKL_OK;
}

std::pair<hipEvent_t, hipEvent_t> *timing_slot(skl_ctx *ctx)
{
    if (ctx->timing_every <= 0) return nullptr;
    if (ctx->events_used == ctx->events.size() && ctx->events.size() < 4096) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess) return nullptr;
        if (hipEventCreate(&b) != hipSuccess) {
            (void)hipEventDestroy(a);
            return nullptr;
        }
        ctx->events.emplace_back(a, b);
    }
    return ctx->events_used < ctx->events.size() ? &ctx->events[ctx->events_used++] : nullptr;
}

// ---------------------------------------------------------------------------
// the host libm's logarithm on the device (completeness path; glibc_log.hpp)
// ---------------------------------------------------------------------------

static double probe_uniform(uint64_t &state)
{
    state = state * 6364136223846793005ull + 1442695040888963407ull;
    return (double)(state >> 11) * 0x1p-53;
}

int host_log_variant()
{
    static const int variant = [] {
#ifdef SKL_AB
        // A/B build only: SKL_FORCE_LOG_VARIANT=-1 | 0 | 1 takes the place of the probe (tests of the "neither form" branch)
        const int forced = ab_forced_log_variant();
        if (forced >= -1) return forced;
#endif
        // arguments of the kind the path takes logarithms of: Jaccard values in (0, 1], the
        // |x - 1| < 1/16 branch, and the values every k-mer length of a sketch size can give
        uint64_t st = 0x5EED0001ull;
        size_t bad[2] = {0, 0};
        auto check = [&](double x) {
            volatile double xv = x;   // a run-time call into libm, never folded
            const double ref = std::log(xv);
            for (int v = 0; v < 2; ++v) {
                if (skl_as_u64(glibc_log(x, v)) != skl_as_u64(ref)) ++bad[v];
            }
        };
        for (int i = 0; i < 150000; ++i) check(probe_uniform(st));
        for (int i = 0; i < 150000; ++i) check(0.9375 + 0.13 * probe_uniform(st));
        for (uint32_t b = 1; b <= 4096; ++b) check((double)b / 4096.0);
        if (bad[SKL_LOG_FMA] == 0) return (int)SKL_LOG_FMA;
        if (bad[SKL_LOG_SSE2] == 0) return (int)SKL_LOG_SSE2;
        return -1;   // reported through skl_ctx_flags() / skl_log_variant(); the caller decides what to print
    }();
    return variant;
}

extern "C" int skl_log_variant(void) { return host_log_variant(); }

extern "C" unsigned skl_ctx_flags(const skl_ctx *ctx)
{
    unsigned flags = 0;
    if (host_log_variant() < 0) flags |= SKL_CTX_FLAG_LOG_UNMATCHED;
    if (ctx && ctx->n_cu != 256) flags |= SKL_CTX_FLAG_NOT_SPX;
    return flags;
}

extern "C" int skl_device_log(skl_ctx *ctx, const double *x_host, size_t n, double *out_host)
{
    SKL_TRY(ctx_bind(ctx));
    if (n == 0) return SKL_OK;
    if (!x_host || !out_host) return fail(SKL_ERR_INVALID_ARG, "null argument");
    DevBuf dx, dy;
    HIP_TRY(hipMalloc(&dx.p, n * sizeof(double)));
    HIP_TRY(hipMalloc(&dy.p, n * sizeof(double)));
    HIP_TRY(hipMemcpyAsync(dx.p, x_host, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    const int v = host_log_variant();
    HIP_TRY(launch_device_log((const double *)dx.p, (double *)dy.p, n, v < 0 ? (int)SKL_LOG_FMA : v, ctx->stream));
    HIP_TRY(hipMemcpyAsync(out_host, dy.p, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SKL_OK;
}

// ---------------------------------------------------------------------------
// sketch slabs
// ---------------------------------------------------------------------------


// jaccard.rs:14,26-33 with no completeness: J as a function of samebits alone.
static double host_jaccard(uint32_t samebits, size_t ss64)
{
    const double unionsize = (double)(64u * ss64);
    const uint32_t maxnbits = (uint32_t)ss64 * 64u;
    const uint32_t expected = maxnbits >> BBITS;
    const uint32_t diff = samebits > expected ? samebits - expected : 0u;
    const double intersize = ((double)diff * (double)maxnbits) / (double)(maxnbits - expected);
    return intersize / unionsize;
}

static int ensure_lanes(const skl_sketches *cs)
{
    skl_sketches *s = const_cast<skl_sketches *>(cs);
    if (s->d_lanes || s->n == 0) return SKL_OK;
    const size_t n_jb = (s->n + 63) / 64;
    const size_t bytes = n_jb * s->nk * s->ss64 * 7 * 64 * sizeof(uint4);
    HIP_TRY(hipMalloc((void **)&s->d_lanes, bytes));
    HIP_TRY(launch_relayout(s->d_rows, s->d_lanes, (uint32_t)s->n, (uint32_t)s->nk,
                            (uint32_t)s->ss64, s->ctx->stream));
    return SKL_OK;
}

static int ensure_ytab(const skl_sketches *cs)
{
    skl_sketches *s = const_cast<skl_sketches *>(cs);
    if (s->d_ytab) return SKL_OK;
    const size_t m = 64 * s->ss64 + 1;
    std::vector<double> tab(m);
    for (size_t b = 0; b < m; ++b) tab[b] = std::log(host_jaccard((uint32_t)b, s->ss64));
    {   // the break test of jaccard.rs:89-91 on the count itself: ln J(count) < tolerance <=> count < min_alive -- if the table is
        // monotone there (it is; checked, not assumed: 0xFFFFFFFF makes the kernels ask the table)
        const double tolerance = std::log(2.0 / (double)((s->ss64 * 64ull) * 64ull));
        size_t first = 0;
        while (first < m && tab[first] < tolerance) ++first;
        bool monotone = true;
        for (size_t b = first; b < m; ++b) monotone = monotone && !(tab[b] < tolerance);
        s->min_alive = monotone ? (uint32_t)first : 0xFFFFFFFFu;
    }
    double *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, m * sizeof(double)));
    const hipError_t e = hipMemcpy(d, tab.data(), m * sizeof(double), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(d);
        return fail(SKL_ERR_HIP, "ln J table upload: %s", hipGetErrorString(e));
    }
    s->d_ytab = d;
    return SKL_OK;
}

static int ensure_dtab(const skl_sketches *cs, int jout, size_t k_idx, float **out)
{
    skl_sketches *s = const_cast<skl_sketches *>(cs);
    const size_t key_k = jout == JOUT_DIST ? 0 : k_idx;
    auto it = s->d_dtab.find({jout, key_k});
    if (it != s->d_dtab.end()) {
        *out = it->second;
        return SKL_OK;
    }
    const size_t m = 64 * s->ss64 + 1;
    const double k = (double)s->kmers[k_idx];
    std::vector<float> tab(m);
    for (size_t b = 0; b < m; ++b) {
        const double j = host_jaccard((uint32_t)b, s->ss64);
        if (jout == JOUT_DIST) {
            tab[b] = (float)(1.0 - j);  // mod.rs:99
        } else {
            // jaccard.rs:49-51
            const double ani = std::fmax(0.0, 1.0 + 1.0 / k * std::log((2.0 * j) / (1.0 + j)));
            tab[b] = jout == JOUT_ANI ? (float)ani : (float)(1.0 - ani);  // mod.rs:97 / :173-176
        }
    }
    float *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, m * sizeof(float)));
    const hipError_t e = hipMemcpy(d, tab.data(), m * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(d);
        return fail(SKL_ERR_HIP, "distance table upload: %s", hipGetErrorString(e));
    }
    s->d_dtab[{jout, key_k}] = d;
    *out = d;
    return SKL_OK;
}

extern "C" int skl_sketches_create(skl_ctx *ctx, const uint64_t *bins, int on_device,
                                   size_t n_samples, size_t nk, const size_t *kmers,
                                   size_t sketchsize64, skl_sketches **out)
{
    SKL_TRY(ctx_bind(ctx));
    if (!out) return fail(SKL_ERR_INVALID_ARG, "out is null");
    *out = nullptr;
    if (nk == 0 || sketchsize64 == 0 || !kmers) {
        return fail(SKL_ERR_INVALID_ARG, "need at least one k-mer length and a non-zero sketch size");
    }
    if (n_samples && !bins) return fail(SKL_ERR_INVALID_ARG, "bins is null");
    if (n_samples >= (1ull << 31) || sketchsize64 >= (1ull << 25) || nk >= (1ull << 16)) {
        return fail(SKL_ERR_INVALID_ARG, "dimensions out of range");
    }
    skl_sketches *s = new skl_sketches();
    s->ctx = ctx;
    s->gen = next_generation();
    {
        std::lock_guard<std::mutex> lock(g_registry_mutex);
        g_live_sketches.insert(s);
        ctx->sketches.insert(s);
    }
    s->n = n_samples;
    s->nk = nk;
    s->ss64 = sketchsize64;
    s->kmers.assign(kmers, kmers + nk);
    const size_t words = s->sample_words();
    const size_t total = (n_samples + A_PAD_ROWS) * words;
    hipError_t e = hipMalloc((void **)&s->d_rows, total * sizeof(uint64_t));
    if (e != hipSuccess) {
        skl_sketches_destroy(s);
        return fail(e == hipErrorOutOfMemory ? SKL_ERR_OOM : SKL_ERR_HIP, "hipMalloc(slab %zu B): %s",
                    total * sizeof(uint64_t), hipGetErrorString(e));
    }
    int rc = SKL_OK;
    do {
        e = hipMemsetAsync(s->d_rows + n_samples * words, 0, A_PAD_ROWS * words * sizeof(uint64_t),
                           ctx->stream);
        if (e != hipSuccess) break;
        if (n_samples) {
            e = hipMemcpyAsync(s->d_rows, bins, n_samples * words * sizeof(uint64_t),
                               on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                               ctx->stream);
            if (e != hipSuccess) break;
        }
        std::vector<double> kf(nk);
        for (size_t i = 0; i < nk; ++i) kf[i] = (double)kmers[i];
        e = hipMalloc((void **)&s->d_kf, nk * sizeof(double));
        if (e != hipSuccess) break;
        e = hipMemcpy(s->d_kf, kf.data(), nk * sizeof(double), hipMemcpyHostToDevice);
        if (e != hipSuccess) break;
        e = hipStreamSynchronize(ctx->stream);
    } while (0);
    if (e != hipSuccess) {
        rc = fail(SKL_ERR_HIP, "slab upload: %s", hipGetErrorString(e));
        skl_sketches_destroy(s);
        return rc;
    }
    *out = s;
    return SKL_OK;
}

extern "C" int skl_sketches_set_completeness(skl_sketches *s, const double *comp)
{
    if (!s) return fail(SKL_ERR_INVALID_ARG, "null sketches");
    SKL_TRY(ctx_bind(s->ctx));
    s->gen = next_generation();   // (a sampled early-break decision belongs to the slab AND its completeness vector)
    if (!comp) {
        if (s->d_comp) {
            HIP_TRY(hipStreamSynchronize(s->ctx->stream));
            HIP_TRY(hipFree(s->d_comp));
            s->d_comp = nullptr;
        }
        return SKL_OK;
    }
    if (!s->d_comp) {
        HIP_TRY(hipMalloc((void **)&s->d_comp, (s->n + A_PAD_ROWS + 64) * sizeof(double)));
        HIP_TRY(hipMemset(s->d_comp, 0, (s->n + A_PAD_ROWS + 64) * sizeof(double)));
    }
    HIP_TRY(hipStreamSynchronize(s->ctx->stream));
    HIP_TRY(hipMemcpy(s->d_comp, comp, s->n * sizeof(double), hipMemcpyHostToDevice));
    s->comp_unit = true;
    for (size_t x = 0; x < s->n; ++x) {
        if (!(comp[x] > 0.0 && comp[x] <= 1.0)) {   // (NaN fails both)
            s->comp_unit = false;
            break;
        }
    }
    return SKL_OK;
}

static void free_sketches_locked(skl_sketches *s)
{
    if (!g_live_sketches.erase(s)) return;
    s->ctx->sketches.erase(s);
    (void)hipSetDevice(s->ctx->device);
    (void)hipStreamSynchronize(s->ctx->stream);
    if (s->d_rows) (void)hipFree(s->d_rows);
    if (s->d_lanes) (void)hipFree(s->d_lanes);
    if (s->d_comp) (void)hipFree(s->d_comp);
    if (s->d_ytab) (void)hipFree(s->d_ytab);
    if (s->d_kf) (void)hipFree(s->d_kf);
    for (auto &kv : s->d_dtab) (void)hipFree(kv.second);
    delete s;
}

extern "C" int skl_sketches_destroy(skl_sketches *s)
{
    if (!s) return SKL_OK;
    std::lock_guard<std::mutex> lock(g_registry_mutex);
    free_sketches_locked(s);  // no-op if the handle (or its context) is already gone
    return SKL_OK;
}

extern "C" size_t skl_sketches_n_samples(const skl_sketches *s) { return s ? s->n : 0; }

// mod.rs:25-37
extern "C" int skl_set_k(const skl_sketches *s, size_t kmer, int ani, double cutoff,
                         skl_dist_params *out)
{
    if (!s || !out) return fail(SKL_ERR_INVALID_ARG, "null argument");
    out->completeness_cutoff = cutoff;
    out->k_idx = 0;
    out->ani = 0;
    if (kmer == 0) {
        out->dist_type = SKL_DIST_COREACC;
        return SKL_OK;
    }
    for (size_t i = 0; i < s->nk; ++i) {
        if (s->kmers[i] == kmer) {
            out->dist_type = SKL_DIST_JACCARD;
            out->k_idx = i;
            out->ani = ani ? 1 : 0;
            return SKL_OK;
        }
    }
    return fail(SKL_ERR_KMER_NOT_FOUND, "K-mer size %zu not found in file", kmer);
}

// ---------------------------------------------------------------------------
// shared launch preparation
// ---------------------------------------------------------------------------

int check_params(const skl_sketches *a, const skl_sketches *b, const skl_dist_params *p)
{
    if (!a || !b || !p) return fail(SKL_ERR_INVALID_ARG, "null argument");
    if (a->ctx != b->ctx) return fail(SKL_ERR_INVALID_ARG, "sketches belong to different contexts");
    if (a->nk != b->nk || a->ss64 != b->ss64 || a->kmers != b->kmers) {
        // MultiSketch::is_compatible_with, multisketch.rs:222-226
        return fail(SKL_ERR_INCOMPATIBLE, "reference and query sketches are not compatible (k-mer lengths / sketch size differ)");
    }
    if (p->dist_type == SKL_DIST_COREACC) {
        if (a->nk < 2) {
            return fail(SKL_ERR_KMER_COUNT,
                        "Need at least two k-mer lengths to calculate core/accessory distances");
        }
    } else if (p->dist_type == SKL_DIST_JACCARD) {
        if (p->k_idx >= a->nk) return fail(SKL_ERR_INVALID_ARG, "k_idx %llu out of range", (unsigned long long)p->k_idx);
    } else {
        return fail(SKL_ERR_INVALID_ARG, "unknown dist_type %d", p->dist_type);
    }
    return SKL_OK;
}

bool fused_coreacc_ok(const skl_sketches *s)
{
    return s->nk <= (size_t)MAX_FUSED_K && 64 * s->ss64 <= 0xFFFFu;
}

// Fill the operand / epilogue fields common to every launch.  `rows` is the scalar
// operand (A), `cols` the lane operand (B).
int fill_args(const skl_sketches *rows, const skl_sketches *cols, const skl_dist_params *p,
                     int mode, int jout, PairArgs *g)
{
    memset(g, 0, sizeof *g);
    SKL_TRY(ensure_lanes(cols));
    g->A = rows->d_rows;
    g->B = cols->d_lanes;
    g->nA = (uint32_t)rows->n;
    g->nB = (uint32_t)cols->n;
    g->nk = (uint32_t)rows->nk;
    g->ss64 = (uint32_t)rows->ss64;
    g->has_comp = (rows->d_comp && cols->d_comp) ? 1 : 0;  // both Some, jaccard.rs:36
    if (g->has_comp) {
        const int v = host_log_variant();
        g->log_variant = v < 0 ? (int)SKL_LOG_FMA : v;
    }
    g->compA = rows->d_comp;
    g->compB = cols->d_comp;
    g->cutoff = p ? p->completeness_cutoff : 0.0;
    g->tolerance = std::log(2.0 / (double)((rows->ss64 * 64ull) * 64ull));  // jaccard.rs:75
    g->jout = jout;
    if (mode == MODE_COUNTS) {
        g->k_begin = 0;
        g->k_count = (uint32_t)rows->nk;
        g->cnt_pair_stride = rows->nk;   // [pair][k] records
        g->cnt_k_stride = 1;
    } else if (mode == MODE_JACCARD) {
        g->k_begin = (uint32_t)p->k_idx;
        g->k_count = 1;
        g->kf[0] = (double)rows->kmers[p->k_idx];
        if (!g->has_comp) {
            float *d = nullptr;
            SKL_TRY(ensure_dtab(rows, jout, p->k_idx, &d));
            g->dtab = d;
        }
    } else {
        g->k_begin = 0;
        g->k_count = (uint32_t)rows->nk;
        for (size_t i = 0; i < rows->nk && i < (size_t)MAX_FUSED_K; ++i) g->kf[i] = (double)rows->kmers[i];
        SKL_TRY(ensure_ytab(rows));
        g->ytab = rows->d_ytab;
    }
    return SKL_OK;
}

// Launch-size rule shared with dispatch_pair_kernel: core/acc launches below this many pairs
// run k-sliced (counts + epilogue kernel), larger ones as one fused kernel.
constexpr long long SLICED_MAX_PAIRS = 32ll << 20;   // n ~ 8000 all-vs-all: equal there (scripts/ab_sweep.py)
constexpr size_t COUNTS_SCRATCH_MAX = 4ull << 30;    // bytes of bin-match counts one unfused core/accessory launch may park in HBM
static bool coreacc_runs_sliced(const skl_ctx *ctx, const skl_sketches *s, uint64_t pairs)
{
    const int forced = forced_kernel(ctx);
    if (forced != 0 && forced != 4) return false;   // another kernel forced: never slice
    if (s->ss64 > (size_t)KSLICE_MAX_U16_CHUNKS) return true;   // beyond 65 535 bins the fused form's u16 fields do not hold a count: always counts + epilogue
    const long long limit = ctx->knobs.sliced_max_pairs >= 0 ? ctx->knobs.sliced_max_pairs : SLICED_MAX_PAIRS;
    return pairs < (uint64_t)limit;
}

static uint64_t cond_index(uint64_t i, uint64_t j, uint64_t n)
{
    return n * i - ((i * (i + 1)) >> 1) + j - 1 - i;  // distance_matrix.rs:11-14
}

// Number of pairs in rows [r0, r1) of the condensed triangle of n samples.
static uint64_t self_rows_pairs(uint64_t r0, uint64_t r1, uint64_t n)
{
    if (n < 2) return 0;
    r1 = std::min<uint64_t>(r1, n - 1);
    if (r1 <= r0) return 0;
    auto upto = [n](uint64_t r) { return r * n - r * (r + 1) / 2; };  // pairs with i < r
    return upto(r1) - upto(r0);
}


// k-sliced core/accessory launches: into how many chunk slices to cut each k-mer length (the kernel
// then runs one workgroup per (tile, k, slice) and the epilogue sums the partial counts).  Default: 1.
// Measured on MI355X (profiles/r02_k_slices_experiment.txt): at BASELINE's 1 000 genomes -- 1 400
// whole-k workgroups on 1 024 resident slots, 1.37 rounds that cost 2 -- 2 slices keep every SIMD at 4
// waves for 100 of 163 us instead of 60 of 157 us, but each workgroup pays its fixed 7 + 2 us (first
// row DMA under load, reduction and stores) on half the work, and the launch ends at the same time
// (0.1601 vs 0.1600 ms per step); 4 slices are 3 % slower, and from n = 1 400 up slices only cost.
// SKL_K_SLICES forces a value (tests keep the sliced form bit-exact; A/B runs).
static uint32_t choose_k_slices(const skl_ctx *ctx, size_t ss64)
{
    const uint32_t S = ctx->knobs.k_slices > 0 ? std::min(8u, (uint32_t)ctx->knobs.k_slices) : 1u;
    uint32_t chunks = 0;
    return slice_plan((uint32_t)ss64, S, &chunks);   // (slices that hold something: a short sketch gets fewer)
}

// EARLY BREAK.  core_acc_dist leaves its loop over the k-mer lengths at the first one whose ln J lies below the tolerance
// (jaccard.rs:89-91: J = 0, i.e. no more shared bins than chance -- expected_samebits, :26-31) and a fit over fewer than three
// lengths is (1, 1) (:117): a pair that fails the test at one of its first lengths is decided by them alone, and between
// unrelated genomes that is nearly every pair (a chance match at each of three lengths: 1.1 % of pairs at 4 096 bins,
// 0.2-0.4 % at 2 048).  The counts + epilogue form can then count only the first ke lengths and let the epilogue complete the
// pairs still in the running (epilogue.hip).  Whether that pays depends on the data -- completing a pair costs EB_COST x what
// the tile kernel spends on a pair and length (a whole column slice read for ONE pair), and between close relatives every
// pair stays in the running -- so the first dense call of a slab against a column slab SAMPLES the pair space: it is cut
// into blocks of (row >> shift, column >> shift) sample ids (up to 64 x 64 of them, each a multiple of 256 samples), a few
// dozen pairs of every block run the reference's loop, and every block takes the ke of {2, 3, 4} that minimises
//     ke + EB_COST x share_alive(ke)      if that is at most 0.9 x nk,
// else every length.  A database that is half one species therefore takes the early break between the species and skips
// it within (round 5 decided once per slab pair).  Blocks of one mind give a plain launch; otherwise the pair kernel's
// (tile, k index) workgroups look their block up and leave when k index >= its ke.
// EB_COST, measured (profiles/r06_early_break_forced_lengths.md: whole calls with 2 / 3 lengths forced, T(3) - T(2) = one length's
// kernel time - EB_COST x the difference of the alive shares): 15-22.  Beyond 65 535 bins a completion is a run of thousands of
// dependent trips of one wave and comes to ~60: there the early break is taken only where hardly a pair stays in the running.
// Pair spaces large enough for the blocked epilogue order (below) complete a pair for ~12-15: cfg 3 with 2 / 3 lengths 642 / 733
// ms, n = 16 000 17.2 / 19.3.
constexpr double EB_COST = 20.0, EB_COST_BLOCKED = 12.0, EB_COST_BIG = 60.0;
static double eb_cost_of(const skl_sketches *rows, const skl_sketches *cols, int self_mode)
{
    if (rows->ss64 > (size_t)KSLICE_MAX_U16_CHUNKS) return EB_COST_BIG;
    const uint64_t pairs = self_mode ? (uint64_t)rows->n * (rows->n - 1) / 2 : (uint64_t)rows->n * cols->n;
    return pairs >= (48ull << 20) ? EB_COST_BLOCKED : EB_COST;
}
// BLOCKED EPILOGUE ORDER (epilogue.hip): the early break's epilogue walks the pairs in blocks of 1 024 rows x 256 columns, each
// block on one XCD, whose L2 then holds the block's 256 column slices while its rows pass -- instead of the launch's flat order
// (a row after the other, all its columns), in which a slice's next reader comes a whole row later and every completion is a
// gather from the Infinity Cache or, once the slices of one length outgrow it (cfg 3: 717 MB), from HBM.  Pays where many
// pairs stay in the running and the launch is large: at 4.9 % alive n = 12 000 / 16 000 / 24 000 / 60 000 / 100 000:
// 10.3 -> 9.9, 18.3 -> 17.2, 41.0 -> 38.0, 266 -> 234, 827 -> 642 ms; at 1.4 % alive (2 048 bins) +1 %: not taken.
bool eb_blocked_order(const skl_ctx *ctx, const skl_sketches *rows, const EbPlan *plan, uint64_t pairs)
{
    if (ctx->knobs.eb_blocked >= 0) return ctx->knobs.eb_blocked != 0;   // (A/B build: forced)
    return plan != nullptr && plan->alive_share >= 0.03 && pairs >= (48ull << 20) && rows->ss64 <= (size_t)KSLICE_MAX_U16_CHUNKS;
}

constexpr uint32_t EB_BLOCKS_MAX = 64;      // blocks per side
constexpr uint32_t EB_SAMPLES_MIN = 128;    // sampled pairs per block
constexpr uint32_t EB_SAMPLES_TOTAL = 4096; // ... and at least this many in all

static std::atomic<uint64_t> g_next_gen{1};
uint64_t next_generation() { return g_next_gen.fetch_add(1); }

static void free_plan(EbPlan *p)
{
    if (!p) return;
    if (p->d_block_ke) (void)hipFree(p->d_block_ke);
    delete p;
}

void free_plans(skl_ctx *ctx)
{
    for (EbPlan *p : ctx->eb_plans) free_plan(p);
    ctx->eb_plans.clear();
    ctx->eb_last_plan = nullptr;
}

// ke of {2, 3, 4} with the lowest modelled cost for a histogram of `total` sampled pairs (hist[m]: pairs that pass the test at
// exactly their first m lengths), or 0 when counting every length is cheaper.  `prior` (9 shares, or null) and its weight:
// the block's estimate is pulled towards the pooled sample of the blocks that take the early break -- 128 pairs a block cannot
// tell a 5 % share from a 9 % one, ten thousand can -- so that only a block that really differs decides differently.
static int best_lengths(const uint32_t *hist, uint32_t total, size_t nk, double eb_cost, double *share_out, const double *prior = nullptr,
                        double weight = 0.0, int preferred = 0)
{
    int best_ke = 0;
    double best = 0.9 * (double)nk, cost_of[5] = {0, 0, 0, 0, 0};
    if (total == 0) return 0;
    // (two lengths decide nothing by themselves -- a fit needs three -- but a pair that fails the test at one of them is
    // decided all the same: (1, 1); ke = 2 leaves more pairs to complete and pays where few share a bin at all: 2 048 bins)
    for (int ke = 2; ke <= 4 && ke < (int)nk; ++ke) {
        double still = 0.0, prior_still = 0.0;
        for (int m = ke; m <= 8; ++m) {
            still += hist[m];
            if (prior) prior_still += prior[m];
        }
        const double share = (still + weight * prior_still) / ((double)total + weight), cost = (double)ke + eb_cost * share;
        cost_of[ke] = cost;
        if (cost <= best) {
            best = cost;
            best_ke = ke;
            if (share_out) *share_out = share;
        }
    }
    // (the blocks' common choice stands unless this block's own is clearly better: the costs of 2 and 3 lengths are often a
    // quarter of a length apart, and a plan whose blocks disagree pays for its table)
    if (preferred >= 2 && preferred <= 4 && preferred < (int)nk && best_ke != preferred && cost_of[preferred] <= 0.9 * (double)nk &&
        cost_of[preferred] <= best + 0.5) {
        best_ke = preferred;
    }
    return best_ke;
}

int early_break_plan(skl_ctx *ctx, const skl_sketches *rows, const skl_sketches *cols, int self_mode, double cutoff, const EbPlan **out)
{
    *out = nullptr;
    const int knob = ctx->knobs.early_break;
    if (knob == 0 || rows->nk < 3 || rows->nk > 8) return SKL_OK;
    if (rows->n * cols->n < 65536) return SKL_OK;
    if (!ctx->eb_counter) {
        HIP_TRY(hipMalloc((void **)&ctx->eb_counter, 1024 * sizeof(uint32_t)));   // 1 024 counter slots (diagnostic)
        HIP_TRY(hipMemsetAsync(ctx->eb_counter, 0, 1024 * sizeof(uint32_t), ctx->stream));
    }
    const bool has_comp = rows->d_comp != nullptr && cols->d_comp != nullptr;
    for (const EbPlan *p : ctx->eb_plans) {
        if (p->rows_gen == rows->gen && p->cols_gen == cols->gen && p->self_mode == self_mode && p->knob == knob && (!has_comp || p->cutoff == cutoff)) {
            *out = p;
            return SKL_OK;
        }
    }
    EbPlan *plan = new EbPlan();
    plan->rows_gen = rows->gen;
    plan->cols_gen = cols->gen;
    plan->self_mode = self_mode;
    plan->cutoff = cutoff;
    plan->knob = knob;
    auto keep = [&]() {
        if (ctx->eb_plans.size() >= 8) {
            // (nothing in flight may still read the oldest plan's table: the streams are drained before it goes)
            if (ctx->eb_plans.front()->d_block_ke != nullptr) {
                (void)hipStreamSynchronize(ctx->stream);
                if (ctx->epi_stream) (void)hipStreamSynchronize(ctx->epi_stream);
            }
            if (ctx->eb_last_plan == ctx->eb_plans.front()) ctx->eb_last_plan = nullptr;
            free_plan(ctx->eb_plans.front());
            ctx->eb_plans.erase(ctx->eb_plans.begin());
        }
        ctx->eb_plans.push_back(plan);
        *out = plan;
    };
    if (knob >= 2) {   // forced (A/B build, tests)
        plan->lengths = knob < (int)rows->nk ? knob : 0;
        keep();
        return SKL_OK;
    }
    // blocks: a power of two of samples per side, at least 256, at most EB_BLOCKS_MAX per side
    auto shift_for = [](size_t n) {
        uint32_t sh = 8;
        while (((n + ((size_t)1 << sh) - 1) >> sh) > EB_BLOCKS_MAX) ++sh;
        return sh;
    };
    plan->shift_r = shift_for(rows->n);
    plan->shift_c = self_mode ? plan->shift_r : shift_for(cols->n);
    plan->blk_rows = (uint32_t)((rows->n + ((size_t)1 << plan->shift_r) - 1) >> plan->shift_r);
    plan->blk_cols = (uint32_t)((cols->n + ((size_t)1 << plan->shift_c) - 1) >> plan->shift_c);
    const uint32_t n_blocks = plan->blk_rows * plan->blk_cols;
    const uint32_t live_blocks = self_mode ? plan->blk_rows * (plan->blk_rows + 1) / 2 : n_blocks;
    const uint32_t samples = std::max(EB_SAMPLES_MIN, (EB_SAMPLES_TOTAL + live_blocks - 1) / live_blocks);
    SKL_TRY(ensure_ytab(rows));
    DevBuf d_hist;
    HIP_TRY(hipMalloc(&d_hist.p, (size_t)n_blocks * 9 * sizeof(uint32_t)));
    HIP_TRY(hipMemsetAsync(d_hist.p, 0, (size_t)n_blocks * 9 * sizeof(uint32_t), ctx->stream));
    EbSampleArgs sa;
    memset(&sa, 0, sizeof sa);
    sa.rows_ref = rows->d_rows;
    sa.cols_ref = cols->d_rows;
    sa.n_rows = (uint32_t)rows->n;
    sa.n_cols = (uint32_t)cols->n;
    sa.nk = (uint32_t)rows->nk;
    sa.ss64 = (uint32_t)rows->ss64;
    sa.self_mode = (uint32_t)self_mode;
    sa.samples = samples;
    sa.blk_shift_r = plan->shift_r;
    sa.blk_shift_c = plan->shift_c;
    sa.blk_rows = plan->blk_rows;
    sa.blk_cols = plan->blk_cols;
    sa.min_alive = rows->min_alive;
    sa.has_comp = has_comp ? 1 : 0;
    if (has_comp) {
        const int v = host_log_variant();
        sa.log_variant = v < 0 ? (int)SKL_LOG_FMA : v;
    }
    sa.ytab = rows->d_ytab;
    sa.compA = rows->d_comp;
    sa.compB = cols->d_comp;
    sa.cutoff = cutoff;
    sa.tolerance = std::log(2.0 / (double)((rows->ss64 * 64ull) * 64ull));  // jaccard.rs:75
    sa.hist = (uint32_t *)d_hist.p;
    HIP_TRY(launch_early_break_sample(sa, ctx->stream));
    std::vector<uint32_t> hist((size_t)n_blocks * 9);
    HIP_TRY(hipMemcpyAsync(hist.data(), d_hist.p, hist.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    // the pooled decision (the kNN drivers' and the one-block case's)
    uint32_t pooled[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, pooled_n = 0;
    for (uint32_t b = 0; b < n_blocks; ++b) {
        for (int m = 0; m <= 8; ++m) {
            pooled[m] += hist[(size_t)b * 9 + m];
            pooled_n += hist[(size_t)b * 9 + m];
        }
    }
    const double eb_cost = eb_cost_of(rows, cols, self_mode);
    plan->lengths = best_lengths(pooled, pooled_n, rows->nk, eb_cost, &plan->alive_share);
    if (live_blocks > 1) {
        // per block.  Pass 1: every block's own sample decides whether it takes the early break at all; pass 2: the blocks that
        // do are pooled, and every block decides again with its estimate pulled towards that pool (weight: one block's sample).
        const uint8_t all = (uint8_t)rows->nk;
        std::vector<uint8_t> ke(n_blocks, all);
        std::vector<uint32_t> tot(n_blocks, 0u);
        double cold[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, cold_n = 0.0;
        for (uint32_t b = 0; b < n_blocks; ++b) {
            for (int m = 0; m <= 8; ++m) tot[b] += hist[(size_t)b * 9 + m];
            if (best_lengths(&hist[(size_t)b * 9], tot[b], rows->nk, eb_cost, nullptr) > 0) {
                for (int m = 0; m <= 8; ++m) cold[m] += hist[(size_t)b * 9 + m];
                cold_n += tot[b];
            }
        }
        int common = 0;
        if (cold_n > 0.0) {
            uint32_t cold_u[9];
            for (int m = 0; m <= 8; ++m) cold_u[m] = (uint32_t)std::min(cold[m], 4.0e9);
            common = best_lengths(cold_u, (uint32_t)std::min(cold_n, 4.0e9), rows->nk, eb_cost, nullptr);
            for (int m = 0; m <= 8; ++m) cold[m] /= cold_n;
        }
        bool differ = false;
        uint8_t first = 0;
        for (uint32_t b = 0; b < n_blocks; ++b) {
            if (self_mode && b % plan->blk_cols < b / plan->blk_cols) continue;   // below the diagonal: no pair
            const int own = best_lengths(&hist[(size_t)b * 9], tot[b], rows->nk, eb_cost, nullptr, cold_n > 0.0 ? cold : nullptr, cold_n > 0.0 ? (double)samples : 0.0, common);
            ke[b] = own > 0 ? (uint8_t)own : all;
            if (first == 0) first = ke[b];
            else if (ke[b] != first) differ = true;
        }
        if (differ) {
            if (self_mode) {   // (mirror: a tile on the diagonal may look a block up from either side)
                for (uint32_t r = 0; r < plan->blk_rows; ++r) {
                    for (uint32_t c = 0; c < r && c < plan->blk_cols; ++c) ke[(size_t)r * plan->blk_cols + c] = ke[(size_t)c * plan->blk_cols + r];
                }
            }
            plan->mixed = true;
            plan->block_ke = ke;
            HIP_TRY(hipMalloc((void **)&plan->d_block_ke, ke.size()));
            HIP_TRY(hipMemcpy(plan->d_block_ke, ke.data(), ke.size(), hipMemcpyHostToDevice));
        } else {
            plan->lengths = first == all ? 0 : (int)first;   // one mind: a plain launch
        }
    }
    keep();
    return SKL_OK;
}

int early_break_lengths(skl_ctx *ctx, const skl_sketches *rows, const skl_sketches *cols, int self_mode, int *lengths)
{
    *lengths = 0;
    // (the kNN bands' epilogue has no completeness branch and no segmented counts: capi_knn.cpp)
    if (rows->d_comp != nullptr || cols->d_comp != nullptr || rows->ss64 > (size_t)KSLICE_MAX_U16_CHUNKS) return SKL_OK;
    const EbPlan *plan = nullptr;
    SKL_TRY(early_break_plan(ctx, rows, cols, self_mode, 0.0, &plan));
    if (plan) *lengths = plan->lengths;
    return SKL_OK;
}

extern "C" int skl_ctx_early_break_blocks(skl_ctx *ctx, uint32_t *blk_rows, uint32_t *blk_cols, uint32_t *shift_rows, uint32_t *shift_cols,
                                          int *pooled_lengths, int *mixed, uint8_t *block_lengths, size_t capacity)
{
    SKL_TRY(ctx_bind(ctx));
    const EbPlan *p = ctx->eb_last_plan;
    if (blk_rows) *blk_rows = p ? p->blk_rows : 0u;
    if (blk_cols) *blk_cols = p ? p->blk_cols : 0u;
    if (shift_rows) *shift_rows = p ? p->shift_r : 0u;
    if (shift_cols) *shift_cols = p ? p->shift_c : 0u;
    if (pooled_lengths) *pooled_lengths = p ? p->lengths : 0;
    if (mixed) *mixed = p && p->mixed ? 1 : 0;
    if (p && p->mixed && block_lengths) {
        if (capacity < p->block_ke.size()) return fail(SKL_ERR_INVALID_ARG, "skl_ctx_early_break_blocks: %zu blocks, room for %zu", p->block_ke.size(), capacity);
        memcpy(block_lengths, p->block_ke.data(), p->block_ke.size());
    }
    return SKL_OK;
}

// Core of every dense call: rows [r0, r1) of the pair space into `dst` (device).
// elem_bytes is the output record size per pair.
int dense_band(skl_ctx *ctx, const skl_sketches *rows, const skl_sketches *cols,
                      const skl_dist_params *p, int mode, int jout, int self_mode, uint64_t r0,
                      uint64_t r1, void *dst_dev)
{
    const uint64_t n_cols = cols->n;
    const uint64_t base = self_mode ? cond_index(r0, r0 + 1, n_cols) : r0 * n_cols;
    const uint64_t pairs = self_mode ? self_rows_pairs(r0, r1, n_cols) : (r1 - r0) * n_cols;
    if (pairs == 0) return SKL_OK;
    const RoctxRange range_(mode == MODE_COREACC ? "skl:dense_band core/accessory" : mode == MODE_JACCARD ? "skl:dense_band single k" : "skl:dense_band bin-match counts");

    const bool coreacc = mode == MODE_COREACC;
    // Small core/acc launches run as (tile, k) workgroups producing counts + the epilogue
    // kernel (pair_kslice.hip): 5x the workgroups of the fused kernel and two columns per lane.
    const EbPlan *plan = nullptr;
    if (coreacc && (forced_kernel(ctx) == 0 || forced_kernel(ctx) == 4)) {
        SKL_TRY(early_break_plan(ctx, rows, cols, self_mode, p ? p->completeness_cutoff : 0.0, &plan));
        ctx->eb_last_plan = plan;
    }
    const bool eb_mixed = plan != nullptr && plan->mixed;               // the early break decided block by block
    const int eb_lengths = plan != nullptr && !eb_mixed ? plan->lengths : 0;
    const bool early = eb_lengths > 0 || eb_mixed;
    // (with the early break every launch takes the counts + epilogue form, whatever its size: three of the k-mer lengths,
    // 12 bytes of counts per pair through HBM -- nothing beside the two lengths not walked)
    const bool sliced = coreacc && (coreacc_runs_sliced(ctx, rows, pairs) || early);
    if (coreacc && (sliced || !fused_coreacc_ok(rows))) {
        // unfused: counts -> scratch2 -> epilogue kernel
        // (the counts scratch is bounded: a band whose counts would not fit COUNTS_SCRATCH_MAX is computed in two halves
        // of equal pair count, each into its slice of the destination -- only sketches beyond 65 535 bins or more than 6
        // k-mer lengths come here with that many pairs)
        const size_t nkw = eb_lengths > 0 ? (size_t)eb_lengths : rows->nk;   // k-mer lengths the pair kernel counts (block by block: planes)
        // U16 COUNTS (round 6): sketches of up to 1 023 chunks count at most 65 472 bins per length, so a launch without
        // chunk slices (no plane to add into) parks its counts as u16: half the scratch traffic of the stream
        const bool tiny = pairs * nkw < 2ull * 4ull * (uint64_t)ctx->n_cu * 2048ull;   // (launches that may be tail-sliced keep u32: slices ADD into a plane)
        const bool cnt_u16 = sliced && rows->ss64 <= (size_t)KSLICE_MAX_U16_CHUNKS && !tiny && ctx->knobs.k_slices <= 1 && ctx->knobs.counts_u16 &&
                             !ctx->knobs.fuse_epilogue && !ctx->knobs.epilogue_r5;   // (the A/B build's older epilogues read u32)
        const size_t cnt_bytes = cnt_u16 ? sizeof(uint16_t) : sizeof(uint32_t);
        // BAND PIPELINE (round 6).  The counts scratch is bounded, and a call whose counts do not fit is computed in row bands of
        // equal pair count.  From 64 Mi pairs on the bands are also what hides the epilogue: band i's epilogue (+ completion
        // of the pairs still in the running) is memory-bound, band i + 1's counts kernel is bound by the vector ALUs, so they
        // run side by side -- counts kernels on the context's stream, epilogues on its second stream, two counts buffers.
        // (the side-by-side run costs the counts kernel about what it hides of the epilogue -- an epilogue wave displaces a wave of the
        // counts kernel, which fills the register file by itself -- and pays only where the epilogue is heavy: from ~3 % of the
        // pairs still in the running.  n = 16 000 at 4.9 %: 18.5 against 19.4 ms; cfg 3 at 1.1 %: 782 against 748 ms.)
        // Since the blocked epilogue order (eb_blocked_order) covers that regime better -- n = 16 000: 17.2 ms blocked, 18.3-18.8
        // piped; cfg 3 at two lengths: 642 blocked, 775 piped -- the pipeline was off unless asked for (A/B build,
        // SKL_EB_PIPELINE=1; tests/test_gpu_early_break_r6.py keeps it exact).
        const bool blocked = early && eb_blocked_order(ctx, rows, plan, pairs);
        // ROUND 6, LATE: with the lean epilogue (58 VGPRs, 8 waves per SIMD, a third of the instructions) the side-by-side run pays
        // where it did not: 300 000 x 10 000 at 1.4 % still in the running 161.6 -> 154.0 ms, n = 30 000 at 2 048 bins 23.9 -> 23.2
        // (profiles/r06_epilogue_lean.md) -- on by itself wherever that kernel runs in the flat order.
        if (early) SKL_TRY(ensure_ytab(rows));   // (min_alive)
        const bool lean_like = !eb_mixed && (!(rows->d_comp && cols->d_comp) || (rows->comp_unit && cols->comp_unit)) && rows->min_alive != 0xFFFFFFFFu && nkw >= 2 && nkw <= 4;
        // (together with the blocked order it pays for the largest calls only: cfg 3 586 -> 575 ms, n = 40 000 95.4 -> 94.7, but
        // n = 16 000 in 4 bands 15.5 -> 18.0: from 2^30 pairs)
        bool piping = false;
        if (early) {
            const int pk = ctx->knobs.eb_pipeline;
            if (pk == 1) piping = !blocked && (eb_mixed || (plan != nullptr && plan->alive_share >= 0.03) || ctx->knobs.early_break >= 2);
            else if (pk == 2) piping = lean_like;
            else if (pk == -1) piping = lean_like && (!blocked || pairs >= (1ull << 30));
        }
        if (!ctx->eb_in_pipeline && r1 - r0 > 1 &&
            (pairs * nkw * cnt_bytes > COUNTS_SCRATCH_MAX || (piping && pairs >= (uint64_t)ctx->knobs.eb_pipeline_min))) {
            const uint64_t fit = std::max<uint64_t>(1, COUNTS_SCRATCH_MAX / (nkw * cnt_bytes));
            const uint64_t want = piping ? std::max<uint64_t>((uint64_t)ctx->knobs.eb_pipeline_min / 2, pairs / 8) : fit;
            const uint64_t n_bands = (pairs + std::min(fit, want) - 1) / std::min(fit, want);
            std::vector<uint64_t> cuts(1, r0);
            for (uint64_t b = 1; b < n_bands; ++b) {
                const uint64_t target = pairs * b / n_bands;   // pairs before the cut
                uint64_t cut;
                if (self_mode) {   // the first row whose predecessors hold at least `target` pairs
                    uint64_t lo = cuts.back() + 1, hi = r1 - 1;
                    while (lo < hi) {
                        const uint64_t m = (lo + hi) / 2;
                        if (self_rows_pairs(r0, m, n_cols) < target) lo = m + 1; else hi = m;
                    }
                    cut = lo;
                } else {
                    cut = r0 + (target + n_cols - 1) / n_cols;
                }
                cut = std::min<uint64_t>(std::max<uint64_t>(cut, cuts.back() + 1), r1 - 1);
                if (cut > cuts.back()) cuts.push_back(cut);
            }
            cuts.push_back(r1);
            const bool overlap = piping && cuts.size() > 2;
            if (overlap && !ctx->eb_events[0]) {
                for (auto &ev : ctx->eb_events) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            }
            ctx->eb_in_pipeline = true;
            int rc = SKL_OK;
            for (size_t b = 0; b + 1 < cuts.size() && rc == SKL_OK; ++b) {
                const int buf = (int)(b & 1);
                ctx->eb_pipe_buf = overlap ? buf : 0;
                ctx->eb_pipe_overlap = overlap;
                // (the epilogue that read this counts buffer two bands ago must be done before the counts kernel rewrites it)
                if (overlap && b >= 2) {
                    const hipError_t e = hipStreamWaitEvent(ctx->stream, ctx->eb_events[2 + buf], 0);
                    if (e != hipSuccess) rc = fail(SKL_ERR_HIP, "hipStreamWaitEvent: %s", hipGetErrorString(e));
                }
                const uint64_t base_b = self_mode ? cond_index(cuts[b], cuts[b] + 1, n_cols) : cuts[b] * n_cols;
                if (rc == SKL_OK) rc = dense_band(ctx, rows, cols, p, mode, jout, self_mode, cuts[b], cuts[b + 1], (char *)dst_dev + (base_b - base) * 2 * sizeof(float));
            }
            ctx->eb_in_pipeline = false;
            ctx->eb_pipe_overlap = false;
            ctx->eb_pipe_buf = 0;
            if (overlap) {   // the output belongs to the context's stream again
                for (int buf = 0; buf < 2; ++buf) {
                    const hipError_t e = hipStreamWaitEvent(ctx->stream, ctx->eb_events[2 + buf], 0);
                    if (e != hipSuccess && rc == SKL_OK) rc = fail(SKL_ERR_HIP, "hipStreamWaitEvent: %s", hipGetErrorString(e));
                }
                if (rc == SKL_OK) ctx->last_kernel += "; " + std::to_string(cuts.size() - 1) + " row bands, each band's epilogue beside the next band's counts kernel";
            }
            return rc;
        }
        const bool piped = ctx->eb_in_pipeline && ctx->eb_pipe_overlap;
        // (a stream of its own: the banded host output copies band i back on aux_stream while band i + 1 is computed, and must not
        // queue behind that band's epilogues)
        if (piped && !ctx->epi_stream) HIP_TRY(hipStreamCreateWithFlags(&ctx->epi_stream, hipStreamNonBlocking));
        hipStream_t epi_stream = piped ? ctx->epi_stream : ctx->stream;
        PairArgs g;
        SKL_TRY(fill_args(rows, cols, p, MODE_COUNTS, 0, &g));
        if (early) {
            g.k_count = (uint32_t)nkw;
            g.cnt_pair_stride = nkw;
        }
        if (eb_mixed) {
            g.xcd_interleave = 1;
            g.block_ke = plan->d_block_ke;
            g.blk_shift_r = plan->shift_r;
            g.blk_shift_c = plan->shift_c;
            g.blk_cols = plan->blk_cols;
        }
        void *counts = nullptr;
        const uint32_t k_slices = sliced ? choose_k_slices(ctx, rows->ss64) : 1u;
        // A launch with fewer (tile, k) units than resident workgroup slots cannot fill the chip with one
        // workgroup per unit: most SIMDs hold 0-2 waves and the launch takes the time of ONE unit at a
        // lone wave's issue rate whatever its size (0.055-0.064 ms from 100 to 500 genomes).  Such launches are
        // cut into tail_slices chunk slices per unit -- and so is the last, partial round of any launch
        // SKL_TAIL_MAX_PCT lets through (default 90: launches of up to 0.9 estimated rounds, where the
        // whole launch is that partial round; the partial round of a longer launch gains nothing,
        // profiles/r02_ab_tail_slices.jsonl).  Slice 0 of a unit stores, the others add into a second
        // plane that is zero on entry and re-zeroed by the epilogue.
        const uint64_t est_units = pairs * nkw / 2048;
        const uint64_t slots = 4ull * (uint64_t)ctx->n_cu;
        // (launches of less than 1/16 round -- ~200 genomes -- are cut twice as fine when the sketch allows it)
        // (round 4: any sketch size is cut -- slices of whole stages, the last one shorter, slice_plan() -- e.g. the 157
        // chunks of `-s 10000`; and launches over sketches beyond 65 535 bins slice their last round however many rounds
        // they have: a unit of 1 563+ chunks dwarfs the fixed cost of a workgroup)
        const bool big_sketch = rows->ss64 > (size_t)KSLICE_MAX_U16_CHUNKS;
        const uint32_t tail_wanted = ctx->knobs.tail_slices == 4 && est_units * 16 <= slots && rows->ss64 >= 64 ? 8u : (uint32_t)ctx->knobs.tail_slices;
        uint32_t tail_chunks = 0;
        const uint32_t tail_slices = slice_plan((uint32_t)rows->ss64, tail_wanted, &tail_chunks);
        bool tail = sliced && k_slices == 1u && tail_slices > 1u && forced_kernel(ctx) == 0 &&
                    ((big_sketch && est_units <= 16 * slots) || est_units * 100 <= (uint64_t)std::max(0ll, ctx->knobs.tail_max_pct) * slots);
        // MID BAND (round 3): from half the 32-row threshold up to it (4-8 Mi pair x k evaluations: 1 300-1 790 genomes at
        // 5 k-mer lengths) the launch is a handful of rounds of workgroups whichever tile it takes, and its last, partial
        // round decides: 32 x 128 tiles with THAT round cut into 2 chunk slices are 0.7-4.6 % ahead of 16 x 128 tiles
        // there (profiles/r03_ab_mid_sizes.jsonl, r03_ab_mid_band.jsonl); plain 32-row tiles are not (+-4 %).  Above the
        // band plain 32-row tiles, below it 16-row tiles (cfg 2: 0.156 against 0.162 ms).
        const uint64_t evals = pairs * nkw;
        const long long t32 = ctx->knobs.tile32_min;
        const bool mid_band = ctx->knobs.mid_band && !tail && sliced && k_slices == 1u && t32 > 0 && tail_slices > 1u &&
                              forced_kernel(ctx) == 0 && rows->ss64 >= 16 && evals * 2 >= (uint64_t)t32 && evals < (uint64_t)t32;
        uint32_t tail_slices_eff = tail_slices, tail_chunks_eff = tail_chunks;
        if (mid_band) {
            tail = true;
            tail_slices_eff = slice_plan((uint32_t)rows->ss64, 2u, &tail_chunks_eff);
        }
        const bool two_planes = tail;
        const bool u16_now = cnt_u16 && !tail && k_slices == 1u;
        const size_t plane_bytes = pairs * nkw * (u16_now ? sizeof(uint16_t) : sizeof(uint32_t));
        SKL_TRY(ctx_scratch(ctx, plane_bytes * std::max(two_planes ? 2u : 1u, k_slices), &counts, piped && ctx->eb_pipe_buf ? 15 : 1));
        g.cnt_u16 = u16_now ? 1u : 0u;
        if (sliced) {   // k-major scratch: coalesced stores from the (tile, k[, chunk slice]) workgroups
            g.cnt_pair_stride = 1;
            g.cnt_k_stride = pairs;
            g.k_sliced = 1;
            g.k_slices = k_slices;
            g.tail_slices = tail ? tail_slices_eff : 0u;
            if (tail) g.slice_chunks = tail_chunks_eff;
            else if (k_slices > 1u) (void)slice_plan((uint32_t)rows->ss64, std::min(8u, (uint32_t)ctx->knobs.k_slices), &g.slice_chunks);
            g.mid_band = mid_band ? 1u : 0u;
        }
        if (two_planes) {
            void *plane1 = (char *)counts + plane_bytes;
            if (ctx->clean_plane1 != plane1 || ctx->clean_plane1_bytes != plane_bytes) {
                HIP_TRY(hipMemsetAsync(plane1, 0, plane_bytes, ctx->stream));
                ctx->clean_plane1 = plane1;
                ctx->clean_plane1_bytes = plane_bytes;
            }
        } else {
            ctx->clean_plane1 = nullptr;   // this launch writes the scratch in another layout
        }
        g.row_begin = (uint32_t)r0;
        g.row_end = (uint32_t)r1;
        g.self_mode = self_mode;
        g.out_base = base;
        g.out = counts;
        SKL_TRY(ensure_ytab(rows));   // (before the pair launch: nothing may fail between it and the epilogue)
        // FUSED EPILOGUE (round 5): a plain k-sliced launch -- one workgroup per (tile, k-mer length), no chunk slices -- finishes
        // its pairs itself: the workgroup that completes a tile's k-mer lengths reads the tile's counts back and stores (core, acc)
        // (pair_kslice.hip, FUSE).  The second launch (10 us at cfg 2, 6 of them the cost of any dependent launch) is gone.
        bool fused = false;
#ifdef SKL_AB
        // (A/B build only, SKL_FUSE_EPILOGUE=1: measured SLOWER than the second launch at cfg 2 -- 0.154 against 0.145 ms per step:
        // the arrival pattern itself is free, but with the k-major dispatch order every tile completes in the launch's last round
        // and one workgroup then does a whole tile's regressions alone while the chip empties; profiles/r05_fused_epilogue.md)
        if (sliced && !early && k_slices == 1u && !two_planes && ctx->knobs.fuse_epilogue && forced_kernel(ctx) == 0 &&
            rows->nk <= (size_t)MAX_FUSED_K && kslice_supported(g, MODE_COUNTS, true) && rows->ss64 <= (size_t)KSLICE_MAX_U16_CHUNKS) {
            // arrival counters, one per tile of the launch (16-row tiles at most), counted modulo nk: zero once per (buffer, nk)
            const size_t tiles_max = ((r1 - r0 + 15) / 16 + 1) * ((cols->n + 127) / 128 + 1) + 64;
            void *fc = nullptr;
            const size_t had = ctx->scratch_bytes[11];
            SKL_TRY(ctx_scratch(ctx, tiles_max * sizeof(uint32_t), &fc, 11));
            if (ctx->scratch_bytes[11] != had || ctx->fuse_counter_k != rows->nk) {
                HIP_TRY(hipMemsetAsync(fc, 0, ctx->scratch_bytes[11], ctx->stream));
                ctx->fuse_counter_k = rows->nk;
            }
            g.fuse_counter = (uint32_t *)fc;
            g.fuse_variant = (uint32_t)env_int("SKL_FUSE_VARIANT", 0);
            g.fuse_out = (float *)dst_dev;
            g.ytab = rows->d_ytab;
            for (size_t t = 0; t < rows->nk; ++t) g.kf[t] = (double)rows->kmers[t];
            g.cutoff = p->completeness_cutoff;
            fused = true;
        }
#endif
        // Plane 1 holds partial counts from the pair launch until the epilogue has re-zeroed it: it is
        // "clean" again only once that epilogue is enqueued.  Any early return in between leaves it marked dirty.
        const void *const plane1_clean = ctx->clean_plane1;
        ctx->clean_plane1 = nullptr;
        SKL_TRY(timed_pair_launch(ctx, g, MODE_COUNTS));
        if (fused) {
            ctx->last_kernel += " + fused core/accessory epilogue (last workgroup of a tile)";
            return SKL_OK;
        }
        EpilogueArgs e;
        memset(&e, 0, sizeof e);
        e.counts = (uint32_t *)counts;
        e.pair_stride = g.cnt_pair_stride;
        e.k_stride = g.cnt_k_stride;
        e.n_pairs = pairs;
        e.nk = (uint32_t)nkw;
        e.nk_total = (uint32_t)rows->nk;
        if (early) {
            e.rows_ref = rows->d_rows;
            e.cols_ref = cols->d_rows;
            e.alive_count = ctx->eb_counter;
            ctx->eb_pairs += pairs;
            if (eb_mixed) {
                e.block_ke = plan->d_block_ke;
                e.blk_shift_r = plan->shift_r;
                e.blk_shift_c = plan->shift_c;
                e.blk_cols = plan->blk_cols;
                ctx->last_kernel += " + early break: block by block (" + std::to_string(plan->blk_rows) + " x " + std::to_string(plan->blk_cols) + " blocks of sample ids), the pairs still in the running completed by the epilogue";
            } else {
                ctx->last_kernel += " + early break: " + std::to_string(nkw) + " of " + std::to_string(rows->nk) + " k-mer lengths counted, the pairs still in the running completed by the epilogue";
            }
        }
        e.min_alive = rows->min_alive;
        e.cnt_u16 = g.cnt_u16;
        e.row_end = (uint32_t)r1;
        e.xcd_shift = ctx_xcd_shift(ctx);
        e.blocked = early && eb_blocked_order(ctx, rows, plan, pairs) ? 1u : 0u;
        e.blk_row_shift = (uint32_t)ctx->knobs.eb_blk_row_shift;
        if (e.blocked) ctx->last_kernel += " (epilogue in blocks of 1 024 x 256 pairs per XCD)";
        // (the workgroup's row slices in LDS pay from ~8 completions per workgroup of 256 pairs on: n = 16 000 at 4 096 bins, 4.9 %
        // still in the running: 18.0 against 18.7 ms; at 2 048 bins, 1.4 %: 30.8 against 27.3 -- profiles/r06_epilogue_forms.md)
        e.ahead = ctx->knobs.eb_ahead ? 1u : 0u;
        e.lean = ctx->knobs.eb_lean ? 1u : 0u;
        e.comp_lean = rows->d_comp && cols->d_comp && rows->comp_unit && cols->comp_unit ? 1u : 0u;
        e.lds_rows = ctx->knobs.eb_lds_rows && plan != nullptr && plan->alive_share >= 0.03 ? 1u : 0u;
        e.ss64 = (uint32_t)rows->ss64;
        e.n_slices = sliced ? ctx->last_count_planes : 1u;
        e.rezero_plane1 = sliced && ctx->last_tail ? 1u : 0u;
        e.nA_rows = (uint32_t)rows->n;
        e.nB_cols = (uint32_t)cols->n;
        e.row_begin = (uint32_t)r0;
        e.self_mode = self_mode;
        e.n_total = (uint32_t)cols->n;
        e.out_base = base;
        e.has_comp = g.has_comp;
        e.log_variant = g.log_variant;
        e.ytab = rows->d_ytab;
        e.compA = rows->d_comp;
        e.compB = cols->d_comp;
        e.cutoff = p->completeness_cutoff;
        e.tolerance = g.tolerance;
        e.kf = rows->d_kf;
        e.out = (float *)dst_dev;
        if (piped) {   // this band's epilogue on the second stream, behind its counts kernel
            HIP_TRY(hipEventRecord(ctx->eb_events[ctx->eb_pipe_buf], ctx->stream));
            HIP_TRY(hipStreamWaitEvent(epi_stream, ctx->eb_events[ctx->eb_pipe_buf], 0));
        }
#ifdef SKL_AB
        if (ctx->knobs.epilogue_r5 && !eb_mixed && !(early && (e.has_comp || rows->ss64 > (size_t)KSLICE_MAX_U16_CHUNKS))) {
            if (!early) e.nk_total = 0;
            HIP_TRY(launch_coreacc_epilogue(e, epi_stream));   // round 5's epilogue: alive pairs completed where they are found (A/B timing)
        } else
#endif
        {
            if (coreacc_epilogue_is_lean(e)) ctx->last_kernel += e.cnt_u16 ? " [lean epilogue]" : " [lean epilogue, sliced counts]";
            HIP_TRY(launch_coreacc_epilogue_r6(e, epi_stream));
        }
        if (piped) HIP_TRY(hipEventRecord(ctx->eb_events[2 + ctx->eb_pipe_buf], epi_stream));
        // (an empty launch, or one another kernel took, leaves plane 1 not known to be zero)
        if (two_planes && ctx->last_tail) ctx->clean_plane1 = plane1_clean;
        return SKL_OK;
    }
    PairArgs g;
    SKL_TRY(fill_args(rows, cols, p, mode, jout, &g));
    g.row_begin = (uint32_t)r0;
    g.row_end = (uint32_t)r1;
    g.self_mode = self_mode;
    g.out_base = base;
    g.out = dst_dev;
    if (mode == MODE_JACCARD) {
        // A single-k launch smaller than the chip has the same problem as a small core/accessory one
        // (one workgroup per tile on a quarter of the SIMDs, each wave at its own issue interval: 0.058
        // ms from 200 to 1 000 genomes) and takes the same cure: bin-match counts in tail_slices chunk
        // slices per tile + an epilogue launch that turns the summed counts into the f32 output.
        const uint64_t est_units = pairs / 2048;
        const uint64_t slots = 4ull * (uint64_t)ctx->n_cu;
        const uint32_t tail_wanted = ctx->knobs.tail_slices == 4 && est_units * 16 <= slots && rows->ss64 >= 64 ? 8u : (uint32_t)ctx->knobs.tail_slices;
        uint32_t tail_chunks = 0;
        const uint32_t tail_slices = slice_plan((uint32_t)rows->ss64, tail_wanted, &tail_chunks);
        const bool tail = tail_slices > 1u && forced_kernel(ctx) == 0 &&
                          ((rows->ss64 > (size_t)KSLICE_MAX_U16_CHUNKS && est_units <= 16 * slots) ||
                           est_units * 100 <= (uint64_t)std::max(0ll, ctx->knobs.tail_max_pct) * slots);
        if (tail) {
            void *counts = nullptr;
            const size_t plane_bytes = pairs * sizeof(uint32_t);
            SKL_TRY(ctx_scratch(ctx, plane_bytes * 2, &counts, 1));
            void *plane1 = (char *)counts + plane_bytes;
            if (ctx->clean_plane1 != plane1 || ctx->clean_plane1_bytes != plane_bytes) {
                HIP_TRY(hipMemsetAsync(plane1, 0, plane_bytes, ctx->stream));
                ctx->clean_plane1 = plane1;
                ctx->clean_plane1_bytes = plane_bytes;
            }
            g.cnt_pair_stride = 1;
            g.cnt_k_stride = pairs;
            g.k_sliced = 1;
            g.k_slices = 1;
            g.tail_slices = tail_slices;
            g.slice_chunks = tail_chunks;
            g.out = counts;
            const void *const plane1_clean = ctx->clean_plane1;   // as above: dirty until the epilogue is enqueued
            ctx->clean_plane1 = nullptr;
            SKL_TRY(timed_pair_launch(ctx, g, MODE_COUNTS));
            EpilogueArgs e;
            memset(&e, 0, sizeof e);
            e.counts = (uint32_t *)counts;
            e.pair_stride = 1;
            e.k_stride = pairs;
            e.n_pairs = pairs;
            e.nk = 1;
            e.ss64 = (uint32_t)rows->ss64;
            e.n_slices = 2;
            e.rezero_plane1 = 1;
            e.nA_rows = (uint32_t)rows->n;
            e.nB_cols = (uint32_t)cols->n;
            e.row_begin = (uint32_t)r0;
            e.self_mode = self_mode;
            e.n_total = (uint32_t)cols->n;
            e.out_base = base;
            e.has_comp = g.has_comp;
            e.log_variant = g.log_variant;
            e.compA = rows->d_comp;
            e.compB = cols->d_comp;
            e.cutoff = p->completeness_cutoff;
            e.jaccard_out = 1;
            e.jout = jout;
            e.kf0 = g.kf[0];
            e.dtab = g.dtab;
            e.out = (float *)dst_dev;
            HIP_TRY(launch_coreacc_epilogue(e, ctx->stream));
            if (ctx->last_tail) ctx->clean_plane1 = plane1_clean;
            return SKL_OK;
        }
    }
    return timed_pair_launch(ctx, g, mode);
}

static size_t record_bytes(const skl_sketches *s, int mode)
{
    if (mode == MODE_COUNTS) return s->nk * sizeof(uint32_t);
    return mode == MODE_COREACC ? 2 * sizeof(float) : sizeof(float);
}

// Dense driver: whole row range either straight into a device destination, or banded
// through scratch and copied back to a host destination.
static int dense_rows(skl_ctx *ctx, const skl_sketches *rows, const skl_sketches *cols,
                      const skl_dist_params *p, int mode, int jout, int self_mode, uint64_t r0,
                      uint64_t r1, void *out, int out_on_device)
{
    SKL_TRY(ctx_bind(ctx));
    if (!out) return fail(SKL_ERR_INVALID_ARG, "out is null");
    const uint64_t n_cols = cols->n;
    const uint64_t row_limit = self_mode ? (n_cols ? n_cols - 1 : 0) : rows->n;
    if (r0 > r1 || r1 > (self_mode ? n_cols : rows->n)) {
        return fail(SKL_ERR_INVALID_ARG, "row range [%llu, %llu) out of bounds", (unsigned long long)r0,
                    (unsigned long long)r1);
    }
    r1 = std::min<uint64_t>(r1, row_limit);
    if (r1 <= r0 || n_cols == 0) return SKL_OK;
    const size_t rec = record_bytes(rows, mode);
    if (out_on_device) {
        return dense_band(ctx, rows, cols, p, mode, jout, self_mode, r0, r1, out);
    }
    // host destination: bands of at most BAND_BYTES through two device buffers -- band i is
    // copied back on the auxiliary stream while band i + 1 is computed
    const uint64_t first = self_mode ? cond_index(r0, r0 + 1, n_cols) : r0 * n_cols;
    void *dev[2] = {nullptr, nullptr};
    const uint64_t all_pairs = self_mode ? self_rows_pairs(r0, r1, n_cols) : (r1 - r0) * n_cols;
    const size_t band_alloc = (size_t)std::min<uint64_t>(BAND_BYTES, all_pairs * rec);
    SKL_TRY(ctx_scratch(ctx, band_alloc, &dev[0], 0));
    SKL_TRY(ctx_scratch(ctx, all_pairs * rec > BAND_BYTES ? band_alloc : 16, &dev[1], 3));
    // A copy into pageable host memory does not return before it is done (the runtime stages it), so the copy of band i is
    // ISSUED after band i + 1's kernels are in the queue: the host blocks in the copy while the device computes.
    struct PendingCopy {
        void *dst = nullptr;
        const void *src = nullptr;
        size_t bytes = 0;
        int buf = 0;
    } pending;
    auto issue_copy = [&](const PendingCopy &c) -> int {
        HIP_TRY(hipStreamWaitEvent(ctx->aux_stream, ctx->knn_pair_done[c.buf], 0));
        HIP_TRY(hipMemcpyAsync(c.dst, c.src, c.bytes, hipMemcpyDeviceToHost, ctx->aux_stream));
        HIP_TRY(hipEventRecord(ctx->knn_topk_done[c.buf], ctx->aux_stream));
        return SKL_OK;
    };
    uint64_t b0 = r0;
    size_t it = 0;
    while (b0 < r1) {
        uint64_t b1 = b0;
        uint64_t pairs = 0;
        while (b1 < r1) {
            const uint64_t row_pairs = self_mode ? (n_cols - 1 - b1) : n_cols;
            if (pairs && (pairs + row_pairs) * rec > BAND_BYTES) break;
            pairs += row_pairs;
            ++b1;
        }
        const int buf = (int)(it & 1);
        void *band = dev[buf];
        if (pairs * rec > band_alloc) {   // a single row wider than a band: its own buffer
            if (pending.bytes) {
                SKL_TRY(issue_copy(pending));
                pending.bytes = 0;
            }
            HIP_TRY(hipStreamSynchronize(ctx->aux_stream));
            SKL_TRY(ctx_scratch(ctx, pairs * rec, &dev[buf], buf == 0 ? 0 : 3));
            band = dev[buf];
        }
        // the copy that read this buffer two bands ago must be done before it is overwritten
        if (it >= 2) HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->knn_topk_done[buf], 0));
        SKL_TRY(dense_band(ctx, rows, cols, p, mode, jout, self_mode, b0, b1, band));
        HIP_TRY(hipEventRecord(ctx->knn_pair_done[buf], ctx->stream));
        if (pending.bytes) SKL_TRY(issue_copy(pending));   // the previous band's, behind this band's kernels
        const uint64_t off = (self_mode ? cond_index(b0, b0 + 1, n_cols) : b0 * n_cols) - first;
        pending.dst = (char *)out + off * rec;
        pending.src = band;
        pending.bytes = pairs * rec;
        pending.buf = buf;
        b0 = b1;
        ++it;
    }
    if (pending.bytes) SKL_TRY(issue_copy(pending));
    HIP_TRY(hipStreamSynchronize(ctx->aux_stream));   // host memory is complete on return
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SKL_OK;
}

static int dense_mode(const skl_dist_params *p, int *mode, int *jout)
{
    if (p->dist_type == SKL_DIST_COREACC) {
        *mode = MODE_COREACC;
        *jout = 0;
    } else {
        *mode = MODE_JACCARD;
        *jout = p->ani ? JOUT_ANI : JOUT_DIST;
    }
    return SKL_OK;
}

// ---------------------------------------------------------------------------
// dense entry points
// ---------------------------------------------------------------------------

extern "C" int skl_self_dists_rows(skl_ctx *ctx, const skl_sketches *s, const skl_dist_params *p,
                                   size_t row_begin, size_t row_end, float *out, int out_on_device)
{
    SKL_TRY(check_params(s, s, p));
    int mode, jout;
    dense_mode(p, &mode, &jout);
    return dense_rows(ctx, s, s, p, mode, jout, 1, row_begin, row_end, out, out_on_device);
}

extern "C" int skl_self_dists_all(skl_ctx *ctx, const skl_sketches *s, const skl_dist_params *p,
                                  float *out, int out_on_device)
{
    if (!s) return fail(SKL_ERR_INVALID_ARG, "null sketches");
    if (s->n < 2) {
        SKL_TRY(check_params(s, s, p));
        return SKL_OK;  // empty upper triangle
    }
    return skl_self_dists_rows(ctx, s, p, 0, s->n, out, out_on_device);
}

extern "C" int skl_cross_dists_rows(skl_ctx *ctx, const skl_sketches *ref,
                                    const skl_sketches *query, const skl_dist_params *p,
                                    size_t ref_begin, size_t ref_end, float *out,
                                    int out_on_device)
{
    SKL_TRY(check_params(ref, query, p));
    int mode, jout;
    dense_mode(p, &mode, &jout);
    return dense_rows(ctx, ref, query, p, mode, jout, 0, ref_begin, ref_end, out, out_on_device);
}

extern "C" int skl_cross_dists_all(skl_ctx *ctx, const skl_sketches *ref,
                                   const skl_sketches *query, const skl_dist_params *p, float *out,
                                   int out_on_device)
{
    if (!ref || !query) return fail(SKL_ERR_INVALID_ARG, "null sketches");
    return skl_cross_dists_rows(ctx, ref, query, p, 0, ref->n, out, out_on_device);
}

extern "C" int skl_self_binmatch(skl_ctx *ctx, const skl_sketches *s, uint32_t *out,
                                 int out_on_device)
{
    if (!s) return fail(SKL_ERR_INVALID_ARG, "null sketches");
    if (s->n < 2) return SKL_OK;
    skl_dist_params p = {SKL_DIST_JACCARD, 0, 0, 0.0};
    return dense_rows(ctx, s, s, &p, MODE_COUNTS, 0, 1, 0, s->n, out, out_on_device);
}

extern "C" int skl_cross_binmatch(skl_ctx *ctx, const skl_sketches *ref,
                                  const skl_sketches *query, uint32_t *out, int out_on_device)
{
    skl_dist_params p = {SKL_DIST_JACCARD, 0, 0, 0.0};
    SKL_TRY(check_params(ref, query, &p));
    return dense_rows(ctx, ref, query, &p, MODE_COUNTS, 0, 0, 0, ref->n, out, out_on_device);
}

// ---------------------------------------------------------------------------
// one-shot host forms
// ---------------------------------------------------------------------------

extern "C" int skl_self_dists_all_host(const uint64_t *bins, size_t n_samples, size_t nk,
                                       const size_t *kmers, size_t sketchsize64,
                                       const skl_dist_params *p, const double *completeness,
                                       float *out)
{
    skl_ctx *ctx = nullptr;
    SKL_TRY(skl_ctx_create(0, &ctx));
    skl_sketches *s = nullptr;
    int rc = skl_sketches_create(ctx, bins, 0, n_samples, nk, kmers, sketchsize64, &s);
    if (rc == SKL_OK && completeness) rc = skl_sketches_set_completeness(s, completeness);
    if (rc == SKL_OK) rc = skl_self_dists_all(ctx, s, p, out, 0);
    skl_sketches_destroy(s);
    skl_ctx_destroy(ctx);
    return rc;
}

extern "C" int skl_cross_dists_all_host(const uint64_t *ref_bins, size_t n_ref,
                                        const uint64_t *query_bins, size_t n_query, size_t nk,
                                        const size_t *kmers, size_t sketchsize64,
                                        const skl_dist_params *p, const double *ref_completeness,
                                        const double *query_completeness, float *out)
{
    skl_ctx *ctx = nullptr;
    SKL_TRY(skl_ctx_create(0, &ctx));
    skl_sketches *r = nullptr, *q = nullptr;
    int rc = skl_sketches_create(ctx, ref_bins, 0, n_ref, nk, kmers, sketchsize64, &r);
    if (rc == SKL_OK) rc = skl_sketches_create(ctx, query_bins, 0, n_query, nk, kmers, sketchsize64, &q);
    if (rc == SKL_OK && ref_completeness) rc = skl_sketches_set_completeness(r, ref_completeness);
    if (rc == SKL_OK && query_completeness) rc = skl_sketches_set_completeness(q, query_completeness);
    if (rc == SKL_OK) rc = skl_cross_dists_all(ctx, r, q, p, out, 0);
    skl_sketches_destroy(q);
    skl_sketches_destroy(r);
    skl_ctx_destroy(ctx);
    return rc;
}
