// capi.cpp -- implementation of include/sketchlib_dist.h on top of kernels.hip.
//
// Host-side responsibilities: device slabs (the MultiSketch bins in the two layouts
// the pair kernel wants), the samebits -> ln(J) / distance tables (computed with the
// host libm, i.e. bit-identical to what the reference's f64::ln produces on this
// machine), banding of large pair spaces through bounded scratch, and mapping the
// reference's panics to status codes.
//
// There is deliberately no CPU compute path here: if HIP cannot run, calls fail.
#include "../../include/sketchlib_dist.h"

#include <hip/hip_runtime.h>

#include <algorithm>
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

#include "kernels.h"

using namespace skl;

// ---------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------

static thread_local std::string g_last_error;

static int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            return fail(e_ == hipErrorOutOfMemory ? SKL_ERR_OOM : SKL_ERR_HIP, "%s: %s",   \
                        #expr, hipGetErrorString(e_));                                     \
        }                                                                                  \
    } while (0)

#define SKL_TRY(expr)             \
    do {                          \
        int rc_ = (expr);         \
        if (rc_ != SKL_OK) return rc_; \
    } while (0)

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

struct skl_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    // grow-only scratch
    void *scratch[6] = {};   // 0/3: key bands, 1: counts, 2: kNN staging, 4/5: turned key bands (symmetric kNN)
    size_t scratch_bytes[6] = {};
    hipStream_t aux_stream = nullptr;   // top-k of band i runs here while band i+1 is computed
    // band pipelines (kNN: pair kernel -> top-k; dense to host: pair kernel -> D2H copy):
    // "producer finished buffer b" / "consumer finished buffer b"
    hipEvent_t knn_pair_done[2] = {nullptr, nullptr}, knn_topk_done[2] = {nullptr, nullptr};
    // timing of pair-kernel launches of the last call
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    size_t events_used = 0;
    std::string last_kernel;
    TileScratch tile_scratch;           // device table of the balanced tile enumeration
    std::set<skl_sketches *> sketches;  // slabs created on this context
};

static int ctx_bind(skl_ctx *ctx)
{
    if (!ctx) return fail(SKL_ERR_INVALID_ARG, "null context");
    HIP_TRY(hipSetDevice(ctx->device));
    return SKL_OK;
}

static int ctx_scratch(skl_ctx *ctx, size_t bytes, void **out, int which = 0)
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

extern "C" int skl_ctx_destroy(skl_ctx *ctx)
{
    if (!ctx) return SKL_OK;
    std::lock_guard<std::mutex> lock(g_registry_mutex);
    if (!g_live_ctx.erase(ctx)) return SKL_OK;  // already destroyed
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->aux_stream) (void)hipStreamSynchronize(ctx->aux_stream);
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
    if (ctx->tile_scratch.d_prefix) (void)hipFree(ctx->tile_scratch.d_prefix);
    if (ctx->tile_scratch.h_staging) (void)hipHostFree(ctx->tile_scratch.h_staging);
    if (ctx->tile_scratch.staged) (void)hipEventDestroy(ctx->tile_scratch.staged);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    if (ctx->aux_stream) (void)hipStreamDestroy(ctx->aux_stream);
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

extern "C" const char *skl_ctx_last_kernel(skl_ctx *ctx) { return ctx ? ctx->last_kernel.c_str() : ""; }

extern "C" int skl_ctx_timing_reset(skl_ctx *ctx)
{
    SKL_TRY(ctx_bind(ctx));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->events_used = 0;
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

// SKL_KERNEL = smem | lds | ksplit | kslice forces one implementation (0: dispatcher's choice)
static int forced_kernel()
{
    const char *e = getenv("SKL_KERNEL");
    if (!e) return 0;
    if (strcmp(e, "smem") == 0) return 1;
    if (strcmp(e, "lds") == 0) return 2;
    if (strcmp(e, "ksplit") == 0) return 3;
    if (strcmp(e, "kslice") == 0) return 4;
    return 0;
}

// One tile computation, four implementations (the dispatcher's rule is in DESIGN.md 4.2):
//   kslice (pair_kslice.hip)  default: 16 x 128 tiles, chunks split over the 4 waves, rows by
//                             LDS DMA; one workgroup per (tile, k) for small launches and for
//                             single-k Jaccard, all k + fused regression otherwise
//   ksplit (pair_ksplit.hip)  fallback for shapes kslice does not take (ss64 > 1023)
//   lds    (pair_lds.hip)     R x 256/512 tiles, rows broadcast from LDS (A/B only)
//   smem   (kernels.hip)      rows through the scalar cache (A/B only)
// SKL_KERNEL = kslice | ksplit | lds | smem forces one.
static hipError_t dispatch_pair_kernel(const PairArgs &args, int mode, int na, hipStream_t stream,
                                       std::string *name, TileScratch &tiles)
{
    static const char *mode_names[] = {"COUNTS", "JACCARD", "COREACC"};
    const std::string m = mode_names[mode];
    // (tuning knobs are read on every call so an A/B run can interleave variants in one
    // process: scripts/ab_sweep.py)
    const int forced = forced_kernel();
    const int forced_kslice_shape = env_int("SKL_KSLICE_SHAPE", 0);  // 81, 82, 161, 162
    const int forced_rows = env_int("SKL_KSPLIT_ROWS", 0);           // 4 or 8
    const uint64_t rows = args.row_end - args.row_begin;
    const uint64_t pairs = args.self_mode ? rows * args.nB / 2 : rows * (uint64_t)args.nB;
    const bool small = pairs < (8ull << 20);
    if (forced == 1) {
        *name = "skl::pair_kernel<NA=" + std::to_string(na) + ", " + m + "> (scalar-cache rows)";
        return launch_pair_kernel(args, mode, na, stream);
    }
    // SKL_KERNEL=kslice (or no override): the chunk-split kernel, k-sliced for small launches
    // (core/acc then arrives here as MODE_COUNTS from dense_band), all-k fused otherwise.
    if (forced == 0 || forced == 4) {
        // single-k Jaccard: the sliced and the all-k form are the same work, the sliced one
        // compiles to fewer registers
        const bool sliced = mode == MODE_JACCARD || (mode == MODE_COUNTS && (small || args.k_sliced));
        if (kslice_supported(args, mode, sliced)) {
            const int shape = forced_kslice_shape ? forced_kslice_shape : 162;
            *name = "skl::pair_kernel_kslice<R=" + std::to_string(shape / 10) + ", JL=" + std::to_string(shape % 10) +
                    ", " + m + (sliced ? ", k-sliced" : ", all k") + "> (" + std::to_string(shape / 10) + "x" +
                    std::to_string((shape % 10) * 64) + " tiles, chunks split over 4 waves)";
            return launch_pair_kernel_kslice(args, mode, shape, sliced, tiles, stream);
        }
    }
    if (forced == 3 || (forced != 2 && small)) {
        int r = forced_rows ? forced_rows : 8;  // 8 >= 4 from n = 1000 up once XCDs are balanced (sweep 18)
        *name = "skl::pair_kernel_ksplit<R=" + std::to_string(r) + ", " + m + "> (" + std::to_string(r) +
                "x64 tiles, chunks split over 4 waves)";
        return launch_pair_kernel_ksplit(args, mode, r, tiles, stream);
    }
    const int shape = choose_lds_shape(rows, args.nB, args.self_mode, mode);
    *name = "skl::pair_kernel_lds<R=" + std::to_string(shape / 10) + ", JL=" + std::to_string(shape % 10) +
            ", " + m + "> (" + std::to_string(shape / 10) + "x" + std::to_string((shape % 10) * 256) + " tiles)";
    return launch_pair_kernel_lds(args, mode, shape, tiles, stream);
}

// Launch the pair kernel bracketed by HIP events on the context's stream.
static int timed_pair_launch(skl_ctx *ctx, const PairArgs &args, int mode, int na)
{
    constexpr size_t MAX_EVENTS = 4096;
    if (ctx->events_used >= MAX_EVENTS) {
        HIP_TRY(dispatch_pair_kernel(args, mode, na, ctx->stream, &ctx->last_kernel, ctx->tile_scratch));
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
    HIP_TRY(dispatch_pair_kernel(args, mode, na, ctx->stream, &ctx->last_kernel, ctx->tile_scratch));
    HIP_TRY(hipEventRecord(ev.second, ctx->stream));
    return SKL_OK;
}

// ---------------------------------------------------------------------------
// sketch slabs
// ---------------------------------------------------------------------------

struct skl_sketches {
    skl_ctx *ctx = nullptr;
    size_t n = 0, nk = 0, ss64 = 0;
    std::vector<size_t> kmers;
    uint64_t *d_rows = nullptr;  // reference layout + A_PAD_ROWS zero rows (scalar operand)
    uint4 *d_lanes = nullptr;    // lane-interleaved layout (vector operand), built on demand
    double *d_comp = nullptr;    // completeness or null
    double *d_ytab = nullptr;    // ln J table [64*ss64+1]
    double *d_kf = nullptr;      // k-mer lengths as f64 [nk]
    std::map<std::pair<int, size_t>, float *> d_dtab;  // (jout, k_idx) -> f32 table
    size_t sample_words() const { return nk * ss64 * BBITS; }
};

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

static int check_params(const skl_sketches *a, const skl_sketches *b, const skl_dist_params *p)
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

static bool fused_coreacc_ok(const skl_sketches *s)
{
    return s->nk <= (size_t)MAX_FUSED_K && 64 * s->ss64 <= 0xFFFFu;
}

// Fill the operand / epilogue fields common to every launch.  `rows` is the scalar
// operand (A), `cols` the lane operand (B).
static int fill_args(const skl_sketches *rows, const skl_sketches *cols, const skl_dist_params *p,
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
static bool coreacc_runs_sliced(const skl_sketches *s, uint64_t pairs)
{
    const int forced = forced_kernel();
    if (s->ss64 > 1023 || (forced != 0 && forced != 4)) return false;   // another kernel forced: never slice
    const long long limit = env_int("SKL_SLICED_MAX_PAIRS", SLICED_MAX_PAIRS);
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

constexpr size_t BAND_BYTES = 512ull << 20;  // scratch bound for host-destined / banded output

// Core of every dense call: rows [r0, r1) of the pair space into `dst` (device).
// elem_bytes is the output record size per pair.
static int dense_band(skl_ctx *ctx, const skl_sketches *rows, const skl_sketches *cols,
                      const skl_dist_params *p, int mode, int jout, int self_mode, uint64_t r0,
                      uint64_t r1, void *dst_dev)
{
    const uint64_t n_cols = cols->n;
    const uint64_t base = self_mode ? cond_index(r0, r0 + 1, n_cols) : r0 * n_cols;
    const uint64_t pairs = self_mode ? self_rows_pairs(r0, r1, n_cols) : (r1 - r0) * n_cols;
    if (pairs == 0) return SKL_OK;

    const bool coreacc = mode == MODE_COREACC;
    // Small core/acc launches run as (tile, k) workgroups producing counts + the epilogue
    // kernel (pair_kslice.hip): 5x the workgroups of the fused kernel and two columns per lane.
    const bool sliced = coreacc && coreacc_runs_sliced(rows, pairs);
    if (coreacc && (sliced || !fused_coreacc_ok(rows))) {
        // unfused: counts -> scratch2 -> epilogue kernel
        PairArgs g;
        SKL_TRY(fill_args(rows, cols, p, MODE_COUNTS, 0, &g));
        void *counts = nullptr;
        SKL_TRY(ctx_scratch(ctx, pairs * rows->nk * sizeof(uint32_t), &counts, 1));
        if (sliced) {   // k-major scratch: coalesced stores from the (tile, k) workgroups
            g.cnt_pair_stride = 1;
            g.cnt_k_stride = pairs;
            g.k_sliced = 1;
        }
        g.row_begin = (uint32_t)r0;
        g.row_end = (uint32_t)r1;
        g.self_mode = self_mode;
        g.out_base = base;
        g.out = counts;
        const int na = choose_na(r1 - r0, n_cols, self_mode, MODE_COUNTS);
        SKL_TRY(timed_pair_launch(ctx, g, MODE_COUNTS, na));
        SKL_TRY(ensure_ytab(rows));
        EpilogueArgs e;
        memset(&e, 0, sizeof e);
        e.counts = (const uint32_t *)counts;
        e.pair_stride = g.cnt_pair_stride;
        e.k_stride = g.cnt_k_stride;
        e.n_pairs = pairs;
        e.nk = (uint32_t)rows->nk;
        e.ss64 = (uint32_t)rows->ss64;
        e.nA_rows = (uint32_t)rows->n;
        e.nB_cols = (uint32_t)cols->n;
        e.row_begin = (uint32_t)r0;
        e.self_mode = self_mode;
        e.n_total = (uint32_t)cols->n;
        e.out_base = base;
        e.has_comp = g.has_comp;
        e.ytab = rows->d_ytab;
        e.compA = rows->d_comp;
        e.compB = cols->d_comp;
        e.cutoff = p->completeness_cutoff;
        e.tolerance = g.tolerance;
        e.kf = rows->d_kf;
        e.out = (float *)dst_dev;
        HIP_TRY(launch_coreacc_epilogue(e, ctx->stream));
        return SKL_OK;
    }
    PairArgs g;
    SKL_TRY(fill_args(rows, cols, p, mode, jout, &g));
    g.row_begin = (uint32_t)r0;
    g.row_end = (uint32_t)r1;
    g.self_mode = self_mode;
    g.out_base = base;
    g.out = dst_dev;
    const int na = choose_na(r1 - r0, n_cols, self_mode, mode);
    return timed_pair_launch(ctx, g, mode, na);
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
            HIP_TRY(hipStreamSynchronize(ctx->aux_stream));
            SKL_TRY(ctx_scratch(ctx, pairs * rec, &dev[buf], buf == 0 ? 0 : 3));
            band = dev[buf];
        }
        // the copy that read this buffer two bands ago must be done before it is overwritten
        if (it >= 2) HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->knn_topk_done[buf], 0));
        SKL_TRY(dense_band(ctx, rows, cols, p, mode, jout, self_mode, b0, b1, band));
        HIP_TRY(hipEventRecord(ctx->knn_pair_done[buf], ctx->stream));
        HIP_TRY(hipStreamWaitEvent(ctx->aux_stream, ctx->knn_pair_done[buf], 0));
        const uint64_t off = (self_mode ? cond_index(b0, b0 + 1, n_cols) : b0 * n_cols) - first;
        HIP_TRY(hipMemcpyAsync((char *)out + off * rec, band, pairs * rec, hipMemcpyDeviceToHost, ctx->aux_stream));
        HIP_TRY(hipEventRecord(ctx->knn_topk_done[buf], ctx->aux_stream));
        b0 = b1;
        ++it;
    }
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
// sparse kNN: dense row bands into scratch, then a row-wise top-k kernel
// ---------------------------------------------------------------------------

// Running top-k states of a kNN call: (sortable key, sample id[, second value]) x knn per row.
namespace {
struct KnnState {
    uint32_t *key = nullptr, *idx = nullptr;
    float *d1 = nullptr;
    ~KnnState()
    {
        if (key) (void)hipFree(key);
        if (idx) (void)hipFree(idx);
        if (d1) (void)hipFree(d1);
    }
};
}  // namespace

// Rows per band of the symmetric drivers: about 8 bands per participant (7/16 of the pair
// evaluations saved), each band at least 32 M pairs, four band buffers within `budget` bytes.
static size_t symmetric_band_rows(size_t n, size_t rec, size_t budget, size_t participants)
{
    auto up16 = [](size_t x) { return (x + 15) / 16 * 16; };
    const size_t budget_rows = std::max<size_t>(16, budget / 4 / (n * rec) / 16 * 16);
    const size_t parts = std::max<size_t>(1, participants);
    return std::min(budget_rows, std::max(up16((n + 8 * parts - 1) / (8 * parts)), up16((32ull << 20) / n + 1)));
}

static int knn_state_init(KnnState &st, size_t rows, size_t knn, bool coreacc, hipStream_t stream)
{
    const size_t items = rows * knn;
    HIP_TRY(hipMalloc((void **)&st.key, items * sizeof(uint32_t)));
    HIP_TRY(hipMalloc((void **)&st.idx, items * sizeof(uint32_t)));
    if (coreacc) HIP_TRY(hipMalloc((void **)&st.d1, items * sizeof(float)));
    HIP_TRY(hipMemsetAsync(st.key, 0xFF, items * sizeof(uint32_t), stream));   // empty
    HIP_TRY(hipMemsetAsync(st.idx, 0xFF, items * sizeof(uint32_t), stream));
    return SKL_OK;
}

// Symmetric self kNN (whole matrix in one call): band [b0, b1) is compared with the columns
// from b0 on only.  The pair kernel stores every record twice -- row-major for the rows of the
// band, and turned (pair_kslice.hip, out_t) as candidates of the rows below the band -- and
// both copies are merged into a running per-row top-k (topk_merge_kernel), so each (i, j) is
// evaluated once instead of twice (the reference evaluates both, mod.rs:148-171; distances are
// symmetric).  Same neighbours, same order as the row-by-row form.
static bool knn_symmetric_ok(const skl_sketches *s, const skl_dist_params *p)
{
    if (s->ss64 > 1023) return false;
    if (p->dist_type == SKL_DIST_COREACC && !fused_coreacc_ok(s)) return false;
    const int forced = forced_kernel();
    if (forced != 0 && forced != 4) return false;          // the turned store lives in pair_kslice.hip
    const long long shape = env_int("SKL_KSLICE_SHAPE", 0);
    return shape == 0 || shape == 81 || shape == 82 || shape == 161 || shape == 162;
}

// The bands `bands` (ascending indices; band b = rows [b*band_rows, (b+1)*band_rows)) merged into
// the running states `st` of all n rows.
static int knn_symmetric_bands(skl_ctx *ctx, const skl_sketches *s, const skl_dist_params *p, size_t knn,
                               size_t band_rows, const std::vector<uint32_t> &bands, bool overlap, KnnState &st)
{
    const size_t n = s->n;
    const bool coreacc = p->dist_type == SKL_DIST_COREACC;
    const int mode = coreacc ? MODE_COREACC : MODE_JACCARD;
    const int jout = coreacc ? 0 : (p->ani ? JOUT_ANI_KEY : JOUT_DIST);
    const size_t rec = coreacc ? 2 * sizeof(float) : sizeof(float);
    const size_t t_stride = (band_rows + 15) / 16 * 16;
    void *kband[2] = {nullptr, nullptr}, *tband[2] = {nullptr, nullptr};
    SKL_TRY(ctx_scratch(ctx, band_rows * n * rec, &kband[0], 0));
    SKL_TRY(ctx_scratch(ctx, n * t_stride * rec, &tband[0], 4));
    kband[1] = kband[0];
    tband[1] = tband[0];
    if (overlap) {
        SKL_TRY(ctx_scratch(ctx, band_rows * n * rec, &kband[1], 3));
        SKL_TRY(ctx_scratch(ctx, n * t_stride * rec, &tband[1], 5));
    }
    hipStream_t topk_stream = overlap ? ctx->aux_stream : ctx->stream;
    if (overlap) {   // the states were cleared on the context's stream, the merges run on the other one
        HIP_TRY(hipEventRecord(ctx->knn_pair_done[0], ctx->stream));
        HIP_TRY(hipStreamWaitEvent(topk_stream, ctx->knn_pair_done[0], 0));
    }

    const size_t jb_words = s->nk * s->ss64 * 7 * 64;   // uint4 per 64-column block of the lane slab
    size_t it = 0;
    for (const uint32_t band : bands) {
        const size_t b0 = (size_t)band * band_rows;
        const size_t b1 = std::min(n, b0 + band_rows);
        const int buf = overlap ? (int)(it & 1) : 0;
        if (overlap && it >= 2) HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->knn_topk_done[buf], 0));
        // the band against the column view that starts at the 64-column block holding b0
        const size_t col0 = b0 / 64 * 64;
        PairArgs g;
        SKL_TRY(fill_args(s, s, p, mode, jout, &g));
        g.B += (b0 / 64) * jb_words;
        g.nB = (uint32_t)(n - col0);
        if (g.compB) g.compB += col0;
        g.row_begin = (uint32_t)b0;
        g.row_end = (uint32_t)b1;
        g.self_mode = 0;
        g.out_base = (uint64_t)b0 * g.nB;
        g.out = kband[buf];
        g.out_t = b1 < n ? (float *)tband[buf] : nullptr;
        g.t_col_begin = (uint32_t)(b1 - col0);
        g.t_stride = (uint32_t)t_stride;
        SKL_TRY(timed_pair_launch(ctx, g, mode, choose_na(b1 - b0, g.nB, 0, mode)));
        if (overlap) {
            HIP_TRY(hipEventRecord(ctx->knn_pair_done[buf], ctx->stream));
            HIP_TRY(hipStreamWaitEvent(topk_stream, ctx->knn_pair_done[buf], 0));
        }
        TopkMergeArgs m;
        memset(&m, 0, sizeof m);
        m.knn = (uint32_t)knn;
        m.stride2 = coreacc ? 2 : 1;
        m.run_key = st.key;
        m.run_idx = st.idx;
        m.run_d1 = st.d1;
        m.streaming = env_int("SKL_TOPK_STREAM", 1) != 0;
        // rows of the band: columns [b0, n) minus themselves (the view's first b0 - col0 columns
        // reached them turned, from earlier bands)
        m.keys = (const float *)kband[buf];
        m.key_stride = (uint64_t)g.nB * m.stride2;
        m.rows = (uint32_t)(b1 - b0);
        m.cols = g.nB;
        m.id_base = (uint32_t)col0;
        m.skip_below = (uint32_t)b0;
        m.self_id_base = m.state_row_base = (uint32_t)b0;
        HIP_TRY(launch_topk_merge(m, topk_stream));
        // rows below the band: the band's samples as their candidates
        m.keys = (const float *)tband[buf];
        m.key_stride = (uint64_t)t_stride * m.stride2;
        m.rows = (uint32_t)(n - b1);
        m.cols = (uint32_t)(b1 - b0);
        m.id_base = (uint32_t)b0;
        m.skip_below = 0;
        m.self_id_base = m.state_row_base = (uint32_t)b1;
        HIP_TRY(launch_topk_merge(m, topk_stream));
        if (overlap) HIP_TRY(hipEventRecord(ctx->knn_topk_done[buf], topk_stream));
        ++it;
    }
    if (overlap && it) {   // the states (and the band buffers) belong to the context's stream again
        HIP_TRY(hipEventRecord(ctx->knn_topk_done[0], topk_stream));
        HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->knn_topk_done[0], 0));
    }
    return SKL_OK;
}

static int knn_self_symmetric(skl_ctx *ctx, const skl_sketches *s, const skl_dist_params *p, size_t knn,
                              size_t band_rows, bool overlap, uint64_t *d_idx, float *d_d0, float *d_d1)
{
    const size_t n = s->n;
    const bool coreacc = p->dist_type == SKL_DIST_COREACC;
    KnnState st;
    SKL_TRY(knn_state_init(st, n, knn, coreacc, ctx->stream));
    std::vector<uint32_t> bands((n + band_rows - 1) / band_rows);
    for (size_t b = 0; b < bands.size(); ++b) bands[b] = (uint32_t)b;
    SKL_TRY(knn_symmetric_bands(ctx, s, p, knn, band_rows, bands, overlap, st));
    HIP_TRY(launch_topk_finalize(st.key, st.idx, st.d1, n * knn, (!coreacc && p->ani) ? 1 : 0, d_idx, d_d0, d_d1,
                                 ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));   // the running states are freed on return
    return SKL_OK;
}

// Row-by-row kNN: dense bands of records into scratch, then a per-row top-k.  With two bands
// the top-k of band i (memory / LDS bound, on the auxiliary stream) runs while the pair kernel
// of band i + 1 (VALU bound) fills the other one.  The top-k is the streaming one of the
// symmetric driver (topk_merge_kernel), fed a whole row at once.
static int knn_rows_banded(skl_ctx *ctx, const skl_sketches *rows, const skl_sketches *cands,
                           const skl_dist_params *p, size_t knn, int self_mode, size_t r0, size_t r1,
                           size_t band_rows, bool overlap, uint64_t *d_idx, float *d_d0, float *d_d1)
{
    const bool coreacc = p->dist_type == SKL_DIST_COREACC;
    const int mode = coreacc ? MODE_COREACC : MODE_JACCARD;
    const int jout = coreacc ? 0 : (p->ani ? JOUT_ANI_KEY : JOUT_DIST);
    const size_t rec = coreacc ? 2 * sizeof(float) : sizeof(float);
    const size_t n_cand = cands->n;
    void *band[2] = {nullptr, nullptr};
    SKL_TRY(ctx_scratch(ctx, band_rows * n_cand * rec, &band[0], 0));
    band[1] = band[0];
    if (overlap) SKL_TRY(ctx_scratch(ctx, band_rows * n_cand * rec, &band[1], 3));
    hipStream_t topk_stream = overlap ? ctx->aux_stream : ctx->stream;
    KnnState st;
    SKL_TRY(knn_state_init(st, r1 - r0, knn, coreacc, ctx->stream));
    if (overlap) {   // the states are cleared on the context's stream, the merges run on the other one
        HIP_TRY(hipEventRecord(ctx->knn_pair_done[0], ctx->stream));
        HIP_TRY(hipStreamWaitEvent(topk_stream, ctx->knn_pair_done[0], 0));
    }

    size_t it = 0;
    for (size_t b0 = r0; b0 < r1; b0 += band_rows, ++it) {
        const size_t b1 = std::min(r1, b0 + band_rows);
        const int buf = overlap ? (int)(it & 1) : 0;
        // the top-k that read this buffer two bands ago must be done before it is overwritten
        if (overlap && it >= 2) HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->knn_topk_done[buf], 0));
        SKL_TRY(dense_band(ctx, rows, cands, p, mode, jout, 0, b0, b1, band[buf]));
        if (overlap) {
            HIP_TRY(hipEventRecord(ctx->knn_pair_done[buf], ctx->stream));
            HIP_TRY(hipStreamWaitEvent(topk_stream, ctx->knn_pair_done[buf], 0));
        }
        TopkMergeArgs m;
        memset(&m, 0, sizeof m);
        m.knn = (uint32_t)knn;
        m.stride2 = coreacc ? 2 : 1;
        m.run_key = st.key;
        m.run_idx = st.idx;
        m.run_d1 = st.d1;
        m.streaming = env_int("SKL_TOPK_STREAM", 1) != 0;
        m.key_stride = (uint64_t)n_cand * m.stride2;
        m.rows = (uint32_t)(b1 - b0);
        m.self_id_base = self_mode ? (uint32_t)b0 : 0xFFFFFFFFu;
        m.state_row_base = (uint32_t)(b0 - r0);
        m.keys = (const float *)band[buf];
        m.cols = (uint32_t)n_cand;
        m.id_base = 0;
        HIP_TRY(launch_topk_merge(m, topk_stream));
        if (overlap) HIP_TRY(hipEventRecord(ctx->knn_topk_done[buf], topk_stream));
    }
    HIP_TRY(launch_topk_finalize(st.key, st.idx, st.d1, (r1 - r0) * knn, (!coreacc && p->ani) ? 1 : 0, d_idx, d_d0,
                                 d_d1, topk_stream));
    if (overlap) {   // results (and the band buffers) belong to the context's stream again
        HIP_TRY(hipEventRecord(ctx->knn_topk_done[0], topk_stream));
        HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->knn_topk_done[0], 0));
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));   // the running states are freed on return
    return SKL_OK;
}

static int knn_rows(skl_ctx *ctx, const skl_sketches *rows, const skl_sketches *cands,
                    const skl_dist_params *p, size_t knn, int self_mode, size_t r0, size_t r1,
                    uint64_t *out_idx, float *out_d0, float *out_d1, int out_on_device)
{
    SKL_TRY(ctx_bind(ctx));
    if (!out_idx || !out_d0) return fail(SKL_ERR_INVALID_ARG, "output pointers are null");
    const bool coreacc = p->dist_type == SKL_DIST_COREACC;
    if (coreacc && !out_d1) return fail(SKL_ERR_INVALID_ARG, "out_d1 is required for core/accessory");
    if (r0 > r1 || r1 > rows->n) return fail(SKL_ERR_INVALID_ARG, "row range out of bounds");
    const size_t n_cand = cands->n;
    const size_t max_knn = n_cand > (size_t)(self_mode ? 1 : 0) ? n_cand - (self_mode ? 1 : 0) : 0;
    if (knn == 0 || knn > max_knn) {
        return fail(SKL_ERR_INVALID_ARG, "knn=%zu must be in [1, %zu]", knn, max_knn);
    }
    if (knn > 2048) return fail(SKL_ERR_INVALID_ARG, "knn=%zu exceeds the device limit of 2048", knn);
    if (r1 == r0) return SKL_OK;

    const size_t rec = coreacc ? 2 * sizeof(float) : sizeof(float);
    // the key band lives only on the device: take up to a quarter of the free HBM (<= 8 GiB)
    // so that the row-wise top-k kernel has thousands of rows (= workgroups) per launch
    size_t free_b = 0, total_b = 0;
    size_t band_bytes = BAND_BYTES;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
        band_bytes = std::max(band_bytes, std::min<size_t>(free_b / 4, 8ull << 30));
    }
    size_t band_rows = std::max<size_t>(1, band_bytes / 2 / (n_cand * rec));   // two key bands
    const size_t forced_band_rows = (size_t)std::max(0ll, env_int("SKL_KNN_BAND_ROWS", 0));  // test knob: force several bands
    if (forced_band_rows) band_rows = forced_band_rows;
    band_rows = std::min(band_rows, r1 - r0);
    // The whole self matrix: evaluate each pair once (knn_self_symmetric) when that leaves bands
    // worth launching -- about 8 of them (7/16 of the pair evaluations saved), each at least 32 M
    // pairs, within four band buffers of up to half the free HBM (<= 32 GiB) together.
    bool symmetric = self_mode && r0 == 0 && r1 == n_cand && knn_symmetric_ok(rows, p) &&
                     env_int("SKL_KNN_SYMMETRIC", 1) != 0;   // (0: A/B against the row-by-row form)
    if (symmetric) {
        size_t budget = band_bytes;
        if (free_b) budget = std::max(budget, std::min<size_t>(free_b / 2, 32ull << 30));
        const size_t want = forced_band_rows ? forced_band_rows : symmetric_band_rows(n_cand, rec, budget, 1);
        if (want >= n_cand) symmetric = false;
        else band_rows = want;
    }
    const bool overlap = env_int("SKL_KNN_OVERLAP", 1) != 0 && band_rows < r1 - r0;

    // device staging for host-destined results
    uint64_t *d_idx = out_idx;
    float *d_d0 = out_d0, *d_d1 = out_d1;
    const size_t items = (r1 - r0) * knn;
    if (!out_on_device) {
        void *stage = nullptr;
        SKL_TRY(ctx_scratch(ctx, items * (sizeof(uint64_t) + 2 * sizeof(float)), &stage, 2));
        d_idx = (uint64_t *)stage;
        d_d0 = (float *)(d_idx + items);
        d_d1 = d_d0 + items;
    }
    if (symmetric) {
        SKL_TRY(knn_self_symmetric(ctx, rows, p, knn, band_rows, overlap, d_idx, d_d0, d_d1));
    } else {
        SKL_TRY(knn_rows_banded(ctx, rows, cands, p, knn, self_mode, r0, r1, band_rows, overlap, d_idx, d_d0, d_d1));
    }
    if (!out_on_device) {
        HIP_TRY(hipMemcpyAsync(out_idx, d_idx, items * sizeof(uint64_t), hipMemcpyDeviceToHost,
                               ctx->stream));
        HIP_TRY(hipMemcpyAsync(out_d0, d_d0, items * sizeof(float), hipMemcpyDeviceToHost,
                               ctx->stream));
        if (coreacc) {
            HIP_TRY(hipMemcpyAsync(out_d1, d_d1, items * sizeof(float), hipMemcpyDeviceToHost,
                                   ctx->stream));
        }
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    return SKL_OK;
}

extern "C" int skl_self_dists_knn_rows(skl_ctx *ctx, const skl_sketches *s,
                                       const skl_dist_params *p, size_t knn, size_t row_begin,
                                       size_t row_end, uint64_t *out_idx, float *out_d0,
                                       float *out_d1, int out_on_device)
{
    SKL_TRY(check_params(s, s, p));
    return knn_rows(ctx, s, s, p, knn, 1, row_begin, row_end, out_idx, out_d0, out_d1,
                    out_on_device);
}

extern "C" int skl_self_dists_knn(skl_ctx *ctx, const skl_sketches *s, const skl_dist_params *p,
                                  size_t knn, uint64_t *out_idx, float *out_d0, float *out_d1,
                                  int out_on_device)
{
    if (!s) return fail(SKL_ERR_INVALID_ARG, "null sketches");
    return skl_self_dists_knn_rows(ctx, s, p, knn, 0, s->n, out_idx, out_d0, out_d1, out_on_device);
}

// ---------------------------------------------------------------------------
// GPU sketching
// ---------------------------------------------------------------------------

namespace {
inline uint64_t h_rotl1(uint64_t v) { return (v << 1) | (v >> 63); }
inline uint64_t h_swapbits033(uint64_t v)
{
    const uint64_t x = (v ^ (v >> 33)) & 1ull;
    return v ^ (x | (x << 33));
}
inline uint64_t h_srol(uint64_t v) { return h_swapbits033(h_rotl1(v)); }
}  // namespace

extern "C" int skl_sketch_signs(skl_ctx *ctx, const uint8_t *codes, const uint64_t *code_begin,
                                const uint64_t *offsets, const uint64_t *offset_begin, size_t n_samples,
                                const size_t *kmers, size_t nk, uint64_t num_bins, int rc, uint64_t *out_signs)
{
    SKL_TRY(ctx_bind(ctx));
    if (!code_begin || !offset_begin || !kmers || !out_signs) return fail(SKL_ERR_INVALID_ARG, "null argument");
    if (n_samples == 0 || nk == 0) return SKL_OK;
    if (num_bins == 0) return fail(SKL_ERR_INVALID_ARG, "num_bins is zero");
    const uint64_t n_codes = code_begin[n_samples], n_offs = offset_begin[n_samples];
    if ((n_codes && !codes) || (n_offs && !offsets)) return fail(SKL_ERR_INVALID_ARG, "null argument");
    static const uint64_t seeds_f[4] = {0x3c8bfbb395c60474ull, 0x3193c18562a02b4cull, 0x295549f54be24456ull,
                                        0x20323ed082572324ull};   // src/hashing/nthash_tables.rs:4-16
    std::vector<uint32_t> k32(nk);
    std::vector<uint64_t> top_f(nk * 4), top_r(nk * 4);
    for (size_t ki = 0; ki < nk; ++ki) {
        if (kmers[ki] == 0 || kmers[ki] > 0xFFFFu) return fail(SKL_ERR_INVALID_ARG, "k-mer length out of range");
        k32[ki] = (uint32_t)kmers[ki];
        for (int b = 0; b < 4; ++b) {
            uint64_t f = seeds_f[b], r = seeds_f[b ^ 2];
            for (size_t m = 1; m < kmers[ki]; ++m) {
                f = h_srol(f);
                r = h_srol(r);
            }
            top_f[ki * 4 + b] = f;
            top_r[ki * 4 + b] = r;
        }
    }
    const uint64_t span = (uint64_t)sketch_span();
    std::vector<uint64_t> span_begin(n_samples + 1, 0);
    for (size_t s = 0; s < n_samples; ++s) {
        if (code_begin[s + 1] < code_begin[s] || offset_begin[s + 1] < offset_begin[s]) {
            return fail(SKL_ERR_INVALID_ARG, "sample ranges must not decrease");
        }
        span_begin[s + 1] = span_begin[s] + (code_begin[s + 1] - code_begin[s] + span - 1) / span;
    }
    struct DevBuf {
        void *p = nullptr;
        ~DevBuf() { if (p) (void)hipFree(p); }
    } d_codes, d_cb, d_offs, d_ob, d_sb, d_k, d_tf, d_tr, d_signs;
    auto upload = [&](DevBuf &b, const void *src, size_t bytes) -> int {
        HIP_TRY(hipMalloc(&b.p, std::max<size_t>(bytes, 16)));
        if (bytes) HIP_TRY(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
        return SKL_OK;
    };
    SKL_TRY(upload(d_codes, codes, n_codes));
    SKL_TRY(upload(d_cb, code_begin, (n_samples + 1) * sizeof(uint64_t)));
    SKL_TRY(upload(d_offs, offsets, n_offs * sizeof(uint64_t)));
    SKL_TRY(upload(d_ob, offset_begin, (n_samples + 1) * sizeof(uint64_t)));
    SKL_TRY(upload(d_sb, span_begin.data(), (n_samples + 1) * sizeof(uint64_t)));
    SKL_TRY(upload(d_k, k32.data(), nk * sizeof(uint32_t)));
    SKL_TRY(upload(d_tf, top_f.data(), top_f.size() * sizeof(uint64_t)));
    SKL_TRY(upload(d_tr, top_r.data(), top_r.size() * sizeof(uint64_t)));
    const size_t sign_bytes = n_samples * nk * num_bins * sizeof(uint64_t);
    HIP_TRY(hipMalloc(&d_signs.p, sign_bytes));
    HIP_TRY(hipMemsetAsync(d_signs.p, 0xFF, sign_bytes, ctx->stream));   // u64::MAX
    HIP_TRY(hipStreamSynchronize(ctx->stream));   // pageable uploads done before the vectors die

    SketchArgs a;
    memset(&a, 0, sizeof a);
    a.codes = (const uint8_t *)d_codes.p;
    a.code_begin = (const uint64_t *)d_cb.p;
    a.offsets = (const uint64_t *)d_offs.p;
    a.offset_begin = (const uint64_t *)d_ob.p;
    a.span_begin = (const uint64_t *)d_sb.p;
    a.n_spans = span_begin[n_samples];
    a.n_samples = (uint32_t)n_samples;
    a.nk = (uint32_t)nk;
    a.kmers = (const uint32_t *)d_k.p;
    a.top_f = (const uint64_t *)d_tf.p;
    a.top_r = (const uint64_t *)d_tr.p;
    a.num_bins = num_bins;
    const uint64_t sign_mod = (1ull << 61) - 1;
    a.bin_size = (sign_mod + num_bins - 1) / num_bins;   // SIGN_MOD.div_ceil(num_bins), sketch/mod.rs:170
    a.inv_bin_size = 1.0 / (double)a.bin_size;
    a.rc = rc ? 1 : 0;
    a.signs = (uint64_t *)d_signs.p;
    {
        if (ctx->events_used == ctx->events.size() && ctx->events.size() < 4096) {
            hipEvent_t e0, e1;
            HIP_TRY(hipEventCreate(&e0));
            HIP_TRY(hipEventCreate(&e1));
            ctx->events.emplace_back(e0, e1);
        }
        const bool timed = ctx->events_used < ctx->events.size();
        if (timed) HIP_TRY(hipEventRecord(ctx->events[ctx->events_used].first, ctx->stream));
        HIP_TRY(launch_sketch_signs(a, ctx->stream));
        if (timed) HIP_TRY(hipEventRecord(ctx->events[ctx->events_used++].second, ctx->stream));
    }
    ctx->last_kernel = "skl::nthash_binmin_kernel (256 window starts per thread, rolling canonical ntHash, atomicMin per bin)";
    HIP_TRY(hipMemcpyAsync(out_signs, d_signs.p, sign_bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SKL_OK;
}

// Candidate-list kNN: the device half of the reference's self_dists_knn_precluster
// (src/distances/mod.rs:399-553).  Host pointers in, host pointers out.
namespace {
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
};
}  // namespace

// Distances + ragged top-k for candidate lists that are already on the device.  host_offsets is
// the host copy of the CSR offsets (the 64-candidate work items are cut on the host).
static int knn_from_device_csr(skl_ctx *ctx, const skl_sketches *s, const skl_dist_params *p, size_t knn,
                               const uint64_t *host_offsets, const uint64_t *d_off, const uint32_t *d_cand,
                               uint64_t *out_idx, float *out_d0)
{
    const size_t n = s->n;
    const uint64_t total = host_offsets[n];
    std::vector<uint32_t> work_row;
    std::vector<uint64_t> work_start;
    for (size_t i = 0; i < n; ++i) {
        for (uint64_t c0 = host_offsets[i]; c0 < host_offsets[i + 1]; c0 += 64) {
            work_row.push_back((uint32_t)i);
            work_start.push_back(c0);
        }
    }
    DevBuf d_wrow, d_wstart, d_keys, d_idx, d_d0;
    HIP_TRY(hipMalloc(&d_wrow.p, std::max<size_t>(work_row.size() * sizeof(uint32_t), 16)));
    HIP_TRY(hipMalloc(&d_wstart.p, std::max<size_t>(work_start.size() * sizeof(uint64_t), 16)));
    if (!work_row.empty()) {
        HIP_TRY(hipMemcpyAsync(d_wrow.p, work_row.data(), work_row.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipMemcpyAsync(d_wstart.p, work_start.data(), work_start.size() * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    }
    HIP_TRY(hipMalloc(&d_keys.p, std::max<size_t>(total * sizeof(float), 16)));
    HIP_TRY(hipMalloc(&d_idx.p, n * knn * sizeof(uint64_t)));
    HIP_TRY(hipMalloc(&d_d0.p, n * knn * sizeof(float)));
    HIP_TRY(hipStreamSynchronize(ctx->stream));   // pageable uploads done before the vectors die

    PairArgs g;
    SKL_TRY(fill_args(s, s, p, MODE_JACCARD, p->ani ? JOUT_ANI_KEY : JOUT_DIST, &g));
    CandArgs c;
    memset(&c, 0, sizeof c);
    c.row_offsets = d_off;
    c.cand = d_cand;
    c.work_row = (const uint32_t *)d_wrow.p;
    c.work_start = (const uint64_t *)d_wstart.p;
    c.n_work = work_row.size();
    c.keys = (float *)d_keys.p;
    {   // bracketed like the pair kernels, so skl_ctx_kernel_ms() reports it
        if (ctx->events_used == ctx->events.size() && ctx->events.size() < 4096) {
            hipEvent_t a, b;
            HIP_TRY(hipEventCreate(&a));
            HIP_TRY(hipEventCreate(&b));
            ctx->events.emplace_back(a, b);
        }
        const bool timed = ctx->events_used < ctx->events.size();
        if (timed) HIP_TRY(hipEventRecord(ctx->events[ctx->events_used].first, ctx->stream));
        HIP_TRY(launch_pair_cand(c, g, ctx->stream));
        if (timed) HIP_TRY(hipEventRecord(ctx->events[ctx->events_used++].second, ctx->stream));
    }
    ctx->last_kernel = "skl::pair_cand_kernel (row x 64 candidates per wave, candidate gather from the reference layout)";
    TopkArgs t;
    memset(&t, 0, sizeof t);
    t.keys = (const float *)d_keys.p;
    t.rows = (uint32_t)n;
    t.cols = 0;
    t.stride2 = 1;
    t.knn = (uint32_t)knn;
    t.self_mode = 0;
    t.row_begin = 0;
    t.ani_undo = p->ani ? 1 : 0;
    t.out_idx = (uint64_t *)d_idx.p;
    t.out_d0 = (float *)d_d0.p;
    t.out_d1 = nullptr;
    t.row_offsets = d_off;
    t.col_ids = d_cand;
    HIP_TRY(launch_topk(t, ctx->stream));
    HIP_TRY(hipMemcpyAsync(out_idx, d_idx.p, n * knn * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(out_d0, d_d0.p, n * knn * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SKL_OK;
}

static int check_candidate_call(skl_ctx *ctx, const skl_sketches *s, const skl_dist_params *p, size_t knn)
{
    SKL_TRY(check_params(s, s, p));
    SKL_TRY(ctx_bind(ctx));
    if (p->dist_type != SKL_DIST_JACCARD) {
        return fail(SKL_ERR_INVALID_ARG, "Prefilter only available for single k-mer distances");  // mod.rs:549-551
    }
    if (knn == 0 || knn > 2048) return fail(SKL_ERR_INVALID_ARG, "knn=%zu must be in [1, 2048]", knn);
    return SKL_OK;
}

extern "C" int skl_self_dists_knn_candidates(skl_ctx *ctx, const skl_sketches *s, const skl_dist_params *p,
                                             size_t knn, const uint64_t *row_offsets, const uint32_t *cand,
                                             uint64_t *out_idx, float *out_d0)
{
    SKL_TRY(check_candidate_call(ctx, s, p, knn));
    if (!row_offsets || !out_idx || !out_d0) return fail(SKL_ERR_INVALID_ARG, "null argument");
    const size_t n = s->n;
    if (n == 0) return SKL_OK;
    const uint64_t total = row_offsets[n];
    if (total && !cand) return fail(SKL_ERR_INVALID_ARG, "cand is null");
    for (size_t i = 0; i < n; ++i) {
        if (row_offsets[i + 1] < row_offsets[i]) return fail(SKL_ERR_INVALID_ARG, "row_offsets must not decrease");
    }
    for (uint64_t x = 0; x < total; ++x) {
        if (cand[x] >= n) return fail(SKL_ERR_INVALID_ARG, "candidate id %u out of range", cand[x]);
    }
    DevBuf d_off, d_cand;
    HIP_TRY(hipMalloc(&d_off.p, (n + 1) * sizeof(uint64_t)));
    HIP_TRY(hipMalloc(&d_cand.p, std::max<size_t>(total * sizeof(uint32_t), 16)));
    HIP_TRY(hipMemcpyAsync(d_off.p, row_offsets, (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    if (total) HIP_TRY(hipMemcpyAsync(d_cand.p, cand, total * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    return knn_from_device_csr(ctx, s, p, knn, row_offsets, (const uint64_t *)d_off.p, (const uint32_t *)d_cand.p,
                               out_idx, out_d0);
}

extern "C" size_t skl_shared_bins_max_samples(void) { return MAX_DEVICE_CANDGEN_SAMPLES; }

// The whole precluster kNN on the device: candidate lists from the index sketches (cand_gen.hip),
// then distances and ragged top-k.
extern "C" int skl_self_dists_knn_shared_bins(skl_ctx *ctx, const skl_sketches *s, const skl_dist_params *p,
                                              size_t knn, const uint16_t *skq, size_t sketch_size,
                                              uint64_t *out_idx, float *out_d0, uint64_t *out_n_candidates)
{
    SKL_TRY(check_candidate_call(ctx, s, p, knn));
    if (!skq || !out_idx || !out_d0) return fail(SKL_ERR_INVALID_ARG, "null argument");
    const size_t n = s->n;
    if (out_n_candidates) *out_n_candidates = 0;
    if (n == 0) return SKL_OK;
    if (sketch_size == 0) return fail(SKL_ERR_INVALID_ARG, "sketch_size is zero");
    if (n > MAX_DEVICE_CANDGEN_SAMPLES) {
        return fail(SKL_ERR_INVALID_ARG, "%zu samples exceed the %zu the on-device candidate search handles per call",
                    n, (size_t)MAX_DEVICE_CANDGEN_SAMPLES);
    }
    DevBuf d_skq, d_starts, d_cursor, d_members, d_counts, d_off, d_cand;
    const size_t table = sketch_size * 65536 * sizeof(uint32_t);
    HIP_TRY(hipMalloc(&d_skq.p, n * sketch_size * sizeof(uint16_t)));
    HIP_TRY(hipMalloc(&d_starts.p, table));
    HIP_TRY(hipMalloc(&d_cursor.p, table));
    HIP_TRY(hipMalloc(&d_members.p, n * sketch_size * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&d_counts.p, n * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&d_off.p, (n + 1) * sizeof(uint64_t)));
    HIP_TRY(hipMemcpyAsync(d_skq.p, skq, n * sketch_size * sizeof(uint16_t), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemsetAsync(d_starts.p, 0, table, ctx->stream));
    CandGenArgs cg;
    memset(&cg, 0, sizeof cg);
    cg.skq = (const uint16_t *)d_skq.p;
    cg.n = (uint32_t)n;
    cg.sketch_size = (uint32_t)sketch_size;
    cg.starts = (uint32_t *)d_starts.p;
    cg.cursor = (uint32_t *)d_cursor.p;
    cg.members = (uint32_t *)d_members.p;
    cg.counts = (uint32_t *)d_counts.p;
    HIP_TRY(launch_cand_groups(cg, ctx->stream));
    HIP_TRY(launch_cand_rows(cg, false, ctx->stream));
    std::vector<uint32_t> counts(n);
    HIP_TRY(hipMemcpyAsync(counts.data(), d_counts.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    std::vector<uint64_t> offsets(n + 1, 0);
    for (size_t i = 0; i < n; ++i) offsets[i + 1] = offsets[i] + counts[i];
    const uint64_t total = offsets[n];
    if (out_n_candidates) *out_n_candidates = total;
    HIP_TRY(hipMalloc(&d_cand.p, std::max<size_t>(total * sizeof(uint32_t), 16)));
    HIP_TRY(hipMemcpyAsync(d_off.p, offsets.data(), (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    cg.row_offsets = (const uint64_t *)d_off.p;
    cg.cand = (uint32_t *)d_cand.p;
    HIP_TRY(launch_cand_rows(cg, true, ctx->stream));
    // the group tables are no longer needed: release them before the distance buffers are allocated
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    (void)hipFree(d_members.p); d_members.p = nullptr;
    (void)hipFree(d_starts.p); d_starts.p = nullptr;
    (void)hipFree(d_cursor.p); d_cursor.p = nullptr;
    return knn_from_device_csr(ctx, s, p, knn, offsets.data(), (const uint64_t *)d_off.p, (const uint32_t *)d_cand.p,
                               out_idx, out_d0);
}

extern "C" size_t skl_knn_band_rows(const skl_sketches *s, const skl_dist_params *p, size_t n_participants)
{
    if (!s || !p || s->n == 0) return 0;
    const size_t rec = p->dist_type == SKL_DIST_COREACC ? 2 * sizeof(float) : sizeof(float);
    const long long forced = env_int("SKL_KNN_BAND_ROWS", 0);   // test knob (the same for every participant)
    if (forced > 0) return std::min<size_t>(s->n, (size_t)forced);
    // a fixed budget (no free-memory query): every participant must arrive at the same number
    return std::min(s->n, symmetric_band_rows(s->n, rec, 32ull << 30, n_participants));
}

extern "C" int skl_self_dists_knn_partial(skl_ctx *ctx, const skl_sketches *s, const skl_dist_params *p, size_t knn,
                                          size_t band_rows, const uint32_t *bands, size_t n_bands,
                                          uint32_t *state_key, uint32_t *state_idx, float *state_d1,
                                          int out_on_device)
{
    SKL_TRY(check_params(s, s, p));
    SKL_TRY(ctx_bind(ctx));
    const bool coreacc = p->dist_type == SKL_DIST_COREACC;
    if (!state_key || !state_idx || (coreacc && !state_d1)) return fail(SKL_ERR_INVALID_ARG, "state pointers are null");
    const size_t n = s->n;
    if (n < 2 || knn == 0 || knn > n - 1) return fail(SKL_ERR_INVALID_ARG, "knn=%zu must be in [1, %zu]", knn, n ? n - 1 : 0);
    if (knn > 2048) return fail(SKL_ERR_INVALID_ARG, "knn=%zu exceeds the device limit of 2048", knn);
    if (band_rows == 0) return fail(SKL_ERR_INVALID_ARG, "band_rows is zero");
    if (n_bands && !bands) return fail(SKL_ERR_INVALID_ARG, "bands is null");
    if (!knn_symmetric_ok(s, p)) {
        return fail(SKL_ERR_INVALID_ARG, "no one-evaluation kNN for this configuration; shard rows with skl_self_dists_knn_rows");
    }
    const size_t total_bands = (n + band_rows - 1) / band_rows;
    std::vector<uint32_t> list(bands, bands + n_bands);
    for (size_t x = 0; x < list.size(); ++x) {
        if (list[x] >= total_bands || (x && list[x] <= list[x - 1])) {
            return fail(SKL_ERR_INVALID_ARG, "bands must be ascending and below %zu", total_bands);
        }
    }
    KnnState st;
    SKL_TRY(knn_state_init(st, n, knn, coreacc, ctx->stream));
    const bool overlap = env_int("SKL_KNN_OVERLAP", 1) != 0 && list.size() > 1;
    SKL_TRY(knn_symmetric_bands(ctx, s, p, knn, band_rows, list, overlap, st));
    const hipMemcpyKind kind = out_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    const size_t items = n * knn;
    HIP_TRY(hipMemcpyAsync(state_key, st.key, items * sizeof(uint32_t), kind, ctx->stream));
    HIP_TRY(hipMemcpyAsync(state_idx, st.idx, items * sizeof(uint32_t), kind, ctx->stream));
    if (coreacc) HIP_TRY(hipMemcpyAsync(state_d1, st.d1, items * sizeof(float), kind, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));   // the running states are freed on return
    return SKL_OK;
}

extern "C" int skl_knn_merge_states(skl_ctx *ctx, size_t n_states, size_t rows, size_t knn,
                                    const uint32_t *state_key, const uint32_t *state_idx, const float *state_d1,
                                    int states_on_device, int ani, uint64_t *out_idx, float *out_d0, float *out_d1,
                                    int out_on_device)
{
    SKL_TRY(ctx_bind(ctx));
    if (!state_key || !state_idx || !out_idx || !out_d0) return fail(SKL_ERR_INVALID_ARG, "null argument");
    if (state_d1 && !out_d1) return fail(SKL_ERR_INVALID_ARG, "out_d1 is required with second values");
    if (n_states == 0 || knn == 0 || knn > 2048) return fail(SKL_ERR_INVALID_ARG, "n_states and knn (<= 2048) must be positive");
    if (rows == 0) return SKL_OK;
    const size_t items = rows * knn;
    struct DevBuf {
        void *p = nullptr;
        ~DevBuf() { if (p) (void)hipFree(p); }
    } in_key, in_idx, in_d1, tmp_key[2], tmp_idx[2], tmp_d1[2], o_idx, o_d0, o_d1;
    const uint32_t *d_key = state_key, *d_idx = state_idx;
    const float *d_d1 = state_d1;
    if (!states_on_device) {
        HIP_TRY(hipMalloc(&in_key.p, n_states * items * sizeof(uint32_t)));
        HIP_TRY(hipMalloc(&in_idx.p, n_states * items * sizeof(uint32_t)));
        HIP_TRY(hipMemcpyAsync(in_key.p, state_key, n_states * items * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipMemcpyAsync(in_idx.p, state_idx, n_states * items * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        d_key = (const uint32_t *)in_key.p;
        d_idx = (const uint32_t *)in_idx.p;
        if (state_d1) {
            HIP_TRY(hipMalloc(&in_d1.p, n_states * items * sizeof(float)));
            HIP_TRY(hipMemcpyAsync(in_d1.p, state_d1, n_states * items * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
            d_d1 = (const float *)in_d1.p;
        }
    }
    // fold the states, as many per launch as fit the LDS sort
    std::vector<const uint32_t *> keys, idxs;
    std::vector<const float *> d1s;
    for (size_t x = 0; x < n_states; ++x) {
        keys.push_back(d_key + x * items);
        idxs.push_back(d_idx + x * items);
        d1s.push_back(d_d1 ? d_d1 + x * items : nullptr);
    }
    const size_t group = std::max<size_t>(2, std::min<size_t>(MERGE_STATES_MAX, MERGE_STATES_ITEMS / knn));
    int flip = 0;
    for (;;) {   // (one state: a pass through the kernel is a copy)
        const size_t take = std::min(group, keys.size());
        if (!tmp_key[flip].p) {
            HIP_TRY(hipMalloc(&tmp_key[flip].p, items * sizeof(uint32_t)));
            HIP_TRY(hipMalloc(&tmp_idx[flip].p, items * sizeof(uint32_t)));
            if (d_d1) HIP_TRY(hipMalloc(&tmp_d1[flip].p, items * sizeof(float)));
        }
        MergeStatesArgs m;
        memset(&m, 0, sizeof m);
        for (size_t x = 0; x < take; ++x) {
            m.key[x] = keys[x];
            m.idx[x] = idxs[x];
            m.d1[x] = d1s[x];
        }
        m.n_in = (uint32_t)take;
        m.rows = (uint32_t)rows;
        m.knn = (uint32_t)knn;
        m.out_key = (uint32_t *)tmp_key[flip].p;
        m.out_idx = (uint32_t *)tmp_idx[flip].p;
        m.out_d1 = d_d1 ? (float *)tmp_d1[flip].p : nullptr;
        HIP_TRY(launch_merge_states(m, ctx->stream));
        keys.erase(keys.begin(), keys.begin() + take);
        idxs.erase(idxs.begin(), idxs.begin() + take);
        d1s.erase(d1s.begin(), d1s.begin() + take);
        keys.insert(keys.begin(), m.out_key);
        idxs.insert(idxs.begin(), m.out_idx);
        d1s.insert(d1s.begin(), m.out_d1);
        flip ^= 1;
        if (keys.size() == 1) break;
    }
    uint64_t *r_idx = out_idx;
    float *r_d0 = out_d0, *r_d1 = out_d1;
    if (!out_on_device) {
        HIP_TRY(hipMalloc(&o_idx.p, items * sizeof(uint64_t)));
        HIP_TRY(hipMalloc(&o_d0.p, items * sizeof(float)));
        r_idx = (uint64_t *)o_idx.p;
        r_d0 = (float *)o_d0.p;
        if (d_d1) {
            HIP_TRY(hipMalloc(&o_d1.p, items * sizeof(float)));
            r_d1 = (float *)o_d1.p;
        }
    }
    HIP_TRY(launch_topk_finalize(keys[0], idxs[0], d1s[0], items, (!d_d1 && ani) ? 1 : 0, r_idx, r_d0, r_d1, ctx->stream));
    if (!out_on_device) {
        HIP_TRY(hipMemcpyAsync(out_idx, r_idx, items * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipMemcpyAsync(out_d0, r_d0, items * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
        if (d_d1) HIP_TRY(hipMemcpyAsync(out_d1, r_d1, items * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));   // temporaries are freed on return
    return SKL_OK;
}

extern "C" int skl_cross_dists_knn_rows(skl_ctx *ctx, const skl_sketches *ref,
                                        const skl_sketches *query, const skl_dist_params *p,
                                        size_t knn, size_t query_begin, size_t query_end,
                                        uint64_t *out_idx, float *out_d0, float *out_d1,
                                        int out_on_device)
{
    SKL_TRY(check_params(ref, query, p));
    if (ref->n == 0) return fail(SKL_ERR_EMPTY_DB, "Reference database has no loaded samples");
    if (query->n == 0) return fail(SKL_ERR_EMPTY_DB, "Query database has no loaded samples");
    // rows = queries (scalar operand), candidates = refs (lane operand); samebits and the
    // completeness factor are symmetric in the pair, so core_acc_dist(ref, query, ri, qi)
    // (mod.rs:377-385) is computed with the roles swapped.
    return knn_rows(ctx, query, ref, p, knn, 0, query_begin, query_end, out_idx, out_d0, out_d1,
                    out_on_device);
}

extern "C" int skl_cross_dists_knn(skl_ctx *ctx, const skl_sketches *ref,
                                   const skl_sketches *query, const skl_dist_params *p, size_t knn,
                                   uint64_t *out_idx, float *out_d0, float *out_d1,
                                   int out_on_device)
{
    if (!ref || !query) return fail(SKL_ERR_INVALID_ARG, "null sketches");
    return skl_cross_dists_knn_rows(ctx, ref, query, p, knn, 0, query->n, out_idx, out_d0, out_d1,
                                    out_on_device);
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
