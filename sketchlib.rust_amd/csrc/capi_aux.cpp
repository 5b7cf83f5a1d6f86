// capi_aux.cpp -- the entry points of include/sketchlib_dist.h either side of the distance
// path: GPU sketching and the candidate-list kNN of the precluster mode
// (row f2).  Kernels: sketch_kernel.hip, cand_gen.hip, pair_cand.hip, kernels.hip (topk_kernel).
#include "capi_internal.hpp"

#include <algorithm>
#include <atomic>
#include <cstring>
#include <thread>

using namespace skl;

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

// Sixteen one-byte codes -> one word of 2-bit codes (code c at bits 2 c).
static inline uint32_t pack16(const uint8_t *src)
{
    uint64_t lo, hi;
    memcpy(&lo, src, 8);
    memcpy(&hi, src + 8, 8);
    auto squeeze = [](uint64_t v) -> uint32_t {   // 8 bytes of 2 significant bits -> 16 bits
        v &= 0x0303030303030303ull;
        v = (v | (v >> 6)) & 0x000F000F000F000Full;
        v = (v | (v >> 12)) & 0x000000FF000000FFull;
        v = (v | (v >> 24)) & 0xFFFFull;
        return (uint32_t)v;
    };
    return squeeze(lo) | (squeeze(hi) << 16);
}

// The bases of samples [s0, s1) packed into dst (the words of sample s start at dst + (word_begin[s] - word_begin[s0])).
static void pack_samples(const uint8_t *codes, const uint64_t *code_begin, const std::vector<uint64_t> &word_begin, size_t s0, size_t s1,
                         uint32_t *dst, unsigned threads)
{
    // work items: (sample, word range) pieces of at most 1 Mi words
    struct Piece { size_t s; uint64_t w0, w1; };
    std::vector<Piece> pieces;
    for (size_t s = s0; s < s1; ++s) {
        const uint64_t words = word_begin[s + 1] - word_begin[s];
        for (uint64_t w = 0; w < words; w += (1ull << 20)) pieces.push_back({s, w, std::min<uint64_t>(words, w + (1ull << 20))});
    }
    std::atomic<size_t> next{0};
    auto work = [&] {
        for (;;) {
            const size_t x = next.fetch_add(1);
            if (x >= pieces.size()) break;
            const Piece &pc = pieces[x];
            const uint8_t *src = codes + code_begin[pc.s];
            const uint64_t n_codes = code_begin[pc.s + 1] - code_begin[pc.s];
            uint32_t *out = dst + (word_begin[pc.s] - word_begin[s0]);
            for (uint64_t w = pc.w0; w < pc.w1; ++w) {
                if (16 * w + 16 <= n_codes) {
                    out[w] = pack16(src + 16 * w);
                } else {   // the sample's last word: zero-padded
                    uint32_t word = 0;
                    for (uint64_t c = 16 * w; c < n_codes; ++c) word |= (uint32_t)(src[c] & 3u) << (2u * (uint32_t)(c - 16 * w));
                    out[w] = word;
                }
            }
        }
    };
    const unsigned t = (unsigned)std::max<size_t>(1, std::min<size_t>(threads, pieces.size()));
    std::vector<std::thread> pool;
    for (unsigned i = 1; i < t; ++i) pool.emplace_back(work);
    work();
    for (auto &th : pool) th.join();
}

// skl_sketch_signs / skl_sketch_signs_packed.  Round 5: the bases cross PCIe at 2 bits each; the device buffers belong to the
// context (grow-only) instead of being allocated per call; the samples go in batches of ~8 Mi words, batch i + 1's upload
// (the auxiliary stream) under batch i's kernel, batch i's signs on their way back under batch i + 1's kernel.  One-byte codes
// are packed on host threads straight into a pinned two-batch ring, so their upload is a true DMA.
static int sketch_signs_impl(skl_ctx *ctx, const uint8_t *codes, const uint32_t *packed, const uint64_t *code_begin,
                             const uint64_t *offsets, const uint64_t *offset_begin, size_t n_samples,
                             const size_t *kmers, size_t nk, uint64_t num_bins, int rc, uint64_t *out_signs)
{
    SKL_TRY(ctx_bind(ctx));
    if (!code_begin || !offset_begin || !kmers || !out_signs) return fail(SKL_ERR_INVALID_ARG, "null argument");
    if (n_samples == 0 || nk == 0) return SKL_OK;
    if (num_bins == 0) return fail(SKL_ERR_INVALID_ARG, "num_bins is zero");
    const uint64_t n_codes = code_begin[n_samples], n_offs = offset_begin[n_samples];
    if ((n_codes && !codes && !packed) || (n_offs && !offsets)) return fail(SKL_ERR_INVALID_ARG, "null argument");
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
    // the LDS-staged kernel takes every k-mer length up to its span + 1 (SKL_SKETCH_KERNEL=global: A/B)
    size_t kmax = 0;
    for (size_t ki = 0; ki < nk; ++ki) kmax = std::max(kmax, kmers[ki]);
    const bool lds_form = kmax <= (size_t)sketch_span_lds() + 1 && !ctx->knobs.sketch_global;
    const uint64_t span = (uint64_t)(lds_form ? sketch_span_lds() : sketch_span());
    const uint64_t wg = lds_form ? (uint64_t)sketch_wg_lds() : 256u;
    std::vector<uint64_t> span_begin(n_samples + 1, 0), word_begin(n_samples + 1, 0);
    for (size_t s = 0; s < n_samples; ++s) {
        if (code_begin[s + 1] < code_begin[s] || offset_begin[s + 1] < offset_begin[s]) {
            return fail(SKL_ERR_INVALID_ARG, "sample ranges must not decrease");
        }
        const uint64_t len = code_begin[s + 1] - code_begin[s];
        uint64_t spans = (len + span - 1) / span;
        spans = (spans + wg - 1) / wg * wg;   // whole workgroups per sample (the staged kernel needs it; batches of samples then
                                              // start on workgroup boundaries in either form)
        span_begin[s + 1] = span_begin[s] + spans;
        word_begin[s + 1] = word_begin[s] + (len + 15) / 16;
    }
    const uint64_t total_words = word_begin[n_samples];
    const size_t sign_words = n_samples * nk * num_bins;
    // device buffers of the context: packed codes | signs | the small arrays
    void *d_packed = nullptr, *d_signs = nullptr, *d_small = nullptr;
    SKL_TRY(ctx_scratch(ctx, std::max<uint64_t>(total_words, 4) * sizeof(uint32_t), &d_packed, 12));
    SKL_TRY(ctx_scratch(ctx, sign_words * sizeof(uint64_t), &d_signs, 13));
    const size_t small_words = 4 * (n_samples + 1) + n_offs + 9 * nk + 8;   // u64 each (the k-mer lengths: two per word)
    SKL_TRY(ctx_scratch(ctx, small_words * sizeof(uint64_t), &d_small, 14));
    std::vector<uint64_t> small;
    small.reserve(small_words);
    auto put = [&](const uint64_t *v, size_t count) {
        const size_t at = small.size();
        small.insert(small.end(), v, v + count);
        return at;
    };
    const size_t at_cb = put(code_begin, n_samples + 1), at_ob = put(offset_begin, n_samples + 1), at_sb = put(span_begin.data(), n_samples + 1);
    const size_t at_wb = put(word_begin.data(), n_samples), at_offs = put(offsets, n_offs);
    const size_t at_tf = put(top_f.data(), top_f.size()), at_tr = put(top_r.data(), top_r.size());
    const size_t at_k = small.size();
    small.resize(at_k + (nk + 1) / 2, 0);
    memcpy(small.data() + at_k, k32.data(), nk * sizeof(uint32_t));
    if (small.size() > small_words) return fail(SKL_ERR_INVALID_ARG, "internal: small-array layout");
    HIP_TRY(hipMemcpyAsync(d_small, small.data(), small.size() * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemsetAsync(d_signs, 0xFF, sign_words * sizeof(uint64_t), ctx->stream));   // u64::MAX
    HIP_TRY(hipStreamSynchronize(ctx->stream));   // (`small` is pageable: uploaded before it dies; the aux stream starts behind this)

    SketchArgs a;
    memset(&a, 0, sizeof a);
    const uint64_t *ds = (const uint64_t *)d_small;
    a.packed = (const uint32_t *)d_packed;
    a.word_begin = ds + at_wb;
    a.code_begin = ds + at_cb;
    a.offsets = ds + at_offs;
    a.offset_begin = ds + at_ob;
    a.span_begin = ds + at_sb;
    a.n_samples = (uint32_t)n_samples;
    a.nk = (uint32_t)nk;
    a.kmers = (const uint32_t *)(ds + at_k);
    a.top_f = ds + at_tf;
    a.top_r = ds + at_tr;
    a.num_bins = num_bins;
    const uint64_t sign_mod = (1ull << 61) - 1;
    a.bin_size = (sign_mod + num_bins - 1) / num_bins;   // SIGN_MOD.div_ceil(num_bins), sketch/mod.rs:170
    a.inv_bin_size = 1.0 / (double)a.bin_size;
    a.rc = rc ? 1 : 0;
    a.signs = (uint64_t *)d_signs;
    a.lds_form = lds_form ? 1u : 0u;

    // batches of whole samples, ~8 Mi words (128 Mi bases) each
    constexpr uint64_t BATCH_WORDS = 8ull << 20;
    std::vector<size_t> cuts{0};
    for (size_t s = 0; s < n_samples; ++s) {
        if (word_begin[s + 1] - word_begin[cuts.back()] >= BATCH_WORDS && s + 1 < n_samples) cuts.push_back(s + 1);
    }
    cuts.push_back(n_samples);
    const size_t n_batches = cuts.size() - 1;
    // one-byte codes: packed on host threads into a pinned ring of two batches (grow-only, kept by the context)
    uint64_t ring_words = 0;
    if (!packed) {
        for (size_t b = 0; b < n_batches; ++b) ring_words = std::max(ring_words, word_begin[cuts[b + 1]] - word_begin[cuts[b]]);
        if (ctx->pinned_words < 2 * ring_words) {
            if (ctx->pinned) HIP_TRY(hipHostFree(ctx->pinned));
            ctx->pinned = nullptr;
            ctx->pinned_words = 0;
            HIP_TRY(hipHostMalloc((void **)&ctx->pinned, std::max<uint64_t>(2 * ring_words, 4) * sizeof(uint32_t), hipHostMallocDefault));
            ctx->pinned_words = 2 * ring_words;
        }
    }
    std::vector<hipEvent_t> ev(3 * n_batches, nullptr);   // per batch: uploaded, hashed, (ring slot) free again
    struct EventGuard {
        std::vector<hipEvent_t> &e;
        ~EventGuard() { for (hipEvent_t x : e) if (x) (void)hipEventDestroy(x); }
    } guard{ev};
    for (auto &e : ev) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    const unsigned pack_threads = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    auto upload = [&](size_t b) -> int {
        const RoctxRange range_("skl:sketch_batch pack + upload");
        const size_t s0 = cuts[b], s1 = cuts[b + 1];
        const uint64_t w0 = word_begin[s0], words = word_begin[s1] - w0;
        if (words == 0) {
            HIP_TRY(hipEventRecord(ev[3 * b], ctx->aux_stream));
            return SKL_OK;
        }
        const uint32_t *src;
        if (packed) {
            src = packed + w0;
        } else {
            uint32_t *slot = ctx->pinned + (b & 1) * ring_words;
            if (b >= 2) HIP_TRY(hipEventSynchronize(ev[3 * (b - 2) + 2]));   // that slot's previous upload has left it
            pack_samples(codes, code_begin, word_begin, s0, s1, slot, pack_threads);
            src = slot;
        }
        HIP_TRY(hipMemcpyAsync((uint32_t *)d_packed + w0, src, words * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->aux_stream));
        HIP_TRY(hipEventRecord(ev[3 * b], ctx->aux_stream));
        HIP_TRY(hipEventRecord(ev[3 * b + 2], ctx->aux_stream));
        return SKL_OK;
    };
    auto launch = [&](size_t b) -> int {
        const RoctxRange range_("skl:sketch_batch hash + bin minima");
        HIP_TRY(hipStreamWaitEvent(ctx->stream, ev[3 * b], 0));
        a.first_span = span_begin[cuts[b]];
        a.n_spans = span_begin[cuts[b + 1]] - a.first_span;
        std::pair<hipEvent_t, hipEvent_t> *tev = timing_slot(ctx);   // (bracketed like the pair kernels, batch by batch)
        if (tev) HIP_TRY(hipEventRecord(tev->first, ctx->stream));
        HIP_TRY(launch_sketch_signs(a, ctx->stream));
        if (tev) HIP_TRY(hipEventRecord(tev->second, ctx->stream));
        HIP_TRY(hipEventRecord(ev[3 * b + 1], ctx->stream));
        return SKL_OK;
    };
    auto download = [&](size_t b) -> int {   // (a copy into pageable memory blocks its caller: issued after the next batch's launch)
        const size_t s0 = cuts[b], s1 = cuts[b + 1];
        HIP_TRY(hipStreamWaitEvent(ctx->aux_stream, ev[3 * b + 1], 0));
        HIP_TRY(hipMemcpyAsync(out_signs + s0 * nk * num_bins, (const uint64_t *)d_signs + s0 * nk * num_bins,
                               (s1 - s0) * nk * num_bins * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->aux_stream));
        return SKL_OK;
    };
    SKL_TRY(upload(0));
    SKL_TRY(launch(0));
    for (size_t b = 0; b < n_batches; ++b) {
        if (b + 1 < n_batches) {
            SKL_TRY(upload(b + 1));
            SKL_TRY(launch(b + 1));
        }
        SKL_TRY(download(b));
    }
    ctx->last_kernel = lds_form ? "skl::nthash_binmin_lds_kernel (bases staged in LDS as 2-bit codes, 128 window starts per thread, rolling "
                                  "canonical ntHash through a fused 16-entry step table, bin minima in LDS)"
                                : "skl::nthash_binmin_kernel (256 window starts per thread, rolling canonical ntHash, atomicMin per bin)";
    HIP_TRY(hipStreamSynchronize(ctx->aux_stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    // The sketching buffers belong to the context so that a run of sketch calls does not allocate per call -- but beyond 1 GiB
    // they would sit in the way of what follows (the kNN drivers size their bands by the free memory): released, with the
    // pinned ring.  (Both streams are idle here.)
    if (ctx->scratch_bytes[12] + ctx->scratch_bytes[13] > (1ull << 30)) {
        for (int slot : {12, 13}) {
            if (ctx->scratch[slot]) HIP_TRY(hipFree(ctx->scratch[slot]));
            ctx->scratch[slot] = nullptr;
            ctx->scratch_bytes[slot] = 0;
        }
        if (ctx->pinned) {
            HIP_TRY(hipHostFree(ctx->pinned));
            ctx->pinned = nullptr;
            ctx->pinned_words = 0;
        }
    }
    return SKL_OK;
}

extern "C" int skl_sketch_signs(skl_ctx *ctx, const uint8_t *codes, const uint64_t *code_begin,
                                const uint64_t *offsets, const uint64_t *offset_begin, size_t n_samples,
                                const size_t *kmers, size_t nk, uint64_t num_bins, int rc, uint64_t *out_signs)
{
    return sketch_signs_impl(ctx, codes, nullptr, code_begin, offsets, offset_begin, n_samples, kmers, nk, num_bins, rc, out_signs);
}

extern "C" int skl_sketch_signs_packed(skl_ctx *ctx, const uint32_t *packed, const uint64_t *code_begin,
                                       const uint64_t *offsets, const uint64_t *offset_begin, size_t n_samples,
                                       const size_t *kmers, size_t nk, uint64_t num_bins, int rc, uint64_t *out_signs)
{
    if (!packed && code_begin && n_samples && code_begin[n_samples]) return fail(SKL_ERR_INVALID_ARG, "null argument");
    return sketch_signs_impl(ctx, nullptr, packed, code_begin, offsets, offset_begin, n_samples, kmers, nk, num_bins, rc, out_signs);
}

// The last step of a candidate-list call: the per-row selection of rows [r, r + rows) -- `launch(first_row, rows)` --
// and the results' way back to the host.  A copy into pageable memory blocks its caller, so large results go in four row
// batches, batch b's copy issued on the auxiliary stream behind batch b + 1's selection.
template <class Launch>
static int select_and_copy_back(skl_ctx *ctx, size_t n, size_t knn, const Launch &launch, const void *d_idx, const void *d_d0,
                                uint64_t *out_idx, float *out_d0)
{
    const size_t n_batches = n * knn * (sizeof(uint64_t) + sizeof(float)) >= (32u << 20) ? 4 : 1;
    size_t prev0 = 0, prev_rows = 0;
    auto copy_back = [&](size_t r0, size_t rows, int ev) -> int {
        HIP_TRY(hipStreamWaitEvent(ctx->aux_stream, ctx->knn_pair_done[ev], 0));
        HIP_TRY(hipMemcpyAsync(out_idx + r0 * knn, (const uint64_t *)d_idx + r0 * knn, rows * knn * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->aux_stream));
        HIP_TRY(hipMemcpyAsync(out_d0 + r0 * knn, (const float *)d_d0 + r0 * knn, rows * knn * sizeof(float), hipMemcpyDeviceToHost, ctx->aux_stream));
        return SKL_OK;
    };
    for (size_t b = 0; b < n_batches; ++b) {
        const size_t r0 = n * b / n_batches, r1 = n * (b + 1) / n_batches;
        if (r1 == r0) continue;
        SKL_TRY(launch(r0, r1 - r0));
        HIP_TRY(hipEventRecord(ctx->knn_pair_done[b & 1], ctx->stream));
        if (prev_rows) SKL_TRY(copy_back(prev0, prev_rows, (int)((b - 1) & 1)));
        prev0 = r0;
        prev_rows = r1 - r0;
    }
    if (prev_rows) SKL_TRY(copy_back(prev0, prev_rows, (int)((n_batches - 1) & 1)));
    HIP_TRY(hipStreamSynchronize(ctx->aux_stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SKL_OK;
}

// Candidate-list kNN: the device half of the reference's self_dists_knn_precluster
// (src/distances/mod.rs:399-553).  Host pointers in, host pointers out.
// Distances + ragged top-k for candidate lists that are already on the device.  host_offsets is
// the host copy of the CSR offsets (the 64-candidate work items are cut on the host).
static int knn_from_device_csr(skl_ctx *ctx, const skl_sketches *s, const skl_dist_params *p, size_t knn,
                               const uint64_t *host_offsets, const uint64_t *d_off, const uint32_t *d_cand,
                               bool symmetric_lists, uint64_t *out_idx, float *out_d0)
{
    const size_t n = s->n;
    const uint64_t total = host_offsets[n];
    const bool symmetric = symmetric_lists && ctx->knobs.cand_symmetric;
    // symmetric lists: only the candidates with a larger id than the row are evaluated (the
    // kernel stores each key for both rows), so a row's work items start at its first such candidate
    std::vector<uint64_t> first;
    DevBuf d_first;
    if (symmetric && n) {
        HIP_TRY(hipMalloc(&d_first.p, n * sizeof(uint64_t)));
        HIP_TRY(launch_first_greater(d_off, d_cand, (uint32_t)n, (uint64_t *)d_first.p, ctx->stream));
        first.resize(n);
        HIP_TRY(hipMemcpyAsync(first.data(), d_first.p, n * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    // Launch order of the rows (round 4).  pair_cand_kernel is bound by the gather of the candidates' sketches, and the rows
    // of one cluster gather nearly the SAME candidates: dispatched next to each other they find them in the L2 of their XCD;
    // in sample order -- cluster members scattered over the database -- every pair goes to HBM.  Rows are therefore taken
    // in the order of their first candidate (the lowest member of the cluster, as a rule): a counting sort, no list is
    // looked at.  (SKL_CAND_ROW_ORDER=0: sample order, A/B.)
    std::vector<uint32_t> order(n);
    for (size_t i = 0; i < n; ++i) order[i] = (uint32_t)i;
    if (ctx->knobs.cand_row_order && n > 1) {
        DevBuf d_key;
        HIP_TRY(hipMalloc(&d_key.p, n * sizeof(uint32_t)));
        HIP_TRY(launch_first_candidate(d_off, d_cand, (uint32_t)n, (uint32_t *)d_key.p, ctx->stream));
        std::vector<uint32_t> key(n);
        HIP_TRY(hipMemcpyAsync(key.data(), d_key.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        std::vector<uint32_t> begin(n + 2, 0);   // bucket n: empty rows
        for (size_t i = 0; i < n; ++i) ++begin[(key[i] < n ? key[i] : n) + 1];
        for (size_t b = 0; b <= n; ++b) begin[b + 1] += begin[b];
        for (size_t i = 0; i < n; ++i) order[begin[key[i] < n ? key[i] : n]++] = (uint32_t)i;
    }
    // The work items (a row and up to 64 of its candidates each; millions at a few hundred thousand rows): the host only
    // counts them per row, in launch order; a kernel writes them out.
    std::vector<uint64_t> item_begin(n + 1, 0);
    for (size_t x = 0; x < n; ++x) {
        const size_t i = order[x];
        item_begin[x + 1] = item_begin[x] + (host_offsets[i + 1] - (symmetric ? first[i] : host_offsets[i]) + 63) / 64;
    }
    const uint64_t n_items = item_begin[n];
    DevBuf d_wrow, d_wstart, d_order, d_ibegin, d_keys, d_idx, d_d0;
    HIP_TRY(hipMalloc(&d_wrow.p, std::max<size_t>(n_items * sizeof(uint32_t), 16)));
    HIP_TRY(hipMalloc(&d_wstart.p, std::max<size_t>(n_items * sizeof(uint64_t), 16)));
    HIP_TRY(hipMalloc(&d_order.p, std::max<size_t>(n * sizeof(uint32_t), 16)));
    HIP_TRY(hipMalloc(&d_ibegin.p, (n + 1) * sizeof(uint64_t)));
    if (n) HIP_TRY(hipMemcpyAsync(d_order.p, order.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(d_ibegin.p, item_begin.data(), (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(launch_cand_work_items((const uint32_t *)d_order.p, (const uint64_t *)d_ibegin.p, d_off, (const uint64_t *)d_first.p,
                                   (uint32_t)n, (uint32_t *)d_wrow.p, (uint64_t *)d_wstart.p, ctx->stream));
    HIP_TRY(hipMalloc(&d_keys.p, std::max<size_t>(total * sizeof(float), 16)));
    HIP_TRY(hipMalloc(&d_idx.p, n * knn * sizeof(uint64_t)));
    HIP_TRY(hipMalloc(&d_d0.p, n * knn * sizeof(float)));
    HIP_TRY(hipStreamSynchronize(ctx->stream));   // pageable uploads done before the vectors die

    PairArgs g;
    SKL_TRY(fill_args(s, s, p, MODE_JACCARD, p->ani ? JOUT_ANI_KEY : JOUT_DIST, &g));
    g.xcd_shift = ctx->knobs.cand_row_order ? ctx_xcd_shift(ctx) : 0u;
    CandArgs c;
    memset(&c, 0, sizeof c);
    c.row_offsets = d_off;
    c.cand = d_cand;
    c.work_row = (const uint32_t *)d_wrow.p;
    c.work_start = (const uint64_t *)d_wstart.p;
    c.n_work = n_items;
    c.keys = (float *)d_keys.p;
    c.symmetric = symmetric ? 1u : 0u;
    c.lanes_over_candidates = ctx->knobs.cand_lanes ? 1u : 0u;
    {   // bracketed like the pair kernels, so skl_ctx_kernel_ms() reports it
        std::pair<hipEvent_t, hipEvent_t> *ev = timing_slot(ctx);
        if (ev) HIP_TRY(hipEventRecord(ev->first, ctx->stream));
        HIP_TRY(launch_pair_cand(c, g, ctx->stream));
        if (ev) HIP_TRY(hipEventRecord(ev->second, ctx->stream));
    }
    ctx->last_kernel = "skl::pair_cand_rows_kernel (row x 64 candidates per wave, one after the other, each read as one contiguous run)";
#ifdef SKL_AB
    if (ctx->knobs.cand_lanes) ctx->last_kernel = "skl::pair_cand_kernel (row x 64 candidates per wave, candidate gather from the reference layout)";
#endif
    if (ctx->knn_ties == SKL_KNN_TIES_REFERENCE) {
        // The reference's tie order: its BinaryHeap replayed over each row's candidates IN THE ORDER THEY ARE LISTED
        // (mod.rs:459-487 pushes in the order Inverted::any_shared_bins returns them: ascending .ski index -- the
        // caller lists them that way; lists built on the device are ascending in the slab's sample order).
        RefHeapArgs h;
        memset(&h, 0, sizeof h);
        h.keys = (const float *)d_keys.p;
        h.stride2 = 1;
        h.rows = (uint32_t)n;
        h.self_id_base = 0xFFFFFFFFu;
        h.knn = (uint32_t)knn;
        h.ani_undo = p->ani ? 1 : 0;
        h.out_idx = (uint64_t *)d_idx.p;
        h.out_d0 = (float *)d_d0.p;
        h.row_offsets = d_off;
        h.col_ids = d_cand;
        h.force_workgroup_form = ctx->knobs.refheap_wave ? 0u : 1u;
        if (knn <= (size_t)REFHEAP_LDS_MAX) {
            return select_and_copy_back(
                ctx, n, knn,
                [&](size_t r0, size_t rows) -> int {
                    h.first_row = (uint32_t)r0;
                    h.rows = (uint32_t)rows;
                    HIP_TRY(launch_topk_refheap(h, ctx->stream));
                    return SKL_OK;
                },
                d_idx.p, d_d0.p, out_idx, out_d0);
        } else {   // heaps in global memory, rows in batches of at most 1 GiB of it
            DevBuf heaps;
            const size_t per_row = 3 * (knn + 1) * sizeof(float);
            const size_t batch = std::max<size_t>(1, std::min<size_t>(n, (1ull << 30) / per_row));
            HIP_TRY(hipMalloc(&heaps.p, batch * per_row));
            h.heap_scratch = (float *)heaps.p;
            for (size_t r = 0; r < n; r += batch) {
                h.first_row = (uint32_t)r;
                h.rows = (uint32_t)std::min(batch, n - r);
                HIP_TRY(launch_topk_refheap(h, ctx->stream));
            }
            HIP_TRY(hipStreamSynchronize(ctx->stream));   // `heaps` is freed on scope exit
        }
        HIP_TRY(hipMemcpyAsync(out_idx, d_idx.p, n * knn * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipMemcpyAsync(out_d0, d_d0.p, n * knn * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        return SKL_OK;
    }
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
    if (knn <= (size_t)TOPK_LDS_MAX) {
        return select_and_copy_back(
            ctx, n, knn,
            [&](size_t r0, size_t rows) -> int {
                t.first_row = (uint32_t)r0;
                t.rows = (uint32_t)rows;
                HIP_TRY(launch_topk(t, ctx->stream));
                return SKL_OK;
            },
            d_idx.p, d_d0.p, out_idx, out_d0);
    } else {
        // more neighbours than the LDS array holds: the selected items of a row are collected and sorted in
        // global memory, rows in batches of at most 1 GiB of it
        DevBuf items;
        t.items_pitch = topk_items_pitch(knn);
        const size_t batch = std::max<size_t>(1, std::min<size_t>(n, (1ull << 30) / (t.items_pitch * sizeof(uint64_t))));
        HIP_TRY(hipMalloc(&items.p, batch * t.items_pitch * sizeof(uint64_t)));
        t.items_scratch = (uint64_t *)items.p;
        for (size_t r = 0; r < n; r += batch) {
            t.first_row = (uint32_t)r;
            t.rows = (uint32_t)std::min(batch, n - r);
            HIP_TRY(launch_topk(t, ctx->stream));
        }
        HIP_TRY(hipStreamSynchronize(ctx->stream));   // `items` is freed on scope exit
    }
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
    if (knn == 0) return fail(SKL_ERR_INVALID_ARG, "knn must be positive");
    return SKL_OK;
}

extern "C" int skl_self_dists_knn_candidates(skl_ctx *ctx, const skl_sketches *s, const skl_dist_params *p,
                                             size_t knn, const uint64_t *row_offsets, const uint32_t *cand,
                                             uint64_t *out_idx, float *out_d0)
{
    const RoctxRange range_("skl:candidate-list kNN (precluster)");
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
    // caller-supplied lists: no symmetry assumed, every listed pair is evaluated
    return knn_from_device_csr(ctx, s, p, knn, row_offsets, (const uint64_t *)d_off.p, (const uint32_t *)d_cand.p,
                               false, out_idx, out_d0);
}

extern "C" size_t skl_shared_bins_max_samples(void) { return MAX_DEVICE_CANDGEN_SAMPLES; }

// The whole precluster kNN on the device: candidate lists from the index sketches (cand_gen.hip),
// then distances and ragged top-k.
extern "C" int skl_self_dists_knn_shared_bins(skl_ctx *ctx, const skl_sketches *s, const skl_dist_params *p,
                                              size_t knn, const uint16_t *skq, size_t sketch_size,
                                              uint64_t *out_idx, float *out_d0, uint64_t *out_n_candidates)
{
    const RoctxRange range_("skl:shared-bins candidates + kNN (precluster)");
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
    HIP_TRY(hipMalloc(&d_members.p, n * sketch_size * sizeof(uint32_t) + 16));   // (+16: cand_rows_kernel reads whole 16-byte blocks)
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
    // "any shared bin" is a symmetric relation: each candidate pair is evaluated once
    return knn_from_device_csr(ctx, s, p, knn, offsets.data(), (const uint64_t *)d_off.p, (const uint32_t *)d_cand.p,
                               true, out_idx, out_d0);
}


// ---------------------------------------------------------------------------
// RCCL gather of row bands computed on several devices of one process ("grouped ncclSend/ncclRecv to root
// for unequal counts").  The reference has no counterpart (one address space); BASELINE.json's north star names it: "a RCCL
// gather over xGMI to assemble the output matrix".  The library is looked up at run time (librccl.so.1: part of ROCm, but the
// product library must load without it); the communicators of a device list are made once per process and kept.
// ---------------------------------------------------------------------------
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>

namespace {
struct RcclApi {
    ncclResult_t (*comm_init_all)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*comm_destroy)(ncclComm_t) = nullptr;
    ncclResult_t (*group_start)() = nullptr;
    ncclResult_t (*group_end)() = nullptr;
    ncclResult_t (*send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*error_string)(ncclResult_t) = nullptr;
    bool ok = false;
    RcclApi()
    {
        for (const char *lib : {"librccl.so.1", "librccl.so"}) {
            void *h = dlopen(lib, RTLD_LAZY | RTLD_LOCAL);
            if (!h) continue;
            comm_init_all = reinterpret_cast<decltype(comm_init_all)>(dlsym(h, "ncclCommInitAll"));
            comm_destroy = reinterpret_cast<decltype(comm_destroy)>(dlsym(h, "ncclCommDestroy"));
            group_start = reinterpret_cast<decltype(group_start)>(dlsym(h, "ncclGroupStart"));
            group_end = reinterpret_cast<decltype(group_end)>(dlsym(h, "ncclGroupEnd"));
            send = reinterpret_cast<decltype(send)>(dlsym(h, "ncclSend"));
            recv = reinterpret_cast<decltype(recv)>(dlsym(h, "ncclRecv"));
            error_string = reinterpret_cast<decltype(error_string)>(dlsym(h, "ncclGetErrorString"));
            ok = comm_init_all && comm_destroy && group_start && group_end && send && recv && error_string;
            if (ok) return;
        }
    }
    static const RcclApi &get()
    {
        static const RcclApi api;
        return api;
    }
};

std::mutex g_rccl_mutex;
std::map<std::vector<int>, std::vector<ncclComm_t>> g_rccl_comms;   // device list -> one communicator per entry (kept for the process)
}  // namespace

#define RCCL_TRY(expr)                                                                          \
    do {                                                                                        \
        const ncclResult_t r_ = (expr);                                                         \
        if (r_ != ncclSuccess) return fail(SKL_ERR_HIP, "%s: %s", #expr, api.error_string(r_)); \
    } while (0)

extern "C" int skl_gather_bands_rccl(skl_ctx *const *ctxs, size_t n_ctx, const void *const *band_dev, const size_t *band_bytes,
                                     void *dst_dev_root, const size_t *dst_offsets, int loopback_through_rccl)
{
    if (!ctxs || n_ctx == 0 || !band_dev || !band_bytes || !dst_dev_root || !dst_offsets) return fail(SKL_ERR_INVALID_ARG, "null argument");
    if (n_ctx > 64) return fail(SKL_ERR_INVALID_ARG, "at most 64 devices");
    std::vector<int> devs(n_ctx);
    for (size_t d = 0; d < n_ctx; ++d) {
        if (!ctxs[d]) return fail(SKL_ERR_INVALID_ARG, "null context");
        devs[d] = ctxs[d]->device;
        for (size_t e = 0; e < d; ++e) {
            if (devs[e] == devs[d]) return fail(SKL_ERR_INVALID_ARG, "skl_gather_bands_rccl: device %d listed twice (RCCL takes one rank per device)", devs[d]);
        }
    }
    skl_ctx *root = ctxs[0];
    SKL_TRY(ctx_bind(root));
    // the root's own band: a copy inside its HBM (or, on request, through RCCL to itself: the one-device test of the transport)
    const bool use_rccl = n_ctx > 1 || loopback_through_rccl;
    if (!use_rccl) {
        if (band_bytes[0]) HIP_TRY(hipMemcpyAsync((char *)dst_dev_root + dst_offsets[0], band_dev[0], band_bytes[0], hipMemcpyDeviceToDevice, root->stream));
        return SKL_OK;
    }
    const RcclApi &api = RcclApi::get();
    if (!api.ok) return fail(SKL_ERR_HIP, "librccl.so.1 not found: the RCCL gather is not available on this host");
    std::vector<ncclComm_t> comms;
    {
        std::lock_guard<std::mutex> lock(g_rccl_mutex);
        auto it = g_rccl_comms.find(devs);
        if (it == g_rccl_comms.end()) {
            std::vector<ncclComm_t> fresh(n_ctx);
            RCCL_TRY(api.comm_init_all(fresh.data(), (int)n_ctx, devs.data()));
            it = g_rccl_comms.emplace(devs, std::move(fresh)).first;
        }
        comms = it->second;
    }
    if (n_ctx > 1 && band_bytes[0]) {
        HIP_TRY(hipMemcpyAsync((char *)dst_dev_root + dst_offsets[0], band_dev[0], band_bytes[0], hipMemcpyDeviceToDevice, root->stream));
    }
    // messages of at most 1 GiB (a cfg-3 band is 5-20 GB), every message of a round in ONE group: rank d sends on its context's
    // stream -- behind the kernels that made the band -- and the root receives on its own
    constexpr size_t MSG = 1ull << 30;
    size_t longest = 0;
    for (size_t d = n_ctx > 1 ? 1 : 0; d < n_ctx; ++d) longest = std::max(longest, band_bytes[d]);
    for (size_t off = 0; off < longest; off += MSG) {
        RCCL_TRY(api.group_start());
        for (size_t d = n_ctx > 1 ? 1 : 0; d < n_ctx; ++d) {
            if (off >= band_bytes[d]) continue;
            const size_t cnt = std::min(MSG, band_bytes[d] - off);
            HIP_TRY(hipSetDevice(devs[d]));
            RCCL_TRY(api.send((const char *)band_dev[d] + off, cnt, ncclUint8, 0, comms[d], ctxs[d]->stream));
            HIP_TRY(hipSetDevice(devs[0]));
            RCCL_TRY(api.recv((char *)dst_dev_root + dst_offsets[d] + off, cnt, ncclUint8, (int)d, comms[0], root->stream));
        }
        RCCL_TRY(api.group_end());
    }
    HIP_TRY(hipSetDevice(devs[0]));
    return SKL_OK;
}
