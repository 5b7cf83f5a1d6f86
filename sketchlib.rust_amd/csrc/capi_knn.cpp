// capi_knn.cpp -- the sparse (k nearest neighbours) entry points of include/sketchlib_dist.h:
// row-by-row bands, the one-evaluation self kNN, its multi-GPU split and the merge of partial
// states.  Kernels: pair_kslice.hip (pair distances, turned second store), kernels.hip
// (topk_merge_kernel, merge_states_kernel).
#include "capi_internal.hpp"

#include <algorithm>
#include <cstring>

using namespace skl;

// ---------------------------------------------------------------------------
// sparse kNN: dense row bands into scratch, then a row-wise top-k kernel
// ---------------------------------------------------------------------------

// Running top-k states of a kNN call: (sortable key, sample id[, second value]) x knn per row.
namespace {
struct KnnState {
    uint32_t *key = nullptr, *idx = nullptr;
    float *d1 = nullptr;
    // Reference tie order (skl_ctx_set_knn_ties): the state of a row is the reference's BinaryHeap itself -- heap-ordered
    // (key, id[, second value]) arrays of h_len items -- and thr its maximum once full, in sortable bits
    // (refheap_merge_kernel, topk.hip).  Null in the canonical mode.
    float *h_key = nullptr, *h_d1 = nullptr;
    uint32_t *h_id = nullptr, *h_len = nullptr, *thr = nullptr;
    bool borrowed = false;   // the arrays belong to the caller (skl_self_dists_knn_window)
    // accept log of the heap replays (skl_self_dists_knn_window_logged; RefHeapMergeArgs::log_*), the caller's arrays
    float *log_rec = nullptr;
    uint32_t *log_id = nullptr, *log_len = nullptr;
    uint32_t log_cap = 0;
    ~KnnState()
    {
        if (borrowed) return;
        for (void *p : {(void *)key, (void *)idx, (void *)d1, (void *)h_key, (void *)h_d1, (void *)h_id, (void *)h_len, (void *)thr}) {
            if (p) (void)hipFree(p);
        }
    }
};
}  // namespace

// Rows per band of the symmetric drivers: about 8 bands per participant (7/16 of the pair
// evaluations saved), each band at least 32 M pairs, four band buffers within `budget` bytes.
// TILE-PRUNING COUNTERS (diagnostic): 1 024 slots of 4 words on the device (scratch slot 10), added to by the pair kernels of
// every band of a call, read back LAZILY by skl_ctx_knn_prune_stats -- never on the launch path: the drivers that feed bands one
// call at a time (column windows, column panels) must not stall the host once per call.
static int prune_stats_reset(skl_ctx *ctx)
{
    ctx->knn_tiles = ctx->knn_tiles_sparse = ctx->knn_tiles_pruned = ctx->knn_tiles_probe_pruned = ctx->knn_pruned_stages = ctx->knn_tile_stages = 0;
    if (ctx->scratch[10] != nullptr) HIP_TRY(hipMemsetAsync(ctx->scratch[10], 0, 4096 * sizeof(uint32_t), ctx->stream));
    return SKL_OK;
}

static int prune_stats_collect(skl_ctx *ctx)
{
    if (ctx->scratch[10] == nullptr || !ctx->knn_prune_pending) return SKL_OK;
    std::vector<uint32_t> counted(4096, 0u);
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->aux_stream) HIP_TRY(hipStreamSynchronize(ctx->aux_stream));
    HIP_TRY(hipMemcpy(counted.data(), ctx->scratch[10], counted.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    ctx->knn_tiles_pruned = ctx->knn_tiles_probe_pruned = ctx->knn_pruned_stages = ctx->knn_tiles_sparse = 0;
    for (size_t x = 0; x < 1024; ++x) {
        ctx->knn_tiles_pruned += (uint64_t)counted[4 * x] + counted[4 * x + 1];
        ctx->knn_tiles_probe_pruned += counted[4 * x];
        ctx->knn_pruned_stages += counted[4 * x + 2];
        ctx->knn_tiles_sparse += counted[4 * x + 3];
    }
    return SKL_OK;
}

// Bytes per pair and band buffer the symmetric drivers' budget counts: the record, and for core/accessory keys the share of
// the early break's two counts buffers (up to 4 lengths x 2 bytes each against four band buffers: + 4 bytes per buffer).
static size_t coreacc_rec_with_counts(const skl_sketches *s, const skl_dist_params *p)
{
    if (p->dist_type != SKL_DIST_COREACC) return sizeof(float);
    return 2 * sizeof(float) + (s->nk >= 3 && s->nk <= 8 && fused_coreacc_ok(s) ? 4 : 0);
}

static size_t symmetric_band_rows(size_t n, size_t rec, size_t budget, size_t participants)
{
    auto up16 = [](size_t x) { return (x + 15) / 16 * 16; };
    const size_t budget_rows = std::max<size_t>(16, budget / 4 / (n * rec) / 16 * 16);
    const size_t parts = std::max<size_t>(1, participants);
    return std::min(budget_rows, std::max(up16((n + 8 * parts - 1) / (8 * parts)), up16((32ull << 20) / n + 1)));
}

static int knn_state_init(KnnState &st, size_t rows, size_t knn, bool coreacc, hipStream_t stream, bool ref_heap = false)
{
    const size_t items = rows * knn;
    if (ref_heap) {
        HIP_TRY(hipMalloc((void **)&st.h_key, items * sizeof(float)));
        HIP_TRY(hipMalloc((void **)&st.h_id, items * sizeof(uint32_t)));
        if (coreacc) HIP_TRY(hipMalloc((void **)&st.h_d1, items * sizeof(float)));
        HIP_TRY(hipMalloc((void **)&st.h_len, rows * sizeof(uint32_t)));
        HIP_TRY(hipMalloc((void **)&st.thr, rows * sizeof(uint32_t)));
        HIP_TRY(hipMemsetAsync(st.h_len, 0, rows * sizeof(uint32_t), stream));     // empty heaps
        HIP_TRY(hipMemsetAsync(st.thr, 0xFF, rows * sizeof(uint32_t), stream));    // not full: everything may enter
        return SKL_OK;
    }
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
    // (beyond 65 535 bins: single-k only -- the k-sliced form walks the k-mer length in segments, the fused
    // core/accessory form has no such walk)
    if (p->dist_type == SKL_DIST_COREACC && !fused_coreacc_ok(s)) return false;
    const int forced = forced_kernel(s->ctx);
    if (forced != 0 && forced != 4) return false;          // the turned store lives in pair_kslice.hip
    return true;
}

// The bands `bands` (ascending indices; band b = rows [b*band_rows, (b+1)*band_rows)) merged into
// the running states `st` of all n rows.
// COLUMN WINDOW (win_lo, win_hi; default: all columns): only the pairs whose COLUMN sample lies in [win_lo, win_hi) are
// evaluated -- band rows against columns [max(b0, win_lo), win_hi), turned copies to the rows [max(b1, win_lo), win_hi) -- which
// is one participant's share of the reference-order pipeline over several devices (skl_self_dists_knn_window).
// CROSS FORM (`cols` given): the rows of `s` against the columns [win_lo, win_hi) of another slab (or of the same one, row
// ranges of the self kNN: self_rows), no symmetry, nothing turned -- one COLUMN PANEL of the row-by-row kNN.  The drivers
// call it panel after panel, ascending, so a row meets its candidates in ascending id, its list tightens from panel to panel,
// and from the second panel on the pair kernel prunes against the rows' bounds (the columns have no lists: bound 0).
struct KnnCross {
    const skl_sketches *cols = nullptr;   // null: the symmetric form
    bool self_rows = false;               // rows and columns are the same sample set: a row is not its own candidate
    size_t row_lo = 0, row_hi = ~(size_t)0;   // rows of the call (bands are clipped to them)
    // (symmetric form, one band per call -- skl_self_dists_knn_window: every list the band meets already holds knn candidates,
    // as from a whole-matrix call's second band on: the early break of the core/accessory keys may start with the call's
    // first band)
    bool lists_hold_knn = false;
};

static int knn_symmetric_bands(skl_ctx *ctx, const skl_sketches *s, const skl_dist_params *p, size_t knn,
                               size_t band_rows, const std::vector<uint32_t> &bands, bool overlap, KnnState &st,
                               size_t win_lo = 0, size_t win_hi = ~(size_t)0, const KnnCross &cross = KnnCross())
{
    const bool is_cross = cross.cols != nullptr;
    const skl_sketches *cs = is_cross ? cross.cols : s;
    const size_t n = cs->n;            // columns (= samples in the symmetric form)
    const size_t n_rows = s->n;
    win_hi = std::min(win_hi, n);
    const bool coreacc = p->dist_type == SKL_DIST_COREACC;
    const bool ref = st.h_key != nullptr;   // the reference's tie order: heaps replayed (a row's candidates arrive in ascending id
                                            // over the bands -- turned from the bands above its own, then its own band's columns)
    const int mode = coreacc ? MODE_COREACC : MODE_JACCARD;
    const int jout = coreacc ? 0 : (p->ani ? JOUT_ANI_KEY : JOUT_DIST);
    const size_t rec = coreacc ? 2 * sizeof(float) : sizeof(float);
    const size_t t_stride = (band_rows + 31) / 32 * 32;   // whole tiles of either height (16 or 32 rows: dispatch_pair_kernel)
    void *kband[2] = {nullptr, nullptr}, *tband[2] = {nullptr, nullptr};
    const size_t k_cols = is_cross ? win_hi - win_lo / 64 * 64 : n;   // columns of a band's records (cross form: the panel's)
    SKL_TRY(ctx_scratch(ctx, band_rows * k_cols * rec, &kband[0], 0));
    if (!is_cross) SKL_TRY(ctx_scratch(ctx, n * t_stride * rec, &tband[0], 4));
    kband[1] = kband[0];
    tband[1] = tband[0];
    if (overlap) {
        SKL_TRY(ctx_scratch(ctx, band_rows * k_cols * rec, &kband[1], 3));
        if (!is_cross) SKL_TRY(ctx_scratch(ctx, n * t_stride * rec, &tband[1], 5));
    }
    // Row flags of the transposed band (one array per band buffer): the pair kernel marks the rows that
    // received a record below their knn-th best so far with the band's number, and the merge of the
    // transposed band (n - b1 workgroups reading band_rows records each: 2/3 of the merge time at cfg 5)
    // returns at once for the others.  The threshold the pair kernel compares with is read from the
    // running state while merges of earlier bands may still be updating it on the other stream: a
    // stale value is a higher one (a row's knn-th best only ever improves), so it flags too many rows,
    // never too few; ties never count (a band's sample ids are above every id a lower row holds).
    void *flag_mem = nullptr;
    SKL_TRY(ctx_scratch(ctx, 2 * n * sizeof(uint32_t), &flag_mem, 6));
    uint32_t *flags[2] = {(uint32_t *)flag_mem, (uint32_t *)flag_mem + (overlap ? n : 0)};
    HIP_TRY(hipMemsetAsync(flag_mem, 0, 2 * n * sizeof(uint32_t), ctx->stream));
    // ... and for the band's own rows one bit per 64-column block (the merge of the band reads only the
    // marked stretches of a row): band_rows x ceil(columns / 2048) words per band buffer, behind the flags
    const size_t bit_words = ((is_cross ? k_cols : n) / 64 + 1 + 31) / 32;   // (bit = 64-column block of the band's view)
    void *bits_mem = nullptr;
    SKL_TRY(ctx_scratch(ctx, 2 * band_rows * bit_words * sizeof(uint32_t), &bits_mem, 7));
    uint32_t *row_bits[2] = {(uint32_t *)bits_mem, (uint32_t *)bits_mem + (overlap ? band_rows * bit_words : 0)};
    // ... and for the turned band one bit per (column, 32-row stretch of the band), so that BOTH merges read marked stretches
    // only and a tile without a mark need not exist: TILE PRUNING (pair_kslice_walk.inc).  Single-k keys are monotone in the
    // mismatch count, so before each band a small kernel turns every sample's current knn-th best into the mismatch count
    // beyond which a pair cannot enter its list, and the pair kernel leaves a tile once every pair of it is beyond both its
    // samples' bounds on the chunks walked so far.  The bounds come from the same (possibly stale, i.e. too high) thresholds
    // as the flags: a pair pruned now would be rejected by both lists whenever it arrived, so the lists -- ids AND order, in
    // either tie rule -- are those of the unpruned run.  Not with a completeness correction (the key then depends on the pair).
    const size_t tbit_words = is_cross ? 0 : (t_stride / 32 + 31) / 32;   // (cross form: nothing turned; the array is only a non-null mark)
    const bool prune = ctx->knobs.knn_prune && ctx->knobs.knn_row_flags && !coreacc && !(s->d_comp != nullptr) &&
                       s->ss64 <= (size_t)KSLICE_MAX_U16_CHUNKS;
    uint32_t *tbits[2] = {nullptr, nullptr}, *prune_q = nullptr, *prune_stats = nullptr;
    if (ctx->knobs.knn_row_flags) {
        void *tb = nullptr;
        SKL_TRY(ctx_scratch(ctx, std::max<size_t>(2 * n * tbit_words, 64) * sizeof(uint32_t), &tb, 9));
        tbits[0] = (uint32_t *)tb;
        tbits[1] = (uint32_t *)tb + (overlap ? n * tbit_words : 0);
    }
    if (prune) {
        void *pq = nullptr, *ps = nullptr;
        SKL_TRY(ctx_scratch(ctx, (n_rows + 64 + (is_cross ? n + 64 : 0)) * sizeof(uint32_t), &pq, 8));
        const bool fresh = ctx->scratch[10] == nullptr;
        SKL_TRY(ctx_scratch(ctx, 4096 * sizeof(uint32_t), &ps, 10));
        prune_q = (uint32_t *)pq;
        prune_stats = (uint32_t *)ps;
        if (fresh) HIP_TRY(hipMemsetAsync(prune_stats, 0, 4096 * sizeof(uint32_t), ctx->stream));   // (afterwards: prune_stats_reset, at the start of a call)
        ctx->knn_prune_pending = true;
        if (is_cross) HIP_TRY(hipMemsetAsync(prune_q + n_rows + 64, 0, (n + 64) * sizeof(uint32_t), ctx->stream));   // the columns have no lists: bound 0
    }
    hipStream_t topk_stream = overlap ? ctx->aux_stream : ctx->stream;
    if (overlap) {   // the states were cleared on the context's stream, the merges run on the other one
        HIP_TRY(hipEventRecord(ctx->knn_pair_done[0], ctx->stream));
        HIP_TRY(hipStreamWaitEvent(topk_stream, ctx->knn_pair_done[0], 0));
    }

    // EARLY BREAK (core/accessory keys; capi.cpp early_break_lengths): from the call's second band on -- every list then holds
    // knn candidates, so a pair that left the reference's loop early, (1, 1), marks nothing -- the band is COUNTED at its first
    // eb_lengths k-mer lengths (k-sliced counts launch) and coreacc_epilogue_knn_kernel writes the records, the marks and the
    // turned copy (pre-filled with (1, 1)), completing the pairs still in the running.  Same records as the fused kernel's.
    // The counts kernel stays on the context's stream; the band's epilogue goes with the merges (the other stream when bands
    // overlap): it is bound by memory and by the latency of the one-by-one completions, the counts kernel by the VALUs.
    int eb_lengths = 0;
    void *eb_counts[2] = {nullptr, nullptr};
    if (coreacc && !is_cross && fused_coreacc_ok(s) && forced_kernel(ctx) == 0 && ctx->knobs.knn_row_flags &&
        (bands.size() > 1 || cross.lists_hold_knn)) {
        SKL_TRY(early_break_lengths(ctx, s, cs, 1, &eb_lengths));
        if (eb_lengths > 0) {
            // (the band heights count these buffers in -- coreacc_rec_with_counts() -- but a band height the CALLER chose, or a
            // device short of memory, must not fail the call: without the counts the bands take the fused kernel as before)
            const size_t bytes = band_rows * n * (size_t)eb_lengths * sizeof(uint16_t);   // (the largest view a band can have; u16 records: fused_coreacc_ok means at most 65 472 bins)
            int rc = ctx_scratch(ctx, bytes, &eb_counts[0], 1);
            ctx->clean_plane1 = nullptr;   // (the counts scratch holds another layout now)
            eb_counts[1] = eb_counts[0];
            if (rc == SKL_OK && overlap) rc = ctx_scratch(ctx, bytes, &eb_counts[1], 15);
            if (rc == SKL_ERR_OOM) {
                (void)hipGetLastError();   // (cleared: the call goes on)
                eb_lengths = 0;
            } else if (rc != SKL_OK) {
                return rc;
            }
        }
    }
    const size_t jb_words = s->nk * s->ss64 * 7 * 64;   // uint4 per 64-column block of the lane slab
    size_t it = 0;
    for (const uint32_t band : bands) {
        const size_t b0 = std::max((size_t)band * band_rows, cross.row_lo);
        const size_t b1 = std::min(std::min(n_rows, (size_t)band * band_rows + band_rows), cross.row_hi);
        if (b1 <= b0) continue;
        const size_t c_first = is_cross ? win_lo : std::max(b0, win_lo);   // first candidate column of the band's own rows
        const size_t t_first = is_cross ? win_hi : std::max(b1, win_lo);   // first row that receives the band turned (cross form: none)
        if (c_first >= win_hi) continue;               // the window lies left of this band: nothing of it here
        const RoctxRange range_("skl:knn_band pair kernel + merges (every pair once)");
        const int buf = overlap ? (int)(it & 1) : 0;
        if (overlap && it >= 2) HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->knn_topk_done[buf], 0));
        // the band against the column view that starts at the 64-column block holding its first candidate column
        const size_t col0 = c_first / 64 * 64;
        PairArgs g;
        SKL_TRY(fill_args(s, cs, p, mode, jout, &g));
        g.B += (col0 / 64) * jb_words;
        g.nB = (uint32_t)(win_hi - col0);
        if (g.compB) g.compB += col0;
        g.row_begin = (uint32_t)b0;
        g.row_end = (uint32_t)b1;
        g.self_mode = 0;
        g.out_base = (uint64_t)b0 * g.nB;
        g.out = kband[buf];
        g.out_t = t_first < win_hi ? (float *)tband[buf] : nullptr;
        g.t_col_begin = (uint32_t)(t_first - col0);
        g.t_stride = (uint32_t)t_stride;
        const uint32_t flag_value = (uint32_t)(it + 1);      // never 0, distinct per band of this call
        if (ctx->knobs.knn_row_flags) {
            HIP_TRY(hipMemsetAsync(row_bits[buf], 0, band_rows * bit_words * sizeof(uint32_t), ctx->stream));
            g.r_bits = row_bits[buf];
            g.r_bits_stride = (uint32_t)bit_words;
            g.r_thr = ref ? st.thr + b0 : st.key + b0 * knn + (knn - 1);        // knn-th best of sample b0 + r
            g.r_thr_stride = ref ? 1u : (uint32_t)knn;
        }
        if (g.out_t && ctx->knobs.knn_row_flags) {
            g.t_flag = flags[buf] + col0;                   // indexed by the view's column number, like t_col_begin
            g.t_flag_value = flag_value;
            g.t_thr = ref ? st.thr + col0 : st.key + col0 * knn + (knn - 1);      // knn-th best of sample col0 + c
            g.t_thr_stride = ref ? 1u : (uint32_t)knn;
            HIP_TRY(hipMemsetAsync(tbits[buf] + t_first * tbit_words, 0, (win_hi - t_first) * tbit_words * sizeof(uint32_t), ctx->stream));
            g.t_bits = tbits[buf] + col0 * tbit_words;
            g.t_bits_stride = (uint32_t)tbit_words;
        }
        if (prune) {
            // every sample's bound as of now (the merges of earlier bands may still be lowering thresholds: stale = too high = safe)
            HIP_TRY(launch_prune_thresholds(ref ? st.thr : st.key + (knn - 1), ref ? 1u : (uint32_t)knn, (uint32_t)n_rows, g.dtab,
                                            (uint32_t)(64 * s->ss64), prune_q, ctx->stream));
            g.prune_q_rows = prune_q;
            g.prune_q_cols = (is_cross ? prune_q + n_rows + 64 : prune_q) + col0;
            g.prune_stats = prune_stats;
            g.prune_flags = ctx->knobs.knn_sparse ? 0u : 1u;

            if (!g.t_bits) {   // (the last band has no turned copy; the kernel takes "both bit sets given" as the sign that the merges mask)
                g.t_bits = tbits[buf] + col0 * tbit_words;
                g.t_bits_stride = (uint32_t)tbit_words;
            }
            ctx->knn_tiles += (uint64_t)((b1 - b0 + 31) / 32) * ((g.nB + 127) / 128);
        }
        const bool eb_band = eb_lengths > 0 && (it >= 1 || cross.lists_hold_knn);
        EpilogueKnnArgs e;
        if (eb_band) {
            const size_t pairs_view = (b1 - b0) * (size_t)g.nB;
            void *counts = eb_counts[buf];
            PairArgs c;
            SKL_TRY(fill_args(s, cs, p, MODE_COUNTS, 0, &c));
            c.B += (col0 / 64) * jb_words;
            c.nB = g.nB;
            c.row_begin = g.row_begin;
            c.row_end = g.row_end;
            c.self_mode = 0;
            c.out_base = g.out_base;
            c.k_count = (uint32_t)eb_lengths;
            c.cnt_pair_stride = 1;
            c.cnt_k_stride = pairs_view;
            c.k_sliced = 1;
            c.k_slices = 1;
            c.cnt_u16 = 1;
            c.out = counts;
            SKL_TRY(timed_pair_launch(ctx, c, MODE_COUNTS));
            memset(&e, 0, sizeof e);
            e.counts = (const uint32_t *)counts;
            e.n_pairs = pairs_view;
            e.rows = (uint32_t)(b1 - b0);
            e.nB = g.nB;
            e.nk = (uint32_t)eb_lengths;
            e.nk_total = (uint32_t)s->nk;
            e.ss64 = (uint32_t)s->ss64;
            e.row_sample0 = (uint32_t)b0;
            e.col_sample0 = (uint32_t)col0;
            e.ytab = s->d_ytab;
            e.kf = s->d_kf;
            e.tolerance = g.tolerance;
            e.rows_ref = s->d_rows;
            e.cols_ref = cs->d_rows;
            e.out = (float *)kband[buf];
            e.r_thr = g.r_thr;
            e.r_thr_stride = g.r_thr_stride;
            e.r_bits = g.r_bits;
            e.r_bits_stride = g.r_bits_stride;
            e.out_t = g.out_t;
            e.t_col_begin = g.t_col_begin;
            e.t_stride = g.t_stride;
            e.t_thr = g.t_thr;
            e.t_thr_stride = g.t_thr_stride;
            e.t_flag = g.t_flag;
            e.t_flag_value = g.t_flag_value;
            e.t_bits = g.t_bits;
            e.t_bits_stride = g.t_bits_stride;
            e.alive_count = ctx->eb_counter;
            e.min_alive = s->min_alive;
            e.xcd_blocked = (uint32_t)ctx->knobs.knn_epi_blocked;
            e.cnt_u16 = 1;
            // (bands ascend: the `it` bands before this one each gave band_rows candidates to every row the turned copy reaches,
            // and their merges run before this launch on the same stream)
            e.plain_marks_nothing = (cross.lists_hold_knn || it * band_rows >= knn) ? 1u : 0u;
            ctx->eb_pairs += pairs_view;
            ctx->last_kernel += " + early break: " + std::to_string(eb_lengths) + " of " + std::to_string(s->nk) + " k-mer lengths counted, the pairs still in the running completed by the band's epilogue";
        } else {
            SKL_TRY(timed_pair_launch(ctx, g, mode));
        }
        if (overlap) {
            HIP_TRY(hipEventRecord(ctx->knn_pair_done[buf], ctx->stream));
            HIP_TRY(hipStreamWaitEvent(topk_stream, ctx->knn_pair_done[buf], 0));
        }
        if (eb_band) {   // counts -> records, marks, turned copy: with the merges, behind the counts kernel
            if (g.out_t != nullptr) {   // (1, 1): every pair that left the loop before its third length
                HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)tband[buf], 0x3F800000, (win_hi - t_first) * t_stride * 2, topk_stream));
            }
            HIP_TRY(launch_coreacc_epilogue_knn(e, topk_stream));
        }
        if (ref) {
            RefHeapMergeArgs m;
            memset(&m, 0, sizeof m);
            m.knn = (uint32_t)knn;
            m.stride2 = coreacc ? 2 : 1;
            m.h_key = st.h_key;
            m.h_id = st.h_id;
            m.h_d1 = st.h_d1;
            m.h_len = st.h_len;
            m.thr = st.thr;
            m.log_rec = st.log_rec;
            m.log_id = st.log_id;
            m.log_len = st.log_len;
            m.log_cap = st.log_cap;
            m.force_workgroup_form = ctx->knobs.refheap_wave ? 0u : 1u;
            // rows below the band FIRST: for them the band's samples are the next candidates in ascending id, and their
            // own band comes later; then the band's own rows (columns [b0, n) minus themselves: everything below b0 reached
            // them turned, from the bands above).  Either way a row is fed ascending ids over the sequence of launches.
            m.keys = (const float *)tband[buf];
            m.key_stride = (uint64_t)t_stride * m.stride2;
            m.rows = g.out_t ? (uint32_t)(win_hi - t_first) : 0u;
            m.cols = (uint32_t)(b1 - b0);
            m.id_base = (uint32_t)b0;
            m.skip_below = 0;
            m.self_id_base = m.state_row_base = (uint32_t)t_first;
            m.flag = ctx->knobs.knn_row_flags ? flags[buf] + t_first : nullptr;
            m.flag_value = flag_value;
            m.seg_bits = ctx->knobs.knn_row_flags ? tbits[buf] + t_first * tbit_words : nullptr;   // (row r of this launch = sample t_first + r)
            m.seg_bits_stride = (uint32_t)tbit_words;
            m.seg_shift = 5;
            HIP_TRY(launch_refheap_merge(m, topk_stream));
            m.flag = nullptr;
            m.seg_shift = 6;
            m.keys = (const float *)kband[buf];
            m.key_stride = (uint64_t)g.nB * m.stride2;
            m.rows = (uint32_t)(b1 - b0);
            m.cols = g.nB;
            m.id_base = (uint32_t)col0;
            m.skip_below = (uint32_t)c_first;
            m.state_row_base = (uint32_t)b0;
            m.self_id_base = (is_cross && !cross.self_rows) ? 0xFFFFFFFFu : (uint32_t)b0;
            m.seg_bits = ctx->knobs.knn_row_flags ? row_bits[buf] : nullptr;
            m.seg_bits_stride = (uint32_t)bit_words;
            HIP_TRY(launch_refheap_merge(m, topk_stream));
            if (overlap) HIP_TRY(hipEventRecord(ctx->knn_topk_done[buf], topk_stream));
            ++it;
            continue;
        }
        TopkMergeArgs m;
        memset(&m, 0, sizeof m);
        m.knn = (uint32_t)knn;
        m.stride2 = coreacc ? 2 : 1;
        m.run_key = st.key;
        m.run_idx = st.idx;
        m.run_d1 = st.d1;
        m.streaming = ctx->knobs.topk_stream;
        // rows of the band: columns [b0, n) minus themselves (the view's first b0 - col0 columns
        // reached them turned, from earlier bands)
        m.keys = (const float *)kband[buf];
        m.key_stride = (uint64_t)g.nB * m.stride2;
        m.rows = (uint32_t)(b1 - b0);
        m.cols = g.nB;
        m.id_base = (uint32_t)col0;
        m.skip_below = (uint32_t)c_first;
        m.state_row_base = (uint32_t)b0;
        m.self_id_base = (is_cross && !cross.self_rows) ? 0xFFFFFFFFu : (uint32_t)b0;
        m.seg_bits = ctx->knobs.knn_row_flags ? row_bits[buf] : nullptr;
        m.seg_bits_stride = (uint32_t)bit_words;
        HIP_TRY(launch_topk_merge(m, topk_stream));
        m.seg_bits = nullptr;
        // rows below the band: the band's samples as their candidates
        m.keys = (const float *)tband[buf];
        m.key_stride = (uint64_t)t_stride * m.stride2;
        m.rows = g.out_t ? (uint32_t)(win_hi - t_first) : 0u;
        m.cols = (uint32_t)(b1 - b0);
        m.id_base = (uint32_t)b0;
        m.skip_below = 0;
        m.self_id_base = m.state_row_base = (uint32_t)t_first;
        m.flag = ctx->knobs.knn_row_flags ? flags[buf] + t_first : nullptr;
        m.flag_value = flag_value;
        m.seg_bits = ctx->knobs.knn_row_flags ? tbits[buf] + t_first * tbit_words : nullptr;   // (row r of this launch = sample t_first + r)
        m.seg_bits_stride = (uint32_t)tbit_words;
        m.seg_shift = 5;
        HIP_TRY(launch_topk_merge(m, topk_stream));
        if (overlap) HIP_TRY(hipEventRecord(ctx->knn_topk_done[buf], topk_stream));
        ++it;
    }
    if (prune) ctx->knn_tile_stages = (s->ss64 + 3) / 4;   // stages of a whole 32 x 128 tile: 4 waves, one chunk each per stage
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
    const bool ref = ctx->knn_ties == SKL_KNN_TIES_REFERENCE;
    KnnState st;
    SKL_TRY(knn_state_init(st, n, knn, coreacc, ctx->stream, ref));
    std::vector<uint32_t> bands((n + band_rows - 1) / band_rows);
    for (size_t b = 0; b < bands.size(); ++b) bands[b] = (uint32_t)b;
    SKL_TRY(knn_symmetric_bands(ctx, s, p, knn, band_rows, bands, overlap, st));
    if (ref) {
        HIP_TRY(launch_refheap_finalize(st.h_key, st.h_id, st.h_d1, st.h_len, (uint32_t)n, (uint32_t)knn, (!coreacc && p->ani) ? 1 : 0,
                                        d_idx, d_d0, d_d1, ctx->stream));
    } else {
        HIP_TRY(launch_topk_finalize(st.key, st.idx, st.d1, n * knn, (!coreacc && p->ani) ? 1 : 0, d_idx, d_d0, d_d1,
                                     ctx->stream));
    }
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
    // Three ways from a band of records to neighbour lists:
    //   * the streaming running top-k (topk_merge_kernel) + finalize: canonical ties, knn <= TOPK_LDS_MAX;
    //   * the radix select with its items in global memory (topk_kernel, dense form): canonical ties, any knn;
    //   * the BinaryHeap replay (topk_refheap_kernel): the reference binary's tie order, any knn.
    const bool ref_ties = ctx->knn_ties == SKL_KNN_TIES_REFERENCE;
    const bool big = knn > (size_t)TOPK_LDS_MAX;
    const bool streaming_state = !ref_ties && !big;
    // COLUMN PANELS (round 5): a row-by-row kNN over many candidates -- cross kNN against a large reference set, a row range of
    // the self kNN -- is fed its candidates in ascending panels of columns instead of all at once: the rows' lists tighten
    // from panel to panel, and from the second panel on the pair kernel leaves the tiles whose pairs are beyond their ROW's
    // bound (tile pruning, as in the symmetric driver; the columns have no lists here).  Same lists in either tie rule: a
    // row still meets its candidates in ascending id.  Single-k keys without a completeness correction, lists that fit the
    // LDS forms, at least 4 panels of 32 Ki columns and launches large enough for the prunable 32 x 128 tiles.
    {
        // (A/B build: SKL_KNN_PANEL forces a panel width -- and lifts the size conditions -- so that tests reach this path on
        // inputs small enough for the oracle)
        const size_t forced_panel = (size_t)std::max(0ll, ctx->knobs.knn_panel) / 128 * 128;
        const size_t panel = forced_panel ? forced_panel : std::max<size_t>(32768, (n_cand / 8 + 127) / 128 * 128);
        const bool eligible = ctx->knobs.knn_prune && ctx->knobs.knn_row_flags && !coreacc && !(rows->d_comp && cands->d_comp) &&
                              rows->ss64 <= (size_t)KSLICE_MAX_U16_CHUNKS && !big && knn <= (size_t)REFHEAP_LDS_MAX && forced_kernel(ctx) == 0 &&
                              (forced_panel ? n_cand > panel : (n_cand >= 4 * panel && (r1 - r0) * panel >= (size_t)(16u << 20)));
        if (eligible) {
            // bands of rows whose records of one panel fit a quarter of the budget the caller sized `band_rows` for
            size_t rows_per = std::max<size_t>(32, std::min<size_t>(r1 - r0, band_rows * n_cand / panel) / 32 * 32);
            if (ctx->knobs.knn_band_rows) rows_per = std::max<size_t>(1, (size_t)ctx->knobs.knn_band_rows);   // (test knob)
            rows_per = std::min(rows_per, (size_t)1 << 20);
            KnnState pst;
            SKL_TRY(knn_state_init(pst, rows->n, knn, false, ctx->stream, ref_ties));
            std::vector<uint32_t> bands;
            for (size_t b = r0 / rows_per; b * rows_per < r1; ++b) bands.push_back((uint32_t)b);
            KnnCross cross;
            cross.cols = cands;
            cross.self_rows = self_mode != 0;
            cross.row_lo = r0;
            cross.row_hi = r1;
            SKL_TRY(prune_stats_reset(ctx));
            for (size_t c0 = 0; c0 < n_cand; c0 += panel) {
                SKL_TRY(knn_symmetric_bands(ctx, rows, p, knn, rows_per, bands, overlap && bands.size() > 1, pst, c0, std::min(n_cand, c0 + panel), cross));
            }
            const int ani_undo = p->ani ? 1 : 0;
            if (ref_ties) {
                HIP_TRY(launch_refheap_finalize(pst.h_key + r0 * knn, pst.h_id + r0 * knn, nullptr, pst.h_len + r0, (uint32_t)(r1 - r0), (uint32_t)knn,
                                                ani_undo, d_idx, d_d0, d_d1, ctx->stream));
            } else {
                HIP_TRY(launch_topk_finalize(pst.key + r0 * knn, pst.idx + r0 * knn, nullptr, (r1 - r0) * knn, ani_undo, d_idx, d_d0, d_d1, ctx->stream));
            }
            HIP_TRY(hipStreamSynchronize(ctx->stream));   // the running states are freed on return
            return SKL_OK;
        }
    }
    KnnState st;
    DevBuf big_scratch;
    if (streaming_state) {
        SKL_TRY(knn_state_init(st, r1 - r0, knn, coreacc, ctx->stream));
    } else if (ref_ties && knn > (size_t)REFHEAP_LDS_MAX) {
        HIP_TRY(hipMalloc(&big_scratch.p, band_rows * 3 * (knn + 1) * sizeof(float)));
    } else if (!ref_ties && big) {
        HIP_TRY(hipMalloc(&big_scratch.p, band_rows * topk_items_pitch(knn) * sizeof(uint64_t)));
    }
    if (overlap) {   // the states are cleared on the context's stream, the merges run on the other one
        HIP_TRY(hipEventRecord(ctx->knn_pair_done[0], ctx->stream));
        HIP_TRY(hipStreamWaitEvent(topk_stream, ctx->knn_pair_done[0], 0));
    }

    size_t it = 0;
    for (size_t b0 = r0; b0 < r1; b0 += band_rows, ++it) {
        const size_t b1 = std::min(r1, b0 + band_rows);
        const RoctxRange range_("skl:knn_band pair kernel + top-k (row by row)");
        const int buf = overlap ? (int)(it & 1) : 0;
        // the top-k that read this buffer two bands ago must be done before it is overwritten
        if (overlap && it >= 2) HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->knn_topk_done[buf], 0));
        SKL_TRY(dense_band(ctx, rows, cands, p, mode, jout, 0, b0, b1, band[buf]));
        if (overlap) {
            HIP_TRY(hipEventRecord(ctx->knn_pair_done[buf], ctx->stream));
            HIP_TRY(hipStreamWaitEvent(topk_stream, ctx->knn_pair_done[buf], 0));
        }
        const size_t o = (b0 - r0) * knn;   // first output item of the band
        const int ani_undo = (!coreacc && p->ani) ? 1 : 0;
        if (streaming_state) {
            TopkMergeArgs m;
            memset(&m, 0, sizeof m);
            m.knn = (uint32_t)knn;
            m.stride2 = coreacc ? 2 : 1;
            m.run_key = st.key;
            m.run_idx = st.idx;
            m.run_d1 = st.d1;
            m.streaming = ctx->knobs.topk_stream;
            m.key_stride = (uint64_t)n_cand * m.stride2;
            m.rows = (uint32_t)(b1 - b0);
            m.self_id_base = self_mode ? (uint32_t)b0 : 0xFFFFFFFFu;
            m.state_row_base = (uint32_t)(b0 - r0);
            m.keys = (const float *)band[buf];
            m.cols = (uint32_t)n_cand;
            m.id_base = 0;
            HIP_TRY(launch_topk_merge(m, topk_stream));
        } else if (ref_ties) {
            RefHeapArgs h;
            memset(&h, 0, sizeof h);
            h.keys = (const float *)band[buf];
            h.stride2 = coreacc ? 2 : 1;
            h.key_stride = (uint64_t)n_cand * h.stride2;
            h.rows = (uint32_t)(b1 - b0);
            h.cols = (uint32_t)n_cand;
            h.self_id_base = self_mode ? (uint32_t)b0 : 0xFFFFFFFFu;
            h.knn = (uint32_t)knn;
            h.ani_undo = ani_undo;
            h.out_idx = d_idx + o;
            h.out_d0 = d_d0 + o;
            h.out_d1 = coreacc ? d_d1 + o : nullptr;
            h.heap_scratch = (float *)big_scratch.p;
            h.force_workgroup_form = ctx->knobs.refheap_wave ? 0u : 1u;
            HIP_TRY(launch_topk_refheap(h, topk_stream));
        } else {
            TopkArgs t;
            memset(&t, 0, sizeof t);
            t.keys = (const float *)band[buf];
            t.rows = (uint32_t)(b1 - b0);
            t.cols = (uint32_t)n_cand;
            t.stride2 = coreacc ? 2 : 1;
            t.knn = (uint32_t)knn;
            t.self_mode = self_mode;
            t.row_begin = (uint32_t)b0;
            t.ani_undo = ani_undo;
            t.out_idx = d_idx + o;
            t.out_d0 = d_d0 + o;
            t.out_d1 = coreacc ? d_d1 + o : nullptr;
            t.items_scratch = (uint64_t *)big_scratch.p;
            t.items_pitch = topk_items_pitch(knn);
            HIP_TRY(launch_topk(t, topk_stream));
        }
        if (overlap) HIP_TRY(hipEventRecord(ctx->knn_topk_done[buf], topk_stream));
    }
    if (streaming_state) {
        HIP_TRY(launch_topk_finalize(st.key, st.idx, st.d1, (r1 - r0) * knn, (!coreacc && p->ani) ? 1 : 0, d_idx, d_d0,
                                     d_d1, topk_stream));
    }
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
    if (r1 == r0) return SKL_OK;
    // (no upper bound on knn beyond the candidates there are, as in the reference, lib.rs:379-382 / mod.rs:325: up to
    // TOPK_LDS_MAX neighbours the lists live in LDS; more go through global memory, row by row)
    const bool ref_ties = ctx->knn_ties == SKL_KNN_TIES_REFERENCE;
    const bool big_knn = knn > (size_t)TOPK_LDS_MAX;

    const size_t rec = coreacc ? 2 * sizeof(float) : sizeof(float);
    // the key band lives only on the device: take up to a quarter of the free HBM (<= 8 GiB)
    // so that the row-wise top-k kernel has thousands of rows (= workgroups) per launch
    size_t free_b = 0, total_b = 0;
    size_t band_bytes = BAND_BYTES;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
        band_bytes = std::max(band_bytes, std::min<size_t>(free_b / 4, 8ull << 30));
    }
    size_t band_rows = std::max<size_t>(1, band_bytes / 2 / (n_cand * rec));   // two key bands
    const size_t forced_band_rows = (size_t)ctx->knobs.knn_band_rows;  // test knob: force several bands
    if (forced_band_rows) band_rows = forced_band_rows;
    band_rows = std::min(band_rows, r1 - r0);
    // The whole self matrix: evaluate each pair once (knn_self_symmetric) when that leaves bands
    // worth launching -- about 8 of them (7/16 of the pair evaluations saved), each at least 32 M
    // pairs, within four band buffers of up to half the free HBM (<= 32 GiB; core/accessory keys: <= 96 GiB) together.
    // (the reference's tie order depends on the ORDER candidates arrive in, ascending j for every row: the symmetric driver
    // delivers exactly that order, band by band, to a heap that lives in global memory between the bands; lists too long for
    // the LDS-resident running state go row by row)
    bool symmetric = self_mode && r0 == 0 && r1 == n_cand && knn_symmetric_ok(rows, p) && !big_knn &&
                     ctx->knobs.knn_symmetric;   // (SKL_KNN_SYMMETRIC=0: A/B against the row-by-row form)
    if (symmetric) {
        size_t budget = band_bytes;
        // (core/accessory keys -- no tile pruning, whose thresholds want short bands -- take taller bands: the bands' epilogue
        // finds a column group's slices in L2 for more rows, and there are fewer launches and heap replays: cfg 5 in
        // core/accessory mode, 704 / 1 408 / 2 048 / 2 816 / 4 096 rows: 25.1 / 24.7 / 24.6 / 24.6 / 24.6 s)
        const size_t budget_cap = p->dist_type == SKL_DIST_COREACC ? 96ull << 30 : 32ull << 30;
        if (free_b) budget = std::max(budget, std::min<size_t>(free_b / 2, budget_cap));
        const size_t want = forced_band_rows ? forced_band_rows : symmetric_band_rows(n_cand, coreacc_rec_with_counts(rows, p), budget, 1);
        if (want >= n_cand) symmetric = false;
        else band_rows = want;
    }
    if (!symmetric && (big_knn || ref_ties)) {
        // per-row working arrays in global memory (knn beyond the LDS forms): keep them within 1 GiB
        const size_t per_row = big_knn ? std::max<size_t>((size_t)topk_items_pitch(knn) * sizeof(uint64_t), 3 * (knn + 1) * sizeof(float)) : 0;
        if (per_row) band_rows = std::max<size_t>(1, std::min<size_t>(band_rows, (size_t)(1ull << 30) / per_row));
    }
    const bool overlap = ctx->knobs.knn_overlap && band_rows < r1 - r0;

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
    SKL_TRY(prune_stats_reset(ctx));
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

extern "C" int skl_ctx_knn_prune_stats(skl_ctx *ctx, uint64_t *tiles, uint64_t *tiles_pruned, uint64_t *stages_per_tile,
                                       uint64_t *stages_walked_in_pruned_tiles, uint64_t *tiles_sparse)
{
    SKL_TRY(ctx_bind(ctx));
    SKL_TRY(prune_stats_collect(ctx));
    if (tiles_sparse) *tiles_sparse = ctx->knn_tiles_sparse;
    if (tiles) *tiles = ctx->knn_tiles;
    if (tiles_pruned) *tiles_pruned = ctx->knn_tiles_pruned;
    if (stages_per_tile) *stages_per_tile = ctx->knn_tile_stages;
    if (stages_walked_in_pruned_tiles) *stages_walked_in_pruned_tiles = ctx->knn_pruned_stages;
    return SKL_OK;
}

extern "C" int skl_ctx_set_knn_ties(skl_ctx *ctx, int mode)
{
    SKL_TRY(ctx_bind(ctx));
    if (mode != SKL_KNN_TIES_CANONICAL && mode != SKL_KNN_TIES_REFERENCE) return fail(SKL_ERR_INVALID_ARG, "unknown kNN tie mode %d", mode);
    ctx->knn_ties = mode;
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

extern "C" size_t skl_knn_band_rows(const skl_sketches *s, const skl_dist_params *p, size_t n_participants)
{
    if (!s || !p || s->n == 0) return 0;
    const size_t rec = coreacc_rec_with_counts(s, p);
    const long long forced = s->ctx->knobs.knn_band_rows;   // test knob (the same for every participant)
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
    if (knn > (size_t)TOPK_LDS_MAX) return fail(SKL_ERR_INVALID_ARG, "the one-evaluation kNN keeps its running lists in LDS: knn=%zu exceeds %u; shard rows with skl_self_dists_knn_rows", knn, TOPK_LDS_MAX);
    if (ctx->knn_ties == SKL_KNN_TIES_REFERENCE) {
        return fail(SKL_ERR_INVALID_ARG, "the reference's tie order follows from the order candidates arrive in: no one-evaluation form; "
                                         "shard rows with skl_self_dists_knn_rows");
    }
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
    SKL_TRY(prune_stats_reset(ctx));
    const bool overlap = ctx->knobs.knn_overlap && list.size() > 1;
    SKL_TRY(knn_symmetric_bands(ctx, s, p, knn, band_rows, list, overlap, st));
    const hipMemcpyKind kind = out_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    const size_t items = n * knn;
    HIP_TRY(hipMemcpyAsync(state_key, st.key, items * sizeof(uint32_t), kind, ctx->stream));
    HIP_TRY(hipMemcpyAsync(state_idx, st.idx, items * sizeof(uint32_t), kind, ctx->stream));
    if (coreacc) HIP_TRY(hipMemcpyAsync(state_d1, st.d1, items * sizeof(float), kind, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));   // the running states are freed on return
    return SKL_OK;
}

// ---------------------------------------------------------------------------
// The reference's tie order over several devices, every pair evaluated once: the column-window pipeline.
// ---------------------------------------------------------------------------
// A row's BinaryHeap must meet its candidates in ascending id (mod.rs:156-181), so partial heaps cannot be merged; but a heap
// can TRAVEL.  Participant r owns the column window [lo_r, hi_r) (windows ascending with r, cut where the pair counts
// balance) and evaluates exactly the pairs (i, j) with i < j and j in its window: for each row band [b0, b1) with b0 < hi_r,
// ascending, the band's rows against the columns [max(b0, lo_r), hi_r) -- the band's own rows take them as candidates, and the
// window's rows below the band take the band's samples turned.  A row of the window therefore meets, on its owner, every id
// below its own (turned from the bands above it, ascending) and then its window's ids above; its heap then moves to
// participant r + 1 -- which feeds it the columns of ITS window when it reaches that row's band -- and so on to the last one,
// where every heap ends.  The heaps of rows [b0, b1) are handed on as soon as the band is done, so the participants work one
// band behind each other.  This call is one band on one participant; the caller owns the heap arrays (device memory, the
// RefHeap layout of skl_knn_heaps_*: h_key / h_id / h_d1 [n][knn], h_len [n], thr [n]) and moves row slices between devices.
static int knn_window_impl(skl_ctx *ctx, const skl_sketches *s, const skl_dist_params *p, size_t knn,
                           size_t band_rows, size_t band, size_t col_lo, size_t col_hi,
                           float *h_key, uint32_t *h_id, float *h_d1, uint32_t *h_len, uint32_t *thr,
                           float *log_rec, uint32_t *log_id, uint32_t *log_len, size_t log_cap)
{
    SKL_TRY(check_params(s, s, p));
    SKL_TRY(ctx_bind(ctx));
    const bool coreacc = p->dist_type == SKL_DIST_COREACC;
    if (!h_key || !h_id || !h_len || !thr || (coreacc && !h_d1)) return fail(SKL_ERR_INVALID_ARG, "heap state pointers are null");
    const size_t n = s->n;
    if (n < 2 || knn == 0 || knn > n - 1) return fail(SKL_ERR_INVALID_ARG, "knn=%zu must be in [1, %zu]", knn, n ? n - 1 : 0);
    if (knn > (size_t)REFHEAP_LDS_MAX) return fail(SKL_ERR_INVALID_ARG, "the travelling heaps live in LDS while they are fed: knn=%zu exceeds %u", knn, REFHEAP_LDS_MAX);
    if (band_rows == 0 || band * band_rows >= n || col_lo > col_hi || col_hi > n) return fail(SKL_ERR_INVALID_ARG, "row band / column window out of range");
    // (a window that starts inside a band would give that band's rows turned candidates on this window's owner which the heaps
    // arriving from the upstream participant then overwrite: candidates silently lost)
    if (col_lo % band_rows != 0 || (col_hi % band_rows != 0 && col_hi != n)) {
        return fail(SKL_ERR_INVALID_ARG, "column window [%zu, %zu) must be cut on band boundaries (multiples of band_rows = %zu; col_hi may equal n)", col_lo, col_hi, band_rows);
    }
    if (!knn_symmetric_ok(s, p)) return fail(SKL_ERR_INVALID_ARG, "no one-evaluation kNN for this configuration; shard rows with skl_self_dists_knn_rows");
    KnnState st;
    st.borrowed = true;
    st.h_key = h_key;
    st.h_id = h_id;
    st.h_d1 = h_d1;
    st.h_len = h_len;
    st.thr = thr;
    st.log_rec = log_rec;
    st.log_id = log_id;
    st.log_len = log_len;
    st.log_cap = (uint32_t)log_cap;
    const std::vector<uint32_t> one{(uint32_t)band};
    // (the counters run on over the bands of a window: skl_ctx_knn_prune_stats reports everything since the last kNN
    // call of another kind -- no read-back, no reset here: this call must not stall the hand-over of the heaps)
    KnnCross form;   // (the symmetric form)
    form.lists_hold_knn = band >= 1 && band_rows >= knn;   // (bands 0 .. band - 1 of this window, or the windows before it, have fed every list)
    return knn_symmetric_bands(ctx, s, p, knn, band_rows, one, false, st, col_lo, col_hi, form);
}

// Empty heaps (h_len = 0, thr = "not full") for rows [row_begin, row_end) of caller-owned state arrays.
extern "C" int skl_self_dists_knn_window(skl_ctx *ctx, const skl_sketches *s, const skl_dist_params *p, size_t knn,
                                         size_t band_rows, size_t band, size_t col_lo, size_t col_hi,
                                         float *h_key, uint32_t *h_id, float *h_d1, uint32_t *h_len, uint32_t *thr)
{
    return knn_window_impl(ctx, s, p, knn, band_rows, band, col_lo, col_hi, h_key, h_id, h_d1, h_len, thr, nullptr, nullptr, nullptr, 0);
}

// DECOUPLED COLUMN WINDOWS (round 6).  The travelling heaps make participant r wait for r - 1.  A heap that starts EMPTY on a
// window takes a superset of what the row's true heap -- the one that has already met every earlier window -- would take
// there (its maximum is never lower, and push_heap's test is `key < maximum`, mod.rs:41-48), so every participant can run its
// window against empty heaps at once and LOG what they take, in order; the row's true list is then the replay of the logs in
// window order (skl_knn_heaps_replay): a candidate missing from a log was refused by a heap with a higher maximum, so the true
// heap refuses it too, and the logged ones reach it in the order the reference would have shown them.
extern "C" int skl_self_dists_knn_window_logged(skl_ctx *ctx, const skl_sketches *s, const skl_dist_params *p, size_t knn,
                                                size_t band_rows, size_t band, size_t col_lo, size_t col_hi,
                                                float *h_key, uint32_t *h_id, float *h_d1, uint32_t *h_len, uint32_t *thr,
                                                float *log_rec, uint32_t *log_id, uint32_t *log_len, size_t log_cap)
{
    if (!log_rec || !log_id || !log_len || log_cap == 0 || log_cap >= (1ull << 31)) return fail(SKL_ERR_INVALID_ARG, "accept-log pointers / capacity");
    return knn_window_impl(ctx, s, p, knn, band_rows, band, col_lo, col_hi, h_key, h_id, h_d1, h_len, thr, log_rec, log_id, log_len, log_cap);
}

extern "C" int skl_knn_heaps_replay(skl_ctx *ctx, size_t rows, size_t knn, int coreacc, const float *log_rec, const uint32_t *log_id,
                                    const uint32_t *log_len, size_t log_cap, float *h_key, uint32_t *h_id, float *h_d1, uint32_t *h_len,
                                    uint32_t *thr)
{
    SKL_TRY(ctx_bind(ctx));
    if (rows == 0) return SKL_OK;
    if (!log_rec || !log_id || !log_len || !h_key || !h_id || !h_len || !thr || (coreacc && !h_d1)) return fail(SKL_ERR_INVALID_ARG, "null argument");
    if (knn == 0 || knn > (size_t)REFHEAP_LDS_MAX || log_cap == 0 || log_cap >= (1ull << 31) || rows >= (1ull << 31)) return fail(SKL_ERR_INVALID_ARG, "knn / capacity / rows out of range");
    RefHeapMergeArgs m;
    memset(&m, 0, sizeof m);
    m.keys = log_rec;
    m.stride2 = coreacc ? 2u : 1u;
    m.key_stride = (uint64_t)log_cap * m.stride2;
    m.rows = (uint32_t)rows;
    m.cols = (uint32_t)log_cap;
    m.self_id_base = 0xFFFFFFFFu;   // (a log never holds the row itself)
    m.knn = (uint32_t)knn;
    m.h_key = h_key;
    m.h_id = h_id;
    m.h_d1 = h_d1;
    m.h_len = h_len;
    m.thr = thr;
    m.cand_ids = log_id;
    m.row_cols = log_len;
    m.force_workgroup_form = ctx->knobs.refheap_wave ? 0u : 1u;
    HIP_TRY(launch_refheap_merge(m, ctx->stream));
    return SKL_OK;
}

extern "C" int skl_knn_heaps_clear(skl_ctx *ctx, size_t row_begin, size_t row_end, uint32_t *h_len, uint32_t *thr)
{
    SKL_TRY(ctx_bind(ctx));
    if (!h_len || !thr || row_begin > row_end) return fail(SKL_ERR_INVALID_ARG, "bad heap range");
    if (row_begin == row_end) return SKL_OK;
    HIP_TRY(hipMemsetAsync(h_len + row_begin, 0, (row_end - row_begin) * sizeof(uint32_t), ctx->stream));
    HIP_TRY(hipMemsetAsync(thr + row_begin, 0xFF, (row_end - row_begin) * sizeof(uint32_t), ctx->stream));
    return SKL_OK;
}

extern "C" int skl_device_malloc(skl_ctx *ctx, size_t bytes, void **out)
{
    SKL_TRY(ctx_bind(ctx));
    if (!out) return fail(SKL_ERR_INVALID_ARG, "out is null");
    *out = nullptr;
    HIP_TRY(hipMalloc(out, bytes ? bytes : 1));
    return SKL_OK;
}

extern "C" int skl_device_free(skl_ctx *ctx, void *ptr)
{
    SKL_TRY(ctx_bind(ctx));
    if (!ptr) return SKL_OK;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipFree(ptr));
    return SKL_OK;
}

extern "C" int skl_device_memcpy(skl_ctx *ctx, void *dst, const void *src, size_t bytes, int to_device)
{
    SKL_TRY(ctx_bind(ctx));
    if (bytes == 0) return SKL_OK;
    if (!dst || !src) return fail(SKL_ERR_INVALID_ARG, "null pointer");
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return SKL_OK;
}

extern "C" int skl_ctx_get_knn_ties(const skl_ctx *ctx) { return ctx ? ctx->knn_ties : SKL_KNN_TIES_REFERENCE; }

// into_sorted_vec of `rows` heaps (arrays pointing at the first of them) -> the public output form (device pointers).
extern "C" int skl_knn_heaps_finalize(skl_ctx *ctx, size_t rows, size_t knn, const float *h_key, const uint32_t *h_id, const float *h_d1,
                                      const uint32_t *h_len, int ani, uint64_t *out_idx, float *out_d0, float *out_d1)
{
    SKL_TRY(ctx_bind(ctx));
    if (!h_key || !h_id || !h_len || !out_idx || !out_d0 || (h_d1 && !out_d1)) return fail(SKL_ERR_INVALID_ARG, "null argument");
    if (knn == 0 || knn > (size_t)REFHEAP_LDS_MAX) return fail(SKL_ERR_INVALID_ARG, "knn out of range");
    HIP_TRY(launch_refheap_finalize(h_key, h_id, h_d1, h_len, (uint32_t)rows, (uint32_t)knn, (!h_d1 && ani) ? 1 : 0, out_idx, out_d0, out_d1, ctx->stream));
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
    DevBuf in_key, in_idx, in_d1, tmp_key[2], tmp_idx[2], tmp_d1[2], o_idx, o_d0, o_d1;
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
