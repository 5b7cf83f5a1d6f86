/*
 * sketchlib_dist.h -- C ABI of the MI355X (gfx950) pairwise-distance engine.
 *
 * This is the drop-in boundary for bacpop/sketchlib.rust's `distances` module:
 * each entry point names the reference function it replaces (file:line relative
 * to the reference tree, v0.3.0).  Signatures use plain pointers and sizes only,
 * so the same library binds from Rust (`extern "C"`), C++
 * (sketchlib.rust_amd/csrc/host) and Python ctypes (sketchlib.rust_amd/capi.py).
 *
 * Sketch bins cross the boundary in the reference's own layout -- the `.skd`
 * byte order, MultiSketch::sketch_bins (src/sketch/multisketch.rs:36-40,
 * 213-219): little-endian u64,
 *     word(sample, k_idx, chunk, plane) =
 *         sample*sample_stride + k_idx*kmer_stride + chunk*14 + plane
 *     kmer_stride = sketchsize64*14 (BBITS, src/sketch/mod.rs:34),
 *     sample_stride = kmer_stride*nk.
 * Any device-side re-layout is internal to the library.
 *
 * Error handling: every function returns SKL_OK (0) or a positive SKL_ERR_*;
 * skl_last_error() returns a thread-local message.  Where the reference
 * panics (jaccard.rs:70-72, mod.rs:318-323) the ABI returns the matching code
 * and the same message text.  There is no CPU fallback: without a usable GPU
 * every compute entry point fails with SKL_ERR_NO_DEVICE.
 *
 * LIMITS (one place; each is checked and reported with SKL_ERR_INVALID_ARG / SKL_ERR_OOM, never silently):
 *   - samples per slab: < 2^31; the library keeps TWO copies of every slab in HBM (row layout + lane layout,
 *     2 x n x nk x sketchsize64 x 112 B): ~7 M genomes per 288 GB GPU at sketchsize64 = 32, nk = 5.
 *   - sketchsize64: < 2^25.  Up to 1 023 (65 472 bins) a k-mer length's counts live in u16 fields; larger sketches
 *     (the reference's "100000-1000000 for SNP level resolution", lib.rs:41-42) take the same kernel, walked in
 *     segments of 1 016 chunks; their core/accessory distances go through a bin-match count scratch of at most
 *     4 GiB per launch (bands of rows).  nk x sketchsize64 < 1 198 372 for the fast kernel (32-bit row offsets).
 *   - k-mer lengths: < 2^16; the fused core/accessory kernel takes up to 6, more go counts + epilogue.
 *   - knn: 1 <= knn <= candidates.  Up to 2 048 neighbours the running lists live in LDS (one-evaluation self kNN,
 *     skl_self_dists_knn_partial, skl_knn_merge_states); longer lists go row by row through global memory.
 *   - skl_self_dists_knn_shared_bins: at most skl_shared_bins_max_samples() = 1 294 336 samples per call.
 *   - timing: at most 4 096 bracketed pair-kernel launches between two skl_ctx_timing_reset();
 *     skl_clock_sampler_start: interval_us x max_samples <= 10 s.
 */
#ifndef SKETCHLIB_DIST_H
#define SKETCHLIB_DIST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SKL_ABI_VERSION 1

/* ---- status codes ---- */
#define SKL_OK 0
#define SKL_ERR_INVALID_ARG 1
#define SKL_ERR_NO_DEVICE 2   /* no HIP device / wrong architecture */
#define SKL_ERR_HIP 3         /* a HIP runtime call failed */
#define SKL_ERR_OOM 4
#define SKL_ERR_KMER_COUNT 5  /* "Need at least two k-mer lengths ..." jaccard.rs:70-72 */
#define SKL_ERR_EMPTY_DB 6    /* "... database has no loaded samples" mod.rs:318-323 */
#define SKL_ERR_KMER_NOT_FOUND 7 /* "K-mer size {k} not found in file" mod.rs:28-30 */
#define SKL_ERR_INCOMPATIBLE 8   /* ref/query differ in k-mers or sketch size */

/* DistType (src/distances/distance_matrix.rs:55-61) */
#define SKL_DIST_COREACC 0
#define SKL_DIST_JACCARD 1

typedef struct skl_ctx skl_ctx;           /* one per (host thread, device) */
typedef struct skl_sketches skl_sketches; /* device-resident MultiSketch bins */

/* Parameters shared by all distance calls.  Mirrors DistType::Jaccard(k_idx, k,
 * ani) / DistType::CoreAcc plus the completeness_cutoff argument every driver in
 * src/distances/mod.rs takes. */
typedef struct {
    int32_t dist_type;           /* SKL_DIST_COREACC | SKL_DIST_JACCARD */
    int32_t ani;                 /* Jaccard only: report ANI (jaccard.rs:49-51) */
    uint64_t k_idx;              /* Jaccard only: index into the k-mer list */
    double completeness_cutoff;  /* cli.rs:229 default 0.64 */
} skl_dist_params;

const char *skl_last_error(void);
int skl_abi_version(void);
/* Number of visible gfx950 devices (0 if none; never fails). */
int skl_device_count(void);

/* ---- context ---- */
int skl_ctx_create(int device, skl_ctx **out);
int skl_ctx_destroy(skl_ctx *ctx);
/* Run all subsequent work of this context on a caller-owned hipStream_t
 * (e.g. torch.cuda.current_stream().cuda_stream).  NULL restores the context's
 * own stream. */
int skl_ctx_set_stream(skl_ctx *ctx, void *hip_stream);
/* Run on the device's (legacy) default stream -- hipStream_t 0, which is what
 * torch.cuda.current_stream().cuda_stream is unless the caller entered a stream context.
 * The context's own stream is non-blocking, i.e. NOT ordered with work queued on the default
 * stream; a caller that fills or reads buffers with default-stream work and does not
 * synchronise explicitly wants this. */
int skl_ctx_use_default_stream(skl_ctx *ctx);
int skl_ctx_synchronize(skl_ctx *ctx);
/* The library's few environment switches (INTEGRATION.md "Building") are read when a context is
 * created and never on the launch path; this re-reads them for an existing context (tests and
 * benchmarks that compare two drivers inside one process). */
int skl_ctx_reload_env(skl_ctx *ctx);
/* Pair-kernel timing -- a diagnostic, OFF by default (an event record is a barrier packet on the queue: two per
 * launch cost a sub-millisecond launch ~5 us).  enable(every): every = 0 turns it off, N >= 1 brackets every N-th
 * launch of the pair kernel (dense / binmatch / knn calls; also the candidate-list and the sketching kernel) with HIP
 * events recorded on the context's stream (the environment variable SKL_TIMING_EVERY sets the initial value).
 * reset() forgets the recorded events; kernel_ms() synchronises and returns the summed device time and the number of
 * launches recorded since the last reset (at most 4096 are kept). */
int skl_ctx_timing_enable(skl_ctx *ctx, int every);
int skl_ctx_timing_reset(skl_ctx *ctx);
int skl_ctx_kernel_ms(skl_ctx *ctx, float *total_ms, int *n_launches);
/* Name (with tile shape) of the pair kernel the last call dispatched, e.g.
 * "pair_kernel_ksplit<R=4>" -- for benchmark reports.  Valid until the next call. */
const char *skl_ctx_last_kernel(skl_ctx *ctx);

/* Diagnostic (no reference counterpart): the shader clock the chip holds while the context's kernels
 * run.  start() launches a one-wave sampler on a stream of its own (it stamps the shader-clock counter
 * against the constant 100 MHz counter every interval_us, at most max_samples times, then ends by
 * itself); run the work to be measured, wait for it with skl_ctx_synchronize() -- NOT a device-wide
 * synchronisation, which would wait for the sampler -- and call stop(): clock readings in GHz, one
 * per interval (median, 10th / 90th percentile, and the mean over the whole window). */
int skl_clock_sampler_start(skl_ctx *ctx, uint32_t interval_us, uint32_t max_samples);
int skl_clock_sampler_stop(skl_ctx *ctx, double *ghz_median, double *ghz_p10, double *ghz_p90, double *ghz_mean,
                           int *n_intervals);

/* The reference takes ln J with Rust's f64::ln = the platform libm's log() (jaccard.rs:51,88), and
 * the core/accessory regression amplifies its last bit without bound on flat fits, so the
 * device evaluates a restatement of glibc's log() (csrc/glibc_log.hpp) whenever the argument is
 * not a function of the bin-match count alone (completeness correction).  skl_log_variant(): the
 * form that reproduces THIS host's log() bit for bit on a probe set: 0 = glibc 2.35 x86-64 FMA
 * form, 1 = its SSE2 form, -1 = neither (form 0 is used; flat-fit pairs may then differ from a
 * CPU run on this host).  skl_device_log(): y[i] = that logarithm of x[i], evaluated on the
 * device (host pointers) -- for tests. */
int skl_log_variant(void);
/* Conditions a caller may want to tell its user about (a null context is accepted and reports the flags that are
 * properties of the host process).  SKL_CTX_FLAG_LOG_UNMATCHED: skl_log_variant() == -1 -- this host's libm
 * log() is neither form the device can reproduce, so completeness-corrected core distances of FLAT fits
 * (jaccard.rs:120-133: the same bin-match count at every k-mer length) may come out 0 where a CPU run of
 * the reference on this host gives 1, or the reverse.  Everything else is unaffected. */
#define SKL_CTX_FLAG_LOG_UNMATCHED 1u
/* SKL_CTX_FLAG_NOT_SPX (needs a context): the device does not report the 256 compute units of an unpartitioned (SPX)
 * MI355X.  Results do not depend on it.  The tile order follows the device (workgroups are dealt to CUs / 32 XCDs, the
 * resident-round sizes use the CU count); the launch sizes at which tile shapes and chunk slicing switch were tuned on
 * SPX only (DESIGN.md "Topology assumptions"). */
#define SKL_CTX_FLAG_NOT_SPX 2u
unsigned skl_ctx_flags(const skl_ctx *ctx);
int skl_device_log(skl_ctx *ctx, const double *x_host, size_t n, double *out_host);

/* ---- sketch slabs: MultiSketch::read_sketch_data / get_sketch_slice
 *      (src/sketch/multisketch.rs:167-219) ---- */
/* bins: n_samples*nk*sketchsize64*14 u64 in the reference layout.  `on_device`
 * != 0 means `bins` is already a device pointer on the context's device. */
int skl_sketches_create(skl_ctx *ctx, const uint64_t *bins, int on_device, size_t n_samples,
                        size_t nk, const size_t *kmers, size_t sketchsize64, skl_sketches **out);
/* Option<&Vec<f64>> completeness vector (src/io.rs:240-324); NULL == None. */
int skl_sketches_set_completeness(skl_sketches *s, const double *completeness_host);
int skl_sketches_destroy(skl_sketches *s);
size_t skl_sketches_n_samples(const skl_sketches *s);

/* set_k (src/distances/mod.rs:25-37): kmer == 0 means Option::None -> CoreAcc. */
int skl_set_k(const skl_sketches *s, size_t kmer, int ani, double completeness_cutoff,
              skl_dist_params *out);

/* ---- dense distances ----
 * `out_on_device` != 0: `out` is a device pointer and the call only enqueues work
 * on the context's stream; == 0: `out` is host memory and the call returns after
 * the copy back. */

/* self_dists_all (src/distances/mod.rs:58-130).  out: n(n-1)/2 * ncols f32,
 * condensed upper triangle, ncols = 2 (core, acc interleaved) or 1. */
int skl_self_dists_all(skl_ctx *ctx, const skl_sketches *s, const skl_dist_params *p, float *out,
                       int out_on_device);
/* Rows [row_begin, row_end) of the same condensed matrix -- the slice
 * [square_to_condensed(row_begin,row_begin+1), square_to_condensed(row_end,row_end+1))
 * -- written to out[0..]; the multi-GPU pair-block partition. */
int skl_self_dists_rows(skl_ctx *ctx, const skl_sketches *s, const skl_dist_params *p,
                        size_t row_begin, size_t row_end, float *out, int out_on_device);
/* cross_dists_all (src/distances/mod.rs:227-297).  out: n_ref*n_query*ncols,
 * index (i_ref*n_query + j_query)*ncols. */
int skl_cross_dists_all(skl_ctx *ctx, const skl_sketches *ref, const skl_sketches *query,
                        const skl_dist_params *p, float *out, int out_on_device);
/* Reference rows [ref_begin, ref_end) of the same matrix. */
int skl_cross_dists_rows(skl_ctx *ctx, const skl_sketches *ref, const skl_sketches *query,
                         const skl_dist_params *p, size_t ref_begin, size_t ref_end, float *out,
                         int out_on_device);

/* ---- sparse k-nearest-neighbour distances ----
 * Output rows hold `knn` items sorted ascending by key, where key = d0 for Jaccard / core for
 * CoreAcc, or 1-ANI for ANI (mod.rs:173-176).  out_d1 is written for CoreAcc only and may be NULL
 * otherwise.  knn is bounded only by the number of candidates (lib.rs:379-382, mod.rs:325): up to
 * 2 048 neighbours the running lists live in LDS, longer ones go through global memory.
 *
 * Equal keys (common: every genome without knn relatives ties at 1.0, and single-k Jaccard values are
 * quantised) -- skl_ctx_set_knn_ties() chooses between two rules for every kNN call of the context:
 *   SKL_KNN_TIES_REFERENCE (default)  the list the reference's self_dists_knn / cross_dists_knn return: candidates j
 *                                     ascending through push_heap (mod.rs:41-48: strict `<` against the heap's
 *                                     maximum) into std::collections::BinaryHeap, then into_sorted_vec (mod.rs:156-191,
 *                                     :335-391) -- which of several equal keys survive, and their order, follow from
 *                                     the heap's history, which the device replays.  skl_self_dists_knn over the whole
 *                                     matrix still evaluates every pair ONCE: a row's heap lives in device memory
 *                                     between the row bands and is fed its candidates in ascending id (rows above its
 *                                     band reach it through the band's turned copy); row ranges, cross kNN and lists
 *                                     longer than 2 048 go row by row (every pair of those rows, as the reference does).
 *   SKL_KNN_TIES_CANONICAL            smallest (key, neighbour index) first: a property of the data alone.  The
 *                                     distance multiset of every row equals the reference's. */
#define SKL_KNN_TIES_CANONICAL 0
#define SKL_KNN_TIES_REFERENCE 1
int skl_ctx_set_knn_ties(skl_ctx *ctx, int mode);

/* Diagnostic (no reference counterpart): the early break of the core/accessory calls.  core_acc_dist leaves its loop over
 * the k-mer lengths at the first one whose ln J lies below the tolerance -- a pair that shares no more bins than chance,
 * expected_samebits (src/distances/jaccard.rs:26-31, :89-91) -- and a fit over fewer than three lengths is (1, 1) (:117).
 * With three to eight lengths the dense calls therefore count only the first two to four for the pairs of a BLOCK of the
 * pair space and complete the pairs still in the running afterwards, grouped by row (csrc/epilogue.hip) -- where a sample
 * of the block's pairs, taken the first time a slab meets a column slab, says that pays (between unrelated genomes ~1 % of
 * the pairs stay in the running at 4 096 bins, 0.2 % at 2 048; between close relatives all of them: such blocks count every
 * length as before).  With or without a completeness correction, any sketch size.  Same (core, acc) bit for bit either way.
 * pairs: pairs of the early-break launches since the context was made; completed_one_by_one: those among them that were
 * still in the running (counter wraps at 2^32).  Either argument may be null.  (The A/B build reads SKL_EARLY_BREAK=0: off.) */
int skl_ctx_early_break_stats(skl_ctx *ctx, uint64_t *pairs, uint64_t *completed_one_by_one);
/* ... and what the last dense core/accessory call of the context decided: the pair space is cut into blk_rows x blk_cols blocks
 * of (row sample >> shift_rows, column sample >> shift_cols); pooled_lengths = the k-mer lengths counted when all blocks agree
 * (0: every length, no early break); mixed = 1 when they do not, and block_lengths[r * blk_cols + c] (capacity >= blk_rows x
 * blk_cols bytes) then holds each block's count (nk: every length).  Any pointer may be null.  Reference behaviour matched:
 * the per-pair `break` of jaccard.rs:89-91 -- the result never depends on the decision, only the time does. */
int skl_ctx_early_break_blocks(skl_ctx *ctx, uint32_t *blk_rows, uint32_t *blk_cols, uint32_t *shift_rows, uint32_t *shift_cols,
                               int *pooled_lengths, int *mixed, uint8_t *block_lengths, size_t capacity);

/* Diagnostic (no reference counterpart): tile pruning of the last skl_self_dists_knn / _partial call of the context.  The
 * whole-matrix self kNN with single-k keys (Jaccard / ANI, no completeness correction) leaves a 32 x 128 tile of the pair
 * space unfinished once every pair of it is, on the bins compared so far, already beyond both its samples' current knn-th
 * best (the key is monotone in the mismatch count).  The neighbour lists are those of the unpruned run in
 * either tie rule.  tiles: tiles of the launches that could prune; tiles_pruned: those left early -- most of them by a probe
 * over two of the fourteen planes before the walk begins (no stage walked), the rest at a stage boundary of the walk;
 * stages_per_tile: stage boundaries of a whole tile's walk (4 chunks each); stages_walked_in_pruned_tiles: how many of them
 * the pruned tiles had walked when they left (so the share of the pair space's full bin comparisons actually made is known);
 * tiles_sparse: tiles the probe could NOT dismiss but in which at most 4 rows held a pair still in the running -- typically one
 * relative among the tile's 4 096 pairs -- and whose walk was made for those rows only (every other pair of the tile gets the
 * worst key there is: it lies beyond both its samples' bounds, like a pruned tile's pairs).  Any argument may be null.  (The A/B
 * build of the library reads SKL_KNN_PRUNE=0: every tile walked, same lists; SKL_KNN_SPARSE=0: survivors walked whole.) */
int skl_ctx_knn_prune_stats(skl_ctx *ctx, uint64_t *tiles, uint64_t *tiles_pruned, uint64_t *stages_per_tile,
                            uint64_t *stages_walked_in_pruned_tiles, uint64_t *tiles_sparse);

/* self_dists_knn (src/distances/mod.rs:133-224); requires 1 <= knn < n. */
int skl_self_dists_knn(skl_ctx *ctx, const skl_sketches *s, const skl_dist_params *p, size_t knn,
                       uint64_t *out_idx, float *out_d0, float *out_d1, int out_on_device);
int skl_self_dists_knn_rows(skl_ctx *ctx, const skl_sketches *s, const skl_dist_params *p,
                            size_t knn, size_t row_begin, size_t row_end, uint64_t *out_idx,
                            float *out_d0, float *out_d1, int out_on_device);
/* self_dists_knn split over several devices with every pair evaluated ONCE (the reference
 * evaluates (i, j) and (j, i), mod.rs:148-171; distances are symmetric).  The rows are cut into
 * bands of band_rows samples (skl_knn_band_rows gives a size every participant agrees on); a
 * participant takes a subset of the bands (ascending band indices; deal them so that costs
 * balance: band b costs ~ n - b*band_rows) and gets back, for ALL n rows, the knn best
 * candidates seen in its share: state_key = order-preserving u32 image of the f32 key (empty
 * entries 0xFFFFFFFF), state_idx = neighbour, state_d1 = accessory distance (CoreAcc only, may
 * be NULL otherwise), each [n][knn], sorted ascending per row.  The participants then exchange
 * row shards of these states (an all-to-all: the one data-path collective of this library) and
 * skl_knn_merge_states turns the n_states partial states of a row shard -- stacked
 * [n_states][rows][knn] -- into the rows of skl_self_dists_knn's output.
 * Returns SKL_ERR_INVALID_ARG when the configuration has no one-evaluation form (CoreAcc with more than 6
 * k-mer lengths or sketchsize64 > 1023; knn > 2 048): shard rows with skl_self_dists_knn_rows then; and in the reference
 * tie order (partial heaps do not merge): use skl_self_dists_knn_window below. */
size_t skl_knn_band_rows(const skl_sketches *s, const skl_dist_params *p, size_t n_participants);
int skl_self_dists_knn_partial(skl_ctx *ctx, const skl_sketches *s, const skl_dist_params *p, size_t knn,
                               size_t band_rows, const uint32_t *bands, size_t n_bands,
                               uint32_t *state_key, uint32_t *state_idx, float *state_d1, int out_on_device);
int skl_knn_merge_states(skl_ctx *ctx, size_t n_states, size_t rows, size_t knn, const uint32_t *state_key,
                         const uint32_t *state_idx, const float *state_d1, int states_on_device, int ani,
                         uint64_t *out_idx, float *out_d0, float *out_d1, int out_on_device);

/* self_dists_knn in the REFERENCE's tie order over several devices, every pair evaluated once: the column-window pipeline.
 * A row's BinaryHeap must meet its candidates in ascending id (mod.rs:156-181), so partial heaps of disjoint candidate sets
 * cannot be merged -- but a heap can travel.  Participant r owns the column window [lo_r, hi_r) (windows ascending with r;
 * cut them at n * sqrt(r / R) so that the pair counts balance, ROUNDED TO BAND BOUNDARIES: col_lo and col_hi must be multiples
 * of band_rows -- col_hi may also be n -- or the call fails with SKL_ERR_INVALID_ARG: a window starting inside a band would
 * lose candidates) and, for every row band b = rows [b * band_rows, ...) that
 * starts below hi_r, in ascending order, calls skl_self_dists_knn_window: the band's rows take the columns
 * [max(band start, lo_r), hi_r) as candidates, the window's rows below the band take the band's samples.  Before a band
 * whose rows lie below lo_r, participant r receives those rows' heaps from participant r - 1 (which has finished that band);
 * after any band it sends the band's rows on to r + 1; the last participant ends up with every row's final heap
 * (skl_knn_heaps_finalize).  Rows of a participant's own window start empty (skl_knn_heaps_clear).  The heap arrays are the
 * caller's (device memory): h_key f32 / h_id u32 / h_d1 f32 (CoreAcc only) [n][knn] in heap order, h_len u32 [n], thr u32 [n]
 * (order-preserving image of the heap's maximum once it is full, 0xFFFFFFFF before).  Tile pruning applies as in the
 * single-device call.  sketchlib.rust_amd/multi_gpu.py (self_knn_once_reference) is the driver over torch.distributed. */
int skl_self_dists_knn_window(skl_ctx *ctx, const skl_sketches *s, const skl_dist_params *p, size_t knn,
                              size_t band_rows, size_t band, size_t col_lo, size_t col_hi,
                              float *h_key, uint32_t *h_id, float *h_d1, uint32_t *h_len, uint32_t *thr);
int skl_knn_heaps_clear(skl_ctx *ctx, size_t row_begin, size_t row_end, uint32_t *h_len, uint32_t *thr);
/* The same pipeline DECOUPLED (round 6): no participant waits for another.  A BinaryHeap that starts EMPTY on a window takes a
 * superset of what the row's true heap (the one that has met every earlier window) would take there -- its maximum is never
 * lower, and push_heap pushes on `key < maximum` (mod.rs:41-48).  So every participant clears ALL its heaps, runs its window
 * band by band with skl_self_dists_knn_window_logged, which also appends every candidate a heap takes to the row's ACCEPT LOG,
 * in order: log_rec [n][log_cap] records of 1 float (the key) or 2 (CoreAcc: key, second distance), log_id [n][log_cap],
 * log_len [n] (zeroed by the caller; it counts past log_cap -- an overflowed row's log is useless and the caller falls back
 * to the travelling heaps; random candidate order takes ~knn (1 + ln(window / knn)) entries).  The row's true list is then
 * skl_knn_heaps_replay of the participants' logs for that row in WINDOW ORDER into one empty heap (rows: the pointers are at
 * the first of them; logs of participants that own no column of a row's candidates are empty), then skl_knn_heaps_finalize.
 * Exactness: a candidate missing from a log was refused by a heap whose maximum was at least the true heap's.  What it
 * costs: tile pruning sees only the window's own relatives (scripts/knn_pipeline_model.py). */
int skl_self_dists_knn_window_logged(skl_ctx *ctx, const skl_sketches *s, const skl_dist_params *p, size_t knn,
                                     size_t band_rows, size_t band, size_t col_lo, size_t col_hi,
                                     float *h_key, uint32_t *h_id, float *h_d1, uint32_t *h_len, uint32_t *thr,
                                     float *log_rec, uint32_t *log_id, uint32_t *log_len, size_t log_cap);
int skl_knn_heaps_replay(skl_ctx *ctx, size_t rows, size_t knn, int coreacc, const float *log_rec, const uint32_t *log_id,
                         const uint32_t *log_len, size_t log_cap, float *h_key, uint32_t *h_id, float *h_d1, uint32_t *h_len,
                         uint32_t *thr);
/* For callers without a device runtime of their own (the C++ host layer sees only this header): device memory on the
 * context's device for the heap arrays above, and copies between it and host memory (to_device != 0: host -> device),
 * ordered with the context's stream and complete on return. */
int skl_device_malloc(skl_ctx *ctx, size_t bytes, void **out);
int skl_device_free(skl_ctx *ctx, void *ptr);
int skl_device_memcpy(skl_ctx *ctx, void *dst, const void *src, size_t bytes, int to_device);
int skl_ctx_get_knn_ties(const skl_ctx *ctx);   /* SKL_KNN_TIES_* of the context */
/* into_sorted_vec of `rows` heaps (the arrays point at the first of them) -> rows of skl_self_dists_knn's output; device pointers. */
int skl_knn_heaps_finalize(skl_ctx *ctx, size_t rows, size_t knn, const float *h_key, const uint32_t *h_id, const float *h_d1,
                           const uint32_t *h_len, int ani, uint64_t *out_idx, float *out_d0, float *out_d1);

/* Assembling a dense matrix that several devices of ONE process computed in row bands (skl_self_dists_rows /
 * skl_cross_dists_rows with device outputs) on the first context's device, over xGMI: "the N x N pair space block-partitioned
 * across the GPUs of one node with a RCCL gather to assemble the output matrix" (the reference has one address space and no
 * counterpart).  ctxs[d] made band_dev[d] (band_bytes[d] bytes, device memory of ITS device, produced on ITS stream); the bands
 * land at dst_dev_root + dst_offsets[d] on ctxs[0]'s device: the root's own band by a copy inside its HBM, the others by
 * grouped ncclSend / ncclRecv in messages of at most 1 GiB, each send behind the kernels of its context's stream, the receives
 * on the root's stream.  Asynchronous: skl_ctx_synchronize(ctxs[0]) completes it.  One context per device (RCCL takes one rank
 * per device: a device listed twice is SKL_ERR_INVALID_ARG).  librccl.so.1 is looked up at run time; the communicators of a
 * device list are made once per process.  loopback_through_rccl != 0 with ONE context sends that band to itself through RCCL
 * (what a one-GPU box can test of the transport).  The torch.distributed rendering (one process per GPU) is
 * sketchlib.rust_amd/multi_gpu.py. */
int skl_gather_bands_rccl(skl_ctx *const *ctxs, size_t n_ctx, const void *const *band_dev, const size_t *band_bytes,
                          void *dst_dev_root, const size_t *dst_offsets, int loopback_through_rccl);

/* The same with the candidate lists built on the device as well: skq holds the index sketch
 * (u16 bins, `.skq` layout [sample][sketch_size]) of every sample of `s`, row i = sample i (the
 * caller applies the .ski -> .skd order map); candidates of i = samples sharing at least one
 * bin value at the same position, i excluded (Inverted::any_shared_bins, inverted.rs:259-268).
 * At most 1 294 336 samples per call (an n-bit bitmap per row lives in LDS).
 * out_n_candidates (nullable): total number of candidate pairs found. */
size_t skl_shared_bins_max_samples(void);
int skl_self_dists_knn_shared_bins(skl_ctx *ctx, const skl_sketches *s, const skl_dist_params *p,
                                   size_t knn, const uint16_t *skq, size_t sketch_size,
                                   uint64_t *out_idx, float *out_d0, uint64_t *out_n_candidates);

/* GPU sketching: bin minima of `canonical ntHash % SIGN_MOD` over every
 * valid k-mer of DNA samples -- Sketch::get_signs_no_densify (src/sketch/mod.rs:156-176) over
 * NtHashIterator (src/hashing/nthash_iterator.rs:325-523), all samples and k-mer lengths of a
 * batch in one launch.  `codes`: 2-bit base codes ((byte >> 1) & 3, hashing/mod.rs:82-85), one
 * byte per valid base, samples concatenated; sample s owns codes[code_begin[s] .. code_begin[s+1]).
 * `offsets`: for each sample the ascending positions (in its own code coordinates) of every
 * invalid base and record end (NtHashIterator::add_dna_seq, nthash_iterator.rs:205-251); a
 * window [p, p+k) is hashed iff no offset o has p < o < p+k.  out_signs: [n_samples][nk][num_bins]
 * u64, u64::MAX for an empty bin; densify_bin and the 14-plane transpose are the caller's
 * (they touch num_bins words, not the genome).  Host pointers. */
int skl_sketch_signs(skl_ctx *ctx, const uint8_t *codes, const uint64_t *code_begin,
                     const uint64_t *offsets, const uint64_t *offset_begin, size_t n_samples,
                     const size_t *kmers, size_t nk, uint64_t num_bins, int rc, uint64_t *out_signs);
/* The same with the bases already at 2 bits each -- what crosses PCIe either way (the one-byte form above is packed by the
 * library on host threads): `packed` holds, sample after sample, ceil(len / 16) little-endian u32 words per sample, code c of
 * the sample at bits 2 (c % 16) of its word c / 16, a sample's last word zero-padded; code_begin still counts CODES
 * (sample s has code_begin[s + 1] - code_begin[s] of them).  A caller that parses sequence files packs as it goes
 * (csrc/host/sketch_gpu.cpp).  Either form uploads batch i + 1 of the samples under the kernel of batch i. */
int skl_sketch_signs_packed(skl_ctx *ctx, const uint32_t *packed, const uint64_t *code_begin,
                            const uint64_t *offsets, const uint64_t *offset_begin, size_t n_samples,
                            const size_t *kmers, size_t nk, uint64_t num_bins, int rc, uint64_t *out_signs);

/* Candidate-list form of skl_self_dists_knn: the device half of self_dists_knn_precluster
 * (src/distances/mod.rs:399-553).  Row i is compared only with the samples
 * cand[row_offsets[i] .. row_offsets[i+1]) (ascending sample ids, i itself excluded) -- what
 * Inverted::any_shared_bins (src/inverted.rs:259-268) returns for it.  Single-k Jaccard / ANI
 * only (the reference's CoreAcc arm is unimplemented!(), mod.rs:549-551).  Output as
 * skl_self_dists_knn, rows of fewer than knn candidates padded with (i, 1.0) (mod.rs:535-546).
 * All pointers are host pointers. */
int skl_self_dists_knn_candidates(skl_ctx *ctx, const skl_sketches *s, const skl_dist_params *p,
                                  size_t knn, const uint64_t *row_offsets, const uint32_t *cand,
                                  uint64_t *out_idx, float *out_d0);

/* cross_dists_knn (src/distances/mod.rs:306-395): rows = queries, neighbours
 * index refs; knn must already be clamped to <= n_ref (mod.rs:325).
 * Against 131 072 references or more (single-k keys, no completeness correction) the references reach a query in ascending
 * panels of columns and the tiles of the later panels are pruned against the queries' running lists, as in the self kNN
 * (skl_ctx_knn_prune_stats reports them): same lists in either tie rule, 16 384 x 1 M top-50 in 0.16 s instead of 0.36 s. */
int skl_cross_dists_knn(skl_ctx *ctx, const skl_sketches *ref, const skl_sketches *query,
                        const skl_dist_params *p, size_t knn, uint64_t *out_idx, float *out_d0,
                        float *out_d1, int out_on_device);
int skl_cross_dists_knn_rows(skl_ctx *ctx, const skl_sketches *ref, const skl_sketches *query,
                             const skl_dist_params *p, size_t knn, size_t query_begin,
                             size_t query_end, uint64_t *out_idx, float *out_d0, float *out_d1,
                             int out_on_device);

/* ---- raw bin-match counts (`samebits`, src/distances/jaccard.rs:15-25) ----
 * self: out[cond(i,j)*nk + k]; cross: out[(i_ref*n_query + j_query)*nk + k]. */
int skl_self_binmatch(skl_ctx *ctx, const skl_sketches *s, uint32_t *out, int out_on_device);
int skl_cross_binmatch(skl_ctx *ctx, const skl_sketches *ref, const skl_sketches *query,
                       uint32_t *out, int out_on_device);

/* ---- one-shot forms: host pointers in, host pointers out; the argument lists
 *      of distances::self_dists_all / cross_dists_all flattened ---- */
int skl_self_dists_all_host(const uint64_t *bins, size_t n_samples, size_t nk,
                            const size_t *kmers, size_t sketchsize64,
                            const skl_dist_params *p, const double *completeness, float *out);
int skl_cross_dists_all_host(const uint64_t *ref_bins, size_t n_ref, const uint64_t *query_bins,
                             size_t n_query, size_t nk, const size_t *kmers, size_t sketchsize64,
                             const skl_dist_params *p, const double *ref_completeness,
                             const double *query_completeness, float *out);

#ifdef __cplusplus
}
#endif
#endif /* SKETCHLIB_DIST_H */
