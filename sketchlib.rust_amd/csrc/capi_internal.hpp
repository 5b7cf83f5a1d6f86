// capi_internal.hpp -- shared between the translation units that implement
// include/sketchlib_dist.h (capi.cpp: contexts, slabs, dense calls; capi_knn.cpp: the kNN
// drivers; capi_aux.cpp: candidate lists and sketching).  Not part of the public boundary.
#pragma once

#include "../../include/sketchlib_dist.h"

#include <hip/hip_runtime.h>

#include <map>
#include <set>
#include <string>
#include <utility>
#include <vector>

#include "kernels.h"
#include "roctx_ranges.hpp"

#define SKL_INTERNAL __attribute__((visibility("hidden")))

// ---- error plumbing: status code + message for skl_last_error() ----
SKL_INTERNAL int fail(int code, const char *fmt, ...);

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

struct skl_sketches;

// Environment switches.  They are read ONCE, when a context is created (skl_ctx_create), never on
// the launch path.  The product library READS eight of them (capi.cpp read_knobs: timing cadence, topology,
// the knobs that force the banded / sliced / 32-row forms on small test inputs); every "A/B only, results
// identical" switch below keeps its default there and is read by the A/B build alone (-DSKL_AB), as are kernel
// selection, tile shapes and the timing-only ablations.
struct Knobs {
    long long timing_every = 0;       // SKL_TIMING_EVERY: bracket every N-th pair-kernel launch with events (0: none; skl_ctx_timing_enable overrides)
    long long sliced_max_pairs = -1;  // SKL_SLICED_MAX_PAIRS: core/acc launches below this run k-sliced (-1: default)
    long long knn_band_rows = 0;      // SKL_KNN_BAND_ROWS: force the band height of the kNN drivers (tests)
    int k_slices = 0;                 // SKL_K_SLICES: chunk slices per k of k-sliced core/acc launches (0: chosen per launch)
    int xcds = 0;                       // SKL_XCDS: XCDs the tile order assumes (0: from the device's CU count: 256 CUs = 8, a 32-CU partition = 1)
    int group_span = 2;                 // SKL_GROUP_SPAN: column groups whose tiles are numbered side by side (device_common.hpp lookup_tile_at)
    long long tile32_min = 8ll << 20;   // SKL_TILE32_MIN: pair x k evaluations from which launches use 32 x 128 tiles (-1: never, 0: always); 8 Mi since the k-sliced 32-row form holds 4 waves per SIMD (profiles/r03_ab_tile32_threshold.jsonl)
    bool mid_band = true;               // SKL_MID_BAND=0: no mid-band rule (32-row tiles + 2 slices of the last round at 0.5-1 x tile32_min evaluations; A/B only, results are identical)
    int tail_slices = 4;                // SKL_TAIL_SLICES: chunk slices per unit in the last, partial round of a k-sliced core/acc launch (0/1: off)
    bool half_tiles = true;             // SKL_HALF_TILES=0: 64-column blocks of a tile without a pair of the launch are walked anyway (A/B only, results are identical)
    bool round_priority = true;         // SKL_ROUND_PRIORITY=0: k-sliced workgroups of later rounds keep the default wave priority (A/B only, results are identical)
    long long tail_max_pct = 90;        // SKL_TAIL_MAX_PCT: ... for launches of up to this many estimated rounds of workgroups (in percent)
    bool knn_symmetric = true;        // SKL_KNN_SYMMETRIC=0: row-by-row self kNN
    bool knn_row_flags = true;        // SKL_KNN_ROW_FLAGS=0: the merge of the transposed band visits every row (A/B only, results are identical)
    bool knn_overlap = true;          // SKL_KNN_OVERLAP=0: top-k and pair kernel on one stream
    bool fuse_epilogue = false;       // A/B build, SKL_FUSE_EPILOGUE=1: the core/accessory epilogue of plain k-sliced launches inside the pair kernel (results identical; slower: profiles/r05_fused_epilogue.md)
    int early_break = 1;              // A/B build, SKL_EARLY_BREAK: 0 core/accessory launches count every k-mer length; 1 (default) the early break where a
                                      // sample of the pairs says it pays; 2..7 forced with that many lengths counted (tests).  Results identical.
    int eb_pipeline = -1;             // -1: the row bands of a large early-break call overlap (band i's epilogue beside band i + 1's counts kernel) where the lean epilogue runs in the flat order; A/B build, SKL_EB_PIPELINE=0 / 1 / 2: never / the old rule (general kernel too: from 3 % still in the running, or forced lengths; never with the blocked order) / wherever the lean kernel runs
    long long eb_pipeline_min = 64ll << 20;  // A/B build, SKL_EB_PIPELINE_MIN: pairs from which an early-break call is cut into overlapping row bands (tests force it low)
    int eb_blocked = -1;              // A/B build, SKL_EB_BLOCKED=0|1: the early break's epilogue walks the pairs in flat order / in 256 x 256 blocks per XCD (-1: by the size of the column slices)
    int eb_blk_row_shift = 10;        // A/B build, SKL_EB_BLK_ROW_SHIFT: rows per block (log2) of the blocked epilogue
    int knn_epi_blocked = 1;          // A/B build, SKL_KNN_EPI_BLOCKED=0 / 2: the kNN bands' early-break epilogue in row-major order / column-group-major per XCD whatever the view's width (default: from 16 384 columns; 2.52 -> 2.43 s at n = 300 000)
    bool eb_lean = true;              // A/B build, SKL_EB_LEAN=0: every early-break launch through the general epilogue kernel
    bool eb_ahead = true;             // A/B build, SKL_EB_AHEAD=0: completions one after the other, nothing requested ahead
    bool eb_lds_rows = true;          // A/B build, SKL_EB_LDS_ROWS=0: completions read the row sample's slice from memory, not from the workgroup's LDS copy
    bool counts_u16 = true;           // A/B build, SKL_COUNTS_U16=0: the counts scratch keeps u32 records
    bool epilogue_r5 = false;         // A/B build, SKL_EPILOGUE_R5=1: round 5's epilogue (alive pairs completed where they are found; timing)
    bool knn_sparse = true;           // A/B build, SKL_KNN_SPARSE=0: tiles that survive the probe are walked whole (results identical)
    long long knn_panel = 0;          // A/B build, SKL_KNN_PANEL: column-panel width of the row-by-row kNN forced (tests; 0: by size)
    bool knn_prune = true;            // SKL_KNN_PRUNE=0: the symmetric self kNN finishes every tile (A/B; results are identical)
    bool refheap_wave = true;         // SKL_REFHEAP_WAVE=0: the heap replays (one-shot and resumable) run one workgroup per row even for knn <= 256 (A/B only, results are identical)
    bool topk_stream = true;          // SKL_TOPK_STREAM=0: radix select instead of the streaming merge
    bool cand_symmetric = true;       // SKL_CAND_SYMMETRIC=0: evaluate symmetric candidate lists in full
    bool inline_prefix = true;        // SKL_INLINE_PREFIX=0: the tile lookup always searches the prefix table in global memory (A/B only, results are identical)
    bool cand_lanes = false;          // SKL_CAND_KERNEL=lanes: round 3's candidate-list kernel (lanes over the candidates; A/B only, results are identical)
    bool cand_row_order = true;       // SKL_CAND_ROW_ORDER=0: candidate-list rows dispatched in sample order, not by first candidate (A/B only, results are identical)
    bool sketch_global = false;       // SKL_SKETCH_KERNEL=global: the unstaged sketching kernel
#ifdef SKL_AB
    int kernel = 0;                   // SKL_KERNEL: 0 none, 3 ksplit, 4 kslice
    int kslice_shape = 0;             // SKL_KSLICE_SHAPE: 165 / 325 (shipped), 1651 / 1652 / 3254 / 3255 (their round-2/3 forms)
    int ksplit_rows = 0;              // SKL_KSPLIT_ROWS: 4 or 8
    int kslice_ablate = 0;            // SKL_KSLICE_ABLATE: timing only, outputs wrong by construction
#endif
};
SKL_INTERNAL Knobs read_knobs();

// EARLY BREAK of the core/accessory calls, as decided for one (row slab, column slab) pair (capi.cpp early_break_plan): how many
// k-mer lengths the pair kernel counts before the epilogue completes the pairs still in the running -- for the whole pair
// space (`lengths`; what the kNN drivers take) and, when its blocks of (row >> shift_r, column >> shift_c) sample ids
// disagree, per block.  Kept by the CONTEXT, keyed by the slabs' generation ids (never reused), not written through the
// caller's const slab.
struct EbPlan {
    uint64_t rows_gen = 0, cols_gen = 0;
    int self_mode = 0, knob = 1;        // (knob: SKL_EARLY_BREAK as of the decision -- the A/B build may change it between calls)
    double cutoff = 0.0;                // completeness cutoff the sample was taken with (a correction changes ln J)
    int lengths = 0;                    // pooled decision: lengths to count (0: all of them, no early break)
    double alive_share = 0.0;           // sampled share of the pairs still in the running after them
    bool mixed = false;                 // the blocks disagree: d_block_ke holds each block's count (nk: all of them)
    uint32_t shift_r = 31, shift_c = 31, blk_rows = 1, blk_cols = 1;
    std::vector<uint8_t> block_ke;      // [blk_rows * blk_cols], host copy (skl_ctx_early_break_blocks)
    uint8_t *d_block_ke = nullptr;
};

struct skl_ctx {
    int device = 0;
    int n_cu = 256;                     // compute units of the device (MI355X: 256)
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    // grow-only scratch
    void *scratch[16] = {};   // 0/3: key bands, 1: counts, 2: kNN staging, 4/5: turned key bands (symmetric kNN), 6: its row flags (2 x n u32), 7: its block bits,
                              // 8: tile-pruning bounds (n u32), 9: bits of the turned bands, 10: pruning counters, 11: arrival counters of the fused epilogue, 12-14: GPU sketching (packed bases, signs, small arrays), 15: second counts band (early break of the core/accessory kNN)
    size_t scratch_bytes[16] = {};
    uint32_t *pinned = nullptr;         // pinned host ring of the sketching upload (two batches of packed bases; grow-only)
    uint64_t pinned_words = 0;
    size_t fuse_counter_k = 0;          // k-mer lengths the arrival counters of slot 11 count modulo (fused epilogue)
    hipStream_t aux_stream = nullptr;   // top-k of band i runs here while band i+1 is computed
    hipStream_t epi_stream = nullptr;   // band pipeline of the early break (capi.cpp dense_band): band i's epilogue beside band i+1's counts kernel
    // band pipelines (kNN: pair kernel -> top-k; dense to host: pair kernel -> D2H copy):
    // "producer finished buffer b" / "consumer finished buffer b"
    hipEvent_t knn_pair_done[2] = {nullptr, nullptr}, knn_topk_done[2] = {nullptr, nullptr};
    // timing of pair-kernel launches of the last call
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    size_t events_used = 0;
    long long timing_every = 0;         // skl_ctx_timing_enable / SKL_TIMING_EVERY: 0 = launches are not bracketed (default)
    size_t launches_seen = 0;           // pair-kernel launches since the last skl_ctx_timing_reset
    std::string last_kernel;
    uint32_t last_count_planes = 1;     // planes the last MODE_COUNTS k-sliced launch wrote (epilogue: n_slices)
    bool last_tail = false;             // ... or the tail-sliced one-workgroup-per-unit launch (plane 1 added to as well)
    // plane 1 of the counts scratch as the tail slices need it: all zero.  Valid for exactly
    // this (pointer, bytes) until anything else writes the scratch.
    const void *clean_plane1 = nullptr;
    size_t clean_plane1_bytes = 0;
    // skl_clock_sampler_start/stop: one-wave shader-clock sampler on its own stream (diagnostic)
    hipStream_t sampler_stream = nullptr;
    uint32_t *sampler_stop = nullptr;   // pinned host flag the kernel polls
    uint64_t *sampler_buf = nullptr;    // [max][2] (s_memtime, s_memrealtime)
    uint32_t *sampler_count = nullptr;
    uint32_t sampler_max = 0;
    bool sampler_running = false;
    uint64_t knn_tiles_sparse = 0;         // tiles that survived the probe and were finished by the sparse walk (alive rows only)
    uint64_t knn_tiles_probe_pruned = 0;   // ... of the pruned tiles, those the plane-pair probe settled before the walk began
    uint64_t knn_pruned_stages = 0, knn_tile_stages = 0;   // ... stages the pruned tiles had walked / stages of a whole tile
    uint32_t *eb_counter = nullptr;        // device word: pairs the early-break epilogue completed (skl_ctx_early_break_stats)
    uint64_t eb_pairs = 0;                 // ... out of this many pairs of early-break launches since the context was made
    // band pipeline of a large early-break call (capi.cpp dense_band): counts kernels on `stream`, epilogues on `epi_stream`
    bool eb_in_pipeline = false, eb_pipe_overlap = false;
    int eb_pipe_buf = 0;
    hipEvent_t eb_events[4] = {nullptr, nullptr, nullptr, nullptr};   // counts of buffer b done / epilogue of buffer b done
    std::vector<EbPlan *> eb_plans;        // early-break decisions of the last few slab pairs (newest last)
    const EbPlan *eb_last_plan = nullptr;  // the plan of the last dense core/accessory call (skl_ctx_early_break_blocks)
    bool knn_prune_pending = false;        // the device counters (scratch slot 10) hold counts not yet read back
    uint64_t knn_tiles = 0, knn_tiles_pruned = 0;   // tile pruning of the last self kNN call (skl_ctx_knn_prune_stats)
    int knn_ties = SKL_KNN_TIES_REFERENCE;   // what self_dists_knn returns (mod.rs:133-224); skl_ctx_set_knn_ties(CANONICAL) opts out
    Knobs knobs;                        // environment switches as of skl_ctx_create
    skl::TileScratch tile_scratch;      // device table of the balanced tile enumeration
    std::set<skl_sketches *> sketches;  // slabs created on this context
};

struct skl_sketches {
    skl_ctx *ctx = nullptr;
    size_t n = 0, nk = 0, ss64 = 0;
    std::vector<size_t> kmers;
    uint64_t *d_rows = nullptr;  // reference layout + A_PAD_ROWS zero rows (scalar operand)
    uint4 *d_lanes = nullptr;    // lane-interleaved layout (vector operand), built on demand
    double *d_comp = nullptr;    // completeness or null
    bool comp_unit = false;      // every completeness value is finite and in (0, 1] (what the lean early-break epilogue's integer tests need)
    double *d_ytab = nullptr;    // ln J table [64*ss64+1]
    double *d_kf = nullptr;      // k-mer lengths as f64 [nk]
    std::map<std::pair<int, size_t>, float *> d_dtab;  // (jout, k_idx) -> f32 table
    uint64_t gen = 0;            // generation id, unique per slab AND per completeness vector over the life of the process
    uint32_t min_alive = 0xFFFFFFFFu;   // ln J(count) < tolerance <=> count < min_alive (set with d_ytab; 0xFFFFFFFF: ask the table)
    size_t sample_words() const { return nk * ss64 * skl::BBITS; }
};

// device allocation freed at scope exit
struct DevBuf {
    void *p = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { if (p) (void)hipFree(p); }
};

constexpr size_t BAND_BYTES = 512ull << 20;  // scratch bound for host-destined / banded output

SKL_INTERNAL int ctx_bind(skl_ctx *ctx);
// which glibc_log.hpp form reproduces this host's libm log(): probed once per process against
// std::log; SKL_LOG_FMA / SKL_LOG_SSE2, or -1 when neither does (the FMA form is then used and a
// warning printed once)
SKL_INTERNAL int host_log_variant();
// grow-only scratch slot `which` of the context, at least `bytes` large
SKL_INTERNAL int ctx_scratch(skl_ctx *ctx, size_t bytes, void **out, int which = 0);
SKL_INTERNAL uint32_t ctx_xcd_shift(const skl_ctx *ctx);   // log2 of the XCDs the tile order deals workgroups to
SKL_INTERNAL int forced_kernel(const skl_ctx *ctx);   // A/B build: SKL_KERNEL; product library: always 0
// the pair kernel bracketed by HIP events on the context's stream (skl_ctx_kernel_ms)
SKL_INTERNAL int timed_pair_launch(skl_ctx *ctx, const skl::PairArgs &args, int mode);
// records [first, second) events around a launch of another kernel the same way; returns the
// slot to record into or null when the event budget is used up
SKL_INTERNAL std::pair<hipEvent_t, hipEvent_t> *timing_slot(skl_ctx *ctx);
SKL_INTERNAL int check_params(const skl_sketches *a, const skl_sketches *b, const skl_dist_params *p);
SKL_INTERNAL bool fused_coreacc_ok(const skl_sketches *s);
// early break of the core/accessory calls (capi.cpp): the decision for this slab pair (sampled once, kept by the context);
// *plan = null: not applicable (fewer than 3 or more than 8 k-mer lengths, a tiny pair space, switched off)
SKL_INTERNAL int early_break_plan(skl_ctx *ctx, const skl_sketches *rows, const skl_sketches *cols, int self_mode, double cutoff,
                                  const EbPlan **plan);
// does the early break's epilogue walk the pairs in blocks kept on one XCD each?  (capi.cpp: large launches with many pairs still in the running)
SKL_INTERNAL bool eb_blocked_order(const skl_ctx *ctx, const skl_sketches *rows, const EbPlan *plan, uint64_t pairs);
// ... its pooled form, for the kNN drivers: k-mer lengths the pair kernel should count (0: all of them)
SKL_INTERNAL int early_break_lengths(skl_ctx *ctx, const skl_sketches *rows, const skl_sketches *cols, int self_mode, int *lengths);
// operand / epilogue fields common to every launch: `rows` is the scalar operand (A), `cols` the lane operand (B)
SKL_INTERNAL int fill_args(const skl_sketches *rows, const skl_sketches *cols, const skl_dist_params *p, int mode,
                           int jout, skl::PairArgs *g);
// rows [r0, r1) of the pair space into `dst_dev` (device memory)
SKL_INTERNAL int dense_band(skl_ctx *ctx, const skl_sketches *rows, const skl_sketches *cols,
                            const skl_dist_params *p, int mode, int jout, int self_mode, uint64_t r0, uint64_t r1,
                            void *dst_dev);
