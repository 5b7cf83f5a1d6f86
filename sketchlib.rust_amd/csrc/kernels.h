// kernels.h -- internal interface between the C ABI (capi*.cpp) and the gfx950
// device code (*.hip).  Not part of the public boundary.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <algorithm>
#include <vector>

namespace skl {

constexpr int BBITS = 14;          // src/sketch/mod.rs:34
constexpr int LANES = 64;          // wavefront width on gfx950
constexpr int WAVES_PER_WG = 4;    // 256-thread workgroups
constexpr int MAX_FUSED_K = 6;     // k-mer lengths the fused core/acc epilogue packs (3 x 2 x u16)
constexpr int TILE_PREFIX_INLINE = 16;
constexpr int A_PAD_ROWS = 64;     // rows the scalar-operand slab is over-allocated by
// pair_kslice.hip keeps a k-mer length's mismatch counts in u16 fields: whole sketches up to 1 023 chunks (65 472 bins);
// larger ones are walked in segments of 1 016 chunks (a multiple of 8 = the chunks per stage of both tile heights)
constexpr int KSLICE_MAX_U16_CHUNKS = 1023;
constexpr int KSLICE_SEG_CHUNKS = 1016;

// What the pair kernel does with the per-(pair,k) mismatch counts.
enum PairMode : int {
    MODE_COUNTS = 0,   // store samebits as u32 [pair][k]            (jaccard.rs:15-25)
    MODE_JACCARD = 1,  // one k: store f32 distance / ANI / kNN key  (mod.rs:83-100)
    MODE_COREACC = 2,  // all k: fused regression, store (core, acc) (jaccard.rs:61-142)
};

// How MODE_JACCARD turns J into the stored f32.
enum JaccardOut : int {
    JOUT_DIST = 0,     // (1 - J) as f32                    mod.rs:99
    JOUT_ANI = 1,      // ani_pois(J,k) as f32              mod.rs:97
    JOUT_ANI_KEY = 2,  // (1 - ani_pois(J,k)) as f32        mod.rs:173-176 (kNN sort key)
};

struct PairArgs {
    // operands
    const uint64_t *A;   // scalar operand: reference layout [nA + pad][nk][ss64][14]
    const uint4 *B;      // lane operand: [ceil(nB/64)][nk][ss64][7][64] x (2 planes x u64)
    uint32_t nA, nB;     // valid samples on each side
    uint32_t nk, ss64;
    uint32_t k_begin, k_count;
    // pair space
    uint32_t row_begin, row_end;  // A rows computed by this launch
    uint32_t self_mode;           // 1: A == B sample set, only i < j
    uint32_t a_tiles;             // workgroup tiles along rows
    uint32_t n_jblocks;           // 64-wide column blocks
    // balanced tile enumeration: active tiles are numbered column
    // group by column group and each XCD takes one contiguous eighth of the numbering
    uint32_t n_active_tiles;
    uint32_t tiles_per_xcd;       // ceil(n_active_tiles / XCDs)
    uint32_t xcd_interleave;      // 1: the XCDs take the tile numbering in turns, 32 tiles at a time, instead of one contiguous share each
                                  // (launches whose tiles differ in cost by REGION -- the early break decided per block: a contiguous
                                  // share would give one XCD the blocks that count every length and another the ones that count two)
    uint32_t xcd_shift;           // log2 of the XCDs the device shows as one (SPX MI355X: 3; CPX: 0): workgroup b runs on XCD
                                  // b mod 2^xcd_shift (MI355X_MICROARCH.md), set by the C ABI from the device's CU count
    uint32_t n_groups;            // column groups
    uint32_t group_span;          // column groups per super-group: the tiles of a super-group are numbered row tile by
                                  // row tile, its groups side by side (1: column group by column group)
    uint32_t tile_rows, group_cols;   // tile height / columns per group the numbering was planned for
    const uint32_t *tile_prefix;  // [ceil(n_groups / group_span) + 1] first tile number of each super-group (self mode)
    // ... and a copy of its first entries IN the kernel arguments when it has at most TILE_PREFIX_INLINE of them (launches of
    // up to ~3 800 genomes): a workgroup's tile lookup is then scalar loads from the kernarg segment (scalar cache) instead of
    // a binary search of dependent global loads -- the first thing every workgroup of a small launch waits for
    uint32_t inline_prefix_ok;    // host-side request (SKL_INLINE_PREFIX=0 clears it: A/B)
    uint32_t n_prefix_inline;     // entries valid in tile_prefix_inline (0: search tile_prefix)
    uint32_t tile_prefix_inline[16];
    uint64_t out_base;            // flat index of the first pair of this launch
    void *out;
    // MODE_COUNTS record layout: count of (pair p, k index kk) at out[p*cnt_pair_stride + kk*cnt_k_stride]
    // ([pair][k] for the public bin-match calls, k-major for the internal counts scratch)
    uint64_t cnt_pair_stride, cnt_k_stride;
    uint32_t cnt_u16;             // MODE_COUNTS: the records are u16 (k-sliced launches without chunk slices, sketches of up to 1 023 chunks)
    uint32_t k_sliced;            // host-side request: one workgroup per (tile, k-mer length)
    uint32_t slice_chunks;        // chunks per chunk slice (k_slices / tail_slices > 1): a multiple of 8; the last slice takes what is
                                  // left, so any sketch size can be cut (slice_plan(); 0: ss64 / slices, the even split)
    uint32_t seg_chunks;          // pair_kslice.hip, set by its launcher: chunks per segment of the segmented walk (0: whole k-mer lengths)
    uint32_t k_slices;            // k-sliced MODE_COUNTS: chunk slices per k-mer length (0/1: none); slice s of k index kk
                                  // stores the matches of ITS bins at "k index" s * k_count + kk
    // Tail slicing of a k-sliced MODE_COUNTS launch (pair_kslice.hip): on every XCD the workgroups from
    // index tail_first on are chunk slices of units (tail_slices per unit, adjacent) instead of whole
    // units; slice 0 stores its counts to plane 0, the others ADD theirs to plane 1 (k index
    // k_count + kk: zero on entry, summed and re-zeroed by the epilogue).
    // tail_slices <= 1: off.  The launcher sets tail_first from tail_resident = workgroups resident per XCD.
    uint32_t tail_slices, tail_first, tail_resident;
    // pair_kslice.hip: workgroups of round r of an XCD (index / round_size) run at wave priority min(r, 3),
    // so that a later round is not served after every older wave on its SIMD (0: off)
    uint32_t round_size;
    uint32_t mid_band;            // host-side request (dense_band): 32 x 128 tiles whatever the launch size
    uint32_t no_half_tiles;       // 1: walk both 64-column blocks of every tile (A/B timing of the half tiles; results identical)
    // Symmetric self kNN (pair_kslice.hip: k-sliced MODE_JACCARD and all-k MODE_COREACC): besides
    // out, the record of (row i, column j >= t_col_begin) also goes to record
    // (j - t_col_begin) * t_stride + (i - row_begin) of out_t, i.e. as a candidate of row j.
    // t_stride (records) is a multiple of 4 and >= the rows of the launch rounded up to the
    // tile height.  Null = off.
    float *out_t;
    uint32_t t_col_begin, t_stride;
    // ... and the rows that receive such a candidate BELOW their current knn-th best are flagged, so
    // that the merge of the transposed band visits only them: t_thr[c * t_thr_stride] is the sortable
    // key (topk.hip) of column c's knn-th best so far (0xFFFFFFFF while it has fewer; it may be stale,
    // i.e. too high, never too low), t_flag[c] is set to t_flag_value.  Column indices are the view's
    // (like t_col_begin).  Null = off.
    const uint32_t *t_thr;
    uint32_t *t_flag;
    uint32_t t_thr_stride, t_flag_value;
    // The same for the band's OWN rows, per 64-column block: bit b of r_bits[(i - row_begin) * r_bits_stride
    // + b / 32] is set when row i received a record in columns [64 b, 64 b + 64) of the view below its
    // knn-th best so far, r_thr[(i - row_begin) * r_thr_stride]; the merge of the band skips the blocks
    // without a bit.  Zeroed by the host before the launch.  Null = off.
    const uint32_t *r_thr;
    uint32_t *r_bits;
    uint32_t r_thr_stride, r_bits_stride;
    // ... and for the turned records one bit per (column, 32-row stretch of the band): bit t of t_bits[c * t_bits_stride + t / 32]
    // is set when column c received, from rows [row_begin + 32 t, + 32), a record below t_thr.  With it the merges of BOTH
    // copies read marked stretches only, which is what lets a workgroup leave its tile unwritten (below).  Zeroed by the host.
    uint32_t *t_bits;
    uint32_t t_bits_stride;
    // TILE PRUNING (symmetric self kNN, single-k keys, 32 x 128 k-sliced form; pair_kslice_walk.inc).  The key is monotone in
    // the mismatch count, so a pair whose count over the chunks walked SO FAR already exceeds allow = the largest count whose
    // key is still below BOTH samples' knn-th best can enter neither list, and a tile all of whose pairs are in that state
    // need not be finished: the workgroup leaves without storing (no bit above is set for it, so nothing reads its records).
    // prune_q_rows[i] / prune_q_cols[c] (row sample id / view column; null = off) = floor(allow / 4) + 1 of that sample (0:
    // nothing can enter), a per-WAVE bound: the 4 waves of a workgroup split the chunks, and if every wave's own count of a
    // pair is >= max(q_row, q_col) the pair's total exceeds max(allow_row, allow_col).  Written by launch_prune_thresholds
    // from the running states before the launch; a stale (too high) threshold prunes less, never wrongly.
    const uint32_t *prune_q_rows, *prune_q_cols;
    uint32_t *prune_stats;        // 1 024 slots of 4 words (slot = blockIdx & 1023: adds to one address would queue up): [0] += workgroups that
                                  // left their tile at the probe, [1] += ... at a stage boundary, [2] += the stages those had walked,
                                  // [3] += workgroups that finished their tile by the sparse walk (null: not counted)
    uint32_t prune_flags;         // bit 0: no sparse walk (A/B build, SKL_KNN_SPARSE=0: the probe's survivors are walked whole)
    // EARLY BREAK DECIDED PER BLOCK (k-sliced MODE_COUNTS; capi.cpp early_break_plan): the workgroup of (tile, k index kk)
    // leaves at once when kk >= the k-mer lengths the tile's block(s) of (row >> blk_shift_r, column >> blk_shift_c) sample ids
    // count -- block_ke[row block * blk_cols + column block]; blocks are multiples of 256 samples, so a tile's 128 columns lie in
    // one and its rows in at most two (the larger count wins).  Null: every tile counts k_count lengths.
    const uint8_t *block_ke;
    uint32_t blk_shift_r, blk_shift_c, blk_cols;
    // FUSED CORE/ACCESSORY EPILOGUE of a k-sliced MODE_COUNTS launch (one workgroup per (tile, k-mer length), no chunk slices;
    // pair_kslice.hip, FUSE): every workgroup stores its k-mer length's counts write-through, then ONE lane adds 1 to
    // fuse_counter[tile] (agent scope, returning); the workgroup whose add completes the tile's k_count arrivals reads the
    // tile's k_count count planes back (agent-scope loads) and stores (core, acc) of the tile's pairs to fuse_out -- no
    // second launch.  The counter is never reset: arrivals are counted modulo k_count (zeroed by the host whenever k_count
    // or the buffer changes).  Null = off (counts only).
    uint32_t *fuse_counter;
    uint32_t fuse_variant;        // EXPERIMENT (SKL_FUSE_VARIANT): 1 plain count stores, 2 the last arriver skips the epilogue (timing only)
    float *fuse_out;              // [pair] (core, acc), same pair index as the counts
    // epilogue
    int32_t jout;                 // JaccardOut
    int32_t has_comp;
    int32_t log_variant;          // glibc_log.hpp form the host libm uses (completeness path only)
    const double *ytab;           // [64*ss64 + 1] ln(J(samebits)), host libm
    const float *dtab;            // [64*ss64 + 1] f32 Jaccard-mode output, host libm
    const double *compA, *compB;  // completeness per sample (device) or null
    double cutoff;
    double tolerance;             // ln(2 / (sketch_size * 64))   jaccard.rs:75
    double kf[MAX_FUSED_K];       // k-mer lengths as f64
};

// Device buffer the launchers may use for the tile-prefix table (owned by the context).
struct TileScratch {
    uint32_t *d_prefix = nullptr;
    uint32_t *h_staging = nullptr; // pinned host copy the upload reads from
    hipEvent_t staged = nullptr;   // recorded after each upload: h_staging may be rewritten once it fired
    size_t capacity = 0;           // entries
    uint64_t cached_key[4] = {~0ull, ~0ull, ~0ull, ~0ull};
};

// Fills n_active_tiles / tiles_per_xcd / n_groups / tile_prefix of `args` for tiles of
// rows_per_tile x cols_per_group; uploads the prefix table (self mode) on `stream`.
// Returns the grid size (0 = nothing to do).
hipError_t plan_tiles(PairArgs &args, uint32_t rows_per_tile, uint32_t cols_per_group,
                      TileScratch &scratch, hipStream_t stream, uint64_t *grid_out);

// K-split variant for small launches (pair_ksplit.hip): rows_per_tile in {4, 8}.
hipError_t launch_pair_kernel_ksplit(const PairArgs &args, int mode, int rows_per_tile,
                                     TileScratch &scratch, hipStream_t stream);
// Chunk-split kernel (pair_kslice.hip): R x 64*JL tiles, the 4 waves of a workgroup split the
// chunks.  k_sliced = one workgroup per (tile, k-mer length), MODE_COUNTS / MODE_JACCARD only;
// otherwise one workgroup walks all k-mer lengths and runs the fused epilogue.
// shape: 165 / 325 (16 x 128 and 32 x 128 tiles in 128 registers) in the product library; the A/B build also has
// their round-2/3 forms (1651, 1652, 3254, 3255) and two timing-only ablations (ablate = 4, 8: outputs wrong
// by construction).
hipError_t launch_pair_kernel_kslice(const PairArgs &args, int mode, int shape, bool k_sliced, int ablate,
                                     TileScratch &scratch, hipStream_t stream);
bool kslice_supported(const PairArgs &args, int mode, bool k_sliced);
// A sketch of ss64 chunks cut into at most `wanted` chunk slices of whole stages: *chunks per slice (a multiple of 8, the
// last slice shorter) -> number of slices that hold something (1: the sketch is too short to cut)
inline uint32_t slice_plan(uint32_t ss64, uint32_t wanted, uint32_t *chunks)
{
    if (wanted < 2u || ss64 < 16u) {
        *chunks = 0;
        return 1u;
    }
    const uint32_t per = ((ss64 + wanted - 1u) / wanted + 7u) / 8u * 8u;
    *chunks = per;
    return (ss64 + per - 1u) / per;
}

// reference layout -> lane-interleaved layout (B operand); n_pad = 64*ceil(n/64)
hipError_t launch_relayout(const uint64_t *ref_layout, uint4 *lane_layout, uint32_t n,
                           uint32_t nk, uint32_t ss64, hipStream_t stream);

// Unfused core/acc epilogue: counts [pair][nk] u32 -> (core, acc) f32 pairs.
struct EpilogueArgs {
    uint32_t *counts;           // count of (pair p, k t) at counts[p*pair_stride + t*k_stride]
    uint64_t pair_stride, k_stride;
    uint64_t n_pairs;
    uint32_t nk, ss64;
    uint32_t n_slices;          // counts come in this many chunk slices per k (k index s * nk + t): summed here
    uint32_t rezero_plane1;     // 1: after reading, write 0 to the slice-1 records (the tail slices add into them)
    uint32_t nA_rows, nB_cols;  // to recover (i, j) for completeness lookups
    uint32_t row_begin;
    uint32_t self_mode;
    uint32_t n_total;           // n (self mode condensed indexing)
    uint64_t out_base;
    int32_t has_comp;
    int32_t log_variant;        // glibc_log.hpp form (completeness path only)
    const double *ytab;
    const double *compA, *compB;
    double cutoff, tolerance;
    const double *kf;           // device array [nk]
    float *out;
    // single-k Jaccard launches that ran as counts (chunk slices of launches smaller than the chip):
    // nk == 1, and the output is one f32 per pair as pair_kslice.hip's MODE_JACCARD would have stored it
    int32_t jaccard_out;        // 0: core/accessory; 1: f32 per pair, value chosen by jout
    int32_t jout;               // JaccardOut
    double kf0;                 // k-mer length of the one k (ANI outputs)
    const float *dtab;          // [64*ss64 + 1] f32 output per bin-match count (no completeness correction)
    // EARLY BREAK (round 5; core/accessory, no completeness correction).  The reference's loop over the k-mer lengths leaves
    // at the first one whose Jaccard index is 0 (jaccard.rs:89-91), and a fit over fewer than three lengths is (1, 1)
    // (jaccard.rs:117): a pair without a shared bin at one of the FIRST THREE lengths is decided by them alone -- 98.9 % of
    // unrelated pairs at 4 096 bins, 99.8 % at 2 048.  The pair kernel then counts only `nk` = 3 of the `nk_total` lengths,
    // and the few pairs that are still in the running get their remaining counts here: the wave that holds such a pair
    // compares its two sketches cooperatively (a lane per chunk, reference layout) length by length until the break.
    // nk_total == 0 or == nk: off.  rows_ref / cols_ref: the two slabs in the reference's layout.
    uint32_t nk_total;
    const uint64_t *rows_ref, *cols_ref;
    uint32_t *alive_count;      // 1 024 words, slot blockIdx & 1023 += pairs completed here (diagnostic; may be null)
    // BLOCKED ORDER (epilogue.hip, early-break launches): workgroups walk the pair space in blocks of 256 rows x 256 columns kept
    // on one XCD each, so that the column slices the completions read are reused from that XCD's L2.  row_end: one past the
    // launch's last row; blk_rb / blk_cb: set by the launcher
    uint32_t blocked, row_end, xcd_shift, blk_rb, blk_cb;
    uint32_t blk_row_shift;     // log2 of the rows per block (out of 5 ... 12: 10)
    uint32_t blk_cb0, wg_base;  // (launcher) first column block that holds a pair; workgroup offset of this launch (a launch carries < 2^32 work-items)
    uint32_t comp_lean;         // 1: both completeness vectors lie in (0, 1] (checked on the host): the lean form's integer tests hold under the correction
    uint32_t lean;              // 1: launches the lean form can take go to it (epilogue.hip coreacc_epilogue_lean_kernel)
    uint32_t ahead;             // 1: with the row slices in LDS, the first length not counted is completed with the column slices requested a trip ahead
    uint32_t lds_rows;          // host-side request (epilogue.hip): a workgroup with a pair still in the running stages its two row slices in LDS
    uint32_t min_alive;         // no completeness correction: ln J(count) < tolerance <=> count < min_alive (0xFFFFFFFF: ask ytab)
    uint32_t cnt_u16;           // 1: the counts are u16 records (sketches of up to 1 023 chunks, no chunk slices)
    // EARLY BREAK DECIDED PER BLOCK: block_ke[(i >> blk_shift_r) * blk_cols + (j >> blk_shift_c)] = k-mer lengths the pair kernel
    // counted for the pairs of that block of (row sample, column sample) ids (nk_total: all of them); `nk` is then the number
    // of count planes (= nk_total).  Null: `nk` lengths for every pair.
    const uint8_t *block_ke;
    uint32_t blk_shift_r, blk_shift_c, blk_cols;
};
hipError_t launch_coreacc_epilogue(const EpilogueArgs &args, hipStream_t stream);
// core/accessory records only, round 6 (epilogue.hip)
hipError_t launch_coreacc_epilogue_r6(const EpilogueArgs &args, hipStream_t stream);
bool coreacc_epilogue_is_lean(const EpilogueArgs &args);   // will that launch take the lean kernel?
// EARLY BREAK in the one-evaluation core/accessory self kNN (capi_knn.cpp): a row band's view of the columns was counted at
// its first `nk` k-mer lengths (k-major counts, counts[t * n_pairs + row * nB + c]); this launch turns them into the band's
// (core, acc) records -- completing the pairs still in the running like coreacc_epilogue_kernel -- and does what the fused
// pair kernel's epilogue does for the symmetric driver: the band's records row-major, one bit per (row, 64-column block)
// that holds a key below the row's knn-th best, and the TURNED copy.  The turned band is pre-filled with (1, 1) by the host
// (a memset: what every pair that left the loop early is); only the other records are stored into it, with their marks.
struct EpilogueKnnArgs {
    uint32_t xcd_blocked;        // column-group-major order per XCD (epilogue.hip): 0 never, 1 from 32 column groups on, 2 always
    const uint32_t *counts;
    uint64_t n_pairs;            // rows * nB
    uint32_t rows, nB;           // the band's rows, the view's columns
    uint32_t row_base;           // (set by the launcher: a launch covers at most 32 768 rows)
    uint32_t nk, nk_total, ss64;
    uint32_t row_sample0, col_sample0;   // sample ids of row 0 / view column 0
    const double *ytab, *kf;
    double tolerance;
    const uint64_t *rows_ref, *cols_ref;
    float *out;                  // [rows][nB] float2
    const uint32_t *r_thr;       // knn-th best (sortable bits) of row r at r_thr[r * r_thr_stride]; null: no marks
    uint32_t r_thr_stride;
    uint32_t *r_bits;
    uint32_t r_bits_stride;
    float *out_t;                // turned band [view column - t_col_begin][t_stride] float2, or null
    uint32_t t_col_begin, t_stride;
    const uint32_t *t_thr;       // indexed by view column (as PairArgs::t_thr)
    uint32_t t_thr_stride;
    uint32_t *t_flag;
    uint32_t t_flag_value;
    uint32_t *t_bits;
    uint32_t t_bits_stride;
    uint32_t *alive_count;
    uint32_t min_alive;          // as EpilogueArgs::min_alive
    uint32_t cnt_u16;            // 1: the counts are u16 records
    uint32_t plain_marks_nothing;   // 1: every list the band meets holds knn candidates (keys <= 1), so a (1, 1) record can enter none
};
hipError_t launch_coreacc_epilogue_knn(const EpilogueKnnArgs &args, hipStream_t stream);
// EARLY BREAK, the driver's question before it counts only the first few k-mer lengths of a block of pairs: how many of them
// would still be in the running after those?  The pair space is cut into blocks of (row >> blk_shift_r, column >> blk_shift_c)
// sample ids; `samples` pairs of every block (a fixed pseudo-random choice; one wave each) run the reference's loop,
// jaccard.rs:77-91, and hist[block * 9 + m] += the pairs that pass its test at each of their first m lengths and not at the
// next (m = nk: at all of them; m <= 8).  Self mode: blocks below the diagonal stay empty.  The test is the epilogue's
// (min_alive / the restated logarithm with a completeness correction).
struct EbSampleArgs {
    const uint64_t *rows_ref, *cols_ref;
    uint32_t n_rows, n_cols, nk, ss64, self_mode;
    uint32_t samples, blk_shift_r, blk_shift_c, blk_rows, blk_cols;   // blk_rows x blk_cols blocks
    uint32_t min_alive;
    int32_t has_comp, log_variant;
    const double *ytab, *compA, *compB;
    double cutoff, tolerance;
    uint32_t *hist;
};
hipError_t launch_early_break_sample(const EbSampleArgs &args, hipStream_t stream);
// one-wave shader-clock sampler (kernels.hip): (s_memtime, s_memrealtime) pairs until *stop != 0 or max_samples
hipError_t launch_clock_sampler(const uint32_t *stop, uint64_t *samples, uint32_t max_samples, uint32_t sleeps,
                                uint32_t *count, hipStream_t stream);
// y[i] = glibc_log(x[i], variant) on the device (glibc_log.hpp)
hipError_t launch_device_log(const double *x, double *y, uint64_t n, int variant, hipStream_t stream);

// Row-wise top-k over a dense [rows][cols] band of keys (and optional second value).
// Candidate-list pair kernel (pair_cand.hip): row i against cand[row_offsets[i] .. row_offsets[i+1]).
struct CandArgs {
    const uint64_t *row_offsets;  // [n_rows + 1]
    const uint32_t *cand;         // candidate sample ids (ascending inside a row)
    const uint32_t *work_row;     // [n_work] row of each 64-candidate work item
    const uint64_t *work_start;   // [n_work] first candidate position of the work item
    uint64_t n_work;
    float *keys;                  // [n_candidates] MODE_JACCARD output per candidate
    uint32_t symmetric;           // 1: the lists are symmetric -- evaluate j > i only, store both copies
    // consecutive work items (rows of one cluster, capi_aux.cpp) stay on ONE XCD, whose L2 then holds the candidates they
    // share: workgroup b takes the items of block (b mod XCDs) * blocks_per_xcd + b / XCDs (set by the launcher)
    uint32_t xcd_shift, blocks_per_xcd;
    uint32_t lanes_over_candidates;   // 1: round 3's form (every lane walks its own candidate); 0: lanes across the sketch (round 4)
};
hipError_t launch_pair_cand(const CandArgs &c, const PairArgs &g, hipStream_t stream);

// GPU sketching (sketch_kernel.hip): bin minima of the canonical ntHash of every valid k-mer.
struct SketchArgs {
    const uint32_t *packed;        // 2-bit base codes, 16 per word (code c of a sample at bits 2 (c % 16) of its word c / 16)
    const uint64_t *word_begin;    // [n_samples] first word of each sample (a sample's last word is zero-padded)
    uint64_t first_span;           // this launch handles spans first_span .. first_span + n_spans - 1 (a batch of samples)
    const uint64_t *code_begin;    // [n_samples + 1] in codes: sample s has code_begin[s + 1] - code_begin[s] of them
    const uint64_t *offsets;       // breaks (N / record ends) in each sample's own coordinates
    const uint64_t *offset_begin;  // [n_samples + 1]
    const uint64_t *span_begin;    // [n_samples + 1] prefix sum of ceil(len / span) per sample
    uint64_t n_spans;
    uint32_t n_samples, nk;
    const uint32_t *kmers;         // [nk]
    const uint64_t *top_f, *top_r; // [nk][4] srol^(k-1) of the forward / reverse seeds
    uint64_t num_bins, bin_size;
    double inv_bin_size;
    int32_t rc;
    uint64_t *signs;               // [n_samples][nk][num_bins], pre-set to UINT64_MAX
    // 1: LDS-staged kernel -- span_begin counts spans of sketch_span_lds() window starts, every
    // sample's count padded to a multiple of sketch_wg_lds() (a workgroup never straddles samples), every
    // k-mer length <= sketch_span_lds() + 1
    uint32_t lds_form;
};
hipError_t launch_sketch_signs(const SketchArgs &args, hipStream_t stream);
int sketch_span();
int sketch_span_lds();
int sketch_wg_lds();   // threads (= spans) per workgroup of the LDS-staged kernel: a sample's span count is padded to a multiple

// Candidate lists on the device (cand_gen.hip): any shared bin between index sketches.
struct CandGenArgs {
    const uint16_t *skq;      // [n][sketch_size] index sketches, row = sample id
    uint32_t n, sketch_size;
    uint32_t *starts, *cursor;  // [sketch_size][65536] group begin / (after the scatter) end
    uint32_t *members;          // [sketch_size][n] sample ids grouped by value
    uint32_t *counts;           // [n] candidates per row (pass 1)
    const uint64_t *row_offsets;  // [n + 1] (pass 2)
    uint32_t *cand;             // candidate ids, ascending inside a row (pass 2)
};
hipError_t launch_cand_groups(const CandGenArgs &g, hipStream_t stream);
hipError_t launch_cand_rows(const CandGenArgs &g, bool fill, hipStream_t stream);
// first[i] = position of the first candidate of row i with an id greater than i
hipError_t launch_first_greater(const uint64_t *row_offsets, const uint32_t *cand, uint32_t n, uint64_t *first,
                                hipStream_t stream);
// key[i] = first candidate of row i, 0xFFFFFFFF for an empty row
hipError_t launch_first_candidate(const uint64_t *row_offsets, const uint32_t *cand, uint32_t n, uint32_t *key, hipStream_t stream);
// work items of the candidate-list kernel, from the per-row item counts the host made (launch order `order`)
hipError_t launch_cand_work_items(const uint32_t *order, const uint64_t *item_begin, const uint64_t *row_offsets, const uint64_t *first,
                                  uint32_t n, uint32_t *work_row, uint64_t *work_start, hipStream_t stream);
constexpr size_t MAX_DEVICE_CANDGEN_SAMPLES = 158ull * 1024 * 8;   // n-bit bitmap in LDS

struct TopkArgs {
    const float *keys;      // [rows][cols] or [rows][cols][2] when stride2
    uint32_t rows, cols;
    uint32_t stride2;       // 1: plain keys; 2: (core, acc) interleaved, key = core
    uint32_t knn;
    uint32_t self_mode;     // skip col == row_begin + row
    uint32_t row_begin;
    int32_t ani_undo;       // write 1.0f - key   (mod.rs:183-189)
    uint64_t *out_idx;      // [rows][knn]
    float *out_d0, *out_d1;
    // Ragged form (candidate lists): row r owns keys[row_offsets[r] .. row_offsets[r+1]) and
    // col_ids maps a position to the sample id written to out_idx; rows with fewer than knn
    // candidates are padded with (row id, 1.0f) as mod.rs:535-546 does.  Null = dense form.
    const uint64_t *row_offsets;
    const uint32_t *col_ids;
    // knn > TOPK_LDS_MAX: the selected items of row r are collected and sorted in
    // items_scratch[r * items_pitch ...) (global memory; items_pitch >= topk_items_pitch(knn)) instead of LDS
    uint64_t *items_scratch;
    uint64_t items_pitch;
    uint32_t first_row;      // this launch handles rows first_row .. first_row + rows - 1 (scratch slices are per launch)
};
constexpr uint32_t TOPK_LDS_MAX = 2048;   // neighbours per row the LDS-resident top-k kernels hold
inline uint64_t topk_items_pitch(uint64_t knn)   // next power of two >= knn (the bitonic sort's array)
{
    uint64_t m = 1;
    while (m < knn) m <<= 1;
    return m;
}
hipError_t launch_topk(const TopkArgs &args, hipStream_t stream);

// Reference tie order (opt-in): the neighbour list the reference binary prints -- std::collections::BinaryHeap
// driven by push_heap (mod.rs:41-48: push when not full or STRICTLY below the maximum, then pop the maximum) over
// the candidates in ascending index order (mod.rs:156-181, :335-369), then into_sorted_vec.  Which of several
// equal keys survive, and in what order equal keys are listed, is decided by the heap's history; the kernel
// replays it (topk.hip: topk_refheap_kernel).  One workgroup per row of a dense [rows][cols] band.
struct RefHeapArgs {
    const float *keys;       // row r's records at keys + r * key_stride (floats), ascending candidate id
    uint64_t key_stride;
    uint32_t stride2;        // 1: plain keys; 2: (core, acc) records, key = core
    uint32_t rows, cols;
    uint32_t self_id_base;   // row r is candidate self_id_base + r and skipped; 0xFFFFFFFF = off
    uint32_t knn;
    int32_t ani_undo;        // write 1.0f - key (mod.rs:183-189)
    uint64_t *out_idx;       // [rows][knn]
    float *out_d0, *out_d1;
    float *heap_scratch;     // knn > REFHEAP_LDS_MAX: [rows of this launch][3 * (knn + 1)] floats in global memory, else null
    // Ragged form (candidate lists, precluster): row r owns keys[row_offsets[r] .. row_offsets[r+1]) and col_ids maps a
    // position to the sample id; the candidates are pushed in the order they are LISTED; rows with fewer than knn
    // candidates are padded with (row id, 1.0f) (mod.rs:535-546).  Null = dense form.  stride2 == 1 only.
    const uint64_t *row_offsets;
    const uint32_t *col_ids;
    uint32_t first_row;      // this launch handles rows first_row .. first_row + rows - 1 (heap_scratch slices are per launch)
    uint32_t force_workgroup_form;   // A/B: one workgroup per row even where one wave per row applies (knn <= 256)
};
constexpr uint32_t REFHEAP_LDS_MAX = 2048;
hipError_t launch_topk_refheap(const RefHeapArgs &args, hipStream_t stream);
// The same replay, resumable: the heap of row r (RefHeap layout: h_key / h_id / h_d1 [.][knn], h_len [.]) is read from and
// written back to global memory, and thr[.] keeps the sortable key bits of its maximum once it is full (0xFFFFFFFF before),
// which is what the pair kernel's row / block flags compare with.  One launch feeds rows [state_row_base, + rows) the
// records keys[r * key_stride ...] as candidates id_base + position; the caller guarantees ascending ids per row over the
// sequence of launches (topk.hip: refheap_merge_kernel).  knn <= REFHEAP_LDS_MAX.
struct RefHeapMergeArgs {
    const float *keys;
    uint64_t key_stride;
    uint32_t stride2;
    uint32_t rows, cols;
    uint32_t id_base, skip_below, self_id_base, state_row_base, knn;
    float *h_key;
    uint32_t *h_id;
    float *h_d1;              // stride2 == 2, else null
    uint32_t *h_len, *thr;
    const uint32_t *flag;     // row r is fed only if flag[r] == flag_value (null: every row)
    uint32_t flag_value;
    const uint32_t *seg_bits; // as TopkMergeArgs::seg_bits
    uint32_t seg_bits_stride, seg_shift;
    uint32_t force_workgroup_form;   // 1: one workgroup per row whatever knn is (default: one WAVE per row up to 256 neighbours; A/B, tests)
    // ACCEPT LOG (decoupled column windows, capi_knn.cpp): every candidate push_heap takes is appended, in order, to the row's
    // log -- log_rec[row * log_cap + x] (stride2 floats: key[, second value]), log_id[row * log_cap + x]; log_len[row] counts on
    // past log_cap (overflow: the caller checks); rows indexed like the heaps (state_row_base + r).  Null: no log.
    float *log_rec;
    uint32_t *log_id, *log_len;
    uint32_t log_cap;
    // EXPLICIT CANDIDATES (the replay of such logs): record q of row r stands for sample cand_ids[r * cols + q] (null: id_base + q),
    // and row r has row_cols[r] records (null: cols; clamped to cols).
    const uint32_t *cand_ids, *row_cols;
};
hipError_t launch_refheap_merge(const RefHeapMergeArgs &args, hipStream_t stream);
hipError_t launch_refheap_finalize(const float *h_key, const uint32_t *h_id, const float *h_d1, const uint32_t *h_len, uint32_t rows,
                                   uint32_t knn, int ani_undo, uint64_t *out_idx, float *out_d0, float *out_d1, hipStream_t stream);

// Running per-row top-k: merges a row's new keys into its sorted state of knn (sortable key,
// sample id[, second value]) entries.  New ids must all be larger than the ids already in the
// state (drivers feed columns / bands in ascending order), which makes "smallest (key, id)" the
// same as "smallest (key, position)" with the state placed first.
struct TopkMergeArgs {
    const float *keys;        // row r's new records at keys + r * key_stride (floats)
    uint64_t key_stride;
    uint32_t stride2;         // 1: plain keys; 2: (core, acc) records, key = core
    uint32_t rows, cols;      // rows of this launch, new records per row
    uint32_t id_base;         // sample id of record position 0
    uint32_t skip_below;      // positions with id < skip_below are not candidates
    uint32_t self_id_base;    // row r is sample self_id_base + r and not its own candidate; 0xFFFFFFFF = off
    uint32_t state_row_base;  // row r's state is row state_row_base + r of run_*
    uint32_t knn;
    uint32_t *run_key;        // [.][knn] sortable key bits, ascending; 0xFFFFFFFF = empty
    uint32_t *run_idx;        // [.][knn]
    float *run_d1;            // [.][knn] second values (stride2 == 2) or null
    uint32_t streaming;       // 1: one-pass streaming merge (default); 0: radix select only (A/B)
    const uint32_t *flag;     // row r is merged only if flag[r] == flag_value (null: every row)
    uint32_t flag_value;
    const uint32_t *seg_bits; // row r: bit b of seg_bits[r * seg_bits_stride + b / 32] clear = no record at positions
    uint32_t seg_bits_stride; // [b << seg_shift, (b + 1) << seg_shift) can enter the state: not read, whatever they hold (null: everything is read)
    uint32_t seg_shift;       // 6 (64 positions per bit; 0 means 6) or 5
};
hipError_t launch_topk_merge(const TopkMergeArgs &args, hipStream_t stream);
// Tile pruning thresholds (PairArgs::prune_q_*): q[i] = floor(allow_i / 4) + 1 with allow_i = the largest mismatch count m
// whose key dtab[total_bins - m] is still STRICTLY below sample i's threshold thr[i * thr_stride] (sortable bits; 0xFFFFFFFF
// = list not full: everything may enter), 0 when no count is.  dtab: the launch's key table over the bin-match count.
hipError_t launch_prune_thresholds(const uint32_t *thr, uint32_t thr_stride, uint32_t n, const float *dtab, uint32_t total_bins,
                                   uint32_t *q, hipStream_t stream);
// Union of up to MERGE_STATES_MAX partial states of the same rows (disjoint candidate sets) ->
// one state: the knn smallest (key, id) per row.  n_in * knn <= MERGE_STATES_ITEMS.
constexpr int MERGE_STATES_MAX = 16;
constexpr int MERGE_STATES_ITEMS = 4096;
struct MergeStatesArgs {
    const uint32_t *key[MERGE_STATES_MAX];
    const uint32_t *idx[MERGE_STATES_MAX];
    const float *d1[MERGE_STATES_MAX];   // all null or all set
    uint32_t n_in, rows, knn;
    uint32_t *out_key, *out_idx;
    float *out_d1;
};
hipError_t launch_merge_states(const MergeStatesArgs &args, hipStream_t stream);
// state -> (out_idx, out_d0) in the public form
hipError_t launch_topk_finalize(const uint32_t *run_key, const uint32_t *run_idx, const float *run_d1, uint64_t items,
                                int ani_undo, uint64_t *out_idx, float *out_d0, float *out_d1, hipStream_t stream);

}  // namespace skl
