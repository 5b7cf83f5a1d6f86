// pair_kslice.hip -- the chunk-split pair kernel (gfx950): the default kernel of the path.
//
// Measured on MI355X (scripts/microbench/lds_bcast.hip): one CU returns ~0.39 wave-level
// ds_read_b128 per ns, broadcast or not, against ~3.4 wave-level v_bitop3 per ns over its
// 4 SIMDs.  A row read of 7 x b128 feeds JL x 28 v_bitop3, so with one column per lane the
// LDS pipe caps a kernel at ~46 % of the VALU rate; two columns per lane lift the cap to
// ~92 %.  The workgroup tile is therefore R rows x 128 columns, every lane owning 2 columns,
// and the 4 waves of the workgroup split the CHUNK axis of that one tile:
//
//   * wave w takes chunk pairs {2w, 2w+1} + 8t of every k-mer length and stages the rows of
//     its own chunks into a private, double-buffered LDS region (global_load_lds DMA): no
//     other wave reads them, so waves never wait for each other while streaming;
//   * at the end of a k-mer length the 4 partial counts per pair are summed through LDS
//     (u16 pairs, written into the row buffer the wave has just finished with) and wave w
//     keeps the totals of packed pair slots x = w (mod 4), finishing those pairs itself.
//
// Two workgroup shapes of the same code (template flag KSL):
//   * KSL = false: one workgroup walks all k-mer lengths of its tile and runs the fused
//     epilogue (counts / Jaccard / core-accessory regression).  Large launches.
//   * KSL = true ("k-sliced"): one workgroup = one (tile, k-mer length); counts or single-k
//     Jaccard only.  BASELINE.json's 1 000-genome configuration has too few tiles to fill
//     256 CUs; slicing the k axis gives 5x the workgroups, and core/accessory launches of
//     that size run MODE_COUNTS into a k-major scratch array followed by
//     coreacc_epilogue_kernel (kernels.hip) -- 28 bytes per pair of extra traffic.
//
// Two tile heights in the product library: R = 16 in 128 registers (4 waves per SIMD) for
// launches below 8 M pair x k evaluations, R = 32 above (one chunk per wave and stage; every column
// register then meets 32 rows, which halves the lane-slab bytes per pair), in 128 registers = 4 waves per
// SIMD (chunks walked in plane-major blocks of MB = 2 rows): k-sliced with 32 KB of LDS, all k + fused
// regression with 32-36 KB (per-k reduction in two phases).
// Launches smaller than the chip (k-sliced counts): the units of the last, partial round of
// workgroups are cut into chunk slices, slice 0 storing and the others adding into a second counts
// plane (PairArgs::tail_slices).
//
// Per (row, column, chunk) the instruction stream is 2 v_xor + 26 v_bitop3 (all-VGPR,
// bank-conflict-free, see device_common.hpp) + 2 fused v_bcnt.
// Wave timelines of this kernel: scripts/microbench/kslice_trace.hip.
#include "device_common.hpp"

#include <algorithm>
#include <cstdlib>

namespace skl {

constexpr uint32_t KSL_TILE_BLOCK = 32;   // k-sliced launches: tiles that walk a k-mer length together

#ifdef SKL_TRACE
// scripts/microbench/kslice_trace.hip: per-wave timeline (100 MHz wall clock) + hardware id,
// and the shader-clock counter (s_memtime) at marks 1 and 2, i.e. around the streaming phase:
// d(s_memtime) / d(wall clock) x 100 MHz is the clock the chip held meanwhile.  Record = 8 words:
// marks 0..3, hardware id, s_memtime at marks 1 and 2.  The trace buffer rides in the dtab field
// (unused by MODE_COUNTS); no output value depends on a stamp.
#define skl_trace ((uint64_t *)g.dtab)
#define SKL_TRACE_WORDS 8u
#define SKL_TRACE_MARK(SLOT)                                                                  \
    do {                                                                                      \
        if (lane == 0) {                                                                      \
            uint64_t *rec_ = &skl_trace[((size_t)blockIdx.x * WAVES_PER_WG + wave) * SKL_TRACE_WORDS]; \
            rec_[(SLOT)] = wall_clock64();                                                    \
            if ((SLOT) == 1 || (SLOT) == 2) rec_[4 + (SLOT)] = __builtin_amdgcn_s_memtime();  \
        }                                                                                     \
    } while (0)
#else
#define SKL_TRACE_MARK(SLOT) do { } while (0)
#endif

// (TIGHT: the 2-column form squeezed into 128 VGPRs -- packed counts, 4-deep row ring -- for 4 waves per SIMD;
// the only form left, the parameter is kept for the kernels' names.)
// MB: rows walked plane by plane together (1: row by row).  A chunk is walked in blocks of MB rows, and
// inside a block plane pair q of all MB rows comes before plane pair q + 1 of any.  Column register
// b[.][q] is then free -- and re-loaded with the next chunk's data -- 6/7 of a BLOCK before its first
// use there instead of one row before: 3.4 rows of VALU work with MB = 4.  Costs 4 (MB - 1) registers
// for the blocks' mismatch accumulators, so only the 32-row form (3 waves per SIMD, registers to
// spare) has it.  Worth 1.2-1.4 % at n = 8 000 ... 16 000 (MB = 1 / 2 / 4 / 8: 30.34 / 29.95 / 29.98 /
// 30.07 ms at n = 16 000, profiles/r02_ab_row_blocks.jsonl): the s_waitcnt share of the SQ counters is
// mostly the LDS reads of a kernel that runs at the LDS's rate, not the column loads.
// OCC: waves per SIMD the register allocator is held to (0: 4 for the tight 16-row form, 3 for the tight
// 32-row form and for 3 columns per lane, 1 otherwise).
// BIG: sketches beyond 65 535 bins (k-sliced counts / single-k forms only): the k-mer length is walked in segments
// whose totals fit the u16 fields and are added up in 32 bits (pair_kslice_walk.inc).
// PRUNE: tile pruning of the symmetric self kNN (PairArgs::prune_q_*; the single-k 32 x 128 form only): at a few stage
// boundaries every wave asks whether each pair's count on ITS chunks has reached the pair's per-wave bound, the 4 answers
// meet in LDS, and a tile that is hopeless as a whole is left unfinished and unwritten (pair_kslice_walk.inc).
// FUSE: the core/accessory epilogue of a k-sliced counts launch inside the launch (PairArgs::fuse_counter): the workgroup that
// completes a tile's k-mer lengths turns the tile's counts into (core, acc).
template <int R, int JL, int MODE, bool KSL, int ABL = 0, bool TIGHT = false, int MB = 1, int OCC = 0, bool BIG = false, bool PRUNE = false, bool FUSE = false>
__global__ __launch_bounds__(LANES *WAVES_PER_WG, OCC != 0 ? OCC : (TIGHT ? (R > 16 ? 3 : 4) : (JL == 3 ? 3 : 1))) void pair_kernel_kslice(const PairArgs g)
{
    static_assert(!PRUNE || (KSL && MODE == MODE_JACCARD && !BIG && JL == 2), "tile pruning: single-k keys, two columns per lane");
    static_assert(!FUSE || (KSL && MODE == MODE_COUNTS && !BIG && JL == 2), "fused epilogue: k-sliced counts, whole k-mer lengths");
    constexpr int W = WAVES_PER_WG;
    constexpr int CH = R > 16 ? 1 : 2;            // chunks per wave per stage
    constexpr int PIECES = R * CH * 7;            // 16-byte pieces per wave-stage
    constexpr int PPL = (PIECES + LANES - 1) / LANES;   // DMA instructions per wave-stage
    constexpr int P = R * JL;                     // pairs per lane
    constexpr int PX = P / 2;                     // packed (2 x u16) partial counts per lane
    constexpr int SLOTS = PX / W;                 // packed slots finished by each wave
    static_assert(P % 2 == 0 && PX % W == 0, "packed slots must split over the waves");
    static_assert(PPL * LANES - PIECES < PIECES, "tail pieces wrap at most once");
    static_assert(!KSL || MODE != MODE_COREACC, "k-sliced core/acc runs COUNTS + the epilogue kernel");
    // LDS: the row buffers [W][2][PPL * LANES] x 16 B.  The per-k reduction words of a wave
    // (PX per lane) go into the row buffer it has just finished with, XIN per lane; with 3 and 4
    // columns per lane the rest (PX - XIN per lane) go to an overflow region behind the row buffers.
    constexpr uint32_t BUF_U4 = PPL * LANES;
    constexpr uint32_t ROWS_U4 = W * 2 * BUF_U4;
    // (a k-sliced workgroup walks ONE k-mer length: no next stage is in flight when it reduces, so both row
    // buffers of a wave are free -- the 32-row form then needs no overflow region: 32 KB instead of 48 KB of
    // LDS per workgroup, i.e. room for 4 workgroups per CU)
    // (not with segments: the next segment's first stage IS in flight at the end of a segment)
    constexpr uint32_t RED_BUFS = (KSL && !BIG) ? 2u : 1u;
    // RED2 (the all-k 32-row form held to 4 waves per SIMD): the words that do not fit the one free buffer go
    // through it in a second phase (two more barriers per k-mer length) instead of an overflow region -- 32 KB
    // of LDS (36 KB with core/accessory's turned tile) instead of 48 KB, i.e. room for 4 workgroups per CU
    constexpr bool RED2 = TIGHT && (!KSL || BIG) && R == 32 && OCC == 4;
    constexpr int RED_PHASES = RED2 ? 2 : 1;
    // Row DMA addressed as scalar base + 32-bit per-lane offset (pair_kslice_walk.inc): the lane part of the address -- which
    // row, chunk of the stage and plane pair a lane fetches -- never changes.  Kept as 64-bit per-lane pointers it was what the 4-wave all-k form spilled
    // at every stage (a scratch reload waits for the column prefetch too): 5-6 % slower than 3 waves with them, 6 % FASTER without.
    constexpr bool SADDR_DMA = TIGHT && (ABL & 4) == 0;
    constexpr int XIN_FIT = (RED_BUFS * BUF_U4 * 4 / LANES) < (uint32_t)PX ? (int)(RED_BUFS * BUF_U4 * 4 / LANES) : PX;
    constexpr int XIN = RED2 ? PX / RED_PHASES : XIN_FIT;
    static_assert(!RED2 || (XIN <= XIN_FIT && SLOTS % RED_PHASES == 0), "a phase's words fit the free row buffer");
    constexpr int XOV = RED2 ? 0 : PX - XIN;
    constexpr uint32_t TURNED_U4 = (MODE == MODE_COREACC) ? (uint32_t)(JL * 64 * (R + 4) * 8 + 15) / 16u : 0u;   // see the turned tile below
    constexpr uint32_t RED_U4 = RED2 ? (TURNED_U4 > ROWS_U4 ? TURNED_U4 - ROWS_U4 : 0u) : (uint32_t)(W * XOV * LANES) / 4u;
    // PRUNE: behind the row buffers, the column samples' per-wave bounds (JL * 64 words, the same values written by every
    // wave) and two rotating rows of 4 votes
    constexpr uint32_t PRUNE_U4 = PRUNE ? (uint32_t)(JL * LANES) / 4u + 2u + (uint32_t)R / 4u + 2u : 0u;   // column bounds, votes, row bounds, the probe's masks
    __shared__ uint4 lds_all[ROWS_U4 + RED_U4 + PRUNE_U4];
    uint4 (*lds_rows)[2][BUF_U4] = reinterpret_cast<uint4 (*)[2][BUF_U4]>(&lds_all[0]);

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef SKL_TRACE
    if (lane == 0) {
        uint32_t hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        skl_trace[((size_t)blockIdx.x * W + wave) * SKL_TRACE_WORDS + 4] = ((uint64_t)xcc << 32) | hw;
    }
#endif
    SKL_TRACE_MARK(0);

    // blockIdx -> (XCD, tile slot on that XCD[, k-mer length]): the k slices of a tile are
    // neighbours in the per-XCD order
    const uint32_t xcd = blockIdx.x & ((1u << g.xcd_shift) - 1u), s_idx = blockIdx.x >> g.xcd_shift;
    // k-sliced: blocks of KSL_TILE_BLOCK consecutive tiles of an XCD walk one k-mer length together
    // (tile index fastest, then k, then block), so that the workgroups resident on an XCD at one time
    // share a (column group, k) plane of the lane slab in its L2: 2 % at n = 4 000 ... 8 000 against
    // k fastest (profiles/r02_ab_korder_l2prefetch.jsonl)
    constexpr uint32_t KB = KSL_TILE_BLOCK;
    // (MODE_COUNTS launches may also cut a k-mer length into g.k_slices chunk ranges, one workgroup
    // each -- more, shorter workgroups for launches that would otherwise fill the chip 1.4 times;
    // slice s of k index kk stores its counts as "k index" s * k_count + kk, summed by the epilogue)
    // TAIL SLICING (g.tail_slices > 1, instead of the uniform slices): the workgroups of an XCD up to
    // index tail_first -- its whole rounds of resident workgroups -- are whole units, the ones after
    // are chunk slices of the remaining units.  The last round of a launch is then made of short
    // workgroups that spread over all SIMDs instead of a few long ones that run 1-2 per SIMD at a lone
    // wave's issue interval, and the rounds before it pay nothing.  Slice 0 stores, the others add
    // into plane 1 (see kernels.h).
    const bool tail_mode = KSL && MODE == MODE_COUNTS && g.tail_slices > 1u;
    const bool in_tail = tail_mode && s_idx >= g.tail_first;
    const uint32_t n_slices = in_tail ? g.tail_slices : (KSL && MODE == MODE_COUNTS && !tail_mode ? g.k_slices : 1u);
    const uint32_t u_idx = in_tail ? g.tail_first + (s_idx - g.tail_first) / n_slices : s_idx;   // unit index (tail mode) / workgroup index
    // (the last block of an XCD may be short: the grid has exactly tiles_per_xcd x k x slices workgroups
    // per XCD, no padding slots that would be dispatched only to exit)
    const uint32_t per_blk = KB * g.k_count * (tail_mode ? 1u : n_slices);
    const uint32_t blk_ = u_idx / per_blk, rem_ = u_idx - blk_ * per_blk;
    const uint32_t in_blk = KSL ? min(KB, g.tiles_per_xcd - min(g.tiles_per_xcd, blk_ * KB)) : 1u;
    if (KSL && in_blk == 0u) return;
    const uint32_t slot = KSL ? blk_ * KB + rem_ % in_blk : s_idx;
    const uint32_t kslot = KSL ? rem_ / in_blk : 0u;
    const uint32_t kk0 = tail_mode ? kslot : kslot / n_slices;  // first k index of this workgroup
    const uint32_t slice = in_tail ? (s_idx - g.tail_first) % n_slices : (tail_mode ? 0u : kslot - kk0 * n_slices);
    const uint32_t nkk = KSL ? 1u : g.k_count;                  // k-mer lengths it walks
    // WAVE PRIORITY BY ROUND.  A SIMD issues for its oldest wave first, so the workgroups of a launch's
    // second round -- dispatched into the slots the first round's oldest waves free -- get what three
    // older waves leave them until those finish, and then run on alone at a lone wave's issue interval:
    // that is the tail of every launch of 1-2.5 rounds.  Raising the priority of each later round by one
    // (s_setprio, capped at 3) lets them work from the moment they arrive: -3 ... -4 % at 900-1 200
    // genomes core/accessory, -7 % at 2 000 single-k (a second round of a few workgroups), +2 % at
    // 2.5-2.7 rounds and nothing beyond, so the launcher asks for it up to 2.25 rounds
    // (profiles/r02_ab_round_priority.jsonl).  Not for sliced launches (four quarter-rounds of short
    // workgroups: +6 % at 800 genomes).
    if (KSL && !tail_mode && g.round_size != 0u) {
        const uint32_t round_ = s_idx / g.round_size;
        if (round_ == 1u) __builtin_amdgcn_s_setprio(1);
        else if (round_ == 2u) __builtin_amdgcn_s_setprio(2);
        else if (round_ >= 3u) __builtin_amdgcn_s_setprio(3);
    }
    // chunk range of this workgroup: whole stages per slice, the last slice takes what is left (any sketch size)
    const uint32_t per_slice = g.slice_chunks != 0u ? g.slice_chunks : g.ss64 / n_slices;
    const uint32_t c_begin = n_slices > 1u ? min(g.ss64, slice * per_slice) : 0u;
    const uint32_t c_end = n_slices > 1u ? min(g.ss64, c_begin + per_slice) : g.ss64;
    uint32_t jg, at;  // column group (JL blocks of 64), row tile
    if (!lookup_tile_at(g, xcd, slot, jg, at)) return;
    if constexpr ((ABL & 8) != 0) {   // timing only: every workgroup computes tile (0, group 1): all operands cache-hot
        jg = 1;
        at = 0;
    }
    const uint32_t jb0 = jg * JL;
    const uint32_t a0 = g.row_begin + at * R;
    if (jb0 >= g.n_jblocks) return;
    if (a0 >= g.row_end) return;
    if (g.self_mode && a0 >= (jg + 1u) * JL * 64u - 1u) return;   // tile entirely on/below the diagonal
    if constexpr (KSL && MODE == MODE_COUNTS && !FUSE) {
        // EARLY BREAK DECIDED PER BLOCK (PairArgs::block_ke): this workgroup counts k index kk0 of its tile; the tile's block of
        // sample ids may want fewer lengths than the launch carries planes for
        if (g.block_ke != nullptr) {
            const uint32_t r_last = min(a0 + (uint32_t)R, g.row_end) - 1u;
            const uint32_t bc = (jb0 * 64u) >> g.blk_shift_c;
            const uint32_t ke_a = g.block_ke[(size_t)(a0 >> g.blk_shift_r) * g.blk_cols + bc];
            const uint32_t ke_b = g.block_ke[(size_t)(r_last >> g.blk_shift_r) * g.blk_cols + bc];
            if (kk0 >= max(ke_a, ke_b)) return;
        }
    }
    // HALF TILES (2 columns per lane).  A 64-column block of the tile that holds no pair of the launch is
    // not walked: block 0 of a tile that straddles the diagonal when every column of it is <= the
    // tile's first row (self mode: 4 of the 8 diagonal row tiles of every column group at 16 rows, 2 of
    // 4 at 32 -- 6.5 % of a 1 000-genome launch's lane-pairs), block 1 when the launch's columns end
    // in block 0 (an odd number of 64-column blocks).  The chunk walk below is instantiated for the
    // three cases and chosen per workgroup (uniform branch); the skipped block's column registers are
    // never loaded and its count fields stay 0 (its pairs are invalid and never stored).
    constexpr bool HALF_TILES = R == 16 && JL == 2 && TIGHT && ABL == 0;
#ifdef SKL_AB
    // (A/B build only: SKL_HALF_TILES=0 walks every block.  The run-time condition costs the 16-row k-sliced
    // form 11 registers spilled outside its loops -- 44 B of scratch per lane, 26 MB of HBM traffic per
    // 1 000-genome launch -- so the product library does not carry it.)
    const bool half_ok = HALF_TILES && g.no_half_tiles == 0u;
#else
    constexpr bool half_ok = HALF_TILES;
#endif
    const bool skip0 = half_ok && g.self_mode && a0 + 1u >= (jb0 + 1u) * 64u;
    const bool skip1 = half_ok && jb0 + 1u >= g.n_jblocks;
    // The walk is TEXTUALLY included once per case (pair_kslice_walk.inc; SKL_J0 / SKL_J1 = the column blocks
    // [J0, J1) it walks), not a lambda or a function template: wrapping it in either changes hipcc's register
    // allocation of the 32-row form (168 VGPRs + 144 B of spills per lane instead of 144 VGPRs).  No value of
    // one case is live in another, so the cases do not add to each other's register pressure.
    if constexpr (HALF_TILES) {
        if (skip0) {
#define SKL_J0 1
#define SKL_J1 JL
#include "pair_kslice_walk.inc"
#undef SKL_J0
#undef SKL_J1
        } else if (skip1) {
#define SKL_J0 0
#define SKL_J1 1
#include "pair_kslice_walk.inc"
#undef SKL_J0
#undef SKL_J1
        } else {
#define SKL_J0 0
#define SKL_J1 JL
#include "pair_kslice_walk.inc"
#undef SKL_J0
#undef SKL_J1
        }
    } else {
#define SKL_J0 0
#define SKL_J1 JL
#include "pair_kslice_walk.inc"
#undef SKL_J0
#undef SKL_J1
    }
    SKL_TRACE_MARK(3);
    if constexpr (FUSE) {
        // FUSED EPILOGUE.  This workgroup's counts (one k-mer length of the tile) were stored write-through (agent scope);
        // every storing wave drains its stores, the workgroup meets, ONE lane adds to the tile's arrival counter (agent scope,
        // returning), and the workgroup whose add was the tile's k_count-th reads all k_count planes of the tile back with
        // agent-scope loads -- it loads only after its add has returned, its other waves after the barrier that lane then
        // joins -- and finishes the tile's pairs.  No fence on either side, no dependence on where the tile's workgroups
        // ran (MI355X_MICROARCH.md, inter-workgroup visibility: write-through stores + drained counter add + agent loads).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        uint32_t *flag = reinterpret_cast<uint32_t *>(&lds_all[0]);   // (the row buffers are dead: every wave is past its last LDS read)
        if (tid == 0u) {
            const uint32_t old = __hip_atomic_fetch_add(&g.fuse_counter[xcd * g.tiles_per_xcd + slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            flag[0] = (old + 1u) % g.k_count == 0u ? 1u : 0u;
        }
        __syncthreads();
        if (flag[0] != 0u && !(g.fuse_variant & 2u)) {
            constexpr uint32_t PER = (uint32_t)(R * JL * 64) / (LANES * W);   // pairs per thread
            constexpr uint32_t BATCH = 8;                                     // ... taken 8 at a time (registers)
            constexpr uint32_t ROW_STEP = (LANES * W) / (uint32_t)(JL * 64);
            const uint32_t *counts = (const uint32_t *)g.out;
            const uint32_t c = tid % (uint32_t)(JL * 64), jc_ = jb0 * 64u + c;
#pragma clang loop unroll(disable)
            for (uint32_t m0 = 0; m0 < PER; m0 += BATCH) {
                uint32_t same[BATCH][MAX_FUSED_K];
                uint64_t at_[BATCH];
                bool ok[BATCH];
#pragma unroll
                for (uint32_t m = 0; m < BATCH; ++m) {   // every load of the batch first: independent, all in flight together
                    const uint32_t i_ = a0 + tid / (uint32_t)(JL * 64) + (m0 + m) * ROW_STEP;
                    ok[m] = pair_valid(g, i_, jc_);
                    at_[m] = ok[m] ? pair_out_index(g, i_, jc_) : 0ull;
#pragma unroll
                    for (uint32_t t = 0; t < (uint32_t)MAX_FUSED_K; ++t) {
                        same[m][t] = (ok[m] && t < g.k_count)
                                         ? __hip_atomic_load(&counts[at_[m] * g.cnt_pair_stride + (uint64_t)t * g.cnt_k_stride], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                         : 0u;
                    }
                }
#pragma unroll
                for (uint32_t m = 0; m < BATCH; ++m) {
                    const uint32_t i_ = a0 + tid / (uint32_t)(JL * 64) + (m0 + m) * ROW_STEP;
                    if (ok[m]) {
                        if (g.fuse_variant & 4u) ((float2 *)g.fuse_out)[at_[m]] = make_float2((float)(same[m][0] + same[m][1] + same[m][2]), (float)(same[m][3] + same[m][4] + same[m][5]));   // EXPERIMENT
                        else ((float2 *)g.fuse_out)[at_[m]] = coreacc_value_counts(g, i_, jc_, same[m]);
                    }
                }
            }
        }
    }
}

template <int R, int JL, bool KSL, bool TIGHT = false, int MB = 1, int OCC = 0>
static hipError_t launch_rjk(const PairArgs &args, int mode, dim3 grid, hipStream_t stream)
{
    const dim3 block(LANES * WAVES_PER_WG);
    if constexpr (KSL && R == 32 && OCC == 4) {   // tile pruning: the single-k 32 x 128 form (the symmetric self kNN's bands)
        if (args.prune_q_rows != nullptr && args.seg_chunks == 0u && mode == MODE_JACCARD && !args.has_comp) {   // (keys of the count alone)
            hipLaunchKernelGGL((pair_kernel_kslice<R, JL, MODE_JACCARD, true, 0, TIGHT, MB, OCC, false, true>), grid, block, 0, stream, args);
            return hipGetLastError();
        }
    }
#ifdef SKL_AB
    // (A/B build only: the fused core/accessory epilogue LOST to the second launch -- profiles/r05_fused_epilogue.md -- and is
    // kept as the measured record of that)
    if constexpr (KSL && OCC == (R == 32 ? 4 : 0)) {   // fused core/accessory epilogue: the shipped k-sliced counts forms
        if (args.fuse_counter != nullptr && args.seg_chunks == 0u && mode == MODE_COUNTS) {
            hipLaunchKernelGGL((pair_kernel_kslice<R, JL, MODE_COUNTS, true, 0, TIGHT, MB, OCC, false, false, true>), grid, block, 0, stream, args);
            return hipGetLastError();
        }
    }
#endif
    if constexpr (KSL && OCC == (R == 32 ? 4 : 0)) {   // (the shipped k-sliced forms only)
        if (args.seg_chunks != 0u) {   // sketches beyond 65 535 bins: the segmented walk
            if (mode == MODE_COUNTS) hipLaunchKernelGGL((pair_kernel_kslice<R, JL, MODE_COUNTS, true, 0, TIGHT, MB, OCC, true>), grid, block, 0, stream, args);
            else if (mode == MODE_JACCARD) hipLaunchKernelGGL((pair_kernel_kslice<R, JL, MODE_JACCARD, true, 0, TIGHT, MB, OCC, true>), grid, block, 0, stream, args);
            else return hipErrorInvalidValue;
            return hipGetLastError();
        }
    }
    if (args.seg_chunks != 0u) return hipErrorInvalidValue;
    switch (mode) {
        case MODE_COUNTS:
            hipLaunchKernelGGL((pair_kernel_kslice<R, JL, MODE_COUNTS, KSL, 0, TIGHT, MB, OCC>), grid, block, 0, stream, args);
            break;
        case MODE_JACCARD:
            hipLaunchKernelGGL((pair_kernel_kslice<R, JL, MODE_JACCARD, KSL, 0, TIGHT, MB, OCC>), grid, block, 0, stream, args);
            break;
        case MODE_COREACC:
            if constexpr (!KSL) {
                hipLaunchKernelGGL((pair_kernel_kslice<R, JL, MODE_COREACC, false, 0, TIGHT, MB, OCC>), grid, block, 0, stream, args);
                break;
            }
            return hipErrorInvalidValue;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

bool kslice_supported(const PairArgs &args, int mode, bool k_sliced)
{
    if (args.k_count < 1u) return false;
    // u16 count fields: beyond 65 535 bins only the k-sliced counts / single-k forms, walked in segments
    if (args.ss64 > (uint32_t)KSLICE_MAX_U16_CHUNKS && !(k_sliced && (mode == MODE_COUNTS || mode == MODE_JACCARD))) return false;
    // the row DMA's per-lane byte offsets (up to 31 rows + 1 chunk from the tile's first row) are 32-bit
    if ((uint64_t)32u * args.nk * args.ss64 * BBITS * sizeof(uint64_t) >= (1ull << 32)) return false;
    if (mode == MODE_COREACC) return !k_sliced && args.k_count <= (uint32_t)MAX_FUSED_K;
    return mode == MODE_COUNTS || mode == MODE_JACCARD;
}

// shape = 165 / 325: 16 x 128 and 32 x 128 tiles (the product library's shapes); the A/B build (-DSKL_AB) also has their
// round-2/3 forms (1651, 1652, 3254, 3255) and two timing-only ablations
hipError_t launch_pair_kernel_kslice(const PairArgs &args_in, int mode, int shape, bool k_sliced, int ablate,
                                     TileScratch &scratch, hipStream_t stream)
{
    PairArgs args = args_in;
    if (args.row_end <= args.row_begin || args.nB == 0) return hipSuccess;
    if (!kslice_supported(args, mode, k_sliced)) return hipErrorInvalidValue;
    // beyond 65 535 bins: segments of KSLICE_SEG_CHUNKS chunks (a multiple of every form's chunks per stage)
    args.seg_chunks = args.ss64 > (uint32_t)KSLICE_MAX_U16_CHUNKS ? (uint32_t)KSLICE_SEG_CHUNKS : 0u;
    // tile pruning exists in the single-k 32 x 128 form, cross-mode launches, both bounds given
    if (!(shape == 325 && k_sliced && mode == MODE_JACCARD && args.seg_chunks == 0u && args.self_mode == 0u && args.prune_q_cols != nullptr &&
          args.t_bits != nullptr && args.r_bits != nullptr)) {
        args.prune_q_rows = args.prune_q_cols = nullptr;
    }
    if (args.seg_chunks != 0u && shape != 165 && shape != 325) return hipErrorInvalidValue;
    const int R = shape > 1000 ? shape / 100 : shape / 10;   // 165 / 325 (and 165x, 325x in the A/B build): 16 x 128 and 32 x 128 tiles
    const int JL = 2;
    uint64_t n_wg = 0;
    const hipError_t pe = plan_tiles(args, (uint32_t)R, (uint32_t)JL * 64u, scratch, stream, &n_wg);
    if (pe != hipSuccess) return pe;
    if (n_wg == 0) return hipSuccess;
    // k-sliced: exactly tiles_per_xcd x k [x slices] workgroups per XCD (the last tile block is short)
    if (!(k_sliced && mode == MODE_COUNTS) || args.k_slices == 0) args.k_slices = 1;
    if (!(k_sliced && mode == MODE_COUNTS)) args.tail_slices = 0;
    // the fused epilogue (A/B build): whole k-mer lengths only (no chunk slices), the two shipped shapes
#ifndef SKL_AB
    if (args.fuse_counter != nullptr) return hipErrorInvalidValue;
#endif
    if (args.fuse_counter != nullptr && !(k_sliced && mode == MODE_COUNTS && args.k_slices == 1u && args.tail_slices <= 1u && args.seg_chunks == 0u &&
                                          (shape == 165 || shape == 325) && args.k_count <= (uint32_t)MAX_FUSED_K && args.fuse_out != nullptr)) {
        return hipErrorInvalidValue;   // (the caller skips the epilogue launch when it asks for the fused one)
    }
    if (args.tail_slices > 1u) {
        // whole units for the XCD's whole rounds of resident workgroups, slices for the rest
        // (counted on the real units: the padding slots of the last tile block exit at once)
        if (args.tail_resident == 0) return hipErrorInvalidValue;
        const uint32_t units_x = args.tiles_per_xcd * args.k_count;
        args.tail_first = units_x / args.tail_resident * args.tail_resident;
        args.k_slices = 1;
    }
    {   // whole stages per slice, every slice holds something
        const uint32_t S = args.tail_slices > 1u ? args.tail_slices : args.k_slices;
        if (S > 1u) {
            if (args.slice_chunks == 0u) {
                if (args.ss64 % (S * 8u) != 0) return hipErrorInvalidValue;
            } else if (args.slice_chunks % 8u != 0 || (uint64_t)args.slice_chunks * (S - 1u) >= args.ss64 || (uint64_t)args.slice_chunks * S < args.ss64) {
                return hipErrorInvalidValue;
            }
        } else {
            args.slice_chunks = 0;
        }
    }
    // wave priority by round: for launches of up to 2.25 rounds of workgroups (it costs 2 % at 2.5-2.7 rounds
    // and is neutral beyond; profiles/r02_ab_round_priority.jsonl)
    if (!k_sliced || (uint64_t)args.tiles_per_xcd * args.k_count * args.k_slices * 4u > 9ull * args.round_size) args.round_size = 0;
    if (k_sliced) {
        const uint64_t units_pad = (uint64_t)args.tiles_per_xcd * args.k_count;   // exact: the last tile block of an XCD is short
        n_wg = ((uint64_t)units_pad << args.xcd_shift) * args.k_slices;
        if (args.tail_slices > 1u) {
            const uint64_t first = std::min<uint64_t>(args.tail_first, units_pad);
            args.tail_first = (uint32_t)first;
            n_wg = (first + (units_pad - first) * args.tail_slices) << args.xcd_shift;
        }
    }
    if (n_wg >= (1ull << 31)) return hipErrorInvalidValue;
    const dim3 grid((unsigned)n_wg);
#ifdef SKL_AB
    // timing-only: rows read as if the row slab were tile-major ([row tile][k][chunk][row][plane]: a wave's
    // stage is one contiguous 3.5 KB run instead of 16-32 runs of 112 B, one per sample)
    // timing-only ablations (outputs wrong by construction): SKL_KSLICE_ABLATE = 8 all workgroups on one hot tile, 4 a tile-major
    // row slab, 16 the all-k form without its per-k totals in private memory
    if (ablate == 8 && shape == 165 && k_sliced && mode == MODE_COUNTS) {
        hipLaunchKernelGGL((pair_kernel_kslice<16, 2, MODE_COUNTS, true, 8, true, 1>), grid, dim3(LANES * WAVES_PER_WG), 0, stream, args);
        return hipGetLastError();
    }
    if (ablate == 16 && shape == 325 && !k_sliced && mode == MODE_COREACC) {   // timing only: per-k totals not parked
        hipLaunchKernelGGL((pair_kernel_kslice<32, 2, MODE_COREACC, false, 16, true, 2, 4>), grid, dim3(LANES * WAVES_PER_WG), 0, stream, args);
        return hipGetLastError();
    }
    if (ablate == 4 && (shape == 165 || shape == 325)) {
        const dim3 block(LANES * WAVES_PER_WG);
        if (shape == 165 && k_sliced && mode == MODE_COUNTS) hipLaunchKernelGGL((pair_kernel_kslice<16, 2, MODE_COUNTS, true, 4, true, 1>), grid, block, 0, stream, args);
        else if (shape == 325 && k_sliced && mode == MODE_COUNTS) hipLaunchKernelGGL((pair_kernel_kslice<32, 2, MODE_COUNTS, true, 4, true, 4>), grid, block, 0, stream, args);
        else if (shape == 325 && !k_sliced && mode == MODE_COREACC) hipLaunchKernelGGL((pair_kernel_kslice<32, 2, MODE_COREACC, false, 4, true, 4>), grid, block, 0, stream, args);
        else return hipErrorInvalidValue;
        return hipGetLastError();
    }
#else
    (void)ablate;
#endif
    switch (shape) {
        case 165:   // the product shape for launches below 8 Mi evaluations: 16 x 128 tiles, 4 waves per SIMD; k-sliced: walked in
                    // plane-major blocks of 4 rows (121-123 registers; the 14 the scalar-base DMA freed: -1.6 % at cfg 2,
                    // profiles/r03_ab_mb16.jsonl)
            return k_sliced ? launch_rjk<16, 2, true, true, 4>(args, mode, grid, stream)
                            : launch_rjk<16, 2, false, true>(args, mode, grid, stream);
        case 325:   // large launches: 32 x 128 tiles, packed counts, half the column traffic per pair, blocks of 2 rows, 128
                    // registers = 4 waves per SIMD.  k-sliced: 32 KB of LDS (-3 ... -5 % against round 2's 3-wave form,
                    // profiles/r03_ab_occ4.jsonl); all k: per-k reduction in two phases, 32-36 KB of LDS, row DMA with a scalar
                    // base (-6 % against the 3-wave form, profiles/r03_ab_allk_occ4.jsonl)
            return k_sliced ? launch_rjk<32, 2, true, true, 2, 4>(args, mode, grid, stream)
                            : launch_rjk<32, 2, false, true, 2, 4>(args, mode, grid, stream);
#ifdef SKL_AB
        case 1651:   // the 16-row tight form walked row by row (what shipped until the scalar-base DMA freed registers) / in blocks of 2
            return k_sliced ? launch_rjk<16, 2, true, true, 1>(args, mode, grid, stream)
                            : launch_rjk<16, 2, false, true, 1>(args, mode, grid, stream);
        case 1652:
            return k_sliced ? launch_rjk<16, 2, true, true, 2>(args, mode, grid, stream)
                            : launch_rjk<16, 2, false, true, 2>(args, mode, grid, stream);
        case 3255:   // the all-k 32 x 128 form of rounds 2-3a: blocks of 4 rows, 168 registers, 48 KB of LDS, 3 waves per SIMD
            return k_sliced ? launch_rjk<32, 2, true, true, 2, 4>(args, mode, grid, stream)
                            : launch_rjk<32, 2, false, true, 4>(args, mode, grid, stream);
        case 3254:   // the k-sliced 32 x 128 form of round 2: blocks of 4 rows, 143 registers, 3 waves per SIMD
            return k_sliced ? launch_rjk<32, 2, true, true, 4>(args, mode, grid, stream)
                            : launch_rjk<32, 2, false, true, 4>(args, mode, grid, stream);
#endif
        default: return hipErrorInvalidValue;
    }
}

}  // namespace skl
