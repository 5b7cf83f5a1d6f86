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
// Two tile heights in the product library: R = 16 in 128 registers (4 waves per
// SIMD) for launches below 16 M pair x k evaluations, R = 32 (130-162 registers, 3 waves, one chunk
// per wave and stage, chunks walked in plane-major blocks of MB = 4 rows) above: every column
// register then meets 32 rows, which halves the lane-slab bytes per pair.
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

// (3 columns per lane: the register allocator is held to 3 waves per SIMD, 168 VGPRs.  TIGHT: the
// 2-column form squeezed into 128 VGPRs -- packed counts, 4-deep row ring -- for 4 waves per SIMD.)
// MB: rows walked plane by plane together (1: row by row).  A chunk is walked in blocks of MB rows, and
// inside a block plane pair q of all MB rows comes before plane pair q + 1 of any.  Column register
// b[.][q] is then free -- and re-loaded with the next chunk's data -- 6/7 of a BLOCK before its first
// use there instead of one row before: 3.4 rows of VALU work with MB = 4.  Costs 4 (MB - 1) registers
// for the blocks' mismatch accumulators, so only the 32-row form (3 waves per SIMD, registers to
// spare) has it.  Worth 1.2-1.4 % at n = 8 000 ... 16 000 (MB = 1 / 2 / 4 / 8: 30.34 / 29.95 / 29.98 /
// 30.07 ms at n = 16 000, profiles/r02_ab_row_blocks.jsonl): the s_waitcnt share of the SQ counters is
// mostly the LDS reads of a kernel that runs at the LDS's rate, not the column loads.
template <int R, int JL, int MODE, bool KSL, int ABL = 0, bool TIGHT = false, int MB = 1>
__global__ __launch_bounds__(LANES *WAVES_PER_WG, TIGHT ? (R > 16 ? 3 : 4) : (JL == 3 ? 3 : 1)) void pair_kernel_kslice(const PairArgs g)
{
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
    constexpr int XIN = (BUF_U4 * 4 / LANES) < (uint32_t)PX ? (int)(BUF_U4 * 4 / LANES) : PX;
    constexpr int XOV = PX - XIN;
    constexpr uint32_t RED_U4 = (uint32_t)(W * XOV * LANES) / 4u;
    __shared__ uint4 lds_all[ROWS_U4 + RED_U4];
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
    const uint32_t xcd = blockIdx.x & 7u, s_idx = blockIdx.x >> 3;
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
    const uint32_t c_begin = slice * (g.ss64 / n_slices);       // chunk range of this workgroup
    const uint32_t c_end = n_slices > 1u ? c_begin + g.ss64 / n_slices : g.ss64;
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

    const size_t kmer_stride = (size_t)g.ss64 * BBITS;
    const size_t sample_stride = kmer_stride * g.nk;
    // wave w owns chunks w*CH + i + ts*(W*CH), i < CH, of every k-mer length
    const uint32_t stages_per_k = (c_end - c_begin + W * CH - 1) / (W * CH);
    const uint32_t n_stages = stages_per_k * nkk;

    // This wave's partial mismatch counts of the current k (its chunks only).  One register per
    // pair, except with 3 columns per lane, where registers are what decides between 2 and 3 waves
    // per SIMD: columns 0 and 1 of a row then share one register as u16 fields (a wave's share of
    // one k-mer length is at most 64 * 256 mismatches per pair: kslice_supported) and column 2 has
    // its own -- 2 R registers instead of 3 R, for one v_lshl_add_u32 more per (row, chunk).
    static_assert(!TIGHT || JL == 2, "the tight form exists for 2 columns per lane");
    constexpr bool PACK01 = JL == 3 || TIGHT;
    constexpr int NCNT = PACK01 ? (JL == 3 ? 2 * R : R) : P;
    uint32_t cnt[NCNT];
#pragma unroll
    for (int x = 0; x < NCNT; ++x) cnt[x] = 0;
    // packed word x (two u16 fields) of the per-k reduction, and the pair its field h stands for
    auto packed_word = [&](int x) -> uint32_t {
        if constexpr (PACK01 && JL == 2) {
            return cnt[x];
        } else if constexpr (PACK01) {
            return x < R ? cnt[x] : (cnt[R + 2 * (x - R)] | (cnt[R + 2 * (x - R) + 1] << 16));
        } else {
            return cnt[2 * x] | (cnt[2 * x + 1] << 16);
        }
    };
    auto field_pair = [](uint32_t x, uint32_t h, uint32_t &r, uint32_t &j) {
        if constexpr (PACK01 && JL == 2) {
            r = x;
            j = h;
        } else if constexpr (PACK01) {
            if (x < (uint32_t)R) {
                r = x;
                j = h;
            } else {
                r = 2u * (x - (uint32_t)R) + h;
                j = 2u;
            }
        } else {
            const uint32_t pair = 2u * x + h;
            r = pair / JL;
            j = pair % JL;
        }
    };
    // MODE_COREACC: totals of this wave's packed slots, one word per k-mer length (two u16
    // fields per word: the slot's two pairs).  Private (scratch) memory on purpose: written
    // once per k-mer length, read once at the end, and 24 registers cheaper.
    volatile uint32_t hist[(MODE == MODE_COREACC) ? SLOTS * MAX_FUSED_K : 1];

    // Row staging: global -> LDS DMA (global_load_lds_dwordx4): piece p = (chunk*R + row)*7 + q
    // of this wave's stage lands at slot p of the wave's buffer.  Chunks past the end of the
    // sketch are clamped to the last one and never used.
#define SKL_STAGE_DMA(T, BUF)                                                                \
    do {                                                                                     \
        const uint32_t k_ = g.k_begin + kk0 + (T) / stages_per_k;                            \
        const uint32_t c0_ = c_begin + ((T) % stages_per_k) * (W * CH) + wave * CH;          \
        _Pragma("unroll") for (int u = 0; u < PPL; ++u)                                      \
        {                                                                                    \
            const uint32_t pp_ = lane + u * 64u;                                             \
            const uint32_t p_ = pp_ < (uint32_t)PIECES ? pp_ : pp_ - (uint32_t)PIECES;       \
            const uint32_t q_ = p_ % 7u, rc_ = p_ / 7u;                                      \
            const uint32_t r_ = rc_ % R, c_ = rc_ / R;                                       \
            const uint32_t cc_ = (c0_ + c_) < g.ss64 ? (c0_ + c_) : (g.ss64 - 1u);           \
            const uint64_t *src_ = (ABL & 4)                                                 \
                ? g.A + ((((size_t)(a0 / R) * g.nk + k_) * g.ss64 + cc_) * R + r_) * BBITS + 2u * q_ /* timing only: a tile-major row slab */ \
                : g.A + (size_t)(a0 + r_) * sample_stride +                                  \
                      (size_t)k_ * kmer_stride + (size_t)cc_ * BBITS + 2u * q_;               \
            skl_dma16(src_, lds_base + (((uint32_t)wave * 2u + (BUF)) * (PPL * LANES) + u * 64u) * 16u); \
        }                                                                                    \
    } while (0)

    const uint32_t lds_base = __builtin_amdgcn_readfirstlane(skl_lds_addr(&lds_all[0]));
    SKL_STAGE_DMA(0u, 0);

    // Next valid chunk of this wave after (kl, ts, ci) in its walk over k-mer lengths, stages
    // and chunks (wave-uniform scalar code); false when the walk is over.
    auto next_chunk = [&](uint32_t kl_, uint32_t ts_, int ci_, uint32_t &k_out, uint32_t &c_out) {
        for (;;) {
            if (++ci_ >= CH) {
                ci_ = 0;
                if (++ts_ >= stages_per_k) {
                    ts_ = 0;
                    ++kl_;
                }
            }
            if (kl_ >= nkk) return false;
            const uint32_t c_ = c_begin + ts_ * (W * CH) + wave * CH + (uint32_t)ci_;
            if (c_ < c_end) {
                k_out = g.k_begin + kk0 + kl_;
                c_out = c_;
                return true;
            }
        }
    };
    auto column_ptr = [&](int j, uint32_t k_, uint32_t c_) {
        const uint32_t jb = (jb0 + j) < g.n_jblocks ? (jb0 + j) : (g.n_jblocks - 1u);   // clamped, never stored
        return g.B + (((size_t)jb * g.nk + k_) * g.ss64 + c_) * (7 * LANES) + lane;
    };

    // Column operand: 2 x 7 x 16 B per lane and chunk, always ONE CHUNK AHEAD with no extra
    // registers: during the last row of a chunk every b[j][q] is re-loaded with the next
    // chunk's data right after its last use (the first of them gets most of a row of lead,
    // the loads return in order, and the compiler's counted vmcnt before each use is exact).
    uint4 b[JL][7];
    {
        uint32_t k1 = g.k_begin + kk0, c1 = c_begin;
        next_chunk(0u, 0u, -1, k1, c1);
#pragma unroll
        for (int j = 0; j < JL; ++j) {
            const uint4 *bp = column_ptr(j, k1, c1);
#pragma unroll
            for (int q = 0; q < 7; ++q) b[j][q] = bp[q * LANES];
        }
    }
    uint32_t b_younger = 1;   // chunks' worth of column loads (JL*7 each) issued after the newest row DMA

    uint32_t t = 0;   // flat stage counter (k-mer lengths x stages)
    for (uint32_t kl = 0; kl < nkk; ++kl) {
        const uint32_t kk = kk0 + kl;
        for (uint32_t ts = 0; ts < stages_per_k; ++ts, ++t) {
            const uint32_t buf = t & 1u;
            const uint32_t c0 = c_begin + ts * (W * CH) + wave * CH;
            // This wave's DMA of stage t must have landed.  VMEM returns in order, so it is
            // enough that only the younger column loads may still be in flight.
            if (b_younger == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else if (b_younger == 1) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(JL * 7) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * JL * 7) : "memory");
            }
            if (t == 0) SKL_TRACE_MARK(1);
            // The next stage's rows are requested after row 0 of this stage's first chunk (all
            // column registers consumed once, so no column load is in flight then); a stage
            // in which this wave has no chunk requests them here.
            const bool want_dma = t + 1 < n_stages;
            if (want_dma && c0 >= c_end) {
                SKL_STAGE_DMA(t + 1, buf ^ 1u);
                b_younger = 0;
            }

            // (unrolled: the column loads issued in the last row of chunk 0 and their first use
            // in chunk 1 are then straight-line code, and the compiler's waits on them are counted
            // instead of the vmcnt(0) it falls back to across a loop back-edge)
#pragma unroll
            for (uint32_t ci = 0; ci < (uint32_t)CH; ++ci) {
                if (c0 + ci >= c_end) break;
                // where the columns of the next chunk are (the current ones again if none follows)
                uint32_t kn = g.k_begin + kk, cn = c0 + ci;
                next_chunk(kl, ts, (int)ci, kn, cn);
                const uint4 *bn[JL];
#pragma unroll
                for (int j = 0; j < JL; ++j) bn[j] = column_ptr(j, kn, cn);
                const uint4 *rows = &lds_rows[wave][buf][(size_t)ci * R * 7];
                // Row operand: a ring of AD plane pairs.  Step s = r * 7 + q of the chunk reads
                // rows[s]; right after its use the register is re-loaded with step s + AD.  AD = 7
                // (a whole row ahead) with up to 2 columns per lane; 4 with 3 or more, where a step
                // is 12+ VALU instructions long and the 12 registers decide the occupancy.
                constexpr int AD = (JL >= 3 || TIGHT) ? 4 : 7;
                uint4 a[AD];
                static_assert(R % MB == 0, "rows per tile must be a multiple of the block height");
                // step s of the chunk -> index of its plane pair in the row buffer ([row][plane pair])
                auto row_slot = [](int s) constexpr { return ((s / (7 * MB)) * MB + s % MB) * 7 + (s % (7 * MB)) / MB; };
#pragma unroll
                for (int q = 0; q < AD; ++q) a[q] = rows[row_slot(q)];
#pragma unroll
                for (int rb = 0; rb < R / MB; ++rb) {
                    uint32_t mlo[MB][JL], mhi[MB][JL];
#pragma unroll
                    for (int q = 0; q < 7; ++q) {
#pragma unroll
                        for (int m = 0; m < MB; ++m) {
                            const int s_ = (rb * 7 + q) * MB + m;      // compile-time after unrolling
                            uint4 &ar = a[s_ % AD];
#pragma unroll
                            for (int j = 0; j < JL; ++j) {
                                // b is stored (hi, lo) per plane: see device_common.hpp "VGPR banks"
                                if (q == 0) {
                                    mlo[m][j] = ar.x ^ b[j][0].y;
                                    mhi[m][j] = ar.y ^ b[j][0].x;
                                } else {
                                    mlo[m][j] = acc_mismatch_vvv(mlo[m][j], ar.x, b[j][q].y);
                                    mhi[m][j] = acc_mismatch_vvv(mhi[m][j], ar.y, b[j][q].x);
                                }
                                mlo[m][j] = acc_mismatch_vvv(mlo[m][j], ar.z, b[j][q].w);
                                mhi[m][j] = acc_mismatch_vvv(mhi[m][j], ar.w, b[j][q].z);
                            }
                            // rolling prefetch of the plane pair AD steps ahead (no extra registers)
                            __builtin_amdgcn_sched_barrier(0);
                            if constexpr (ABL & 1) {   // timing-only: no re-read, but opaque to CSE
                                asm volatile("" : "+v"(ar.x), "+v"(ar.y), "+v"(ar.z), "+v"(ar.w));
                            } else {
                                if (s_ + AD < R * 7) ar = rows[row_slot(s_ + AD)];
                            }
                            if constexpr (!(ABL & 2)) {
                                if (rb == R / MB - 1 && m == MB - 1) {   // last use of b[.][q] in this chunk: fetch the next chunk's
#pragma unroll
                                    for (int j = 0; j < JL; ++j) b[j][q] = bn[j][q * LANES];
                                }
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
#pragma unroll
                  for (int m = 0; m < MB; ++m) {
                    const int r = rb * MB + m;
                    // popcount with the add fused (v_bcnt_u32_b32 d, m, d)
                    if constexpr (PACK01) {
                        asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(cnt[r]) : "v"(mlo[m][0]));
                        asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(cnt[r]) : "v"(mhi[m][0]));
                        uint32_t t1;
                        asm("v_bcnt_u32_b32 %0, %1, 0" : "=v"(t1) : "v"(mlo[m][1]));
                        asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(t1) : "v"(mhi[m][1]));
                        cnt[r] = (t1 << 16) + cnt[r];   // v_lshl_add_u32
                        if constexpr (JL == 3) {
                            asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(cnt[R + r]) : "v"(mlo[m][2]));
                            asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(cnt[R + r]) : "v"(mhi[m][2]));
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < JL; ++j) {
                            asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(cnt[r * JL + j]) : "v"(mlo[m][j]));
                            asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(cnt[r * JL + j]) : "v"(mhi[m][j]));
                        }
                    }
                  }
                    if (rb == 0 && ci == 0 && want_dma) {
                        SKL_STAGE_DMA(t + 1, buf ^ 1u);  // lands under this stage's VALU work
                        b_younger = 0;
                    }
                }
                if constexpr (!(ABL & 2)) ++b_younger;
            }
        }

        // ---- end of a k-mer length: sum the 4 partial counts per pair through LDS ----
        if (kl + 1 == nkk) SKL_TRACE_MARK(2);
        // The buffer this wave consumed last is dead (the next stage was prefetched into the
        // other one), so every wave publishes into its own: [wave][buf][PX][LANES] words.
        const uint32_t dead = (t - 1u) & 1u;
        // word x of wave w: red_at(w, x)
        uint32_t *red_rows = reinterpret_cast<uint32_t *>(&lds_all[0]);
        uint32_t *red_over = reinterpret_cast<uint32_t *>(&lds_all[ROWS_U4]);
        auto red_at = [&](uint32_t w, uint32_t x) -> uint32_t & {
            return x < (uint32_t)XIN ? red_rows[(w * 2u + dead) * (BUF_U4 * 4u) + x * LANES + lane]
                                     : red_over[(w * (uint32_t)XOV + (x - (uint32_t)XIN)) * LANES + lane];
        };
#pragma unroll
        for (int x = 0; x < PX; ++x) red_at(wave, (uint32_t)x) = packed_word(x);
#pragma unroll
        for (int x = 0; x < NCNT; ++x) cnt[x] = 0;
        __syncthreads();
        // wave w finishes packed slots x = w (mod 4); fields stay below 2^16 (ss64 <= 1023)
        float tval[(MODE == MODE_JACCARD && KSL) ? SLOTS * 2 : 1];   // this wave's keys, for out_t
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) {
            const uint32_t x = (uint32_t)i * W + wave;
            uint32_t total = 0;
#pragma unroll
            for (int w = 0; w < W; ++w) total += red_at((uint32_t)w, x);
            if constexpr (MODE == MODE_COREACC) {
                hist[i * MAX_FUSED_K + kl] = total;
            } else {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    uint32_t r, j;
                    field_pair(x, (uint32_t)h, r, j);
                    const uint32_t mism = h ? (total >> 16) : (total & 0xFFFFu);
                    if constexpr (MODE == MODE_COUNTS) {
                        if (in_tail && slice != 0u) {   // tail slices 1.. add into plane 1
                            const uint32_t i_ = a0 + r, jc_ = (jb0 + j) * 64u + lane;
                            if (pair_valid(g, i_, jc_)) {
                                atomicAdd(&((uint32_t *)g.out)[pair_out_index(g, i_, jc_) * g.cnt_pair_stride +
                                                               (uint64_t)(g.k_count + kk) * g.cnt_k_stride],
                                          (c_end - c_begin) * 64u - mism);
                            }
                        } else {
                            store_count(g, a0 + r, (jb0 + j) * 64u + lane, (tail_mode ? 0u : slice * g.k_count) + kk, (c_end - c_begin) * 64u, mism);
                        }
                    } else if constexpr (KSL) {
                        const uint32_t i_ = a0 + r, jc_ = (jb0 + j) * 64u + lane;
                        float v = __builtin_inff();
                        const bool valid_ = pair_valid(g, i_, jc_);
                        if (valid_) {
                            v = jaccard_out_value(g, i_, jc_, mism);
                            ((float *)g.out)[pair_out_index(g, i_, jc_)] = v;
                        }
                        if (g.r_bits != nullptr && i_ < g.row_end) {   // (wave-uniform: i_ is)
                            // symmetric self kNN: does this 64-column block bring row i_ anything below its knn-th best?
                            const uint32_t thr_ = g.r_thr[(size_t)(i_ - g.row_begin) * g.r_thr_stride];
                            if (__ballot(valid_ && sortable_bits(v) < thr_) != 0ull && lane == 0u) {
                                atomicOr(&g.r_bits[(size_t)(i_ - g.row_begin) * g.r_bits_stride + ((jb0 + j) >> 5)], 1u << ((jb0 + j) & 31u));
                            }
                        }
                        tval[i * 2 + h] = v;
                    } else {
                        store_jaccard(g, a0 + r, (jb0 + j) * 64u + lane, mism);
                    }
                }
            }
        }
        if constexpr (MODE == MODE_JACCARD && KSL) {
            // Symmetric self kNN: the keys of columns >= t_col_begin are also candidates of the
            // ROW with that sample id.  The tile is turned through LDS so that a column's R keys
            // leave as R/4 16-byte stores (one 64-byte run per column at R = 16).
            if (g.out_t != nullptr && min((jb0 + (uint32_t)JL) * 64u, g.nB) > g.t_col_begin) {   // workgroup-uniform
                constexpr uint32_t TP = R + 4;   // padded column pitch (floats), keeps 16-byte alignment
                static_assert(JL * 64 * TP * 4 <= (ROWS_U4 + RED_U4) * 16, "the turned tile fits the LDS of the workgroup");
                float *tt = reinterpret_cast<float *>(&lds_all[0]);
                __syncthreads();   // every wave is done with the reduction words
#pragma unroll
                for (int i = 0; i < SLOTS; ++i) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        uint32_t r, j;
                        field_pair((uint32_t)i * W + wave, (uint32_t)h, r, j);
                        tt[(j * 64u + lane) * TP + r] = tval[i * 2 + h];
                    }
                }
                __syncthreads();
                constexpr uint32_t QUADS = R / 4;
                for (uint32_t item = tid; item < (uint32_t)JL * 64u * QUADS; item += LANES * W) {
                    const uint32_t c = item / QUADS, q = item % QUADS;
                    const uint32_t jc = jb0 * 64u + c;
                    if (jc >= g.t_col_begin && jc < g.nB) {
                        const float4 v = *reinterpret_cast<const float4 *>(&tt[c * TP + 4u * q]);
                        *reinterpret_cast<float4 *>(&g.out_t[(size_t)(jc - g.t_col_begin) * g.t_stride +
                                                             (a0 - g.row_begin) + 4u * q]) = v;
                        if (g.t_flag != nullptr) {   // does any of the four beat column jc's knn-th best? (invalid records are +inf)
                            const float best = fminf(fminf(v.x, v.y), fminf(v.z, v.w));
                            if (sortable_bits(best) < g.t_thr[(size_t)jc * g.t_thr_stride]) g.t_flag[jc] = g.t_flag_value;
                        }
                    }
                }
            }
        }
        // the next stage's DMA of every wave goes into the buffer just read: fence the reads
        if (kl + 1 < nkk) __syncthreads();
    }
#undef SKL_STAGE_DMA

    if constexpr (MODE == MODE_COREACC) {
        // symmetric self kNN: (core, acc) of columns >= t_col_begin also leave turned, see MODE_JACCARD
        constexpr uint32_t TP = R + 4;
        static_assert(JL * 64 * TP * 8 <= (ROWS_U4 + RED_U4) * 16, "the turned tile fits the LDS of the workgroup");
        const bool turned = g.out_t != nullptr && min((jb0 + (uint32_t)JL) * 64u, g.nB) > g.t_col_begin;   // workgroup-uniform
        float2 *tt = reinterpret_cast<float2 *>(&lds_all[0]);
        if (turned) __syncthreads();   // every wave is done with the reduction words
        // store_coreacc() takes u16 fields, newest k lowest, as s2:s1:s0
#pragma clang loop unroll(disable)
        for (int i = 0; i < SLOTS; ++i) {
            const uint32_t x = (uint32_t)i * W + wave;
            uint32_t word[MAX_FUSED_K];   // word[f]: totals of the k-mer length f steps from the newest
#pragma unroll
            for (int f = 0; f < MAX_FUSED_K; ++f) {
                word[f] = (uint32_t)f < nkk ? hist[i * MAX_FUSED_K + (nkk - 1u - f)] : 0u;
            }
#pragma clang loop unroll(disable)
            for (int h = 0; h < 2; ++h) {
                uint32_t r, j;
                field_pair(x, (uint32_t)h, r, j);
                const uint32_t sh = h * 16u;
                const uint32_t s0 = ((word[0] >> sh) & 0xFFFFu) | (((word[1] >> sh) & 0xFFFFu) << 16);
                const uint32_t s1 = ((word[2] >> sh) & 0xFFFFu) | (((word[3] >> sh) & 0xFFFFu) << 16);
                const uint32_t s2 = ((word[4] >> sh) & 0xFFFFu) | (((word[5] >> sh) & 0xFFFFu) << 16);
                const uint32_t i_ = a0 + r, jc_ = (jb0 + j) * 64u + lane;
                float2 v = make_float2(__builtin_inff(), __builtin_inff());
                const bool valid_ = pair_valid(g, i_, jc_);
                if (valid_) {
                    v = coreacc_value(g, i_, jc_, s0, s1, s2);
                    ((float2 *)g.out)[pair_out_index(g, i_, jc_)] = v;
                }
                if (g.r_bits != nullptr && i_ < g.row_end) {   // see MODE_JACCARD; the key is the core distance
                    const uint32_t thr_ = g.r_thr[(size_t)(i_ - g.row_begin) * g.r_thr_stride];
                    if (__ballot(valid_ && sortable_bits(v.x) < thr_) != 0ull && lane == 0u) {
                        atomicOr(&g.r_bits[(size_t)(i_ - g.row_begin) * g.r_bits_stride + ((jb0 + j) >> 5)], 1u << ((jb0 + j) & 31u));
                    }
                }
                if (turned) tt[(j * 64u + lane) * TP + r] = v;
            }
        }
        if (turned) {
            __syncthreads();
            constexpr uint32_t DUOS = R / 2;   // two records = one 16-byte store
            for (uint32_t item = tid; item < (uint32_t)JL * 64u * DUOS; item += LANES * W) {
                const uint32_t c = item / DUOS, q = item % DUOS;
                const uint32_t jc = jb0 * 64u + c;
                if (jc >= g.t_col_begin && jc < g.nB) {
                    const float4 v = *reinterpret_cast<const float4 *>(&tt[c * TP + 2u * q]);
                    *reinterpret_cast<float4 *>(reinterpret_cast<float2 *>(g.out_t) +
                                                (size_t)(jc - g.t_col_begin) * g.t_stride + (a0 - g.row_begin) + 2u * q) = v;
                    if (g.t_flag != nullptr) {   // two (core, acc) records: the key is the core distance
                        if (sortable_bits(fminf(v.x, v.z)) < g.t_thr[(size_t)jc * g.t_thr_stride]) g.t_flag[jc] = g.t_flag_value;
                    }
                }
            }
        }
    }
    SKL_TRACE_MARK(3);
}

template <int R, int JL, bool KSL, bool TIGHT = false, int MB = 1>
static hipError_t launch_rjk(const PairArgs &args, int mode, dim3 grid, hipStream_t stream)
{
    const dim3 block(LANES * WAVES_PER_WG);
    switch (mode) {
        case MODE_COUNTS:
            hipLaunchKernelGGL((pair_kernel_kslice<R, JL, MODE_COUNTS, KSL, 0, TIGHT, MB>), grid, block, 0, stream, args);
            break;
        case MODE_JACCARD:
            hipLaunchKernelGGL((pair_kernel_kslice<R, JL, MODE_JACCARD, KSL, 0, TIGHT, MB>), grid, block, 0, stream, args);
            break;
        case MODE_COREACC:
            if constexpr (!KSL) {
                hipLaunchKernelGGL((pair_kernel_kslice<R, JL, MODE_COREACC, false, 0, TIGHT, MB>), grid, block, 0, stream, args);
                break;
            }
            return hipErrorInvalidValue;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

bool kslice_supported(const PairArgs &args, int mode, bool k_sliced)
{
    if (args.ss64 > 1023u || args.k_count < 1u) return false;   // u16 count fields
    if (mode == MODE_COREACC) return !k_sliced && args.k_count <= (uint32_t)MAX_FUSED_K;
    return mode == MODE_COUNTS || mode == MODE_JACCARD;
}

// shape = R*10 + JL, or 165 / 325 = the tight forms of 16 x 128 and 32 x 128 (the product library's shapes); the A/B
// build (-DSKL_AB) has the others and the ablations
hipError_t launch_pair_kernel_kslice(const PairArgs &args_in, int mode, int shape, bool k_sliced, int ablate,
                                     TileScratch &scratch, hipStream_t stream)
{
    PairArgs args = args_in;
    if (args.row_end <= args.row_begin || args.nB == 0) return hipSuccess;
    if (!kslice_supported(args, mode, k_sliced)) return hipErrorInvalidValue;
    const int R = shape > 1000 ? shape / 100 : shape / 10;   // 165 / 325 (and 1652, 325x in the A/B build): the tight forms of 16 x 128 and 32 x 128
    const int JL = (shape == 165 || shape == 325 || shape > 1000) ? 2 : shape % 10;
    uint64_t n_wg = 0;
    const hipError_t pe = plan_tiles(args, (uint32_t)R, (uint32_t)JL * 64u, scratch, stream, &n_wg);
    if (pe != hipSuccess) return pe;
    if (n_wg == 0) return hipSuccess;
    // k-sliced: exactly tiles_per_xcd x k [x slices] workgroups per XCD (the last tile block is short)
    if (!(k_sliced && mode == MODE_COUNTS) || args.k_slices == 0) args.k_slices = 1;
    if (!(k_sliced && mode == MODE_COUNTS)) args.tail_slices = 0;
    if (args.tail_slices > 1u) {
        // whole units for the XCD's whole rounds of resident workgroups, slices for the rest
        // (counted on the real units: the padding slots of the last tile block exit at once)
        if (args.ss64 % (args.tail_slices * 8u) != 0 || args.tail_resident == 0) return hipErrorInvalidValue;   // whole stages per slice
        const uint32_t units_x = args.tiles_per_xcd * args.k_count;
        args.tail_first = units_x / args.tail_resident * args.tail_resident;
        args.k_slices = 1;
    }
    if (args.ss64 % (args.k_slices * 8u) != 0 && args.k_slices != 1) return hipErrorInvalidValue;   // whole stages per slice
    // wave priority by round: for launches of up to 2.25 rounds of workgroups (it costs 2 % at 2.5-2.7 rounds
    // and is neutral beyond; profiles/r02_ab_round_priority.jsonl)
    if (!k_sliced || (uint64_t)args.tiles_per_xcd * args.k_count * args.k_slices * 4u > 9ull * args.round_size) args.round_size = 0;
    if (k_sliced) {
        const uint64_t units_pad = (uint64_t)args.tiles_per_xcd * args.k_count;   // exact: the last tile block of an XCD is short
        n_wg = 8ull * units_pad * args.k_slices;
        if (args.tail_slices > 1u) {
            const uint64_t first = std::min<uint64_t>(args.tail_first, units_pad);
            args.tail_first = (uint32_t)first;
            n_wg = 8ull * (first + (units_pad - first) * args.tail_slices);
        }
    }
    if (n_wg >= (1ull << 31)) return hipErrorInvalidValue;
    const dim3 grid((unsigned)n_wg);
#ifdef SKL_AB
    // timing-only: rows read as if the row slab were tile-major ([row tile][k][chunk][row][plane]: a wave's
    // stage is one contiguous 3.5 KB run instead of 16-32 runs of 112 B, one per sample)
    if (ablate == 8 && shape == 165 && k_sliced && mode == MODE_COUNTS) {   // timing only: all workgroups on one hot tile
        hipLaunchKernelGGL((pair_kernel_kslice<16, 2, MODE_COUNTS, true, 8, true, 1>), grid, dim3(LANES * WAVES_PER_WG), 0, stream, args);
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
    // timing-only ablations of the sliced COUNTS kernel (outputs wrong by construction):
    // SKL_KSLICE_ABLATE = 1 no row re-reads from LDS, 2 no column reloads, 3 both
    if (ablate && (shape == 162 || shape == 165) && k_sliced && mode == MODE_COUNTS) {
        const dim3 block(LANES * WAVES_PER_WG);
        if (ablate == 1) hipLaunchKernelGGL((pair_kernel_kslice<16, 2, MODE_COUNTS, true, 1>), grid, block, 0, stream, args);
        else if (ablate == 2) hipLaunchKernelGGL((pair_kernel_kslice<16, 2, MODE_COUNTS, true, 2>), grid, block, 0, stream, args);
        else hipLaunchKernelGGL((pair_kernel_kslice<16, 2, MODE_COUNTS, true, 3>), grid, block, 0, stream, args);
        return hipGetLastError();
    }
#else
    (void)ablate;
#endif
#define SKL_SHAPE(SH, RR, JJ)                                                                  \
    case SH:                                                                                   \
        return k_sliced ? launch_rjk<RR, JJ, true>(args, mode, grid, stream)                   \
                        : launch_rjk<RR, JJ, false>(args, mode, grid, stream);
    switch (shape) {
        case 165:   // the product shape: 16 x 128 tiles, 128 VGPRs
            return k_sliced ? launch_rjk<16, 2, true, true>(args, mode, grid, stream)
                            : launch_rjk<16, 2, false, true>(args, mode, grid, stream);
        case 325:   // large launches: 32 x 128 tiles, packed counts, 3 waves per SIMD, half the column traffic per pair
            return k_sliced ? launch_rjk<32, 2, true, true, 4>(args, mode, grid, stream)
                            : launch_rjk<32, 2, false, true, 4>(args, mode, grid, stream);
#ifdef SKL_AB
        case 3251:   // ... walked row by row
            return k_sliced ? launch_rjk<32, 2, true, true, 1>(args, mode, grid, stream)
                            : launch_rjk<32, 2, false, true, 1>(args, mode, grid, stream);
        SKL_SHAPE(162, 16, 2)
        SKL_SHAPE(81, 8, 1)
        SKL_SHAPE(82, 8, 2)
        SKL_SHAPE(84, 8, 4)
        SKL_SHAPE(161, 16, 1)
        SKL_SHAPE(163, 16, 3)
        SKL_SHAPE(164, 16, 4)
#endif
        default: return hipErrorInvalidValue;
    }
#undef SKL_SHAPE
}

}  // namespace skl
