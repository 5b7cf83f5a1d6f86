// epilogue.hip -- the core/accessory epilogue of the counts + epilogue form, with the EARLY BREAK of the reference's loop
// over the k-mer lengths (gfx950).
//
// Reference: core_acc_dist, src/distances/jaccard.rs:61-101.  The loop over the k-mer lengths leaves at the first one whose
// ln J lies below the tolerance (:89-91) -- J = 0 for a pair that shares no more bins than chance (expected_samebits,
// :26-31) -- and a fit over fewer than three lengths is (1, 1) (:117).  The pair kernel therefore counts only the first
// `ke` lengths of a block of pairs (pair_kslice.hip, k-sliced MODE_COUNTS), and this kernel
//   1. runs the reference's loop over those counts for every pair: a pair that leaves inside them is finished at once;
//   2. parks the pairs STILL IN THE RUNNING in LDS lists (one per wave), which the workgroup then completes together:
//      the four waves share the lists evenly, a wave takes its entries in pair order -- consecutive entries are columns of
//      one row -- holds the ROW sample's slice of the next length in registers across them, and reads each column
//      sample's slice as ONE contiguous run (lane l the l-th half chunk of 7 planes; the two halves of a chunk meet by
//      DPP): the form pair_cand.hip measured at 0.88 of the HBM peak;
//   3. finishes those pairs one per lane, all at once: the same sums in the same order as the reference.
// Round 5 completed a pair where it was found -- a whole wave re-reading BOTH samples' slices per pair, 64 lanes 112
// bytes apart, the other pairs of the wave waiting: 21-31 % of every core/accessory step.
//
// The break test is the reference's: y < tolerance with y = ln J.  Without a completeness correction y is a function of
// the bin-match count alone and non-decreasing in it, so the test is `count < min_alive` (the host finds min_alive in the
// table it uploads and checks the monotonicity; EB_NONE: ask the table).  With a correction (jaccard.rs:36-41) J is scaled
// per pair -- 0 stays 0 -- and y is evaluated with the restated libm logarithm, per pair, as everywhere else.
#include "device_common.hpp"

#include <algorithm>

namespace skl {

namespace {

constexpr uint32_t EB_CAP = 128;       // alive pairs a wave parks before its workgroup completes them
constexpr uint32_t EB_MAXK = 8;        // k-mer lengths of an early-break launch (the driver refuses more)
constexpr uint32_t EB_NONE = 0xFFFFFFFFu;

struct EbLists {                        // LDS of one workgroup: one list per wave
    uint32_t i[4][EB_CAP], j[4][EB_CAP], slot[4][EB_CAP], ke[4][EB_CAP];
    uint32_t same[4][EB_CAP][EB_MAXK];  // bin-match count per k-mer length: the first ke from the pair kernel, the rest from the completion
    uint32_t count[4];
};

// row i's condensed start, inverted: the (i, position in row i) of flat index `flat` (distance_matrix.rs:46-51 with the
// f64 guess fixed up by a search, as coreacc_epilogue_kernel has always done)
__device__ __forceinline__ void eb_locate_self(uint64_t flat, uint64_t n_total, uint32_t &i_out, uint32_t &pos_out)
{
    const double nn = (double)n_total;
    const double guess = nn - 2.0 - floor(sqrt(-8.0 * (double)flat + 4.0 * nn * (nn - 1.0) - 7.0) / 2.0 - 0.5);
    uint64_t i = (uint64_t)(guess < 0.0 ? 0.0 : guess);
    if (i > n_total - 2ull) i = n_total - 2ull;
    while (i > 0 && square_to_condensed_dev(i, i + 1, n_total) > flat) --i;
    while (i + 2 < n_total && square_to_condensed_dev(i + 1, i + 2, n_total) <= flat) ++i;
    i_out = (uint32_t)i;
    pos_out = (uint32_t)(flat - square_to_condensed_dev(i, i + 1, n_total));
}

__device__ __forceinline__ uint32_t eb_count_at(const EpilogueArgs &g, uint64_t idx)
{
    return g.cnt_u16 ? (uint32_t)reinterpret_cast<const uint16_t *>(g.counts)[idx] : g.counts[idx];
}

// ln J of a bin-match count (jaccard.rs:26-44, :88)
template <bool COMP>
__device__ __forceinline__ double eb_lnj(const EpilogueArgs &g, uint32_t same, double c1, double c2)
{
    if constexpr (COMP) return glibc_log(jaccard_from_samebits_dev(same, g.ss64, true, c1, c2, g.cutoff), g.log_variant);
    const uint32_t maxnbits = g.ss64 * 64u;
    return g.ytab[same <= maxnbits ? same : maxnbits];
}

struct EbSums {
    double xsum = 0.0, ysum = 0.0, xysum = 0.0, xsquaresum = 0.0, ysquaresum = 0.0, n = 0.0;
    __device__ __forceinline__ void add(double k_fl, double y)   // jaccard.rs:92-97, in that order
    {
        xsum += k_fl;
        ysum += y;
        xysum += k_fl * y;
        xsquaresum += k_fl * k_fl;
        ysquaresum += y * y;
        n += 1.0;
    }
};

// (one copy per kernel: the regression -- f64 divisions, square roots, exp -- inlined at both of its sites took the kernel
// past 128 registers)
__device__ __noinline__ float2 eb_regress(double xsum, double ysum, double xysum, double xsquaresum, double ysquaresum, double n)
{
    return simple_linear_regression_dev(xsum, ysum, xysum, xsquaresum, ysquaresum, n);
}

#define SKL_DPP_ADD(v, ctrl) ((v) + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), (ctrl), 0xF, 0xF, true))

// Bins two (sample, k) slices share, counted by the whole wave: lane l takes half chunk h = trip * 64 + l (7 planes,
// 56 contiguous bytes; a wave's trip is one run of 3 584 bytes), the other seven planes of its chunk sit in lane l ^ 1.
// `a`: the row slice's seven planes for this lane and trip.  Returns the MISMATCHES of this lane's share (even lanes only).
__device__ __forceinline__ uint32_t eb_trip(const uint2 *a, const uint2 *pj, uint32_t h, uint32_t halves, uint32_t lane)
{
    uint32_t mlo = 0, mhi = 0;
    if (h < halves) {
        uint2 b[7];
#pragma unroll
        for (int q = 0; q < 7; ++q) b[q] = pj[(size_t)h * 7 + q];
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            mlo = acc_mismatch<true>(mlo, a[q].x, b[q].x);
            mhi = acc_mismatch<true>(mhi, a[q].y, b[q].y);
        }
    }
    mlo |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mlo, 0xB1, 0xF, 0xF, true);   // quad_perm [1, 0, 3, 2]
    mhi |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mhi, 0xB1, 0xF, 0xF, true);
    return ((lane & 1u) == 0u && h < halves) ? (uint32_t)__builtin_popcount(mlo) + (uint32_t)__builtin_popcount(mhi) : 0u;
}

__device__ __forceinline__ uint32_t eb_wave_sum(uint32_t part)
{
    part = SKL_DPP_ADD(part, 0x111);   // row_shr:1
    part = SKL_DPP_ADD(part, 0x112);   // row_shr:2
    part = SKL_DPP_ADD(part, 0x114);   // row_shr:4
    part = SKL_DPP_ADD(part, 0x118);   // row_shr:8
    return (uint32_t)__builtin_amdgcn_readlane((int)part, 15) + (uint32_t)__builtin_amdgcn_readlane((int)part, 31) +
           (uint32_t)__builtin_amdgcn_readlane((int)part, 47) + (uint32_t)__builtin_amdgcn_readlane((int)part, 63);
}

// The row sample's slice of ONE k-mer length, kept in registers across the alive columns of that row (TRIPS x 7 x 8 bytes
// per lane: sketches of up to 32 x TRIPS chunks; TRIPS = 0: any size, nothing kept)
template <int TRIPS>
struct EbRow {
    uint2 a[TRIPS > 0 ? TRIPS : 1][7];
    uint32_t i = EB_NONE, t = EB_NONE;
};

template <int TRIPS>
__device__ __forceinline__ uint32_t eb_same_bins(const uint64_t *rows_ref, const uint64_t *cols_ref, uint32_t nk_total, uint32_t ss64,
                                                 EbRow<TRIPS> &row, uint32_t i, uint32_t j, uint32_t t, bool keep, uint32_t lane)
{
    const uint32_t halves = ss64 * 2u;
    const uint2 *pi = reinterpret_cast<const uint2 *>(rows_ref + (((uint64_t)i * nk_total + t) * ss64) * BBITS);
    const uint2 *pj = reinterpret_cast<const uint2 *>(cols_ref + (((uint64_t)j * nk_total + t) * ss64) * BBITS);
    uint32_t part = 0;
    if constexpr (TRIPS > 0) {
        if (keep) {
            if (row.i != i || row.t != t) {   // (wave-uniform)
#pragma unroll
                for (int tr = 0; tr < TRIPS; ++tr) {
#pragma unroll
                    for (int q = 0; q < 7; ++q) row.a[tr][q] = (uint32_t)tr * 64u + lane < halves ? pi[((size_t)tr * 64u + lane) * 7 + q] : make_uint2(0u, 0u);
                }
                row.i = i;
                row.t = t;
            }
#pragma unroll
            for (int tr = 0; tr < TRIPS; ++tr) part += eb_trip(row.a[tr], pj, (uint32_t)tr * 64u + lane, halves, lane);
            return ss64 * 64u - eb_wave_sum(part);
        }
    }
    for (uint32_t h0 = 0; h0 < halves; h0 += 64u) {
        const uint32_t h = h0 + lane;
        uint2 a[7];
#pragma unroll
        for (int q = 0; q < 7; ++q) a[q] = h < halves ? pi[(size_t)h * 7 + q] : make_uint2(0u, 0u);
        part += eb_trip(a, pj, h, halves, lane);
    }
    return ss64 * 64u - eb_wave_sum(part);
}

// Entry e of wave w_src's list, completed by the calling wave: the counts of the lengths from its ke on, up to and including
// the one that ends the reference's loop (jaccard.rs:89-91).
struct EbCompleteArgs {                 // what the completion needs of EpilogueArgs (passed by value: registers)
    const uint64_t *rows_ref, *cols_ref;
    const double *compA, *compB, *ytab;
    double cutoff, tolerance;
    uint32_t nk_total, ss64, min_alive;
    int32_t log_variant;
};

template <int TRIPS, bool COMP>
__device__ __forceinline__ void eb_complete(const EbCompleteArgs &g, EbLists &L, uint32_t w_src, uint32_t e, EbRow<TRIPS> &row, uint32_t lane)
{
    const uint32_t i = (uint32_t)__builtin_amdgcn_readfirstlane((int)L.i[w_src][e]);
    const uint32_t j = (uint32_t)__builtin_amdgcn_readfirstlane((int)L.j[w_src][e]);
    const uint32_t ke = (uint32_t)__builtin_amdgcn_readfirstlane((int)L.ke[w_src][e]);
    double c1 = 0.0, c2 = 0.0;
    if constexpr (COMP) {
        c1 = g.compA[i];
        c2 = g.compB[j];
    }
    for (uint32_t t = ke; t < g.nk_total; ++t) {
        // (the row's slice is kept for the first of the remaining lengths, where four completions of five end)
        const uint32_t same = eb_same_bins<TRIPS>(g.rows_ref, g.cols_ref, g.nk_total, g.ss64, row, i, j, t, t == ke, lane);
        if (lane == 0u) L.same[w_src][e][t] = same;
        bool stop;
        if (!COMP && g.min_alive != EB_NONE) stop = same < g.min_alive;
        else if constexpr (COMP) stop = glibc_log(jaccard_from_samebits_dev(same, g.ss64, true, c1, c2, g.cutoff), g.log_variant) < g.tolerance;
        else stop = g.ytab[same <= g.ss64 * 64u ? same : g.ss64 * 64u] < g.tolerance;
        if (stop) break;   // (wave-uniform)
    }
}

// Entries [e_lo, e_hi) of the workgroup's concatenated lists (c0, c1, c2: the lengths of the first three; `shared` = false: of
// wave w_own's list alone), completed by the calling wave.  Not inlined: its registers -- the row slice, a column slice in
// flight -- are then allotted apart from the kernel's other phases.
template <int TRIPS, bool COMP>
__device__ __noinline__ void eb_complete_range(EbCompleteArgs g, EbLists *L, uint32_t e_lo, uint32_t e_hi, uint32_t c0, uint32_t c1, uint32_t c2,
                                               uint32_t w_own, bool shared)
{
    const uint32_t lane = threadIdx.x & 63u;
    EbRow<TRIPS> row;
    for (uint32_t e = e_lo; e < e_hi; ++e) {
        uint32_t w_src = w_own, idx = e;
        if (shared && idx >= c0) {
            idx -= c0;
            w_src = 1;
            if (idx >= c1) {
                idx -= c1;
                w_src = 2;
                if (idx >= c2) {
                    idx -= c2;
                    w_src = 3;
                }
            }
        }
        eb_complete<TRIPS, COMP>(g, *L, w_src, idx, row, lane);
    }
}

// a wave's LDS writes (list entries, completed counts) before its other lanes read them
__device__ __forceinline__ void eb_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

}  // namespace

// One wave = 64 x span consecutive pairs of the launch's flat order (consecutive columns of a row, then the next row).
template <int TRIPS, bool COMP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void coreacc_epilogue_rows_kernel(const EpilogueArgs g)
{
    __shared__ EbLists L;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint64_t wave_p0 = ((uint64_t)blockIdx.x * 4u + wave) * 64ull * g.span;
    const bool early = g.nk_total > g.nk || g.block_ke != nullptr;   // (pairs may be left in the running)
    const bool need_ij = early || COMP;
    const uint32_t maxnbits = g.ss64 * 64u;
    uint32_t i0 = 0, pos0 = 0;
    if (need_ij && wave_p0 < g.n_pairs) {
        const uint64_t flat = wave_p0 + g.out_base;
        if (g.self_mode) {
            eb_locate_self(flat, g.n_total, i0, pos0);
        } else {
            i0 = (uint32_t)(flat / g.nB_cols);
            pos0 = (uint32_t)(flat % g.nB_cols);
        }
    }
    // the (i, j) of the pair `slot` positions behind the wave's first
    auto pair_at = [&](uint32_t slot, uint32_t &i, uint32_t &j) {
        if (g.self_mode) {
            uint32_t pos = pos0 + slot, len = g.n_total - 1u - i0;
            i = i0;
            while (pos >= len) {
                pos -= len;
                ++i;
                --len;
            }
            j = i + 1u + pos;
        } else {
            const uint64_t at = (uint64_t)pos0 + slot;
            i = i0 + (uint32_t)(at / g.nB_cols);
            j = (uint32_t)(at % g.nB_cols);
        }
    };
    uint32_t parked = 0;                // pairs this wave has parked in all
    uint32_t it = 0;
    bool last;
    // A wave walks its groups of 64 pairs until they end (`last`) or its list may not hold another group's alive pairs; then
    // the parked pairs are completed and finished -- after the last group by the whole workgroup together, before (a list
    // that filled up: closely related samples the block's sample did not show) by the wave alone.
    do {
        uint32_t count = 0;             // entries of this wave's list
        for (; it < g.span && count + 64u <= EB_CAP; ++it) {
            const uint64_t p0_it = wave_p0 + (uint64_t)it * 64u;
            if (p0_it >= g.n_pairs) {   // (wave-uniform)
                it = g.span;
                break;
            }
            const uint32_t slot = it * 64u + lane;
            const uint64_t p = p0_it + lane;
            const bool in_range = p < g.n_pairs;
            uint32_t i = 0, j = 0, ke = g.nk;
            double c1 = 0.0, c2 = 0.0;
            if (in_range && (COMP || g.block_ke != nullptr)) {
                pair_at(slot, i, j);
                if constexpr (COMP) {
                    c1 = g.compA[i];
                    c2 = g.compB[j];
                }
                if (g.block_ke != nullptr) ke = g.block_ke[(size_t)(i >> g.blk_shift_r) * g.blk_cols + (j >> g.blk_shift_c)];
            }
            EbSums s;
            bool stopped = !in_range;
            // The reference's loop is a chain of 2 nk dependent loads when taken literally (count, then table entry): the
            // counts of up to KB lengths and their table entries are loaded up front, independent of each other; the sums
            // then stop where the reference's do.
            constexpr uint32_t KB = 8;
            uint32_t first_same[KB];
            for (uint32_t t0 = 0; t0 < g.nk; t0 += KB) {
                uint32_t same[KB];
#pragma unroll
                for (uint32_t u = 0; u < KB; ++u) {
                    const uint32_t t = t0 + u;
                    same[u] = 0u;
                    if (t < g.nk && in_range) {
                        if (t < ke) {
                            same[u] = eb_count_at(g, p * g.pair_stride + (uint64_t)t * g.k_stride);
                            for (uint32_t sl = 1; sl < g.n_slices; ++sl) same[u] += eb_count_at(g, p * g.pair_stride + ((uint64_t)sl * g.nk + t) * g.k_stride);
                        }
                        // plane 1 goes back to zero for the next tail-sliced launch, whether or not k index t is used
                        if (g.rezero_plane1) g.counts[p * g.pair_stride + ((uint64_t)g.nk + t) * g.k_stride] = 0u;
                    }
                    if (t0 == 0u) first_same[u] = same[u];
                }
                if constexpr (!COMP) {
                    double yt[KB];
#pragma unroll
                    for (uint32_t u = 0; u < KB; ++u) yt[u] = g.ytab[same[u] <= maxnbits ? same[u] : maxnbits];
#pragma unroll
                    for (uint32_t u = 0; u < KB; ++u) {
                        const uint32_t t = t0 + u;
                        if (t >= ke || t >= g.nk || stopped) continue;
                        if (yt[u] < g.tolerance) {   // jaccard.rs:89-91: break
                            stopped = true;
                            continue;
                        }
                        s.add(g.kf[t], yt[u]);
                    }
                } else {
#pragma clang loop unroll(disable)
                    for (uint32_t u = 0; u < KB; ++u) {
                        const uint32_t t = t0 + u;
                        if (t >= ke || t >= g.nk || stopped) break;
                        uint32_t same_u = same[0];
#pragma unroll
                        for (uint32_t x = 1; x < KB; ++x) same_u = u == x ? same[x] : same_u;
                        const double y = eb_lnj<COMP>(g, same_u, c1, c2);
                        if (y < g.tolerance) {
                            stopped = true;
                            break;
                        }
                        s.add(g.kf[t], y);
                    }
                }
            }
            const bool alive = early && in_range && !stopped && ke < g.nk_total;
            if (in_range && !alive) ((float2 *)g.out)[p] = eb_regress(s.xsum, s.ysum, s.xysum, s.xsquaresum, s.ysquaresum, s.n);
            if (!early) continue;
            const uint64_t mask = __ballot(alive);
            if (mask == 0ull) continue;     // (wave-uniform)
            if (alive) {
                if (!(COMP || g.block_ke != nullptr)) pair_at(slot, i, j);
                const uint32_t e = count + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
                L.i[wave][e] = i;
                L.j[wave][e] = j;
                L.slot[wave][e] = slot;
                L.ke[wave][e] = ke;
#pragma unroll
                for (uint32_t t = 0; t < EB_MAXK; ++t) L.same[wave][e][t] = t < ke ? first_same[t] : EB_NONE;
            }
            count += (uint32_t)__popcll(mask);
        }
        if (!early) return;
        last = it >= g.span;
        parked += count;
        // completion: after the last group the four waves take equal shares of the workgroup's concatenated lists, each
        // its entries in list order (a wave's list is in pair order: runs of columns of one row)
        uint32_t c0 = 0, c1_ = 0, c2_ = 0, e_lo = 0, e_hi = count, w_own = wave;
        if (last) {
            if (lane == 0u) L.count[wave] = count;
            __syncthreads();
            c0 = L.count[0];
            c1_ = L.count[1];
            c2_ = L.count[2];
            const uint32_t total = c0 + c1_ + c2_ + L.count[3];
            e_lo = (uint32_t)(((uint64_t)total * wave) >> 2);
            e_hi = (uint32_t)(((uint64_t)total * (wave + 1u)) >> 2);
            w_own = 0u;
        } else {
            eb_wave_sync();
        }
        if (e_hi > e_lo) {
            EbCompleteArgs ca;
            ca.rows_ref = g.rows_ref;
            ca.cols_ref = g.cols_ref;
            ca.compA = g.compA;
            ca.compB = g.compB;
            ca.ytab = g.ytab;
            ca.cutoff = g.cutoff;
            ca.tolerance = g.tolerance;
            ca.nk_total = g.nk_total;
            ca.ss64 = g.ss64;
            ca.min_alive = g.min_alive;
            ca.log_variant = g.log_variant;
            eb_complete_range<TRIPS, COMP>(ca, &L, e_lo, e_hi, c0, c1_, c2_, w_own, last);
        }
        if (last) __syncthreads();
        else eb_wave_sync();
        // the parked pairs, one per lane: the reference's loop over the counts of every length looked at
        for (uint32_t q0 = 0; q0 < count; q0 += 64u) {
            const uint32_t q = q0 + lane;
            if (q < count) {
                const uint32_t i = L.i[wave][q], j = L.j[wave][q];
                double c1 = 0.0, c2 = 0.0;
                if constexpr (COMP) {
                    c1 = g.compA[i];
                    c2 = g.compB[j];
                }
                EbSums s;
#pragma clang loop unroll(disable)
                for (uint32_t t = 0; t < g.nk_total; ++t) {
                    const uint32_t same = L.same[wave][q][t];
                    if (same == EB_NONE) break;
                    const double y = eb_lnj<COMP>(g, same, c1, c2);
                    if (y < g.tolerance) break;   // jaccard.rs:89-91
                    s.add(g.kf[t], y);
                }
                ((float2 *)g.out)[wave_p0 + L.slot[wave][q]] = eb_regress(s.xsum, s.ysum, s.xysum, s.xsquaresum, s.ysquaresum, s.n);
            }
        }
        if (!last) eb_wave_sync();      // (the list is rewritten by the next groups)
    } while (!last);
    if (g.alive_count != nullptr && parked != 0u && lane == 0u) atomicAdd(&g.alive_count[blockIdx.x & 1023u], parked);   // (1 024 slots: adds to ONE address queue up)
}

// One wave per (block, sample): see EbSampleArgs (kernels.h).
__global__ __launch_bounds__(256) void early_break_sample_kernel(const EbSampleArgs g)
{
    const uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
    const uint32_t blk = w / g.samples, s = w - blk * g.samples;
    if (blk >= g.blk_rows * g.blk_cols) return;
    const uint32_t br = blk / g.blk_cols, bc = blk - br * g.blk_cols;
    if (g.self_mode && bc < br) return;                       // (below the diagonal: no pair of the launch)
    const uint32_t r_lo = br << g.blk_shift_r, r_hi = min(g.n_rows, (br + 1u) << g.blk_shift_r);
    const uint32_t c_lo = bc << g.blk_shift_c, c_hi = min(g.n_cols, (bc + 1u) << g.blk_shift_c);
    if (r_hi <= r_lo || c_hi <= c_lo) return;
    uint64_t h = ((uint64_t)w + 1u) * 0x9E3779B97F4A7C15ull;
    h ^= h >> 29;
    h *= 0xBF58476D1CE4E5B9ull;
    h ^= h >> 32;
    uint32_t i = r_lo + (uint32_t)(h % (r_hi - r_lo));
    uint32_t j = c_lo + (uint32_t)((h >> 20) % (c_hi - c_lo));
    (void)s;
    if (g.self_mode) {                                         // (only a diagonal block can hold i >= j: both ranges are the same)
        if (i == j) {
            if (j + 1u < c_hi) ++j;
            else if (i > r_lo) --i;
            else return;                                       // a block of one sample
        }
        if (i > j) {
            const uint32_t x = i;
            i = j;
            j = x;
        }
    }
    double c1 = 0.0, c2 = 0.0;
    if (g.has_comp) {
        c1 = g.compA[i];
        c2 = g.compB[j];
    }
    const uint32_t maxnbits = g.ss64 * 64u;
    EbRow<0> none;
    uint32_t lead = 0u;
    for (uint32_t t = 0; t < g.nk && t < 8u; ++t) {
        const uint32_t same = eb_same_bins<0>(g.rows_ref, g.cols_ref, g.nk, g.ss64, none, i, j, t, false, lane);
        bool stop;
        if (g.has_comp) stop = glibc_log(jaccard_from_samebits_dev(same, g.ss64, true, c1, c2, g.cutoff), g.log_variant) < g.tolerance;
        else if (g.min_alive != EB_NONE) stop = same < g.min_alive;
        else stop = g.ytab[same <= maxnbits ? same : maxnbits] < g.tolerance;
        if (stop) break;                                       // jaccard.rs:89-91
        ++lead;
    }
    if (lane == 0u) atomicAdd(&g.hist[(size_t)blk * 9u + lead], 1u);
}

#undef SKL_DPP_ADD

hipError_t launch_early_break_sample(const EbSampleArgs &args, hipStream_t stream)
{
    const uint64_t waves = (uint64_t)args.blk_rows * args.blk_cols * args.samples;
    if (waves == 0 || args.n_rows == 0 || args.n_cols == 0) return hipSuccess;
    const uint64_t blocks = (waves * 64u + 255u) / 256u;
    if (blocks >= (1ull << 31)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(early_break_sample_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, args);
    return hipGetLastError();
}

hipError_t launch_coreacc_epilogue_rows(const EpilogueArgs &args_in, hipStream_t stream)
{
    EpilogueArgs args = args_in;
    if (args.n_pairs == 0) return hipSuccess;
    if (args.span == 0u) args.span = 1u;
    if (args.nk_total != 0u && args.nk_total > EB_MAXK && (args.nk_total > args.nk || args.block_ke != nullptr)) return hipErrorInvalidValue;
    const uint64_t per_wg = 256ull * args.span;
    const uint64_t blocks = (args.n_pairs + per_wg - 1) / per_wg;
    if (blocks >= (1ull << 31)) return hipErrorInvalidValue;
    const dim3 gr((unsigned)blocks), bl(256);
    const bool early = args.nk_total > args.nk || args.block_ke != nullptr;
    const uint32_t trips = early ? (args.ss64 + 31u) / 32u : 0u;   // trips of 32 chunks the row slice is kept for (0: not kept)
#define SKL_EPI_LAUNCH(T)                                                                                               \
    do {                                                                                                                \
        if (args.has_comp) hipLaunchKernelGGL((coreacc_epilogue_rows_kernel<T, true>), gr, bl, 0, stream, args);        \
        else hipLaunchKernelGGL((coreacc_epilogue_rows_kernel<T, false>), gr, bl, 0, stream, args);                     \
    } while (0)
    switch (trips) {
        case 1: SKL_EPI_LAUNCH(1); break;
        case 2: SKL_EPI_LAUNCH(2); break;
        case 3: SKL_EPI_LAUNCH(3); break;
        case 4: SKL_EPI_LAUNCH(4); break;
        default: SKL_EPI_LAUNCH(0); break;
    }
#undef SKL_EPI_LAUNCH
    return hipGetLastError();
}

}  // namespace skl
