// epilogue.hip -- the core/accessory epilogue of the counts + epilogue form, with the EARLY BREAK of the reference's loop
// over the k-mer lengths (gfx950).
//
// Reference: core_acc_dist, src/distances/jaccard.rs:61-101.  The loop over the k-mer lengths leaves at the first one whose
// ln J lies below the tolerance (:89-91) -- J = 0 for a pair that shares no more bins than chance (expected_samebits,
// :26-31) -- and a fit over fewer than three lengths is (1, 1) (:117).  The pair kernel therefore counts only the first
// `ke` lengths of a block of pairs (pair_kslice.hip, k-sliced MODE_COUNTS), and
// coreacc_epilogue_kernel_r6 (one thread per pair, a pure stream: 4-16 bytes of counts in, 8 bytes out) runs the reference's
// loop over those counts: a pair that leaves inside them is finished at once; a pair STILL IN THE RUNNING is completed on the
// spot by its wave -- all 64 lanes count the bins the pair shares at the next length, each sample's slice read as ONE
// contiguous run (lane l the l-th half chunk of 7 planes, 56 bytes; the two halves of a chunk meet by DPP: the form
// pair_cand.hip measured at 0.88 of the HBM peak), until the reference's break.
// Round 5 read the two slices a lane per chunk, 64 lanes 112 bytes apart, and asked the ln J table after every length; this
// form tests the count itself (below) and is 5-13 % faster on whole calls (profiles/r06_epilogue_forms.md).  Two forms that
// set the completions apart were built and measured SLOWER than completing on the spot, and are gone again: per-workgroup
// LDS lists completed row by row inside the epilogue (the completions halve, the stream loses it again: 128 registers and 24
// KB of LDS where the plain epilogue runs at 5 waves per SIMD), and a work list in global memory walked by a second kernel
// with the row's slice held in registers (n = 16 000: 22.4 against 18.7 ms -- the completions are bound by the column
// slices' bytes either way, and on the spot they overlap with the other waves' streaming for free).
// Late in round 6 the counters showed what binds the kernel: the instructions it issues (profiles/r06_epilogue_lean.md).
// Launches with one ke for every pair -- all but block-by-block plans and completeness vectors with a value outside (0, 1] --
// now go to coreacc_epilogue_lean_kernel (below); coreacc_epilogue_kernel_r6 keeps the general case.
//
// The break test is the reference's: y < tolerance with y = ln J.  Without a completeness correction y is a function of
// the bin-match count alone and non-decreasing in it, so the test is `count < min_alive` (the host finds min_alive in the
// table it uploads and checks the monotonicity; EB_NONE: ask the table).  With a correction (jaccard.rs:36-41) J is scaled
// per pair -- 0 stays 0 -- and y is evaluated with the restated libm logarithm, per pair, as everywhere else.
#include "device_common.hpp"

#include <algorithm>

namespace skl {

namespace {

constexpr uint32_t EB_MAXK = 8;        // k-mer lengths of an early-break launch (the driver refuses more)
constexpr uint32_t EB_NONE = 0xFFFFFFFFu;

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

// Bins the pair (row sample i, column sample j) shares at k-mer length index t, counted by the whole wave.
__device__ __forceinline__ uint32_t eb_same_bins(const uint64_t *rows_ref, const uint64_t *cols_ref, uint32_t nk_total, uint32_t ss64, uint32_t i, uint32_t j,
                                                 uint32_t t, uint32_t lane)
{
    const uint32_t halves = ss64 * 2u;
    const uint2 *pi = reinterpret_cast<const uint2 *>(rows_ref + (((uint64_t)i * nk_total + t) * ss64) * BBITS);
    const uint2 *pj = reinterpret_cast<const uint2 *>(cols_ref + (((uint64_t)j * nk_total + t) * ss64) * BBITS);
    uint32_t part = 0;
    // (every lane loads: those past the slices' end re-read the last half chunk and count nothing -- a load under `h < halves`
    // is a branch around it, 8 of them per trip)
    for (uint32_t h0 = 0; h0 < halves; h0 += 64u) {
        const uint32_t h_raw = h0 + lane, h = min(h_raw, halves - 1u);
        uint2 a[7], b[7];
#pragma unroll
        for (int q = 0; q < 7; ++q) a[q] = pi[(size_t)h * 7 + q];
#pragma unroll
        for (int q = 0; q < 7; ++q) b[q] = pj[(size_t)h * 7 + q];
        uint32_t mlo = 0, mhi = 0;
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            mlo = acc_mismatch<true>(mlo, a[q].x, b[q].x);
            mhi = acc_mismatch<true>(mhi, a[q].y, b[q].y);
        }
        mlo |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mlo, 0xB1, 0xF, 0xF, true);   // quad_perm [1, 0, 3, 2]
        mhi |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mhi, 0xB1, 0xF, 0xF, true);
        part += ((lane & 1u) == 0u && h_raw < halves) ? (uint32_t)__builtin_popcount(mlo) + (uint32_t)__builtin_popcount(mhi) : 0u;
    }
    return ss64 * 64u - eb_wave_sum(part);
}

// the (i, j) of flat pair index `flat` of the launch's pair space
__device__ __forceinline__ void eb_pair_of(const EpilogueArgs &g, uint64_t flat, uint32_t &i, uint32_t &j)
{
    if (g.self_mode) {
        uint32_t pos;
        eb_locate_self(flat, g.n_total, i, pos);
        j = i + 1u + pos;
    } else {
        // (a 64-bit division is ~100 instructions for every wave that holds a pair still in the running: the quotient from the f64
        // product -- flat < 2^53 is exact in a double -- put right by one step either way)
        uint64_t q = (uint64_t)((double)flat * (1.0 / (double)g.nB_cols));
        int64_t r = (int64_t)(flat - q * g.nB_cols);
        if (r < 0) {
            --q;
            r += g.nB_cols;
        } else if (r >= (int64_t)g.nB_cols) {
            ++q;
            r -= g.nB_cols;
        }
        i = (uint32_t)q;
        j = (uint32_t)r;
    }
}

// does the reference's loop leave at a length with this bin-match count?  (jaccard.rs:88-91; wave-uniform where `same` is)
template <bool COMP>
__device__ __forceinline__ bool eb_stops(const EpilogueArgs &g, uint32_t same, double c1, double c2)
{
    if (!COMP && g.min_alive != EB_NONE) return same < g.min_alive;
    return eb_lnj<COMP>(g, same, c1, c2) < g.tolerance;
}

// THE FIRST LENGTH NOT COUNTED, for all the pairs of a wave that are still in the running (bit l of `mask`: lane l's pair), with
// the column slices requested ONE TRIP AHEAD.  Completing the pairs one after the other leaves a wave with one trip of 3.5 KB
// in flight, then a reduction, then the next request: the completions are bound by that chain, not by the L2 the blocked
// order serves them from (9 TB/s where MI355X_MICROARCH.md's gather reads 17-19).  Here trip n + 1 (the same pair's next
// run, or the next pair's first) is requested before trip n is counted, two register sets in turns; the row's slice
// comes from the workgroup's LDS copy (row_off: 0 / 1, which of the staged rows).  Every request is unconditional (lanes
// past the slice's end re-read its last half chunk and count nothing), so that the wait before a trip is counted stands for
// the OLDER request only.  Returns, in lane l of a pair, the bins the pair shares at length index g.nk.
struct EbTripIt {
    uint64_t mask;
    uint32_t l, trip;
    __device__ __forceinline__ explicit EbTripIt(uint64_t m) : mask(m), l((uint32_t)__builtin_ctzll(m)), trip(0u) {}
    __device__ __forceinline__ bool valid() const { return mask != 0ull; }
    __device__ __forceinline__ void advance(uint32_t trips)
    {
        if (++trip == trips) {
            trip = 0u;
            mask &= mask - 1ull;
            l = mask != 0ull ? (uint32_t)__builtin_ctzll(mask) : 0u;
        }
    }
};

__device__ __forceinline__ void eb_request_cols(uint2 (&b)[7], const uint2 *cols_t, size_t sample_stride, uint32_t j, const EbTripIt &it, uint32_t halves, uint32_t lane)
{
    const uint32_t j_l = (uint32_t)__builtin_amdgcn_readlane((int)j, (int)it.l);
    const uint32_t h = min(it.trip * 64u + lane, halves - 1u);
    const uint2 *pj = cols_t + (size_t)j_l * sample_stride + (size_t)h * 7;
#pragma unroll
    for (int q = 0; q < 7; ++q) b[q] = pj[q];
}

__device__ __forceinline__ void eb_count_trip(const uint2 (&b)[7], const uint2 *lds_rows, uint32_t row_off, const EbTripIt &it, uint32_t ss64, uint32_t lane,
                                              uint32_t &part, uint32_t &result)
{
    const uint32_t halves = ss64 * 2u, trips = (halves + 63u) >> 6;
    const uint32_t r_l = (uint32_t)__builtin_amdgcn_readlane((int)row_off, (int)it.l);
    const uint32_t h_raw = it.trip * 64u + lane, h = min(h_raw, halves - 1u);
    const uint2 *pa = lds_rows + (size_t)r_l * ss64 * 14u + (size_t)h * 7;
    uint32_t mlo = 0, mhi = 0;
#pragma unroll
    for (int q = 0; q < 7; ++q) {
        const uint2 a = pa[q];
        mlo = acc_mismatch<true>(mlo, a.x, b[q].x);
        mhi = acc_mismatch<true>(mhi, a.y, b[q].y);
    }
    mlo |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mlo, 0xB1, 0xF, 0xF, true);   // quad_perm [1, 0, 3, 2]
    mhi |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mhi, 0xB1, 0xF, 0xF, true);
    part += ((lane & 1u) == 0u && h_raw < halves) ? (uint32_t)__builtin_popcount(mlo) + (uint32_t)__builtin_popcount(mhi) : 0u;
    if (it.trip + 1u == trips) {   // (wave-uniform) the pair's last trip
        const uint32_t same = ss64 * 64u - eb_wave_sum(part);
        if (lane == it.l) result = same;
        part = 0u;
    }
}

__device__ __forceinline__ uint32_t eb_first_length_ahead(const EpilogueArgs &g, uint64_t mask, uint32_t j, uint32_t row_off, const uint2 *lds_rows, uint32_t lane)
{
    const uint32_t halves = g.ss64 * 2u, trips = (halves + 63u) >> 6;
    const size_t sample_stride = (size_t)g.nk_total * g.ss64 * BBITS;   // uint2 between the same slice of consecutive samples
    const uint2 *cols_t = reinterpret_cast<const uint2 *>(g.cols_ref) + (size_t)g.nk * g.ss64 * BBITS;
    uint2 b0[7], b1[7];
    uint32_t part = 0u, result = 0u;
    EbTripIt rq(mask), ct(mask);
    uint32_t left = (uint32_t)__popcll(mask) * trips;   // trips not yet counted; b0 holds the first of them
    eb_request_cols(b0, cols_t, sample_stride, j, rq, halves, lane);
    rq.advance(trips);
    // (one exit, nothing conditional around a request: a wait in this loop stands for the older of the two requests in flight)
    while (left > 2u) {
        eb_request_cols(b1, cols_t, sample_stride, j, rq, halves, lane);
        rq.advance(trips);
        eb_count_trip(b0, lds_rows, row_off, ct, g.ss64, lane, part, result);
        ct.advance(trips);
        eb_request_cols(b0, cols_t, sample_stride, j, rq, halves, lane);
        rq.advance(trips);
        eb_count_trip(b1, lds_rows, row_off, ct, g.ss64, lane, part, result);
        ct.advance(trips);
        left -= 2u;
    }
    if (left == 2u) eb_request_cols(b1, cols_t, sample_stride, j, rq, halves, lane);
    eb_count_trip(b0, lds_rows, row_off, ct, g.ss64, lane, part, result);
    if (left == 2u) {
        ct.advance(trips);
        eb_count_trip(b1, lds_rows, row_off, ct, g.ss64, lane, part, result);
    }
    return result;
}

}  // namespace

// One thread per pair of the launch.
template <bool COMP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(COMP ? 4 : 5, COMP ? 4 : 5))) void coreacc_epilogue_kernel_r6(const EpilogueArgs g)
{
    const bool early = g.nk_total > g.nk || g.block_ke != nullptr;   // (pairs may be left in the running)
    const uint32_t maxnbits = g.ss64 * 64u;
    uint32_t i = 0, j = 0, ke = g.nk;
    bool in_range, have_ij = false;
    uint64_t p;
    if (g.blocked) {
        // BLOCKED ORDER (early-break launches whose column slices do not fit the Infinity Cache): the pair space is walked in
        // blocks of 1 024 rows x 256 columns (cfg 3: 256 / 512 / 1 024 / 2 048 rows: 659 / 649 / 642 / 639 ms), a workgroup = one
        // row's 256 columns of a block, the workgroups of a block consecutive ON ONE XCD (blockIdx mod XCDs is the XCD, MI355X_MICROARCH.md).  A completion reads its column sample's
        // slice; in the flat order (a row after the other, all its columns) a slice's next reader comes a whole row later and
        // every completion is a 7 KB gather from HBM.  Here the block's 256 column slices (1.8 MB at 4 096 bins) stay in that
        // XCD's 4 MB L2 while the block's rows pass.
        const uint32_t wg = blockIdx.x + g.wg_base;              // (a launch carries at most 2^31 work-items: wg_base, a multiple of the XCDs)
        const uint32_t xcd = wg & ((1u << g.xcd_shift) - 1u), slot = wg >> g.xcd_shift;
        const uint32_t rs = g.blk_row_shift, lb = slot >> rs;      // rows per block = 1 << rs; lb: this XCD's lb-th block
        const uint32_t blk = (lb << g.xcd_shift) + xcd;          // blocks dealt to the XCDs in turns, column block fastest
        if (blk >= g.blk_rb * g.blk_cb) return;                    // (workgroup-uniform)
        const uint32_t rb = blk / g.blk_cb, cb = g.blk_cb0 + (blk - rb * g.blk_cb);   // (self mode: the column blocks left of the launch's first row hold no pair)
        i = g.row_begin + (rb << rs) + (slot & ((1u << rs) - 1u));
        j = cb * 256u + threadIdx.x;
        if (i >= g.row_end || (g.self_mode && cb * 256u + 255u <= i)) return;   // (workgroup-uniform: no pair of the launch in this row of the block)
        in_range = j < g.nB_cols && (!g.self_mode || j > i);
        have_ij = true;
        p = in_range ? (g.self_mode ? square_to_condensed_dev(i, j, g.n_total) : (uint64_t)i * g.nB_cols + j) - g.out_base : 0ull;   // (lanes without a pair shadow pair 0 and store nothing)
    } else {
        const uint64_t p_raw = ((uint64_t)blockIdx.x + g.wg_base) * blockDim.x + threadIdx.x;
        if (p_raw >= g.n_pairs && !early) return;   // (early break: every lane of a wave stays, the completion below is cooperative)
        in_range = p_raw < g.n_pairs;
        p = in_range ? p_raw : g.n_pairs - 1;       // (lanes past the end shadow the last pair and store nothing)
    }
    double c1 = 0.0, c2 = 0.0;
    if (COMP || g.block_ke != nullptr) {
        if (!have_ij) eb_pair_of(g, p + g.out_base, i, j);
        have_ij = true;
        if constexpr (COMP) {
            c1 = g.compA[i];
            c2 = g.compB[min(j, g.nB_cols - 1u)];
        }
        if (g.block_ke != nullptr) ke = g.block_ke[(size_t)(i >> g.blk_shift_r) * g.blk_cols + (min(j, g.nB_cols - 1u) >> g.blk_shift_c)];
    }
    const uint64_t cnt0 = p * g.pair_stride;
    EbSums s;
    // The reference's loop leaves at the first k-mer length whose ln J is below the tolerance (jaccard.rs:89-91), so a literal
    // loop is a chain of 2 nk dependent loads (count, then table entry).  The counts of up to KB k-mer lengths and their table
    // entries are therefore loaded up front, independent of each other; the sums then stop at the same length as before.
    constexpr uint32_t KB = 8;
    bool stopped = false;
    for (uint32_t t0 = 0; t0 < g.nk; t0 += KB) {
        uint32_t same[KB];
        double yt[KB];
#pragma unroll
        for (uint32_t u = 0; u < KB; ++u) {
            const uint32_t t = t0 + u;
            same[u] = 0u;
            if (t < g.nk) {
                if (t < ke) {
                    same[u] = eb_count_at(g, cnt0 + (uint64_t)t * g.k_stride);
                    for (uint32_t sl = 1; sl < g.n_slices; ++sl) same[u] += eb_count_at(g, cnt0 + ((uint64_t)sl * g.nk + t) * g.k_stride);
                    // (a pair still in the running is taken up again from plane 0: it gets the sum)
                    if (early && g.n_slices > 1u && in_range) g.counts[cnt0 + (uint64_t)t * g.k_stride] = same[u];
                }
                // plane 1 goes back to zero for the next tail-sliced launch, whether or not k index t is used
                if (g.rezero_plane1 && in_range) g.counts[cnt0 + ((uint64_t)g.nk + t) * g.k_stride] = 0u;   // (a lane that shadows the last pair must not clear what that pair's own lane has yet to read)
            }
        }
        if constexpr (!COMP) {
#pragma unroll
            for (uint32_t u = 0; u < KB; ++u) yt[u] = g.ytab[same[u] <= maxnbits ? same[u] : maxnbits];
        }
#pragma unroll
        for (uint32_t u = 0; u < KB; ++u) {
            const uint32_t t = t0 + u;
            if (t >= ke || t >= g.nk || stopped) continue;
            double y;
            if constexpr (COMP) y = eb_lnj<true>(g, same[u], c1, c2);
            else y = yt[u];
            if (y < g.tolerance) {   // jaccard.rs:89-91: break
                stopped = true;
                continue;
            }
            s.add(g.kf[t], y);
        }
    }
    if (!early) {
        ((float2 *)g.out)[p] = simple_linear_regression_dev(s.xsum, s.ysum, s.xysum, s.xsquaresum, s.ysquaresum, s.n);
        return;
    }
    bool alive = in_range && !stopped && ke < g.nk_total;
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t alive_mask = __ballot(alive);
    if (alive_mask != 0ull) {           // (wave-uniform)
        if (alive && !have_ij) eb_pair_of(g, p + g.out_base, i, j);
        if (g.alive_count != nullptr && lane == 0u) atomicAdd(&g.alive_count[blockIdx.x & 1023u], (uint32_t)__popcll(alive_mask));   // (1 024 slots: a million adds to ONE address queue up)
    }
    // ROWS IN LDS (g.lds_rows: launches with one ke for every pair, sketches whose two slices fit).  The 256 pairs of a workgroup
    // are consecutive in the launch's flat order -- columns of one row, then of the next -- so if any of them is still in the
    // running, the workgroup stages the slices (of the first length not counted) of its first pair's row and of the row behind
    // it ONCE, and a completion reads only its column sample's slice from memory: half the L2 traffic of the completions.
    extern __shared__ __attribute__((aligned(16))) uint2 eb_lds_rows[];   // [2][2 ss64][7]
    uint32_t i_wg = 0;
    bool staged = false;
    if (g.lds_rows) {                       // (workgroup-uniform; every thread is still here: see the top)
        if (__syncthreads_or(alive ? 1 : 0)) {
            uint32_t j_wg;
            if (g.blocked) i_wg = i;      // (blocked order: the workgroup's one row)
            else eb_pair_of(g, ((uint64_t)blockIdx.x + g.wg_base) * blockDim.x + g.out_base, i_wg, j_wg);
            const uint32_t per_row = g.ss64 * 14u;   // uint2 per slice
            const uint2 *src = reinterpret_cast<const uint2 *>(g.rows_ref + (((uint64_t)i_wg * g.nk_total + g.nk) * g.ss64) * BBITS);
            const size_t next_row = (size_t)g.nk_total * g.ss64 * BBITS;   // uint2 between the same slice of consecutive samples
            const uint32_t staged_rows = g.blocked ? 1u : 2u;   // (blocked order: the workgroup's pairs are one row's)
            for (uint32_t x = threadIdx.x; x < staged_rows * per_row; x += blockDim.x) {
                const uint32_t r = x >= per_row ? 1u : 0u;   // (the row slab ends in pad rows: row i_wg + 1 always exists)
                eb_lds_rows[x] = src[(size_t)r * next_row + (x - r * per_row)];
            }
            __syncthreads();
            staged = true;
        }
    }
    // the pairs of this wave still in the running, one after the other -- all 64 lanes count the bins the pair shares at the next
    // length, until the reference's break
    bool more = alive;            // this lane's pair is still in the running, from length index t_first on
    uint32_t t_first = ke;
    if constexpr (!COMP) {
        if (staged && g.ahead) {   // (workgroup-uniform) the first length not counted: every such pair of the wave, requests one trip ahead
            const uint32_t row_off = i - i_wg;
            const bool fast = alive && ke == g.nk && row_off < (g.blocked ? 1u : 2u);
            const uint64_t fast_mask = __ballot(fast);
            if (fast_mask != 0ull) {
                const uint32_t same0 = eb_first_length_ahead(g, fast_mask, j, row_off, eb_lds_rows, lane);
                if (fast) {
                    const double y0 = g.ytab[same0 <= maxnbits ? same0 : maxnbits];
                    if (g.min_alive != EB_NONE ? same0 < g.min_alive : y0 < g.tolerance) {
                        more = false;                        // jaccard.rs:89-91: break
                    } else {
                        s.add(g.kf[g.nk], y0);
                        t_first = g.nk + 1u;
                        more = t_first < g.nk_total;
                    }
                }
            }
        }
    }
    uint64_t todo = __ballot(more);
    while (todo != 0ull) {
        const int l = __builtin_ctzll(todo);
        todo &= todo - 1ull;
        const uint32_t i_l = (uint32_t)__builtin_amdgcn_readfirstlane(__shfl((int)i, l)), j_l = (uint32_t)__builtin_amdgcn_readfirstlane(__shfl((int)j, l));
        const uint32_t ke_l = (uint32_t)__builtin_amdgcn_readfirstlane(__shfl((int)t_first, l));
        double c1_l = 0.0, c2_l = 0.0;
        if constexpr (COMP) {
            c1_l = g.compA[i_l];
            c2_l = g.compB[j_l];
        }
        for (uint32_t t = ke_l; t < g.nk_total; ++t) {
            uint32_t same;
            if (staged && t == g.nk && i_l - i_wg < (g.blocked ? 1u : 2u)) {   // the row's slice from LDS, the column's as one contiguous run
                const uint32_t halves = g.ss64 * 2u;
                const uint2 *a_lds = eb_lds_rows + (size_t)(i_l - i_wg) * g.ss64 * 14u;
                const uint2 *pj = reinterpret_cast<const uint2 *>(g.cols_ref + (((uint64_t)j_l * g.nk_total + t) * g.ss64) * BBITS);
                uint32_t part = 0;
                for (uint32_t h0 = 0; h0 < halves; h0 += 64u) {
                    const uint32_t h = h0 + lane;
                    uint2 a[7];
#pragma unroll
                    for (int q = 0; q < 7; ++q) a[q] = h < halves ? a_lds[(size_t)h * 7 + q] : make_uint2(0u, 0u);
                    part += eb_trip(a, pj, h, halves, lane);
                }
                same = g.ss64 * 64u - eb_wave_sum(part);
            } else {
                same = eb_same_bins(g.rows_ref, g.cols_ref, g.nk_total, g.ss64, i_l, j_l, t, lane);
            }
            if (eb_stops<COMP>(g, same, c1_l, c2_l)) break;   // (wave-uniform)
            if ((int)lane == l) s.add(g.kf[t], eb_lnj<COMP>(g, same, c1, c2));
        }
    }
    if (in_range && (alive || stopped || ke >= g.nk_total)) {
        ((float2 *)g.out)[p] = simple_linear_regression_dev(s.xsum, s.ysum, s.xysum, s.xsquaresum, s.ysquaresum, s.n);
    }
}

// (core, acc) of a pair with n_pts >= 3 points, from the bin-match counts of its first n_pts lengths: ln J looked up (all at once),
// the reference's sums in the reference's order (jaccard.rs:92-97), the regression.  ONE copy per kernel (the regression alone
// is ~600 instructions, and the callers meet it for one pair in a hundred or a thousand): the kNN bands' epilogue inlined it
// four times and ran to 47 KB of code.
__device__ __attribute__((noinline)) float2 eb_fit_of_counts(const double *ytab, const double *kf, uint32_t maxnbits, uint32_t n_pts, uint32_t c0, uint32_t c1,
                                                             uint32_t c2, uint32_t c3, uint32_t c4, uint32_t c5, uint32_t c6, uint32_t c7)
{
    const uint32_t c[EB_MAXK] = {c0, c1, c2, c3, c4, c5, c6, c7};
    double y[EB_MAXK];
#pragma unroll
    for (uint32_t t = 0; t < EB_MAXK; ++t) y[t] = t < n_pts ? ytab[c[t] <= maxnbits ? c[t] : maxnbits] : 0.0;
    EbSums s;
#pragma unroll
    for (uint32_t t = 0; t < EB_MAXK; ++t) {
        if (t < n_pts) s.add(kf[t], y[t]);
    }
    return simple_linear_regression_dev(s.xsum, s.ysum, s.xysum, s.xsquaresum, s.ysquaresum, s.n);
}

// ... the same with a completeness correction (jaccard.rs:36-41): J is scaled per pair, so ln J is the restated libm logarithm of each
// point instead of a table entry
__device__ __attribute__((noinline)) float2 eb_fit_of_counts_comp(const double *kf, uint32_t ss64, double c1, double c2, double cutoff, int log_variant, uint32_t n_pts,
                                                                  uint32_t c0, uint32_t c1n, uint32_t c2n, uint32_t c3, uint32_t c4, uint32_t c5, uint32_t c6, uint32_t c7)
{
    const uint32_t c[EB_MAXK] = {c0, c1n, c2n, c3, c4, c5, c6, c7};
    EbSums s;
#pragma unroll 1
    for (uint32_t t = 0; t < EB_MAXK; ++t) {
        if (t < n_pts) s.add(kf[t], glibc_log(jaccard_from_samebits_dev(c[t], ss64, true, c1, c2, cutoff), log_variant));
    }
    return simple_linear_regression_dev(s.xsum, s.ysum, s.xysum, s.xsquaresum, s.ysquaresum, s.n);
}

// does a count pass the reference's test (jaccard.rs:88-91) under a completeness correction?  The correction divides J by
// c1 c2 / (c1 + c2 - c1 c2) <= 1 for completeness values in (0, 1] (the host checks the vectors: EpilogueArgs::comp_lean), so a
// count that passes uncorrected passes corrected, and a count at or below the chance level gives J = 0 either way; only the counts
// in between -- none at most sketch sizes -- ask the logarithm.
__device__ __attribute__((noinline)) bool eb_passes_comp_exact(uint32_t same, uint32_t ss64, double c1, double c2, double cutoff, int log_variant, double tolerance)
{
    return !(glibc_log(jaccard_from_samebits_dev(same, ss64, true, c1, c2, cutoff), log_variant) < tolerance);
}

__device__ __forceinline__ bool eb_passes_comp(const EpilogueArgs &g, uint32_t same, uint32_t expected, double c1, double c2)
{
    if (same >= g.min_alive) return true;
    if (same <= expected) return false;
    return eb_passes_comp_exact(same, g.ss64, c1, c2, g.cutoff, g.log_variant, g.tolerance);
}

// THE LEAN FORM of the kernel above, for the launches that matter (one ke for every pair, the break decided on the count
// itself -- COMP: under a completeness correction with every value in (0, 1], see eb_passes_comp -- NK = 2 ... 4 lengths counted).  The general kernel is bound by the instructions it issues,
// not by memory: 273 vector + 393 scalar instructions per wave at cfg 4's sketch size, 697 + 850 at cfg 3's
// (profiles/r06_epilogue_lean.md), most of them spent on pairs that end as (1, 1) -- table look-ups, f64 sums and a regression
// with three divisions and three square roots for a pair whose fit has fewer than three points.  Here a pair that leaves the
// reference's loop with fewer than three lengths (jaccard.rs:89-91, :117) is decided by NK integer compares and stored; the
// pairs still in the running are completed as above (first length: requests one trip ahead from the LDS rows; later lengths one
// after the other), their counts kept in registers; only a pair with three or more points looks its ln J up and runs the
// reference's sums and regression, in the reference's order.  SLICED: u32 counts in n_slices planes (tail-sliced launches,
// cfg 2), plane 1 re-zeroed; else u16 counts in one plane.
template <bool SLICED, int NK, bool COMP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5))) void coreacc_epilogue_lean_kernel(const EpilogueArgs g)
{
    constexpr int EXT = (int)EB_MAXK - NK;
    const uint32_t maxnbits = g.ss64 * 64u;
    const uint32_t expected = maxnbits >> BBITS;   // (COMP) bins two unrelated sketches share by chance: J = 0 up to here
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t i = 0, j = 0;
    bool in_range, have_ij = false;
    uint64_t p;
    if (g.blocked) {   // (the blocked order of the kernel above)
        const uint32_t wg = blockIdx.x + g.wg_base;
        const uint32_t xcd = wg & ((1u << g.xcd_shift) - 1u), slot = wg >> g.xcd_shift;
        const uint32_t rs = g.blk_row_shift, lb = slot >> rs;
        const uint32_t blk = (lb << g.xcd_shift) + xcd;
        if (blk >= g.blk_rb * g.blk_cb) return;
        const uint32_t rb = blk / g.blk_cb, cb = g.blk_cb0 + (blk - rb * g.blk_cb);
        i = g.row_begin + (rb << rs) + (slot & ((1u << rs) - 1u));
        j = cb * 256u + threadIdx.x;
        if (i >= g.row_end || (g.self_mode && cb * 256u + 255u <= i)) return;
        in_range = j < g.nB_cols && (!g.self_mode || j > i);
        have_ij = true;
        p = in_range ? (g.self_mode ? square_to_condensed_dev(i, j, g.n_total) : (uint64_t)i * g.nB_cols + j) - g.out_base : 0ull;
    } else {
        const uint64_t p_raw = ((uint64_t)blockIdx.x + g.wg_base) * blockDim.x + threadIdx.x;
        in_range = p_raw < g.n_pairs;
        p = in_range ? p_raw : g.n_pairs - 1;       // (lanes past the end shadow the last pair and store nothing)
    }
    // the counted lengths: how many pass before the first that does not (jaccard.rs:89-91 on the counts: count < min_alive <=> ln J < tolerance)
    uint32_t all[EB_MAXK];
#pragma unroll
    for (int t = 0; t < (int)EB_MAXK; ++t) all[t] = 0u;
#pragma unroll
    for (int t = 0; t < NK; ++t) {
        if constexpr (SLICED) {
            all[t] = g.counts[p + (uint64_t)t * g.k_stride];
            for (uint32_t sl = 1; sl < g.n_slices; ++sl) all[t] += g.counts[p + ((uint64_t)sl * NK + t) * g.k_stride];
            if (g.rezero_plane1 && in_range) g.counts[p + ((uint64_t)NK + t) * g.k_stride] = 0u;   // plane 1 back to zero for the next tail-sliced launch
        } else {
            all[t] = (uint32_t)reinterpret_cast<const uint16_t *>(g.counts)[p + (uint64_t)t * g.k_stride];
        }
    }
    uint32_t passed = 0;
    bool run = true;
#pragma unroll
    for (int t = 0; t < NK; ++t) {
        run = run && all[t] >= g.min_alive;
        passed += run ? 1u : 0u;
    }
    bool ij_lane = false;                   // (COMP) this lane computed its own (i, j) already
    if constexpr (COMP) {
        // a count between the chance level and min_alive is decided by the pair's own completeness values: redo such a lane's walk exactly
        bool unsure = false;
        {
            bool r = true;
#pragma unroll
            for (int t = 0; t < NK; ++t) {
                if (r && all[t] < g.min_alive) {
                    unsure = all[t] > expected;
                    r = false;
                }
            }
        }
        unsure = unsure && in_range;
        if (__ballot(unsure) != 0ull) {      // (rare; never where min_alive <= expected + 1)
            if (unsure) {
                if (!have_ij) eb_pair_of(g, p + g.out_base, i, j);
                ij_lane = true;
                const double c1 = g.compA[i], c2 = g.compB[j];
                passed = 0;
                run = true;
#pragma unroll
                for (int t = 0; t < NK; ++t) {
                    run = run && eb_passes_comp(g, all[t], expected, c1, c2);
                    passed += run ? 1u : 0u;
                }
            }
        }
    }
    const bool alive = in_range && run;     // (NK < nk_total: the host sends nothing else here)
    const uint64_t alive_mask = __ballot(alive);
    // (i, j) in the flat order: the WORKGROUP's first pair is located once per wave (eb_pair_of: an f64 square root and its
    // fix-ups in self mode, ~100 instructions), a lane's own pair follows from it by whole rows -- a workgroup's 256 pairs span two
    // rows, a few more at the matrix' end -- instead of a second square root per wave
    uint32_t i_wg = 0, j_wg = 0;
    bool have_wg = false;
    auto locate_wg = [&]() {
        if (have_wg) return;
        if (g.blocked) {
            i_wg = i;
        } else {
            eb_pair_of(g, ((uint64_t)blockIdx.x + g.wg_base) * blockDim.x + g.out_base, i_wg, j_wg);
        }
        have_wg = true;
    };
    if (alive_mask != 0ull) {           // (wave-uniform)
        if (!have_ij) {
            if (g.self_mode || g.nB_cols >= 64u) {
                locate_wg();
                if (alive) {
                    if (g.self_mode) {
                        uint32_t pos = j_wg - i_wg - 1u + threadIdx.x, len = g.n_total - 1u - i_wg, ii = i_wg;
                        while (pos >= len) {
                            pos -= len;
                            ++ii;
                            --len;
                        }
                        i = ii;
                        j = ii + 1u + pos;
                    } else {
                        uint32_t jj = j_wg + threadIdx.x, ii = i_wg;
                        while (jj >= g.nB_cols) {
                            jj -= g.nB_cols;
                            ++ii;
                        }
                        i = ii;
                        j = jj;
                    }
                }
            } else if (alive) {
                eb_pair_of(g, p + g.out_base, i, j);
            }
        }
        if (g.alive_count != nullptr && lane == 0u) atomicAdd(&g.alive_count[blockIdx.x & 1023u], (uint32_t)__popcll(alive_mask));
    }
    extern __shared__ __attribute__((aligned(16))) uint2 eb_lds_rows[];   // [2][2 ss64][7]: the workgroup's row slices of length index NK (see above)
    bool staged = false;
    if (g.lds_rows) {                       // (workgroup-uniform; every thread is still here)
        if (__syncthreads_or(alive ? 1 : 0)) {
            locate_wg();
            const uint32_t per_row = g.ss64 * 14u;
            const uint2 *src = reinterpret_cast<const uint2 *>(g.rows_ref + (((uint64_t)i_wg * g.nk_total + NK) * g.ss64) * BBITS);
            const size_t next_row = (size_t)g.nk_total * g.ss64 * BBITS;
            const uint32_t staged_rows = g.blocked ? 1u : 2u;
            for (uint32_t x = threadIdx.x; x < staged_rows * per_row; x += blockDim.x) {
                const uint32_t r = x >= per_row ? 1u : 0u;   // (the row slab ends in pad rows: row i_wg + 1 always exists)
                eb_lds_rows[x] = src[(size_t)r * next_row + (x - r * per_row)];
            }
            __syncthreads();
            staged = true;
        }
    }
    if (alive_mask != 0ull) {
        bool more = alive;
        uint32_t u_first = 0;             // this lane's first length index beyond NK not yet counted
        if (staged && g.ahead) {
            const uint32_t row_off = i - i_wg;
            const bool fast = alive && row_off < (g.blocked ? 1u : 2u);
            const uint64_t fast_mask = __ballot(fast);
            if (fast_mask != 0ull) {
                const uint32_t same0 = eb_first_length_ahead(g, fast_mask, j, row_off, eb_lds_rows, lane);
                if (fast) {
                    bool pass0 = same0 >= g.min_alive;
                    if constexpr (COMP) {
                        if (!pass0 && same0 > expected) pass0 = eb_passes_comp(g, same0, expected, g.compA[i], g.compB[j]);
                    }
                    if (!pass0) {
                        more = false;                        // jaccard.rs:89-91: break
                    } else {
                        all[NK] = same0;
                        ++passed;
                        u_first = 1u;
                        more = NK + 1u < g.nk_total;
                    }
                }
            }
        }
        uint64_t todo = __ballot(more);
        while (todo != 0ull) {
            const int l = __builtin_ctzll(todo);
            todo &= todo - 1ull;
            const uint32_t i_l = (uint32_t)__builtin_amdgcn_readlane((int)i, l), j_l = (uint32_t)__builtin_amdgcn_readlane((int)j, l);
            const uint32_t u_l = (uint32_t)__builtin_amdgcn_readlane((int)u_first, l);
#pragma unroll
            for (int u = 0; u < EXT; ++u) {
                if ((uint32_t)u < u_l) continue;                    // (wave-uniform)
                if ((uint32_t)(NK + u) >= g.nk_total) break;
                const uint32_t same = eb_same_bins(g.rows_ref, g.cols_ref, g.nk_total, g.ss64, i_l, j_l, (uint32_t)(NK + u), lane);
                if constexpr (COMP) {
                    if (same < g.min_alive && (same <= expected || !eb_passes_comp(g, same, expected, g.compA[i_l], g.compB[j_l]))) break;   // (wave-uniform)
                } else {
                    if (same < g.min_alive) break;                  // jaccard.rs:89-91: break (wave-uniform)
                }
                if ((int)lane == l) {
                    all[NK + u] = same;
                    ++passed;
                }
            }
        }
    }
    float2 res = make_float2(1.0f, 1.0f);     // a fit over fewer than three lengths (jaccard.rs:117)
    if (passed >= 3u) {
        if constexpr (COMP) {
            if (!have_ij && !alive && !ij_lane) eb_pair_of(g, p + g.out_base, i, j);   // (a pair that left inside the counted lengths with three points)
            res = eb_fit_of_counts_comp(g.kf, g.ss64, g.compA[i], g.compB[min(j, g.nB_cols - 1u)], g.cutoff, g.log_variant, passed, all[0], all[1], all[2], all[3], all[4],
                                        all[5], all[6], all[7]);
        } else {
            res = eb_fit_of_counts(g.ytab, g.kf, maxnbits, passed, all[0], all[1], all[2], all[3], all[4], all[5], all[6], all[7]);
        }
    }
    if (in_range) ((float2 *)g.out)[p] = res;
}

// The early-break epilogue of a row band of the symmetric core/accessory self kNN: see EpilogueKnnArgs (kernels.h).
// A wave = KNN_BLOCKS consecutive 64-column blocks of ONE row of the band (the blocks the row bits stand for; the view starts on a
// block boundary), a lane one column of each.  This kernel runs BESIDE the next band's counts kernel, and every wave of it
// that is resident displaces a wave of that kernel (which fills the register file by itself): what it costs is its waves'
// residence time, i.e. the length of its chains of dependent loads (profiles/r06_cfg5_coreacc.md).  So
//   A. the counts of all the wave's blocks are requested at once, and whether a pair leaves the reference's loop is decided
//      on the counts themselves (min_alive): a block all of whose pairs left with fewer than three lengths -- nearly every
//      block between unrelated genomes -- is (1, 1) throughout: no table look-up, no regression, and once every list is
//      full no store and no mark either;
//   B. the pairs still in the running -- of ALL the wave's blocks -- are completed one after the other with the NEXT pair's
//      column slice already requested: the row's slice of the first length not counted is read once per wave (every pair of
//      the wave shares the row) and each column slice is one contiguous run (eb_trip); the completed counts wait in LDS;
//   C. block by block, the pairs that have a fit run the reference's sums and regression; records, marks, turned copy.
// TRIPS: trips of 32 chunks of a slice (1 or 2: the row's slice and the next column's are kept in registers; 0: any sketch
// size, nothing kept, nothing requested ahead).
constexpr uint32_t KNN_BLOCKS = 4;
constexpr uint32_t KNN_MAXKE = 4;
constexpr uint32_t KNN_MAXEXT = 6;      // lengths beyond the counted ones (nk_total <= 8, nk >= 2)

template <int TRIPS>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(4, TRIPS == 0 ? 8 : 5))) void coreacc_epilogue_knn_kernel(const EpilogueKnnArgs g)
{
    __shared__ uint16_t ext[2][KNN_BLOCKS][KNN_MAXEXT][64];   // completed bin-match counts of the pairs still in the running (0xFFFF: not looked at)
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // COLUMN-GROUP-MAJOR ORDER PER XCD (g.xcd_blocked; gridDim.x is a multiple of the XCDs then): the workgroups of the launch go to the
    // XCDs in turns of their linear index; XCD x takes the column groups x, x + 8, ... one after the other, each for ALL the
    // launch's rows before the next -- the group's 512 column slices (a workgroup = 2 waves x 256 columns; 1.75 MB at 2 048 bins) stay in that XCD's L2 while the rows
    // pass, and a slice read for one row's completion is found there by the ~30 other rows of the band that need it.  (Row-major, a
    // slice's next reader comes ~67 rows later -- 1.5 % of the pairs are still in the running -- and by then the XCD has read
    // 130 MB of other slices: every completion is an HBM gather.)
    uint32_t bx = blockIdx.x, by = blockIdx.y;
    if (g.xcd_blocked) {
        const uint32_t lin = blockIdx.y * gridDim.x + blockIdx.x, xcd = lin & 7u, slot = lin >> 3;
        const uint32_t cg_l = slot / gridDim.y;
        by = slot - cg_l * gridDim.y;
        bx = cg_l * 8u + xcd;
    }
    const uint32_t row = g.row_base + by;
    const uint32_t c_wave = (bx * (blockDim.x >> 6) + wave) * (64u * KNN_BLOCKS);
    const uint32_t maxnbits = g.ss64 * 64u, halves = g.ss64 * 2u;
    const uint32_t i_s = g.row_sample0 + row;
    if (c_wave >= g.nB) return;
    // the counts of block u's pairs at the counted lengths.  They are NOT kept across phase B: a register this kernel holds is a
    // register the counts kernel beside it cannot have (that kernel fills the file with 4 waves of 128), and two waves of 64
    // displace what one wave of 65 ... 128 does; the rare pair with a fit reads its counts again in phase C.
    auto load_counts = [&](uint32_t u, uint32_t (&cn)[KNN_MAXKE]) {
        const uint32_t c = min(c_wave + u * 64u + lane, g.nB - 1u);   // (lanes past the row's end shadow its last pair and store nothing)
        const uint64_t p = (uint64_t)row * g.nB + c;
#pragma unroll
        for (uint32_t t = 0; t < KNN_MAXKE; ++t) {
            const uint64_t at = (uint64_t)t * g.n_pairs + p;
            cn[t] = t < g.nk ? (g.cnt_u16 ? (uint32_t)reinterpret_cast<const uint16_t *>(g.counts)[at] : g.counts[at]) : 0u;
        }
    };
    const bool by_count = g.min_alive != EB_NONE;
    const uint32_t thr_row = g.r_bits != nullptr ? g.r_thr[(size_t)row * g.r_thr_stride] : 0u;
    auto stops = [&](uint32_t same) { return by_count ? same < g.min_alive : g.ytab[same <= maxnbits ? same : maxnbits] < g.tolerance; };
    // A. the reference's loop over the first g.nk lengths, on the counts: how many lengths pass before the first that does not
    uint32_t passed_pk = 0;                 // passed[u] in byte u
    uint64_t alive_mask[KNN_BLOCKS];
    uint32_t n_alive = 0;
    {
        uint32_t cnt[KNN_BLOCKS][KNN_MAXKE];
#pragma unroll
        for (uint32_t u = 0; u < KNN_BLOCKS; ++u) load_counts(u, cnt[u]);
        if (by_count) {   // (the usual case, decided ONCE: integer compares and nothing else)
#pragma unroll
            for (uint32_t u = 0; u < KNN_BLOCKS; ++u) {
                const bool in_range = c_wave + u * 64u + lane < g.nB;
                uint32_t passed = 0;
                bool run = true;
#pragma unroll
                for (uint32_t t = 0; t < KNN_MAXKE; ++t) {
                    run = run && (t >= g.nk || cnt[u][t] >= g.min_alive);   // jaccard.rs:89-91: break
                    passed += run && t < g.nk ? 1u : 0u;
                }
                passed_pk |= passed << (8u * u);
                alive_mask[u] = __ballot(in_range && run && g.nk_total > g.nk);
                n_alive += (uint32_t)__popcll(alive_mask[u]);
            }
        } else {
#pragma unroll
            for (uint32_t u = 0; u < KNN_BLOCKS; ++u) {
                const bool in_range = c_wave + u * 64u + lane < g.nB;
                uint32_t passed = 0;
                bool stopped = false;
#pragma unroll
                for (uint32_t t = 0; t < KNN_MAXKE; ++t) {
                    if (t < g.nk && !stopped) {
                        if (stops(cnt[u][t])) stopped = true;   // jaccard.rs:89-91: break
                        else ++passed;
                    }
                }
                passed_pk |= passed << (8u * u);
                alive_mask[u] = __ballot(in_range && !stopped && g.nk_total > g.nk);
                n_alive += (uint32_t)__popcll(alive_mask[u]);
            }
        }
    }
    // B. the pairs still in the running, in block order (pass_mask: those that passed the first length not counted -- their counts wait in LDS)
    uint64_t pass_mask[KNN_BLOCKS];
#pragma unroll
    for (uint32_t u = 0; u < KNN_BLOCKS; ++u) pass_mask[u] = 0ull;
    if (n_alive != 0u) {                // (wave-uniform)
        if (g.alive_count != nullptr && lane == 0u) atomicAdd(&g.alive_count[(bx + by * 7u) & 1023u], n_alive);   // (1 024 slots)
        const uint2 *pi = reinterpret_cast<const uint2 *>(g.rows_ref + (((uint64_t)i_s * g.nk_total + g.nk) * g.ss64) * BBITS);
        constexpr int KEPT = TRIPS > 0 ? TRIPS : 1;
        uint2 a_row[KEPT][7], b_next[KEPT][7];
        if constexpr (TRIPS > 0) {
#pragma unroll
            for (int tr = 0; tr < TRIPS; ++tr) {
#pragma unroll
                for (int q = 0; q < 7; ++q) a_row[tr][q] = (uint32_t)tr * 64u + lane < halves ? pi[((size_t)tr * 64u + lane) * 7 + q] : make_uint2(0u, 0u);
            }
        }
        // (u, l) of the next pair still in the running: lane u of vm_lo / vm_hi holds block u's mask (a register array indexed by
        // a running block number turned the walk into a scalar state machine of ~40 instructions per step)
        uint32_t vm_lo = 0u, vm_hi = 0u;
#pragma unroll
        for (uint32_t u = 0; u < KNN_BLOCKS; ++u) {
            if (lane == u) {
                vm_lo = (uint32_t)alive_mask[u];
                vm_hi = (uint32_t)(alive_mask[u] >> 32);
            }
        }
        uint32_t cur_u = 0;
        uint64_t cur_m = alive_mask[0];
        auto advance = [&](uint32_t &u_out, uint32_t &l_out) {   // (the caller counts: never called past the last pair)
            while (cur_m == 0ull) {
                ++cur_u;
                cur_m = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)vm_hi, (int)cur_u) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)vm_lo, (int)cur_u);
            }
            l_out = (uint32_t)__builtin_ctzll(cur_m);
            cur_m &= cur_m - 1ull;
            u_out = cur_u;
        };
        auto column_of = [&](uint32_t u, uint32_t l) {   // view column -> sample id (wave-uniform)
            return g.col_sample0 + c_wave + u * 64u + l;
        };
        // the reference's loop from the first length not counted on, given that length's count (the later ones, reached by one pair
        // in eight, are read when they are needed)
        auto finish = [&](uint32_t same, uint32_t u_e, uint32_t l_e) {
            // nearly every pair that survived the counted lengths by chance leaves here: nothing is written for it (it has
            // g.nk points; phase C knows)
            if (stops(same)) return;                           // (wave-uniform) jaccard.rs:89-91
            const uint64_t bit = 1ull << l_e;
#pragma unroll
            for (uint32_t u = 0; u < KNN_BLOCKS; ++u) pass_mask[u] |= u == u_e ? bit : 0ull;
            if (lane == l_e) {
                ext[wave][u_e][0][lane] = (uint16_t)same;       // (u_e: wave-uniform)
#pragma unroll
                for (uint32_t x = 1; x < KNN_MAXEXT; ++x) ext[wave][u_e][x][lane] = 0xFFFFu;
            }
            const uint32_t j_e = column_of(u_e, l_e);
            for (uint32_t t = g.nk + 1u; t < g.nk_total; ++t) {
                same = eb_same_bins(g.rows_ref, g.cols_ref, g.nk_total, g.ss64, i_s, j_e, t, lane);
                if (stops(same)) break;                        // (wave-uniform)
                if (lane == l_e) ext[wave][u_e][t - g.nk][lane] = (uint16_t)same;
            }
        };
        if constexpr (TRIPS == 1) {
            // The next pair's slice is requested as soon as this one's registers are free (after the 14 bit operations, before the
            // wave sum); every lane loads (those past the slice's end re-read its last half chunk and count nothing).  TWO register
            // sets -- pair n + 1 requested before pair n is waited for -- were measured too: 2.53 s either way at n = 300 000,
            // and 15 registers dearer.
            const uint32_t h_c = min(lane, halves - 1u);
            auto request1 = [&](uint2 (&bq)[7], uint32_t j) {
                const uint2 *pj = reinterpret_cast<const uint2 *>(g.cols_ref + (((uint64_t)j * g.nk_total + g.nk) * g.ss64) * BBITS) + (size_t)h_c * 7;
#pragma unroll
                for (int q = 0; q < 7; ++q) bq[q] = pj[q];
            };
            auto count_tail = [&](uint32_t mlo, uint32_t mhi) {
                mlo |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mlo, 0xB1, 0xF, 0xF, true);   // quad_perm [1, 0, 3, 2]
                mhi |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mhi, 0xB1, 0xF, 0xF, true);
                const uint32_t part = ((lane & 1u) == 0u && lane < halves) ? (uint32_t)__builtin_popcount(mlo) + (uint32_t)__builtin_popcount(mhi) : 0u;
                return maxnbits - eb_wave_sum(part);
            };
            uint2 b0[7];
            uint32_t u0 = 0, l0 = 0, u1 = 0, l1 = 0;
            uint32_t left = n_alive;                       // pairs not yet counted; b0 holds the first of them
            advance(u0, l0);
            request1(b0, column_of(u0, l0));
            while (left > 1u) {
                uint32_t mlo = 0, mhi = 0;
#pragma unroll
                for (int q = 0; q < 7; ++q) {
                    mlo = acc_mismatch<true>(mlo, a_row[0][q].x, b0[q].x);
                    mhi = acc_mismatch<true>(mhi, a_row[0][q].y, b0[q].y);
                }
                advance(u1, l1);                            // the next pair's slice is on its way while this one is summed up
                request1(b0, column_of(u1, l1));
                finish(count_tail(mlo, mhi), u0, l0);
                u0 = u1;
                l0 = l1;
                --left;
            }
            {
                uint32_t mlo = 0, mhi = 0;
#pragma unroll
                for (int q = 0; q < 7; ++q) {
                    mlo = acc_mismatch<true>(mlo, a_row[0][q].x, b0[q].x);
                    mhi = acc_mismatch<true>(mhi, a_row[0][q].y, b0[q].y);
                }
                finish(count_tail(mlo, mhi), u0, l0);
            }
        } else {
            auto request = [&](uint32_t j) {                 // the column sample's slice of the first length not counted
                if constexpr (TRIPS > 0) {
                    const uint2 *pj = reinterpret_cast<const uint2 *>(g.cols_ref + (((uint64_t)j * g.nk_total + g.nk) * g.ss64) * BBITS);
#pragma unroll
                    for (int tr = 0; tr < TRIPS; ++tr) {
#pragma unroll
                        for (int q = 0; q < 7; ++q) b_next[tr][q] = (uint32_t)tr * 64u + lane < halves ? pj[((size_t)tr * 64u + lane) * 7 + q] : make_uint2(0u, 0u);
                    }
                }
            };
            uint32_t u_e = 0, l_e = 0, u_n = 0, l_n = 0;
            uint32_t left = n_alive;
            advance(u_e, l_e);
            request(column_of(u_e, l_e));
            while (left != 0u) {
                const uint32_t j_e = column_of(u_e, l_e);
                uint32_t same;
                --left;
                if constexpr (TRIPS > 0) {
                    uint32_t part = 0;
#pragma unroll
                    for (int tr = 0; tr < TRIPS; ++tr) {
                        const uint32_t h = (uint32_t)tr * 64u + lane;
                        uint32_t mlo = 0, mhi = 0;
#pragma unroll
                        for (int q = 0; q < 7; ++q) {
                            mlo = acc_mismatch<true>(mlo, a_row[tr][q].x, b_next[tr][q].x);
                            mhi = acc_mismatch<true>(mhi, a_row[tr][q].y, b_next[tr][q].y);
                        }
                        mlo |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mlo, 0xB1, 0xF, 0xF, true);   // quad_perm [1, 0, 3, 2]
                        mhi |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mhi, 0xB1, 0xF, 0xF, true);
                        part += ((lane & 1u) == 0u && h < halves) ? (uint32_t)__builtin_popcount(mlo) + (uint32_t)__builtin_popcount(mhi) : 0u;
                    }
                    // the next pair's slice is on its way while this one is summed up
                    if (left != 0u) {
                        advance(u_n, l_n);
                        request(column_of(u_n, l_n));
                    }
                    same = maxnbits - eb_wave_sum(part);
                } else {
                    same = eb_same_bins(g.rows_ref, g.cols_ref, g.nk_total, g.ss64, i_s, j_e, g.nk, lane);
                    if (left != 0u) advance(u_n, l_n);
                }
                finish(same, u_e, l_e);
                u_e = u_n;
                l_e = l_n;
            }
        }
    }
    // C. block by block: sums and regression where there is a fit, records, marks, the turned copy
#pragma unroll
    for (uint32_t u = 0; u < KNN_BLOCKS; ++u) {
        const uint32_t c_raw = c_wave + u * 64u + lane;
        if (c_wave + u * 64u >= g.nB) break;                    // (wave-uniform)
        const bool in_range = c_raw < g.nB;
        const uint32_t c = in_range ? c_raw : g.nB - 1u;
        const uint64_t p = (uint64_t)row * g.nB + c;
        const bool alive = (alive_mask[u] >> lane) & 1ull;
        // a pair has a fit (three or more points, jaccard.rs:117) if its counted lengths gave three, or if it was still in the running
        // and passed the first completed length (g.nk >= 2 counted ones + that one)
        const bool ext_ok = alive && ((pass_mask[u] >> lane) & 1ull);
        const uint32_t passed_u = (passed_pk >> (8u * u)) & 0xFFu;
        const bool fit = in_range && (passed_u >= 3u || ext_ok);
        const bool any_fit = __ballot(fit) != 0ull;
        // NEARLY EVERY BLOCK: all 64 pairs are (1, 1), the row's list is full (a 1 marks nothing), so is every column's: nothing to
        // store, nothing to mark
        if (!any_fit && g.r_bits != nullptr && !(sortable_bits(1.0f) < thr_row) && (g.out_t == nullptr || g.plain_marks_nothing)) continue;
        float2 v = make_float2(1.0f, 1.0f);
        if (any_fit) {
            // the lengths passed beyond the counted ones, from the counts phase B left in LDS
            uint32_t e[KNN_MAXEXT];
            uint32_t n_ext = 0;
#pragma unroll
            for (uint32_t x = 0; x < KNN_MAXEXT; ++x) e[x] = 0u;
            if (ext_ok) {
                bool run = true;
#pragma unroll
                for (uint32_t x = 0; x < KNN_MAXEXT; ++x) {
                    e[x] = ext[wave][u][x][lane];
                    run = run && e[x] != 0xFFFFu && !stops(e[x]);
                    n_ext += run ? 1u : 0u;
                }
            }
            if (fit) {
                // the pair's counts in length order: the counted ones, then the completed ones (g.nk is 2 ... 4)
                uint32_t cn[KNN_MAXKE], cc[EB_MAXK];
                load_counts(u, cn);
#pragma unroll
                for (uint32_t t = 0; t < EB_MAXK; ++t) {
                    cc[t] = t < KNN_MAXKE ? cn[t < KNN_MAXKE ? t : 0u] : 0u;
#pragma unroll
                    for (uint32_t x = 0; x < KNN_MAXEXT; ++x) {
                        if (t == g.nk + x) cc[t] = e[x];
                    }
                }
                v = eb_fit_of_counts(g.ytab, g.kf, maxnbits, passed_u + n_ext, cc[0], cc[1], cc[2], cc[3], cc[4], cc[5], cc[6], cc[7]);
            }
        }
        // does this 64-column block bring the row anything below its knn-th best?  (the key is the core distance)  The merge
        // reads the marked blocks of a row only, so an unmarked block's 64 records are not even stored.
        bool store = in_range;
        if (g.r_bits != nullptr) {
            const bool marked = __ballot(in_range && sortable_bits(v.x) < thr_row) != 0ull;
            if (marked && lane == 0u) {
                const uint32_t blk = c_raw >> 6;
                atomicOr(&g.r_bits[(size_t)row * g.r_bits_stride + (blk >> 5)], 1u << (blk & 31u));
            }
            store = store && marked;
        }
        if (store) ((float2 *)g.out)[p] = v;
        // the turned copy: pre-filled with (1, 1); everything else is stored, and marked where it beats the column's knn-th
        // best.  Once every list holds knn candidates (none above 1: plain_marks_nothing, the host knows) a (1, 1) marks
        // nothing, so a block of nothing but (1, 1) -- nearly every block -- is done here: no look at its columns' thresholds.
        const bool plain = __float_as_uint(v.x) == 0x3F800000u && __float_as_uint(v.y) == 0x3F800000u;
        const bool turn = in_range && c >= g.t_col_begin && !(plain && g.plain_marks_nothing);
        if (g.out_t == nullptr || __ballot(turn) == 0ull) continue;
        if (turn) {
            if (!plain) reinterpret_cast<float2 *>(g.out_t)[(size_t)(c - g.t_col_begin) * g.t_stride + row] = v;
            if (g.t_flag != nullptr && sortable_bits(v.x) < g.t_thr[(size_t)c * g.t_thr_stride]) {
                g.t_flag[c] = g.t_flag_value;
                if (g.t_bits != nullptr) {
                    const uint32_t tb = row >> 5;
                    atomicOr(&g.t_bits[(size_t)c * g.t_bits_stride + (tb >> 5)], 1u << (tb & 31u));
                }
            }
        }
    }
}

hipError_t launch_coreacc_epilogue_knn(const EpilogueKnnArgs &args, hipStream_t stream)
{
    if (args.rows == 0 || args.nB == 0) return hipSuccess;
    if (args.nk > KNN_MAXKE) return hipErrorInvalidValue;
    const uint32_t wg_threads = 128u;   // (2 waves = 512 columns = 1.75 MB of slices per column group at 2 048 bins; 4 waves: +0.4 %)
    const uint32_t per_wg = wg_threads * KNN_BLOCKS;
    for (uint32_t r0 = 0; r0 < args.rows; r0 += 32768u) {
        EpilogueKnnArgs a = args;
        a.row_base = r0;
        // (gridDim.x rounded up to a multiple of the XCDs alone -- a column group on the same XCD in every row, rows still the slow
        // index -- changed nothing: 2.51 s either way at n = 300 000; see the kernel for the order that goes with it)
        uint32_t gx = (args.nB + per_wg - 1u) / per_wg;
        a.xcd_blocked = a.xcd_blocked == 2u || (a.xcd_blocked == 1u && gx >= 32u) ? 1u : 0u;   // (narrow views: the padding to a multiple of 8 would be mostly empty workgroups; 2: forced, tests)
        if (a.xcd_blocked) gx = (gx + 7u) & ~7u;
        const dim3 gr(gx, std::min(32768u, args.rows - r0)), bl(wg_threads);
        if (args.ss64 <= 32u) hipLaunchKernelGGL(coreacc_epilogue_knn_kernel<1>, gr, bl, 0, stream, a);
        else if (args.ss64 <= 64u) hipLaunchKernelGGL(coreacc_epilogue_knn_kernel<2>, gr, bl, 0, stream, a);
        else hipLaunchKernelGGL(coreacc_epilogue_knn_kernel<0>, gr, bl, 0, stream, a);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
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
    uint32_t lead = 0u;
    for (uint32_t t = 0; t < g.nk && t < 8u; ++t) {
        const uint32_t same = eb_same_bins(g.rows_ref, g.cols_ref, g.nk, g.ss64, i, j, t, lane);
        bool stop;
        if (g.has_comp) stop = glibc_log(jaccard_from_samebits_dev(same, g.ss64, true, c1, c2, g.cutoff), g.log_variant) < g.tolerance;
        else if (g.min_alive != EB_NONE) stop = same < g.min_alive;
        else stop = g.ytab[same <= maxnbits ? same : maxnbits] < g.tolerance;
        if (stop) break;                                       // jaccard.rs:89-91
        ++lead;
    }
    if (lane == 0u) atomicAdd(&g.hist[(size_t)blk * 9u + lead], 1u);
}

hipError_t launch_early_break_sample(const EbSampleArgs &args, hipStream_t stream)
{
    const uint64_t waves = (uint64_t)args.blk_rows * args.blk_cols * args.samples;
    if (waves == 0 || args.n_rows == 0 || args.n_cols == 0) return hipSuccess;
    const uint64_t blocks = (waves * 64u + 255u) / 256u;
    if (blocks >= (1ull << 31)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(early_break_sample_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, args);
    return hipGetLastError();
}

#undef SKL_DPP_ADD

template <int N>
static void launch_lean_n(bool sliced, bool comp, dim3 gr, dim3 bl, size_t lds, hipStream_t stream, const EpilogueArgs &a)
{
    if (sliced) {
        if (comp) hipLaunchKernelGGL((coreacc_epilogue_lean_kernel<true, N, true>), gr, bl, lds, stream, a);
        else hipLaunchKernelGGL((coreacc_epilogue_lean_kernel<true, N, false>), gr, bl, lds, stream, a);
    } else {
        if (comp) hipLaunchKernelGGL((coreacc_epilogue_lean_kernel<false, N, true>), gr, bl, lds, stream, a);
        else hipLaunchKernelGGL((coreacc_epilogue_lean_kernel<false, N, false>), gr, bl, lds, stream, a);
    }
}

// the lean form takes: one ke for every pair, no completeness correction or one with every value in (0, 1], the break decided on the count, 2 ... 4 lengths counted, k-major
// counts (u16 in one plane, or u32 in the planes of a tail-sliced launch)
bool coreacc_epilogue_is_lean(const EpilogueArgs &a)
{
    const bool early = a.nk_total > a.nk || a.block_ke != nullptr;
    return a.lean != 0u && early && a.block_ke == nullptr && (!a.has_comp || a.comp_lean != 0u) && a.min_alive != EB_NONE && a.nk >= 2u && a.nk <= 4u && a.nk_total > a.nk &&
           a.nk_total <= EB_MAXK && a.pair_stride == 1u && (a.cnt_u16 != 0u ? a.n_slices == 1u && a.rezero_plane1 == 0u : a.n_slices >= 1u);
}

hipError_t launch_coreacc_epilogue_r6(const EpilogueArgs &args, hipStream_t stream)
{
    if (args.n_pairs == 0) return hipSuccess;
    const bool early = args.nk_total > args.nk || args.block_ke != nullptr;
    if (early && args.nk_total > EB_MAXK) return hipErrorInvalidValue;
    EpilogueArgs a = args;
    uint64_t blocks = (args.n_pairs + 255) / 256;
    if (!early) a.blocked = 0u;
    if (a.blocked) {   // blocks of (1 << blk_row_shift) rows x 256 columns, one workgroup per row of a block, dealt to the XCDs whole
        if (a.blk_row_shift < 5u || a.blk_row_shift > 12u) a.blk_row_shift = 10u;
        const uint32_t br = 1u << a.blk_row_shift;
        a.blk_rb = (a.row_end - a.row_begin + br - 1u) / br;
        a.blk_cb0 = a.self_mode ? (a.row_begin + 1u) / 256u : 0u;
        a.blk_cb = (a.nB_cols + 255u) / 256u - a.blk_cb0;
        const uint64_t n_xcd = 1ull << a.xcd_shift;
        const uint64_t per_xcd = ((uint64_t)a.blk_rb * a.blk_cb + n_xcd - 1) / n_xcd;
        blocks = (per_xcd * br) << a.xcd_shift;
        if (blocks >= (1ull << 32)) return hipErrorInvalidValue;
    }
    const size_t lds = early && a.block_ke == nullptr && a.ss64 * 224ull <= 16384ull && a.lds_rows != 0u ? (size_t)a.ss64 * 224u : 0u;
    a.lds_rows = lds != 0 ? 1u : 0u;
    const bool lean = coreacc_epilogue_is_lean(a);
    // (a dispatch packet counts WORK-ITEMS in 32 bits: 2^23 workgroups of 256 per launch at most)
    constexpr uint64_t MAX_WG = 1ull << 23;
    for (uint64_t w0 = 0; w0 < blocks; w0 += MAX_WG) {
        a.wg_base = (uint32_t)w0;
        const dim3 gr((unsigned)std::min(MAX_WG, blocks - w0)), bl(256);
        if (lean) {
            const bool sl = a.cnt_u16 == 0u, cm = a.has_comp != 0u;
            switch (a.nk) {
            case 2: launch_lean_n<2>(sl, cm, gr, bl, lds, stream, a); break;
            case 3: launch_lean_n<3>(sl, cm, gr, bl, lds, stream, a); break;
            default: launch_lean_n<4>(sl, cm, gr, bl, lds, stream, a); break;
            }
        } else if (a.has_comp) hipLaunchKernelGGL(coreacc_epilogue_kernel_r6<true>, gr, bl, lds, stream, a);
        else hipLaunchKernelGGL(coreacc_epilogue_kernel_r6<false>, gr, bl, lds, stream, a);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace skl
