// pair_cand.hip -- distances for an explicit candidate list (gfx950).
//
// The reference's `self_dists_knn_precluster` (src/distances/mod.rs:399-553) computes, for
// every sample i, the single-k Jaccard / ANI key against the samples an inverted index
// returns for it (any shared bin), instead of against all n.  The pair space is then a
// ragged list: row_offsets[i] .. row_offsets[i+1] into an array of candidate sample ids.
//
// One wave = one row and up to 64 of its candidates.  The candidate sample lives in the lane
// and is read straight from the reference layout (7 x 16 B per chunk, scattered between
// lanes -- the list has no structure to tile); the row sample is wave-uniform and arrives
// through the scalar cache.  This kernel is bound by the gather (7 168 B per pair at
// sketchsize64 = 64), not by the VALU, so the half-rate scalar operand is the right
// trade: no LDS, no staging, any list shape.
#include "device_common.hpp"

namespace skl {

typedef const __attribute__((address_space(4))) uint32_t *const_u32_ptr_c;

#ifdef SKL_AB   // (round 3's form, SKL_CAND_KERNEL=lanes: A/B build only)
__global__ __launch_bounds__(LANES *WAVES_PER_WG) void pair_cand_kernel(const CandArgs c, const PairArgs g)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t blk = (blockIdx.x & ((1u << c.xcd_shift) - 1u)) * c.blocks_per_xcd + (blockIdx.x >> c.xcd_shift);
    if ((blockIdx.x >> c.xcd_shift) >= c.blocks_per_xcd) return;
    const uint64_t w = (uint64_t)blk * WAVES_PER_WG + wave;
    if (w >= c.n_work) return;
    const uint32_t row = c.work_row[w];
    const uint64_t start = c.work_start[w];
    const uint64_t row_end = c.row_offsets[row + 1];
    const uint32_t cnt = (uint32_t)((row_end - start) < 64ull ? (row_end - start) : 64ull);
    uint32_t j = lane < cnt ? c.cand[start + lane] : row;   // idle lanes redo the diagonal
    // Symmetric lists (j in cand(i) <=> i in cand(j), what "any shared bin" yields): only the
    // j > i half is evaluated, and the key is stored for both rows.  The kernel is gather-bound,
    // so the lanes that sit out (their j becomes the row itself: a broadcast load) save their
    // 3.5-7 KB of scattered traffic per pair.
    bool active = lane < cnt;
    if (c.symmetric) {
        active = active && j > row;
        if (__ballot(active) == 0) return;
        if (!active) j = row;
    }

    const size_t kmer_stride = (size_t)g.ss64 * BBITS;
    const size_t sample_stride = kmer_stride * g.nk;
    const uint64_t *pi = g.A + (size_t)row * sample_stride + (size_t)g.k_begin * kmer_stride;
    const uint4 *pj = reinterpret_cast<const uint4 *>(g.A + (size_t)j * sample_stride + (size_t)g.k_begin * kmer_stride);

    uint32_t mism = 0;
    uint4 b[7];
#pragma unroll
    for (int q = 0; q < 7; ++q) b[q] = pj[q];
    for (uint32_t ch = 0; ch < g.ss64; ++ch) {
        const uint32_t cn = ch + 1 < g.ss64 ? ch + 1 : ch;
        uint4 bn[7];   // next chunk of the candidate: in flight under this chunk's work
#pragma unroll
        for (int q = 0; q < 7; ++q) bn[q] = pj[(size_t)cn * 7 + q];
        const_u32_ptr_c a = (const_u32_ptr_c)(uintptr_t)(pi + (size_t)ch * BBITS);
        uint32_t mlo = 0, mhi = 0;
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            // reference layout: plane p = (lo, hi) dwords; b[q] = planes 2q, 2q+1
            mlo = acc_mismatch<true>(mlo, a[4 * q + 0], b[q].x);
            mhi = acc_mismatch<true>(mhi, a[4 * q + 1], b[q].y);
            mlo = acc_mismatch<true>(mlo, a[4 * q + 2], b[q].z);
            mhi = acc_mismatch<true>(mhi, a[4 * q + 3], b[q].w);
        }
        mism += __builtin_popcount(mlo) + __builtin_popcount(mhi);
#pragma unroll
        for (int q = 0; q < 7; ++q) b[q] = bn[q];
    }
    if (active) {
        const float key = jaccard_out_value(g, row, j, mism);
        c.keys[start + lane] = key;
        if (c.symmetric) {   // row j's copy: position of `row` in its (ascending) list
            uint64_t lo = c.row_offsets[j], hi = c.row_offsets[j + 1];
            while (lo < hi) {
                const uint64_t mid = (lo + hi) >> 1;
                if (c.cand[mid] < row) lo = mid + 1; else hi = mid;
            }
            if (lo < c.row_offsets[j + 1] && c.cand[lo] == row) c.keys[lo] = key;
        }
    }
}
#endif

// Round 4: the same work item -- a row and up to 64 of its candidates -- with the LANES ACROSS THE SKETCH instead of across the
// candidates.  The form above has every lane walk its own candidate, 112 bytes per chunk: a wave's request is 64 scattered
// lines at a time, and the HBM-side traffic runs at 4.5 TB/s for the 3.5 KB blocks it actually wants.  Here a wave takes
// the candidates one after the other and reads each as ONE contiguous run (lane l takes the l-th half chunk: planes 0-6 or
// 7-13 of a chunk, 56 bytes; sketches of more than 32 chunks take several trips), ORs the two halves of a chunk between
// neighbouring lanes (DPP), counts, and sums the 64 partial counts with four DPP row shifts + four v_readlane.  ~3x the
// VALU instructions per pair of the form above, which is nothing: the kernel waits for memory either way.
#define SKL_DPP_ADD(v, ctrl) ((v) + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), (ctrl), 0xF, 0xF, true))

// TRIPS = 1 ... 5: sketches of up to 32 x TRIPS chunks -- the ROW's planes are read once per work item and stay in
// registers (7 x 8 bytes per lane and trip), so a candidate costs seven loads per trip and nothing else; TRIPS = 0: any
// size, the row's planes re-read (from L1) beside every candidate's.
template <int TRIPS>
__global__ __launch_bounds__(LANES *WAVES_PER_WG) void pair_cand_rows_kernel(const CandArgs c, const PairArgs g)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t blk = (blockIdx.x & ((1u << c.xcd_shift) - 1u)) * c.blocks_per_xcd + (blockIdx.x >> c.xcd_shift);
    if ((blockIdx.x >> c.xcd_shift) >= c.blocks_per_xcd) return;
    const uint64_t w = (uint64_t)blk * WAVES_PER_WG + wave;
    if (w >= c.n_work) return;
    const uint32_t row = c.work_row[w];
    const uint64_t start = c.work_start[w];
    const uint64_t row_end = c.row_offsets[row + 1];
    const uint32_t cnt = (uint32_t)((row_end - start) < 64ull ? (row_end - start) : 64ull);
    const uint32_t j_mine = lane < cnt ? c.cand[start + lane] : row;
    const size_t kmer_stride = (size_t)g.ss64 * BBITS;
    const size_t sample_stride = kmer_stride * g.nk;
    const uint32_t halves = g.ss64 * 2u;                     // half chunks of 7 planes (56 bytes) per sketch
    const uint2 *pi = reinterpret_cast<const uint2 *>(g.A + (size_t)row * sample_stride + (size_t)g.k_begin * kmer_stride);
    uint32_t mism_mine = 0;
    constexpr int KEPT = TRIPS > 0 ? TRIPS : 1;
    uint2 a_row[KEPT][7];
    if constexpr (TRIPS > 0) {
#pragma unroll
        for (int t = 0; t < TRIPS; ++t) {
#pragma unroll
            for (int q = 0; q < 7; ++q) a_row[t][q] = (uint32_t)t * 64u + lane < halves ? pi[((size_t)t * 64u + lane) * 7 + q] : make_uint2(0u, 0u);
        }
    }
    for (uint32_t cc = 0; cc < cnt; ++cc) {
        const uint32_t j = (uint32_t)__builtin_amdgcn_readlane((int)j_mine, (int)cc);   // (cc is wave-uniform)
        if (c.symmetric && j <= row) continue;               // the other half of a symmetric list: stored from row j's side
        const uint2 *pj = reinterpret_cast<const uint2 *>(g.A + (size_t)j * sample_stride + (size_t)g.k_begin * kmer_stride);
        uint32_t part = 0;
        auto one_trip = [&](uint32_t h, const uint2 *a) {    // half chunk h of the candidate against the row's (a: 7 planes)
            uint32_t mlo = 0, mhi = 0;
            if (h < halves) {
                uint2 b[7];
#pragma unroll
                for (int q = 0; q < 7; ++q) b[q] = pj[(size_t)h * 7 + q];      // one plane (lo, hi) each: the wave's 64 x 56 B are contiguous
#pragma unroll
                for (int q = 0; q < 7; ++q) {
                    mlo = acc_mismatch<true>(mlo, a[q].x, b[q].x);
                    mhi = acc_mismatch<true>(mhi, a[q].y, b[q].y);
                }
            }
            // the other seven planes of this chunk sit in the neighbouring lane (h ^ 1): a bin matches iff all 14 agree
            mlo |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mlo, 0xB1, 0xF, 0xF, true);   // quad_perm [1, 0, 3, 2]
            mhi |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mhi, 0xB1, 0xF, 0xF, true);
            if ((lane & 1u) == 0u && h < halves) part += (uint32_t)__builtin_popcount(mlo) + (uint32_t)__builtin_popcount(mhi);
        };
        if constexpr (TRIPS > 0) {
#pragma unroll
            for (int t = 0; t < TRIPS; ++t) one_trip((uint32_t)t * 64u + lane, a_row[t]);
        } else {
            for (uint32_t h0 = 0; h0 < halves; h0 += 64u) {
                const uint32_t h = h0 + lane;
                uint2 a[7];
#pragma unroll
                for (int q = 0; q < 7; ++q) a[q] = h < halves ? pi[(size_t)h * 7 + q] : make_uint2(0u, 0u);   // the row's: the same addresses for every candidate (L1)
                one_trip(h, a);
            }
        }
        // sum over the wave: four row shifts leave each row of 16 lanes' total in its last lane
        part = SKL_DPP_ADD(part, 0x111);   // row_shr:1
        part = SKL_DPP_ADD(part, 0x112);   // row_shr:2
        part = SKL_DPP_ADD(part, 0x114);   // row_shr:4
        part = SKL_DPP_ADD(part, 0x118);   // row_shr:8
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)part, 15) + (uint32_t)__builtin_amdgcn_readlane((int)part, 31) +
                               (uint32_t)__builtin_amdgcn_readlane((int)part, 47) + (uint32_t)__builtin_amdgcn_readlane((int)part, 63);
        if (lane == cc) mism_mine = total;
    }
    const bool active = lane < cnt && (!c.symmetric || j_mine > row);
    if (active) {
        const float key = jaccard_out_value(g, row, j_mine, mism_mine);
        c.keys[start + lane] = key;
        if (c.symmetric) {   // row j's copy: position of `row` in its (ascending) list
            uint64_t lo = c.row_offsets[j_mine], hi = c.row_offsets[j_mine + 1];
            while (lo < hi) {
                const uint64_t mid = (lo + hi) >> 1;
                if (c.cand[mid] < row) lo = mid + 1; else hi = mid;
            }
            if (lo < c.row_offsets[j_mine + 1] && c.cand[lo] == row) c.keys[lo] = key;
        }
    }
}
#undef SKL_DPP_ADD

hipError_t launch_pair_cand(const CandArgs &c_in, const PairArgs &g, hipStream_t stream)
{
    CandArgs c = c_in;
    if (c.n_work == 0) return hipSuccess;
    const uint64_t blocks = (c.n_work + WAVES_PER_WG - 1) / WAVES_PER_WG;
    c.xcd_shift = g.xcd_shift;
    c.blocks_per_xcd = (uint32_t)((blocks + (1ull << c.xcd_shift) - 1) >> c.xcd_shift);
    const uint64_t grid = (uint64_t)c.blocks_per_xcd << c.xcd_shift;
    if (grid >= (1ull << 31)) return hipErrorInvalidValue;
#ifdef SKL_AB
    if (c.lanes_over_candidates) {
        hipLaunchKernelGGL(pair_cand_kernel, dim3((unsigned)grid), dim3(LANES * WAVES_PER_WG), 0, stream, c, g);
    } else
#endif
    {
        const dim3 gr((unsigned)grid), bl(LANES * WAVES_PER_WG);
        switch ((g.ss64 + 31u) / 32u) {      // trips of 32 chunks
            case 1: hipLaunchKernelGGL(pair_cand_rows_kernel<1>, gr, bl, 0, stream, c, g); break;
            case 2: hipLaunchKernelGGL(pair_cand_rows_kernel<2>, gr, bl, 0, stream, c, g); break;
            case 3: hipLaunchKernelGGL(pair_cand_rows_kernel<3>, gr, bl, 0, stream, c, g); break;
            case 4: hipLaunchKernelGGL(pair_cand_rows_kernel<4>, gr, bl, 0, stream, c, g); break;
            case 5: hipLaunchKernelGGL(pair_cand_rows_kernel<5>, gr, bl, 0, stream, c, g); break;   // (157 chunks: `sketch -s 10000`)
            default: hipLaunchKernelGGL(pair_cand_rows_kernel<0>, gr, bl, 0, stream, c, g); break;
        }
    }
    return hipGetLastError();
}

}  // namespace skl
