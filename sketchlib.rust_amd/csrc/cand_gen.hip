// cand_gen.hip -- Inverted::any_shared_bins (src/inverted.rs:259-268) for every sample at once,
// on the GPU (gfx950): sample j is a candidate of sample i iff their index sketches hold the same
// u16 value at some bin position.  The reference (and this library's host path) walks an inverted
// index per row; with the per-row work at ~sketch_size x cluster_size list entries that is
// billions of scattered updates for a few hundred thousand genomes -- seconds on 64 host
// threads, tens of milliseconds here.
//
//   1. group the samples of every bin by value (counting sort: histogram with global atomics,
//      one exclusive scan of 65 536 counters per bin, scatter);
//   2. one workgroup per row: an n-bit bitmap in LDS, every bin's group of the row's value is
//      OR-ed into it (one wave per bin, lanes over the members), the row itself cleared; then
//      either the population count (pass 1) or the ascending list of set bits written at the
//      row's offset (pass 2; offsets = prefix sum of the counts).
// The n-bit bitmap bounds n at 160 KB * 8 = 1.3 M samples per call.
#include "kernels.h"

namespace skl {

constexpr int CG_THREADS = 256;

__global__ __launch_bounds__(CG_THREADS) void bin_hist_kernel(const CandGenArgs g)
{
    const uint64_t total = (uint64_t)g.n * g.sketch_size;
    for (uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; x < total; x += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t b = (uint32_t)(x % g.sketch_size);
        atomicAdd(&g.starts[(uint64_t)b * 65536u + g.skq[x]], 1u);
    }
}

// one workgroup per bin: counts -> exclusive starts (and a copy that the scatter advances)
__global__ __launch_bounds__(CG_THREADS) void bin_scan_kernel(const CandGenArgs g)
{
    __shared__ uint32_t part[CG_THREADS];
    uint32_t *starts = g.starts + (uint64_t)blockIdx.x * 65536u;
    uint32_t *cursor = g.cursor + (uint64_t)blockIdx.x * 65536u;
    const uint32_t tid = threadIdx.x;
    constexpr uint32_t PER = 65536u / CG_THREADS;
    uint32_t sum = 0;
    for (uint32_t k = 0; k < PER; ++k) sum += starts[tid * PER + k];
    part[tid] = sum;
    __syncthreads();
    if (tid == 0) {
        uint32_t acc = 0;
        for (int t = 0; t < CG_THREADS; ++t) {
            const uint32_t v = part[t];
            part[t] = acc;
            acc += v;
        }
    }
    __syncthreads();
    uint32_t acc = part[tid];
    for (uint32_t k = 0; k < PER; ++k) {
        const uint32_t c = starts[tid * PER + k];
        starts[tid * PER + k] = acc;
        cursor[tid * PER + k] = acc;
        acc += c;
    }
}

__global__ __launch_bounds__(CG_THREADS) void bin_scatter_kernel(const CandGenArgs g)
{
    const uint64_t total = (uint64_t)g.n * g.sketch_size;
    for (uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; x < total; x += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t b = (uint32_t)(x % g.sketch_size);
        const uint32_t i = (uint32_t)(x / g.sketch_size);
        const uint32_t pos = atomicAdd(&g.cursor[(uint64_t)b * 65536u + g.skq[x]], 1u);
        g.members[(uint64_t)b * g.n + pos] = i;
    }
}

// After the scatter cursor[b][v] is the END of the group that starts at starts[b][v].
//
// Round 4.  Round 3's form walked a row's bins one after the other, each a chain of three dependent loads (the row's
// value -> the group's bounds -> its members, 64 four-byte loads per wave and trip): 82-86 % of the wave-cycles on
// s_waitcnt, 0.18 of HBM's rate for the 78 KB of member lists a row of the benchmark reads.  Now:
//   A. every bin's group bounds are looked up at once, one THREAD per bin (two dependent loads for the whole row), and
//      parked in LDS;
//   B. a wave takes U_BINS bins per trip and reads each group with ONE 16-byte load per lane (256 members per wave
//      and load, aligned down to 16 bytes and masked by the group's bounds; longer groups loop) -- U_BINS independent
//      loads in flight per lane before the first bitmap update of the trip;
//   C. the per-thread counts are scanned with wave shuffles instead of a serial loop of thread 0.
constexpr int CG_U_BINS = 4;
constexpr uint32_t CG_BIN_BLOCK = 192;   // bins whose bounds are parked at a time (1.5 KB of LDS beside a bitmap of up to 158 KB)

template <bool FILL>
__global__ __launch_bounds__(CG_THREADS) void cand_rows_kernel(const CandGenArgs g)
{
    extern __shared__ uint32_t cg_lds[];   // [2 * CG_BIN_BLOCK] group bounds (later the waves' totals), then the bitmap: ceil(n / 32) words
    const uint32_t i = blockIdx.x;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t S = g.sketch_size;
    uint32_t *gb = cg_lds, *ge = cg_lds + CG_BIN_BLOCK;
    uint32_t *wave_tot = cg_lds;
    uint32_t *bitmap = cg_lds + 2u * CG_BIN_BLOCK;
    const uint32_t n_words = (g.n + 31u) / 32u;
    const uint32_t n_quads = (n_words + 3u) / 4u;
    for (uint32_t q = tid; q < n_quads; q += CG_THREADS) reinterpret_cast<uint4 *>(bitmap)[q] = make_uint4(0u, 0u, 0u, 0u);
    const uint16_t *sig = g.skq + (uint64_t)i * S;
    constexpr uint32_t W = CG_THREADS / 64, U = CG_U_BINS;
    for (uint32_t blk = 0; blk < S; blk += CG_BIN_BLOCK) {
    const uint32_t SB = min(CG_BIN_BLOCK, S - blk);   // bins of this block
    if (blk) __syncthreads();                          // the previous block's bounds have been used
    // A: the bounds of this row's group in every bin of the block
    for (uint32_t b = tid; b < SB; b += CG_THREADS) {
        const uint64_t slot = (uint64_t)(blk + b) * 65536u + sig[blk + b];
        gb[b] = g.starts[slot];
        ge[b] = g.cursor[slot];
    }
    __syncthreads();
    // B: OR the members of every group into the bitmap
    for (uint32_t b0 = wave; b0 < SB; b0 += W * U) {
        uint4 v[U];
        uint64_t e0[U];
        uint32_t lo[U], hi[U];
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) {
            const uint32_t b = b0 + u * W;
            v[u] = make_uint4(0u, 0u, 0u, 0u);
            lo[u] = hi[u] = 0u;
            e0[u] = 0ull;
            if (b < SB) {   // (wave-uniform)
                const uint64_t base = (uint64_t)(blk + b) * g.n;
                const uint64_t first = (base + gb[b]) & ~3ull;     // element index of the 16-byte block the group starts in
                e0[u] = first + 4ull * lane;                        // this lane's four elements
                lo[u] = gb[b];
                hi[u] = ge[b];
                if (e0[u] < base + hi[u]) v[u] = *reinterpret_cast<const uint4 *>(g.members + e0[u]);
            }
        }
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) {
            const uint32_t b = b0 + u * W;
            if (b >= SB) continue;
            const uint64_t base = (uint64_t)(blk + b) * g.n;
            const uint64_t begin = base + lo[u], end = base + hi[u];
            uint64_t e = e0[u];
            uint4 x = v[u];
            for (;;) {
                const uint32_t xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
                for (uint32_t c = 0; c < 4; ++c) {
                    if (e + c >= begin && e + c < end) atomicOr(&bitmap[xs[c] >> 5], 1u << (xs[c] & 31u));
                }
                e += 256u;   // (groups of more than ~256 members: further trips of this wave)
                if (e - 4ull * lane >= end) break;   // (wave-uniform: the trip's first element)
                x = e < end ? *reinterpret_cast<const uint4 *>(g.members + e) : make_uint4(0u, 0u, 0u, 0u);
            }
        }
    }
    }
    __syncthreads();
    if (tid == 0) bitmap[i >> 5] &= ~(1u << (i & 31u));   // a sample is not its own candidate (mod.rs:458-461)
    __syncthreads();
    // C: contiguous word range per thread, so that the output is ascending; exclusive scan of the per-thread counts
    const uint32_t per = (n_words + CG_THREADS - 1) / CG_THREADS;
    const uint32_t w0 = tid * per, w1 = (w0 + per < n_words) ? w0 + per : n_words;
    uint32_t cnt = 0;
    for (uint32_t w = w0; w < w1; ++w) cnt += __popc(bitmap[w]);
    uint32_t incl = cnt;
#pragma unroll
    for (uint32_t d = 1; d < 64u; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d);
        if (lane >= d) incl += up;
    }
    if (lane == 63u) wave_tot[wave] = incl;
    __syncthreads();
    uint32_t before = incl - cnt;
    uint32_t total = 0;
#pragma unroll
    for (uint32_t w = 0; w < W; ++w) {
        if (w < wave) before += wave_tot[w];
        total += wave_tot[w];
    }
    if constexpr (!FILL) {
        if (tid == 0) g.counts[i] = total;
    } else {
        uint32_t *out = g.cand + g.row_offsets[i] + before;
        for (uint32_t w = w0; w < w1; ++w) {
            uint32_t bits = bitmap[w];
            while (bits) {
                const uint32_t bit = __ffs(bits) - 1u;
                *out++ = w * 32u + bit;
                bits &= bits - 1u;
            }
        }
    }
}

hipError_t launch_cand_groups(const CandGenArgs &g, hipStream_t stream)
{
    const uint64_t total = (uint64_t)g.n * g.sketch_size;
    if (total == 0) return hipSuccess;
    const unsigned blocks = (unsigned)std::min<uint64_t>((total + CG_THREADS - 1) / CG_THREADS, 1u << 16);
    hipLaunchKernelGGL(bin_hist_kernel, dim3(blocks), dim3(CG_THREADS), 0, stream, g);
    hipLaunchKernelGGL(bin_scan_kernel, dim3(g.sketch_size), dim3(CG_THREADS), 0, stream, g);
    hipLaunchKernelGGL(bin_scatter_kernel, dim3(blocks), dim3(CG_THREADS), 0, stream, g);
    return hipGetLastError();
}

// first position of a candidate id greater than the row's own id in each (ascending) list
__global__ void first_greater_kernel(const uint64_t *row_offsets, const uint32_t *cand, uint32_t n, uint64_t *first)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t lo = row_offsets[i], hi = row_offsets[i + 1];
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (cand[mid] <= i) lo = mid + 1; else hi = mid;
    }
    first[i] = lo;
}

hipError_t launch_first_greater(const uint64_t *row_offsets, const uint32_t *cand, uint32_t n, uint64_t *first,
                                hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(first_greater_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, row_offsets, cand, n, first);
    return hipGetLastError();
}

// key[i] = the first candidate of row i (0xFFFFFFFF for an empty row): rows whose lists start with the same sample are,
// as a rule, members of one cluster with nearly the same list -- the launch order of pair_cand_kernel's work items
__global__ void first_candidate_kernel(const uint64_t *row_offsets, const uint32_t *cand, uint32_t n, uint32_t *key)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    key[i] = row_offsets[i + 1] > row_offsets[i] ? cand[row_offsets[i]] : 0xFFFFFFFFu;
}

hipError_t launch_first_candidate(const uint64_t *row_offsets, const uint32_t *cand, uint32_t n, uint32_t *key, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(first_candidate_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, row_offsets, cand, n, key);
    return hipGetLastError();
}

// The work items of pair_cand_kernel, written where the host counted them: the x-th row of the launch order (row order[x])
// owns items item_begin[x] .. item_begin[x + 1], one per 64 candidates from `first[row]` (symmetric lists: its first
// candidate with a larger id; null: the start of its list).
__global__ void cand_work_items_kernel(const uint32_t *order, const uint64_t *item_begin, const uint64_t *row_offsets,
                                       const uint64_t *first, uint32_t n, uint32_t *work_row, uint64_t *work_start)
{
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= n) return;
    const uint32_t row = order[x];
    const uint64_t end = row_offsets[row + 1];
    uint64_t at = item_begin[x];
    for (uint64_t c0 = first ? first[row] : row_offsets[row]; c0 < end; c0 += 64, ++at) {
        work_row[at] = row;
        work_start[at] = c0;
    }
}

hipError_t launch_cand_work_items(const uint32_t *order, const uint64_t *item_begin, const uint64_t *row_offsets, const uint64_t *first,
                                  uint32_t n, uint32_t *work_row, uint64_t *work_start, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(cand_work_items_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, order, item_begin, row_offsets, first, n,
                       work_row, work_start);
    return hipGetLastError();
}

hipError_t launch_cand_rows(const CandGenArgs &g, bool fill, hipStream_t stream)
{
    if (g.n == 0) return hipSuccess;
    // the group bounds of a block of bins + the bitmap rounded up to whole 16-byte blocks
    const size_t lds = ((size_t)2u * CG_BIN_BLOCK + (size_t)(((g.n + 31u) / 32u + 3u) & ~3u)) * sizeof(uint32_t);
    if (lds > 158 * 1024 + 2 * CG_BIN_BLOCK * sizeof(uint32_t)) return hipErrorInvalidValue;
    if (fill) {
        (void)hipFuncSetAttribute((const void *)cand_rows_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(cand_rows_kernel<true>, dim3(g.n), dim3(CG_THREADS), lds, stream, g);
    } else {
        (void)hipFuncSetAttribute((const void *)cand_rows_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(cand_rows_kernel<false>, dim3(g.n), dim3(CG_THREADS), lds, stream, g);
    }
    return hipGetLastError();
}

}  // namespace skl
