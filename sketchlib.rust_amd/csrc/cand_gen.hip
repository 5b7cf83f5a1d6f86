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
template <bool FILL>
__global__ __launch_bounds__(CG_THREADS) void cand_rows_kernel(const CandGenArgs g)
{
    extern __shared__ uint32_t bitmap[];   // ceil(n / 32) words
    __shared__ uint32_t part[CG_THREADS];
    const uint32_t i = blockIdx.x;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t n_words = (g.n + 31u) / 32u;
    for (uint32_t w = tid; w < n_words; w += CG_THREADS) bitmap[w] = 0;
    __syncthreads();
    const uint16_t *sig = g.skq + (uint64_t)i * g.sketch_size;
    for (uint32_t b = wave; b < g.sketch_size; b += CG_THREADS / 64) {   // one wave per bin
        const uint64_t slot = (uint64_t)b * 65536u + sig[b];
        const uint32_t begin = g.starts[slot], end = g.cursor[slot];
        const uint32_t *mem = g.members + (uint64_t)b * g.n;
        for (uint32_t m = begin + lane; m < end; m += 64) {
            const uint32_t j = mem[m];
            atomicOr(&bitmap[j >> 5], 1u << (j & 31u));
        }
    }
    __syncthreads();
    if (tid == 0) bitmap[i >> 5] &= ~(1u << (i & 31u));   // a sample is not its own candidate (mod.rs:458-461)
    __syncthreads();
    // contiguous word range per thread, so that the output is ascending
    const uint32_t per = (n_words + CG_THREADS - 1) / CG_THREADS;
    const uint32_t w0 = tid * per, w1 = (w0 + per < n_words) ? w0 + per : n_words;
    uint32_t cnt = 0;
    for (uint32_t w = w0; w < w1; ++w) cnt += __popc(bitmap[w]);
    part[tid] = cnt;
    __syncthreads();
    if (tid == 0) {
        uint32_t acc = 0;
        for (int t = 0; t < CG_THREADS; ++t) {
            const uint32_t v = part[t];
            part[t] = acc;
            acc += v;
        }
        if (!FILL) g.counts[i] = acc;
    }
    if constexpr (FILL) {
        __syncthreads();
        uint32_t *out = g.cand + g.row_offsets[i] + part[tid];
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

hipError_t launch_cand_rows(const CandGenArgs &g, bool fill, hipStream_t stream)
{
    if (g.n == 0) return hipSuccess;
    const size_t lds = (size_t)((g.n + 31u) / 32u) * sizeof(uint32_t);
    if (lds > 158 * 1024) return hipErrorInvalidValue;
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
