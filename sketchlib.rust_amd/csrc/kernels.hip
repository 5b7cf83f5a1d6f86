// kernels.hip -- gfx950 (CDNA4 / MI355X) device code shared by the pair kernels of the
// pairwise-distance path: the balanced tile enumeration, the slab re-layout, the unfused
// core/accessory epilogue and the restated libm logarithm as a diagnostic kernel.
//
// What the path computes (reference: bacpop/sketchlib.rust v0.3.0):
//   * bin-match counts            src/distances/jaccard.rs:15-25
//   * Jaccard / ANI per pair       src/distances/jaccard.rs:26-51, mod.rs:83-100
//   * core/accessory regression    src/distances/jaccard.rs:61-142
//   * per-row k nearest neighbours src/distances/mod.rs:41-48,133-224,306-395
//
// A sketch is bit-sliced: 14 u64 planes per 64-bin chunk.  A bin matches iff all 14 planes
// agree, so per (pair, k, chunk) the work is  m = OR_p (a_p ^ b_p);  matches += 64 - popcount(m)
// i.e. pure 32-bit integer VALU work (no MFMA shape exists for it).  The pair kernels are in
// pair_kslice.hip (default), pair_ksplit.hip (fallback for sketches beyond 65 535 bins),
// pair_cand.hip (candidate lists); DESIGN.md "Kernels".
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (the regression must not
// be contracted into FMAs: the reference's f64 arithmetic is unfused).
#include "device_common.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace skl {
// ---------------------------------------------------------------------------
// balanced tile enumeration (host side of device_common.hpp::lookup_tile)
// ---------------------------------------------------------------------------

hipError_t plan_tiles(PairArgs &args, uint32_t rows_per_tile, uint32_t cols_per_group,
                      TileScratch &scratch, hipStream_t stream, uint64_t *grid_out)
{
    *grid_out = 0;
    if (args.group_span == 0) args.group_span = 1;
    args.tile_rows = rows_per_tile;
    args.group_cols = cols_per_group;
    const uint32_t rows = args.row_end - args.row_begin;
    args.a_tiles = (rows + rows_per_tile - 1) / rows_per_tile;
    args.n_jblocks = (args.nB + 63u) / 64u;
    args.n_groups = (args.nB + cols_per_group - 1) / cols_per_group;
    args.tile_prefix = nullptr;
    uint64_t total;
    if (!args.self_mode) {
        total = (uint64_t)args.a_tiles * args.n_groups;
    } else {
        // group g is needed by the row tiles with a0 < (g+1)*W - 1 (some i < j exists)
        const uint32_t n_super = (args.n_groups + args.group_span - 1) / args.group_span;
        std::vector<uint32_t> prefix(n_super + 1);
        total = 0;
        for (uint32_t gi = 0; gi < args.n_groups; ++gi) {
            if (gi % args.group_span == 0) prefix[gi / args.group_span] = (uint32_t)total;
            const uint64_t last_col = (uint64_t)(gi + 1) * cols_per_group - 1;
            const uint32_t lim = (uint32_t)std::min<uint64_t>(args.row_end, last_col);
            total += lim > args.row_begin ? (lim - args.row_begin + rows_per_tile - 1) / rows_per_tile : 0u;
        }
        prefix[n_super] = (uint32_t)total;
        if (total >= (1ull << 32)) return hipErrorInvalidValue;
        const uint64_t key[4] = {((uint64_t)args.row_begin << 32) | args.row_end, ((uint64_t)args.group_span << 32) | args.nB,
                                 ((uint64_t)rows_per_tile << 32) | cols_per_group, total};
        if (prefix.size() > scratch.capacity) {
            if (scratch.d_prefix) (void)hipFree(scratch.d_prefix);          // (synchronises the device)
            if (scratch.h_staging) (void)hipHostFree(scratch.h_staging);
            scratch.d_prefix = nullptr;
            scratch.h_staging = nullptr;
            scratch.capacity = 0;
            const size_t cap = std::max<size_t>(prefix.size(), 4096);
            hipError_t e = hipMalloc((void **)&scratch.d_prefix, cap * sizeof(uint32_t));
            if (e != hipSuccess) return e;
            e = hipHostMalloc((void **)&scratch.h_staging, cap * sizeof(uint32_t), hipHostMallocDefault);
            if (e != hipSuccess) return e;
            if (!scratch.staged) {
                e = hipEventCreateWithFlags(&scratch.staged, hipEventDisableTiming);
                if (e != hipSuccess) return e;
            }
            scratch.capacity = cap;
            scratch.cached_key[0] = ~0ull;
        }
        if (memcmp(key, scratch.cached_key, sizeof key) != 0) {
            // The upload is asynchronous on `stream`: it reads a PINNED staging buffer owned by the
            // context (never a local that dies with this call), which is rewritten only after
            // the previous upload has fired its event; the device table itself is ordered with
            // the kernels that read it by the stream.
            hipError_t e = hipEventSynchronize(scratch.staged);
            if (e != hipSuccess) return e;
            memcpy(scratch.h_staging, prefix.data(), prefix.size() * sizeof(uint32_t));
            e = hipMemcpyAsync(scratch.d_prefix, scratch.h_staging, prefix.size() * sizeof(uint32_t),
                               hipMemcpyHostToDevice, stream);
            if (e != hipSuccess) return e;
            e = hipEventRecord(scratch.staged, stream);
            if (e != hipSuccess) return e;
            memcpy(scratch.cached_key, key, sizeof key);
        }
        args.tile_prefix = scratch.d_prefix;
        args.n_prefix_inline = 0;
        if (args.inline_prefix_ok && prefix.size() <= (size_t)TILE_PREFIX_INLINE) {
            args.n_prefix_inline = (uint32_t)prefix.size();
            for (size_t x = 0; x < prefix.size(); ++x) args.tile_prefix_inline[x] = prefix[x];
        }
    }
    if (total == 0) return hipSuccess;
    if (total >= (1ull << 31)) return hipErrorInvalidValue;
    args.n_active_tiles = (uint32_t)total;
    const uint64_t n_xcd = 1ull << args.xcd_shift;
    args.tiles_per_xcd = (uint32_t)((total + n_xcd - 1) / n_xcd);
    if (args.xcd_interleave) args.tiles_per_xcd = (uint32_t)(((total + 31) / 32 + n_xcd - 1) / n_xcd * 32);   // whole blocks of 32 tiles, dealt in turns
    *grid_out = n_xcd * args.tiles_per_xcd;
    return hipSuccess;
}

// ---------------------------------------------------------------------------
// slab re-layout: reference [sample][k][chunk][plane] -> [jb][k][chunk][q][lane]{2 planes, hi:lo}
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(256) void relayout_kernel(const uint64_t *__restrict__ src,
                                                       uint4 *__restrict__ dst, uint32_t n,
                                                       uint32_t nk, uint32_t ss64,
                                                       uint64_t total)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t o = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += stride) {
        const uint32_t lane = (uint32_t)(o & 63u);
        uint64_t r = o >> 6;
        const uint32_t q = (uint32_t)(r % 7u);
        r /= 7u;
        const uint32_t c = (uint32_t)(r % ss64);
        r /= ss64;
        const uint32_t k = (uint32_t)(r % nk);
        const uint64_t jb = r / nk;
        const uint64_t j = jb * 64u + lane;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (j < n) {
            const uint64_t w = ((j * nk + k) * ss64 + c) * BBITS + 2u * q;
            const uint64_t p0 = src[w], p1 = src[w + 1];
            // (hi, lo) order inside each plane: see "VGPR banks" in pair_lds.hip -- the dword
            // that meets row-operand component .x (even register) sits in .y (odd register)
            v = make_uint4((uint32_t)(p0 >> 32), (uint32_t)p0, (uint32_t)(p1 >> 32), (uint32_t)p1);
        }
        dst[o] = v;
    }
}

hipError_t launch_relayout(const uint64_t *ref_layout, uint4 *lane_layout, uint32_t n, uint32_t nk,
                           uint32_t ss64, hipStream_t stream)
{
    const uint64_t n_jb = (n + 63u) / 64u;
    const uint64_t total = n_jb * nk * ss64 * 7ull * 64ull;
    if (total == 0) return hipSuccess;
    uint64_t blocks = (total + 255) / 256;
    if (blocks > 256ull * 32ull) blocks = 256ull * 32ull;
    hipLaunchKernelGGL(relayout_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, ref_layout,
                       lane_layout, n, nk, ss64, total);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// unfused core/acc epilogue (any number of k-mer lengths / any sketch size)
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void coreacc_epilogue_kernel(const EpilogueArgs g)
{
    const uint64_t p_raw = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool early = g.nk_total > g.nk;     // (early break: every lane of a wave stays, the completion below is cooperative)
    if (p_raw >= g.n_pairs && !early) return;
    const bool in_range = p_raw < g.n_pairs;
    const uint64_t p = in_range ? p_raw : g.n_pairs - 1;   // (lanes past the end shadow the last pair and store nothing)
    double c1 = 0.0, c2 = 0.0;
    // (i, j) of this thread's pair from the flat index of this launch
    auto pair_of = [&](uint64_t &i, uint64_t &j) {
        const uint64_t flat = p + g.out_base;
        if (g.self_mode) {
            // invert the condensed index by a row search (rows are short ranges; the
            // closed form of distance_matrix.rs:46-51 needs an f64 sqrt and a fix-up)
            const double nn = (double)g.n_total;
            double guess = nn - 2.0 -
                           floor(sqrt(-8.0 * (double)flat + 4.0 * nn * (nn - 1.0) - 7.0) / 2.0 -
                                 0.5);
            i = (uint64_t)(guess < 0.0 ? 0.0 : guess);
            if (i > g.n_total - 2ull) i = g.n_total - 2ull;
            while (i > 0 && square_to_condensed_dev(i, i + 1, g.n_total) > flat) --i;
            while (i + 2 < g.n_total && square_to_condensed_dev(i + 1, i + 2, g.n_total) <= flat)
                ++i;
            j = flat - square_to_condensed_dev(i, i + 1, g.n_total) + i + 1;
        } else {
            i = flat / g.nB_cols;
            j = flat % g.nB_cols;
        }
    };
    if (g.has_comp) {
        uint64_t i, j;
        pair_of(i, j);
        c1 = g.compA[i];
        c2 = g.compB[j];
    }
    const uint32_t maxnbits = g.ss64 * 64u;
    uint32_t *cnt = g.counts + p * g.pair_stride;
    if (g.jaccard_out) {
        // one k-mer length, counted in chunk slices: the value pair_kslice.hip's MODE_JACCARD stores
        // (device_common.hpp jaccard_out_value; mod.rs:83-100, :173-176)
        uint32_t same = cnt[0];
        for (uint32_t sl = 1; sl < g.n_slices; ++sl) same += cnt[(uint64_t)sl * g.k_stride];
        if (g.rezero_plane1) cnt[g.k_stride] = 0u;
        if (same > maxnbits) same = maxnbits;
        float v;
        if (!g.has_comp) {
            v = g.dtab[same];
        } else {
            const double jac = jaccard_from_samebits_dev(same, g.ss64, true, c1, c2, g.cutoff);
            if (g.jout == JOUT_DIST) v = (float)(1.0 - jac);
            else if (g.jout == JOUT_ANI) v = (float)ani_pois_dev(jac, g.kf0, g.log_variant);
            else v = (float)(1.0 - ani_pois_dev(jac, g.kf0, g.log_variant));
        }
        g.out[p] = v;
        return;
    }
    double xsum = 0.0, ysum = 0.0, xysum = 0.0, xsquaresum = 0.0, ysquaresum = 0.0, n = 0.0;
    // The reference's loop leaves at the first k-mer length whose ln J is below the tolerance
    // (jaccard.rs:89-91), so a literal loop is a chain of 2 nk dependent loads (count, then table
    // entry) -- 10 L2 round trips at cfg 2, which is what this kernel's 10 us were.  The counts of up
    // to KB k-mer lengths and their table entries are therefore loaded up front, independent of each
    // other; the sums then stop at the same k-mer length as before.
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
                same[u] = cnt[t * g.k_stride];
                for (uint32_t sl = 1; sl < g.n_slices; ++sl) same[u] += cnt[((uint64_t)sl * g.nk + t) * g.k_stride];
                // plane 1 goes back to zero for the next tail-sliced launch, whether or not k index t is used
                if (g.rezero_plane1 && in_range) cnt[((uint64_t)g.nk + t) * g.k_stride] = 0u;   // (a lane that shadows the last pair must not clear what that pair's own lane has yet to read)
            }
        }
        if (!g.has_comp) {
#pragma unroll
            for (uint32_t u = 0; u < KB; ++u) yt[u] = g.ytab[same[u] <= maxnbits ? same[u] : maxnbits];
        }
#pragma unroll
        for (uint32_t u = 0; u < KB; ++u) {
            const uint32_t t = t0 + u;
            if (t >= g.nk || stopped) continue;
            const double y = g.has_comp ? glibc_log(jaccard_from_samebits_dev(same[u], g.ss64, true, c1, c2, g.cutoff), g.log_variant)
                                        : yt[u];
            if (y < g.tolerance) {   // jaccard.rs:89-91: break
                stopped = true;
                continue;
            }
            const double k_fl = g.kf[t];
            xsum += k_fl;
            ysum += y;
            xysum += k_fl * y;
            xsquaresum += k_fl * k_fl;
            ysquaresum += y * y;
            n += 1.0;
        }
    }
    if (early) {
        // the pairs of this wave that are still in the running (a shared bin at each of the first g.nk lengths), one after the
        // other: all 64 lanes count the bins the pair shares at the next length -- lane c the chunks c, c + 64, ... of the two
        // samples' slices in the reference's layout -- until a length without one (the reference's break) or the last
        const bool alive = in_range && !stopped;
        uint64_t i_mine = 0, j_mine = 0;
        if (alive) pair_of(i_mine, j_mine);
        uint64_t todo = __ballot(alive);
        // (counted in one of 1 024 slots: a million adds to ONE address per launch queue up behind each other -- 10 of this
        // kernel's 12 ms at n = 16 000, the lesson of the kNN's pruning counters once more)
        if (g.alive_count != nullptr && todo != 0ull && (threadIdx.x & 63u) == 0u) atomicAdd(&g.alive_count[blockIdx.x & 1023u], (uint32_t)__popcll(todo));
        const uint32_t lane = threadIdx.x & 63u;
        while (todo != 0ull) {
            const int l = __builtin_ctzll(todo);
            todo &= todo - 1ull;
            const uint64_t i_l = __shfl(i_mine, l), j_l = __shfl(j_mine, l);
            for (uint32_t t = g.nk; t < g.nk_total; ++t) {
                const uint64_t *a = g.rows_ref + ((i_l * g.nk_total + t) * g.ss64) * BBITS;
                const uint64_t *b = g.cols_ref + ((j_l * g.nk_total + t) * g.ss64) * BBITS;
                uint32_t m = 0u;
                for (uint32_t c = lane; c < g.ss64; c += 64u) {
                    uint64_t differ = 0ull;
#pragma unroll
                    for (uint32_t pl = 0; pl < (uint32_t)BBITS; ++pl) differ |= a[(uint64_t)c * BBITS + pl] ^ b[(uint64_t)c * BBITS + pl];
                    m += (uint32_t)__popcll(~differ);
                }
#pragma unroll
                for (int sh = 32; sh >= 1; sh >>= 1) m += (uint32_t)__shfl_xor((int)m, sh);
                const double y = g.ytab[m <= maxnbits ? m : maxnbits];
                if (y < g.tolerance) break;   // (wave-uniform: m is) jaccard.rs:89-91
                if ((int)lane == l) {
                    const double k_fl = g.kf[t];
                    xsum += k_fl;
                    ysum += y;
                    xysum += k_fl * y;
                    xsquaresum += k_fl * k_fl;
                    ysquaresum += y * y;
                    n += 1.0;
                }
            }
        }
        if (!in_range) return;
    }
    ((float2 *)g.out)[p] =
        simple_linear_regression_dev(xsum, ysum, xysum, xsquaresum, ysquaresum, n);
}

// skl_device_log: the restated libm logarithm as the kernels evaluate it (diagnostic / tests)
__global__ void device_log_kernel(const double *x, double *y, uint64_t n, int variant)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = glibc_log(x[i], variant);
}

hipError_t launch_device_log(const double *x, double *y, uint64_t n, int variant, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    const uint64_t blocks = (n + 255) / 256;
    if (blocks >= (1ull << 31)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(device_log_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, y, n, variant);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// shader clock under load (diagnostic): ONE wave that stamps the shader-clock counter (s_memtime)
// and the constant 100 MHz counter (s_memrealtime) every `sleeps` x s_sleep 127 (~4 us each) until
// the host raises *stop (pinned host memory) or max_samples are taken.  Launched on its own stream
// next to the pair kernels, it reads the clock the chip holds WHILE they run; the ratio of the two
// counters' increments is the clock in units of 100 MHz.  It occupies one wave slot of one CU.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void clock_sampler_kernel(const uint32_t *stop, uint64_t *samples, uint32_t max_samples,
                                                           uint32_t sleeps, uint32_t *count)
{
    if (threadIdx.x != 0) return;
    uint32_t i = 0;
    while (i < max_samples) {
        samples[2 * i] = __builtin_amdgcn_s_memtime();
        samples[2 * i + 1] = __builtin_amdgcn_s_memrealtime();
        ++i;
        if (__hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) break;
        for (uint32_t s = 0; s < sleeps; ++s) __builtin_amdgcn_s_sleep(127);
    }
    *count = i;
}

hipError_t launch_clock_sampler(const uint32_t *stop, uint64_t *samples, uint32_t max_samples, uint32_t sleeps,
                                uint32_t *count, hipStream_t stream)
{
    hipLaunchKernelGGL(clock_sampler_kernel, dim3(1), dim3(64), 0, stream, stop, samples, max_samples, sleeps, count);
    return hipGetLastError();
}

hipError_t launch_coreacc_epilogue(const EpilogueArgs &args, hipStream_t stream)
{
    if (args.n_pairs == 0) return hipSuccess;
    const uint64_t blocks = (args.n_pairs + 255) / 256;
    if (blocks >= (1ull << 31)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(coreacc_epilogue_kernel, dim3((unsigned)blocks), dim3(256), 0, stream,
                       args);
    return hipGetLastError();
}

}  // namespace skl
