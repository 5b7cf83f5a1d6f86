// kernels.hip -- gfx950 (CDNA4 / MI355X) device code of the pairwise-distance path.
//
// What is computed (reference: bacpop/sketchlib.rust v0.3.0):
//   * bin-match counts            src/distances/jaccard.rs:15-25
//   * Jaccard / ANI per pair       src/distances/jaccard.rs:26-51, mod.rs:83-100
//   * core/accessory regression    src/distances/jaccard.rs:61-142
//   * per-row k nearest neighbours src/distances/mod.rs:41-48,133-224,306-395
//
// How it is laid out for the machine (see DESIGN.md "Kernels"):
//   A sketch is bit-sliced: 14 u64 planes per 64-bin chunk.  A bin matches iff all
//   14 planes agree, so per (pair, k, chunk) the work is
//       m = OR_p (a_p ^ b_p);  matches += 64 - popcount(m)
//   i.e. pure 32-bit integer VALU work (no MFMA shape exists for it).
//
//   One wavefront owns NA "row" samples x 64 "column" samples:
//     * the column sample lives in the lane: its 14 planes of the current chunk are
//       28 VGPRs, loaded with 7 fully coalesced global_load_dwordx4 per lane from a
//       lane-interleaved copy of the slab (1 KiB contiguous per wave-instruction);
//     * the row sample is wave-uniform: its 28 dwords arrive through the scalar
//       cache with s_load_dwordx16/x8/x4 into SGPRs and feed the VALU as the scalar
//       operand -- the hardware broadcast path, no LDS traffic and no cross-lane
//       reduction at all;
//     * m |= a ^ b is a single v_bitop3_b32 (gfx950 3-input LUT op), so a chunk of
//       one pair costs 28 bitop3 + 2 v_bcnt_u32_b32 (popcount with fused accumulate).
//   Counts stay in VGPRs; the Jaccard / regression epilogue runs in the same kernel,
//   every lane finishing its own pairs.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (the regression must not
// be contracted into FMAs: the reference's f64 arithmetic is unfused).
#include "device_common.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace skl {

// ---------------------------------------------------------------------------
// helpers
// ---------------------------------------------------------------------------

struct A28 {
    uint32_t w[28];
};
// Constant address space => the compiler emits scalar (SMEM) loads for uniform
// addresses; the slab is read-only for the lifetime of the launch.
typedef const __attribute__((address_space(4))) uint32_t *const_u32_ptr;

__device__ __forceinline__ A28 load_row_chunk(const uint64_t *p)
{
    const_u32_ptr q = (const_u32_ptr)(uintptr_t)p;
    A28 r;
#pragma unroll
    for (int x = 0; x < 28; ++x) r.w[x] = q[x];
    return r;
}

// ---------------------------------------------------------------------------
// the pair kernel
// ---------------------------------------------------------------------------

template <int NA, int MODE, bool BITOP3>
__global__ __launch_bounds__(LANES *WAVES_PER_WG) void pair_kernel(const PairArgs g)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    // XCD-aware tile mapping.  Workgroups are dealt round-robin over the 8 XCDs, so
    // blockIdx % 8 labels the XCD; consecutive workgroups of one XCD walk the row
    // tiles of ONE column block, whose lane-operand slab then stays in that XCD's L2.
    // Column blocks are dealt to XCDs modulo 8, which also balances the triangle.
    // The 4 waves of a workgroup share the column block and take 4 adjacent row tiles.
    const uint32_t xcd = blockIdx.x & 7u;
    const uint32_t slot = blockIdx.x >> 3;
    const uint32_t gseq = slot / g.a_tiles;
    const uint32_t jb = gseq * 8u + ((gseq & 1u) ? 7u - xcd : xcd);  // boustrophedon deal
    const uint32_t at = slot % g.a_tiles;
    const uint32_t a0 = g.row_begin + (at * WAVES_PER_WG + wave) * NA;
    if (jb >= g.n_jblocks) return;
    if (a0 >= g.row_end) return;
    if (g.self_mode && a0 >= jb * 64u + 63u) return;  // no i < j in this wave tile
    const uint32_t jcol = jb * 64u + lane;

    const size_t kmer_stride = (size_t)g.ss64 * BBITS;
    const size_t sample_stride = kmer_stride * g.nk;
    const uint64_t *abase = g.A + (size_t)a0 * sample_stride;
    const uint32_t last_chunk = g.ss64 - 1u;

    // Per pair, the mismatch counts.  MODE_COREACC keeps up to 6 k-mer lengths as a 96-bit
    // shift register of u16 fields (st2:st1:st0, newest k in the low field of st0) and
    // accumulates the current k directly into that low field (counts <= 64*ss64 <= 65535
    // on this path, so the popcount adds never carry into the neighbouring field).
    uint32_t st0[NA], st1[NA], st2[NA];
#pragma unroll
    for (int ia = 0; ia < NA; ++ia) {
        st0[ia] = 0;
        st1[ia] = 0;
        st2[ia] = 0;
    }

    for (uint32_t kk = 0; kk < g.k_count; ++kk) {
        const uint32_t k = g.k_begin + kk;
#pragma unroll
        for (int ia = 0; ia < NA; ++ia) {
            if constexpr (MODE == MODE_COREACC) {
                st2[ia] = __builtin_amdgcn_alignbit(st2[ia], st1[ia], 16);
                st1[ia] = __builtin_amdgcn_alignbit(st1[ia], st0[ia], 16);
                st0[ia] = st0[ia] << 16;
            } else {
                st0[ia] = 0;
            }
        }

        const uint4 *bp = g.B + ((size_t)(jb * g.nk + k) * g.ss64) * (7 * LANES) + lane;
        const uint64_t *ap = abase + (size_t)k * kmer_stride;

        uint4 b[7];
#pragma unroll
        for (int q = 0; q < 7; ++q) b[q] = bp[q * LANES];
        A28 a_cur = load_row_chunk(ap);

        for (uint32_t c = 0; c <= last_chunk; ++c) {
            const uint32_t cn = c < last_chunk ? c + 1u : c;
            // next chunk of the lane operand: in flight under this chunk's VALU work
            uint4 bn[7];
            const uint4 *bpn = bp + (size_t)cn * (7 * LANES);
#pragma unroll
            for (int q = 0; q < 7; ++q) bn[q] = bpn[q * LANES];

#pragma unroll
            for (int ia = 0; ia < NA; ++ia) {
                // First touch of this row's SGPRs: the compiler's s_waitcnt lgkmcnt(0)
                // lands here, BEFORE the next row's loads are issued (SMEM returns out of
                // order, so a wait placed after them would wait for them too).
                uint32_t mlo = a_cur.w[0] ^ b[0].y;
                uint32_t mhi = a_cur.w[1] ^ b[0].x;
                __builtin_amdgcn_sched_barrier(0);
                // next row's chunk (or row 0 of the next chunk): in flight under this
                // row's 28 VALU ops
                const A28 a_nxt = (ia + 1 < NA)
                                      ? load_row_chunk(ap + (size_t)(ia + 1) * sample_stride +
                                                       (size_t)c * BBITS)
                                      : load_row_chunk(ap + (size_t)cn * BBITS);
                __builtin_amdgcn_sched_barrier(0);
                mlo = acc_mismatch<BITOP3>(mlo, a_cur.w[2], b[0].w);
                mhi = acc_mismatch<BITOP3>(mhi, a_cur.w[3], b[0].z);
#pragma unroll
                for (int q = 1; q < 7; ++q) {
                    mlo = acc_mismatch<BITOP3>(mlo, a_cur.w[4 * q + 0], b[q].y);
                    mhi = acc_mismatch<BITOP3>(mhi, a_cur.w[4 * q + 1], b[q].x);
                    mlo = acc_mismatch<BITOP3>(mlo, a_cur.w[4 * q + 2], b[q].w);
                    mhi = acc_mismatch<BITOP3>(mhi, a_cur.w[4 * q + 3], b[q].z);
                }
                st0[ia] += __builtin_popcount(mlo);
                st0[ia] += __builtin_popcount(mhi);
                a_cur = a_nxt;
            }
#pragma unroll
            for (int q = 0; q < 7; ++q) b[q] = bn[q];
        }

        // ---- per-k epilogue ----
        if constexpr (MODE == MODE_COUNTS) {
#pragma unroll
            for (int ia = 0; ia < NA; ++ia) store_count(g, a0 + ia, jcol, kk, st0[ia]);
        } else if constexpr (MODE == MODE_JACCARD) {
#pragma unroll
            for (int ia = 0; ia < NA; ++ia) store_jaccard(g, a0 + ia, jcol, st0[ia]);
        }
    }

    // ---- fused core/accessory regression (jaccard.rs:61-142) ----
    if constexpr (MODE == MODE_COREACC) {
        // One body for all NA rows: take slot 0, then rotate the register window.
#pragma clang loop unroll(disable)
        for (int r = 0; r < NA; ++r) {
            store_coreacc(g, a0 + (uint32_t)r, jcol, st0[0], st1[0], st2[0]);
#pragma unroll
            for (int x = 0; x + 1 < NA; ++x) {
                st0[x] = st0[x + 1];
                st1[x] = st1[x + 1];
                st2[x] = st2[x + 1];
            }
        }
    }
}

// ---------------------------------------------------------------------------
// launch
// ---------------------------------------------------------------------------

int choose_na(uint64_t n_rows, uint64_t n_cols, int self_mode, int mode)
{
    (void)mode;
    static const int forced = [] {
        const char *e = getenv("SKL_FORCE_NA");  // tuning knob: 2, 4 or 8
        return e ? atoi(e) : 0;
    }();
    if (forced == 2 || forced == 4 || forced == 8) return forced;
    const uint64_t pairs = self_mode ? n_rows * n_cols / 2 : n_rows * n_cols;
    // this kernel serves small launches: aim for >= 2 waves on each of the 1024 SIMDs
    // (measured on MI355X: NA = 4 beats 2 and 8 from n = 1000 to n = 3000, sweep5.log)
    if (pairs / (4ull * 64ull) >= 1024) return 4;
    return 2;
}

template <int NA, int MODE>
static hipError_t launch_t(const PairArgs &args, bool bitop3, dim3 grid, hipStream_t stream)
{
    if (bitop3) {
        hipLaunchKernelGGL((pair_kernel<NA, MODE, true>), grid, dim3(LANES * WAVES_PER_WG), 0,
                           stream, args);
    } else {
        hipLaunchKernelGGL((pair_kernel<NA, MODE, false>), grid, dim3(LANES * WAVES_PER_WG), 0,
                           stream, args);
    }
    return hipGetLastError();
}

template <int MODE>
static hipError_t launch_m(const PairArgs &args, int na, bool bitop3, dim3 grid,
                           hipStream_t stream)
{
    switch (na) {
        case 2: return launch_t<2, MODE>(args, bitop3, grid, stream);
        case 4: return launch_t<4, MODE>(args, bitop3, grid, stream);
        case 8: return launch_t<8, MODE>(args, bitop3, grid, stream);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_pair_kernel(const PairArgs &args_in, int mode, int na, hipStream_t stream)
{
    PairArgs args = args_in;
    if (args.row_end <= args.row_begin || args.nB == 0) return hipSuccess;
    const uint32_t rows = args.row_end - args.row_begin;
    args.share_rows = 0;
    args.n_jblocks = (args.nB + 63u) / 64u;
    const uint32_t rows_per_wg = (uint32_t)na * WAVES_PER_WG;
    args.a_tiles = (rows + rows_per_wg - 1) / rows_per_wg;
    const uint64_t n_wg = 8ull * ((args.n_jblocks + 7u) / 8u) * args.a_tiles;
    if (n_wg >= (1ull << 31)) return hipErrorInvalidValue;
    static const bool bitop3 = [] {
        const char *e = getenv("SKL_PAIR_VARIANT");
        return !(e && strcmp(e, "or3") == 0);
    }();
    const dim3 grid((unsigned)n_wg);
    switch (mode) {
        case MODE_COUNTS: return launch_m<MODE_COUNTS>(args, na, bitop3, grid, stream);
        case MODE_JACCARD: return launch_m<MODE_JACCARD>(args, na, bitop3, grid, stream);
        case MODE_COREACC: return launch_m<MODE_COREACC>(args, na, bitop3, grid, stream);
        default: return hipErrorInvalidValue;
    }
}

// ---------------------------------------------------------------------------
// balanced tile enumeration (host side of device_common.hpp::lookup_tile)
// ---------------------------------------------------------------------------

hipError_t plan_tiles(PairArgs &args, uint32_t rows_per_tile, uint32_t cols_per_group,
                      TileScratch &scratch, hipStream_t stream, uint64_t *grid_out)
{
    *grid_out = 0;
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
        std::vector<uint32_t> prefix(args.n_groups + 1);
        total = 0;
        for (uint32_t gi = 0; gi < args.n_groups; ++gi) {
            prefix[gi] = (uint32_t)total;
            const uint64_t last_col = (uint64_t)(gi + 1) * cols_per_group - 1;
            const uint32_t lim = (uint32_t)std::min<uint64_t>(args.row_end, last_col);
            total += lim > args.row_begin ? (lim - args.row_begin + rows_per_tile - 1) / rows_per_tile : 0u;
        }
        prefix[args.n_groups] = (uint32_t)total;
        if (total >= (1ull << 32)) return hipErrorInvalidValue;
        const uint64_t key[4] = {((uint64_t)args.row_begin << 32) | args.row_end, args.nB,
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
    }
    if (total == 0) return hipSuccess;
    if (total >= (1ull << 31)) return hipErrorInvalidValue;
    args.n_active_tiles = (uint32_t)total;
    args.tiles_per_xcd = (uint32_t)((total + 7) / 8);
    *grid_out = 8ull * args.tiles_per_xcd;
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

__global__ __launch_bounds__(256) void coreacc_epilogue_kernel(const EpilogueArgs g)
{
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= g.n_pairs) return;
    double c1 = 0.0, c2 = 0.0;
    if (g.has_comp) {
        // recover (i, j) from the flat index of this launch
        const uint64_t flat = p + g.out_base;
        uint64_t i, j;
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
        c1 = g.compA[i];
        c2 = g.compB[j];
    }
    const uint32_t maxnbits = g.ss64 * 64u;
    double xsum = 0.0, ysum = 0.0, xysum = 0.0, xsquaresum = 0.0, ysquaresum = 0.0, n = 0.0;
    const uint32_t *cnt = g.counts + p * g.pair_stride;
    for (uint32_t t = 0; t < g.nk; ++t) {
        const uint32_t same = cnt[t * g.k_stride];
        double y;
        if (!g.has_comp) {
            y = g.ytab[same <= maxnbits ? same : maxnbits];
        } else {
            y = glibc_log(jaccard_from_samebits_dev(same, g.ss64, true, c1, c2, g.cutoff), g.log_variant);
        }
        if (y < g.tolerance) break;
        const double k_fl = g.kf[t];
        xsum += k_fl;
        ysum += y;
        xysum += k_fl * y;
        xsquaresum += k_fl * k_fl;
        ysquaresum += y * y;
        n += 1.0;
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
