// pair_smem.hip -- the FIRST design of the pair kernel, kept for A/B runs only (built with -DSKL_AB,
// `make AB=1`; not part of the product library): one wavefront owns NA "row" samples x 64 "column"
// samples, the column sample lives in the lane (28 VGPRs per chunk from the lane slab), the row
// sample is wave-uniform and arrives through the scalar cache (s_load_dwordx16/x8/x4 into SGPRs)
// to feed the VALU as the scalar operand.  It is what exposed that a VALU instruction with an SGPR
// source issues at about two thirds of the all-VGPR rate (scripts/microbench/valu_clock.hip).
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
            for (int ia = 0; ia < NA; ++ia) store_count(g, a0 + ia, jcol, kk, g.ss64 * 64u, st0[ia]);
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

}  // namespace skl
