// device_common.hpp -- device-side helpers shared by the pair kernels: the bin-match
// primitive and the per-pair epilogues (Jaccard / ANI / core-accessory regression),
// each a restatement of the reference function it cites.
#pragma once

#include "glibc_log.hpp"
#include "kernels.h"

namespace skl {

// m | (a ^ b) in one instruction: v_bitop3_b32 with truth table 0xDE for
// (src0 = a, src1 = m, src2 = b).
template <bool BITOP3>
__device__ __forceinline__ uint32_t acc_mismatch(uint32_t m, uint32_t a, uint32_t b)
{
    if constexpr (BITOP3) {
        return __builtin_amdgcn_bitop3_b32(a, m, b, 0xDE);
    } else {
        return m | (a ^ b);
    }
}

// Same function with the operands placed for the register file (all-VGPR form used by the
// LDS kernel): src0 = row dword, src1 = column dword, src2 = accumulator; truth table
// (F0 ^ CC) | AA = 0xBE.  Measured on MI355X (scripts/microbench/vgpr_banks.hip):
// v_bitop3_b32 issues at half rate exactly when src0 and src1 come from the same VGPR bank
// (register index mod 4); src2 and the destination are free.  The column slab stores each
// plane as (hi, lo), so the dword that pairs with the row's even-register component always
// sits in an odd register and vice versa (128-bit VGPR tuples are even-aligned on gfx950):
// the conflict cannot occur, whatever the register allocator does.
__device__ __forceinline__ uint32_t acc_mismatch_vvv(uint32_t m, uint32_t a, uint32_t b)
{
    return __builtin_amdgcn_bitop3_b32(a, b, m, 0xBE);
}

// 16 bytes per lane, global -> LDS, no VGPR destination (global_load_lds_dwordx4): lane l's
// 16 bytes land at LDS byte address m0 + 16*l.  Issued through inline asm ON PURPOSE: the
// compiler treats the builtin form as an LDS write it cannot disambiguate and puts
// `s_waitcnt vmcnt(0)` in front of every later LDS read -- which also waits for the NEXT
// stage's DMA and the column prefetch, i.e. exposes their full latency once per chunk.
// Hidden from its bookkeeping, the ordering is ours: the explicit counted vmcnt wait at the
// top of each stage (VMEM returns in order).  Hidden VMEM ops can only make the compiler's
// own counted waits stricter, never laxer, and the issue points below keep every hidden op
// OLDER than the column loads in flight, so they stay exact.
__device__ __forceinline__ void skl_dma16(const void *src, uint32_t lds_byte_addr)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                 :
                 : "v"(src), "s"(__builtin_amdgcn_readfirstlane(lds_byte_addr))
                 : "memory");   // m0 is reserved (cannot be listed); nothing else in these kernels uses it
#else
    (void)src;
    (void)lds_byte_addr;
#endif
}

// The same with the address split into a wave-uniform base (SGPR pair) and a 32-bit per-lane byte offset:
// the lane offsets of a wave's row pieces never change, so they are computed once (one register per DMA
// instruction) and a stage only supplies the base.
__device__ __forceinline__ void skl_dma16_saddr(const void *uniform_base, uint32_t lane_byte_offset, uint32_t lds_byte_addr)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const uint64_t b = (uint64_t)uniform_base;
    // (the builtin returns int: widen as unsigned, or a low half with bit 31 set smears into the high one)
    const uint64_t sb = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32)) << 32) |
                        (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :
                 : "v"(lane_byte_offset), "s"(sb), "s"(__builtin_amdgcn_readfirstlane(lds_byte_addr))
                 : "memory");
#else
    (void)uniform_base;
    (void)lane_byte_offset;
    (void)lds_byte_addr;
#endif
}

__device__ __forceinline__ uint32_t skl_lds_addr(const void *p)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
#else
    (void)p;
    return 0;
#endif
}

// jaccard.rs:14,26-44 on the device (used when a completeness correction makes the
// host-built tables inapplicable).
__device__ __forceinline__ double jaccard_from_samebits_dev(uint32_t samebits, uint32_t ss64,
                                                            bool has_c, double c1, double c2,
                                                            double cutoff)
{
    const double unionsize = (double)(64u * ss64);
    const uint32_t maxnbits = ss64 * 64u;
    const uint32_t expected = maxnbits >> BBITS;
    const uint32_t diff = samebits > expected ? samebits - expected : 0u;
    const double intersize = ((double)diff * (double)maxnbits) / (double)(maxnbits - expected);
    double j = intersize / unionsize;
    if (has_c) {
        if (c1 * c2 >= cutoff) {
            j = j / (c1 * c2 / (c1 + c2 - c1 * c2));  // jaccard.rs:55-57
            j = fmin(j, 1.0);
        }
    }
    return j;
}

// jaccard.rs:49-51; ln = the host libm's, restated (glibc_log.hpp)
__device__ __forceinline__ double ani_pois_dev(double j, double k, int log_variant)
{
    return fmax(0.0, 1.0 + 1.0 / k * glibc_log((2.0 * j) / (1.0 + j), log_variant));
}

// jaccard.rs:105-142, operation for operation.
__device__ __forceinline__ float2 simple_linear_regression_dev(double xsum, double ysum,
                                                               double xysum, double xsquaresum,
                                                               double ysquaresum, double n)
{
    if (isnan(ysum) || ysum == -INFINITY || n < 3.0) {
        return make_float2(1.0f, 1.0f);
    }
    const double xbar = xsum / n;
    const double ybar = ysum / n;
    const double x_diff = xsquaresum - xsum * xsum / n;
    const double y_diff = ysquaresum - ysum * ysum / n;
    const double xstddev = sqrt((xsquaresum - xsum * xsum / n) / n);
    const double ystddev = sqrt((ysquaresum - ysum * ysum / n) / n);
    const double r = (xysum - xsum * ysum / n) / sqrt(x_diff * y_diff);
    const double beta = r * ystddev / xstddev;
    const double alpha = -beta * xbar + ybar;
    double core = 0.0, acc = 0.0;
    if (beta < 0.0) {
        core = 1.0 - exp(beta);
    } else if (r > 0.0) {
        core = 1.0;
    }
    if (alpha < 0.0) {
        acc = 1.0 - exp(alpha);
    }
    return make_float2((float)core, (float)acc);
}

// distance_matrix.rs:11-14
__device__ __forceinline__ uint64_t square_to_condensed_dev(uint64_t i, uint64_t j, uint64_t n)
{
    return n * i - ((i * (i + 1)) >> 1) + j - 1 - i;
}


// Balanced, XCD-aware tile lookup.  Workgroups are dealt round-robin to the 8 XCDs, so
// Row tiles column group `group` needs in self mode: those with some i < j, i.e. first row below the
// group's last column (the host's plan_tiles counts the same way).
__device__ __forceinline__ uint32_t group_row_tiles(const PairArgs &g, uint32_t group)
{
    const uint64_t last_col = (uint64_t)(group + 1u) * g.group_cols - 1u;
    const uint32_t lim = last_col < g.row_end ? (uint32_t)last_col : g.row_end;
    return lim > g.row_begin ? (lim - g.row_begin + g.tile_rows - 1u) / g.tile_rows : 0u;
}

// Tile u of super-group sg, self mode.  A super-group is group_span consecutive column groups; its
// tiles are numbered row tile by row tile, the groups that need that row tile side by side.  Later
// groups need more row tiles (the triangle), so row tile `at` belongs to the LAST groups of the
// super-group: all of them up to the first group's count, one fewer up to the second's, ...
__device__ __forceinline__ void tile_in_supergroup_self(const PairArgs &g, uint32_t sg, uint32_t u, uint32_t &group,
                                                        uint32_t &row_tile)
{
    const uint32_t first = sg * g.group_span;
    const uint32_t gcount = min(g.group_span, g.n_groups - first);
    uint32_t lo_at = 0;
    for (uint32_t gi = 0; gi + 1u < gcount; ++gi) {
        const uint32_t width = gcount - gi;
        const uint32_t n_gi = group_row_tiles(g, first + gi);
        const uint32_t span = (n_gi - lo_at) * width;
        if (u < span) {
            row_tile = lo_at + u / width;
            group = first + gi + (u - (u / width) * width);
            return;
        }
        u -= span;
        lo_at = n_gi;
    }
    row_tile = lo_at + u;
    group = first + gcount - 1u;
}

// ... cross mode: every group needs all a_tiles row tiles
__device__ __forceinline__ void tile_in_supergroup_cross(const PairArgs &g, uint32_t t, uint32_t &group, uint32_t &row_tile)
{
    const uint32_t per = g.group_span * g.a_tiles;
    const uint32_t sg = t / per, u = t - sg * per;
    const uint32_t first = sg * g.group_span;
    const uint32_t gcount = min(g.group_span, g.n_groups - first);
    row_tile = u / gcount;
    group = first + (u - row_tile * gcount);
}

// f32 -> u32 whose unsigned order is the float order (the running top-k keeps its keys this way)
__device__ __forceinline__ uint32_t sortable_bits(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// blockIdx % (number of XCDs) labels the XCD (MI355X_MICROARCH.md; 8 on an unpartitioned MI355X, PairArgs::xcd_shift);
// XCD x takes tiles
// [x*tiles_per_xcd, (x+1)*tiles_per_xcd) of the super-group-major numbering of the ACTIVE
// tiles: every XCD gets the same number of (equal-cost) tiles.  The ~100 workgroups resident on an
// XCD are consecutive tiles = row tiles x group_span column groups, each row tile shared by
// group_span neighbouring workgroups.  HBM bytes per launch at n = 16 000 with 32 x 128 tiles:
// 44.8 GB numbered group by group, 31.6 GB with group_span = 2 (the default), 32.0 GB with 4; at
// cfg 2 (k-sliced, 16 x 128): 261 / 257 / 321 MB (profiles/r02_tile32_*.md, r02c_*).
// Returns false when this workgroup has no tile.
__device__ __forceinline__ bool lookup_tile_at(const PairArgs &g, uint32_t xcd, uint32_t slot,
                                               uint32_t &group, uint32_t &row_tile)
{
    if (slot >= g.tiles_per_xcd) return false;
    const uint32_t t = g.xcd_interleave ? ((((slot >> 5) << g.xcd_shift) + xcd) << 5) + (slot & 31u) : xcd * g.tiles_per_xcd + slot;
    if (t >= g.n_active_tiles) return false;
    if (!g.self_mode) {
        tile_in_supergroup_cross(g, t, group, row_tile);
        return true;
    }
    const uint32_t n_super = (g.n_groups + g.group_span - 1u) / g.group_span;
    if (g.n_prefix_inline != 0u) {   // the table rides in the kernel arguments: no global load before the first row DMA
        uint32_t lo = 0, base = 0;
#pragma unroll
        for (int x = 1; x < TILE_PREFIX_INLINE; ++x) {
            if ((uint32_t)x < n_super && g.tile_prefix_inline[x] <= t) {
                lo = (uint32_t)x;
                base = g.tile_prefix_inline[x];
            }
        }
        tile_in_supergroup_self(g, lo, t - base, group, row_tile);
        return true;
    }
    uint32_t lo = 0, hi = n_super;  // largest lo with prefix[lo] <= t
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (g.tile_prefix[mid] <= t) lo = mid; else hi = mid;
    }
    tile_in_supergroup_self(g, lo, t - g.tile_prefix[lo], group, row_tile);
    return true;
}

__device__ __forceinline__ bool lookup_tile(const PairArgs &g, uint32_t &group, uint32_t &row_tile)
{
    return lookup_tile_at(g, blockIdx.x & ((1u << g.xcd_shift) - 1u), blockIdx.x >> g.xcd_shift, group, row_tile);
}

__device__ __forceinline__ bool pair_valid(const PairArgs &g, uint32_t i, uint32_t jcol)
{
    return i < g.row_end && jcol < g.nB && (!g.self_mode || i < jcol);
}

__device__ __forceinline__ uint64_t pair_out_index(const PairArgs &g, uint32_t i, uint32_t jcol)
{
    return (g.self_mode ? square_to_condensed_dev(i, jcol, g.nB) : (uint64_t)i * g.nB + jcol) -
           g.out_base;
}

// MODE_COUNTS: samebits of k index kk (jaccard.rs:15-25) over `bins` bins (the whole sketch, or one
// chunk slice of it: pair_kslice.hip, k_slices)
// (U16_OK: the forms whose launches may park u16 records -- PairArgs::cnt_u16 -- carry the run-time choice)
template <bool U16_OK = false>
__device__ __forceinline__ void store_count(const PairArgs &g, uint32_t i, uint32_t jcol,
                                            uint32_t kk, uint32_t bins, uint32_t mismatches)
{
    if (pair_valid(g, i, jcol)) {
        const uint64_t at = pair_out_index(g, i, jcol) * g.cnt_pair_stride + kk * g.cnt_k_stride;
        if (U16_OK && g.cnt_u16) ((uint16_t *)g.out)[at] = (uint16_t)(bins - mismatches);
        else ((uint32_t *)g.out)[at] = bins - mismatches;
    }
}

// The same store write-through at agent scope (global_store_dword ... sc1): the counts a fused epilogue reads back from
// another workgroup (pair_kslice.hip, FUSE)
__device__ __forceinline__ void store_count_agent(const PairArgs &g, uint32_t i, uint32_t jcol,
                                                  uint32_t kk, uint32_t bins, uint32_t mismatches)
{
    if (pair_valid(g, i, jcol)) {
        __hip_atomic_store(&((uint32_t *)g.out)[pair_out_index(g, i, jcol) * g.cnt_pair_stride + kk * g.cnt_k_stride],
                           bins - mismatches, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// MODE_JACCARD: mod.rs:83-100 (dense) / :173-176 (kNN key)
__device__ __forceinline__ float jaccard_out_value(const PairArgs &g, uint32_t i, uint32_t jcol,
                                                   uint32_t mismatches)
{
    const uint32_t same = g.ss64 * 64u - mismatches;
    if (!g.has_comp) return g.dtab[same];
    const double jac =
        jaccard_from_samebits_dev(same, g.ss64, true, g.compA[i], g.compB[jcol], g.cutoff);
    if (g.jout == JOUT_DIST) return (float)(1.0 - jac);
    if (g.jout == JOUT_ANI) return (float)ani_pois_dev(jac, g.kf[0], g.log_variant);
    return (float)(1.0 - ani_pois_dev(jac, g.kf[0], g.log_variant));
}

__device__ __forceinline__ void store_jaccard(const PairArgs &g, uint32_t i, uint32_t jcol,
                                              uint32_t mismatches)
{
    if (!pair_valid(g, i, jcol)) return;
    ((float *)g.out)[pair_out_index(g, i, jcol)] = jaccard_out_value(g, i, jcol, mismatches);
}

// MODE_COREACC: core_acc_dist + simple_linear_regression (jaccard.rs:61-142) from the
// packed per-k mismatch counts (u16 fields, newest k lowest, s2:s1:s0).
__device__ __forceinline__ float2 coreacc_value(const PairArgs &g, uint32_t i, uint32_t jcol,
                                                uint32_t s0, uint32_t s1, uint32_t s2)
{
    const uint32_t maxnbits = g.ss64 * 64u;
    double xsum = 0.0, ysum = 0.0, xysum = 0.0, xsquaresum = 0.0, ysquaresum = 0.0, n = 0.0;
    double c1 = 0.0, c2 = 0.0;
    if (g.has_comp) {
        c1 = g.compA[i];
        c2 = g.compB[jcol];
    }
    bool alive = true;
    for (uint32_t t = 0; t < g.k_count; ++t) {
        const uint32_t f = g.k_count - 1u - t;  // field holding k index t
        const uint32_t word = (f >> 1) == 0u ? s0 : ((f >> 1) == 1u ? s1 : s2);
        const uint32_t same = maxnbits - ((word >> ((f & 1u) * 16u)) & 0xFFFFu);
        double y;
        if (!g.has_comp) {
            y = g.ytab[same];
        } else {
            y = glibc_log(jaccard_from_samebits_dev(same, g.ss64, true, c1, c2, g.cutoff), g.log_variant);
        }
        if (alive) {
            if (y < g.tolerance) {
                alive = false;  // jaccard.rs:89-91: break
            } else {
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
    return simple_linear_regression_dev(xsum, ysum, xysum, xsquaresum, ysquaresum, n);
}

// ... the same from the bin-match counts themselves, same[t] = samebits of k index t (the fused epilogue of a k-sliced
// counts launch): core_acc_dist + simple_linear_regression, jaccard.rs:61-142, operation for operation as above
__device__ __forceinline__ float2 coreacc_value_counts(const PairArgs &g, uint32_t i, uint32_t jcol, const uint32_t *same_k)
{
    const uint32_t maxnbits = g.ss64 * 64u;
    double xsum = 0.0, ysum = 0.0, xysum = 0.0, xsquaresum = 0.0, ysquaresum = 0.0, n = 0.0;
    double c1 = 0.0, c2 = 0.0;
    if (g.has_comp) {
        c1 = g.compA[i];
        c2 = g.compB[jcol];
    }
    bool alive = true;
#pragma unroll
    for (uint32_t t = 0; t < (uint32_t)MAX_FUSED_K; ++t) {
        if (t < g.k_count) {
            const uint32_t same = same_k[t] <= maxnbits ? same_k[t] : maxnbits;
            double y;
            if (!g.has_comp) {
                y = g.ytab[same];
            } else {
                y = glibc_log(jaccard_from_samebits_dev(same, g.ss64, true, c1, c2, g.cutoff), g.log_variant);
            }
            if (alive) {
                if (y < g.tolerance) {
                    alive = false;  // jaccard.rs:89-91: break
                } else {
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
    }
    return simple_linear_regression_dev(xsum, ysum, xysum, xsquaresum, ysquaresum, n);
}

__device__ __forceinline__ void store_coreacc(const PairArgs &g, uint32_t i, uint32_t jcol,
                                              uint32_t s0, uint32_t s1, uint32_t s2)
{
    if (!pair_valid(g, i, jcol)) return;
    ((float2 *)g.out)[pair_out_index(g, i, jcol)] = coreacc_value(g, i, jcol, s0, s1, s2);
}

}  // namespace skl
