// pair_lds.hip -- the LDS-staged pair kernel (gfx950).
//
// Measured on MI355X (scripts/microbench/valu_rates.hip, profiles/): a VALU instruction
// with an SGPR source issues at HALF the rate of the same instruction with VGPR sources
// (v_xor_b32 v,s,v 4.5 clk vs v,v,v 2.7 clk per wave-instruction per SIMD), while
// v_bitop3_b32 with three VGPR sources runs at full rate.  So the row operand must reach
// the VALU from a VGPR, not an SGPR.  The hardware path that puts one value into all 64
// lanes of a VGPR is an LDS read at a wave-uniform address (broadcast, no bank conflict).
//
// Tile: one workgroup (4 waves) = R row samples x (4 * JL * 64) column samples.
//   * rows: the R x 112-byte row chunks of S consecutive chunks are staged into LDS by
//     all 256 threads (global -> registers -> ds_write_b128, double buffered, one
//     barrier per stage); every wave reads them back with 7 uniform-address
//     ds_read_b128 per row = 28 VGPRs holding the row's planes in every lane;
//   * columns: each lane owns JL column samples; their planes for the current chunk are
//     JL x 7 coalesced global_load_dwordx4 (lane-interleaved slab);
//   * per (row, column, chunk): 2 v_xor + 26 v_bitop3 (m |= a ^ b, all-VGPR, full rate)
//     + 2 v_bcnt (popcount fused with the accumulate).
// LDS traffic: 28 LDS cycles per wave-row-step against JL x ~87 VALU cycles, so JL = 2
// keeps the LDS pipe at ~2/3 of the VALU time and the kernel VALU-bound.
// Template flag ADB ("ahead"): rolling one-row prefetch of the row registers (default);
// false = read, wait, compute per row (kept for A/B measurements).
#include "device_common.hpp"

#include <cstdlib>

namespace skl {

// 16 bytes per lane, global -> LDS, no VGPR destination (global_load_lds_dwordx4).  The
// builtin only exists in the device pass; the host pass needs the kernel body to parse.
__device__ __forceinline__ void skl_dma16(const void *src, void *lds_wave_base)
{
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
#else
    (void)src;
    (void)lds_wave_base;
#endif
}

constexpr int STAGE_CHUNKS = 8;  // chunks per LDS stage
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

template <int R, int JL, int MODE, bool ADB, int ORD, int ABL = 0>
__global__ __launch_bounds__(LANES *WAVES_PER_WG) void pair_kernel_lds(const PairArgs g)
{
    constexpr int S = R > 16 ? STAGE_CHUNKS / 2 : STAGE_CHUNKS;
    constexpr int PIECES = R * S * 7;                     // 16-byte pieces per stage
    constexpr int PPT = (PIECES + 255) / 256;             // pieces per thread
    constexpr int P = R * JL;                             // pairs per lane
    static_assert(PPT * 256 - PIECES < PIECES, "tail pieces wrap at most once");
    __shared__ uint4 lds_rows[2][PPT * 256];  // PIECES rounded up: the tail holds duplicates

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    uint32_t jg, at;  // column group (4*JL blocks of 64), row tile
    if (!lookup_tile(g, jg, at)) return;
    const uint32_t jb0 = (jg * WAVES_PER_WG + wave) * JL;  // this wave's first column block
    const uint32_t a0 = g.row_begin + at * R;
    if (jg * WAVES_PER_WG * JL >= g.n_jblocks) return;     // whole workgroup out of range
    if (a0 >= g.row_end) return;
    if (g.self_mode && a0 >= (jg + 1u) * WAVES_PER_WG * JL * 64u - 1u) return;  // below diagonal
    // wave-level skip (the wave still helps staging and joins every barrier)
    const bool active = jb0 < g.n_jblocks && !(g.self_mode && a0 >= (jb0 + JL) * 64u - 1u);

    const size_t kmer_stride = (size_t)g.ss64 * BBITS;
    const size_t sample_stride = kmer_stride * g.nk;
    const uint32_t stages_per_k = (g.ss64 + S - 1) / S;
    const uint32_t n_stages = stages_per_k * g.k_count;

    // Per pair: the live mismatch counter is the low u16 of cnt[]; the previous k-mer
    // length's count sits in its high u16.  Every second k the full word is parked in
    // `cold` -- private (scratch) memory on purpose: it is touched once per two k-mer
    // lengths, and keeping it out of the register file is worth a wave per SIMD.
    uint32_t cnt[P];
    volatile uint32_t cold[(MODE == MODE_COREACC) ? 2 * P : 1];
#pragma unroll
    for (int x = 0; x < P; ++x) cnt[x] = 0;

    // this thread's share of a stage: global -> registers (early), registers -> LDS (late)
    // Row staging: global -> LDS DMA (global_load_lds_dwordx4), no VGPR round trip.  Each
    // wave-instruction writes 64 x 16 B linearly at a wave-uniform LDS base; the source
    // address is per lane, so piece p = (chunk*R + row)*7 + plane_pair lands at LDS slot p.
    // Pieces past the end re-load an early piece into the unused tail (same bytes, harmless).
#define SKL_STAGE_DMA(T, BUF)                                                                \
    do {                                                                                     \
        const uint32_t k_ = g.k_begin + (T) / stages_per_k;                                  \
        const uint32_t c0_ = ((T) % stages_per_k) * S;                                       \
        _Pragma("unroll") for (int u = 0; u < PPT; ++u)                                      \
        {                                                                                    \
            const uint32_t pp_ = tid + u * 256u;                                             \
            const uint32_t p_ = pp_ < (uint32_t)PIECES ? pp_ : pp_ - (uint32_t)PIECES;       \
            const uint32_t q_ = p_ % 7u, rc_ = p_ / 7u;                                      \
            const uint32_t r_ = rc_ % R, c_ = rc_ / R;                                       \
            const uint64_t *src_ = g.A + (size_t)(a0 + r_) * sample_stride +                 \
                                   (size_t)k_ * kmer_stride + (size_t)(c0_ + c_) * BBITS +   \
                                   2u * q_;                                                  \
            skl_dma16(src_, &lds_rows[BUF][u * 256u + wave * 64u]);                          \
        }                                                                                    \
    } while (0)

    SKL_STAGE_DMA(0u, 0);
    __syncthreads();

    for (uint32_t t = 0; t < n_stages; ++t) {
        const uint32_t buf = t & 1u;
        const uint32_t kk = t / stages_per_k;
        const uint32_t k = g.k_begin + kk;
        const uint32_t c0 = (t % stages_per_k) * S;
        const bool have_next = t + 1 < n_stages;
        if (have_next) SKL_STAGE_DMA(t + 1, buf ^ 1u);  // lands under this stage's VALU work

        if (c0 == 0 && active) {
            // start of a k-mer length
            if constexpr (MODE == MODE_COREACC) {
                if (kk > 0 && (kk & 1u) == 0u) {
#pragma unroll
                    for (int x = 0; x < P; ++x) {
                        cold[(kk / 2u - 1u) * P + x] = cnt[x];
                        cnt[x] = 0;
                    }
                } else {
#pragma unroll
                    for (int x = 0; x < P; ++x) cnt[x] <<= 16;
                }
            } else {
#pragma unroll
                for (int x = 0; x < P; ++x) cnt[x] = 0;
            }
        }

        if (active) {
            const uint32_t c_end = (g.ss64 - c0) < (uint32_t)S ? (g.ss64 - c0) : (uint32_t)S;
            for (uint32_t c = 0; c < c_end; ++c) {
                // column operand of this chunk (blocks past the end are clamped; their
                // results are never stored)
                uint4 b[JL][7];
#pragma unroll
                for (int j = 0; j < JL; ++j) {
                    if ((ABL & 2) && c > 0) {  // timing-only ablation: reuse stale registers
#pragma unroll
                        for (int q = 0; q < 7; ++q) asm volatile("" : "=v"(b[j][q]));
                        continue;
                    }
                    const uint32_t jb = (jb0 + j) < g.n_jblocks ? (jb0 + j) : (g.n_jblocks - 1u);
                    const uint4 *bp =
                        g.B + (((size_t)jb * g.nk + k) * g.ss64 + (c0 + c)) * (7 * LANES) + lane;
#pragma unroll
                    for (int q = 0; q < 7; ++q) b[j][q] = bp[q * LANES];
                }
                const uint4 *rows = &lds_rows[buf][(size_t)c * R * 7];
                // Row operand: 7 uniform-address ds_read_b128 per row put the row's 14 planes
                // into 28 VGPRs of every lane.  Rolling prefetch with no extra registers: the
                // read of row r+1's plane pair q is issued right after row r's last use of
                // a[q] (LDS returns in order, so the counted lgkmcnt the compiler inserts
                // before each use is exact).  Prefetch distance: most of one row's VALU work.
                uint4 a[7];
                if constexpr (ADB) {
#pragma unroll
                    for (int q = 0; q < 7; ++q) a[q] = rows[q];
                }
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if constexpr (!ADB) {
#pragma unroll
                        for (int q = 0; q < 7; ++q) a[q] = rows[r * 7 + q];
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    uint32_t mlo[JL], mhi[JL];
#pragma unroll
                    for (int q = 0; q < 7; ++q) {
#pragma unroll
                        for (int j = 0; j < JL; ++j) {
                            // b is stored (hi, lo) per plane: .y/.w are the low halves
                            if (q == 0) {
                                mlo[j] = a[0].x ^ b[j][0].y;
                                mhi[j] = a[0].y ^ b[j][0].x;
                            } else {
                                mlo[j] = ORD ? acc_mismatch_vvv(mlo[j], a[q].x, b[j][q].y)
                                             : acc_mismatch<true>(mlo[j], a[q].x, b[j][q].y);
                                mhi[j] = ORD ? acc_mismatch_vvv(mhi[j], a[q].y, b[j][q].x)
                                             : acc_mismatch<true>(mhi[j], a[q].y, b[j][q].x);
                            }
                            mlo[j] = ORD ? acc_mismatch_vvv(mlo[j], a[q].z, b[j][q].w)
                                         : acc_mismatch<true>(mlo[j], a[q].z, b[j][q].w);
                            mhi[j] = ORD ? acc_mismatch_vvv(mhi[j], a[q].w, b[j][q].z)
                                         : acc_mismatch<true>(mhi[j], a[q].w, b[j][q].z);
                        }
                        if constexpr (ADB) {
                            // a[q] is dead for this row: start fetching the next row's
                            __builtin_amdgcn_sched_barrier(0);
                            if (r + 1 < R && !(ABL & 1)) a[q] = rows[(r + 1) * 7 + q];
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
#pragma unroll
                    for (int j = 0; j < JL; ++j) {
                        // popcount with the add fused (v_bcnt_u32_b32 d, m, d): hipcc otherwise
                        // emits two bcnt + one add3 -- three half-rate ops instead of two
                        asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(cnt[r * JL + j]) : "v"(mlo[j]));
                        asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(cnt[r * JL + j]) : "v"(mhi[j]));
                    }
                    if constexpr (!ADB) __builtin_amdgcn_sched_barrier(0);
                }
            }
        }

        __syncthreads();

        // ---- per-k epilogue ----
        if (active && (t % stages_per_k) == stages_per_k - 1u) {
            if constexpr (MODE != MODE_COREACC) {
                // one store body, rotating the counter window (they are reset next k anyway)
#pragma clang loop unroll(disable)
                for (int x = 0; x < P; ++x) {
                    const uint32_t r = (uint32_t)x / JL, j = (uint32_t)x % JL;
                    if constexpr (MODE == MODE_COUNTS) {
                        store_count(g, a0 + r, (jb0 + j) * 64u + lane, kk, g.ss64 * 64u, cnt[0]);
                    } else {
                        store_jaccard(g, a0 + r, (jb0 + j) * 64u + lane, cnt[0]);
                    }
#pragma unroll
                    for (int y = 0; y + 1 < P; ++y) cnt[y] = cnt[y + 1];
                }
            }
        }
    }
#undef SKL_STAGE_DMA

    if constexpr (MODE == MODE_COREACC) {
        if (active) {
            // Re-assemble the shift-register view store_coreacc() expects: newest k in the
            // low field of s0.  nk = 2m + e: cnt holds the last (2 - ... ) fields.
            const uint32_t nk = g.k_count;
            const uint32_t in_cnt = (nk & 1u) ? 1u : 2u;          // fields held by cnt
            const uint32_t n_cold = (nk - in_cnt) / 2u;           // parked words per pair
            // one epilogue body: take slot 0, rotate the register window (P iterations)
#pragma clang loop unroll(disable)
            for (int x = 0; x < P; ++x) {
                const uint32_t r = (uint32_t)x / JL, j = (uint32_t)x % JL;
                // fields oldest -> newest: cold[0] (hi, lo), cold[1] (hi, lo), cnt (hi?, lo)
                uint32_t f[6] = {0, 0, 0, 0, 0, 0};
                uint32_t nf = 0;
                for (uint32_t w = 0; w < n_cold; ++w) {
                    const uint32_t v = cold[w * P + x];
                    f[nf++] = v >> 16;
                    f[nf++] = v & 0xFFFFu;
                }
                if (in_cnt == 2u) f[nf++] = cnt[0] >> 16;
                f[nf++] = cnt[0] & 0xFFFFu;
                // pack newest-lowest: field index from the newest end
                uint32_t s[3] = {0, 0, 0};
                for (uint32_t u = 0; u < nf; ++u) {
                    const uint32_t from_new = nf - 1u - u;
                    s[from_new >> 1] |= f[u] << ((from_new & 1u) * 16u);
                }
                store_coreacc(g, a0 + r, (jb0 + j) * 64u + lane, s[0], s[1], s[2]);
#pragma unroll
                for (int y = 0; y + 1 < P; ++y) cnt[y] = cnt[y + 1];
            }
        }
    }
}

template <int R, int JL, bool ADB, int ORD>
static hipError_t launch_rjao(const PairArgs &args, int mode, dim3 grid, hipStream_t stream)
{
    const dim3 block(LANES * WAVES_PER_WG);
    switch (mode) {
        case MODE_COUNTS:
            hipLaunchKernelGGL((pair_kernel_lds<R, JL, MODE_COUNTS, ADB, ORD>), grid, block, 0, stream,
                               args);
            break;
        case MODE_JACCARD:
            hipLaunchKernelGGL((pair_kernel_lds<R, JL, MODE_JACCARD, ADB, ORD>), grid, block, 0, stream,
                               args);
            break;
        case MODE_COREACC:
            hipLaunchKernelGGL((pair_kernel_lds<R, JL, MODE_COREACC, ADB, ORD>), grid, block, 0, stream,
                               args);
            break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

template <int R, int JL>
static hipError_t launch_rj(const PairArgs &args, int mode, dim3 grid, hipStream_t stream)
{
    // ADB = true (rolling row prefetch), ORD = 1 (accumulator in src2): the A/B runs that
    // chose them are in profiles/r01_kernel_sweeps.jsonl
    return launch_rjao<R, JL, true, 1>(args, mode, grid, stream);
}

// shape: rows R and columns-per-lane JL, encoded as R*10 + JL
hipError_t launch_pair_kernel_lds(const PairArgs &args_in, int mode, int shape, TileScratch &scratch,
                                  hipStream_t stream)
{
    PairArgs args = args_in;
    if (args.row_end <= args.row_begin || args.nB == 0) return hipSuccess;
    const int R = shape / 10, JL = shape % 10;
    uint64_t n_wg = 0;
    const hipError_t pe = plan_tiles(args, (uint32_t)R, WAVES_PER_WG * (uint32_t)JL * 64u, scratch, stream, &n_wg);
    if (pe != hipSuccess) return pe;
    if (n_wg == 0) return hipSuccess;
    const dim3 grid((unsigned)n_wg);
    static const int abl = [] {
        const char *e = getenv("SKL_LDS_ABLATE");  // timing-only: 1 no row re-reads, 2 no column reloads
        return e ? atoi(e) : 0;
    }();
    if (abl && shape == 162 && mode == MODE_COREACC) {
        const dim3 block(LANES * WAVES_PER_WG);
        if (abl == 1) hipLaunchKernelGGL((pair_kernel_lds<16, 2, MODE_COREACC, true, 1, 1>), grid, block, 0, stream, args);
        else if (abl == 2) hipLaunchKernelGGL((pair_kernel_lds<16, 2, MODE_COREACC, true, 1, 2>), grid, block, 0, stream, args);
        else hipLaunchKernelGGL((pair_kernel_lds<16, 2, MODE_COREACC, true, 1, 3>), grid, block, 0, stream, args);
        return hipGetLastError();
    }
    switch (shape) {
        case 41: return launch_rj<4, 1>(args, mode, grid, stream);
        case 81: return launch_rj<8, 1>(args, mode, grid, stream);
        case 82: return launch_rj<8, 2>(args, mode, grid, stream);
        case 162: return launch_rj<16, 2>(args, mode, grid, stream);
        default: return hipErrorInvalidValue;
    }
}

int choose_lds_shape(uint64_t n_rows, uint64_t n_cols, int self_mode, int mode)
{
    const char *e = getenv("SKL_LDS_SHAPE");  // tuning knob: 41, 81, 82, 162 (read per call)
    const int forced = e ? atoi(e) : 0;
    if (forced) return forced;
    const uint64_t pairs = self_mode ? n_rows * n_cols / 2 : n_rows * n_cols;
    (void)mode;
    // with the balanced tile enumeration the 16x512 tile is at least as fast as 8x256 from
    // ~8M pairs up (profiles/r01_kernel_sweeps.jsonl, sweeps 21/22); smaller launches go
    // to pair_ksplit.hip
    (void)pairs;
    return 162;
}

}  // namespace skl
