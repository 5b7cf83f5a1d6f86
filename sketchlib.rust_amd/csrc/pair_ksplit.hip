// pair_ksplit.hip -- LDS-staged pair kernel for SMALL launches (gfx950).
//
// BASELINE.json's single-GPU configuration is 1 000 genomes = 499 500 pairs: 1.6e8
// (pair, k, chunk) steps, ~0.1 ms of VALU time for the whole chip.  With one wave per
// (rows x 64 columns) tile walking all chunks there are not enough waves to give each of
// the 1 024 SIMDs the 2-3 resident waves the VALU needs to issue at full rate (one wave
// alone issues one VALU per ~7 clk, scripts/microbench/vgpr_banks.hip).  So here the
// chunk ("K") dimension is split across the 4 waves of a workgroup:
//
//   workgroup tile = R rows x 64 columns; wave w takes chunks c = w (mod 4) of every
//   k-mer length; rows are staged through LDS and broadcast exactly as in pair_lds.hip;
//   at the end of each k-mer length the 4 partial counts per pair are summed through LDS,
//   wave w keeping the totals of pair slots x = w (mod 4) and later running their epilogue.
//
// Same instruction stream per (row, column, chunk) as pair_lds.hip (2 v_xor + 26
// v_bitop3 + 2 v_bcnt, all-VGPR, bank-conflict-free); 4x as many waves per pair.
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

template <int R, int MODE, int S, bool BPF>
__global__ __launch_bounds__(LANES *WAVES_PER_WG) void pair_kernel_ksplit(const PairArgs g)
{
    constexpr int PIECES = R * S * 7;            // 16-byte pieces per stage
    constexpr int PPT = (PIECES + 255) / 256;
    constexpr int SLOTS = R / WAVES_PER_WG;      // pair slots (rows) finished by each wave
    static_assert(R % WAVES_PER_WG == 0, "rows per tile must be a multiple of the wave count");
    static_assert(PPT * 256 - PIECES < PIECES, "tail pieces wrap at most once");
    __shared__ uint4 lds_rows[2][PPT * 256];
    __shared__ uint32_t lds_red[2][WAVES_PER_WG][R][LANES];  // double-buffered by k parity

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    uint32_t jb, at;  // 64-wide column block, row tile
    if (!lookup_tile(g, jb, at)) return;
    const uint32_t a0 = g.row_begin + at * R;
    if (jb >= g.n_jblocks) return;
    if (a0 >= g.row_end) return;
    if (g.self_mode && a0 >= jb * 64u + 63u) return;     // tile entirely on/below the diagonal
    const uint32_t jcol = jb * 64u + lane;

    const size_t kmer_stride = (size_t)g.ss64 * BBITS;
    const size_t sample_stride = kmer_stride * g.nk;
    const uint32_t stages_per_k = (g.ss64 + S - 1) / S;
    const uint32_t n_stages = stages_per_k * g.k_count;

    uint32_t cnt[R];                 // this wave's partial mismatch counts (its chunks only)
    uint32_t st0[SLOTS], st1[SLOTS], st2[SLOTS];   // totals of this wave's slots, packed per k
#pragma unroll
    for (int x = 0; x < R; ++x) cnt[x] = 0;
#pragma unroll
    for (int x = 0; x < SLOTS; ++x) {
        st0[x] = 0;
        st1[x] = 0;
        st2[x] = 0;
    }

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
    // first column chunk of this wave
    uint4 b_nxt[7];
    if constexpr (BPF) {
        const uint32_t c_first = wave < g.ss64 ? wave : g.ss64 - 1u;
        const uint4 *bp0 = g.B + (((size_t)jb * g.nk + g.k_begin) * g.ss64 + c_first) * (7 * LANES) + lane;
#pragma unroll
        for (int q = 0; q < 7; ++q) b_nxt[q] = bp0[q * LANES];
    }
    __syncthreads();

    for (uint32_t t = 0; t < n_stages; ++t) {
        const uint32_t buf = t & 1u;
        const uint32_t kk = t / stages_per_k;
        const uint32_t k = g.k_begin + kk;
        const uint32_t c0 = (t % stages_per_k) * S;
        const bool have_next = t + 1 < n_stages;
        if (have_next) SKL_STAGE_DMA(t + 1, buf ^ 1u);  // lands under this stage's VALU work

        const uint32_t c_end = (g.ss64 - c0) < (uint32_t)S ? (g.ss64 - c0) : (uint32_t)S;
        for (uint32_t c = wave; c < c_end; c += WAVES_PER_WG) {   // this wave's chunks
            // column operand: the NEXT chunk of this wave (same stage, or the first one of the
            // next stage) is requested now and lands under this chunk's VALU work
            uint4 b[7];
            if constexpr (!BPF) {
                const uint4 *bp = g.B + (((size_t)jb * g.nk + k) * g.ss64 + (c0 + c)) * (7 * LANES) + lane;
#pragma unroll
                for (int q = 0; q < 7; ++q) b[q] = bp[q * LANES];
            } else {
#pragma unroll
                for (int q = 0; q < 7; ++q) b[q] = b_nxt[q];
                uint32_t nk_ = k, nc_ = c0 + c + WAVES_PER_WG;
                if (c + WAVES_PER_WG >= c_end) {          // first chunk of the next stage
                    const uint32_t tn = t + 1 < n_stages ? t + 1 : t;
                    nk_ = g.k_begin + tn / stages_per_k;
                    nc_ = (tn % stages_per_k) * S + wave;
                    if (nc_ >= g.ss64) nc_ = g.ss64 - 1u;  // wave has no chunk there: harmless re-read
                }
                const uint4 *bpn = g.B + (((size_t)jb * g.nk + nk_) * g.ss64 + nc_) * (7 * LANES) + lane;
#pragma unroll
                for (int q = 0; q < 7; ++q) b_nxt[q] = bpn[q * LANES];
            }
            const uint4 *rows = &lds_rows[buf][(size_t)c * R * 7];
            uint4 a[7];
#pragma unroll
            for (int q = 0; q < 7; ++q) a[q] = rows[q];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                uint32_t mlo, mhi;
#pragma unroll
                for (int q = 0; q < 7; ++q) {
                    // column slab stores each plane as (hi, lo): see pair_lds.hip "VGPR banks"
                    if (q == 0) {
                        mlo = a[0].x ^ b[0].y;
                        mhi = a[0].y ^ b[0].x;
                    } else {
                        mlo = acc_mismatch_vvv(mlo, a[q].x, b[q].y);
                        mhi = acc_mismatch_vvv(mhi, a[q].y, b[q].x);
                    }
                    mlo = acc_mismatch_vvv(mlo, a[q].z, b[q].w);
                    mhi = acc_mismatch_vvv(mhi, a[q].w, b[q].z);
                    // rolling prefetch of the next row's plane pair (no extra registers)
                    __builtin_amdgcn_sched_barrier(0);
                    if (r + 1 < R) a[q] = rows[(r + 1) * 7 + q];
                    __builtin_amdgcn_sched_barrier(0);
                }
                asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(cnt[r]) : "v"(mlo));
                asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(cnt[r]) : "v"(mhi));
            }
        }


        const bool k_done = (t % stages_per_k) == stages_per_k - 1u;
        if (k_done) {
            // publish this wave's partial counts of the finished k-mer length
#pragma unroll
            for (int x = 0; x < R; ++x) {
                lds_red[kk & 1u][wave][x][lane] = cnt[x];
                cnt[x] = 0;
            }
        }
        __syncthreads();
        if (k_done) {
            // wave w owns pair slots (rows) w, w + 4, ...: sum the 4 partials
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) {
                const uint32_t x = (uint32_t)s * WAVES_PER_WG + wave;
                uint32_t total = 0;
#pragma unroll
                for (int w = 0; w < WAVES_PER_WG; ++w) total += lds_red[kk & 1u][w][x][lane];
                if constexpr (MODE == MODE_COUNTS) {
                    store_count(g, a0 + x, jcol, kk, g.ss64 * 64u, total);
                } else if constexpr (MODE == MODE_JACCARD) {
                    store_jaccard(g, a0 + x, jcol, total);
                } else {
                    st2[s] = __builtin_amdgcn_alignbit(st2[s], st1[s], 16);
                    st1[s] = __builtin_amdgcn_alignbit(st1[s], st0[s], 16);
                    st0[s] = (st0[s] << 16) | total;
                }
            }
            // lds_red[kk & 1] is rewritten two k-mer lengths later, i.e. at least two barriers
            // after these reads: no extra barrier needed.
        }
    }
#undef SKL_STAGE_DMA

    if constexpr (MODE == MODE_COREACC) {
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            store_coreacc(g, a0 + (uint32_t)s * WAVES_PER_WG + wave, jcol, st0[s], st1[s], st2[s]);
        }
    }
}

template <int R, int S, bool BPF>
static hipError_t launch_rsb(const PairArgs &args, int mode, dim3 grid, hipStream_t stream)
{
    const dim3 block(LANES * WAVES_PER_WG);
    switch (mode) {
        case MODE_COUNTS:
            hipLaunchKernelGGL((pair_kernel_ksplit<R, MODE_COUNTS, S, BPF>), grid, block, 0, stream, args);
            break;
        case MODE_JACCARD:
            hipLaunchKernelGGL((pair_kernel_ksplit<R, MODE_JACCARD, S, BPF>), grid, block, 0, stream, args);
            break;
        case MODE_COREACC:
            hipLaunchKernelGGL((pair_kernel_ksplit<R, MODE_COREACC, S, BPF>), grid, block, 0, stream, args);
            break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

template <int R>
static hipError_t launch_r(const PairArgs &args, int mode, dim3 grid, hipStream_t stream)
{
    // S = 8 chunks per stage, no column prefetch: the fastest of the four (S, prefetch)
    // combinations measured on MI355X (profiles/r01_kernel_sweeps.jsonl, sweep 14) --
    // extra registers / LDS cost more occupancy than the prefetch buys at this launch size.
    return launch_rsb<R, 8, false>(args, mode, grid, stream);
}

hipError_t launch_pair_kernel_ksplit(const PairArgs &args_in, int mode, int rows_per_tile,
                                     TileScratch &scratch, hipStream_t stream)
{
    PairArgs args = args_in;
    if (args.row_end <= args.row_begin || args.nB == 0) return hipSuccess;
    uint64_t n_wg = 0;
    const hipError_t pe = plan_tiles(args, (uint32_t)rows_per_tile, 64u, scratch, stream, &n_wg);
    if (pe != hipSuccess) return pe;
    if (n_wg == 0) return hipSuccess;
    const dim3 grid((unsigned)n_wg);
    switch (rows_per_tile) {
        case 8: return launch_r<8>(args, mode, grid, stream);
#ifdef SKL_AB
        case 4: return launch_r<4>(args, mode, grid, stream);
        case 16: return launch_r<16>(args, mode, grid, stream);
#endif
        default: return hipErrorInvalidValue;
    }
}

}  // namespace skl
