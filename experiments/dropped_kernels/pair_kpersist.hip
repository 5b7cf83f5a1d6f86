// pair_kpersist.hip -- the chunk-split pair kernel as a PERSISTENT launch (gfx950), for SMALL
// launches of the k-sliced core/accessory form (bin-match counts into a k-major scratch array +
// epilogue kernel).  A/B BUILD ONLY (-DSKL_AB, SKL_PERSIST=2): plain chunk slices of such launches
// (pair_kslice.hip, PairArgs::tail_slices) do the same job as well or better with a tenth of the code
// (profiles/r02_ab_tail_slices.jsonl), so that is what the product library ships;
// this kernel stays as the measured answer to "a dynamic work queue, one atomic per fetch".
//
// Why.  pair_kslice.hip's k-sliced form launches one workgroup per (tile, k-mer length): U units of
// equal cost on G = 4 x CUs resident workgroup slots.  A unit is 8 stages of 4 waves (sketchsize64 =
// 64); one wave alone on its SIMD issues a VALU instruction every 5.1 cycles, three or more share
// the pipe's 1.9 (scripts/microbench/valu_clock.hip).  With U < G most SIMDs hold 0-2 waves and
// the launch takes the time of ONE unit walked at a lone wave's rate whatever U is: 0.100 ms from
// 100 to 600 genomes (k = 5).  Here the launch is exactly the resident set and its workgroups share
// the STAGES of all units, so every SIMD has 4 waves however few units there are: 0.033 ms at 100
// genomes, 0.041 at 200, 0.070 at 500 (profiles/r02_ab_persist.jsonl).  From U ~ 0.7 G on the
// one-workgroup-per-unit launch fills the chip by itself and is as fast or faster (the parts a unit
// is cut into cost a reduction each).
//
// Work distribution.  The workgroups of an XCD (gx of them) walk that XCD's stage sequence (a stage =
// 8 chunks of one unit, 2 per wave; units in the k-sliced launch's order) in PIECES, a fixed function
// of the piece index: whole units while more than two per workgroup remain, then 2^j stages with
// gx pieces per size, down to single stages.  A workgroup starts with the piece of its own index
// and takes every further index from one atomic counter per XCD.  What was measured on the way
// (1 000 genomes, U = 1.37 G, one-workgroup-per-unit launch = 1.00):
//   equal contiguous shares, no queue            1.17-1.29  a SIMD serves its OLDEST wave first: of four
//                                                           waves with equal work the youngest gets what
//                                                           the others leave and finishes last, alone
//   equal shares, units strided                  1.14-1.17
//   queue, CAS per fetch (size from what is left) 4.6       failed compare-and-swaps serialise
//   queue, fixed piece order, one atomicAdd each  1.08      each new piece started cold
//   + the walk continues across pieces           1.07
//   + wave priority rotated with the stage       0.98-1.00  (s_setprio: the four waves of a SIMD take turns
//                                                           at the head of the queue instead of by age)
//   first-round sizes by age rank instead        1.05
// Part-end accounting at that size (kslice_trace 993): 3.2 parts per wave, per part 2 300 cycles
// waiting at the first barrier, 2 400 for reduction + stores, 1 800 to set up the next part: 8 % of
// a wave's time; the stages themselves run at the LDS-bound 2.25 cycles per issue slot.
//
// The walk is seamless -- the next stage's row DMA and the next chunk's column loads are requested
// one stage / one chunk ahead across unit AND piece boundaries (the piece after the current one is
// always known: its index is fetched one piece ahead), as the all-k form of pair_kslice.hip crosses
// k-mer lengths.  Only a workgroup's first part starts cold.
//
// Counts.  A unit is cut into parts by the piece boundaries.  The part that holds the unit's first
// stage (every whole unit, for one) stores its counts to plane 0 of the scratch array with
// plain stores; any other part ADDS its counts to plane 1 (u32 atomics, at "k index"
// k_count + kk: the layout of pair_kslice.hip's chunk slices).  coreacc_epilogue_kernel sums the two
// planes and writes plane 1 back to zero, so the next launch finds it clean (the host zeroes it
// when the scratch is allocated or was used for something else); it also resets the 8 counters.
// Every record of plane 0 is written by exactly one workgroup in every launch.
//
// The inner loop is pair_kslice.hip's 128-register form: 16 x 128 tiles, 2 columns per lane in
// registers, rows by LDS DMA into a private double buffer per wave, a 4-deep ring of row plane
// pairs, packed u16 counts, v_bitop3 / fused v_bcnt.  Requires sketchsize64 % 8 == 0 (every wave
// has its 2 chunks in every stage); other launches take pair_kslice.hip.
#include "device_common.hpp"

namespace skl {

namespace {

constexpr int KP_R = 16, KP_JL = 2, KP_W = WAVES_PER_WG, KP_CH = 2;
constexpr int KP_PIECES = KP_R * KP_CH * 7;                       // 16-byte pieces per wave-stage
constexpr int KP_PPL = (KP_PIECES + LANES - 1) / LANES;           // DMA instructions per wave-stage
constexpr uint32_t KP_BUF_U4 = KP_PPL * LANES;
constexpr uint32_t KP_TILE_BLOCK = 32;                            // tiles that walk a k-mer length together (pair_kslice.hip)

struct Unit {
    uint32_t a0, jb0, k, kk;   // first row, first 64-column block, absolute k index, k index of the launch
};

}  // namespace

__global__ __launch_bounds__(LANES *WAVES_PER_WG, 4) void pair_kernel_kpersist(const PairArgs g)
{
    constexpr int R = KP_R, JL = KP_JL, W = KP_W, CH = KP_CH, PIECES = KP_PIECES, PPL = KP_PPL;
    constexpr uint32_t BUF_U4 = KP_BUF_U4;
    constexpr int PX = R;            // packed words per lane (word x: row x, fields = the 2 columns)
    constexpr int SLOTS = PX / W;    // packed words finished by each wave
    __shared__ uint4 lds_all[W * 2 * BUF_U4];
    uint4 (*lds_rows)[2][BUF_U4] = reinterpret_cast<uint4 (*)[2][BUF_U4]>(&lds_all[0]);

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- this workgroup's share of its XCD's stage sequence ----
    const uint32_t xcd = blockIdx.x & 7u, wg = blockIdx.x >> 3, gx = gridDim.x >> 3;
    const uint32_t t_lo = xcd * g.tiles_per_xcd;
    if (t_lo >= g.n_active_tiles) return;
    const uint32_t tiles_x = min(g.tiles_per_xcd, g.n_active_tiles - t_lo);
    const uint32_t spk = g.ss64 / (W * CH);                        // stages per unit
    const uint32_t units_x = tiles_x * g.k_count;
    const uint32_t T = units_x * spk;                              // stages of this XCD
    __shared__ uint32_t lds_q;

    const size_t kmer_stride = (size_t)g.ss64 * BBITS;
    const size_t sample_stride = kmer_stride * g.nk;

    // self mode, up to 64 super-groups of column groups (every k-sliced launch): the whole tile
    // prefix table in one register, one entry per lane
    const uint32_t n_super = (g.n_groups + g.group_span - 1u) / g.group_span;
    const uint32_t prefix_lane = (g.self_mode && n_super <= 64u) ? g.tile_prefix[lane < n_super ? lane : n_super - 1u] : 0u;

    // unit index on this XCD -> (tile, k): blocks of KP_TILE_BLOCK tiles walk one k-mer length
    // together (tile fastest, then k, then block), as in the k-sliced launch of pair_kslice.hip
    auto decode = [&](uint32_t u) -> Unit {
        const uint32_t per_blk = KP_TILE_BLOCK * g.k_count;
        const uint32_t blk = u / per_blk, rem = u - blk * per_blk;
        const uint32_t in_blk = min(KP_TILE_BLOCK, tiles_x - blk * KP_TILE_BLOCK);   // the last block may be short
        const uint32_t kk = rem / in_blk, slot = blk * KP_TILE_BLOCK + (rem - kk * in_blk);
        uint32_t jg = 0, at = 0;
        const uint32_t t = t_lo + slot;
        if (!g.self_mode) {
            tile_in_supergroup_cross(g, t, jg, at);
        } else if (n_super <= 64u) {
            // no memory access: lane l holds the first tile of super-group l (loaded once, above); the
            // super-groups' first tiles ascend
            const uint64_t below = __ballot(lane < n_super && prefix_lane <= t);
            const uint32_t sg = (uint32_t)__popcll(below) - 1u;
            tile_in_supergroup_self(g, sg, t - (uint32_t)__builtin_amdgcn_readlane((int)prefix_lane, (int)sg), jg, at);
        } else {
            uint32_t lo = 0, hi = n_super;   // largest lo with prefix[lo] <= t
            while (hi - lo > 1u) {
                const uint32_t mid = (lo + hi) >> 1;
                if (g.tile_prefix[mid] <= t) lo = mid; else hi = mid;
            }
            tile_in_supergroup_self(g, lo, t - g.tile_prefix[lo], jg, at);
        }
        Unit d;
        d.a0 = __builtin_amdgcn_readfirstlane(g.row_begin + at * R);
        d.jb0 = __builtin_amdgcn_readfirstlane(jg * JL);
        d.kk = __builtin_amdgcn_readfirstlane(kk);
        d.k = d.kk + g.k_begin;
        return d;
    };

    const uint32_t lds_base = __builtin_amdgcn_readfirstlane(skl_lds_addr(&lds_all[0]));
    // rows of stage `ts` of unit `d` -> this wave's buffer `buf` (global -> LDS DMA, 16 B per lane)
    auto stage_dma = [&](const Unit &d, uint32_t ts, uint32_t buf) {
        const uint32_t c0 = ts * (W * CH) + wave * CH;
#pragma unroll
        for (int u = 0; u < PPL; ++u) {
            const uint32_t pp = lane + u * 64u;
            const uint32_t p = pp < (uint32_t)PIECES ? pp : pp - (uint32_t)PIECES;
            const uint32_t q = p % 7u, rc = p / 7u;
            const uint32_t r = rc % R, c = rc / R;
            const uint64_t *src = g.A + (size_t)(d.a0 + r) * sample_stride + (size_t)d.k * kmer_stride +
                                  (size_t)(c0 + c) * BBITS + 2u * q;
            skl_dma16(src, lds_base + (((uint32_t)wave * 2u + buf) * BUF_U4 + u * 64u) * 16u);
        }
    };
    auto column_ptr = [&](const Unit &d, int j, uint32_t c) {
        const uint32_t jb = (d.jb0 + j) < g.n_jblocks ? (d.jb0 + j) : (g.n_jblocks - 1u);   // clamped, never stored
        return g.B + (((size_t)jb * g.nk + d.k) * g.ss64 + c) * (7 * LANES) + lane;
    };

    uint32_t cnt[R];   // packed: column 0 low, column 1 high (u16 fields; a part is at most 64 * 256 mismatches per pair)
#pragma unroll
    for (int x = 0; x < R; ++x) cnt[x] = 0;

#ifdef SKL_TRACE
    // scripts/microbench/kslice_trace.hip: per wave {s_memtime, wall clock} at the start and the end of
    // its walk, and the stages it walked (the buffer rides in the dtab field; no output depends on it)
    uint64_t *trace_rec = (uint64_t *)g.dtab + ((size_t)blockIdx.x * W + wave) * 8u;
    if (lane == 0) {
        trace_rec[0] = wall_clock64();
        trace_rec[5] = __builtin_amdgcn_s_memtime();
        trace_rec[1] = trace_rec[0];
        uint32_t hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        trace_rec[4] = ((uint64_t)xcc << 32) | hw;
        trace_rec[7] = 0;
    }
    // second region (after 4 096 records): cycles between the marks of a part's end, summed per wave
    uint64_t *trace_ext = (uint64_t *)g.dtab + 4096u * 8u + ((size_t)blockIdx.x * W + wave) * 8u;
    uint64_t tx_b1 = 0, tx_red = 0, tx_b2 = 0, tx_gen = 0, tx_parts = 0, tx_mark = 0;
#endif
    uint32_t t = 0;            // stages walked: buffer parity
    uint4 b[JL][7];

    // ---- the XCD's queue: pieces of its stage sequence in a fixed, guided order ----
    // Piece p of the sequence is a fixed function of p.  First come rounds of whole units, one unit per
    // piece and gx pieces per round, for as long as more than two units per workgroup remain: that is
    // the order (and the L2 footprint) the hardware dispatcher gives pair_kernel_kslice.  The remainder
    // R goes out in rounds of gx pieces of sz = min(spk, 2^floor(log2(R / (1.3 gx)))) stages -- sizes
    // only ever shrink, so for a power-of-two spk every piece stays inside one unit -- and finally in
    // single stages.  A workgroup's first piece is p = its own index; every further one is the next value
    // of one atomic counter per XCD (one returning atomicAdd per piece, requested a piece ahead).  A wave
    // that finishes early -- a SIMD serves its oldest wave first -- simply comes back for more, so every
    // SIMD keeps 4 working waves until the sequence is used up.
    const uint32_t unit_rounds = T / (gx * spk) >= 2u ? T / (gx * spk) - 1u : 0u;
    const uint32_t age_rank = (wg * 4u / gx) & 3u;
    auto piece_range = [&](uint32_t pc, uint32_t &lo, uint32_t &hi) {
        if (pc < unit_rounds * gx) {
            lo = pc * spk;
            hi = lo + spk;
            return;
        }
        uint32_t q = pc - unit_rounds * gx, start = unit_rounds * gx * spk;
        for (;;) {
            const uint32_t rem = start < T ? T - start : 0u;
            const uint32_t want = (rem * 10u) / (13u * gx);
            const uint32_t sz = want >= 2u ? min(spk, 1u << (31 - __builtin_clz(want))) : 1u;
            if (sz == 1u || q < gx || rem == 0u) {
                lo = start + q * sz;
                hi = min(T, lo + sz);
                return;
            }
            q -= gx;
            start += gx * sz;
        }
    };
    // ---- the stream of parts ----
    // A part = consecutive stages of one unit inside one piece.  gen() yields the next part of this
    // workgroup's stream and moves to the following piece when one is used up.  The piece after the
    // current one is always known (lds_q[]), so the walk is seamless ACROSS pieces too: the first
    // rows and columns of a part are requested during the last stage of the part before it, whichever
    // piece that belongs to, and only the very first part of a workgroup starts cold.
    // The index of the piece after the current one travels through lds_q: the prologue requests the
    // first from the queue counter; every later one is requested by tid 0 (one returning atomic) at the
    // end of the part that uses its piece up -- before that part-end's first barrier -- and published
    // between its two barriers, when every wave has read the previous value and none can read the new
    // one yet.
    struct Part { Unit u; uint32_t ts0, len; };
    uint32_t gv = 0, gv_hi = 0;          // generator: next stage position / end of its piece
    // the prologue takes two indices at once: `spare` serves the one switch that may come before the
    // first part-end (a first piece of a single part)
    uint32_t piece_next = wg;
    if (tid == 0) lds_q = gx + atomicAdd(g.work_counter + xcd * 32u, 2u);
    __syncthreads();
    uint32_t spare = __builtin_amdgcn_readfirstlane(lds_q) + 1u;
    bool spare_valid = true;
    bool first_switch = true;
    auto gen = [&]() -> Part {
        Part p;
        p.u = Unit{0, 0, 0, 0};
        p.ts0 = 0;
        p.len = 0;
        if (gv >= gv_hi) {
            if (piece_next == 0xFFFFFFFFu) return p;
            piece_range(piece_next, gv, gv_hi);
            gv = __builtin_amdgcn_readfirstlane(gv);
            gv_hi = __builtin_amdgcn_readfirstlane(gv_hi);
            if (gv >= T) {               // the sequence is used up (pieces ascend: so are all later ones)
                piece_next = 0xFFFFFFFFu;
                gv = gv_hi = T;
                return p;
            }
            if (first_switch) {
                piece_next = spare - 1u;
                first_switch = false;
            } else if (spare_valid) {
                piece_next = spare;
                spare_valid = false;
            } else {
                piece_next = __builtin_amdgcn_readfirstlane(lds_q);
            }
#ifdef SKL_TRACE
            if (lane == 0) trace_rec[7] += gv_hi - gv;
#endif
        }
        p.ts0 = __builtin_amdgcn_readfirstlane(gv % spk);
        p.len = __builtin_amdgcn_readfirstlane(min(spk - p.ts0, gv_hi - gv));
        p.u = decode(gv / spk);
        gv += p.len;
        return p;
    };

    Part cur = gen();
    if (cur.len == 0u) return;
    stage_dma(cur.u, cur.ts0, t & 1u);
    {
        const uint32_t c1 = cur.ts0 * (W * CH) + wave * CH;
#pragma unroll
        for (int j = 0; j < JL; ++j) {
            const uint4 *bp = column_ptr(cur.u, j, c1);
#pragma unroll
            for (int q = 0; q < 7; ++q) b[j][q] = bp[q * LANES];
        }
    }
    // Column loads issued after the newest row DMA.  Only these are counted: the stores of a part's
    // end, the queue request and the loads of decode() are younger than the DMA too, but a store may
    // be skipped (no valid lane), and counting an operation that was not issued would make the wait
    // below too lax; leaving them out makes it stricter by at most that many column loads, which are
    // ~2 us old by then.
    uint32_t young = JL * 7;

    for (;;) {
        const Part np = gen();
        const bool has_next = np.len != 0u;
#ifdef SKL_TRACE
        if (tx_mark) tx_gen += __builtin_amdgcn_s_memtime() - tx_mark;
#endif
        const Unit nxt = np.u;
        const uint32_t ts_n = np.ts0;
        const uint32_t ts0 = cur.ts0, ts1 = cur.ts0 + cur.len;
        const Unit cu = cur.u;

        for (uint32_t ts = ts0; ts < ts1; ++ts, ++t) {
            const uint32_t buf = t & 1u;
            const uint32_t c0 = ts * (W * CH) + wave * CH;
            // A SIMD issues for its highest-priority wave first and, among equals, for the oldest: left
            // alone, the youngest of four waves with equal work gets what the others leave.  Rotating the
            // priority with the stage number gives the four equal turns.
            switch ((t + age_rank) & 3u) {
            case 0: __builtin_amdgcn_s_setprio(0); break;
            case 1: __builtin_amdgcn_s_setprio(1); break;
            case 2: __builtin_amdgcn_s_setprio(2); break;
            default: __builtin_amdgcn_s_setprio(3); break;
            }
            // This wave's DMA of this stage must have landed.  VMEM returns in order, so it is enough
            // that only the `young` younger operations may still be in flight.
            if (young < (uint32_t)(2 * JL * 7)) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(JL * 7) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * JL * 7) : "memory");
            }
            const bool last_of_part = ts + 1u == ts1;
            const bool want_dma = !last_of_part || has_next;

#pragma unroll
            for (uint32_t ci = 0; ci < (uint32_t)CH; ++ci) {
                // where the columns of the next chunk are (the current ones again if none follows)
                const bool into_next = ci + 1u == (uint32_t)CH && last_of_part;
                const Unit &dn = into_next ? nxt : cu;
                const uint32_t cn = ci + 1u < (uint32_t)CH ? c0 + ci + 1u
                                    : (!last_of_part ? c0 + (uint32_t)(W * CH)
                                                     : (has_next ? ts_n * (uint32_t)(W * CH) + wave * CH : c0 + ci));
                const uint4 *bn[JL];
#pragma unroll
                for (int j = 0; j < JL; ++j) bn[j] = column_ptr(dn, j, cn);
                const uint4 *rows = &lds_rows[wave][buf][(size_t)ci * R * 7];
                constexpr int AD = 4;   // ring of row plane pairs: step s_ = r * 7 + q reads rows[s_], re-loaded AD steps ahead
                uint4 a[AD];
#pragma unroll
                for (int q = 0; q < AD; ++q) a[q] = rows[q];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    uint32_t mlo[JL], mhi[JL];
#pragma unroll
                    for (int q = 0; q < 7; ++q) {
                        const int s_ = r * 7 + q;
                        uint4 &ar = a[s_ % AD];
#pragma unroll
                        for (int j = 0; j < JL; ++j) {
                            // b is stored (hi, lo) per plane: see device_common.hpp "VGPR banks"
                            if (q == 0) {
                                mlo[j] = ar.x ^ b[j][0].y;
                                mhi[j] = ar.y ^ b[j][0].x;
                            } else {
                                mlo[j] = acc_mismatch_vvv(mlo[j], ar.x, b[j][q].y);
                                mhi[j] = acc_mismatch_vvv(mhi[j], ar.y, b[j][q].x);
                            }
                            mlo[j] = acc_mismatch_vvv(mlo[j], ar.z, b[j][q].w);
                            mhi[j] = acc_mismatch_vvv(mhi[j], ar.w, b[j][q].z);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        if (s_ + AD < R * 7) ar = rows[s_ + AD];
                        if (r == R - 1) {   // last use of b[.][q] in this chunk: fetch the next chunk's
#pragma unroll
                            for (int j = 0; j < JL; ++j) b[j][q] = bn[j][q * LANES];
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    // popcounts, the add fused; column 1 goes to the high field
                    asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(cnt[r]) : "v"(mlo[0]));
                    asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(cnt[r]) : "v"(mhi[0]));
                    uint32_t t1;
                    asm("v_bcnt_u32_b32 %0, %1, 0" : "=v"(t1) : "v"(mlo[1]));
                    asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(t1) : "v"(mhi[1]));
                    cnt[r] = (t1 << 16) + cnt[r];
                    if (r == 0 && ci == 0 && want_dma) {
                        // the next stage's rows, requested when no column load is in flight (all column
                        // registers consumed once): lands under this stage's VALU work
                        if (!last_of_part) stage_dma(cu, ts + 1u, buf ^ 1u);
                        else stage_dma(nxt, ts_n, buf ^ 1u);
                        young = 0;
                    }
                }
                young += JL * 7;
            }
        }

#ifdef SKL_TRACE
        tx_mark = __builtin_amdgcn_s_memtime();
        ++tx_parts;
#endif
        // ---- end of this part of the unit: sum the 4 waves' partial counts through LDS, store ----
        const uint32_t dead = (t - 1u) & 1u;   // the buffer this wave consumed last (the next stage sits in the other)
        uint32_t *red = reinterpret_cast<uint32_t *>(&lds_all[0]);
#pragma unroll
        for (int x = 0; x < PX; ++x) {
            red[((uint32_t)wave * 2u + dead) * (BUF_U4 * 4u) + (uint32_t)x * LANES + lane] = cnt[x];
            cnt[x] = 0;
        }
        // the generator is at the end of its piece and there is a next one: the next gen() switches
        const bool refill = gv >= gv_hi && piece_next != 0xFFFFFFFFu && !spare_valid;
        uint32_t pend = 0;
        if (refill && tid == 0) pend = gx + atomicAdd(g.work_counter + xcd * 32u, 1u);
        __syncthreads();
#ifdef SKL_TRACE
        { const uint64_t m = __builtin_amdgcn_s_memtime(); tx_b1 += m - tx_mark; tx_mark = m; }
#endif
        const bool first_part = ts0 == 0u;   // plane 0, plain stores; any other part: added to plane 1
        const uint32_t part_bins = (ts1 - ts0) * (uint32_t)(W * CH) * 64u;
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) {
            const uint32_t x = (uint32_t)i * W + wave;   // row x of the tile, both columns
            uint32_t total = 0;
#pragma unroll
            for (int w = 0; w < W; ++w) total += red[((uint32_t)w * 2u + dead) * (BUF_U4 * 4u) + x * LANES + lane];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t mism = h ? (total >> 16) : (total & 0xFFFFu);
                const uint32_t i_ = cu.a0 + x, jc_ = (cu.jb0 + (uint32_t)h) * 64u + lane;
                if (first_part) {
                    store_count(g, i_, jc_, cu.kk, part_bins, mism);
                } else if (pair_valid(g, i_, jc_)) {
                    atomicAdd(&((uint32_t *)g.out)[pair_out_index(g, i_, jc_) * g.cnt_pair_stride +
                                                   (uint64_t)(g.k_count + cu.kk) * g.cnt_k_stride],
                              part_bins - mism);
                }
            }
        }
#ifdef SKL_TRACE
        { const uint64_t m = __builtin_amdgcn_s_memtime(); tx_red += m - tx_mark; tx_mark = m; }
#endif
        if (!has_next) break;
        cur = np;
        if (refill && tid == 0) lds_q = pend;
        // the next stage's DMA of every wave goes into the buffer just read: fence the reads
        __syncthreads();
#ifdef SKL_TRACE
        { const uint64_t m = __builtin_amdgcn_s_memtime(); tx_b2 += m - tx_mark; tx_mark = m; }
#endif
    }
#ifdef SKL_TRACE
    if (lane == 0) {
        trace_ext[0] = tx_b1; trace_ext[1] = tx_red; trace_ext[2] = tx_b2; trace_ext[3] = tx_gen; trace_ext[4] = tx_parts;
        trace_rec[6] = __builtin_amdgcn_s_memtime();
        trace_rec[2] = wall_clock64();
        trace_rec[3] = trace_rec[2];
    }
#endif
}

bool kpersist_supported(const PairArgs &args, int mode, uint32_t slots)
{
    if (mode != MODE_COUNTS || !args.k_sliced || args.k_count < 1u) return false;
    if (args.ss64 % (uint32_t)(KP_W * KP_CH) != 0u || args.ss64 > 1023u) return false;
    (void)slots;
    return true;
}

// Grid = the resident set (slots = 4 workgroups per CU, a multiple of 8).  The scratch array has two
// planes; plane 1 must be zero on entry and the epilogue must be told to sum and re-zero it.
// *used = false (nothing launched) when the launch is empty.
hipError_t launch_pair_kernel_kpersist(const PairArgs &args_in, uint32_t slots, TileScratch &scratch,
                                       hipStream_t stream, bool *used)
{
    *used = false;
    PairArgs args = args_in;
    if (args.row_end <= args.row_begin || args.nB == 0) return hipSuccess;
    uint64_t n_wg = 0;
    const hipError_t pe = plan_tiles(args, (uint32_t)KP_R, (uint32_t)KP_JL * 64u, scratch, stream, &n_wg);
    if (pe != hipSuccess) return pe;
    if (n_wg == 0) return hipSuccess;
    slots &= ~7u;
    if (slots < 8u) return hipSuccess;
    args.k_slices = 2;
    *used = true;
    hipLaunchKernelGGL(pair_kernel_kpersist, dim3(slots), dim3(LANES * WAVES_PER_WG), 0, stream, args);
    return hipGetLastError();
}

}  // namespace skl
