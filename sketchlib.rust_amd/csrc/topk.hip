// topk.hip -- k nearest neighbours per row on the device: the radix select (select_smallest,
// topk_kernel for ragged candidate rows), the streaming running top-k of the dense kNN drivers
// (topk_merge_kernel), the merge of partial states of a multi-GPU run (merge_states_kernel)
// and the conversion of a state to the public output form.  mod.rs:41-48 semantics with the
// canonical tie rule: smallest (key, index) first.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "kernels.h"

namespace skl {

// ---------------------------------------------------------------------------
// row-wise k nearest neighbours over a dense band (mod.rs:41-48 semantics with the
// canonical tie rule: smallest (key, index) first)
// ---------------------------------------------------------------------------

constexpr int TOPK_THREADS = 256;
constexpr int TOPK_MAX = 2048;  // knn upper bound handled on device

// (the pair kernel compares with the same mapping: device_common.hpp)
__device__ __forceinline__ uint32_t sortable_bits(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ float from_sortable_bits(uint32_t s)
{
    return __uint_as_float((s & 0x80000000u) ? (s & 0x7FFFFFFFu) : ~s);
}

struct TopkShared {
    uint32_t hist[256];
    uint32_t prefix, remaining, count, taken;
    uint32_t wave_cnt[TOPK_THREADS / 64];
    uint64_t items[TOPK_MAX];  // (sortable key << 32) | position
};

// Bitonic sort of items[0, m) (m a power of two), ascending.  `items` is the workgroup's own array: LDS, or --
// for more neighbours than fit there -- global memory (a barrier orders a workgroup's global accesses too).
__device__ __forceinline__ void sort_items(uint64_t *items, uint32_t m)
{
    const uint32_t tid = threadIdx.x;
    for (uint32_t size = 2; size <= m; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t x = tid; x < m / 2; x += TOPK_THREADS) {
                const uint32_t lo = 2 * x - (x & (stride - 1));
                const uint32_t hi = lo + stride;
                const bool up = (lo & size) == 0;
                const uint64_t a = items[lo], b = items[hi];
                if ((a > b) == up) {
                    items[lo] = b;
                    items[hi] = a;
                }
            }
            __syncthreads();
        }
    }
}

// The knn_eff smallest (key, position) of positions [0, n_items) for which item(c, u) is true
// (u = sortable key bits), left sorted in items[0, knn_eff); `items` holds the next power of two
// >= knn_eff entries (sh.items, or a global array of the workgroup's own when knn_eff > TOPK_MAX).
// At least knn_eff positions must be valid.  Called by the whole workgroup.
template <class Item>
__device__ __forceinline__ void select_smallest(const Item &item, uint32_t n_items, uint32_t knn_eff, TopkShared &sh,
                                                uint64_t *items)
{
    const uint32_t tid = threadIdx.x;
    // ---- radix select: key value of the knn-th smallest ----
    if (tid == 0) {
        sh.prefix = 0;
        sh.remaining = knn_eff;
    }
    __syncthreads();
    for (int pass = 3; pass >= 0; --pass) {
        sh.hist[tid] = 0;
        __syncthreads();
        const uint32_t prefix = sh.prefix;
        const uint32_t hi_mask = pass == 3 ? 0u : (0xFFFFFFFFu << ((pass + 1) * 8));
        for (uint32_t c = tid; c < n_items; c += TOPK_THREADS) {
            uint32_t u;
            if (!item(c, u)) continue;
            const bool in = (u & hi_mask) == (prefix & hi_mask);
            const uint32_t bin = (u >> (pass * 8)) & 0xFFu;
            // wave-aggregated update: distances cluster (unrelated genomes all sit at 1.0), so most
            // lanes of a wave hit the same counter -- one lane adds the whole group
            const uint32_t lead_bin = __builtin_amdgcn_readfirstlane(bin);
            const uint64_t same = __ballot(in && bin == lead_bin);
            if (in) {
                if (bin == lead_bin) {
                    if ((threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(same)) atomicAdd(&sh.hist[bin], (uint32_t)__popcll(same));
                } else {
                    atomicAdd(&sh.hist[bin], 1u);
                }
            }
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t rem = sh.remaining, b = 0;
            for (; b < 256; ++b) {
                if (sh.hist[b] >= rem) break;
                rem -= sh.hist[b];
            }
            if (b > 255) b = 255;
            sh.prefix = prefix | (b << (pass * 8));
            sh.remaining = rem;
        }
        __syncthreads();
    }
    const uint32_t thresh = sh.prefix;      // exact key bits of the knn-th smallest
    const uint32_t take_eq = sh.remaining;  // how many == thresh to take, lowest position first

    // ---- collect: everything below the threshold (any order) ----
    if (tid == 0) {
        sh.count = 0;
        sh.taken = 0;
    }
    __syncthreads();
    for (uint32_t c = tid; c < n_items; c += TOPK_THREADS) {
        uint32_t u;
        if (!item(c, u)) continue;
        if (u < thresh) {
            const uint32_t pos = atomicAdd(&sh.count, 1u);
            items[pos] = ((uint64_t)u << 32) | c;
        }
    }
    __syncthreads();
    // ---- ties at the threshold: ordered compaction, lowest position first ----
    const uint32_t n_less = sh.count;
    for (uint32_t base = 0; base < n_items && sh.taken < take_eq; base += TOPK_THREADS) {
        const uint32_t c = base + tid;
        bool eq = false;
        if (c < n_items) {
            uint32_t u;
            eq = item(c, u) && u == thresh;
        }
        // block-wide exclusive prefix of `eq` via per-wave ballots
        const uint64_t ballot = __ballot(eq);
        const uint32_t wave = tid >> 6, lane = tid & 63u;
        if (lane == 0) sh.wave_cnt[wave] = (uint32_t)__popcll(ballot);
        __syncthreads();
        uint32_t before = (uint32_t)__popcll(ballot & ((1ull << lane) - 1ull));
        for (uint32_t w = 0; w < wave; ++w) before += sh.wave_cnt[w];
        const uint32_t taken = sh.taken;
        if (eq && taken + before < take_eq) {
            items[n_less + taken + before] = ((uint64_t)thresh << 32) | c;
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t tot = 0;
            for (uint32_t w = 0; w < TOPK_THREADS / 64; ++w) tot += sh.wave_cnt[w];
            sh.taken = taken + tot;
        }
        __syncthreads();
    }

    // ---- bitonic sort of the knn items by (key, position) ----
    uint32_t m = 1;
    while (m < knn_eff) m <<= 1;
    for (uint32_t x = knn_eff + tid; x < m; x += TOPK_THREADS) items[x] = ~0ull;
    __syncthreads();
    sort_items(items, m);
}

__global__ __launch_bounds__(TOPK_THREADS) void topk_kernel(const TopkArgs g)
{
    __shared__ TopkShared sh;

    const uint32_t row = blockIdx.x + g.first_row;
    const uint32_t tid = threadIdx.x;
    const bool ragged = g.row_offsets != nullptr;
    const uint64_t row_base = ragged ? g.row_offsets[row] : (uint64_t)row * g.cols;
    const uint32_t n_cols = ragged ? (uint32_t)(g.row_offsets[row + 1] - row_base) : g.cols;
    const uint32_t knn_eff = ragged ? (n_cols < g.knn ? n_cols : g.knn) : g.knn;
    const float *keys = g.keys + row_base * g.stride2;
    const uint32_t self_col = g.self_mode ? g.row_begin + row : 0xFFFFFFFFu;
    if (ragged) {
        // padding entries (fewer candidates than knn): (this row, 1.0)
        for (uint32_t x = knn_eff + tid; x < g.knn; x += TOPK_THREADS) {
            const size_t o = (size_t)row * g.knn + x;
            g.out_idx[o] = g.row_begin + row;
            g.out_d0[o] = 1.0f;
        }
        if (knn_eff == 0) return;
    }
    const uint32_t stride2 = g.stride2;
    // more neighbours than the LDS array holds: the row's own slice of a global scratch array
    uint64_t *items = g.items_scratch ? g.items_scratch + (size_t)blockIdx.x * g.items_pitch : sh.items;
    select_smallest(
        [&](uint32_t c, uint32_t &u) {
            if (c == self_col) return false;
            u = sortable_bits(keys[(size_t)c * stride2]);
            return true;
        },
        n_cols, knn_eff, sh, items);
    for (uint32_t x = tid; x < knn_eff; x += TOPK_THREADS) {
        const uint32_t col = (uint32_t)(items[x] & 0xFFFFFFFFu);
        const size_t o = (size_t)row * g.knn + x;
        const float key = keys[(size_t)col * g.stride2];
        g.out_idx[o] = g.col_ids ? g.col_ids[row_base + col] : col;
        g.out_d0[o] = g.ani_undo ? 1.0f - key : key;
        if (g.stride2 == 2 && g.out_d1) g.out_d1[o] = keys[(size_t)col * 2 + 1];
    }
}

hipError_t launch_topk(const TopkArgs &args, hipStream_t stream)
{
    if (args.rows == 0) return hipSuccess;
    if (args.knn == 0) return hipErrorInvalidValue;
    if (args.knn > TOPK_MAX && (args.items_scratch == nullptr || args.items_pitch < topk_items_pitch(args.knn))) return hipErrorInvalidValue;
    hipLaunchKernelGGL(topk_kernel, dim3(args.rows), dim3(TOPK_THREADS), 0, stream, args);
    return hipGetLastError();
}

// Running top-k of one row (TopkMergeArgs): a streaming merge.  The state sits in LDS as sorted
// (sortable key << 32 | sample id) items -- that composite IS the canonical order, smallest
// (key, id) first.  The row's new records are scanned ONCE, in segments: an item below the
// state's current knn-th one is appended behind the state (up to TOPK_MAX items together), the
// lot is sorted, the first knn are the new state.  A cold state starts with a segment that fits
// the buffer outright; segments grow while few items qualify and shrink (the segment is
// scanned again) when the buffer overflows.  Data that keeps overflowing -- keys arriving in
// descending order -- falls back to the radix select over state ++ remaining records, which
// needs the drivers' guarantee that new ids are larger than the ids already in the state.
__device__ __forceinline__ void topk_merge_row(const TopkMergeArgs &g, const uint32_t row, TopkShared &sh, float *second)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t self_id = g.self_id_base == 0xFFFFFFFFu ? 0xFFFFFFFFu : g.self_id_base + row;
    const uint32_t knn = g.knn;
    const size_t srow = (size_t)(g.state_row_base + row) * knn;
    uint32_t *sk = g.run_key + srow;
    uint32_t *si = g.run_idx + srow;
    const uint32_t stride2 = g.stride2;
    const float *keys = g.keys + (size_t)row * g.key_stride;
    const uint32_t cols = g.cols;
    // the pair kernel marked the stretches of (1 << seg_shift) records that hold something below this row's knn-th best (as
    // of a moment ago: never too few).  Unmarked stretches are NOT candidates -- they may not even have been written (a
    // pruned tile, pair_kslice_walk.inc)
    const uint32_t *seg_bits = g.seg_bits != nullptr ? g.seg_bits + (size_t)row * g.seg_bits_stride : nullptr;
    const uint32_t seg_shift = g.seg_shift != 0u ? g.seg_shift : 6u;
    auto marked = [&](uint32_t q) {
        const uint32_t b = q >> seg_shift;
        return seg_bits == nullptr || ((seg_bits[b >> 5] >> (b & 31u)) & 1u) != 0u;
    };

    auto fresh = [&](uint32_t q, uint32_t &u) {   // new key at position q of this launch
        const uint32_t id = g.id_base + q;
        if (id < g.skip_below || id == self_id || !marked(q)) return false;
        u = sortable_bits(__builtin_nontemporal_load(&keys[(size_t)q * stride2]));   // read once
        return true;
    };
    // second value of a state item: from this launch's records, or from the old state (still
    // untouched in global memory, sorted by the same composite: binary search)
    auto second_of = [&](uint64_t item) {
        const uint32_t id = (uint32_t)(item & 0xFFFFFFFFu);
        const uint32_t q = id - g.id_base;
        if (id >= g.id_base && q < cols && id >= g.skip_below && id != self_id) return keys[(size_t)q * 2 + 1];
        uint32_t lo = 0, hi = knn;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            const uint64_t old = ((uint64_t)sk[mid] << 32) | si[mid];
            if (old < item) lo = mid + 1; else hi = mid;
        }
        return g.run_d1[srow + (lo < knn ? lo : knn - 1)];
    };
    auto write_state = [&]() {   // items[0, knn) (composite) -> global state; whole workgroup
        if (stride2 == 2) {
            for (uint32_t x = tid; x < knn; x += TOPK_THREADS) second[x] = second_of(sh.items[x]);
        }
        __syncthreads();
        for (uint32_t x = tid; x < knn; x += TOPK_THREADS) {
            sk[x] = (uint32_t)(sh.items[x] >> 32);
            si[x] = (uint32_t)(sh.items[x] & 0xFFFFFFFFu);
            if (stride2 == 2) g.run_d1[srow + x] = second[x];
        }
        __syncthreads();
    };

    const uint32_t cap = (uint32_t)TOPK_MAX - knn;   // room behind the state
    uint32_t begin = 0;
    bool changed = false;
    if (cap >= 256u && g.streaming) {
        for (uint32_t x = tid; x < knn; x += TOPK_THREADS) sh.items[x] = ((uint64_t)sk[x] << 32) | si[x];
        __syncthreads();
        uint64_t worst = sh.items[knn - 1];           // ~0 while the state is not full
        uint32_t seg = worst == ~0ull ? cap : 4u * cap;
        uint32_t overflows = 0;
        while (begin < cols) {
            const uint32_t len = min(seg, cols - begin);
            if (tid == 0) sh.count = 0;
            __syncthreads();
            // (UNROLL independent loads per thread before any of them is looked at: one workgroup
            // walks a whole row, so its memory-level parallelism is what the scan runs at)
            constexpr uint32_t UNROLL = 4;
            for (uint32_t q0 = begin; q0 < begin + len; q0 += TOPK_THREADS * UNROLL) {
                if (seg_bits != nullptr) {
                    // a stretch of 1 024 positions without a mark is not read at all
                    const uint32_t *bits = seg_bits;
                    const uint32_t b_lo = q0 >> seg_shift, b_hi = (min(q0 + TOPK_THREADS * UNROLL, begin + len) - 1u) >> seg_shift;
                    bool any = false;
                    for (uint32_t w = b_lo >> 5; w <= (b_hi >> 5); ++w) {
                        uint32_t word = bits[w];
                        if (w == (b_lo >> 5)) word &= ~0u << (b_lo & 31u);
                        if (w == (b_hi >> 5) && (b_hi & 31u) != 31u) word &= (2u << (b_hi & 31u)) - 1u;
                        any |= word != 0u;
                    }
                    if (!any) continue;   // (workgroup-uniform)
                }
                uint64_t item[UNROLL];
                float raw[UNROLL];
#pragma unroll
                for (uint32_t j = 0; j < UNROLL; ++j) {   // unconditional (clamped) loads, all in flight together
                    const uint32_t q = min(q0 + j * TOPK_THREADS + tid, cols - 1u);
                    raw[j] = __builtin_nontemporal_load(&keys[(size_t)q * stride2]);
                }
#pragma unroll
                for (uint32_t j = 0; j < UNROLL; ++j) {
                    const uint32_t q = q0 + j * TOPK_THREADS + tid;
                    const uint32_t id = g.id_base + q;
                    const bool valid = q < begin + len && id >= g.skip_below && id != self_id && marked(q);
                    item[j] = valid ? ((uint64_t)sortable_bits(raw[j]) << 32) | id : ~0ull;
                }
#pragma unroll
                for (uint32_t j = 0; j < UNROLL; ++j) {
                    const bool take = item[j] < worst;
                    const uint64_t votes = __ballot(take);
                    if (votes) {   // one LDS atomic per wave
                        const uint32_t leader = (uint32_t)__builtin_ctzll(votes);
                        uint32_t base = 0;
                        if (lane == leader) base = atomicAdd(&sh.count, (uint32_t)__popcll(votes));
                        base = __shfl(base, leader);
                        const uint32_t pos = base + (uint32_t)__popcll(votes & ((1ull << lane) - 1ull));
                        if (take && pos < cap) sh.items[knn + pos] = item[j];
                    }
                }
            }
            __syncthreads();
            const uint32_t cnt = sh.count;
            __syncthreads();   // (everyone has read the count before the next segment clears it)
            if (cnt > cap) {   // too many for the buffer: scan a shorter segment again
                if (++overflows > 6u) break;
                seg = max(cap, len / 4u);
                continue;
            }
            if (cnt) {
                const uint32_t total = knn + cnt;
                uint32_t m = 1;
                while (m < total) m <<= 1;
                for (uint32_t x = total + tid; x < m; x += TOPK_THREADS) sh.items[x] = ~0ull;
                __syncthreads();
                sort_items(sh.items, m);
                worst = sh.items[knn - 1];
                changed = true;
            }
            begin += len;
            if (cnt * 8u <= cap) seg = min(seg * 2u, 1u << 20);
            else if (cnt * 2u > cap) seg = max(cap, seg / 2u);
        }
        if (begin >= cols) {
            if (changed) write_state();
            return;
        }
        if (changed) write_state();   // the select below reads the state from global memory
    }

    // ---- radix select over state ++ records [begin, cols) (position order = id order) ----
    const uint32_t rest = cols - begin;
    select_smallest(
        [&](uint32_t c, uint32_t &u) {
            if (c < knn) {
                u = sk[c];
                return true;
            }
            return fresh(begin + (c - knn), u);
        },
        knn + rest, knn, sh, sh.items);
    for (uint32_t x = tid; x < knn; x += TOPK_THREADS) {
        const uint32_t pos = (uint32_t)(sh.items[x] & 0xFFFFFFFFu);
        const uint32_t id = pos < knn ? si[pos] : g.id_base + begin + (pos - knn);
        if (stride2 == 2) second[x] = pos < knn ? g.run_d1[srow + pos] : keys[(size_t)(begin + pos - knn) * 2 + 1];
        sh.items[x] = (sh.items[x] & 0xFFFFFFFF00000000ull) | id;
    }
    __syncthreads();
    for (uint32_t x = tid; x < knn; x += TOPK_THREADS) {
        sk[x] = (uint32_t)(sh.items[x] >> 32);
        si[x] = (uint32_t)(sh.items[x] & 0xFFFFFFFFu);
        if (stride2 == 2) g.run_d1[srow + x] = second[x];
    }
}

// One workgroup per row.  (Round 3 also built "one workgroup per 64 rows that merges the flagged ones in turn", so that
// the ~1 M workgroups per band of cfg 5 that only read their flag and exit are never dispatched: wall time and pair-kernel
// time of cfg 5 did not move -- 12.009 against 12.006 s, interleaved in one process, profiles/r03_ab_merge_block.jsonl --
// because these merges run beside the pair kernel on a second stream and are hidden already.  Not kept.)
__global__ __launch_bounds__(TOPK_THREADS) void topk_merge_kernel(const TopkMergeArgs g)
{
    __shared__ TopkShared sh;
    __shared__ float second[TOPK_MAX];   // second values of the new state (stride2 == 2)
    // the pair kernel flagged the rows that received a record below their knn-th best: nothing to do for the others
    if (g.flag != nullptr && g.flag[blockIdx.x] != g.flag_value) return;
    topk_merge_row(g, blockIdx.x, sh, second);
}

hipError_t launch_topk_merge(const TopkMergeArgs &args, hipStream_t stream)
{
    if (args.rows == 0 || args.cols == 0) return hipSuccess;
    if (args.knn == 0 || args.knn > TOPK_MAX) return hipErrorInvalidValue;
    if (args.stride2 != 1 && !(args.stride2 == 2 && args.run_d1)) return hipErrorInvalidValue;
    if (args.seg_shift != 0u && args.seg_shift != 5u && args.seg_shift != 6u) return hipErrorInvalidValue;
    hipLaunchKernelGGL(topk_merge_kernel, dim3(args.rows), dim3(TOPK_THREADS), 0, stream, args);
    return hipGetLastError();
}

__global__ __launch_bounds__(TOPK_THREADS) void merge_states_kernel(const MergeStatesArgs g)
{
    __shared__ uint64_t items[MERGE_STATES_ITEMS];   // (sortable key << 32) | sample id
    __shared__ uint16_t slot[MERGE_STATES_ITEMS];    // where the item came from: state * knn + position
    const uint32_t row = blockIdx.x, tid = threadIdx.x, knn = g.knn;
    const uint32_t total = g.n_in * knn;
    uint32_t m = 1;
    while (m < total) m <<= 1;
    for (uint32_t x = tid; x < m; x += TOPK_THREADS) {
        if (x < total) {
            const uint32_t st = x / knn, pos = x - st * knn;
            const size_t o = (size_t)row * knn + pos;
            items[x] = ((uint64_t)g.key[st][o] << 32) | g.idx[st][o];
        } else {
            items[x] = ~0ull;
        }
        slot[x] = (uint16_t)x;
    }
    __syncthreads();
    for (uint32_t size = 2; size <= m; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t x = tid; x < m / 2; x += TOPK_THREADS) {
                const uint32_t lo = 2 * x - (x & (stride - 1));
                const uint32_t hi = lo + stride;
                const bool up = (lo & size) == 0;
                const uint64_t a = items[lo], b = items[hi];
                if ((a > b) == up) {
                    items[lo] = b;
                    items[hi] = a;
                    const uint16_t sa = slot[lo];
                    slot[lo] = slot[hi];
                    slot[hi] = sa;
                }
            }
            __syncthreads();
        }
    }
    for (uint32_t x = tid; x < knn; x += TOPK_THREADS) {
        const size_t o = (size_t)row * knn + x;
        g.out_key[o] = (uint32_t)(items[x] >> 32);
        g.out_idx[o] = (uint32_t)(items[x] & 0xFFFFFFFFu);
        if (g.out_d1) {
            const uint32_t st = slot[x] / knn, pos = slot[x] - st * knn;
            g.out_d1[o] = st < g.n_in ? g.d1[st][(size_t)row * knn + pos] : 0.0f;
        }
    }
}

hipError_t launch_merge_states(const MergeStatesArgs &args, hipStream_t stream)
{
    if (args.rows == 0) return hipSuccess;
    if (args.n_in == 0 || args.n_in > (uint32_t)MERGE_STATES_MAX || args.knn == 0 ||
        (uint64_t)args.n_in * args.knn > (uint64_t)MERGE_STATES_ITEMS) {
        return hipErrorInvalidValue;
    }
    hipLaunchKernelGGL(merge_states_kernel, dim3(args.rows), dim3(TOPK_THREADS), 0, stream, args);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Reference tie order (RefHeapArgs, kernels.h): std::collections::BinaryHeap replayed.
//
// The reference keeps a row's neighbours in a max-heap of at most knn items: candidate j (ascending) is pushed
// when the heap is not full or its key is STRICTLY below the heap's maximum, and the maximum is then popped
// (mod.rs:41-48); the row is printed in into_sorted_vec order.  Among equal keys, which candidates survive and in
// what order they are listed is therefore a function of the heap's history.  Rust's std is not part of the
// reference tree: the routines below restate the published algorithm of alloc::collections::binary_heap (push =
// sift_up; pop = move the last element to the root, sift_down_to_bottom, sift_up; into_sorted_vec = repeated
// swap(0, end) + sift_down_range); the test suite checks it against an independent CPU restatement.  Keys compare as f32
// (SparseJaccard / SparseCoreAcc order on the distance alone, distance_matrix.rs:214-262).
//
// One workgroup per row.  All threads scan the row's records in candidate order and compact, IN ORDER, the ones
// that could enter the heap as it stood at the last drain (not full, or key below its maximum: the maximum only
// falls, so this is a superset of what the heap accepts) into an LDS buffer; thread 0 drains the buffer through
// the exact push / pop sequence, which rejects the rest.  The heap lives in LDS (global memory beyond
// REFHEAP_LDS_MAX neighbours).
// ---------------------------------------------------------------------------
namespace {
struct RefHeap {
    float *key;       // [knn + 1]
    float *d1;        // [knn + 1] second values (or unused)
    uint32_t *id;     // [knn + 1]
    uint32_t len;
    bool two;

    struct Elt { float k, d; uint32_t i; };
    __device__ __forceinline__ Elt get(uint32_t p) const { return Elt{key[p], two ? d1[p] : 0.0f, id[p]}; }
    __device__ __forceinline__ void set(uint32_t p, const Elt &e) { key[p] = e.k; id[p] = e.i; if (two) d1[p] = e.d; }
    __device__ __forceinline__ void move(uint32_t dst, uint32_t src) { key[dst] = key[src]; id[dst] = id[src]; if (two) d1[dst] = d1[src]; }

    __device__ void sift_up(uint32_t start, uint32_t pos)
    {
        const Elt elt = get(pos);
        while (pos > start) {
            const uint32_t parent = (pos - 1u) / 2u;
            if (elt.k <= key[parent]) break;
            move(pos, parent);
            pos = parent;
        }
        set(pos, elt);
    }
    __device__ void sift_down_range(uint32_t pos, uint32_t end)
    {
        const Elt elt = get(pos);
        uint32_t child = 2u * pos + 1u;
        while (child <= (end >= 2u ? end - 2u : 0u)) {   // end.saturating_sub(2)
            child += key[child] <= key[child + 1u] ? 1u : 0u;
            if (elt.k >= key[child]) {
                set(pos, elt);
                return;
            }
            move(pos, child);
            pos = child;
            child = 2u * pos + 1u;
        }
        if (child == end - 1u && elt.k < key[child]) {
            move(pos, child);
            pos = child;
        }
        set(pos, elt);
    }
    __device__ void sift_down_to_bottom(uint32_t pos)
    {
        const uint32_t end = len, start = pos;
        const Elt elt = get(pos);
        uint32_t child = 2u * pos + 1u;
        while (child <= (end >= 2u ? end - 2u : 0u)) {
            child += key[child] <= key[child + 1u] ? 1u : 0u;
            move(pos, child);
            pos = child;
            child = 2u * pos + 1u;
        }
        if (child == end - 1u) {
            move(pos, child);
            pos = child;
        }
        set(pos, elt);
        sift_up(start, pos);
    }
    // mod.rs:41-48; -> whether the item entered the heap (the accept log of the decoupled column windows)
    __device__ bool push_heap(const Elt &item, uint32_t knn)
    {
        if (len < knn || item.k < key[0]) {
            set(len, item);
            ++len;
            sift_up(0u, len - 1u);
            if (len > knn) {   // pop the maximum
                --len;
                if (len > 0u) {
                    move(0u, len);
                    sift_down_to_bottom(0u);
                }
            }
            return true;
        }
        return false;
    }
    __device__ void into_sorted()
    {
        uint32_t end = len;
        while (end > 1u) {
            --end;
            const Elt a = get(0u), b = get(end);
            set(0u, b);
            set(end, a);
            sift_down_range(0u, end);
        }
    }
};
}  // namespace

// Shared scratch of the replay kernels: the candidates that passed the pre-filter since the last drain, in order.
constexpr uint32_t REFHEAP_CAP = 2048;
// ACCEPT LOG (RefHeapMergeArgs::log_*): every candidate a row's heap takes is appended to the row's log, in order -- what a
// heap that arrives LATER with the row's earlier candidates already in it has to be shown of this launch's candidates (it
// takes a subset: its maximum is never higher).  One thread writes (the one that pushes); len counts past cap (overflow).
struct RefHeapLog {
    float *rec = nullptr;
    uint32_t *id = nullptr;
    uint32_t cap = 0, len = 0, stride2 = 1;
    __device__ __forceinline__ void add(const RefHeap::Elt &e)
    {
        if (rec == nullptr) return;
        if (len < cap) {
            rec[(size_t)len * stride2] = e.k;
            if (stride2 == 2u) rec[(size_t)len * 2u + 1u] = e.d;
            id[len] = e.i;
        }
        ++len;
    }
};

struct RefHeapShared {
    float cand_key[REFHEAP_CAP], cand_d1[REFHEAP_CAP];
    uint32_t cand_id[REFHEAP_CAP];
    uint32_t wave_cnt[TOPK_THREADS / 64];
    uint32_t len, ncand;
    float thr;
};

// The records keys[0, cols) of one row (stride2 floats apiece) fed to the heap in position order.  id_of(q, id): is
// position q a candidate, and under which sample id.  `bits` (nullable): bit b clear = positions [64 b, 64 b + 64) hold
// nothing below the heap's maximum (set by the pair kernel against a maximum that may be stale, i.e. too high: never too
// few bits), such stretches are not read.  sh.len / sh.thr describe the heap on entry and on return (whole workgroup).
template <class IdOf>
__device__ __forceinline__ void refheap_feed(RefHeap &h, RefHeapShared &sh, const float *keys, uint32_t stride2, uint32_t cols,
                                             uint32_t knn, const IdOf &id_of, const uint32_t *bits, uint32_t seg_shift = 6u,
                                             RefHeapLog *alog = nullptr)
{
    constexpr uint32_t UNROLL = 4;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    auto drain = [&]() {   // thread 0: the buffered candidates through the exact push / pop sequence, in order
        if (tid == 0) {
            h.len = sh.len;
            const uint32_t m = sh.ncand;
            for (uint32_t c = 0; c < m; ++c) {
                const RefHeap::Elt e{sh.cand_key[c], sh.cand_d1[c], sh.cand_id[c]};
                if (h.push_heap(e, knn) && alog != nullptr) alog->add(e);
            }
            sh.len = h.len;
            sh.ncand = 0;
            sh.thr = h.len < knn ? __builtin_inff() : h.key[0];
        }
        __syncthreads();
    };
    for (uint32_t q0 = 0; q0 < cols; q0 += TOPK_THREADS * UNROLL) {
        if (bits != nullptr) {
            const uint32_t b_lo = q0 >> seg_shift, b_hi = (min(q0 + TOPK_THREADS * UNROLL, cols) - 1u) >> seg_shift;
            bool marked = false;
            for (uint32_t w = b_lo >> 5; w <= (b_hi >> 5); ++w) {
                uint32_t word = bits[w];
                if (w == (b_lo >> 5)) word &= ~0u << (b_lo & 31u);
                if (w == (b_hi >> 5) && (b_hi & 31u) != 31u) word &= (2u << (b_hi & 31u)) - 1u;
                marked |= word != 0u;
            }
            if (!marked) continue;   // (workgroup-uniform)
        }
        float k[UNROLL], d[UNROLL];
#pragma unroll
        for (uint32_t j = 0; j < UNROLL; ++j) {   // unconditional (clamped) loads, all in flight together
            const uint32_t q = min(q0 + j * TOPK_THREADS + tid, cols - 1u);
            k[j] = __builtin_nontemporal_load(&keys[(size_t)q * stride2]);
            d[j] = stride2 == 2u ? __builtin_nontemporal_load(&keys[(size_t)q * 2u + 1u]) : 0.0f;
        }
        const float thr = sh.thr;
        const bool open = sh.len < knn;   // not full: everything is pushed
        bool take[UNROLL];
        uint32_t id[UNROLL];
        int any = 0;
#pragma unroll
        for (uint32_t j = 0; j < UNROLL; ++j) {
            const uint32_t q = q0 + j * TOPK_THREADS + tid;
            id[j] = 0u;
            // (an unmarked stretch is not a candidate: it may not have been written at all -- a pruned tile)
            const uint32_t sb = q >> seg_shift;
            const bool seg_ok = bits == nullptr || (q < cols && ((bits[sb >> 5] >> (sb & 31u)) & 1u) != 0u);
            take[j] = q < cols && seg_ok && id_of(q, id[j]) && (open || k[j] < thr);
            any |= take[j] ? 1 : 0;
        }
        if (!__syncthreads_or(any)) continue;   // (the common case once the heap has settled)
#pragma unroll
        for (uint32_t j = 0; j < UNROLL; ++j) {
            // ordered compaction: position order = candidate order
            const uint64_t votes = __ballot(take[j]);
            if (lane == 0) sh.wave_cnt[wave] = (uint32_t)__popcll(votes);
            __syncthreads();
            uint32_t before = sh.ncand + (uint32_t)__popcll(votes & ((1ull << lane) - 1ull));
            uint32_t total = 0;
            for (uint32_t w = 0; w < TOPK_THREADS / 64; ++w) {
                if (w < wave) before += sh.wave_cnt[w];
                total += sh.wave_cnt[w];
            }
            if (take[j]) {
                sh.cand_key[before] = k[j];
                sh.cand_d1[before] = d[j];
                sh.cand_id[before] = id[j];
            }
            __syncthreads();
            if (tid == 0) sh.ncand += total;
            __syncthreads();
            if (sh.ncand + TOPK_THREADS > REFHEAP_CAP || open) {
                // (while the heap fills, drain after every round: the threshold must exist before more is buffered)
                drain();
                if (open) {   // the threshold may have appeared: re-test what this thread still holds
                    const float t2 = sh.thr;
                    const bool open2 = sh.len < knn;
#pragma unroll
                    for (uint32_t jj = 0; jj < UNROLL; ++jj) {
                        if (jj > j) take[jj] = take[jj] && (open2 || k[jj] < t2);
                    }
                }
            }
        }
    }
    drain();
}

// One-shot form: a whole row of a dense band, or (row_offsets != null) a ragged candidate row whose position p stands for
// sample col_ids[row_offsets[row] + p] -- the candidates in the order they are LISTED (the reference pushes in the order
// Inverted::any_shared_bins returns them, mod.rs:459-487) -- padded with (row, 1.0) behind fewer than knn candidates
// (mod.rs:535-546).
__global__ __launch_bounds__(TOPK_THREADS) void topk_refheap_kernel(const RefHeapArgs g)
{
    __shared__ float lds_heap[3 * (REFHEAP_LDS_MAX + 1)];
    __shared__ RefHeapShared sh;
    const uint32_t row = blockIdx.x + g.first_row, tid = threadIdx.x;
    const uint32_t knn = g.knn, stride2 = g.stride2;
    const bool ragged = g.row_offsets != nullptr;
    const uint64_t row_base = ragged ? g.row_offsets[row] : 0ull;
    const uint32_t cols = ragged ? (uint32_t)(g.row_offsets[row + 1] - row_base) : g.cols;
    const uint32_t self_id = g.self_id_base == 0xFFFFFFFFu ? 0xFFFFFFFFu : g.self_id_base + row;
    const float *keys = ragged ? g.keys + row_base * stride2 : g.keys + (size_t)row * g.key_stride;
    const uint32_t *ids = ragged ? g.col_ids + row_base : nullptr;
    float *base = g.heap_scratch ? g.heap_scratch + (size_t)blockIdx.x * 3u * (knn + 1u) : lds_heap;
    RefHeap h{base, base + (knn + 1u), reinterpret_cast<uint32_t *>(base + 2u * (knn + 1u)), 0u, stride2 == 2u};
    if (tid == 0) {
        sh.len = 0;
        sh.ncand = 0;
        sh.thr = __builtin_inff();
    }
    __syncthreads();
    if (cols != 0u) {
        refheap_feed(h, sh, keys, stride2, cols, knn,
                     [&](uint32_t q, uint32_t &id) {
                         id = ids ? ids[q] : q;
                         return q != self_id || ids != nullptr;
                     },
                     nullptr);
    }
    if (tid == 0) {
        h.len = sh.len;
        h.into_sorted();
    }
    __syncthreads();
    const uint32_t len = sh.len;
    for (uint32_t x = tid; x < knn; x += TOPK_THREADS) {
        const size_t o = (size_t)row * knn + x;
        if (x < len) {
            g.out_idx[o] = h.id[x];
            g.out_d0[o] = g.ani_undo ? 1.0f - h.key[x] : h.key[x];
            if (stride2 == 2u && g.out_d1) g.out_d1[o] = h.d1[x];
        } else if (ragged) {
            g.out_idx[o] = row;
            g.out_d0[o] = 1.0f;
        }
    }
}

constexpr uint32_t REFHEAP_WAVE_KNN_ONE_SHOT = 256;   // (= REFHEAP_WAVE_KNN below)
__global__ void topk_refheap_wave_kernel(const RefHeapArgs g);

hipError_t launch_topk_refheap(const RefHeapArgs &args, hipStream_t stream)
{
    if (args.rows == 0 || (args.cols == 0 && args.row_offsets == nullptr)) return hipSuccess;
    if (args.knn == 0 || (args.knn > REFHEAP_LDS_MAX && args.heap_scratch == nullptr)) return hipErrorInvalidValue;
    if (args.stride2 != 1 && args.stride2 != 2) return hipErrorInvalidValue;
    if (args.row_offsets != nullptr && (args.col_ids == nullptr || args.stride2 != 1)) return hipErrorInvalidValue;
    if (args.knn <= REFHEAP_WAVE_KNN_ONE_SHOT && !args.force_workgroup_form) {
        hipLaunchKernelGGL(topk_refheap_wave_kernel, dim3((args.rows + 3u) / 4u), dim3(256), 0, stream, args);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(topk_refheap_kernel, dim3(args.rows), dim3(TOPK_THREADS), 0, stream, args);
    return hipGetLastError();
}

// Resumable form (the one-evaluation self kNN in the reference's tie order): a row's heap lives in global memory
// between launches -- RefHeapMergeArgs -- and each launch continues the replay with the row's next batch of candidates.
// The drivers feed a row its candidates in ascending sample id over the sequence of launches, which is the order the
// reference pushes them in (mod.rs:156-181), so the heap goes through the same states.
__global__ __launch_bounds__(TOPK_THREADS) void refheap_merge_kernel(const RefHeapMergeArgs g)
{
    __shared__ float lds_heap[3 * (REFHEAP_LDS_MAX + 1)];
    __shared__ RefHeapShared sh;
    const uint32_t row = blockIdx.x, tid = threadIdx.x;
    if (g.flag != nullptr && g.flag[row] != g.flag_value) return;
    const uint32_t knn = g.knn, stride2 = g.stride2;
    const size_t srow = (size_t)(g.state_row_base + row);
    const uint32_t self_id = g.self_id_base == 0xFFFFFFFFu ? 0xFFFFFFFFu : g.self_id_base + row;
    RefHeap h{lds_heap, lds_heap + (knn + 1u), reinterpret_cast<uint32_t *>(lds_heap + 2u * (knn + 1u)), 0u, stride2 == 2u};
    const uint32_t len0 = g.h_len[srow];
    for (uint32_t x = tid; x < len0; x += TOPK_THREADS) {
        h.key[x] = g.h_key[srow * knn + x];
        h.id[x] = g.h_id[srow * knn + x];
        if (stride2 == 2u) h.d1[x] = g.h_d1[srow * knn + x];
    }
    if (tid == 0) {
        sh.len = len0;
        sh.ncand = 0;
    }
    __syncthreads();
    if (tid == 0) sh.thr = len0 < knn ? __builtin_inff() : h.key[0];
    __syncthreads();
    RefHeapLog alog;   // (thread 0's copy is the one that counts)
    if (g.log_rec != nullptr) {
        alog.rec = g.log_rec + srow * g.log_cap * stride2;
        alog.id = g.log_id + srow * g.log_cap;
        alog.cap = g.log_cap;
        alog.len = g.log_len[srow];
        alog.stride2 = stride2;
    }
    const uint32_t cols = g.row_cols != nullptr ? min(g.cols, g.row_cols[row]) : g.cols;
    const uint32_t *ids = g.cand_ids != nullptr ? g.cand_ids + (size_t)row * g.cols : nullptr;
    refheap_feed(h, sh, g.keys + (size_t)row * g.key_stride, stride2, cols, knn,
                 [&](uint32_t q, uint32_t &id) {
                     id = ids != nullptr ? ids[q] : g.id_base + q;
                     return id >= g.skip_below && id != self_id;
                 },
                 g.seg_bits ? g.seg_bits + (size_t)row * g.seg_bits_stride : nullptr, g.seg_shift != 0u ? g.seg_shift : 6u, &alog);
    if (tid == 0 && g.log_rec != nullptr) g.log_len[srow] = alog.len;
    const uint32_t len = sh.len;
    for (uint32_t x = tid; x < len; x += TOPK_THREADS) {
        g.h_key[srow * knn + x] = h.key[x];
        g.h_id[srow * knn + x] = h.id[x];
        if (stride2 == 2u) g.h_d1[srow * knn + x] = h.d1[x];
    }
    if (tid == 0) {
        g.h_len[srow] = len;
        g.thr[srow] = len < knn ? 0xFFFFFFFFu : sortable_bits(h.key[0]);
    }
}

// The same with ONE WAVE per row (knn <= REFHEAP_WAVE_KNN): what the one-evaluation driver's merges look like at scale is
// a million rows per band, each fed a few hundred to a few thousand records of which a handful enter the heap -- a
// 256-thread workgroup per row spends its time in barriers and in the serial drain of one thread while three rows fit
// a CU.  Here a workgroup is 4 independent rows: no barrier anywhere (a wave's LDS operations execute in order), the heap
// and the candidate buffer of a row take 4.6 KB of LDS, and 24-32 rows are in flight per CU to hide each other's drains.
constexpr uint32_t REFHEAP_WAVE_KNN = 256, REFHEAP_WAVE_CAND = 128;
static_assert(REFHEAP_WAVE_KNN == REFHEAP_WAVE_KNN_ONE_SHOT, "one constant");

__global__ __launch_bounds__(256) void refheap_merge_wave_kernel(const RefHeapMergeArgs g)
{
    __shared__ float heap_mem[4][3 * (REFHEAP_WAVE_KNN + 1)];
    __shared__ float cand_key[4][REFHEAP_WAVE_CAND], cand_d1[4][REFHEAP_WAVE_CAND];
    __shared__ uint32_t cand_id[4][REFHEAP_WAVE_CAND];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t row = blockIdx.x * 4u + wave;
    if (row >= g.rows) return;
    if (g.flag != nullptr && g.flag[row] != g.flag_value) return;
    const uint32_t knn = g.knn, stride2 = g.stride2;
    const uint32_t cols = g.row_cols != nullptr ? min(g.cols, g.row_cols[row]) : g.cols;
    const uint32_t *ids = g.cand_ids != nullptr ? g.cand_ids + (size_t)row * g.cols : nullptr;
    const size_t srow = (size_t)(g.state_row_base + row);
    const uint32_t self_id = g.self_id_base == 0xFFFFFFFFu ? 0xFFFFFFFFu : g.self_id_base + row;
    RefHeapLog alog;   // (lane 0's copy is the one that counts)
    if (g.log_rec != nullptr) {
        alog.rec = g.log_rec + srow * g.log_cap * stride2;
        alog.id = g.log_id + srow * g.log_cap;
        alog.cap = g.log_cap;
        alog.len = g.log_len[srow];
        alog.stride2 = stride2;
    }
    float *hm = heap_mem[wave];
    RefHeap h{hm, hm + (knn + 1u), reinterpret_cast<uint32_t *>(hm + 2u * (knn + 1u)), 0u, stride2 == 2u};
    auto wave_sync = [] {   // orders this wave's LDS traffic as the program states it
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    uint32_t len = g.h_len[srow];
    for (uint32_t x = lane; x < len; x += 64u) {
        h.key[x] = g.h_key[srow * knn + x];
        h.id[x] = g.h_id[srow * knn + x];
        if (stride2 == 2u) h.d1[x] = g.h_d1[srow * knn + x];
    }
    wave_sync();
    float thr = len < knn ? __builtin_inff() : h.key[0];
    uint32_t ncand = 0;
    bool dirty = false;
    auto drain = [&]() {   // lane 0: the buffered candidates through the exact push / pop sequence, in order
        wave_sync();
        if (lane == 0u) {
            h.len = len;
            for (uint32_t c = 0; c < ncand; ++c) {
                const RefHeap::Elt e{cand_key[wave][c], cand_d1[wave][c], cand_id[wave][c]};
                if (h.push_heap(e, knn)) alog.add(e);
            }
            len = h.len;
            thr = h.len < knn ? __builtin_inff() : h.key[0];
        }
        wave_sync();
        len = __shfl(len, 0);
        thr = __shfl(thr, 0);
        dirty = dirty || ncand != 0u;
        ncand = 0;
    };
    const float *keys = g.keys + (size_t)row * g.key_stride;
    const uint32_t *bits = g.seg_bits ? g.seg_bits + (size_t)row * g.seg_bits_stride : nullptr;
    constexpr uint32_t UNROLL = 4;
    const uint32_t seg_shift = g.seg_shift != 0u ? g.seg_shift : 6u;   // 6: one bit per 64 records, 5: per 32
    for (uint32_t q0 = 0; q0 < cols; q0 += 64u * UNROLL) {
        uint32_t mask = (1u << UNROLL) - 1u;   // which 64-record blocks of this trip are read
        uint32_t lane_ok = mask;               // ... and, per block, whether THIS lane's record is in a marked stretch
        if (bits != nullptr) {
            // (64 * UNROLL = 256 records = bits b .. b + 3 (or b .. b + 7 at 32 records per bit), inside one word)
            const uint32_t b = q0 >> seg_shift;
            const uint32_t w = bits[b >> 5] >> (b & 31u);
            if (seg_shift == 6u) {
                mask &= w;
                lane_ok = mask;
            } else {
                const uint32_t any2 = (w | (w >> 1)) & 0x55u;            // block j: bits 2 j, 2 j + 1
                mask &= (any2 & 1u) | ((any2 >> 1) & 2u) | ((any2 >> 2) & 4u) | ((any2 >> 3) & 8u);
                const uint32_t mine = w >> (lane >> 5);                  // this lane's half of each block
                lane_ok = (mine & 1u) | ((mine >> 1) & 2u) | ((mine >> 2) & 4u) | ((mine >> 3) & 8u);
            }
            if (mask == 0u) continue;
        }
        float k[UNROLL], d[UNROLL];
#pragma unroll
        for (uint32_t j = 0; j < UNROLL; ++j) {
            const uint32_t q = min(q0 + j * 64u + lane, cols - 1u);
            k[j] = (mask >> j) & 1u ? __builtin_nontemporal_load(&keys[(size_t)q * stride2]) : 0.0f;
            d[j] = (stride2 == 2u && ((mask >> j) & 1u)) ? __builtin_nontemporal_load(&keys[(size_t)q * 2u + 1u]) : 0.0f;
        }
#pragma unroll
        for (uint32_t j = 0; j < UNROLL; ++j) {
            if (!((mask >> j) & 1u)) continue;
            const uint32_t q = q0 + j * 64u + lane;
            const uint32_t id = ids != nullptr ? ids[min(q, cols - 1u)] : g.id_base + q;
            const bool open = len < knn;
            const bool take = q < cols && ((lane_ok >> j) & 1u) != 0u && id >= g.skip_below && id != self_id && (open || k[j] < thr);
            const uint64_t votes = __ballot(take);
            if (votes == 0ull) continue;
            if (take) {
                const uint32_t pos = ncand + (uint32_t)__popcll(votes & ((1ull << lane) - 1ull));
                cand_key[wave][pos] = k[j];
                cand_d1[wave][pos] = d[j];
                cand_id[wave][pos] = id;
            }
            ncand += (uint32_t)__popcll(votes);
            // (while the heap fills, drain at once: the threshold must exist before more is buffered)
            if (open || ncand + 64u > REFHEAP_WAVE_CAND) drain();
        }
    }
    if (ncand != 0u) drain();
    if (lane == 0u && g.log_rec != nullptr) g.log_len[srow] = alog.len;
    if (dirty) {
        for (uint32_t x = lane; x < len; x += 64u) {
            g.h_key[srow * knn + x] = h.key[x];
            g.h_id[srow * knn + x] = h.id[x];
            if (stride2 == 2u) g.h_d1[srow * knn + x] = h.d1[x];
        }
        if (lane == 0u) {
            g.h_len[srow] = len;
            g.thr[srow] = len < knn ? 0xFFFFFFFFu : sortable_bits(h.key[0]);
        }
    }
}

// The ONE-SHOT replay (topk_refheap_kernel's job: a whole dense row, or a ragged candidate row of the precluster mode) with
// one wave per row, for the same reason: the precluster call at 400 000 rows x ~800 listed candidates spent 64 ms in the
// one-workgroup-per-row form -- as long as the distances themselves -- against 7 ms for the canonical radix select.
__global__ __launch_bounds__(256) void topk_refheap_wave_kernel(const RefHeapArgs g)
{
    __shared__ float heap_mem[4][3 * (REFHEAP_WAVE_KNN + 1)];
    __shared__ float cand_key[4][REFHEAP_WAVE_CAND], cand_d1[4][REFHEAP_WAVE_CAND];
    __shared__ uint32_t cand_id[4][REFHEAP_WAVE_CAND];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (blockIdx.x * 4u + wave >= g.rows) return;
    const uint32_t row = blockIdx.x * 4u + wave + g.first_row;
    const uint32_t knn = g.knn, stride2 = g.stride2;
    const bool ragged = g.row_offsets != nullptr;
    const uint64_t row_base = ragged ? g.row_offsets[row] : 0ull;
    const uint32_t cols = ragged ? (uint32_t)(g.row_offsets[row + 1] - row_base) : g.cols;
    const uint32_t self_id = g.self_id_base == 0xFFFFFFFFu ? 0xFFFFFFFFu : g.self_id_base + row;
    const float *keys = ragged ? g.keys + row_base * stride2 : g.keys + (size_t)row * g.key_stride;
    const uint32_t *ids = ragged ? g.col_ids + row_base : nullptr;
    float *hm = heap_mem[wave];
    RefHeap h{hm, hm + (knn + 1u), reinterpret_cast<uint32_t *>(hm + 2u * (knn + 1u)), 0u, stride2 == 2u};
    auto wave_sync = [] {   // orders this wave's LDS traffic as the program states it
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    uint32_t len = 0, ncand = 0;
    float thr = __builtin_inff();
    auto drain = [&]() {   // lane 0: the buffered candidates through the exact push / pop sequence, in order
        wave_sync();
        if (lane == 0u) {
            h.len = len;
            for (uint32_t c = 0; c < ncand; ++c) (void)h.push_heap(RefHeap::Elt{cand_key[wave][c], cand_d1[wave][c], cand_id[wave][c]}, knn);
            len = h.len;
            thr = h.len < knn ? __builtin_inff() : h.key[0];
        }
        wave_sync();
        len = __shfl(len, 0);
        thr = __shfl(thr, 0);
        ncand = 0;
    };
    constexpr uint32_t UNROLL = 4;
    for (uint32_t q0 = 0; q0 < cols; q0 += 64u * UNROLL) {
        float k[UNROLL], d[UNROLL];
        uint32_t id[UNROLL];
#pragma unroll
        for (uint32_t j = 0; j < UNROLL; ++j) {   // unconditional (clamped) loads, all in flight together
            const uint32_t q = min(q0 + j * 64u + lane, cols - 1u);
            k[j] = __builtin_nontemporal_load(&keys[(size_t)q * stride2]);
            d[j] = stride2 == 2u ? __builtin_nontemporal_load(&keys[(size_t)q * 2u + 1u]) : 0.0f;
            id[j] = ids ? ids[q] : q;
        }
#pragma unroll
        for (uint32_t j = 0; j < UNROLL; ++j) {
            const uint32_t q = q0 + j * 64u + lane;
            const bool open = len < knn;
            const bool take = q < cols && (ids != nullptr || q != self_id) && (open || k[j] < thr);
            const uint64_t votes = __ballot(take);
            if (votes == 0ull) continue;
            if (take) {
                const uint32_t pos = ncand + (uint32_t)__popcll(votes & ((1ull << lane) - 1ull));
                cand_key[wave][pos] = k[j];
                cand_d1[wave][pos] = d[j];
                cand_id[wave][pos] = id[j];
            }
            ncand += (uint32_t)__popcll(votes);
            // (while the heap fills, drain at once: the threshold must exist before more is buffered)
            if (open || ncand + 64u > REFHEAP_WAVE_CAND) drain();
        }
    }
    if (ncand != 0u) drain();
    wave_sync();
    if (lane == 0u) {
        h.len = len;
        h.into_sorted();
    }
    wave_sync();
    for (uint32_t x = lane; x < knn; x += 64u) {
        const size_t o = (size_t)row * knn + x;
        if (x < len) {
            g.out_idx[o] = h.id[x];
            g.out_d0[o] = g.ani_undo ? 1.0f - h.key[x] : h.key[x];
            if (stride2 == 2u && g.out_d1) g.out_d1[o] = h.d1[x];
        } else if (ragged) {
            g.out_idx[o] = row;
            g.out_d0[o] = 1.0f;
        }
    }
}

hipError_t launch_refheap_merge(const RefHeapMergeArgs &args, hipStream_t stream)
{
    if (args.rows == 0 || args.cols == 0) return hipSuccess;
    if (args.knn == 0 || args.knn > REFHEAP_LDS_MAX) return hipErrorInvalidValue;
    if (args.stride2 != 1 && !(args.stride2 == 2 && args.h_d1)) return hipErrorInvalidValue;
    if (args.seg_shift != 0u && args.seg_shift != 5u && args.seg_shift != 6u) return hipErrorInvalidValue;
    if (args.knn <= REFHEAP_WAVE_KNN && !args.force_workgroup_form) {
        hipLaunchKernelGGL(refheap_merge_wave_kernel, dim3((args.rows + 3u) / 4u), dim3(256), 0, stream, args);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(refheap_merge_kernel, dim3(args.rows), dim3(TOPK_THREADS), 0, stream, args);
    return hipGetLastError();
}

// ... and its last step: into_sorted_vec of every row's heap -> the public output form.
__global__ __launch_bounds__(64) void refheap_finalize_kernel(const float *h_key, const uint32_t *h_id, const float *h_d1, const uint32_t *h_len,
                                                              uint32_t rows, uint32_t knn, int ani_undo, uint64_t *out_idx, float *out_d0,
                                                              float *out_d1)
{
    __shared__ float lds_heap[3 * (REFHEAP_LDS_MAX + 1)];
    const uint32_t row = blockIdx.x, tid = threadIdx.x;
    RefHeap h{lds_heap, lds_heap + (knn + 1u), reinterpret_cast<uint32_t *>(lds_heap + 2u * (knn + 1u)), 0u, h_d1 != nullptr};
    const uint32_t len = h_len[row];
    for (uint32_t x = tid; x < len; x += 64u) {
        h.key[x] = h_key[(size_t)row * knn + x];
        h.id[x] = h_id[(size_t)row * knn + x];
        if (h_d1) h.d1[x] = h_d1[(size_t)row * knn + x];
    }
    __syncthreads();
    if (tid == 0) {
        h.len = len;
        h.into_sorted();
    }
    __syncthreads();
    for (uint32_t x = tid; x < len; x += 64u) {
        const size_t o = (size_t)row * knn + x;
        out_idx[o] = h.id[x];
        out_d0[o] = ani_undo ? 1.0f - h.key[x] : h.key[x];
        if (h_d1 && out_d1) out_d1[o] = h.d1[x];
    }
}

hipError_t launch_refheap_finalize(const float *h_key, const uint32_t *h_id, const float *h_d1, const uint32_t *h_len, uint32_t rows,
                                   uint32_t knn, int ani_undo, uint64_t *out_idx, float *out_d0, float *out_d1, hipStream_t stream)
{
    if (rows == 0) return hipSuccess;
    if (knn == 0 || knn > REFHEAP_LDS_MAX || (h_d1 && !out_d1)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(refheap_finalize_kernel, dim3(rows), dim3(64), 0, stream, h_key, h_id, h_d1, h_len, rows, knn, ani_undo,
                       out_idx, out_d0, out_d1);
    return hipGetLastError();
}

// Tile-pruning thresholds of the symmetric self kNN (PairArgs::prune_q_*): one thread per sample.  The launch's key is
// dtab[bin matches] (f32, non-increasing in the matches, so non-decreasing in the mismatch count m = total_bins - matches):
// allow = the largest m whose key is STRICTLY below the sample's knn-th best (records at or above it never enter a list:
// push_heap's strict `<`, mod.rs:42; canonical ties: a later id loses), found by bisection; q = floor(allow / 4) + 1 is the
// count EACH of the 4 waves of a workgroup must reach on its own chunks for the pair's total to exceed allow.
__global__ void prune_thresholds_kernel(const uint32_t *thr, uint32_t thr_stride, uint32_t n, const float *dtab, uint32_t total_bins,
                                        uint32_t *q)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t t = thr[(size_t)i * thr_stride];
    // below(m): key(m) < t.  Monotone: true for m = 0 .. allow, false beyond.
    auto below = [&](uint32_t m) { return sortable_bits(dtab[total_bins - m]) < t; };
    if (!below(0u)) {
        q[i] = 0u;   // not even an identical sketch would enter: every pair of this sample is hopeless
        return;
    }
    uint32_t lo = 0u, hi = total_bins + 1u;   // below(lo) holds, below(hi) does not (hi = total_bins + 1: out of range, never asked)
    while (hi - lo > 1u) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (below(mid)) lo = mid; else hi = mid;
    }
    q[i] = lo / 4u + 1u;
}

hipError_t launch_prune_thresholds(const uint32_t *thr, uint32_t thr_stride, uint32_t n, const float *dtab, uint32_t total_bins,
                                   uint32_t *q, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    if (!thr || !dtab || !q || thr_stride == 0u) return hipErrorInvalidValue;
    hipLaunchKernelGGL(prune_thresholds_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, thr, thr_stride, n, dtab, total_bins, q);
    return hipGetLastError();
}

__global__ void topk_finalize_kernel(const uint32_t *run_key, const uint32_t *run_idx, const float *run_d1,
                                     uint64_t items, int ani_undo, uint64_t *out_idx, float *out_d0, float *out_d1)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; x < items; x += stride) {
        const float key = from_sortable_bits(run_key[x]);
        out_idx[x] = run_idx[x];
        out_d0[x] = ani_undo ? 1.0f - key : key;
        if (run_d1) out_d1[x] = run_d1[x];
    }
}

hipError_t launch_topk_finalize(const uint32_t *run_key, const uint32_t *run_idx, const float *run_d1, uint64_t items,
                                int ani_undo, uint64_t *out_idx, float *out_d0, float *out_d1, hipStream_t stream)
{
    if (items == 0) return hipSuccess;
    if (run_d1 && !out_d1) return hipErrorInvalidValue;
    const uint64_t blocks = std::min<uint64_t>((items + 255) / 256, 65536);
    hipLaunchKernelGGL(topk_finalize_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, run_key, run_idx, run_d1,
                       items, ani_undo, out_idx, out_d0, out_d1);
    return hipGetLastError();
}

}  // namespace skl
