// sketch_kernel.hip -- the sketching hot loop on the GPU:
// canonical ntHash of every valid k-mer, `% SIGN_MOD`, bin minimum -- i.e.
// Sketch::get_signs_no_densify (src/sketch/mod.rs:156-176) over the NtHashIterator
// (src/hashing/nthash_iterator.rs:325-523) for DNA assemblies, all samples and all k-mer
// lengths of a batch in one launch.  Densification and the 14-plane transpose stay on the
// host (csrc/host/sketch.cpp): they touch num_bins words per (sample, k), not the genome.
//
// One thread = SPAN consecutive window starts of one sample: it finds the first break
// (N / record boundary) after its first position by binary search, then walks, rolling the
// forward and reverse-complement hashes (one split rotation + two XORs each) while windows stay
// valid and re-seeding them (k steps) after a break.  Each window's sign goes to its bin with
// a 64-bit atomicMin, issued only when it would lower the bin (after the first few hundred
// windows of a bin almost none does).
#include "kernels.h"

namespace skl {

namespace {
__device__ __forceinline__ uint64_t rotl1(uint64_t v) { return (v << 1) | (v >> 63); }
__device__ __forceinline__ uint64_t rotr1(uint64_t v) { return (v >> 1) | (v << 63); }
// swapbits033, src/hashing/mod.rs:99-103
__device__ __forceinline__ uint64_t swapbits033(uint64_t v)
{
    const uint64_t x = (v ^ (v >> 33)) & 1ull;
    return v ^ (x | (x << 33));
}
__device__ __forceinline__ uint64_t srol(uint64_t v) { return swapbits033(rotl1(v)); }
__device__ __forceinline__ uint64_t sror(uint64_t v) { return rotr1(swapbits033(v)); }

// src/hashing/nthash_tables.rs:4-16 (index = 2-bit base code)
__device__ __forceinline__ uint64_t hash_fwd(uint32_t c)
{
    return c == 0 ? 0x3c8bfbb395c60474ull : c == 1 ? 0x3193c18562a02b4cull : c == 2 ? 0x295549f54be24456ull : 0x20323ed082572324ull;
}
__device__ __forceinline__ uint64_t hash_rc(uint32_t c) { return hash_fwd(c ^ 2u); }   // complement = code ^ 2

constexpr uint64_t SIGN_MOD_DEV = (1ull << 61) - 1;   // src/sketch/mod.rs:36
__device__ __forceinline__ uint64_t mod_sign(uint64_t h)
{
    uint64_t r = (h & SIGN_MOD_DEV) + (h >> 61);        // 2^61 = 1 (mod 2^61 - 1)
    return r >= SIGN_MOD_DEV ? r - SIGN_MOD_DEV : r;
}
}  // namespace

constexpr int SKETCH_SPAN = 256;   // window starts per thread

__global__ __launch_bounds__(256) void nthash_binmin_kernel(const SketchArgs g)
{
    if ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x >= g.n_spans) return;
    const uint64_t t = g.first_span + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // span -> sample: span_begin is the prefix sum of spans per sample
    uint32_t lo = 0, hi = g.n_samples;
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (g.span_begin[mid] <= t) lo = mid; else hi = mid;
    }
    const uint32_t sample = lo;
    const uint64_t code0 = g.code_begin[sample], n_codes = g.code_begin[sample + 1] - code0;
    const uint64_t *offs = g.offsets + g.offset_begin[sample];
    const uint32_t n_offs = (uint32_t)(g.offset_begin[sample + 1] - g.offset_begin[sample]);
    const uint32_t *pk = g.packed + g.word_begin[sample];
    auto codes = [&](uint64_t x) -> uint32_t { return (pk[x >> 4] >> ((uint32_t)(x & 15u) * 2u)) & 3u; };
    (void)code0;
    const uint64_t p0 = (t - g.span_begin[sample]) * SKETCH_SPAN;
    const uint64_t p1 = p0 + SKETCH_SPAN < n_codes ? p0 + SKETCH_SPAN : n_codes;

    for (uint32_t ki = 0; ki < g.nk; ++ki) {
        const uint32_t k = g.kmers[ki];
        // weight of the base that leaves / enters at distance k-1 (srol^(k-1) of its seed)
        uint64_t top_f[4], top_r[4];
#pragma unroll
        for (uint32_t b = 0; b < 4; ++b) {
            top_f[b] = g.top_f[ki * 4 + b];
            top_r[b] = g.top_r[ki * 4 + b];
        }
        uint64_t *bins = g.signs + ((uint64_t)sample * g.nk + ki) * g.num_bins;
        // first break strictly after p0
        uint32_t oi;
        {
            uint32_t a = 0, b = n_offs;   // first index with offs[idx] > p0
            while (a < b) {
                const uint32_t mid = (a + b) >> 1;
                if (offs[mid] > p0) b = mid; else a = mid + 1;
            }
            oi = a;
        }
        bool have = false;
        uint64_t fh = 0, rh = 0;
        for (uint64_t s = p0; s < p1; ++s) {
            while (oi < n_offs && offs[oi] <= s) ++oi;
            const uint64_t next_off = oi < n_offs ? offs[oi] : n_codes;
            if (s + k > next_off) {   // the window would span a break (next_iterator, :325-346)
                have = false;
                continue;
            }
            if (have) {
                const uint32_t old_b = codes(s - 1), new_b = codes(s + k - 1);
                fh = srol(fh ^ top_f[old_b]) ^ hash_fwd(new_b);
                if (g.rc) rh = sror(rh ^ hash_rc(old_b)) ^ top_r[new_b];
            } else {
                fh = 0;
                rh = 0;
                for (uint32_t i = 0; i < k; ++i) fh = srol(fh) ^ hash_fwd(codes(s + i));
                if (g.rc) {
                    for (uint32_t i = k; i-- > 0;) rh = srol(rh) ^ hash_rc(codes(s + i));
                }
                have = true;
            }
            const uint64_t h = g.rc ? (fh < rh ? fh : rh) : fh;      // nthash_iterator.rs:62-68
            const uint64_t sign = mod_sign(h);
            // bin = sign / bin_size: reciprocal estimate, then exact fix-up
            uint64_t bin = (uint64_t)((double)sign * g.inv_bin_size);
            if (bin >= g.num_bins) bin = g.num_bins - 1;
            while (bin * g.bin_size > sign) --bin;
            while ((bin + 1) * g.bin_size <= sign) ++bin;
            if (sign < bins[bin]) atomicMin((unsigned long long *)&bins[bin], (unsigned long long)sign);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// LDS-staged form (the default).  The form above reads two bytes per window straight from
// global memory, each lane from its own cache line, and checks the bin in global memory: ~190
// cache-line requests per wave and window, which is what it runs at.  Here a workgroup owns
// WG2 consecutive spans OF ONE SAMPLE (the host pads every sample's span count to a multiple of
// WG2), stages their bases once into LDS and walks them for every k-mer length; the bin minima of
// the current k-mer length live in LDS too (64-bit ds_min) and are flushed with one global
// atomicMin per touched bin.
//
// Round 4 (round 3's form: 140 VALU lane-instructions per window, 0.30 of the VALU issue peak, 42 % of the
// wave-cycles waiting at 2 waves per SIMD):
//   * the bases are staged PACKED, 16 two-bit codes per dword (row pitch 9 dwords: lanes that read the same
//     column of consecutive rows hit different banks): 18 KB instead of 34 KB per 256 threads, so a workgroup
//     of 512 threads + its 32 KB of bin minima fit a CU three times (6 waves per SIMD instead of 2);
//   * a thread keeps the codes that leave and enter its next 16 windows in two registers (one LDS dword
//     and one v_alignbit per stream and 16 windows) instead of two byte reads with their address
//     arithmetic per window;
//   * srol / sror are linear over XOR, so  fh' = srol(fh ^ top_f[old]) ^ seed_f[new] = srol(fh) ^ Tf[old][new]
//     and  rh' = sror(rh) ^ Tr[old][new]:  the four 4-way 64-bit selects per window become ONE 16-byte LDS
//     read from a 16-entry table per k-mer length;
//   * a window that spans a break (N, record end) is not hashed into a bin, but the hashes roll on through
//     it -- `codes` holds valid bases only, so the roll stays exact -- and nothing is re-seeded after a break
//     (one seed of k steps per thread and k-mer length);
//   * the bin of a sign comes from the reciprocal estimate and ONE 64-bit multiply (the estimate is off by at
//     most one: the two `while` loops of round 3 evaluated two).
// ---------------------------------------------------------------------------------------------

constexpr int SPAN2 = 128;                    // window starts per thread
constexpr int WG2 = 512;                      // threads (= spans) per workgroup
constexpr int PITCH2_DW = SPAN2 / 16 + 1;     // dwords per staged row (8 of codes + 1 of padding)
constexpr int ROWS2 = WG2 + 1;                // one more row: a window reaches k - 1 <= SPAN2 codes past its span
constexpr int LDS_BINS_MAX = 4096;

template <bool LDS_BINS>
__global__ __launch_bounds__(WG2) void nthash_binmin_lds_kernel(const SketchArgs g)
{
    __shared__ uint32_t staged[ROWS2 * PITCH2_DW];
    __shared__ uint4 tabs[16];                 // {Tf.lo, Tf.hi, Tr.lo, Tr.hi} by (old << 2) | new, for the current k-mer length
    __shared__ unsigned long long lbins[LDS_BINS ? LDS_BINS_MAX : 1];
    const uint32_t tid = threadIdx.x;
    const uint64_t t0 = g.first_span + (uint64_t)blockIdx.x * WG2;   // first span of this workgroup
    uint32_t lo = 0, hi = g.n_samples;
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (g.span_begin[mid] <= t0) lo = mid; else hi = mid;
    }
    const uint32_t sample = lo;
    const uint64_t code0 = g.code_begin[sample], n_codes = g.code_begin[sample + 1] - code0;
    const uint64_t *offs = g.offsets + g.offset_begin[sample];
    const uint32_t n_offs = (uint32_t)(g.offset_begin[sample + 1] - g.offset_begin[sample]);
    const uint32_t *pk = g.packed + g.word_begin[sample];
    (void)code0;
    const uint64_t wg0 = (t0 - g.span_begin[sample]) * SPAN2;   // first base of the workgroup (a multiple of 16: whole words)
    if (wg0 >= n_codes) return;                                // (padding spans of the sample)
    const uint64_t wg1 = wg0 + (uint64_t)ROWS2 * SPAN2 < n_codes ? wg0 + (uint64_t)ROWS2 * SPAN2 : n_codes;
    const uint32_t n_staged = (uint32_t)(wg1 - wg0);
    // stage: one packed dword (16 codes) per thread and trip, as the host packed them (round 5: the bases cross PCIe at 2
    // bits each; a sample's last word is zero-padded and words past it are not read)
    for (uint32_t d = tid; d < (uint32_t)ROWS2 * (SPAN2 / 16); d += WG2) {
        const uint32_t word = d * 16u < n_staged ? pk[(wg0 >> 4) + d] : 0u;
        staged[(d >> 3) * PITCH2_DW + (d & 7u)] = word;
    }
    if (LDS_BINS) {
        for (uint32_t b = tid; b < (uint32_t)g.num_bins; b += WG2) lbins[b] = ~0ull;
    }
    // flat dword m of the staged codes (dword m holds codes 16 m .. 16 m + 15 from wg0)
    auto dword_at = [&](uint32_t m) -> uint32_t { return staged[(m >> 3) * PITCH2_DW + (m & 7u)]; };
    auto code_at = [&](uint32_t x) -> uint32_t { return (dword_at(x >> 4) >> ((x & 15u) * 2u)) & 3u; };   // code x from wg0
    const uint64_t p0 = wg0 + (uint64_t)tid * SPAN2;
    const uint32_t x0 = tid * SPAN2;              // this thread's first code, from wg0
    const uint32_t last_bin = (uint32_t)(g.num_bins - 1u);

    for (uint32_t ki = 0; ki < g.nk; ++ki) {
        const uint32_t k = g.kmers[ki];
        __syncthreads();   // staged / lbins ready (first k); everyone done with the previous k's table
        if (tid < 16u) {
            const uint32_t old_b = tid >> 2, new_b = tid & 3u;
            const uint64_t tf = srol(g.top_f[ki * 4 + old_b]) ^ hash_fwd(new_b);
            const uint64_t tr = sror(hash_rc(old_b)) ^ g.top_r[ki * 4 + new_b];
            tabs[tid] = make_uint4((uint32_t)tf, (uint32_t)(tf >> 32), (uint32_t)tr, (uint32_t)(tr >> 32));
        }
        __syncthreads();
        uint64_t *bins = g.signs + ((uint64_t)sample * g.nk + ki) * g.num_bins;
        // windows of this thread: starts p0 .. p0 + n_win - 1 (a start needs k codes of the sample)
        const uint64_t last_start_excl = n_codes >= k ? n_codes - k + 1u : 0u;
        const uint32_t n_win = p0 < last_start_excl ? (uint32_t)(last_start_excl - p0 < SPAN2 ? last_start_excl - p0 : SPAN2) : 0u;
        if (n_win != 0u) {
            // first break strictly after p0; positions from here on are relative to p0 (32-bit)
            uint32_t oi;
            {
                uint32_t a = 0, b = n_offs;   // first index with offs[idx] > p0
                while (a < b) {
                    const uint32_t mid = (a + b) >> 1;
                    if (offs[mid] > p0) b = mid; else a = mid + 1;
                }
                oi = a;
            }
            constexpr uint32_t FAR = 0x7FFFFFFFu;
            auto rel = [&](uint32_t idx) -> uint32_t {
                if (idx >= n_offs) return FAR;
                const uint64_t d = offs[idx] - p0;
                return d < (uint64_t)FAR ? (uint32_t)d : FAR;
            };
            uint32_t next_rel = rel(oi);   // first break after window start j (relative): the window is hashed iff j + k <= next_rel
            // seed: the window at p0 (nthash_iterator.rs: forward fold, reverse-complement fold from the far end)
            uint64_t fh = 0, rh = 0;
            for (uint32_t i = 0; i < k; ++i) fh = srol(fh) ^ hash_fwd(code_at(x0 + i));
            if (g.rc) {
                for (uint32_t i = k; i-- > 0;) rh = srol(rh) ^ hash_rc(code_at(x0 + i));
            }
            const uint32_t a = k - 1u;                        // lag of the entering code
            const uint32_t da = a >> 4, pa = (a & 15u) * 2u;
            uint32_t d_prev = 0u;                             // dword x0/16 + q - 1 of the leaving stream (unused for q = 0)
            uint32_t e_prev = dword_at((x0 >> 4) + da);       // dword of the entering stream
            for (uint32_t q = 0; q < SPAN2 / 16 && q * 16u < n_win; ++q) {
                const uint32_t d_cur = dword_at((x0 >> 4) + q);
                // (k = 129: the entering stream starts exactly one row on and its last look-ahead dword is not used -- nor staged)
                const uint32_t e_cur = dword_at(min((x0 >> 4) + q + da + 1u, (uint32_t)(ROWS2 * (SPAN2 / 16)) - 1u));
                // 16 codes from index 16 q - 1 (leaving) and 16 q + k - 1 (entering) of this thread's row
                const uint32_t old_reg = __builtin_amdgcn_alignbit(d_cur, d_prev, 30);
                const uint32_t new_reg = pa ? __builtin_amdgcn_alignbit(e_cur, e_prev, pa) : e_prev;
                d_prev = d_cur;
                e_prev = e_cur;
#pragma unroll
                for (uint32_t jj = 0; jj < 16u; ++jj) {
                    const uint32_t j = q * 16u + jj;
                    if (j >= n_win) break;
                    if (j != 0u) {   // roll from window j - 1 to window j
                        const uint32_t idx = (((old_reg >> (2u * jj)) & 3u) << 2) | ((new_reg >> (2u * jj)) & 3u);
                        const uint4 t = tabs[idx];
                        fh = srol(fh) ^ (((uint64_t)t.y << 32) | t.x);
                        if (g.rc) rh = sror(rh) ^ (((uint64_t)t.w << 32) | t.z);
                    }
                    if (next_rel <= j) {   // (breaks are rare: a few per span at most)
                        while (oi < n_offs && offs[oi] <= p0 + j) ++oi;
                        next_rel = rel(oi);
                    }
                    if (j + k > next_rel) continue;   // the window spans a break (next_iterator, :325-346): not hashed into a bin
                    const uint64_t h = g.rc ? (fh < rh ? fh : rh) : fh;      // nthash_iterator.rs:62-68
                    const uint64_t sign = mod_sign(h);
                    // bin = sign / bin_size: the reciprocal estimate is off by at most one (relative error 2^-52 on a
                    // quotient below 2^32), one exact product settles it
                    uint32_t bin = (uint32_t)((double)sign * g.inv_bin_size);
                    if (bin > last_bin) bin = last_bin;
                    const uint64_t prod = (uint64_t)bin * g.bin_size;
                    if (prod > sign) --bin;
                    else if (sign - prod >= g.bin_size && bin < last_bin) ++bin;
                    if (LDS_BINS) {
                        if (sign < lbins[bin]) atomicMin(&lbins[bin], (unsigned long long)sign);
                    } else {
                        if (sign < bins[bin]) atomicMin((unsigned long long *)&bins[bin], (unsigned long long)sign);
                    }
                }
            }
        }
        if (LDS_BINS) {   // one global atomic per bin this workgroup touched
            __syncthreads();
            for (uint32_t b = tid; b < (uint32_t)g.num_bins; b += WG2) {
                const unsigned long long v = lbins[b];
                if (v != ~0ull) {
                    if (v < bins[b]) atomicMin((unsigned long long *)&bins[b], v);
                    lbins[b] = ~0ull;
                }
            }
        }
    }
}

hipError_t launch_sketch_signs(const SketchArgs &args, hipStream_t stream)
{
    if (args.n_spans == 0) return hipSuccess;
    const uint64_t wg = args.lds_form ? (uint64_t)WG2 : 256u;
    const uint64_t blocks = (args.n_spans + wg - 1) / wg;
    if (blocks >= (1ull << 31)) return hipErrorInvalidValue;
    if (args.lds_form) {
        if (args.num_bins <= (uint64_t)LDS_BINS_MAX) {
            hipLaunchKernelGGL(nthash_binmin_lds_kernel<true>, dim3((unsigned)blocks), dim3(WG2), 0, stream, args);
        } else {
            hipLaunchKernelGGL(nthash_binmin_lds_kernel<false>, dim3((unsigned)blocks), dim3(WG2), 0, stream, args);
        }
    } else {
        hipLaunchKernelGGL(nthash_binmin_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, args);
    }
    return hipGetLastError();
}

int sketch_span() { return SKETCH_SPAN; }
int sketch_span_lds() { return SPAN2; }
int sketch_wg_lds() { return WG2; }

}  // namespace skl
