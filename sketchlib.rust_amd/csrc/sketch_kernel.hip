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
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= g.n_spans) return;
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
    const uint8_t *codes = g.codes + code0;
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
                const uint32_t old_b = codes[s - 1], new_b = codes[s + k - 1];
                fh = srol(fh ^ top_f[old_b]) ^ hash_fwd(new_b);
                if (g.rc) rh = sror(rh ^ hash_rc(old_b)) ^ top_r[new_b];
            } else {
                fh = 0;
                rh = 0;
                for (uint32_t i = 0; i < k; ++i) fh = srol(fh) ^ hash_fwd(codes[s + i]);
                if (g.rc) {
                    for (uint32_t i = k; i-- > 0;) rh = srol(rh) ^ hash_rc(codes[s + i]);
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
// cache-line requests per wave and window, which is what it runs at.  Here a workgroup owns 256
// consecutive spans OF ONE SAMPLE (the host pads every sample's span count to a multiple of 256),
// stages their bases once into LDS (row pitch 132 bytes: lanes that read the same column of
// consecutive rows hit different banks) and walks them for every k-mer length; the bin minima of
// the current k-mer length live in LDS too (64-bit ds_min) and are flushed with one global
// atomicMin per touched bin.
// ---------------------------------------------------------------------------------------------

constexpr int SPAN2 = 128;                    // window starts per thread
constexpr int PITCH2 = SPAN2 + 4;             // bytes per staged row
constexpr int ROWS2 = 256 + 1;                // one more row: a window reaches k - 1 <= SPAN2 bytes past its span
constexpr int LDS_BINS_MAX = 4096;

template <bool LDS_BINS>
__global__ __launch_bounds__(256) void nthash_binmin_lds_kernel(const SketchArgs g)
{
    __shared__ uint8_t staged[ROWS2 * PITCH2];
    __shared__ unsigned long long lbins[LDS_BINS ? LDS_BINS_MAX : 1];
    const uint32_t tid = threadIdx.x;
    const uint64_t t0 = (uint64_t)blockIdx.x * 256u;   // first span of this workgroup
    uint32_t lo = 0, hi = g.n_samples;
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (g.span_begin[mid] <= t0) lo = mid; else hi = mid;
    }
    const uint32_t sample = lo;
    const uint64_t code0 = g.code_begin[sample], n_codes = g.code_begin[sample + 1] - code0;
    const uint64_t *offs = g.offsets + g.offset_begin[sample];
    const uint32_t n_offs = (uint32_t)(g.offset_begin[sample + 1] - g.offset_begin[sample]);
    const uint8_t *codes = g.codes + code0;
    const uint64_t wg0 = (t0 - g.span_begin[sample]) * SPAN2;   // first base of the workgroup
    if (wg0 >= n_codes) return;                                // (padding spans of the sample)
    const uint64_t wg1 = wg0 + (uint64_t)ROWS2 * SPAN2 < n_codes ? wg0 + (uint64_t)ROWS2 * SPAN2 : n_codes;
    for (uint32_t x = tid; x < (uint32_t)(wg1 - wg0); x += 256u) {
        staged[(x >> 7) * PITCH2 + (x & 127u)] = codes[wg0 + x];
    }
    if (LDS_BINS) {
        for (uint32_t b = tid; b < (uint32_t)g.num_bins; b += 256u) lbins[b] = ~0ull;
    }
    __syncthreads();
    auto code_at = [&](uint64_t pos) -> uint32_t {   // base at sample position pos (inside the staged range)
        const uint32_t x = (uint32_t)(pos - wg0);
        return staged[(x >> 7) * PITCH2 + (x & 127u)];
    };
    const uint64_t p0 = wg0 + (uint64_t)tid * SPAN2;
    const uint64_t p1 = p0 + SPAN2 < n_codes ? p0 + SPAN2 : n_codes;

    for (uint32_t ki = 0; ki < g.nk; ++ki) {
        const uint32_t k = g.kmers[ki];
        uint64_t top_f[4], top_r[4];
#pragma unroll
        for (uint32_t b = 0; b < 4; ++b) {
            top_f[b] = g.top_f[ki * 4 + b];
            top_r[b] = g.top_r[ki * 4 + b];
        }
        uint64_t *bins = g.signs + ((uint64_t)sample * g.nk + ki) * g.num_bins;
        if (p0 < n_codes) {
            uint32_t oi;
            {
                uint32_t a = 0, b = n_offs;   // first index with offs[idx] > p0
                while (a < b) {
                    const uint32_t mid = (a + b) >> 1;
                    if (offs[mid] > p0) b = mid; else a = mid + 1;
                }
                oi = a;
            }
            uint64_t next_off = oi < n_offs ? offs[oi] : n_codes;
            bool have = false;
            uint64_t fh = 0, rh = 0;
            for (uint64_t s = p0; s < p1; ++s) {
                if (next_off <= s) {
                    while (oi < n_offs && offs[oi] <= s) ++oi;
                    next_off = oi < n_offs ? offs[oi] : n_codes;
                }
                if (s + k > next_off) {   // the window would span a break (next_iterator, :325-346)
                    have = false;
                    continue;
                }
                if (have) {
                    const uint32_t old_b = code_at(s - 1), new_b = code_at(s + k - 1);
                    fh = srol(fh ^ top_f[old_b]) ^ hash_fwd(new_b);
                    if (g.rc) rh = sror(rh ^ hash_rc(old_b)) ^ top_r[new_b];
                } else {
                    fh = 0;
                    rh = 0;
                    for (uint32_t i = 0; i < k; ++i) fh = srol(fh) ^ hash_fwd(code_at(s + i));
                    if (g.rc) {
                        for (uint32_t i = k; i-- > 0;) rh = srol(rh) ^ hash_rc(code_at(s + i));
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
                if (LDS_BINS) {
                    if (sign < lbins[bin]) atomicMin(&lbins[bin], (unsigned long long)sign);
                } else {
                    if (sign < bins[bin]) atomicMin((unsigned long long *)&bins[bin], (unsigned long long)sign);
                }
            }
        }
        if (LDS_BINS) {   // one global atomic per bin this workgroup touched
            __syncthreads();
            for (uint32_t b = tid; b < (uint32_t)g.num_bins; b += 256u) {
                const unsigned long long v = lbins[b];
                if (v != ~0ull) {
                    if (v < bins[b]) atomicMin((unsigned long long *)&bins[b], v);
                    lbins[b] = ~0ull;
                }
            }
            __syncthreads();
        }
    }
}

hipError_t launch_sketch_signs(const SketchArgs &args, hipStream_t stream)
{
    if (args.n_spans == 0) return hipSuccess;
    const uint64_t blocks = (args.n_spans + 255) / 256;
    if (blocks >= (1ull << 31)) return hipErrorInvalidValue;
    if (args.lds_form) {
        if (args.num_bins <= (uint64_t)LDS_BINS_MAX) {
            hipLaunchKernelGGL(nthash_binmin_lds_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, stream, args);
        } else {
            hipLaunchKernelGGL(nthash_binmin_lds_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, stream, args);
        }
    } else {
        hipLaunchKernelGGL(nthash_binmin_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, args);
    }
    return hipGetLastError();
}

int sketch_span() { return SKETCH_SPAN; }
int sketch_span_lds() { return SPAN2; }

}  // namespace skl
