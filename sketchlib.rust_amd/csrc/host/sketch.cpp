#include "sketch.hpp"
#include "inverted.hpp"

#include <zlib.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <cstring>
#include <fstream>
#include <mutex>
#include <stdexcept>
#include <thread>

namespace skl_host {

namespace {

// src/hashing/nthash_tables.rs:4-16
const uint64_t HASH_LOOKUP[4] = {0x3c8bfbb395c60474ull, 0x3193c18562a02b4cull, 0x295549f54be24456ull,
                                 0x20323ed082572324ull};
const uint64_t RC_HASH_LOOKUP[4] = {0x295549f54be24456ull, 0x20323ed082572324ull, 0x3c8bfbb395c60474ull,
                                    0x3193c18562a02b4cull};

inline uint64_t rotl1(uint64_t v) { return (v << 1) | (v >> 63); }
inline uint64_t rotr1(uint64_t v) { return (v >> 1) | (v << 63); }
// swapbits033, src/hashing/mod.rs:99-103
inline uint64_t swapbits033(uint64_t v)
{
    const uint64_t x = (v ^ (v >> 33)) & 1;
    return v ^ (x | (x << 33));
}
// the split rotation ntHash rolls with (nthash_iterator.rs:368-370) and its inverse
inline uint64_t srol(uint64_t v) { return swapbits033(rotl1(v)); }
inline uint64_t sror(uint64_t v) { return rotr1(swapbits033(v)); }

inline bool valid_base(uint8_t b)
{
    b |= 0x20;
    return b == 'a' || b == 'c' || b == 'g' || b == 't' || b == 'u';  // hashing/mod.rs:93-97
}
inline uint8_t encode_base(uint8_t b) { return (b >> 1) & 0x3; }  // hashing/mod.rs:82-85

std::string read_maybe_gz(const std::string &path)
{
    gzFile f = gzopen(path.c_str(), "rb");  // transparently reads plain files too
    if (!f) throw std::runtime_error("Invalid path/file: " + path);
    gzbuffer(f, 1 << 20);
    std::string data;
    {
        std::ifstream probe(path, std::ios::binary | std::ios::ate);
        if (probe) data.reserve((size_t)probe.tellg() * (gzdirect(f) ? 1 : 4) + 16);
    }
    std::vector<char> buf(1 << 20);
    int n;
    while ((n = gzread(f, buf.data(), (unsigned)buf.size())) > 0) data.append(buf.data(), (size_t)n);
    gzclose(f);
    if (n < 0) throw std::runtime_error("Invalid FASTA/Q record in " + path);
    return data;
}

}  // namespace

void add_fasta(const std::string &path, Sequence &s)
{
    const std::string data = read_maybe_gz(path);
    if (!data.empty() && data[0] == '@') {
        throw std::runtime_error(path + ": FASTQ input (reads) is not supported by this build");
    }
    // byte classes: 0-3 = base code, 4 = line break (skipped), 5 = anything else inside a record
    // (an invalid base: recorded as a break)
    static const auto table = [] {
        std::array<uint8_t, 256> t;
        t.fill(5);
        t[(uint8_t)'\n'] = 4;
        t[(uint8_t)'\r'] = 4;
        for (int c = 0; c < 256; ++c) {
            if (valid_base((uint8_t)c)) t[c] = encode_base((uint8_t)c);
        }
        return t;
    }();
    const size_t n = data.size();
    const size_t base0 = s.codes.size();
    s.codes.resize(base0 + n);          // upper bound; trimmed below (one allocation, no per-base growth)
    uint8_t *out = s.codes.data() + base0;
    size_t n_out = 0;
    uint64_t counts[6] = {0, 0, 0, 0, 0, 0};
    size_t i = 0;
    bool in_record = false;
    while (i < n) {
        if (data[i] == '>') {
            if (in_record) s.offsets.push_back(base0 + n_out);  // record boundary
            in_record = true;
            const void *nl = std::memchr(data.data() + i, '\n', n - i);   // skip header line
            i = nl ? (size_t)((const char *)nl - data.data()) : n;
            continue;
        }
        if (!in_record) {
            ++i;
            continue;
        }
        // sequence bytes up to the next header
        const void *gt = std::memchr(data.data() + i, '>', n - i);
        const size_t end = gt ? (size_t)((const char *)gt - data.data()) : n;
        for (; i < end; ++i) {
            const uint8_t cls = table[(uint8_t)data[i]];
            if (cls < 4) {
                out[n_out++] = cls;
                ++counts[cls];
            } else if (cls == 5) {
                ++counts[5];
                s.offsets.push_back(base0 + n_out);
            }
        }
    }
    if (in_record) s.offsets.push_back(base0 + n_out);
    s.codes.resize(base0 + n_out);
    for (int b = 0; b < 4; ++b) s.acgt[b] += counts[b];
    s.non_acgt += counts[5];
}

namespace {

// Bin minima of hash % SIGN_MOD over every valid k-mer (get_signs, sketch/mod.rs:132-153);
// canonical hash = min(forward, reverse-complement) when rc (nthash_iterator.rs:62-68).
// A window [s, s+k) is valid iff no offset o satisfies s < o < s+k (next_iterator, :325-346).
void bin_minima(const Sequence &seq, size_t k, bool rc, std::vector<uint64_t> &signs)
{
    const size_t n = seq.codes.size();
    const uint64_t num_bins = signs.size();
    const uint64_t bin_size = (SIGN_MOD + num_bins - 1) / num_bins;
    if (n < k) throw std::runtime_error("K-mer larger than smallest valid sequence");
    // srol^(k-1) of each seed: the weight of the oldest base (forward) / newest base (reverse)
    uint64_t top_f[4], top_r[4];
    for (int b = 0; b < 4; ++b) {
        top_f[b] = HASH_LOOKUP[b];
        top_r[b] = RC_HASH_LOOKUP[b];
        for (size_t m = 1; m < k; ++m) {
            top_f[b] = srol(top_f[b]);
            top_r[b] = srol(top_r[b]);
        }
    }
    size_t off_idx = 0;
    size_t start = 0;
    bool any = false;
    while (start + k <= n) {
        // skip windows containing an offset strictly inside
        while (off_idx < seq.offsets.size() && seq.offsets[off_idx] <= start) ++off_idx;
        if (off_idx < seq.offsets.size() && seq.offsets[off_idx] < start + k) {
            start = seq.offsets[off_idx];  // restart right after the N / boundary
            continue;
        }
        // run of valid windows: from `start` until the next offset
        const size_t run_end = off_idx < seq.offsets.size() ? seq.offsets[off_idx] : n;  // exclusive base bound
        uint64_t fh = 0, rh = 0;
        for (size_t i = 0; i < k; ++i) {
            fh = srol(fh) ^ HASH_LOOKUP[seq.codes[start + i]];
        }
        if (rc) {
            for (size_t i = k; i-- > 0;) rh = srol(rh) ^ RC_HASH_LOOKUP[seq.codes[start + i]];
        }
        size_t s = start;
        for (;;) {
            const uint64_t h = rc ? std::min(fh, rh) : fh;
            const uint64_t sign = h % SIGN_MOD;
            uint64_t &slot = signs[sign / bin_size];
            if (sign < slot) slot = sign;
            any = true;
            if (s + k >= run_end) break;
            const uint8_t old_b = seq.codes[s], new_b = seq.codes[s + k];
            fh = srol(fh ^ top_f[old_b]) ^ HASH_LOOKUP[new_b];
            if (rc) rh = sror(rh ^ RC_HASH_LOOKUP[old_b]) ^ top_r[new_b];
            ++s;
        }
        start = run_end;
    }
    if (!any) throw std::runtime_error("K-mer larger than smallest valid sequence");
}

// sketch/mod.rs:225-231
inline uint64_t universal_hash(uint64_t s, uint64_t t)
{
    const uint64_t x = s * 1009ull + t * (1000ull * 1000ull + 3ull);
    return (x * 48271ull + 11ull) % ((1ull << 31) - 1);
}

}  // namespace

// densify_bin, sketch/mod.rs:237-258
bool densify_bin(std::vector<uint64_t> &signs)
{
    uint64_t maxval = 0;
    for (uint64_t s : signs) maxval = std::max(maxval, s);
    if (maxval != UINT64_MAX) return false;
    for (size_t i = 0; i < signs.size(); ++i) {
        size_t j = i;
        uint64_t attempts = 0;
        while (signs[j] == UINT64_MAX) {
            j = (size_t)(universal_hash(i, attempts) % signs.size());
            ++attempts;
        }
        signs[i] = signs[j];
    }
    return true;
}

// fill_usigs, sketch/mod.rs:215-223
void fill_usigs(uint64_t *usigs, const std::vector<uint64_t> &signs)
{
    for (size_t idx = 0; idx < signs.size(); ++idx) {
        const size_t leftshift = idx % 64;
        for (uint64_t p = 0; p < BBITS; ++p) {
            usigs[idx / 64 * BBITS + p] |= ((signs[idx] >> p) & 1ull) << leftshift;
        }
    }
}

std::vector<InputFastx> read_input_fastas(const std::vector<std::string> &seq_files)
{
    // the reference strips the directory when the extension is a known FASTA/FASTQ one
    static const char *exts[] = {".fa", ".fasta", ".fa.gz", ".fasta.gz", ".fastq", ".fastq.gz", ".fq", ".fq.gz"};
    std::vector<InputFastx> out;
    for (const auto &file : seq_files) {
        std::string name = file;
        bool known = false;
        for (const char *e : exts) {
            const size_t el = strlen(e);
            if (file.size() > el && file.compare(file.size() - el, el, e) == 0) known = true;
        }
        const size_t slash = file.rfind('/');
        if (known && slash != std::string::npos) name = file.substr(slash + 1);
        out.push_back({name, {file}});
    }
    return out;
}

std::vector<InputFastx> read_rfile(const std::string &file_list)
{
    std::ifstream f(file_list);
    if (!f) throw std::runtime_error("Unable to open file_list " + file_list);
    std::vector<InputFastx> out;
    std::string line;
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty()) continue;
        std::vector<std::string> fields;
        size_t pos = 0;
        for (;;) {
            const size_t tab = line.find_first_of("\t ", pos);
            fields.push_back(line.substr(pos, tab == std::string::npos ? std::string::npos : tab - pos));
            if (tab == std::string::npos) break;
            pos = tab + 1;
        }
        if (fields.size() < 2) throw std::runtime_error("Unable to parse line in file_list: " + line);
        out.push_back({fields[0], std::vector<std::string>(fields.begin() + 1, fields.end())});
    }
    return out;
}

std::vector<size_t> parse_kmers(const std::vector<size_t> &k_vals, const std::vector<size_t> &k_seq)
{
    std::vector<size_t> kmers;
    if (!k_vals.empty()) {
        kmers = k_vals;
    } else if (k_seq.size() == 3 && k_seq[2] > 0) {
        for (size_t k = k_seq[0]; k <= k_seq[1]; k += k_seq[2]) kmers.push_back(k);
    } else {
        throw std::runtime_error("Must specify --k-vals or --k-seq");
    }
    std::sort(kmers.begin(), kmers.end());
    for (size_t k : kmers) {
        if (k < 3) throw std::runtime_error("K-mers must be >=3");
    }
    return kmers;
}

SketchResult sketch_sample(const InputFastx &input, const std::vector<size_t> &kmers, uint64_t sketch_size,
                           bool rc)
{
    Sequence seq;
    for (const auto &file : input.second) add_fasta(file, seq);
    uint64_t total = 0;
    for (uint64_t c : seq.acgt) total += c;
    if (total == 0) throw std::runtime_error(input.first + " has no valid sequence");
    const uint64_t ss64 = (sketch_size + 63) / 64;  // num_bins, sketch/mod.rs:49-54
    const uint64_t num_bins = ss64 * 64;
    SketchResult out;
    out.usigs.assign((size_t)(ss64 * BBITS * kmers.size()), 0);
    bool densified = false;
    for (size_t ki = 0; ki < kmers.size(); ++ki) {
        std::vector<uint64_t> signs((size_t)num_bins, UINT64_MAX);
        bin_minima(seq, kmers[ki], rc, signs);
        densified |= densify_bin(signs);
        fill_usigs(out.usigs.data() + ki * ss64 * BBITS, signs);
    }
    out.meta.name = input.first;
    out.meta.rc = rc;
    out.meta.reads = false;
    out.meta.seq_length = total;
    out.meta.densified = densified;
    for (int b = 0; b < 4; ++b) out.meta.acgt[b] = seq.acgt[b];
    out.meta.non_acgt = seq.non_acgt;
    return out;
}

std::vector<uint16_t> sketch_sample_inverted(const InputFastx &input, size_t k, uint64_t sketch_size, bool rc)
{
    Sequence seq;
    for (const auto &file : input.second) add_fasta(file, seq);
    uint64_t total = 0;
    for (uint64_t c : seq.acgt) total += c;
    if (total == 0) throw std::runtime_error(input.first + " has no valid sequence");
    std::vector<uint64_t> signs((size_t)sketch_size, UINT64_MAX);   // exactly sketch_size bins, inverted.rs:352-357
    bin_minima(seq, k, rc, signs);
    densify_bin(signs);
    std::vector<uint16_t> out(signs.size());
    for (size_t i = 0; i < signs.size(); ++i) out[i] = (uint16_t)signs[i];   // `*h as u16`, inverted.rs:380
    return out;
}

MultiSketch sketch_files(const std::string &output_prefix, const std::vector<InputFastx> &inputs,
                         const std::vector<size_t> &kmers, uint64_t sketch_size, bool rc, size_t threads)
{
    const uint64_t ss64 = (sketch_size + 63) / 64;
    const size_t sample_words = (size_t)(ss64 * BBITS * kmers.size());
    std::vector<SketchResult> results(inputs.size());
    std::atomic<size_t> next{0};
    std::string error;
    std::mutex *err_mutex = nullptr;
    (void)err_mutex;
    auto worker = [&]() {
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= inputs.size()) break;
            try {
                results[i] = sketch_sample(inputs[i], kmers, sketch_size, rc);
            } catch (const std::exception &e) {
                results[i].meta.name.clear();
                results[i].usigs.clear();
                static std::mutex m;
                std::lock_guard<std::mutex> lock(m);
                if (error.empty()) error = e.what();
            }
        }
    };
    threads = std::max<size_t>(1, std::min(threads, inputs.size()));
    std::vector<std::thread> pool;
    for (size_t t = 1; t < threads; ++t) pool.emplace_back(worker);
    worker();
    for (auto &t : pool) t.join();
    if (!error.empty()) throw std::runtime_error(error);

    std::vector<uint64_t> bins(sample_words * inputs.size());
    std::vector<SketchMeta> meta;
    for (size_t i = 0; i < inputs.size(); ++i) {
        std::copy(results[i].usigs.begin(), results[i].usigs.end(), bins.begin() + i * sample_words);
        results[i].meta.index = i;
        meta.push_back(results[i].meta);
    }
    MultiSketch::write_sketch_data(output_prefix, bins.data(), bins.size());
    MultiSketch m(std::move(meta), ss64 * 64, kmers);
    m.save_metadata(output_prefix);
    m.set_bins(std::move(bins));
    return m;
}

}  // namespace skl_host
