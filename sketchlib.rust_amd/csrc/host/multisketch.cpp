#include "multisketch.hpp"
#include <algorithm>
#include <unistd.h>
#include <sys/stat.h>
#include <sys/mman.h>
#include <fcntl.h>
#include <set>
#include <thread>
#include <atomic>

#include <cstring>
#include <fstream>
#include <stdexcept>

#include "cbor.hpp"
#include "snappy_frame.hpp"

namespace skl_host {

static std::vector<uint8_t> read_file(const std::string &path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open " + path);
    f.seekg(0, std::ios::end);
    const std::streamoff n = f.tellg();
    f.seekg(0);
    std::vector<uint8_t> buf((size_t)n);
    if (n > 0) f.read((char *)buf.data(), n);
    if (!f) throw std::runtime_error("cannot read " + path);
    return buf;
}

MultiSketch::MultiSketch(std::vector<SketchMeta> meta, uint64_t sketch_size_bins,
                         std::vector<size_t> kmers, std::string version)
{
    // MultiSketch::new, multisketch.rs:50-77
    sketch_size = sketch_size_bins;
    sketchsize64 = sketch_size_bins / 64;
    kmer_lengths_ = std::move(kmers);
    sketch_metadata_ = std::move(meta);
    for (size_t i = 0; i < sketch_metadata_.size(); ++i) {
        if (!sketch_metadata_[i].index) sketch_metadata_[i].index = i;
        name_map_order_.emplace_back(sketch_metadata_[i].name, (size_t)*sketch_metadata_[i].index);
    }
    bin_stride_ = 1;
    kmer_stride_ = (size_t)(sketchsize64 * BBITS);
    sample_stride_ = kmer_stride_ * kmer_lengths_.size();
    sketch_version_ = std::move(version);
    hash_type_ = "DNA";
}

MultiSketch MultiSketch::load_metadata(const std::string &file_prefix)
{
    // (a pull parse straight into the struct: the generic CborValue tree of a 400 000-sample .skm
    // is ~8 million nodes and took longer to build than the .skd took to read)
    const std::string filename = file_prefix + ".skm";
    const std::vector<uint8_t> doc = snappy_frame_decode(read_file(filename));
    CborCursor c(doc.data(), doc.size());
    using Head = CborCursor::Head;
    auto want_uint = [&](const char *what) -> uint64_t {
        const Head h = c.head();
        if (h.major != 0) throw std::runtime_error(std::string("CBOR: expected unsigned int for ") + what);
        return h.value;
    };
    auto want_bool = [&](const char *what) -> bool {
        const Head h = c.head();
        if (h.major != 7 || (h.info != 20 && h.info != 21)) throw std::runtime_error(std::string("CBOR: expected a bool for ") + what);
        return h.info == 21;
    };
    auto want_text = [&](const char *what) -> std::string {
        const Head h = c.head();
        if (h.major != 3) throw std::runtime_error(std::string("CBOR: expected text for ") + what);
        return c.text(h);
    };
    const Head root = c.head();
    if (root.major != 5) throw std::runtime_error(filename + ": not a CBOR map");
    MultiSketch m;
    std::set<std::string> seen;
    for (uint64_t f = 0; c.more(root, f); ++f) {
        const std::string key = want_text("a field name");
        seen.insert(key);
        if (key == "sketch_size") {
            m.sketch_size = want_uint("sketch_size");
        } else if (key == "sketchsize64") {   // #[serde(default)]
            m.sketchsize64 = want_uint("sketchsize64");
        } else if (key == "kmer_lengths") {
            const Head a = c.head();
            if (a.major != 4) throw std::runtime_error("CBOR: expected an array for kmer_lengths");
            for (uint64_t k = 0; c.more(a, k); ++k) m.kmer_lengths_.push_back((size_t)want_uint("kmer_lengths"));
            c.close(a);
        } else if (key == "sketch_metadata") {
            const Head a = c.head();
            if (a.major != 4) throw std::runtime_error("CBOR: expected an array for sketch_metadata");
            if (!a.indefinite) m.sketch_metadata_.reserve((size_t)a.value);
            for (uint64_t k = 0; c.more(a, k); ++k) {
                const Head o = c.head();
                if (o.major != 5) throw std::runtime_error("CBOR: expected a map for a sketch_metadata entry");
                SketchMeta sm;
                for (uint64_t q = 0; c.more(o, q); ++q) {
                    const std::string field = want_text("a sketch field name");
                    if (field == "name") {
                        sm.name = want_text("name");
                    } else if (field == "index") {
                        const Head h = c.head();   // uint | null
                        if (h.major == 0) sm.index = h.value;
                    } else if (field == "rc") {
                        sm.rc = want_bool("rc");
                    } else if (field == "reads") {
                        sm.reads = want_bool("reads");
                    } else if (field == "seq_length") {
                        sm.seq_length = want_uint("seq_length");
                    } else if (field == "densified") {
                        sm.densified = want_bool("densified");
                    } else if (field == "acgt") {
                        const Head v = c.head();
                        if (v.major != 4) throw std::runtime_error("CBOR: expected an array for acgt");
                        for (uint64_t x = 0; c.more(v, x); ++x) {
                            const uint64_t count = want_uint("acgt");
                            if (x < 4) sm.acgt[x] = count;
                        }
                        c.close(v);
                    } else if (field == "non_acgt") {
                        sm.non_acgt = want_uint("non_acgt");
                    } else {
                        c.skip();
                    }
                }
                c.close(o);
                m.sketch_metadata_.push_back(std::move(sm));
            }
            c.close(a);
        } else if (key == "name_map") {
            const Head o = c.head();
            if (o.major != 5) throw std::runtime_error("CBOR: expected a map for name_map");
            if (!o.indefinite) m.name_map_order_.reserve((size_t)o.value);
            for (uint64_t k = 0; c.more(o, k); ++k) {
                std::string name = want_text("name_map");
                const size_t idx = (size_t)want_uint("name_map");
                m.name_map_order_.emplace_back(std::move(name), idx);
            }
            c.close(o);
        } else if (key == "bin_stride") {
            m.bin_stride_ = (size_t)want_uint("bin_stride");
        } else if (key == "kmer_stride") {
            m.kmer_stride_ = (size_t)want_uint("kmer_stride");
        } else if (key == "sample_stride") {
            m.sample_stride_ = (size_t)want_uint("sample_stride");
        } else if (key == "sketch_version") {
            m.sketch_version_ = want_text("sketch_version");
        } else if (key == "hash_type") {
            const CborValue ht = c.value();
            if (ht.kind == CborValue::TEXT) {
                m.hash_type_ = ht.s;
            } else if (ht.kind == CborValue::MAP && !ht.map.empty()) {
                m.hash_type_ = ht.map[0].first.s + ":" + ht.map[0].second.s;  // {"AA": "Level1"}
            }
        } else {
            c.skip();
        }
    }
    c.close(root);
    for (const char *need : {"sketch_size", "kmer_lengths", "sketch_metadata", "name_map", "bin_stride", "kmer_stride",
                             "sample_stride", "sketch_version", "hash_type"}) {
        if (!seen.count(need)) throw std::runtime_error(filename + ": missing field " + need);
    }
    // For backwards compatibility (field added in v0.2.0), multisketch.rs:96-100
    if (m.sketchsize64 == 0) {
        m.sketchsize64 = m.sketch_size;
        m.sketch_size *= 64;
    }
    // the strides are what multisketch.rs:61-73 computes; a file that says otherwise cannot be indexed
    if (m.bin_stride_ != 1 || m.kmer_stride_ != (size_t)m.sketchsize64 * BBITS ||
        m.sample_stride_ != m.kmer_stride_ * m.kmer_lengths_.size() || m.kmer_lengths_.empty() || m.sketchsize64 == 0) {
        throw std::runtime_error(filename + ": strides do not match sketch size and k-mer lengths");
    }
    for (const auto &sm : m.sketch_metadata_) {
        if (sm.index && *sm.index >= m.sketch_metadata_.size()) throw std::runtime_error(filename + ": sample index out of range");
    }
    // name_map values (the samples' block indices, multisketch.rs:55-58) index per-sample arrays
    // later (get_sample_index -> the completeness vectors of io.cpp read_completeness_file)
    for (const auto &kv : m.name_map_order_) {
        if (kv.second >= m.sketch_metadata_.size()) throw std::runtime_error(filename + ": name_map index out of range");
    }
    return m;
}

void MultiSketch::save_metadata(const std::string &file_prefix) const
{
    // the fields in the order serde writes the struct (multisketch.rs:21-44), straight into the byte stream
    CborWriter w;
    w.map(10);
    w.key("sketch_size");
    w.uint(sketch_size);
    w.key("sketchsize64");
    w.uint(sketchsize64);
    w.key("kmer_lengths");
    w.array(kmer_lengths_.size());
    for (size_t k : kmer_lengths_) w.uint(k);
    w.key("sketch_metadata");
    w.array(sketch_metadata_.size());
    for (const auto &s : sketch_metadata_) {
        w.map(8);
        w.key("name");
        w.text(s.name);
        w.key("index");
        if (s.index) w.uint(*s.index);
        else w.null();
        w.key("rc");
        w.boolean(s.rc);
        w.key("reads");
        w.boolean(s.reads);
        w.key("seq_length");
        w.uint(s.seq_length);
        w.key("densified");
        w.boolean(s.densified);
        w.key("acgt");
        w.array(4);
        for (uint64_t c : s.acgt) w.uint(c);
        w.key("non_acgt");
        w.uint(s.non_acgt);
    }
    w.key("name_map");
    w.map(name_map_order_.size());
    for (const auto &kv : name_map_order_) {
        w.text(kv.first);
        w.uint(kv.second);
    }
    w.key("bin_stride");
    w.uint(bin_stride_);
    w.key("kmer_stride");
    w.uint(kmer_stride_);
    w.key("sample_stride");
    w.uint(sample_stride_);
    w.key("sketch_version");
    w.text(sketch_version_);
    w.key("hash_type");
    const size_t colon = hash_type_.find(':');
    if (colon == std::string::npos) {
        w.text(hash_type_);
    } else {
        w.map(1);
        w.text(hash_type_.substr(0, colon));
        w.text(hash_type_.substr(colon + 1));
    }
    const std::vector<uint8_t> framed = snappy_frame_encode(w.bytes());
    std::ofstream f(file_prefix + ".skm", std::ios::binary);
    if (!f) throw std::runtime_error("cannot create " + file_prefix + ".skm");
    f.write((const char *)framed.data(), (std::streamsize)framed.size());
}

void MultiSketch::write_sketch_data(const std::string &file_prefix, const uint64_t *bins, size_t n_words)
{
    std::ofstream f(file_prefix + ".skd", std::ios::binary);
    if (!f) throw std::runtime_error("cannot create " + file_prefix + ".skd");
    f.write((const char *)bins, (std::streamsize)(n_words * sizeof(uint64_t)));  // little-endian host
}

static std::atomic<bool> g_no_mapping{false};
void MultiSketch::testing_read_slices_without_mapping(bool on) { g_no_mapping = on; }

void MultiSketch::select_kmer(size_t k_idx)
{
    if (k_idx >= kmer_lengths_.size()) throw std::runtime_error("select_kmer: no such k-mer length");
    if (file_k_idx_) throw std::runtime_error("select_kmer: already selected");
    if (!sketch_bins_.empty()) throw std::runtime_error("select_kmer: the sketch data has been read already");
    file_k_idx_ = k_idx;
    file_sample_stride_ = sample_stride_;
    kmer_lengths_ = {kmer_lengths_[k_idx]};
    sample_stride_ = kmer_stride_;
}

void MultiSketch::read_sketch_data(const std::string &file_prefix)
{
    if (file_k_idx_) {
        // one slice per sample, picked out of a mapping of the file by several threads (a file that cannot be mapped:
        // one positional read per slice)
        const std::string path = file_prefix + ".skd";
        const int fd = ::open(path.c_str(), O_RDONLY | O_CLOEXEC);
        if (fd < 0) throw std::runtime_error("cannot open " + path);
        struct stat st;
        if (::fstat(fd, &st) != 0) {
            ::close(fd);
            throw std::runtime_error("cannot stat " + path);
        }
        const size_t n = sketch_metadata_.size();
        const size_t file_bytes = (size_t)st.st_size;
        if (file_bytes / sizeof(uint64_t) < n * file_sample_stride_) {
            ::close(fd);
            throw std::runtime_error(path + " is shorter than its metadata says");
        }
        // Positional reads by default (round 5): a file truncated or replaced while it is read -- NFS, a concurrent `sketch`
        // run -- then gives the clean "error reading" below instead of a SIGBUS out of a mapping; the slices are read by
        // several threads either way.  SKL_SKD_MMAP=1 maps the file instead.
        static const bool want_mapping = [] { const char *e = std::getenv("SKL_SKD_MMAP"); return e && e[0] == '1'; }();
        void *m = file_bytes && want_mapping && !g_no_mapping ? ::mmap(nullptr, file_bytes, PROT_READ, MAP_PRIVATE, fd, 0) : MAP_FAILED;
        const char *mapped = m == MAP_FAILED ? nullptr : (const char *)m;
        sketch_bins_.resize(n * kmer_stride_);
        const size_t slice_bytes = kmer_stride_ * sizeof(uint64_t);
        const size_t n_threads = n * slice_bytes >= (64u << 20) ? std::min<size_t>(32, std::max(1u, std::thread::hardware_concurrency())) : 1;
        std::atomic<bool> ok{true};
        auto copy_range = [&](size_t t) {
            char *dst = reinterpret_cast<char *>(sketch_bins_.data());
            for (size_t i = n * t / n_threads, end = n * (t + 1) / n_threads; i < end; ++i) {
                const size_t at = (i * file_sample_stride_ + *file_k_idx_ * kmer_stride_) * sizeof(uint64_t);
                if (mapped) {
                    memcpy(dst + i * slice_bytes, mapped + at, slice_bytes);
                    continue;
                }
                for (size_t got_all = 0; got_all < slice_bytes;) {
                    const ssize_t got = ::pread(fd, dst + i * slice_bytes + got_all, slice_bytes - got_all, (off_t)(at + got_all));
                    if (got <= 0) {
                        ok = false;
                        return;
                    }
                    got_all += (size_t)got;
                }
            }
        };
        std::vector<std::thread> pool;
        for (size_t t = 1; t < n_threads; ++t) pool.emplace_back(copy_range, t);
        copy_range(0);
        for (auto &t : pool) t.join();
        if (mapped) ::munmap(m, file_bytes);
        ::close(fd);
        if (!ok) throw std::runtime_error("error reading " + path);
        block_reindex_.reset();
        return;
    }
    // read_all_from_skd streams the whole file (sketch_datafile.rs:159-168); here straight into
    // the bins, in slices read concurrently (a GB-sized .skd is otherwise a second of memcpy)
    const std::string path = file_prefix + ".skd";
    const int fd = ::open(path.c_str(), O_RDONLY | O_CLOEXEC);
    if (fd < 0) throw std::runtime_error("cannot open " + path);
    struct stat st;
    if (::fstat(fd, &st) != 0) {
        ::close(fd);
        throw std::runtime_error("cannot stat " + path);
    }
    const size_t bytes = (size_t)st.st_size / sizeof(uint64_t) * sizeof(uint64_t);
    if (bytes / sizeof(uint64_t) < sketch_metadata_.size() * sample_stride_) {   // every later index trusts this
        ::close(fd);
        throw std::runtime_error(path + " is shorter than its metadata says");
    }
    sketch_bins_.resize(bytes / sizeof(uint64_t));
    char *dst = reinterpret_cast<char *>(sketch_bins_.data());
    const size_t n_slices = bytes >= (64u << 20) ? std::min<size_t>(32, std::max(1u, std::thread::hardware_concurrency())) : 1;
    std::atomic<bool> ok{true};
    auto read_slice = [&](size_t sl) {
        size_t off = bytes * sl / n_slices;
        const size_t end = bytes * (sl + 1) / n_slices;
        while (off < end) {
            const ssize_t got = ::pread(fd, dst + off, end - off, (off_t)off);
            if (got <= 0) {
                ok = false;
                return;
            }
            off += (size_t)got;
        }
    };
    std::vector<std::thread> pool;
    for (size_t sl = 1; sl < n_slices; ++sl) pool.emplace_back(read_slice, sl);
    read_slice(0);
    for (auto &t : pool) t.join();
    ::close(fd);
    if (!ok) throw std::runtime_error("error reading " + path);
    block_reindex_.reset();
}

void MultiSketch::read_sketch_data_block(const std::string &file_prefix,
                                         const std::vector<std::string> &names)
{
    std::vector<size_t> block_reindex, read_indices;
    for (const auto &name : names) {
        auto it = name_map().find(name);
        if (it == name_map().end()) {
            throw std::runtime_error("Could not find requested sample " + name + " in sketch metadata");
        }
        const size_t sketch_idx = it->second;
        read_indices.push_back((size_t)sketch_metadata_.at(sketch_idx).index.value_or(sketch_idx));
        block_reindex.push_back(sketch_idx);
    }
    std::ifstream f(file_prefix + ".skd", std::ios::binary);
    if (!f) throw std::runtime_error("cannot open " + file_prefix + ".skd");
    sketch_bins_.assign(sample_stride_ * read_indices.size(), 0);
    const size_t in_file_stride = file_k_idx_ ? file_sample_stride_ : sample_stride_;    // (select_kmer: one slice of each sample)
    const size_t in_sample = file_k_idx_ ? *file_k_idx_ * kmer_stride_ : 0;
    for (size_t i = 0; i < read_indices.size(); ++i) {
        f.seekg((std::streamoff)((read_indices[i] * in_file_stride + in_sample) * sizeof(uint64_t)));
        f.read((char *)(sketch_bins_.data() + i * sample_stride_),
               (std::streamsize)(sample_stride_ * sizeof(uint64_t)));
        if (!f) throw std::runtime_error(file_prefix + ".skd is shorter than its metadata says");
    }
    block_reindex_ = std::move(block_reindex);
}

const std::unordered_map<std::string, size_t> &MultiSketch::name_map() const
{
    std::call_once(name_lookup_->once, [&] {
        name_lookup_->map.reserve(name_map_order_.size());
        for (const auto &kv : name_map_order_) name_lookup_->map[kv.first] = kv.second;   // (a repeated name: the last entry, as in a serde map)
    });
    return name_lookup_->map;
}

size_t MultiSketch::number_samples_loaded() const
{
    return block_reindex_ ? block_reindex_->size() : sketch_metadata_.size();
}

std::optional<size_t> MultiSketch::get_k_idx(size_t k) const
{
    for (size_t i = 0; i < kmer_lengths_.size(); ++i) {
        if (kmer_lengths_[i] == k) return i;
    }
    return std::nullopt;
}

const std::string &MultiSketch::sketch_name(size_t index) const
{
    return block_reindex_ ? sketch_metadata_.at((*block_reindex_).at(index)).name
                          : sketch_metadata_.at(index).name;
}

std::optional<size_t> MultiSketch::get_sample_index(const std::string &name) const
{
    if (block_reindex_) {
        for (size_t logical = 0; logical < block_reindex_->size(); ++logical) {
            if (sketch_metadata_[(*block_reindex_)[logical]].name == name) return logical;
        }
        return std::nullopt;
    }
    auto it = name_map().find(name);
    if (it == name_map().end()) return std::nullopt;
    return it->second;
}

const uint64_t *MultiSketch::get_sketch_slice(size_t sketch_idx, size_t k_idx) const
{
    return sketch_bins_.data() + sketch_idx * sample_stride_ + k_idx * kmer_stride_;
}

bool MultiSketch::is_compatible_with(const MultiSketch &o) const
{
    return kmer_lengths_ == o.kmer_lengths_ && sketch_size == o.sketch_size && hash_type_ == o.hash_type_;
}

}  // namespace skl_host
