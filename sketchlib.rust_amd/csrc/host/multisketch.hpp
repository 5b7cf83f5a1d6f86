// multisketch.hpp -- C++ mirror of the read side of the reference's MultiSketch
// (src/sketch/multisketch.rs) plus the writer needed to emit databases in the
// reference's on-disk layout: `.skm` = snappy-framed CBOR of the struct, `.skd` = the
// bins as little-endian u64 (src/sketch/sketch_datafile.rs:159-194).
#pragma once

#include <cstdint>
#include <map>
#include <mutex>
#include <unordered_map>
#include <optional>
#include <memory>
#include <new>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

namespace skl_host {

constexpr uint64_t BBITS = 14;  // src/sketch/mod.rs:34

// Sketch metadata as serialised by the reference (src/sketch/mod.rs:56-71)
struct SketchMeta {
    std::string name;
    std::optional<uint64_t> index;  // position of the sample's block in the .skd
    bool rc = true;
    bool reads = false;
    uint64_t seq_length = 0;
    bool densified = false;
    uint64_t acgt[4] = {0, 0, 0, 0};
    uint64_t non_acgt = 0;
};

// std::vector whose resize() leaves new elements uninitialised: a GB-sized .skd is read straight
// into it, and value-initialising it first is a second, single-threaded pass over the memory.
template <class T>
struct DefaultInitAllocator : std::allocator<T> {
    template <class U>
    struct rebind {
        using other = DefaultInitAllocator<U>;
    };
    using std::allocator<T>::allocator;
    template <class U>
    void construct(U *p) noexcept(std::is_nothrow_default_constructible<U>::value)
    {
        ::new (static_cast<void *>(p)) U;
    }
    template <class U, class... Args>
    void construct(U *p, Args &&...args)
    {
        ::new (static_cast<void *>(p)) U(std::forward<Args>(args)...);
    }
};
using BinVec = std::vector<uint64_t, DefaultInitAllocator<uint64_t>>;

class MultiSketch {
  public:
    uint64_t sketch_size = 0;    // bins (a multiple of 64)
    uint64_t sketchsize64 = 0;   // sketch_size / 64

    MultiSketch() = default;
    // New in-memory database (used to write synthetic / converted databases).
    MultiSketch(std::vector<SketchMeta> meta, uint64_t sketch_size, std::vector<size_t> kmers,
                std::string version = "0.3.0");

    // multisketch.rs:90-103 (including the pre-0.2.0 compatibility rule)
    static MultiSketch load_metadata(const std::string &file_prefix);
    void save_metadata(const std::string &file_prefix) const;     // :80-87
    // A distance at ONE k-mer length (`dist -k`, precluster) never looks at the other slices of a sample: called before the
    // data is read, this keeps the k_idx-th k-mer length only -- kmer_lengths() becomes that one length (index 0), and
    // read_sketch_data* pick its slice out of every sample of the file (a fifth of the bytes of a five-k database to read,
    // hold and upload).  The file and its metadata on disk are not touched.
    void select_kmer(size_t k_idx);
    static void testing_read_slices_without_mapping(bool on);   // (tests: take the positional-read path of a file that cannot be mapped)
    void read_sketch_data(const std::string &file_prefix);        // :167-184
    void read_sketch_data_block(const std::string &file_prefix,   // :187-210
                                const std::vector<std::string> &names);
    static void write_sketch_data(const std::string &file_prefix, const uint64_t *bins, size_t n_words);

    size_t number_samples_loaded() const;                         // :106-111
    std::optional<size_t> get_k_idx(size_t k) const;              // :116-121
    const std::vector<size_t> &kmer_lengths() const { return kmer_lengths_; }
    const std::string &sketch_name(size_t index) const;           // :139-144
    std::optional<size_t> get_sample_index(const std::string &name) const;  // :148-164
    const uint64_t *get_sketch_slice(size_t sketch_idx, size_t k_idx) const;  // :213-219
    bool is_compatible_with(const MultiSketch &other) const;      // :222-226
    const BinVec &bins() const { return sketch_bins_; }
    void set_bins(const std::vector<uint64_t> &bins) { sketch_bins_.assign(bins.begin(), bins.end()); }
    size_t sample_stride() const { return sample_stride_; }
    size_t kmer_stride() const { return kmer_stride_; }
    const std::string &hash_type() const { return hash_type_; }
    const std::string &version() const { return sketch_version_; }
    const std::vector<SketchMeta> &metadata() const { return sketch_metadata_; }

  private:
    std::vector<size_t> kmer_lengths_;
    std::vector<SketchMeta> sketch_metadata_;
    std::vector<std::pair<std::string, size_t>> name_map_order_;  // serialisation order
    // name -> block index, for lookups by name (--subset, completeness files): built from name_map_order_ on first use --
    // a dist run over a million samples never asks, and a million tree inserts are a third of its .skm load
    struct LazyNameMap {
        std::once_flag once;
        std::unordered_map<std::string, size_t> map;
    };
    mutable std::shared_ptr<LazyNameMap> name_lookup_ = std::make_shared<LazyNameMap>();
    const std::unordered_map<std::string, size_t> &name_map() const;
    std::optional<std::vector<size_t>> block_reindex_;
    BinVec sketch_bins_;
    size_t bin_stride_ = 1, kmer_stride_ = 0, sample_stride_ = 0;
    std::optional<size_t> file_k_idx_;     // select_kmer: the slice kept, and the stride of a sample IN THE FILE
    size_t file_sample_stride_ = 0;
    std::string sketch_version_;
    std::string hash_type_ = "DNA";  // "DNA" | "PDB" | "AA:<level>"
};

}  // namespace skl_host
