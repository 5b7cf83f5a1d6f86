// inverted.hpp -- the inverted index of single-k sketches the reference's precluster mode uses
// to pick kNN candidates: `Inverted` (src/inverted.rs:46-58), its on-disk
// forms `.ski` (snappy frame around the MessagePack document rmp-serde writes for the struct: a
// 9-element array in field order, bitmaps as bin blobs in Roaring's portable serialisation,
// inverted.rs:194-216; msgpack.hpp) and `.skq` (raw little-endian u16 bins, [sample][sketch_size],
// inverted.rs:88-98), `any_shared_bins` (:259-268) and the pair count of
// `precluster --count` (:271-300).
#pragma once

#include <cstdint>
#include <optional>
#include <string>
#include <unordered_map>
#include <vector>

#include "sketch.hpp"

namespace skl_host {

// Roaring "portable" serialisation of an ascending list of u32 (the format `roaring` 0.10's
// serde impl hands to the serialiser as bytes; RoaringFormatSpec).  The reader takes array, bitmap
// and run containers; the writer emits array / bitmap containers without the run cookie.
std::string roaring_serialize(const std::vector<uint32_t> &sorted_values);
std::vector<uint32_t> roaring_deserialize(const std::string &bytes);

class Inverted {
  public:
    // index[bin][value] = ascending sample ids (index order) whose sketch has `value` at `bin`
    std::vector<std::unordered_map<uint16_t, std::vector<uint32_t>>> index;
    size_t n_samples = 0;
    std::vector<std::string> sample_names;
    std::optional<std::vector<std::string>> metadata, labels;
    size_t kmer_size = 0;
    std::string sketch_version = "0.3.0";
    bool rc = true;
    std::string hash_type = "DNA";

    size_t sketch_size() const { return index.empty() ? sketch_size_hint : index.size(); }
    bool has_index() const { return !index.empty() || sketch_size_hint == 0; }
    size_t sketch_size_hint = 0;   // set by load(prefix, false): number of bins of the skipped index

    // Inverted::new minus the sketching (inverted.rs:99-112,467-499); sketches in index order.
    static Inverted from_sketches(const std::vector<std::vector<uint16_t>> &sketches,
                                  std::vector<std::string> names, size_t k, bool rc);
    void save(const std::string &file_prefix) const;          // <prefix>.ski
    // with_index = false reads everything but the bitmaps (names, k, sketch size): enough when
    // the candidate search runs on the device from the .skq
    static Inverted load(const std::string &file_prefix, bool with_index = true);   // throws std::runtime_error

    // Samples sharing at least one bin with the query sketch, ascending (inverted.rs:259-268).
    std::vector<uint32_t> any_shared_bins(const uint16_t *query_sigs) const;
    // The same with caller-owned scratch, for loops over many queries: `stamp` has n_samples
    // entries (zero-initialised once), `epoch` must differ between calls (and be non-zero).
    // Cost is proportional to the lists touched, not to n_samples.
    void any_shared_bins(const uint16_t *query_sigs, std::vector<uint32_t> &stamp, uint32_t epoch,
                         std::vector<uint32_t> &out) const;
    // Number of distinct sample pairs sharing at least one bin (inverted.rs:271-300).
    uint64_t any_shared_bin_pairs(size_t threads) const;
};

// Inverted::sketch_files_inverted for one single-entry sample (inverted.rs:303-395):
// get_signs_no_densify over exactly `sketch_size` bins, densify_bin, `as u16`.
std::vector<uint16_t> sketch_sample_inverted(const InputFastx &input, size_t k, uint64_t sketch_size, bool rc);

void write_skq(const std::string &path, const std::vector<std::vector<uint16_t>> &sketches);
// [n_samples * sketch_size] u16; throws if the file does not hold exactly that many.
std::vector<uint16_t> read_skq(const std::string &path, size_t n_samples, size_t sketch_size);

// reorder_input_files (src/io.rs:40-115): index position of each input sample when samples are
// grouped by the label of a `name<TAB>label` file (labels in order of first appearance, inputs
// without a label last).  Equal labels keep the order of the label file (the reference sorts
// with sort_unstable_by_key; its order among equal labels is unspecified).
std::vector<size_t> reorder_by_labels(const std::vector<InputFastx> &inputs, const std::string &label_file,
                                      std::optional<std::vector<std::string>> *labels_out);

}  // namespace skl_host
