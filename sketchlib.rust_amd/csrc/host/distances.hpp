// distances.hpp -- C++ mirror of the reference's `distances` module API
// (src/distances/mod.rs): same function names, argument meaning and error behaviour,
// implemented by calling the gfx950 engine through the C ABI (include/sketchlib_dist.h).
#pragma once

#include <iosfwd>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "distance_matrix.hpp"
#include "inverted.hpp"
#include "multisketch.hpp"

struct skl_ctx;

namespace skl_host {

// What a Rust panic is in the reference (exit code 101 in the CLI).
struct Panic : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// One GPU context (skl_ctx) -- the stand-in for the reference's global rayon pool.
class Device {
  public:
    explicit Device(int device = 0);
    ~Device();
    Device(const Device &) = delete;
    Device &operator=(const Device &) = delete;
    skl_ctx *ctx() const { return ctx_; }

  private:
    skl_ctx *ctx_ = nullptr;
};

// Several GPUs of one node driven from one process: one host thread and one context per
// device, every device holding the whole slab, the pair space split into contiguous row
// bands whose results are copied straight into the right offsets of the host output (the
// bands are contiguous slices of the reference's output arrays, so no exchange is needed).
// The same device may be listed more than once (used by the tests on a 1-GPU box).
class DeviceSet {
  public:
    explicit DeviceSet(const std::vector<int> &devices);
    size_t size() const { return devs_.size(); }
    Device &operator[](size_t i) { return *devs_[i]; }

  private:
    std::vector<std::unique_ptr<Device>> devs_;
};

namespace distances {

// Multi-GPU forms of the dense / sparse drivers (same results as the single-device ones).
DistanceMatrix self_dists_all(DeviceSet &devs, const MultiSketch &sketches, size_t n, const DistType &dist_type,
                              bool quiet, const std::vector<double> *completeness_vec,
                              double completeness_cutoff);
DistanceMatrix cross_dists_all(DeviceSet &devs, const MultiSketch &ref_sketches,
                               const MultiSketch &query_sketches, size_t n, size_t n_query,
                               const DistType &dist_type, bool quiet,
                               const std::vector<double> *ref_completeness_vec,
                               const std::vector<double> *query_completeness_vec,
                               double completeness_cutoff);
SparseDistanceMatrix self_dists_knn(DeviceSet &devs, const MultiSketch &sketches, size_t n, size_t knn,
                                    const DistType &dist_type, bool quiet,
                                    const std::vector<double> *completeness_vec,
                                    double completeness_cutoff);
SparseDistanceMatrix cross_dists_knn(DeviceSet &devs, const MultiSketch &ref_sketches,
                                     const MultiSketch &query_sketches, size_t n, size_t n_query,
                                     size_t knn, const DistType &dist_type, bool quiet,
                                     const std::vector<double> *ref_completeness_vec,
                                     const std::vector<double> *query_completeness_vec,
                                     double completeness_cutoff);

// Streaming dense drivers (the reference muses about it at mod.rs:50-55):
// the matrix is produced in row bands of at most `band_bytes`, each band is written as text
// as soon as it is back on the host while the GPU computes the next one -- bounded host
// memory instead of n(n-1)/2 records, and formatting overlapped with compute.  Output is
// byte-identical to DistanceMatrix::write of the whole matrix.
// `npy` = write the matrix as a NumPy .npy v1.0 array instead of text: '<f4', C order, shape
// (n_pairs, ncols), rows in the same (condensed / ref-major) order as the text lines.
void self_dists_all_streamed(Device &dev, const MultiSketch &sketches, size_t n, const DistType &dist_type,
                             const std::vector<double> *completeness_vec, double completeness_cutoff,
                             TextSink &sink, size_t threads, size_t band_bytes, bool npy = false);
void cross_dists_all_streamed(Device &dev, const MultiSketch &ref_sketches, const MultiSketch &query_sketches,
                              size_t n, size_t n_query, const DistType &dist_type,
                              const std::vector<double> *ref_completeness_vec,
                              const std::vector<double> *query_completeness_vec, double completeness_cutoff,
                              TextSink &sink, size_t threads, size_t band_bytes, bool npy = false);
// The .npy v1.0 header of a (rows, cols) little-endian f32 array.
std::string npy_header(size_t rows, size_t cols);

// self_dists_knn_precluster (mod.rs:399-553): kNN restricted to the candidates an inverted
// index returns (any shared bin).  The candidate lists are built on the host from the index
// (`threads` workers), the distances and the per-row top-k run on the device
// (skl_self_dists_knn_candidates); rows without candidates follow --retain-unmatched.  If the
// index was loaded without its bitmaps (Inverted::load(prefix, false)) the candidate lists are
// built on the device too, from the .skq alone (skl_self_dists_knn_shared_bins).
// knn_ties = SKL_KNN_TIES_REFERENCE: ids and order of equal keys as the reference binary prints them -- its BinaryHeap
// replayed over each row's candidates in ascending .ski index (mod.rs:459-487).  Lists built on the device are ascending
// in .skd order: with a .ski that orders the samples differently the caller must load the index (host lists), or this
// throws.
enum class RetainUnmatched { None, Singleton, Bruteforce };
SparseDistanceMatrix self_dists_knn_precluster(Device &dev, const MultiSketch &sketches, const Inverted &inverted_index,
                                               const std::vector<uint16_t> &skq_bins, size_t skq_stride, size_t n,
                                               size_t knn, const DistType &dist_type,
                                               const std::vector<double> *completeness_vec,
                                               double completeness_cutoff, RetainUnmatched retain_unmatched,
                                               size_t threads, int knn_ties = 1 /* SKL_KNN_TIES_REFERENCE */);
// Whether the .ski lists the .skd's samples in the .skd's own order (then "ascending .ski index", the order the
// reference pushes a row's candidates in, is "ascending sample id", the order lists built on the device have).
bool ski_order_is_skd_order(const MultiSketch &sketches, const Inverted &inverted_index);

// mod.rs:25-37.  Throws std::runtime_error("K-mer size {k} not found in file").
DistType set_k(const MultiSketch &sketches, std::optional<size_t> kmer, bool ani);

// mod.rs:58-130
DistanceMatrix self_dists_all(Device &dev, const MultiSketch &sketches, size_t n, const DistType &dist_type,
                              bool quiet, const std::vector<double> *completeness_vec,
                              double completeness_cutoff);
// mod.rs:133-224
SparseDistanceMatrix self_dists_knn(Device &dev, const MultiSketch &sketches, size_t n, size_t knn,
                                    const DistType &dist_type, bool quiet,
                                    const std::vector<double> *completeness_vec,
                                    double completeness_cutoff);
// mod.rs:227-297
DistanceMatrix cross_dists_all(Device &dev, const MultiSketch &ref_sketches,
                               const MultiSketch &query_sketches, size_t n, size_t n_query,
                               const DistType &dist_type, bool quiet,
                               const std::vector<double> *ref_completeness_vec,
                               const std::vector<double> *query_completeness_vec,
                               double completeness_cutoff);
// mod.rs:306-395
SparseDistanceMatrix cross_dists_knn(Device &dev, const MultiSketch &ref_sketches,
                                     const MultiSketch &query_sketches, size_t n, size_t n_query,
                                     size_t knn, const DistType &dist_type, bool quiet,
                                     const std::vector<double> *ref_completeness_vec,
                                     const std::vector<double> *query_completeness_vec,
                                     double completeness_cutoff);

}  // namespace distances
}  // namespace skl_host
