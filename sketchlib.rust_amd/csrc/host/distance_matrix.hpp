// distance_matrix.hpp -- C++ mirror of src/distances/distance_matrix.rs: DistType,
// the dense / sparse result containers and their text output (the reference's Display
// impls, distance_matrix.rs:175-209 and :354-401).
#pragma once

#include <cstdint>
#include <iosfwd>
#include <optional>
#include <string>
#include <vector>

namespace skl_host {

// distance_matrix.rs:11-51
inline size_t square_to_condensed(size_t i, size_t j, size_t n)
{
    return n * i - ((i * (i + 1)) >> 1) + j - 1 - i;
}
inline std::pair<size_t, size_t> calc_query_indices(size_t k, size_t n) { return {k / n, k % n}; }
size_t calc_row_idx(size_t k, size_t n);
size_t calc_col_idx(size_t k, size_t i, size_t n);

// DistType, distance_matrix.rs:55-77
struct DistType {
    enum Kind { Jaccard, CoreAcc } kind = CoreAcc;
    size_t k_idx = 0;
    double k = 0.0;
    bool ani = false;
    std::string describe() const;  // the Display impl (logged by set_k)
    size_t n_dist_cols() const { return kind == CoreAcc ? 2 : 1; }
};

// Rust's `{}` for f32: shortest digits that round-trip, positional notation.
std::string format_f32(float v);

// DistanceMatrix, distance_matrix.rs:120-209
class DistanceMatrix {
  public:
    size_t n_distances = 0;
    DistType jaccard;
    std::vector<float> distances;
    std::vector<std::string> ref_names;
    std::optional<std::vector<std::string>> query_names;

    // The reference's Display impl (distance_matrix.rs:175-209).  `threads` > 1 formats row
    // blocks concurrently into per-thread buffers and writes them in order -- the reference
    // formats single-threaded, which dwarfs the GPU time at large n.
    void write(std::ostream &os, size_t threads = 1) const;
};

struct SparseJaccard {   // distance_matrix.rs:214
    size_t idx;
    float dist;
};
struct SparseCoreAcc {   // distance_matrix.rs:243
    size_t idx;
    float core, acc;
};

// SparseDistanceMatrix, distance_matrix.rs:285-401
class SparseDistanceMatrix {
  public:
    size_t n_distances = 0;
    size_t knn = 0;
    DistType jaccard;
    std::vector<SparseJaccard> jaccard_dists;   // DistVec::Jaccard
    std::vector<SparseCoreAcc> coreacc_dists;   // DistVec::CoreAcc
    std::vector<std::string> ref_names;
    std::optional<std::vector<std::string>> query_names;

    void write(std::ostream &os) const;
};

}  // namespace skl_host
