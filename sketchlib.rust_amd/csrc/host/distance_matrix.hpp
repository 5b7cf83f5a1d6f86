// distance_matrix.hpp -- C++ mirror of src/distances/distance_matrix.rs: DistType,
// the dense / sparse result containers and their text output (the reference's Display
// impls, distance_matrix.rs:175-209 and :354-401).
#pragma once

#include <cstdint>
#include <iosfwd>
#include <optional>
#include <string>
#include <vector>

#include "multisketch.hpp"   // DefaultInitAllocator

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

// Wall-clock split of the dense text path (seconds), reported by `sketchlib dist` when
// SKL_CLI_TIMING is set: time formatting blocks, time handing them to the sink, time the
// writer waited for the GPU band.
struct OutputTiming {
    double format_s = 0, sink_s = 0, wait_s = 0;
};
OutputTiming &output_timing();

// Where formatted text blocks go.  A stream takes them one after the other; a regular file
// takes them at computed offsets from several threads at once (page-cache copies of a multi-GB
// listing are otherwise the serial tail of `sketchlib dist -o`).
class TextSink {
  public:
    virtual ~TextSink() = default;
    // Called once per text block, in output order, one caller at a time.  Returns a token
    // for finish().
    virtual uint64_t begin(const char *p, size_t len) = 0;
    // Called after begin() for the same block, from any thread, in any order.
    virtual void finish(uint64_t /*token*/, const char * /*p*/, size_t /*len*/) {}
    // true: the token is a byte offset and finish(token + d, p + d, len - d) may be called piecewise (a regular file)
    virtual bool positional() const { return false; }
};
// `len` raw bytes to the sink (a band of a .npy array).
void write_raw(TextSink &sink, const char *bytes, size_t len);
class StreamSink : public TextSink {
  public:
    explicit StreamSink(std::ostream &os) : os_(os) {}
    uint64_t begin(const char *p, size_t len) override;
  private:
    std::ostream &os_;
};
class FileSink : public TextSink {
  public:
    explicit FileSink(const std::string &path);   // throws std::runtime_error if it cannot be created
    ~FileSink() override;
    uint64_t begin(const char *p, size_t len) override;
    void finish(uint64_t token, const char *p, size_t len) override;
    bool positional() const override { return true; }
  private:
    int fd_ = -1;
    uint64_t offset_ = 0;
};

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

    // Streaming form: write reference rows [r0, r1) whose distances start at `band`
    // (the slice of the full matrix that skl_*_dists_rows produces for that row range).
    void write_rows(TextSink &sink, size_t r0, size_t r1, const float *band, size_t threads) const;
};

// CPUs this process may use (hardware threads, affinity mask, cgroup quota): output workers are capped at twice this
size_t host_cpu_budget();
void testing_set_host_cpu_budget(size_t n);   // (tests: 0 = measure; n = pretend)

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
    // (resize() leaves new records uninitialised: 50 M of them at a million samples are filled by several threads, and
    // value-initialising 800 MB first is a second, single-threaded pass over the memory)
    std::vector<SparseJaccard, DefaultInitAllocator<SparseJaccard>> jaccard_dists;   // DistVec::Jaccard
    std::vector<SparseCoreAcc, DefaultInitAllocator<SparseCoreAcc>> coreacc_dists;   // DistVec::CoreAcc
    std::vector<std::string> ref_names;
    std::optional<std::vector<std::string>> query_names;

    void write(std::ostream &os) const;
    // block-parallel form (same bytes): `threads` formatters, ordered hand-over to the sink
    void write(TextSink &sink, size_t threads) const;
};

}  // namespace skl_host
