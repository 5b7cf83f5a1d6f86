#include "distances.hpp"

#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <deque>
#include <cstdlib>
#include <algorithm>
#include <atomic>
#include <memory>
#include <mutex>
#include <unordered_map>
#include <exception>
#include <ostream>
#include <string>
#include <thread>

#include "../../../include/sketchlib_dist.h"

namespace skl_host {

static void check(int rc)
{
    if (rc == SKL_OK) return;
    const std::string msg = skl_last_error();
    // reference panics keep their panic status (exit 101); the rest are ordinary errors
    if (rc == SKL_ERR_KMER_COUNT || rc == SKL_ERR_EMPTY_DB) throw Panic(msg);
    throw std::runtime_error(msg);
}

Device::Device(int device) { check(skl_ctx_create(device, &ctx_)); }
Device::~Device() { skl_ctx_destroy(ctx_); }

DeviceSet::DeviceSet(const std::vector<int> &devices)
{
    if (devices.empty()) throw std::runtime_error("no devices given");
    for (int d : devices) devs_.push_back(std::make_unique<Device>(d));
}

namespace {
// the result arrays of the C ABI: written whole by the call, never read before
template <class T>
using RawVec = std::vector<T, DefaultInitAllocator<T>>;

// records [0, n) by several threads (a million samples x 50 neighbours: 50 M records)
template <class Fill>
void fill_records(size_t n, const Fill &fill)
{
    const size_t n_threads = n >= (1u << 20) ? std::min<size_t>(16, std::max(1u, std::thread::hardware_concurrency())) : 1;
    std::vector<std::thread> pool;
    for (size_t t = 1; t < n_threads; ++t) pool.emplace_back([&, t] { fill(n * t / n_threads, n * (t + 1) / n_threads); });
    fill(0, n / n_threads);
    for (auto &th : pool) th.join();
}

SparseDistanceMatrix assemble_knn(const DistType &dist_type, size_t knn, const uint64_t *idx, const float *d0, const float *d1, size_t n_records)
{
    SparseDistanceMatrix out;
    out.jaccard = dist_type;
    out.knn = knn;
    out.n_distances = n_records;
    if (dist_type.kind == DistType::CoreAcc) {
        out.coreacc_dists.resize(n_records);
        fill_records(n_records, [&](size_t a, size_t b) {
            for (size_t i = a; i < b; ++i) out.coreacc_dists[i] = {(size_t)idx[i], d0[i], d1[i]};
        });
    } else {
        out.jaccard_dists.resize(n_records);
        fill_records(n_records, [&](size_t a, size_t b) {
            for (size_t i = a; i < b; ++i) out.jaccard_dists[i] = {(size_t)idx[i], d0[i]};
        });
    }
    return out;
}

// RAII for a device-resident MultiSketch
struct Slab {
    skl_sketches *h = nullptr;
    Slab(Device &dev, const MultiSketch &m, const std::vector<double> *comp)
    {
        const size_t n = m.number_samples_loaded();
        check(skl_sketches_create(dev.ctx(), m.bins().data(), 0, n, m.kmer_lengths().size(),
                                  m.kmer_lengths().data(), (size_t)m.sketchsize64, &h));
        if (comp) check(skl_sketches_set_completeness(h, comp->data()));
    }
    ~Slab() { skl_sketches_destroy(h); }
};

skl_dist_params to_params(const DistType &d, double cutoff)
{
    skl_dist_params p;
    p.dist_type = d.kind == DistType::CoreAcc ? SKL_DIST_COREACC : SKL_DIST_JACCARD;
    p.ani = d.ani ? 1 : 0;
    p.k_idx = d.k_idx;
    p.completeness_cutoff = cutoff;
    return p;
}

std::vector<std::string> sketch_names(const MultiSketch &m)
{
    std::vector<std::string> names;
    for (size_t i = 0; i < m.number_samples_loaded(); ++i) names.push_back(m.sketch_name(i));
    return names;
}
}  // namespace

namespace distances {

DistType set_k(const MultiSketch &sketches, std::optional<size_t> kmer, bool ani)
{
    DistType d;
    if (kmer) {
        const auto k_idx = sketches.get_k_idx(*kmer);
        if (!k_idx) throw std::runtime_error("K-mer size " + std::to_string(*kmer) + " not found in file");
        d.kind = DistType::Jaccard;
        d.k_idx = *k_idx;
        d.k = (double)*kmer;
        d.ani = ani;
    } else {
        d.kind = DistType::CoreAcc;
    }
    return d;
}

DistanceMatrix self_dists_all(Device &dev, const MultiSketch &sketches, size_t n, const DistType &dist_type,
                              bool /*quiet*/, const std::vector<double> *completeness_vec,
                              double completeness_cutoff)
{
    DistanceMatrix out;  // DistanceMatrix::new, distance_matrix.rs:132-158
    out.jaccard = dist_type;
    out.ref_names = sketch_names(sketches);
    out.n_distances = n * (n - 1) / 2;
    out.distances.assign(out.n_distances * dist_type.n_dist_cols(), 0.0f);
    Slab s(dev, sketches, completeness_vec);
    const skl_dist_params p = to_params(dist_type, completeness_cutoff);
    check(skl_self_dists_all(dev.ctx(), s.h, &p, out.distances.data(), 0));
    return out;
}

DistanceMatrix cross_dists_all(Device &dev, const MultiSketch &ref_sketches,
                               const MultiSketch &query_sketches, size_t n, size_t n_query,
                               const DistType &dist_type, bool /*quiet*/,
                               const std::vector<double> *ref_completeness_vec,
                               const std::vector<double> *query_completeness_vec,
                               double completeness_cutoff)
{
    DistanceMatrix out;
    out.jaccard = dist_type;
    out.ref_names = sketch_names(ref_sketches);
    out.query_names = sketch_names(query_sketches);
    out.n_distances = n * n_query;
    out.distances.assign(out.n_distances * dist_type.n_dist_cols(), 0.0f);
    Slab r(dev, ref_sketches, ref_completeness_vec);
    Slab q(dev, query_sketches, query_completeness_vec);
    const skl_dist_params p = to_params(dist_type, completeness_cutoff);
    check(skl_cross_dists_all(dev.ctx(), r.h, q.h, &p, out.distances.data(), 0));
    return out;
}

static SparseDistanceMatrix run_knn(Device &dev, skl_sketches *ref, skl_sketches *query, size_t rows,
                                    size_t knn, const DistType &dist_type, double cutoff)
{
    const size_t n_records = rows * knn;
    RawVec<uint64_t> idx(n_records);
    RawVec<float> d0(n_records), d1(n_records);
    const skl_dist_params p = to_params(dist_type, cutoff);
    if (query) {
        check(skl_cross_dists_knn(dev.ctx(), ref, query, &p, knn, idx.data(), d0.data(), d1.data(), 0));
    } else {
        check(skl_self_dists_knn(dev.ctx(), ref, &p, knn, idx.data(), d0.data(), d1.data(), 0));
    }
    return assemble_knn(dist_type, knn, idx.data(), d0.data(), d1.data(), n_records);
}

SparseDistanceMatrix self_dists_knn(Device &dev, const MultiSketch &sketches, size_t n, size_t knn,
                                    const DistType &dist_type, bool /*quiet*/,
                                    const std::vector<double> *completeness_vec,
                                    double completeness_cutoff)
{
    Slab s(dev, sketches, completeness_vec);
    SparseDistanceMatrix out = run_knn(dev, s.h, nullptr, n, knn, dist_type, completeness_cutoff);
    out.ref_names = sketch_names(sketches);
    return out;
}

SparseDistanceMatrix cross_dists_knn(Device &dev, const MultiSketch &ref_sketches,
                                     const MultiSketch &query_sketches, size_t n, size_t n_query,
                                     size_t knn, const DistType &dist_type, bool /*quiet*/,
                                     const std::vector<double> *ref_completeness_vec,
                                     const std::vector<double> *query_completeness_vec,
                                     double completeness_cutoff)
{
    if (n == 0) throw Panic("Reference database has no loaded samples");   // mod.rs:318-320
    if (n_query == 0) throw Panic("Query database has no loaded samples");  // mod.rs:321-323
    knn = std::min(knn, n);                                                 // mod.rs:325
    Slab r(dev, ref_sketches, ref_completeness_vec);
    Slab q(dev, query_sketches, query_completeness_vec);
    SparseDistanceMatrix out = run_knn(dev, r.h, q.h, n_query, knn, dist_type, completeness_cutoff);
    out.ref_names = sketch_names(ref_sketches);
    out.query_names = sketch_names(query_sketches);
    return out;
}

// ---------------------------------------------------------------------------
// precluster
// ---------------------------------------------------------------------------

SparseDistanceMatrix self_dists_knn_precluster(Device &dev, const MultiSketch &sketches, const Inverted &inv,
                                               const std::vector<uint16_t> &skq_bins, size_t skq_stride, size_t n,
                                               size_t knn, const DistType &dist_type,
                                               const std::vector<double> *completeness_vec,
                                               double completeness_cutoff, RetainUnmatched retain, size_t threads, int knn_ties)
{
    // sample sets of .ski and .skm must agree; i <-> j lookups (mod.rs:412-440)
    std::unordered_map<std::string, size_t> skq_lookup;
    for (size_t i = 0; i < inv.sample_names.size(); ++i) skq_lookup.emplace(inv.sample_names[i], i);
    std::vector<size_t> ski_of_skd;
    std::string not_found;
    for (size_t i = 0; i < sketches.number_samples_loaded(); ++i) {
        const auto it = skq_lookup.find(sketches.sketch_name(i));
        if (it != skq_lookup.end()) ski_of_skd.push_back(it->second);
        else not_found += (not_found.empty() ? "\"" : ", \"") + sketches.sketch_name(i) + "\"";
    }
    if (!not_found.empty()) {
        throw Panic("The following samples in the .skd could not be found in the .ski:\n[" + not_found + "]");
    }
    std::vector<size_t> skd_of_ski(std::max(n, inv.sample_names.size()), 0);
    for (size_t i = 0; i < ski_of_skd.size(); ++i) skd_of_ski[ski_of_skd[i]] = i;
    if (dist_type.kind == DistType::CoreAcc) {
        throw Panic("not implemented: Prefilter only available for single k-mer distances");   // mod.rs:549-551
    }

    const bool timing = std::getenv("SKL_CLI_TIMING") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    auto since = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count(); };
    const bool reference_order = knn_ties == SKL_KNN_TIES_REFERENCE;
    check(skl_ctx_set_knn_ties(dev.ctx(), knn_ties));
    bool monotone = true;   // does ascending .skd id mean ascending .ski index?
    for (size_t i = 1; i < ski_of_skd.size(); ++i) monotone = monotone && ski_of_skd[i - 1] < ski_of_skd[i];
    if (reference_order && !monotone && !inv.has_index()) {
        throw std::runtime_error("the reference's tie order pushes a row's candidates in ascending .ski index, and this .ski orders the "
                                 "samples differently from the .skd: load the index (host candidate lists)");
    }
    Slab s(dev, sketches, completeness_vec);
    const skl_dist_params p = to_params(dist_type, completeness_cutoff);
    std::vector<uint64_t> idx(n * knn);
    std::vector<float> d0(n * knn), d1;
    std::vector<uint8_t> unmatched(n, 0);
    if (!inv.has_index()) {
        // candidate lists on the device: index sketches re-ordered to .skd order
        std::vector<uint16_t> skq_skd(n * skq_stride);
        for (size_t i = 0; i < n; ++i) {
            std::copy(skq_bins.begin() + ski_of_skd[i] * skq_stride, skq_bins.begin() + (ski_of_skd[i] + 1) * skq_stride,
                      skq_skd.begin() + i * skq_stride);
        }
        uint64_t total = 0;
        check(skl_self_dists_knn_shared_bins(dev.ctx(), s.h, &p, knn, skq_skd.data(), skq_stride, idx.data(), d0.data(),
                                             &total));
        // a row whose first entry is padding (itself at distance 1) had no candidate at all
        for (size_t i = 0; i < n; ++i) unmatched[i] = idx[i * knn] == i;
        if (timing) {
            std::fprintf(stderr, "TIMING precluster: %llu candidate pairs; device candidate search + distances + top-k=%.3fs\n",
                         (unsigned long long)total, since());
        }
    } else {
    // candidate lists (inverted.rs:259-268), in .skd ids, ascending, self excluded.  Every
    // worker appends its rows' lists to ONE growing buffer of its own (a few large allocations
    // instead of one per row: 64 threads faulting in millions of small blocks serialise on the
    // address-space lock) and records where each row went.
    const size_t n_workers = std::max<size_t>(1, threads);
    struct RowRef {
        uint32_t worker = 0;
        uint32_t len = 0;
        uint64_t begin = 0;
    };
    std::vector<RowRef> rows(n);
    std::vector<std::vector<uint32_t>> arena(n_workers);
    {
        std::atomic<size_t> next{0};
        std::exception_ptr err;
        std::mutex mu;
        auto work = [&](size_t wid) {
            try {
                std::vector<uint32_t> stamp(inv.n_samples, 0), hits;   // per-thread scratch
                std::vector<uint32_t> &buf = arena[wid];
                uint32_t epoch = 0;
                for (;;) {
                    const size_t i0 = next.fetch_add(256);
                    if (i0 >= n) break;
                    for (size_t i = i0; i < std::min(n, i0 + 256); ++i) {
                        const size_t ski_i = ski_of_skd[i];
                        if (++epoch == 0) {   // wrapped: start over
                            std::fill(stamp.begin(), stamp.end(), 0u);
                            epoch = 1;
                        }
                        inv.any_shared_bins(skq_bins.data() + ski_i * skq_stride, stamp, epoch, hits);
                        const size_t begin = buf.size();
                        for (uint32_t j : hits) {   // (ascending .ski index)
                            if (j != ski_i) buf.push_back((uint32_t)skd_of_ski[j]);   // mod.rs:458-461
                        }
                        // canonical ties: ascending .skd id (the lists are a set); the reference's tie order: the order the
                        // reference pushes them in, i.e. as any_shared_bins returned them
                        if (!reference_order) std::sort(buf.begin() + begin, buf.end());
                        rows[i].worker = (uint32_t)wid;
                        rows[i].begin = begin;
                        rows[i].len = (uint32_t)(buf.size() - begin);
                    }
                }
            } catch (...) {
                std::lock_guard<std::mutex> lk(mu);
                err = std::current_exception();
            }
        };
        std::vector<std::thread> pool;
        for (size_t t = 1; t < n_workers; ++t) pool.emplace_back(work, t);
        work(0);
        for (auto &t : pool) t.join();
        if (err) std::rethrow_exception(err);
    }
    const double t_lists = since();
    std::vector<uint64_t> offsets(n + 1, 0);
    for (size_t i = 0; i < n; ++i) offsets[i + 1] = offsets[i] + rows[i].len;
    std::unique_ptr<uint32_t[]> cand(new uint32_t[std::max<uint64_t>(offsets[n], 1)]);   // not zero-filled
    {
        std::atomic<size_t> next{0};
        auto copy_work = [&] {
            for (;;) {
                const size_t i0 = next.fetch_add(1024);
                if (i0 >= n) break;
                for (size_t i = i0; i < std::min(n, i0 + 1024); ++i) {
                    const uint32_t *src = arena[rows[i].worker].data() + rows[i].begin;
                    std::copy(src, src + rows[i].len, cand.get() + offsets[i]);
                }
            }
        };
        std::vector<std::thread> pool;
        for (size_t t = 1; t < n_workers; ++t) pool.emplace_back(copy_work);
        copy_work();
        for (auto &t : pool) t.join();
    }
    std::vector<std::vector<uint32_t>>().swap(arena);
    for (size_t i = 0; i < n; ++i) unmatched[i] = rows[i].len == 0;
    const double t_csr = since();

    const double t_slab = since();
    check(skl_self_dists_knn_candidates(dev.ctx(), s.h, &p, knn, offsets.data(), cand.get(), idx.data(), d0.data()));
    if (timing) {
        std::fprintf(stderr, "TIMING precluster: %llu candidate pairs; lists=%.3fs csr=%.3fs slab upload=%.3fs "
                             "candidate upload+distances+top-k=%.3fs\n",
                     (unsigned long long)offsets[n], t_lists, t_csr - t_lists, t_slab - t_csr, since() - t_slab);
    }

    }

    // genomes without a prefilter match (mod.rs:487-527)
    if (retain != RetainUnmatched::None) {
        std::vector<uint64_t> bi(knn);
        std::vector<float> b0(knn), b1(knn);
        for (size_t i = 0; i < n; ++i) {
            if (!unmatched[i]) continue;
            if (retain == RetainUnmatched::Singleton) {
                for (size_t t = 0; t < knn; ++t) {
                    idx[i * knn + t] = i;
                    d0[i * knn + t] = t == 0 ? 0.0f : 1.0f;
                }
            } else {
                check(skl_self_dists_knn_rows(dev.ctx(), s.h, &p, knn, i, i + 1, bi.data(), b0.data(), b1.data(), 0));
                std::copy(bi.begin(), bi.end(), idx.begin() + i * knn);
                std::copy(b0.begin(), b0.end(), d0.begin() + i * knn);
            }
        }
    }
    SparseDistanceMatrix out = assemble_knn(dist_type, knn, idx.data(), d0.data(), d1.data(), idx.size());
    out.ref_names = sketch_names(sketches);
    return out;
}

bool ski_order_is_skd_order(const MultiSketch &sketches, const Inverted &inv)
{
    std::unordered_map<std::string, size_t> pos;
    for (size_t i = 0; i < inv.sample_names.size(); ++i) pos.emplace(inv.sample_names[i], i);
    size_t last = 0;
    for (size_t i = 0; i < sketches.number_samples_loaded(); ++i) {
        const auto it = pos.find(sketches.sketch_name(i));
        if (it == pos.end()) return true;   // (reported by self_dists_knn_precluster itself)
        if (i && it->second <= last) return false;
        last = it->second;
    }
    return true;
}

// ---------------------------------------------------------------------------
// several devices: contiguous row bands, one host thread per device
// ---------------------------------------------------------------------------

namespace {
// rows [0, n) of the condensed triangle split into `parts` bands of ~equal pair count
std::vector<size_t> self_row_bounds(size_t n, size_t parts)
{
    const size_t total = n * (n - 1) / 2;
    std::vector<size_t> b = {0};
    size_t row = 0, acc = 0;
    for (size_t w = 1; w < parts; ++w) {
        const size_t target = (total * w + parts - 1) / parts;
        while (row < n && acc < target) {
            acc += n - 1 - row;
            ++row;
        }
        b.push_back(row);
    }
    b.push_back(n);
    return b;
}

std::vector<size_t> even_bounds(size_t rows, size_t parts)
{
    std::vector<size_t> b;
    for (size_t w = 0; w <= parts; ++w) b.push_back(rows * w / parts);
    return b;
}

// run fn(device index) on one thread per device; rethrow the first failure
template <class F>
void for_each_device(DeviceSet &devs, F fn)
{
    std::vector<std::exception_ptr> errs(devs.size());
    std::vector<std::thread> pool;
    for (size_t d = 0; d < devs.size(); ++d) {
        pool.emplace_back([&, d] {
            try {
                fn(d);
            } catch (...) {
                errs[d] = std::current_exception();
            }
        });
    }
    for (auto &t : pool) t.join();
    for (auto &e : errs) {
        if (e) std::rethrow_exception(e);
    }
}
}  // namespace

DistanceMatrix self_dists_all(DeviceSet &devs, const MultiSketch &sketches, size_t n, const DistType &dist_type,
                              bool quiet, const std::vector<double> *completeness_vec,
                              double completeness_cutoff)
{
    if (devs.size() == 1) return self_dists_all(devs[0], sketches, n, dist_type, quiet, completeness_vec, completeness_cutoff);
    DistanceMatrix out;
    out.jaccard = dist_type;
    out.ref_names = sketch_names(sketches);
    out.n_distances = n * (n - 1) / 2;
    const size_t ncols = dist_type.n_dist_cols();
    out.distances.assign(out.n_distances * ncols, 0.0f);
    const skl_dist_params p = to_params(dist_type, completeness_cutoff);
    const std::vector<size_t> b = self_row_bounds(n, devs.size());
    for_each_device(devs, [&](size_t d) {
        if (b[d + 1] <= b[d]) return;
        Slab s(devs[d], sketches, completeness_vec);
        const size_t first = b[d] + 1 < n ? square_to_condensed(b[d], b[d] + 1, n) : out.n_distances;
        check(skl_self_dists_rows(devs[d].ctx(), s.h, &p, b[d], b[d + 1], out.distances.data() + first * ncols, 0));
    });
    return out;
}

DistanceMatrix cross_dists_all(DeviceSet &devs, const MultiSketch &ref_sketches,
                               const MultiSketch &query_sketches, size_t n, size_t n_query,
                               const DistType &dist_type, bool quiet,
                               const std::vector<double> *ref_completeness_vec,
                               const std::vector<double> *query_completeness_vec,
                               double completeness_cutoff)
{
    if (devs.size() == 1) {
        return cross_dists_all(devs[0], ref_sketches, query_sketches, n, n_query, dist_type, quiet,
                               ref_completeness_vec, query_completeness_vec, completeness_cutoff);
    }
    DistanceMatrix out;
    out.jaccard = dist_type;
    out.ref_names = sketch_names(ref_sketches);
    out.query_names = sketch_names(query_sketches);
    out.n_distances = n * n_query;
    const size_t ncols = dist_type.n_dist_cols();
    out.distances.assign(out.n_distances * ncols, 0.0f);
    const skl_dist_params p = to_params(dist_type, completeness_cutoff);
    const std::vector<size_t> b = even_bounds(n, devs.size());
    for_each_device(devs, [&](size_t d) {
        if (b[d + 1] <= b[d]) return;
        Slab r(devs[d], ref_sketches, ref_completeness_vec);
        Slab q(devs[d], query_sketches, query_completeness_vec);
        check(skl_cross_dists_rows(devs[d].ctx(), r.h, q.h, &p, b[d], b[d + 1],
                                   out.distances.data() + b[d] * n_query * ncols, 0));
    });
    return out;
}


SparseDistanceMatrix self_dists_knn(DeviceSet &devs, const MultiSketch &sketches, size_t n, size_t knn,
                                    const DistType &dist_type, bool quiet,
                                    const std::vector<double> *completeness_vec,
                                    double completeness_cutoff)
{
    if (devs.size() == 1) return self_dists_knn(devs[0], sketches, n, knn, dist_type, quiet, completeness_vec, completeness_cutoff);
    std::vector<uint64_t> idx(n * knn);
    std::vector<float> d0(n * knn), d1(n * knn);
    const skl_dist_params p = to_params(dist_type, completeness_cutoff);
    const std::vector<size_t> b = even_bounds(n, devs.size());
    // THE REFERENCE'S TIE ORDER over several devices, every pair once: partial heaps of disjoint candidate sets do not merge (a
    // BinaryHeap's state depends on the order its candidates arrive in), so the heaps TRAVEL: device d owns the column window
    // [cut[d], cut[d + 1]), evaluates the pairs whose column lies in it band by band (skl_self_dists_knn_window), and hands each
    // band's heaps to device d + 1 as soon as the band is done; the last device ends up with every row's final heap.  Decided up
    // front: nothing is uploaded for a form the configuration does not have.
    // DECOUPLED WINDOWS first (round 6; the header's skl_self_dists_knn_window_logged): every device runs its column window against
    // heaps it has cleared itself and logs what they take; no device waits for another.  The logs pass through host memory
    // (this layer sees only the C ABI), so the form is taken while they fit comfortably -- 16 knn entries per row and device --
    // and falls through to the travelling heaps when a log overflows or the logs would not fit.
    if (skl_ctx_get_knn_ties(devs[0].ctx()) == SKL_KNN_TIES_REFERENCE && knn <= 2048 && n >= 2) {
        const size_t W = devs.size();
        const bool coreacc = p.dist_type == SKL_DIST_COREACC;
        const size_t rec_w = coreacc ? 2 : 1, cap = std::max<size_t>(64, 16 * knn);
        const char *off = std::getenv("SKL_KNN_DECOUPLED");   // (=0: the travelling heaps, for tests and A/B timing)
        if (!(off && off[0] == '0') && n * cap * (rec_w + 1) * 4 <= (4ull << 30)) {
            struct DevMem {
                skl_ctx *c;
                void *p = nullptr;
                DevMem(skl_ctx *ctx_, size_t bytes) : c(ctx_) { check(skl_device_malloc(c, bytes, &p)); }
                ~DevMem() { skl_device_free(c, p); }
                DevMem(const DevMem &) = delete;
                DevMem &operator=(const DevMem &) = delete;
            };
            std::vector<std::vector<float>> log_rec(W);
            std::vector<std::vector<uint32_t>> log_id(W), log_len(W);
            std::vector<size_t> cut;
            size_t band_rows = 0;
            std::mutex plan_m;
            std::atomic<bool> give_up{false};
            for_each_device(devs, [&](size_t d) {
                Slab s(devs[d], sketches, completeness_vec);
                skl_ctx *ctx = devs[d].ctx();
                {
                    std::lock_guard<std::mutex> g(plan_m);
                    if (band_rows == 0) {   // (every device arrives at the same plan)
                        band_rows = skl_knn_band_rows(s.h, &p, W);
                        if (band_rows == 0) throw std::runtime_error("skl_knn_band_rows failed");
                        cut.assign(1, 0);
                        for (size_t r = 1; r < W; ++r) {   // n * sqrt(r / W) on band boundaries: equal pair counts
                            const size_t c = (size_t)std::llround((double)n * std::sqrt((double)r / (double)W) / (double)band_rows) * band_rows;
                            cut.push_back(std::min(std::max(c, cut.back()), n));
                        }
                        cut.push_back(n);
                    }
                }
                const size_t lo = cut[d], hi = cut[d + 1];
                if (hi == 0) return;
                DevMem h_key(ctx, n * knn * 4), h_id(ctx, n * knn * 4), h_d1(ctx, coreacc ? n * knn * 4 : 4), h_len(ctx, n * 4), thr(ctx, n * 4);
                DevMem l_rec(ctx, hi * cap * rec_w * 4), l_id(ctx, hi * cap * 4), l_len(ctx, hi * 4);
                check(skl_knn_heaps_clear(ctx, 0, n, (uint32_t *)h_len.p, (uint32_t *)thr.p));
                {
                    const std::vector<uint32_t> zeros(hi, 0u);
                    check(skl_device_memcpy(ctx, l_len.p, zeros.data(), hi * 4, 1));
                }
                const size_t n_bands = (n + band_rows - 1) / band_rows;
                for (size_t band = 0; band < n_bands && !give_up; ++band) {
                    if (band * band_rows >= hi) break;
                    // (the log arrays cover rows [0, hi): no row at or behind the window's end meets a pair of it)
                    const int r = skl_self_dists_knn_window_logged(ctx, s.h, &p, knn, band_rows, band, lo, hi, (float *)h_key.p, (uint32_t *)h_id.p,
                                                                   coreacc ? (float *)h_d1.p : nullptr, (uint32_t *)h_len.p, (uint32_t *)thr.p,
                                                                   (float *)l_rec.p, (uint32_t *)l_id.p, (uint32_t *)l_len.p, cap);
                    if (r == SKL_ERR_INVALID_ARG && band == 0) {   // no one-evaluation form for this configuration
                        give_up = true;
                        return;
                    }
                    check(r);
                }
                log_len[d].resize(hi);
                check(skl_device_memcpy(ctx, log_len[d].data(), l_len.p, hi * 4, 0));
                for (const uint32_t v : log_len[d]) {
                    if (v > cap) give_up = true;   // an overflowed log is useless: the travelling heaps below
                }
                if (give_up) return;
                log_rec[d].resize(hi * cap * rec_w);
                log_id[d].resize(hi * cap);
                check(skl_device_memcpy(ctx, log_rec[d].data(), l_rec.p, hi * cap * rec_w * 4, 0));
                check(skl_device_memcpy(ctx, log_id[d].data(), l_id.p, hi * cap * 4, 0));
            });
            if (!give_up) {
                // every device replays the logs of ITS row shard in window order into empty heaps, and sorts them
                for_each_device(devs, [&](size_t d) {
                    const size_t r0 = b[d], r1 = b[d + 1], rows = r1 - r0;
                    if (rows == 0) return;
                    skl_ctx *ctx = devs[d].ctx();
                    DevMem h_key(ctx, rows * knn * 4), h_id(ctx, rows * knn * 4), h_d1(ctx, coreacc ? rows * knn * 4 : 4), h_len(ctx, rows * 4), thr(ctx, rows * 4);
                    DevMem u_rec(ctx, rows * cap * rec_w * 4), u_id(ctx, rows * cap * 4), u_len(ctx, rows * 4);
                    check(skl_knn_heaps_clear(ctx, 0, rows, (uint32_t *)h_len.p, (uint32_t *)thr.p));
                    for (size_t src = 0; src < W; ++src) {
                        const size_t a = r0, e = std::min(r1, cut[src + 1]);
                        if (e <= a) continue;   // (rows at or behind window src's end: nothing of it)
                        check(skl_device_memcpy(ctx, u_rec.p, log_rec[src].data() + a * cap * rec_w, (e - a) * cap * rec_w * 4, 1));
                        check(skl_device_memcpy(ctx, u_id.p, log_id[src].data() + a * cap, (e - a) * cap * 4, 1));
                        check(skl_device_memcpy(ctx, u_len.p, log_len[src].data() + a, (e - a) * 4, 1));
                        check(skl_knn_heaps_replay(ctx, e - a, knn, coreacc ? 1 : 0, (const float *)u_rec.p, (const uint32_t *)u_id.p, (const uint32_t *)u_len.p, cap,
                                                   (float *)h_key.p, (uint32_t *)h_id.p, coreacc ? (float *)h_d1.p : nullptr, (uint32_t *)h_len.p, (uint32_t *)thr.p));
                    }
                    DevMem o_idx(ctx, rows * knn * 8), o_d0(ctx, rows * knn * 4), o_d1(ctx, coreacc ? rows * knn * 4 : 4);
                    check(skl_knn_heaps_finalize(ctx, rows, knn, (const float *)h_key.p, (const uint32_t *)h_id.p, coreacc ? (const float *)h_d1.p : nullptr,
                                                 (const uint32_t *)h_len.p, p.ani, (uint64_t *)o_idx.p, (float *)o_d0.p, coreacc ? (float *)o_d1.p : nullptr));
                    check(skl_device_memcpy(ctx, idx.data() + r0 * knn, o_idx.p, rows * knn * 8, 0));
                    check(skl_device_memcpy(ctx, d0.data() + r0 * knn, o_d0.p, rows * knn * 4, 0));
                    if (coreacc) check(skl_device_memcpy(ctx, d1.data() + r0 * knn, o_d1.p, rows * knn * 4, 0));
                });
                SparseDistanceMatrix out = assemble_knn(dist_type, knn, idx.data(), d0.data(), d1.data(), idx.size());
                out.ref_names = sketch_names(sketches);
                return out;
            }
        }
    }
    if (skl_ctx_get_knn_ties(devs[0].ctx()) == SKL_KNN_TIES_REFERENCE && knn <= 2048 && n >= 2) {
        const size_t W = devs.size();
        const bool coreacc = p.dist_type == SKL_DIST_COREACC;
        const size_t words = (coreacc ? 3 : 2) * knn + 2;   // a row's heap on the wire: keys, ids[, second values], length, threshold
        struct Link {   // device d -> d + 1: bands in order
            std::mutex m;
            std::condition_variable cv;
            std::deque<std::vector<uint32_t>> q;
            bool broken = false;
        };
        std::vector<Link> link(W);
        std::vector<int> rc(W, SKL_OK);
        std::vector<std::string> msg(W);
        std::atomic<bool> no_form{false};
        size_t band_rows = 0;
        std::vector<size_t> cut;
        std::mutex plan_m;
        auto break_links = [&] {
            for (auto &l : link) {
                std::lock_guard<std::mutex> g(l.m);
                l.broken = true;
                l.cv.notify_all();
            }
        };
        for_each_device(devs, [&](size_t d) {
            try {
                Slab s(devs[d], sketches, completeness_vec);
                skl_ctx *ctx = devs[d].ctx();
                {
                    std::lock_guard<std::mutex> g(plan_m);
                    if (band_rows == 0) {   // (every device arrives at the same plan)
                        band_rows = skl_knn_band_rows(s.h, &p, W);
                        if (band_rows == 0) throw std::runtime_error("skl_knn_band_rows failed");
                        cut.assign(1, 0);
                        for (size_t r = 1; r < W; ++r) {   // n * sqrt(r / W) on band boundaries: equal pair counts
                            const size_t c = (size_t)std::llround((double)n * std::sqrt((double)r / (double)W) / (double)band_rows) * band_rows;
                            cut.push_back(std::min(std::max(c, cut.back()), n));
                        }
                        cut.push_back(n);
                    }
                }
                const size_t lo = cut[d], hi = cut[d + 1];
                struct DevMem {
                    skl_ctx *c;
                    void *p = nullptr;
                    DevMem(skl_ctx *ctx_, size_t bytes) : c(ctx_) { check(skl_device_malloc(c, bytes, &p)); }
                    ~DevMem() { skl_device_free(c, p); }
                };
                DevMem h_key(ctx, n * knn * 4), h_id(ctx, n * knn * 4), h_d1(ctx, coreacc ? n * knn * 4 : 4), h_len(ctx, n * 4), thr(ctx, n * 4);
                check(skl_knn_heaps_clear(ctx, 0, n, (uint32_t *)h_len.p, (uint32_t *)thr.p));
                auto row_ptr = [&](DevMem &m, size_t row, size_t per_row) { return (char *)m.p + row * per_row * 4; };
                auto export_rows = [&](size_t r0, size_t r1) {
                    const size_t rows = r1 - r0;
                    std::vector<uint32_t> buf(rows * words);
                    uint32_t *w = buf.data();
                    check(skl_device_memcpy(ctx, w, row_ptr(h_key, r0, knn), rows * knn * 4, 0));
                    w += rows * knn;
                    check(skl_device_memcpy(ctx, w, row_ptr(h_id, r0, knn), rows * knn * 4, 0));
                    w += rows * knn;
                    if (coreacc) {
                        check(skl_device_memcpy(ctx, w, row_ptr(h_d1, r0, knn), rows * knn * 4, 0));
                        w += rows * knn;
                    }
                    check(skl_device_memcpy(ctx, w, row_ptr(h_len, r0, 1), rows * 4, 0));
                    w += rows;
                    check(skl_device_memcpy(ctx, w, row_ptr(thr, r0, 1), rows * 4, 0));
                    return buf;
                };
                auto import_rows = [&](size_t r0, size_t r1, const std::vector<uint32_t> &buf) {
                    const size_t rows = r1 - r0;
                    const uint32_t *w = buf.data();
                    check(skl_device_memcpy(ctx, row_ptr(h_key, r0, knn), w, rows * knn * 4, 1));
                    w += rows * knn;
                    check(skl_device_memcpy(ctx, row_ptr(h_id, r0, knn), w, rows * knn * 4, 1));
                    w += rows * knn;
                    if (coreacc) {
                        check(skl_device_memcpy(ctx, row_ptr(h_d1, r0, knn), w, rows * knn * 4, 1));
                        w += rows * knn;
                    }
                    check(skl_device_memcpy(ctx, row_ptr(h_len, r0, 1), w, rows * 4, 1));
                    w += rows;
                    check(skl_device_memcpy(ctx, row_ptr(thr, r0, 1), w, rows * 4, 1));
                };
                const size_t n_bands = (n + band_rows - 1) / band_rows;
                for (size_t band = 0; band < n_bands && !no_form; ++band) {
                    const size_t b0 = band * band_rows, b1 = std::min(n, b0 + band_rows);
                    if (b0 >= hi) break;
                    if (b0 < lo) {   // rows of an earlier window: from the device that has just finished them
                        Link &in = link[d - 1];
                        std::unique_lock<std::mutex> g(in.m);
                        in.cv.wait(g, [&] { return !in.q.empty() || in.broken; });
                        if (in.q.empty()) {
                            if (no_form) return;   // (upstream found that the configuration has no one-evaluation form)
                            throw std::runtime_error("the device upstream of this one failed");
                        }
                        std::vector<uint32_t> buf = std::move(in.q.front());
                        in.q.pop_front();
                        g.unlock();
                        import_rows(b0, b1, buf);
                    }
                    const int r = skl_self_dists_knn_window(ctx, s.h, &p, knn, band_rows, band, lo, hi, (float *)h_key.p, (uint32_t *)h_id.p,
                                                            coreacc ? (float *)h_d1.p : nullptr, (uint32_t *)h_len.p, (uint32_t *)thr.p);
                    if (r == SKL_ERR_INVALID_ARG && band == 0) {   // no one-evaluation form for this configuration: row shards below
                        no_form = true;
                        break_links();
                        return;
                    }
                    check(r);
                    if (d + 1 < W) {
                        std::vector<uint32_t> buf = export_rows(b0, b1);
                        Link &out_l = link[d];
                        std::lock_guard<std::mutex> g(out_l.m);
                        out_l.q.push_back(std::move(buf));
                        out_l.cv.notify_all();
                    }
                }
                if (d + 1 == W && !no_form) {   // every heap has ended here: into_sorted_vec, lists to the host
                    DevMem o_idx(ctx, n * knn * 8), o_d0(ctx, n * knn * 4), o_d1(ctx, coreacc ? n * knn * 4 : 4);
                    check(skl_knn_heaps_finalize(ctx, n, knn, (const float *)h_key.p, (const uint32_t *)h_id.p, coreacc ? (const float *)h_d1.p : nullptr,
                                                 (const uint32_t *)h_len.p, p.ani, (uint64_t *)o_idx.p, (float *)o_d0.p, coreacc ? (float *)o_d1.p : nullptr));
                    check(skl_device_memcpy(ctx, idx.data(), o_idx.p, n * knn * 8, 0));
                    check(skl_device_memcpy(ctx, d0.data(), o_d0.p, n * knn * 4, 0));
                    if (coreacc) check(skl_device_memcpy(ctx, d1.data(), o_d1.p, n * knn * 4, 0));
                }
            } catch (...) {
                break_links();   // nobody waits for a band that will not come
                throw;
            }
        });
        if (!no_form) {
            SparseDistanceMatrix out = assemble_knn(dist_type, knn, idx.data(), d0.data(), d1.data(), idx.size());
            out.ref_names = sketch_names(sketches);
            return out;
        }
    } else
    // CANONICAL TIES: every pair once across the devices (the reference evaluates (i, j) and (j, i), mod.rs:148-171):
    // the row bands are dealt back and forth over the devices (band b costs ~ n - b*band_rows), each
    // device returns its partial top-k states for ALL rows, and every device then merges one row
    // shard of the stacked states.  Configurations without that form fall through to row shards.
    if (knn <= 2048) {
        const size_t W = devs.size();
        const bool coreacc = p.dist_type == SKL_DIST_COREACC;
        std::vector<std::vector<uint32_t>> key(W), sid(W);
        std::vector<std::vector<float>> sd1(W);
        std::vector<int> rc(W, SKL_OK);
        std::vector<std::string> msg(W);
        for_each_device(devs, [&](size_t d) {
            Slab s(devs[d], sketches, completeness_vec);
            const size_t band_rows = skl_knn_band_rows(s.h, &p, W);
            if (band_rows == 0) {
                rc[d] = SKL_ERR_INVALID_ARG;
                return;
            }
            const size_t n_bands = (n + band_rows - 1) / band_rows;
            std::vector<uint32_t> mine;
            for (size_t band = 0; band < n_bands; ++band) {
                const size_t lap = band / W, pos = band % W;
                if ((lap % 2 == 0 ? pos : W - 1 - pos) == d) mine.push_back((uint32_t)band);
            }
            key[d].resize(n * knn);
            sid[d].resize(n * knn);
            if (coreacc) sd1[d].resize(n * knn);
            rc[d] = skl_self_dists_knn_partial(devs[d].ctx(), s.h, &p, knn, band_rows, mine.data(), mine.size(),
                                               key[d].data(), sid[d].data(), coreacc ? sd1[d].data() : nullptr, 0);
            if (rc[d] != SKL_OK) msg[d] = skl_last_error();
        });
        bool once = true;
        for (size_t d = 0; d < W; ++d) {
            if (rc[d] == SKL_ERR_INVALID_ARG) once = false;
            else if (rc[d] == SKL_ERR_KMER_COUNT || rc[d] == SKL_ERR_EMPTY_DB) throw Panic(msg[d]);
            else if (rc[d] != SKL_OK) throw std::runtime_error(msg[d]);
        }
        if (once) {
            for_each_device(devs, [&](size_t d) {
                const size_t rows = b[d + 1] - b[d];
                if (rows == 0) return;
                std::vector<uint32_t> k_all(W * rows * knn), i_all(W * rows * knn);
                std::vector<float> d_all(coreacc ? W * rows * knn : 0);
                for (size_t w = 0; w < W; ++w) {
                    std::copy_n(key[w].data() + b[d] * knn, rows * knn, k_all.data() + w * rows * knn);
                    std::copy_n(sid[w].data() + b[d] * knn, rows * knn, i_all.data() + w * rows * knn);
                    if (coreacc) std::copy_n(sd1[w].data() + b[d] * knn, rows * knn, d_all.data() + w * rows * knn);
                }
                check(skl_knn_merge_states(devs[d].ctx(), W, rows, knn, k_all.data(), i_all.data(),
                                           coreacc ? d_all.data() : nullptr, 0, p.ani, idx.data() + b[d] * knn,
                                           d0.data() + b[d] * knn, d1.data() + b[d] * knn, 0));
            });
            SparseDistanceMatrix out = assemble_knn(dist_type, knn, idx.data(), d0.data(), d1.data(), idx.size());
            out.ref_names = sketch_names(sketches);
            return out;
        }
    }
    for_each_device(devs, [&](size_t d) {
        if (b[d + 1] <= b[d]) return;
        Slab s(devs[d], sketches, completeness_vec);
        check(skl_self_dists_knn_rows(devs[d].ctx(), s.h, &p, knn, b[d], b[d + 1], idx.data() + b[d] * knn,
                                      d0.data() + b[d] * knn, d1.data() + b[d] * knn, 0));
    });
    SparseDistanceMatrix out = assemble_knn(dist_type, knn, idx.data(), d0.data(), d1.data(), idx.size());
    out.ref_names = sketch_names(sketches);
    return out;
}

SparseDistanceMatrix cross_dists_knn(DeviceSet &devs, const MultiSketch &ref_sketches,
                                     const MultiSketch &query_sketches, size_t n, size_t n_query,
                                     size_t knn, const DistType &dist_type, bool quiet,
                                     const std::vector<double> *ref_completeness_vec,
                                     const std::vector<double> *query_completeness_vec,
                                     double completeness_cutoff)
{
    if (devs.size() == 1) {
        return cross_dists_knn(devs[0], ref_sketches, query_sketches, n, n_query, knn, dist_type, quiet,
                               ref_completeness_vec, query_completeness_vec, completeness_cutoff);
    }
    if (n == 0) throw Panic("Reference database has no loaded samples");
    if (n_query == 0) throw Panic("Query database has no loaded samples");
    knn = std::min(knn, n);
    std::vector<uint64_t> idx(n_query * knn);
    std::vector<float> d0(n_query * knn), d1(n_query * knn);
    const skl_dist_params p = to_params(dist_type, completeness_cutoff);
    const std::vector<size_t> b = even_bounds(n_query, devs.size());
    for_each_device(devs, [&](size_t d) {
        if (b[d + 1] <= b[d]) return;
        Slab r(devs[d], ref_sketches, ref_completeness_vec);
        Slab q(devs[d], query_sketches, query_completeness_vec);
        check(skl_cross_dists_knn_rows(devs[d].ctx(), r.h, q.h, &p, knn, b[d], b[d + 1], idx.data() + b[d] * knn,
                                       d0.data() + b[d] * knn, d1.data() + b[d] * knn, 0));
    });
    SparseDistanceMatrix out = assemble_knn(dist_type, knn, idx.data(), d0.data(), d1.data(), idx.size());
    out.ref_names = sketch_names(ref_sketches);
    out.query_names = sketch_names(query_sketches);
    return out;
}

// ---------------------------------------------------------------------------
// streaming dense drivers
// ---------------------------------------------------------------------------

namespace {
// Compute bands [b[i], b[i+1]) one after the other; while band i is formatted and written,
// band i+1 is already being computed (one helper thread drives the GPU call).
template <class Compute>
void stream_bands(const DistanceMatrix &shape, const std::vector<size_t> &b, size_t max_band_floats,
                  Compute compute, TextSink &sink, size_t threads, bool npy)
{
    const size_t ncols = shape.jaccard.n_dist_cols();
    const size_t n_rows = shape.ref_names.size();
    auto band_floats = [&](size_t r0, size_t r1) {
        if (shape.query_names) return (r1 - r0) * shape.query_names->size() * ncols;
        r1 = std::min(r1, n_rows ? n_rows - 1 : 0);
        if (r1 <= r0) return (size_t)0;
        auto upto = [&](size_t r) { return r * n_rows - r * (r + 1) / 2; };   // pairs with i < r
        return (upto(r1) - upto(r0)) * ncols;
    };
    if (npy) {
        const std::string h = npy_header(shape.n_distances, ncols);
        sink.finish(sink.begin(h.data(), h.size()), h.data(), h.size());
    }
    std::vector<float> buf[2];
    buf[0].resize(max_band_floats);
    buf[1].resize(max_band_floats);
    std::exception_ptr err;
    auto launch = [&](size_t i) {
        return std::thread([&, i] {
            try {
                compute(b[i], b[i + 1], buf[i & 1].data());
            } catch (...) {
                err = std::current_exception();
            }
        });
    };
    const size_t n_bands = b.size() - 1;
    if (n_bands == 0) return;
    std::thread worker = launch(0);
    for (size_t i = 0; i < n_bands; ++i) {
        const auto t0 = std::chrono::steady_clock::now();
        worker.join();
        output_timing().wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (err) std::rethrow_exception(err);
        if (i + 1 < n_bands) worker = launch(i + 1);
        if (npy) {
            const char *bytes = reinterpret_cast<const char *>(buf[i & 1].data());
            const size_t len = band_floats(b[i], b[i + 1]) * sizeof(float);
            const auto t1 = std::chrono::steady_clock::now();
            write_raw(sink, bytes, len);
            output_timing().sink_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
        } else {
            shape.write_rows(sink, b[i], b[i + 1], buf[i & 1].data(), threads);
        }
    }
}

}  // namespace

std::string npy_header(size_t rows, size_t cols)
{
    std::string dict = "{'descr': '<f4', 'fortran_order': False, 'shape': (" + std::to_string(rows) + ", " +
                       std::to_string(cols) + "), }";
    // magic(6) + version(2) + header length(2) + dict + padding + '\n' must be a multiple of 64
    size_t total = 10 + dict.size() + 1;
    const size_t pad = (64 - total % 64) % 64;
    dict.append(pad, ' ');
    dict.push_back('\n');
    std::string h("\x93NUMPY", 6);
    h.push_back('\x01');
    h.push_back('\x00');
    h.push_back((char)(dict.size() & 0xFF));
    h.push_back((char)((dict.size() >> 8) & 0xFF));
    return h + dict;
}

void self_dists_all_streamed(Device &dev, const MultiSketch &sketches, size_t n, const DistType &dist_type,
                             const std::vector<double> *completeness_vec, double completeness_cutoff,
                             TextSink &sink, size_t threads, size_t band_bytes, bool npy)
{
    if (n < 2) return;
    DistanceMatrix shape;   // names + type only; no n^2 storage
    shape.jaccard = dist_type;
    shape.ref_names = sketch_names(sketches);
    shape.n_distances = n * (n - 1) / 2;
    const size_t ncols = dist_type.n_dist_cols();
    const size_t band_pairs = std::max<size_t>(n, band_bytes / (ncols * sizeof(float)));
    std::vector<size_t> b = {0};
    size_t acc = 0, max_pairs = 0;
    for (size_t i = 0; i + 1 < n; ++i) {
        const size_t row = n - 1 - i;
        if (acc && acc + row > band_pairs) {
            b.push_back(i);
            max_pairs = std::max(max_pairs, acc);
            acc = 0;
        }
        acc += row;
    }
    b.push_back(n);
    max_pairs = std::max(max_pairs, acc);
    Slab s(dev, sketches, completeness_vec);
    const skl_dist_params p = to_params(dist_type, completeness_cutoff);
    stream_bands(shape, b, max_pairs * ncols,
                 [&](size_t r0, size_t r1, float *out) { check(skl_self_dists_rows(dev.ctx(), s.h, &p, r0, r1, out, 0)); },
                 sink, threads, npy);
}

void cross_dists_all_streamed(Device &dev, const MultiSketch &ref_sketches, const MultiSketch &query_sketches,
                              size_t n, size_t n_query, const DistType &dist_type,
                              const std::vector<double> *ref_completeness_vec,
                              const std::vector<double> *query_completeness_vec, double completeness_cutoff,
                              TextSink &sink, size_t threads, size_t band_bytes, bool npy)
{
    if (n == 0 || n_query == 0) return;
    DistanceMatrix shape;
    shape.jaccard = dist_type;
    shape.ref_names = sketch_names(ref_sketches);
    shape.query_names = sketch_names(query_sketches);
    shape.n_distances = n * n_query;
    const size_t ncols = dist_type.n_dist_cols();
    const size_t rows_per_band = std::max<size_t>(1, band_bytes / (n_query * ncols * sizeof(float)));
    std::vector<size_t> b;
    for (size_t r = 0; r < n; r += rows_per_band) b.push_back(r);
    b.push_back(n);
    Slab r(dev, ref_sketches, ref_completeness_vec);
    Slab q(dev, query_sketches, query_completeness_vec);
    const skl_dist_params p = to_params(dist_type, completeness_cutoff);
    stream_bands(shape, b, std::min(rows_per_band, n) * n_query * ncols,
                 [&](size_t r0, size_t r1, float *out) {
                     check(skl_cross_dists_rows(dev.ctx(), r.h, q.h, &p, r0, r1, out, 0));
                 },
                 sink, threads, npy);
}

}  // namespace distances
}  // namespace skl_host
