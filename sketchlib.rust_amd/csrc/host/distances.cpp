#include "distances.hpp"

#include <string>

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

namespace {
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
    SparseDistanceMatrix out;
    out.jaccard = dist_type;
    out.knn = knn;
    out.n_distances = rows * knn;
    std::vector<uint64_t> idx(out.n_distances);
    std::vector<float> d0(out.n_distances), d1(out.n_distances);
    const skl_dist_params p = to_params(dist_type, cutoff);
    if (query) {
        check(skl_cross_dists_knn(dev.ctx(), ref, query, &p, knn, idx.data(), d0.data(), d1.data(), 0));
    } else {
        check(skl_self_dists_knn(dev.ctx(), ref, &p, knn, idx.data(), d0.data(), d1.data(), 0));
    }
    if (dist_type.kind == DistType::CoreAcc) {
        out.coreacc_dists.resize(out.n_distances);
        for (size_t i = 0; i < out.n_distances; ++i) out.coreacc_dists[i] = {(size_t)idx[i], d0[i], d1[i]};
    } else {
        out.jaccard_dists.resize(out.n_distances);
        for (size_t i = 0; i < out.n_distances; ++i) out.jaccard_dists[i] = {(size_t)idx[i], d0[i]};
    }
    return out;
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

}  // namespace distances
}  // namespace skl_host
