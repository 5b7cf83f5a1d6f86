#include "distance_matrix.hpp"

#include <charconv>
#include <cmath>
#include <ostream>

namespace skl_host {

size_t calc_row_idx(size_t k, size_t n)
{
    const int64_t k_i = (int64_t)k, n_i = (int64_t)n;
    return n - 2 -
           (size_t)std::floor(std::sqrt((double)(-8 * k_i + 4 * n_i * (n_i - 1) - 7)) / 2.0 - 0.5);
}

size_t calc_col_idx(size_t k, size_t i, size_t n)
{
    const int64_t k_i = (int64_t)k, i_i = (int64_t)i, n_i = (int64_t)n;
    return (size_t)(k_i + i_i + 1 - n_i * (n_i - 1) / 2 + (n_i - i_i) * ((n_i - i_i) - 1) / 2);
}

std::string DistType::describe() const
{
    if (kind == CoreAcc) return "Distances: core/accessory regression";
    const size_t ki = (size_t)k;
    return ani ? "Distances: ANI at k=" + std::to_string(ki)
               : "Distances: Jaccard distances at k=" + std::to_string(ki);
}

std::string format_f32(float v)
{
    if (std::isnan(v)) return "NaN";
    if (std::isinf(v)) return v > 0 ? "inf" : "-inf";
    char buf[128];
    const auto res = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::fixed);
    return std::string(buf, res.ptr);
}

void DistanceMatrix::write(std::ostream &os) const
{
    size_t dist_idx = 0;
    const bool coreacc = jaccard.kind == DistType::CoreAcc;
    if (query_names) {
        for (const auto &ref_name : ref_names) {
            for (const auto &query_name : *query_names) {
                os << ref_name << '\t' << query_name << '\t' << format_f32(distances[dist_idx]);
                if (coreacc) {
                    os << '\t' << format_f32(distances[dist_idx + 1]);
                    dist_idx += 1;
                }
                os << '\n';
                dist_idx += 1;
            }
        }
    } else {
        for (size_t i = 0; i < ref_names.size(); ++i) {
            for (size_t j = i + 1; j < ref_names.size(); ++j) {
                os << ref_names[i] << '\t' << ref_names[j] << '\t' << format_f32(distances[dist_idx]);
                if (coreacc) {
                    os << '\t' << format_f32(distances[dist_idx + 1]);
                    dist_idx += 1;
                }
                os << '\n';
                dist_idx += 1;
            }
        }
    }
}

void SparseDistanceMatrix::write(std::ostream &os) const
{
    // rows are labelled by query names in cross mode, ref names otherwise
    const std::vector<std::string> &rows = query_names ? *query_names : ref_names;
    if (jaccard.kind == DistType::Jaccard) {
        for (size_t x = 0; x < jaccard_dists.size(); ++x) {
            const std::string &row_name = rows[x / knn];
            const std::string &col_name = ref_names[jaccard_dists[x].idx];
            // Padding entries (dist == 1.0, col == row) are skipped, distance_matrix.rs:379-381
            if (jaccard_dists[x].dist < 1.0f || col_name != row_name) {
                os << row_name << '\t' << col_name << '\t' << format_f32(jaccard_dists[x].dist) << '\n';
            }
        }
    } else {
        for (size_t x = 0; x < coreacc_dists.size(); ++x) {
            os << rows[x / knn] << '\t' << ref_names[coreacc_dists[x].idx] << '\t'
               << format_f32(coreacc_dists[x].core) << '\t' << format_f32(coreacc_dists[x].acc)
               << '\n';
        }
    }
}

}  // namespace skl_host
