#include "distance_matrix.hpp"

#include <charconv>
#include <cmath>
#include <algorithm>
#include <ostream>
#include <thread>
#include <vector>

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

// Append one output row block: reference rows [r0, r1) of the dense matrix.
static void format_rows(const DistanceMatrix &m, size_t r0, size_t r1, std::string &out)
{
    const bool coreacc = m.jaccard.kind == DistType::CoreAcc;
    const size_t ncols = coreacc ? 2 : 1;
    const size_t n = m.ref_names.size();
    char buf[128];
    auto put_f32 = [&](float v) {
        if (std::isnan(v)) { out += "NaN"; return; }
        if (std::isinf(v)) { out += v > 0 ? "inf" : "-inf"; return; }
        const auto res = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::fixed);
        out.append(buf, res.ptr);
    };
    for (size_t i = r0; i < r1; ++i) {
        size_t dist_idx;
        size_t j_begin, j_end;
        if (m.query_names) {
            j_begin = 0;
            j_end = m.query_names->size();
            dist_idx = i * j_end * ncols;
        } else {
            j_begin = i + 1;
            j_end = n;
            dist_idx = (i + 1 < n ? square_to_condensed(i, i + 1, n) : 0) * ncols;
        }
        for (size_t j = j_begin; j < j_end; ++j) {
            out += m.ref_names[i];
            out += '\t';
            out += m.query_names ? (*m.query_names)[j] : m.ref_names[j];
            out += '\t';
            put_f32(m.distances[dist_idx]);
            if (coreacc) {
                out += '\t';
                put_f32(m.distances[dist_idx + 1]);
            }
            out += '\n';
            dist_idx += ncols;
        }
    }
}

void DistanceMatrix::write(std::ostream &os, size_t threads) const
{
    const size_t n_rows = ref_names.size();
    if (n_rows == 0) return;
    threads = std::max<size_t>(1, std::min(threads, n_rows));
    // row blocks of roughly equal line count, formatted `threads` at a time and written in order
    const size_t lines_total = n_distances;
    const size_t target = std::max<size_t>(1 << 14, lines_total / (threads * 8) + 1);
    std::vector<size_t> bounds = {0};
    size_t acc = 0;
    for (size_t i = 0; i < n_rows; ++i) {
        acc += query_names ? query_names->size() : n_rows - 1 - i;
        if (acc >= target || i + 1 == n_rows) {
            bounds.push_back(i + 1);
            acc = 0;
        }
    }
    const size_t n_blocks = bounds.size() - 1;
    for (size_t b0 = 0; b0 < n_blocks; b0 += threads) {
        const size_t b1 = std::min(n_blocks, b0 + threads);
        std::vector<std::string> bufs(b1 - b0);
        if (b1 - b0 == 1) {
            format_rows(*this, bounds[b0], bounds[b0 + 1], bufs[0]);
        } else {
            std::vector<std::thread> pool;
            for (size_t b = b0; b < b1; ++b) {
                pool.emplace_back([&, b] { format_rows(*this, bounds[b], bounds[b + 1], bufs[b - b0]); });
            }
            for (auto &t : pool) t.join();
        }
        for (const auto &s : bufs) os.write(s.data(), (std::streamsize)s.size());
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
