#include "distance_matrix.hpp"

#include <atomic>
#include <charconv>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <chrono>
#include <cmath>
#include <algorithm>
#include <ostream>
#include <stdexcept>
#include <fcntl.h>
#include <unistd.h>
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

// Longest text of one f32 in positional notation: "-0." + 44 zeros + 9 digits (denormal), or
// 39 digits (FLT_MAX); 64 covers both.
constexpr size_t F32_TEXT_MAX = 64;

// Rust's Display for f32 (what distance_matrix.rs:160-209 prints with `{}`): the shortest
// digit string that round-trips, laid out positionally, never in exponent form.  Writes at
// most F32_TEXT_MAX chars to dst and returns the end.
static char *format_f32_to(char *dst, float v)
{
    if (std::isnan(v)) {
        std::memcpy(dst, "NaN", 3);
        return dst + 3;
    }
    if (std::isinf(v)) {
        const size_t l = v > 0 ? 3 : 4;
        std::memcpy(dst, v > 0 ? "inf" : "-inf", l);
        return dst + l;
    }
    if (std::fabs(v) < 16777216.0f) {
        // below 2^24 the shortest fixed form has no digits beyond the shortest round-trip ones
        return std::to_chars(dst, dst + F32_TEXT_MAX, v, std::chars_format::fixed).ptr;
    }
    // Large values: to_chars(fixed) prints the exact integer; Rust prints the shortest digits
    // and pads with zeros.  Take digits + exponent from the scientific form.
    char sci[32];
    const char *end = std::to_chars(sci, sci + sizeof sci, v, std::chars_format::scientific).ptr;
    const char *q = sci;
    if (*q == '-') *dst++ = *q++;
    const char *e = q;
    while (e < end && *e != 'e') ++e;
    int exp10 = 0;
    std::from_chars(e + 1 + (e[1] == '+'), end, exp10);
    int written = 0;   // digits emitted so far (integer part needs exp10 + 1 of them)
    for (const char *c = q; c < e; ++c) {
        if (*c == '.') continue;
        *dst++ = *c;
        ++written;
    }
    for (; written < exp10 + 1; ++written) *dst++ = '0';
    return dst;
}

std::string format_f32(float v)
{
    char buf[F32_TEXT_MAX];
    return std::string(buf, format_f32_to(buf, v));
}

// A reusable text block: grows geometrically, keeps its capacity between blocks so a
// formatting thread stops page-faulting fresh memory after its first few blocks.
struct TextBlock {
    char *p = nullptr;
    size_t len = 0, cap = 0;
    TextBlock() = default;
    TextBlock(const TextBlock &) = delete;
    TextBlock &operator=(const TextBlock &) = delete;
    ~TextBlock() { std::free(p); }
    void need(size_t extra)
    {
        if (cap - len >= extra) return;
        size_t c = std::max<size_t>(cap * 2, 1 << 20);
        while (c - len < extra) c *= 2;
        char *q = static_cast<char *>(std::realloc(p, c));
        if (!q) throw std::bad_alloc();
        p = q;
        cap = c;
    }
    void put(const std::string &s)
    {
        std::memcpy(p + len, s.data(), s.size());
        len += s.size();
    }
    void put(char c) { p[len++] = c; }
};

static inline void put_f32(TextBlock &out, float v)
{
    out.len = (size_t)(format_f32_to(out.p + out.len, v) - out.p);
}

// Format one output row block: reference rows [r0, r1) of the dense matrix.
// `dist` points at element `dist_base` of the full (flat) distance array.
static void format_rows(const DistanceMatrix &m, size_t r0, size_t r1, const float *dist, size_t dist_base,
                        TextBlock &out)
{
    const bool coreacc = m.jaccard.kind == DistType::CoreAcc;
    const size_t ncols = coreacc ? 2 : 1;
    const size_t n = m.ref_names.size();
    out.len = 0;
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
        const std::string &row_name = m.ref_names[i];
        const float *d = dist + (dist_idx - dist_base);
        for (size_t j = j_begin; j < j_end; ++j) {
            const std::string &col_name = m.query_names ? (*m.query_names)[j] : m.ref_names[j];
            out.need(row_name.size() + col_name.size() + 2 * F32_TEXT_MAX + 4);
            out.put(row_name);
            out.put('\t');
            out.put(col_name);
            out.put('\t');
            put_f32(out, d[0]);
            if (coreacc) {
                out.put('\t');
                put_f32(out, d[1]);
            }
            out.put('\n');
            d += ncols;
        }
    }
}

OutputTiming &output_timing()
{
    static OutputTiming t;
    return t;
}

static double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ---- sinks ----

uint64_t StreamSink::begin(const char *p, size_t len)
{
    os_.write(p, (std::streamsize)len);   // called in block order: the stream is the order
    return 0;
}

FileSink::FileSink(const std::string &path)
{
    fd_ = ::open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0644);
    if (fd_ < 0) throw std::runtime_error("cannot create output file " + path);
}

FileSink::~FileSink()
{
    if (fd_ >= 0) ::close(fd_);
}

uint64_t FileSink::begin(const char *, size_t len)
{
    const uint64_t at = offset_;   // called in block order: reserve the byte range only
    offset_ += len;
    return at;
}

void FileSink::finish(uint64_t off, const char *p, size_t len)
{
    while (len) {   // any thread, any order
        const ssize_t w = ::pwrite(fd_, p, len, (off_t)off);
        if (w <= 0) throw std::runtime_error("write to output file failed");
        p += w;
        len -= (size_t)w;
        off += (uint64_t)w;
    }
}

// Blocks 0..n_blocks-1 formatted by `threads` workers (each into a private reusable buffer)
// and handed to the sink in block order: threads take blocks from a shared counter, format,
// then pass through an ordered section (stream: write there; file: reserve the byte range there
// and pwrite outside it).
template <class Format>
static void write_blocks_in_order(TextSink &sink, size_t n_blocks, size_t threads, Format format)
{
    if (n_blocks == 0) return;
    threads = std::max<size_t>(1, std::min(threads, n_blocks));
    const double t_begin = now_s();
    std::atomic<size_t> next{0};
    std::mutex mu;
    std::condition_variable cv;
    size_t turn = 0;          // block whose ordered section may run
    bool failed = false;
    std::exception_ptr err;
    std::vector<double> sink_s(threads, 0.0);
    auto work = [&](size_t tid) {
        TextBlock block;
        try {
            for (;;) {
                const size_t b = next.fetch_add(1);
                if (b >= n_blocks) break;
                block.len = 0;
                format(b, block);
                const double t0 = now_s();
                uint64_t token;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return turn == b || failed; });
                    if (failed) return;
                    token = sink.begin(block.p, block.len);
                    turn = b + 1;
                }
                cv.notify_all();
                sink.finish(token, block.p, block.len);
                sink_s[tid] += now_s() - t0;
            }
        } catch (...) {
            {
                std::lock_guard<std::mutex> lk(mu);
                if (!failed) err = std::current_exception();
                failed = true;
            }
            cv.notify_all();
        }
    };
    if (threads == 1) {
        work(0);
    } else {
        std::vector<std::thread> pool;
        for (size_t t = 0; t < threads; ++t) pool.emplace_back(work, t);
        for (auto &t : pool) t.join();
    }
    if (err) std::rethrow_exception(err);
    // wall time of this call, split by the share of thread time spent in/waiting for the sink
    const double wall = now_s() - t_begin;
    double sink_share = 0;
    for (double v : sink_s) sink_share += v;
    sink_share = std::min(1.0, sink_share / (wall * (double)threads + 1e-12));
    output_timing().sink_s += wall * sink_share;
    output_timing().format_s += wall * (1.0 - sink_share);
}

void DistanceMatrix::write(std::ostream &os, size_t threads) const
{
    StreamSink sink(os);
    write_rows(sink, 0, ref_names.size(), distances.data(), threads);
}

void DistanceMatrix::write_rows(TextSink &sink, size_t r0, size_t r1, const float *band, size_t threads) const
{
    const size_t n_rows = ref_names.size();
    r1 = std::min(r1, n_rows);
    if (r1 <= r0) return;
    const size_t ncols = jaccard.n_dist_cols();
    const size_t dist_base = (query_names ? r0 * query_names->size()
                                          : (r0 + 1 < n_rows ? square_to_condensed(r0, r0 + 1, n_rows) : 0)) * ncols;
    // Blocks of whole rows, ~32K lines each (about a megabyte of text: stays cache-resident
    // between formatting and the write).  Threads take blocks from a shared counter, format
    // into a private reusable buffer, then pass through an ordered section in block order
    // (stream: write there; file: reserve the byte range there and pwrite outside it).
    constexpr size_t BLOCK_LINES = 1 << 15;
    std::vector<size_t> bounds = {r0};
    size_t acc = 0;
    for (size_t i = r0; i < r1; ++i) {
        acc += query_names ? query_names->size() : n_rows - 1 - i;
        if (acc >= BLOCK_LINES || i + 1 == r1) {
            bounds.push_back(i + 1);
            acc = 0;
        }
    }
    write_blocks_in_order(sink, bounds.size() - 1, threads, [&](size_t b, TextBlock &block) {
        format_rows(*this, bounds[b], bounds[b + 1], band, dist_base, block);
    });
}

void SparseDistanceMatrix::write(std::ostream &os) const
{
    StreamSink sink(os);
    write(sink, 1);
}

void SparseDistanceMatrix::write(TextSink &sink, size_t threads) const
{
    // rows are labelled by query names in cross mode, ref names otherwise
    const std::vector<std::string> &rows = query_names ? *query_names : ref_names;
    const bool jac = jaccard.kind == DistType::Jaccard;
    const size_t n_items = jac ? jaccard_dists.size() : coreacc_dists.size();
    if (n_items == 0 || knn == 0) return;
    const size_t n_rows = n_items / knn;
    const size_t rows_per_block = std::max<size_t>(1, (1 << 15) / knn);   // ~32K lines per block
    const size_t n_blocks = (n_rows + rows_per_block - 1) / rows_per_block;
    write_blocks_in_order(sink, n_blocks, threads, [&](size_t b, TextBlock &out) {
        const size_t x1 = std::min(n_items, (b + 1) * rows_per_block * knn);
        for (size_t x = b * rows_per_block * knn; x < x1; ++x) {
            const std::string &row_name = rows[x / knn];
            const std::string &col_name = ref_names[jac ? jaccard_dists[x].idx : coreacc_dists[x].idx];
            // Padding entries (dist == 1.0, col == row) are skipped, distance_matrix.rs:379-381
            if (jac && !(jaccard_dists[x].dist < 1.0f || col_name != row_name)) continue;
            out.need(row_name.size() + col_name.size() + 2 * F32_TEXT_MAX + 4);
            out.put(row_name);
            out.put('\t');
            out.put(col_name);
            out.put('\t');
            if (jac) {
                put_f32(out, jaccard_dists[x].dist);
            } else {
                put_f32(out, coreacc_dists[x].core);
                out.put('\t');
                put_f32(out, coreacc_dists[x].acc);
            }
            out.put('\n');
        }
    });
}

}  // namespace skl_host
