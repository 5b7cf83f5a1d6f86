#include "distance_matrix.hpp"
#include <fstream>
#include <sched.h>

#include <atomic>
#include <charconv>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
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
    for (size_t i = r0; i < r1; ++i) {   // (appends to `out`)
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

// Output workers that live for the whole process: a streamed listing hands them one band after the other (160 bands of
// 256 MB at 100 000 genomes), and a worker's text buffer -- a megabyte and more, grown by need() -- is faulted in ONCE
// instead of once per band (round 3 started `threads` fresh threads with fresh buffers for every band: at 256 threads
// the page faults and the condition-variable wake-ups of the ordered section were 55 of the 68 s the text listing of
// BASELINE configs[2] took, profiles/r04_e2e_cfg3.txt).
namespace {
class OutputPool {
  public:
    static OutputPool &instance()
    {
        static OutputPool pool;
        return pool;
    }
    // fn(tid) on workers 0 .. threads-1 (the caller runs as worker 0); returns when all are done.  One job at a time: a
    // second caller (two matrices written at once from two threads of a binding) waits for the first one's job to finish.
    void run(size_t threads, const std::function<void(size_t)> &fn)
    {
        threads = std::max<size_t>(1, threads);
        if (threads == 1) {
            fn(0);
            return;
        }
        std::lock_guard<std::mutex> one_job(run_mu_);
        {
            std::lock_guard<std::mutex> lk(mu_);
            while (workers_.size() + 1 < threads) {
                const size_t tid = workers_.size() + 1;
                workers_.emplace_back([this, tid] { loop(tid); });
            }
            job_ = &fn;
            job_threads_ = threads;
            pending_ = threads - 1;
            ++generation_;
        }
        cv_.notify_all();
        fn(0);
        std::unique_lock<std::mutex> lk(mu_);
        done_cv_.wait(lk, [&] { return pending_ == 0; });
        job_ = nullptr;
    }
    TextBlock &block(size_t tid)
    {
        std::lock_guard<std::mutex> lk(mu_);
        if (blocks_.size() <= tid) blocks_.resize(tid + 1);
        if (!blocks_[tid]) blocks_[tid] = std::make_unique<TextBlock>();
        return *blocks_[tid];
    }
    ~OutputPool()
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : workers_) t.join();
    }

  private:
    void loop(size_t tid)
    {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(size_t)> *job = nullptr;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || generation_ != seen; });
                if (stop_) return;
                seen = generation_;
                if (tid < job_threads_) job = job_;
            }
            if (!job) continue;
            (*job)(tid);
            bool last;
            {
                std::lock_guard<std::mutex> lk(mu_);
                last = --pending_ == 0;
            }
            if (last) done_cv_.notify_one();
        }
    }
    std::mutex mu_;
    std::mutex run_mu_;   // held for the whole of run(): the job slot (job_, pending_, generation_) serves one caller
    std::condition_variable cv_, done_cv_;
    std::vector<std::thread> workers_;
    std::vector<std::unique_ptr<TextBlock>> blocks_;
    const std::function<void(size_t)> *job_ = nullptr;
    size_t job_threads_ = 0, pending_ = 0;
    uint64_t generation_ = 0;
    bool stop_ = false;
};
}  // namespace

// Blocks 0..n_blocks-1 of text to the sink, in block order, formatted by `threads` workers.  `format(b, out)` APPENDS
// block b to `out`.  Groups of up to 8 blocks per worker go through two phases with nothing ordered inside either:
//   1. the workers take blocks from a shared counter and format them one behind the other into their own buffers;
//   2. the lengths give every block its place: a positional sink (file) reserves the group's byte range once and the
//      workers pwrite their own blocks there; a stream gets the blocks in order from the calling thread.
// (Rounds 2-3 passed every block through an ordered section, condition variable or ticket: with 256 workers that
// section -- 150 000 blocks at 100 000 genomes -- was most of the listing's wall time, profiles/r04_e2e_cfg3.txt.)
// The CPUs this process may really use: the hardware threads, its affinity mask, and the cgroup's quota (a container that
// shows 256 hardware threads may be granted 16 CPUs' worth of time: formatting workers beyond twice that only add
// throttling -- 31 s instead of 16 s for the listing of BASELINE configs[2] at --threads 256, profiles/r04_e2e_cfg3.txt).
static std::atomic<size_t> g_cpu_budget_override{0};
void testing_set_host_cpu_budget(size_t n) { g_cpu_budget_override = n; }

size_t host_cpu_budget()
{
    if (const size_t forced = g_cpu_budget_override.load()) return forced;
    static const size_t budget = [] {
        size_t n = std::max(1u, std::thread::hardware_concurrency());
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min<size_t>(n, (size_t)std::max(1, CPU_COUNT(&set)));
        auto quota_from = [&](const char *quota_path, const char *period_path) {
            std::ifstream q(quota_path);
            std::string a, b;
            if (!(q >> a)) return;
            if (period_path) {   // cgroup v1: two files
                std::ifstream pf(period_path);
                if (!(pf >> b)) return;
            } else if (!(q >> b)) {   // cgroup v2: "<quota|max> <period>"
                return;
            }
            if (a == "max" || a == "-1") return;
            const double quota = std::strtod(a.c_str(), nullptr), period = std::strtod(b.c_str(), nullptr);
            if (quota > 0 && period > 0) n = std::min<size_t>(n, (size_t)std::max(1.0, std::ceil(quota / period)));
        };
        quota_from("/sys/fs/cgroup/cpu.max", nullptr);
        quota_from("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us");
        return n;
    }();
    return budget;
}

template <class Format>
static void write_blocks_in_order(TextSink &sink, size_t n_blocks, size_t threads, Format format)
{
    if (n_blocks == 0) return;
    threads = std::max<size_t>(1, std::min({threads, n_blocks, 2 * host_cpu_budget()}));
    OutputPool &pool = OutputPool::instance();
    struct Piece {
        size_t worker, off, len;
    };
    constexpr size_t BLOCKS_PER_WORKER = 8;
    const size_t group = threads * BLOCKS_PER_WORKER;
    std::vector<Piece> pieces(std::min(group, n_blocks));
    std::vector<uint64_t> place(pieces.size());
    std::mutex err_mu;
    std::exception_ptr err;
    auto guarded = [&](auto &&body) {
        try {
            body();
        } catch (...) {
            std::lock_guard<std::mutex> lk(err_mu);
            if (!err) err = std::current_exception();
        }
    };
    for (size_t g0 = 0; g0 < n_blocks; g0 += group) {
        const size_t g1 = std::min(n_blocks, g0 + group);
        const double t0 = now_s();
        std::atomic<size_t> next{g0};
        pool.run(threads, [&](size_t tid) {
            guarded([&] {
                TextBlock &buf = pool.block(tid);
                buf.len = 0;
                for (;;) {
                    const size_t b = next.fetch_add(1);
                    if (b >= g1) break;
                    const size_t start = buf.len;
                    format(b, buf);
                    pieces[b - g0] = Piece{tid, start, buf.len - start};
                }
            });
        });
        if (err) std::rethrow_exception(err);
        const double t1 = now_s();
        output_timing().format_s += t1 - t0;
        if (sink.positional()) {
            uint64_t total = 0;
            for (size_t x = 0; x < g1 - g0; ++x) {
                place[x] = total;
                total += pieces[x].len;
            }
            const uint64_t token = sink.begin(nullptr, total);
            pool.run(threads, [&](size_t tid) {
                guarded([&] {
                    const TextBlock &buf = pool.block(tid);
                    for (size_t x = 0; x < g1 - g0; ++x) {
                        if (pieces[x].worker == tid && pieces[x].len) sink.finish(token + place[x], buf.p + pieces[x].off, pieces[x].len);
                    }
                });
            });
            if (err) std::rethrow_exception(err);
        } else {
            for (size_t x = 0; x < g1 - g0; ++x) {
                if (pieces[x].len) sink.begin(pool.block(pieces[x].worker).p + pieces[x].off, pieces[x].len);
            }
        }
        output_timing().sink_s += now_s() - t1;
    }
}

// Raw bytes (a band of a .npy array) to the sink: ONE write.  Measured at BASELINE configs[2] (40 GB into a tmpfs file,
// 2 x EPYC 9575F, profiles/r04_e2e_cfg3.txt): one pwrite per 256 MB band copies into the page cache at 7.1 GB/s; 8 MB
// pieces pwritten by the worker pool at their offsets 6.0 GB/s (a file's writes queue behind its inode lock however
// many threads issue them); stores through a shared mapping from 64 / 256 workers 1.4 / 0.7 GB/s (page faults under
// the mapping's lock).  The copy into the page cache, not the GPU or PCIe, is what `--npy` waits for.
void write_raw(TextSink &sink, const char *bytes, size_t len)
{
    if (len) sink.finish(sink.begin(bytes, len), bytes, len);
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
