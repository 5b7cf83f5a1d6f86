// sketch_gpu.cpp -- `sketchlib sketch --gpu`: FASTA parsing on host threads, hashing and bin
// minima on the device (skl_sketch_signs), densify / transpose / file writers on the host.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <thread>

#include "distances.hpp"
#include "sketch.hpp"
#include "sketchlib_dist.h"

namespace skl_host {

namespace {
template <class F>
void parallel_for(size_t n, size_t threads, F f)
{
    std::atomic<size_t> next{0};
    auto work = [&] {
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= n) break;
            f(i);
        }
    };
    std::vector<std::thread> pool;
    for (size_t t = 1; t < std::max<size_t>(1, std::min(threads, n)); ++t) pool.emplace_back(work);
    work();
    for (auto &t : pool) t.join();
}
}  // namespace

MultiSketch sketch_files_gpu(Device &dev, const std::string &output_prefix, const std::vector<InputFastx> &inputs,
                             const std::vector<size_t> &kmers, uint64_t sketch_size, bool rc, size_t threads)
{
    const size_t n = inputs.size(), nk = kmers.size();
    const uint64_t ss64 = (sketch_size + 63) / 64;   // num_bins, sketch/mod.rs:49-54
    const uint64_t num_bins = ss64 * 64;
    const size_t sample_words = (size_t)(ss64 * BBITS * nk);

    const bool timing = std::getenv("SKL_CLI_TIMING") != nullptr;
    const auto t_start = std::chrono::steady_clock::now();
    auto since = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count(); };
    double t_parse = 0, t_pack = 0, t_gpu = 0, t_finish = 0;

    // 1. parse (threads)
    std::vector<Sequence> seqs(n);
    {
        std::atomic<size_t> next{0};
        std::string error;
        std::mutex mu;
        auto work = [&] {
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= n) break;
                try {
                    for (const auto &file : inputs[i].second) add_fasta(file, seqs[i]);
                    uint64_t total = 0;
                    for (uint64_t c : seqs[i].acgt) total += c;
                    if (total == 0) throw std::runtime_error(inputs[i].first + " has no valid sequence");
                } catch (const std::exception &e) {
                    std::lock_guard<std::mutex> lk(mu);
                    if (error.empty()) error = e.what();
                }
            }
        };
        std::vector<std::thread> pool;
        for (size_t t = 1; t < std::max<size_t>(1, std::min(threads, n)); ++t) pool.emplace_back(work);
        work();
        for (auto &t : pool) t.join();
        if (!error.empty()) throw std::runtime_error(error);
    }

    t_parse = since();
    // 2. hash + bin minima on the device, in batches of at most ~4 G bases
    std::vector<uint64_t> bins(sample_words * n, 0);
    std::vector<SketchMeta> meta(n);
    constexpr uint64_t BATCH_CODES = 4ull << 30;
    for (size_t b0 = 0; b0 < n;) {
        size_t b1 = b0;
        uint64_t codes_in_batch = 0;
        while (b1 < n && (b1 == b0 || codes_in_batch + seqs[b1].codes.size() <= BATCH_CODES)) {
            codes_in_batch += seqs[b1].codes.size();
            ++b1;
        }
        const size_t nb = b1 - b0;
        // the bases at 2 bits each, 16 per word, every sample on a word boundary (skl_sketch_signs_packed): a quarter of the
        // bytes to gather here and to send over PCIe
        std::vector<uint64_t> code_begin(nb + 1, 0), word_begin(nb + 1, 0), offset_begin(nb + 1, 0), offsets;
        for (size_t i = 0; i < nb; ++i) {
            const Sequence &s = seqs[b0 + i];
            code_begin[i + 1] = code_begin[i] + s.codes.size();
            word_begin[i + 1] = word_begin[i] + (s.codes.size() + 15) / 16;
            offsets.insert(offsets.end(), s.offsets.begin(), s.offsets.end());
            offset_begin[i + 1] = offsets.size();
        }
        std::unique_ptr<uint32_t[]> packed(new uint32_t[std::max<uint64_t>(word_begin[nb], 1)]);   // not zero-filled
        parallel_for(nb, threads, [&](size_t i) {
            const std::vector<uint8_t> &c = seqs[b0 + i].codes;
            uint32_t *out = packed.get() + word_begin[i];
            const size_t whole = c.size() / 16;
            for (size_t w = 0; w < whole; ++w) {
                uint64_t lo, hi;
                memcpy(&lo, c.data() + 16 * w, 8);
                memcpy(&hi, c.data() + 16 * w + 8, 8);
                auto squeeze = [](uint64_t v) -> uint32_t {   // 8 bytes of 2 significant bits -> 16 bits
                    v &= 0x0303030303030303ull;
                    v = (v | (v >> 6)) & 0x000F000F000F000Full;
                    v = (v | (v >> 12)) & 0x000000FF000000FFull;
                    v = (v | (v >> 24)) & 0xFFFFull;
                    return (uint32_t)v;
                };
                out[w] = squeeze(lo) | (squeeze(hi) << 16);
            }
            if (c.size() % 16) {
                uint32_t word = 0;
                for (size_t x = 16 * whole; x < c.size(); ++x) word |= (uint32_t)(c[x] & 3u) << (2u * (uint32_t)(x - 16 * whole));
                out[whole] = word;
            }
        });
        std::vector<uint64_t> signs(nb * nk * num_bins);
        const double t0 = since();
        t_pack += t0 - (t_parse + t_pack + t_gpu + t_finish);
        const int rc_ = skl_sketch_signs_packed(dev.ctx(), packed.get(), code_begin.data(), offsets.data(), offset_begin.data(),
                                                nb, kmers.data(), nk, num_bins, rc ? 1 : 0, signs.data());
        if (rc_ != SKL_OK) throw std::runtime_error(skl_last_error());
        t_gpu += since() - t0;
        // 3. densify + transpose (host threads)
        std::string finish_error;
        std::mutex finish_mu;
        parallel_for(nb, threads, [&](size_t i) {
          try {
            const Sequence &s = seqs[b0 + i];
            bool densified = false;
            for (size_t ki = 0; ki < nk; ++ki) {
                const uint64_t *src = signs.data() + (i * nk + ki) * num_bins;
                std::vector<uint64_t> sg(src, src + num_bins);
                if (std::all_of(sg.begin(), sg.end(), [](uint64_t v) { return v == UINT64_MAX; })) {
                    throw std::runtime_error("K-mer larger than smallest valid sequence");   // as the CPU path
                }
                densified |= densify_bin(sg);
                fill_usigs(bins.data() + (b0 + i) * sample_words + ki * ss64 * BBITS, sg);
            }
            SketchMeta &m = meta[b0 + i];
            m.name = inputs[b0 + i].first;
            m.rc = rc;
            m.reads = false;
            uint64_t total = 0;
            for (uint64_t c : s.acgt) total += c;
            m.seq_length = total;
            m.densified = densified;
            for (int x = 0; x < 4; ++x) m.acgt[x] = s.acgt[x];
            m.non_acgt = s.non_acgt;
            m.index = b0 + i;
          } catch (const std::exception &e) {
            std::lock_guard<std::mutex> lk(finish_mu);
            if (finish_error.empty()) finish_error = e.what();
          }
        });
        if (!finish_error.empty()) throw std::runtime_error(finish_error);
        for (size_t i = b0; i < b1; ++i) Sequence().codes.swap(seqs[i].codes);   // release
        b0 = b1;
        t_finish = since() - t_parse - t_pack - t_gpu;
    }
    const double t_before_write = since();
    MultiSketch::write_sketch_data(output_prefix, bins.data(), bins.size());
    MultiSketch m(std::move(meta), ss64 * 64, kmers);
    m.save_metadata(output_prefix);
    m.set_bins(std::move(bins));
    if (timing) {
        std::fprintf(stderr, "TIMING sketch --gpu: parse=%.3fs pack=%.3fs upload+kernel+download=%.3fs densify+transpose=%.3fs write=%.3fs\n",
                     t_parse, t_pack, t_gpu, t_finish, since() - t_before_write);
    }
    return m;
}

}  // namespace skl_host
