// The CLI's text writer (csrc/host/distance_matrix.cpp: persistent worker pool, two-phase block groups) on its own, no GPU:
// the dense listing of a 1 500-sample matrix through a file sink and a stream sink, cut into row bands of many heights and
// formatted by 1 ... 186 workers, must be byte-identical to the single-threaded listing.  argv[1]: a scratch directory;
// argv[2]: the number of samples (default 1 500; the ThreadSanitizer build of the test takes 700).
#include "distance_matrix.hpp"
#include <iostream>
#include <sstream>
#include <fstream>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace skl_host;
static std::string slurp(const char *p) { std::ifstream f(p, std::ios::binary); std::stringstream ss; ss << f.rdbuf(); return ss.str(); }
int main(int argc, char **argv) {
    const std::string scratch = std::string(argc > 1 ? argv[1] : ".") + "/out3.txt";
    size_t n = argc > 2 ? (size_t)atol(argv[2]) : 1500;
    DistanceMatrix m;
    m.jaccard.kind = DistType::CoreAcc;
    for (size_t i = 0; i < n; ++i) m.ref_names.push_back("s" + std::to_string(i));
    m.n_distances = n * (n - 1) / 2;
    m.distances.resize(m.n_distances * 2);
    for (size_t i = 0; i < m.distances.size(); ++i) m.distances[i] = (float)((i * 2654435761u) % 1000003) / 1000003.0f;
    std::string ref;
    testing_set_host_cpu_budget(1000);      // (the cap at twice the CPU budget would hide the large worker counts below)
    fprintf(stderr, "host_cpu_budget forced to %zu\n", host_cpu_budget());
    for (int rep = 0; rep < 12; ++rep) {
        size_t threads = rep == 0 ? 1 : 1 + (rep * 37) % 200;
        size_t rows_per = rep == 0 ? n : 1 + (rep * 131) % 700;
        std::string got;
        if (rep & 1) {
            std::ostringstream os;
            StreamSink ss(os);
            for (size_t r0 = 0; r0 < n; r0 += rows_per) {
                size_t r1 = std::min(n, r0 + rows_per);
                size_t base = (r0 + 1 < n ? (n * r0 - r0 * (r0 + 1) / 2) : 0) * 2;
                m.write_rows(ss, r0, r1, m.distances.data() + base, threads);
            }
            got = os.str();
        } else {
            {
                FileSink fs(scratch);
                for (size_t r0 = 0; r0 < n; r0 += rows_per) {
                    size_t r1 = std::min(n, r0 + rows_per);
                    size_t base = (r0 + 1 < n ? (n * r0 - r0 * (r0 + 1) / 2) : 0) * 2;
                    m.write_rows(fs, r0, r1, m.distances.data() + base, threads);
                }
            }
            got = slurp(scratch.c_str());
        }
        if (rep == 0) ref = got;
        fprintf(stderr, "rep %d threads %zu rows_per %zu bytes %zu %s\n", rep, threads, rows_per, got.size(), got == ref ? "same" : "DIFFERENT");
        if (got != ref) return 1;
    }
    testing_set_host_cpu_budget(0);
    if (host_cpu_budget() < 1 || host_cpu_budget() > 4096) return 2;   // the measured budget: hardware threads / affinity / cgroup quota
    fprintf(stderr, "host_cpu_budget measured: %zu\n", host_cpu_budget());
    // raw bytes (a .npy band) behind a header
    {
        std::vector<char> raw((20u << 20) + 3);
        for (size_t i = 0; i < raw.size(); ++i) raw[i] = (char)((i * 131u + (i >> 13)) & 0xFF);
        {
            FileSink fs(scratch);
            const std::string header(128, 'h');
            fs.finish(fs.begin(header.data(), header.size()), header.data(), header.size());
            write_raw(fs, raw.data(), raw.size());
            write_raw(fs, raw.data(), 1000);
        }
        const std::string got = slurp(scratch.c_str());
        const bool ok = got.size() == 128 + raw.size() + 1000 && got.compare(0, 128, std::string(128, 'h')) == 0 &&
                        std::memcmp(got.data() + 128, raw.data(), raw.size()) == 0 &&
                        std::memcmp(got.data() + 128 + raw.size(), raw.data(), 1000) == 0;
        fprintf(stderr, "raw %s\n", ok ? "same" : "DIFFERENT");
        if (!ok) return 1;
    }
    return 0;
}
