// How does the CLI's text writer (csrc/host/distance_matrix.cpp) scale with --threads on this host?  Formats the dense
// listing of n samples (all distances 1, as Set U gives) into /dev/null with 8 ... 256 workers.  No GPU.
//   g++ -O2 -std=c++17 -Isketchlib.rust_amd/csrc/host -Iinclude scripts/native/text_writer_scaling.cpp sketchlib.rust_amd/csrc/host/distance_matrix.cpp -lpthread -o /tmp/tws && /tmp/tws 20000
#include "distance_matrix.hpp"
#include <chrono>
#include <cstdio>
#include <cstdlib>
using namespace skl_host;
int main(int argc, char **argv)
{
    const size_t n = argc > 1 ? (size_t)atoll(argv[1]) : 20000;
    DistanceMatrix m;
    m.jaccard.kind = DistType::CoreAcc;
    for (size_t i = 0; i < n; ++i) m.ref_names.push_back("s" + std::to_string(i));
    m.n_distances = n * (n - 1) / 2;
    m.distances.assign(m.n_distances * 2, 1.0f);
    for (size_t threads : {8, 16, 32, 64, 128, 256, 64, 8}) {
        FileSink fs("/dev/null");
        const auto t0 = std::chrono::steady_clock::now();
        m.write_rows(fs, 0, n, m.distances.data(), threads);
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("threads %3zu: %.3f s, %.3g lines/s, %.3g per thread (format %.3f s, sink %.3f s so far)\n", threads, s, m.n_distances / s,
               m.n_distances / s / threads, output_timing().format_s, output_timing().sink_s);
    }
    return 0;
}
