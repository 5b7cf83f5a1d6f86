// Per-wave timeline of pair_kernel_kslice on the BASELINE configs[1] shape (n genomes, 5
// k-mer lengths, sketchsize64 = 64): when does every wave start / get its first rows /
// finish streaming / finish, and on which CU.  Answers "where does the time of a 0.17 ms
// launch go": dispatch ramp, rounds, tail.  (Measurement tool, not product code.)
// Build: see scripts/microbench/build.sh
#define SKL_TRACE 1
#define SKL_AB 1   // every tile shape and the timing-only ablations (SKL_KSLICE_ABLATE)
// distinct symbol names: the product library exports the untraced kernel under the original ones
#define pair_kernel_kslice pair_kernel_kslice_traced
#define launch_pair_kernel_kslice launch_pair_kernel_kslice_traced
#define kslice_supported kslice_supported_traced
#include "../../sketchlib.rust_amd/csrc/pair_kslice.hip"
// (the persistent form this tool also traced in round 2 is archived: experiments/dropped_kernels/pair_kpersist.hip)

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <random>
#include <vector>

using namespace skl;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

int main(int argc, char **argv)
{
    const uint32_t n = argc > 1 ? atoi(argv[1]) : 1000;
    int shape = argc > 2 ? atoi(argv[2]) : 165;
    const uint32_t nk = 5, ss64 = 64;
    constexpr size_t TW = 8;   // words per trace record (SKL_TRACE_WORDS)
    const size_t sample_words = (size_t)nk * ss64 * BBITS;
    std::vector<uint64_t> h((size_t)(n + A_PAD_ROWS) * sample_words, 0);
    std::mt19937_64 rng(1);
    // argv[3]: "zero" = all-zero sketches (same instruction stream, no bit toggling in the datapath)
    const bool zero_data = argc > 3 && !strcmp(argv[3], "zero");
    if (!zero_data) for (size_t i = 0; i < (size_t)n * sample_words; ++i) h[i] = rng();
    uint64_t *dA;
    uint4 *dB;
    CK(hipMalloc(&dA, h.size() * 8));
    CK(hipMemcpy(dA, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    const size_t n_jb = (n + 63) / 64;
    CK(hipMalloc(&dB, n_jb * nk * ss64 * 7 * 64 * sizeof(uint4)));
    fprintf(stderr, "relayout\n");
    CK(launch_relayout(dA, dB, n, nk, ss64, 0));
    CK(hipDeviceSynchronize());
    fprintf(stderr, "relayout done\n");
    const uint64_t pairs = (uint64_t)n * (n - 1) / 2;
    uint32_t *dOut;
    CK(hipMalloc(&dOut, pairs * nk * 4 * std::max<size_t>(2, argc > 5 ? (size_t)atoi(argv[5]) : 1)));
    PairArgs g;
    memset(&g, 0, sizeof g);
    g.A = dA; g.B = dB; g.nA = n; g.nB = n; g.nk = nk; g.ss64 = ss64;
    g.k_begin = 0; g.k_count = nk; g.row_begin = 0; g.row_end = n - 1; g.self_mode = 1;
    g.out_base = 0; g.out = dOut; g.cnt_pair_stride = 1; g.cnt_k_stride = pairs;
    g.k_slices = argc > 5 ? (uint32_t)atoi(argv[5]) : 1u;
    g.xcd_shift = 3;   // an unpartitioned MI355X: 8 XCDs
    uint64_t *dTrace;
    const size_t trace_words = (size_t)TW << 20;   // up to 1 M waves
    CK(hipMalloc(&dTrace, trace_words * 8));
    CK(hipMemset(dTrace, 0, trace_words * 8));
    g.dtab = (const float *)dTrace;
    TileScratch ts;
    fprintf(stderr, "launch\n");
    // argv[4]: untimed launches before the traced one (the clock settles under sustained load)
    const int warm = argc > 4 ? atoi(argv[4]) : 3;
    const int ablate = getenv("SKL_KSLICE_ABLATE") ? atoi(getenv("SKL_KSLICE_ABLATE")) : 0;
    // argv[5]: chunk slices per k-mer length (k_slices of the k-sliced COUNTS launch)
    const uint32_t slices = argc > 5 ? (uint32_t)atoi(argv[5]) : 1u;
    const bool persistent = false;
    if (getenv("KT_ROUND")) g.round_size = (uint32_t)atoi(getenv("KT_ROUND"));   // wave priority by round of workgroups (PairArgs::round_size)
    // KT_TAIL=S: tail slicing of the one-workgroup-per-unit launch (pair_kslice.hip, PairArgs::tail_slices)
    if (getenv("KT_TAIL") && !persistent) {
        g.tail_slices = (uint32_t)atoi(getenv("KT_TAIL"));
        g.tail_resident = (shape == 325 ? 3u : 4u) * 256u / 8u;
        CK(hipMemset(dOut, 0, pairs * nk * 4 * 2));
    }
    for (int i = 0; i < warm; ++i) CK(launch_pair_kernel_kslice(g, MODE_COUNTS, shape, true, ablate, ts, 0));
    CK(hipDeviceSynchronize());
    fprintf(stderr, "warm done\n");
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    CK(launch_pair_kernel_kslice(g, MODE_COUNTS, shape, true, ablate, ts, 0));
    hipEventRecord(e1);
    CK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const int R = shape > 1000 ? shape / 100 : shape / 10, JL = 2;
    PairArgs gp = g; uint64_t n_wg = 0;
    CK(plan_tiles(gp, R, JL * 64, ts, 0, &n_wg));
    n_wg = 8ull * gp.tiles_per_xcd * nk * slices;   // as launch_pair_kernel_kslice sizes it
    if (shape == 999) n_wg = 1024;
    if (g.tail_slices > 1) n_wg *= g.tail_slices;   // upper bound: workgroup indices beyond the launch have empty records
    const int wpw = getenv("SKL_KSLICE_WAVES") ? atoi(getenv("SKL_KSLICE_WAVES")) : 4;
    const size_t n_waves = n_wg * wpw;
    fprintf(stderr, "n_wg %llu\n", (unsigned long long)n_wg);
    if (n_waves * TW > trace_words) { printf("too many waves for the trace buffer\n"); return 1; }
    std::vector<uint64_t> tr(n_waves * TW);
    CK(hipMemcpy(tr.data(), dTrace, tr.size() * 8, hipMemcpyDeviceToHost));
    // 100 MHz ticks -> us
    uint64_t t_min = ~0ull, t_max = 0;
    size_t active = 0;
    for (size_t w = 0; w < n_waves; ++w) {
        const uint64_t *r = &tr[w * TW];
        if (r[3] == 0 || r[0] == 0) continue;   // wave exited early (no tile)
        ++active;
        t_min = std::min(t_min, r[0]);
        t_max = std::max(t_max, r[3]);
    }
    printf("n=%u shape=%d workgroups=%llu waves=%zu (with work: %zu) event time %.1f us, trace span %.1f us\n", n, shape,
           (unsigned long long)n_wg, n_waves, active, ms * 1e3, (t_max - t_min) / 100.0);
    {   // the clock the chip held during the streaming phase, and what the SIMDs issued in it
        std::vector<double> clk, cyc;
        for (size_t w = 0; w < n_waves; ++w) {
            const uint64_t *r = &tr[w * TW];
            if (r[3] == 0 || r[0] == 0 || r[2] <= r[1]) continue;
            clk.push_back((double)(r[6] - r[5]) / (double)(r[2] - r[1]) * 0.1);
            cyc.push_back((double)(r[6] - r[5]));
        }
        std::sort(clk.begin(), clk.end());
        std::sort(cyc.begin(), cyc.end());
        // one wave of a k-sliced workgroup: ss64 / 4 chunks x R rows x JL columns x (28 + 2 half-rate) = 32 issue slots each
        double slots = (double)(ss64 / 4 / slices) * R * JL * 32.0;
        if (shape == 999) {   // a wave walks tr[7] stages of 2 chunks each (median over waves)
            std::vector<double> st;
            for (size_t w = 0; w < n_waves; ++w) if (tr[w * TW + 3]) st.push_back((double)tr[w * TW + 7]);
            std::sort(st.begin(), st.end());
            slots = st[st.size() / 2] * 2.0 * R * JL * 32.0;
            printf("  persistent: stages per wave min %.0f median %.0f max %.0f\n", st.front(), st[st.size() / 2], st.back());
            // part-end accounting (second trace region): mean cycles per wave in each section
            std::vector<uint64_t> ext(4096 * 8);
            CK(hipMemcpy(ext.data(), (uint64_t *)dTrace + 4096 * 8, ext.size() * 8, hipMemcpyDeviceToHost));
            double sums[5] = {0, 0, 0, 0, 0};
            for (size_t w = 0; w < 4096; ++w) for (int x = 0; x < 5; ++x) sums[x] += (double)ext[w * 8 + x];
            printf("  persistent part ends, mean per wave: parts %.2f; cycles to barrier 1 %.0f, reduce+store %.0f, to barrier 2 %.0f, next-part setup %.0f\n",
                   sums[4] / 4096, sums[0] / 4096, sums[1] / 4096, sums[2] / 4096, sums[3] / 4096);
        }
        printf("  in-kernel clock while streaming: p10 %.3f median %.3f p90 %.3f GHz; streaming phase of a wave: median %.0f cycles = %.2f cycles per issue slot per wave (%.0f slots)\n",
               clk[clk.size() / 10], clk[clk.size() / 2], clk[clk.size() * 9 / 10], cyc[cyc.size() / 2],
               cyc[cyc.size() / 2] / slots, slots);
    }
    std::vector<double> start, wait0, stream, tail, total;
    std::map<uint64_t, int> per_cu;   // (xcc, se, cu) -> waves
    for (size_t w = 0; w < n_waves; ++w) {
        const uint64_t *r = &tr[w * TW];
        if (r[3] == 0 || r[0] == 0) continue;
        start.push_back((r[0] - t_min) / 100.0);
        wait0.push_back((r[1] - r[0]) / 100.0);
        stream.push_back((r[2] - r[1]) / 100.0);
        tail.push_back((r[3] - r[2]) / 100.0);
        total.push_back((r[3] - r[0]) / 100.0);
        const uint32_t hw = (uint32_t)r[4], xcc = (uint32_t)(r[4] >> 32) & 0xF;
        const uint32_t cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;
        per_cu[((uint64_t)xcc << 16) | (se << 8) | (sh << 4) | cu]++;
    }
    if (start.empty()) { printf("no trace records\n"); return 1; }
    {   // resident waves over time, 10 us buckets
        printf("  resident waves at t =");
        for (double t = 5.0; t < (t_max - t_min) / 100.0; t += 10.0) {
            size_t live = 0;
            for (size_t w = 0; w < n_waves; ++w) {
                const uint64_t *r = &tr[w * TW];
                if (r[3] == 0 || r[0] == 0) continue;
                const double a = (r[0] - t_min) / 100.0, b = (r[3] - t_min) / 100.0;
                if (a <= t && t < b) ++live;
            }
            printf(" %.0f:%zu", t, live);
        }
        printf("\n");
    }
    auto stats = [](std::vector<double> v, const char *name) {
        std::sort(v.begin(), v.end());
        double sum = 0; for (double x : v) sum += x;
        printf("  %-28s min %7.2f  p10 %7.2f  median %7.2f  p90 %7.2f  max %7.2f  mean %7.2f us\n", name, v.front(),
               v[v.size() / 10], v[v.size() / 2], v[v.size() * 9 / 10], v.back(), sum / v.size());
    };
    stats(start, "start after first wave");
    stats(wait0, "wait for first rows (DMA)");
    stats(stream, "streaming chunks");
    stats(tail, "reduce + store");
    stats(total, "wave lifetime");
    // who is slow?  streaming time of first-round waves by XCD, by k-mer length, by wave-in-workgroup
    {
        std::map<int, std::pair<double, int>> by_xcc, by_k, by_wave, by_se;
        const int wpw_ = getenv("SKL_KSLICE_WAVES") ? atoi(getenv("SKL_KSLICE_WAVES")) : 4;
        for (size_t w = 0; w < n_waves; ++w) {
            const uint64_t *r = &tr[w * TW];
            if (r[3] == 0 || r[0] == 0) continue;
            if ((r[0] - t_min) / 100.0 > 10.0) continue;   // first round only
            const double st = (r[2] - r[1]) / 100.0;
            const uint32_t hw = (uint32_t)r[4], xcc = (uint32_t)(r[4] >> 32) & 0xF;
            const size_t wg = w / wpw_;
            auto add = [&](std::map<int, std::pair<double, int>> &m, int key) { m[key].first += st; m[key].second++; };
            add(by_xcc, (int)xcc);
            add(by_k, (int)(((wg >> 3) % (KSL_TILE_BLOCK * nk)) / KSL_TILE_BLOCK));
            add(by_wave, (int)(w % wpw_));
            add(by_se, (int)((hw >> 13) & 0x7));
        }
        auto show = [](const char *name, std::map<int, std::pair<double, int>> &m) {
            printf("  first-round streaming us by %s:", name);
            for (auto &kv : m) printf(" %d:%.1f(n=%d)", kv.first, kv.second.first / kv.second.second, kv.second.second);
            printf("\n");
        };
        // per-CU means of first-round streaming time, and the spread inside one workgroup
        {
            std::map<uint64_t, std::pair<double, int>> cu;
            std::map<size_t, std::pair<double, double>> wgmm;
            std::vector<double> first;
            for (size_t w = 0; w < n_waves; ++w) {
                const uint64_t *r = &tr[w * TW];
                if (r[3] == 0 || r[0] == 0) continue;
                if ((r[0] - t_min) / 100.0 > 10.0) continue;
                const double st = (r[2] - r[1]) / 100.0;
                first.push_back(st);
                const uint32_t hw = (uint32_t)r[4], xcc = (uint32_t)(r[4] >> 32) & 0xF;
                const uint64_t key = ((uint64_t)xcc << 16) | (hw & 0xFF00);   // XCC + SE/SH/CU bits
                cu[key].first += st;
                cu[key].second++;
                auto &mm = wgmm[w / wpw_];
                if (mm.first == 0) mm = {st, st};
                mm.first = std::min(mm.first, st);
                mm.second = std::max(mm.second, st);
            }
            std::sort(first.begin(), first.end());
            printf("  first-round streaming: min %.1f p10 %.1f median %.1f p90 %.1f max %.1f\n", first.front(),
                   first[first.size() / 10], first[first.size() / 2], first[first.size() * 9 / 10], first.back());
            std::vector<double> cum;
            for (auto &kv : cu) cum.push_back(kv.second.first / kv.second.second);
            std::sort(cum.begin(), cum.end());
            printf("  per-CU mean (n=%zu CUs): min %.1f p10 %.1f median %.1f p90 %.1f max %.1f\n", cum.size(), cum.front(),
                   cum[cum.size() / 10], cum[cum.size() / 2], cum[cum.size() * 9 / 10], cum.back());
            double spread = 0;
            for (auto &kv : wgmm) spread += kv.second.second - kv.second.first;
            printf("  mean (max - min) inside a workgroup: %.2f us\n", spread / wgmm.size());
        }
        show("XCD", by_xcc);
        show("k index", by_k);
        show("wave in workgroup", by_wave);
        show("shader engine", by_se);
    }
    // concurrency over time
    const double span = (t_max - t_min) / 100.0;
    const int bins = 20;
    printf("  waves in flight per %.1f us bin:", span / bins);
    for (int b = 0; b < bins; ++b) {
        const double t = (b + 0.5) * span / bins;
        size_t c = 0;
        for (size_t i = 0; i < start.size(); ++i) c += (start[i] <= t && start[i] + total[i] > t);
        printf(" %zu", c);
    }
    printf("\n  CUs used: %zu; waves per CU min/max:", per_cu.size());
    int mn = 1 << 30, mx = 0;
    for (auto &kv : per_cu) { mn = std::min(mn, kv.second); mx = std::max(mx, kv.second); }
    printf(" %d / %d\n", mn, mx);
    return 0;
}
