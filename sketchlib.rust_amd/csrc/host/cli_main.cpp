// cli_main.cpp -- `sketchlib dist`, the reference's command line for this path
// (src/cli.rs:185-231, src/lib.rs:303-453) on top of the gfx950 engine.
//
// Same positional arguments, flags, defaults, stdout format and exit behaviour
// (panic -> 101, error -> 1, usage -> 2).  `--threads` (default 1, as in the reference) sets
// the host threads that format the dense text output; the distances themselves run on the
// GPU.  `--device` is the one addition.
#include <atomic>
#include <mutex>
#include <future>
#include <thread>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <algorithm>
#include <iostream>
#include <map>
#include <memory>
#include <optional>
#include <string>
#include <vector>

#include "distances.hpp"
#include "inverted.hpp"
#include "sketchlib_dist.h"
#include "io.hpp"
#include "multisketch.hpp"
#include "sketch.hpp"

using namespace skl_host;

namespace {

struct DistArgs {
    std::string ref_db;
    std::optional<std::string> query_db, output, subset, ref_completeness_file, query_completeness_file;
    std::optional<size_t> knn, kmer;
    int knn_ties = SKL_KNN_TIES_REFERENCE;
    bool ani = false;
    size_t threads = 1;
    double completeness_cutoff = 0.64;
    bool verbose = false, quiet = false;
    std::vector<int> devices = {0};   // --device D | --devices a,b,.. | --gpus N
    bool npy = false;                  // --npy: dense output as a NumPy .npy array instead of text
    size_t band_bytes = 0;             // --band-mb: host memory per streamed output band (0: default)
    size_t band() const
    {
        // finer-than-MB override, for tests that want many bands on a small database
        if (const char *e = std::getenv("SKL_DIST_BAND_BYTES")) return std::max<size_t>(1, std::strtoull(e, nullptr, 10));
        // default: 256 MB, and at least 8 MB per output thread (a band is a barrier for the workers that format it)
        return band_bytes ? band_bytes : std::max<size_t>(256ull << 20, threads * (8ull << 20));
    }
};

const char *g_usage = "sketchlib dist [OPTIONS] <REF_DB> [QUERY_DB]";

[[noreturn]] void usage_error(const std::string &msg)
{
    std::cerr << "error: " << msg << "\n\n"
              << "Usage: " << g_usage << "\n\n"
              << "For more information, try '--help'.\n";
    std::exit(2);
}

void print_help()
{
    std::cout <<
        "Calculate pairwise distances using sketches\n\n"
        "Usage: sketchlib dist [OPTIONS] <REF_DB> [QUERY_DB]\n\n"
        "Arguments:\n"
        "  <REF_DB>    The .skm file used as the reference\n"
        "  [QUERY_DB]  The .skm file used as the query (omit for ref v ref)\n\n"
        "Options:\n"
        "  -o <OUTPUT>                     Output filename (omit to output to stdout)\n"
        "      --knn <KNN>                 Calculate sparse distances with k nearest-neighbours (ref-vs-ref or ref-vs-query)\n"
        "      --knn-ties <RULE>           Neighbours at EQUAL distance: reference (default) = exactly the ids and order\n"
        "                                  the reference binary prints (its BinaryHeap replayed on the GPU);\n"
        "                                  canonical = lowest index first (a property of the data alone; the distances\n"
        "                                  per row are the reference's, tied rows may list other ids).  Either rule\n"
        "                                  evaluates every pair once on any number of GPUs\n"
        "      --subset <SUBSET>           Sample names to analyse\n"
        "  -k <KMER>                       K-mer length (if provided only calculate Jaccard distance)\n"
        "      --ani                       Calculate ANI rather than Jaccard dists, using Poisson model\n"
        "      --threads <THREADS>         Number of CPU threads [default: 1]\n"
        "      --ref-completeness-file <F> File listing reference sample completeness estimates 0.0-1.0\n"
        "      --query-completeness-file <F> File listing query sample completeness estimates 0.0-1.0\n"
        "      --completeness-cutoff <C>   minimum completeness product for the correction [default: 0.64]\n"
        "      --device <D>                GPU to run on [default: 0]\n"
        "      --npy                       Dense output as a NumPy .npy array (f32, one row per pair in the\n"
        "                                  order of the text lines) instead of text; needs -o\n"
        "      --band-mb <MB>              Host memory per streamed dense output band [default: 256, and at least 8 per thread]\n"
        "      --gpus <N>                  Split the pair space over GPUs 0..N-1 (row bands)\n"
        "      --devices <LIST>            Same, with an explicit comma separated device list\n"
        "  -v, --verbose                   Show progress messages\n"
        "      --quiet                     Don't show any messages\n"
        "  -h, --help                      Print help\n";
}

size_t parse_usize(const std::string &flag, const std::string &v)
{
    char *end = nullptr;
    if (v.empty() || v[0] == '-') usage_error("invalid value '" + v + "' for '" + flag + "': invalid digit found in string");
    const unsigned long long x = std::strtoull(v.c_str(), &end, 10);
    if (end != v.c_str() + v.size()) usage_error("invalid value '" + v + "' for '" + flag + "': invalid digit found in string");
    return (size_t)x;
}

DistArgs parse_dist(int argc, char **argv, int first)
{
    DistArgs a;
    std::vector<std::string> positional;
    for (int i = first; i < argc; ++i) {
        std::string arg = argv[i];
        std::optional<std::string> inline_val;
        if (arg.rfind("--", 0) == 0) {
            const size_t eq = arg.find('=');
            if (eq != std::string::npos) {
                inline_val = arg.substr(eq + 1);
                arg = arg.substr(0, eq);
            }
        }
        auto value = [&](const std::string &flag) -> std::string {
            if (inline_val) return *inline_val;
            if (i + 1 >= argc) usage_error("a value is required for '" + flag + "' but none was supplied");
            return argv[++i];
        };
        if (arg == "-h" || arg == "--help") { print_help(); std::exit(0); }
        else if (arg == "-v" || arg == "--verbose") a.verbose = true;
        else if (arg == "--quiet") a.quiet = true;
        else if (arg == "-o") a.output = value("-o <OUTPUT>");
        else if (arg == "--knn") a.knn = parse_usize("--knn <KNN>", value("--knn <KNN>"));
        else if (arg == "--knn-ties") {
            const std::string v = value("--knn-ties <RULE>");
            if (v == "canonical") a.knn_ties = SKL_KNN_TIES_CANONICAL;
            else if (v == "reference") a.knn_ties = SKL_KNN_TIES_REFERENCE;
            else usage_error("invalid value '" + v + "' for '--knn-ties <RULE>': possible values: canonical, reference");
        }
        else if (arg == "--subset") a.subset = value("--subset <SUBSET>");
        else if (arg == "-k") a.kmer = parse_usize("-k <KMER>", value("-k <KMER>"));
        else if (arg == "--ani") a.ani = true;
        else if (arg == "--threads") {
            const std::string v = value("--threads <THREADS>");
            char *end = nullptr;
            const long long t = std::strtoll(v.c_str(), &end, 10);
            if (v.empty() || end != v.c_str() + v.size()) {
                usage_error("invalid value '" + v + "' for '--threads <THREADS>': `" + v + "` isn't a valid number of cores");
            }
            if (t < 1) usage_error("invalid value '" + v + "' for '--threads <THREADS>': Threads must be one or higher");
            a.threads = (size_t)t;
        }
        else if (arg == "--ref-completeness-file") a.ref_completeness_file = value(arg);
        else if (arg == "--query-completeness-file") a.query_completeness_file = value(arg);
        else if (arg == "--completeness-cutoff") {
            const std::string v = value(arg);
            char *end = nullptr;
            a.completeness_cutoff = std::strtod(v.c_str(), &end);
            if (v.empty() || end != v.c_str() + v.size()) usage_error("invalid value '" + v + "' for '--completeness-cutoff <COMPLETENESS_CUTOFF>': invalid float literal");
        }
        else if (arg == "--device") a.devices = {(int)parse_usize("--device <D>", value(arg))};
        else if (arg == "--npy") a.npy = true;
        else if (arg == "--band-mb") a.band_bytes = std::max<size_t>(1, parse_usize("--band-mb <MB>", value(arg))) << 20;
        else if (arg == "--gpus") {
            const size_t ngpu = parse_usize("--gpus <N>", value(arg));
            if (ngpu < 1) usage_error("invalid value for '--gpus <N>': must be one or higher");
            a.devices.clear();
            for (size_t d = 0; d < ngpu; ++d) a.devices.push_back((int)d);
        }
        else if (arg == "--devices") {
            a.devices.clear();
            const std::string v = value(arg);
            size_t pos = 0;
            while (pos <= v.size()) {
                const size_t comma = v.find(',', pos);
                a.devices.push_back((int)parse_usize("--devices <LIST>", v.substr(pos, comma == std::string::npos ? std::string::npos : comma - pos)));
                if (comma == std::string::npos) break;
                pos = comma + 1;
            }
        }
        else if (arg.size() > 2 && arg[0] == '-' && arg[1] != '-' && (arg[1] == 'o' || arg[1] == 'k')) {
            // clap accepts -k21 / -oFILE
            if (arg[1] == 'o') a.output = arg.substr(2);
            else a.kmer = parse_usize("-k <KMER>", arg.substr(2));
        }
        else if (arg.size() > 1 && arg[0] == '-') usage_error("unexpected argument '" + arg + "' found");
        else positional.push_back(arg);
    }
    if (positional.empty()) usage_error("the following required arguments were not provided:\n  <REF_DB>");
    if (positional.size() > 2) usage_error("unexpected argument '" + positional[2] + "' found");
    a.ref_db = positional[0];
    if (positional.size() == 2) a.query_db = positional[1];
    if (a.ani && !a.kmer) {  // #[arg(long, requires("kmer"))]
        usage_error("the following required arguments were not provided:\n  -k <KMER>");
    }
    return a;
}

// skl_ctx_flags() & SKL_CTX_FLAG_LOG_UNMATCHED, printed once when a completeness correction is asked for
const char *const LOG_UNMATCHED_WARNING =
    "this host's libm log() is neither form of glibc's x86-64 log the GPU reproduces bit for bit: with a completeness "
    "correction, core distances of flat fits (the same bin-match count at every k-mer length) may be 0 where a CPU run "
    "of sketchlib on this host prints 1, or the reverse; all other values agree to 1e-6";

struct Logger {
    bool info_on, warn_on;
    void info(const std::string &m) const { if (info_on) std::cerr << "INFO  [sketchlib] " << m << "\n"; }
    void warn(const std::string &m) const { if (warn_on) std::cerr << "WARN  [sketchlib] " << m << "\n"; }
};

// A dense matrix held whole in memory (multi-device path): text or .npy.
void write_whole(const DistanceMatrix &d, TextSink &sink, size_t n, const DistArgs &a)
{
    if (!a.npy) {
        d.write_rows(sink, 0, n, d.distances.data(), a.threads);
        return;
    }
    const std::string h = distances::npy_header(d.n_distances, d.jaccard.n_dist_cols());
    sink.finish(sink.begin(h.data(), h.size()), h.data(), h.size());
    const char *bytes = reinterpret_cast<const char *>(d.distances.data());
    const size_t len = d.distances.size() * sizeof(float);
    sink.finish(sink.begin(bytes, len), bytes, len);
}

// The listing is complete and flushed: leave without tearing down what the operating system reclaims anyway -- the slabs
// in HBM, gigabytes of host vectors, the worker pool, the HIP runtime (0.05 s of a 0.19 s run on 1 000 genomes, 0.5 s
// after a million).  Files are written with pwrite (nothing buffered in the process); SKL_CLI_FAST_EXIT=0 keeps the
// orderly exit.
// (called by main() AFTER its last line of output -- the verbose "Complete in" line -- with the command's exit code; a
// failed flush of the listing (ENOSPC, EPIPE) turns a success into exit code 1 and takes the orderly way out)
bool g_listing_complete = false;   // set by run_dist / run_inverted once the listing is written

int leave_after_success(int rc)
{
    std::cout.flush();
    std::cerr.flush();
    const bool flushed = std::cout.good() && fflush(nullptr) == 0;
    if (!flushed) {
        std::fprintf(stderr, "Error: writing the output failed\n");
        return rc ? rc : 1;
    }
    if (rc != 0 || !g_listing_complete) return rc;
    const char *e = std::getenv("SKL_CLI_FAST_EXIT");
    if (e && e[0] == '0') return rc;
    // a profiler or another preloaded tool writes its results from exit handlers: leave in order for it
    for (const char *tool : {"LD_PRELOAD", "HSA_TOOLS_LIB", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD"}) {
        const char *v = std::getenv(tool);
        if (v && v[0]) return rc;
    }
    std::_Exit(0);
}

int run_dist(const DistArgs &a)
{
    if (a.npy && (!a.output || a.knn)) {
        usage_error("--npy needs -o <file> and a dense (no --knn) output");
    }
    const bool timing = std::getenv("SKL_CLI_TIMING") != nullptr;
    const auto t_start = std::chrono::steady_clock::now();
    auto since_start = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count(); };
    double t_loaded = 0, t_device = 0;
    // lib.rs:230-237: verbose -> Info, quiet -> Error, default -> Warn
    const Logger log{a.verbose && !a.quiet, !a.quiet};
    log.info("Using " + std::to_string(a.threads) + " threads");   // cli.rs:75-86 (host threads unused)
    if (a.threads > 2 * host_cpu_budget()) {
        log.info("This process may use " + std::to_string(host_cpu_budget()) + " CPUs (affinity / cgroup quota): the output is formatted by " +
                 std::to_string(2 * host_cpu_budget()) + " threads");
    }
    // The device contexts (HIP runtime start, stream, first allocations: 0.1-0.25 s) come up on their own thread while the
    // database is read; whatever goes wrong there is reported where the contexts are first needed, after the loading errors.
    std::future<std::unique_ptr<DeviceSet>> dev_starting =
        std::async(std::launch::async, [devices = a.devices] { return std::make_unique<DeviceSet>(devices); });

    // listings go through a TextSink (regular file: blocks written at offsets from all
    // formatting threads; stdout: in order)
    std::ostream *os = &std::cout;
    std::unique_ptr<TextSink> sink;
    if (a.output) {
        try {
            sink = std::make_unique<FileSink>(*a.output);
        } catch (const std::exception &e) {
            throw Panic(e.what());
        }
    }
    if (!sink) sink = std::make_unique<StreamSink>(*os);

    const std::string ref_db_name = strip_sketch_extension(a.ref_db);
    MultiSketch references;
    try {
        references = MultiSketch::load_metadata(ref_db_name);
    } catch (const std::exception &) {
        throw Panic("Could not read sketch metadata from " + a.ref_db + ".skm");  // lib.rs:322-323
    }
    // A distance at one k-mer length only ever reads that slice of each sample (jaccard.rs:6-45 through get_sketch_slice):
    // of a database with several lengths, only that slice is read, held and uploaded.  Not when the query database lists
    // other lengths than the reference database: that pair of databases is refused by the library (SKL_ERR_INCOMPATIBLE;
    // the reference would apply the reference database's k INDEX to the queries, mod.rs:253-269), as before.
    std::optional<MultiSketch> queries_peeked;
    if (a.kmer && references.kmer_lengths().size() > 1 && references.get_k_idx(*a.kmer)) {
        bool same_lengths = true;
        if (a.query_db) {
            try {
                queries_peeked = MultiSketch::load_metadata(strip_sketch_extension(*a.query_db));
                same_lengths = queries_peeked->kmer_lengths() == references.kmer_lengths();
            } catch (const std::exception &) {
                same_lengths = false;   // (reported where the reference reports it, after the reference database is read)
            }
        }
        if (same_lengths) {
            const size_t k_idx = *references.get_k_idx(*a.kmer);
            references.select_kmer(k_idx);
            if (queries_peeked) queries_peeked->select_kmer(k_idx);
        }
    }
    log.info("Loading sketch data from " + ref_db_name + ".skd");
    try {
        if (a.subset) {
            references.read_sketch_data_block(ref_db_name, read_subset_names(*a.subset));
        } else {
            references.read_sketch_data(ref_db_name);
        }
    } catch (const std::exception &e) {
        throw Panic(e.what());
    }
    const size_t n = references.number_samples_loaded();

    std::vector<std::string> warnings;
    std::optional<std::vector<double>> ref_comp;
    if (a.ref_completeness_file) {
        // an unreadable / malformed file is an Err returned from main in the reference (lib.rs:334-339,
        // `?`): "Error: ..." and exit code 1, which is what main() here does with std::exception
        ref_comp = read_completeness_file(*a.ref_completeness_file, references, &warnings);
    }

    DistType dist_type;
    try {
        dist_type = distances::set_k(references, a.kmer, a.ani);
    } catch (const std::exception &e) {
        throw Panic(std::string("Error setting k size: ") + e.what());  // lib.rs:341-343
    }
    log.info(dist_type.describe());

    std::optional<MultiSketch> queries;
    if (a.query_db) {
        const std::string query_db_name = strip_sketch_extension(*a.query_db);
        try {
            if (queries_peeked) queries = std::move(*queries_peeked);
            else queries = MultiSketch::load_metadata(query_db_name);
        } catch (const std::exception &) {
            throw Panic("Could not read sketch metadata from " + *a.query_db + ".skm");
        }
        log.info("Loading query sketch data from " + query_db_name + ".skd");
        try {
            queries->read_sketch_data(query_db_name);   // panics in the reference too (lib.rs:351)
        } catch (const std::exception &e) {
            throw Panic(e.what());
        }
    }
    std::optional<std::vector<double>> query_comp;
    if (queries && a.query_completeness_file) {
        query_comp = read_completeness_file(*a.query_completeness_file, *queries, &warnings);
    }
    for (const auto &w : warnings) log.warn(w);

    t_loaded = since_start();
    const std::unique_ptr<DeviceSet> dev_owner = dev_starting.get();
    DeviceSet &dev = *dev_owner;
    t_device = since_start();
    if (a.devices.size() > 1) log.info("Using " + std::to_string(a.devices.size()) + " GPU contexts (row-band partition)");
    for (size_t d = 0; d < dev.size(); ++d) {
        if (skl_ctx_set_knn_ties(dev[d].ctx(), a.knn_ties) != SKL_OK) throw std::runtime_error(skl_last_error());
    }
    if ((ref_comp || query_comp) && (skl_ctx_flags(dev[0].ctx()) & SKL_CTX_FLAG_LOG_UNMATCHED)) log.warn(LOG_UNMATCHED_WARNING);
    const std::vector<double> *rc = ref_comp ? &*ref_comp : nullptr;
    const std::vector<double> *qc = query_comp ? &*query_comp : nullptr;
    if (!queries) {
        if (!a.knn) {
            log.info("Calculating all ref vs ref distances");
            if (n < 2) {
                if (dist_type.kind == DistType::CoreAcc && references.kmer_lengths().size() < 2) {
                    throw Panic("Need at least two k-mer lengths to calculate core/accessory distances");
                }
                return 0;  // empty upper triangle
            }
            if (a.devices.size() == 1) {
                // one device: stream row bands (compute band i+1 while band i is written)
                log.info("Writing out in long matrix form");
                distances::self_dists_all_streamed(dev[0], references, n, dist_type, rc, a.completeness_cutoff,
                                                   *sink, a.threads, a.band(), a.npy);
            } else {
                const DistanceMatrix d = distances::self_dists_all(dev, references, n, dist_type, a.quiet, rc,
                                                                   a.completeness_cutoff);
                log.info("Writing out in long matrix form");
                write_whole(d, *sink, n, a);
            }
        } else {
            size_t nn = *a.knn;
            if (nn >= n) {  // lib.rs:379-382
                log.warn("knn=" + std::to_string(nn) + " is higher than number of samples=" + std::to_string(n));
                nn = n - 1;
            }
            if (nn == 0) throw Panic("chunk size must be non-zero");  // par_chunks_mut(0)
            log.info("Calculating sparse ref vs ref distances with " + std::to_string(nn) + " nearest neighbours");
            const SparseDistanceMatrix d = distances::self_dists_knn(dev, references, n, nn, dist_type, a.quiet,
                                                                     rc, a.completeness_cutoff);
            log.info("Writing out in sparse matrix form");
            d.write(*sink, a.threads);
        }
    } else {
        const size_t n_query = queries->number_samples_loaded();
        if (a.knn) {
            size_t nn = *a.knn;
            if (nn > n) {  // lib.rs:411-414
                log.warn("knn=" + std::to_string(nn) + " is higher than number of reference samples=" + std::to_string(n));
                nn = n;
            }
            log.info("Calculating sparse ref vs query distances with " + std::to_string(nn) + " nearest neighbours");
            const SparseDistanceMatrix d = distances::cross_dists_knn(dev, references, *queries, n, n_query, nn,
                                                                      dist_type, a.quiet, rc, qc,
                                                                      a.completeness_cutoff);
            log.info("Writing out in sparse matrix form");
            d.write(*sink, a.threads);
        } else {
            log.info("Calculating all ref vs query distances");
            if (a.devices.size() == 1) {
                log.info("Writing out in long matrix form");
                distances::cross_dists_all_streamed(dev[0], references, *queries, n, n_query, dist_type, rc, qc,
                                                    a.completeness_cutoff, *sink, a.threads, a.band(), a.npy);
            } else {
                const DistanceMatrix d = distances::cross_dists_all(dev, references, *queries, n, n_query, dist_type,
                                                                    a.quiet, rc, qc, a.completeness_cutoff);
                log.info("Writing out in long matrix form");
                write_whole(d, *sink, n, a);
            }
        }
    }
    os->flush();
    if (timing) {
        const OutputTiming &t = output_timing();
        std::fprintf(stderr, "TIMING load=%.3fs device_wait=%.3fs dist+output=%.3fs (gpu_wait=%.3fs format=%.3fs sink=%.3fs)\n",
                     t_loaded, t_device - t_loaded, since_start() - t_device, t.wait_s, t.format_s, t.sink_s);
    }
    g_listing_complete = true;
    return 0;
}

// ---- `sketchlib sketch` (src/cli.rs:121-183, src/lib.rs:242-301), DNA assemblies ----
struct SketchArgs {
    std::vector<std::string> seq_files;
    std::optional<std::string> file_list, output;
    std::vector<size_t> k_vals, k_seq;
    uint64_t sketch_size = 1000;   // DEFAULT_SKETCHSIZE, cli.rs:17
    bool single_strand = false;
    size_t threads = 1;
    bool verbose = false, quiet = false;
    int gpu = -1;   // --gpu [D]: hash and take bin minima on device D (default: CPU)
};

std::vector<size_t> parse_list(const std::string &flag, const std::string &v)
{
    std::vector<size_t> out;
    size_t pos = 0;
    while (pos <= v.size()) {
        const size_t comma = v.find(',', pos);
        const std::string tok = v.substr(pos, comma == std::string::npos ? std::string::npos : comma - pos);
        out.push_back(parse_usize(flag, tok));
        if (comma == std::string::npos) break;
        pos = comma + 1;
    }
    return out;
}

int run_sketch(int argc, char **argv, int first, bool verbose, bool quiet)
{
    g_usage = "sketchlib sketch [OPTIONS] -o <OUTPUT> <--k-vals <K_VALS>|--k-seq <K_SEQ>> <SEQ_FILES|-f <FILE_LIST>>";
    SketchArgs a;
    a.verbose = verbose;
    a.quiet = quiet;
    for (int i = first; i < argc; ++i) {
        std::string arg = argv[i];
        auto value = [&](const std::string &flag) -> std::string {
            if (i + 1 >= argc) usage_error("a value is required for '" + flag + "' but none was supplied");
            return argv[++i];
        };
        if (arg == "-v" || arg == "--verbose") a.verbose = true;
        else if (arg == "--quiet") a.quiet = true;
        else if (arg == "-f") a.file_list = value("-f <FILE_LIST>");
        else if (arg == "-o") a.output = value("-o <OUTPUT>");
        else if (arg == "-k" || arg == "--k-vals") a.k_vals = parse_list("--k-vals <K_VALS>", value(arg));
        else if (arg == "--k-seq") a.k_seq = parse_list("--k-seq <K_SEQ>", value(arg));
        else if (arg == "-s" || arg == "--sketch-size") a.sketch_size = parse_usize("--sketch-size <SKETCH_SIZE>", value(arg));
        else if (arg == "--single-strand") a.single_strand = true;
        else if (arg == "--threads") a.threads = std::max<size_t>(1, parse_usize("--threads <THREADS>", value(arg)));
        else if (arg == "--gpu") a.gpu = 0;
        else if (arg == "--device") a.gpu = (int)parse_usize("--device <D>", value(arg));
        else if (arg == "--seq-type") {
            const std::string v = value(arg);
            if (v != "dna") { std::cerr << "error: this build sketches DNA assemblies only (--seq-type " << v << ")\n"; return 2; }
        }
        else if (arg == "--min-count" || arg == "--min-qual" || arg == "--level") (void)value(arg);
        else if (arg.size() > 1 && arg[0] == '-') usage_error("unexpected argument '" + arg + "' found");
        else a.seq_files.push_back(arg);
    }
    if (!a.output) usage_error("the following required arguments were not provided:\n  -o <OUTPUT>");
    if (a.seq_files.empty() == !a.file_list.has_value()) {
        usage_error("exactly one of <SEQ_FILES>... or -f <FILE_LIST> must be given");
    }
    if (a.k_vals.empty() == a.k_seq.empty()) usage_error("exactly one of --k-vals or --k-seq must be given");
    const Logger log{a.verbose && !a.quiet, !a.quiet};
    const std::vector<InputFastx> inputs = a.file_list ? read_rfile(*a.file_list) : read_input_fastas(a.seq_files);
    log.info("Parsed " + std::to_string(inputs.size()) + " samples in input list");
    const std::vector<size_t> kmers = parse_kmers(a.k_vals, a.k_seq);
    const uint64_t bins = (a.sketch_size + 63) / 64 * 64;
    log.info("Running sketching: sketch_size:" + std::to_string(bins) + " threads:" + std::to_string(a.threads));
    try {
        if (a.gpu >= 0) {
            log.info("Hashing on GPU " + std::to_string(a.gpu));
            Device dev(a.gpu);
            sketch_files_gpu(dev, *a.output, inputs, kmers, a.sketch_size, !a.single_strand, a.threads);
        } else {
            sketch_files(*a.output, inputs, kmers, a.sketch_size, !a.single_strand, a.threads);
        }
    } catch (const std::exception &e) {
        throw Panic(e.what());   // the reference panics on unreadable / empty input
    }
    return 0;
}

// ---- `sketchlib inverted build|precluster` (src/cli.rs:329-461, src/lib.rs:485-600,682-789) ----
int run_inverted(int argc, char **argv, int first, bool verbose, bool quiet)
{
    if (first >= argc) usage_error("'sketchlib inverted' requires a subcommand: build | precluster");
    const std::string sub = argv[first];
    auto next_value = [&](int &i, const std::string &flag) -> std::string {
        if (i + 1 >= argc) usage_error("a value is required for '" + flag + "' but none was supplied");
        return argv[++i];
    };
    if (sub == "build") {
        g_usage = "sketchlib inverted build [OPTIONS] -o <OUTPUT> <SEQ_FILES|-f <FILE_LIST>>";
        std::vector<std::string> seq_files;
        std::optional<std::string> file_list, output, species_names, metadata_file;
        bool write_skq_flag = false, single_strand = false;
        uint64_t sketch_size = 1000;   // DEFAULT_SKETCHSIZE, cli.rs:17
        size_t kmer = 21, threads = 1; // DEFAULT_KMER, cli.rs:13
        for (int i = first + 1; i < argc; ++i) {
            const std::string arg = argv[i];
            if (arg == "-v" || arg == "--verbose") verbose = true;
            else if (arg == "--quiet") quiet = true;
            else if (arg == "-f") file_list = next_value(i, "-f <FILE_LIST>");
            else if (arg == "-o") output = next_value(i, "-o <OUTPUT>");
            else if (arg == "--write-skq") write_skq_flag = true;
            else if (arg == "--species-names") species_names = next_value(i, arg);
            else if (arg == "--metadata") metadata_file = next_value(i, arg);
            else if (arg == "-s" || arg == "--sketch-size") sketch_size = parse_usize("--sketch-size <SKETCH_SIZE>", next_value(i, arg));
            else if (arg == "-k" || arg == "--kmer-length") kmer = parse_usize("--kmer-length <KMER_LENGTH>", next_value(i, arg));
            else if (arg == "--single-strand") single_strand = true;
            else if (arg == "--threads") threads = std::max<size_t>(1, parse_usize("--threads <THREADS>", next_value(i, arg)));
            else if (arg == "--min-count" || arg == "--min-qual") (void)next_value(i, arg);
            else if (arg.size() > 1 && arg[0] == '-') usage_error("unexpected argument '" + arg + "' found");
            else seq_files.push_back(arg);
        }
        if (!output) usage_error("the following required arguments were not provided:\n  -o <OUTPUT>");
        if (seq_files.empty() == !file_list.has_value()) {
            usage_error("exactly one of <SEQ_FILES>... or -f <FILE_LIST> must be given");
        }
        const Logger log{verbose && !quiet, !quiet};
        log.info("Getting input files");
        const std::vector<InputFastx> inputs = file_list ? read_rfile(*file_list) : read_input_fastas(seq_files);
        log.info("Parsed " + std::to_string(inputs.size()) + " samples in input list");
        {
            std::vector<std::string> names;
            for (const auto &in : inputs) names.push_back(in.first);
            std::sort(names.begin(), names.end());
            if (std::adjacent_find(names.begin(), names.end()) != names.end()) {
                std::cerr << "error: samples listed more than once (multi-entry samples) are not part of this build\n";
                return 2;
            }
        }
        std::optional<std::vector<std::string>> labels;
        std::vector<size_t> order(inputs.size());
        for (size_t i = 0; i < order.size(); ++i) order[i] = i;
        try {
            if (species_names) {
                log.info("Reordering samples using labels in " + *species_names);
                order = reorder_by_labels(inputs, *species_names, &labels);
            }
            log.info("Creating sketches");
            std::vector<std::vector<uint16_t>> sketches(inputs.size());
            std::vector<std::string> names(inputs.size());
            {
                std::atomic<size_t> next{0};
                std::exception_ptr err;
                std::mutex mu;
                auto work = [&] {
                    try {
                        for (;;) {
                            const size_t i = next.fetch_add(1);
                            if (i >= inputs.size()) break;
                            sketches[order[i]] = sketch_sample_inverted(inputs[i], kmer, sketch_size, !single_strand);
                            names[order[i]] = inputs[i].first;
                        }
                    } catch (...) {
                        std::lock_guard<std::mutex> lk(mu);
                        err = std::current_exception();
                    }
                };
                std::vector<std::thread> pool;
                for (size_t t = 1; t < threads; ++t) pool.emplace_back(work);
                work();
                for (auto &t : pool) t.join();
                if (err) std::rethrow_exception(err);
            }
            if (write_skq_flag) {
                log.info("Writing bins for use with precluster as " + *output + ".skq");
                write_skq(*output + ".skq", sketches);
            }
            log.info("Inverting sketch order");
            Inverted inv = Inverted::from_sketches(sketches, names, kmer, !single_strand);
            inv.labels = labels;
            if (metadata_file) {   // parse_metadata_info, src/io.rs:118-137; lib.rs:557-568
                std::ifstream mf(*metadata_file);
                if (!mf) throw std::runtime_error("Unable to open species name file " + *metadata_file);
                std::map<std::string, std::string> dict;
                std::string line;
                while (std::getline(mf, line)) {
                    const size_t tab = line.find('\t');
                    const std::string key = line.substr(0, tab);
                    std::string val = tab == std::string::npos ? "" : line.substr(tab + 1);
                    const size_t tab2 = val.find('\t');
                    if (tab2 != std::string::npos) val.resize(tab2);
                    if (!dict.emplace(key, val).second) throw std::runtime_error("Some entry in metadata is duplicated");
                }
                log.info("Got metadata for " + std::to_string(dict.size()) + " labels");
                std::vector<std::string> md(inputs.size());
                for (size_t i = 0; i < inputs.size(); ++i) {
                    const auto it = dict.find(inputs[i].first);
                    if (it == dict.end()) throw std::runtime_error("no metadata for sample " + inputs[i].first);
                    md[order[i]] = it->second;
                }
                inv.metadata = md;
            }
            inv.save(*output);
        } catch (const std::exception &e) {
            throw Panic(e.what());
        }
        return 0;
    }
    if (sub != "precluster") {
        std::cerr << "error: unrecognized subcommand 'inverted " << sub << "' (this build provides `inverted build` and `inverted precluster`)\n";
        return 2;
    }
    g_usage = "sketchlib inverted precluster [OPTIONS] <SKI> <--skd <SKD>|--count>";
    std::optional<std::string> ski, skd, output, completeness_file, retain;
    bool count = false, ani = false, host_candidates = false;
    int knn_ties = SKL_KNN_TIES_REFERENCE;
    size_t knn = 50, threads = 1;   // DEFAULT_KNN, cli.rs
    double cutoff = 0.64;
    int device = 0;
    for (int i = first + 1; i < argc; ++i) {
        const std::string arg = argv[i];
        if (arg == "-v" || arg == "--verbose") verbose = true;
        else if (arg == "--quiet") quiet = true;
        else if (arg == "--skd") skd = next_value(i, "--skd <SKD>");
        else if (arg == "-o") output = next_value(i, "-o <OUTPUT>");
        else if (arg == "--count") count = true;
        else if (arg == "--knn") knn = parse_usize("--knn <KNN>", next_value(i, arg));
        else if (arg == "--ani") ani = true;
        else if (arg == "--threads") threads = std::max<size_t>(1, parse_usize("--threads <THREADS>", next_value(i, arg)));
        else if (arg == "--ref-completeness-file") completeness_file = next_value(i, arg);
        else if (arg == "--completeness-cutoff") cutoff = std::strtod(next_value(i, arg).c_str(), nullptr);
        else if (arg == "--retain-unmatched") retain = next_value(i, arg);
        else if (arg == "--device") device = (int)parse_usize("--device <D>", next_value(i, arg));
        else if (arg == "--host-candidates") host_candidates = true;   // candidate lists from the .ski on host threads
        else if (arg == "--knn-ties") {   // as `dist --knn-ties`: reference (default) | canonical
            const std::string v = next_value(i, "--knn-ties <RULE>");
            if (v == "canonical") knn_ties = SKL_KNN_TIES_CANONICAL;
            else if (v == "reference") knn_ties = SKL_KNN_TIES_REFERENCE;
            else usage_error("invalid value '" + v + "' for '--knn-ties <RULE>': possible values: canonical, reference");
        }
        else if (arg.size() > 1 && arg[0] == '-') usage_error("unexpected argument '" + arg + "' found");
        else if (!ski) ski = arg;
        else usage_error("unexpected argument '" + arg + "' found");
    }
    if (!ski) usage_error("the following required arguments were not provided:\n  <SKI>");
    if (count && skd) usage_error("the argument '--skd <SKD>' cannot be used with '--count'");
    using distances::RetainUnmatched;
    RetainUnmatched retain_mode = RetainUnmatched::None;
    if (retain) {
        if (*retain == "singleton") retain_mode = RetainUnmatched::Singleton;
        else if (*retain == "bruteforce") retain_mode = RetainUnmatched::Bruteforce;
        else usage_error("invalid value '" + *retain + "' for '--retain-unmatched <RETAIN_UNMATCHED>'\n  [possible values: singleton, bruteforce]");
    }
    const Logger log{verbose && !quiet, !quiet};
    log.info("Using " + std::to_string(threads) + " threads");
    const bool timing = std::getenv("SKL_CLI_TIMING") != nullptr;
    const auto t_start = std::chrono::steady_clock::now();
    auto since_start = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count(); };
    const std::string input_prefix = strip_sketch_extension(*ski);
    // (as in `dist`: the device context comes up while the .ski, the .skq and the .skd are read)
    std::future<std::unique_ptr<Device>> dev_starting;
    if (skd && !count) dev_starting = std::async(std::launch::async, [device] { return std::make_unique<Device>(device); });
    // The bitmaps of the index are only needed for --count and for the host candidate search;
    // by default the candidates are found on the device from the .skq alone.
    Inverted inv = Inverted::load(input_prefix, count || host_candidates);   // `?` in the reference: Error, exit 1
    if (!inv.has_index() && inv.sample_names.size() > skl_shared_bins_max_samples()) {
        log.info("More samples than the on-device candidate search takes: using the index on the host");
        inv = Inverted::load(input_prefix, true);
    }
    const double t_ski = since_start();
    if (count) {
        const size_t ns = inv.sample_names.size();
        std::cout << "Identified " << inv.any_shared_bin_pairs(threads) << " prefilter pairs from a max of "
                  << ns * (ns - 1) / 2 << "\n";
        return 0;
    }
    if (!skd) return 0;   // neither mode: the reference does nothing (lib.rs:713)
    std::ostream *os = &std::cout;
    std::unique_ptr<TextSink> sink;
    if (output) {
        try {
            sink = std::make_unique<FileSink>(*output);
        } catch (const std::exception &e) {
            throw Panic(e.what());
        }
    } else {
        sink = std::make_unique<StreamSink>(*os);
    }
    const std::string skq_filename = input_prefix + ".skq";
    log.info("Loading queries from " + skq_filename);
    std::vector<uint16_t> skq_bins;
    try {
        skq_bins = read_skq(skq_filename, inv.sample_names.size(), inv.sketch_size());
    } catch (const std::exception &e) {
        throw Panic(e.what());
    }
    const std::string ref_db_name = strip_sketch_extension(*skd);
    MultiSketch references;
    try {
        references = MultiSketch::load_metadata(ref_db_name);
    } catch (const std::exception &) {
        throw Panic("Could not read sketch metadata from " + ref_db_name + ".skm");
    }
    if (references.kmer_lengths().size() > 1 && references.get_k_idx(inv.kmer_size)) {
        references.select_kmer(*references.get_k_idx(inv.kmer_size));   // (as in `dist -k`: the one slice the distances read)
    }
    log.info("Loading sketch data from " + ref_db_name + ".skd");
    references.read_sketch_data(ref_db_name);
    const size_t n = references.number_samples_loaded();
    if (knn_ties == SKL_KNN_TIES_REFERENCE && !inv.has_index() && !distances::ski_order_is_skd_order(references, inv)) {
        // the reference pushes a row's candidates in ascending .ski index (mod.rs:459-487); lists built on the device are
        // ascending in .skd order, so with a .ski that orders the samples differently they come from the index, on the host
        log.info("The .ski orders the samples differently from the .skd: candidate lists from the index on the host "
                 "(--knn-ties canonical keeps them on the device)");
        inv = Inverted::load(input_prefix, true);
    }
    if (knn >= n) {   // lib.rs:737-740
        log.warn("knn=" + std::to_string(knn) + " is higher than number of samples=" + std::to_string(n));
        knn = n - 1;
    }
    if (knn == 0) throw Panic("chunk size must be non-zero");
    DistType dist_type;
    try {
        dist_type = distances::set_k(references, inv.kmer_size, ani);
    } catch (const std::exception &e) {
        throw Panic("K-mer size " + std::to_string(inv.kmer_size) + " used for .ski not found in .skd: " + e.what());
    }
    std::optional<std::vector<double>> comp;
    if (completeness_file) {
        std::vector<std::string> warnings;
        comp = read_completeness_file(*completeness_file, references, &warnings);
        for (const auto &w : warnings) log.warn(w);
    }
    log.info("Calculating sparse ref vs ref distances with " + std::to_string(knn) + " nearest neighbours");
    log.info("Preclustering with k=" + std::to_string(inv.kmer_size) + " and s=" + std::to_string(inv.sketch_size()));
    if (retain) log.info("Retain unmatched mode: " + *retain);
    const double t_loaded = since_start();
    const std::unique_ptr<Device> dev_owner = dev_starting.get();
    Device &dev = *dev_owner;
    const double t_device = since_start();
    if (comp && (skl_ctx_flags(dev.ctx()) & SKL_CTX_FLAG_LOG_UNMATCHED)) log.warn(LOG_UNMATCHED_WARNING);
    const SparseDistanceMatrix d = distances::self_dists_knn_precluster(
        dev, references, inv, skq_bins, inv.sketch_size(), n, knn, dist_type, comp ? &*comp : nullptr, cutoff,
        retain_mode, threads, knn_ties);
    const double t_dist = since_start();
    log.info("Writing out in sparse matrix form");
    d.write(*sink, threads);
    os->flush();
    if (timing) {
        std::fprintf(stderr, "TIMING precluster: load_ski=%.3fs load_skq+skd=%.3fs device_wait=%.3fs candidates+distances=%.3fs write=%.3fs\n",
                     t_ski, t_loaded - t_ski, t_device - t_loaded, t_dist - t_device, since_start() - t_dist);
    }
    g_listing_complete = true;
    return 0;
}

}  // namespace

int main(int argc, char **argv)
{
    const auto start = std::chrono::steady_clock::now();
    // global flags may precede the subcommand
    int sub = 1;
    while (sub < argc && (strcmp(argv[sub], "-v") == 0 || strcmp(argv[sub], "--verbose") == 0 ||
                          strcmp(argv[sub], "--quiet") == 0)) {
        ++sub;
    }
    if (sub >= argc || strcmp(argv[sub], "-h") == 0 || strcmp(argv[sub], "--help") == 0) {
        std::cout << "Usage: sketchlib [OPTIONS] <COMMAND>\n\nCommands:\n"
                     "  sketch  Create sketches from input data (DNA assemblies, CPU)\n"
                     "  dist    Calculate pairwise distances using sketches (GPU)\n"
                     "  inverted build|precluster  Inverted index of single-k sketches; kNN restricted to its candidates (GPU)\n";
        return sub >= argc ? 2 : 0;
    }
    if (strcmp(argv[sub], "sketch") == 0) {
        bool verbose = false, quiet = false;
        for (int i = 1; i < sub; ++i) {
            if (strcmp(argv[i], "--quiet") == 0) quiet = true;
            else verbose = true;
        }
        try {
            return run_sketch(argc, argv, sub + 1, verbose, quiet);
        } catch (const Panic &p) {
            std::cerr << "thread 'main' panicked:\n" << p.what() << "\n";
            return 101;
        } catch (const std::exception &e) {
            std::cerr << "Error: " << e.what() << "\n";
            return 1;
        }
    }
    if (strcmp(argv[sub], "inverted") == 0) {
        bool verbose = false, quiet = false;
        for (int i = 1; i < sub; ++i) {
            if (strcmp(argv[i], "--quiet") == 0) quiet = true;
            else verbose = true;
        }
        try {
            const int rc = run_inverted(argc, argv, sub + 1, verbose, quiet);
            if (verbose && !quiet) {
                const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
                std::cerr << "INFO  [sketchlib] Complete in " << s << "s\n";  // lib.rs:949-957
            }
            return leave_after_success(rc);
        } catch (const Panic &p) {
            std::cerr << "thread 'main' panicked:\n" << p.what() << "\n";
            return 101;
        } catch (const std::exception &e) {
            std::cerr << "Error: " << e.what() << "\n";
            return 1;
        }
    }
    if (strcmp(argv[sub], "dist") != 0) {
        std::cerr << "error: unrecognized subcommand '" << argv[sub]
                  << "' (this build provides `sketch` (DNA assemblies, CPU), `dist` (GPU) and `inverted build|precluster`)\n";
        return 2;
    }
    DistArgs args = parse_dist(argc, argv, sub + 1);
    for (int i = 1; i < sub; ++i) {
        if (strcmp(argv[i], "--quiet") == 0) args.quiet = true;
        else args.verbose = true;
    }
    try {
        const int rc = run_dist(args);
        if (args.verbose && !args.quiet) {
            const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
            std::cerr << "INFO  [sketchlib] Complete in " << s << "s\n";  // lib.rs:949-957
        }
        return leave_after_success(rc);
    } catch (const Panic &p) {
        std::cerr << "thread 'main' panicked:\n" << p.what() << "\n";
        return 101;
    } catch (const std::exception &e) {
        std::cerr << "Error: " << e.what() << "\n";
        return 1;
    }
}
