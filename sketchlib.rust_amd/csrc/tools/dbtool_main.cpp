// skl_dbtool -- small database utility around the C++ MultiSketch mirror (CPU only):
//   skl_dbtool info <prefix>            dump the .skm fields (one `key<TAB>value` per line)
//   skl_dbtool roundtrip <in> <out>     load <in>.skm/.skd and write them back as <out>.*
//   skl_dbtool unframe <in> <out>       decode a snappy-framed file (.skm / .ski) to its payload
//   skl_dbtool slice <prefix> <i> <k>   print the u64 words of get_sketch_slice(i, k_idx)
//   skl_dbtool slice-selected <prefix> <sample> <k_idx> [name...]   a slice read through select_kmer
//   skl_dbtool make <prefix> <bins> <k1,k2,..> <name>...  write <prefix>.skm for an existing .skd
//   skl_dbtool format <self|cross> <coreacc|jaccard> <n> <nq> <threads> <band_rows> <dists.f32> [out]
//                                       print a raw f32 distance array as the dense long-form text
//                                       (names s0,s1,.. / q0,q1,..), in row bands of <band_rows> rows,
//                                       to stdout or (block-parallel) to <out>
//   skl_dbtool make-ski <prefix> <k> <sketch_size> <name>...   write <prefix>.ski for an existing
//                                       <prefix>.skq ([sample][sketch_size] u16) whose rows are in
//                                       the order of the names
// Used by the test-suite to pin the file-format code without a GPU, and to write
// synthetic databases in the reference's on-disk layout.
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include <fstream>

#include "../host/distance_matrix.hpp"
#include "../host/inverted.hpp"
#include "../host/io.hpp"
#include "../host/multisketch.hpp"
#include "../host/snappy_frame.hpp"

using namespace skl_host;

// names given as arguments, or `@file` = one name per line (command lines have a length limit)
static std::vector<std::string> collect_names(int argc, char **argv, int first)
{
    std::vector<std::string> names;
    for (int i = first; i < argc; ++i) {
        if (argv[i][0] == '@') {
            std::ifstream f(argv[i] + 1);
            std::string line;
            while (std::getline(f, line)) {
                if (!line.empty()) names.push_back(line);
            }
        } else {
            names.push_back(argv[i]);
        }
    }
    return names;
}

int main(int argc, char **argv)
{
    try {
        if (argc >= 3 && std::string(argv[1]) == "info") {
            const std::string prefix = strip_sketch_extension(argv[2]);
            MultiSketch m = MultiSketch::load_metadata(prefix);
            std::cout << "sketch_size\t" << m.sketch_size << "\n";
            std::cout << "sketchsize64\t" << m.sketchsize64 << "\n";
            std::cout << "kmer_lengths\t";
            for (size_t i = 0; i < m.kmer_lengths().size(); ++i) std::cout << (i ? "," : "") << m.kmer_lengths()[i];
            std::cout << "\nn_samples\t" << m.number_samples_loaded() << "\n";
            std::cout << "kmer_stride\t" << m.kmer_stride() << "\nsample_stride\t" << m.sample_stride() << "\n";
            std::cout << "sketch_version\t" << m.version() << "\nhash_type\t" << m.hash_type() << "\n";
            for (size_t i = 0; i < m.metadata().size(); ++i) {
                const auto &s = m.metadata()[i];
                std::cout << "sample\t" << i << "\t" << s.name << "\t" << (s.index ? (long long)*s.index : -1)
                          << "\t" << s.seq_length << "\t" << s.rc << s.reads << s.densified << "\t"
                          << s.acgt[0] << "," << s.acgt[1] << "," << s.acgt[2] << "," << s.acgt[3] << "\t"
                          << s.non_acgt << "\n";
            }
            return 0;
        }
        if (argc >= 4 && std::string(argv[1]) == "unframe") {   // snappy-framed file -> its payload
            std::ifstream f(argv[2], std::ios::binary);
            if (!f) throw std::runtime_error(std::string("cannot open ") + argv[2]);
            f.seekg(0, std::ios::end);
            std::vector<uint8_t> framed((size_t)f.tellg());
            f.seekg(0);
            f.read((char *)framed.data(), (std::streamsize)framed.size());
            const std::vector<uint8_t> raw = snappy_frame_decode(framed);
            std::ofstream o(argv[3], std::ios::binary);
            o.write((const char *)raw.data(), (std::streamsize)raw.size());
            return 0;
        }
        if (argc >= 4 && std::string(argv[1]) == "roundtrip") {
            const std::string in = strip_sketch_extension(argv[2]), out = strip_sketch_extension(argv[3]);
            MultiSketch m = MultiSketch::load_metadata(in);
            m.read_sketch_data(in);
            m.save_metadata(out);
            MultiSketch::write_sketch_data(out, m.bins().data(), m.bins().size());
            return 0;
        }
        if (argc >= 5 && std::string(argv[1]) == "slice") {
            const std::string prefix = strip_sketch_extension(argv[2]);
            MultiSketch m = MultiSketch::load_metadata(prefix);
            m.read_sketch_data(prefix);
            const uint64_t *p = m.get_sketch_slice((size_t)atoll(argv[3]), (size_t)atoll(argv[4]));
            for (size_t w = 0; w < m.kmer_stride(); ++w) std::cout << p[w] << "\n";
            return 0;
        }
        if (argc >= 5 && std::string(argv[1]) == "slice-selected") {
            // slice-selected <prefix> <logical sample> <k_idx> [name...]: the same slice through MultiSketch::select_kmer
            // (only that k-mer length is read from the file); with names, through read_sketch_data_block of those samples
            const std::string prefix = strip_sketch_extension(argv[2]);
            MultiSketch m = MultiSketch::load_metadata(prefix);
            m.select_kmer((size_t)atoll(argv[4]));
            std::vector<std::string> names = collect_names(argc, argv, 5);
            if (!names.empty() && names[0] == "--no-mmap") {   // the positional-read path of a file that cannot be mapped
                MultiSketch::testing_read_slices_without_mapping(true);
                names.erase(names.begin());
            }
            if (names.empty()) m.read_sketch_data(prefix);
            else m.read_sketch_data_block(prefix, names);
            if (m.kmer_lengths().size() != 1 || m.bins().size() != m.number_samples_loaded() * m.kmer_stride()) return 3;
            const uint64_t *p = m.get_sketch_slice((size_t)atoll(argv[3]), 0);
            for (size_t w = 0; w < m.kmer_stride(); ++w) std::cout << p[w] << "\n";
            return 0;
        }
        if (argc >= 9 && std::string(argv[1]) == "format") {
            const bool cross = std::string(argv[2]) == "cross";
            DistanceMatrix m;
            m.jaccard.kind = std::string(argv[3]) == "coreacc" ? DistType::CoreAcc : DistType::Jaccard;
            const size_t n = strtoull(argv[4], nullptr, 10), nq = strtoull(argv[5], nullptr, 10);
            const size_t threads = strtoull(argv[6], nullptr, 10), band_rows = strtoull(argv[7], nullptr, 10);
            for (size_t i = 0; i < n; ++i) m.ref_names.push_back("s" + std::to_string(i));
            if (cross) {
                m.query_names.emplace();
                for (size_t i = 0; i < nq; ++i) m.query_names->push_back("q" + std::to_string(i));
            }
            std::ifstream f(argv[8], std::ios::binary);
            f.seekg(0, std::ios::end);
            const size_t bytes = (size_t)f.tellg();
            f.seekg(0);
            m.distances.resize(bytes / sizeof(float));
            f.read(reinterpret_cast<char *>(m.distances.data()), (std::streamsize)bytes);
            const size_t ncols = m.jaccard.n_dist_cols();
            m.n_distances = m.distances.size() / ncols;
            std::unique_ptr<TextSink> sink;
            if (argc >= 10) sink = std::make_unique<FileSink>(argv[9]);
            else sink = std::make_unique<StreamSink>(std::cout);
            for (size_t r0 = 0; r0 < n; r0 += band_rows) {
                const size_t first = cross ? r0 * nq : (r0 + 1 < n ? square_to_condensed(r0, r0 + 1, n) : 0);
                m.write_rows(*sink, r0, std::min(n, r0 + band_rows), m.distances.data() + first * ncols, threads);
            }
            std::cout.flush();
            return 0;
        }
        if (argc >= 4 && std::string(argv[1]) == "reski") {   // <in prefix> <out prefix>: load a .ski and write it again
            Inverted::load(strip_sketch_extension(argv[2])).save(argv[3]);
            return 0;
        }
        if (argc >= 6 && std::string(argv[1]) == "make-ski") {
            const std::string prefix = argv[2];
            const size_t k = strtoull(argv[3], nullptr, 10), sketch_size = strtoull(argv[4], nullptr, 10);
            const std::vector<std::string> names = collect_names(argc, argv, 5);
            const std::vector<uint16_t> flat = read_skq(prefix + ".skq", names.size(), sketch_size);
            std::vector<std::vector<uint16_t>> sketches(names.size());
            for (size_t i = 0; i < names.size(); ++i) {
                sketches[i].assign(flat.begin() + i * sketch_size, flat.begin() + (i + 1) * sketch_size);
            }
            Inverted::from_sketches(sketches, names, k, true).save(prefix);
            return 0;
        }
        if (argc >= 6 && std::string(argv[1]) == "make") {
            // make <prefix> <sketch_size_bins> <k1,k2,...> <name>...: write <prefix>.skm for an
            // existing <prefix>.skd whose sample blocks are in the order of the names
            const std::string prefix = strip_sketch_extension(argv[2]);
            const uint64_t bins = strtoull(argv[3], nullptr, 10);
            std::vector<size_t> kmers;
            std::string ks = argv[4];
            size_t pos = 0;
            while (pos <= ks.size()) {
                const size_t comma = ks.find(',', pos);
                const std::string tok = ks.substr(pos, comma == std::string::npos ? std::string::npos : comma - pos);
                if (!tok.empty()) kmers.push_back((size_t)strtoull(tok.c_str(), nullptr, 10));
                if (comma == std::string::npos) break;
                pos = comma + 1;
            }
            std::vector<SketchMeta> meta;
            const std::vector<std::string> names = collect_names(argc, argv, 5);
            for (size_t i = 0; i < names.size(); ++i) {
                SketchMeta sm;
                sm.name = names[i];
                sm.index = (uint64_t)i;
                meta.push_back(sm);
            }
            MultiSketch m(std::move(meta), bins, kmers);
            m.save_metadata(prefix);
            return 0;
        }
        std::cerr << "usage: skl_dbtool info|roundtrip|slice|make ...\n";
        return 2;
    } catch (const std::exception &e) {
        std::cerr << "Error: " << e.what() << "\n";
        return 1;
    }
}
