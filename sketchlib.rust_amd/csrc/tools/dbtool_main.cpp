// skl_dbtool -- small database utility around the C++ MultiSketch mirror (CPU only):
//   skl_dbtool info <prefix>            dump the .skm fields (one `key<TAB>value` per line)
//   skl_dbtool roundtrip <in> <out>     load <in>.skm/.skd and write them back as <out>.*
//   skl_dbtool slice <prefix> <i> <k>   print the u64 words of get_sketch_slice(i, k_idx)
//   skl_dbtool make <prefix> <bins> <k1,k2,..> <name>...  write <prefix>.skm for an existing .skd
// Used by the test-suite to pin the file-format code without a GPU, and to write
// synthetic databases in the reference's on-disk layout.
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>
#include <vector>

#include "../host/io.hpp"
#include "../host/multisketch.hpp"

using namespace skl_host;

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
            for (int i = 5; i < argc; ++i) {
                SketchMeta sm;
                sm.name = argv[i];
                sm.index = (uint64_t)(i - 5);
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
