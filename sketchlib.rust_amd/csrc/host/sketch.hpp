// sketch.hpp -- CPU assembly sketcher: gz/plain FASTA -> canonical ntHash -> bin minima ->
// densify -> 14-plane transpose, i.e. the producer of the `.skm/.skd` files the distance
// path consumes.  Mirrors the reference's `sketch` command for DNA
// assemblies (src/sketch/mod.rs:74-258,283-391, src/hashing/nthash_iterator.rs); reads
// (FASTQ + k-mer count filter), amino-acid and structure alphabets are not part of this
// build.
#pragma once

#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include "multisketch.hpp"

namespace skl_host {

constexpr uint64_t SIGN_MOD = (1ull << 61) - 1;  // src/sketch/mod.rs:36

// (sample name, sequence files), one entry per sample (src/io.rs:20-40, rfile parsing)
using InputFastx = std::pair<std::string, std::vector<std::string>>;

std::vector<InputFastx> read_input_fastas(const std::vector<std::string> &seq_files);  // io.rs:20-40
std::vector<InputFastx> read_rfile(const std::string &file_list);                      // name<TAB>file[<TAB>file]
std::vector<size_t> parse_kmers(const std::vector<size_t> &k_vals, const std::vector<size_t> &k_seq);  // io.rs:140-159

// NtHashIterator::add_dna_seq (nthash_iterator.rs:205-251): valid bases as 2-bit codes, plus
// the valid-base coordinates of every N and record end.
struct Sequence {
    std::vector<uint8_t> codes;
    std::vector<size_t> offsets;
    uint64_t acgt[4] = {0, 0, 0, 0};
    uint64_t non_acgt = 0;
};
void add_fasta(const std::string &path, Sequence &s);
bool densify_bin(std::vector<uint64_t> &signs);                     // sketch/mod.rs:237-258
void fill_usigs(uint64_t *usigs, const std::vector<uint64_t> &signs);  // sketch/mod.rs:215-223

struct SketchResult {
    SketchMeta meta;
    std::vector<uint64_t> usigs;  // [k][chunk][plane]
};

// Sketch::new (src/sketch/mod.rs:74-129) for one sample.
SketchResult sketch_sample(const InputFastx &input, const std::vector<size_t> &kmers, uint64_t sketch_size,
                           bool rc);

// sketch_files (src/sketch/mod.rs:283-391): writes <output_prefix>.skd and .skm; samples
// keep their input order (what the reference yields with --threads 1).
MultiSketch sketch_files(const std::string &output_prefix, const std::vector<InputFastx> &inputs,
                         const std::vector<size_t> &kmers, uint64_t sketch_size, bool rc, size_t threads);

class Device;
// The same with the hashing / bin-minimum loop on the GPU (SURVEY 8f row f4,
// skl_sketch_signs): FASTA parsing, densification, transpose and the file writers stay on the
// host.  Output files are byte-identical to sketch_files'.
MultiSketch sketch_files_gpu(Device &dev, const std::string &output_prefix, const std::vector<InputFastx> &inputs,
                             const std::vector<size_t> &kmers, uint64_t sketch_size, bool rc, size_t threads);

}  // namespace skl_host
