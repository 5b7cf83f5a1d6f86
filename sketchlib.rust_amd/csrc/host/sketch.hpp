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

}  // namespace skl_host
