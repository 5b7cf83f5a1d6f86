// io.hpp -- the dist-side file helpers of the reference (src/io.rs:164-173,227-324;
// src/utils.rs:9-15).
#pragma once

#include <string>
#include <vector>

#include "multisketch.hpp"

namespace skl_host {

// utils.rs:9-15
std::string strip_sketch_extension(const std::string &file_name);
// io.rs:227-236 (one sample name per line)
std::vector<std::string> read_subset_names(const std::string &subset_file);
// io.rs:240-324.  Genomes not in the file default to 1.0; values outside [0, 1] are an
// error ("Completeness values must be in [0.0, 1.0], not percentages. ...").
// Warnings go to `warnings` (the reference logs them at warn level).
std::vector<double> read_completeness_file(const std::string &completeness_file,
                                           const MultiSketch &sketches,
                                           std::vector<std::string> *warnings);

}  // namespace skl_host
