#include "io.hpp"

#include <charconv>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <stdexcept>

namespace skl_host {

std::string strip_sketch_extension(const std::string &f)
{
    auto ends = [&](const char *e) { return f.size() >= 4 && f.compare(f.size() - 4, 4, e) == 0; };
    if (ends(".skm") || ends(".skd") || ends(".ski")) return f.substr(0, f.size() - 4);
    return f;
}

std::vector<std::string> read_subset_names(const std::string &subset_file)
{
    std::ifstream f(subset_file);
    if (!f) throw std::runtime_error("Unable to open " + subset_file);
    std::vector<std::string> names;
    std::string line;
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        names.push_back(line);
    }
    return names;
}

static std::string trim(const std::string &s)
{
    size_t a = 0, b = s.size();
    while (a < b && isspace((unsigned char)s[a])) ++a;
    while (b > a && isspace((unsigned char)s[b - 1])) --b;
    return s.substr(a, b - a);
}

static std::string fmt_f64(double v)
{
    char buf[64];
    const auto r = std::to_chars(buf, buf + sizeof buf, v);
    return std::string(buf, r.ptr);
}

std::vector<double> read_completeness_file(const std::string &completeness_file,
                                           const MultiSketch &sketches,
                                           std::vector<std::string> *warnings)
{
    const size_t n = sketches.number_samples_loaded();
    std::vector<double> completeness(n, 1.0);
    std::ifstream f(completeness_file);
    if (!f) throw std::runtime_error("Failed to open completeness file: " + completeness_file);
    std::vector<std::string> not_in_sketch, out_of_range;
    std::vector<bool> matched(n, false);
    std::string line;
    while (std::getline(f, line)) {
        const size_t tab = line.find('\t');
        if (tab == std::string::npos) continue;  // split_once('\t')? -> skipped
        const std::string genome_id = line.substr(0, tab);
        const std::string value_str = trim(line.substr(tab + 1));
        char *end = nullptr;
        const double value = std::strtod(value_str.c_str(), &end);
        if (value_str.empty() || end != value_str.c_str() + value_str.size()) {
            if (warnings) warnings->push_back("Could not parse completeness value for '" + genome_id + "': '" +
                                              line.substr(tab + 1) + "' — skipping");
            continue;
        }
        if (!(value >= 0.0 && value <= 1.0)) {
            out_of_range.push_back(genome_id + ": " + fmt_f64(value));
            continue;
        }
        if (auto idx = sketches.get_sample_index(genome_id)) {
            completeness[*idx] = value;
            matched[*idx] = true;
        } else {
            not_in_sketch.push_back(genome_id);
        }
    }
    if (!out_of_range.empty()) {
        std::ostringstream msg;
        msg << "Completeness values must be in [0.0, 1.0], not percentages. Found " << out_of_range.size()
            << " out-of-range value(s) in " << completeness_file << ":";
        for (const auto &s : out_of_range) msg << "\n  " << s;
        throw std::runtime_error(msg.str());
    }
    if (warnings) {
        if (!not_in_sketch.empty()) {
            std::string w = std::to_string(not_in_sketch.size()) +
                            " genome(s) in completeness file not found in sketch database (ignored): ";
            for (size_t i = 0; i < not_in_sketch.size(); ++i) w += (i ? ", " : "") + not_in_sketch[i];
            warnings->push_back(w);
        }
        std::string missing;
        size_t n_missing = 0;
        for (size_t i = 0; i < n; ++i) {
            if (!matched[i]) {
                missing += (n_missing ? ", " : "") + sketches.sketch_name(i);
                ++n_missing;
            }
        }
        if (n_missing) {
            warnings->push_back(std::to_string(n_missing) +
                                " genome(s) not found in completeness file, using default 1.0: " + missing);
        }
    }
    return completeness;
}

}  // namespace skl_host
