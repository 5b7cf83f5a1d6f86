#include "inverted.hpp"

#include <algorithm>
#include <atomic>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <thread>

#include "cbor.hpp"
#include "msgpack.hpp"
#include "msgpack.hpp"
#include "snappy_frame.hpp"

namespace skl_host {

// ---------------------------------------------------------------------------
// Roaring portable format
// ---------------------------------------------------------------------------

namespace {
constexpr uint32_t SERIAL_COOKIE_NO_RUN = 12346, SERIAL_COOKIE = 12347, NO_OFFSET_THRESHOLD = 4;

void put16(std::string &o, uint16_t v) { o.push_back((char)(v & 0xFF)); o.push_back((char)(v >> 8)); }
void put32(std::string &o, uint32_t v) { for (int s = 0; s < 32; s += 8) o.push_back((char)((v >> s) & 0xFF)); }

struct Reader {
    const std::string &b;
    size_t pos = 0;
    void need(size_t n) const { if (pos + n > b.size()) throw std::runtime_error("roaring: truncated bitmap"); }
    uint16_t u16() { need(2); const uint16_t v = (uint8_t)b[pos] | ((uint16_t)(uint8_t)b[pos + 1] << 8); pos += 2; return v; }
    uint32_t u32() { const uint32_t lo = u16(); return lo | ((uint32_t)u16() << 16); }
};
}  // namespace

std::string roaring_serialize(const std::vector<uint32_t> &v)
{
    // containers: one per distinct high half, in ascending order
    std::vector<std::pair<uint16_t, std::pair<size_t, size_t>>> cont;   // key, [begin, end) in v
    for (size_t i = 0; i < v.size();) {
        const uint16_t key = (uint16_t)(v[i] >> 16);
        size_t j = i;
        while (j < v.size() && (uint16_t)(v[j] >> 16) == key) ++j;
        cont.push_back({key, {i, j}});
        i = j;
    }
    std::string o;
    put32(o, SERIAL_COOKIE_NO_RUN);
    put32(o, (uint32_t)cont.size());
    for (const auto &c : cont) {
        put16(o, c.first);
        put16(o, (uint16_t)(c.second.second - c.second.first - 1));
    }
    uint32_t offset = (uint32_t)(8 + 8 * cont.size());
    for (const auto &c : cont) {
        put32(o, offset);
        const size_t card = c.second.second - c.second.first;
        offset += card > 4096 ? 8192u : (uint32_t)(2 * card);
    }
    for (const auto &c : cont) {
        const size_t card = c.second.second - c.second.first;
        if (card > 4096) {
            std::string bits(8192, '\0');
            for (size_t i = c.second.first; i < c.second.second; ++i) {
                const uint16_t low = (uint16_t)(v[i] & 0xFFFF);
                bits[low >> 3] = (char)((uint8_t)bits[low >> 3] | (1u << (low & 7)));
            }
            o += bits;
        } else {
            for (size_t i = c.second.first; i < c.second.second; ++i) put16(o, (uint16_t)(v[i] & 0xFFFF));
        }
    }
    return o;
}

std::vector<uint32_t> roaring_deserialize(const std::string &bytes)
{
    Reader r{bytes};
    const uint32_t cookie = r.u32();
    size_t size;
    std::string run_flags;
    bool has_runs = false;
    if ((cookie & 0xFFFF) == SERIAL_COOKIE) {
        has_runs = true;
        size = (cookie >> 16) + 1;
        const size_t nb = (size + 7) / 8;
        r.need(nb);
        run_flags = bytes.substr(r.pos, nb);
        r.pos += nb;
    } else if (cookie == SERIAL_COOKIE_NO_RUN) {
        size = r.u32();
    } else {
        throw std::runtime_error("roaring: unknown cookie");
    }
    if (size > 65536) throw std::runtime_error("roaring: more containers than 16-bit keys");
    std::vector<std::pair<uint16_t, uint32_t>> desc(size);   // key, cardinality
    for (auto &d : desc) {
        d.first = r.u16();
        d.second = (uint32_t)r.u16() + 1;
    }
    if (!has_runs || size >= NO_OFFSET_THRESHOLD) {
        r.need(4 * size);
        r.pos += 4 * size;   // containers follow in order: the offsets are redundant
    }
    std::vector<uint32_t> out;
    for (size_t c = 0; c < size; ++c) {
        const uint32_t hi = (uint32_t)desc[c].first << 16;
        const bool is_run = has_runs && (((uint8_t)run_flags[c >> 3] >> (c & 7)) & 1);
        if (is_run) {
            const uint16_t n_runs = r.u16();
            for (uint16_t k = 0; k < n_runs; ++k) {
                const uint32_t start = r.u16(), len = r.u16();
                if (start + len > 0xFFFFu) throw std::runtime_error("roaring: run leaves its container");
                for (uint32_t x = start; x <= start + len; ++x) out.push_back(hi | x);
            }
        } else if (desc[c].second > 4096) {
            r.need(8192);
            for (uint32_t x = 0; x < 65536; ++x) {
                if (((uint8_t)bytes[r.pos + (x >> 3)] >> (x & 7)) & 1) out.push_back(hi | x);
            }
            r.pos += 8192;
        } else {
            for (uint32_t k = 0; k < desc[c].second; ++k) out.push_back(hi | r.u16());
        }
    }
    return out;
}

// ---------------------------------------------------------------------------
// Inverted
// ---------------------------------------------------------------------------

Inverted Inverted::from_sketches(const std::vector<std::vector<uint16_t>> &sketches,
                                 std::vector<std::string> names, size_t k, bool rc_)
{
    Inverted inv;
    inv.n_samples = names.size();
    inv.sample_names = std::move(names);
    inv.kmer_size = k;
    inv.rc = rc_;
    const size_t sketch_size = sketches.empty() ? 0 : sketches[0].size();
    inv.index.resize(sketch_size);
    // build_inverted_index, inverted.rs:467-499: genomes in ascending order, so the lists are sorted
    for (size_t g = 0; g < sketches.size(); ++g) {
        for (size_t b = 0; b < sketch_size; ++b) inv.index[b][sketches[g][b]].push_back((uint32_t)g);
    }
    return inv;
}

// `.ski` = snappy frame around the MessagePack form of struct Inverted that rmp-serde writes
// (inverted.rs:194-201): an ARRAY of the nine fields in declaration order (msgpack.hpp).
void Inverted::save(const std::string &file_prefix) const
{
    CborValue root = CborValue::array();
    CborValue idx = CborValue::array();
    for (const auto &bin : index) {
        CborValue m = CborValue::object();
        for (const auto &kv : bin) {
            CborValue bytes;
            bytes.kind = CborValue::BYTES;
            bytes.s = roaring_serialize(kv.second);
            m.map.emplace_back(CborValue::uint(kv.first), std::move(bytes));
        }
        idx.arr.push_back(std::move(m));
    }
    root.arr.push_back(std::move(idx));                                   // index
    root.arr.push_back(CborValue::uint(n_samples));                       // n_samples
    CborValue names = CborValue::array();
    for (const auto &n : sample_names) names.arr.push_back(CborValue::text(n));
    root.arr.push_back(std::move(names));                                 // sample_names
    auto opt_list = [](const std::optional<std::vector<std::string>> &v) {
        if (!v) return CborValue::null();
        CborValue a = CborValue::array();
        for (const auto &s : *v) a.arr.push_back(CborValue::text(s));
        return a;
    };
    root.arr.push_back(opt_list(metadata));
    root.arr.push_back(opt_list(labels));
    root.arr.push_back(CborValue::uint(kmer_size));
    root.arr.push_back(CborValue::text(sketch_version));
    root.arr.push_back(CborValue::boolean(rc));
    {   // HashType as rmp-serde writes it: a unit variant (DNA, PDB) is its name; the newtype variant AA(level),
        // held here as "AA(LevelN)", is the one-entry map {"AA": "LevelN"}
        const size_t open = hash_type.find('(');
        if (open == std::string::npos || hash_type.empty() || hash_type.back() != ')') {
            root.arr.push_back(CborValue::text(hash_type));
        } else {
            CborValue o = CborValue::object();
            o.put(hash_type.substr(0, open), CborValue::text(hash_type.substr(open + 1, hash_type.size() - open - 2)));
            root.arr.push_back(o);
        }
    }
    const std::vector<uint8_t> framed = snappy_frame_encode(msgpack_encode(root));
    std::ofstream f(file_prefix + ".ski", std::ios::binary);
    if (!f) throw std::runtime_error("Couldn't write to " + file_prefix + ".ski");
    f.write(reinterpret_cast<const char *>(framed.data()), (std::streamsize)framed.size());
}

// whole file in one read (an istreambuf_iterator copy is a byte at a time: 0.3 s for a 250 MB index)
static std::vector<uint8_t> slurp(const std::string &path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("Could not open " + path);
    f.seekg(0, std::ios::end);
    const std::streamoff n = f.tellg();
    f.seekg(0);
    std::vector<uint8_t> buf((size_t)std::max<std::streamoff>(n, 0));
    if (n > 0) f.read((char *)buf.data(), n);
    if (!f) throw std::runtime_error("Could not read " + path);
    return buf;
}

Inverted Inverted::load(const std::string &file_prefix, bool with_index)
{
    const std::string path = file_prefix + ".ski";
    const std::vector<uint8_t> framed = slurp(path);
    uint64_t n_bins = 0;
    const std::vector<uint8_t> doc = snappy_frame_decode(framed);
    if (doc.empty()) throw std::runtime_error(path + ": empty document");
    // rmp-serde writes the struct as a 9-element array (0x99).  Two other forms are read: the same
    // fields as a string-keyed MessagePack map (rmp-serde's `with_struct_map`, 0x89), and the
    // snappy-framed CBOR map that round 1 of this code base wrote (any other first byte).
    static const char *const FIELDS[9] = {"index", "n_samples", "sample_names", "metadata", "labels",
                                          "kmer_size", "sketch_version", "rc", "hash_type"};
    CborValue root;
    if (doc[0] == 0x99) {
        const CborValue arr = with_index ? msgpack_decode(doc) : msgpack_decode_array_skipping(doc, 0, &n_bins);
        root = CborValue::object();
        for (size_t f = 0; f < 9; ++f) root.put(FIELDS[f], arr.arr[f]);
    } else if (doc[0] == 0x89) {
        root = msgpack_decode(doc);
        if (!with_index && root.get("index")) n_bins = root.get("index")->arr.size();
    } else {
        root = with_index ? cbor_decode(doc) : cbor_decode_map_skipping(doc, "index", &n_bins);
    }
    if (root.kind != CborValue::MAP) throw std::runtime_error(path + ": not an inverted index");
    auto need = [&](const char *k) -> const CborValue & {
        const CborValue *v = root.get(k);
        if (!v) throw std::runtime_error(path + ": missing field " + k);
        return *v;
    };
    if (with_index && need("index").kind != CborValue::ARRAY) throw std::runtime_error(path + ": index is not an array");
    Inverted inv;
    inv.sketch_size_hint = (size_t)n_bins;
    if (with_index)
    for (const auto &bin : need("index").arr) {
        if (bin.kind != CborValue::MAP) throw std::runtime_error(path + ": index entry is not a map");
        inv.index.emplace_back();
        for (const auto &kv : bin.map) {
            if (kv.first.as_u64("bin value") > 0xFFFFu) throw std::runtime_error(path + ": bin value does not fit u16");
            // serde's serialize_bytes gives a bin / byte string; a generic Vec<u8> would be an array
            std::string bytes;
            if (kv.second.kind == CborValue::BYTES) {
                bytes = kv.second.s;
            } else if (kv.second.kind == CborValue::ARRAY) {
                for (const auto &x : kv.second.arr) bytes.push_back((char)x.as_u64("bitmap byte"));
            } else {
                throw std::runtime_error(path + ": unexpected bitmap encoding");
            }
            inv.index.back()[(uint16_t)kv.first.as_u64("bin value")] = roaring_deserialize(bytes);
        }
    }
    inv.n_samples = need("n_samples").as_u64("n_samples");
    for (const auto &n : need("sample_names").arr) {
        if (n.kind != CborValue::TEXT) throw std::runtime_error(path + ": sample name is not a string");
        inv.sample_names.push_back(n.s);
    }
    auto opt_list = [](const CborValue *v) -> std::optional<std::vector<std::string>> {
        if (!v || v->kind != CborValue::ARRAY) return std::nullopt;
        std::vector<std::string> out;
        for (const auto &s : v->arr) out.push_back(s.s);
        return out;
    };
    inv.metadata = opt_list(root.get("metadata"));
    inv.labels = opt_list(root.get("labels"));
    inv.kmer_size = need("kmer_size").as_u64("kmer_size");
    inv.sketch_version = need("sketch_version").s;
    if (need("rc").kind != CborValue::BOOL) throw std::runtime_error(path + ": rc is not a bool");
    inv.rc = need("rc").b;
    // HashType: "DNA" / "PDB" (unit variants), {"AA": "LevelN"} (newtype variant)
    const CborValue &ht = need("hash_type");
    if (ht.kind == CborValue::TEXT) {
        inv.hash_type = ht.s;
    } else if (ht.kind == CborValue::MAP && ht.map.size() == 1 && ht.map[0].first.kind == CborValue::TEXT) {
        inv.hash_type = ht.map[0].first.s + "(" + (ht.map[0].second.kind == CborValue::TEXT ? ht.map[0].second.s : "?") + ")";
    } else {
        throw std::runtime_error(path + ": unexpected hash_type encoding");
    }
    // a file is untrusted input: everything later indexes per-sample arrays with these ids
    if (inv.n_samples != inv.sample_names.size()) throw std::runtime_error(path + ": n_samples does not match sample_names");
    if (inv.metadata && inv.metadata->size() != inv.n_samples) throw std::runtime_error(path + ": metadata does not match n_samples");
    if (inv.labels && inv.labels->size() != inv.n_samples) throw std::runtime_error(path + ": labels do not match n_samples");
    for (const auto &bin : inv.index) {
        for (const auto &kv : bin) {
            uint64_t prev = 0;
            bool first = true;
            for (uint32_t id : kv.second) {
                if (id >= inv.n_samples || (!first && id <= prev)) throw std::runtime_error(path + ": bitmap holds an invalid sample id");
                prev = id;
                first = false;
            }
        }
    }
    return inv;
}

void Inverted::any_shared_bins(const uint16_t *query_sigs, std::vector<uint32_t> &stamp, uint32_t epoch,
                               std::vector<uint32_t> &out) const
{
    out.clear();
    for (size_t b = 0; b < index.size(); ++b) {
        const auto it = index[b].find(query_sigs[b]);
        if (it == index[b].end()) continue;
        for (uint32_t s : it->second) {
            if (stamp[s] != epoch) {
                stamp[s] = epoch;
                out.push_back(s);
            }
        }
    }
    std::sort(out.begin(), out.end());
}

std::vector<uint32_t> Inverted::any_shared_bins(const uint16_t *query_sigs) const
{
    std::vector<uint32_t> stamp(n_samples, 0), out;
    any_shared_bins(query_sigs, stamp, 1u, out);
    return out;
}

uint64_t Inverted::any_shared_bin_pairs(size_t threads) const
{
    // pair (i, j), i < j, counted once however many bins it shares: per sample i, mark the j > i
    // that appear with it in any bin list
    threads = std::max<size_t>(1, threads);
    std::vector<std::vector<std::pair<uint32_t, const std::vector<uint32_t> *>>> per_sample(n_samples);
    for (const auto &bin : index) {
        for (const auto &kv : bin) {
            if (kv.second.size() < 2) continue;
            for (uint32_t s : kv.second) per_sample[s].push_back({0, &kv.second});
        }
    }
    std::atomic<size_t> next{0};
    std::atomic<uint64_t> total{0};
    auto work = [&] {
        std::vector<uint8_t> hit(n_samples, 0);
        uint64_t mine = 0;
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= n_samples) break;
            std::vector<uint32_t> touched;
            for (const auto &pl : per_sample[i]) {
                for (uint32_t j : *pl.second) {
                    if (j > i && !hit[j]) {
                        hit[j] = 1;
                        touched.push_back(j);
                    }
                }
            }
            mine += touched.size();
            for (uint32_t j : touched) hit[j] = 0;
        }
        total += mine;
    };
    std::vector<std::thread> pool;
    for (size_t t = 1; t < threads; ++t) pool.emplace_back(work);
    work();
    for (auto &t : pool) t.join();
    return total.load();
}

// ---------------------------------------------------------------------------
// .skq
// ---------------------------------------------------------------------------

void write_skq(const std::string &path, const std::vector<std::vector<uint16_t>> &sketches)
{
    std::ofstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("Couldn't write to " + path);
    std::vector<char> row;
    for (const auto &s : sketches) {   // one write per sample (little-endian u16, inverted.rs:94-99)
        row.resize(s.size() * 2);
        for (size_t i = 0; i < s.size(); ++i) {
            row[2 * i] = (char)(s[i] & 0xFF);
            row[2 * i + 1] = (char)(s[i] >> 8);
        }
        f.write(row.data(), (std::streamsize)row.size());
    }
    if (!f) throw std::runtime_error("Couldn't write to " + path);
}

std::vector<uint16_t> read_skq(const std::string &path, size_t n_samples, size_t sketch_size)
{
    const std::vector<uint8_t> raw = slurp(path);
    if (raw.size() != n_samples * sketch_size * 2) {
        throw std::runtime_error(path + " does not hold " + std::to_string(n_samples) + " x " +
                                 std::to_string(sketch_size) + " bins");
    }
    std::vector<uint16_t> out(n_samples * sketch_size);
    for (size_t i = 0; i < out.size(); ++i) out[i] = (uint16_t)(raw[2 * i] | (raw[2 * i + 1] << 8));
    return out;
}

std::vector<size_t> reorder_by_labels(const std::vector<InputFastx> &inputs, const std::string &label_file,
                                      std::optional<std::vector<std::string>> *labels_out)
{
    std::ifstream f(label_file);
    if (!f) throw std::runtime_error("Unable to open species name file " + label_file);
    std::map<std::string, bool> input_names;
    for (const auto &in : inputs) input_names[in.first] = true;
    std::map<std::string, size_t> species_order;                  // label -> order of first appearance
    std::map<std::string, std::string> label_of;                  // sample -> label
    std::vector<std::pair<std::string, size_t>> label_order;      // (sample, label order)
    std::string line;
    while (std::getline(f, line)) {
        const size_t tab = line.find('\t');
        if (tab == std::string::npos) continue;
        const std::string name = line.substr(0, tab);
        std::string label = line.substr(tab + 1);
        const size_t tab2 = label.find('\t');
        if (tab2 != std::string::npos) label.resize(tab2);
        if (input_names.count(name)) {
            auto it = species_order.find(label);
            if (it == species_order.end()) it = species_order.emplace(label, species_order.size()).first;
            label_order.push_back({name, it->second});
        }
        label_of[name] = label;
    }
    std::stable_sort(label_order.begin(), label_order.end(),
                     [](const auto &a, const auto &b) { return a.second < b.second; });
    std::map<std::string, size_t> new_index;
    for (size_t i = 0; i < label_order.size(); ++i) new_index.emplace(label_order[i].first, i);
    std::vector<size_t> order(inputs.size());
    if (new_index.empty()) {      // "Could not find any sample names": identity, no labels
        for (size_t i = 0; i < inputs.size(); ++i) order[i] = i;
        if (labels_out) labels_out->reset();
        return order;
    }
    size_t next = new_index.size();
    for (size_t i = 0; i < inputs.size(); ++i) {
        const auto it = new_index.find(inputs[i].first);
        order[i] = it != new_index.end() ? it->second : next++;
    }
    if (labels_out) {
        std::vector<std::string> labels(inputs.size());
        for (size_t i = 0; i < inputs.size(); ++i) {
            const auto it = label_of.find(inputs[i].first);
            labels[order[i]] = it != label_of.end() ? it->second : "";
        }
        *labels_out = std::move(labels);
    }
    return order;
}

}  // namespace skl_host
