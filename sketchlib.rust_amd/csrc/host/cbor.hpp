// cbor.hpp -- the subset of CBOR (RFC 8949) that serde/ciborium emits for the
// reference's MultiSketch struct (src/sketch/multisketch.rs:21-44): unsigned/negative
// ints, text and byte strings, arrays, maps (definite or indefinite), bool, null.
#pragma once

#include <cstdint>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace skl_host {

struct CborValue {
    enum Kind { UINT, NINT, BYTES, TEXT, ARRAY, MAP, BOOL, NIL, FLOAT } kind = NIL;
    uint64_t u = 0;   // UINT value, or (-1 - n) magnitude for NINT
    bool b = false;
    double f = 0.0;
    std::string s;    // TEXT / BYTES
    std::vector<CborValue> arr;
    std::vector<std::pair<CborValue, CborValue>> map;  // insertion order preserved

    const CborValue *get(const std::string &key) const
    {
        for (const auto &kv : map) {
            if (kv.first.kind == TEXT && kv.first.s == key) return &kv.second;
        }
        return nullptr;
    }
    uint64_t as_u64(const char *what) const
    {
        if (kind != UINT) throw std::runtime_error(std::string("CBOR: expected unsigned int for ") + what);
        return u;
    }
    static CborValue uint(uint64_t v) { CborValue c; c.kind = UINT; c.u = v; return c; }
    static CborValue text(const std::string &v) { CborValue c; c.kind = TEXT; c.s = v; return c; }
    static CborValue boolean(bool v) { CborValue c; c.kind = BOOL; c.b = v; return c; }
    static CborValue null() { return CborValue(); }
    static CborValue array() { CborValue c; c.kind = ARRAY; return c; }
    static CborValue object() { CborValue c; c.kind = MAP; return c; }
    void put(const std::string &k, CborValue v) { map.emplace_back(text(k), std::move(v)); }
};

CborValue cbor_decode(const std::vector<uint8_t> &bytes);
// Decode a top-level map but do not materialise the value stored under `skip_key`; its element
// count (array / map length) is returned through skipped_count.
CborValue cbor_decode_map_skipping(const std::vector<uint8_t> &bytes, const std::string &skip_key,
                                   uint64_t *skipped_count);
std::vector<uint8_t> cbor_encode(const CborValue &v);

}  // namespace skl_host
