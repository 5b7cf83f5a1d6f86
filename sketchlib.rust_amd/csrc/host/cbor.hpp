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

// Pull parser over the same subset, for documents too large to be worth a CborValue tree (the
// .skm of a 400 000-sample database holds ~8 million values).  Usage: h = head(); then, by
// h.major: 0/1 -> h.value is the number; 2/3 -> text(h); 4/5 -> iterate h.value items (pairs),
// or until at_break() when h.indefinite, then take_break(); 7 -> h.info 20/21 = false/true,
// 22/23 = null.  Tags are skipped transparently by head().
class CborCursor {
public:
    struct Head {
        int major = 0, info = 0;
        uint64_t value = 0;
        bool indefinite = false;
    };
    CborCursor(const uint8_t *p, size_t n) : p_(p), n_(n) {}
    Head head();
    bool at_break() const { return i_ < n_ && p_[i_] == 0xFF; }
    void take_break();
    std::string text(const Head &h);   // body of a text / byte string whose head was just read
    CborValue value();                 // one whole value, generically
    void skip();                       // one whole value, unread
    // true while a container opened by `h` has another item; k counts the items taken so far
    bool more(const Head &h, uint64_t k) const { return h.indefinite ? !at_break() : k < h.value; }
    void close(const Head &h) { if (h.indefinite) take_break(); }

private:
    const uint8_t *p_;
    size_t n_, i_ = 0;
};

// Push writer over the same subset (definite lengths, as cbor_encode writes them): a document is written value by value
// into one byte vector -- the CborValue tree of a million-sample .skm is gigabytes of nodes.
class CborWriter {
public:
    void uint(uint64_t v) { head(0, v); }
    void text(const std::string &s)
    {
        head(3, s.size());
        out_.insert(out_.end(), s.begin(), s.end());
    }
    void boolean(bool b) { out_.push_back(b ? 0xF5 : 0xF4); }
    void null() { out_.push_back(0xF6); }
    void array(uint64_t n_items) { head(4, n_items); }   // followed by n_items values
    void map(uint64_t n_pairs) { head(5, n_pairs); }     // followed by n_pairs (key, value) pairs
    void key(const char *k) { text(k); }
    std::vector<uint8_t> &bytes() { return out_; }

private:
    void head(int major, uint64_t v);
    std::vector<uint8_t> out_;
};

CborValue cbor_decode(const std::vector<uint8_t> &bytes);
// Decode a top-level map but do not materialise the value stored under `skip_key`; its element
// count (array / map length) is returned through skipped_count.
CborValue cbor_decode_map_skipping(const std::vector<uint8_t> &bytes, const std::string &skip_key,
                                   uint64_t *skipped_count);
std::vector<uint8_t> cbor_encode(const CborValue &v);

}  // namespace skl_host
