#include "cbor.hpp"

#include <cstring>

namespace skl_host {

namespace {
struct Reader {
    const uint8_t *p;
    size_t n, i = 0;
    uint8_t byte()
    {
        if (i >= n) throw std::runtime_error("CBOR: unexpected end of input");
        return p[i++];
    }
    uint64_t be(int bytes)
    {
        uint64_t v = 0;
        for (int k = 0; k < bytes; ++k) v = (v << 8) | byte();
        return v;
    }
    // skip one value without materialising it; returns the element count of an array / map
    // header (0 for anything else)
    uint64_t skip(int depth = 0)
    {
        if (depth > 64) throw std::runtime_error("CBOR: nesting too deep");
        const uint8_t ib = byte();
        const int mt = ib >> 5, ai = ib & 31;
        uint64_t v = 0;
        bool indefinite = false;
        if (ai < 24) v = ai;
        else if (ai == 24) v = be(1);
        else if (ai == 25) v = be(2);
        else if (ai == 26) v = be(4);
        else if (ai == 27) v = be(8);
        else if (ai == 31) indefinite = true;
        else throw std::runtime_error("CBOR: reserved additional info");
        switch (mt) {
            case 0:
            case 1: return 0;
            case 2:
            case 3:
                if (indefinite) {
                    while (i < n && p[i] != 0xFF) skip(depth + 1);
                    byte();
                } else {
                    if (v > n - i) throw std::runtime_error("CBOR: string runs past the end");
                    i += (size_t)v;
                }
                return 0;
            case 4:
            case 5: {
                const uint64_t per = mt == 5 ? 2 : 1;
                uint64_t count = 0;
                if (indefinite) {
                    while (i < n && p[i] != 0xFF) {
                        for (uint64_t k = 0; k < per; ++k) skip(depth + 1);
                        ++count;
                    }
                    byte();
                } else {
                    for (uint64_t k = 0; k < v * per; ++k) skip(depth + 1);
                    count = v;
                }
                return count;
            }
            case 6: return skip(depth + 1);
            default: return 0;   // simple values / floats: the argument bytes are already consumed
        }
    }
    CborValue value(int depth = 0)
    {
        if (depth > 64) throw std::runtime_error("CBOR: nesting too deep");
        const uint8_t ib = byte();
        const int mt = ib >> 5, ai = ib & 31;
        uint64_t v = 0;
        bool indefinite = false;
        if (ai < 24) v = ai;
        else if (ai == 24) v = be(1);
        else if (ai == 25) v = be(2);
        else if (ai == 26) v = be(4);
        else if (ai == 27) v = be(8);
        else if (ai == 31) indefinite = true;
        else throw std::runtime_error("CBOR: reserved additional info");
        CborValue c;
        switch (mt) {
            case 0: c.kind = CborValue::UINT; c.u = v; return c;
            case 1: c.kind = CborValue::NINT; c.u = v; return c;
            case 2:
            case 3: {
                c.kind = mt == 2 ? CborValue::BYTES : CborValue::TEXT;
                if (indefinite) {
                    while (i < n && p[i] != 0xFF) {
                        CborValue part = value(depth + 1);
                        c.s += part.s;
                    }
                    byte();   // the break (throws at the end of input)
                } else {
                    if (v > n - i) throw std::runtime_error("CBOR: string runs past the end");
                    c.s.assign((const char *)p + i, (size_t)v);
                    i += (size_t)v;
                }
                return c;
            }
            case 4: {
                c.kind = CborValue::ARRAY;
                if (indefinite) {
                    while (i < n && p[i] != 0xFF) c.arr.push_back(value(depth + 1));
                    byte();
                } else {
                    for (uint64_t k = 0; k < v; ++k) c.arr.push_back(value(depth + 1));
                }
                return c;
            }
            case 5: {
                c.kind = CborValue::MAP;
                if (indefinite) {
                    while (i < n && p[i] != 0xFF) {
                        CborValue k = value(depth + 1);
                        CborValue x = value(depth + 1);
                        c.map.emplace_back(std::move(k), std::move(x));
                    }
                    byte();
                } else {
                    for (uint64_t k = 0; k < v; ++k) {
                        CborValue key = value(depth + 1);
                        CborValue x = value(depth + 1);
                        c.map.emplace_back(std::move(key), std::move(x));
                    }
                }
                return c;
            }
            case 6: return value(depth + 1);  // tag: ignore, keep the tagged value
            default: {
                if (ai == 20 || ai == 21) { c.kind = CborValue::BOOL; c.b = ai == 21; return c; }
                if (ai == 22 || ai == 23) { c.kind = CborValue::NIL; return c; }
                if (ai == 26) { float f; uint32_t u = (uint32_t)v; memcpy(&f, &u, 4); c.kind = CborValue::FLOAT; c.f = f; return c; }
                if (ai == 27) { double d; memcpy(&d, &v, 8); c.kind = CborValue::FLOAT; c.f = d; return c; }
                throw std::runtime_error("CBOR: unsupported simple value");
            }
        }
    }
};

void head(std::vector<uint8_t> &o, int mt, uint64_t v)
{
    const uint8_t m = (uint8_t)(mt << 5);
    if (v < 24) o.push_back(m | (uint8_t)v);
    else if (v <= 0xFF) { o.push_back(m | 24); o.push_back((uint8_t)v); }
    else if (v <= 0xFFFF) { o.push_back(m | 25); o.push_back((uint8_t)(v >> 8)); o.push_back((uint8_t)v); }
    else if (v <= 0xFFFFFFFFull) { o.push_back(m | 26); for (int s = 24; s >= 0; s -= 8) o.push_back((uint8_t)(v >> s)); }
    else { o.push_back(m | 27); for (int s = 56; s >= 0; s -= 8) o.push_back((uint8_t)(v >> s)); }
}

void enc(std::vector<uint8_t> &o, const CborValue &v)
{
    switch (v.kind) {
        case CborValue::UINT: head(o, 0, v.u); break;
        case CborValue::NINT: head(o, 1, v.u); break;
        case CborValue::BYTES: head(o, 2, v.s.size()); o.insert(o.end(), v.s.begin(), v.s.end()); break;
        case CborValue::TEXT: head(o, 3, v.s.size()); o.insert(o.end(), v.s.begin(), v.s.end()); break;
        case CborValue::ARRAY: head(o, 4, v.arr.size()); for (const auto &x : v.arr) enc(o, x); break;
        case CborValue::MAP: head(o, 5, v.map.size()); for (const auto &kv : v.map) { enc(o, kv.first); enc(o, kv.second); } break;
        case CborValue::BOOL: o.push_back(v.b ? 0xF5 : 0xF4); break;
        case CborValue::NIL: o.push_back(0xF6); break;
        case CborValue::FLOAT: { o.push_back(0xFB); uint64_t u; memcpy(&u, &v.f, 8); for (int s = 56; s >= 0; s -= 8) o.push_back((uint8_t)(u >> s)); break; }
    }
}
}  // namespace

void CborWriter::head(int major, uint64_t v) { ::skl_host::head(out_, major, v); }

CborCursor::Head CborCursor::head()
{
    for (;;) {
        Reader r{p_, n_, i_};
        const uint8_t ib = r.byte();
        Head h;
        h.major = ib >> 5;
        h.info = ib & 31;
        if (h.info < 24) h.value = (uint64_t)h.info;
        else if (h.info == 24) h.value = r.be(1);
        else if (h.info == 25) h.value = r.be(2);
        else if (h.info == 26) h.value = r.be(4);
        else if (h.info == 27) h.value = r.be(8);
        else if (h.info == 31) h.indefinite = true;
        else throw std::runtime_error("CBOR: reserved additional info");
        i_ = r.i;
        if (h.major != 6) return h;   // a tag: ignore it, the tagged value follows
    }
}

void CborCursor::take_break()
{
    if (!at_break()) throw std::runtime_error("CBOR: expected the end of an indefinite-length item");
    ++i_;
}

std::string CborCursor::text(const Head &h)
{
    if (h.major != 2 && h.major != 3) throw std::runtime_error("CBOR: expected a string");
    std::string out;
    if (h.indefinite) {
        while (!at_break()) {
            const Head part = head();
            out += text(part);
        }
        take_break();
    } else {
        if (h.value > n_ - i_) throw std::runtime_error("CBOR: string runs past the end");
        out.assign((const char *)p_ + i_, (size_t)h.value);
        i_ += (size_t)h.value;
    }
    return out;
}

CborValue CborCursor::value()
{
    Reader r{p_, n_, i_};
    CborValue v = r.value();
    i_ = r.i;
    return v;
}

void CborCursor::skip()
{
    Reader r{p_, n_, i_};
    r.skip();
    i_ = r.i;
}

CborValue cbor_decode_map_skipping(const std::vector<uint8_t> &bytes, const std::string &skip_key,
                                   uint64_t *skipped_count)
{
    Reader r{bytes.data(), bytes.size()};
    const uint8_t ib = r.byte();
    if ((ib >> 5) != 5) throw std::runtime_error("CBOR: top-level value is not a map");
    const int ai = ib & 31;
    uint64_t v = 0;
    bool indefinite = false;
    if (ai < 24) v = ai;
    else if (ai == 24) v = r.be(1);
    else if (ai == 25) v = r.be(2);
    else if (ai == 26) v = r.be(4);
    else if (ai == 27) v = r.be(8);
    else if (ai == 31) indefinite = true;
    else throw std::runtime_error("CBOR: reserved additional info");
    CborValue root = CborValue::object();
    if (skipped_count) *skipped_count = 0;
    for (uint64_t k = 0; indefinite ? (r.i < r.n && r.p[r.i] != 0xFF) : k < v; ++k) {
        CborValue key = r.value(1);
        if (key.kind == CborValue::TEXT && key.s == skip_key) {
            const uint64_t c = r.skip(1);
            if (skipped_count) *skipped_count = c;
        } else {
            CborValue x = r.value(1);
            root.map.emplace_back(std::move(key), std::move(x));
        }
    }
    return root;
}

CborValue cbor_decode(const std::vector<uint8_t> &bytes)
{
    Reader r{bytes.data(), bytes.size()};
    return r.value();
}

std::vector<uint8_t> cbor_encode(const CborValue &v)
{
    std::vector<uint8_t> o;
    enc(o, v);
    return o;
}

}  // namespace skl_host
