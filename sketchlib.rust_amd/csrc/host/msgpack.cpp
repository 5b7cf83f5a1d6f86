// msgpack.cpp -- see msgpack.hpp.
#include "msgpack.hpp"

#include <cstring>

namespace skl_host {

namespace {

void put_be(std::vector<uint8_t> &out, uint64_t v, int bytes)
{
    for (int i = bytes - 1; i >= 0; --i) out.push_back((uint8_t)(v >> (8 * i)));
}

void encode(std::vector<uint8_t> &out, const CborValue &v)
{
    switch (v.kind) {
        case CborValue::UINT:
            if (v.u < 128) out.push_back((uint8_t)v.u);
            else if (v.u <= 0xFF) { out.push_back(0xcc); put_be(out, v.u, 1); }
            else if (v.u <= 0xFFFF) { out.push_back(0xcd); put_be(out, v.u, 2); }
            else if (v.u <= 0xFFFFFFFFull) { out.push_back(0xce); put_be(out, v.u, 4); }
            else { out.push_back(0xcf); put_be(out, v.u, 8); }
            break;
        case CborValue::TEXT: {
            const size_t n = v.s.size();
            if (n < 32) out.push_back((uint8_t)(0xa0 | n));
            else if (n <= 0xFF) { out.push_back(0xd9); put_be(out, n, 1); }
            else if (n <= 0xFFFF) { out.push_back(0xda); put_be(out, n, 2); }
            else { out.push_back(0xdb); put_be(out, n, 4); }
            out.insert(out.end(), v.s.begin(), v.s.end());
            break;
        }
        case CborValue::BYTES: {
            const size_t n = v.s.size();
            if (n <= 0xFF) { out.push_back(0xc4); put_be(out, n, 1); }
            else if (n <= 0xFFFF) { out.push_back(0xc5); put_be(out, n, 2); }
            else { out.push_back(0xc6); put_be(out, n, 4); }
            out.insert(out.end(), v.s.begin(), v.s.end());
            break;
        }
        case CborValue::ARRAY: {
            const size_t n = v.arr.size();
            if (n < 16) out.push_back((uint8_t)(0x90 | n));
            else if (n <= 0xFFFF) { out.push_back(0xdc); put_be(out, n, 2); }
            else { out.push_back(0xdd); put_be(out, n, 4); }
            for (const auto &x : v.arr) encode(out, x);
            break;
        }
        case CborValue::MAP: {
            const size_t n = v.map.size();
            if (n < 16) out.push_back((uint8_t)(0x80 | n));
            else if (n <= 0xFFFF) { out.push_back(0xde); put_be(out, n, 2); }
            else { out.push_back(0xdf); put_be(out, n, 4); }
            for (const auto &kv : v.map) {
                encode(out, kv.first);
                encode(out, kv.second);
            }
            break;
        }
        case CborValue::BOOL: out.push_back(v.b ? 0xc3 : 0xc2); break;
        case CborValue::NIL: out.push_back(0xc0); break;
        default: throw std::runtime_error("MessagePack: value kind not supported by the writer");
    }
}

struct Reader {
    const uint8_t *p;
    size_t n, i = 0;

    uint8_t byte()
    {
        if (i >= n) throw std::runtime_error("MessagePack: truncated input");
        return p[i++];
    }
    uint64_t be(int bytes)
    {
        if (n - i < (size_t)bytes) throw std::runtime_error("MessagePack: truncated input");
        uint64_t v = 0;
        for (int k = 0; k < bytes; ++k) v = (v << 8) | p[i++];
        return v;
    }
    std::string raw(uint64_t len)
    {
        if (len > n - i) throw std::runtime_error("MessagePack: string longer than the input");
        std::string s((const char *)p + i, (size_t)len);
        i += (size_t)len;
        return s;
    }
    // every element takes at least one byte: a count beyond the remaining input is malformed
    void check_count(uint64_t count, uint64_t per) const
    {
        if (count > (n - i) / per + 1 && count * per > n - i) throw std::runtime_error("MessagePack: container longer than the input");
    }

    // kind of the next value's container/len without materialising: returns through `v` for scalars
    CborValue value(int depth = 0)
    {
        if (depth > 64) throw std::runtime_error("MessagePack: nesting too deep");
        const uint8_t t = byte();
        CborValue v;
        auto array = [&](uint64_t count) {
            check_count(count, 1);
            v.kind = CborValue::ARRAY;
            v.arr.reserve((size_t)count);
            for (uint64_t k = 0; k < count; ++k) v.arr.push_back(value(depth + 1));
        };
        auto map = [&](uint64_t count) {
            check_count(count, 2);
            v.kind = CborValue::MAP;
            v.map.reserve((size_t)count);
            for (uint64_t k = 0; k < count; ++k) {
                CborValue key = value(depth + 1);
                v.map.emplace_back(std::move(key), value(depth + 1));
            }
        };
        if (t < 0x80) return CborValue::uint(t);
        if (t >= 0xe0) { v.kind = CborValue::NINT; v.u = (uint64_t)(0xFF - t); return v; }   // negative fixint: -1 - u
        if ((t & 0xf0) == 0x80) { map(t & 0x0f); return v; }
        if ((t & 0xf0) == 0x90) { array(t & 0x0f); return v; }
        if ((t & 0xe0) == 0xa0) return CborValue::text(raw(t & 0x1f));
        switch (t) {
            case 0xc0: return CborValue::null();
            case 0xc2: return CborValue::boolean(false);
            case 0xc3: return CborValue::boolean(true);
            case 0xc4: case 0xc5: case 0xc6: {
                v.kind = CborValue::BYTES;
                v.s = raw(be(1 << (t - 0xc4)));
                return v;
            }
            case 0xcc: return CborValue::uint(be(1));
            case 0xcd: return CborValue::uint(be(2));
            case 0xce: return CborValue::uint(be(4));
            case 0xcf: return CborValue::uint(be(8));
            case 0xd0: case 0xd1: case 0xd2: case 0xd3: {   // signed: serde writes them for negative values only
                const int bytes = 1 << (t - 0xd0);
                const uint64_t u = be(bytes);
                const int64_t s = bytes == 8 ? (int64_t)u : (int64_t)(u << (64 - 8 * bytes)) >> (64 - 8 * bytes);
                if (s >= 0) return CborValue::uint((uint64_t)s);
                v.kind = CborValue::NINT;
                v.u = (uint64_t)(-1 - s);
                return v;
            }
            case 0xd9: return CborValue::text(raw(be(1)));
            case 0xda: return CborValue::text(raw(be(2)));
            case 0xdb: return CborValue::text(raw(be(4)));
            case 0xdc: array(be(2)); return v;
            case 0xdd: array(be(4)); return v;
            case 0xde: map(be(2)); return v;
            case 0xdf: map(be(4)); return v;
            default: throw std::runtime_error("MessagePack: unsupported type byte");   // floats, ext: serde never writes them here
        }
    }

    // skip one value; returns the element count if it is a container (0 otherwise)
    uint64_t skip(int depth = 0)
    {
        if (depth > 64) throw std::runtime_error("MessagePack: nesting too deep");
        const uint8_t t = byte();
        auto items = [&](uint64_t count, int per) {
            check_count(count, (uint64_t)per);
            for (uint64_t k = 0; k < count * (uint64_t)per; ++k) skip(depth + 1);
            return count;
        };
        if (t < 0x80 || t >= 0xe0) return 0;
        if ((t & 0xf0) == 0x80) return items(t & 0x0f, 2);
        if ((t & 0xf0) == 0x90) return items(t & 0x0f, 1);
        if ((t & 0xe0) == 0xa0) { raw(t & 0x1f); return 0; }
        switch (t) {
            case 0xc0: case 0xc2: case 0xc3: return 0;
            case 0xc4: case 0xc5: case 0xc6: raw(be(1 << (t - 0xc4))); return 0;
            case 0xcc: case 0xd0: be(1); return 0;
            case 0xcd: case 0xd1: be(2); return 0;
            case 0xce: case 0xd2: be(4); return 0;
            case 0xcf: case 0xd3: be(8); return 0;
            case 0xd9: raw(be(1)); return 0;
            case 0xda: raw(be(2)); return 0;
            case 0xdb: raw(be(4)); return 0;
            case 0xdc: return items(be(2), 1);
            case 0xdd: return items(be(4), 1);
            case 0xde: return items(be(2), 2);
            case 0xdf: return items(be(4), 2);
            default: throw std::runtime_error("MessagePack: unsupported type byte");
        }
    }
};

}  // namespace

std::vector<uint8_t> msgpack_encode(const CborValue &v)
{
    std::vector<uint8_t> out;
    encode(out, v);
    return out;
}

CborValue msgpack_decode(const std::vector<uint8_t> &bytes)
{
    Reader r{bytes.data(), bytes.size()};
    CborValue v = r.value();
    if (r.i != bytes.size()) throw std::runtime_error("MessagePack: trailing bytes after the document");
    return v;
}

CborValue msgpack_decode_array_skipping(const std::vector<uint8_t> &bytes, size_t skip_index, uint64_t *skipped_count)
{
    Reader r{bytes.data(), bytes.size()};
    const uint8_t t = r.byte();
    uint64_t count;
    if ((t & 0xf0) == 0x90) count = t & 0x0f;
    else if (t == 0xdc) count = r.be(2);
    else if (t == 0xdd) count = r.be(4);
    else throw std::runtime_error("MessagePack: expected an array");
    r.check_count(count, 1);
    CborValue v = CborValue::array();
    for (uint64_t k = 0; k < count; ++k) {
        if (k == skip_index) {
            const uint64_t c = r.skip();
            if (skipped_count) *skipped_count = c;
            v.arr.push_back(CborValue::null());
        } else {
            v.arr.push_back(r.value(1));
        }
    }
    if (r.i != bytes.size()) throw std::runtime_error("MessagePack: trailing bytes after the document");
    return v;
}

}  // namespace skl_host
