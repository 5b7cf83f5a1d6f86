#include "snappy_frame.hpp"

#include <cstring>
#include <stdexcept>

namespace skl_host {

uint32_t crc32c(const uint8_t *data, size_t n)
{
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ 0x82F63B78u : (c >> 1);
            table[i] = c;
        }
        init = true;
    }
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; ++i) c = table[(c ^ data[i]) & 0xFFu] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

static uint32_t mask_crc(uint32_t crc) { return ((crc >> 15) | (crc << 17)) + 0xA282EAD8u; }

// raw snappy block: varint uncompressed length, then literal / copy elements
static void snappy_raw_decode(const uint8_t *p, size_t n, std::vector<uint8_t> &out)
{
    size_t i = 0;
    uint64_t len = 0;
    int shift = 0;
    for (;;) {
        if (i >= n) throw std::runtime_error("snappy: truncated length");
        const uint8_t b = p[i++];
        len |= (uint64_t)(b & 0x7F) << shift;
        if (!(b & 0x80)) break;
        shift += 7;
        if (shift > 35) throw std::runtime_error("snappy: bad length varint");
    }
    const size_t base = out.size();
    out.reserve(base + len);
    while (i < n) {
        const uint8_t tag = p[i++];
        const int type = tag & 3;
        if (type == 0) {
            size_t l = tag >> 2;
            if (l >= 60) {
                const size_t nb = l - 59;
                if (i + nb > n) throw std::runtime_error("snappy: truncated literal length");
                l = 0;
                for (size_t k = 0; k < nb; ++k) l |= (size_t)p[i + k] << (8 * k);
                i += nb;
            }
            l += 1;
            if (i + l > n) throw std::runtime_error("snappy: truncated literal");
            out.insert(out.end(), p + i, p + i + l);
            i += l;
        } else {
            size_t l, off;
            if (type == 1) {
                if (i + 1 > n) throw std::runtime_error("snappy: truncated copy");
                l = ((tag >> 2) & 7) + 4;
                off = ((size_t)(tag >> 5) << 8) | p[i];
                i += 1;
            } else if (type == 2) {
                if (i + 2 > n) throw std::runtime_error("snappy: truncated copy");
                l = (tag >> 2) + 1;
                off = (size_t)p[i] | ((size_t)p[i + 1] << 8);
                i += 2;
            } else {
                if (i + 4 > n) throw std::runtime_error("snappy: truncated copy");
                l = (tag >> 2) + 1;
                off = (size_t)p[i] | ((size_t)p[i + 1] << 8) | ((size_t)p[i + 2] << 16) |
                      ((size_t)p[i + 3] << 24);
                i += 4;
            }
            if (off == 0 || off > out.size() - base) throw std::runtime_error("snappy: bad copy offset");
            for (size_t k = 0; k < l; ++k) out.push_back(out[out.size() - off]);
        }
    }
    if (out.size() - base != len) throw std::runtime_error("snappy: length mismatch");
}

std::vector<uint8_t> snappy_frame_decode(const std::vector<uint8_t> &f)
{
    std::vector<uint8_t> out;
    size_t i = 0;
    bool seen_id = false;
    while (i < f.size()) {
        if (i + 4 > f.size()) throw std::runtime_error("snappy frame: truncated chunk header");
        const uint8_t type = f[i];
        const size_t len = (size_t)f[i + 1] | ((size_t)f[i + 2] << 8) | ((size_t)f[i + 3] << 16);
        i += 4;
        if (i + len > f.size()) throw std::runtime_error("snappy frame: truncated chunk");
        const uint8_t *body = f.data() + i;
        if (type == 0xFF) {
            if (len != 6 || memcmp(body, "sNaPpY", 6) != 0) throw std::runtime_error("snappy frame: bad stream identifier");
            seen_id = true;
        } else if (type == 0x00 || type == 0x01) {
            if (!seen_id) throw std::runtime_error("snappy frame: data before stream identifier");
            if (len < 4) throw std::runtime_error("snappy frame: chunk too short");
            const uint32_t want = (uint32_t)body[0] | ((uint32_t)body[1] << 8) |
                                  ((uint32_t)body[2] << 16) | ((uint32_t)body[3] << 24);
            const size_t before = out.size();
            if (type == 0x00) {
                snappy_raw_decode(body + 4, len - 4, out);
            } else {
                out.insert(out.end(), body + 4, body + len);
            }
            if (mask_crc(crc32c(out.data() + before, out.size() - before)) != want) {
                throw std::runtime_error("snappy frame: checksum mismatch");
            }
        } else if (type >= 0x02 && type <= 0x7F) {
            throw std::runtime_error("snappy frame: reserved unskippable chunk");
        }  // 0x80..0xfe: skippable / padding
        i += len;
    }
    if (!seen_id) throw std::runtime_error("snappy frame: missing stream identifier");
    return out;
}

std::vector<uint8_t> snappy_frame_encode(const std::vector<uint8_t> &raw)
{
    std::vector<uint8_t> f = {0xFF, 0x06, 0x00, 0x00, 's', 'N', 'a', 'P', 'p', 'Y'};
    const size_t kMax = 65536;
    size_t i = 0;
    do {
        const size_t n = std::min(kMax, raw.size() - i);
        const size_t len = n + 4;
        const uint32_t crc = mask_crc(crc32c(raw.data() + i, n));
        f.push_back(0x01);
        f.push_back((uint8_t)(len & 0xFF));
        f.push_back((uint8_t)((len >> 8) & 0xFF));
        f.push_back((uint8_t)((len >> 16) & 0xFF));
        for (int k = 0; k < 4; ++k) f.push_back((uint8_t)((crc >> (8 * k)) & 0xFF));
        f.insert(f.end(), raw.begin() + i, raw.begin() + i + n);
        i += n;
    } while (i < raw.size());
    return f;
}

}  // namespace skl_host
