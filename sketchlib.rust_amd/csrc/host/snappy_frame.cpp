#include "snappy_frame.hpp"

#include <algorithm>
#include <atomic>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <thread>

namespace skl_host {

static uint32_t crc32c_table(uint32_t c, const uint8_t *data, size_t n)
{
    static const struct Table {
        uint32_t t[256];
        Table()
        {
            for (uint32_t i = 0; i < 256; ++i) {
                uint32_t c = i;
                for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ 0x82F63B78u : (c >> 1);
                t[i] = c;
            }
        }
    } table;
    for (size_t i = 0; i < n; ++i) c = table.t[(c ^ data[i]) & 0xFFu] ^ (c >> 8);
    return c;
}

#if defined(__x86_64__)
// the CRC32C instruction (SSE4.2): ~20x the table walk; index files are hundreds of MB
__attribute__((target("sse4.2"))) static uint32_t crc32c_hw(uint32_t c, const uint8_t *data, size_t n)
{
    uint64_t c64 = c;
    while (n >= 8) {
        uint64_t v;
        memcpy(&v, data, 8);
        c64 = __builtin_ia32_crc32di(c64, v);
        data += 8;
        n -= 8;
    }
    c = (uint32_t)c64;
    while (n--) c = __builtin_ia32_crc32qi(c, *data++);
    return c;
}
#endif

uint32_t crc32c(const uint8_t *data, size_t n)
{
#if defined(__x86_64__)
    static const bool hw = __builtin_cpu_supports("sse4.2");
    if (hw) return crc32c_hw(0xFFFFFFFFu, data, n) ^ 0xFFFFFFFFu;
#endif
    return crc32c_table(0xFFFFFFFFu, data, n) ^ 0xFFFFFFFFu;
}

static uint32_t mask_crc(uint32_t crc) { return ((crc >> 15) | (crc << 17)) + 0xA282EAD8u; }

// varint uncompressed length at the head of a raw snappy block; returns the bytes it took
static size_t snappy_raw_length(const uint8_t *p, size_t n, uint64_t &len)
{
    size_t i = 0;
    int shift = 0;
    len = 0;
    for (;;) {
        if (i >= n) throw std::runtime_error("snappy: truncated length");
        const uint8_t b = p[i++];
        len |= (uint64_t)(b & 0x7F) << shift;
        if (!(b & 0x80)) break;
        shift += 7;
        if (shift > 35) throw std::runtime_error("snappy: bad length varint");
    }
    return i;
}

// raw snappy block (varint uncompressed length, then literal / copy elements) into out[0, out_len)
static void snappy_raw_decode(const uint8_t *p, size_t n, uint8_t *out, size_t out_len)
{
    uint64_t len = 0;
    size_t i = snappy_raw_length(p, n, len);
    if (len != out_len) throw std::runtime_error("snappy: length mismatch");
    size_t o = 0;
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
            if (o + l > out_len) throw std::runtime_error("snappy: length mismatch");
            memcpy(out + o, p + i, l);
            o += l;
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
            if (off == 0 || off > o) throw std::runtime_error("snappy: bad copy offset");
            if (o + l > out_len) throw std::runtime_error("snappy: length mismatch");
            if (off >= l) {
                memcpy(out + o, out + o - off, l);
            } else {
                for (size_t k = 0; k < l; ++k) out[o + k] = out[o + k - off];   // overlapping run
            }
            o += l;
        }
    }
    if (o != out_len) throw std::runtime_error("snappy: length mismatch");
}

std::vector<uint8_t> snappy_frame_decode(const std::vector<uint8_t> &f)
{
    // pass 1: the chunk list (chunks are independent: each carries its own checksum and, when
    // compressed, its own uncompressed length)
    struct Chunk {
        const uint8_t *body;   // after the 4-byte checksum
        size_t len, out_off, out_len;
        uint32_t want;
        bool compressed;
    };
    std::vector<Chunk> chunks;
    size_t i = 0, total = 0;
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
            Chunk c;
            c.want = (uint32_t)body[0] | ((uint32_t)body[1] << 8) | ((uint32_t)body[2] << 16) | ((uint32_t)body[3] << 24);
            c.body = body + 4;
            c.len = len - 4;
            c.compressed = type == 0x00;
            uint64_t ulen = c.len;
            if (c.compressed) snappy_raw_length(c.body, c.len, ulen);
            // the framing format caps a chunk at 65536 uncompressed bytes (and the `snap` crate the
            // reference uses enforces it): a 25-byte file may not claim gigabytes of output
            if (ulen > 65536) throw std::runtime_error("snappy frame: chunk claims more than 65536 uncompressed bytes");
            c.out_off = total;
            c.out_len = (size_t)ulen;
            total += c.out_len;
            chunks.push_back(c);
        } else if (type >= 0x02 && type <= 0x7F) {
            throw std::runtime_error("snappy frame: reserved unskippable chunk");
        }  // 0x80..0xfe: skippable / padding
        i += len;
    }
    if (!seen_id) throw std::runtime_error("snappy frame: missing stream identifier");

    // pass 2: decode + verify, in parallel for large streams
    std::vector<uint8_t> out(total);
    std::atomic<size_t> next{0};
    std::mutex err_mutex;
    std::string err;
    auto worker = [&]() {
        try {
            for (;;) {
                const size_t first = next.fetch_add(64);
                if (first >= chunks.size()) break;
                for (size_t x = first; x < std::min(chunks.size(), first + 64); ++x) {
                    const Chunk &c = chunks[x];
                    uint8_t *dst = out.data() + c.out_off;
                    if (c.compressed) snappy_raw_decode(c.body, c.len, dst, c.out_len);
                    else memcpy(dst, c.body, c.out_len);
                    if (mask_crc(crc32c(dst, c.out_len)) != c.want) throw std::runtime_error("snappy frame: checksum mismatch");
                }
            }
        } catch (const std::exception &e) {
            std::lock_guard<std::mutex> lock(err_mutex);
            if (err.empty()) err = e.what();
        }
    };
    const size_t hw = std::max(1u, std::thread::hardware_concurrency());
    const size_t n_threads = total < (8u << 20) ? 1 : std::min<size_t>({hw, 16, chunks.size() / 64 + 1});
    std::vector<std::thread> pool;
    for (size_t t = 1; t < n_threads; ++t) pool.emplace_back(worker);
    worker();
    for (auto &t : pool) t.join();
    if (!err.empty()) throw std::runtime_error(err);
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
