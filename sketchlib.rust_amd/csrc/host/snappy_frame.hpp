// snappy_frame.hpp -- the Snappy *framing format* (what Rust's snap::read::FrameDecoder /
// snap::write::FrameEncoder speak; the reference wraps the .skm CBOR in it,
// src/sketch/multisketch.rs:80-103).  Self-contained: no libsnappy in this image.
//
// Frame: stream identifier chunk ff 06 00 00 "sNaPpY", then chunks
//   [type:1][length:3 LE][masked CRC32C of the uncompressed data:4 LE][payload]
// with type 0x00 = raw-snappy compressed payload, 0x01 = uncompressed payload,
// 0xfe = padding, 0x80..0xfd = skippable.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace skl_host {

// Decode a whole framed stream.  Throws std::runtime_error on malformed input or CRC
// mismatch.
std::vector<uint8_t> snappy_frame_decode(const std::vector<uint8_t> &framed);

// Encode as a framed stream of uncompressed (0x01) chunks -- valid input for any
// conforming decoder (snap verifies the checksums).
std::vector<uint8_t> snappy_frame_encode(const std::vector<uint8_t> &raw);

uint32_t crc32c(const uint8_t *data, size_t n);

}  // namespace skl_host
