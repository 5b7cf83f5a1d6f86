// msgpack.hpp -- the subset of MessagePack that rmp-serde 1.3 emits for the reference's
// `struct Inverted` (src/inverted.rs:46-58, written with rmp_serde::encode::write, :194-201):
// unsigned ints in their shortest form, str, bin, arrays, maps, bool, nil.  rmp-serde is a
// third-party crate absent from the reference tree; its documented encoding is restated here:
//   * a struct is an ARRAY of its fields in declaration order (rmp-serde's default, "compact");
//   * Vec<T> -> array, HashMap<u16, V> -> map with unsigned-int keys, String -> str,
//     Option<T> -> nil | T, bool -> true/false, usize/u16 -> the shortest unsigned form;
//   * RoaringBitmap (roaring 0.10, feature "serde") -> bin holding the portable serialisation;
//   * a unit enum variant (HashType::DNA) -> its name as a str; a newtype variant
//     (HashType::AA(level)) -> a 1-entry map {name: value}.
// Values are held in the CborValue tree (cbor.hpp): the two formats share their data model.
#pragma once

#include "cbor.hpp"

namespace skl_host {

std::vector<uint8_t> msgpack_encode(const CborValue &v);
// Throws std::runtime_error on truncated or unsupported input.  Containers are bounded by the
// input length, so a hostile length field cannot make the decoder allocate more than it reads.
CborValue msgpack_decode(const std::vector<uint8_t> &bytes);
// Decode a top-level array but do not materialise element `skip_index`; its element count
// (array / map length) is returned through skipped_count.
CborValue msgpack_decode_array_skipping(const std::vector<uint8_t> &bytes, size_t skip_index, uint64_t *skipped_count);

}  // namespace skl_host
