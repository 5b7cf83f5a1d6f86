"""`sketchlib inverted build` and the `.ski` / `.skq` formats without a GPU:
the `.skq` our native sketcher writes is byte-identical to the reference's golden
(tests/inverted.rs:260-270), `precluster --count` prints the reference's line (:279-285), and
the `.ski` is decoded here by independent Python readers (snappy frame -> MessagePack via the
`msgpack` package -> Roaring portable format) and compared with the index built from the `.skq`.
The reference writes the index with rmp_serde::encode::write (inverted.rs:194-201): struct Inverted
as a 9-element MessagePack array in field order."""
import os
import shutil
import subprocess

import msgpack
import numpy as np
import pytest

from conftest import REF_FIXTURES, ROOT
from helpers import FIXTURE_NAMES
from test_fileformat_cpu import _py_unframe

CLI = os.path.join(ROOT, "sketchlib.rust_amd", "csrc", "_build", "sketchlib")


@pytest.fixture(scope="module", autouse=True)
def _built(skl):
    assert os.path.exists(CLI)


@pytest.fixture()
def wd(tmp_path):
    for f in FIXTURE_NAMES + ["rfile.txt"]:
        shutil.copy(os.path.join(REF_FIXTURES, f), tmp_path / f)
    return tmp_path


def run(wd, *args, ok=True):
    res = subprocess.run([CLI, *args], cwd=wd, capture_output=True, text=True)
    if ok:
        assert res.returncode == 0, res.stderr
    return res


FIELDS = ["index", "n_samples", "sample_names", "metadata", "labels", "kmer_size", "sketch_version", "rc",
          "hash_type"]                     # field order of struct Inverted (inverted.rs:46-58)


def ski_decode(path):
    """.ski -> dict by field name; asserts the document is rmp-serde's array form in its canonical
    (shortest) encodings: re-packing the decoded document gives the same bytes."""
    doc = _py_unframe(path.read_bytes())
    arr = msgpack.unpackb(doc, raw=False, strict_map_key=False)
    assert isinstance(arr, list) and len(arr) == 9 and doc[0] == 0x99
    assert msgpack.packb(arr, use_bin_type=True) == doc
    return dict(zip(FIELDS, arr))


def cbor_decode(b, i=0):
    """Minimal CBOR reader (definite lengths), enough for serde's output."""
    mt, ai = b[i] >> 5, b[i] & 31
    i += 1
    if ai < 24:
        v = ai
    elif ai in (24, 25, 26, 27):
        ln = 1 << (ai - 24)
        v = int.from_bytes(b[i:i + ln], "big")
        i += ln
    else:
        assert mt == 7
        v = None
    if mt == 0:
        return v, i
    if mt in (2, 3):
        s = b[i:i + v]
        return (s if mt == 2 else s.decode()), i + v
    if mt == 4:
        out = []
        for _ in range(v):
            x, i = cbor_decode(b, i)
            out.append(x)
        return out, i
    if mt == 5:
        out = {}
        for _ in range(v):
            k, i = cbor_decode(b, i)
            x, i = cbor_decode(b, i)
            out[k] = x
        return out, i
    if mt == 7:
        return {20: False, 21: True, 22: None}[ai], i
    raise AssertionError(f"unexpected CBOR major type {mt}")


def roaring_decode(b):
    """Roaring portable format without run containers (RoaringFormatSpec)."""
    cookie = int.from_bytes(b[0:4], "little")
    assert cookie == 12346
    size = int.from_bytes(b[4:8], "little")
    desc = [(int.from_bytes(b[8 + 4 * c:10 + 4 * c], "little"), int.from_bytes(b[10 + 4 * c:12 + 4 * c], "little") + 1)
            for c in range(size)]
    pos = 8 + 4 * size
    offsets = [int.from_bytes(b[pos + 4 * c:pos + 4 * c + 4], "little") for c in range(size)]
    out = []
    for (key, card), off in zip(desc, offsets):
        if card > 4096:
            bits = np.unpackbits(np.frombuffer(b[off:off + 8192], dtype=np.uint8), bitorder="little")
            out += [(key << 16) | int(x) for x in np.nonzero(bits)[0]]
        else:
            out += [(key << 16) | int(x) for x in np.frombuffer(b[off:off + 2 * card], dtype="<u2")]
    return out


def test_build_writes_reference_skq_and_a_decodable_ski(wd):
    run(wd, "inverted", "build", "-o", "inverted", "-v", "-k", "21", "-s", "10", "-f", "rfile.txt", "--write-skq")
    assert (wd / "inverted.skq").read_bytes() == open(os.path.join(REF_FIXTURES, "inverted.skq"), "rb").read()
    res = run(wd, "inverted", "precluster", "-v", "--count", "inverted.ski")
    assert res.stdout == "Identified 2 prefilter pairs from a max of 6\n"
    ski = ski_decode(wd / "inverted.ski")
    assert ski["sample_names"] == FIXTURE_NAMES and ski["n_samples"] == 4 and ski["kmer_size"] == 21
    assert ski["rc"] is True and ski["hash_type"] == "DNA" and ski["metadata"] is None and ski["labels"] is None
    skq = np.fromfile(wd / "inverted.skq", dtype="<u2").reshape(4, 10)
    assert len(ski["index"]) == 10
    for b, table in enumerate(ski["index"]):
        expect = {}
        for s in range(4):
            expect.setdefault(int(skq[s, b]), []).append(s)
            assert all(isinstance(v, bytes) for v in table.values())      # bin blobs, as serde's serialize_bytes gives
        assert {k: roaring_decode(v) for k, v in table.items()} == expect


def test_large_bitmaps_round_trip_through_the_cli(wd, tmp_path):
    """A bin value shared by > 4096 samples is a bitmap container; ids beyond 65535 need a second
    container.  A .ski with both is assembled here from the documented layout (struct Inverted as
    rmp-serde's MessagePack array, Roaring portable bitmaps, snappy frame) and read by the C++ loader
    via --count -- and once more as the snappy-framed CBOR map that round 1 of this repository wrote,
    which the loader still accepts."""
    # 70 000 samples x 2 bins: bin 0 is the same for everyone, bin 1 is shared by neighbours
    n = 70000

    def roaring_encode(vals):
        by_key = {}
        for v in vals:
            by_key.setdefault(v >> 16, []).append(v & 0xFFFF)
        keys = sorted(by_key)
        head = (12346).to_bytes(4, "little") + len(keys).to_bytes(4, "little")
        desc = b"".join(k.to_bytes(2, "little") + (len(by_key[k]) - 1).to_bytes(2, "little") for k in keys)
        off, offs, data = 8 + 8 * len(keys), b"", b""
        for k in keys:
            offs += off.to_bytes(4, "little")
            lows = by_key[k]
            if len(lows) > 4096:
                bits = np.zeros(65536, dtype=np.uint8)
                bits[lows] = 1
                blob = np.packbits(bits, bitorder="little").tobytes()
            else:
                blob = np.array(lows, dtype="<u2").tobytes()
            data += blob
            off += len(blob)
        return head + desc + offs + data

    def enc_uint(mt, v):
        if v < 24:
            return bytes([(mt << 5) | v])
        for ai, ln in ((24, 1), (25, 2), (26, 4), (27, 8)):
            if v < 1 << (8 * ln):
                return bytes([(mt << 5) | ai]) + v.to_bytes(ln, "big")
        raise AssertionError

    def enc(x):
        if x is None:
            return b"\xf6"
        if x is True or x is False:
            return b"\xf5" if x else b"\xf4"
        if isinstance(x, int):
            return enc_uint(0, x)
        if isinstance(x, bytes):
            return enc_uint(2, len(x)) + x
        if isinstance(x, str):
            e = x.encode()
            return enc_uint(3, len(e)) + e
        if isinstance(x, list):
            return enc_uint(4, len(x)) + b"".join(enc(y) for y in x)
        if isinstance(x, dict):
            return enc_uint(5, len(x)) + b"".join(enc(k) + enc(v) for k, v in x.items())
        raise AssertionError(type(x))

    index = [{7: roaring_encode(list(range(n)))}, {}]
    for s in range(n):
        index[1].setdefault((s // 2) % 65536, []).append(s)
    index[1] = {k: roaring_encode(v) for k, v in index[1].items()}
    ski = {"index": index, "n_samples": n, "sample_names": [f"s{i}" for i in range(n)], "metadata": None,
           "labels": None, "kmer_size": 21, "sketch_version": "0.3.0", "rc": True, "hash_type": "DNA"}
    # snappy frame, uncompressed chunks
    table = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        table.append(c)

    def crc32c(b):
        c = 0xFFFFFFFF
        for x in b:
            c = table[(c ^ x) & 0xFF] ^ (c >> 8)
        return c ^ 0xFFFFFFFF

    def frame(raw):
        framed = b"\xff\x06\x00\x00sNaPpY"
        for o in range(0, len(raw), 60000):
            chunk = raw[o:o + 60000]
            c = crc32c(chunk)
            masked = ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF
            framed += b"\x01" + (len(chunk) + 4).to_bytes(3, "little") + masked.to_bytes(4, "little") + chunk
        return framed

    expect = f"Identified {n * (n - 1) // 2} prefilter pairs from a max of {n * (n - 1) // 2}\n"
    (tmp_path / "big.ski").write_bytes(frame(msgpack.packb([ski[f] for f in FIELDS], use_bin_type=True)))
    assert run(tmp_path, "inverted", "precluster", "--count", "big.ski", "--threads", "4").stdout == expect
    (tmp_path / "legacy.ski").write_bytes(frame(enc(ski)))
    assert run(tmp_path, "inverted", "precluster", "--count", "legacy.ski", "--threads", "4").stdout == expect


def test_species_names_reorder_the_index(wd):
    (wd / "species.txt").write_text("R6.fa.gz\tpneumo\n14412_3#84.contigs_velvet.fa.gz\tother\nTIGR4.fa.gz\tpneumo\n")
    run(wd, "inverted", "build", "-o", "reordered", "-k", "21", "-s", "10", "-f", "rfile.txt", "--write-skq",
        "--species-names", "species.txt")
    ski = ski_decode(wd / "reordered.ski")
    # labels in order of first appearance, unlabelled inputs last (io.rs:40-115)
    assert ski["sample_names"] == ["R6.fa.gz", "TIGR4.fa.gz", "14412_3#84.contigs_velvet.fa.gz",
                                   "14412_3#82.contigs_velvet.fa.gz"]
    assert ski["labels"] == ["pneumo", "pneumo", "other", ""]
    golden = np.fromfile(os.path.join(REF_FIXTURES, "inverted.skq"), dtype="<u2").reshape(4, 10)
    got = np.fromfile(wd / "reordered.skq", dtype="<u2").reshape(4, 10)
    assert np.array_equal(got, golden[[2, 3, 1, 0]])
    assert run(wd, "inverted", "precluster", "--count", "reordered.ski").stdout == \
        "Identified 2 prefilter pairs from a max of 6\n"


def test_metadata_is_stored_in_index_order(wd):
    """tests/inverted.rs:52-77 (`inverted build --species-names --metadata`)."""
    (wd / "metadata.txt").write_text("TIGR4.fa.gz\tMetadata of TIGR4\nR6.fa.gz\tMetadata of R6\n"
                                     "14412_3#82.contigs_velvet.fa.gz\tMetadata of 14412_3 82\n"
                                     "14412_3#84.contigs_velvet.fa.gz\tMetadata of 14412_3 84\n")
    (wd / "species.txt").write_text("R6.fa.gz\tpneumo\nTIGR4.fa.gz\tpneumo\n")
    run(wd, "inverted", "build", "-o", "meta", "-k", "31", "-f", "rfile.txt", "--species-names", "species.txt",
        "--metadata", "metadata.txt")
    ski = ski_decode(wd / "meta.ski")
    assert ski["sample_names"][:2] == ["R6.fa.gz", "TIGR4.fa.gz"]
    assert ski["metadata"] == ["Metadata of " + {"R6.fa.gz": "R6", "TIGR4.fa.gz": "TIGR4",
                                                  "14412_3#82.contigs_velvet.fa.gz": "14412_3 82",
                                                  "14412_3#84.contigs_velvet.fa.gz": "14412_3 84"}[n]
                               for n in ski["sample_names"]]
    (wd / "dup.txt").write_text("R6.fa.gz\ta\nR6.fa.gz\tb\n")
    res = run(wd, "inverted", "build", "-o", "bad", "-k", "31", "-f", "rfile.txt", "--metadata", "dup.txt", ok=False)
    assert res.returncode == 101 and "duplicated" in res.stderr


def test_usage_and_errors(wd):
    assert run(wd, "inverted", ok=False).returncode == 2
    assert run(wd, "inverted", "query", "x.ski", ok=False).returncode == 2
    assert run(wd, "inverted", "build", "R6.fa.gz", ok=False).returncode == 2            # -o missing
    res = run(wd, "inverted", "precluster", "--count", "missing.ski", ok=False)
    assert res.returncode == 1 and "missing.ski" in res.stderr
    res = run(wd, "inverted", "precluster", "x.ski", "--retain-unmatched", "maybe", ok=False)
    assert res.returncode == 2 and "singleton, bruteforce" in res.stderr


def test_gpu_paths_refuse_without_a_device(wd, skl):
    """No silent CPU fallback: the commands whose compute runs on the device fail loudly on a
    box without one (the product has no CPU distance / candidate / hashing path behind them)."""
    if skl.device_count() > 0:
        pytest.skip("a GPU is present; the refusal path is only reachable without one")
    res = run(wd, "sketch", "--gpu", "-o", "x", "-k", "21", "R6.fa.gz", ok=False)
    assert res.returncode != 0 and "no CPU path" in res.stderr
    run(wd, "inverted", "build", "-o", "idx", "-k", "21", "-s", "10", "--write-skq", "R6.fa.gz", "TIGR4.fa.gz")
    run(wd, "sketch", "-o", "db", "-k", "21", "R6.fa.gz", "TIGR4.fa.gz")
    for extra in ((), ("--host-candidates",)):
        res = run(wd, "inverted", "precluster", "idx.ski", "--skd", "db", *extra, ok=False)
        assert res.returncode != 0 and "no CPU path" in res.stderr


def test_rmp_serde_byte_layout_is_read(tmp_path):
    """A `.ski` document typed out byte by byte from rmp-serde's encoding rules (no encoder of ours or
    of the msgpack package involved): 2 bins, 3 samples, metadata None, labels Some, HashType::DNA.
    Also the newtype-variant form of hash_type ({"AA": "Level2"}) and rmp-serde's struct-as-map form."""
    from test_fileformat_cpu import _py_frame

    def roaring(vals):   # one array container, portable format without run cookie
        return ((12346).to_bytes(4, "little") + (1).to_bytes(4, "little") + (0).to_bytes(2, "little") +
                (len(vals) - 1).to_bytes(2, "little") + (16).to_bytes(4, "little") +
                b"".join(v.to_bytes(2, "little") for v in vals))

    def binblob(b):
        return b"\xc4" + bytes([len(b)]) + b

    def fixstr(t):
        return bytes([0xa0 | len(t)]) + t.encode()

    def body(hash_type):
        return (b"\x92"                                                     # index: array(2)
                + b"\x82" + b"\x05" + binblob(roaring([0, 2])) + b"\xcd\x12\x34" + binblob(roaring([1]))   # {5: .., 0x1234: ..}
                + b"\x81" + b"\xcc\xc8" + binblob(roaring([0, 1, 2]))         # {200: ..}   (u8 form)
                + b"\x03"                                                   # n_samples
                + b"\x93" + fixstr("a") + fixstr("b") + fixstr("c")         # sample_names
                + b"\xc0"                                                   # metadata: None
                + b"\x93" + fixstr("x") + fixstr("x") + fixstr("y")         # labels: Some
                + b"\x15" + fixstr("0.3.0") + b"\xc3" + hash_type)          # kmer_size 21, version, rc true

    for name, ht, ok in [("dna", fixstr("DNA"), True), ("aa", b"\x81" + fixstr("AA") + fixstr("Level2"), True),
                         ("bad", b"\x07", False)]:
        (tmp_path / f"{name}.ski").write_bytes(_py_frame(b"\x99" + body(ht)))
        res = run(tmp_path, "inverted", "precluster", "--count", f"{name}.ski", ok=ok)
        if ok:
            assert res.stdout == "Identified 3 prefilter pairs from a max of 3\n", name
        else:
            assert res.returncode == 1 and "hash_type" in res.stderr
    # load -> save keeps the variant's encoding: {"AA": "Level2"} stays a one-entry map (rmp-serde's newtype variant),
    # "DNA" stays a string -- a flattened "AA(Level2)" string would be a file the reference cannot deserialize
    dbtool = os.path.join(ROOT, "sketchlib.rust_amd", "csrc", "_build", "skl_dbtool")
    for name, want in (("aa", {"AA": "Level2"}), ("dna", "DNA")):
        subprocess.check_call([dbtool, "reski", str(tmp_path / name), str(tmp_path / (name + "_again"))])
        payload = subprocess.check_output([dbtool, "unframe", str(tmp_path / (name + "_again.ski")), "/dev/stdout"])
        again = msgpack.unpackb(payload, raw=False, strict_map_key=False)
        first = msgpack.unpackb(b"\x99" + body(b"\x81" + fixstr("AA") + fixstr("Level2") if name == "aa" else fixstr("DNA")),
                                raw=False, strict_map_key=False)
        assert again[8] == want and again[:8] == first[:8], name
    # with_struct_map: the same nine values keyed by field name
    vals = msgpack.unpackb(b"\x99" + body(fixstr("DNA")), raw=False, strict_map_key=False)
    (tmp_path / "map.ski").write_bytes(_py_frame(msgpack.packb(dict(zip(FIELDS, vals)), use_bin_type=True)))
    assert run(tmp_path, "inverted", "precluster", "--count", "map.ski").stdout == \
        "Identified 3 prefilter pairs from a max of 3\n"
    # malformed: truncated document, container length beyond the input
    doc = b"\x99" + body(fixstr("DNA"))
    for bad in (doc[:-3], b"\x99\xdd\xff\xff\xff\xff", b"\x99\x92\x81\x05\xc6\xff\xff\xff\xff"):
        (tmp_path / "trunc.ski").write_bytes(_py_frame(bad))
        res = run(tmp_path, "inverted", "precluster", "--count", "trunc.ski", ok=False)
        assert res.returncode == 1 and "MessagePack" in res.stderr, res.stderr
