"""The reference's on-disk layout (.skm snappy-framed CBOR, .skd raw LE u64) through the
C++ host layer, without a GPU: parsing the committed fixtures (incl. the pre-0.2.0
legacy_db rule, multisketch.rs:96-100 / tests/sketch.rs:145), write -> read round trips,
and an independent Python decode of what the C++ writer emits."""
import os
import subprocess

import numpy as np
import pytest

from conftest import REF_FIXTURES, ROOT
from helpers import FIXTURE_DBS, FIXTURE_NAMES

BUILD = os.path.join(ROOT, "sketchlib.rust_amd", "csrc", "_build")
DBTOOL = os.path.join(BUILD, "skl_dbtool")
CLI = os.path.join(BUILD, "sketchlib")


@pytest.fixture(scope="module", autouse=True)
def _built(skl):
    assert os.path.exists(DBTOOL) and os.path.exists(CLI)


def _info(prefix):
    out = subprocess.check_output([DBTOOL, "info", prefix], text=True)
    d, samples = {}, []
    for line in out.splitlines():
        parts = line.split("\t")
        if parts[0] == "sample":
            samples.append(parts[1:])
        else:
            d[parts[0]] = parts[1]
    return d, samples


@pytest.mark.parametrize("name", sorted(FIXTURE_DBS))
def test_skm_fields(name):
    n, kmers, ss64 = FIXTURE_DBS[name]
    d, samples = _info(os.path.join(REF_FIXTURES, name))
    assert int(d["sketchsize64"]) == ss64
    assert int(d["sketch_size"]) == ss64 * 64          # legacy: sketch_size *= 64
    assert [int(k) for k in d["kmer_lengths"].split(",")] == kmers
    assert int(d["n_samples"]) == n == len(samples)
    assert int(d["kmer_stride"]) == ss64 * 14
    assert int(d["sample_stride"]) == ss64 * 14 * len(kmers)
    assert d["hash_type"] == "DNA"
    names = [s[1] for s in samples]
    assert names == (FIXTURE_NAMES if n == 4 else ["R6.fa.gz", "TIGR4.fa.gz"])
    assert [int(s[2]) for s in samples] == list(range(n))   # .skd block positions


def test_legacy_version_string():
    d, _ = _info(os.path.join(REF_FIXTURES, "legacy_db.skm"))  # prefix given with extension
    assert d["sketch_version"] == "0.1.3"


def test_get_sketch_slice_matches_skd_bytes():
    prefix = os.path.join(REF_FIXTURES, "legacy_db")
    raw = np.fromfile(prefix + ".skd", dtype="<u8")
    out = subprocess.check_output([DBTOOL, "slice", prefix, "1", "2"], text=True)
    words = np.array([int(x) for x in out.split()], dtype=np.uint64)
    kmer_stride, sample_stride = 28, 84
    assert np.array_equal(words, raw[1 * sample_stride + 2 * kmer_stride:][:kmer_stride])


def test_select_kmer_reads_the_one_slice_of_every_sample():
    """`dist -k` / precluster read one k-mer length: MultiSketch::select_kmer picks that slice out of each sample of the
    file (whole database and --subset blocks); it must be the bytes get_sketch_slice returns from the whole database."""
    prefix = os.path.join(REF_FIXTURES, "legacy_db")
    raw = np.fromfile(prefix + ".skd", dtype="<u8")
    kmer_stride, sample_stride = 28, 84
    names = ["R6.fa.gz", "TIGR4.fa.gz"] if raw.size == 2 * sample_stride else FIXTURE_NAMES
    for sample in range(len(names)):
        for k_idx in range(3):
            want = raw[sample * sample_stride + k_idx * kmer_stride:][:kmer_stride]
            for extra in ([], ["--no-mmap"]):      # through a mapping of the file / one positional read per slice
                out = subprocess.check_output([DBTOOL, "slice-selected", prefix, str(sample), str(k_idx), *extra], text=True)
                assert np.array_equal(np.array([int(x) for x in out.split()], dtype=np.uint64), want), (sample, k_idx, extra)
    # a subset in another order: logical sample 0 is the LAST sample of the file
    order = names[::-1]
    for k_idx in (0, 2):
        out = subprocess.check_output([DBTOOL, "slice-selected", prefix, "0", str(k_idx), *order], text=True)
        last = len(names) - 1
        assert np.array_equal(np.array([int(x) for x in out.split()], dtype=np.uint64),
                              raw[last * sample_stride + k_idx * kmer_stride:][:kmer_stride])
    bad = subprocess.run([DBTOOL, "slice-selected", prefix, "0", "3"], capture_output=True, text=True)
    assert bad.returncode != 0 and "no such k-mer length" in bad.stderr


def _py_unframe(data):
    """Independent check of the writer: snappy frame with uncompressed chunks + masked CRC32C."""
    import struct
    assert data[:10] == b"\xff\x06\x00\x00sNaPpY"
    i, out = 10, b""
    table = []
    for n in range(256):
        c = n
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        table.append(c)

    def crc32c(b):
        c = 0xFFFFFFFF
        for x in b:
            c = table[(c ^ x) & 0xFF] ^ (c >> 8)
        return c ^ 0xFFFFFFFF

    while i < len(data):
        t = data[i]
        ln = int.from_bytes(data[i + 1:i + 4], "little")
        body = data[i + 4:i + 4 + ln]
        i += 4 + ln
        assert t == 0x01, "writer emits uncompressed chunks"
        crc = struct.unpack("<I", body[:4])[0]
        payload = body[4:]
        c = crc32c(payload)
        assert crc == ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF
        out += payload
    return out


def _py_frame(raw):
    """The inverse: a snappy frame of uncompressed chunks (<= 65536 bytes each, as the format requires)."""
    table = []
    for n in range(256):
        c = n
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        table.append(c)

    def crc32c(b):
        c = 0xFFFFFFFF
        for x in b:
            c = table[(c ^ x) & 0xFF] ^ (c >> 8)
        return c ^ 0xFFFFFFFF

    framed = b"\xff\x06\x00\x00sNaPpY"
    for o in range(0, max(len(raw), 1), 60000):
        chunk = raw[o:o + 60000]
        c = crc32c(chunk)
        masked = ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF
        framed += b"\x01" + (len(chunk) + 4).to_bytes(3, "little") + masked.to_bytes(4, "little") + chunk
    return framed


@pytest.mark.parametrize("name", sorted(FIXTURE_DBS))
def test_roundtrip_write_read(tmp_path, name):
    src = os.path.join(REF_FIXTURES, name)
    dst = str(tmp_path / "copy")
    subprocess.check_call([DBTOOL, "roundtrip", src, dst])
    assert open(dst + ".skd", "rb").read() == open(src + ".skd", "rb").read()
    a, sa = _info(src)
    b, sb = _info(dst)
    assert a == b and sa == sb
    cbor = _py_unframe(open(dst + ".skm", "rb").read())
    assert cbor[0] >> 5 == 5 and b"sketch_metadata" in cbor and b"name_map" in cbor


def test_corrupt_skm_is_rejected(tmp_path):
    data = bytearray(open(os.path.join(REF_FIXTURES, "sketches1.skm"), "rb").read())
    data[40] ^= 0xFF
    bad = tmp_path / "bad.skm"
    bad.write_bytes(bytes(data))
    res = subprocess.run([DBTOOL, "info", str(bad)], capture_output=True, text=True)
    assert res.returncode == 1 and "Error:" in res.stderr


def test_cli_usage_errors():
    db = os.path.join(REF_FIXTURES, "sketches3")
    res = subprocess.run([CLI, "dist", db, "--ani"], capture_output=True, text=True)
    assert res.returncode == 2 and "-k <KMER>" in res.stderr       # --ani requires -k (cli.rs:211-213)
    res = subprocess.run([CLI, "dist", db, "--threads", "0"], capture_output=True, text=True)
    assert res.returncode == 2 and "Threads must be one or higher" in res.stderr   # cli.rs:63-73
    res = subprocess.run([CLI, "dist"], capture_output=True, text=True)
    assert res.returncode == 2 and "<REF_DB>" in res.stderr
    res = subprocess.run([CLI, "dist", "/nonexistent/db"], capture_output=True, text=True)
    assert res.returncode == 101 and "Could not read sketch metadata" in res.stderr   # lib.rs:322-323
    res = subprocess.run([CLI, "dist", db, "-k", "33"], capture_output=True, text=True)
    assert res.returncode == 101 and "K-mer size 33 not found in file" in res.stderr   # lib.rs:341-343


def test_cli_rejects_percent_completeness(tmp_path):
    """tests/distance.rs:619-623: percentages are an error mentioning [0.0, 1.0]."""
    comp = tmp_path / "comp.txt"
    comp.write_text("R6.fa.gz\t95.0\nTIGR4.fa.gz\t0.9\n")
    res = subprocess.run([CLI, "dist", os.path.join(REF_FIXTURES, "sketches3"), "-k", "21",
                          "--ref-completeness-file", str(comp)], capture_output=True, text=True)
    assert res.returncode == 1 and "[0.0, 1.0]" in res.stderr and "R6.fa.gz: 95" in res.stderr


def test_large_skm_parallel_frame_decode_and_corruption(tmp_path):
    """A .skm above the 8 MB threshold where the snappy frames are decoded (and CRC-checked) on
    several threads and the CBOR is pull-parsed: round trip, then one flipped payload byte must be
    caught by the checksum of its frame."""
    n, ss64 = 120_000, 1
    prefix = str(tmp_path / "big")
    np.zeros(n * ss64 * 14, dtype="<u8").tofile(prefix + ".skd")
    names = tmp_path / "names.txt"
    names.write_text("\n".join(f"sample_with_a_longer_name_{i:07d}.fa.gz" for i in range(n)))
    subprocess.run([DBTOOL, "make", prefix, str(ss64 * 64), "17,21", "@" + str(names)], check=True)
    assert os.path.getsize(prefix + ".skm") > (8 << 20)
    # the .skd written above holds 1 k-mer length per sample; make one that matches two
    np.zeros(n * ss64 * 14 * 2, dtype="<u8").tofile(prefix + ".skd")
    out = str(tmp_path / "copy")
    subprocess.run([DBTOOL, "roundtrip", prefix, out], check=True)
    assert open(prefix + ".skm", "rb").read() == open(out + ".skm", "rb").read()
    info = subprocess.run([DBTOOL, "info", out], check=True, capture_output=True, text=True).stdout
    assert f"n_samples\t{n}" in info or str(n) in info
    raw = bytearray(open(prefix + ".skm", "rb").read())
    raw[len(raw) // 2] ^= 0x40
    open(prefix + ".skm", "wb").write(raw)
    res = subprocess.run([DBTOOL, "info", prefix], capture_output=True, text=True)
    assert res.returncode != 0 and "checksum" in (res.stderr + res.stdout)


def _crc32c_masked(b):
    table = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        table.append(c)
    c = 0xFFFFFFFF
    for x in b:
        c = table[(c ^ x) & 0xFF] ^ (c >> 8)
    c ^= 0xFFFFFFFF
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def test_snappy_compressed_chunks_by_hand(tmp_path):
    """Raw snappy blocks written by hand (the writer here only emits uncompressed chunks): long
    literals, 1-, 2- and 4-byte-offset copies, an overlapping run (offset < length), a skippable
    chunk between them; then malformed blocks must be refused, not read out of bounds."""
    def chunk(kind, payload, raw):
        body = _crc32c_masked(raw).to_bytes(4, "little") + payload
        return bytes([kind]) + len(body).to_bytes(3, "little") + body

    want1 = b"ab" + b"ab" * 5                       # literal "ab", copy offset 2 length 10 (overlapping)
    blk1 = bytes([12, 0x04]) + b"ab" + bytes([((10 - 4) << 2) | 1, 2])
    lit = bytes(range(256)) * 2                     # 512-byte literal: 2-byte length form (tag 61)
    want2 = lit + lit[100:164] + lit[:40]
    blk2 = (bytes([0x80 | (len(want2) & 0x7F), len(want2) >> 7]) if len(want2) >= 128 else bytes([len(want2)]))
    blk2 += bytes([61 << 2]) + (len(lit) - 1).to_bytes(2, "little") + lit
    blk2 += bytes([((64 - 1) << 2) | 2]) + (512 - 100).to_bytes(2, "little")          # copy 64 from offset 412 back
    blk2 += bytes([((40 - 1) << 2) | 3]) + (512 + 64).to_bytes(4, "little")           # copy 40 from the very start
    stream = b"\xff\x06\x00\x00sNaPpY" + chunk(0x00, blk1, want1) + b"\x80\x03\x00\x00xyz" + chunk(0x00, blk2, want2) \
        + chunk(0x01, b"tail", b"tail")
    (tmp_path / "s.bin").write_bytes(stream)
    subprocess.run([DBTOOL, "unframe", str(tmp_path / "s.bin"), str(tmp_path / "s.out")], check=True)
    assert (tmp_path / "s.out").read_bytes() == want1 + want2 + b"tail"
    bad_blocks = [bytes([12, 0x04]) + b"ab" + bytes([((10 - 4) << 2) | 1, 3]),      # copy reaches before the block
                  bytes([12, 0x04]) + b"ab" + bytes([((11 - 4) << 2) | 1, 2]),      # one byte too many
                  bytes([12, 0x04]) + b"ab",                                        # too short
                  bytes([12, 61 << 2, 0xFF])]                                       # truncated literal length
    for bad in bad_blocks:
        (tmp_path / "b.bin").write_bytes(b"\xff\x06\x00\x00sNaPpY" + chunk(0x00, bad, want1))
        res = subprocess.run([DBTOOL, "unframe", str(tmp_path / "b.bin"), str(tmp_path / "b.out")], capture_output=True, text=True)
        assert res.returncode != 0 and "snappy" in res.stderr


def _frame(raw):
    out = b"\xff\x06\x00\x00sNaPpY"
    for i in range(0, max(len(raw), 1), 65536):
        part = raw[i:i + 65536]
        body = _crc32c_masked(part).to_bytes(4, "little") + part
        out += b"\x01" + len(body).to_bytes(3, "little") + body
    return out


def test_skm_pull_parser_accepts_what_cbor_allows_and_refuses_the_rest(tmp_path):
    """Hand-written .skm documents: fields in another order, unknown fields of every shape,
    indefinite-length maps / arrays / strings, a tag in front of a value; then a missing field, a
    wrong type and a truncated document must be errors, not crashes."""
    def head(major, v):
        if v < 24:
            return bytes([major << 5 | v])
        if v < 256:
            return bytes([major << 5 | 24, v])
        if v < 65536:
            return bytes([major << 5 | 25]) + v.to_bytes(2, "big")
        return bytes([major << 5 | 26]) + v.to_bytes(4, "big")

    def u(v): return head(0, v)
    def t(s): return head(3, len(s)) + s.encode()
    BREAK = b"\xff"
    sample = lambda name, idx: (b"\xbf" + t("reads") + b"\xf4" + t("name") + b"\x7f" + t(name[:3]) + t(name[3:]) + BREAK +
                                t("index") + (u(idx) if idx is not None else b"\xf6") + t("extra") + head(4, 2) + u(1) + head(5, 1) + t("x") + b"\xf5" +
                                t("rc") + b"\xf5" + t("seq_length") + u(1234) + t("densified") + b"\xf4" +
                                t("acgt") + b"\x9f" + u(1) + u(2) + u(3) + u(4) + BREAK + t("non_acgt") + u(7) + BREAK)
    fields = {
        "hash_type": t("DNA"), "sketch_version": t("0.3.0"), "sample_stride": u(28), "kmer_stride": u(14), "bin_stride": u(1),
        "name_map": b"\xbf" + t("genomeA") + u(0) + t("genomeB") + u(1) + BREAK,
        "sketch_metadata": b"\x9f" + sample("genomeA", 0) + sample("genomeB", None) + BREAK,
        "kmer_lengths": head(4, 2) + b"\xc1" + u(17) + u(21),       # a tag in front of the first length
        "unknown_blob": head(2, 5) + b"hello", "sketchsize64": u(1), "sketch_size": u(64),
    }
    def doc(keys): return b"\xbf" + b"".join(t(k) + fields[k] for k in keys) + BREAK
    order = list(fields)
    prefix = str(tmp_path / "hand")
    np.zeros(2 * 28, dtype="<u8").tofile(prefix + ".skd")
    (tmp_path / "hand.skm").write_bytes(_frame(doc(order)))
    info = subprocess.run([DBTOOL, "info", prefix], check=True, capture_output=True, text=True).stdout
    assert "kmer_lengths\t17,21" in info and "n_samples\t2" in info and "hash_type\tDNA" in info
    assert "sample\t0\tgenomeA\t0\t1234\t100\t1,2,3,4\t7" in info and "sample\t1\tgenomeB\t-1\t1234" in info
    for label, raw in (("missing field", doc([k for k in order if k != "kmer_stride"])),
                       ("wrong type", doc(order).replace(t("sample_stride") + u(28), t("sample_stride") + t("28"))),
                       ("truncated", doc(order)[:-40])):
        (tmp_path / "hand.skm").write_bytes(_frame(raw))
        res = subprocess.run([DBTOOL, "info", prefix], capture_output=True, text=True)
        assert res.returncode != 0 and res.returncode > 0, label      # an error exit, not a signal


def test_untrusted_files_are_refused_cleanly(tmp_path):
    """Findings of a mutation fuzz of the loaders under AddressSanitizer, pinned: an
    indefinite-length string that never ends, a string longer than the file, an index whose
    n_samples disagrees with its names or whose bitmaps name samples that do not exist."""
    prefix = str(tmp_path / "x")
    np.zeros(28, dtype="<u8").tofile(prefix + ".skd")
    for raw in (b"\xbf\x7f\x61a",                                   # {_ (_ "a" ...  and the input ends
                b"\xa1\x7b\xff\xff\xff\xff\xff\xff\xff\xf0abc",     # text of 2^64 - 16 bytes
                b"\xa1\x69n_samples\x9f\x9f\x9f"):                  # nested arrays cut short
        (tmp_path / "x.skm").write_bytes(_frame(raw))
        res = subprocess.run([DBTOOL, "info", prefix], capture_output=True, text=True)
        assert res.returncode > 0, raw
    # name_map values index the completeness vectors (io.cpp): an out-of-range one used to write 400 MB
    # past a 2-element array in `dist --ref-completeness-file` (exit 139); now the .skm is refused
    import shutil
    for ext in (".skm", ".skd"):
        shutil.copy(os.path.join(REF_FIXTURES, "legacy_db" + ext), tmp_path / ("nm" + ext))
    subprocess.run([DBTOOL, "unframe", str(tmp_path / "nm.skm"), str(tmp_path / "nm.raw")], check=True)
    rawm = (tmp_path / "nm.raw").read_bytes()
    key = b"\x6bTIGR4.fa.gz\x01"                 # name_map entry "TIGR4.fa.gz": 1
    assert rawm.count(key) == 1
    (tmp_path / "nm.skm").write_bytes(_frame(rawm.replace(key, b"\x6bTIGR4.fa.gz\x1a\x02\xfa\xf0\x80")))   # 50 000 000
    (tmp_path / "comp.txt").write_text("TIGR4.fa.gz\t0.9\nR6.fa.gz\t0.8\n")
    for cmd in ([DBTOOL, "info", str(tmp_path / "nm")],
                [CLI, "dist", str(tmp_path / "nm"), "--ref-completeness-file", str(tmp_path / "comp.txt")]):
        res = subprocess.run(cmd, capture_output=True, text=True)
        # (the CLI reports it as the reference does a .skm it cannot read: a panic, exit 101)
        assert res.returncode > 0 and ("name_map index out of range" in res.stderr or
                                       (res.returncode == 101 and "Could not read sketch metadata" in res.stderr)), \
            (cmd, res.returncode, res.stderr)
    # a chunk that claims more than the format's 65536 uncompressed bytes (here 3 GiB from 25 bytes)
    bomb = b"\xff\x06\x00\x00sNaPpY" + b"\x00" + (4 + 7).to_bytes(3, "little") + b"\x00\x00\x00\x00" + \
        b"\x80\x80\x80\x80\x0c" + b"\x00\x61"
    (tmp_path / "nm.skm").write_bytes(bomb)
    res = subprocess.run([DBTOOL, "info", str(tmp_path / "nm")], capture_output=True, text=True)
    assert res.returncode > 0 and "65536" in res.stderr, res.stderr
    # a real index, then its n_samples / a bitmap entry patched in the (re-framed) payload
    wd = tmp_path / "idx"
    wd.mkdir()
    for f in FIXTURE_NAMES:
        shutil.copy(os.path.join(REF_FIXTURES, f), wd / f)
    subprocess.run([CLI, "inverted", "build", "-o", "i", "-k", "21", "-s", "20", *FIXTURE_NAMES], cwd=wd, check=True)
    subprocess.run([DBTOOL, "unframe", str(wd / "i.ski"), str(wd / "i.raw")], check=True)
    raw = (wd / "i.raw").read_bytes()
    # MessagePack array of struct Inverted: n_samples (4) sits between the index and the 4 names
    key = b"\x04\x94\xbf14412_3#82"
    assert raw[0] == 0x99 and raw.count(key) == 1
    ok = subprocess.run([CLI, "inverted", "precluster", "i.ski", "--count"], cwd=wd, capture_output=True, text=True)
    assert ok.returncode == 0 and "prefilter pairs" in ok.stdout
    (wd / "bad.ski").write_bytes(_frame(raw.replace(key, b"\x05" + key[1:])))
    res = subprocess.run([CLI, "inverted", "precluster", "bad.ski", "--count"], cwd=wd, capture_output=True, text=True)
    assert res.returncode > 0 and "n_samples" in res.stderr
    # roaring array container of one value [cookie 12346, 1 container, key 0, card-1 = 0, offset, value]:
    one = (12346).to_bytes(4, "little") + (1).to_bytes(4, "little") + b"\x00\x00\x00\x00" + (16).to_bytes(4, "little")
    hits = [i for i in range(len(raw)) if raw.startswith(one, i)]
    assert hits
    at = hits[0] + len(one)
    patched = raw[:at] + (9).to_bytes(2, "little") + raw[at + 2:]   # sample id 9 of 4
    (wd / "bad.ski").write_bytes(_frame(patched))
    res = subprocess.run([CLI, "inverted", "precluster", "bad.ski", "--count"], cwd=wd, capture_output=True, text=True)
    assert res.returncode > 0 and "invalid sample id" in res.stderr


def test_short_skd_and_wrong_strides_are_refused(tmp_path):
    import shutil
    for ext in (".skm", ".skd"):
        shutil.copy(os.path.join(REF_FIXTURES, "sketches1" + ext), tmp_path / ("s" + ext))
    data = (tmp_path / "s.skd").read_bytes()
    (tmp_path / "s.skd").write_bytes(data[:len(data) - 8 * 14])
    res = subprocess.run([DBTOOL, "roundtrip", str(tmp_path / "s"), str(tmp_path / "t")], capture_output=True, text=True)
    assert res.returncode > 0 and "shorter than its metadata" in res.stderr
    res = subprocess.run([CLI, "dist", str(tmp_path / "s")], capture_output=True, text=True)
    assert res.returncode > 0 and "shorter than its metadata" in res.stderr      # before any device is touched
    subprocess.run([DBTOOL, "unframe", str(tmp_path / "s.skm"), str(tmp_path / "s.raw")], check=True)
    raw = (tmp_path / "s.raw").read_bytes()
    key = b"\x6bkmer_stride\x18\xe0"          # 16 * 14 = 224
    assert raw.count(key) == 1
    (tmp_path / "s.skm").write_bytes(_frame(raw.replace(key, b"\x6bkmer_stride\x18\xe1")))
    res = subprocess.run([DBTOOL, "info", str(tmp_path / "s")], capture_output=True, text=True)
    assert res.returncode > 0 and "strides" in res.stderr


@pytest.mark.parametrize("name", ["sketches1", "sketches2", "sketches3"])
def test_skm_writer_emits_the_reference_cbor(tmp_path, name):
    """What our .skm writer puts inside the snappy frame is, byte for byte, the CBOR document that
    ciborium wrote into the reference's own file (multisketch.rs:80-88): same field order, same
    shortest-form integers, definite lengths, name_map in the file's order.  (The frames differ: the
    reference's chunks are snappy-compressed, ours are stored -- type 0x01 -- which every framing-format
    reader, `snap::read::FrameDecoder` included, must accept.)  The `.skm` the native sketcher writes
    from the genomes decodes to the same document up to the order of name_map, a HashMap."""
    src = os.path.join(REF_FIXTURES, name)
    subprocess.check_call([DBTOOL, "unframe", src + ".skm", str(tmp_path / "ref.raw")])
    subprocess.check_call([DBTOOL, "roundtrip", src, str(tmp_path / "copy")])
    ours = _py_unframe((tmp_path / "copy.skm").read_bytes())
    assert ours == (tmp_path / "ref.raw").read_bytes()
    from test_inverted_cli_cpu import cbor_decode
    doc, end = cbor_decode(ours)
    assert end == len(ours) and list(doc)[:3] == ["sketch_size", "sketchsize64", "kmer_lengths"]
    k, size = {"sketches1": ("31", "1000"), "sketches2": ("31", "10000"), "sketches3": ("21", "1000")}[name]
    subprocess.check_call([CLI, "sketch", "-o", str(tmp_path / "fresh"), "-k", k, "-s", size, *FIXTURE_NAMES], cwd=REF_FIXTURES,
                          stderr=subprocess.DEVNULL)
    fresh, _ = cbor_decode(_py_unframe((tmp_path / "fresh.skm").read_bytes()))
    assert list(fresh) == list(doc)
    for key in doc:
        if key == "sketch_version":
            continue
        assert fresh[key] == doc[key], key        # dict comparison ignores name_map's order
    assert (tmp_path / "fresh.skd").read_bytes() == open(src + ".skd", "rb").read()
