"""GPU sketching: skl_sketch_signs -- rolling canonical ntHash, `% SIGN_MOD`
and bin minima on the device -- against the oracle's non-rolling numpy restatement
(oracle/sketcher.py, itself pinned bit-exactly on the reference's sketches{1,2,3}.skd), and
`sketchlib sketch --gpu` against the reference's committed `.skd` files byte for byte."""
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, REF_FIXTURES, ROOT
from helpers import FIXTURE_NAMES
from oracle import sketcher

pytestmark = pytest.mark.gpu
CLI = os.path.join(ROOT, "sketchlib.rust_amd", "csrc", "_build", "sketchlib")
U64_MAX = np.uint64(0xFFFFFFFFFFFFFFFF)


def oracle_signs(codes, offsets, k, num_bins, rc):
    h = sketcher.kmer_hashes(codes, offsets, k, rc) % np.uint64(sketcher.SIGN_MOD)
    bin_size = -(-sketcher.SIGN_MOD // num_bins)
    signs = np.full(num_bins, U64_MAX, dtype=np.uint64)
    if h.size:
        np.minimum.at(signs, (h // np.uint64(bin_size)).astype(np.int64), h)
    return signs


def pack(samples):
    codes = np.concatenate([c for c, _ in samples]) if samples else np.zeros(0, np.uint8)
    cb = np.cumsum([0] + [len(c) for c, _ in samples])
    offs = np.concatenate([o for _, o in samples]) if samples else np.zeros(0, np.int64)
    ob = np.cumsum([0] + [len(o) for _, o in samples])
    return codes, cb, offs, ob


@pytest.mark.parametrize("rc", [True, False])
def test_signs_of_reference_genomes(skl, gpu_ctx, rc):
    samples = [sketcher.read_fasta_bases(os.path.join(REF_FIXTURES, n)) for n in FIXTURE_NAMES[:2]]
    kmers, num_bins = [17, 21, 31], 1024
    got = skl.sketch_signs(gpu_ctx, *pack(samples), kmers, num_bins, rc)
    assert "nthash_binmin_lds_kernel" in gpu_ctx.last_kernel()
    for s, (codes, offsets) in enumerate(samples):
        for ki, k in enumerate(kmers):
            assert np.array_equal(got[s, ki], oracle_signs(codes, offsets, k, num_bins, rc)), (s, k)


def test_signs_of_synthetic_sequences_with_breaks(skl, gpu_ctx):
    """Ns, records shorter than k, a record of exactly k, breaks at span boundaries (multiples of
    256), a sample with no valid window at the largest k, an odd bin count."""
    rng = np.random.default_rng(5)

    def sample(lengths, n_frac):
        codes, offsets, pos = [], [], 0
        for ln in lengths:
            seq = rng.integers(0, 4, size=ln, dtype=np.uint8)
            invalid = rng.random(ln) < n_frac
            keep = ~invalid
            before = np.cumsum(keep) - keep
            offsets.append(pos + before[invalid])
            codes.append(seq[keep])
            pos += int(keep.sum())
            offsets.append(np.array([pos]))
        return np.concatenate(codes).astype(np.uint8), np.concatenate(offsets).astype(np.int64)

    samples = [
        sample([5000, 12, 31, 700, 256, 512, 3], 0.002),
        sample([256] * 9, 0.0),                               # breaks exactly at span boundaries
        sample([40, 25], 0.0),                                # nothing for k = 61
        sample([100000], 0.0005),
        sample([1, 2, 3, 64000], 0.01),
    ]
    kmers = [3, 15, 31, 61]
    for num_bins in (64, 1000, 4096):
        got = skl.sketch_signs(gpu_ctx, *pack(samples), kmers, num_bins, True)
        for s, (codes, offsets) in enumerate(samples):
            for ki, k in enumerate(kmers):
                assert np.array_equal(got[s, ki], oracle_signs(codes, offsets, k, num_bins, True)), (num_bins, s, k)
    assert (got[2, 3] == U64_MAX).all()


@pytest.mark.parametrize("name,args", [
    ("sketches1", ["-k", "31", "-s", "1000", "-f", "rfile.txt"]),
    ("sketches3", ["--k-vals", "21", "-s", "1000", "-f", "rfile.txt", "--threads", "3"]),
    ("sketches2", ["-k", "31", "-s", "10000", *FIXTURE_NAMES]),
])
def test_cli_gpu_sketch_is_byte_identical_to_reference(gpu_ctx, tmp_path, name, args):
    out = str(tmp_path / name)
    subprocess.check_call([CLI, "sketch", "--gpu", "-o", out, *args], cwd=REF_FIXTURES)
    assert open(out + ".skd", "rb").read() == open(os.path.join(REF_FIXTURES, name + ".skd"), "rb").read()
    cpu = str(tmp_path / (name + "_cpu"))
    subprocess.check_call([CLI, "sketch", "-o", cpu, *args], cwd=REF_FIXTURES)
    dbtool = os.path.join(os.path.dirname(CLI), "skl_dbtool")
    assert subprocess.check_output([dbtool, "info", out]) == subprocess.check_output([dbtool, "info", cpu])


def test_cli_gpu_sketch_4k_database_and_errors(gpu_ctx, tmp_path):
    out = str(tmp_path / "db4k")
    subprocess.check_call([CLI, "sketch", "--gpu", "-o", out, "--k-seq", "17,31,4", "-s", "10000", "-f", "rfile.txt",
                           "--threads", "4"], cwd=REF_FIXTURES)
    assert np.array_equal(np.fromfile(out + ".skd", dtype="<u8"),
                          np.fromfile(os.path.join(GOLDEN, "generated", "sketch_db_4k.skd"), dtype="<u8"))
    res = subprocess.run([CLI, "sketch", "--gpu", "-o", str(tmp_path / "x"), "-k", "60", "short_sequence.fa"],
                         cwd=REF_FIXTURES, capture_output=True, text=True)
    cpu = subprocess.run([CLI, "sketch", "-o", str(tmp_path / "y"), "-k", "60", "short_sequence.fa"],
                         cwd=REF_FIXTURES, capture_output=True, text=True)
    assert res.returncode == cpu.returncode == 101 and "K-mer larger than smallest valid sequence" in res.stderr


@pytest.mark.ab_library
@pytest.mark.parametrize("num_bins", [1000, 5000])
def test_both_kernel_forms_and_bin_counts(skl, gpu_ctx, monkeypatch, num_bins):
    """The LDS-staged kernel with bin minima in LDS (<= 4096 bins) and in global memory (more),
    the global-memory kernel (SKL_SKETCH_KERNEL=global, also what k > 129 takes), many short
    samples (workgroups padded per sample) and k at the staged kernel's limit."""
    rng = np.random.default_rng(11)
    samples = []
    for ln in (40, 129, 130, 5000, 128 * 256 + 77, 3 * 128 * 256 + 5):
        codes = rng.integers(0, 4, size=ln, dtype=np.uint8)
        cuts = np.sort(rng.choice(np.arange(1, ln), size=min(6, ln // 20), replace=False)).astype(np.int64)
        samples.append((codes, cuts))
    kmers = [11, 31, 129]
    got = skl.sketch_signs(gpu_ctx, *pack(samples), kmers, num_bins, True)
    assert "lds_kernel" in gpu_ctx.last_kernel()
    monkeypatch.setenv("SKL_SKETCH_KERNEL", "global")
    gpu_ctx.reload_env()
    ref = skl.sketch_signs(gpu_ctx, *pack(samples), kmers, num_bins, True)
    assert "lds_kernel" not in gpu_ctx.last_kernel()
    assert np.array_equal(got, ref)
    monkeypatch.delenv("SKL_SKETCH_KERNEL")
    gpu_ctx.reload_env()
    big_k = skl.sketch_signs(gpu_ctx, *pack(samples), [130], num_bins, True)      # past the staged limit
    assert "lds_kernel" not in gpu_ctx.last_kernel()
    for s_, (codes, offsets) in enumerate(samples):
        assert np.array_equal(got[s_, 1], oracle_signs(codes, offsets, 31, num_bins, True)), s_
        assert np.array_equal(got[s_, 2], oracle_signs(codes, offsets, 129, num_bins, True)), s_
        assert np.array_equal(big_k[s_, 0], oracle_signs(codes, offsets, 130, num_bins, True)), s_


def test_packed_form_and_several_batches(skl, gpu_ctx):
    """skl_sketch_signs_packed (bases at 2 bits each, every sample on a word boundary, ragged last words) equals the
    one-byte form, which the library packs itself -- over enough bases for several upload batches (8 Mi words each), so
    that batch i + 1's upload really runs under batch i's kernel and the pinned two-slot ring is reused."""
    rng = np.random.default_rng(23)
    lengths = [45_000_007, 31, 16, 17, 38_000_001, 52_000_003, 5, 44_000_000, 41_234_567, 12_345]     # ~220 M bases: several batches
    samples = []
    for ln in lengths:
        codes = rng.integers(0, 4, size=ln, dtype=np.uint8)
        cuts = np.sort(rng.choice(np.arange(1, max(ln, 2)), size=min(5, max(ln // 3, 0)), replace=False)).astype(np.int64) if ln > 3 else np.zeros(0, np.int64)
        samples.append((codes, cuts))
    kmers, num_bins = [15, 31], 4096
    codes, cb, offs, ob = pack(samples)
    byte_form = skl.sketch_signs(gpu_ctx, codes, cb, offs, ob, kmers, num_bins, True)
    packed = skl.pack_codes(codes, cb)
    assert packed.size == sum((ln + 15) // 16 for ln in lengths)
    packed_form = skl.sketch_signs_packed(gpu_ctx, packed, cb, offs, ob, kmers, num_bins, True)
    assert np.array_equal(byte_form, packed_form)
    again = skl.sketch_signs(gpu_ctx, codes, cb, offs, ob, kmers, num_bins, True)      # (buffers and the ring reused)
    assert np.array_equal(again, byte_form)
    for s_ in (1, 2, 3, 6, 9):      # the short samples against the oracle sketcher
        for ki, k in enumerate(kmers):
            assert np.array_equal(byte_form[s_, ki], oracle_signs(*samples[s_], k, num_bins, True)), (s_, k)
    # a long one: its first megabase on its own must give bin minima >= the whole sample's (a subset of its windows)
    head = (samples[0][0][:1_000_000], samples[0][1][samples[0][1] < 1_000_000])
    part = oracle_signs(*head, 15, num_bins, True)
    assert np.all(byte_form[0, 0] <= part)
