"""`sketchlib sketch` (native C++ CPU sketcher) against the reference's
committed sketches: the `.skd` it writes must be byte-identical to sketches{1,2,3}.skd
(which the reference produced from the same FASTA files), the `.skm` must carry the same
per-sample metadata, and it must agree with the independent numpy restatement used to
build the 4-k fixture."""
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, REF_FIXTURES, ROOT
from helpers import FIXTURE_NAMES

BUILD = os.path.join(ROOT, "sketchlib.rust_amd", "csrc", "_build")
CLI = os.path.join(BUILD, "sketchlib")
DBTOOL = os.path.join(BUILD, "skl_dbtool")


@pytest.fixture(scope="module", autouse=True)
def _built(skl):
    assert os.path.exists(CLI)


def sketch(tmp_path, name, *args):
    out = str(tmp_path / name)
    subprocess.check_call([CLI, "sketch", "-o", out, *args], cwd=REF_FIXTURES)
    return out


@pytest.mark.parametrize("name,args", [
    ("sketches1", ["-k", "31", "-s", "1000", "-f", "rfile.txt"]),
    ("sketches3", ["--k-vals", "21", "-s", "1000", "-f", "rfile.txt", "--threads", "3"]),
    ("sketches2", ["-k", "31", "-s", "10000", *FIXTURE_NAMES]),      # positional files
])
def test_skd_byte_identical_to_reference(tmp_path, name, args):
    out = sketch(tmp_path, name, *args)
    assert open(out + ".skd", "rb").read() == open(os.path.join(REF_FIXTURES, name + ".skd"), "rb").read()


def test_skm_metadata_matches_reference(tmp_path):
    out = sketch(tmp_path, "meta", "-k", "31", "-s", "1000", "-f", "rfile.txt")

    def samples(prefix):
        txt = subprocess.check_output([DBTOOL, "info", prefix], text=True)
        return [l for l in txt.splitlines() if l.startswith("sample")], txt

    mine, info = samples(out)
    ref, ref_info = samples(os.path.join(REF_FIXTURES, "sketches1"))
    assert mine == ref            # name, index, seq_length, rc/reads/densified, acgt, non_acgt
    for key in ("sketch_size\t1024", "sketchsize64\t16", "kmer_lengths\t31", "kmer_stride\t224"):
        assert key in info and key in ref_info


def test_agrees_with_numpy_restatement_on_4k_database(tmp_path):
    """The reference's knn_dists database: --k-seq 17,31,4 -s 10000 (tests/distance.rs:281-290)."""
    out = sketch(tmp_path, "db4k", "--k-seq", "17,31,4", "-s", "10000", "-f", "rfile.txt", "--threads", "4")
    mine = np.fromfile(out + ".skd", dtype="<u8")
    committed = np.fromfile(os.path.join(GOLDEN, "generated", "sketch_db_4k.skd"), dtype="<u8")
    assert np.array_equal(mine, committed)


def test_single_strand_and_errors(tmp_path):
    a = sketch(tmp_path, "rc", "-k", "21", "-s", "1000", "R6.fa.gz")
    b = sketch(tmp_path, "ss", "-k", "21", "-s", "1000", "--single-strand", "R6.fa.gz")
    assert open(a + ".skd", "rb").read() != open(b + ".skd", "rb").read()
    res = subprocess.run([CLI, "sketch", "-o", str(tmp_path / "x"), "-k", "21", "missing.fa"],
                         capture_output=True, text=True, cwd=REF_FIXTURES)
    assert res.returncode == 101 and "Invalid path/file" in res.stderr
    res = subprocess.run([CLI, "sketch", "-o", str(tmp_path / "x"), "-k", "60", "short_sequence.fa"],
                         capture_output=True, text=True, cwd=REF_FIXTURES)
    assert res.returncode == 101 and "K-mer larger than smallest valid sequence" in res.stderr
    res = subprocess.run([CLI, "sketch", "-k", "21", "R6.fa.gz"], capture_output=True, text=True, cwd=REF_FIXTURES)
    assert res.returncode == 2 and "-o <OUTPUT>" in res.stderr
