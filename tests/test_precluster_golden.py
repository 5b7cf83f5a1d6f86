"""the oracle's restatement of `self_dists_knn_precluster`
(src/distances/mod.rs:399-553) and of the inverted-index sketch pinned on the reference's own
goldens (tests/inverted.rs:244-349): the `.skq` bytes written by
`sketchlib inverted build -k 21 -s 10 --write-skq`, the `--count` answer, and the
`inverted precluster --knn 1 [--ani]` / `--knn 50` stdout."""
import os

import numpy as np
import pytest

from conftest import REF_FIXTURES
from helpers import FIXTURE_NAMES, load_fixture_bins, rust_f32
from oracle import sketcher

FASTA = [os.path.join(REF_FIXTURES, n) for n in FIXTURE_NAMES]


@pytest.fixture(scope="module")
def skq():
    return sketcher.inverted_sketch_files(FASTA, 21, 10)


def test_skq_bytes_match_reference_golden(skq):
    golden = np.fromfile(os.path.join(REF_FIXTURES, "inverted.skq"), dtype="<u2").reshape(4, 10)
    assert np.array_equal(skq, golden)


def test_prefilter_pair_count(oracle, skq):
    assert oracle.prefilter_pair_count(skq) == 2          # "Identified 2 prefilter pairs from a max of 6"


def _text(rows, ani):
    lines = []
    for i, row in enumerate(rows):
        for item in row:
            j, d = int(item["idx"]), float(item["d0"])
            if d < 1.0 or j != i:                              # padding rows are not printed (distance_matrix.rs:379-381)
                lines.append(f"{FIXTURE_NAMES[i]}\t{FIXTURE_NAMES[j]}\t{rust_f32(d)}")
    return sorted(lines)


@pytest.mark.parametrize("ani,golden", [(False, "inverted_precluster.stdout"), (True, "inverted_precluster_ani.stdout")])
@pytest.mark.parametrize("knn", [1, 3])   # --knn 50 is clamped to n - 1 = 3 and prints the same lines
def test_precluster_stdout(oracle, skq, ani, golden, knn):
    bins, n, kmers, ss64 = load_fixture_bins("sketches3")     # `sketch --k-vals 21 -s 1000 -f rfile.txt`
    s = oracle.Sketches(bins, n, kmers, ss64)
    expected = sorted(open(os.path.join(REF_FIXTURES, golden)).read().splitlines())
    for ties in (oracle.TIES_RUST_HEAP, oracle.TIES_CANONICAL):
        rows = oracle.self_dists_knn_precluster(s, skq, knn, k_idx=0, ani=ani, ties=ties)
        assert _text(rows, ani) == expected


def test_retain_unmatched_and_reordered_index(oracle, skq):
    bins, n, kmers, ss64 = load_fixture_bins("sketches3")
    s = oracle.Sketches(bins, n, kmers, ss64)
    # an index whose sample order differs from the .skd (tests/inverted.rs:352-452): same answer
    perm = np.array([2, 0, 3, 1])                              # ski position of skd sample i
    skq_perm = np.empty_like(skq)
    skq_perm[perm] = skq
    a = oracle.self_dists_knn_precluster(s, skq, 2)
    b = oracle.self_dists_knn_precluster(s, skq_perm, 2, ski_of_skd=perm)
    assert np.array_equal(a, b)
    # a sample with no shared bin: nothing / singleton / brute force (mod.rs:487-527)
    lonely = skq.copy()
    lonely[1] = np.arange(10) + 60000
    none = oracle.self_dists_knn_precluster(s, lonely, 2)
    assert [(int(x["idx"]), float(x["d0"])) for x in none[1]] == [(1, 1.0), (1, 1.0)]
    single = oracle.self_dists_knn_precluster(s, lonely, 2, retain=oracle.RETAIN_SINGLETON)
    assert [(int(x["idx"]), float(x["d0"])) for x in single[1]] == [(1, 0.0), (1, 1.0)]
    brute = oracle.self_dists_knn_precluster(s, lonely, 2, retain=oracle.RETAIN_BRUTEFORCE)
    full = oracle.self_dists_knn(s, 2, oracle.JACCARD, 0)
    assert np.array_equal(brute[1], full[1])
