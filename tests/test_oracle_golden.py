"""Pins the CPU oracle against the reference's own goldens (no GPU needed).

Sources (reference tree): tests/test_results_correct/inverted_precluster{,_ani}.stdout
(k=21 self-kNN(1) over sketches3.skd, tests/inverted.rs:300-346),
sketchlib_output_true.txt `multiple_genomes` (tests/distance.rs:16-30,193-266, abs tol 0.05
after rounding to 3 dp), and the App. A table of SURVEY.md.
"""
import os

import numpy as np
import pytest

from conftest import REF_FIXTURES
from helpers import FIXTURE_NAMES, load_fixture_bins, rust_f32


def _knn_text(oracle, s, names, knn, dist_type, k_idx, ani):
    rows = oracle.self_dists_knn(s, knn, dist_type, k_idx, ani, ties=oracle.TIES_RUST_HEAP)
    lines = []
    for i in range(s.n):
        for item in rows[i]:
            col = names[int(item["idx"])]
            if dist_type == oracle.JACCARD:
                # distance_matrix.rs:379-381: padding rows are suppressed
                if item["d0"] < 1.0 or col != names[i]:
                    lines.append(f"{names[i]}\t{col}\t{rust_f32(item['d0'])}")
            else:
                lines.append(f"{names[i]}\t{col}\t{rust_f32(item['d0'])}\t{rust_f32(item['d1'])}")
    return "\n".join(lines) + "\n"


@pytest.mark.parametrize("ani,golden", [(False, "inverted_precluster.stdout"),
                                        (True, "inverted_precluster_ani.stdout")])
def test_sketches3_knn1_matches_reference_stdout(oracle, ani, golden):
    bins, n, kmers, ss64 = load_fixture_bins("sketches3")
    s = oracle.Sketches(bins, n, kmers, ss64)
    text = _knn_text(oracle, s, FIXTURE_NAMES, 1, oracle.JACCARD, 0, ani)
    expected = open(os.path.join(REF_FIXTURES, golden)).read()
    # the precluster golden is compared unordered in the reference (tests/inverted.rs:300-346)
    assert sorted(text.splitlines()) == sorted(expected.splitlines())


def test_sketches2_dense_jaccard_within_reference_tolerance(oracle):
    bins, n, kmers, ss64 = load_fixture_bins("sketches2")
    s = oracle.Sketches(bins, n, kmers, ss64)
    got = oracle.self_dists_all(s, oracle.JACCARD, 0).ravel()
    expected = [0.40755, 1.0, 1.0, 1.0, 1.0, 0.37049]  # multiple_genomes
    for a, e in zip(got, expected):
        assert abs(round(float(a), 3) - round(e, 3)) <= 0.05


APPENDIX_A = {
    # db: (samebits per condensed pair, 1-J as printed, ANI as printed)
    "sketches3": ([660, 0, 0, 0, 0, 722],
                  ["0.35546875", "1", "1", "1", "1", "0.29492188"],
                  ["0.98840284", "0", "0", "0", "0", "0.99095666"]),
    "sketches1": ([597, 0, 0, 0, 0, 678],
                  ["0.4169922", "1", "1", "1", "1", "0.33789062"],
                  ["0.9901376", "0", "0", "0", "0", "0.99266887"]),
    "sketches2": ([6104, 1, 0, 1, 0, 6325],
                  ["0.39251593", "0.99990046", "1", "0.99990046", "1", "0.3705215"],
                  ["0.9909693", "0.72509414", "0", "0.72509414", "0", "0.9916782"]),
}


@pytest.mark.parametrize("name", sorted(APPENDIX_A))
def test_appendix_a_known_answers(oracle, name):
    bins, n, kmers, ss64 = load_fixture_bins(name)
    s = oracle.Sketches(bins, n, kmers, ss64)
    bits, dist, ani = APPENDIX_A[name]
    assert oracle.self_binmatch(s).ravel().tolist() == bits
    assert [rust_f32(v) for v in oracle.self_dists_all(s, oracle.JACCARD, 0).ravel()] == dist
    assert [rust_f32(v) for v in oracle.self_dists_all(s, oracle.JACCARD, 0, True).ravel()] == ani


def test_legacy_db_coreacc(oracle):
    """alpha >= 0 => acc = 0 branch (jaccard.rs:138-140); derived value, SURVEY App. A."""
    bins, n, kmers, ss64 = load_fixture_bins("legacy_db")
    s = oracle.Sketches(bins, n, kmers, ss64)
    assert oracle.self_binmatch(s).ravel().tolist() == [98, 100, 82]
    core, acc = oracle.self_dists_all(s)[0]
    assert rust_f32(core) == "0.02203464" and rust_f32(acc) == "0"


def test_regression_edge_cases(oracle):
    # n < 3 -> (1, 1)   jaccard.rs:117
    assert oracle.regression(30.0, -2.0, -31.0, 500.0, 2.5, 2.0) == (1.0, 1.0)
    # -inf / NaN ysum -> (1, 1)
    assert oracle.regression(60.0, float("-inf"), 0.0, 1300.0, 0.0, 3.0) == (1.0, 1.0)
    assert oracle.regression(60.0, float("nan"), 0.0, 1300.0, 0.0, 3.0) == (1.0, 1.0)
    # all y equal: y_diff == 0 -> r = NaN -> both comparisons false -> (0, 0)   :127-140
    x = [15.0, 19.0, 23.0]
    y = [-0.5, -0.5, -0.5]
    args = (sum(x), sum(y), sum(a * b for a, b in zip(x, y)), sum(a * a for a in x),
            sum(b * b for b in y), 3.0)
    assert oracle.regression(*args) == (0.0, 0.0)


def test_index_helpers_roundtrip(oracle):
    L = oracle.lib()
    for n in (2, 3, 7, 100, 1001):
        k = 0
        for i in range(min(n - 1, 40)):
            for j in range(i + 1, n):
                idx = L.sko_square_to_condensed(i, j, n)
                assert L.sko_calc_row_idx(idx, n) == i
                assert L.sko_calc_col_idx(idx, i, n) == j
                if i == 0:
                    assert idx == k
                    k += 1


def test_completeness_formula(oracle):
    """tests/completeness.rs:468-592: J' = min(1, J / (c1c2/(c1+c2-c1c2))) when c1c2 >= cutoff."""
    L = oracle.lib()
    ss64, bits = 16, 660
    j = L.sko_jaccard_from_samebits(bits, ss64, 0, 0.0, 0.0, 0.64)
    assert j == pytest.approx(bits / 1024.0, abs=1e-15)
    c1, c2 = 0.9, 0.85
    expect = min(1.0, j / (c1 * c2 / (c1 + c2 - c1 * c2)))
    assert L.sko_jaccard_from_samebits(bits, ss64, 1, c1, c2, 0.64) == pytest.approx(expect, abs=1e-12)
    # below the cutoff: untouched
    assert L.sko_jaccard_from_samebits(bits, ss64, 1, 0.5, 0.5, 0.64) == j
    # capped at 1
    assert L.sko_jaccard_from_samebits(1024, ss64, 1, 0.8, 0.8, 0.64) == 1.0


def test_knn_tie_modes_agree_on_distances(oracle):
    from sketchlib.rust_amd import synth

    kmers, ss64, n = [17, 21, 25, 29], 8, 40
    s = oracle.Sketches(synth.set_r(n, kmers, ss64, n_clusters=4), n, kmers, ss64)
    a = oracle.self_dists_knn(s, 6, oracle.JACCARD, 1, False, ties=oracle.TIES_RUST_HEAP)
    b = oracle.self_dists_knn(s, 6, oracle.JACCARD, 1, False, ties=oracle.TIES_CANONICAL)
    assert np.array_equal(np.sort(a["d0"], axis=1), np.sort(b["d0"], axis=1))
    assert np.all(np.diff(b["d0"], axis=1) >= 0)
    # heap order is ascending too (into_sorted_vec)
    assert np.all(np.diff(a["d0"], axis=1) >= 0)


# ---------------------------------------------------------------------------
# The reference's exact-text goldens over the 4-genome, 4-k database
# (tests/distance.rs:270-328 knn_dists, :690-721 subset_dists).  The database is
# regenerated from the reference's FASTA by oracle/sketcher.py, itself pinned bit-exactly
# on sketches{1,2,3}.skd -- see tests/golden/make_generated_fixtures.py.
# ---------------------------------------------------------------------------
GEN = os.path.join(os.path.dirname(REF_FIXTURES), "generated")
K4 = [17, 21, 25, 29]


def _db4(oracle, subset=None):
    bins = np.fromfile(os.path.join(GEN, "sketch_db_4k.skd"), dtype="<u8").reshape(4, -1)
    names = list(FIXTURE_NAMES)
    if subset is not None:
        bins = bins[[names.index(s) for s in subset]]
        names = subset
    return oracle.Sketches(bins, len(names), K4, 157), names


@pytest.mark.parametrize("dist,golden", [("coreacc", "dists_knn_ca.stdout"),
                                         ("jaccard", "dists_knn_jaccard.stdout"),
                                         ("ani", "dists_knn_ani.stdout")])
def test_knn_goldens_exact_text(oracle, dist, golden):
    s, names = _db4(oracle)
    if dist == "coreacc":
        text = _knn_text(oracle, s, names, 1, oracle.COREACC, 0, False)
    else:
        text = _knn_text(oracle, s, names, 1, oracle.JACCARD, 1, dist == "ani")
    assert text == open(os.path.join(REF_FIXTURES, golden)).read()


def test_subset_golden_exact_text(oracle):
    subset = open(os.path.join(REF_FIXTURES, "subset.txt")).read().split()
    s, names = _db4(oracle, subset)
    d = oracle.self_dists_all(s)
    lines, x = [], 0
    for i in range(len(names)):
        for j in range(i + 1, len(names)):
            lines.append(f"{names[i]}\t{names[j]}\t{rust_f32(d[x][0])}\t{rust_f32(d[x][1])}")
            x += 1
    assert "\n".join(lines) + "\n" == open(os.path.join(REF_FIXTURES, "dists_subset.stdout")).read()


def test_db4_samebits_appendix_a(oracle):
    s, _ = _db4(oracle)
    bm = oracle.self_binmatch(s)
    assert bm[0].tolist() == [6811, 6547, 6310, 6127]      # (0,1)
    assert bm[5].tolist() == [7167, 6871, 6623, 6342]      # (2,3)
    assert bm[4].tolist() == [7, 1, 1, 3]                  # (1,3): regression on noise
    assert bm[1].tolist() == [12, 0, 1, 0]                 # (0,2): early break


def test_binary_heap_restatement_on_hand_worked_ties(oracle):
    """The BinaryHeap restatement (oracle/sketchlib_oracle.c) against vectors worked BY HAND from the algorithm Rust's std
    documents and publishes (alloc::collections::binary_heap: push = sift_up with `elt <= parent => stop`; pop = swap the
    last element into the root, sift_down_to_bottom -- always descend to the child chosen by `left <= right => right`, then
    sift_up; into_sorted_vec = repeated swap(0, end) + sift_down_range, which stops at `elt >= child`), driven by push_heap
    (mod.rs:41-48: push when the heap holds fewer than knn items or the key is STRICTLY below its maximum, then pop the
    maximum).  None of the expected lists below was produced by running this code.

    (a) knn = 3, five equal keys, ids 0..4.  Pushes 0, 1, 2 leave the array [0, 1, 2] (no sift: equal keys stop at once);
        3 and 4 are not strictly below the maximum.  into_sorted_vec: end = 2: swap(0, 2) -> [2, 1 | 0], sift_down_range(0, 2)
        has no child pair and `elt < child` fails on equal keys; end = 1: swap(0, 1) -> [1 | 2, 0].  Listed: 1, 2, 0.
    (b) knn = 3, keys .5 .5 .5 .2 .5 .5.  After 0, 1, 2: [0, 1, 2].  Candidate 3 (.2) is pushed to the end (its parent 1 is
        larger: no move) -> [0, 1, 2, 3]; pop: 3 goes to the root, the maximum 0 leaves; sift_down_to_bottom: children 1, 2
        equal -> right child 2 moves up, 3 lands at the bottom -> [2, 1, 3]; sift_up stops (.2 <= .5).  4 and 5 are not
        strictly below .5.  into_sorted_vec: swap(0, 2) -> [3, 1 | 2]; .2 < .5 -> [1, 3 | 2]; swap(0, 1) -> [3 | 1, 2].
        Listed: 3, 1, 2 -- of the equal keys it is 1 and 2 that survive, not 0 and 1.
    (c) knn = 3, keys .7 .5 .5 .5 .5 .5 .3.  [0, 1, 2]; candidate 3 replaces the maximum 0 -> [2, 1, 3]; 4, 5 rejected;
        candidate 6 (.3): pushed to the end, popped into the root, the maximum 2 leaves, right child 3 moves up ->
        [3, 1, 6].  into_sorted_vec: swap(0, 2) -> [6, 1 | 3], .3 < .5 -> [1, 6 | 3]; swap(0, 1) -> [6 | 1, 3].
        Listed: 6, 1, 3 (smallest (key, id) would be 6, 1, 2)."""
    cases = [
        ([0.5] * 5, 3, [1, 2, 0], [0.5, 0.5, 0.5]),
        ([0.5, 0.5, 0.5, 0.2, 0.5, 0.5], 3, [3, 1, 2], [0.2, 0.5, 0.5]),
        ([0.7, 0.5, 0.5, 0.5, 0.5, 0.5, 0.3], 3, [6, 1, 3], [0.3, 0.5, 0.5]),
    ]
    for keys, knn, ids, d0 in cases:
        got = oracle.heap_replay(np.array(keys, dtype=np.float32), knn)
        assert got["idx"].tolist() == ids, (keys, got["idx"].tolist())
        assert got["d0"].tolist() == np.array(d0, dtype=np.float32).tolist()
    # fewer candidates than knn: everything is kept, ascending
    got = oracle.heap_replay(np.array([0.9, 0.1], dtype=np.float32), 5)
    assert got["idx"].tolist() == [1, 0]
