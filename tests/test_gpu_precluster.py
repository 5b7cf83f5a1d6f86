"""Candidate-list kNN on the GPU (skl_self_dists_knn_candidates: the device half of the
reference's self_dists_knn_precluster, src/distances/mod.rs:399-553) against the oracle's
restatement: reference fixtures, synthetic clustered databases with ragged candidate lists
(empty rows, rows shorter and much longer than knn, a reordered index), Jaccard and ANI."""
import os

import numpy as np
import pytest

from conftest import REF_FIXTURES
from helpers import FIXTURE_NAMES, load_fixture_bins
from sketchlib.rust_amd import synth

pytestmark = pytest.mark.gpu


def candidates(skq):
    """CSR of `any_shared_bins` (inverted.rs:259-268) per row, self excluded, ascending ids."""
    n = skq.shape[0]
    offs, cols = [0], []
    for i in range(n):
        shared = (skq == skq[i]).any(axis=1)
        shared[i] = False
        j = np.nonzero(shared)[0]
        cols.append(j)
        offs.append(offs[-1] + j.size)
    return np.array(offs, dtype=np.uint64), (np.concatenate(cols) if cols else np.zeros(0)).astype(np.uint32)


def as_pairs(idx, d0):
    return [[(int(a), float(b)) for a, b in zip(r_i, r_d)] for r_i, r_d in zip(idx, d0)]


def oracle_pairs(rows):
    return [[(int(x["idx"]), float(x["d0"])) for x in r] for r in rows]


@pytest.mark.parametrize("ani", [False, True])
@pytest.mark.parametrize("knn", [1, 3])
def test_reference_fixture(oracle, skl, gpu_ctx, ani, knn):
    from oracle import sketcher
    skq = sketcher.inverted_sketch_files([os.path.join(REF_FIXTURES, n) for n in FIXTURE_NAMES], 21, 10)
    bins, n, kmers, ss64 = load_fixture_bins("sketches3")
    o, g = oracle.Sketches(bins, n, kmers, ss64), gpu_ctx.sketches(bins, n, kmers, ss64)
    offs, cols = candidates(skq)
    idx, d0 = skl.self_dists_knn_candidates(gpu_ctx, g, g.set_k(21, ani), knn, offs, cols)
    assert "pair_cand_rows_kernel" in gpu_ctx.last_kernel()
    assert as_pairs(idx, d0) == oracle_pairs(oracle.self_dists_knn_precluster(o, skq, knn, 0, ani))


@pytest.mark.parametrize("ani,comp", [(False, False), (True, False), (False, True)])
def test_synthetic_ragged_lists(oracle, skl, gpu_ctx, ani, comp):
    kmers, ss64, n = [17, 21, 25], 16, 700
    bins = synth.set_r(n, kmers, ss64, n_clusters=23)
    rng = np.random.default_rng(11)
    # an index sketch of 12 bins: cluster members share most bins, singletons share none
    cluster = np.arange(n) % 5                                 # ~140 candidates per row
    parents = rng.integers(0, 60000, size=(5, 12), dtype=np.uint16)
    skq = parents[cluster].copy()
    mutate = rng.random(skq.shape) < 0.35
    skq[mutate] = rng.integers(0, 65535, size=int(mutate.sum()), dtype=np.uint16)
    skq[5] = 65535 - np.arange(12, dtype=np.uint16)            # no shared bin at all
    skq[::97, :] = skq[1]                                      # exact copies of one index sketch
    skq[10:13] = 61000 + np.arange(12, dtype=np.uint16)        # a triple that only sees itself
    completeness = np.linspace(0.75, 1.0, n) if comp else None
    o = oracle.Sketches(bins, n, kmers, ss64, completeness=completeness)
    g = gpu_ctx.sketches(bins, n, kmers, ss64, completeness=completeness)
    offs, cols = candidates(skq)
    lens = np.diff(offs.astype(np.int64))
    assert lens.min() == 0 and lens.max() > 64 and (lens < 5).any()
    for knn in (1, 5, 40):
        idx, d0 = skl.self_dists_knn_candidates(gpu_ctx, g, g.set_k(21, ani), knn, offs, cols)
        # the same with the candidate lists built on the device from the index sketches
        idx2, d02, total = skl.self_dists_knn_shared_bins(gpu_ctx, g, g.set_k(21, ani), knn, skq)
        assert total == cols.size and np.array_equal(idx2, idx) and np.array_equal(d02, d0)
        exp = oracle.self_dists_knn_precluster(o, skq, knn, 1, ani, threads=8)
        if comp:
            assert np.array_equal(idx, exp["idx"].astype(np.uint64))
            np.testing.assert_allclose(d0, exp["d0"], atol=1e-6, rtol=0)
        else:
            assert as_pairs(idx, d0) == oracle_pairs(exp)


@pytest.mark.ab_library
@pytest.mark.parametrize("ss64", [1, 3, 16, 24, 32, 50, 64, 80, 128, 157, 170])
def test_candidate_kernels_at_many_sketch_sizes(oracle, skl, gpu_ctx, set_switch, ss64):
    """pair_cand_rows_kernel (lanes across the sketch: half chunks per lane; the row's planes stay in registers up to 32
    chunks, several trips beyond, idle lanes below) and round 3's pair_cand_kernel (SKL_CAND_KERNEL=lanes) against the
    oracle and each other, host lists (every pair evaluated) and device lists (symmetric halves)."""
    kmers, n, knn = [17, 21], 300, 6
    bins = synth.set_r(n, kmers, ss64, n_clusters=11)
    rng = np.random.default_rng(ss64)
    skq = rng.integers(0, 25, size=(n, 4), dtype=np.uint16)
    o, g = oracle.Sketches(bins, n, kmers, ss64), gpu_ctx.sketches(bins, n, kmers, ss64)
    offs, cols = candidates(skq)
    exp = oracle.self_dists_knn_precluster(o, skq, knn, 1, False, threads=8)
    got = {}
    for kernel in (None, "lanes"):
        set_switch("SKL_CAND_KERNEL", kernel)
        idx, d0 = skl.self_dists_knn_candidates(gpu_ctx, g, g.set_k(21), knn, offs, cols)
        assert ("pair_cand_kernel" if kernel else "pair_cand_rows_kernel") in gpu_ctx.last_kernel()
        assert as_pairs(idx, d0) == oracle_pairs(exp), (ss64, kernel)
        idx2, d02, _ = skl.self_dists_knn_shared_bins(gpu_ctx, g, g.set_k(21), knn, skq)
        assert np.array_equal(idx2, idx) and np.array_equal(d02, d0), (ss64, kernel)
        got[kernel] = (idx, d0)
    g.close()


def test_reordered_index_and_errors(oracle, skl, gpu_ctx):
    kmers, ss64, n = [21], 8, 130
    bins = synth.set_r(n, kmers, ss64, n_clusters=7)
    rng = np.random.default_rng(3)
    skq = rng.integers(0, 50, size=(n, 6), dtype=np.uint16)     # small alphabet: many shared bins
    o, g = oracle.Sketches(bins, n, kmers, ss64), gpu_ctx.sketches(bins, n, kmers, ss64)
    perm = rng.permutation(n)                                   # ski position of skd sample i
    skq_ski = np.empty_like(skq)
    skq_ski[perm] = skq
    offs, cols = candidates(skq)                                # the host driver maps ski ids back to skd ids
    idx, d0 = skl.self_dists_knn_candidates(gpu_ctx, g, g.set_k(21), 9, offs, cols)
    exp = oracle.self_dists_knn_precluster(o, skq_ski, 9, 0, ski_of_skd=perm, threads=4)
    assert as_pairs(idx, d0) == oracle_pairs(exp)
    with pytest.raises(skl.SklError) as e:                       # the reference's CoreAcc arm is unimplemented!()
        skl.self_dists_knn_candidates(gpu_ctx, gpu_ctx.sketches(synth.set_r(4, [15, 19], 2), 4, [15, 19], 2),
                                      skl.params(), 1, np.zeros(5, dtype=np.uint64), np.zeros(0, dtype=np.uint32))
    assert "single k-mer" in str(e.value)


@pytest.fixture()
def ref_ties(skl, gpu_ctx):
    gpu_ctx.set_knn_ties(skl.TIES_REFERENCE)
    yield
    gpu_ctx.set_knn_ties(skl.TIES_CANONICAL)


@pytest.mark.ab_library
@pytest.mark.parametrize("ani,wave", [(False, "1"), (True, "1"), (False, "0")])
def test_reference_tie_order_over_candidate_lists(oracle, skl, gpu_ctx, ref_ties, set_switch, ani, wave):
    """skl_ctx_set_knn_ties(REFERENCE) on the candidate-list path: the reference's BinaryHeap replayed over each row's
    candidates in the order they are listed (mod.rs:459-487: the order any_shared_bins returns them).  A database of
    near-copies (most keys tie) with ragged lists; host lists and lists built on the device; one wave per row (the
    default up to 256 neighbours) and one workgroup per row (SKL_REFHEAP_WAVE=0)."""
    set_switch("SKL_REFHEAP_WAVE", wave)
    gpu_ctx.set_knn_ties(skl.TIES_REFERENCE)      # (set_switch re-reads the environment)
    kmers, ss64, n = [17, 21, 25], 2, 500
    bins = synth.set_r(n, kmers, ss64, n_clusters=9)
    bins[::3] = bins[0]                                          # a third of the database is one sketch: rows of ties
    rng = np.random.default_rng(5)
    skq = rng.integers(0, 40, size=(n, 5), dtype=np.uint16)      # small alphabet: ~60 candidates per row
    skq[7] = 60000 + np.arange(5, dtype=np.uint16)               # no candidate at all
    o, g = oracle.Sketches(bins, n, kmers, ss64), gpu_ctx.sketches(bins, n, kmers, ss64)
    offs, cols = candidates(skq)
    differs = False
    for knn in (1, 4, 30, 100):
        idx, d0 = skl.self_dists_knn_candidates(gpu_ctx, g, g.set_k(21, ani), knn, offs, cols)
        exp = oracle.self_dists_knn_precluster(o, skq, knn, 1, ani, ties=oracle.TIES_RUST_HEAP, threads=8)
        assert as_pairs(idx, d0) == oracle_pairs(exp), knn
        idx2, d02, _total = skl.self_dists_knn_shared_bins(gpu_ctx, g, g.set_k(21, ani), knn, skq)
        assert np.array_equal(idx2, idx) and np.array_equal(d02, d0)
        canon = oracle.self_dists_knn_precluster(o, skq, knn, 1, ani, ties=oracle.TIES_CANONICAL, threads=8)
        differs = differs or oracle_pairs(canon) != oracle_pairs(exp)
    assert differs, "the data set is meant to make the two tie rules disagree"
    g.close()


def test_reference_tie_order_with_a_reordered_index(oracle, skl, gpu_ctx, ref_ties):
    """A .ski that orders the samples differently from the .skd: the reference pushes in ascending .ski index, so the
    caller lists each row's candidates in that order (what the host driver does from the index)."""
    kmers, ss64, n = [21], 2, 160
    bins = synth.set_r(n, kmers, ss64, n_clusters=4)
    bins[::2] = bins[1]
    rng = np.random.default_rng(8)
    skq = rng.integers(0, 30, size=(n, 4), dtype=np.uint16)
    o, g = oracle.Sketches(bins, n, kmers, ss64), gpu_ctx.sketches(bins, n, kmers, ss64)
    perm = rng.permutation(n)                                    # ski position of skd sample i
    skq_ski = np.empty_like(skq)
    skq_ski[perm] = skq
    offs, cols = candidates(skq)
    cols_ski = cols.copy()
    for i in range(n):                                           # each row in ascending .ski index
        a, b = int(offs[i]), int(offs[i + 1])
        cols_ski[a:b] = cols[a:b][np.argsort(perm[cols[a:b]], kind="stable")]
    idx, d0 = skl.self_dists_knn_candidates(gpu_ctx, g, g.set_k(21), 12, offs, cols_ski)
    exp = oracle.self_dists_knn_precluster(o, skq_ski, 12, 0, ski_of_skd=perm, ties=oracle.TIES_RUST_HEAP, threads=4)
    assert as_pairs(idx, d0) == oracle_pairs(exp)
    g.close()


def test_device_candidate_lists_many_bins_and_samples(oracle, skl, gpu_ctx):
    """cand_gen.hip on its own terms: 5 000 samples (157 bitmap words per row, 20 per thread),
    300 bins (more bins than waves), values that collide in one bin only, a value shared by
    everybody in one bin (a 5 000-member group), against the numpy definition."""
    kmers, ss64, n = [21], 2, 5000
    bins = synth.set_u(n, 1, ss64)
    rng = np.random.default_rng(9)
    skq = rng.integers(0, 65536, size=(n, 300), dtype=np.uint16)
    skq[:, 7] = rng.integers(0, 400, size=n)            # small alphabet in bin 7: ~12 partners each
    skq[::50, 250] = 4242                               # a 100-member group in bin 250
    g = gpu_ctx.sketches(bins, n, kmers, ss64)
    offs, cols = candidates(skq)
    idx, d0, total = skl.self_dists_knn_shared_bins(gpu_ctx, g, g.set_k(21), 3, skq)
    assert total == cols.size
    idx_ref, d0_ref = skl.self_dists_knn_candidates(gpu_ctx, g, g.set_k(21), 3, offs, cols)
    assert np.array_equal(idx, idx_ref) and np.array_equal(d0, d0_ref)
    skq[:, 0] = 1                                       # everybody shares bin 0: all-vs-all
    idx, d0, total = skl.self_dists_knn_shared_bins(gpu_ctx, g, g.set_k(21), 4, skq)
    assert total == n * (n - 1)
    fi, fd, _ = skl.self_dists_knn(gpu_ctx, g, g.set_k(21), 4)
    assert np.array_equal(idx, fi) and np.array_equal(d0, fd)
