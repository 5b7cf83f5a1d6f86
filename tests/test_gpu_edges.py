"""Edge cases of the path through the C ABI on the GPU: degenerate sizes, unusual sketch
shapes, the unfused fall-backs, clamping and completeness corner cases."""
import numpy as np
import pytest

from sketchlib.rust_amd import synth

pytestmark = pytest.mark.gpu
TOL = 1e-6


def both(oracle, ctx, bins, n, kmers, ss64, comp=None):
    return oracle.Sketches(bins, n, kmers, ss64, comp), ctx.sketches(bins, n, kmers, ss64, comp)


@pytest.mark.parametrize("n", [0, 1, 2, 3])
def test_tiny_sample_counts(oracle, skl, gpu_ctx, n):
    kmers, ss64 = [17, 21, 25], 4
    bins = synth.set_r(max(n, 1), kmers, ss64, n_clusters=1)[:n]
    g = gpu_ctx.sketches(bins, n, kmers, ss64)
    got = skl.self_dists_all(gpu_ctx, g, g.set_k())
    assert got.shape == (n * (n - 1) // 2 if n > 1 else 0, 2)
    if n >= 2:
        o = oracle.Sketches(bins, n, kmers, ss64)
        assert np.array_equal(got, oracle.self_dists_all(o))


def test_single_kmer_jaccard_and_one_chunk(oracle, skl, gpu_ctx):
    bins = synth.set_r(70, [31], 1, n_clusters=3)        # sketchsize64 = 1: one chunk per sketch
    o, g = both(oracle, gpu_ctx, bins, 70, [31], 1)
    for ani in (False, True):
        assert np.array_equal(skl.self_dists_all(gpu_ctx, g, g.set_k(31, ani)),
                              oracle.self_dists_all(o, oracle.JACCARD, 0, ani))


def test_large_sketch_unfused_counts(oracle, skl, gpu_ctx):
    """sketchsize64 = 1030 (65 920 bins): counts no longer fit the u16 fields of the fused
    epilogue -> counts kernel + regression kernel; expected_samebits = 4 is exercised too."""
    kmers, ss64, n = [17, 21, 25, 29], 1030, 18
    bins = synth.set_r(n, kmers, ss64, n_clusters=3)
    o, g = both(oracle, gpu_ctx, bins, n, kmers, ss64)
    assert np.array_equal(skl.self_binmatch(gpu_ctx, g), oracle.self_binmatch(o, threads=8))
    np.testing.assert_allclose(skl.self_dists_all(gpu_ctx, g, g.set_k()),
                               oracle.self_dists_all(o, threads=8), atol=TOL, rtol=0)
    assert np.array_equal(skl.self_dists_all(gpu_ctx, g, g.set_k(21)),
                          oracle.self_dists_all(o, oracle.JACCARD, 1, threads=8))


def test_expected_samebits_nonzero(oracle, skl, gpu_ctx):
    """sketchsize64 >= 256 -> expected_samebits = maxnbits >> 14 >= 1 (jaccard.rs:26-31)."""
    kmers, ss64, n = [21, 25, 29], 300, 40
    bins = synth.set_u(n, 3, ss64)        # random bins: samebits ~ 1, saturating_sub hits 0
    o, g = both(oracle, gpu_ctx, bins, n, kmers, ss64)
    assert np.array_equal(skl.self_dists_all(gpu_ctx, g, g.set_k(25)), oracle.self_dists_all(o, oracle.JACCARD, 1))
    assert np.array_equal(skl.self_dists_all(gpu_ctx, g, g.set_k()), oracle.self_dists_all(o))


def test_cross_with_unfused_and_completeness(oracle, skl, gpu_ctx):
    kmers = [13, 15, 17, 19, 21, 23, 25]      # 7 k-mer lengths > fused limit of 6
    ss64 = 6
    rb = synth.set_r(50, kmers, ss64, n_clusters=4)
    qb = synth.set_r(23, kmers, ss64, n_clusters=4, first_sample=300)
    rng = np.random.default_rng(1)
    rc, qc = rng.uniform(0.6, 1.0, 50), rng.uniform(0.6, 1.0, 23)
    o_r, g_r = both(oracle, gpu_ctx, rb, 50, kmers, ss64, rc)
    o_q, g_q = both(oracle, gpu_ctx, qb, 23, kmers, ss64, qc)
    got = skl.cross_dists_all(gpu_ctx, g_r, g_q, g_r.set_k())
    np.testing.assert_allclose(got, oracle.cross_dists_all(o_r, o_q, threads=8), atol=TOL, rtol=0)
    # only one side has completeness -> None semantics (jaccard.rs:36 needs both)
    g_q.set_completeness(None)
    o_q2 = oracle.Sketches(qb, 23, kmers, ss64)
    got = skl.cross_dists_all(gpu_ctx, g_r, g_q, g_r.set_k())
    np.testing.assert_allclose(got, oracle.cross_dists_all(o_r, o_q2, threads=8), atol=TOL, rtol=0)


def test_completeness_zero_gives_reference_nan_semantics(oracle, skl, gpu_ctx):
    """c = 0 with cutoff 0: the correction factor is 0/0 = NaN; the reference's f64 code then
    propagates NaN into ysum -> (1, 1).  Same on the device."""
    kmers, ss64, n = [17, 21, 25], 8, 12
    bins = synth.set_r(n, kmers, ss64, n_clusters=2)
    comp = np.ones(n)
    comp[3] = 0.0
    o, g = both(oracle, gpu_ctx, bins, n, kmers, ss64, comp)
    got = skl.self_dists_all(gpu_ctx, g, g.set_k(cutoff=0.0))
    exp = oracle.self_dists_all(o, cutoff=0.0)
    np.testing.assert_allclose(got, exp, atol=TOL, rtol=0, equal_nan=True)


def test_knn_bounds_and_all_neighbours(oracle, skl, gpu_ctx):
    kmers, ss64, n = [17, 21, 25], 8, 30
    bins = synth.set_r(n, kmers, ss64, n_clusters=3)
    o, g = both(oracle, gpu_ctx, bins, n, kmers, ss64)
    p = g.set_k(21)
    with pytest.raises(skl.SklError):
        skl.self_dists_knn(gpu_ctx, g, p, n)          # must be < n (lib.rs:379-382 clamps first)
    with pytest.raises(skl.SklError):
        skl.self_dists_knn(gpu_ctx, g, p, 0)
    idx, d0, _ = skl.self_dists_knn(gpu_ctx, g, p, n - 1)   # every other sample
    exp = oracle.self_dists_knn(o, n - 1, oracle.JACCARD, 1, False, ties=oracle.TIES_CANONICAL)
    assert np.array_equal(idx, exp["idx"]) and np.array_equal(d0, exp["d0"])
    for row in range(n):
        assert sorted(idx[row].tolist()) == [j for j in range(n) if j != row]
    # cross: knn == n_ref allowed (mod.rs:325)
    idx, d0, _ = skl.cross_dists_knn(gpu_ctx, g, g, p, n)
    assert idx.shape == (n, n) and np.all(d0[:, 0] == 0.0) and np.all(idx[:, 0] == np.arange(n))


def test_incompatible_and_empty_databases(skl, gpu_ctx):
    a = gpu_ctx.sketches(synth.set_u(4, 2, 2), 4, [17, 21], 2)
    b = gpu_ctx.sketches(synth.set_u(4, 2, 2), 4, [17, 25], 2)
    with pytest.raises(skl.SklError) as e:
        skl.cross_dists_all(gpu_ctx, a, b, a.set_k())
    assert e.value.code == skl.ERR_INCOMPATIBLE
    empty = gpu_ctx.sketches(np.zeros(0, dtype=np.uint64), 0, [17, 21], 2)
    with pytest.raises(skl.SklError) as e:
        skl.cross_dists_knn(gpu_ctx, empty, a, a.set_k(), 1)
    assert e.value.code in (skl.ERR_EMPTY_DB, skl.ERR_INVALID_ARG)
    with pytest.raises(skl.SklError) as e:
        skl.cross_dists_knn(gpu_ctx, a, empty, a.set_k(), 1)
    assert e.value.code == skl.ERR_EMPTY_DB and "Query database has no loaded samples" in e.value.message
    # self kNN over zero or one sample has no neighbour to report: an argument error, not a crash
    one = gpu_ctx.sketches(synth.set_u(1, 2, 2), 1, [17, 21], 2)
    for db in (empty, one):
        with pytest.raises(skl.SklError) as e:
            skl.self_dists_knn(gpu_ctx, db, db.set_k(), 1)
        assert e.value.code == skl.ERR_INVALID_ARG


def test_one_shot_host_entry_point(oracle, skl, gpu_ctx):
    kmers, ss64, n = [17, 21, 25], 8, 25
    bins = synth.set_r(n, kmers, ss64, n_clusters=3)
    got = skl.self_dists_all_host(bins, n, kmers, ss64, skl.params(skl.COREACC))
    assert np.array_equal(got, oracle.self_dists_all(oracle.Sketches(bins, n, kmers, ss64)))


def test_kernel_choice_is_reported(skl, gpu_ctx):
    g = gpu_ctx.sketches(synth.set_u(200, 5, 64), 200, [15, 19, 23, 27, 31], 64)
    skl.self_dists_all(gpu_ctx, g, g.set_k())
    assert "pair_kernel" in gpu_ctx.last_kernel()
