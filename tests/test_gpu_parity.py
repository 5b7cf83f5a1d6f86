"""HIP path vs the CPU oracle through the C ABI (run with -m gpu on an MI355X).

Bar: bit-exact u32 bin-match counts; f32 distances within 1e-6 (in fact bit-identical
without a completeness correction, because ln J comes from host-libm tables).
"""
import numpy as np
import pytest

from helpers import FIXTURE_DBS, load_fixture_bins
from sketchlib.rust_amd import synth

pytestmark = pytest.mark.gpu

TOL = 1e-6
K5 = [15, 19, 23, 27, 31]


def _both(oracle, skl, ctx, bins, n, kmers, ss64, comp=None):
    return (oracle.Sketches(bins, n, kmers, ss64, comp), ctx.sketches(bins, n, kmers, ss64, comp))


@pytest.mark.parametrize("name", sorted(FIXTURE_DBS))
def test_fixture_binmatch_and_jaccard(oracle, skl, gpu_ctx, name):
    bins, n, kmers, ss64 = load_fixture_bins(name)
    o, g = _both(oracle, skl, gpu_ctx, bins, n, kmers, ss64)
    assert np.array_equal(skl.self_binmatch(gpu_ctx, g), oracle.self_binmatch(o))
    for ani in (False, True):
        p = g.set_k(kmers[0], ani)
        got = skl.self_dists_all(gpu_ctx, g, p)
        exp = oracle.self_dists_all(o, oracle.JACCARD, 0, ani)
        assert np.array_equal(got, exp)


def test_legacy_db_coreacc(oracle, skl, gpu_ctx):
    bins, n, kmers, ss64 = load_fixture_bins("legacy_db")
    o, g = _both(oracle, skl, gpu_ctx, bins, n, kmers, ss64)
    got = skl.self_dists_all(gpu_ctx, g, g.set_k())
    exp = oracle.self_dists_all(o)
    np.testing.assert_allclose(got, exp, atol=TOL, rtol=0)


@pytest.mark.parametrize("n,ss64,kmers", [
    (130, 64, K5),          # cfg-2 sketch shape, ragged tile edges
    (67, 16, [17, 21, 25, 29]),
    (33, 5, [13, 17]),      # odd chunk count, nk = 2
    (257, 3, [21, 25, 29, 33, 37, 41]),  # nk = 6: fused limit
])
def test_related_self_coreacc(oracle, skl, gpu_ctx, n, ss64, kmers):
    bins = synth.set_r(n, kmers, ss64, n_clusters=7)
    o, g = _both(oracle, skl, gpu_ctx, bins, n, kmers, ss64)
    assert np.array_equal(skl.self_binmatch(gpu_ctx, g), oracle.self_binmatch(o, threads=8))
    got = skl.self_dists_all(gpu_ctx, g, g.set_k())
    exp = oracle.self_dists_all(o, threads=8)
    np.testing.assert_allclose(got, exp, atol=TOL, rtol=0)
    assert np.array_equal(got, exp), "table-driven path should be bit-identical"


def test_unfused_coreacc_many_kmers(oracle, skl, gpu_ctx):
    kmers = list(range(13, 13 + 2 * 9, 2))  # nk = 9 > fused limit
    bins = synth.set_r(70, kmers, 8, n_clusters=5)
    o, g = _both(oracle, skl, gpu_ctx, bins, 70, kmers, 8)
    got = skl.self_dists_all(gpu_ctx, g, g.set_k())
    np.testing.assert_allclose(got, oracle.self_dists_all(o, threads=8), atol=TOL, rtol=0)


def test_random_bins_all_ones(oracle, skl, gpu_ctx):
    """Set U: J ~ 0, so (almost) every pair breaks at the first k -> (1, 1); the few that
    match a bin at every k regress on noise, identically on both sides."""
    bins = synth.set_u(100, 5, 64)
    o, g = _both(oracle, skl, gpu_ctx, bins, 100, K5, 64)
    got = skl.self_dists_all(gpu_ctx, g, g.set_k())
    assert np.array_equal(got, oracle.self_dists_all(o, threads=8))
    assert np.mean(got == 1.0) > 0.9


@pytest.mark.parametrize("ani", [False, True])
def test_cross_dense(oracle, skl, gpu_ctx, ani):
    kmers, ss64 = [17, 21, 25, 29], 16
    rb = synth.set_r(90, kmers, ss64, n_clusters=6)
    qb = synth.set_r(41, kmers, ss64, n_clusters=6, first_sample=1000)
    o_r, g_r = _both(oracle, skl, gpu_ctx, rb, 90, kmers, ss64)
    o_q, g_q = _both(oracle, skl, gpu_ctx, qb, 41, kmers, ss64)
    assert np.array_equal(skl.cross_binmatch(gpu_ctx, g_r, g_q),
                          oracle.cross_binmatch(o_r, o_q, threads=8))
    got = skl.cross_dists_all(gpu_ctx, g_r, g_q, g_r.set_k())
    np.testing.assert_allclose(got, oracle.cross_dists_all(o_r, o_q, threads=8), atol=TOL, rtol=0)
    p = g_r.set_k(21, ani)
    got = skl.cross_dists_all(gpu_ctx, g_r, g_q, p)
    exp = oracle.cross_dists_all(o_r, o_q, oracle.JACCARD, 1, ani, threads=8)
    assert np.array_equal(got, exp)


def test_completeness_correction(oracle, skl, gpu_ctx):
    kmers, ss64, n = [17, 21, 25, 29], 32, 60
    bins = synth.set_r(n, kmers, ss64, n_clusters=4)
    comp = np.random.default_rng(5).uniform(0.5, 1.0, n)
    comp[::7] = 1.0
    o, g = _both(oracle, skl, gpu_ctx, bins, n, kmers, ss64, comp)
    for cutoff in (0.64, 0.0, 0.99):
        got = skl.self_dists_all(gpu_ctx, g, g.set_k(cutoff=cutoff))
        exp = oracle.self_dists_all(o, cutoff=cutoff, threads=8)
        np.testing.assert_allclose(got, exp, atol=TOL, rtol=0)
        for ani in (False, True):
            got = skl.self_dists_all(gpu_ctx, g, g.set_k(21, ani, cutoff))
            exp = oracle.self_dists_all(o, oracle.JACCARD, 1, ani, cutoff, threads=8)
            np.testing.assert_allclose(got, exp, atol=TOL, rtol=0)


def test_row_bands_concatenate(oracle, skl, gpu_ctx):
    """skl_self_dists_rows over a partition of the rows == the whole condensed matrix."""
    kmers, ss64, n = K5, 8, 150
    bins = synth.set_r(n, kmers, ss64, n_clusters=9)
    g = gpu_ctx.sketches(bins, n, kmers, ss64)
    p = g.set_k()
    whole = skl.self_dists_all(gpu_ctx, g, p)
    parts = [skl.self_dists_rows(gpu_ctx, g, p, a, b) for a, b in [(0, 17), (17, 64), (64, 149), (149, 150)]]
    assert np.array_equal(np.concatenate(parts), whole)


@pytest.mark.parametrize("dist", ["coreacc", "jaccard", "ani"])
def test_self_knn(oracle, skl, gpu_ctx, dist):
    kmers, ss64, n, knn = [17, 21, 25, 29], 16, 120, 7
    bins = synth.set_r(n, kmers, ss64, n_clusters=10)
    o, g = _both(oracle, skl, gpu_ctx, bins, n, kmers, ss64)
    if dist == "coreacc":
        p, oargs = g.set_k(), (oracle.COREACC, 0, False)
    else:
        p, oargs = g.set_k(21, dist == "ani"), (oracle.JACCARD, 1, dist == "ani")
    idx, d0, d1 = skl.self_dists_knn(gpu_ctx, g, p, knn)
    exp = oracle.self_dists_knn(o, knn, *oargs, ties=oracle.TIES_CANONICAL, threads=8)
    assert np.array_equal(idx, exp["idx"])
    np.testing.assert_allclose(d0, exp["d0"], atol=TOL, rtol=0)
    if dist == "coreacc":
        np.testing.assert_allclose(d1, exp["d1"], atol=TOL, rtol=0)
    # and the same distance multiset as the reference's heap order
    heap = oracle.self_dists_knn(o, knn, *oargs, ties=oracle.TIES_RUST_HEAP, threads=8)
    np.testing.assert_allclose(np.sort(d0, axis=1), np.sort(heap["d0"], axis=1), atol=TOL, rtol=0)


def test_cross_knn(oracle, skl, gpu_ctx):
    kmers, ss64 = [17, 21, 25, 29], 16
    rb = synth.set_r(75, kmers, ss64, n_clusters=6)
    qb = synth.set_r(20, kmers, ss64, n_clusters=6, first_sample=500)
    o_r, g_r = _both(oracle, skl, gpu_ctx, rb, 75, kmers, ss64)
    o_q, g_q = _both(oracle, skl, gpu_ctx, qb, 20, kmers, ss64)
    for p, oargs in [(g_r.set_k(), (oracle.COREACC, 0, False)),
                     (g_r.set_k(25, True), (oracle.JACCARD, 2, True))]:
        idx, d0, d1 = skl.cross_dists_knn(gpu_ctx, g_r, g_q, p, 5)
        exp = oracle.cross_dists_knn(o_r, o_q, 5, *oargs, ties=oracle.TIES_CANONICAL, threads=8)
        assert np.array_equal(idx, exp["idx"])
        np.testing.assert_allclose(d0, exp["d0"], atol=TOL, rtol=0)


def test_error_codes(skl, gpu_ctx):
    bins = synth.set_u(4, 1, 2)
    g = gpu_ctx.sketches(bins, 4, [21], 2)
    with pytest.raises(skl.SklError) as e:
        skl.self_dists_all(gpu_ctx, g, g.set_k())
    assert e.value.code == skl.ERR_KMER_COUNT
    assert "Need at least two k-mer lengths" in e.value.message
    with pytest.raises(skl.SklError) as e:
        g.set_k(33)
    assert e.value.code == skl.ERR_KMER_NOT_FOUND and "K-mer size 33 not found" in e.value.message
