"""BASELINE.json configs[3] and configs[4] on their OWN sketch shape -- sketchsize64 = 32,
k = {13,17,21,25,29} -- at sizes the oracle finishes in seconds: dense ref x query ("subset mode",
cross_dists_all mod.rs:227-297) and kNN-50 (cross_dists_knn mod.rs:306-395, self_dists_knn
mod.rs:133-224), every output against the oracle.  The full-size runs of the same configurations
are in test_gpu_fullsize_configs.py."""
import numpy as np
import pytest

from sketchlib.rust_amd import synth

pytestmark = pytest.mark.gpu
K4 = [13, 17, 21, 25, 29]
SS64 = 32
N_REF, N_QUERY, KNN = 2000, 500, 50


@pytest.fixture(scope="module")
def dbs(oracle, skl, _product_ctx):
    gpu_ctx = _product_ctx
    # refs and queries drawn from the same 25 clusters (sample s is in cluster s % 25), so a query has
    # ~80 related references with a real regression and ~1 900 unrelated ones at (1, 1)
    rb = synth.set_r(N_REF, K4, SS64, n_clusters=25)
    qb = synth.set_r(N_QUERY, K4, SS64, n_clusters=25, first_sample=50_000)
    qb[7] = rb[1234]                       # a query that IS a reference: distance (0, 0), Jaccard 0
    o_r, o_q = oracle.Sketches(rb, N_REF, K4, SS64), oracle.Sketches(qb, N_QUERY, K4, SS64)
    g_r, g_q = gpu_ctx.sketches(rb, N_REF, K4, SS64), gpu_ctx.sketches(qb, N_QUERY, K4, SS64)
    yield rb, qb, o_r, o_q, g_r, g_q
    g_r.close()
    g_q.close()


def test_cfg4_dense_cross_coreacc(oracle, skl, gpu_ctx, dbs):
    _rb, _qb, o_r, o_q, g_r, g_q = dbs
    got = skl.cross_dists_all(gpu_ctx, g_r, g_q, g_r.set_k())
    exp = oracle.cross_dists_all(o_r, o_q, threads=8)
    assert got.shape == (N_REF, N_QUERY, 2)
    assert np.array_equal(got, exp)                       # f32 bit-identical (bar: 1e-6)
    assert got[1234, 7].tolist() == [0.0, 0.0]
    fitted = (got[..., 0] > 0) & (got[..., 0] < 1)
    assert 0.02 < fitted.mean() < 0.2                     # the regression is exercised, and so is (1, 1)
    assert np.array_equal(skl.cross_binmatch(gpu_ctx, g_r, g_q), oracle.cross_binmatch(o_r, o_q, threads=8))


@pytest.mark.parametrize("ani", [False, True])
def test_cfg4_dense_cross_single_k(oracle, skl, gpu_ctx, dbs, ani):
    _rb, _qb, o_r, o_q, g_r, g_q = dbs
    for k_idx, k in enumerate(K4):
        got = skl.cross_dists_all(gpu_ctx, g_r, g_q, g_r.set_k(k, ani=ani))
        assert np.array_equal(got, oracle.cross_dists_all(o_r, o_q, oracle.JACCARD, k_idx, ani, threads=8)), k


def test_cfg4_row_bands_are_the_whole(skl, gpu_ctx, dbs):
    """The 8-GPU partition of configs[3]: contiguous reference bands, concatenated."""
    from sketchlib.rust_amd import multi_gpu

    _rb, _qb, _o_r, _o_q, g_r, g_q = dbs
    p = g_r.set_k()
    whole = skl.cross_dists_all(gpu_ctx, g_r, g_q, p)
    for world in (2, 8):
        b = multi_gpu.even_row_bounds(N_REF, world)
        parts = [skl.cross_dists_rows(gpu_ctx, g_r, g_q, p, b[w], b[w + 1]) for w in range(world)]
        assert np.array_equal(np.concatenate(parts), whole)


def test_cfg4_with_completeness(oracle, skl, gpu_ctx, dbs):
    rb, qb, _o_r, _o_q, g_r, g_q = dbs
    cr, cq = np.linspace(0.55, 1.0, N_REF), np.linspace(1.0, 0.7, N_QUERY)
    o_r, o_q = oracle.Sketches(rb, N_REF, K4, SS64, completeness=cr), oracle.Sketches(qb, N_QUERY, K4, SS64, completeness=cq)
    g_r.set_completeness(cr)
    g_q.set_completeness(cq)
    try:
        for cutoff in (0.64, 0.0, 0.95):
            got = skl.cross_dists_all(gpu_ctx, g_r, g_q, g_r.set_k(cutoff=cutoff))
            exp = oracle.cross_dists_all(o_r, o_q, cutoff=cutoff, threads=8)
            np.testing.assert_allclose(got, exp, rtol=0, atol=1e-6)      # every pair, flat fits included
            got = skl.cross_dists_all(gpu_ctx, g_r, g_q, g_r.set_k(21, cutoff=cutoff))
            np.testing.assert_allclose(got, oracle.cross_dists_all(o_r, o_q, oracle.JACCARD, 2, False, cutoff, threads=8),
                                       rtol=0, atol=1e-6)
    finally:
        g_r.set_completeness(None)
        g_q.set_completeness(None)


def _check_knn(oracle, got, exp_canonical, exp_heap, ani=False):
    idx, d0, d1 = got
    assert np.array_equal(idx, exp_canonical["idx"])
    np.testing.assert_allclose(d0, exp_canonical["d0"], rtol=0, atol=1e-6)
    # the reference's BinaryHeap keeps the same distances, whatever ids it keeps among ties
    np.testing.assert_allclose(np.sort(d0, axis=1), np.sort(exp_heap["d0"], axis=1), rtol=0, atol=1e-6)
    # ascending distances; ANI rows are sorted on 1 - ANI and un-transformed (mod.rs:183-189): descending
    assert np.all(np.diff(d0, axis=1) <= 0) if ani else np.all(np.diff(d0, axis=1) >= 0)


@pytest.mark.parametrize("mode", ["coreacc", "jaccard", "ani"])
def test_cfg5_cross_knn50(oracle, skl, gpu_ctx, dbs, mode):
    _rb, _qb, o_r, o_q, g_r, g_q = dbs
    p, oargs = {"coreacc": (g_r.set_k(), (oracle.COREACC, 0, False)),
                "jaccard": (g_r.set_k(21), (oracle.JACCARD, 2, False)),
                "ani": (g_r.set_k(17, ani=True), (oracle.JACCARD, 1, True))}[mode]
    got = skl.cross_dists_knn(gpu_ctx, g_r, g_q, p, KNN)
    exp = oracle.cross_dists_knn(o_r, o_q, KNN, *oargs, ties=oracle.TIES_CANONICAL, threads=8)
    heap = oracle.cross_dists_knn(o_r, o_q, KNN, *oargs, ties=oracle.TIES_RUST_HEAP, threads=8)
    _check_knn(oracle, got, exp, heap, ani=mode == "ani")
    if mode == "coreacc":
        np.testing.assert_allclose(got[2], exp["d1"], rtol=0, atol=1e-6)
    # the planted reference is a nearest neighbour of query 7 (core distance 0 can tie with flat fits,
    # and ties go to the lowest id)
    at = np.flatnonzero(got[0][7] == 1234)
    assert at.size == 1 and got[1][7, at[0]] == got[1][7, 0]


@pytest.mark.parametrize("mode", ["coreacc", "jaccard"])
def test_cfg5_self_knn50(oracle, skl, gpu_ctx, dbs, mode):
    _rb, _qb, o_r, _o_q, g_r, _g_q = dbs
    p, oargs = {"coreacc": (g_r.set_k(), (oracle.COREACC, 0, False)),
                "jaccard": (g_r.set_k(21), (oracle.JACCARD, 2, False))}[mode]
    got = skl.self_dists_knn(gpu_ctx, g_r, p, KNN)
    exp = oracle.self_dists_knn(o_r, KNN, *oargs, ties=oracle.TIES_CANONICAL, threads=8)
    heap = oracle.self_dists_knn(o_r, KNN, *oargs, ties=oracle.TIES_RUST_HEAP, threads=8)
    _check_knn(oracle, got, exp, heap)
    assert not np.any(got[0] == np.arange(N_REF)[:, None])      # self skipped (mod.rs:150)
