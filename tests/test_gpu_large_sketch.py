"""Sketches beyond 65 535 bins -- the sizes the reference recommends "for SNP level resolution"
(`-s 100000` ... `-s 1000000`, /root/reference/src/lib.rs:41-42: sketchsize64 = 1 563 ... 15 625) -- through the C ABI:
the chunk-split kernel walks such a k-mer length in segments of 1 016 chunks (pair_kslice_walk.inc) instead of
falling back to the one-column kernel.  Bin-match counts bit-exact, f32 Jaccard / ANI bit-exact, core/accessory
within 1e-6 (bit-exact without completeness), self / cross / row bands / kNN (row by row and one evaluation)."""
import numpy as np
import pytest

from sketchlib.rust_amd import synth

pytestmark = pytest.mark.gpu
TOL = 1e-6


def _db(oracle, ctx, n, kmers, ss64, first=0, clusters=3, comp=None):
    bins = synth.set_r(n, kmers, ss64, n_clusters=clusters, first_sample=first)
    return bins, oracle.Sketches(bins, n, kmers, ss64, comp), ctx.sketches(bins, n, kmers, ss64, comp)


@pytest.mark.parametrize("ss64,n,nq", [(1024, 150, 40), (1563, 140, 33), (2032, 70, 20), (15625, 36, 9)])
def test_large_sketches_take_the_chunk_split_kernel(oracle, skl, gpu_ctx, ss64, n, nq):
    """1 024: the first size past the u16 fields (2 segments, the second of 8 chunks); 1 563 = 100 000 bins (odd: the
    last stage is short); 2 032 = exactly 2 segments; 15 625 = 1 000 000 bins (16 segments)."""
    kmers = [17, 21, 25, 29]
    _b, o, g = _db(oracle, gpu_ctx, n, kmers, ss64)
    _q, oq, gq = _db(oracle, gpu_ctx, nq, kmers, ss64, first=500)
    # raw counts, self and cross
    assert np.array_equal(skl.self_binmatch(gpu_ctx, g), oracle.self_binmatch(o, threads=8))
    assert "pair_kernel_kslice" in gpu_ctx.last_kernel() and "segments of 1016" in gpu_ctx.last_kernel(), gpu_ctx.last_kernel()
    assert np.array_equal(skl.cross_binmatch(gpu_ctx, g, gq), oracle.cross_binmatch(o, oq, threads=8))
    # core/accessory: counts + epilogue
    got = skl.self_dists_all(gpu_ctx, g, g.set_k())
    assert "pair_kernel_kslice" in gpu_ctx.last_kernel() and "COUNTS, k-sliced" in gpu_ctx.last_kernel(), gpu_ctx.last_kernel()
    assert np.array_equal(got, oracle.self_dists_all(o, threads=8))
    assert np.array_equal(skl.cross_dists_all(gpu_ctx, g, gq, g.set_k()), oracle.cross_dists_all(o, oq, threads=8))
    r0, r1 = n // 3, n // 3 + n // 4
    assert np.array_equal(skl.self_dists_rows(gpu_ctx, g, g.set_k(), r0, r1), oracle.self_dists_all(o, threads=8)[_cond(r0, n):_cond(r1, n)])
    # single k, distance and ANI
    for ani in (False, True):
        got = skl.self_dists_all(gpu_ctx, g, g.set_k(21, ani))
        # (a launch of a few rounds over such a sketch: bin-match counts with its last round in chunk slices + the epilogue)
        assert "pair_kernel_kslice" in gpu_ctx.last_kernel() and "segments of 1016" in gpu_ctx.last_kernel(), gpu_ctx.last_kernel()
        assert np.array_equal(got, oracle.self_dists_all(o, oracle.JACCARD, 1, ani, threads=8))
        assert np.array_equal(skl.cross_dists_all(gpu_ctx, g, gq, g.set_k(21, ani)),
                              oracle.cross_dists_all(o, oq, oracle.JACCARD, 1, ani, threads=8))
    g.close()
    gq.close()


def _cond(r, n):
    r = min(r, n - 1)
    return r * n - r * (r + 1) // 2


def test_large_sketch_32_row_tiles(oracle, skl, gpu_ctx, set_switch):
    """The 32 x 128 form (large launches) with segments: forced onto a small database."""
    kmers, ss64, n = [15, 19, 23], 1563, 170
    _b, o, g = _db(oracle, gpu_ctx, n, kmers, ss64, clusters=4)
    set_switch("SKL_TILE32_MIN", "0")
    assert np.array_equal(skl.self_binmatch(gpu_ctx, g), oracle.self_binmatch(o, threads=8))
    assert "R=32" in gpu_ctx.last_kernel() and "segments of 1016" in gpu_ctx.last_kernel(), gpu_ctx.last_kernel()
    assert np.array_equal(skl.self_dists_all(gpu_ctx, g, g.set_k()), oracle.self_dists_all(o, threads=8))
    for ani in (False, True):
        assert np.array_equal(skl.self_dists_all(gpu_ctx, g, g.set_k(19, ani)), oracle.self_dists_all(o, oracle.JACCARD, 1, ani, threads=8))
        assert "R=32" in gpu_ctx.last_kernel()
    g.close()


def test_large_sketch_with_completeness(oracle, skl, gpu_ctx):
    kmers, ss64, n = [17, 21, 25, 29], 1563, 60
    comp = np.linspace(0.65, 1.0, n)
    _b, o, g = _db(oracle, gpu_ctx, n, kmers, ss64, comp=comp)
    np.testing.assert_allclose(skl.self_dists_all(gpu_ctx, g, g.set_k()), oracle.self_dists_all(o, threads=8), atol=TOL, rtol=0)
    for ani in (False, True):
        np.testing.assert_allclose(skl.self_dists_all(gpu_ctx, g, g.set_k(25, ani)),
                                   oracle.self_dists_all(o, oracle.JACCARD, 2, ani, threads=8), atol=TOL, rtol=0)
    g.close()


@pytest.mark.ab_library
@pytest.mark.parametrize("tile32", [False, True])
def test_large_sketch_knn(oracle, skl, gpu_ctx, set_switch, tile32):
    """Self kNN (single k): the one-evaluation driver with its turned second store, several bands; cross kNN and
    core/accessory kNN row by row (counts + epilogue per band)."""
    kmers, ss64, n, knn = [17, 21, 25], 1563, 150, 7
    _b, o, g = _db(oracle, gpu_ctx, n, kmers, ss64, clusters=5)
    _q, oq, gq = _db(oracle, gpu_ctx, 20, kmers, ss64, first=700, clusters=5)
    set_switch("SKL_KNN_BAND_ROWS", "48")
    if tile32:
        set_switch("SKL_TILE32_MIN", "0")
    for ani in (False, True):
        idx, d0, _ = skl.self_dists_knn(gpu_ctx, g, g.set_k(21, ani), knn)
        assert "pair_kernel_kslice" in gpu_ctx.last_kernel() and "segments" in gpu_ctx.last_kernel(), gpu_ctx.last_kernel()
        exp = oracle.self_dists_knn(o, knn, oracle.JACCARD, 1, ani, ties=oracle.TIES_CANONICAL, threads=8)
        assert np.array_equal(idx, exp["idx"])
        np.testing.assert_allclose(d0, exp["d0"], atol=TOL, rtol=0)
    set_switch("SKL_KNN_SYMMETRIC", "0")     # ... and row by row: same lists
    idx2, d02, _ = skl.self_dists_knn(gpu_ctx, g, g.set_k(21, True), knn)
    assert np.array_equal(idx2, idx) and np.array_equal(d02, d0)
    set_switch("SKL_KNN_SYMMETRIC", None)
    idx, d0, _ = skl.cross_dists_knn(gpu_ctx, g, gq, g.set_k(21), knn)
    exp = oracle.cross_dists_knn(o, oq, knn, oracle.JACCARD, 1, False, ties=oracle.TIES_CANONICAL, threads=8)
    assert np.array_equal(idx, exp["idx"])
    np.testing.assert_allclose(d0, exp["d0"], atol=TOL, rtol=0)
    idx, d0, d1 = skl.self_dists_knn(gpu_ctx, g, g.set_k(), knn)
    exp = oracle.self_dists_knn(o, knn, oracle.COREACC, 0, False, ties=oracle.TIES_CANONICAL, threads=8)
    assert np.array_equal(idx, exp["idx"])
    np.testing.assert_allclose(d0, exp["d0"], atol=TOL, rtol=0)
    np.testing.assert_allclose(d1, exp["d1"], atol=TOL, rtol=0)
    g.close()
    gq.close()


def test_counts_scratch_is_bounded(oracle, skl, gpu_ctx):
    """Core/accessory over a large sketch parks its bin-match counts in HBM: at most 4 GiB per launch, a bigger call is
    computed in row bands.  20 000 x 20 000 pairs x 3 k x 4 B = 4.8 GB -> at least two launches (round 6: eight bands of
    50 M pairs, each band's epilogue beside the next band's counts kernel); spot-checked against the oracle."""
    import torch

    kmers, ss64, n = [17, 21, 25], 1024, 20000
    dev = torch.device("cuda", 0)
    bins = synth.set_clustered_device(n, len(kmers), ss64, dev, cluster_size=50, keep=[0.97, 0.95, 0.93])
    torch.cuda.synchronize()      # (torch fills on ITS stream; the session's context runs on a stream of its own)
    g = gpu_ctx.sketches(bins, n, kmers, ss64)
    out = torch.zeros((n, n, 2), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    gpu_ctx.timing_enable()
    gpu_ctx.timing_reset()
    skl.cross_dists_all(gpu_ctx, g, g, g.set_k(), out=out)
    torch.cuda.synchronize()
    _ms, launches = gpu_ctx.kernel_ms()
    rng = np.random.default_rng(3)
    ii = np.concatenate([rng.integers(0, n, 60), [0, n // 2 - 1, n // 2, n - 1]])
    jj = np.concatenate([ii[:60] % 400 + 400 * rng.integers(0, 50, 60), [1, n // 2, n // 2 - 1, n - 2]])   # (cluster = id % 400)
    ids = np.unique(np.concatenate([ii, jj]))
    o = oracle.Sketches(bins[torch.from_numpy(ids).to(dev)].cpu().numpy().view(np.uint64), len(ids), kmers, ss64)
    pos = {int(s): k for k, s in enumerate(ids)}
    got = out[torch.from_numpy(ii).to(dev), torch.from_numpy(jj).to(dev)].cpu().numpy()
    fitted = 0
    for t in range(len(ii)):
        exp = oracle.core_acc_pair(o, o, pos[int(ii[t])], pos[int(jj[t])])
        assert abs(got[t, 0] - exp[0]) <= TOL and abs(got[t, 1] - exp[1]) <= TOL, (ii[t], jj[t], got[t], exp)
        fitted += 0.0 < exp[0] < 1.0
    assert fitted >= 20
    g.close()
    # (what the default dispatch does, asserted LAST: scripts/forced_switch_suites.sh)
    assert 2 <= launches <= 16, launches
