"""Symmetric self kNN (every pair evaluated once, running per-row top-k) against the oracle and
against the row-by-row form of the same library: same neighbours in the same order.

The reference evaluates both (i, j) and (j, i) (src/distances/mod.rs:148-171); the results it
defines are what is compared here.
"""
import numpy as np
import pytest

from sketchlib.rust_amd import synth

pytestmark = [pytest.mark.gpu, pytest.mark.ab_library]   # (SKL_KNN_SYMMETRIC / SKL_KNN_PRUNE ...: switches of the A/B build)

TOL = 1e-6


def _knn(skl, ctx, g, p, knn, monkeypatch, band_rows, symmetric, with_d1=False):
    monkeypatch.setenv("SKL_KNN_BAND_ROWS", str(band_rows))
    monkeypatch.setenv("SKL_KNN_SYMMETRIC", "1" if symmetric else "0")
    ctx.reload_env()
    idx, d0, d1 = skl.self_dists_knn(ctx, g, p, knn)
    return (idx, d0, d1) if with_d1 else (idx, d0)


@pytest.mark.parametrize("tile_rows", [16, 32])
@pytest.mark.parametrize("band_rows", [7, 16, 40, 64, 100, 332])
@pytest.mark.parametrize("ani", [False, True], ids=["dist", "ani"])
def test_bands_of_every_shape_match_the_oracle(oracle, skl, gpu_ctx, monkeypatch, band_rows, ani, tile_rows):
    """Both tile heights of the pair kernel turn their tiles into the transposed band (large bands take
    32 x 128 tiles: SKL_TILE32_MIN; forced here either way)."""
    kmers, ss64, n, knn = [17, 21, 25], 8, 333, 11
    bins = synth.set_r(n, kmers, ss64, n_clusters=9)
    o, g = oracle.Sketches(bins, n, kmers, ss64), gpu_ctx.sketches(bins, n, kmers, ss64)
    p = g.set_k(21, ani)
    monkeypatch.setenv("SKL_TILE32_MIN", "0" if tile_rows == 32 else "-1")
    idx, d0 = _knn(skl, gpu_ctx, g, p, knn, monkeypatch, band_rows, True)
    assert "k-sliced" in gpu_ctx.last_kernel() and f"R={tile_rows}," in gpu_ctx.last_kernel()
    exp = oracle.self_dists_knn(o, knn, oracle.JACCARD, 1, ani, ties=oracle.TIES_CANONICAL, threads=8)
    assert np.array_equal(idx, exp["idx"])
    np.testing.assert_allclose(d0, exp["d0"], atol=TOL, rtol=0)


@pytest.mark.parametrize("tile_rows", [16, 32])
@pytest.mark.parametrize("band_rows", [7, 16, 50, 128])
def test_core_accessory_keys(oracle, skl, gpu_ctx, monkeypatch, band_rows, tile_rows):
    """(core, acc) records: the key is the core distance, the accessory distance rides along
    (distance_matrix.rs:245-248), through both copies of a record and the running top-k."""
    kmers, ss64, n, knn = [15, 19, 23, 27, 31], 8, 211, 9
    bins = synth.set_r(n, kmers, ss64, n_clusters=6)
    o, g = oracle.Sketches(bins, n, kmers, ss64), gpu_ctx.sketches(bins, n, kmers, ss64)
    monkeypatch.setenv("SKL_TILE32_MIN", "0" if tile_rows == 32 else "-1")
    idx, d0, d1 = _knn(skl, gpu_ctx, g, g.set_k(), knn, monkeypatch, band_rows, True, with_d1=True)
    assert "COREACC, all k" in gpu_ctx.last_kernel() and f"R={tile_rows}," in gpu_ctx.last_kernel()
    ref = _knn(skl, gpu_ctx, g, g.set_k(), knn, monkeypatch, band_rows, False, with_d1=True)
    assert np.array_equal(idx, ref[0]) and np.array_equal(d0, ref[1]) and np.array_equal(d1, ref[2])
    exp = oracle.self_dists_knn(o, knn, oracle.COREACC, 0, False, ties=oracle.TIES_CANONICAL, threads=8)
    assert np.array_equal(idx, exp["idx"])
    np.testing.assert_allclose(d0, exp["d0"], atol=TOL, rtol=0)
    np.testing.assert_allclose(d1, exp["d1"], atol=TOL, rtol=0)


def test_all_keys_tie(oracle, skl, gpu_ctx, monkeypatch):
    """n copies of one sketch: every distance is 0, so the neighbours are decided by the tie rule
    alone (smallest sample id first), whichever band or copy of a key reaches a row first."""
    kmers, ss64, n, knn = [21], 4, 200, 5
    bins = np.tile(synth.set_u(1, len(kmers), ss64), (n, 1))
    g = gpu_ctx.sketches(bins, n, kmers, ss64)
    for band_rows in (48, 7):
        idx, d0 = _knn(skl, gpu_ctx, g, g.set_k(21), knn, monkeypatch, band_rows, True)
        assert np.all(d0 == 0.0)
        for row in range(n):
            assert idx[row].tolist() == [j for j in range(n) if j != row][:knn]


def test_completeness_correction(oracle, skl, gpu_ctx, monkeypatch):
    kmers, ss64, n, knn = [17, 21], 8, 150, 6
    bins = synth.set_r(n, kmers, ss64, n_clusters=5)
    comp = np.random.default_rng(5).uniform(0.7, 1.0, n)
    o, g = oracle.Sketches(bins, n, kmers, ss64, comp), gpu_ctx.sketches(bins, n, kmers, ss64, comp)
    idx, d0 = _knn(skl, gpu_ctx, g, g.set_k(21), knn, monkeypatch, 32, True)
    ref_idx, ref_d0 = _knn(skl, gpu_ctx, g, g.set_k(21), knn, monkeypatch, 32, False)
    assert np.array_equal(idx, ref_idx) and np.array_equal(d0, ref_d0)
    exp = oracle.self_dists_knn(o, knn, oracle.JACCARD, 1, False, ties=oracle.TIES_CANONICAL, threads=8)
    np.testing.assert_allclose(d0, exp["d0"], atol=TOL, rtol=0)


@pytest.mark.parametrize("n,band_rows,knn", [(3000, 256, 10), (5000, 2100, 20), (4099, 1000, 64)])
def test_same_as_row_by_row(oracle, skl, gpu_ctx, monkeypatch, n, band_rows, knn):
    """Cold states with more than 2048 candidates take the radix-select branch of the merge; warm
    ones the in-LDS sort; both forms of the driver must agree bit for bit."""
    kmers, ss64 = [21], 3
    bins = synth.set_r(n, kmers, ss64, n_clusters=40)
    g = gpu_ctx.sketches(bins, n, kmers, ss64)
    p = g.set_k(21)
    idx, d0 = _knn(skl, gpu_ctx, g, p, knn, monkeypatch, band_rows, True)
    ref_idx, ref_d0 = _knn(skl, gpu_ctx, g, p, knn, monkeypatch, band_rows, False)
    assert np.array_equal(idx, ref_idx) and np.array_equal(d0, ref_d0)
    o = oracle.Sketches(bins, n, kmers, ss64)
    exp = oracle.self_dists_knn(o, knn, oracle.JACCARD, 0, False, ties=oracle.TIES_CANONICAL, threads=8)
    assert np.array_equal(idx, exp["idx"])


@pytest.mark.parametrize("dist", ["jaccard", "coreacc"])
def test_row_flags_skip_only_rows_without_news(oracle, skl, gpu_ctx, monkeypatch, dist):
    """The pair kernel flags the rows a band improves and the merge of the transposed band skips the
    rest (SKL_KNN_ROW_FLAGS=0: every row is visited).  Clustered data, many small bands: most rows are
    final after their cluster's bands, so most visits are skipped -- and nothing may change."""
    kmers, ss64, n, knn = ([21], 3, 2600, 12) if dist == "jaccard" else ([15, 19, 23], 8, 900, 7)
    bins = synth.set_r(n, kmers, ss64, n_clusters=25)
    g = gpu_ctx.sketches(bins, n, kmers, ss64)
    p = g.set_k(21) if dist == "jaccard" else g.set_k()
    monkeypatch.setenv("SKL_KNN_ROW_FLAGS", "1")
    on = _knn(skl, gpu_ctx, g, p, knn, monkeypatch, 48, True, with_d1=True)
    monkeypatch.setenv("SKL_KNN_ROW_FLAGS", "0")
    off = _knn(skl, gpu_ctx, g, p, knn, monkeypatch, 48, True, with_d1=True)
    for a, b in zip(on, off):
        assert (a is None and b is None) or np.array_equal(a, b)
    o = oracle.Sketches(bins, n, kmers, ss64)
    exp = oracle.self_dists_knn(o, knn, oracle.JACCARD if dist == "jaccard" else oracle.COREACC, 0, False,
                                ties=oracle.TIES_CANONICAL, threads=8)
    assert np.array_equal(on[0], exp["idx"])


def test_device_output_and_default_band_size(skl, gpu_ctx, monkeypatch):
    """No knobs: the driver sizes the bands itself; one band means the row-by-row form runs."""
    monkeypatch.delenv("SKL_KNN_BAND_ROWS", raising=False)
    monkeypatch.delenv("SKL_KNN_SYMMETRIC", raising=False)
    kmers, ss64, n, knn = [21], 2, 500, 4
    bins = synth.set_r(n, kmers, ss64, n_clusters=7)
    g = gpu_ctx.sketches(bins, n, kmers, ss64)
    idx, d0, _ = skl.self_dists_knn(gpu_ctx, g, g.set_k(21), knn)
    assert idx.shape == (n, knn) and np.all(np.diff(d0, axis=1) >= 0)


@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("dist", ["jaccard", "ani", "coreacc"])
def test_partial_states_of_every_rank_merge_to_the_whole(skl, gpu_ctx, monkeypatch, world, dist):
    """The multi-GPU form on one device: each 'rank' computes its dealt bands (every pair once
    across the ranks), the row shards of the partial states are stacked as the all-to-all would
    deliver them, and skl_knn_merge_states must give the rows of the single-call result."""
    from sketchlib.rust_amd import multi_gpu
    kmers, ss64, n, knn, band_rows = [15, 19, 23, 27], 4, 203, 7, 16
    bins = synth.set_r(n, kmers, ss64, n_clusters=5)
    g = gpu_ctx.sketches(bins, n, kmers, ss64)
    p = g.set_k() if dist == "coreacc" else g.set_k(23, dist == "ani")
    monkeypatch.setenv("SKL_KNN_SYMMETRIC", "0")
    gpu_ctx.reload_env()
    whole = skl.self_dists_knn(gpu_ctx, g, p, knn)
    n_bands = (n + band_rows - 1) // band_rows
    deal = multi_gpu.knn_band_deal(n_bands, world)
    states = [skl.self_dists_knn_partial(gpu_ctx, g, p, knn, band_rows, deal[r]) for r in range(world)]
    bounds = multi_gpu.even_row_bounds(n, world)
    for r in range(world):
        r0, r1 = bounds[r], bounds[r + 1]
        if r1 == r0:
            continue
        stack = [np.ascontiguousarray(np.stack([st[x][r0:r1] for st in states])) if states[0][x] is not None else None
                 for x in range(3)]
        idx, d0, d1 = skl.knn_merge_states(gpu_ctx, stack[0], stack[1], stack[2], ani=dist == "ani")
        assert np.array_equal(idx, whole[0][r0:r1]) and np.array_equal(d0, whole[1][r0:r1])
        if dist == "coreacc":
            assert np.array_equal(d1, whole[2][r0:r1])


def test_merge_states_folds_more_than_one_launch(skl, gpu_ctx):
    """knn = 1500: two states fill the 4096-item sort, so 5 states are folded in several launches."""
    rng = np.random.default_rng(11)
    n_states, rows, knn = 5, 9, 1500
    keys = rng.integers(0x80000000, 0xBF800000, size=(n_states, rows, knn), dtype=np.uint32)
    keys.sort(axis=2)
    ids = rng.permutation(n_states * rows * knn).astype(np.uint32).reshape(n_states, rows, knn)
    idx, d0, _ = skl.knn_merge_states(gpu_ctx, keys, ids, None)
    for r in range(rows):
        union = sorted(zip(keys[:, r].ravel().tolist(), ids[:, r].ravel().tolist()))[:knn]
        assert idx[r].tolist() == [j for _, j in union]
        exp = (np.array([k for k, _ in union], dtype=np.uint32) & np.uint32(0x7FFFFFFF)).view(np.float32)
        assert np.array_equal(d0[r], exp)


def test_partial_rejects_bad_band_lists(skl, gpu_ctx):
    kmers, ss64, n = [21], 2, 100
    g = gpu_ctx.sketches(synth.set_u(n, 1, ss64), n, kmers, ss64)
    p = g.set_k(21)
    for bands in ([3, 2], [0, 0], [7]):
        with pytest.raises(skl.SklError) as e:
            skl.self_dists_knn_partial(gpu_ctx, g, p, 5, 16, bands)
        assert e.value.code == skl.ERR_INVALID_ARG
    assert skl.knn_band_rows(g, p, 8) == skl.knn_band_rows(g, p, 8) > 0


def test_streaming_merge_against_the_radix_select(oracle, skl, gpu_ctx, monkeypatch):
    """The one-pass streaming top-k and the radix-select-only form (SKL_TOPK_STREAM=0) on data
    that exercises growth, overflow + rescan and the give-up branch: keys in random order, in
    ascending order (nothing qualifies after the first segment) and in descending order (every
    key qualifies: the segments shrink and the kernel falls back to the select)."""
    kmers, ss64, knn = [21], 8, 25
    rng = np.random.default_rng(17)
    base = synth.set_u(1, 1, ss64)[0]
    n = 9000
    # sample s differs from sample 0 in m(s) bins: distance to sample 0 is a known, strictly
    # monotone function of m
    order = {"random": rng.permutation(n - 1) % 500, "ascending": np.arange(n - 1) * 500 // (n - 1),
             "descending": 499 - np.arange(n - 1) * 500 // (n - 1)}
    for name, m_of in order.items():
        vals = np.zeros((n, ss64 * 64), dtype=np.uint16)
        for s_ in range(1, n):
            vals[s_, :int(m_of[s_ - 1]) + 1] = 1 + (s_ % 3)      # that many bins differ from sample 0 (all zero)
        bins = synth.bitslice(vals).reshape(n, -1)
        g = gpu_ctx.sketches(bins, n, kmers, ss64)
        q = gpu_ctx.sketches(bins[:1].copy(), 1, kmers, ss64)
        got = {}
        for stream in ("1", "0"):
            monkeypatch.setenv("SKL_TOPK_STREAM", stream)
            gpu_ctx.reload_env()
            got[stream] = skl.cross_dists_knn(gpu_ctx, g, q, g.set_k(21), knn)
        assert np.array_equal(got["1"][0], got["0"][0]) and np.array_equal(got["1"][1], got["0"][1]), name
        o_r, o_q = oracle.Sketches(bins, n, kmers, ss64), oracle.Sketches(bins[:1].copy(), 1, kmers, ss64)
        exp = oracle.cross_dists_knn(o_r, o_q, knn, oracle.JACCARD, 0, False, ties=oracle.TIES_CANONICAL)
        assert np.array_equal(got["1"][0], exp["idx"]), name


@pytest.mark.parametrize("knn", [1000, 1900, 2048])
def test_large_knn(oracle, skl, gpu_ctx, monkeypatch, knn):
    """knn close to the 2048 the LDS buffer holds: little or no room behind the state, so the
    streaming merge hands over to the radix select; both drivers against the oracle."""
    kmers, ss64, n = [21], 2, 2300
    bins = synth.set_r(n, kmers, ss64, n_clusters=3)
    o, g = oracle.Sketches(bins, n, kmers, ss64), gpu_ctx.sketches(bins, n, kmers, ss64)
    exp = oracle.self_dists_knn(o, knn, oracle.JACCARD, 0, False, ties=oracle.TIES_CANONICAL, threads=8)
    for symmetric in (True, False):
        idx, d0 = _knn(skl, gpu_ctx, g, g.set_k(21), knn, monkeypatch, 512, symmetric)
        assert np.array_equal(idx, exp["idx"]), symmetric
        assert np.array_equal(d0, exp["d0"])


FUZZ_SEEDS = int(__import__("os").environ.get("SKL_FUZZ_SEEDS", "16"))


@pytest.mark.parametrize("seed", range(FUZZ_SEEDS))
def test_random_knn_configuration(oracle, skl, gpu_ctx, monkeypatch, seed):
    """Random sample count, sketch size, k-mer list, knn, band height, key type and completeness:
    the one-evaluation driver, the row-by-row driver and a 3-way split into partial states
    against the oracle."""
    from sketchlib.rust_amd import multi_gpu
    rng = np.random.default_rng(9000 + seed)
    nk = int(rng.integers(2, 7))
    kmers = sorted(rng.choice(np.arange(9, 60), size=nk, replace=False).tolist())
    ss64 = int(rng.choice([1, 2, 5, 16, 33]))
    n = int(rng.integers(40, 700))
    knn = int(rng.integers(1, min(n - 1, 70) + 1))
    band_rows = int(rng.integers(3, n))
    coreacc = rng.random() < 0.4
    ani = (not coreacc) and rng.random() < 0.4
    comp = rng.uniform(0.6, 1.0, n) if rng.random() < 0.3 else None
    bins = synth.set_r(n, kmers, ss64, n_clusters=int(rng.integers(1, 12)), seed=77 + seed)
    o, g = oracle.Sketches(bins, n, kmers, ss64, comp), gpu_ctx.sketches(bins, n, kmers, ss64, comp)
    k_idx = int(rng.integers(0, nk))
    p = g.set_k() if coreacc else g.set_k(kmers[k_idx], ani)
    exp = oracle.self_dists_knn(o, knn, oracle.COREACC if coreacc else oracle.JACCARD, 0 if coreacc else k_idx, ani,
                                ties=oracle.TIES_CANONICAL, threads=8)
    runs = {"once": _knn(skl, gpu_ctx, g, p, knn, monkeypatch, band_rows, True, with_d1=True),
            "rows": _knn(skl, gpu_ctx, g, p, knn, monkeypatch, band_rows, False, with_d1=True)}
    deal = multi_gpu.knn_band_deal((n + band_rows - 1) // band_rows, 3)
    states = [skl.self_dists_knn_partial(gpu_ctx, g, p, knn, band_rows, deal[r]) for r in range(3)]
    stack = [np.ascontiguousarray(np.stack([st[x] for st in states])) if states[0][x] is not None else None
             for x in range(3)]
    runs["split"] = skl.knn_merge_states(gpu_ctx, stack[0], stack[1], stack[2], ani=ani)
    for name, (idx, d0, d1) in runs.items():
        np.testing.assert_allclose(d0, exp["d0"], atol=TOL, rtol=0, err_msg=name)
        if comp is None:
            assert np.array_equal(idx, exp["idx"]), name
            if coreacc:
                np.testing.assert_allclose(d1, exp["d1"], atol=TOL, rtol=0, err_msg=name)
    assert np.array_equal(runs["once"][0], runs["rows"][0]) and np.array_equal(runs["once"][0], runs["split"][0])


@pytest.mark.parametrize("shape", ["1651", "3254", "3255"])
def test_other_tile_shapes_turn_their_tiles_too(oracle, skl, gpu_ctx, monkeypatch, shape):
    """The turned second store for the forms of the A/B build (the round-2/3 forms of the 16- and 32-row tiles),
    single-k and core/accessory records."""
    import sketchlib.rust_amd as pkg

    kmers, ss64, n, knn = [15, 19, 23], 4, 301, 8
    bins = synth.set_r(n, kmers, ss64, n_clusters=6)
    o = oracle.Sketches(bins, n, kmers, ss64)
    monkeypatch.setenv("SKL_KSLICE_SHAPE", shape)
    with skl.using_library(pkg.build_ab_library()):
        ctx = skl.Context(0)
        ctx.set_knn_ties(skl.TIES_CANONICAL)
        g = ctx.sketches(bins, n, kmers, ss64)
        for p, oargs in ((g.set_k(19), (oracle.JACCARD, 1, False)), (g.set_k(), (oracle.COREACC, 0, False))):
            idx, d0, d1 = _knn(skl, ctx, g, p, knn, monkeypatch, 40, True, with_d1=True)
            assert f"R={shape[:2]}, JL=2" in ctx.last_kernel()
            exp = oracle.self_dists_knn(o, knn, *oargs, ties=oracle.TIES_CANONICAL, threads=8)
            assert np.array_equal(idx, exp["idx"])
            np.testing.assert_allclose(d0, exp["d0"], atol=TOL, rtol=0)
            if oargs[0] == oracle.COREACC:
                np.testing.assert_allclose(d1, exp["d1"], atol=TOL, rtol=0)
        g.close()
        ctx.close()
