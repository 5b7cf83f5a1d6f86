"""kNN output against the reference BINARY's tie order, and neighbour lists longer than the LDS forms hold.

The reference keeps a row's neighbours in std::collections::BinaryHeap through push_heap (strict `<` against
the heap's maximum, src/distances/mod.rs:41-48), candidates j ascending (mod.rs:156-181, :335-369), and prints
into_sorted_vec: with equal keys -- every genome with fewer than knn relatives ties at 1.0, single-k Jaccard
values are quantised -- which ids survive and in what order is decided by the heap's history.  The oracle
replays that heap (ties=TIES_RUST_HEAP); `skl_ctx_set_knn_ties(SKL_KNN_TIES_REFERENCE)` must reproduce ids,
order and distances exactly.  The default (canonical) mode must be unchanged.

knn is bounded only by the candidates there are (src/lib.rs:379-382, mod.rs:325): > 2048 goes through global
memory."""
import numpy as np
import pytest

from sketchlib.rust_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ref_ties(skl, gpu_ctx):
    gpu_ctx.set_knn_ties(skl.TIES_REFERENCE)
    yield
    gpu_ctx.set_knn_ties(skl.TIES_CANONICAL)


def _check_self(oracle, skl, ctx, bins, n, kmers, ss64, knn, dist, ties):
    o, g = oracle.Sketches(bins, n, kmers, ss64), ctx.sketches(bins, n, kmers, ss64)
    if dist == "coreacc":
        p, oargs = g.set_k(), (oracle.COREACC, 0, False)
    else:
        p, oargs = g.set_k(kmers[1 if len(kmers) > 1 else 0], dist == "ani"), (oracle.JACCARD, 1 if len(kmers) > 1 else 0, dist == "ani")
    idx, d0, d1 = skl.self_dists_knn(ctx, g, p, knn)
    exp = oracle.self_dists_knn(o, knn, *oargs, ties=ties, threads=8)
    assert np.array_equal(idx, exp["idx"]), np.argwhere(idx != exp["idx"])[:5]
    assert np.array_equal(d0, exp["d0"])
    if dist == "coreacc":
        assert np.array_equal(d1, exp["d1"])
    g.close()
    return idx


@pytest.mark.parametrize("knn", [1, 7, 50])
@pytest.mark.parametrize("dist", ["jaccard", "ani", "coreacc"])
def test_reference_order_on_related_genomes(oracle, skl, gpu_ctx, ref_ties, knn, dist):
    """Clusters smaller than knn: every row's tail ties at 1.0 (or at (1, 1)), and inside a cluster the quantised
    single-k distances repeat."""
    kmers, ss64, n = [17, 21, 25, 29], 4, 600
    bins = synth.set_r(n, kmers, ss64, n_clusters=40)
    _check_self(oracle, skl, gpu_ctx, bins, n, kmers, ss64, knn, dist, oracle.TIES_RUST_HEAP)


@pytest.mark.parametrize("knn", [1, 7, 50, 199])
def test_reference_order_when_every_key_ties(oracle, skl, gpu_ctx, ref_ties, knn):
    """n copies of one sketch: the neighbour lists are the heap's history and nothing else -- and they are NOT
    the canonical lists (lowest index first), which is why the mode exists."""
    kmers, ss64, n = [21], 4, 200
    bins = np.tile(synth.set_u(1, 1, ss64), (n, 1))
    idx = _check_self(oracle, skl, gpu_ctx, bins, n, kmers, ss64, knn, "jaccard", oracle.TIES_RUST_HEAP)
    if 1 < knn < n - 1:
        canon = np.array([[j for j in range(n) if j != r][:knn] for r in range(n)], dtype=np.uint64)
        assert not np.array_equal(idx, canon)


def test_reference_order_on_random_sketches(oracle, skl, gpu_ctx, ref_ties):
    """Set U at BASELINE configs[4]'s sketch shape (sketchsize64 = 32, one of k = {13..29}): nearly every key is 1.0,
    a few pairs match a bin by chance -- long runs of rejected candidates between accepted ones."""
    kmers, ss64, n, knn = [13, 17, 21, 25, 29], 32, 3000, 50
    bins = synth.set_u(n, len(kmers), ss64)
    o, g = oracle.Sketches(bins, n, kmers, ss64), gpu_ctx.sketches(bins, n, kmers, ss64)
    idx, d0, _ = skl.self_dists_knn(gpu_ctx, g, g.set_k(21), knn)
    exp = oracle.self_dists_knn(o, knn, oracle.JACCARD, 2, False, ties=oracle.TIES_RUST_HEAP, threads=8)
    assert np.array_equal(idx, exp["idx"]) and np.array_equal(d0, exp["d0"])
    g.close()


@pytest.mark.parametrize("dist", ["jaccard", "coreacc"])
def test_reference_order_cross(oracle, skl, gpu_ctx, ref_ties, dist):
    kmers, ss64, nr, nq, knn = [17, 21, 25, 29], 4, 700, 90, 20
    rb = synth.set_r(nr, kmers, ss64, n_clusters=30)
    qb = synth.set_r(nq, kmers, ss64, n_clusters=30, first_sample=5000)
    o_r, o_q = oracle.Sketches(rb, nr, kmers, ss64), oracle.Sketches(qb, nq, kmers, ss64)
    g_r, g_q = gpu_ctx.sketches(rb, nr, kmers, ss64), gpu_ctx.sketches(qb, nq, kmers, ss64)
    p, oargs = (g_r.set_k(), (oracle.COREACC, 0, False)) if dist == "coreacc" else (g_r.set_k(21), (oracle.JACCARD, 1, False))
    idx, d0, d1 = skl.cross_dists_knn(gpu_ctx, g_r, g_q, p, knn)
    exp = oracle.cross_dists_knn(o_r, o_q, knn, *oargs, ties=oracle.TIES_RUST_HEAP, threads=8)
    assert np.array_equal(idx, exp["idx"]) and np.array_equal(d0, exp["d0"])
    if dist == "coreacc":
        assert np.array_equal(d1, exp["d1"])


def test_several_bands_and_row_ranges(oracle, skl, gpu_ctx, ref_ties, set_switch):
    """The reference order does not depend on how the rows are cut into bands or calls."""
    kmers, ss64, n, knn = [21], 4, 500, 9
    bins = synth.set_r(n, kmers, ss64, n_clusters=50)
    o, g = oracle.Sketches(bins, n, kmers, ss64), gpu_ctx.sketches(bins, n, kmers, ss64)
    exp = oracle.self_dists_knn(o, knn, oracle.JACCARD, 0, False, ties=oracle.TIES_RUST_HEAP, threads=8)
    set_switch("SKL_KNN_BAND_ROWS", 37)
    gpu_ctx.set_knn_ties(skl.TIES_REFERENCE)      # (reload_env keeps the mode; set again to be explicit)
    idx, d0, _ = skl.self_dists_knn(gpu_ctx, g, g.set_k(21), knn)
    assert np.array_equal(idx, exp["idx"]) and np.array_equal(d0, exp["d0"])
    idx, d0, _ = skl.self_dists_knn(gpu_ctx, g, g.set_k(21), knn, 123, 301)
    assert np.array_equal(idx, exp["idx"][123:301]) and np.array_equal(d0, exp["d0"][123:301])
    g.close()


@pytest.mark.ab_library
@pytest.mark.parametrize("dist", ["jaccard", "ani", "coreacc"])
@pytest.mark.parametrize("knn,band,flags,wave", [(1, 48, "1", "1"), (7, 64, "1", "1"), (50, 100, "1", "1"), (7, 37, "0", "1"), (50, 64, "1", "0"),
                                                 (300, 96, "1", "1")])
def test_one_evaluation_driver_in_reference_order(oracle, skl, gpu_ctx, ref_ties, set_switch, dist, knn, band, flags, wave):
    """The whole self matrix with every pair evaluated ONCE (round 4): the heap of a row lives in global memory between the
    bands and is fed the row's candidates in ascending id -- turned from the bands above its own, then its own band's
    columns -- so it goes through the reference's states.  Band heights that are / are not multiples of the tile, with
    and without the pair kernel's row and block flags."""
    kmers, ss64, n = [17, 21, 25, 29], 4, 600
    bins = synth.set_r(n, kmers, ss64, n_clusters=40)
    set_switch("SKL_KNN_BAND_ROWS", band)
    set_switch("SKL_KNN_ROW_FLAGS", flags)
    set_switch("SKL_REFHEAP_WAVE", wave)     # one wave per row (knn <= 256; default) / one workgroup per row (also what knn = 300 takes)
    gpu_ctx.set_knn_ties(skl.TIES_REFERENCE)
    _check_self(oracle, skl, gpu_ctx, bins, n, kmers, ss64, knn, dist, oracle.TIES_RUST_HEAP)
    assert "k-sliced" in gpu_ctx.last_kernel() or "all k" in gpu_ctx.last_kernel()
    # the row-by-row form (every pair twice, as the reference does) gives the same lists
    set_switch("SKL_KNN_SYMMETRIC", "0")
    gpu_ctx.set_knn_ties(skl.TIES_REFERENCE)
    _check_self(oracle, skl, gpu_ctx, bins, n, kmers, ss64, knn, dist, oracle.TIES_RUST_HEAP)


@pytest.mark.parametrize("knn", [1, 7, 50, 199])
def test_one_evaluation_driver_when_every_key_ties(oracle, skl, gpu_ctx, ref_ties, set_switch, knn):
    kmers, ss64, n = [21], 4, 200
    bins = np.tile(synth.set_u(1, 1, ss64), (n, 1))
    set_switch("SKL_KNN_BAND_ROWS", 32)
    gpu_ctx.set_knn_ties(skl.TIES_REFERENCE)
    _check_self(oracle, skl, gpu_ctx, bins, n, kmers, ss64, knn, "jaccard", oracle.TIES_RUST_HEAP)


def test_one_evaluation_reference_order_on_random_sketches(oracle, skl, gpu_ctx, ref_ties):
    """Default band rule (no switch): 12 000 random sketches are enough for the one-evaluation driver to be chosen."""
    kmers, ss64, n, knn = [13, 17, 21], 8, 12000, 20
    bins = synth.set_u(n, len(kmers), ss64)
    o, g = oracle.Sketches(bins, n, kmers, ss64), gpu_ctx.sketches(bins, n, kmers, ss64)
    gpu_ctx.timing_enable()
    gpu_ctx.timing_reset()
    idx, d0, _ = skl.self_dists_knn(gpu_ctx, g, g.set_k(17), knn)
    _ms, launches = gpu_ctx.kernel_ms()
    assert launches >= 3, launches         # several bands
    exp = oracle.self_dists_knn(o, knn, oracle.JACCARD, 1, False, ties=oracle.TIES_RUST_HEAP, threads=8)
    assert np.array_equal(idx, exp["idx"]) and np.array_equal(d0, exp["d0"])
    g.close()


def test_default_mode_is_unchanged(oracle, skl, gpu_ctx):
    kmers, ss64, n, knn = [21], 4, 200, 7
    bins = np.tile(synth.set_u(1, 1, ss64), (n, 1))
    g = gpu_ctx.sketches(bins, n, kmers, ss64)
    idx, _d0, _ = skl.self_dists_knn(gpu_ctx, g, g.set_k(21), knn)
    for row in range(n):
        assert idx[row].tolist() == [j for j in range(n) if j != row][:knn]
    with pytest.raises(skl.SklError):
        gpu_ctx.set_knn_ties(7)
    g.close()


# ---- more neighbours than the LDS forms hold ----

@pytest.mark.parametrize("knn", [2049, 3000, 4999])
@pytest.mark.parametrize("mode", ["canonical", "reference"])
def test_knn_beyond_2048(oracle, skl, gpu_ctx, knn, mode):
    """`dist --knn 3000` on a 5 000-sample database works in the reference (only clamped to n - 1)."""
    kmers, ss64, n = [21], 2, 5000
    bins = synth.set_r(n, kmers, ss64, n_clusters=7)
    gpu_ctx.set_knn_ties(skl.TIES_REFERENCE if mode == "reference" else skl.TIES_CANONICAL)
    try:
        _check_self(oracle, skl, gpu_ctx, bins, n, kmers, ss64, knn, "jaccard",
                    oracle.TIES_RUST_HEAP if mode == "reference" else oracle.TIES_CANONICAL)
    finally:
        gpu_ctx.set_knn_ties(skl.TIES_CANONICAL)


def test_knn_beyond_2048_core_accessory_and_cross(oracle, skl, gpu_ctx):
    kmers, ss64, nr, nq, knn = [17, 21, 25], 2, 2600, 40, 2500
    rb = synth.set_r(nr, kmers, ss64, n_clusters=5)
    qb = synth.set_r(nq, kmers, ss64, n_clusters=5, first_sample=9000)
    o_r, o_q = oracle.Sketches(rb, nr, kmers, ss64), oracle.Sketches(qb, nq, kmers, ss64)
    g_r, g_q = gpu_ctx.sketches(rb, nr, kmers, ss64), gpu_ctx.sketches(qb, nq, kmers, ss64)
    idx, d0, d1 = skl.cross_dists_knn(gpu_ctx, g_r, g_q, g_r.set_k(), knn)
    exp = oracle.cross_dists_knn(o_r, o_q, knn, oracle.COREACC, 0, False, ties=oracle.TIES_CANONICAL, threads=8)
    assert np.array_equal(idx, exp["idx"]) and np.array_equal(d0, exp["d0"]) and np.array_equal(d1, exp["d1"])
    idx, d0, d1 = skl.self_dists_knn(gpu_ctx, g_r, g_r.set_k(), knn)
    exp = oracle.self_dists_knn(o_r, knn, oracle.COREACC, 0, False, ties=oracle.TIES_CANONICAL, threads=8)
    assert np.array_equal(idx, exp["idx"]) and np.array_equal(d0, exp["d0"]) and np.array_equal(d1, exp["d1"])


def test_candidate_lists_with_knn_beyond_2048(oracle, skl, gpu_ctx):
    """The precluster path (ragged candidate rows): rows with more than 2 048 candidates keep them all."""
    kmers, ss64, n, knn = [21], 2, 3000, 2400
    bins = synth.set_r(n, kmers, ss64, n_clusters=4)
    o, g = oracle.Sketches(bins, n, kmers, ss64), gpu_ctx.sketches(bins, n, kmers, ss64)
    rng = np.random.default_rng(5)
    lists = [np.sort(rng.choice(np.delete(np.arange(n), r), size=int(rng.integers(0, n - 1)), replace=False)).astype(np.uint32)
             if r % 3 else np.delete(np.arange(n, dtype=np.uint32), r) for r in range(0, n, 50)]
    rows = list(range(0, n, 50))
    offs = np.zeros(n + 1, dtype=np.uint64)
    for r, l in zip(rows, lists):
        offs[r + 1] = len(l)
    offs = np.cumsum(offs).astype(np.uint64)
    cand = np.concatenate(lists) if lists else np.zeros(0, dtype=np.uint32)
    idx, d0 = skl.self_dists_knn_candidates(gpu_ctx, g, g.set_k(21), knn, offs, cand)
    dense = oracle.self_dists_all(o, oracle.JACCARD, 0, False, threads=8)[:, 0]

    def dist(i, j):
        a, b = (i, j) if i < j else (j, i)
        return dense[n * a - a * (a + 1) // 2 + b - 1 - a]

    for r, l in zip(rows, lists):
        keys = np.array([dist(r, int(j)) for j in l], dtype=np.float32)
        order = np.lexsort((l, keys))[:knn]
        m = len(order)
        assert np.array_equal(idx[r, :m], l[order].astype(np.uint64)), r
        assert np.array_equal(d0[r, :m], keys[order]), r
        assert np.all(idx[r, m:] == r) and np.all(d0[r, m:] == 1.0)      # padding (mod.rs:535-546)


FUZZ_SEEDS = int(__import__("os").environ.get("SKL_FUZZ_SEEDS", "12"))


@pytest.mark.parametrize("seed", range(FUZZ_SEEDS))
def test_random_reference_tie_configuration(oracle, skl, gpu_ctx, seed):
    """Seeded fuzz of the heap replay: random sizes, knn, sketch sizes (small sketches quantise the distances into few
    levels: long runs of equal keys), cluster structure, self / cross, all three key types, several row bands."""
    rng = np.random.default_rng(1000 + seed)
    ss64 = int(rng.choice([1, 2, 4, 16]))
    kmers = [[21], [17, 21, 25], [15, 19, 23, 27, 31]][int(rng.integers(0, 3))]
    n = int(rng.integers(40, 900))
    knn = int(rng.integers(1, min(n - 1, 130)))
    bins = synth.set_r(n, kmers, ss64, n_clusters=int(rng.integers(1, 40)))
    if rng.random() < 0.3:          # exact duplicates: keys of 0 that tie with each other
        src = rng.integers(0, n, n // 4)
        dst = rng.integers(0, n, n // 4)
        bins = bins.reshape(n, -1).copy()
        bins[dst] = bins[src]
    dist = ["jaccard", "ani", "coreacc"][int(rng.integers(0, 3))]
    if dist == "coreacc" and len(kmers) < 2:
        dist = "jaccard"
    k_idx = int(rng.integers(0, len(kmers)))
    os_env = __import__("os").environ
    old = os_env.get("SKL_KNN_BAND_ROWS")
    os_env["SKL_KNN_BAND_ROWS"] = str(int(rng.integers(5, 400)))
    gpu_ctx.reload_env()
    gpu_ctx.set_knn_ties(skl.TIES_REFERENCE)
    try:
        o, g = oracle.Sketches(bins, n, kmers, ss64), gpu_ctx.sketches(bins, n, kmers, ss64)
        p, oargs = ((g.set_k(), (oracle.COREACC, 0, False)) if dist == "coreacc"
                    else (g.set_k(kmers[k_idx], dist == "ani"), (oracle.JACCARD, k_idx, dist == "ani")))
        if rng.random() < 0.5:
            idx, d0, d1 = skl.self_dists_knn(gpu_ctx, g, p, knn)
            exp = oracle.self_dists_knn(o, knn, *oargs, ties=oracle.TIES_RUST_HEAP, threads=8)
        else:
            nq = int(rng.integers(1, 120))
            qb = synth.set_r(nq, kmers, ss64, n_clusters=int(rng.integers(1, 40)), first_sample=int(rng.integers(0, n)))
            oq, gq = oracle.Sketches(qb, nq, kmers, ss64), gpu_ctx.sketches(qb, nq, kmers, ss64)
            knn = min(knn, n)
            idx, d0, d1 = skl.cross_dists_knn(gpu_ctx, g, gq, p, knn)
            exp = oracle.cross_dists_knn(o, oq, knn, *oargs, ties=oracle.TIES_RUST_HEAP, threads=8)
        assert np.array_equal(idx, exp["idx"]), (seed, n, knn, dist, ss64)
        assert np.array_equal(d0, exp["d0"])
        if dist == "coreacc":
            assert np.array_equal(d1, exp["d1"])
    finally:
        gpu_ctx.set_knn_ties(skl.TIES_CANONICAL)
        if old is None:
            os_env.pop("SKL_KNN_BAND_ROWS", None)
        else:
            os_env["SKL_KNN_BAND_ROWS"] = old
        gpu_ctx.reload_env()


@pytest.mark.parametrize("world,band_rows", [(1, 64), (2, 64), (3, 64), (5, 64), (3, 48), (4, 100), (2, 16)])
@pytest.mark.parametrize("dist", ["jaccard", "ani", "coreacc"])
def test_reference_order_over_column_windows(oracle, skl, gpu_ctx, monkeypatch, world, band_rows, dist):
    """skl_self_dists_knn_window: the reference's lists from heaps that travel through `world` column windows, every pair
    evaluated once.  The participants run one after the other here on one device and one set of heap arrays -- which is
    exact: a participant never touches a row again after the band that holds it, so the state it would have sent on is the
    state it leaves behind (the transport itself: tests/test_multi_gpu_cpu.py over gloo, tests/test_bench_gpu.py)."""
    import torch
    from sketchlib.rust_amd import multi_gpu

    # (band heights that are / are not multiples of the tile and of the 64-column block: a window then starts inside a block)
    kmers, ss64, n, knn = [17, 21, 25, 29], 8, 613, 9
    bins = synth.set_r(n, kmers, ss64, n_clusters=5)
    bins[400] = bins[3]
    bins[401] = bins[3]
    o, g = oracle.Sketches(bins, n, kmers, ss64), gpu_ctx.sketches(bins, n, kmers, ss64)
    if dist == "coreacc":
        p, oargs = g.set_k(), (oracle.COREACC, 0, False)
    else:
        p, oargs = g.set_k(21, dist == "ani"), (oracle.JACCARD, 1, dist == "ani")
    monkeypatch.setenv("SKL_TILE32_MIN", "0")       # the prunable 32 x 128 form
    gpu_ctx.reload_env()
    dev = torch.device("cuda", 0)
    heaps = skl.knn_heaps_alloc(n, knn, dist == "coreacc", dev)
    cuts = multi_gpu.knn_window_cuts(n, band_rows, world)
    for r in range(world):
        lo, hi = cuts[r], cuts[r + 1]
        for band in range((n + band_rows - 1) // band_rows):
            if band * band_rows >= hi:
                break
            skl.self_dists_knn_window(gpu_ctx, g, p, knn, band_rows, band, lo, hi, heaps)
    idx, d0, d1 = skl.knn_heaps_finalize(gpu_ctx, heaps, 0, n, knn, ani=dist == "ani")
    gpu_ctx.synchronize()
    exp = oracle.self_dists_knn(o, knn, *oargs, ties=oracle.TIES_RUST_HEAP, threads=8)
    assert np.array_equal(idx.cpu().numpy().astype(np.uint64), exp["idx"]), np.argwhere(idx.cpu().numpy() != exp["idx"])[:5]
    assert np.array_equal(d0.cpu().numpy(), exp["d0"])
    if dist == "coreacc":
        assert np.array_equal(d1.cpu().numpy(), exp["d1"])
    g.close()


@pytest.mark.parametrize("coreacc", [False, True])
@pytest.mark.parametrize("world,band_rows,knn,cap", [(1, 64, 7, 64), (3, 64, 5, 64), (2, 96, 40, 256), (5, 48, 12, 128), (3, 64, 300, 1024)])
def test_decoupled_column_windows_logs_replayed_in_window_order(oracle, skl, gpu_ctx, world, band_rows, knn, cap, coreacc):
    """skl_self_dists_knn_window_logged + skl_knn_heaps_replay (round 6): every participant's window run against heaps that START
    EMPTY, one participant after the other on the one device, each with its own heaps and accept logs; the replay of the logs in
    window order into one empty heap per row gives the oracle's whole-row BinaryHeap replay -- ids, order, both distances --
    for the one-wave and (knn = 300) the one-workgroup form of the heap kernels, with exact ties in the data."""
    import torch
    from sketchlib.rust_amd import multi_gpu

    kmers, ss64, n = [15, 19, 23, 27, 31], 16, 700
    bins = synth.set_r(n, kmers, ss64, n_clusters=4)
    bins[40] = bins[7]
    bins[399] = bins[7]
    bins[400] = bins[7]
    o, g = oracle.Sketches(bins, n, kmers, ss64), gpu_ctx.sketches(bins, n, kmers, ss64)
    p = g.set_k() if coreacc else g.set_k(23)
    dev = torch.device("cuda", 0)
    cuts = multi_gpu.knn_window_cuts(n, band_rows, world)
    logs = []
    for r in range(world):
        heaps = skl.knn_heaps_alloc(n, knn, coreacc, dev)
        lg = skl.knn_logs_alloc(n, cap, coreacc, dev)
        for band in range((n + band_rows - 1) // band_rows):
            if band * band_rows >= cuts[r + 1]:
                break
            skl.self_dists_knn_window_logged(gpu_ctx, g, p, knn, band_rows, band, cuts[r], cuts[r + 1], heaps, lg)
        gpu_ctx.synchronize()
        assert int(lg["len"].max()) <= cap, "the test's logs are meant to hold"
        assert int(lg["len"][cuts[r + 1]:].sum()) == 0       # rows behind the window meet none of its pairs
        logs.append(lg)
    final = skl.knn_heaps_alloc(n, knn, coreacc, dev)
    for lg in logs:
        m = max(1, int(lg["len"].max()))
        rec, ids = lg["rec"][:, :m].contiguous(), lg["id"][:, :m].contiguous()
        torch.cuda.synchronize()      # (torch cuts the logs on ITS stream; the session's context runs on a stream of its own)
        skl.knn_heaps_replay(gpu_ctx, final, 0, n, knn, rec, ids, lg["len"])
        gpu_ctx.synchronize()
    idx, d0, d1 = skl.knn_heaps_finalize(gpu_ctx, final, 0, n, knn)
    gpu_ctx.synchronize()
    exp = oracle.self_dists_knn(o, knn, oracle.COREACC if coreacc else oracle.JACCARD, 0 if coreacc else 2, False, ties=oracle.TIES_RUST_HEAP, threads=8)
    assert np.array_equal(idx.cpu().numpy().astype(np.uint64), exp["idx"]), np.argwhere(idx.cpu().numpy() != exp["idx"])[:5]
    assert np.array_equal(d0.cpu().numpy().view(np.uint32), exp["d0"].view(np.uint32))
    if coreacc:
        assert np.array_equal(d1.cpu().numpy().view(np.uint32), exp["d1"].view(np.uint32))
    g.close()
