"""Tile pruning of the one-evaluation self kNN (pair_kslice_walk.inc, PRUNE; capi_knn.cpp).

With single-k keys the key is monotone in the mismatch count, so a 32 x 128 tile all of whose pairs are -- on the chunks
walked so far -- beyond both their samples' current knn-th best is left unfinished and unwritten.  The bar: the neighbour
lists (ids, order, distances) are those of the oracle in BOTH tie rules, with and without pruning, and tiles really are
skipped on data where every genome has more than knn relatives."""
import numpy as np
import pytest

from sketchlib.rust_amd import synth

pytestmark = [pytest.mark.gpu, pytest.mark.ab_library]   # (SKL_KNN_SYMMETRIC / SKL_KNN_PRUNE ...: switches of the A/B build)


def _clustered(n, nk, ss64, n_clusters, keep=0.94, shuffle=None):
    """Sample s belongs to cluster s % n_clusters and keeps each of the cluster's bin values with probability `keep`
    (J ~ 0.8 inside a cluster, ~0 outside): synth.set_clustered_device, brought to the host for the oracle.  `shuffle` (a
    seed): the samples in a random order -- relatives at random ids instead of regular id distances."""
    import torch

    t = synth.set_clustered_device(n, nk, ss64, torch.device("cuda", 0), keep=keep, n_clusters=n_clusters)
    bins = t.cpu().numpy().view(np.uint64)
    if shuffle is not None:
        bins = np.ascontiguousarray(bins[np.random.default_rng(shuffle).permutation(n)])
    return bins


def _run(skl, ctx, g, p, knn, monkeypatch, band_rows, prune=True, ties=None, sparse=True):
    monkeypatch.setenv("SKL_KNN_SPARSE", "1" if sparse else "0")
    monkeypatch.setenv("SKL_KNN_BAND_ROWS", str(band_rows))
    monkeypatch.setenv("SKL_TILE32_MIN", "0")          # 32 x 128 tiles whatever the launch size (the prunable form)
    monkeypatch.setenv("SKL_KNN_PRUNE", "1" if prune else "0")
    ctx.reload_env()
    if ties is not None:
        ctx.set_knn_ties(ties)
    try:
        idx, d0, _ = skl.self_dists_knn(ctx, g, p, knn)
    finally:
        ctx.set_knn_ties(skl.TIES_CANONICAL)
    return idx, d0, ctx.knn_prune_stats()


@pytest.mark.parametrize("ties", ["canonical", "reference"])
@pytest.mark.parametrize("ani", [False, True], ids=["dist", "ani"])
@pytest.mark.parametrize("band_rows,ss64", [(32, 32), (64, 32), (96, 16), (160, 37), (64, 157)])
def test_pruned_run_equals_the_oracle(oracle, skl, gpu_ctx, monkeypatch, ties, ani, band_rows, ss64):
    """240 clusters of 6 genomes (sample s belongs to cluster s % 240: a genome's relatives sit at id distances 240, 480, ...),
    knn = 3: once a row has seen three relatives its list holds relatives only, and a tile whose id distances miss every
    multiple of 240 is hopeless from its first or second stage on.  Sketch sizes: whole stages (32, 16 chunks), a ragged
    last stage (37), the reference's `-s 10000` (157)."""
    kmers, n, knn = [17, 21, 25], 1440, 3
    bins = _clustered(n, len(kmers), ss64, 240)
    o, g = oracle.Sketches(bins, n, kmers, ss64), gpu_ctx.sketches(bins, n, kmers, ss64)
    p = g.set_k(21, ani)
    t = skl.TIES_REFERENCE if ties == "reference" else skl.TIES_CANONICAL
    idx, d0, (tiles, pruned) = _run(skl, gpu_ctx, g, p, knn, monkeypatch, band_rows, True, t)
    assert "R=32, JL=2, JACCARD, k-sliced" in gpu_ctx.last_kernel()
    exp = oracle.self_dists_knn(o, knn, oracle.JACCARD, 1, ani,
                                ties=oracle.TIES_RUST_HEAP if ties == "reference" else oracle.TIES_CANONICAL, threads=8)
    assert np.array_equal(idx, exp["idx"]), np.argwhere(idx != exp["idx"])[:5]
    assert np.array_equal(d0, exp["d0"])
    assert tiles > 0 and 0 < pruned < tiles, (tiles, pruned)
    # ... and without pruning: the same lists, nothing skipped
    idx0, d00, (_t0, pruned0) = _run(skl, gpu_ctx, g, p, knn, monkeypatch, band_rows, False, t)
    assert pruned0 == 0 and np.array_equal(idx0, idx) and np.array_equal(d00, d0)
    g.close()


@pytest.mark.parametrize("n,band_rows,ss64,ties,ani", [(8192, 64, 16, "reference", False), (8192, 96, 13, "canonical", True),
                                                        (8192, 160, 16, "canonical", False), (4096, 64, 72, "reference", True),
                                                        (4096, 64, 157, "canonical", False), (4096, 96, 157, "reference", False)])
def test_probe_survivors_take_the_sparse_walk(oracle, skl, gpu_ctx, monkeypatch, n, band_rows, ss64, ties, ani):
    """Clusters of 4 genomes at RANDOM ids, knn = 1, sets large enough that a 32 x 128 tile holds one or two related pairs
    (4 096 x 3 / n): a tile that survives the probe does so for those pairs, and is walked for their rows only (at most 4 alive
    rows; more fall back to the whole walk).  Sketch sizes: one batch of the sparse walk's row staging, waves with unequal
    chunk counts (13), one batch and a bit (72 -> 18 chunks per wave), several batches (157 -> 40)."""
    kmers, knn = [17, 21], 1
    bins = _clustered(n, len(kmers), ss64, n // 4, shuffle=77)
    o, g = oracle.Sketches(bins, n, kmers, ss64), gpu_ctx.sketches(bins, n, kmers, ss64)
    p = g.set_k(21, ani)
    t = skl.TIES_REFERENCE if ties == "reference" else skl.TIES_CANONICAL
    idx, d0, (tiles, pruned) = _run(skl, gpu_ctx, g, p, knn, monkeypatch, band_rows, True, t)
    sparse = gpu_ctx.knn_prune_stats(full=True)["tiles_sparse_walk"]
    exp = oracle.self_dists_knn(o, knn, oracle.JACCARD, 1, ani,
                                ties=oracle.TIES_RUST_HEAP if ties == "reference" else oracle.TIES_CANONICAL, threads=8)
    assert np.array_equal(idx, exp["idx"]), np.argwhere(idx != exp["idx"])[:5]
    assert np.array_equal(d0, exp["d0"])
    assert tiles > 0 and 0 < pruned < tiles and 0 < sparse <= tiles - pruned, (tiles, pruned, sparse)
    # ... the survivors walked whole instead: the same lists
    idx1, d01, _ = _run(skl, gpu_ctx, g, p, knn, monkeypatch, band_rows, True, t, sparse=False)
    assert gpu_ctx.knn_prune_stats(full=True)["tiles_sparse_walk"] == 0
    assert np.array_equal(idx1, idx) and np.array_equal(d01, d0)
    g.close()


@pytest.mark.parametrize("ties", ["canonical", "reference"])
def test_rows_with_fewer_relatives_than_knn_are_never_pruned_wrongly(oracle, skl, gpu_ctx, monkeypatch, ties):
    """Clusters SMALLER than knn: every list ends in chance matches (keys just below 1.0) and exact ties at 1.0 -- the bound
    of such a row admits every count but the very last, so its tiles are walked to the end -- next to clusters larger than
    knn whose rows do prune."""
    kmers, ss64, knn = [21], 32, 3
    big = _clustered(1200, 1, ss64, 240)                           # 5 per cluster
    small = synth.set_u(240, 1, ss64)                              # no relatives at all
    bins = np.concatenate([big[:600], small, big[600:]])
    n = bins.shape[0]
    o, g = oracle.Sketches(bins, n, kmers, ss64), gpu_ctx.sketches(bins, n, kmers, ss64)
    p = g.set_k(21)
    t = skl.TIES_REFERENCE if ties == "reference" else skl.TIES_CANONICAL
    idx, d0, (tiles, pruned) = _run(skl, gpu_ctx, g, p, knn, monkeypatch, 64, True, t)
    exp = oracle.self_dists_knn(o, knn, oracle.JACCARD, 0, False,
                                ties=oracle.TIES_RUST_HEAP if ties == "reference" else oracle.TIES_CANONICAL, threads=8)
    assert np.array_equal(idx, exp["idx"]) and np.array_equal(d0, exp["d0"])
    assert pruned < tiles
    g.close()


@pytest.mark.parametrize("ties", ["canonical", "reference"])
def test_identical_sketches_prune_everything_once_the_lists_are_full(oracle, skl, gpu_ctx, monkeypatch, ties):
    """n copies of one sketch: every key is 0, nothing is ever STRICTLY below a full list's maximum (mod.rs:42), so every
    tile met after that is hopeless before its first stage ends -- and the lists are still the heap's history."""
    kmers, ss64, n, knn = [21], 32, 300, 7
    bins = np.tile(synth.set_u(1, 1, ss64), (n, 1))
    o, g = oracle.Sketches(bins, n, kmers, ss64), gpu_ctx.sketches(bins, n, kmers, ss64)
    t = skl.TIES_REFERENCE if ties == "reference" else skl.TIES_CANONICAL
    idx, d0, (tiles, pruned) = _run(skl, gpu_ctx, g, g.set_k(21), knn, monkeypatch, 32, True, t)
    exp = oracle.self_dists_knn(o, knn, oracle.JACCARD, 0, False,
                                ties=oracle.TIES_RUST_HEAP if ties == "reference" else oracle.TIES_CANONICAL, threads=8)
    assert np.array_equal(idx, exp["idx"]) and np.array_equal(d0, exp["d0"])
    assert pruned > 0
    g.close()


def test_partial_states_with_pruning_merge_to_the_whole(oracle, skl, gpu_ctx, monkeypatch):
    """The multi-GPU split (bands dealt over 3 participants, partial states merged) prunes inside every participant's
    bands against that participant's own running lists: the merged lists are the oracle's."""
    from sketchlib.rust_amd import multi_gpu

    kmers, ss64, n, knn, band_rows = [17, 21, 25], 32, 1440, 1, 64     # (a participant sees a third of a row's relatives)
    bins = _clustered(n, len(kmers), ss64, 240)
    o, g = oracle.Sketches(bins, n, kmers, ss64), gpu_ctx.sketches(bins, n, kmers, ss64)
    p = g.set_k(21)
    monkeypatch.setenv("SKL_TILE32_MIN", "0")
    gpu_ctx.reload_env()
    deal = multi_gpu.knn_band_deal((n + band_rows - 1) // band_rows, 3)
    states, pruned_total = [], 0
    for r in range(3):
        states.append(skl.self_dists_knn_partial(gpu_ctx, g, p, knn, band_rows, deal[r]))
        pruned_total += gpu_ctx.knn_prune_stats()[1]
    key = np.stack([s[0] for s in states])
    idx = np.stack([s[1] for s in states])
    got = skl.knn_merge_states(gpu_ctx, key, idx, None)
    exp = oracle.self_dists_knn(o, knn, oracle.JACCARD, 1, False, ties=oracle.TIES_CANONICAL, threads=8)
    assert np.array_equal(got[0], exp["idx"]) and np.array_equal(got[1], exp["d0"])
    assert pruned_total > 0
    g.close()


def test_pruning_is_off_with_a_completeness_correction(oracle, skl, gpu_ctx, monkeypatch):
    """With completeness the key depends on the pair, not on the mismatch count alone: no bound, no pruning, same lists."""
    kmers, ss64, n, knn = [21], 32, 300, 7
    bins = synth.set_r(n, kmers, ss64, n_clusters=6)
    comp = np.linspace(0.7, 1.0, n)
    o = oracle.Sketches(bins, n, kmers, ss64, completeness=comp)
    g = gpu_ctx.sketches(bins, n, kmers, ss64)
    g.set_completeness(comp)
    idx, d0, (_tiles, pruned) = _run(skl, gpu_ctx, g, g.set_k(21), knn, monkeypatch, 64)
    exp = oracle.self_dists_knn(o, knn, oracle.JACCARD, 0, False, ties=oracle.TIES_CANONICAL, threads=8)
    assert pruned == 0 and np.array_equal(idx, exp["idx"])
    np.testing.assert_allclose(d0, exp["d0"], atol=1e-6, rtol=0)
    g.close()


FUZZ_SEEDS = int(__import__("os").environ.get("SKL_FUZZ_SEEDS", "10"))


@pytest.mark.parametrize("seed", range(FUZZ_SEEDS))
def test_random_pruned_configuration(oracle, skl, gpu_ctx, monkeypatch, seed):
    """Random sizes, cluster structures (relatives at regular id distances, so that whole tiles are free of them), sketch
    sizes (ragged last stages included), band heights, neighbour counts, keys and tie rules: the pruned run's lists are the
    oracle's.  A soak run sets SKL_FUZZ_SEEDS."""
    rng = np.random.default_rng(9000 + seed)
    n = int(rng.integers(300, 1700))
    n_clusters = int(rng.choice([max(2, n // 3), max(2, n // 5), max(2, n // 8), 7, n]))      # n: no relatives at all
    knn = int(rng.integers(1, 6))
    ss64 = int(rng.choice([8, 13, 16, 32, 37, 64]))
    band_rows = int(rng.choice([16, 32, 48, 64, 96, 160, 256]))
    ani = bool(rng.integers(0, 2))
    ties = ["canonical", "reference"][int(rng.integers(0, 2))]
    keep = float(rng.choice([0.98, 0.94, 0.85]))
    kmers = [17, 21]
    shuffle = int(rng.integers(1, 1 << 30)) if rng.integers(0, 2) else None      # relatives at random ids: sparse walks
    bins = _clustered(n, len(kmers), ss64, n_clusters, keep=keep, shuffle=shuffle)
    if rng.integers(0, 2):      # a few exact duplicates: ties at key 0
        for _ in range(3):
            a, b = rng.integers(0, n, 2)
            bins[a] = bins[b]
    o, g = oracle.Sketches(bins, n, kmers, ss64), gpu_ctx.sketches(bins, n, kmers, ss64)
    p = g.set_k(21, ani)
    t = skl.TIES_REFERENCE if ties == "reference" else skl.TIES_CANONICAL
    idx, d0, (tiles, pruned) = _run(skl, gpu_ctx, g, p, knn, monkeypatch, band_rows, True, t)
    exp = oracle.self_dists_knn(o, knn, oracle.JACCARD, 1, ani,
                                ties=oracle.TIES_RUST_HEAP if ties == "reference" else oracle.TIES_CANONICAL, threads=8)
    cfg = dict(n=n, n_clusters=n_clusters, knn=knn, ss64=ss64, band_rows=band_rows, ani=ani, ties=ties, keep=keep, tiles=tiles, pruned=pruned,
               shuffle=shuffle, sparse=gpu_ctx.knn_prune_stats(full=True)["tiles_sparse_walk"])
    assert np.array_equal(idx, exp["idx"]), (cfg, np.argwhere(idx != exp["idx"])[:5])
    assert np.array_equal(d0, exp["d0"]), cfg
    g.close()


@pytest.mark.parametrize("ties", ["canonical", "reference"])
@pytest.mark.parametrize("ani", [False, True], ids=["dist", "ani"])
@pytest.mark.parametrize("panel,band_rows", [(256, 64), (384, 96), (128, 33)])
def test_cross_knn_in_column_panels(oracle, skl, gpu_ctx, monkeypatch, ties, ani, panel, band_rows):
    """Row-by-row kNN (cross mode) fed its candidates in ascending column panels, pruned from the second panel on against the
    query rows' bounds: the oracle's lists in both tie rules.  (SKL_KNN_PANEL -- A/B build -- forces panels on a reference
    set small enough for the oracle; the product takes them from 131 072 references on.)"""
    kmers, ss64, nr, nq, knn = [17, 21], 16, 1500, 333, 4
    rb = _clustered(nr, len(kmers), ss64, 250)                     # 6 per cluster, relatives at id distances 250, 500, ...
    import torch
    qb = synth.set_clustered_device(nq, len(kmers), ss64, torch.device("cuda", 0), keep=0.94, n_clusters=250,
                                    first_sample=10 ** 6).cpu().numpy().view(np.uint64)
    qb[7] = rb[31]                                                  # a query that IS a reference: key 0
    o_r, o_q = oracle.Sketches(rb, nr, kmers, ss64), oracle.Sketches(qb, nq, kmers, ss64)
    g_r, g_q = gpu_ctx.sketches(rb, nr, kmers, ss64), gpu_ctx.sketches(qb, nq, kmers, ss64)
    monkeypatch.setenv("SKL_KNN_PANEL", str(panel))
    monkeypatch.setenv("SKL_KNN_BAND_ROWS", str(band_rows))
    monkeypatch.setenv("SKL_TILE32_MIN", "0")
    gpu_ctx.reload_env()
    gpu_ctx.set_knn_ties(skl.TIES_REFERENCE if ties == "reference" else skl.TIES_CANONICAL)
    try:
        idx, d0, _ = skl.cross_dists_knn(gpu_ctx, g_r, g_q, g_r.set_k(21, ani), knn)
    finally:
        gpu_ctx.set_knn_ties(skl.TIES_CANONICAL)
    tiles, pruned = gpu_ctx.knn_prune_stats()
    exp = oracle.cross_dists_knn(o_r, o_q, knn, oracle.JACCARD, 1, ani,
                                 ties=oracle.TIES_RUST_HEAP if ties == "reference" else oracle.TIES_CANONICAL, threads=8)
    assert np.array_equal(idx, exp["idx"]), np.argwhere(idx != exp["idx"])[:5]
    assert np.array_equal(d0, exp["d0"])
    assert tiles > 0 and pruned > 0, (tiles, pruned)
    g_r.close()
    g_q.close()


@pytest.mark.parametrize("ties", ["canonical", "reference"])
def test_self_knn_row_range_in_column_panels(oracle, skl, gpu_ctx, monkeypatch, ties):
    """A row range of the self kNN (what a rank of a row-sharded run computes) through the same panels: a row is not its own
    candidate, and the panels on either side of it arrive in ascending id."""
    kmers, ss64, n, knn = [21], 16, 1400, 3
    bins = _clustered(n, 1, ss64, 200)
    o, g = oracle.Sketches(bins, n, kmers, ss64), gpu_ctx.sketches(bins, n, kmers, ss64)
    monkeypatch.setenv("SKL_KNN_PANEL", "256")
    monkeypatch.setenv("SKL_KNN_BAND_ROWS", "64")
    monkeypatch.setenv("SKL_TILE32_MIN", "0")
    gpu_ctx.reload_env()
    gpu_ctx.set_knn_ties(skl.TIES_REFERENCE if ties == "reference" else skl.TIES_CANONICAL)
    try:
        idx, d0, _ = skl.self_dists_knn(gpu_ctx, g, g.set_k(21), knn, 301, 777)
    finally:
        gpu_ctx.set_knn_ties(skl.TIES_CANONICAL)
    exp = oracle.self_dists_knn(o, knn, oracle.JACCARD, 0, False,
                                ties=oracle.TIES_RUST_HEAP if ties == "reference" else oracle.TIES_CANONICAL, threads=8)
    assert np.array_equal(idx, exp["idx"][301:777]) and np.array_equal(d0, exp["d0"][301:777])
    assert gpu_ctx.knn_prune_stats()[1] > 0
    g.close()


def test_sparse_walks_inside_column_windows_and_partial_states(oracle, skl, gpu_ctx, monkeypatch):
    """The two multi-device forms over a set whose relatives sit at random ids (tiles that survive the probe for one pair):
    the reference's order from heaps travelling through 3 column windows, and canonical partial states of 3 participants
    merged -- both the oracle's lists, the windows with tiles finished by the sparse walk on the way."""
    import torch
    from sketchlib.rust_amd import multi_gpu

    kmers, ss64, n, knn, band_rows, world = [21], 16, 8192, 1, 96, 3
    bins = _clustered(n, 1, ss64, n // 4, shuffle=5)
    o, g = oracle.Sketches(bins, n, kmers, ss64), gpu_ctx.sketches(bins, n, kmers, ss64)
    p = g.set_k(21)
    monkeypatch.setenv("SKL_TILE32_MIN", "0")
    gpu_ctx.reload_env()
    n_bands = (n + band_rows - 1) // band_rows
    # (a) travelling heaps
    heaps = skl.knn_heaps_alloc(n, knn, False, torch.device("cuda", 0))
    cuts = multi_gpu.knn_window_cuts(n, band_rows, world)
    sparse = 0
    for r in range(world):
        for band in range(n_bands):
            if band * band_rows >= cuts[r + 1]:
                break
            skl.self_dists_knn_window(gpu_ctx, g, p, knn, band_rows, band, cuts[r], cuts[r + 1], heaps)
            sparse += gpu_ctx.knn_prune_stats(full=True)["tiles_sparse_walk"]
    idx, d0, _ = skl.knn_heaps_finalize(gpu_ctx, heaps, 0, n, knn)
    gpu_ctx.synchronize()
    exp = oracle.self_dists_knn(o, knn, oracle.JACCARD, 0, False, ties=oracle.TIES_RUST_HEAP, threads=8)
    assert np.array_equal(idx.cpu().numpy().astype(np.uint64), exp["idx"]) and np.array_equal(d0.cpu().numpy(), exp["d0"])
    assert sparse > 0
    # (b) partial states
    deal = multi_gpu.knn_band_deal(n_bands, world)
    states, sparse = [], 0
    for r in range(world):
        states.append(skl.self_dists_knn_partial(gpu_ctx, g, p, knn, band_rows, deal[r]))
        sparse += gpu_ctx.knn_prune_stats(full=True)["tiles_sparse_walk"]
    got = skl.knn_merge_states(gpu_ctx, np.stack([s[0] for s in states]), np.stack([s[1] for s in states]), None)
    exp = oracle.self_dists_knn(o, knn, oracle.JACCARD, 0, False, ties=oracle.TIES_CANONICAL, threads=8)
    assert np.array_equal(got[0], exp["idx"]) and np.array_equal(got[1], exp["d0"])
    # (a participant's own lists see a third of the ids: few of its tiles have all 160 bounds tight -- parity is the point here)
    g.close()
