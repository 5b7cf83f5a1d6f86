"""BASELINE.json configs[2], [3] and [4] at FULL size on one MI355X (they are specified for 8 GPUs;
one holds them: 40 GB, 80 GB and 0.8 GB of output):

  cfg 3  100 000 genomes all-vs-all, sketchsize64 = 64, k = {15..31}: 5.0e9 core/accessory pairs
  cfg 4  1 000 000 refs x 10 000 queries, sketchsize64 = 32, k = {13..29}: 1.0e10 pairs
  cfg 5  self kNN-50 over 1 000 000 x 1 000 000, single-k Jaccard (1.0e12 pair distances defined)

The oracle cannot recompute these, so each run is checked through (a) size-independent properties --
every record written (NaN-prefilled output), values in range, row bands recomputed on their own equal
the slice of the whole, sortedness / uniqueness of neighbour lists -- and (b) >= 5 000 oracle spot
checks per dense configuration on sampled pairs (random, inside clusters, on tile edges), the sampled
sketches gathered from the device slab.  Clustered synthetic sketches (synth.set_clustered_device):
~200 relatives per genome with J falling with k, everything else at (1, 1)."""
import time

import numpy as np
import pytest

from sketchlib.rust_amd import synth

pytestmark = pytest.mark.gpu
K5, K4 = [15, 19, 23, 27, 31], [13, 17, 21, 25, 29]
KEEP = [0.97, 0.955, 0.94, 0.925, 0.91]      # P(bin kept) per k-mer length: J_k falls with k


def cond(i, j, n):
    return n * i - (i * (i + 1)) // 2 + j - 1 - i


@pytest.fixture(scope="module")
def torch_ctx(skl):
    import torch

    if skl.device_count() == 0:
        pytest.fail("no gfx950 device visible: -m gpu tests must run on the GPU box")
    dev = torch.device("cuda", 0)
    ctx = skl.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    ctx.set_knn_ties(skl.TIES_CANONICAL)   # (the library's default is the reference's order: tests that want it set it)
    yield torch, dev, ctx
    ctx.close()
    torch.cuda.empty_cache()


def gather_oracle(oracle, torch, bins, ids, kmers, ss64):
    """oracle.Sketches over the samples `ids` (sorted unique) of a device slab + id -> position map."""
    ids = np.unique(np.asarray(ids, dtype=np.int64))
    sub = bins[torch.from_numpy(ids).to(bins.device)].cpu().numpy().view(np.uint64)
    return oracle.Sketches(sub, len(ids), kmers, ss64), {int(s): p for p, s in enumerate(ids)}


def all_finite_in_unit_range(torch, out, step=1 << 28):
    flat = out.view(-1)
    for a in range(0, flat.numel(), step):
        c = flat[a:a + step]
        if not bool(((c >= 0) & (c <= 1)).all().item()):     # NaN fails both comparisons
            return False
    return True


def test_cfg3_full_size(oracle, skl, torch_ctx):
    torch, dev, ctx = torch_ctx
    n, ss64 = 100_000, 64
    n_clusters = n // 200
    bins = synth.set_clustered_device(n, 5, ss64, dev, cluster_size=200, keep=KEEP)
    g = ctx.sketches(bins, n, K5, ss64)
    p = g.set_k()
    pairs = n * (n - 1) // 2
    out = torch.full((pairs, 2), float("nan"), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    skl.self_dists_all(ctx, g, p, out=out)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    print(f"cfg3 full: {pairs} pairs in {wall:.3f} s = {pairs / wall:.3g} pairs/s [{ctx.last_kernel()}]")
    assert all_finite_in_unit_range(torch, out), "a record was not written or is out of range"
    # row bands computed on their own (the 8-GPU partition) are bit-equal slices of the whole
    from sketchlib.rust_amd import multi_gpu
    for r0, r1, p0, cnt in multi_gpu.self_band_slices(n, 8)[::3]:
        band = torch.empty((cnt, 2), dtype=torch.float32, device=dev)
        skl.self_dists_rows(ctx, g, p, r0, r1, out=band)
        torch.cuda.synchronize()
        assert bool((band == out[p0:p0 + cnt]).all().item()), (r0, r1)
        del band
    # oracle spot checks
    rng = np.random.default_rng(3)
    ii = rng.integers(0, n - 1, 2000)
    jj = ii + 1 + rng.integers(0, n, 2000) % (n - 1 - ii)
    ci = rng.integers(0, n - n_clusters * 150, 3500)        # inside clusters: j = i + m * n_clusters
    cj = ci + n_clusters * rng.integers(1, 150, 3500)
    edges = [(0, 1), (0, n - 1), (n - 2, n - 1), (15, 16), (16, 127), (16, 128), (127, 128), (255, 256), (4095, 4096),
             (65535, 65536), (n - 129, n - 1), (n - 17, n - 16)]
    pi = np.concatenate([ii, ci, [e[0] for e in edges]])
    pj = np.concatenate([jj, cj, [e[1] for e in edges]])
    assert np.all(pi < pj) and np.all(pj < n) and len(pi) >= 5000
    o, pos = gather_oracle(oracle, torch, bins, np.concatenate([pi, pj]), K5, ss64)
    flat = torch.from_numpy(cond(pi.astype(np.int64), pj.astype(np.int64), n)).to(dev)
    got = out[flat].cpu().numpy()
    fitted = 0
    for t in range(len(pi)):
        exp = oracle.core_acc_pair(o, o, pos[int(pi[t])], pos[int(pj[t])])
        assert tuple(got[t]) == exp, (int(pi[t]), int(pj[t]), tuple(got[t]), exp)
        fitted += 0 < exp[0] < 1
    assert fitted > 2500, "the within-cluster spot checks exercise the regression"
    g.close()
    del out, bins
    torch.cuda.empty_cache()


@pytest.fixture(scope="module")
def million(skl, torch_ctx):
    """cfg 4 / cfg 5 reference database: 1 000 000 sketches, sketchsize64 = 32 (17.9 GB)."""
    torch, dev, ctx = torch_ctx
    nr, ss64 = 1_000_000, 32
    bins = synth.set_clustered_device(nr, 5, ss64, dev, cluster_size=200, keep=KEEP)
    g = ctx.sketches(bins, nr, K4, ss64)
    yield nr, ss64, bins, g
    g.close()
    del bins
    torch.cuda.empty_cache()


def test_cfg4_full_size(oracle, skl, torch_ctx, million):
    torch, dev, ctx = torch_ctx
    nr, ss64, rbins, g_r = million
    nq, n_clusters = 10_000, nr // 200
    qbins = synth.set_clustered_device(nq, 5, ss64, dev, keep=KEEP, first_sample=10_000_000, n_clusters=n_clusters)
    qbins[77] = rbins[123_456]            # a query that is a reference
    g_q = ctx.sketches(qbins, nq, K4, ss64)
    p = g_r.set_k()
    out = torch.full((nr, nq, 2), float("nan"), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    skl.cross_dists_all(ctx, g_r, g_q, p, out=out)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    print(f"cfg4 full: {nr * nq} pairs in {wall:.3f} s = {nr * nq / wall:.3g} pairs/s [{ctx.last_kernel()}]")
    assert all_finite_in_unit_range(torch, out), "a record was not written or is out of range"
    assert out[123_456, 77].tolist() == [0.0, 0.0]
    from sketchlib.rust_amd import multi_gpu
    b = multi_gpu.even_row_bounds(nr, 8)
    for w in (0, 3, 7):                   # reference bands of the 8-GPU partition
        band = torch.empty((b[w + 1] - b[w], nq, 2), dtype=torch.float32, device=dev)
        skl.cross_dists_rows(ctx, g_r, g_q, p, b[w], b[w + 1], out=band)
        torch.cuda.synchronize()
        assert bool((band == out[b[w]:b[w + 1]]).all().item()), w
        del band
    rng = np.random.default_rng(4)
    ri = np.concatenate([rng.integers(0, nr, 2000), [0, nr - 1, 15, 16, 124_999, 125_000, nr - 1]])
    qj = np.concatenate([rng.integers(0, nq, 2000), [0, nq - 1, 127, 128, 9_999, 0, 0]])
    cq = rng.integers(0, nq, 3500)                       # same cluster: ref id = query's cluster + m * n_clusters
    cr = (cq % n_clusters) + n_clusters * rng.integers(0, 200, 3500)
    ri, qj = np.concatenate([ri, cr]), np.concatenate([qj, cq])
    assert len(ri) >= 5000
    o_r, rpos = gather_oracle(oracle, torch, rbins, ri, K4, ss64)
    o_q, qpos = gather_oracle(oracle, torch, qbins, qj, K4, ss64)
    got = out[torch.from_numpy(ri).to(dev), torch.from_numpy(qj).to(dev)].cpu().numpy()
    fitted = 0
    for t in range(len(ri)):
        exp = oracle.core_acc_pair(o_r, o_q, rpos[int(ri[t])], qpos[int(qj[t])])
        assert tuple(got[t]) == exp, (int(ri[t]), int(qj[t]), tuple(got[t]), exp)
        fitted += 0 < exp[0] < 1
    assert fitted > 2500
    g_q.close()
    del out, qbins
    torch.cuda.empty_cache()


def test_cfg5_full_size(oracle, skl, torch_ctx, million):
    torch, dev, ctx = torch_ctx
    n, ss64, bins, g = million
    knn, k_idx = 50, 2
    p = g.set_k(K4[k_idx])
    t0 = time.perf_counter()
    idx, d0, _d1 = skl.self_dists_knn(ctx, g, p, knn)
    wall = time.perf_counter() - t0
    print(f"cfg5 full: self kNN-{knn} over {n} x {n} in {wall:.2f} s = {n * (n - 1) / wall:.3g} pair distances/s")
    # properties of every row
    assert idx.shape == (n, knn) and idx.max() < n
    assert np.all(np.diff(d0, axis=1) >= 0), "rows ascending"
    assert np.all((d0 >= 0) & (d0 <= 1))
    assert not np.any(idx == np.arange(n, dtype=np.uint64)[:, None]), "self excluded (mod.rs:150)"
    srt = np.sort(idx, axis=1)
    assert np.all(srt[:, 1:] != srt[:, :-1]), "no neighbour twice"
    same_cluster = (idx % (n // 200)) == (np.arange(n, dtype=np.uint64)[:, None] % (n // 200))
    assert same_cluster.mean() > 0.99, "neighbours are the cluster's members"
    # sampled rows: the whole row of distances from the dense path, top-50 by (key, id) on the host
    rng = np.random.default_rng(5)
    rows = np.concatenate([rng.integers(0, n, 60), [0, 1, n - 1, 2143, 2144]])
    for i in rows:
        i = int(i)
        dense = skl.cross_dists_rows(ctx, g, g, p, i, i + 1)[0, :, 0]
        dense[i] = np.inf
        order = np.lexsort((np.arange(n), dense))[:knn]
        assert np.array_equal(idx[i], order.astype(np.uint64)), i
        assert np.array_equal(d0[i], dense[order]), i
    # ... and those neighbours' distances against the oracle
    for i in rows[:24]:
        i = int(i)
        o_i, _ = gather_oracle(oracle, torch, bins, [i], K4, ss64)
        nb = np.sort(idx[i].astype(np.int64))
        o_nb, pos = gather_oracle(oracle, torch, bins, nb, K4, ss64)
        exp = oracle.cross_dists_all(o_i, o_nb, oracle.JACCARD, k_idx)[0, :, 0]
        got = {int(j): float(d) for j, d in zip(idx[i], d0[i])}
        for j in nb:
            assert np.float32(got[int(j)]) == exp[pos[int(j)]], (i, int(j))


def test_cfg5_full_size_coreacc_default_mode(oracle, skl, torch_ctx, million):
    """cfg 5 as `sketchlib dist db --knn 50` runs it WITHOUT -k: DistType::CoreAcc (mod.rs:25-37), the kNN arm of mod.rs:195-221
    -- all five k-mer lengths + the regression per pair, rows sorted on the core distance -- in the library's default tie
    rule (the reference's BinaryHeap order).  5.0e11 pairs x 5 k-mer lengths, every pair evaluated once."""
    torch, dev, ctx = torch_ctx
    n, ss64, bins, g = million
    knn = 50
    p = g.set_k()
    ctx.set_knn_ties(skl.TIES_REFERENCE)
    try:
        t0 = time.perf_counter()
        idx, d0, d1 = skl.self_dists_knn(ctx, g, p, knn)
        wall = time.perf_counter() - t0
    finally:
        ctx.set_knn_ties(skl.TIES_CANONICAL)
    print(f"cfg5 core/accessory full: self kNN-{knn} over {n} x {n} in {wall:.2f} s = {n * (n - 1) / wall:.3g} pair distances/s")
    assert "COREACC" in ctx.last_kernel() or "early break" in ctx.last_kernel()   # (fused all-k bands, or counted bands + band epilogue)
    # properties of every row
    assert idx.shape == (n, knn) and idx.max() < n
    assert np.all(np.diff(d0, axis=1) >= 0), "rows ascending on the core distance (distance_matrix.rs:245-248)"
    assert np.all((d0 >= 0) & (d0 <= 1)) and np.all((d1 >= 0) & (d1 <= 1))
    assert not np.any(idx == np.arange(n, dtype=np.uint64)[:, None]), "self excluded (mod.rs:203)"
    srt = np.sort(idx, axis=1)
    assert np.all(srt[:, 1:] != srt[:, :-1]), "no neighbour twice"
    same_cluster = (idx % (n // 200)) == (np.arange(n, dtype=np.uint64)[:, None] % (n // 200))
    print(f"cfg5 core/accessory: {same_cluster.mean():.4f} of the listed neighbours are cluster members; "
          f"{(d0 == 0).mean():.4f} of the listed core distances are exactly 0")
    # sampled rows: the whole row of (core, acc) from the dense path, pushed through the oracle's BinaryHeap in ascending id
    rng = np.random.default_rng(55)
    rows = np.concatenate([rng.integers(0, n, 60), [0, 1, n - 1, 2047, 2048]])
    for i in rows:
        i = int(i)
        dense = skl.cross_dists_rows(ctx, g, g, p, i, i + 1)[0]          # [n, 2]
        ids = np.delete(np.arange(n, dtype=np.uint64), i)
        exp = oracle.heap_replay(np.delete(dense[:, 0], i), knn, ids=ids)
        assert np.array_equal(idx[i], exp["idx"]), i
        assert np.array_equal(d0[i], exp["d0"]), i
        assert np.array_equal(d1[i], dense[idx[i].astype(np.int64), 1]), i
    # ... and those neighbours' distances against the oracle's core_acc_dist
    for i in rows[:24]:
        i = int(i)
        o_i, _ = gather_oracle(oracle, torch, bins, [i], K4, ss64)
        nb = np.sort(idx[i].astype(np.int64))
        o_nb, pos = gather_oracle(oracle, torch, bins, nb, K4, ss64)
        exp = oracle.cross_dists_all(o_i, o_nb, oracle.COREACC)[0]         # [50, 2]
        for j, c, a in zip(idx[i], d0[i], d1[i]):
            assert np.float32(c) == exp[pos[int(j)], 0] and np.float32(a) == exp[pos[int(j)], 1], (i, int(j))
