"""The early break of the core/accessory calls (capi.cpp dense_band, kernels.hip coreacc_epilogue_kernel): the reference's loop
over the k-mer lengths leaves at the first one whose Jaccard index is 0 (jaccard.rs:89-91) and a fit over fewer than three
lengths is (1, 1) (jaccard.rs:117), so the pair kernel counts only the first three or four lengths for every pair and the
epilogue completes the pairs that are still in the running.  Same (core, acc), bit for bit, as counting every length --
checked here on sketches built so that pairs leave the loop at EVERY position, against the oracle's core_acc_dist."""
import numpy as np
import pytest

from sketchlib.rust_amd import synth

pytestmark = pytest.mark.gpu


def _graded(n, nk, ss64, seed=1):
    """Sketches without a single chance match (bin b of sample s at k index t holds (17 s + 5 b + t) mod 2^14: 17 is odd, so two
    samples never agree), into which shared bins are planted pair by pair: pair q = (2q, 2q + 1) shares 1..40 bins at each of
    its first q % (nk + 1) k-mer lengths -- and, every third pair, also at the LAST length, behind the gap (the reference never
    looks there).  Samples 3 apart in the second half form chains sharing bins at every length (fits over all nk points)."""
    nb = ss64 * 64
    s = np.arange(n, dtype=np.int64)[:, None, None]
    t = np.arange(nk, dtype=np.int64)[None, :, None]
    b = np.arange(nb, dtype=np.int64)[None, None, :]
    vals = ((17 * s + 5 * b + t) & 0x3FFF).astype(np.uint16)
    rng = np.random.default_rng(seed)
    lead = {}
    for q in range(n // 4):
        a, c = 2 * q, 2 * q + 1
        m = q % (nk + 1)
        lead[(a, c)] = m
        for tt in list(range(m)) + ([nk - 1] if q % 3 == 0 else []):
            bins = rng.choice(nb, size=int(rng.integers(1, 41)), replace=False)
            vals[c, tt, bins] = vals[a, tt, bins]
    for a in range(n // 2, n - 3):       # chains: a and a + 3 share a fifth of their bins at every length, fewer at longer k
        for tt in range(nk):
            bins = rng.choice(nb, size=max(1, nb // (5 + 2 * tt)), replace=False)
            vals[a + 3, tt, bins] = vals[a, tt, bins]
    return np.ascontiguousarray(synth.bitslice(vals).reshape(n, -1)), lead


CASES = [(520, [15, 19, 23, 27, 31], 64), (520, [13, 17, 21, 25], 8), (300, [11, 15, 19, 23, 27, 31], 70), (700, [15, 19, 23, 27, 31], 16),
         (600, [15, 23, 31], 32), (400, [11, 13, 15, 17, 19, 21, 23], 8)]


@pytest.mark.ab_library
@pytest.mark.parametrize("lengths", [2, 3, 4, 1, 0], ids=["two", "three", "four", "sampled", "off"])
@pytest.mark.parametrize("n,kmers,ss64", CASES)
def test_every_break_position_self_and_cross(oracle, skl, gpu_ctx, monkeypatch, n, kmers, ss64, lengths):
    """SKL_EARLY_BREAK (A/B build): 3 / 4 lengths counted whatever the sample says, the sampled choice, off -- the whole self
    matrix (device and banded host output) and a cross matrix, bit for bit the oracle's."""
    bins, lead = _graded(n, len(kmers), ss64)
    o = oracle.Sketches(bins, n, kmers, ss64)
    exp = oracle.self_dists_all(o, oracle.COREACC, threads=8).reshape(-1, 2)
    # the construction does what it says: pairs leave the loop at every position, and fits over 3, 4, ... nk lengths exist
    fitted = int(((exp[:, 0] != 1.0) | (exp[:, 1] != 1.0)).sum())
    assert fitted >= n // 2 and len({m for m in lead.values()}) == len(kmers) + 1
    monkeypatch.setenv("SKL_EARLY_BREAK", str(lengths))
    gpu_ctx.reload_env()
    g = gpu_ctx.sketches(bins, n, kmers, ss64)
    before = gpu_ctx.early_break_stats()
    got = skl.self_dists_all(gpu_ctx, g, g.set_k())
    after = gpu_ctx.early_break_stats()
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), np.argwhere(got != exp)[:5]
    if lengths >= 2 and lengths < len(kmers):
        assert "early break: %d of %d" % (lengths, len(kmers)) in gpu_ctx.last_kernel()
        assert after[0] - before[0] == n * (n - 1) // 2 and after[1] > before[1]      # pairs were completed one by one
    if lengths == 0:
        assert after == before
    # rows 37 .. n - 50 through the row-range call, and a cross matrix (references = the first 300 samples)
    part = skl.self_dists_rows(gpu_ctx, g, g.set_k(), 37, n - 50)
    lo = 37 * n - 37 * 38 // 2
    assert np.array_equal(part.view(np.uint32), exp[lo:lo + part.shape[0]].view(np.uint32))
    nr = 300 if n > 300 else 150
    g_r, g_q = gpu_ctx.sketches(bins[:nr], nr, kmers, ss64), gpu_ctx.sketches(bins[nr:], n - nr, kmers, ss64)
    o_r, o_q = oracle.Sketches(bins[:nr], nr, kmers, ss64), oracle.Sketches(bins[nr:], n - nr, kmers, ss64)
    cross = skl.cross_dists_all(gpu_ctx, g_r, g_q, g_r.set_k())
    assert np.array_equal(cross.view(np.uint32), oracle.cross_dists_all(o_r, o_q, oracle.COREACC, threads=8).view(np.uint32))
    for x in (g, g_r, g_q):
        x.close()


def test_the_product_library_samples_and_decides(oracle, skl, gpu_ctx):
    """Nothing forced (the product library): sketches with hardly a shared bin take the early break, a set of close relatives
    -- every pair in the running to the last length -- does not; both give the oracle's matrix."""
    kmers, ss64, n = [15, 19, 23, 27, 31], 32, 600
    bins, _ = _graded(n, len(kmers), ss64, seed=3)
    o, g = oracle.Sketches(bins, n, kmers, ss64), gpu_ctx.sketches(bins, n, kmers, ss64)
    before = gpu_ctx.early_break_stats()
    got = skl.self_dists_all(gpu_ctx, g, g.set_k())
    mid = gpu_ctx.early_break_stats()
    assert np.array_equal(got.view(np.uint32), oracle.self_dists_all(o, oracle.COREACC, threads=8).reshape(-1, 2).view(np.uint32))
    assert "early break: " in gpu_ctx.last_kernel() and mid[0] - before[0] == n * (n - 1) // 2
    g.close()
    rel = synth.set_r(n, kmers, ss64, n_clusters=1)
    o, g = oracle.Sketches(rel, n, kmers, ss64), gpu_ctx.sketches(rel, n, kmers, ss64)
    got = skl.self_dists_all(gpu_ctx, g, g.set_k())
    assert np.array_equal(got.view(np.uint32), oracle.self_dists_all(o, oracle.COREACC, threads=8).reshape(-1, 2).view(np.uint32))
    assert "early break" not in gpu_ctx.last_kernel() and gpu_ctx.early_break_stats() == mid
    g.close()


@pytest.mark.parametrize("ties", ["reference", "canonical"])
@pytest.mark.parametrize("band_rows,knn", [(64, 5), (96, 12), (16, 40), (200, 3)])
def test_symmetric_core_accessory_knn_takes_the_early_break(oracle, skl, gpu_ctx, set_switch, ties, band_rows, knn):
    """The one-evaluation core/accessory self kNN, several row bands: from the second band on the band is counted at its first
    lengths and the band's epilogue writes records, marks and the turned copy (pre-filled with (1, 1)).  Most lists of this
    set END in (1, 1) entries -- the first that arrived, in the reference's order -- and with bands lower than knn (16 rows,
    knn = 40) lists are still filling when the early break starts, so (1, 1) records themselves are candidates.  Ids, order
    and both distances = the oracle's."""
    kmers, ss64, n = [15, 19, 23, 27, 31], 16, 700
    bins, _ = _graded(n, len(kmers), ss64, seed=5)
    o, g = oracle.Sketches(bins, n, kmers, ss64), gpu_ctx.sketches(bins, n, kmers, ss64)
    set_switch("SKL_KNN_BAND_ROWS", band_rows)
    gpu_ctx.set_knn_ties(skl.TIES_REFERENCE if ties == "reference" else skl.TIES_CANONICAL)
    before = gpu_ctx.early_break_stats()
    try:
        idx, d0, d1 = skl.self_dists_knn(gpu_ctx, g, g.set_k(), knn)
    finally:
        gpu_ctx.set_knn_ties(skl.TIES_CANONICAL)
    after = gpu_ctx.early_break_stats()
    exp = oracle.self_dists_knn(o, knn, oracle.COREACC, 0, False,
                                ties=oracle.TIES_RUST_HEAP if ties == "reference" else oracle.TIES_CANONICAL, threads=8)
    assert np.array_equal(idx, exp["idx"]), np.argwhere(idx != exp["idx"])[:5]
    assert np.array_equal(d0.view(np.uint32), exp["d0"].view(np.uint32)) and np.array_equal(d1.view(np.uint32), exp["d1"].view(np.uint32))
    assert "early break: " in gpu_ctx.last_kernel() and after[0] > before[0] and after[1] > before[1]
    g.close()


@pytest.mark.parametrize("world,band_rows,knn", [(3, 64, 5), (2, 96, 12), (4, 48, 7)])
def test_column_windows_take_the_early_break_from_their_second_band(oracle, skl, gpu_ctx, world, band_rows, knn):
    """skl_self_dists_knn_window (the reference's order over several devices: heaps that travel through column windows), core/
    accessory keys: from band 1 on each call counts its band at the first lengths and finishes it with the band epilogue.
    The participants run one after the other on one device and one set of heap arrays (exact: tests/test_gpu_knn_ties.py)."""
    import torch
    from sketchlib.rust_amd import multi_gpu

    kmers, ss64, n = [15, 19, 23, 27, 31], 16, 700
    bins, _ = _graded(n, len(kmers), ss64, seed=9)
    o, g = oracle.Sketches(bins, n, kmers, ss64), gpu_ctx.sketches(bins, n, kmers, ss64)
    p = g.set_k()
    heaps = skl.knn_heaps_alloc(n, knn, True, torch.device("cuda", 0))
    cuts = multi_gpu.knn_window_cuts(n, band_rows, world)
    before = gpu_ctx.early_break_stats()
    names = set()
    for r in range(world):
        for band in range((n + band_rows - 1) // band_rows):
            if band * band_rows >= cuts[r + 1]:
                break
            skl.self_dists_knn_window(gpu_ctx, g, p, knn, band_rows, band, cuts[r], cuts[r + 1], heaps)
            names.add("early break" in gpu_ctx.last_kernel())
    idx, d0, d1 = skl.knn_heaps_finalize(gpu_ctx, heaps, 0, n, knn)
    gpu_ctx.synchronize()
    after = gpu_ctx.early_break_stats()
    exp = oracle.self_dists_knn(o, knn, oracle.COREACC, 0, False, ties=oracle.TIES_RUST_HEAP, threads=8)
    assert np.array_equal(idx.cpu().numpy().astype(np.uint64), exp["idx"]), np.argwhere(idx.cpu().numpy() != exp["idx"])[:5]
    assert np.array_equal(d0.cpu().numpy().view(np.uint32), exp["d0"].view(np.uint32))
    assert np.array_equal(d1.cpu().numpy().view(np.uint32), exp["d1"].view(np.uint32))
    assert names == {True, False} and after[0] > before[0]      # band 0 of a window fused, the later ones counted + epilogue
    g.close()
