"""Round 6 of the early break (capi.cpp early_break_plan, epilogue.hip): the regimes round 5 left untested or declined.

* sketchsize64 256 ... 1023: expected_samebits = maxnbits >> 14 >= 1 (jaccard.rs:26-31), so a pair leaves the reference's
  loop when it shares NO MORE bins than chance -- 1, 2 or 3 of them -- not when it shares none.  Random sketches (Poisson(1-4)
  chance matches per length) plus planted relatives, on every path: self, row ranges, cross, the symmetric core/accessory
  kNN's bands, column windows; forced 2 / 3 / 4 lengths (A/B build) and the sampled choice (product library).
* with a completeness correction (jaccard.rs:36-41: J is scaled per pair, 0 stays 0): <= 1e-6 on every pair.
* sketches beyond 65 535 bins (the segmented counts form).
* the decision block by block: a database that is half one species takes the early break between the species and counts
  every length within.
Everything against the oracle's core_acc_dist."""
import numpy as np
import pytest

from sketchlib.rust_amd import synth

pytestmark = pytest.mark.gpu

KMERS = [15, 19, 23, 27, 31]


def _mixed(n, kmers, ss64, n_random, n_clusters=3, seed=11):
    """The first n_random samples are random sketches (Set U: chance matches only), the others come in n_clusters clusters of
    relatives (Set R), interleaved by cluster."""
    u = synth.set_u(n_random, len(kmers), ss64, seed=synth.SEED_U + seed)
    r = synth.set_r(n - n_random, kmers, ss64, n_clusters=n_clusters, seed=synth.SEED_R + seed)
    return np.ascontiguousarray(np.concatenate([u, r], axis=0))


def _chance_regime_is_exercised(oracle, o, exp, ss64):
    """The data does what the test is about: some pair's FIRST count lies in [1, expected] -- it leaves the loop at once with a
    non-zero count, (1, 1) -- some pair is above the expectation at its first lengths and falls to it later, some fit exists."""
    expected = (ss64 * 64) >> 14
    assert expected >= 1
    same = oracle.self_binmatch(o, threads=8)
    first = same[:, 0]
    leaves_nonzero = (first >= 1) & (first <= expected)
    assert leaves_nonzero.any() and np.all(exp[leaves_nonzero] == 1.0)
    above = same > expected
    later_fall = above[:, 0] & above[:, 1] & above[:, 2] & ~above[:, 3:].all(axis=1)
    assert later_fall.any()
    assert ((exp[:, 0] != 1.0) | (exp[:, 1] != 1.0)).any()


CHANCE_CASES = [(320, 256), (320, 300), (320, 512), (300, 1023)]


@pytest.mark.parametrize("n,ss64", CHANCE_CASES)
def test_expected_samebits_nonzero_product_library(oracle, skl, gpu_ctx, n, ss64):
    """The product library, nothing forced: the sample runs the reference's test (ln J < tolerance, not `no shared bin`), so
    random sketches at 256+ chunks TAKE the early break; self matrix, a row range and a cross matrix, bit for bit."""
    bins = _mixed(n, KMERS, ss64, n_random=n - 60)
    o = oracle.Sketches(bins, n, KMERS, ss64)
    exp = oracle.self_dists_all(o, oracle.COREACC, threads=8).reshape(-1, 2)
    _chance_regime_is_exercised(oracle, o, exp, ss64)
    g = gpu_ctx.sketches(bins, n, KMERS, ss64)
    before = gpu_ctx.early_break_stats()
    got = skl.self_dists_all(gpu_ctx, g, g.set_k())
    after = gpu_ctx.early_break_stats()
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), np.argwhere(got != exp)[:5]
    # (at 1 023 chunks a random pair shares Poisson(4) bins per length against an expectation of 3: 57 % pass each test, a
    # tenth of the pairs would still be in the running after four lengths -- the sample says so and every length is counted)
    took = ss64 <= 512
    assert ("early break: " in gpu_ctx.last_kernel()) == took, gpu_ctx.last_kernel()
    if took:
        assert after[0] - before[0] == n * (n - 1) // 2 and after[1] > before[1]
    part = skl.self_dists_rows(gpu_ctx, g, g.set_k(), 29, n - 41)
    lo = 29 * n - 29 * 30 // 2
    assert np.array_equal(part.view(np.uint32), exp[lo:lo + part.shape[0]].view(np.uint32))
    # the random samples alone (what Set U is): the early break is taken there as well
    nr = n - 60
    g_u = gpu_ctx.sketches(bins[:nr], nr, KMERS, ss64)
    got_u = skl.self_dists_all(gpu_ctx, g_u, g_u.set_k())
    assert ("early break: " in gpu_ctx.last_kernel()) == took, gpu_ctx.last_kernel()
    keep = np.array([i * n - i * (i + 1) // 2 + j - 1 - i for i in range(nr) for j in range(i + 1, nr)])
    assert np.array_equal(got_u.view(np.uint32), exp[keep].view(np.uint32))
    g_q = gpu_ctx.sketches(bins[nr - 100:], n - nr + 100, KMERS, ss64)
    o_u, o_q = oracle.Sketches(bins[:nr], nr, KMERS, ss64), oracle.Sketches(bins[nr - 100:], n - nr + 100, KMERS, ss64)
    cross = skl.cross_dists_all(gpu_ctx, g_u, g_q, g_u.set_k())
    assert np.array_equal(cross.view(np.uint32), oracle.cross_dists_all(o_u, o_q, oracle.COREACC, threads=8).view(np.uint32))
    for x in (g, g_u, g_q):
        x.close()


@pytest.mark.ab_library
@pytest.mark.parametrize("lengths", [2, 3, 4])
@pytest.mark.parametrize("n,ss64", CHANCE_CASES)
def test_expected_samebits_nonzero_forced_lengths(oracle, skl, gpu_ctx, monkeypatch, n, ss64, lengths):
    """SKL_EARLY_BREAK = 2 / 3 / 4 (A/B build): that many lengths counted whatever the sample says."""
    bins = _mixed(n, KMERS, ss64, n_random=n - 60)
    o = oracle.Sketches(bins, n, KMERS, ss64)
    exp = oracle.self_dists_all(o, oracle.COREACC, threads=8).reshape(-1, 2)
    monkeypatch.setenv("SKL_EARLY_BREAK", str(lengths))
    gpu_ctx.reload_env()
    g = gpu_ctx.sketches(bins, n, KMERS, ss64)
    got = skl.self_dists_all(gpu_ctx, g, g.set_k())
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), np.argwhere(got != exp)[:5]
    assert "early break: %d of 5" % lengths in gpu_ctx.last_kernel()
    part = skl.self_dists_rows(gpu_ctx, g, g.set_k(), 100, 250)
    lo = 100 * n - 100 * 101 // 2
    assert np.array_equal(part.view(np.uint32), exp[lo:lo + part.shape[0]].view(np.uint32))
    # (a cross matrix of at least 65 536 pairs -- the early break's floor: queries = the samples from 70 on)
    g_q, o_q = gpu_ctx.sketches(bins[70:], n - 70, KMERS, ss64), oracle.Sketches(bins[70:], n - 70, KMERS, ss64)
    cross = skl.cross_dists_all(gpu_ctx, g, g_q, g.set_k())
    assert "early break: %d of 5" % lengths in gpu_ctx.last_kernel()
    assert np.array_equal(cross.view(np.uint32), oracle.cross_dists_all(o, o_q, oracle.COREACC, threads=8).view(np.uint32))
    for x in (g, g_q):
        x.close()


@pytest.mark.parametrize("ties", ["reference", "canonical"])
@pytest.mark.parametrize("ss64,band_rows,knn", [(256, 64, 7), (512, 96, 12), (300, 48, 40)])
def test_symmetric_knn_bands_where_chance_matches_are_expected(oracle, skl, gpu_ctx, set_switch, ties, ss64, band_rows, knn):
    """The one-evaluation core/accessory self kNN over several row bands (the bands' own early-break epilogue) at
    expected_samebits >= 1: ids, order and both distances = the oracle's."""
    n = 330
    bins = _mixed(n, KMERS, ss64, n_random=n - 30, seed=17)
    o, g = oracle.Sketches(bins, n, KMERS, ss64), gpu_ctx.sketches(bins, n, KMERS, ss64)
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
    # (at 512 chunks -- Poisson(2) chance matches against an expectation of 2 -- whether three lengths pay is a close call the
    # sample may decide either way; at 256 and 300 it is not)
    if ss64 <= 300:
        assert "early break: " in gpu_ctx.last_kernel() and after[0] > before[0] and after[1] > before[1]
    g.close()


@pytest.mark.parametrize("world,ss64,band_rows,knn", [(3, 256, 64, 5), (2, 300, 96, 12)])
def test_column_windows_where_chance_matches_are_expected(oracle, skl, gpu_ctx, world, ss64, band_rows, knn):
    """skl_self_dists_knn_window (heaps that travel through column windows), core/accessory keys, expected_samebits >= 1."""
    import torch
    from sketchlib.rust_amd import multi_gpu

    n = 330
    bins = _mixed(n, KMERS, ss64, n_random=n - 30, seed=19)
    o, g = oracle.Sketches(bins, n, KMERS, ss64), gpu_ctx.sketches(bins, n, KMERS, ss64)
    p = g.set_k()
    heaps = skl.knn_heaps_alloc(n, knn, True, torch.device("cuda", 0))
    cuts = multi_gpu.knn_window_cuts(n, band_rows, world)
    names = set()
    for r in range(world):
        for band in range((n + band_rows - 1) // band_rows):
            if band * band_rows >= cuts[r + 1]:
                break
            skl.self_dists_knn_window(gpu_ctx, g, p, knn, band_rows, band, cuts[r], cuts[r + 1], heaps)
            names.add("early break" in gpu_ctx.last_kernel())
    idx, d0, d1 = skl.knn_heaps_finalize(gpu_ctx, heaps, 0, n, knn)
    gpu_ctx.synchronize()
    exp = oracle.self_dists_knn(o, knn, oracle.COREACC, 0, False, ties=oracle.TIES_RUST_HEAP, threads=8)
    assert np.array_equal(idx.cpu().numpy().astype(np.uint64), exp["idx"]), np.argwhere(idx.cpu().numpy() != exp["idx"])[:5]
    assert np.array_equal(d0.cpu().numpy().view(np.uint32), exp["d0"].view(np.uint32))
    assert np.array_equal(d1.cpu().numpy().view(np.uint32), exp["d1"].view(np.uint32))
    assert True in names
    g.close()


@pytest.mark.parametrize("n,ss64,cutoff", [(600, 64, 0.64), (400, 32, 0.3), (320, 300, 0.64), (500, 16, 0.9)])
def test_early_break_with_a_completeness_correction(oracle, skl, gpu_ctx, n, ss64, cutoff):
    """jaccard.rs:36-41 scales J by the pair's completeness factor when c1 c2 >= cutoff; 0 stays 0, so the same break fires.
    The dense calls take the early break with a completeness vector too: every pair within 1e-6 of the oracle (the regression
    of a corrected pair goes through the restated libm logarithm), self and cross."""
    bins = _mixed(n, KMERS, ss64, n_random=n - 120, n_clusters=4, seed=23)
    rng = np.random.default_rng(5)
    comp = rng.uniform(0.45, 1.0, n)
    comp[::7] = 1.0
    o = oracle.Sketches(bins, n, KMERS, ss64, completeness=comp)
    exp = oracle.self_dists_all(o, oracle.COREACC, cutoff=cutoff, threads=8).reshape(-1, 2)
    assert ((exp[:, 0] != 1.0) | (exp[:, 1] != 1.0)).sum() >= 100
    g = gpu_ctx.sketches(bins, n, KMERS, ss64, completeness=comp)
    p = g.set_k(cutoff=cutoff)
    before = gpu_ctx.early_break_stats()
    got = skl.self_dists_all(gpu_ctx, g, p)
    after = gpu_ctx.early_break_stats()
    name = gpu_ctx.last_kernel()
    assert np.max(np.abs(got.astype(np.float64) - exp.astype(np.float64))) <= 1e-6
    assert np.array_equal(np.isnan(got), np.isnan(exp))
    nr = n // 2
    g_r = gpu_ctx.sketches(bins[:nr], nr, KMERS, ss64, completeness=comp[:nr])
    g_q = gpu_ctx.sketches(bins[nr:], n - nr, KMERS, ss64, completeness=comp[nr:])
    o_r = oracle.Sketches(bins[:nr], nr, KMERS, ss64, completeness=comp[:nr])
    o_q = oracle.Sketches(bins[nr:], n - nr, KMERS, ss64, completeness=comp[nr:])
    cross = skl.cross_dists_all(gpu_ctx, g_r, g_q, g_r.set_k(cutoff=cutoff))
    exp_c = oracle.cross_dists_all(o_r, o_q, oracle.COREACC, cutoff=cutoff, threads=8)
    assert np.max(np.abs(cross.astype(np.float64) - exp_c.astype(np.float64))) <= 1e-6
    # the same slab without its vector afterwards: another decision, bit-identical again
    g.set_completeness(None)
    got = skl.self_dists_all(gpu_ctx, g, g.set_k())
    exp = oracle.self_dists_all(oracle.Sketches(bins, n, KMERS, ss64), oracle.COREACC, threads=8).reshape(-1, 2)
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))
    for x in (g, g_r, g_q):
        x.close()
    # (what the default dispatch does, asserted LAST: under scripts/forced_switch_suites.sh every parity assertion above must
    # hold whatever the switches make of the dispatch)
    assert "early break: " in name, name
    assert after[0] - before[0] == n * (n - 1) // 2 and after[1] > before[1]


@pytest.mark.ab_library
@pytest.mark.parametrize("lengths", [1, 3])
def test_early_break_beyond_65535_bins(oracle, skl, gpu_ctx, monkeypatch, lengths):
    """`sketch -s 100000` (sketchsize64 = 1 563: the segmented counts form, expected_samebits = 6).  Forced (3 lengths): the
    first lengths are counted in segments, the pairs still in the running completed by the same epilogue, bit for bit.  Sampled
    (1): between random sketches 44 % of the pairs pass each length's test (Poisson(6.1) > 6), a completion there is a run of
    thousands of dependent trips -- the cost model says no, and every length is counted (profiles/r06_early_break_forced_lengths.md)."""
    n, ss64 = 300, 1563
    bins = _mixed(n, KMERS, ss64, n_random=n - 40, n_clusters=2, seed=29)
    o = oracle.Sketches(bins, n, KMERS, ss64)
    exp = oracle.self_dists_all(o, oracle.COREACC, threads=8).reshape(-1, 2)
    assert ((exp[:, 0] != 1.0) | (exp[:, 1] != 1.0)).sum() >= 100
    monkeypatch.setenv("SKL_EARLY_BREAK", str(lengths))
    gpu_ctx.reload_env()
    g = gpu_ctx.sketches(bins, n, KMERS, ss64)
    before = gpu_ctx.early_break_stats()
    got = skl.self_dists_all(gpu_ctx, g, g.set_k())
    after = gpu_ctx.early_break_stats()
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), np.argwhere(got != exp)[:5]
    if lengths == 3:
        assert "early break: 3 of 5" in gpu_ctx.last_kernel() and after[1] > before[1]
    else:
        assert "early break" not in gpu_ctx.last_kernel()
    g.close()


def _species_db(sizes, kmers, ss64, seed=31):
    """Consecutive groups of samples: size > 0 = one species of that many close relatives (its own parent sketch), size < 0 =
    that many unrelated random sketches."""
    parts = []
    for x, size in enumerate(sizes):
        if size > 0:
            parts.append(synth.set_r(size, kmers, ss64, n_clusters=1, seed=synth.SEED_R + seed + 101 * x))
        else:
            parts.append(synth.set_u(-size, len(kmers), ss64, seed=synth.SEED_U + seed + 101 * x))
    return np.ascontiguousarray(np.concatenate(parts, axis=0))


@pytest.mark.parametrize("sizes,ss64", [((512, -512), 16), ((-300, 468, -256), 8), ((256,) * 8, 8), ((-256, 256, 256, -512, 256), 32)])
def test_the_early_break_is_decided_block_by_block(oracle, skl, gpu_ctx, sizes, ss64):
    """A database that is half one species (and one sorted by species): the blocks of the pair space within a species count every
    length, the blocks between species -- and between unrelated genomes -- take the early break; bit for bit the oracle's
    matrix, self and cross, and the plan says which block did what."""
    bins = _species_db(sizes, KMERS, ss64)
    n = bins.shape[0]
    o = oracle.Sketches(bins, n, KMERS, ss64)
    exp = oracle.self_dists_all(o, oracle.COREACC, threads=8).reshape(-1, 2)
    g = gpu_ctx.sketches(bins, n, KMERS, ss64)
    before = gpu_ctx.early_break_stats()
    got = skl.self_dists_all(gpu_ctx, g, g.set_k())
    after = gpu_ctx.early_break_stats()
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), np.argwhere(got != exp)[:5]
    plan = gpu_ctx.early_break_blocks()
    assert plan["mixed"] and "block by block" in gpu_ctx.last_kernel(), (plan, gpu_ctx.last_kernel())
    assert plan["shifts"] == (8, 8)
    table = plan["block_lengths"]
    # which species does a block of 256 sample ids lie in?  (0: none / several)
    owner = np.zeros(n, dtype=np.int64)
    at = 0
    for x, size in enumerate(sizes):
        if size > 0:
            owner[at:at + size] = x + 1
        at += abs(size)
    nb = (n + 255) // 256
    for r in range(nb):
        for c in range(r, nb):
            rows, cols = owner[r * 256:(r + 1) * 256], owner[c * 256:(c + 1) * 256]
            if rows.min() == rows.max() and cols.min() == cols.max():      # (blocks that straddle a boundary may go either way)
                same_species = rows[0] != 0 and rows[0] == cols[0]
                assert (table[r, c] == len(KMERS)) == same_species, (r, c, table)
    assert after[0] - before[0] == n * (n - 1) // 2 and after[1] > before[1]
    # rows through the banded call, and a cross matrix of the same samples (queries = the database's second half)
    part = skl.self_dists_rows(gpu_ctx, g, g.set_k(), 100, n - 77)
    lo = 100 * n - 100 * 101 // 2
    assert np.array_equal(part.view(np.uint32), exp[lo:lo + part.shape[0]].view(np.uint32))
    h = n // 2
    g_q, o_q = gpu_ctx.sketches(bins[h:], n - h, KMERS, ss64), oracle.Sketches(bins[h:], n - h, KMERS, ss64)
    cross = skl.cross_dists_all(gpu_ctx, g, g_q, g.set_k())
    assert np.array_equal(cross.view(np.uint32), oracle.cross_dists_all(o, o_q, oracle.COREACC, threads=8).view(np.uint32))
    g.close()
    g_q.close()


@pytest.mark.ab_library
@pytest.mark.parametrize("n,ss64,comp", [(700, 64, False), (600, 32, True), (500, 157, False), (420, 300, False)])
def test_band_pipeline_forced_on_small_inputs(oracle, skl, gpu_ctx, monkeypatch, n, ss64, comp):
    """The form large calls take, forced onto inputs the oracle can check whole (A/B build, SKL_EB_PIPELINE_MIN): u16 counts, the
    call cut into row bands, each band's epilogue on the second stream beside the next band's counts kernel."""
    bins = _mixed(n, KMERS, ss64, n_random=n - 200, n_clusters=2, seed=37)
    cvec = np.random.default_rng(3).uniform(0.5, 1.0, n) if comp else None
    o = oracle.Sketches(bins, n, KMERS, ss64, completeness=cvec)
    exp = oracle.self_dists_all(o, oracle.COREACC, threads=8).reshape(-1, 2)
    monkeypatch.setenv("SKL_EARLY_BREAK", "3")
    monkeypatch.setenv("SKL_TAIL_SLICES", "0")       # (chunk slices keep u32 counts)
    monkeypatch.setenv("SKL_EB_PIPELINE", "1")
    monkeypatch.setenv("SKL_EB_PIPELINE_MIN", "30000")
    gpu_ctx.reload_env()
    g = gpu_ctx.sketches(bins, n, KMERS, ss64, completeness=cvec)

    def check(got, want):
        if comp:
            assert np.max(np.abs(got.astype(np.float64) - want.astype(np.float64))) <= 1e-6
        else:
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), np.argwhere(got != want)[:5]

    check(skl.self_dists_all(gpu_ctx, g, g.set_k()), exp)
    assert "row bands, each band's epilogue beside the next band's counts kernel" in gpu_ctx.last_kernel(), gpu_ctx.last_kernel()
    g_q = gpu_ctx.sketches(bins[150:], n - 150, KMERS, ss64, completeness=None if cvec is None else cvec[150:])
    o_q = oracle.Sketches(bins[150:], n - 150, KMERS, ss64, completeness=None if cvec is None else cvec[150:])
    check(skl.cross_dists_all(gpu_ctx, g, g_q, g.set_k()), oracle.cross_dists_all(o, o_q, oracle.COREACC, threads=8))
    assert "row bands" in gpu_ctx.last_kernel()
    g.close()
    g_q.close()


@pytest.mark.ab_library
@pytest.mark.parametrize("lengths", [2, 3])
@pytest.mark.parametrize("n,ss64,comp", [(700, 64, False), (601, 32, True), (530, 16, False), (420, 300, False)])
def test_blocked_epilogue_order_forced_on_small_inputs(oracle, skl, gpu_ctx, monkeypatch, n, ss64, comp, lengths):
    """SKL_EB_BLOCKED=1 (A/B build): the early break's epilogue walking the pair space in blocks of 256 x 256 pairs, each block on
    one XCD -- what the library does by itself when the column samples' slices of one k-mer length outgrow the Infinity Cache
    (cfg 3) -- on inputs the oracle checks whole: self matrix (ragged last blocks, the diagonal), a row range, a cross matrix."""
    bins = _mixed(n, KMERS, ss64, n_random=n - 200, n_clusters=2, seed=41)
    cvec = np.random.default_rng(3).uniform(0.5, 1.0, n) if comp else None
    o = oracle.Sketches(bins, n, KMERS, ss64, completeness=cvec)
    exp = oracle.self_dists_all(o, oracle.COREACC, threads=8).reshape(-1, 2)
    monkeypatch.setenv("SKL_EARLY_BREAK", str(lengths))
    monkeypatch.setenv("SKL_EB_BLOCKED", "1")
    monkeypatch.setenv("SKL_EB_BLK_ROW_SHIFT", "7" if n % 2 else "10")     # (several row blocks at these sizes / one)
    gpu_ctx.reload_env()
    g = gpu_ctx.sketches(bins, n, KMERS, ss64, completeness=cvec)

    def check(got, want):
        if comp:
            assert np.max(np.abs(got.astype(np.float64) - want.astype(np.float64))) <= 1e-6
        else:
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), np.argwhere(got != want)[:5]

    check(skl.self_dists_all(gpu_ctx, g, g.set_k()), exp)
    assert "pairs per XCD" in gpu_ctx.last_kernel(), gpu_ctx.last_kernel()
    part = skl.self_dists_rows(gpu_ctx, g, g.set_k(), 131, n - 57)
    lo = 131 * n - 131 * 132 // 2
    check(part, exp[lo:lo + part.shape[0]])
    g_q = gpu_ctx.sketches(bins[150:], n - 150, KMERS, ss64, completeness=None if cvec is None else cvec[150:])
    o_q = oracle.Sketches(bins[150:], n - 150, KMERS, ss64, completeness=None if cvec is None else cvec[150:])
    check(skl.cross_dists_all(gpu_ctx, g, g_q, g.set_k()), oracle.cross_dists_all(o, o_q, oracle.COREACC, threads=8))
    g.close()
    g_q.close()


@pytest.mark.ab_library
@pytest.mark.parametrize("lean", [1, 0])
@pytest.mark.parametrize("lengths", [2, 3, 4])
@pytest.mark.parametrize("n,ss64,tail", [(700, 64, True), (700, 64, False), (2100, 32, False), (2100, 16, False), (530, 16, True), (450, 73, False), (420, 300, False)])
def test_lean_epilogue_against_the_general_one(oracle, skl, gpu_ctx, monkeypatch, n, ss64, tail, lengths, lean):
    """The lean early-break epilogue (coreacc_epilogue_lean_kernel<SLICED, NK>: pairs that end as (1, 1) decided on their counts,
    f64 work only for fits of three or more points, the first length not counted completed with requests a trip ahead from the LDS
    rows) and the general kernel (SKL_EB_LEAN=0, A/B build) on the same inputs, both bit for bit against the oracle: 2 / 3 / 4
    lengths counted; u32 counts (small launches, in one plane or in the planes of a tail-sliced launch) and u16 counts (from
    4 Mi pair x length evaluations on: the cases of 2 100 genomes); sketch sizes of 1, 2 and 3 trips with the rows in LDS and one
    beyond them (300 chunks); self matrix, a row range, a cross matrix."""
    bins = _mixed(n, KMERS, ss64, n_random=n - 200, n_clusters=2, seed=43)
    o = oracle.Sketches(bins, n, KMERS, ss64)
    exp = oracle.self_dists_all(o, oracle.COREACC, threads=8).reshape(-1, 2)
    monkeypatch.setenv("SKL_EARLY_BREAK", str(lengths))
    monkeypatch.setenv("SKL_EB_LEAN", str(lean))
    if not tail:
        monkeypatch.setenv("SKL_TAIL_SLICES", "0")
    gpu_ctx.reload_env()
    g = gpu_ctx.sketches(bins, n, KMERS, ss64)
    got = skl.self_dists_all(gpu_ctx, g, g.set_k())
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), np.argwhere(got != exp)[:5]
    name = gpu_ctx.last_kernel()
    part = skl.self_dists_rows(gpu_ctx, g, g.set_k(), 97, n - 33)
    lo = 97 * n - 97 * 98 // 2
    assert np.array_equal(part.view(np.uint32), exp[lo:lo + part.shape[0]].view(np.uint32))
    g_q = gpu_ctx.sketches(bins[150:], n - 150, KMERS, ss64)
    o_q = oracle.Sketches(bins[150:], n - 150, KMERS, ss64)
    cross = skl.cross_dists_all(gpu_ctx, g, g_q, g.set_k())
    assert np.array_equal(cross.view(np.uint32), oracle.cross_dists_all(o, o_q, oracle.COREACC, threads=8).view(np.uint32))
    g.close()
    g_q.close()
    assert ("[lean epilogue" in name) == bool(lean), name
    if lean and n >= 2000:
        assert "[lean epilogue]" in name, name          # (u16 counts)


@pytest.mark.ab_library
@pytest.mark.parametrize("n,ss64,band_rows,knn,order", [(330, 256, 64, 7, 2), (9300, 16, 512, 20, 2), (9300, 16, 512, 20, 0)])
def test_knn_band_epilogue_order(oracle, skl, gpu_ctx, set_switch, n, ss64, band_rows, knn, order):
    """The kNN bands' early-break epilogue walks a launch column group by column group on each XCD (all the band's rows of one group
    of 512 columns before the next, so that the group's column slices stay in that L2) from 32 column groups on;
    SKL_KNN_EPI_BLOCKED=2 (A/B build) forces that order onto views the oracle can check whole -- one group padded to eight, and nineteen
    groups of which the last is ragged, padded to twenty-four -- and 0 the row-major order: ids, order and both distances = the oracle's."""
    bins = _mixed(n, KMERS, ss64, n_random=n - 300, n_clusters=3, seed=29)
    o, g = oracle.Sketches(bins, n, KMERS, ss64), gpu_ctx.sketches(bins, n, KMERS, ss64)
    set_switch("SKL_KNN_BAND_ROWS", band_rows)
    set_switch("SKL_KNN_EPI_BLOCKED", order)
    set_switch("SKL_EARLY_BREAK", 2)
    gpu_ctx.set_knn_ties(skl.TIES_REFERENCE)
    try:
        idx, d0, d1 = skl.self_dists_knn(gpu_ctx, g, g.set_k(), knn)
    finally:
        gpu_ctx.set_knn_ties(skl.TIES_CANONICAL)
    exp = oracle.self_dists_knn(o, knn, oracle.COREACC, 0, False, ties=oracle.TIES_RUST_HEAP, threads=8)
    assert "early break: 2 of 5" in gpu_ctx.last_kernel(), gpu_ctx.last_kernel()
    assert np.array_equal(idx, exp["idx"]), np.argwhere(idx != exp["idx"])[:5]
    assert np.array_equal(d0.view(np.uint32), exp["d0"].view(np.uint32)) and np.array_equal(d1.view(np.uint32), exp["d1"].view(np.uint32))
    g.close()


@pytest.mark.ab_library
@pytest.mark.parametrize("lean", [1, 0])
@pytest.mark.parametrize("lengths", [2, 3, 4])
@pytest.mark.parametrize("n,ss64,cutoff,unit", [(600, 64, 0.64, True), (500, 256, 0.3, True), (430, 300, 0.64, True), (520, 32, 0.5, False)])
def test_lean_epilogue_with_a_completeness_correction(oracle, skl, gpu_ctx, monkeypatch, n, ss64, cutoff, unit, lengths, lean):
    """The lean epilogue under a completeness correction (coreacc_epilogue_lean_kernel<.., COMP = true>): the correction divides J
    by a factor <= 1 when every completeness value lies in (0, 1] -- the host checks the vectors -- so the integer tests stand (a
    count that passes uncorrected passes corrected, a count at or below the chance level is J = 0 either way) and only the
    counts in between (sketchsize64 256 and 300: expected_samebits = 1, min_alive above 2) ask the pair's own values.  A vector with
    a value outside (0, 1] keeps the general kernel.  2 / 3 / 4 lengths forced (A/B build), against the oracle within the
    completeness bar (1e-6), the same (1, 1) / NaN pattern; and the same call with SKL_EB_LEAN=0."""
    bins = _mixed(n, KMERS, ss64, n_random=n - 150, n_clusters=3, seed=47)
    rng = np.random.default_rng(9)
    comp = rng.uniform(0.4, 1.0, n)
    comp[::5] = 1.0
    if not unit:
        comp[3] = 1.25
        comp[77] = 1.6
    o = oracle.Sketches(bins, n, KMERS, ss64, completeness=comp)
    exp = oracle.self_dists_all(o, oracle.COREACC, cutoff=cutoff, threads=8).reshape(-1, 2)
    monkeypatch.setenv("SKL_EARLY_BREAK", str(lengths))
    monkeypatch.setenv("SKL_EB_LEAN", str(lean))
    gpu_ctx.reload_env()
    g = gpu_ctx.sketches(bins, n, KMERS, ss64, completeness=comp)
    p = g.set_k(cutoff=cutoff)
    got = skl.self_dists_all(gpu_ctx, g, p)
    name = gpu_ctx.last_kernel()
    assert np.max(np.abs(got.astype(np.float64) - exp.astype(np.float64))) <= 1e-6
    assert np.array_equal(np.isnan(got), np.isnan(exp))
    assert np.array_equal((got[:, 0] == 1.0) & (got[:, 1] == 1.0), (exp[:, 0] == 1.0) & (exp[:, 1] == 1.0))
    nr = n // 3
    g_r = gpu_ctx.sketches(bins[:nr], nr, KMERS, ss64, completeness=comp[:nr])
    g_q = gpu_ctx.sketches(bins[nr:], n - nr, KMERS, ss64, completeness=comp[nr:])
    o_r = oracle.Sketches(bins[:nr], nr, KMERS, ss64, completeness=comp[:nr])
    o_q = oracle.Sketches(bins[nr:], n - nr, KMERS, ss64, completeness=comp[nr:])
    cross = skl.cross_dists_all(gpu_ctx, g_r, g_q, g_r.set_k(cutoff=cutoff))
    exp_c = oracle.cross_dists_all(o_r, o_q, oracle.COREACC, cutoff=cutoff, threads=8)
    assert np.max(np.abs(cross.astype(np.float64) - exp_c.astype(np.float64))) <= 1e-6
    for x in (g, g_r, g_q):
        x.close()
    assert ("[lean epilogue" in name) == bool(lean and unit), name
