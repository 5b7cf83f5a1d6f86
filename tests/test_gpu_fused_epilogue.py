"""The fused core/accessory epilogue of k-sliced launches (pair_kslice.hip, FUSE; capi.cpp dense_band) -- A/B BUILD ONLY.

Launches of roughly 900 ... 8 000 genomes run one workgroup per (tile, k-mer length) and hand their bin-match counts to a
second kernel.  Round 5 built the one-launch form the round-4 verdict asked for: every workgroup stores its counts
write-through, adds to its tile's arrival counter, and the workgroup that completes a tile finishes the tile's pairs
itself.  It is bit-identical and the arrival pattern costs nothing, but it LOST on time (profiles/r05_fused_epilogue.md), so
it lives in the A/B library behind SKL_FUSE_EPILOGUE=1 as the record of that measurement.  The bar here: bit-identical to the
two-launch form and to the oracle, launch after launch (the counters are never reset: arrivals count modulo the
number of k-mer lengths), across databases with other numbers of k-mer lengths, with half tiles, ragged edges, both
tile heights, self and cross mode, and with a completeness correction (<= 1e-6)."""
import numpy as np
import pytest

from sketchlib.rust_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ab_ctx(skl, monkeypatch):
    """A context of the A/B library (its switches are re-read at every entry point)."""
    import sketchlib.rust_amd as pkg

    monkeypatch.setenv("SKL_EARLY_BREAK", "0")   # (the fused epilogue finishes launches that count every k-mer length)
    with skl.using_library(pkg.build_ab_library()):
        ctx = skl.Context(0)
        yield ctx
        ctx.close()


@pytest.fixture()
def set_switch(monkeypatch):
    def _set(name, value):
        if value is None:
            monkeypatch.delenv(name, raising=False)
        else:
            monkeypatch.setenv(name, str(value))
    return _set


def _both(skl, ctx, call, set_switch):
    set_switch("SKL_FUSE_EPILOGUE", "1")
    fused = call()
    name = ctx.last_kernel()
    set_switch("SKL_FUSE_EPILOGUE", "0")
    plain = call()
    name0 = ctx.last_kernel()
    set_switch("SKL_FUSE_EPILOGUE", "1")
    return fused, plain, (name, name0)


def _names_last(names):
    """(checked after the parity assertions: under scripts/forced_switch_suites.sh other launch forms are forced, which
    have no fused epilogue -- every parity assertion must still hold)"""
    name, name0 = names
    assert "fused core/accessory epilogue" in name and "fused" not in name0, "KERNEL NAME (all parity assertions passed): " + name


@pytest.mark.parametrize("n,kmers,ss64", [(1000, [15, 19, 23, 27, 31], 64), (1250, [17, 21, 25], 37), (1100, [13, 17, 21, 25, 29, 33], 16),
                                          (2500, [15, 19, 23, 27, 31], 8)])
def test_self_matrix_fused_equals_two_launches_and_the_oracle(oracle, skl, ab_ctx, set_switch, n, kmers, ss64):
    bins = synth.set_r(n, kmers, ss64, n_clusters=23)
    bins[n - 3] = bins[5]                       # identical sketches: the flat fit (0, 0)
    o, g = oracle.Sketches(bins, n, kmers, ss64), ab_ctx.sketches(bins, n, kmers, ss64)
    p = g.set_k()
    fused, plain, names = _both(skl, ab_ctx, lambda: skl.self_dists_all(ab_ctx, g, p), set_switch)
    assert np.array_equal(fused, plain)
    assert np.array_equal(fused, oracle.self_dists_all(o, threads=8))
    # launch after launch: the arrival counters run on (modulo the number of k-mer lengths)
    for _ in range(7):
        assert np.array_equal(skl.self_dists_all(ab_ctx, g, p), fused)
    g.close()
    _names_last(names)


def test_32_row_tiles_and_cross_mode(oracle, skl, ab_ctx, set_switch):
    """From 8 Mi pair x k evaluations the k-sliced launches take 32 x 128 tiles; cross mode has no half tiles and ragged
    edges on both sides."""
    kmers, ss64, nr, nq = [15, 19, 23, 27, 31], 16, 2111, 1009
    rb, qb = synth.set_r(nr, kmers, ss64, n_clusters=31), synth.set_r(nq, kmers, ss64, n_clusters=31, first_sample=5000)
    o_r, o_q = oracle.Sketches(rb, nr, kmers, ss64), oracle.Sketches(qb, nq, kmers, ss64)
    g_r, g_q = ab_ctx.sketches(rb, nr, kmers, ss64), ab_ctx.sketches(qb, nq, kmers, ss64)
    p = g_r.set_k()
    fused, plain, names = _both(skl, ab_ctx, lambda: skl.cross_dists_all(ab_ctx, g_r, g_q, p), set_switch)
    assert np.array_equal(fused, plain) and np.array_equal(fused, oracle.cross_dists_all(o_r, o_q, threads=8))
    fused_s, plain_s, names_s = _both(skl, ab_ctx, lambda: skl.self_dists_all(ab_ctx, g_r, p), set_switch)
    assert np.array_equal(fused_s, plain_s) and np.array_equal(fused_s, oracle.self_dists_all(o_r, threads=8))
    g_r.close()
    g_q.close()
    _names_last(names)
    _names_last(names_s)


def test_counters_survive_a_change_of_database(oracle, skl, ab_ctx, set_switch):
    """Arrivals are counted modulo the number of k-mer lengths: a database with another number re-zeroes the counters."""
    set_switch("SKL_FUSE_EPILOGUE", "1")
    outs, names = [], []
    for kmers in ([15, 19, 23], [13, 17, 21, 25, 29], [15, 19, 23], [17, 21]):
        n, ss64 = {3: 1300, 5: 1100, 2: 1500}[len(kmers)], 32      # (sizes at which a launch is neither tail-sliced nor in the mid band)
        bins = synth.set_r(n, kmers, ss64, n_clusters=17)
        o, g = oracle.Sketches(bins, n, kmers, ss64), ab_ctx.sketches(bins, n, kmers, ss64)
        got = skl.self_dists_all(ab_ctx, g, g.set_k())
        names.append(ab_ctx.last_kernel())
        assert np.array_equal(got, oracle.self_dists_all(o, threads=8)), kmers
        outs.append(got)
        g.close()
    assert np.array_equal(outs[0], outs[2])
    assert all("fused" in x for x in names), "KERNEL NAME (all parity assertions passed): " + names[0]


def test_fused_with_a_completeness_correction(oracle, skl, ab_ctx, set_switch):
    kmers, ss64, n = [15, 19, 23, 27, 31], 16, 1200
    bins = synth.set_r(n, kmers, ss64, n_clusters=19)
    comp = np.linspace(0.65, 1.0, n)
    o = oracle.Sketches(bins, n, kmers, ss64, completeness=comp)
    g = ab_ctx.sketches(bins, n, kmers, ss64)
    g.set_completeness(comp)
    fused, plain, names = _both(skl, ab_ctx, lambda: skl.self_dists_all(ab_ctx, g, g.set_k()), set_switch)
    assert np.array_equal(fused, plain)
    np.testing.assert_allclose(fused, oracle.self_dists_all(o, threads=8), atol=1e-6, rtol=0)
    g.close()
    _names_last(names)


def test_row_bands_of_the_multi_gpu_partition(oracle, skl, ab_ctx, set_switch):
    """skl_self_dists_rows (a rank's band of the condensed triangle) through the fused form equals the whole matrix's slice."""
    set_switch("SKL_FUSE_EPILOGUE", "1")
    kmers, ss64, n = [15, 19, 23, 27, 31], 16, 1200
    bins = synth.set_r(n, kmers, ss64, n_clusters=19)
    g = ab_ctx.sketches(bins, n, kmers, ss64)
    p = g.set_k()
    whole = skl.self_dists_all(ab_ctx, g, p)
    off = 0
    for r0, r1 in ((0, 400), (400, 409), (409, 1100), (1100, n)):
        band = skl.self_dists_rows(ab_ctx, g, p, r0, r1)
        assert np.array_equal(band, whole[off:off + band.shape[0]]), (r0, r1)
        off += band.shape[0]
    assert off == whole.shape[0]
    g.close()
