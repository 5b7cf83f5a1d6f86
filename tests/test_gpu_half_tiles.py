"""Half tiles of pair_kernel_kslice (16 x 128 form): a 64-column block of a tile that holds no pair of the
launch is neither loaded nor computed -- block 0 of a diagonal tile whose rows all lie at or below its
columns (self mode, reference: only i < j is evaluated, src/distances/mod.rs:77-127), block 1 when the
columns end in block 0.  Every shape below makes some workgroups take each of the three walks; results
must equal the oracle bit for bit (u32 counts) / exactly (table-driven f32)."""
import numpy as np
import pytest

from sketchlib.rust_amd import synth

pytestmark = pytest.mark.gpu

K5 = [15, 19, 23, 27, 31]


def _both(oracle, ctx, bins, n, kmers, ss64):
    return oracle.Sketches(bins, n, kmers, ss64), ctx.sketches(bins, n, kmers, ss64)


@pytest.mark.parametrize("n", [64, 65, 127, 129, 191, 193, 257, 448, 900, 1000])
def test_self_counts_and_coreacc(oracle, skl, gpu_ctx, n):
    """n picks the mix: 65..128 -> one column group (diagonal tiles only); 129, 193, 257, 900 -> an odd
    number of 64-column blocks (block 1 of the last group empty); 1000 -> BASELINE configs[1]."""
    ss64, kmers = (64, K5) if n >= 900 else (8, [17, 21, 25])
    bins = synth.set_r(n, kmers, ss64, n_clusters=max(2, n // 40))
    o, g = _both(oracle, gpu_ctx, bins, n, kmers, ss64)
    assert np.array_equal(skl.self_binmatch(gpu_ctx, g), oracle.self_binmatch(o, threads=8))
    assert np.array_equal(skl.self_dists_all(gpu_ctx, g, g.set_k()), oracle.self_dists_all(o, threads=8))
    p = g.set_k(kmers[1])
    assert np.array_equal(skl.self_dists_all(gpu_ctx, g, p), oracle.self_dists_all(o, oracle.JACCARD, 1, False, threads=8))


@pytest.mark.parametrize("r0,r1", [(0, 17), (1, 80), (63, 64), (65, 130), (77, 299), (127, 300), (250, 300)])
def test_self_row_ranges(oracle, skl, gpu_ctx, r0, r1):
    """Row ranges move the tile grid off the 16-row raster (a0 = r0 + 16 t), so the 'every column of block
    0 is at or below the tile's first row' test is exercised at every alignment."""
    n, ss64, kmers = 300, 8, [17, 21, 25, 29]
    bins = synth.set_r(n, kmers, ss64, n_clusters=9)
    o, g = _both(oracle, gpu_ctx, bins, n, kmers, ss64)
    full = oracle.self_dists_all(o, threads=8)
    lo = skl.self_pairs(n, 0, r0)
    cnt = skl.self_pairs(n, r0, r1)
    got = skl.self_dists_rows(gpu_ctx, g, g.set_k(), r0, r1)
    assert got.shape[0] == cnt
    assert np.array_equal(got, full[lo:lo + cnt])


@pytest.mark.parametrize("nq", [1, 64, 65, 130, 200, 321])
def test_cross_odd_block_counts(oracle, skl, gpu_ctx, nq):
    """Cross mode never skips block 0; block 1 is skipped in the last group when ceil(nq / 64) is odd."""
    kmers, ss64, nr = [17, 21, 25, 29], 16, 75
    rb = synth.set_r(nr, kmers, ss64, n_clusters=5)
    qb = synth.set_r(nq, kmers, ss64, n_clusters=5, first_sample=500)
    orf, grf = _both(oracle, gpu_ctx, rb, nr, kmers, ss64)
    oq, gq = _both(oracle, gpu_ctx, qb, nq, kmers, ss64)
    assert np.array_equal(skl.cross_binmatch(gpu_ctx, grf, gq), oracle.cross_binmatch(orf, oq, threads=8))
    assert np.array_equal(skl.cross_dists_all(gpu_ctx, grf, gq, grf.set_k()), oracle.cross_dists_all(orf, oq, threads=8))


def test_self_knn_through_half_tiles(oracle, skl, gpu_ctx):
    """The symmetric self kNN stores turned tiles and row flags from the same workgroups."""
    n, ss64, kmers, knn = 700, 16, [17, 21, 25, 29], 9
    bins = synth.set_r(n, kmers, ss64, n_clusters=20)
    o, g = _both(oracle, gpu_ctx, bins, n, kmers, ss64)
    for p, args in ((g.set_k(21), (oracle.JACCARD, 1, False)), (g.set_k(), (oracle.COREACC, 0, False))):
        idx, d0, d1 = skl.self_dists_knn(gpu_ctx, g, p, knn)
        exp = oracle.self_dists_knn(o, knn, *args, ties=oracle.TIES_CANONICAL, threads=8)
        assert np.array_equal(idx, exp["idx"])
        assert np.array_equal(d0, exp["d0"])
