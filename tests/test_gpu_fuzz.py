"""Seeded random configurations of the whole path against the oracle (GPU): sample counts,
sketch sizes, k-mer lists, modes, completeness and kNN sizes drawn at random; plus the
multi-band kNN path and a cross launch large enough for the 16x512 LDS tiles."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from sketchlib.rust_amd import synth

pytestmark = pytest.mark.gpu
TOL = 1e-6


def _case(seed):
    rng = np.random.default_rng(seed)
    nk = int(rng.integers(1, 8))
    kmers = sorted(rng.choice(np.arange(7, 64), size=nk, replace=False).tolist())
    ss64 = int(rng.choice([1, 2, 3, 7, 16, 33, 64, 100]))
    n = int(rng.integers(2, 260))
    nq = int(rng.integers(1, 130))
    comp = rng.random() < 0.4
    return rng, kmers, ss64, n, nq, comp


@pytest.mark.parametrize("seed", range(int(os.environ.get("SKL_FUZZ_SEEDS", "24"))))   # a soak run sets SKL_FUZZ_SEEDS=400
def test_random_configuration(oracle, skl, gpu_ctx, seed):
    rng, kmers, ss64, n, nq, use_comp = _case(seed)
    clusters = int(rng.integers(1, 9))
    rb = synth.set_r(n, kmers, ss64, n_clusters=clusters, seed=1000 + seed)
    qb = synth.set_r(nq, kmers, ss64, n_clusters=clusters, first_sample=5000, seed=1000 + seed)
    rc = rng.uniform(0.55, 1.0, n) if use_comp else None
    qc = rng.uniform(0.55, 1.0, nq) if use_comp else None
    cutoff = float(rng.choice([0.64, 0.3, 0.9]))
    o_r, g_r = oracle.Sketches(rb, n, kmers, ss64, rc), gpu_ctx.sketches(rb, n, kmers, ss64, rc)
    o_q, g_q = oracle.Sketches(qb, nq, kmers, ss64, qc), gpu_ctx.sketches(qb, nq, kmers, ss64, qc)
    # raw counts: always bit-exact
    assert np.array_equal(skl.self_binmatch(gpu_ctx, g_r), oracle.self_binmatch(o_r, threads=4))
    assert np.array_equal(skl.cross_binmatch(gpu_ctx, g_r, g_q), oracle.cross_binmatch(o_r, o_q, threads=4))
    # single-k modes
    k_idx = int(rng.integers(0, len(kmers)))
    for ani in (False, True):
        p = g_r.set_k(kmers[k_idx], ani, cutoff)
        np.testing.assert_allclose(skl.self_dists_all(gpu_ctx, g_r, p),
                                   oracle.self_dists_all(o_r, oracle.JACCARD, k_idx, ani, cutoff, threads=4),
                                   atol=TOL, rtol=0)
        np.testing.assert_allclose(skl.cross_dists_all(gpu_ctx, g_r, g_q, p),
                                   oracle.cross_dists_all(o_r, o_q, oracle.JACCARD, k_idx, ani, cutoff, threads=4),
                                   atol=TOL, rtol=0)
    if len(kmers) >= 2:
        p = g_r.set_k(cutoff=cutoff)
        got = skl.self_dists_all(gpu_ctx, g_r, p)
        exp = oracle.self_dists_all(o_r, cutoff=cutoff, threads=4)
        np.testing.assert_allclose(got, exp, atol=TOL, rtol=0, equal_nan=True)
        if not use_comp:
            assert np.array_equal(got, exp)      # table-driven ln: bit-identical
        np.testing.assert_allclose(skl.cross_dists_all(gpu_ctx, g_r, g_q, p),
                                   oracle.cross_dists_all(o_r, o_q, cutoff=cutoff, threads=4),
                                   atol=TOL, rtol=0, equal_nan=True)
    # kNN (Jaccard keys), self and cross
    if n >= 3:
        knn = int(rng.integers(1, min(n - 1, 40) + 1))
        p = g_r.set_k(kmers[k_idx], False, cutoff)
        idx, d0, _ = skl.self_dists_knn(gpu_ctx, g_r, p, knn)
        exp = oracle.self_dists_knn(o_r, knn, oracle.JACCARD, k_idx, False, cutoff, ties=oracle.TIES_CANONICAL, threads=4)
        np.testing.assert_allclose(d0, exp["d0"], atol=TOL, rtol=0)
        if not use_comp:
            assert np.array_equal(idx, exp["idx"])
        knn = int(rng.integers(1, min(n, 30) + 1))
        idx, d0, _ = skl.cross_dists_knn(gpu_ctx, g_r, g_q, p, knn)
        exp = oracle.cross_dists_knn(o_r, o_q, knn, oracle.JACCARD, k_idx, False, cutoff, ties=oracle.TIES_CANONICAL, threads=4)
        np.testing.assert_allclose(d0, exp["d0"], atol=TOL, rtol=0)


def test_knn_multiple_bands_subprocess(oracle):
    """Force 7-row bands (child process: the knob stays out of the other tests) and compare with the oracle."""
    code = r"""
import sys; sys.path.insert(0, %r)
import numpy as np
from sketchlib.rust_amd import capi, synth
from oracle import oracle as O
kmers, ss64, n = [17, 21, 25, 29], 8, 61
bins = synth.set_r(n, kmers, ss64, n_clusters=5)
ctx = capi.Context(0); ctx.set_knn_ties(capi.TIES_CANONICAL); g = ctx.sketches(bins, n, kmers, ss64); o = O.Sketches(bins, n, kmers, ss64)
for p, oa in [(g.set_k(), (O.COREACC, 0, False)), (g.set_k(25, True), (O.JACCARD, 2, True))]:
    idx, d0, d1 = capi.self_dists_knn(ctx, g, p, 9)
    exp = O.self_dists_knn(o, 9, *oa, ties=O.TIES_CANONICAL)
    assert np.array_equal(idx, exp["idx"]) and np.allclose(d0, exp["d0"], atol=1e-6, rtol=0)
print("BANDS_OK")
""" % ROOT
    env = dict(os.environ, SKL_KNN_BAND_ROWS="7")
    res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "BANDS_OK" in res.stdout, res.stderr[-2000:]


@pytest.mark.ab_library      # (SKL_EARLY_BREAK: the "fused" cases are the all-k kernel with every k-mer length counted)
@pytest.mark.parametrize("sliced_max,tail,tile32", [("0", "0", None), ("1000000000000", "0", None), ("1000000000000", "1", None),
                                                     ("0", "0", "-1"), ("1000000000000", "0", "-1"), ("1000000000000", "1", "-1")],
                         ids=["fused", "k_sliced", "k_sliced_tail", "fused_16_rows", "k_sliced_16_rows", "k_sliced_16_rows_tail"])
def test_cross_large_launch_ragged_tiles(oracle, skl, gpu_ctx, monkeypatch, sliced_max, tail, tile32):
    """5 000 refs x 2 000 queries = 1e7 pairs in cross mode, ragged on both axes: the large-launch kernel
    (all k-mer lengths fused in one workgroup), the one-workgroup-per-(tile, k) launch, and that launch
    with its last, partial round of workgroups cut into chunk slices (whole units before it, slices that
    add into the second counts plane after: forced, the default rule takes it for short launches only).
    At this size the dispatcher takes 32 x 128 tiles; SKL_TILE32_MIN=-1 keeps the 16 x 128 ones."""
    kmers, ss64 = [15, 19, 23, 27, 31], 64
    nr, nq = 5003, 2001
    rb = np.concatenate([synth.set_r(min(500, nr - s), kmers, ss64, n_clusters=50, first_sample=s) for s in range(0, nr, 500)])
    qb = np.concatenate([synth.set_r(min(500, nq - s), kmers, ss64, n_clusters=50, first_sample=20000 + s) for s in range(0, nq, 500)])
    o_r, g_r = oracle.Sketches(rb, nr, kmers, ss64), gpu_ctx.sketches(rb, nr, kmers, ss64)
    o_q, g_q = oracle.Sketches(qb, nq, kmers, ss64), gpu_ctx.sketches(qb, nq, kmers, ss64)
    monkeypatch.setenv("SKL_SLICED_MAX_PAIRS", sliced_max)
    if sliced_max == "0":
        monkeypatch.setenv("SKL_EARLY_BREAK", "0")
    monkeypatch.setenv("SKL_TAIL_MAX_PCT", "100000000" if tail == "1" else "90")
    if tile32 is not None:
        monkeypatch.setenv("SKL_TILE32_MIN", tile32)
    gpu_ctx.reload_env()
    got = skl.cross_dists_all(gpu_ctx, g_r, g_q, g_r.set_k())
    name = gpu_ctx.last_kernel()      # (checked last: the parity assertions run whatever a forced switch setting made of the dispatch)
    rng = np.random.default_rng(3)
    pairs = list(zip(rng.integers(0, nr, 2500), rng.integers(0, nq, 2500))) + [(0, 0), (nr - 1, nq - 1), (nr - 1, 0),
                                                                               (0, nq - 1), (15, 511), (16, 512), (4999, 1999)]
    for i, j in pairs:
        assert tuple(got[int(i), int(j)]) == oracle.core_acc_pair(o_r, o_q, int(i), int(j)), (i, j)
    # sample (i, j) with j in the same cluster must not be (1, 1): sample s is in cluster s % 50
    same = [(i, j) for i, j in pairs if (int(i) % 50) == ((20000 + int(j)) % 50)]
    assert any(tuple(got[int(i), int(j)]) != (1.0, 1.0) for i, j in same)
    assert ("all k" if sliced_max == "0" else "k-sliced") in name, name
    assert ("chunk slices" in name) == (tail == "1"), name
    assert ("R=16" if tile32 == "-1" else "R=32") in name, name
