"""The row-by-row kNN over a LARGE candidate set, as the product library dispatches it by itself: from 131 072 candidates on
(single-k keys) the candidates reach a row in ascending column panels and the later panels' tiles are pruned against the
rows' running lists (capi_knn.cpp, knn_rows_banded).  tests/test_gpu_knn_prune.py forces small panels with a
switch of the A/B build; here nothing is forced -- the sizes are the smallest that take the path -- and the oracle checks
every row."""
import numpy as np
import pytest

from sketchlib.rust_amd import synth

pytestmark = pytest.mark.gpu

N_REF, N_QUERY, SS64, KMERS = 140_000, 640, 8, [21]


@pytest.fixture(scope="module")
def big_refs(_product_ctx, oracle):
    import torch

    dev = torch.device("cuda", 0)
    rb = synth.set_clustered_device(N_REF, 1, SS64, dev, cluster_size=100, keep=0.93, scatter=True).cpu().numpy().view(np.uint64)
    qb = synth.set_clustered_device(N_QUERY, 1, SS64, dev, cluster_size=100, keep=0.93, scatter=True, first_sample=N_REF,
                                    n_clusters=N_REF // 100).cpu().numpy().view(np.uint64)
    qb[5] = rb[77_777]           # a query that IS a reference (key 0), and two queries that are each other
    qb[9] = qb[8]
    return rb, qb


@pytest.mark.parametrize("ties", ["reference", "canonical"])
@pytest.mark.parametrize("ani", [False, True], ids=["dist", "ani"])
def test_cross_knn_against_140k_references(oracle, skl, gpu_ctx, big_refs, ties, ani):
    rb, qb = big_refs
    knn = 12
    o_r, o_q = oracle.Sketches(rb, N_REF, KMERS, SS64), oracle.Sketches(qb, N_QUERY, KMERS, SS64)
    g_r, g_q = gpu_ctx.sketches(rb, N_REF, KMERS, SS64), gpu_ctx.sketches(qb, N_QUERY, KMERS, SS64)
    gpu_ctx.set_knn_ties(skl.TIES_REFERENCE if ties == "reference" else skl.TIES_CANONICAL)
    try:
        idx, d0, _ = skl.cross_dists_knn(gpu_ctx, g_r, g_q, g_r.set_k(21, ani), knn)
    finally:
        gpu_ctx.set_knn_ties(skl.TIES_CANONICAL)
    st = gpu_ctx.knn_prune_stats(full=True)
    exp = oracle.cross_dists_knn(o_r, o_q, knn, oracle.JACCARD, 0, ani,
                                 ties=oracle.TIES_RUST_HEAP if ties == "reference" else oracle.TIES_CANONICAL, threads=8)
    assert np.array_equal(idx, exp["idx"]), np.argwhere(idx != exp["idx"])[:5]
    assert np.array_equal(d0, exp["d0"])
    assert idx[5, 0] == 77_777 and d0[5, 0] == (1.0 if ani else 0.0)      # (ani: the identity itself)
    # the panels were taken (4 of 32 768 columns and the rest), and tiles of the later ones left early
    assert st["tiles"] == (N_QUERY // 32) * (4 * (32768 // 128) + (N_REF - 4 * 32768 + 127) // 128) and st["tiles_left_early"] > 0, st
    g_r.close()
    g_q.close()


def test_row_range_of_the_self_knn_over_140k_samples(oracle, skl, gpu_ctx, big_refs):
    """Rows [100 000, 100 640) of the self kNN (a rank's share of a row-sharded run): the canonical list of a row is the cross
    list of that sample against all samples with itself taken out (no two samples of this set are identical)."""
    rb, _ = big_refs
    knn, r0, r1 = 10, 100_000, 100_640
    o_r = oracle.Sketches(rb, N_REF, KMERS, SS64)
    o_q = oracle.Sketches(np.ascontiguousarray(rb[r0:r1]), r1 - r0, KMERS, SS64)
    g = gpu_ctx.sketches(rb, N_REF, KMERS, SS64)
    idx, d0, _ = skl.self_dists_knn(gpu_ctx, g, g.set_k(21), knn, r0, r1)
    st = gpu_ctx.knn_prune_stats(full=True)
    exp = oracle.cross_dists_knn(o_r, o_q, knn + 1, oracle.JACCARD, 0, False, ties=oracle.TIES_CANONICAL, threads=8)
    for r in range(r1 - r0):
        keep = exp["idx"][r] != r0 + r
        assert keep.sum() == knn, r                                  # (the sample itself is its own nearest candidate)
        assert np.array_equal(idx[r], exp["idx"][r][keep]) and np.array_equal(d0[r], exp["d0"][r][keep]), r
    assert st["tiles_left_early"] > 0, st
    g.close()
