"""pair_kernel_kpersist (the persistent form of the k-sliced core/accessory launch, default for launches
with fewer (tile, k) units than workgroup slots): bit-identical to the one-workgroup-per-unit launch and
to the oracle over sequences of launches in ONE context -- the second counts plane it adds into must be
found zero by every launch whatever ran before (another size, the other form, a single-k launch or
raw counts through the same scratch), and the queue counters must be back at zero."""
import numpy as np
import pytest

from sketchlib.rust_amd import synth

pytestmark = pytest.mark.gpu


def _both_forms(skl, ctx, set_switch, call):
    set_switch("SKL_PERSIST", "2")
    a = call()
    name = ctx.last_kernel()
    set_switch("SKL_PERSIST", "0")
    b = call()
    assert "kpersist" not in ctx.last_kernel()
    return a, b, name


@pytest.mark.parametrize("seed", range(6))
def test_sequences_of_launches_in_one_context(oracle, skl, gpu_ctx, set_switch, seed):
    rng = np.random.default_rng(100 + seed)
    took_persistent = 0
    for step in range(7):
        nk = int(rng.integers(2, 7))
        kmers = sorted(rng.choice(np.arange(9, 40), size=nk, replace=False).tolist())
        ss64 = int(rng.choice([8, 16, 24, 64, 128]))
        n = int(rng.integers(3, 700))
        nq = int(rng.integers(1, 300))
        rb = synth.set_r(n, kmers, ss64, n_clusters=int(rng.integers(1, 12)), seed=seed * 31 + step)
        qb = synth.set_r(nq, kmers, ss64, n_clusters=3, first_sample=4000, seed=seed * 31 + step)
        o_r, g_r = oracle.Sketches(rb, n, kmers, ss64), gpu_ctx.sketches(rb, n, kmers, ss64)
        o_q, g_q = oracle.Sketches(qb, nq, kmers, ss64), gpu_ctx.sketches(qb, nq, kmers, ss64)
        p = g_r.set_k()
        a, b, name = _both_forms(skl, gpu_ctx, set_switch, lambda: skl.self_dists_all(gpu_ctx, g_r, p))
        took_persistent += "kpersist" in name
        assert np.array_equal(a, b), (seed, step, "self", n, ss64, kmers)
        assert np.array_equal(a, oracle.self_dists_all(o_r, threads=8)), (seed, step, "self vs oracle")
        a, b, _ = _both_forms(skl, gpu_ctx, set_switch, lambda: skl.cross_dists_all(gpu_ctx, g_r, g_q, p))
        assert np.array_equal(a, b), (seed, step, "cross", n, nq, ss64, kmers)
        assert np.array_equal(a, oracle.cross_dists_all(o_r, o_q, threads=8)), (seed, step, "cross vs oracle")
        if n >= 8:   # a row band in the middle of the triangle
            r0, r1 = n // 3, n // 3 + max(1, n // 4)
            a, b, _ = _both_forms(skl, gpu_ctx, set_switch, lambda: skl.self_dists_rows(gpu_ctx, g_r, p, r0, r1))
            assert np.array_equal(a, b), (seed, step, "rows", n, r0, r1)
        # other users of the same scratch in between: single-k Jaccard, raw counts
        if step % 2 == 0:
            set_switch("SKL_PERSIST", "2")
            j = skl.self_dists_all(gpu_ctx, g_r, g_r.set_k(kmers[0]))
            assert np.array_equal(j, oracle.self_dists_all(o_r, oracle.JACCARD, 0, False, threads=8))
            assert np.array_equal(skl.self_binmatch(gpu_ctx, g_r), oracle.self_binmatch(o_r, threads=8))
        g_r.close()
        g_q.close()
    assert took_persistent >= 5, took_persistent


def test_default_rule_takes_it_for_small_launches_only(oracle, skl, gpu_ctx, set_switch):
    kmers, ss64 = [15, 19, 23, 27, 31], 64
    set_switch("SKL_PERSIST", "1")
    for n, expect in [(90, True), (400, True), (1000, False)]:
        bins = synth.set_r(n, kmers, ss64, n_clusters=7)
        g = gpu_ctx.sketches(bins, n, kmers, ss64)
        got = skl.self_dists_all(gpu_ctx, g, g.set_k())
        assert ("kpersist" in gpu_ctx.last_kernel()) == expect, (n, gpu_ctx.last_kernel())
        if n <= 400:
            assert np.array_equal(got, oracle.self_dists_all(oracle.Sketches(bins, n, kmers, ss64), threads=8))
        g.close()


def test_unsupported_shapes_fall_back(oracle, skl, gpu_ctx, set_switch):
    """sketchsize64 not a multiple of 8: the persistent form does not apply, whatever the switch says."""
    kmers, ss64, n = [13, 17, 21], 20, 150
    bins = synth.set_r(n, kmers, ss64, n_clusters=4)
    g = gpu_ctx.sketches(bins, n, kmers, ss64)
    set_switch("SKL_PERSIST", "2")
    got = skl.self_dists_all(gpu_ctx, g, g.set_k())
    assert "kpersist" not in gpu_ctx.last_kernel()
    assert np.array_equal(got, oracle.self_dists_all(oracle.Sketches(bins, n, kmers, ss64), threads=8))
    g.close()
