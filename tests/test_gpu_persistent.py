"""The two-plane form of the k-sliced core/accessory launch against the plain one and the oracle, over
sequences of launches in ONE context: chunk slices (default for launches of less than one round of
workgroups, forced onto the last round of longer ones here) -- slice 0 of a unit stores its counts, the
others add into a second plane.

Plane 1 must be found zero by every such launch whatever ran before (another size, a single-k launch or
raw counts through the same scratch)."""
import os

import numpy as np
import pytest

from sketchlib.rust_amd import synth

pytestmark = pytest.mark.gpu


def _sequence(oracle, skl, ctx, switch, seed, on, off, marker):
    """`on` / `off`: environment of the form under test and of the plain launch; `marker`: what the
    form's kernel name contains."""
    rng = np.random.default_rng(100 + seed)
    took = 0

    def both(call):
        switch(on)
        a = call()
        name = ctx.last_kernel()
        switch(off)
        b = call()
        assert marker not in ctx.last_kernel()
        return a, b, name

    for step in range(7):
        nk = int(rng.integers(2, 7))
        kmers = sorted(rng.choice(np.arange(9, 40), size=nk, replace=False).tolist())
        ss64 = int(rng.choice([8, 16, 24, 32, 64, 64, 128]))
        if step == 0:
            ss64 = 128      # (every form under test applies to at least one launch of the sequence)
        n = int(rng.integers(3, 700))
        nq = int(rng.integers(1, 300))
        rb = synth.set_r(n, kmers, ss64, n_clusters=int(rng.integers(1, 12)), seed=seed * 31 + step)
        qb = synth.set_r(nq, kmers, ss64, n_clusters=3, first_sample=4000, seed=seed * 31 + step)
        o_r, g_r = oracle.Sketches(rb, n, kmers, ss64), ctx.sketches(rb, n, kmers, ss64)
        o_q, g_q = oracle.Sketches(qb, nq, kmers, ss64), ctx.sketches(qb, nq, kmers, ss64)
        p = g_r.set_k()
        a, b, name = both(lambda: skl.self_dists_all(ctx, g_r, p))
        took += marker in name
        assert np.array_equal(a, b), (seed, step, "self", n, ss64, kmers)
        assert np.array_equal(a, oracle.self_dists_all(o_r, threads=8)), (seed, step, "self vs oracle")
        a, b, _ = both(lambda: skl.cross_dists_all(ctx, g_r, g_q, p))
        assert np.array_equal(a, b), (seed, step, "cross", n, nq, ss64, kmers)
        assert np.array_equal(a, oracle.cross_dists_all(o_r, o_q, threads=8)), (seed, step, "cross vs oracle")
        if n >= 8:   # a row band in the middle of the triangle
            r0, r1 = n // 3, n // 3 + max(1, n // 4)
            a, b, _ = both(lambda: skl.self_dists_rows(ctx, g_r, p, r0, r1))
            assert np.array_equal(a, b), (seed, step, "rows", n, r0, r1)
        if step % 2 == 0:   # other users of the same scratch in between: single-k Jaccard, raw counts
            switch(on)
            j = skl.self_dists_all(ctx, g_r, g_r.set_k(kmers[0]))
            assert np.array_equal(j, oracle.self_dists_all(o_r, oracle.JACCARD, 0, False, threads=8))
            assert np.array_equal(skl.self_binmatch(ctx, g_r), oracle.self_binmatch(o_r, threads=8))
        g_r.close()
        g_q.close()
    return took


SOAK = int(os.environ.get("SKL_FUZZ_SEEDS", "0"))   # a soak run sets SKL_FUZZ_SEEDS=400


@pytest.mark.parametrize("slices", [2, 4, 8])
@pytest.mark.parametrize("seed", range(max(3, SOAK // 20)))
def test_chunk_slices_in_sequences_of_launches(oracle, skl, gpu_ctx, monkeypatch, seed, slices):
    def switch(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        gpu_ctx.reload_env()

    on = {"SKL_TAIL_SLICES": str(slices), "SKL_TAIL_MAX_PCT": "100000000"}
    took = _sequence(oracle, skl, gpu_ctx, switch, seed * 3 + slices, on, {"SKL_TAIL_SLICES": "0", "SKL_TAIL_MAX_PCT": "90"},
                     "chunk slices")
    assert took >= 1, took     # (sketch sizes that are not a multiple of 8 x slices run plain)


def test_default_rule_slices_launches_of_less_than_a_round(oracle, skl, gpu_ctx):
    kmers, ss64 = [15, 19, 23, 27, 31], 64
    names = []
    for n, expect in [(90, True), (400, True), (800, True), (1000, False)]:
        bins = synth.set_r(n, kmers, ss64, n_clusters=7)
        g = gpu_ctx.sketches(bins, n, kmers, ss64)
        got = skl.self_dists_all(gpu_ctx, g, g.set_k())
        names.append((n, expect, gpu_ctx.last_kernel()))
        if n <= 400:
            assert np.array_equal(got, oracle.self_dists_all(oracle.Sketches(bins, n, kmers, ss64), threads=8))
        g.close()
    for n, expect, name in names:      # (after the parity assertions, so that they run under a forced switch setting too)
        assert ("chunk slices" in name) == expect, (n, name)


@pytest.mark.ab_library
@pytest.mark.parametrize("ss64,expect", [(12, None), (20, "3 chunk slices"), (37, "3 chunk slices"), (157, "7 chunk slices")])
def test_any_sketch_size_is_cut_into_whole_stages(oracle, skl, gpu_ctx, set_switch, ss64, expect):
    """Round 4: slices of whole stages (multiples of 8 chunks), the last one shorter -- 20 chunks = 8 + 8 + 4, the 157
    chunks of `-s 10000`  in a launch this small = 6 x 24 + 13; a sketch of fewer than 16 chunks runs plain.  Self, cross and a row band."""
    kmers, n, nq = [13, 17, 21], 150, 40
    bins = synth.set_r(n, kmers, ss64, n_clusters=4)
    qb = synth.set_r(nq, kmers, ss64, n_clusters=4, first_sample=900)
    g, gq = gpu_ctx.sketches(bins, n, kmers, ss64), gpu_ctx.sketches(qb, nq, kmers, ss64)
    o, oq = oracle.Sketches(bins, n, kmers, ss64), oracle.Sketches(qb, nq, kmers, ss64)
    set_switch("SKL_TAIL_MAX_PCT", "100000000")
    got = skl.self_dists_all(gpu_ctx, g, g.set_k())
    name = gpu_ctx.last_kernel()      # (checked last: parity first, whatever a forced switch setting made of the slicing)
    assert np.array_equal(got, oracle.self_dists_all(o, threads=8))
    assert np.array_equal(skl.cross_dists_all(gpu_ctx, g, gq, g.set_k()), oracle.cross_dists_all(o, oq, threads=8))
    assert np.array_equal(skl.self_dists_rows(gpu_ctx, g, g.set_k(), 40, 93), oracle.self_dists_all(o, threads=8)[40 * n - 40 * 41 // 2:93 * n - 93 * 94 // 2])
    for ani in (False, True):
        assert np.array_equal(skl.self_dists_all(gpu_ctx, g, g.set_k(17, ani)), oracle.self_dists_all(o, oracle.JACCARD, 1, ani, threads=8))
    assert np.array_equal(skl.self_binmatch(gpu_ctx, g), oracle.self_binmatch(o, threads=8))
    set_switch("SKL_K_SLICES", "3")        # ... and uniform slices of every unit (three planes of counts)
    set_switch("SKL_TAIL_SLICES", "0")
    assert np.array_equal(skl.self_dists_all(gpu_ctx, g, g.set_k()), oracle.self_dists_all(o, threads=8))
    g.close()
    gq.close()
    assert (expect in name) if expect else ("chunk slices" not in name), name


@pytest.mark.parametrize("ss64,slices", [(64, 4), (32, 4), (32, 2), (128, 8)])
def test_single_k_launches_smaller_than_the_chip(oracle, skl, gpu_ctx, set_switch, ss64, slices):
    """Single-k Jaccard / ANI launches of less than a round of workgroups run as bin-match counts in
    chunk slices + an epilogue launch; same values as the one-workgroup-per-tile kernel and the oracle,
    with and without a completeness correction, self, cross and row bands."""
    kmers, n, nq = [17, 21, 25], 317, 90
    rb = synth.set_r(n, kmers, ss64, n_clusters=6)
    qb = synth.set_r(nq, kmers, ss64, n_clusters=6, first_sample=2000)
    for comp in (False, True):
        rc = np.linspace(0.6, 1.0, n) if comp else None
        qc = np.linspace(1.0, 0.7, nq) if comp else None
        o_r, g_r = oracle.Sketches(rb, n, kmers, ss64, rc), gpu_ctx.sketches(rb, n, kmers, ss64, rc)
        o_q, g_q = oracle.Sketches(qb, nq, kmers, ss64, qc), gpu_ctx.sketches(qb, nq, kmers, ss64, qc)
        for ani in (False, True):
            p = g_r.set_k(21, ani)
            got = {}
            for form, env in (("sliced", str(slices)), ("plain", "0")):
                set_switch("SKL_TAIL_SLICES", env)
                got[form] = (skl.self_dists_all(gpu_ctx, g_r, p), skl.cross_dists_all(gpu_ctx, g_r, g_q, p),
                             skl.self_dists_rows(gpu_ctx, g_r, p, 100, 171))
                assert ("JACCARD" in gpu_ctx.last_kernel()) == (form == "plain"), gpu_ctx.last_kernel()
            for a, b in zip(got["sliced"], got["plain"]):
                assert np.array_equal(a, b), (ss64, slices, comp, ani)
            np.testing.assert_allclose(got["sliced"][0], oracle.self_dists_all(o_r, oracle.JACCARD, 1, ani, threads=8), atol=1e-6, rtol=0)
            np.testing.assert_allclose(got["sliced"][1], oracle.cross_dists_all(o_r, o_q, oracle.JACCARD, 1, ani, threads=8),
                                       atol=1e-6, rtol=0)
        g_r.close()
        g_q.close()
