"""Every pair-kernel implementation that is built, not only the one the dispatcher picks.  The
PRODUCT library has the chunk-split kernel in both forms (k-sliced + epilogue kernel, all-k fused)
and the ksplit fallback; the A/B library (`make AB=1`, -DSKL_AB; never loaded by the product) adds
the round-2/3 forms of the two tile shapes (SKL_KSLICE_SHAPE) and SKL_KERNEL=ksplit.  The switches
are read when a context is created, so each variant runs in a child process; results must equal the
oracle bit for bit (counts, Jaccard f32, regression outputs) and, with a completeness correction,
within 1e-6 on EVERY pair."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

CHILD = r"""
import sys
sys.path.insert(0, %(root)r)
import numpy as np, torch
from sketchlib.rust_amd import capi, synth
from oracle import oracle
expect = sys.argv[1]
kmers, ss64 = [15, 19, 23, 27, 31], 64
n, nq = 413, 150
bins = synth.set_r(n, kmers, ss64, n_clusters=9)
qb = synth.set_r(nq, kmers, ss64, n_clusters=9, first_sample=1000)
ctx = capi.Context(0)
g, o = ctx.sketches(bins, n, kmers, ss64), oracle.Sketches(bins, n, kmers, ss64)
gq, oq = ctx.sketches(qb, nq, kmers, ss64), oracle.Sketches(qb, nq, kmers, ss64)
# core/accessory, self and cross
got = capi.self_dists_all(ctx, g, g.set_k())
first_kernel = ctx.last_kernel()      # (its name is checked LAST: under a forced switch setting, scripts/forced_switch_suites.sh,
assert np.array_equal(got, oracle.self_dists_all(o, threads=8)), "self coreacc"   #  every parity assertion below still runs)
got = capi.cross_dists_all(ctx, g, gq, g.set_k())
assert np.array_equal(got, oracle.cross_dists_all(o, oq, threads=8)), "cross coreacc"
# single-k Jaccard and ANI
for ani in (False, True):
    got = capi.self_dists_all(ctx, g, g.set_k(23, ani=ani))
    assert np.array_equal(got, oracle.self_dists_all(o, oracle.JACCARD, 2, ani, threads=8)), "jaccard"
# raw bin-match counts
assert np.array_equal(capi.self_binmatch(ctx, g), oracle.self_binmatch(o, threads=8)), "self counts"
assert np.array_equal(capi.cross_binmatch(ctx, g, gq), oracle.cross_binmatch(o, oq, threads=8)), "cross counts"
# completeness correction: ln J is taken on the device with the host libm's own algorithm
# (csrc/glibc_log.hpp), so EVERY pair must agree -- including the flat fits (the same bin-match count
# at every k-mer length the regression takes), whose core distance is 0 or 1 on the last bit of ln J
# (jaccard.rs:120-133; DESIGN.md "Parity bar").  Bar: north_star's 1e-6 on both columns, no pair excluded.
assert capi.log_variant() in (0, 1), "host libm log() is not one of the restated glibc forms"
comp = np.linspace(0.7, 1.0, n)
g.set_completeness(comp)
oc = oracle.Sketches(bins, n, kmers, ss64, completeness=comp)
got = capi.self_dists_all(ctx, g, g.set_k())
ref = oracle.self_dists_all(oc, threads=8)
counts = oracle.self_binmatch(o, threads=8)
used = np.cumprod(counts > 0, axis=1).astype(bool)            # prefix before the first zero count
flat = (used.sum(axis=1) >= 3) & ((counts == counts[:, :1]) | ~used).all(axis=1) & (counts[:, 0] < 64 * ss64)
assert flat.any(), "the data set is meant to contain flat fits"
assert np.allclose(got, ref, rtol=0, atol=1e-6), "completeness, all pairs incl. %%d flat fits" %% int(flat.sum())
gq.set_completeness(np.linspace(1.0, 0.6, nq))
oqc = oracle.Sketches(qb, nq, kmers, ss64, completeness=np.linspace(1.0, 0.6, nq))
got = capi.cross_dists_all(ctx, g, gq, g.set_k())
assert np.allclose(got, oracle.cross_dists_all(oc, oqc, threads=8), rtol=0, atol=1e-6), "completeness, cross"
for ani in (False, True):
    got = capi.self_dists_all(ctx, g, g.set_k(23, ani=ani))
    ref = oracle.self_dists_all(oc, oracle.JACCARD, 2, ani, threads=8)
    assert np.allclose(got, ref, rtol=0, atol=1e-6), "completeness, single k"
assert expect in first_kernel, "KERNEL NAME (all parity assertions passed): " + first_kernel
print("VARIANT_OK", ctx.last_kernel())
""" % {"root": ROOT}

AB = {"SKL_LIBRARY": os.path.join(ROOT, "sketchlib.rust_amd", "csrc", "_build_ab", "libsketchlib_dist_hip.so")}
VARIANTS = [
    # product library
    ({}, "4 chunk slices"),                                        # dispatcher's choice at this size: less than one round of workgroups
    ({"SKL_TAIL_SLICES": "0"}, "k-sliced"),                        # one workgroup per (tile, k)
    ({"SKL_TAIL_SLICES": "2"}, "2 chunk slices"),
    ({"SKL_SLICED_MAX_PAIRS": "0"}, "all k"),                      # all-k fused form
    ({**AB, "SKL_K_SLICES": "1"}, "k-sliced"),                     # k-sliced, whole k-mer lengths (the uniform slices: an A/B switch)
    ({"SKL_TILE32_MIN": "0"}, "R=32, JL=2, COUNTS, k-sliced"),     # 32 x 128 tiles (large launches' shape)
    ({"SKL_TILE32_MIN": "0", "SKL_SLICED_MAX_PAIRS": "0"}, "R=32, JL=2, COREACC, all k"),
    ({"SKL_GROUP_SPAN": "1"}, "k-sliced"),                         # tile numbering: column group by column group
    ({"SKL_GROUP_SPAN": "3", "SKL_TAIL_SLICES": "0"}, "k-sliced"), # ... 3 groups side by side (default: 2)
    ({"SKL_GROUP_SPAN": "4"}, "k-sliced"),
    ({**AB, "SKL_K_SLICES": "2"}, "k-sliced"),                     # ... cut into 2 / 4 / 8 chunk slices
    ({**AB, "SKL_K_SLICES": "4"}, "k-sliced"),
    ({**AB, "SKL_K_SLICES": "8"}, "k-sliced"),
    # tile order for devices that show fewer than 8 XCDs (partitioned MI355X: the C ABI derives it from the CU count)
    ({"SKL_XCDS": "1"}, "4 chunk slices"),
    ({"SKL_XCDS": "2", "SKL_SLICED_MAX_PAIRS": "0"}, "all k"),
    ({"SKL_XCDS": "4", "SKL_TILE32_MIN": "0"}, "R=32, JL=2, COUNTS, k-sliced"),
    # A/B library
    ({**AB}, "4 chunk slices"),
    ({**AB, "SKL_KSLICE_SHAPE": "1651"}, "R=16, JL=2, COUNTS, k-sliced"),   # the 16-row form walked row by row (round 3a)
    ({**AB, "SKL_KSLICE_SHAPE": "1652", "SKL_SLICED_MAX_PAIRS": "0"}, "R=16, JL=2, COREACC, all k"),
    ({**AB, "SKL_KSLICE_SHAPE": "3254"}, "R=32, JL=2, COUNTS, k-sliced"),    # round 2's k-sliced 32-row form (3 waves per SIMD)
    ({**AB, "SKL_KSLICE_SHAPE": "3255", "SKL_SLICED_MAX_PAIRS": "0"}, "R=32, JL=2, COREACC, all k"),   # the all-k 32-row form of round 2 (3 waves per SIMD)
    ({**AB, "SKL_KERNEL": "ksplit"}, "pair_kernel_ksplit"),
]


@pytest.mark.parametrize("env,expect", VARIANTS,
                         ids=[",".join("AB" if k == "SKL_LIBRARY" else f"{k}={v}" for k, v in e.items()) or "default"
                              for e, _ in VARIANTS])
def test_kernel_variant_parity(gpu_ctx, env, expect):
    if "SKL_LIBRARY" in env:
        import sketchlib.rust_amd as pkg
        pkg.build_ab_library()
    res = subprocess.run([sys.executable, "-c", CHILD, expect], env={**os.environ, **env}, capture_output=True,
                         text=True, timeout=600)
    name_line = [l for l in res.stderr.splitlines() if "KERNEL NAME" in l]     # (the child checks the kernel's name last)
    assert res.returncode == 0 and "VARIANT_OK" in res.stdout, name_line[-1] if name_line else (res.stdout[-500:], res.stderr[-2000:])
