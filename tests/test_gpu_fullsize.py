"""BASELINE.json-sized runs (cfg 2: n = 1000, sketchsize64 = 64, k = {15,19,23,27,31}) and
larger launches that exercise the LDS kernel's tile shapes, checked through
size-independent properties and oracle spot checks (the oracle finishes these in seconds
because only sampled pairs / the n = 1000 case are recomputed on the CPU)."""
import numpy as np
import pytest

from sketchlib.rust_amd import synth

pytestmark = pytest.mark.gpu
K5 = [15, 19, 23, 27, 31]
SS64 = 64


def cond(i, j, n):
    return n * i - (i * (i + 1)) // 2 + j - 1 - i


def make_related(n, n_clusters=40, chunk=500):
    parts = [synth.set_r(min(chunk, n - s0), K5, SS64, n_clusters=n_clusters, first_sample=s0)
             for s0 in range(0, n, chunk)]
    return np.concatenate(parts)


@pytest.fixture(scope="module")
def cfg2(oracle, skl, _product_ctx):
    gpu_ctx = _product_ctx
    n = 1000
    bins = make_related(n)
    bins[777] = bins[3]          # exact duplicates: J = 1 at every k
    bins[998] = bins[999]
    return n, bins, oracle.Sketches(bins, n, K5, SS64), gpu_ctx.sketches(bins, n, K5, SS64)


def test_cfg2_full_parity(oracle, skl, gpu_ctx, cfg2):
    """The whole cfg-2 matrix against the oracle: counts bit-exact, distances <= 1e-6."""
    n, _bins, o, g = cfg2
    got = skl.self_dists_all(gpu_ctx, g, g.set_k())
    exp = oracle.self_dists_all(o, threads=8)
    np.testing.assert_allclose(got, exp, atol=1e-6, rtol=0)
    assert np.array_equal(got, exp)
    assert np.array_equal(skl.self_binmatch(gpu_ctx, g), oracle.self_binmatch(o, threads=8))
    # identical sketches: y = ln 1 = 0 at every k -> y_diff = 0 -> r = NaN -> (0, 0)
    assert got[cond(3, 777, n)].tolist() == [0.0, 0.0]
    assert got[cond(998, 999, n)].tolist() == [0.0, 0.0]
    # the regression is really exercised: a healthy share of pairs is neither (1,1) nor (0,0)
    interesting = np.mean((got[:, 0] > 0) & (got[:, 0] < 1))
    assert interesting > 0.01


def test_cfg2_self_vs_cross_and_symmetry(skl, gpu_ctx, cfg2):
    n, _bins, _o, g = cfg2
    p = g.set_k()
    self_d = skl.self_dists_all(gpu_ctx, g, p)
    cross = skl.cross_dists_all(gpu_ctx, g, g, p)
    iu = np.triu_indices(n, 1)
    assert np.array_equal(cross[iu], self_d)           # condensed order == row-major upper triangle
    assert np.array_equal(cross, cross.transpose(1, 0, 2))
    pj = g.set_k(23)
    dj = skl.cross_dists_all(gpu_ctx, g, g, pj)[:, :, 0]
    assert np.all(np.diag(dj) == 0.0) and np.array_equal(dj, dj.T)
    ani = skl.cross_dists_all(gpu_ctx, g, g, g.set_k(23, True))[:, :, 0]
    assert np.all(np.diag(ani) == 1.0)


def test_cfg2_row_bands_and_device_output(skl, gpu_ctx, cfg2):
    import torch

    from sketchlib.rust_amd import multi_gpu

    n, _bins, _o, g = cfg2
    p = g.set_k()
    whole = skl.self_dists_all(gpu_ctx, g, p)
    for world in (2, 3, 8):
        parts = []
        for r0, r1, p0, cnt in multi_gpu.self_band_slices(n, world):
            out = torch.zeros((cnt, 2), dtype=torch.float32, device="cuda")
            skl.self_dists_rows(gpu_ctx, g, p, r0, r1, out=out)
            gpu_ctx.synchronize()
            parts.append(out.cpu().numpy())
        assert np.array_equal(np.concatenate(parts), whole)


def test_cfg2_knn50(oracle, skl, gpu_ctx, cfg2):
    n, _bins, o, g = cfg2
    for p, oargs in [(g.set_k(), (oracle.COREACC, 0, False)), (g.set_k(19, True), (oracle.JACCARD, 1, True))]:
        idx, d0, d1 = skl.self_dists_knn(gpu_ctx, g, p, 50)
        exp = oracle.self_dists_knn(o, 50, *oargs, ties=oracle.TIES_CANONICAL, threads=8)
        assert np.array_equal(idx, exp["idx"])
        np.testing.assert_allclose(d0, exp["d0"], atol=1e-6, rtol=0)
        heap = oracle.self_dists_knn(o, 50, *oargs, ties=oracle.TIES_RUST_HEAP, threads=8)
        np.testing.assert_allclose(np.sort(d0, axis=1), np.sort(heap["d0"], axis=1), atol=1e-6, rtol=0)


@pytest.mark.parametrize("n", [4000, 7500])   # 8.0e6 pairs -> 8x256 LDS tiles; 2.8e7 -> 16x512
def test_large_launch_spot_checks(oracle, skl, gpu_ctx, n):
    bins = make_related(n, n_clusters=100)
    o = oracle.Sketches(bins, n, K5, SS64)
    g = gpu_ctx.sketches(bins, n, K5, SS64)
    got = skl.self_dists_all(gpu_ctx, g, g.set_k())
    assert np.all(np.isfinite(got))
    rng = np.random.default_rng(n)
    # random pairs + pairs inside clusters (sample s is in cluster s % 100) + tile edges
    ii = rng.integers(0, n - 1, 1500)
    jj = np.minimum(ii + 1 + rng.integers(0, n, 1500) % (n - 1 - ii), n - 1)
    same = rng.integers(0, n - 200, 1500)
    pairs = list(zip(ii, jj)) + list(zip(same, same + 100)) + [(0, 1), (0, n - 1), (n - 2, n - 1),
                                                               (15, 16), (16, 511), (16, 512), (255, 256)]
    for i, j in pairs:
        i, j = int(i), int(j)
        assert tuple(got[cond(i, j, n)]) == oracle.core_acc_pair(o, o, i, j), (i, j)
    # Jaccard mode on the same slab: a full row against the oracle
    dj = skl.self_dists_rows(gpu_ctx, g, g.set_k(27), 0, 1)
    r0 = oracle.Sketches(bins[:1], 1, K5, SS64)
    exp = oracle.cross_dists_all(r0, o, oracle.JACCARD, 3, threads=8)[0, 1:, 0]
    assert np.array_equal(dj[:, 0], exp)


@pytest.mark.ab_library      # (SKL_EARLY_BREAK=0: the all-k fused kernel on Set U, which the sampled early break replaces there)
@pytest.mark.parametrize("early_break", ["0", "1"], ids=["every_length_counted", "early_break"])
def test_every_pair_written_when_kernels_alternate(skl, gpu_ctx, monkeypatch, early_break):
    """4.5e8 pairs into NaN-filled device buffers while the forms of the pair kernel alternate on ONE
    context, as they do in a caller's process: all-k fused core/accessory, k-sliced single-k Jaccard,
    raw counts, core/accessory over row bands small enough to run k-sliced + epilogue.  Every launch
    must write every pair, and the banded results must equal the whole.  (A per-launch tile table that
    was uploaded from a dying host buffer once left 0.6 % of a 5e9-pair launch unwritten when two tile
    shapes alternated; the table is now a closed form evaluated on the device.)"""
    import torch

    from sketchlib.rust_amd import multi_gpu

    n = 30000
    dev = torch.device("cuda", 0)
    monkeypatch.setenv("SKL_EARLY_BREAK", early_break)
    gpu_ctx.reload_env()
    # the NaN fill is torch work on the default stream: run the context there too
    gpu_ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    g = gpu_ctx.sketches(synth.set_u_device(n, 5, SS64, dev), n, K5, SS64)
    pairs = n * (n - 1) // 2
    whole = torch.full((pairs, 2), float("nan"), dtype=torch.float32, device=dev)
    skl.self_dists_all(gpu_ctx, g, g.set_k(), out=whole)
    first_kernel = gpu_ctx.last_kernel()      # (names are checked last: every-pair-written holds under any forced dispatch)
    torch.cuda.synchronize()
    assert not bool(torch.isnan(whole).any().item())
    seen = set()
    for round_ in range(2):
        jac = torch.full((pairs, 1), float("nan"), dtype=torch.float32, device=dev)
        skl.self_dists_all(gpu_ctx, g, g.set_k(23), out=jac)
        seen.add(gpu_ctx.last_kernel().split(" (")[0])
        torch.cuda.synchronize()
        assert not bool(torch.isnan(jac).any().item()), gpu_ctx.last_kernel()
        del jac
        banded = torch.full((pairs, 2), float("nan"), dtype=torch.float32, device=dev)
        for r0, r1, p0, cnt in multi_gpu.self_band_slices(n, 40):       # 1.1e7 pairs per band: k-sliced + epilogue
            skl.self_dists_rows(gpu_ctx, g, g.set_k(), r0, r1, out=banded[p0:p0 + cnt])
            seen.add(gpu_ctx.last_kernel().split(" (")[0])
        again = torch.full((pairs, 2), float("nan"), dtype=torch.float32, device=dev)
        skl.self_dists_all(gpu_ctx, g, g.set_k(), out=again)
        seen.add(gpu_ctx.last_kernel().split(" (")[0])
        torch.cuda.synchronize()
        assert bool((banded == whole).all().item()) and bool((again == whole).all().item()), round_
        del banded, again
    if early_break == "0":
        assert "all k" in first_kernel, first_kernel
        assert len(seen) >= 3, seen      # COREACC all k, JACCARD k-sliced, COUNTS k-sliced
    else:
        assert "early break" in first_kernel, first_kernel
        assert len(seen) >= 2, seen      # COUNTS k-sliced (+ the completing epilogue), JACCARD k-sliced
    g.close()
    gpu_ctx.set_stream(None)


def test_host_output_in_several_bands_equals_device_output(skl, gpu_ctx):
    """A host destination is filled in bands of 512 MB through two device buffers, band i's copy issued behind band i + 1's
    kernels (csrc/capi.cpp dense_rows): 1.6 GB of (core, acc) records = 4 bands (self mode), 0.72 GB of single-k
    distances = 2 bands (cross mode), and a row range that starts mid-matrix, against the same call left on the device."""
    import torch

    dev = torch.device("cuda", 0)
    gpu_ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    n, nq = 20000, 60000
    g = gpu_ctx.sketches(synth.set_u_device(n, 5, SS64, dev), n, K5, SS64)
    pairs = n * (n - 1) // 2
    on_dev = torch.empty((pairs, 2), dtype=torch.float32, device=dev)
    skl.self_dists_all(gpu_ctx, g, g.set_k(), out=on_dev)
    torch.cuda.synchronize()
    host = np.full((pairs, 2), np.nan, dtype=np.float32)
    skl.self_dists_all(gpu_ctx, g, g.set_k(), out=host)
    assert bool(np.array_equal(host, on_dev.cpu().numpy()))
    r0, r1 = 3001, 17003                                   # a row range: its own first band offset
    lo, hi = cond(r0, r0 + 1, n), cond(r1, r1 + 1, n)
    part = np.full((hi - lo, 2), np.nan, dtype=np.float32)
    skl.self_dists_rows(gpu_ctx, g, g.set_k(), r0, r1, out=part)
    assert bool(np.array_equal(part, host[lo:hi]))
    del on_dev, host, part
    q = gpu_ctx.sketches(synth.set_u_device(nq, 5, SS64, dev, first_sample=10**6), nq, K5, SS64)
    nr = 3000
    cross_dev = torch.empty((nr, nq, 1), dtype=torch.float32, device=dev)
    skl.cross_dists_rows(gpu_ctx, g, q, g.set_k(23), 0, nr, out=cross_dev)
    torch.cuda.synchronize()
    cross_host = np.full((nr, nq, 1), np.nan, dtype=np.float32)
    skl.cross_dists_rows(gpu_ctx, g, q, g.set_k(23), 0, nr, out=cross_host)
    assert bool(np.array_equal(cross_host, cross_dev.cpu().numpy()))
    q.close()
    g.close()
