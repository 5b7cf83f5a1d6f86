"""The N > 1 path on CPU: row-band partition maths, and a world_size-2 gloo run of the
gather that assembles the condensed matrix on rank 0."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from sketchlib.rust_amd import multi_gpu


@pytest.mark.parametrize("n", [2, 3, 10, 1000, 2829])
@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_self_band_slices_cover_triangle(n, world):
    slices = multi_gpu.self_band_slices(n, world)
    total = n * (n - 1) // 2
    assert len(slices) == world
    pos, row = 0, 0
    for r0, r1, p0, cnt in slices:
        assert r0 == row and p0 == pos and r1 >= r0
        # pairs in rows [r0, r1): sum_{i=r0}^{r1-1} (n-1-i)
        assert cnt == sum(n - 1 - i for i in range(r0, min(r1, n)))
        pos += cnt
        row = r1
    assert pos == total and row == n
    if n >= 1000:
        counts = [s[3] for s in slices]
        assert max(counts) - min(counts) <= 2 * n  # each boundary is within one row of ideal


def test_even_row_bounds():
    assert multi_gpu.even_row_bounds(10, 3) == [0, 3, 6, 10]
    assert multi_gpu.even_row_bounds(0, 2) == [0, 0, 0]


WORKER = r"""
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
from sketchlib.rust_amd import multi_gpu, synth
from oracle import oracle as O
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
n, kmers, ss64 = 61, [17, 21, 25], 4
s = O.Sketches(synth.set_r(n, kmers, ss64, n_clusters=5), n, kmers, ss64)
ref = torch.from_numpy(O.self_dists_all(s))      # every rank can produce the truth
slices = multi_gpu.self_band_slices(n, world)
r0, r1, p0, cnt = slices[rank]
mine = ref[p0:p0 + cnt].clone()                  # stand-in for this rank's GPU band
if rank == 0:
    full = torch.full_like(ref, -1.0)
    full[p0:p0 + cnt] = mine
else:
    full = None
multi_gpu.gather_to_root(full, mine, slices, rank, world, dist)
if rank == 0:
    assert torch.equal(full, ref), "assembled matrix differs"
    print("GATHER_OK")
# the pipelined form bench.py uses with RCCL: several steps in flight, two rotating band buffers
# per sender, a different payload every step (so a buffer overwritten too early would show)
if rank == 0:
    full.fill_(-1.0)
pipe = multi_gpu.PipelinedGather(full, slices, rank, world, dist, depth=2)
bufs = [torch.empty_like(mine), torch.empty_like(mine)]
steps = 7
for it in range(steps):
    b = bufs[it %% 2]
    b.copy_(ref[p0:p0 + cnt] + it)               # "compute" of step `it`
    if rank == 0:
        full[p0:p0 + cnt] = b
    pipe.submit(b)
pipe.drain()
if rank == 0:
    assert torch.equal(full, ref + (steps - 1)), "pipelined gather: last step's matrix differs"
    print("PIPELINE_OK")
# bands cut into messages: a band travels as several sends that pair up with the receives in order
for max_elems in (1, 7, 100, 10 ** 9):
    if rank == 0:
        full.fill_(-1.0)
        full[p0:p0 + cnt] = mine
    multi_gpu.gather_to_root(full, mine, slices, rank, world, dist, max_elems=max_elems)
    if rank == 0:
        assert torch.equal(full, ref), max_elems
    pipe = multi_gpu.PipelinedGather(full, slices, rank, world, dist, depth=2, max_elems=max_elems)
    for it in range(3):
        b = bufs[it %% 2]
        b.copy_(ref[p0:p0 + cnt] + 10 * it)
        if rank == 0:
            full[p0:p0 + cnt] = b
        pipe.submit(b)
    pipe.drain()
    if rank == 0:
        assert torch.equal(full, ref + 20), max_elems
if rank == 0:
    print("CHUNKS_OK")
# --gather host: every rank writes its band into its offsets of one shared host buffer
hg = multi_gpu.HostGather(ref.shape[0], ref.shape[1], slices, rank, world, dist, tag="test%%d" %% os.getpid() if world == 1 else "test_w%%d" %% world)
for it in range(4):
    b = bufs[it %% 2]
    b.copy_(ref[p0:p0 + cnt] + it)
    hg.submit(b)
hg.drain()
dist.barrier()
if rank == 0:
    assert torch.equal(hg.assembled(), ref + 3), "host gather: assembled matrix differs"
    print("HOST_GATHER_OK")
hg.close()
assert rank != 0 or not os.path.exists(hg.path)
dist.barrier()
dist.destroy_process_group()
""" % ROOT


def test_gloo_world2_gather(oracle, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    res = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
         "--master-addr", "127.0.0.1", "--master-port", "29531", str(script)],
        env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert "GATHER_OK" in res.stdout and "PIPELINE_OK" in res.stdout
    assert "CHUNKS_OK" in res.stdout and "HOST_GATHER_OK" in res.stdout


def test_message_chunks():
    assert multi_gpu.message_chunks(0) == []
    assert multi_gpu.message_chunks(5, 2) == [(0, 2), (2, 2), (4, 1)]
    assert multi_gpu.message_chunks(4, 2) == [(0, 2), (2, 2)]
    # cfg 3 at N = 2: a 2.5e9-row band goes as 19 messages of at most 1 GiB
    c = multi_gpu.message_chunks(2_499_975_000)
    assert len(c) == 19 and sum(m for _, m in c) == 2_499_975_000 and max(m for _, m in c) * 8 <= 1 << 30


@pytest.mark.parametrize("n_bands,world", [(1, 1), (5, 2), (16, 8), (489, 8), (7, 3)])
def test_knn_band_deal(n_bands, world):
    deal = multi_gpu.knn_band_deal(n_bands, world)
    assert sorted(b for r in deal for b in r) == list(range(n_bands))
    assert all(r == sorted(r) for r in deal)
    if n_bands >= 16 * world:   # band b costs ~ (n_bands - b): the back-and-forth deal evens it out
        cost = [sum(n_bands - b for b in r) for r in deal]
        assert max(cost) - min(cost) <= 2 * n_bands


KNN_WORKER = r"""
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
from sketchlib.rust_amd import multi_gpu, synth
from oracle import oracle as O
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
n, kmers, ss64, knn, band_rows = 97, [17, 21, 25], 4, 6, 10
s = O.Sketches(synth.set_r(n, kmers, ss64, n_clusters=4), n, kmers, ss64)
dense = O.self_dists_all(s, O.JACCARD, 1, False)[:, 0]     # condensed (i < j), f32
sortable = lambda f: (f.view(np.uint32) | np.uint32(0x80000000)).astype(np.uint32)   # keys are >= 0
n_bands = (n + band_rows - 1) // band_rows
mine = multi_gpu.knn_band_deal(n_bands, world)[rank]
# stand-in for skl_self_dists_knn_partial (the GPU half): the pairs {i < j} whose band of i is
# ours, each a candidate of both rows; per row the knn smallest (key, id)
cand = [[] for _ in range(n)]
pos = 0
for i in range(n):
    for j in range(i + 1, n):
        if i // band_rows in mine:
            k = int(sortable(dense[pos:pos + 1])[0])
            cand[i].append((k, j)); cand[j].append((k, i))
        pos += 1
key = np.full((n, knn), 0xFFFFFFFF, dtype=np.uint32); idx = np.full((n, knn), 0xFFFFFFFF, dtype=np.uint32)
for r in range(n):
    best = sorted(cand[r])[:knn]
    for x, (k, j) in enumerate(best):
        key[r, x], idx[r, x] = k, j
bounds = multi_gpu.even_row_bounds(n, world)
tk = torch.from_numpy(key.view(np.int32)); ti = torch.from_numpy(idx.view(np.int32))
k_all, i_all, none = multi_gpu.exchange_knn_states([tk, ti, None], bounds, rank, world, dist)
assert none is None and tuple(k_all.shape) == (world, bounds[rank + 1] - bounds[rank], knn)
# stand-in for skl_knn_merge_states: knn smallest (key, id) of the union
ka = k_all.numpy().view(np.uint32); ia = i_all.numpy().view(np.uint32)
exp = O.self_dists_knn(s, knn, O.JACCARD, 1, False, ties=O.TIES_CANONICAL)
for r in range(bounds[rank], bounds[rank + 1]):
    union = sorted((int(ka[w, r - bounds[rank], x]), int(ia[w, r - bounds[rank], x]))
                   for w in range(world) for x in range(knn))[:knn]
    assert [j for _, j in union] == exp["idx"][r].tolist(), (rank, r)
    assert [k for k, _ in union] == sortable(exp["d0"][r]).tolist()
print("KNN_ONCE_OK", rank)
dist.barrier()
dist.destroy_process_group()
""" % ROOT


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_knn_every_pair_once_exchange(oracle, tmp_path, world):
    """The all-to-all of partial kNN states: deal of the bands, routing of the row shards, and
    that the union of the ranks' partial lists holds every row's true neighbours."""
    script = tmp_path / "knn_worker.py"
    script.write_text(KNN_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    res = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
         "--master-addr", "127.0.0.1", "--master-port", str(29540 + world), str(script)],
        env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert res.stdout.count("KNN_ONCE_OK") == world


@pytest.mark.parametrize("n,band_rows,world", [(1000000, 2048, 8), (700, 64, 3), (97, 10, 2), (50, 64, 4), (1000, 16, 1)])
def test_knn_window_cuts(n, band_rows, world):
    """Column windows of the travelling-heaps pipeline: ascending, on band boundaries, balanced by pair count."""
    cuts = multi_gpu.knn_window_cuts(n, band_rows, world)
    assert len(cuts) == world + 1 and cuts[0] == 0 and cuts[-1] == n and cuts == sorted(cuts)
    assert all(c % band_rows == 0 or c == n for c in cuts[:-1])
    if n >= 100 * band_rows * world:      # rank r evaluates ~ (hi^2 - lo^2) / 2 pairs
        area = [(cuts[r + 1] ** 2 - cuts[r] ** 2) / 2 for r in range(world)]
        assert max(area) / min(area) < 1.05


HEAP_WORKER = r"""
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
from sketchlib.rust_amd import multi_gpu, synth
from oracle import oracle as O
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
mode = sys.argv[1]
form = sys.argv[2] if len(sys.argv) > 2 else "travelling"     # travelling | decoupled | overflow
n, kmers, ss64, knn, band_rows = 131, [17, 21, 25], 4, 7, 16
bins = synth.set_r(n, kmers, ss64, n_clusters=3)
bins[40] = bins[7]; bins[99] = bins[7]; bins[100] = bins[7]       # exact ties inside a cluster
s = O.Sketches(bins, n, kmers, ss64)
coreacc = mode == "coreacc"
dtype, k_idx = (O.COREACC, 0) if coreacc else (O.JACCARD, 1)
def sub(a, b):
    return O.Sketches(bins[a:b], b - a, kmers, ss64)

class P:      # what the driver reads of skl_dist_params
    dist_type = 0 if coreacc else 1
    ani = 0
class SK:
    pass
SK.n = n

evaluated = [0]
def load_row(h, row):
    hp = O.Heaps(1, knn)
    m = int(h["h_len"][row])
    hp.len[0] = m
    hp.items["d0"][0, :m] = h["h_key"][row, :m].numpy()
    hp.items["idx"][0, :m] = h["h_id"][row, :m].numpy().astype(np.uint64)
    if coreacc:
        hp.items["d1"][0, :m] = h["h_d1"][row, :m].numpy()
    return hp
def store_row(h, row, hp):
    m = int(hp.len[0])
    h["h_len"][row] = m
    h["h_key"][row, :m] = torch.from_numpy(hp.items["d0"][0, :m].copy())
    h["h_id"][row, :m] = torch.from_numpy(hp.items["idx"][0, :m].astype(np.int32))
    if coreacc:
        h["h_d1"][row, :m] = torch.from_numpy(hp.items["d1"][0, :m].copy())

def log_taken(lg, row, took, ids, keys, d1):
    # append what the heap took, in order (skl_self_dists_knn_window_logged's accept log; the length counts past the capacity)
    for c in np.flatnonzero(took):
        m = int(lg["len"][row])
        if m < lg["cap"]:
            lg["rec"][row, m, 0] = float(keys[c])
            if coreacc:
                lg["rec"][row, m, 1] = float(d1[c])
            lg["id"][row, m] = int(ids[c])
        lg["len"][row] = m + 1

def stage(band, lo, hi, h, lg=None):
    # stand-in for skl_self_dists_knn_window[_logged] (the GPU half), from the oracle's distances and its resumable BinaryHeap
    b0, b1 = band * band_rows, min(n, (band + 1) * band_rows)
    c_first, t_first = max(b0, lo), max(b1, lo)
    if c_first >= hi:
        return
    D = O.cross_dists_all(sub(b0, b1), sub(c_first, hi), dtype, k_idx)       # [band, cols, 1 or 2]
    evaluated[0] += sum(1 for i in range(b0, b1) for j in range(c_first, hi) if i < j)
    for j in range(t_first, hi):            # the window's rows below the band: the band's samples, ascending
        hp = load_row(h, j)
        ids_, k_, d_ = np.arange(b0, b1), D[:, j - c_first, 0], (D[:, j - c_first, 1] if coreacc else None)
        took = hp.feed(0, ids_, k_, d_, log=lg is not None)
        if lg is not None:
            log_taken(lg, j, took, ids_, k_, d_)
        store_row(h, j, hp)
    for i in range(b0, b1):                 # the band's own rows: the window's columns from max(b0, lo) on, minus themselves
        cols = np.array([c for c in range(c_first, hi) if c != i], dtype=np.int64)
        if cols.size == 0:
            continue
        hp = load_row(h, i)
        k_, d_ = D[i - b0, cols - c_first, 0], (D[i - b0, cols - c_first, 1] if coreacc else None)
        took = hp.feed(0, cols, k_, d_, log=lg is not None)
        if lg is not None:
            log_taken(lg, i, took, cols, k_, d_)
        store_row(h, i, hp)

def replay(h, r0, r1, rec, ids, lens):
    # stand-in for skl_knn_heaps_replay: rows [r0, r1) of h take their logged candidates in the order logged
    for r in range(r0, r1):
        m = int(lens[r - r0])
        if m == 0:
            continue
        hp = load_row(h, r)
        hp.feed(0, ids[r - r0, :m].numpy().astype(np.int64), rec[r - r0, :m, 0].numpy(), rec[r - r0, :m, 1].numpy() if coreacc else None)
        store_row(h, r, hp)

def finalize(h, r0, r1):
    idx = torch.zeros((r1 - r0, knn), dtype=torch.int64); d0 = torch.zeros((r1 - r0, knn)); d1 = torch.zeros((r1 - r0, knn)) if coreacc else None
    for r in range(r0, r1):
        hp = load_row(h, r)
        rows = hp.sorted_rows()[0]
        idx[r - r0] = torch.from_numpy(rows["idx"].astype(np.int64)); d0[r - r0] = torch.from_numpy(rows["d0"].copy())
        if coreacc:
            d1[r - r0] = torch.from_numpy(rows["d1"].copy())
    return idx, d0, d1

heaps = {"h_key": torch.zeros((n, knn)), "h_id": torch.zeros((n, knn), dtype=torch.int32),
         "h_d1": torch.zeros((n, knn)) if coreacc else None, "h_len": torch.zeros((n,), dtype=torch.int32),
         "thr": torch.full((n,), -1, dtype=torch.int32)}
def new_logs(cap):
    return {"rec": torch.zeros((n, cap, 2 if coreacc else 1)), "id": torch.zeros((n, cap), dtype=torch.int32),
            "len": torch.zeros((n,), dtype=torch.int32), "cap": cap}
if form == "travelling":
    res = multi_gpu.self_knn_once_reference(None, SK, P, knn, rank, world, dist, torch.device("cpu"),
                                            band_rows=band_rows, stage=stage, finalize=finalize, heaps=heaps)
else:
    # the decoupled form: every rank its own window against empty heaps, logs exchanged, replayed in window order
    cap = 3 if form == "overflow" else 64
    res = multi_gpu.self_knn_once_reference_decoupled(None, SK, P, knn, rank, world, dist, torch.device("cpu"), band_rows=band_rows,
                                                      stage=stage, replay=replay, finalize=finalize, heaps=heaps, logs=new_logs(cap),
                                                      log_cap=cap, host_staged=True)
    if form == "overflow":
        assert res is None, "a log of 3 entries cannot hold what a heap of 7 takes"
        evaluated[0] = 0
        for k in heaps:
            if heaps[k] is not None:
                heaps[k].zero_()
        heaps["thr"].fill_(-1)
        res = multi_gpu.self_knn_once_reference(None, SK, P, knn, rank, world, dist, torch.device("cpu"),
                                                band_rows=band_rows, stage=stage, finalize=finalize, heaps=heaps)
r0, r1, idx, d0, d1 = res
exp = O.self_dists_knn(s, knn, dtype, k_idx, False, ties=O.TIES_RUST_HEAP)
assert (r0, r1) == tuple(multi_gpu.even_row_bounds(n, world)[rank:rank + 2])
assert np.array_equal(idx.numpy().astype(np.uint64), exp["idx"][r0:r1]), (rank, "ids / order differ from the reference's heap")
assert np.array_equal(d0.numpy(), exp["d0"][r0:r1])
if coreacc:
    assert np.array_equal(d1.numpy(), exp["d1"][r0:r1])
canon = O.self_dists_knn(s, knn, dtype, k_idx, False, ties=O.TIES_CANONICAL)
tot = torch.tensor([evaluated[0], int((exp["idx"] != canon["idx"]).any())], dtype=torch.int64)
dist.all_reduce(tot)
assert int(tot[0]) == n * (n - 1) // 2, "every pair evaluated exactly once over the ranks"
assert int(tot[1]) > 0, "the data set is meant to have ties the two rules resolve differently"
print("HEAPS_OK", rank)
dist.barrier()
dist.destroy_process_group()
""" % ROOT


@pytest.mark.parametrize("form", ["decoupled", "overflow"])
@pytest.mark.parametrize("mode", ["jaccard", "coreacc"])
@pytest.mark.parametrize("world", [1, 2, 3, 4])
def test_gloo_reference_order_decoupled_windows(oracle, tmp_path, world, mode, form):
    """The same lists with NO rank waiting for another (round 6): every rank runs its column window against heaps that start
    empty and logs what they take; a row's true list is the replay of its logs in window order on the rank that finalises it
    (multi_gpu.self_knn_once_reference_decoupled).  ids, order, both distances = the oracle's whole-row BinaryHeap replay;
    every pair evaluated exactly once; a log too short for what a heap takes sends every rank back to the travelling heaps."""
    if form == "overflow" and world == 4:
        pytest.skip("one overflow case per world size up to 3 is enough")
    script = tmp_path / "heap_worker.py"
    script.write_text(HEAP_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    res = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
         "--master-addr", "127.0.0.1", "--master-port", str(29600 + world + (10 if mode == "coreacc" else 0) + (20 if form == "overflow" else 0)),
         str(script), mode, form],
        env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    assert res.stdout.count("HEAPS_OK") == world


@pytest.mark.parametrize("mode", ["jaccard", "coreacc"])
@pytest.mark.parametrize("world", [1, 2, 3])
def test_gloo_reference_order_travelling_heaps(oracle, tmp_path, world, mode):
    """The reference's tie order over several ranks with every pair evaluated once: column windows, heaps handed from rank
    to rank band by band, lists gathered on the last rank and dealt back -- against the oracle's BinaryHeap replay of whole
    rows (mod.rs:133-224), with the GPU half replaced by the oracle's distances and its resumable heap."""
    script = tmp_path / "heap_worker.py"
    script.write_text(HEAP_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    res = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
         "--master-addr", "127.0.0.1", "--master-port", str(29560 + world + (10 if mode == "coreacc" else 0)), str(script), mode],
        env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    assert res.stdout.count("HEAPS_OK") == world
