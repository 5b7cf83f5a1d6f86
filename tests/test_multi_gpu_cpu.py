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
