"""Multi-GPU partition of the pair space: one process per GPU (torch.distributed, backend
"nccl" == RCCL over xGMI), every rank holding the whole sketch slab.

The path shards by independent row bands:
  * dense self: rows of the condensed upper triangle, balanced by pair count -- each
    band is one contiguous slice of the reference's condensed output array
    (distance_matrix.rs:11-14);
  * dense cross: contiguous bands of reference rows (output index i_ref*n_query + j);
  * kNN: contiguous bands of query rows.
There is no reduction there; the only exchange is assembling the output on rank 0,
done as grouped point-to-point send/recv straight into the final offsets (what
ncclGather does internally, but with per-rank counts and no staging copy).

The one-evaluation self kNN (every pair {i, j} computed once instead of twice) is the
exception: a pair's key is a candidate of BOTH rows, which belong to different ranks, so every
rank keeps partial top-k states for all rows and the ranks exchange row shards of them with
one all-to-all before the final merge (knn_band_deal / exchange_knn_states / self_knn_once).
"""
import numpy as np


def self_row_bounds(n, world):
    """Split rows 0..n-1 of the condensed triangle into `world` contiguous bands with
    (nearly) equal pair counts.  Returns world+1 row boundaries."""
    total = n * (n - 1) // 2
    bounds = [0]
    # pairs in rows [0, r) = r*n - r(r+1)/2; smallest r reaching target w*total/world
    rows = np.arange(n + 1, dtype=np.int64)
    cum = rows * n - rows * (rows + 1) // 2
    cum[-1] = total  # row n-1 has no pairs
    for w in range(1, world):
        target = (total * w + world - 1) // world
        r = int(np.searchsorted(cum, target, side="left"))
        bounds.append(max(min(r, n), bounds[-1]))
    bounds.append(n)
    return bounds


def even_row_bounds(n_rows, world):
    """Contiguous, near-equal bands (dense cross: reference rows; kNN: query rows)."""
    return [(n_rows * w) // world for w in range(world + 1)]


def self_pairs_before(n, r):
    """Number of condensed pairs in rows [0, r)."""
    r = min(r, n)
    if r >= n - 1:
        return n * (n - 1) // 2
    return r * n - r * (r + 1) // 2


def self_band_slices(n, world):
    """[(row0, row1, first_pair, n_pairs)] per rank."""
    b = self_row_bounds(n, world)
    out = []
    for w in range(world):
        p0 = self_pairs_before(n, b[w])
        p1 = self_pairs_before(n, b[w + 1])
        out.append((b[w], b[w + 1], p0, p1 - p0))
    return out


def gather_to_root(full, local, slices, rank, world, dist, root=0):
    """Assemble per-rank slices on `root`.

    full:   root's [total_pairs, ncols] tensor (ignored elsewhere); root's own band is
            expected to be computed directly into its slice of `full`.
    local:  this rank's [n_pairs_rank, ncols] tensor (non-root ranks).
    slices: self_band_slices()/cross equivalents: (.., .., first_pair, n_pairs) per rank.
    """
    if world == 1:
        return
    ops = []
    if rank == root:
        for w in range(world):
            if w == root or slices[w][3] == 0:
                continue
            p0, cnt = slices[w][2], slices[w][3]
            ops.append(dist.P2POp(dist.irecv, full[p0:p0 + cnt], w))
    elif slices[rank][3] > 0:
        ops.append(dist.P2POp(dist.isend, local, root))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()


class PipelinedGather:
    """The same gather, one step behind the compute.

    submit(local) starts the send / receives of a step and returns at once; before it returns
    it waits for the gather of `depth` steps ago, so a sender that rotates `depth` band buffers
    never overwrites one that is still being sent.  With "nccl" (RCCL) the transfers run on
    the communicator's stream and `wait()` is a stream dependency, not a host block: the next
    step's kernel overlaps the previous step's xGMI traffic.  drain() waits for everything.
    Root receives every step into the same `full` (its own band is computed in place); the
    receives of consecutive steps are ordered by the communicator.
    """

    def __init__(self, full, slices, rank, world, dist, root=0, depth=2, loopback=False):
        """loopback: the root also SENDS its own band to itself (a separate `local` buffer received
        into its slice of `full`) -- with one rank that is the whole exchange, which lets a 1-GPU box
        run the RCCL send/recv path of the N > 1 gather (tests/test_bench_gpu.py)."""
        import collections

        self.full, self.slices, self.rank, self.world, self.dist, self.root = full, slices, rank, world, dist, root
        self.depth = max(1, depth)
        self.pending = collections.deque()
        self.loopback = loopback

    def submit(self, local):
        if self.world == 1 and not self.loopback:
            return
        ops = []
        if self.rank == self.root and self.loopback and self.slices[self.root][3] > 0:
            p0, cnt = self.slices[self.root][2], self.slices[self.root][3]
            ops.append(self.dist.P2POp(self.dist.isend, local, self.root))
            ops.append(self.dist.P2POp(self.dist.irecv, self.full[p0:p0 + cnt], self.root))
        if self.rank == self.root:
            for w in range(self.world):
                if w == self.root or self.slices[w][3] == 0:
                    continue
                p0, cnt = self.slices[w][2], self.slices[w][3]
                ops.append(self.dist.P2POp(self.dist.irecv, self.full[p0:p0 + cnt], w))
        elif self.slices[self.rank][3] > 0:
            ops.append(self.dist.P2POp(self.dist.isend, local, self.root))
        self.pending.append(self.dist.batch_isend_irecv(ops) if ops else [])
        while len(self.pending) >= self.depth:
            for req in self.pending.popleft():
                req.wait()

    def drain(self):
        while self.pending:
            for req in self.pending.popleft():
                req.wait()


# ---------------------------------------------------------------------------
# self kNN, every pair once
# ---------------------------------------------------------------------------

def knn_band_deal(n_bands, world):
    """Band indices per rank, ascending.  Band b costs ~ (n_bands - b): deal them back and forth
    (0..W-1, W-1..0, ...) so that every rank gets the same cost to first order."""
    out = [[] for _ in range(world)]
    for b in range(n_bands):
        lap, pos = divmod(b, world)
        out[pos if lap % 2 == 0 else world - 1 - pos].append(b)
    return out


def exchange_knn_states(states, bounds, rank, world, dist):
    """All-to-all of row shards: `states` = this rank's [n, knn] tensors (some may be None);
    returns, for each, the [world, rows_of_this_rank, knn] stack of every rank's view of the rows
    bounds[rank]..bounds[rank+1]."""
    import torch

    rows = [bounds[w + 1] - bounds[w] for w in range(world)]
    out = []
    for t in states:
        if t is None:
            out.append(None)
            continue
        knn = t.shape[1]
        recv = torch.empty((world * rows[rank], knn), dtype=t.dtype, device=t.device)
        if dist is None:
            recv.copy_(t[bounds[0]:bounds[1]])
        else:   # (also with one rank: the collective then runs against itself)
            dist.all_to_all_single(recv, t.contiguous(), output_split_sizes=[rows[rank]] * world,
                                   input_split_sizes=rows)
        out.append(recv.view(world, rows[rank], knn))
    return out


def self_knn_once(ctx, sk, p, knn, rank, world, dist, device):
    """One-evaluation self kNN over `world` ranks: -> (row0, row1, idx, d0, d1) for this rank's
    row shard, as device tensors.  Every rank must call it."""
    import torch

    from . import capi

    n = sk.n
    band_rows = capi.knn_band_rows(sk, p, world)
    n_bands = (n + band_rows - 1) // band_rows
    mine = knn_band_deal(n_bands, world)[rank]
    coreacc = p.dist_type == capi.COREACC
    key = torch.empty((n, knn), dtype=torch.int32, device=device)
    idx = torch.empty((n, knn), dtype=torch.int32, device=device)
    d1 = torch.empty((n, knn), dtype=torch.float32, device=device) if coreacc else None
    capi.self_dists_knn_partial(ctx, sk, p, knn, band_rows, mine, out=(key, idx, d1))
    bounds = even_row_bounds(n, world)
    k_all, i_all, d_all = exchange_knn_states([key, idx, d1], bounds, rank, world, dist)
    rows = bounds[rank + 1] - bounds[rank]
    out = (torch.empty((rows, knn), dtype=torch.int64, device=device),
           torch.empty((rows, knn), dtype=torch.float32, device=device),
           torch.empty((rows, knn), dtype=torch.float32, device=device) if coreacc else None)
    if rows:
        capi.knn_merge_states(ctx, k_all, i_all, d_all, ani=bool(p.ani), out=out)
    return bounds[rank], bounds[rank + 1], out[0], out[1], out[2]
