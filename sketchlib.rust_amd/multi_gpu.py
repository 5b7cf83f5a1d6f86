"""Multi-GPU partition of the pair space: one process per GPU (torch.distributed, backend
"nccl" == RCCL over xGMI), every rank holding the whole sketch slab.

The path shards by independent row bands:
  * dense self: rows of the condensed upper triangle, balanced by pair count -- each
    band is one contiguous slice of the reference's condensed output array
    (distance_matrix.rs:11-14);
  * dense cross: contiguous bands of reference rows (output index i_ref*n_query + j);
  * kNN: contiguous bands of query rows.
There is no reduction anywhere; the only exchange is assembling the output on rank 0,
done as grouped point-to-point send/recv straight into the final offsets (what
ncclGather does internally, but with per-rank counts and no staging copy).
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

    def __init__(self, full, slices, rank, world, dist, root=0, depth=2):
        import collections

        self.full, self.slices, self.rank, self.world, self.dist, self.root = full, slices, rank, world, dist, root
        self.depth = max(1, depth)
        self.pending = collections.deque()

    def submit(self, local):
        if self.world == 1:
            return
        ops = []
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
