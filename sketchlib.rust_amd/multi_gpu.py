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
That merge is defined for the canonical tie rule.  In the REFERENCE's tie order a row's BinaryHeap must meet its
candidates in ascending id, so heaps travel instead: rank r owns a window of COLUMNS, feeds every row band the columns
of its window, and hands each band's heaps to rank r + 1 -- a pipeline one band deep per rank, every pair still evaluated
once (knn_window_cuts / self_knn_once_reference; include/sketchlib_dist.h, skl_self_dists_knn_window).
"""
import os

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


MAX_MESSAGE_ELEMS = 1 << 27   # rows ((core, acc) records: 8 B) per message: 1 GiB


def message_chunks(cnt, max_elems=MAX_MESSAGE_ELEMS):
    """[(first row, rows)] of a band of `cnt` rows cut into messages of at most `max_elems` rows.  Sender and
    receiver cut the same band the same way, so the messages pair up in order.  (A band of cfg 3 is 5 GB per
    rank at N = 8 and 20 GB at N = 2: as ONE message that is a 2.5e9-element send, past anything the transport's
    32-bit counts have been exercised with; 1 GiB pieces also let the root's receives from different peers
    interleave.)"""
    max_elems = max(1, int(max_elems))
    return [(a, min(max_elems, cnt - a)) for a in range(0, cnt, max_elems)]


def gather_to_root(full, local, slices, rank, world, dist, root=0, max_elems=MAX_MESSAGE_ELEMS):
    """Assemble per-rank slices on `root`.

    full:   root's [total_pairs, ncols] tensor (ignored elsewhere); root's own band is
            expected to be computed directly into its slice of `full`.
    local:  this rank's [n_pairs_rank, ncols] tensor (non-root ranks).
    slices: self_band_slices()/cross equivalents: (.., .., first_pair, n_pairs) per rank.
    Bands travel in messages of at most max_elems rows (message_chunks).
    """
    if world == 1:
        return
    ops = []
    if rank == root:
        for w in range(world):
            if w == root or slices[w][3] == 0:
                continue
            p0, cnt = slices[w][2], slices[w][3]
            for a, m in message_chunks(cnt, max_elems):
                ops.append(dist.P2POp(dist.irecv, full[p0 + a:p0 + a + m], w))
    elif slices[rank][3] > 0:
        for a, m in message_chunks(slices[rank][3], max_elems):
            ops.append(dist.P2POp(dist.isend, local[a:a + m], root))
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

    def __init__(self, full, slices, rank, world, dist, root=0, depth=2, loopback=False, max_elems=MAX_MESSAGE_ELEMS):
        """loopback: the root also SENDS its own band to itself (a separate `local` buffer received
        into its slice of `full`) -- with one rank that is the whole exchange, which lets a 1-GPU box
        run the RCCL send/recv path of the N > 1 gather (tests/test_bench_gpu.py)."""
        import collections

        self.full, self.slices, self.rank, self.world, self.dist, self.root = full, slices, rank, world, dist, root
        self.depth = max(1, depth)
        self.pending = collections.deque()
        self.loopback = loopback
        self.max_elems = max_elems

    def submit(self, local):
        if self.world == 1 and not self.loopback:
            return
        ops = []
        P2P, isend, irecv = self.dist.P2POp, self.dist.isend, self.dist.irecv
        if self.rank == self.root and self.loopback and self.slices[self.root][3] > 0:
            p0, cnt = self.slices[self.root][2], self.slices[self.root][3]
            for a, m in message_chunks(cnt, self.max_elems):
                ops.append(P2P(isend, local[a:a + m], self.root))
                ops.append(P2P(irecv, self.full[p0 + a:p0 + a + m], self.root))
        if self.rank == self.root:
            for w in range(self.world):
                if w == self.root or self.slices[w][3] == 0:
                    continue
                p0, cnt = self.slices[w][2], self.slices[w][3]
                for a, m in message_chunks(cnt, self.max_elems):
                    ops.append(P2P(irecv, self.full[p0 + a:p0 + a + m], w))
        elif self.slices[self.rank][3] > 0:
            for a, m in message_chunks(self.slices[self.rank][3], self.max_elems):
                ops.append(P2P(isend, local[a:a + m], self.root))
        self.pending.append(self.dist.batch_isend_irecv(ops) if ops else [])
        while len(self.pending) >= self.depth:
            for req in self.pending.popleft():
                req.wait()

    def drain(self):
        while self.pending:
            for req in self.pending.popleft():
                req.wait()


class HostGather:
    """The alternative SURVEY 8(e) names for when the host needs the matrix anyway: no transfer between
    GPUs at all -- every rank copies its band device-to-host straight into ITS offsets of one host buffer
    shared by the ranks of the node (a file in /dev/shm mapped by every rank, the rank's own slice
    registered as pinned memory so the copy is one DMA).  The copy of step i runs on a side stream while
    the kernel of step i + 1 runs; the caller rotates two band buffers, and submit() waits for the copy
    that last read the buffer it is handed.  Root ingress over xGMI (35 GB per step into one GPU at cfg 3,
    N = 8) is replaced by 8 independent PCIe streams of 5 GB."""

    def __init__(self, total_rows, ncols, slices, rank, world, dist, tag="0", device=None, directory="/dev/shm"):
        import torch

        self.torch, self.rank, self.world, self.dist = torch, rank, world, dist
        if total_rows <= 0:
            raise ValueError("HostGather: nothing to gather (total_rows == 0)")
        # One file per run, under a random name chosen by rank 0 and created there with O_EXCL | O_NOFOLLOW (never an
        # existing file or a symlink somebody left at a guessable path), mode 0600; the name travels to the other ranks
        # by broadcast.  The ranks must share the node: a /dev/shm file exists on rank 0's node only.
        name = [None]
        if rank == 0:
            import secrets

            name[0] = f"{directory}/skl_bench_gather_{tag}_{secrets.token_hex(8)}.f32"
            fd = os.open(name[0], os.O_CREAT | os.O_EXCL | os.O_NOFOLLOW | os.O_RDWR, 0o600)
            try:
                os.ftruncate(fd, max(total_rows * ncols * 4, 4))
            finally:
                os.close(fd)
        if dist is not None and world > 1:
            dist.broadcast_object_list(name, src=0)
        self.path = name[0]
        self.shape = (total_rows, ncols)
        if dist is not None:
            dist.barrier()
        if not os.path.exists(self.path):
            raise RuntimeError(f"HostGather: rank {rank} does not see {self.path}: the ranks of a host gather must share one node")
        self.map = np.memmap(self.path, dtype=np.float32, mode="r+", shape=self.shape)
        p0, cnt = slices[rank][2], slices[rank][3]
        self.mine = torch.from_numpy(self.map[p0:p0 + cnt])
        self.pinned = False
        self.on_gpu = device is not None and torch.device(device).type == "cuda"
        if self.on_gpu and cnt:
            # page-locked for the lifetime of the object: the D2H copy is then one asynchronous DMA
            rc = torch.cuda.cudart().cudaHostRegister(self.mine.data_ptr(), self.mine.numel() * 4, 0)
            self.pinned = int(rc) == 0
            if not self.pinned:   # the copy still works (staged, synchronous) but the advertised overlap is gone: say so
                import sys

                print(f"[HostGather] rank {rank}: cudaHostRegister failed (rc {int(rc)}): device-to-host copies will be staged, "
                      "not overlapped", file=sys.stderr)
            self.stream = torch.cuda.Stream(device=device)
        self.last = None     # event of the newest copy; self.before: the one before it

    def submit(self, local):
        """Start the copy of `local` (just filled on the current stream) into this rank's slice.  The caller
        rotates TWO band buffers: the next kernel writes the buffer of the previous submit, so the current
        stream is made to wait for THAT copy; the copy started here overlaps the next kernel."""
        torch = self.torch
        if local.shape[0] == 0:
            return
        if not self.on_gpu:
            self.mine.copy_(local)
            return
        cur = torch.cuda.current_stream(local.device)
        ready = torch.cuda.Event()
        ready.record(cur)                         # the kernel that filled `local`
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ready)
            self.mine.copy_(local, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        if self.last is not None:
            cur.wait_event(self.last)             # the other buffer's copy, before the next kernel overwrites it
        self.last = ev

    def drain(self):
        if self.on_gpu and self.last is not None:
            self.last.synchronize()               # (copies are ordered on one stream)

    def assembled(self):
        """The whole matrix as a host tensor (after drain() + a barrier every rank's band is in it)."""
        return self.torch.from_numpy(self.map)

    def close(self):
        self.drain()
        if self.pinned:
            self.torch.cuda.cudart().cudaHostUnregister(self.mine.data_ptr())
            self.pinned = False
        if self.dist is not None:
            self.dist.barrier()
        del self.mine
        del self.map
        if self.rank == 0:
            try:
                os.unlink(self.path)
            except OSError:
                pass


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


# ---------------------------------------------------------------------------
# self kNN, every pair once, in the reference's tie order: travelling heaps
# ---------------------------------------------------------------------------

def knn_window_cuts(n, band_rows, world):
    """Column windows [cuts[r], cuts[r + 1]) per rank: cut where the pair counts balance -- rank r evaluates the pairs (i, j),
    i < j, whose column j lies in its window, ~ (hi^2 - lo^2) / 2 of them -- on band boundaries (a band's rows are then
    either all inside a window or all outside it)."""
    cuts = [0]
    for r in range(1, world):
        c = int(round(n * (r / world) ** 0.5 / band_rows)) * band_rows
        cuts.append(min(max(c, cuts[-1]), n))
    cuts.append(n)
    return cuts


def _pack_heaps(heaps, r0, r1):
    import torch

    parts = [heaps["h_key"][r0:r1].view(torch.int32), heaps["h_id"][r0:r1]]
    if heaps["h_d1"] is not None:
        parts.append(heaps["h_d1"][r0:r1].view(torch.int32))
    parts += [heaps["h_len"][r0:r1, None], heaps["thr"][r0:r1, None]]
    return torch.cat(parts, dim=1).contiguous()


def _unpack_heaps(heaps, r0, r1, packed):
    import torch

    knn = heaps["h_key"].shape[1]
    heaps["h_key"][r0:r1] = packed[:, :knn].view(torch.float32)
    heaps["h_id"][r0:r1] = packed[:, knn:2 * knn]
    o = 2 * knn
    if heaps["h_d1"] is not None:
        heaps["h_d1"][r0:r1] = packed[:, o:o + knn].view(torch.float32)
        o += knn
    heaps["h_len"][r0:r1] = packed[:, o]
    heaps["thr"][r0:r1] = packed[:, o + 1]


def self_knn_once_reference(ctx, sk, p, knn, rank, world, dist, device, band_rows=None, stage=None, finalize=None, heaps=None,
                            host_staged=False):
    """Self kNN in the reference's tie order over `world` ranks, every pair evaluated once: -> (row0, row1, idx, d0, d1) for
    this rank's row shard (device tensors; even_row_bounds), like self_knn_once.  Every rank must call it.

    `stage(band, lo, hi, heaps)` feeds one row band the columns [lo, hi) (default: skl_self_dists_knn_window) and
    `finalize(heaps, r0, r1)` sorts heaps into lists (default: skl_knn_heaps_finalize); the CPU tests of the protocol pass
    oracle-backed ones.  host_staged: messages go through host memory (the gloo debugging backend, which lets several ranks
    share one GPU; RCCL moves device tensors)."""
    import torch

    from . import capi

    n = sk.n
    coreacc = p.dist_type == capi.COREACC
    if band_rows is None:
        band_rows = capi.knn_band_rows(sk, p, world)
    n_bands = (n + band_rows - 1) // band_rows
    lo, hi = knn_window_cuts(n, band_rows, world)[rank:rank + 2]
    if heaps is None:
        heaps = capi.knn_heaps_alloc(n, knn, coreacc, device)
    if stage is None:
        def stage(band, lo_, hi_, h):
            capi.self_dists_knn_window(ctx, sk, p, knn, band_rows, band, lo_, hi_, h)
    if finalize is None:
        def finalize(h, r0, r1):
            return capi.knn_heaps_finalize(ctx, h, r0, r1, knn, ani=bool(p.ani))
    width = (3 if coreacc else 2) * knn + 2
    comm = torch.device("cpu") if host_staged else device
    sends = []
    for band in range(n_bands):
        b0, b1 = band * band_rows, min(n, (band + 1) * band_rows)
        if b0 >= hi:
            break                       # the bands from here on hold no pair of this window
        if b0 < lo:                     # rows of an earlier window: their heaps come from the rank that just finished them
            packed = torch.empty((b1 - b0, width), dtype=torch.int32, device=comm)
            dist.recv(packed, src=rank - 1)
            _unpack_heaps(heaps, b0, b1, packed.to(device))
        stage(band, lo, hi, heaps)
        if rank + 1 < world:            # done with this window: on to the next one
            packed = _pack_heaps(heaps, b0, b1).to(comm)
            sends.append((dist.isend(packed, dst=rank + 1), packed))
    for work, _buf in sends:
        work.wait()
    # every heap has ended on the last rank: lists there, row shards back to their ranks
    bounds = even_row_bounds(n, world)
    r0, r1 = bounds[rank], bounds[rank + 1]
    if world == 1:
        idx, d0, d1 = finalize(heaps, 0, n)
        return 0, n, idx, d0, d1
    if rank == world - 1:
        idx, d0, d1 = finalize(heaps, 0, n)
        out = []
        for r in range(world - 1):
            a, b = bounds[r], bounds[r + 1]
            for t in (idx, d0, d1):
                if t is not None and b > a:
                    shard = t[a:b].contiguous().to(comm)
                    out.append((dist.isend(shard, dst=r), shard))
        for work, _buf in out:
            work.wait()
        return r0, r1, idx[r0:r1], d0[r0:r1], (d1[r0:r1] if d1 is not None else None)
    idx = torch.empty((r1 - r0, knn), dtype=torch.int64, device=comm)
    d0 = torch.empty((r1 - r0, knn), dtype=torch.float32, device=comm)
    d1 = torch.empty((r1 - r0, knn), dtype=torch.float32, device=comm) if coreacc else None
    for t in (idx, d0, d1):
        if t is not None and r1 > r0:
            dist.recv(t, src=world - 1)
    return r0, r1, idx.to(device), d0.to(device), (d1.to(device) if d1 is not None else None)


def self_knn_once_reference_decoupled(ctx, sk, p, knn, rank, world, dist, device, band_rows=None, stage=None, replay=None, finalize=None,
                                      heaps=None, logs=None, log_cap=None, host_staged=False):
    """Self kNN in the reference's tie order over `world` ranks, every pair evaluated once, NO RANK WAITING FOR ANOTHER:
    -> (row0, row1, idx, d0, d1) for this rank's row shard (device tensors; even_row_bounds), or None when some row's accept log
    overflowed (then call self_knn_once_reference, the travelling heaps).  Every rank must call it.

    A BinaryHeap that starts EMPTY on a column window takes a superset of what the row's true heap takes there (its maximum is
    never lower; push_heap pushes on key < maximum, mod.rs:41-48).  So rank r runs ITS window [lo_r, hi_r) -- every row band
    that starts below hi_r, the same calls as the travelling form -- against heaps it has cleared itself, logging per row what
    the heaps take, in order (skl_self_dists_knn_window_logged); row x's true list is the replay of its logs in rank order into
    one empty heap (skl_knn_heaps_replay) on the rank whose shard holds x.  Ranks exchange logs, not heaps: one all-to-all of
    (lengths, records) per pair of ranks after the windows, nothing during them.

    `stage(band, lo, hi, heaps, logs)`, `replay(heaps, r0, r1, rec, ids, lens)` and `finalize(heaps, r0, r1)` default to the
    library's calls; the CPU tests of the protocol pass oracle-backed ones.  host_staged: messages through host memory (gloo)."""
    import torch

    from . import capi

    n = sk.n
    coreacc = p.dist_type == capi.COREACC
    if band_rows is None:
        band_rows = capi.knn_band_rows(sk, p, world)
    n_bands = (n + band_rows - 1) // band_rows
    cuts = knn_window_cuts(n, band_rows, world)
    lo, hi = cuts[rank:rank + 2]
    if log_cap is None:
        log_cap = max(64, 16 * knn)
    if heaps is None:
        heaps = capi.knn_heaps_alloc(n, knn, coreacc, device)
    if logs is None:
        logs = capi.knn_logs_alloc(n, log_cap, coreacc, device)
    if stage is None:
        def stage(band, lo_, hi_, h, lg):
            capi.self_dists_knn_window_logged(ctx, sk, p, knn, band_rows, band, lo_, hi_, h, lg)
    if replay is None:
        def replay(h, r0, r1, rec, ids, lens):
            capi.knn_heaps_replay(ctx, h, r0, r1, knn, rec, ids, lens)
    if finalize is None:
        def finalize(h, r0, r1):
            return capi.knn_heaps_finalize(ctx, h, r0, r1, knn, ani=bool(p.ani))
    comm = torch.device("cpu") if host_staged else device
    # 1. this rank's window against its own empty heaps, every band that holds a pair of it
    for band in range(n_bands):
        if band * band_rows >= hi:
            break
        stage(band, lo, hi, heaps, logs)
    # 2. did every log hold?  (one number per rank; an overflow anywhere sends everybody to the travelling form)
    worst = torch.tensor([int(logs["len"][:hi].max()) if hi > 0 else 0], dtype=torch.int64, device=comm)
    if world > 1:
        dist.all_reduce(worst, op=dist.ReduceOp.MAX)
    if int(worst.item()) > log_cap:
        return None
    # 3. logs to the ranks that finalise the rows: rank s holds candidates of the rows below hi_s
    bounds = even_row_bounds(n, world)
    r0, r1 = bounds[rank], bounds[rank + 1]
    width = 2 if coreacc else 1
    sends, mine = [], None
    for d in range(world):
        a, b = bounds[d], min(bounds[d + 1], hi)
        if b <= a:
            continue
        lens = logs["len"][a:b].to(torch.int32)
        m = max(1, int(lens.max()))
        rec = logs["rec"][a:b, :m].contiguous()
        ids = logs["id"][a:b, :m].contiguous()
        if d == rank:
            mine = (a, b, lens, rec, ids)
            continue
        head = torch.tensor([m], dtype=torch.int32, device=comm)
        for t in (head, lens.to(comm), rec.to(comm), ids.to(comm)):
            sends.append((dist.isend(t, dst=d), t))
    # 4. replay in rank (= window) order into empty heaps of this rank's rows
    final = {k: (None if v is None else torch.zeros_like(v[r0:r1])) for k, v in heaps.items()}
    final["thr"] = torch.full_like(heaps["thr"][r0:r1], -1)
    for s_rank in range(world):
        a, b = r0, min(r1, cuts[s_rank + 1])
        if b <= a:
            continue
        if s_rank == rank:
            _a, _b, lens, rec, ids = mine
        else:
            head = torch.empty((1,), dtype=torch.int32, device=comm)
            dist.recv(head, src=s_rank)
            m = int(head.item())
            lens = torch.empty((b - a,), dtype=torch.int32, device=comm)
            rec = torch.empty((b - a, m, width), dtype=torch.float32, device=comm)
            ids = torch.empty((b - a, m), dtype=torch.int32, device=comm)
            for t in (lens, rec, ids):
                dist.recv(t, src=s_rank)
            lens, rec, ids = lens.to(device), rec.to(device), ids.to(device)
        replay(final, 0, b - a, rec, ids, lens)
    for work, _buf in sends:
        work.wait()
    idx, d0, d1 = finalize(final, 0, r1 - r0)
    return r0, r1, idx, d0, d1
