"""Synthetic MultiSketch bins in the reference layout [sample][k][chunk][plane] u64
(the `.skd` byte order, src/sketch/multisketch.rs:213-219; bit-slicing rule of
Sketch::fill_usigs, src/sketch/mod.rs:215-223).

Two sets:
  * Set U ("random-bin"): every u64 word i.i.d. uniform.  P(bin match) = 2^-14, so
    J ~ 0 and every pair stops at the first k -- exercises the bit-match loop only.
  * Set R ("related"): samples come in clusters that share a parent sketch; each bin of
    a sample keeps the parent's 14-bit value with probability exp((alpha+beta*k)/2), so
    within a cluster E[J_k] ~ exp(alpha + beta*k) and the core/accessory regression has
    something to fit.

Both are counter-based (splitmix64 of a (seed, sample, word) counter), so any slice
can be generated independently -- on the host with numpy or on the device with torch --
and is identical everywhere.
"""
import numpy as np

BBITS = 14
SEED_U = 0x5EED0000
SEED_R = 0xC0DE

_M64 = (1 << 64) - 1
_GOLD = 0x9E3779B97F4A7C15
_C1 = 0xBF58476D1CE4E5B9
_C2 = 0x94D049BB133111EB


def _mix_np(z):
    """splitmix64 finaliser on a uint64 array (wrapping arithmetic)."""
    z = z.astype(np.uint64, copy=True)
    with np.errstate(over="ignore"):
        z ^= z >> np.uint64(30)
        z *= np.uint64(_C1)
        z ^= z >> np.uint64(27)
        z *= np.uint64(_C2)
        z ^= z >> np.uint64(31)
    return z


def _counter_np(seed, sample_ids, n_words):
    """state(seed, sample, w) = mix(seed + sample) + (w+1)*GOLD, then mixed again."""
    with np.errstate(over="ignore"):
        base = _mix_np(np.uint64(seed) + sample_ids.astype(np.uint64))
        w = (np.arange(1, n_words + 1, dtype=np.uint64) * np.uint64(_GOLD))
        return _mix_np(base[:, None] + w[None, :])


def set_u(n, nk, ss64, first_sample=0, seed=SEED_U):
    """[n, nk*ss64*14] uint64, uniform random words."""
    ids = np.arange(first_sample, first_sample + n, dtype=np.uint64)
    return _counter_np(seed, ids, nk * ss64 * BBITS)


def bitslice(values):
    """values [..., nbins] (14-bit ints) -> words [..., nbins/64, 14] uint64: bit (b % 64) of
    word (b/64, plane) is bit `plane` of values[b]  (fill_usigs)."""
    v = np.asarray(values, dtype=np.uint16)
    lead = v.shape[:-1]
    nb = v.shape[-1]
    assert nb % 64 == 0
    v = v.reshape(*lead, nb // 64, 64)
    planes = []
    for p in range(BBITS):
        bits = ((v >> p) & 1).astype(np.uint8)
        packed = np.packbits(bits, axis=-1, bitorder="little")  # [..., chunks, 8] bytes
        planes.append(packed.view("<u8")[..., 0])
    return np.stack(planes, axis=-1)


def set_r_params(n_clusters, kmers, seed=SEED_R):
    """Per-cluster (alpha, beta): core in [0.001, 0.05], accessory in [0.05, 0.5]."""
    rng = np.random.default_rng(seed)
    core = rng.uniform(0.001, 0.05, n_clusters)
    acc = rng.uniform(0.05, 0.5, n_clusters)
    beta = np.log1p(-core)
    alpha = np.log1p(-acc)
    return alpha, beta


def set_r(n, kmers, ss64, n_clusters=100, first_sample=0, seed=SEED_R):
    """[n, nk*ss64*14] uint64, clustered sketches (sample s belongs to cluster s % n_clusters)."""
    kmers = np.asarray(kmers, dtype=np.float64)
    nk = len(kmers)
    nb = ss64 * 64
    alpha, beta = set_r_params(n_clusters, kmers, seed)
    ids = np.arange(first_sample, first_sample + n, dtype=np.uint64)
    cl = (ids % np.uint64(n_clusters)).astype(np.int64)
    # parent bin values per cluster
    parent = _counter_np(seed ^ 0xA5A5, np.arange(n_clusters, dtype=np.uint64), nk * nb)
    parent = (parent & np.uint64(0x3FFF)).astype(np.uint16).reshape(n_clusters, nk, nb)
    # per-sample draws: low 14 bits = own value, high bits = keep/redraw coin
    r = _counter_np(seed ^ 0x5A5A, ids, nk * nb).reshape(n, nk, nb)
    own = (r & np.uint64(0x3FFF)).astype(np.uint16)
    coin = (r >> np.uint64(40)).astype(np.float64) / float(1 << 24)
    keep_p = np.exp((alpha[cl][:, None] + beta[cl][:, None] * kmers[None, :]) / 2.0)  # [n, nk]
    vals = np.where(coin < keep_p[:, :, None], parent[cl], own)
    return bitslice(vals).reshape(n, nk * ss64 * BBITS)


# ---------------------------------------------------------------------------
# device-side Set U for large benchmark slabs (same values as set_u)
# ---------------------------------------------------------------------------

def _mix_t(z, torch):
    def lsr(x, s):  # logical shift right on int64
        return (x >> s) & ((1 << (64 - s)) - 1)

    def c(v):  # two's-complement int64 constant
        return v - (1 << 64) if v >= (1 << 63) else v

    z = z ^ lsr(z, 30)
    z = z * c(_C1)
    z = z ^ lsr(z, 27)
    z = z * c(_C2)
    z = z ^ lsr(z, 31)
    return z


def set_u_device(n, nk, ss64, device, first_sample=0, seed=SEED_U, chunk=4096):
    """Same words as set_u(), generated on `device`; returns an int64 tensor [n, words]
    whose bit pattern is the uint64 slab."""
    import torch

    words = nk * ss64 * BBITS
    out = torch.empty((n, words), dtype=torch.int64, device=device)
    gold = _GOLD - (1 << 64)
    w = torch.arange(1, words + 1, dtype=torch.int64, device=device) * gold
    for s0 in range(0, n, chunk):
        s1 = min(n, s0 + chunk)
        ids = torch.arange(first_sample + s0, first_sample + s1, dtype=torch.int64, device=device)
        base = _mix_t(ids + seed, torch)
        out[s0:s1] = _mix_t(base[:, None] + w[None, :], torch)
    return out


def set_clustered_device(n, nk, ss64, device, cluster_size=200, keep=0.9, seed=SEED_R, chunk=2048,
                         first_sample=0, n_clusters=None, scatter=False):
    """Clustered sketches generated on `device` for large runs: sample s belongs to cluster
    s % n_clusters (default n / cluster_size) and keeps each of its cluster's bin values with
    probability `keep` -- one number, or one per k-mer length so that J falls with k and the
    core/accessory regression has a slope to fit -- and an independent 14-bit value otherwise, so a
    row has ~cluster_size close neighbours scattered over the whole id range and every other
    distance sits at ~1.  `first_sample` offsets the ids the samples' own draws come from (a query
    set drawn from the same clusters as a reference set).  `scatter`: the cluster of sample s is a hash of s (mod n_clusters)
    instead -- relatives at RANDOM ids rather than at regular id distances (s % n_clusters puts a tile's related pairs on a
    diagonal: either every row of a 32 x 128 tile has one or none has; scattered, a tile that holds a relative usually holds
    exactly one).  Returns an int64 tensor [n, words]
    whose bit pattern is the uint64 slab (fill_usigs layout, src/sketch/mod.rs:215-223)."""
    import torch

    nb = ss64 * 64
    words = nk * ss64 * BBITS
    if n_clusters is None:
        n_clusters = max(1, n // cluster_size)
    out = torch.empty((n, words), dtype=torch.int64, device=device)
    gold = _GOLD - (1 << 64)
    w = torch.arange(1, nk * nb + 1, dtype=torch.int64, device=device) * gold
    bit = torch.arange(64, dtype=torch.int64, device=device)
    keeps = [keep] * nk if np.isscalar(keep) else list(keep)
    assert len(keeps) == nk
    thresh = torch.tensor([int(k * 65536) for k in keeps], dtype=torch.int64, device=device).view(1, nk, 1, 1)
    for s0 in range(0, n, chunk):
        s1 = min(n, s0 + chunk)
        ids = torch.arange(first_sample + s0, first_sample + s1, dtype=torch.int64, device=device)
        own = _mix_t(_mix_t(ids + seed, torch)[:, None] + w[None, :], torch)
        cl = ((_mix_t(ids + (seed ^ 0x5C5C), torch) >> 1) & 0x3FFFFFFFFFFF) % n_clusters if scatter else ids % n_clusters
        par = _mix_t(_mix_t(cl + (seed ^ 0xA5A5), torch)[:, None] + w[None, :], torch)
        coin = ((own >> 14) & 0xFFFF).view(s1 - s0, nk, ss64, 64)
        vals = torch.where(coin < thresh, (par & 0x3FFF).view(s1 - s0, nk, ss64, 64),
                           (own & 0x3FFF).view(s1 - s0, nk, ss64, 64))
        # 64 disjoint bits -> one word: an OR tree of element-wise kernels (not .sum(): torch's reduce kernels
        # crash rocprofv3's counter collection, and the profiling recipes generate their data with this)
        def fold(x):
            while x.shape[-1] > 1:
                h = x.shape[-1] // 2
                x = x[..., :h] | x[..., h:]
            return x[..., 0]

        planes = [fold(((vals >> p) & 1) << bit) for p in range(BBITS)]
        out[s0:s1] = torch.stack(planes, dim=-1).view(s1 - s0, words)
    return out
