"""ctypes front-end of the CPU oracle (oracle/sketchlib_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Product code under sketchlib.rust_amd/ must never
import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libsketchlib_oracle.so")

COREACC = 0
JACCARD = 1
TIES_RUST_HEAP = 0
TIES_CANONICAL = 1


def build(force=False):
    """Compile the oracle with gcc (seconds).  Building the checker is not using it."""
    src = [os.path.join(_HERE, f) for f in ("sketchlib_oracle.c", "sketchlib_oracle.h", "Makefile")]
    if not force and os.path.exists(_SO) and all(
        os.path.getmtime(_SO) >= os.path.getmtime(s) for s in src
    ):
        return _SO
    subprocess.check_call(["make", "-s", "-C", _HERE, "-B"])
    return _SO


class _Sketches(C.Structure):
    _fields_ = [
        ("bins", C.c_void_p),
        ("n_samples", C.c_size_t),
        ("nk", C.c_size_t),
        ("kmers", C.c_void_p),
        ("sketchsize64", C.c_uint64),
        ("completeness", C.c_void_p),
    ]


SPARSE_DTYPE = np.dtype([("idx", "<u8"), ("d0", "<f4"), ("d1", "<f4")])

_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.sko_samebits.restype = C.c_uint32
        L.sko_samebits.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.sko_jaccard_from_samebits.restype = C.c_double
        L.sko_jaccard_from_samebits.argtypes = [C.c_uint32, C.c_uint64, C.c_int, C.c_double,
                                                C.c_double, C.c_double]
        L.sko_ani_pois.restype = C.c_double
        L.sko_ani_pois.argtypes = [C.c_double, C.c_double]
        L.sko_completeness_correction.restype = C.c_double
        L.sko_completeness_correction.argtypes = [C.c_double] * 3
        L.sko_simple_linear_regression.restype = None
        L.sko_simple_linear_regression.argtypes = [C.c_double] * 6 + [C.c_void_p, C.c_void_p]
        for f in ("sko_square_to_condensed", "sko_calc_col_idx"):
            getattr(L, f).restype = C.c_size_t
            getattr(L, f).argtypes = [C.c_size_t] * 3
        L.sko_calc_row_idx.restype = C.c_size_t
        L.sko_calc_row_idx.argtypes = [C.c_size_t] * 2
        P = C.POINTER(_Sketches)
        L.sko_self_dists_all.restype = C.c_int
        L.sko_self_dists_all.argtypes = [P, C.c_int, C.c_size_t, C.c_int, C.c_double, C.c_int,
                                         C.c_void_p]
        L.sko_self_dists_all_repeat.restype = C.c_int
        L.sko_self_dists_all_repeat.argtypes = [P, C.c_int, C.c_size_t, C.c_int, C.c_double, C.c_int,
                                                C.c_int, C.c_void_p]
        L.sko_cross_dists_all.restype = C.c_int
        L.sko_cross_dists_all.argtypes = [P, P, C.c_int, C.c_size_t, C.c_int, C.c_double, C.c_int,
                                          C.c_void_p]
        L.sko_self_dists_knn.restype = C.c_int
        L.sko_self_dists_knn.argtypes = [P, C.c_size_t, C.c_int, C.c_size_t, C.c_int, C.c_double,
                                         C.c_int, C.c_int, C.c_void_p]
        L.sko_heap_replay.restype = C.c_size_t
        L.sko_heap_replay.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]
        L.sko_heap_feed.restype = None
        L.sko_heap_feed.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t]
        L.sko_heap_feed_logged.restype = None
        L.sko_heap_feed_logged.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]
        L.sko_heap_sorted.restype = None
        L.sko_heap_sorted.argtypes = [C.c_void_p, C.c_size_t]
        L.sko_cross_dists_knn.restype = C.c_long
        L.sko_cross_dists_knn.argtypes = [P, P, C.c_size_t, C.c_int, C.c_size_t, C.c_int,
                                          C.c_double, C.c_int, C.c_int, C.c_void_p]
        L.sko_self_dists_knn_precluster.restype = C.c_int
        L.sko_self_dists_knn_precluster.argtypes = [P, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                                    C.c_size_t, C.c_int, C.c_double, C.c_int, C.c_int,
                                                    C.c_int, C.c_void_p]
        L.sko_prefilter_pair_count.restype = C.c_uint64
        L.sko_prefilter_pair_count.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t]
        L.sko_core_acc_dist.restype = None
        L.sko_core_acc_dist.argtypes = [P, P, C.c_size_t, C.c_size_t, C.c_double, C.c_void_p, C.c_void_p]
        L.sko_self_binmatch.restype = C.c_int
        L.sko_self_binmatch.argtypes = [P, C.c_int, C.c_void_p]
        L.sko_cross_binmatch.restype = C.c_int
        L.sko_cross_binmatch.argtypes = [P, P, C.c_int, C.c_void_p]
        _lib = L
    return _lib


class Sketches:
    """Borrowed view of a MultiSketch's bins in the reference layout
    [sample][k][chunk][plane] (multisketch.rs:213-219)."""

    def __init__(self, bins, n, kmers, sketchsize64, completeness=None):
        self.bins = np.ascontiguousarray(bins, dtype="<u8").reshape(-1)
        self.kmers = np.ascontiguousarray(kmers, dtype=np.uintp)
        self.n = int(n)
        self.nk = len(self.kmers)
        self.ss64 = int(sketchsize64)
        assert self.bins.size == self.n * self.nk * self.ss64 * 14, "bins size mismatch"
        self.completeness = (
            None if completeness is None else np.ascontiguousarray(completeness, dtype=np.float64)
        )
        self._c = _Sketches(
            self.bins.ctypes.data, self.n, self.nk, self.kmers.ctypes.data, self.ss64,
            None if self.completeness is None else self.completeness.ctypes.data,
        )

    @property
    def ref(self):
        return C.byref(self._c)


def ncols(dist_type):
    return 2 if dist_type == COREACC else 1


def self_dists_all(s, dist_type=COREACC, k_idx=0, ani=False, cutoff=0.64, threads=1):
    n_pairs = s.n * (s.n - 1) // 2
    out = np.zeros(n_pairs * ncols(dist_type), dtype=np.float32)
    rc = lib().sko_self_dists_all(s.ref, dist_type, k_idx, int(ani), cutoff, threads,
                                  out.ctypes.data)
    if rc:
        raise ValueError(f"oracle self_dists_all failed rc={rc}")
    return out.reshape(n_pairs, ncols(dist_type))


def self_dists_all_repeat(s, repeat, dist_type=COREACC, k_idx=0, ani=False, cutoff=0.64, threads=1):
    """`repeat` passes inside one thread pool (timing helper for bench.py)."""
    n_pairs = s.n * (s.n - 1) // 2
    out = np.zeros(n_pairs * ncols(dist_type), dtype=np.float32)
    rc = lib().sko_self_dists_all_repeat(s.ref, dist_type, k_idx, int(ani), cutoff, threads, repeat,
                                         out.ctypes.data)
    if rc:
        raise ValueError(f"oracle self_dists_all_repeat failed rc={rc}")
    return out.reshape(n_pairs, ncols(dist_type))


def cross_dists_all(r, q, dist_type=COREACC, k_idx=0, ani=False, cutoff=0.64, threads=1):
    out = np.zeros(r.n * q.n * ncols(dist_type), dtype=np.float32)
    rc = lib().sko_cross_dists_all(r.ref, q.ref, dist_type, k_idx, int(ani), cutoff, threads,
                                   out.ctypes.data)
    if rc:
        raise ValueError(f"oracle cross_dists_all failed rc={rc}")
    return out.reshape(r.n, q.n, ncols(dist_type))


def self_dists_knn(s, knn, dist_type=COREACC, k_idx=0, ani=False, cutoff=0.64,
                   ties=TIES_CANONICAL, threads=1):
    out = np.zeros(s.n * knn, dtype=SPARSE_DTYPE)
    rc = lib().sko_self_dists_knn(s.ref, knn, dist_type, k_idx, int(ani), cutoff, ties, threads,
                                  out.ctypes.data)
    if rc:
        raise ValueError(f"oracle self_dists_knn failed rc={rc}")
    return out.reshape(s.n, knn)


def heap_replay(keys, knn, ids=None):
    """mod.rs:41-48 + into_sorted_vec over the candidates (ids[c], keys[c]) in the order given -> items (idx, d0, d1)."""
    keys = np.ascontiguousarray(keys, dtype=np.float32)
    ids = np.arange(keys.size, dtype=np.uint64) if ids is None else np.ascontiguousarray(ids, dtype=np.uint64)
    out = np.zeros(max(min(keys.size, knn), 1), dtype=SPARSE_DTYPE)
    m = lib().sko_heap_replay(ids.ctypes.data, keys.ctypes.data, keys.size, knn, out.ctypes.data)
    return out[:m]


class Heaps:
    """n resumable BinaryHeaps (items [n, knn + 1], lengths [n]): the state a row's heap is in between batches of candidates."""

    def __init__(self, n, knn):
        self.knn = knn
        self.items = np.zeros((n, knn + 1), dtype=SPARSE_DTYPE)
        self.len = np.zeros(n, dtype=np.uintp)

    def feed(self, row, ids, keys, d1=None, log=False):
        """push_heap over the candidates in the order given; log=True: -> bool array, which of them the heap took."""
        ids = np.ascontiguousarray(ids, dtype=np.uint64)
        keys = np.ascontiguousarray(keys, dtype=np.float32)
        d1 = None if d1 is None else np.ascontiguousarray(d1, dtype=np.float32)
        if log:
            took = np.zeros(ids.size, dtype=np.uint8)
            lib().sko_heap_feed_logged(self.items[row].ctypes.data, self.len[row:row + 1].ctypes.data, ids.ctypes.data, keys.ctypes.data,
                                       None if d1 is None else d1.ctypes.data, ids.size, self.knn, took.ctypes.data)
            return took.astype(bool)
        lib().sko_heap_feed(self.items[row].ctypes.data, self.len[row:row + 1].ctypes.data, ids.ctypes.data, keys.ctypes.data,
                            None if d1 is None else d1.ctypes.data, ids.size, self.knn)
        return None

    def sorted_rows(self):
        out = self.items.copy()
        for r in range(out.shape[0]):
            lib().sko_heap_sorted(out[r].ctypes.data, int(self.len[r]))
        return out[:, :self.knn]


def cross_dists_knn(r, q, knn, dist_type=COREACC, k_idx=0, ani=False, cutoff=0.64,
                    ties=TIES_CANONICAL, threads=1):
    knn_eff = min(knn, r.n)
    out = np.zeros(q.n * max(knn_eff, 1), dtype=SPARSE_DTYPE)
    rc = lib().sko_cross_dists_knn(r.ref, q.ref, knn, dist_type, k_idx, int(ani), cutoff, ties,
                                   threads, out.ctypes.data)
    if rc < 0:
        raise ValueError(f"oracle cross_dists_knn failed rc={rc}")
    return out.reshape(q.n, knn_eff)


RETAIN_NONE, RETAIN_SINGLETON, RETAIN_BRUTEFORCE = 0, 1, 2


def self_dists_knn_precluster(s, skq, knn, k_idx=0, ani=False, cutoff=0.64, ski_of_skd=None,
                              retain=RETAIN_NONE, ties=TIES_CANONICAL, threads=1):
    """mod.rs:399-553.  skq: [n, sketch_size] u16 bins in index order."""
    skq = np.ascontiguousarray(skq, dtype=np.uint16)
    assert skq.shape[0] == s.n
    lookup = np.arange(s.n, dtype=np.uintp) if ski_of_skd is None else np.ascontiguousarray(ski_of_skd, dtype=np.uintp)
    out = np.zeros(s.n * knn, dtype=SPARSE_DTYPE)
    rc = lib().sko_self_dists_knn_precluster(s.ref, skq.ctypes.data, skq.shape[1], lookup.ctypes.data, knn,
                                             k_idx, int(ani), cutoff, retain, ties, threads, out.ctypes.data)
    if rc:
        raise ValueError(f"oracle self_dists_knn_precluster failed rc={rc}")
    return out.reshape(s.n, knn)


def prefilter_pair_count(skq):
    skq = np.ascontiguousarray(skq, dtype=np.uint16)
    return int(lib().sko_prefilter_pair_count(skq.ctypes.data, skq.shape[0], skq.shape[1]))


def self_binmatch(s, threads=1):
    n_pairs = s.n * (s.n - 1) // 2
    out = np.zeros(n_pairs * s.nk, dtype=np.uint32)
    lib().sko_self_binmatch(s.ref, threads, out.ctypes.data)
    return out.reshape(n_pairs, s.nk)


def cross_binmatch(r, q, threads=1):
    out = np.zeros(r.n * q.n * r.nk, dtype=np.uint32)
    lib().sko_cross_binmatch(r.ref, q.ref, threads, out.ctypes.data)
    return out.reshape(r.n, q.n, r.nk)


def core_acc_pair(r, q, i, j, cutoff=0.64):
    """jaccard.rs:61-101 for one (ref i, query j) pair."""
    core = C.c_float()
    acc = C.c_float()
    lib().sko_core_acc_dist(r.ref, q.ref, i, j, cutoff, C.byref(core), C.byref(acc))
    return core.value, acc.value


def regression(xsum, ysum, xysum, xsq, ysq, n):
    core = C.c_float()
    acc = C.c_float()
    lib().sko_simple_linear_regression(xsum, ysum, xysum, xsq, ysq, n, C.byref(core), C.byref(acc))
    return core.value, acc.value
