"""ctypes binding of include/sketchlib_dist.h.

Used by tests/ and bench.py (and as the model for any other FFI: the Rust binding in
INTEGRATION.md is the same call sequence).  Host buffers are numpy arrays; device
buffers are anything with a `data_ptr()` (torch tensors).  Loading fails loudly when
the in-tree HIP library is missing -- there is no fallback implementation.
"""
import ctypes as C
import os

import numpy as np

from .build import library_path

COREACC = 0
JACCARD = 1

OK = 0
ERR_INVALID_ARG = 1
ERR_NO_DEVICE = 2
ERR_HIP = 3
ERR_OOM = 4
ERR_KMER_COUNT = 5
ERR_EMPTY_DB = 6
ERR_KMER_NOT_FOUND = 7
ERR_INCOMPATIBLE = 8


class SklError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"[skl error {code}] {message}")
        self.code = code
        self.message = message


class DistParams(C.Structure):
    _fields_ = [
        ("dist_type", C.c_int32),
        ("ani", C.c_int32),
        ("k_idx", C.c_uint64),
        ("completeness_cutoff", C.c_double),
    ]


# every symbol include/sketchlib_dist.h declares: (name, restype, argtypes)
_P = C.c_void_p
_SIG = [
    ("skl_last_error", C.c_char_p, []),
    ("skl_abi_version", C.c_int, []),
    ("skl_device_count", C.c_int, []),
    ("skl_ctx_create", C.c_int, [C.c_int, C.POINTER(_P)]),
    ("skl_ctx_destroy", C.c_int, [_P]),
    ("skl_ctx_set_stream", C.c_int, [_P, _P]),
    ("skl_ctx_use_default_stream", C.c_int, [_P]),
    ("skl_ctx_synchronize", C.c_int, [_P]),
    ("skl_ctx_reload_env", C.c_int, [_P]),
    ("skl_ctx_timing_enable", C.c_int, [_P, C.c_int]),
    ("skl_ctx_timing_reset", C.c_int, [_P]),
    ("skl_ctx_kernel_ms", C.c_int, [_P, C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    ("skl_ctx_last_kernel", C.c_char_p, [_P]),
    ("skl_log_variant", C.c_int, []),
    ("skl_ctx_flags", C.c_uint, [_P]),
    ("skl_ctx_set_knn_ties", C.c_int, [_P, C.c_int]),
    ("skl_self_dists_knn_window", C.c_int, [_P, _P, _P, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, _P, _P, _P, _P, _P]),
    ("skl_knn_heaps_clear", C.c_int, [_P, C.c_size_t, C.c_size_t, _P, _P]),
    ("skl_self_dists_knn_window_logged", C.c_int, [_P, _P, _P, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, _P, _P, _P, _P, _P,
                                                   _P, _P, _P, C.c_size_t]),
    ("skl_gather_bands_rccl", C.c_int, [_P, C.c_size_t, _P, _P, _P, _P, C.c_int]),
    ("skl_knn_heaps_replay", C.c_int, [_P, C.c_size_t, C.c_size_t, C.c_int, _P, _P, _P, C.c_size_t, _P, _P, _P, _P, _P]),
    ("skl_knn_heaps_finalize", C.c_int, [_P, C.c_size_t, C.c_size_t, _P, _P, _P, _P, C.c_int, _P, _P, _P]),
    ("skl_device_malloc", C.c_int, [_P, C.c_size_t, C.POINTER(_P)]),
    ("skl_device_free", C.c_int, [_P, _P]),
    ("skl_device_memcpy", C.c_int, [_P, _P, _P, C.c_size_t, C.c_int]),
    ("skl_ctx_get_knn_ties", C.c_int, [_P]),
    ("skl_ctx_early_break_stats", C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("skl_ctx_early_break_blocks", C.c_int, [_P, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                             C.POINTER(C.c_int), C.POINTER(C.c_int), _P, C.c_size_t]),
    ("skl_ctx_knn_prune_stats", C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                          C.POINTER(C.c_uint64)]),
    ("skl_clock_sampler_start", C.c_int, [_P, C.c_uint32, C.c_uint32]),
    ("skl_clock_sampler_stop", C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                         C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    ("skl_device_log", C.c_int, [_P, _P, C.c_size_t, _P]),
    ("skl_sketches_create", C.c_int, [_P, _P, C.c_int, C.c_size_t, C.c_size_t, _P, C.c_size_t,
                                      C.POINTER(_P)]),
    ("skl_sketches_set_completeness", C.c_int, [_P, _P]),
    ("skl_sketches_destroy", C.c_int, [_P]),
    ("skl_sketches_n_samples", C.c_size_t, [_P]),
    ("skl_set_k", C.c_int, [_P, C.c_size_t, C.c_int, C.c_double, C.POINTER(DistParams)]),
    ("skl_self_dists_all", C.c_int, [_P, _P, C.POINTER(DistParams), _P, C.c_int]),
    ("skl_self_dists_rows", C.c_int, [_P, _P, C.POINTER(DistParams), C.c_size_t, C.c_size_t, _P,
                                      C.c_int]),
    ("skl_cross_dists_all", C.c_int, [_P, _P, _P, C.POINTER(DistParams), _P, C.c_int]),
    ("skl_cross_dists_rows", C.c_int, [_P, _P, _P, C.POINTER(DistParams), C.c_size_t, C.c_size_t,
                                       _P, C.c_int]),
    ("skl_self_dists_knn", C.c_int, [_P, _P, C.POINTER(DistParams), C.c_size_t, _P, _P, _P,
                                     C.c_int]),
    ("skl_self_dists_knn_rows", C.c_int, [_P, _P, C.POINTER(DistParams), C.c_size_t, C.c_size_t,
                                          C.c_size_t, _P, _P, _P, C.c_int]),
    ("skl_cross_dists_knn", C.c_int, [_P, _P, _P, C.POINTER(DistParams), C.c_size_t, _P, _P, _P,
                                      C.c_int]),
    ("skl_cross_dists_knn_rows", C.c_int, [_P, _P, _P, C.POINTER(DistParams), C.c_size_t,
                                           C.c_size_t, C.c_size_t, _P, _P, _P, C.c_int]),
    ("skl_self_dists_knn_candidates", C.c_int, [_P, _P, C.POINTER(DistParams), C.c_size_t, _P, _P, _P, _P]),
    ("skl_shared_bins_max_samples", C.c_size_t, []),
    ("skl_self_dists_knn_shared_bins", C.c_int, [_P, _P, C.POINTER(DistParams), C.c_size_t, _P, C.c_size_t, _P, _P, _P]),
    ("skl_knn_band_rows", C.c_size_t, [_P, C.POINTER(DistParams), C.c_size_t]),
    ("skl_self_dists_knn_partial", C.c_int, [_P, _P, C.POINTER(DistParams), C.c_size_t, C.c_size_t, _P, C.c_size_t,
                                             _P, _P, _P, C.c_int]),
    ("skl_knn_merge_states", C.c_int, [_P, C.c_size_t, C.c_size_t, C.c_size_t, _P, _P, _P, C.c_int, C.c_int,
                                       _P, _P, _P, C.c_int]),
    ("skl_sketch_signs", C.c_int, [_P, _P, _P, _P, _P, C.c_size_t, _P, C.c_size_t, C.c_uint64, C.c_int, _P]),
    ("skl_sketch_signs_packed", C.c_int, [_P, _P, _P, _P, _P, C.c_size_t, _P, C.c_size_t, C.c_uint64, C.c_int, _P]),
    ("skl_self_binmatch", C.c_int, [_P, _P, _P, C.c_int]),
    ("skl_cross_binmatch", C.c_int, [_P, _P, _P, _P, C.c_int]),
    ("skl_self_dists_all_host", C.c_int, [_P, C.c_size_t, C.c_size_t, _P, C.c_size_t,
                                          C.POINTER(DistParams), _P, _P]),
    ("skl_cross_dists_all_host", C.c_int, [_P, C.c_size_t, _P, C.c_size_t, C.c_size_t, _P,
                                           C.c_size_t, C.POINTER(DistParams), _P, _P, _P]),
]
DECLARED_SYMBOLS = [s[0] for s in _SIG]

_lib = None


def load():
    """dlopen the in-tree library and attach prototypes.  Raises if it is not built."""
    global _lib
    if _lib is None:
        path = library_path()
        if not os.path.exists(path):
            raise ImportError(
                f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C sketchlib.rust_amd/csrc`).  There is no CPU fallback."
            )
        # If torch is going to share this process (device tensors, torch.distributed), let it
        # load ITS HIP runtime first: two different libamdhip64 copies in one process leave
        # the second one without devices ("No HIP GPUs are available").
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        _lib = _open(path)
    return _lib


def _open(path):
    L = C.CDLL(path)
    # SKL_LIBRARY_OLD=1: an older build of the library (round-over-round timing on one box) may lack newer entry points
    tolerant = bool(os.environ.get("SKL_LIBRARY_OLD")) and bool(os.environ.get("SKL_LIBRARY"))
    for name, restype, argtypes in _SIG:
        try:
            fn = getattr(L, name)  # AttributeError if the export is missing
        except AttributeError:
            if tolerant:
                continue
            raise
        fn.restype = restype
        fn.argtypes = argtypes
    return L


class using_library:
    """with using_library(path): ...  -- the binding calls into another build of the library (the A/B
    build) inside the block.  Handles created inside belong to that build: close them before leaving."""

    def __init__(self, path):
        self.path = path

    def __enter__(self):
        global _lib
        load()
        self.saved = _lib
        _lib = _open(self.path)
        return self

    def __exit__(self, *exc):
        global _lib
        _lib = self.saved
        return False


def _check(rc):
    if rc != OK:
        raise SklError(rc, load().skl_last_error().decode("utf-8", "replace"))


def _ptr(buf):
    """(address, on_device) of a numpy array or a device tensor."""
    if buf is None:
        return None, 0
    if isinstance(buf, np.ndarray):
        return buf.ctypes.data, 0
    if hasattr(buf, "data_ptr"):
        return buf.data_ptr(), 1 if buf.is_cuda else 0
    raise TypeError(f"unsupported buffer type {type(buf)}")


LOG_UNMATCHED = 1
TIES_CANONICAL, TIES_REFERENCE = 0, 1


def ctx_flags(ctx=None):
    """skl_ctx_flags(): conditions worth telling the user about (LOG_UNMATCHED: see the header)."""
    return int(load().skl_ctx_flags(ctx._h if ctx is not None else None))


def log_variant():
    """Which restated form of glibc's log() reproduces this host's libm: 0 FMA, 1 SSE2, -1 neither."""
    return int(load().skl_log_variant())


def device_log(ctx, x):
    """ln(x) as the kernels evaluate it on the completeness path (csrc/glibc_log.hpp), on the device."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    _check(load().skl_device_log(ctx._h, x.ctypes.data, x.size, out.ctypes.data))
    return out


def device_count():
    return load().skl_device_count()


class Context:
    """skl_ctx: one per (thread, device)."""

    def __init__(self, device=0, stream=None):
        self._h = _P()
        _check(load().skl_ctx_create(device, C.byref(self._h)))
        self.device = device
        if stream is not None:
            self.set_stream(stream)

    def set_stream(self, stream):
        """stream: integer hipStream_t handle (torch.cuda.current_stream().cuda_stream): 0 is
        the device's default stream; None restores the context's own (non-blocking) stream."""
        if stream is None:
            _check(load().skl_ctx_set_stream(self._h, None))
        elif int(stream) == 0:
            _check(load().skl_ctx_use_default_stream(self._h))
        else:
            _check(load().skl_ctx_set_stream(self._h, _P(stream)))

    def synchronize(self):
        _check(load().skl_ctx_synchronize(self._h))

    def reload_env(self):
        """Re-read the SKL_* environment switches (they are otherwise read when the context is created)."""
        _check(load().skl_ctx_reload_env(self._h))

    def timing_enable(self, every=1):
        """Bracket every `every`-th pair-kernel launch with HIP events (0: off, the library's default)."""
        _check(load().skl_ctx_timing_enable(self._h, int(every)))

    def timing_reset(self):
        _check(load().skl_ctx_timing_reset(self._h))

    def kernel_ms(self):
        """(summed pair-kernel device ms, launches) since the last timing_reset()."""
        ms = C.c_float()
        n = C.c_int()
        _check(load().skl_ctx_kernel_ms(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def early_break_stats(self):
        """(pairs of early-break core/accessory launches, pairs among them completed one by one) since the context was made."""
        a, b = C.c_uint64(), C.c_uint64()
        _check(load().skl_ctx_early_break_stats(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def early_break_blocks(self):
        """What the last dense core/accessory call decided: {"blocks": (rows, cols), "shifts": (rows, cols), "pooled_lengths": ke or 0,
        "mixed": bool, "block_lengths": uint8 [rows, cols] or None}."""
        br, bc, sr, sc = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        pooled, mixed = C.c_int(), C.c_int()
        _check(load().skl_ctx_early_break_blocks(self._h, C.byref(br), C.byref(bc), C.byref(sr), C.byref(sc), C.byref(pooled), C.byref(mixed), None, 0))
        table = None
        if mixed.value:
            table = np.zeros((br.value, bc.value), dtype=np.uint8)
            _check(load().skl_ctx_early_break_blocks(self._h, None, None, None, None, None, None, table.ctypes.data, table.size))
        return {"blocks": (br.value, bc.value), "shifts": (sr.value, sc.value), "pooled_lengths": pooled.value, "mixed": bool(mixed.value),
                "block_lengths": table}

    def knn_prune_stats(self, full=False):
        """(tiles, tiles left early) of the last self kNN call's prunable launches; full=True: + (stages of a whole tile,
        stages the pruned tiles had walked) and the share of the walk actually made."""
        a, b, c, d, e = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        _check(load().skl_ctx_knn_prune_stats(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d), C.byref(e)))
        if not full:
            return a.value, b.value
        walked = ((a.value - b.value) * c.value + d.value) / (a.value * c.value) if a.value and c.value else 1.0
        return {"tiles": a.value, "tiles_left_early": b.value, "tiles_sparse_walk": e.value, "stages_per_tile": c.value, "stages_walked_in_pruned_tiles": d.value,
                "share_of_the_walk_made": walked}

    def set_knn_ties(self, mode):
        """TIES_REFERENCE (the library's default: the reference binary's lists) or TIES_CANONICAL: see the header."""
        _check(load().skl_ctx_set_knn_ties(self._h, int(mode)))

    def clock_sampler_start(self, interval_us=20, max_samples=1 << 16):
        """One-wave shader-clock sampler next to the context's kernels (diagnostic; see the header)."""
        _check(load().skl_clock_sampler_start(self._h, interval_us, max_samples))

    def clock_sampler_stop(self):
        """-> {"ghz": median, "p10", "p90", "mean", "intervals"}; call after synchronize(), never after a device-wide sync."""
        med, p10, p90, mean, n = C.c_double(), C.c_double(), C.c_double(), C.c_double(), C.c_int()
        _check(load().skl_clock_sampler_stop(self._h, C.byref(med), C.byref(p10), C.byref(p90), C.byref(mean), C.byref(n)))
        return {"ghz": med.value, "p10": p10.value, "p90": p90.value, "mean": mean.value, "intervals": n.value}

    def last_kernel(self):
        return load().skl_ctx_last_kernel(self._h).decode()

    def close(self):
        if self._h:
            load().skl_ctx_destroy(self._h)
            self._h = _P()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- slabs ----
    def sketches(self, bins, n, kmers, sketchsize64, completeness=None):
        return Sketches(self, bins, n, kmers, sketchsize64, completeness)


class Sketches:
    """skl_sketches: a MultiSketch's bins resident on the device."""

    def __init__(self, ctx, bins, n, kmers, sketchsize64, completeness=None):
        self.ctx = ctx
        self.n = int(n)
        self.kmers = np.ascontiguousarray(kmers, dtype=np.uintp)
        self.nk = len(self.kmers)
        self.ss64 = int(sketchsize64)
        if isinstance(bins, np.ndarray):
            if bins.dtype == np.int64:      # (what torch hands over: the same bits, no 8-bytes-per-word conversion pass)
                bins = bins.view(np.uint64)
            bins = np.ascontiguousarray(bins, dtype="<u8").reshape(-1)
            assert bins.size == self.n * self.nk * self.ss64 * 14, "bins size mismatch"
        addr, on_dev = _ptr(bins)
        self._h = _P()
        _check(load().skl_sketches_create(ctx._h, addr, on_dev, self.n, self.nk,
                                          self.kmers.ctypes.data, self.ss64, C.byref(self._h)))
        if completeness is not None:
            self.set_completeness(completeness)

    def set_completeness(self, completeness):
        if completeness is None:
            _check(load().skl_sketches_set_completeness(self._h, None))
            return
        comp = np.ascontiguousarray(completeness, dtype=np.float64)
        assert comp.size == self.n
        _check(load().skl_sketches_set_completeness(self._h, comp.ctypes.data))

    def set_k(self, kmer=None, ani=False, cutoff=0.64):
        """distances::set_k (mod.rs:25-37)."""
        p = DistParams()
        _check(load().skl_set_k(self._h, 0 if kmer is None else int(kmer), int(ani), cutoff,
                                C.byref(p)))
        return p

    def close(self):
        if self._h:
            load().skl_sketches_destroy(self._h)
            self._h = _P()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def params(dist_type=COREACC, k_idx=0, ani=False, cutoff=0.64):
    return DistParams(dist_type, int(ani), k_idx, cutoff)


def ncols(p):
    return 2 if p.dist_type == COREACC else 1


def self_pairs(n, r0=0, r1=None):
    r1 = n if r1 is None else r1
    r1 = min(r1, max(n - 1, 0))
    if r1 <= r0:
        return 0
    upto = lambda r: r * n - r * (r + 1) // 2  # noqa: E731
    return upto(r1) - upto(r0)


# ---- dense ----

def self_dists_all(ctx, s, p, out=None):
    """distances::self_dists_all (mod.rs:58-130) -> [n(n-1)/2, ncols] f32."""
    n_pairs = self_pairs(s.n)
    if out is None:
        out = np.zeros((n_pairs, ncols(p)), dtype=np.float32)
    addr, dev = _ptr(out)
    _check(load().skl_self_dists_all(ctx._h, s._h, C.byref(p), addr, dev))
    return out


def self_dists_rows(ctx, s, p, r0, r1, out=None):
    if out is None:
        out = np.zeros((self_pairs(s.n, r0, r1), ncols(p)), dtype=np.float32)
    addr, dev = _ptr(out)
    _check(load().skl_self_dists_rows(ctx._h, s._h, C.byref(p), r0, r1, addr, dev))
    return out


def cross_dists_all(ctx, r, q, p, out=None):
    """distances::cross_dists_all (mod.rs:227-297) -> [n_ref, n_query, ncols] f32."""
    if out is None:
        out = np.zeros((r.n, q.n, ncols(p)), dtype=np.float32)
    addr, dev = _ptr(out)
    _check(load().skl_cross_dists_all(ctx._h, r._h, q._h, C.byref(p), addr, dev))
    return out


def cross_dists_rows(ctx, r, q, p, r0, r1, out=None):
    if out is None:
        out = np.zeros((r1 - r0, q.n, ncols(p)), dtype=np.float32)
    addr, dev = _ptr(out)
    _check(load().skl_cross_dists_rows(ctx._h, r._h, q._h, C.byref(p), r0, r1, addr, dev))
    return out


# ---- sparse ----

def _knn_out(rows, knn):
    return (np.zeros((rows, knn), dtype=np.uint64), np.zeros((rows, knn), dtype=np.float32),
            np.zeros((rows, knn), dtype=np.float32))


def self_dists_knn(ctx, s, p, knn, r0=0, r1=None):
    """distances::self_dists_knn (mod.rs:133-224) -> (idx, d0, d1) each [rows, knn]."""
    r1 = s.n if r1 is None else r1
    idx, d0, d1 = _knn_out(r1 - r0, knn)
    _check(load().skl_self_dists_knn_rows(ctx._h, s._h, C.byref(p), knn, r0, r1, idx.ctypes.data,
                                          d0.ctypes.data, d1.ctypes.data, 0))
    return idx, d0, d1


def cross_dists_knn(ctx, r, q, p, knn, q0=0, q1=None):
    """distances::cross_dists_knn (mod.rs:306-395); rows = queries, idx into refs."""
    q1 = q.n if q1 is None else q1
    idx, d0, d1 = _knn_out(q1 - q0, knn)
    _check(load().skl_cross_dists_knn_rows(ctx._h, r._h, q._h, C.byref(p), knn, q0, q1,
                                           idx.ctypes.data, d0.ctypes.data, d1.ctypes.data, 0))
    return idx, d0, d1


def knn_band_rows(s, p, participants=1):
    """Band height of the one-evaluation self kNN that every participant agrees on."""
    return int(load().skl_knn_band_rows(s._h, C.byref(p), participants))


def self_dists_knn_partial(ctx, s, p, knn, band_rows, bands, out=None):
    """This participant's share (ascending band indices) of the one-evaluation self kNN:
    -> (state_key u32, state_idx u32, state_d1 f32 | None), each [n, knn], for ALL n rows.
    `out` = a tuple of preallocated numpy arrays or device tensors (int32 bit patterns) instead."""
    bands = np.ascontiguousarray(bands, dtype=np.uint32)
    coreacc = p.dist_type == COREACC
    if out is None:
        out = (np.empty((s.n, knn), dtype=np.uint32), np.empty((s.n, knn), dtype=np.uint32),
               np.empty((s.n, knn), dtype=np.float32) if coreacc else None)
    (ka, dev), (ia, _), (da, _) = _ptr(out[0]), _ptr(out[1]), _ptr(out[2])
    _check(load().skl_self_dists_knn_partial(ctx._h, s._h, C.byref(p), knn, band_rows,
                                             bands.ctypes.data if bands.size else None, bands.size, ka, ia, da, dev))
    return out


def knn_merge_states(ctx, state_key, state_idx, state_d1, ani=False, out=None):
    """Partial states of the same rows, stacked [n_states, rows, knn] (numpy or device tensors)
    -> (idx u64, d0 f32, d1 f32 | None) [rows, knn] as self_dists_knn returns them."""
    n_states, rows, knn = state_key.shape
    (ka, dev), (ia, _), (da, _) = _ptr(state_key), _ptr(state_idx), _ptr(state_d1)
    if out is None:
        out = _knn_out(rows, knn)
    (oi, odev), (o0, _), (o1, _) = _ptr(out[0]), _ptr(out[1]), _ptr(out[2])
    _check(load().skl_knn_merge_states(ctx._h, n_states, rows, knn, ka, ia, da, dev, int(ani), oi, o0,
                                       o1 if state_d1 is not None else None, odev))
    return out


def knn_heaps_alloc(n, knn, coreacc, device):
    """Travelling heaps of the reference-order pipeline (skl_self_dists_knn_window): device tensors for ALL n rows, empty."""
    import torch

    return {"h_key": torch.zeros((n, knn), dtype=torch.float32, device=device),
            "h_id": torch.zeros((n, knn), dtype=torch.int32, device=device),
            "h_d1": torch.zeros((n, knn), dtype=torch.float32, device=device) if coreacc else None,
            "h_len": torch.zeros((n,), dtype=torch.int32, device=device),
            "thr": torch.full((n,), -1, dtype=torch.int32, device=device)}     # 0xFFFFFFFF: not full


def self_dists_knn_window(ctx, s, p, knn, band_rows, band, col_lo, col_hi, heaps):
    """One row band of one participant's column window (the header's skl_self_dists_knn_window): heaps updated in place."""
    ptr = lambda t: _ptr(t)[0]
    _check(load().skl_self_dists_knn_window(ctx._h, s._h, C.byref(p), knn, band_rows, band, col_lo, col_hi, ptr(heaps["h_key"]),
                                            ptr(heaps["h_id"]), ptr(heaps["h_d1"]), ptr(heaps["h_len"]), ptr(heaps["thr"])))


def gather_bands_rccl(ctxs, bands, dst, offsets_bytes, loopback_through_rccl=False):
    """skl_gather_bands_rccl: device tensors bands[d] (made on ctxs[d]'s device and stream) land at byte offsets offsets_bytes[d]
    of the device tensor `dst` on ctxs[0]'s device; asynchronous on the contexts' streams."""
    n = len(ctxs)
    handles = (_P * n)(*[c._h for c in ctxs])
    ptrs = (_P * n)(*[_ptr(b)[0] for b in bands])
    sizes = (C.c_size_t * n)(*[b.numel() * b.element_size() for b in bands])
    offs = (C.c_size_t * n)(*[int(o) for o in offsets_bytes])
    _check(load().skl_gather_bands_rccl(handles, n, ptrs, sizes, _ptr(dst)[0], offs, int(bool(loopback_through_rccl))))


def knn_logs_alloc(n, cap, coreacc, device):
    """Accept logs of the decoupled column windows (skl_self_dists_knn_window_logged): device tensors for n rows, empty."""
    import torch

    return {"rec": torch.zeros((n, cap, 2 if coreacc else 1), dtype=torch.float32, device=device),
            "id": torch.zeros((n, cap), dtype=torch.int32, device=device),
            "len": torch.zeros((n,), dtype=torch.int32, device=device), "cap": cap}


def self_dists_knn_window_logged(ctx, s, p, knn, band_rows, band, col_lo, col_hi, heaps, logs):
    """skl_self_dists_knn_window_logged: as self_dists_knn_window, and every candidate a heap takes is appended to the row's log."""
    ptr = lambda t: _ptr(t)[0]
    _check(load().skl_self_dists_knn_window_logged(ctx._h, s._h, C.byref(p), knn, band_rows, band, col_lo, col_hi, ptr(heaps["h_key"]),
                                                   ptr(heaps["h_id"]), ptr(heaps["h_d1"]), ptr(heaps["h_len"]), ptr(heaps["thr"]),
                                                   ptr(logs["rec"]), ptr(logs["id"]), ptr(logs["len"]), logs["cap"]))


def knn_heaps_replay(ctx, heaps, r0, r1, knn, rec, ids, lens):
    """skl_knn_heaps_replay: the heaps of rows [r0, r1) (of `heaps`, arrays over all rows) are fed the logged candidates
    rec [r1 - r0, cap, 1 | 2] f32 / ids [r1 - r0, cap] i32 / lens [r1 - r0] i32 (device tensors), each row's in the order logged.
    The call is asynchronous on the CONTEXT's stream: tensors made on another stream must be complete (and stay alive) until it ran."""
    if r1 <= r0:
        return
    coreacc = heaps["h_d1"] is not None
    cap = int(rec.shape[1])
    if cap == 0:
        return
    rec, ids, lens = rec.contiguous(), ids.contiguous(), lens.contiguous()
    sl = lambda t: None if t is None else t[r0:r1]
    ptr = lambda t: _ptr(t)[0]
    _check(load().skl_knn_heaps_replay(ctx._h, r1 - r0, knn, int(coreacc), ptr(rec), ptr(ids), ptr(lens), cap, ptr(sl(heaps["h_key"])),
                                       ptr(sl(heaps["h_id"])), ptr(sl(heaps["h_d1"])), ptr(sl(heaps["h_len"])), ptr(sl(heaps["thr"]))))


def knn_heaps_finalize(ctx, heaps, r0, r1, knn, ani=False):
    """into_sorted_vec of the heaps of rows [r0, r1) -> (idx i64, d0 f32, d1 f32 | None) device tensors."""
    import torch

    dev = heaps["h_key"].device
    rows = r1 - r0
    idx = torch.empty((rows, knn), dtype=torch.int64, device=dev)
    d0 = torch.empty((rows, knn), dtype=torch.float32, device=dev)
    d1 = torch.empty((rows, knn), dtype=torch.float32, device=dev) if heaps["h_d1"] is not None else None
    if rows:
        ptr = lambda t: _ptr(t)[0]
        _check(load().skl_knn_heaps_finalize(ctx._h, rows, knn, ptr(heaps["h_key"][r0:]), ptr(heaps["h_id"][r0:]),
                                             ptr(heaps["h_d1"][r0:]) if d1 is not None else None, ptr(heaps["h_len"][r0:]), int(ani),
                                             ptr(idx), ptr(d0), ptr(d1)))
    return idx, d0, d1


def self_dists_knn_candidates(ctx, s, p, knn, row_offsets, cand):
    """Candidate-list kNN (device half of self_dists_knn_precluster, mod.rs:399-553): row i is
    compared with cand[row_offsets[i]:row_offsets[i+1]] only (ascending ids, i excluded)."""
    row_offsets = np.ascontiguousarray(row_offsets, dtype=np.uint64)
    cand = np.ascontiguousarray(cand, dtype=np.uint32)
    assert row_offsets.size == s.n + 1 and int(row_offsets[-1]) == cand.size
    idx, d0, _ = _knn_out(s.n, knn)
    _check(load().skl_self_dists_knn_candidates(ctx._h, s._h, C.byref(p), knn, row_offsets.ctypes.data,
                                                cand.ctypes.data if cand.size else None, idx.ctypes.data,
                                                d0.ctypes.data))
    return idx, d0


def self_dists_knn_shared_bins(ctx, s, p, knn, skq):
    """Precluster kNN with the candidate lists built on the device from the index sketches
    (skq: [n, sketch_size] u16, row i = sample i).  -> (idx, d0, total candidate pairs)."""
    skq = np.ascontiguousarray(skq, dtype=np.uint16)
    assert skq.shape[0] == s.n
    idx, d0, _ = _knn_out(s.n, knn)
    total = C.c_uint64(0)
    _check(load().skl_self_dists_knn_shared_bins(ctx._h, s._h, C.byref(p), knn, skq.ctypes.data, skq.shape[1],
                                                 idx.ctypes.data, d0.ctypes.data, C.byref(total)))
    return idx, d0, int(total.value)


def sketch_signs(ctx, codes, code_begin, offsets, offset_begin, kmers, num_bins, rc=True):
    """GPU bin minima of the canonical ntHash (get_signs_no_densify, sketch/mod.rs:156-176):
    -> [n_samples, nk, num_bins] uint64, u64::MAX for empty bins."""
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    code_begin = np.ascontiguousarray(code_begin, dtype=np.uint64)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    offset_begin = np.ascontiguousarray(offset_begin, dtype=np.uint64)
    kmers = np.ascontiguousarray(kmers, dtype=np.uintp)
    n = code_begin.size - 1
    out = np.zeros((n, kmers.size, num_bins), dtype=np.uint64)
    _check(load().skl_sketch_signs(ctx._h, codes.ctypes.data if codes.size else None, code_begin.ctypes.data,
                                   offsets.ctypes.data if offsets.size else None, offset_begin.ctypes.data, n,
                                   kmers.ctypes.data, kmers.size, num_bins, int(rc), out.ctypes.data))
    return out


def pack_codes(codes, code_begin):
    """One-byte 2-bit codes -> the packed form skl_sketch_signs_packed takes: per sample ceil(len / 16) u32 words, code c at
    bits 2 (c % 16) of word c / 16, the last word zero-padded."""
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    code_begin = np.ascontiguousarray(code_begin, dtype=np.uint64)
    parts = []
    for s in range(code_begin.size - 1):
        c = codes[int(code_begin[s]):int(code_begin[s + 1])] & 3
        pad = (-c.size) % 16
        if pad:
            c = np.concatenate([c, np.zeros(pad, dtype=np.uint8)])
        c = c.reshape(-1, 16).astype(np.uint32)
        parts.append((c << (2 * np.arange(16, dtype=np.uint32))[None, :]).sum(axis=1, dtype=np.uint32))
    return np.concatenate(parts) if parts else np.zeros(0, dtype=np.uint32)


def sketch_signs_packed(ctx, packed, code_begin, offsets, offset_begin, kmers, num_bins, rc=True):
    """skl_sketch_signs_packed: as sketch_signs, the bases already at 2 bits each (pack_codes)."""
    packed = np.ascontiguousarray(packed, dtype=np.uint32)
    code_begin = np.ascontiguousarray(code_begin, dtype=np.uint64)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    offset_begin = np.ascontiguousarray(offset_begin, dtype=np.uint64)
    kmers = np.ascontiguousarray(kmers, dtype=np.uintp)
    n = code_begin.size - 1
    out = np.zeros((n, kmers.size, num_bins), dtype=np.uint64)
    _check(load().skl_sketch_signs_packed(ctx._h, packed.ctypes.data if packed.size else None, code_begin.ctypes.data,
                                          offsets.ctypes.data if offsets.size else None, offset_begin.ctypes.data, n,
                                          kmers.ctypes.data, kmers.size, num_bins, int(rc), out.ctypes.data))
    return out


# ---- raw counts ----

def self_binmatch(ctx, s):
    out = np.zeros((self_pairs(s.n), s.nk), dtype=np.uint32)
    _check(load().skl_self_binmatch(ctx._h, s._h, out.ctypes.data, 0))
    return out


def cross_binmatch(ctx, r, q):
    out = np.zeros((r.n, q.n, r.nk), dtype=np.uint32)
    _check(load().skl_cross_binmatch(ctx._h, r._h, q._h, out.ctypes.data, 0))
    return out


# ---- one-shot host forms ----

def self_dists_all_host(bins, n, kmers, sketchsize64, p, completeness=None):
    bins = np.ascontiguousarray(bins, dtype="<u8").reshape(-1)
    kmers = np.ascontiguousarray(kmers, dtype=np.uintp)
    out = np.zeros((self_pairs(n), ncols(p)), dtype=np.float32)
    comp = None if completeness is None else np.ascontiguousarray(completeness, dtype=np.float64)
    _check(load().skl_self_dists_all_host(bins.ctypes.data, n, len(kmers), kmers.ctypes.data,
                                          sketchsize64, C.byref(p),
                                          None if comp is None else comp.ctypes.data,
                                          out.ctypes.data))
    return out
