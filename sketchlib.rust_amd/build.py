"""Build recipe for the in-tree HIP libraries (hipcc cross-compiles gfx950 without a GPU).

Two libraries come out of sketchlib.rust_amd/csrc:
  * the product library  csrc/_build/libsketchlib_dist_hip.so (+ the CLI and skl_dbtool);
  * the A/B library      csrc/_build_ab/libsketchlib_dist_hip.so, built with -DSKL_AB: the product
    library plus the kernels and environment switches kept only for A/B timing and for the tests
    that pin those kernels' results (csrc/Makefile).

A build is redone when the SOURCES changed, not when a file's mtime did: the key is a SHA-256 over
every file under csrc/ (outside the build directories) and the public header, kept next to the
library.  A library compiled in another container from the same sources is therefore reused as is,
and one compiled from other sources never is."""
import hashlib
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(CSRC, "_build", "libsketchlib_dist_hip.so")
LIB_AB = os.path.join(CSRC, "_build_ab", "libsketchlib_dist_hip.so")
CLI = os.path.join(CSRC, "_build", "sketchlib")


def library_path():
    """The library the ctypes binding loads: SKL_LIBRARY if set (e.g. the A/B build), else the product library."""
    return os.environ.get("SKL_LIBRARY") or LIB


def ab_library_path():
    return LIB_AB


def source_hash():
    h = hashlib.sha256()
    files = [os.path.join(os.path.dirname(_HERE), "include", "sketchlib_dist.h")]
    for root, dirs, names in os.walk(CSRC):
        dirs[:] = sorted(d for d in dirs if not d.startswith("_build"))
        files += [os.path.join(root, f) for f in sorted(names)]
    for f in files:
        h.update(os.path.relpath(f, _HERE).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _build(ab, force, verbose):
    out_dir = os.path.join(CSRC, "_build_ab" if ab else "_build")
    lib = LIB_AB if ab else LIB
    stamp = os.path.join(out_dir, ".source_sha256")
    want = source_hash()
    have = open(stamp).read().strip() if os.path.exists(stamp) else ""
    targets = [lib] + ([] if ab else [CLI])
    if force or have != want or not all(os.path.exists(t) for t in targets):
        cmd = ["make", "-C", CSRC, "-j8"] + (["AB=1"] if ab else []) + (["-B"] if force or have != want else [])
        res = subprocess.run(cmd, capture_output=not verbose, text=True)
        if res.returncode != 0:
            raise RuntimeError("building the HIP library failed:\n" + (res.stdout or "") + (res.stderr or ""))
        with open(stamp, "w") as f:
            f.write(want + "\n")
    return lib


def build_library(force=False, verbose=False):
    """make -C csrc: *.hip + capi*.cpp -> libsketchlib_dist_hip.so (and the CLI)."""
    return _build(False, force, verbose)


def build_ab_library(force=False, verbose=False):
    """make -C csrc AB=1: the A/B library (tests/test_gpu_kernel_variants.py, scripts/ab_sweep.py)."""
    return _build(True, force, verbose)


PROBE_DIR = os.path.join(os.path.dirname(_HERE), "scripts", "microbench")
CLOCK_PROBE = os.path.join(PROBE_DIR, "_build", "kslice_trace")


def build_clock_probe(force=False):
    """scripts/microbench/kslice_trace: the pair kernels built with s_memtime / wall-clock stamps (a
    diagnostic binary, linked against the A/B library for the host-side helpers).  bench.py runs it
    after its timed region to quote the shader clock the chip holds under the kernel on that box."""
    build_ab_library()
    h = hashlib.sha256(source_hash().encode())
    with open(os.path.join(PROBE_DIR, "kslice_trace.hip"), "rb") as fh:
        h.update(fh.read())
    want = h.hexdigest()
    stamp = os.path.join(PROBE_DIR, "_build", ".kslice_trace_sha256")
    have = open(stamp).read().strip() if os.path.exists(stamp) else ""
    if force or have != want or not os.path.exists(CLOCK_PROBE):
        res = subprocess.run(["bash", os.path.join(PROBE_DIR, "build.sh"), "kslice_trace"], capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("building scripts/microbench/kslice_trace failed:\n" + res.stdout + res.stderr)
        with open(stamp, "w") as f:
            f.write(want + "\n")
    return CLOCK_PROBE
