"""Build recipe for the in-tree HIP library (hipcc cross-compiles gfx950 without a GPU)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(CSRC, "_build", "libsketchlib_dist_hip.so")
CLI = os.path.join(CSRC, "_build", "sketchlib")


def library_path():
    return LIB


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources if os.path.exists(s))


def build_library(force=False, verbose=False):
    """make -C csrc: *.hip + capi*.cpp -> libsketchlib_dist_hip.so (and the CLI)."""
    srcs = []
    for root, _dirs, files in os.walk(CSRC):
        if "_build" in root:
            continue
        srcs += [os.path.join(root, f) for f in files]
    srcs.append(os.path.join(os.path.dirname(_HERE), "include", "sketchlib_dist.h"))
    if force or _stale(LIB, srcs) or (os.path.exists(os.path.join(CSRC, "host")) and _stale(CLI, srcs)):
        cmd = ["make", "-C", CSRC, "-j4"] + (["-B"] if force else [])
        res = subprocess.run(cmd, capture_output=not verbose, text=True)
        if res.returncode != 0:
            raise RuntimeError("building the HIP library failed:\n" + (res.stdout or "") + (res.stderr or ""))
    return LIB
