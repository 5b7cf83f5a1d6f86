"""csrc/glibc_log.hpp restates the host libm's log() for the device (completeness path).  Here, on
the CPU: (1) both restated forms against the glibc objects themselves, extracted from the static
libm this image ships (skipped where there is none); (2) the form the library's probe selects against
the running host's log() on millions of arguments.  IEEE fma/mul/add are the same on the device, so
what holds for the host build of the header holds for the kernels (tests/test_gpu_log.py closes
the loop on the GPU)."""
import json
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

SRC = os.path.join(ROOT, "tests", "native", "glibc_log_check.cpp")
LIBM_A = "/usr/lib/x86_64-linux-gnu/libm-2.35.a"


def _build(tmp_path, extra):
    exe = str(tmp_path / "glibc_log_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", SRC, "-o", exe] + extra)
    return exe


def test_probed_form_equals_host_libm(skl, tmp_path):
    v = skl.log_variant()
    assert v in (0, 1), "this host's log() is neither restated glibc form"
    res = json.loads(subprocess.check_output([_build(tmp_path, []), "host", "4000000"], text=True))
    assert res["arguments"] > 4_000_000
    assert res["mismatch_fma" if v == 0 else "mismatch_sse2"] == 0, res


@pytest.mark.skipif(not (os.path.exists(LIBM_A) and shutil.which("ar")), reason="no static glibc libm to compare with")
def test_both_forms_equal_the_glibc_objects(tmp_path):
    objs = ["e_log.o", "e_log-fma.o", "e_log-fma4.o", "e_log-avx.o", "e_log_data.o", "math_err.o"]
    subprocess.check_call(["ar", "x", LIBM_A] + objs, cwd=tmp_path)
    exe = _build(tmp_path, ["-no-pie", "-DWITH_GLIBC_OBJECTS"] + [str(tmp_path / o) for o in objs])
    res = json.loads(subprocess.check_output([exe, "objects", "4000000"], text=True))
    assert res["mismatch_fma"] == 0 and res["mismatch_sse2"] == 0, res


def test_table_is_the_one_the_generator_reads(tmp_path):
    """glibc_log_data.inc is generated, not typed: regenerating it from the image's libm changes nothing."""
    if not (os.path.exists(LIBM_A) and shutil.which("ar") and shutil.which("objcopy")):
        pytest.skip("no static glibc libm")
    inc = os.path.join(ROOT, "sketchlib.rust_amd", "csrc", "glibc_log_data.inc")
    before = [l for l in open(inc) if not l.startswith("//")]
    subprocess.check_call(["ar", "x", LIBM_A, "e_log_data.o"], cwd=tmp_path)
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.rodata", "e_log_data.o", "t.bin"], cwd=tmp_path)
    import struct
    vals = struct.unpack("<530d", open(tmp_path / "t.bin", "rb").read())
    again = ["%s, %s,\n" % (vals[i].hex(), vals[i + 1].hex()) for i in range(0, 530, 2)]
    assert before == again
