"""bench.py's arithmetic (no GPU): the roofline constants are the ones DESIGN.md and the judge use."""
import importlib.util
import os

import numpy as np

from conftest import ROOT


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_roofline_constants():
    b = _bench()
    assert b.algorithmic_bytes_per_pair(5, 64, 2) == 71688          # SURVEY 8(d), cfg 2/3 core/acc
    assert b.algorithmic_bytes_per_pair(5, 32, 2) == 35848          # cfg 4/5
    # per (k, chunk): 28 full-rate instructions + 2 v_bcnt at the measured 3.2 / 1.92 = 5/3 slot each
    slots = (28 + 2 * 5 / 3) * 5 * 64
    assert abs(b.issue_slots_per_pair(5, 64) - slots) < 1e-9 and abs(slots - 10026.67) < 0.01
    assert abs(b.VALU_PEAK_LANE_OPS - 7.8643e13) < 1e10             # 256 CU x 4 SIMD x 32 lanes x 2.4 GHz
    blk = b.valu_block(499500, 0.16e-3, 5, 64, None)
    assert abs(blk["frac"] - (slots * 499500 / 0.16e-3) / b.VALU_PEAK_LANE_OPS) < 1e-12
    assert abs(blk["peak_pairs_per_s"] - 7.8434e9) < 1e6
    assert blk["valu_instructions_per_pair"] == 9600 and "in_kernel_clock" not in blk
    clk = {"ghz": 2.0, "p10": 1.9, "p90": 2.1, "mean": 2.0, "intervals": 100, "source": "test"}
    blk = b.valu_block(499500, 0.16e-3, 5, 64, clk)
    assert abs(blk["frac_at_in_kernel_clock"] - blk["frac"] * 2.4 / 2.0) < 1e-12


def test_the_hidden_child_mode_and_the_driver_arguments_parse():
    """`bench.py --gpus 1 --steps 20 --warmup 5` is what the driver runs; --traffic-probe is the child mode of the
    rocprofv3 --pmc passes.  (Parsing only: no GPU here.)"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    for flag in ("--precondition-s", "--roofline-launches", "--traffic-probe", "--gather", "--msg-mib", "--no-traffic"):
        assert flag in src, flag
    assert "preconditioning_s" in src and "kernel_launches_timed" in src


def test_plain_gpus_2_starts_its_own_ranks_and_fails_only_for_want_of_a_gpu():
    """`python bench.py --gpus 2` typed as is (no torch.distributed.run, WORLD_SIZE unset) must get past the launcher:
    the parent starts the ranks as a child process; here, without a GPU, every rank stops at "needs an MI355X"."""
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["SKL_BENCH_BACKEND"] = "gloo"
    res = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1"], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=600)
    assert "must be launched with torch.distributed.run" not in res.stderr
    assert "torch.distributed.run --nnodes=1 --nproc-per-node=2" in res.stderr, res.stderr[-2000:]
    assert "needs an MI355X" in res.stderr, res.stderr[-3000:]
    assert res.returncode != 0 and not [l for l in res.stdout.splitlines() if l.startswith("{")]


def test_cpu_baseline_sample_is_bounded():
    b = _bench()
    assert b.CPU_SAMPLE_MAX_N == 20_000      # cfg 3's CPU leg: 2e8 pairs, not 5e9
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "n1_same_workload" in src and "cold_pairs_per_s" in src


def test_condensed_index_matches_the_reference_formula():
    b = _bench()
    n = 37
    flat = [b.cond_index(i, j, n) for i in range(n) for j in range(i + 1, n)]
    assert flat == list(range(n * (n - 1) // 2))
    ii, jj = np.array([0, 5, 35]), np.array([1, 9, 36])
    assert b.cond_index(ii, jj, n).tolist() == [0, b.cond_index(5, 9, n), n * (n - 1) // 2 - 1]


def test_counted_lengths_and_the_roofline_block_of_an_early_break_launch():
    """bench.py prices the k-mer lengths the pair kernel COUNTED (the library's kernel description says how many) and puts the
    rate of answers beside it."""
    import bench

    name = ("skl::pair_kernel_kslice<R=16, JL=2, COUNTS, k-sliced, tight> (16x128 tiles) + early break: 3 of 5 k-mer lengths counted, "
            "the pairs still in the running completed by the epilogue")
    assert bench.counted_lengths(name, 5) == 3
    assert bench.counted_lengths(name, 6) == 6                      # (a description of another launch shape: not taken)
    assert bench.counted_lengths("skl::pair_kernel_kslice<R=32, JL=2, COREACC, all k, tight>", 5) == 5
    assert bench.counted_lengths(None, 5) == 5
    full = bench.valu_block(499500, 1.0e-4, 5, 64, None)
    part = bench.valu_block(499500, 1.0e-4, 5, 64, None, counted=3)
    assert part["k_mer_lengths_counted"] == 3 and abs(part["frac"] - 0.6 * full["frac"]) < 1e-12
    assert abs(part["frac_as_if_every_length_were_counted"] - full["frac"]) < 1e-12 and "frac_as_if_every_length_were_counted" not in full
