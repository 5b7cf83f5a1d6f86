"""bench.py and scripts/bench_knn_multi.py end to end on the GPU box, including the N > 1 code path:
a one-rank `torch.distributed.run` launch initialises RCCL (backend "nccl"), and `--loopback` makes
rank 0 send its own band to itself, so init_process_group, the pipelined send/recv gather and the
post-run self-check (assembled matrix == the matrix rank 0 computes alone) all execute on hardware."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _last_json(text):
    lines = [l for l in text.splitlines() if l.startswith("{")]
    assert lines, text[-2000:]
    return json.loads(lines[-1])


def _torchrun(script_args, port):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", str(port)] + script_args
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, (res.stdout[-2000:], res.stderr[-4000:])
    return _last_json(res.stdout)


def test_bench_default_line(gpu_ctx):
    res = subprocess.run([sys.executable, "bench.py", "--steps", "40", "--warmup", "10", "--no-cpu-baseline",
                          "--no-secondary"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-4000:]
    line = _last_json(res.stdout)
    assert line["n_gpus"] == 1 and line["steps"] == 40 and line["unit"] == "pairs/s"
    assert "configs[1]" in line["config"]["workload"] and line["config"]["pairs"] == 499500
    rf = line["roofline"]
    assert rf["bound"] == "valu" and 0.05 < rf["frac"] <= 1.0
    assert abs(rf["achieved"] / rf["peak"] - rf["frac"]) < 1e-9
    assert rf["hbm_no_reuse"]["algorithmic_bytes_per_pair"] == 71688
    assert line["config"]["verified_pairs"] >= 2000 and line["config"]["max_abs_err"] <= 1e-6
    assert rf["kernel_launches_timed"] >= 10
    # the kernel time the roofline uses is consistent with the step time the value uses
    assert rf["kernel_avg_ms"] <= line["ms_per_step"] * 1.05
    # the in-kernel clock beside the fraction is measured on this box, after the timed region
    clk = rf["in_kernel_clock"]
    assert clk["source"].startswith("live:") and 1.0 < clk["ghz"] < 2.6, clk
    assert abs(rf["frac_at_in_kernel_clock"] - rf["frac"] * 2.4 / clk["ghz"]) < 1e-9


def test_bench_rccl_path_with_one_rank(gpu_ctx):
    line = _torchrun(["bench.py", "--gpus", "1", "--workload", "cfg3", "--samples", "6000", "--steps", "3", "--warmup", "1",
                      "--loopback", "--no-cpu-baseline"], 29611)
    assert line["n_gpus"] == 1 and line["config"]["n_samples"] == 6000
    assert "RCCL" in line["config"]["partition"] and "overlapped" in line["config"]["partition"]
    assert line["config"]["verified_pairs"] >= 2000 and line["config"]["max_abs_err"] <= 1e-6
    assert 0 < line["roofline"]["frac"] <= 1.0


def test_knn_multi_rccl_path_with_one_rank(gpu_ctx):
    line = _torchrun(["scripts/bench_knn_multi.py", "--samples", "20000", "--knn", "50", "--clustered", "--check"], 29612)
    assert line["n_gpus"] == 1 and line["shard_equals_row_by_row"] is True
