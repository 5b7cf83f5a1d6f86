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


def _torchrun_world(world, script_args, port):
    """`world` ranks sharing the ONE GPU of the box over gloo (SKL_BENCH_BACKEND=gloo: RCCL refuses two ranks on a
    device): everything of the N > 1 path except the RCCL transport -- the band partition, every rank's `*_rows` call on
    real device buffers, the gather, max-over-ranks timing and the band-by-band self-check on rank 0."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", SKL_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port)] + script_args
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, (res.stdout[-2000:], res.stderr[-4000:])
    return _last_json(res.stdout)


@pytest.mark.parametrize("world,gather,port", [(2, "rccl", 29621), (3, "host", 29623), (8, "rccl", 29625)])
def test_bench_with_several_ranks_on_the_one_gpu(gpu_ctx, world, gather, port):
    """(`--gather rccl` under the gloo debugging backend is the host-staged send/recv gather.  World 8 = the north star's
    partition; its ranks run without the clock sampler: eight spinning one-wave kernels of eight processes on ONE GPU
    time-slice each other for minutes.)"""
    extra = ["--no-clock-sampler"] if world > 3 else []
    line = _torchrun_world(world, ["bench.py", "--gpus", str(world), "--samples", "4000", "--steps", "3", "--warmup", "1",
                                   "--gather", gather, "--no-cpu-baseline", "--precondition-s", "0.05", "--msg-mib", "8"] + extra, port)
    assert line["n_gpus"] == world and line["scaling"] == "strong" and line["config"]["n_samples"] == 4000
    assert f"{world} row band(s)" in line["config"]["partition"]
    assert ("shared, pinned host buffer" if gather == "host" else "gather to rank 0") in line["config"]["partition"]
    # rank 0 compared the assembled matrix band by band with what it computes alone, then 2 010 pairs with the oracle
    assert line["config"]["verified_pairs"] >= 2000 and line["config"]["max_abs_err"] <= 1e-6
    assert line["value"] > 0 and line["roofline"]["pairs_per_launch"] < 4000 * 3999 // 2


def test_plain_bench_gpus_2_launches_itself(gpu_ctx):
    """`python bench.py --gpus 2 ...` as typed -- no torch.distributed.run in front, WORLD_SIZE unset: bench.py starts the
    two ranks itself as a fresh child process (here over gloo, both on the one GPU) and relays rank 0's JSON line, which
    carries the CPU baseline and the N = 1 figure of the same workload."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "TORCHELASTIC_RUN_ID")}
    env["SKL_BENCH_BACKEND"] = "gloo"
    res = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--samples", "4000", "--steps", "3", "--warmup", "1",
                          "--precondition-s", "0.05", "--msg-mib", "8"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, (res.stdout[-2000:], res.stderr[-4000:])
    line = _last_json(res.stdout)
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["config"]["n_samples"] == 4000
    assert line["config"]["verified_pairs"] >= 2000 and line["config"]["max_abs_err"] <= 1e-6
    n1 = line["config"]["n1_same_workload"]
    assert n1["n_gpus"] == 1 and n1["pairs_per_s"] > 0
    cpu = line["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["value"] > 0 and cpu["cores"] >= 1


def test_bench_line_on_the_drivers_arguments(gpu_ctx):
    """`--steps 20 --warmup 5` is what the driver passes: the roofline's kernel time must not depend on it."""
    res = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--steps", "20", "--warmup", "5", "--no-cpu-baseline",
                          "--no-secondary"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-4000:]
    line = _last_json(res.stdout)
    assert line["n_gpus"] == 1 and line["steps"] == 20 and line["warmup"] == 5 and line["unit"] == "pairs/s"
    assert "configs[1]" in line["config"]["workload"] and line["config"]["pairs"] == 499500
    assert 0.5 <= line["config"]["preconditioning_s"] <= 3.0
    # the cold figure (the same 5 + 20 steps as the first launches of the process) rides beside the preconditioned value
    assert 0.5 * line["value"] < line["config"]["cold_pairs_per_s"] < 1.1 * line["value"], line["config"]["cold"]
    assert line["scaling"] == "strong"
    rf = line["roofline"]
    assert rf["bound"] == "valu" and 0.05 < rf["frac"] <= 1.0
    assert abs(rf["achieved"] / rf["peak"] - rf["frac"]) < 1e-9
    assert rf["hbm_no_reuse"]["algorithmic_bytes_per_pair"] == 71688
    assert line["config"]["verified_pairs"] >= 2000 and line["config"]["max_abs_err"] <= 1e-6
    assert rf["kernel_launches_timed"] >= 50           # the fixed pass, whatever --steps is
    # the kernel time the roofline uses is consistent with the step time the value uses (same warm device)
    assert rf["kernel_avg_ms"] <= line["ms_per_step"] * 1.05
    # the timed region runs the library's default configuration: no launch of it is bracketed with events
    assert rf["kernel_launches_bracketed_in_timed_region"] == 0
    # the clock beside the fraction is read DURING the roofline pass's launches, on this box
    clk = rf["in_kernel_clock"]
    assert clk["source"].startswith("live:") and 1.0 < clk["ghz"] < 2.6 and clk["intervals"] >= 50, clk
    assert abs(rf["frac_at_in_kernel_clock"] - rf["frac"] * 2.4 / clk["ghz"]) < 1e-9
    # HBM bytes per launch measured by the two rocprofv3 --pmc child passes of this run
    tr = rf["traffic"]
    assert tr["source"].startswith("live:") and 4.0e7 < tr["bytes"] < 2.0e9, tr


def test_clock_sampler_reads_a_plausible_clock(skl, gpu_ctx):
    import numpy as np
    from sketchlib.rust_amd import synth

    bins = synth.set_u(600, 5, 64)
    g = gpu_ctx.sketches(bins, 600, [15, 19, 23, 27, 31], 64)
    out = np.zeros((600 * 599 // 2, 2), dtype=np.float32)
    skl.self_dists_all(gpu_ctx, g, g.set_k(), out=out)       # warm
    gpu_ctx.clock_sampler_start(20, 1 << 14)
    for _ in range(40):
        skl.self_dists_all(gpu_ctx, g, g.set_k(), out=out)
    gpu_ctx.synchronize()
    clk = gpu_ctx.clock_sampler_stop()
    assert clk["intervals"] >= 20 and 0.8 < clk["p10"] <= clk["ghz"] <= clk["p90"] < 2.6, clk
    # it ends by itself too (max_samples), and a second start after a stop works
    gpu_ctx.clock_sampler_start(4, 8)
    import time
    time.sleep(0.05)
    clk = gpu_ctx.clock_sampler_stop()
    assert clk["intervals"] <= 7
    g.close()


def test_bench_rccl_path_with_one_rank(gpu_ctx):
    line = _torchrun(["bench.py", "--gpus", "1", "--workload", "cfg3", "--samples", "6000", "--steps", "3", "--warmup", "1",
                      "--loopback", "--no-cpu-baseline", "--msg-mib", "16"], 29611)     # 144 MB band = 9 messages
    assert line["n_gpus"] == 1 and line["config"]["n_samples"] == 6000
    assert "RCCL" in line["config"]["partition"] and "overlapped" in line["config"]["partition"] and "16 MiB" in line["config"]["partition"]
    assert line["config"]["verified_pairs"] >= 2000 and line["config"]["max_abs_err"] <= 1e-6
    assert 0 < line["roofline"]["frac"] <= 1.0


def test_bench_host_gather_with_one_rank(gpu_ctx):
    line = _torchrun(["bench.py", "--gpus", "1", "--workload", "cfg3", "--samples", "6000", "--steps", "3", "--warmup", "1",
                      "--gather", "host", "--no-cpu-baseline"], 29613)
    assert "shared, pinned host buffer" in line["config"]["partition"]
    assert line["config"]["verified_pairs"] >= 2000 and line["config"]["max_abs_err"] <= 1e-6


def test_rccl_loopback_gather_at_cfg3_full_size(gpu_ctx):
    """BASELINE configs[2] at FULL size through the N > 1 code path on the one GPU: the 40 GB band is sent by rank
    0 to itself over RCCL in 1 GiB messages (38 send/recv pairs per step, two rotating band buffers), and the
    assembled matrix is compared band by band with what rank 0 computes alone."""
    line = _torchrun(["bench.py", "--gpus", "1", "--workload", "cfg3", "--steps", "2", "--warmup", "1", "--loopback",
                      "--no-cpu-baseline", "--precondition-s", "0"], 29614)
    assert line["config"]["n_samples"] == 100000 and line["config"]["pairs"] == 4999950000
    assert "RCCL" in line["config"]["partition"] and "1024 MiB" in line["config"]["partition"]
    assert line["config"]["verified_pairs"] >= 2000 and line["config"]["max_abs_err"] <= 1e-6
    print(f"cfg3 full through the RCCL loopback gather: {line['value']:.3g} pairs/s, {line['ms_per_step']:.0f} ms per step")


def test_knn_multi_rccl_path_with_one_rank(gpu_ctx):
    line = _torchrun(["scripts/bench_knn_multi.py", "--samples", "20000", "--knn", "50", "--clustered", "--check"], 29612)
    assert line["n_gpus"] == 1 and line["shard_equals_row_by_row"] is True


def test_knn_multi_canonical_all_to_all_with_one_rank(gpu_ctx):
    line = _torchrun(["scripts/bench_knn_multi.py", "--samples", "20000", "--knn", "50", "--clustered", "--check", "--ties", "canonical"], 29616)
    assert line["ties"] == "canonical" and line["shard_equals_row_by_row"] is True


@pytest.mark.parametrize("world,coreacc,port", [(2, False, 29631), (3, True, 29633), (4, False, 29635)])
def test_knn_reference_order_travelling_heaps_with_several_ranks_on_the_one_gpu(gpu_ctx, world, coreacc, port):
    """The reference's tie order over `world` ranks, every pair evaluated once (skl_self_dists_knn_window + heaps handed
    from rank to rank band by band, multi_gpu.self_knn_once_reference), all ranks on the box's one GPU over gloo: rank 0's
    row shard equals the single-device row-by-row replay of the same rows (ids, order, distances)."""
    args = ["scripts/bench_knn_multi.py", "--samples", "12000", "--knn", "20", "--clustered", "--check", "--ties", "reference"]
    line = _torchrun_world(world, args + (["--coreacc"] if coreacc else []), port)
    assert line["n_gpus"] == world and line["ties"] == "reference" and "travelling" in line["mode"]
    assert line["shard_equals_row_by_row"] is True


@pytest.mark.parametrize("world,coreacc,port", [(1, False, 29641), (2, True, 29643), (3, False, 29645), (4, True, 29647)])
def test_knn_reference_order_decoupled_windows_with_several_ranks_on_the_one_gpu(gpu_ctx, world, coreacc, port):
    """The same lists with no rank waiting for another (round 6): every rank runs its column window against heaps it has cleared
    itself, logs what they take (skl_self_dists_knn_window_logged), the logs are exchanged and replayed in window order
    (skl_knn_heaps_replay; multi_gpu.self_knn_once_reference_decoupled) -- all ranks on the box's one GPU over gloo: rank 0's
    row shard equals the single-device row-by-row replay of the same rows (ids, order, distances)."""
    args = ["scripts/bench_knn_multi.py", "--samples", "12000", "--knn", "20", "--clustered", "--check", "--ties", "reference", "--decoupled"]
    line = _torchrun_world(world, args + (["--coreacc"] if coreacc else []), port) if world > 1 else _torchrun(args + (["--coreacc"] if coreacc else []), port)
    assert line["n_gpus"] == world and line["ties"] == "reference" and "decoupled" in line["mode"]
    assert line["shard_equals_row_by_row"] is True


def test_c_abi_rccl_gather_of_row_bands_with_one_device(oracle, skl, gpu_ctx):
    """skl_gather_bands_rccl (the C ABI's rendering of "a RCCL gather over xGMI to assemble the output matrix"): two row bands of
    a self matrix computed with device outputs, assembled on the root (a) by the root's own copy and (b) THROUGH RCCL -- the
    band sent to itself, grouped ncclSend / ncclRecv on the context's stream, the one-GPU test of the transport; the
    assembled matrix is the oracle's.  A device listed twice is refused (RCCL takes one rank per device)."""
    import numpy as np
    import torch
    from sketchlib.rust_amd import synth

    kmers, ss64, n = [15, 19, 23, 27, 31], 16, 400
    bins = synth.set_r(n, kmers, ss64, n_clusters=8)
    o, g = oracle.Sketches(bins, n, kmers, ss64), gpu_ctx.sketches(bins, n, kmers, ss64)
    exp = oracle.self_dists_all(o, oracle.COREACC, threads=8).reshape(-1, 2)
    dev = torch.device("cuda", 0)
    cut = 150
    first = cut * n - cut * (cut + 1) // 2                      # pairs of rows [0, cut)
    b0 = torch.zeros((first, 2), dtype=torch.float32, device=dev)
    b1 = torch.zeros((exp.shape[0] - first, 2), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    skl.self_dists_rows(gpu_ctx, g, g.set_k(), 0, cut, out=b0)
    skl.self_dists_rows(gpu_ctx, g, g.set_k(), cut, n, out=b1)
    for loopback in (False, True):
        full = torch.full((exp.shape[0], 2), -1.0, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        skl.gather_bands_rccl([gpu_ctx], [b0], full, [0], loopback_through_rccl=loopback)
        skl.gather_bands_rccl([gpu_ctx], [b1], full, [first * 8], loopback_through_rccl=loopback)
        gpu_ctx.synchronize()
        assert np.array_equal(full.cpu().numpy().view(np.uint32), exp.view(np.uint32)), loopback
    with pytest.raises(skl.SklError):
        skl.gather_bands_rccl([gpu_ctx, gpu_ctx], [b0, b1], full, [0, first * 8])
    g.close()
