"""bench.py itself, launched the way the driver launches it for N > 1 (round-2 verdict, "Next round" 2: nothing had
ever executed `python -m torch.distributed.run ... bench.py --gpus N`).  The test box has one GPU, so the ranks
share cuda:0 and the messages go through gloo (APK_SHARE_GPU=1 APK_DIST_BACKEND=gloo, the switches bench.py documents
for exactly this); everything else -- rank grid, brick partition, overlapped exchanges, barrier + max-over-ranks
timing, the JSON contract -- is the path an 8-GPU node runs.  N = 1 is checked for the second, copy-inclusive figure."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _launch(world, extra, timeout=900):
    env = dict(os.environ, APK_SHARE_GPU="1", APK_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "bench.py", "--gpus", str(world)] + extra
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, "bench.py failed on %d ranks:\n%s\n%s" % (world, r.stdout[-3000:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 must print exactly ONE JSON line, got %d:\n%s" % (len(lines), r.stdout[-2000:])
    return json.loads(lines[0])


def _check_contract(d, world, steps, warmup):
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == world and d["steps"] == steps and d["warmup"] == warmup
    assert d["unit"] == "cell-updates/s" and d["scaling"] == "weak" and d["dtype"] == "f64" and d["higher_is_better"] is True
    assert d["value"] > 0 and d["value"] == d["value"] and d["ms_per_step"] > 0
    r = d["roofline"]
    # (`bound` names what binds the stage kernels -- the fp64 vector issue rate for the GLM-MHD marches -- while achieved /
    # peak / frac stay priced against HBM bandwidth, as north_star asks: `priced_against`)
    assert r["bound"] in ("hbm", "fp64_valu_issue") and r["priced_against"] == "hbm" and r["unit"] == "GB/s"
    assert 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12


def test_bench_on_two_ranks_through_the_launcher():
    """128^3 per rank in 64^3 meshblocks (2 x 1 x 1 rank grid): the headline workload's scheme and code path, reduced"""
    d = _launch(2, ["--steps", "2", "--warmup", "1", "--brick", "128", "--meshblock", "64", "--no-cpu-baseline"])
    _check_contract(d, 2, 2, 1)
    c = d["config"]
    assert c["mesh"] == [256, 128, 128] and c["blocks_per_gpu"] == 8 and c["parallelism"].endswith("2x1x1 GPU grid")
    assert "gloo" in c["comm_backend"]
    assert c["overlapped_exchanges_per_cycle"] > 0          # the halo messages fly during the next stage's kernels
    assert c["reductions_per_cycle"] == 1.0    # ONE collective per cycle: the time step and the c_h estimate travel together
    assert c["halo_exchanges_per_cycle"] == 2.0  # one per stage
    # both of them with their x1 strips stored into / read from the message buffers by the stage kernels (warm-up included:
    # per cycle of the run)
    assert c["exchanges_with_x1_strips_in_the_buffers_per_cycle"] == 2.0
    assert "cpu_baseline" not in d and "weak_scaling_base_with_ghost_copies" not in d   # N = 1 only
    # value = zones of ALL ranks * steps / max-over-ranks time
    assert abs(d["value"] - 256 * 128 * 128 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]


def test_bench_restarts_on_the_callback_transport_when_a_rank_fails_to_start():
    """a failure on ONE rank during initialisation / warm-up (injected here; on a node it would be the native RCCL
    transport's first real exchange) makes EVERY rank start over with comm='torch' instead of hanging or dying"""
    env = dict(os.environ, APK_SHARE_GPU="1", APK_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1",
               APK_BENCH_INJECT_START_FAILURE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--brick", "64",
           "--meshblock", "32", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "retrying with comm='torch'" in r.stderr and "injected start-up failure on rank 1" in r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    _check_contract(d, 2, 2, 1)
    assert "torch.distributed callbacks" in d["config"]["comm_backend"]


def test_bench_on_eight_ranks_through_the_launcher():
    """the 2 x 2 x 2 rank grid of the 8-GPU run on 64^3 per rank (one GPU shared by eight processes: slow, but every
    rank has 7 peers, three late faces per block and the same message pattern as on the node)"""
    d = _launch(8, ["--steps", "2", "--warmup", "1", "--brick", "64", "--meshblock", "32", "--no-cpu-baseline"], timeout=1500)
    _check_contract(d, 8, 2, 1)
    c = d["config"]
    assert c["mesh"] == [128, 128, 128] and c["blocks_per_gpu"] == 8 and c["parallelism"].endswith("2x2x2 GPU grid")
    assert c["overlapped_exchanges_per_cycle"] > 0 and c["reductions_per_cycle"] == 1.0 and c["halo_exchanges_per_cycle"] == 2.0
    assert c["exchanges_with_x1_strips_in_the_buffers_per_cycle"] == 2.0


def test_bench_on_one_gpu_reports_the_copy_inclusive_base():
    """N = 1 (no launcher): the default figure has no ghost copies at all (direct neighbour addressing); the second
    figure repeats the steps with the same-rank copies an N > 1 run performs between bricks"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--brick", "128", "--meshblock", "64", "--no-cpu-baseline"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    _check_contract(d, 1, 3, 1)
    assert d["config"]["same_rank_ghost_copies_skipped_per_cycle"] == 2.0
    base = d["weak_scaling_base_with_ghost_copies"]
    assert 0 < base["value"] <= d["value"] * 1.05 and base["unit"] == "cell-updates/s"
    assert "fused_m12f_kernel" in d["roofline"]["dominant_kernel"] or "fused_dc3" in d["roofline"]["dominant_kernel"]
    assert d["roofline"]["general_stage"]["ms_per_stage"] > 0
