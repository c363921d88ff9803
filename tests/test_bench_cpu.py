"""bench.py's host logic, without a GPU: rank 0 prints exactly one JSON line on every exit path (no device, wrong
launch, watchdog, a dead peer under the launcher), stage time-outs abandon a hung call, and the static half of the
line follows the driver's contract."""
import json
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
sys.path.insert(0, ROOT)


def one_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


def no_gpu_env():
    # hide any device, so that the test means the same thing on the GPU box
    return dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1", CUDA_VISIBLE_DEVICES="-1")


def test_failure_line_without_a_device():
    r = subprocess.run([sys.executable, BENCH, "--steps", "20", "--warmup", "5"], capture_output=True, text=True, timeout=300,
                       env=no_gpu_env())
    d = one_line(r.stdout)
    assert r.returncode == 1
    assert d["value"] is None and d["ms_per_step"] is None and d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5
    assert d["metric"] == "cg_iterations_per_sec" and d["unit"] == "iterations/s" and d["dtype"] == "f64"
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "N=32768" in d["config"]["workload"] and "configs[2]" in d["config"]["workload"]
    assert d["error"]["kind"] == "AssertionError" and "no CPU path" in d["error"]["message"]


def test_failure_line_when_launched_without_the_launcher():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4"], capture_output=True, text=True, timeout=120, env=no_gpu_env())
    d = one_line(r.stdout)
    assert r.returncode == 2 and d["value"] is None and d["n_gpus"] == 4
    assert d["error"]["kind"] == "launch" and "--nproc-per-node 4" in d["error"]["message"]
    assert "configs[3]" in d["config"]["workload"] and d["config"]["parallelism"] == "rowblock4"


def test_watchdog_line():
    t0 = time.time()
    r = subprocess.run([sys.executable, BENCH, "--watchdog", "0.05"], capture_output=True, text=True, timeout=120, env=no_gpu_env())
    d = one_line(r.stdout)
    assert r.returncode == 3 and d["value"] is None and d["error"]["kind"] == "watchdog"
    assert time.time() - t0 < 60


def test_weak_mode_sizes_follow_the_reference_rule():
    import bench

    class A:
        n, mode = 0, "weak"
    assert [bench.problem_size(A, p) for p in (1, 2, 4, 8)] == [16384, 23170, 32768, 46340]   # code/MPI/cg.run:22-44
    A.mode = "strong"
    assert bench.problem_size(A, 8) == 32768


def test_call_with_timeout():
    import bench
    assert bench.call_with_timeout(lambda: 7, 5.0, "quick") == 7
    with pytest.raises(ValueError):
        bench.call_with_timeout(lambda: (_ for _ in ()).throw(ValueError("x")), 5.0, "raises")
    t0 = time.time()
    with pytest.raises(TimeoutError):
        bench.call_with_timeout(lambda: time.sleep(30), 0.2, "hangs")
    assert time.time() - t0 < 5.0


def test_rank0_prints_a_line_when_a_peer_dies(tmp_path):
    """Two ranks under torch.distributed.run on CPU: rank 1 dies at once (no device), the launcher sends SIGTERM to
    rank 0, whose waiter thread still gets the line out -- or rank 0 fails on its own for the same reason and prints it
    itself.  Either way: exactly one line, value null."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29631", BENCH, "--gpus", "2", "--steps", "5", "--warmup", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(no_gpu_env(), MASTER_ADDR="127.0.0.1"))
    d = one_line(r.stdout)
    assert r.returncode != 0 and d["value"] is None and d["n_gpus"] == 2 and "error" in d
