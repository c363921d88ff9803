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


def test_self_launch_without_a_device_relays_rank0s_failure_line():
    """`python bench.py --gpus 4` with no launcher around it starts its own four ranks (fresh child processes, torch never
    imported in the parent), every rank fails for lack of a device, and the parent relays rank 0's ONE line and the
    child's exit code (code/MPI/cg_main.cc:15-20: the reference's ranks come from srun)."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--steps", "7", "--warmup", "2"], capture_output=True, text=True,
                       timeout=300, env=no_gpu_env())
    d = one_line(r.stdout)
    assert r.returncode != 0 and d["value"] is None and d["n_gpus"] == 4 and d["steps"] == 7 and d["warmup"] == 2
    # rank 0 either fails by itself (no device) or is stopped by the launcher because another rank failed first
    assert (d["error"]["kind"] == "AssertionError" and "no CPU path" in d["error"]["message"]) or d["error"]["kind"] == "signal"
    assert "configs[3]" in d["config"]["workload"] and d["config"]["parallelism"] == "rowblock4"
    assert "starting 4 ranks" in r.stderr and "--nproc-per-node 4" in r.stderr


def test_self_launch_is_skipped_under_a_launcher():
    """With RANK / WORLD_SIZE in the environment bench.py is a rank, never a launcher (the driver's N > 1 command)."""
    env = dict(no_gpu_env(), RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29655")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "5", "--warmup", "1"], capture_output=True, text=True,
                       timeout=300, env=env)
    d = one_line(r.stdout)
    assert d["value"] is None and d["n_gpus"] == 2 and "starting" not in r.stderr


def test_self_launch_parent_prints_a_line_when_the_ranks_print_none(tmp_path):
    """The ranks die without a word (here: the python the parent starts cannot run torch.distributed.run at all): the
    parent still prints one failure line and a non-zero exit code."""
    import bench
    args = type("A", (), {"gpus": 2, "steps": 5, "warmup": 1, "mode": "strong", "n": 0})()
    out = []
    old_exe, old_argv = sys.executable, sys.argv
    sys.executable, sys.argv = "/bin/false", ["bench.py", "--gpus", "2"]
    try:
        rc = bench.self_launch(args, lambda t: out.append(t) or True,
                               lambda kind, msg: dict(bench.base_line(args, 2, 32768), error={"kind": kind, "message": msg}), {})
    finally:
        sys.executable, sys.argv = old_exe, old_argv
    assert rc != 0 and len(out) == 1
    d = json.loads(out[0])
    assert d["value"] is None and d["error"]["kind"] == "launch" and "without a result line" in d["error"]["message"]


def test_self_launch_watchdog_stops_the_ranks():
    """The parent's watchdog (the backstop behind the ranks' own) ends the launcher child and its whole process group."""
    import bench
    proc = subprocess.Popen([sys.executable, "-c", "import subprocess, sys, time; "
                             "subprocess.Popen([sys.executable, '-c', 'import time; time.sleep(600)']); time.sleep(600)"],
                            start_new_session=True)
    time.sleep(1.0)
    kids = subprocess.run(["pgrep", "-g", str(proc.pid)], capture_output=True, text=True).stdout.split()
    assert len(kids) == 2
    bench.LAUNCHED["proc"] = proc
    try:
        t0 = time.time()
        bench.stop_launched()
        assert proc.poll() is not None and time.time() - t0 < 20
        time.sleep(0.5)
        # nothing of the group is left running (an orphan nobody reaps may linger as a zombie in a container)
        left = subprocess.run(["ps", "-o", "stat=", "-g", str(proc.pid)], capture_output=True, text=True).stdout.split()
        assert all(st.startswith("Z") for st in left), left
    finally:
        bench.LAUNCHED["proc"] = None


def test_cpu_baseline_sample_is_bounded_at_every_size():
    import bench
    assert bench.cpu_baseline_iters(32768, 20) == 20 and bench.cpu_baseline_iters(4096, 20) == 20
    assert bench.cpu_baseline_iters(46340, 20) == 10          # configs[4], P = 8: half the bodies at twice the cost
    assert bench.cpu_baseline_iters(131072, 20) == 3


def test_traffic_row_must_match_the_plan_that_ran(tmp_path, monkeypatch):
    import bench
    doc = {"rows": [{"n": 32768, "nranks": 8, "plan": {"R": 8, "U": 2, "light": 1, "split": 8}, "hbm_bytes_per_launch": 123.0},
                    {"n": 32768, "nranks": 1, "plan": {"R": 8, "U": 2, "light": 0, "split": 1}, "hbm_bytes_per_launch": 456.0}]}
    (tmp_path / "profiles").mkdir()
    (tmp_path / "profiles" / "k1_hbm_traffic.json").write_text(json.dumps(doc))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    plan8 = {"variant": 1, "R": 8, "U": 2, "waves": 4, "light": 1, "split": 8, "grid": 4096, "ncols": 32768}
    assert bench.pmc_traffic(32768, 8, plan8) == 123.0
    assert bench.pmc_traffic(32768, 8, dict(plan8, split=4)) is None          # another kernel form: no evidence
    assert bench.pmc_traffic(32768, 8, dict(plan8, R=4, U=4, split=1)) is None
    assert bench.pmc_traffic(32768, 1, dict(plan8, light=0, split=1)) == 456.0
    assert bench.pmc_traffic(16384, 1, dict(plan8, light=0, split=1)) is None
    assert bench.pmc_traffic(32768, 8, None) is None


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


def test_committed_traffic_rows_name_the_kernel_form_they_were_collected_on():
    """profiles/k1_hbm_traffic.json (written by tools/pmc_k1.sh alone): every row carries the K1 plan, the plan agrees with the
    kernel name rocprofv3 recorded (template arguments R, U, ..., light), the traffic is the guide's formula of the two
    counters, and the headline shapes are all there."""
    import re
    doc = json.load(open(os.path.join(ROOT, "profiles", "k1_hbm_traffic.json")))
    assert "rows_unsplit_shards" not in doc
    seen = set()
    for r in doc["rows"]:
        pl = r["plan"]
        m = re.search(r"k_gemv_colsplit<(\d+), (\d+), \d+, 1(?:, (true|false))?(?:, (?:true|false))?>", r["kernel"])
        assert m and int(m.group(1)) == pl["R"] and int(m.group(2)) == pl["U"] and (m.group(3) == "true") == bool(pl["light"]), r["kernel"]
        rows = r["n"] // r["nranks"]
        assert r["algorithmic_bytes_per_launch"] == 8.0 * (rows * r["n"] + r["n"] + rows)
        assert abs(r["hbm_bytes_per_launch"] - (r["FETCH_SIZE_KB_mean"] * 2048 + r["WRITE_SIZE_KB_mean"] * 1024)) < 1.0
        assert 1.0 <= r["traffic_over_algorithmic"] < 1.01 and r["launches_sampled"] >= 100
        seen.add((r["n"], r["nranks"]))
    assert {(32768, 1), (32768, 2), (32768, 4), (32768, 8), (16384, 1)} <= seen


def test_a_counter_pass_in_flight_is_ended_with_its_whole_process_group(tmp_path):
    """the rocprofv3 child of live_pmc_traffic (and the bench.py under it, which holds a matrix on the GPU) runs in
    a process group of its own that the stop paths (timeout, watchdog, SIGTERM) know: stop_launched() ends the group, the
    grandchild included."""
    import bench
    marker = tmp_path / "grandchild.pid"
    # a child that starts a grandchild and waits: the shape of `rocprofv3 -- python bench.py`
    script = "import subprocess, sys, time\np = subprocess.Popen([sys.executable, '-c', 'import time; time.sleep(600)'])\nopen(%r, 'w').write(str(p.pid))\ntime.sleep(600)\n" % str(marker)
    proc = subprocess.Popen([sys.executable, "-c", script], start_new_session=True)
    bench.LAUNCHED["extra"] = proc
    try:
        for _ in range(100):
            if marker.exists() and marker.read_text():
                break
            time.sleep(0.05)
        grandchild = int(marker.read_text())
        t0 = time.time()
        bench.stop_launched()
        assert proc.poll() is not None and time.time() - t0 < 20
        for _ in range(100):
            try:
                os.kill(grandchild, 0)
            except ProcessLookupError:
                break
            # a zombie of a dead process still answers kill(0) until init reaps it: look at its state
            try:
                if open("/proc/%d/stat" % grandchild).read().split()[2] == "Z":
                    break
            except FileNotFoundError:
                break
            time.sleep(0.05)
        else:
            raise AssertionError("the grandchild of the counter pass survived stop_launched()")
    finally:
        bench.LAUNCHED["extra"] = None
        if proc.poll() is None:
            proc.kill()
