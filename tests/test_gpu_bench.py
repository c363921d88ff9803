"""bench.py on the GPU box: the line's contract on a short window (the driver runs --steps 20 --warmup 5), the
launcher paths one GPU can rehearse (RCCL with one rank under torch.distributed.run; two ranks over the IPC mailboxes
with a gloo control plane), and the promise that rank 0 prints a line whatever happens."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def one_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


def torchrun(world, port, args, env=None, timeout=420):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", str(world)] + args
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout,
                          env=dict(os.environ, OMP_NUM_THREADS="1", MASTER_ADDR="127.0.0.1", **(env or {})))


def test_bench_line_contract(gpu_pkg):
    """One JSON line with the driver's keys plus roofline and cpu_baseline (small size so that it takes seconds)."""
    r = subprocess.run([sys.executable, BENCH, "--steps", "30", "--warmup", "5", "--matrix-size", "4096",
                        "--cpu-baseline-iters", "3"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = one_line(r.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 30 and d["warmup"] == 5 and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert d["value"] > 0 and abs(d["ms_per_step"] * d["value"] - 1000.0) < 1e-6 * 1000
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and rf["unit"] == "GB/s" and 0 < rf["frac"] < 1.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and "sample" in cb
    ob = cb["openblas_gemv_1thread"]            # context: the loop's GEMV through the BLAS family the reference linked, one thread
    assert ob.get("skipped") or (ob["cores"] == 1 and ob["value"] > 0 and ob["gemv_GBs"] > 0)
    assert "error" not in d
    # roofline.traffic: by default counters of THIS box (two rocprofv3 --pmc child passes of the same workload), per launch
    if rf["traffic_source"].startswith("LIVE"):
        assert 0.9 < rf["traffic_over_algorithmic"] < 1.2 and abs(rf["traffic"] - rf["traffic_over_algorithmic"] * rf["bytes_per_launch"]) < 1.0
    else:
        # a box that cannot start the profiler (no rocprofv3, no counter access): the line says why and falls back
        assert rf["traffic_live_note"] and rf["traffic"] is None
    assert rf["traffic_committed"] is None          # no committed row for N = 4096
    # the dense-data leg: the same K1 on a matrix in which every element is a different number, same window
    dr = rf["dense_random"]
    assert dr["launches_timed"] == rf["launches_timed"] and dr["median_launch_ms"] > 0
    assert abs(dr["time_ratio_to_generated_matrix"] - dr["median_launch_ms"] / rf["median_launch_ms"]) < 1e-12
    assert abs(dr["frac"] - rf["bytes_per_launch"] / (dr["median_launch_ms"] * 1e-3) / 1e9 / 8000.0) < 1e-12


def test_bench_short_window_statistics(gpu_pkg, oracle):
    """The driver's window (20 timed steps after 5): every launch but the first is a sample, the median is what the
    fraction is computed from, a K1 launch never outlasts the step it is part of, and the run is the reference's
    recurrence (residual after 25 iterations against the oracle)."""
    n = 8192
    r = subprocess.run([sys.executable, BENCH, "--steps", "20", "--warmup", "5", "--matrix-size", str(n), "--no-cpu-baseline", "--no-live-pmc"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = one_line(r.stdout)
    rf = d["roofline"]
    assert rf["launches_timed"] == 19 and rf["launches_discarded"] == 1 and rf["consistency"] == "ok"
    assert rf["min_launch_ms"] <= rf["median_launch_ms"] <= rf["max_launch_ms"]
    assert rf["median_launch_ms"] <= d["ms_per_step"]
    assert abs(rf["achieved"] - rf["bytes_per_launch"] / (rf["median_launch_ms"] * 1e-3) / 1e9) < 1e-6 * rf["achieved"]
    assert rf["bytes_per_launch"] == 8.0 * (n * n + n + n)
    assert "traffic_source" in rf and "NOT a counter of this run" in rf["traffic_source"]
    assert len(d["k1_samples_ms"]) == 19
    assert rf["median_launch_ms"] < d["device_window_ms_per_step"] <= d["ms_per_step"]      # K1 < device window <= host window
    w = d["solve_window"]
    assert w["iterations"] == 20 and abs(d["solve_window_iterations_per_s"] - 20 / w["seconds_solve"]) < 1e-9 * d["value"]
    assert d["solve_window_iterations_per_s"] < d["value"]          # the window also holds set-up and two more GEMVs
    assert d["iterations_done"] == 25
    _, ro = oracle.solve_lap2d(n, 25, 0.0, 1)
    assert abs(d["residual_after_run"] - ro["residual_prev"]) <= 1e-6 * ro["residual_prev"]
    c = d["config"]
    assert c["transport"] == "self" and c["ranks_seen"] == 1 and c["distinct_gpus"] == 1 and c["rccl_nranks"] is None
    assert c["k1_plan"]["variant"] == 1 and c["k1_plan"]["split"] == 1 and c["k1_plan"]["grid"] == n // c["k1_plan"]["R"]
    assert "update_kernel" not in d          # one GPU: K3 is not event-timed unless asked (--profile-update)


def test_bench_rccl_one_rank_under_the_launcher(gpu_pkg, oracle):
    """--transport rccl with one rank under torch.distributed.run: ncclGetUniqueId, the broadcast of the id,
    ncclCommInitRank, one ncclAllGather per iteration on the library's stream, ncclCommCount in the line.  RCCL refuses
    two ranks on one device, so one rank is what a one-GPU box can run of this path."""
    n = 4096
    r = torchrun(1, 29721, ["--steps", "30", "--warmup", "5", "--matrix-size", str(n), "--transport", "rccl", "--no-cpu-baseline"])
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    d = one_line(r.stdout)
    c = d["config"]
    assert c["transport"] == "rccl" and c["rccl_nranks"] == 1 and c["process_group_ranks"] == 1
    assert c["transport_ranks_wired"] == [1] and c["ranks_seen"] == 1
    assert d["value"] > 0 and d["iterations_done"] == 35 and d["roofline"]["consistency"] == "ok"
    _, ro = oracle.solve_lap2d(n, 35, 0.0, 1)
    assert abs(d["residual_after_run"] - ro["residual_prev"]) <= 1e-6 * ro["residual_prev"]


def test_bench_two_ranks_over_the_mailboxes(gpu_pkg, oracle):
    """bench.py's world > 1 flow end to end on one GPU: gloo control plane, both ranks on device 0, transport chosen
    by the calibration among the two mailbox forms (RCCL drops out: two ranks on one device), per-rank K1 statistics."""
    n = 4096
    r = torchrun(2, 29722, ["--steps", "30", "--warmup", "5", "--matrix-size", str(n)], env={"CGX_BENCH_BACKEND": "gloo"})
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    d = one_line(r.stdout)
    c = d["config"]
    assert d["n_gpus"] == 2 and c["transport"] in ("p2p", "p2p-sep") and c["ranks_seen"] == 2
    assert set(c["transport_calibration_ms_per_iteration"]) == {"p2p-tag", "p2p", "p2p-sep"}
    assert c["transport_ranks_wired"] == [2, 2] and c["distinct_gpus"] == 1 and c["process_group_ranks"] == 2
    assert any("rccl" in note for note in c["transport_notes"])
    assert [q["rank"] for q in d["k1_per_rank"]] == [0, 1] and [q["rows"] for q in d["k1_per_rank"]] == [2048, 2048]
    assert d["k1_slowest_rank"] in (0, 1)
    assert all(q["min_ms"] <= q["median_ms"] <= q["max_ms"] for q in d["k1_per_rank"])
    # the update kernel of the timed iterations (on several ranks: the kernel that holds the wait for the peers)
    uk = d["update_kernel"]
    assert ("k_update_xr_p2p" in uk["kernel"]) == (c["transport"] in ("p2p", "p2p-tag")) and [q["rank"] for q in uk["per_rank"]] == [0, 1]
    assert all(q["launches_timed"] >= 5 and 0 < q["min_ms"] <= q["median_ms"] <= q["max_ms"] < d["ms_per_step"] for q in uk["per_rank"])
    _, ro = oracle.solve_lap2d(n, 35, 0.0, 2)
    assert abs(d["residual_after_run"] - ro["residual_prev"]) <= 1e-6 * ro["residual_prev"]


def test_bench_prints_a_failure_line_when_no_transport_works(gpu_pkg):
    """Two ranks on one device with RCCL as the only allowed transport: ncclCommInitRank refuses, nothing can be timed,
    and rank 0 still prints ONE line -- value null, an error object that says why."""
    r = torchrun(2, 29723, ["--steps", "10", "--warmup", "2", "--matrix-size", "1024", "--transport", "rccl", "--wireup-timeout", "60"],
                 env={"CGX_BENCH_BACKEND": "gloo"})
    d = one_line(r.stdout)
    assert d["value"] is None and d["n_gpus"] == 2 and d["steps"] == 10
    assert "no transport produced a result" in d["error"]["message"] and "rccl" in d["error"]["message"]
    assert r.returncode != 0


def test_bench_watchdog_prints_a_failure_line(gpu_pkg):
    """A run that cannot finish inside the watchdog still ends with one parseable line."""
    r = subprocess.run([sys.executable, BENCH, "--steps", "500", "--warmup", "100", "--watchdog", "0.2", "--no-cpu-baseline", "--no-live-pmc"],
                       capture_output=True, text=True, timeout=300)
    d = one_line(r.stdout)
    assert d["value"] is None and d["error"]["kind"] == "watchdog"
    assert r.returncode == 3


def test_bench_auto_transport_one_rank_under_the_launcher(gpu_pkg):
    """The driver's multi-GPU launch shape with one rank: nccl control plane (CUDA control tensors), all three transports
    built, pre-warmed and calibrated against each other, the fastest one timed."""
    r = torchrun(1, 29724, ["--steps", "20", "--warmup", "5", "--matrix-size", "8192", "--no-cpu-baseline"])
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    d = one_line(r.stdout)
    c = d["config"]
    assert set(c["transport_calibration_ms_per_iteration"]) == {"p2p-tag", "p2p", "p2p-sep", "rccl"}
    # the tagged-word form is calibrated for the record but auto never lets it carry the timed run (no multi-GPU run has shown
    # its 16-byte stores untorn over xGMI yet): the fastest of the others does
    assert c["transport"] in ("p2p", "p2p-sep", "rccl")
    assert c["transport_notes"] == ["p2p-tag calibrated for the record only (auto never selects it; --transport p2p-tag runs it)"]
    others = {t: v for t, v in c["transport_calibration_ms_per_iteration"].items() if t != "p2p-tag"}
    assert c["transport"] == min(others, key=others.get)
    assert c["rccl_nranks"] == 1          # from the RCCL candidate of the calibration, whichever transport carried the run
    assert d["value"] > 0 and d["roofline"]["consistency"] == "ok" and d["roofline"]["launches_timed"] == 19


def test_bench_four_ranks_uneven_partition(gpu_pkg, oracle):
    """Four ranks sharing the GPU (gloo control plane), N = 10001: the last rank owns one row more (cg.cc:255-266), so the
    per-rank rows, bytes and K1 statistics differ between ranks and the roofline is taken from the rank furthest below it."""
    n = 10001
    r = torchrun(4, 29725, ["--steps", "24", "--warmup", "4", "--matrix-size", str(n)], env={"CGX_BENCH_BACKEND": "gloo"}, timeout=600)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    d = one_line(r.stdout)
    c = d["config"]
    assert d["n_gpus"] == 4 and c["ranks_seen"] == 4 and c["transport_ranks_wired"] == [4, 4, 4, 4] and c["distinct_gpus"] == 1
    assert [q["rows"] for q in d["k1_per_rank"]] == [2500, 2500, 2500, 2501]
    assert [q["bytes_per_launch"] for q in d["k1_per_rank"]] == [8.0 * (rows * n + n + rows) for rows in (2500, 2500, 2500, 2501)]
    assert all(q["launches_timed"] >= 5 and q["launches_discarded"] == 1 for q in d["k1_per_rank"])     # every 4th of 24
    rf = d["roofline"]
    assert rf["rank"] in (0, 1, 2, 3) and rf["bytes_per_launch"] == d["k1_per_rank"][rf["rank"]]["bytes_per_launch"]
    assert rf["achieved"] == min(q["GBs"] for q in d["k1_per_rank"])
    assert d["iterations_done"] == 28 and d["solve_window"]["iterations"] == 24
    _, ro = oracle.solve_lap2d(n, 28, 0.0, 4)
    assert abs(d["residual_after_run"] - ro["residual_prev"]) <= 1e-6 * ro["residual_prev"]


def test_bench_survives_a_wireup_stage_that_never_returns(gpu_pkg):
    """A hung ncclCommInitRank (simulated: the stage sleeps for an hour) is abandoned after --wireup-timeout, RCCL is
    dropped with a note, the mailbox transports carry the run, and the process still ends (os._exit past the stuck thread)."""
    r = torchrun(1, 29726, ["--steps", "20", "--warmup", "5", "--matrix-size", "4096", "--no-cpu-baseline", "--wireup-timeout", "3", "--test-hang", "rccl:ncclCommInitRank"],
                 timeout=300)
    d = one_line(r.stdout)
    c = d["config"]
    assert d["value"] > 0 and c["transport"] in ("p2p", "p2p-sep")
    assert set(c["transport_calibration_ms_per_iteration"]) == {"p2p-tag", "p2p", "p2p-sep"}
    assert any("rccl" in note and "did not finish within 3 s" in note for note in c["transport_notes"])
    assert r.returncode == 0, r.stderr[-2000:]


def test_bench_two_ranks_n32768_default_column_split(gpu_pkg, oracle):
    """The strong-scaling workload itself with two ranks sharing the GPU: each rank's 16384 x 32768 shard streams from HBM,
    so the fused P2P transport runs K1 with its default XCD-affine column split (2 pieces at P = 2) and the update adds the
    pieces on the fly.  Residual after 25 iterations against the oracle's on-the-fly twin with the same partition."""
    n = 32768
    r = torchrun(2, 29727, ["--steps", "20", "--warmup", "5", "--transport", "p2p", "--no-solve-window", "--cpu-baseline-iters", "3"],
                 env={"CGX_BENCH_BACKEND": "gloo"}, timeout=600)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    d = one_line(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["transport"] == "p2p" and d["config"]["n"] == n and d["iterations_done"] == 25
    assert [q["rows"] for q in d["k1_per_rank"]] == [16384, 16384]
    plans = d["config"]["k1_plan"]
    assert len(plans) == 2 and all(pl["light"] == 1 and pl["split"] >= 2 and pl["R"] == 8 and pl["U"] == 2 for pl in plans)
    cb = d["cpu_baseline"]                 # north_star: the CPU path timed in the same run, on multi-GPU lines too
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and "N=32768" in cb["sample"]
    assert d["roofline"]["traffic"] is not None and d["roofline"]["consistency"] == "ok"
    _, ro = oracle.solve_lap2d_banded(n, 25, 0.0, 2)
    assert abs(d["residual_after_run"] - ro["residual_prev"]) <= 1e-6 * ro["residual_prev"]


def test_bench_self_launch_two_ranks(gpu_pkg, oracle):
    """`python3 bench.py --gpus 2` with NO launcher around it: the parent starts the two ranks itself (fresh children before
    any GPU call), relays rank 0's one line and the child's exit code (cg_main.cc:15-20, cg.run:15-19)."""
    n = 4096
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--matrix-size", str(n), "--steps", "30", "--warmup", "5",
                        "--cpu-baseline-iters", "3"], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, CGX_BENCH_BACKEND="gloo"))
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    d = one_line(r.stdout)
    c = d["config"]
    assert d["n_gpus"] == 2 and c["ranks_seen"] == 2 and c["process_group_ranks"] == 2 and c["transport"] in ("p2p-tag", "p2p", "p2p-sep")
    assert d["value"] > 0 and d["iterations_done"] == 35 and len(c["k1_plan"]) == 2 and "cpu_baseline" in d
    _, ro = oracle.solve_lap2d(n, 35, 0.0, 2)
    assert abs(d["residual_after_run"] - ro["residual_prev"]) <= 1e-6 * ro["residual_prev"]


def test_bench_self_launch_refuses_more_ranks_than_gpus(gpu_pkg):
    """Two ranks over RCCL's control plane on a one-GPU box: every rank leaves before any rendezvous, rank 0's failure line
    says what is missing, the parent relays it and a non-zero exit code -- no hang."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer GPUs than ranks")
    env = {k: v for k, v in os.environ.items() if k != "CGX_BENCH_BACKEND"}
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--matrix-size", "4096", "--steps", "10", "--warmup", "2"],
                       capture_output=True, text=True, timeout=300, env=env)
    d = one_line(r.stdout)
    assert r.returncode != 0 and d["value"] is None and d["n_gpus"] == 2
    assert "needs 2 MI355X, 1 visible" in d["error"]["message"]


def test_bench_update_kernel_timing_on_request(gpu_pkg):
    """--profile-update on one GPU: K3 of every iteration whose K1 is timed gets its own event pair; K1 + K3 stay below the step."""
    r = subprocess.run([sys.executable, BENCH, "--steps", "20", "--warmup", "5", "--matrix-size", "8192", "--no-cpu-baseline",
                        "--no-solve-window", "--profile-update", "--no-live-pmc"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = one_line(r.stdout)
    u = d["update_kernel"]["per_rank"][0]
    assert d["update_kernel"]["kernel"].startswith("k_update_xr (K3)") and u["launches_timed"] == 19
    assert 0 < u["min_ms"] <= u["median_ms"] <= u["max_ms"]
    assert d["roofline"]["median_launch_ms"] + u["median_ms"] <= d["ms_per_step"] and d["roofline"]["consistency"] == "ok"


def test_bench_self_launch_weak_mode_two_ranks(gpu_pkg, oracle):
    """configs[4] at P = 2 through the self-launch path: `python3 bench.py --gpus 2 --mode weak` picks N = floor(16384 sqrt 2) =
    23170 (code/MPI/cg.run:22-44), 11585 rows per rank, scaling "weak"; both ranks share the one GPU (gloo control plane).
    Residual after 25 iterations against the oracle's on-the-fly twin with the same partition."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--mode", "weak", "--steps", "20", "--warmup", "5", "--no-solve-window",
                        "--cpu-baseline-iters", "3"], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, CGX_BENCH_BACKEND="gloo"))
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    d = one_line(r.stdout)
    c = d["config"]
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and c["n"] == 23170 and "configs[4]" in c["workload"]
    assert [q["rows"] for q in d["k1_per_rank"]] == [11585, 11585] and c["ranks_seen"] == 2
    assert all(pl["split"] == 8 and pl["light"] == 1 for pl in c["k1_plan"])
    assert d["iterations_done"] == 25 and "N=23170" in d["cpu_baseline"]["sample"]
    _, ro = oracle.solve_lap2d_banded(23170, 25, 0.0, 2)
    assert abs(d["residual_after_run"] - ro["residual_prev"]) <= 1e-6 * ro["residual_prev"]


def test_bench_keeps_the_measurement_when_an_extra_hangs(gpu_pkg):
    """The CPU baseline (or a live counter pass) never comes back: the watchdog prints the line that was complete before it --
    value, roofline, everything measured -- with a note, and the exit code stays 0."""
    r = subprocess.run([sys.executable, BENCH, "--steps", "20", "--warmup", "5", "--matrix-size", "4096", "--no-live-pmc", "--watchdog", "18",
                        "--test-hang", "extra:cpu baseline"], capture_output=True, text=True, timeout=300)
    d = one_line(r.stdout)
    assert r.returncode == 0, r.stderr[-2000:]
    assert d["value"] > 0 and d["roofline"]["consistency"] == "ok" and "cpu_baseline" not in d and "error" not in d
    assert "cpu baseline" in d["watchdog_note"]


def test_bench_four_ranks_more_chunk_pairs_than_a_wave(gpu_pkg, oracle):
    """Four ranks sharing the GPU at N = 33000 (gloo control plane): 8250 rows per rank, 17 chunks per rank, 68 (peer, chunk)
    pairs -- more than the 64 lanes of one polling wave -- under the column-split K1 with a column count that is no multiple
    of anything.  (Four ranks is the most a test may start: the test process, the launcher and the ranks all hold the GPU, and
    the box allows six.)  Both fused forms are built and calibrated; residual after 25 iterations against the oracle's twin."""
    n = 33000
    r = torchrun(4, 29728, ["--steps", "20", "--warmup", "5", "--matrix-size", str(n), "--no-solve-window", "--cpu-baseline-iters", "3"],
                 env={"CGX_BENCH_BACKEND": "gloo"}, timeout=900)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    d = one_line(r.stdout)
    c = d["config"]
    assert d["n_gpus"] == 4 and c["ranks_seen"] == 4 and [q["rows"] for q in d["k1_per_rank"]] == [8250] * 4
    assert {"p2p-tag", "p2p"} <= set(c["transport_calibration_ms_per_iteration"]) and c["transport"] in ("p2p-tag", "p2p", "p2p-sep")
    assert all(pl["split"] == 8 for pl in c["k1_plan"]) and d["iterations_done"] == 25
    _, ro = oracle.solve_lap2d_banded(n, 25, 0.0, 4)
    assert abs(d["residual_after_run"] - ro["residual_prev"]) <= 1e-6 * ro["residual_prev"]
