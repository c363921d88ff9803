"""libcgx's CGX_COMM_RCCL transport with MORE THAN ONE rank on a one-GPU box.

RCCL itself refuses two ranks on one device, so until round 4 the RCCL path of the library (prefold kernel -> in-place
ncclAllGather of equal segments -> K3; the scalar all-gather of the verification phase; the wire-up through ncclGetUniqueId /
ncclCommInitRank over cgsolver's pipes) had only ever run with one rank.  tests/fake_rccl/fake_rccl.cc is a TEST DOUBLE with
the semantics the NCCL API documents for the entry points libcgx binds, for ranks that are separate processes sharing one
GPU.  It is built here and put in front of the loader's search path (LD_LIBRARY_PATH; libcgx binds librccl.so.1 with dlopen),
for the cgsolver CLI only: a Python process that imported torch already holds the real RCCL.  What this shows: libcgx's use of
the API (counts, in-place offsets, buffer sizes, call order, every rank taking the same break) is right for P > 1.  What it
cannot show: RCCL's own behaviour and latency over xGMI (code/MPI/cg.cc:106,117,135-136 <-> the one all-gather)."""
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "conjugate-gradient_amd", "cgsolver")


@pytest.fixture(scope="module")
def fake_rccl_dir(tmp_path_factory):
    d = tmp_path_factory.mktemp("fake_rccl")
    out = d / "librccl.so.1"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-w", "-O2", "-std=c++17", "-fPIC", "-shared", "-I/opt/rocm/include",
                           os.path.join(ROOT, "tests", "fake_rccl", "fake_rccl.cc"), "-o", str(out), "-Wl,-soname,librccl.so.1"],
                          cwd=str(d))
    return str(d)


def run_cli(fake_dir, args, timeout=300):
    env = dict(os.environ, LD_LIBRARY_PATH=fake_dir + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""))
    return subprocess.run([EXE] + args, capture_output=True, text=True, timeout=timeout, env=env)


def parse_step(stdout):
    m = re.search(r"\[STEP (\d+)\] residual = (\S+), \|\|x\|\| = (\S+), \|\|Ax - b\|\|/\|\|b\|\| = (\S+)", stdout)
    assert m, stdout
    return int(m.group(1)), float(m.group(2)), float(m.group(3)), float(m.group(4))


def test_cgsolver_rccl_transport_three_ranks(gpu_pkg, fake_rccl_dir, tmp_path):
    """`cgsolver 2048 OUT 200 --gpus 3 --transport rccl`: the reference's own numbers for this run."""
    out = tmp_path / "strong.txt"
    r = run_cli(fake_rccl_dir, ["2048", str(out), "200", "--gpus", "3", "--same-device", "--transport", "rccl", "--stats"])
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stderr.count("fake_rccl: rank") == 3 and "fake_rccl: rank 2 of 3 wired" in r.stderr      # the double was what ran
    assert "[STEP 200] residual = 1.331819e-05, ||x|| = 8.808702e+07" in r.stdout
    assert r.stdout.count("[STEP") == 1 and out.read_text().strip().startswith("2048,3,")
    assert "gpus=3" in r.stderr


@pytest.mark.parametrize("n,p,iters", [(1001, 4, 50), (1000, 2, 40), (37, 4, 10), (9, 4, 3), (5000, 4, 30)])   # at most 4 ranks: the box allows 6 processes on its GPU
def test_cgsolver_rccl_transport_partitions(gpu_pkg, oracle, fake_rccl_dir, tmp_path, n, p, iters):
    """Uneven partitions (the last rank owns the remainder, cg.cc:255-266), fewer rows per rank than a chunk, fewer rows than ranks:
    residual, ||x|| and ||Ax-b||/||b|| as printed (seven digits) against the oracle with the same number of row blocks."""
    out = tmp_path / "o.txt"
    r = run_cli(fake_rccl_dir, [str(n), str(out), str(iters), "--gpus", str(p), "--same-device", "--transport", "rccl"])
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stderr.count("fake_rccl: rank") == p
    k, res, xn, rel = parse_step(r.stdout)
    _, ro = oracle.solve_lap2d(n, iters, 1e-10, p)
    assert k == ro["iterations"] == iters
    assert abs(xn - ro["x_norm"]) <= 2e-6 * ro["x_norm"]
    if n >= 16:      # on a 3 x 3 system the residual after two iterations is rounding noise
        assert abs(res - ro["residual_prev"]) <= 2e-6 * ro["residual_prev"] and abs(rel - ro["rel_residual"]) <= 2e-6 * ro["rel_residual"]
    assert out.read_text().strip().startswith("%d,%d," % (n, p))


def test_cgsolver_rccl_transport_runs_to_convergence(gpu_pkg, oracle, fake_rccl_dir, tmp_path):
    """Every rank must take the same `break` (cg.cc:120-121): a run to convergence on three ranks ends, with the oracle's k."""
    out = tmp_path / "o.txt"
    r = run_cli(fake_rccl_dir, ["1024", str(out), "--gpus", "3", "--same-device", "--transport", "rccl"])
    assert r.returncode == 0, r.stdout + r.stderr
    k, res, xn, rel = parse_step(r.stdout)
    _, ro = oracle.solve_lap2d(1024, 1024, 1e-10, 3)
    assert ro["converged"] and abs(k - ro["iterations"]) <= 0.15 * ro["iterations"]
    assert rel < 1e-11 and abs(xn - ro["x_norm"]) <= 2e-6 * ro["x_norm"]


def test_cgsolver_rccl_transport_config4_n32768_500_iterations(gpu_pkg, fake_rccl_dir, tmp_path):
    """BASELINE.json configs[3] shape (N = 32768, 500 iterations, row blocks) with 4 processes over the RCCL transport: the
    reference's recorded line for this run (residual 2.788823e+01, ||x|| 8.873404e+10, 6.7433e-07)."""
    out = tmp_path / "o.txt"
    r = run_cli(fake_rccl_dir, ["32768", str(out), "500", "--gpus", "4", "--same-device", "--transport", "rccl"], timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "[STEP 500] residual = 2.788823e+01, ||x|| = 8.873404e+10, ||Ax - b||/||b|| = 6.7433" in r.stdout, r.stdout
    assert out.read_text().strip().startswith("32768,4,")


def test_a_rank_that_never_reaches_ncclCommInitRank_ends_the_job(gpu_pkg, fake_rccl_dir, tmp_path):
    """--transport rccl by name, with a dead peer: the job ends with exit code 1 inside the bound, nobody is left behind."""
    out = tmp_path / "o.txt"
    r = run_cli(fake_rccl_dir, ["512", str(out), "20", "--gpus", "2", "--same-device", "--transport", "rccl", "--wireup-timeout", "5",
                                "--test-hang-stage", "ncclCommInitRank:1"], timeout=120)
    assert r.returncode == 1 and "wire-up stage 'ncclCommInitRank' did not finish within 5 s" in r.stderr
    assert not out.exists()
