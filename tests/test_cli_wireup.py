"""`cgsolver --gpus P` must never hang on a peer that does not answer: every stage of the wire-up that can
block on another rank runs under --wireup-timeout; on expiry the rank prints one line and leaves with exit code 1, rank 0
ends and reaps the others.  Runs without a GPU: the device probe is the first such stage, and the test-only argument
--test-hang-stage makes one rank never come back from it (an explicit argument: nothing in the environment arms it).  Reference behaviour this replaces: MPI_Init under srun either
returns or the scheduler kills the job (code/MPI/cg_main.cc:15-20,67)."""
import os
import subprocess
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "conjugate-gradient_amd", "cgsolver")


def no_gpu_env(**kw):
    return dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1", CUDA_VISIBLE_DEVICES="-1", **kw)


def run(args, env, limit=90):
    assert os.path.exists(EXE), "cgsolver not built (run __graft_entry__.build())"
    t0 = time.time()
    p = subprocess.Popen([EXE] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, start_new_session=True)
    out, err = p.communicate(timeout=limit)
    # nothing of the job may be left running
    left = subprocess.run(["ps", "-o", "stat=", "-g", str(p.pid)], capture_output=True, text=True).stdout.split()
    assert all(st.startswith("Z") for st in left), left
    return p.returncode, out, err, time.time() - t0


def test_a_rank_that_never_answers_ends_the_job_within_the_bound(tmp_path):
    out = tmp_path / "o.txt"
    rc, so, se, secs = run(["64", str(out), "5", "--gpus", "2", "--wireup-timeout", "2", "--test-hang-stage", "device probe:1"], no_gpu_env())
    assert rc == 1, (so, se)
    assert secs < 30
    assert "wire-up stage 'device probe' did not finish within 2 s" in se
    assert not out.exists()                        # no CSV line from a job that never solved


def test_rank0_stuck_is_bounded_too(tmp_path):
    out = tmp_path / "o.txt"
    rc, so, se, secs = run(["64", str(out), "5", "--gpus", "3", "--test-hang-stage", "device probe:0"], no_gpu_env(CG_WIREUP_TIMEOUT="1.5"))
    assert rc == 1 and secs < 30 and "(rank 0): wire-up stage 'device probe' did not finish within 1.5 s" in se
    assert not out.exists()


def test_without_a_device_every_rank_leaves_at_once(tmp_path):
    out = tmp_path / "o.txt"
    rc, so, se, secs = run(["64", str(out), "5", "--gpus", "2"], no_gpu_env())
    assert rc == 1 and secs < 30
    assert "not every rank has a usable MI355X" in se and "did not finish" not in se
    assert not out.exists()


def test_the_environment_cannot_arm_the_hang_hook(tmp_path):
    """Round 3 read the hook from CG_TEST_HANG_STAGE: a variable in a user's environment must not change what cgsolver does."""
    out = tmp_path / "o.txt"
    rc, so, se, secs = run(["64", str(out), "5", "--gpus", "2", "--wireup-timeout", "20"], no_gpu_env(CG_TEST_HANG_STAGE="device probe:1"))
    assert rc == 1 and secs < 15 and "did not finish" not in se and "not every rank has a usable MI355X" in se
