"""world_size > 1 on CPU (gloo): the row-block exchange protocol of the multi-GPU path, one process per rank."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_world(world, n, max_iter, tmp_path, port):
    out = tmp_path / ("verdict_%d_%d.json" % (world, n))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "dist_worker.py"), str(n), str(max_iter), str(out)]
    env = dict(os.environ, OMP_NUM_THREADS="1", MASTER_ADDR="127.0.0.1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return json.load(open(out))


@pytest.mark.parametrize("world,n,max_iter,port", [(2, 512, 60, 29611), (2, 257, 400, 29612), (3, 200, 50, 29613),
                                                   (2, 2500, 40, 29614)])          # three 512-row chunks per rank
def test_rowblock_protocol_over_gloo(tmp_path, world, n, max_iter, port):
    v = run_world(world, n, max_iter, tmp_path, port)
    assert v["ranks_agree"], v                       # same break decision, bit-identical rsnew on every rank
    assert v["dx"] < 1e-13, v
    if v["converged"]:
        # sqrt(rsold) at exit is rounding noise around 1e-10: r.r is summed over all rows here, per rank in the oracle
        assert abs(v["k"] - v["k_oracle"]) <= 2 and v["residual_rel"] < 1e-2, v
    else:
        assert v["k"] == v["k_oracle"] and v["residual_rel"] < 1e-9, v
    assert sum(v["counts"]) == n
