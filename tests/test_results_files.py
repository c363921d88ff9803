"""the files experiments/cg_mi355x.run leaves under results/ must go through the loaders of the
reference's plots.ipynb.  The loaders are restated here in a few lines each (what they do to a line, which tables they
index), citing the notebook's source lines (numbering of its first code cell); where the notebook itself is present
(/root/reference, this container only) its ALPHAS table is read from it and compared with the restatement."""
import json
import math
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RES = os.path.join(ROOT, "results")
NOTEBOOK = "/root/reference/plots.ipynb"

# plots.ipynb:5-6 -- serial fractions per matrix size; show_hard_MPI_results indexes it with every n of the file
ALPHA_KEYS = [1024, 1148, 2048, 2896, 4096, 5792, 8192, 11585, 16384]


def load_strong(path):
    """plots.ipynb:18-24: results[int(c0)] -> list of [int(c1), float(c2)], keys in first-seen order."""
    results = {}
    with open(path) as f:
        for line in f:
            cur = line.rstrip("\n").split(",")
            results.setdefault(int(cur[0]), []).append([int(cur[1]), float(cur[2])])
    return results


def load_weak(path):
    """plots.ipynb:68-76: three parallel lists points / psizes / times, consumed in blocks of ITEMS rows (:80-82)."""
    points, psizes, times = [], [], []
    with open(path) as f:
        for line in f:
            cur = line.rstrip("\n").split(",")
            points.append(int(cur[0]))
            psizes.append(int(cur[1]))
            times.append(float(cur[2]))
    return points, psizes, times


def load_cuda_t(path):
    """plots.ipynb:144-150: results[int(c1)] -> list of [int(c0), float(c2)] (BLOCK_WIDTH -> [NUM_THREADS, seconds])."""
    results = {}
    with open(path) as f:
        for line in f:
            cur = line.rstrip("\n").split(",")
            results.setdefault(int(cur[1]), []).append([int(cur[0]), float(cur[2])])
    return results


def no_blank_or_ragged_lines(path):
    for line in open(path):
        assert line.endswith("\n") and len(line.rstrip("\n").split(",")) == 3, (path, line)   # a blank line is a ValueError in int()


def test_alphas_table_is_the_notebooks():
    if not os.path.exists(NOTEBOOK):
        pytest.skip("reference notebook not present on this machine")
    src = "".join("".join(c["source"]) for c in json.load(open(NOTEBOOK))["cells"] if c["cell_type"] == "code")
    table = re.search(r"ALPHAS\s*=\s*\{(.*?)\}", src, re.S).group(1)
    assert [int(k) for k in re.findall(r"(\d+)\s*:", table)] == ALPHA_KEYS
    assert "ITEMS = 7" in src and "list(ALPHAS.values())[2:]" in src


@pytest.mark.parametrize("name", ["strong_scaling.txt", "strong_scaling_n32768.txt"])
def test_strong_scaling_files(name):
    path = os.path.join(RES, name)
    no_blank_or_ragged_lines(path)
    res = load_strong(path)
    assert res, name
    for n, rows in res.items():
        psizes = [r[0] for r in rows]
        times = np.array([r[1] for r in rows])
        assert psizes == sorted(set(psizes)) and psizes[0] == 1 and set(psizes) <= {1, 2, 4, 8}     # times[0] is the p=1 time (:34)
        assert times.dtype == np.float64 and np.all(times > 0) and np.all(np.isfinite(times[0] / times))
    if name == "strong_scaling.txt":
        assert set(res) <= set(ALPHA_KEYS), "a size outside ALPHAS is a KeyError at plots.ipynb:35"
        assert {1024, 2048, 4096, 8192} <= set(res)                                                     # the reference's own sizes
    else:
        assert set(res) == {32768}                                                                     # BASELINE.json configs[3]


@pytest.mark.parametrize("name,n0s", [("weak_scaling.txt", [1024, 1448, 2048]), ("weak_scaling_n16384.txt", [16384])])
def test_weak_scaling_files(name, n0s):
    path = os.path.join(RES, name)
    no_blank_or_ragged_lines(path)
    points, psizes, times = load_weak(path)
    assert len(points) == len(psizes) == len(times) > 0
    items = len(times) // len(n0s)                       # the notebook's ITEMS: rows per series (7 there, <= 4 on one node)
    assert items * len(n0s) == len(times) and 1 <= items <= 4
    for i, n0 in enumerate(n0s):
        blk = slice(i * items, (i + 1) * items)
        assert psizes[blk] == [1, 2, 4, 8][:items]
        assert points[blk] == [int(math.floor(n0 * math.sqrt(p))) for p in psizes[blk]]                # code/MPI/cg.run:22-44
        eff = times[i * items] / np.array(times[blk])                                                   # plots.ipynb:81
        assert eff.dtype == np.float64 and eff[0] == 1.0 and np.all(eff > 0)


def test_mtx_file():
    path = os.path.join(RES, "MI355X_mtx.txt")
    no_blank_or_ragged_lines(path)
    res = load_cuda_t(path)
    assert list(res) == [16]                             # BLOCK_WIDTH echoed
    assert [r[0] for r in res[16]] == [32, 256, 1024]    # NUM_THREADS echoed
    assert all(0 < r[1] < 0.822428 for r in res[16])     # the reference's best published CUDA time (results/CUDA_T.txt:48)


def test_plot_script_reads_every_committed_file_without_an_edit(tmp_path):
    """results/plot_mi355x.py restates the notebook's two MPI loaders with ITEMS taken from the file: it
    must give the same numbers as the loaders above on every committed file, for one row per series (a one-GPU box) and
    for a synthetic 4-row node file, and draw both figures."""
    import importlib.util
    import subprocess
    import sys
    spec = importlib.util.spec_from_file_location("plot_mi355x", os.path.join(RES, "plot_mi355x.py"))
    pm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pm)
    assert list(pm.ALPHAS) == ALPHA_KEYS
    for name in ("strong_scaling.txt", "strong_scaling_n32768.txt"):
        mine, ref = pm.strong_series(os.path.join(RES, name)), load_strong(os.path.join(RES, name))
        assert list(mine) == list(ref)
        for n in ref:
            assert [(p, t) for p, t, _, _ in mine[n]] == [tuple(r) for r in ref[n]]
            assert [s for _, _, s, _ in mine[n]] == list(ref[n][0][1] / np.array([r[1] for r in ref[n]]))      # plots.ipynb:34
    for name, n0s in (("weak_scaling.txt", [1024, 1448, 2048]), ("weak_scaling_n16384.txt", [16384])):
        ss = pm.weak_series(os.path.join(RES, name))
        assert [s[0][0] for s in ss] == n0s and all(s[0][3] == 1.0 for s in ss)
    # a node file: three series of four rows (what experiments/cg_mi355x.run leaves on an 8-GPU node)
    node = tmp_path / "weak_scaling.txt"
    node.write_text("".join("%d,%d,%g\n" % (int(math.floor(n0 * math.sqrt(p))), p, 0.01 * (1 + 0.1 * k))
                            for n0 in (1024, 1448, 2048) for k, p in enumerate((1, 2, 4, 8))))
    ss = pm.weak_series(str(node))
    assert [len(s) for s in ss] == [4, 4, 4] and [round(q[3], 6) for q in ss[2]] == [1.0, round(1 / 1.1, 6), round(1 / 1.2, 6), round(1 / 1.3, 6)]
    r = subprocess.run([sys.executable, os.path.join(RES, "plot_mi355x.py"), "--png", str(tmp_path / "fig")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "ITEMS = [1]" in r.stdout and "N=32768" in r.stdout
    assert (tmp_path / "fig_strong.png").stat().st_size > 1000 and (tmp_path / "fig_weak.png").stat().st_size > 1000
