"""Shapes of the persistent kernels for EVERY problem size, checked on the CPU: `cgx_probe_persistent_plan` is host arithmetic
(csrc/cgx_resident.hip plan_resident, csrc/cgx_stream.hip plan_stream), no device, no context.  What the kernels assume about
their launch, and nothing validates on the device: the grid covers all rows and fits the chip (one workgroup per CU: they wait
for each other), the streamed rows are a whole number of ring batches, the LDS layout fits what a workgroup may have, the
exchange buffer has a slot for every column a thread gathers, the own-rows table has room for every pair of a workgroup."""
import ctypes as C

import pytest

LDS = 160 * 1024          # MI355X: 160 KB of LDS per CU, all of it available to one workgroup
CUS = 256


def plan(pkg, n, streaming, cus=CUS, lds=LDS):
    out = (C.c_long * 12)()
    st = pkg.cgx.lib().cgx_probe_persistent_plan(n, cus, lds, streaming, out)
    assert st == 0
    keys = ("fits", "R", "S", "grid", "xslots", "lds_bytes", "RL", "RG", "RB", "l2_rows", "hybrid", "threads")
    return dict(zip(keys, list(out)))


def test_resident_plan_for_every_n(pkg):
    for n in range(1, 4097):
        p = plan(pkg, n, 0)
        assert p["fits"] == 1 and p["threads"] == 256, (n, p)
        assert p["grid"] <= CUS and p["grid"] * p["R"] >= n > (p["grid"] - 1) * p["R"], (n, p)      # every row, no empty workgroup
        assert p["S"] * 512 >= n > (p["S"] - 1) * 512 and p["xslots"] == 512 * p["S"], (n, p)
        assert p["lds_bytes"] <= LDS and p["RL"] + p["RG"] <= p["R"], (n, p)
        if n <= 2048:
            assert p["hybrid"] == 0 and p["RL"] == p["R"] <= 8 and p["R"] & (p["R"] - 1) == 0, (n, p)
        else:
            assert p["hybrid"] == 1 and p["R"] == 16 and p["RL"] >= 1, (n, p)
    assert plan(pkg, 4096, 0, cus=255)["fits"] == 0             # 256 workgroups of 16 rows need 256 CUs
    assert plan(pkg, 2048, 0, lds=64 * 1024)["fits"] == 0       # 8 rows of 16 KB do not fit 64 KB


def test_stream_plan_for_every_n(pkg):
    rows_on_chip = {}
    for n in range(1024, 16385):
        p = plan(pkg, n, 1)
        assert p["fits"] == 1 and p["threads"] == 512, (n, p)
        R, S, RB, res = p["R"], p["S"], p["RB"], p["RL"] + p["RG"]
        assert p["grid"] <= CUS and p["grid"] * R >= n > (p["grid"] - 1) * R, (n, p)
        assert S * 1024 >= n > (S - 1) * 1024 and p["xslots"] == 1024 * S, (n, p)
        assert R < 512, (n, p)                                  # thread t publishes row t of the workgroup
        assert R > res and (R - res) % RB == 0 and (R - res) // RB >= 1, (n, p)      # a whole number of batches, at least one
        assert 0 <= p["l2_rows"] <= R - res, (n, p)
        # the LDS layout of k_cg_stream: [8][R] row sums | 2 x 2 x 8 dot partials | 8 | 4 | parked Ap [S][512] pairs |
        # RL rows [S][512] pairs | own rows 3 x (R / 2 + 2) pairs
        need = (8 * R + 32 + 8 + 4 + 2 * S * 512 * (1 + p["RL"]) + 6 * (R // 2 + 2)) * 8
        assert p["lds_bytes"] == need <= LDS, (n, p)
        rows_on_chip.setdefault(S, set()).add((p["RL"], p["RG"], RB, p["l2_rows"]))
    assert all(len(v) == 1 for v in rows_on_chip.values())       # the shape depends on S only
    assert rows_on_chip[5] == {(2, 7, 1, 2)} and rows_on_chip[8] == {(1, 3, 1, 0)} and rows_on_chip[10] == {(0, 1, 1, 0)}
    assert rows_on_chip[12] == {(0, 0, 1, 0)}
    # the library's default hands n > 4096 to the streaming kernel through plan_resident
    assert plan(pkg, 5000, 0) == plan(pkg, 5000, 1)
    assert plan(pkg, 1023, 1)["fits"] == 0 and plan(pkg, 16385, 1)["fits"] == 0
    # fewer CUs: more rows per workgroup, until the row sums and the rows on the chip no longer fit the LDS
    p = plan(pkg, 8192, 1, cus=128)
    assert p["fits"] == 1 and p["grid"] <= 128 and p["grid"] * p["R"] >= 8192
    assert plan(pkg, 16384, 1, cus=16)["fits"] == 0


@pytest.mark.parametrize("args", [(0, 256, LDS, 0), (100, 0, LDS, 0), (100, 256, -1, 1)])
def test_bad_arguments(pkg, args):
    out = (C.c_long * 12)()
    assert pkg.cgx.lib().cgx_probe_persistent_plan(*args, out) != 0
