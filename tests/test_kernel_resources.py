"""Build-time checks of the persistent kernels' register budgets: `hipcc -Rpass-analysis=kernel-resource-usage`
cross-compiles for gfx950 without a GPU.

csrc/cgx_resident.hip issues the streamed rows of its hybrid shapes (2048 < n <= 4096) with hand-written `global_load ... nt`
and waits for them a whole loop trip later: a register-allocator spill or copy of one of those destination registers inside
that window would read a register whose load has not landed, and nothing validates A's streamed rows.  The kernels sit at
487-511 of 512 registers, so a toolchain bump could introduce exactly that: this test fails the build if ANY instantiation of
k_cg_resident spills a vector register or touches scratch.  csrc/cgx_stream.hip leaves every wait to the compiler (buffer-load
builtins), so a spill there costs time, not correctness: the instantiations that hold rows of A on the chip (n <= 11264: S <= 11;
the library's default uses them up to S = 10: n <= 10000) must be spill-free -- they sit at 227-256 of 256 registers because every register that is free holds a piece
of a row of A; the others may spill outside the sweep (documented in DESIGN.md) but must still build."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


def resources(src):
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"), "-c",
           os.path.join(ROOT, "conjugate-gradient_amd", "csrc", src), "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    rows, cur = [], None
    for line in p.stderr.splitlines():
        m = re.search(r"remark:\s+([A-Za-z][A-Za-z \[\]/]*?):\s*(.*?) \[-Rpass", line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2).strip()
        if "Name" in k:
            name = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()
            cur = {"name": name}
            rows.append(cur)
        elif cur is not None:
            cur[k] = v
    return rows


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_resident_kernels_never_spill():
    rows = [r for r in resources("cgx_resident.hip") if "k_cg_resident" in r["name"]]
    assert len(rows) == 20, [r["name"] for r in rows]           # 4 hybrid shapes + 4 x 4 all-in-LDS shapes
    for r in rows:
        assert int(r["VGPRs Spill"]) == 0 and int(r["ScratchSize [bytes/lane]"]) == 0, r


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_stream_kernels_of_the_default_range_never_spill():
    rows = [r for r in resources("cgx_stream.hip") if "k_cg_stream" in r["name"]]
    assert len(rows) == 16, [r["name"] for r in rows]           # S = 1 ... 16
    for r in rows:
        s = int(re.search(r"k_cg_stream<(\d+),", r["name"]).group(1))
        assert int(r["VGPRs"]) <= 256, r                        # two waves per SIMD: one workgroup of 512 threads per CU
        if s <= 11:
            assert int(r["VGPRs Spill"]) == 0 and int(r["ScratchSize [bytes/lane]"]) == 0, r
