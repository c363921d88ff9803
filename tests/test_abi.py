"""CPU-side tests of the drop-in boundary: libcgx.so loads, exports every symbol include/cgx.h declares,
its host-only entry points work, and everything that needs the GPU fails LOUDLY without one (no fallback)."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _has_gpu():
    import torch
    return torch.cuda.is_available()


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "cgx.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(cgx_[a-z0-9_]+)\s*\(", hdr)))


def test_header_and_binding_agree(pkg):
    assert declared_symbols() == sorted(pkg.cgx.EXPORTS)


def test_library_exports_every_declared_symbol(pkg):
    L = pkg.cgx.lib()
    for name in declared_symbols():
        assert hasattr(L, name), name
    out = subprocess.run(["nm", "-D", "--defined-only", pkg.cgx.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (cgx_\w+)", out))
    assert set(declared_symbols()) <= exported


def test_library_contains_gfx950_code_object(pkg):
    blob = open(pkg.cgx.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"k_gemv_colsplit" in blob


def test_struct_layouts_match_header(pkg):
    assert C.sizeof(pkg.cgx.Config) == 4 * 5 + 128 + 4 * 5 + 4 * 8
    assert C.sizeof(pkg.cgx.Result) == 8 + 8 * 8 + 8 + 8 + 8 * 4


def test_partition_through_abi_matches_oracle(pkg, oracle):
    for n, p in [(10, 1), (10, 3), (1000, 3), (32768, 8), (46340, 8), (23170, 2), (5, 8)]:
        assert pkg.partition(n, p) == oracle.partition(n, p)


def test_status_strings(pkg):
    L = pkg.cgx.lib()
    assert L.cgx_status_string(0) == b"ok"
    assert b"no CPU fallback" in L.cgx_status_string(6)


def test_bad_arguments_are_rejected_not_crashed(pkg):
    L = pkg.cgx.lib()
    assert L.cgx_partition(10, 0, None, None) == 1
    assert L.cgx_set_max_iter(None, 3) == 1
    assert L.cgx_solve(None, None, None) == 1
    assert L.cgx_read_matrix(None, b"/nonexistent.mtx") == 1


def test_no_gpu_means_loud_failure(pkg):
    if _has_gpu():
        pytest.skip("this machine has a GPU")
    with pytest.raises(pkg.CgxError) as e:
        pkg.CGSolver()
    assert e.value.status == 6 and "no CPU fallback" in str(e.value)


def test_cli_usage_and_failure_exit_codes(pkg):
    exe = os.path.join(ROOT, "conjugate-gradient_amd", "cgsolver")
    assert os.path.exists(exe), "cgsolver not built (run __graft_entry__.build())"
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 1 and "Usage" in r.stderr          # code/MPI/cg_main.cc:22-26
    r = subprocess.run([exe, "64", "/tmp/cgx_never_written.txt", "--cpu"], capture_output=True, text=True)
    assert r.returncode == 1 and "no CPU fallback" in r.stderr and not os.path.exists("/tmp/cgx_never_written.txt")
    if not _has_gpu():
        r = subprocess.run([exe, "64", "/tmp/cgx_never_written.txt"], capture_output=True, text=True)
        assert r.returncode == 1 and "no CPU fallback" in r.stderr
        assert not os.path.exists("/tmp/cgx_never_written.txt")
        # forked ranks: every rank reports, nobody hangs in a collective wire-up
        r = subprocess.run([exe, "64", "/tmp/cgx_never_written.txt", "--gpus", "2"], capture_output=True, text=True, timeout=60)
        assert r.returncode == 1 and "not every rank has a usable MI355X" in r.stderr


def test_product_package_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under conjugate-gradient_amd/ or include/ may reference it."""
    bad = []
    for base in ("conjugate-gradient_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            if "build" in dp.split(os.sep):
                continue
            for f in files:
                if f.endswith((".py", ".cc", ".cpp", ".hip", ".h", ".hh", "Makefile")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"cg_oracle|oracle\.py|from oracle|import oracle|libcg_oracle|load_oracle", txt):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_header_is_plain_c_and_a_c_client_links(pkg, tmp_path):
    """include/cgx.h is the boundary a C / cgo / JNI / ctypes client binds: it must compile as C99 with nothing but
    itself, and a C program using only its host-side entry points must link against libcgx.so and run without a GPU."""
    src = tmp_path / "client.c"
    src.write_text(r'''
#include "cgx.h"
#include <stdio.h>
#include <string.h>
int main(void) {
    int start[3], rows[3];
    cgx_config cfg;
    cgx_ctx *ctx = 0;
    cgx_config_init(&cfg);
    if (cfg.struct_version != CGX_VERSION || cfg.matrix_format != CGX_MATRIX_DENSE) return 2;
    if (cgx_partition(10, 3, start, rows) != CGX_OK || rows[2] != 4) return 3;         /* cg.cc:255-266 */
    cgx_status st = cgx_create(&ctx, &cfg);
    if (st == CGX_OK) { cgx_destroy(ctx); printf("gpu\n"); return 0; }
    if (st != CGX_ERR_NO_DEVICE || !strstr(cgx_last_error(0), "no CPU fallback")) return 4;
    printf("%s\n", cgx_status_string(st));
    return 0;
}
''')
    exe = tmp_path / "client"
    inc = os.path.join(ROOT, "include")
    libdir = os.path.dirname(pkg.cgx.LIB_PATH)
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", inc, str(src), "-o", str(exe),
                        "-L", libdir, "-lcgx", "-Wl,-rpath," + libdir], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert r.stdout.strip() in ("gpu", "no usable GPU (libcgx has no CPU fallback)")


def test_mirror_class_keeps_the_reference_members_virtual(pkg, tmp_path):
    """code/MPI/cg.hh:17-32 declares read_matrix, partition_matrix, generate_lap2d_matrix, solve and set_max_iter virtual: a
    caller that derives from CGSolver and overrides them must keep working with the drop-in header (host/cg.hh).  Compiled
    with the reference's own warning flags (code/MPI/Makefile:3); the override is reached through a base reference."""
    src = tmp_path / "derived.cc"
    src.write_text(r'''
#include "cg.hh"
#include <cstdio>
#include <type_traits>
struct Mine : CGSolver {
    int calls = 0;
    void read_matrix(const std::string &) override { ++calls; }
    void partition_matrix(int, int, int[], int[]) override { ++calls; }
    void generate_lap2d_matrix(int) override { ++calls; }
    void solve(std::vector<double> &) override { ++calls; }
    void set_max_iter(int) override { ++calls; }
};
static_assert(std::has_virtual_destructor<CGSolver>::value, "a polymorphic base needs a virtual destructor");
int through_base(CGSolver &s)
{
    std::vector<double> x;
    int a[1], b[1];
    s.read_matrix("f");
    s.partition_matrix(1, 1, a, b);
    s.generate_lap2d_matrix(4);
    s.solve(x);
    s.set_max_iter(3);
    return 0;
}
int main() { std::printf("%d\n", (int)std::is_polymorphic<CGSolver>::value); return 0; }
''')
    exe = tmp_path / "derived"
    pkgdir = os.path.dirname(pkg.cgx.LIB_PATH)
    r = subprocess.run(["g++", "-std=c++11", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                        "-I", os.path.join(pkgdir, "host"), str(src), os.path.join(pkgdir, "host", "cg.cc"), "-o", str(exe),
                        "-L", pkgdir, "-lcgx", "-Wl,-rpath," + pkgdir], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip() == "1"


def test_build_does_not_load_the_library_into_the_calling_process():
    """__graft_entry__.build() checks that libcgx.so loads -- in a child process.  Loaded into the caller, the library binds
    the process to /opt/rocm's HIP runtime before torch brings its own, and a smoke() in the same process then finds no
    device (seen on the GPU box: `python -c "import __graft_entry__ as g; g.build(); g.smoke()"`)."""
    code = ("import __graft_entry__ as g; g.build(); "
            "print('MAPPED' if any('libcgx' in l for l in open('/proc/self/maps')) else 'CLEAN')")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.strip().endswith("CLEAN")
