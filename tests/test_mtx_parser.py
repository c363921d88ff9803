"""The library's Matrix-Market parser (cgx_read_matrix's host half: header checks + the entries parsed from the mapped file
on several host threads) WITHOUT a GPU: against the oracle's restatement of MatrixCOO::read / Matrix::read
(code/MPI/matrix_coo.cc:7-60, matrix.cc:6-22), against the reference's own reader compiled from its sources where it exists
(oracle/_ref), and with the file cut into many thread ranges so that range boundaries fall inside tokens and inside entries."""
import os

import numpy as np
import pytest

from test_oracle import MTX_CASES

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def densify(m, n, sym, I, J, a):
    """Matrix::read, matrix.cc:12-21: sequential assignment, a later entry wins, the mirror follows its original."""
    A = np.zeros((m, n))
    for i, j, v in zip(I, J, a):
        A[i, j] = v
        if sym:
            A[j, i] = v
    return A


@pytest.mark.parametrize("threads", [1, -2, -3, -7, -64, 0])
def test_parser_matches_the_oracle_and_the_reference_reader(pkg, oracle, mtx_path, tmp_path, threads):
    files = {"fixture": mtx_path} if threads in (1, -7) else {}      # the 10000 x 10000 fixture densifies to 800 MB: twice is enough
    for name, text in MTX_CASES.items():
        f = tmp_path / name
        f.write_text(text)
        files[name] = str(f)
    # white space the reference's fscanf("%d %d %lg") accepts: tabs, blank lines, CRLF, several entries on one line
    odd = tmp_path / "odd_space.mtx"
    odd.write_bytes(b"%%MatrixMarket matrix coordinate real general\r\n%c\r\n3 3 4\r\n1\t1  2.5\r\n\r\n2 2 -1e-3   3 3 7 \n 3\n1\n0.5\n")
    files["odd_space"] = str(odd)
    for name, path in files.items():
        m, n, sym, I, J, a = pkg.cgx.parse_matrix_market(path, threads)
        A = densify(m, n, sym, I, J, a)
        Ao, nz, symo = oracle.read_mtx_dense(path)
        assert (m, n) == Ao.shape and sym == symo and len(a) == nz, name
        assert np.array_equal(A, Ao), name
        if oracle.ref_available():
            assert np.array_equal(A, oracle.ref_read_mtx_dense(path)), name


def test_parser_threads_agree_on_a_larger_file(pkg, tmp_path):
    """20 000 entries with values of every printed length: every forced thread count gives the same entries in file order."""
    rng = np.random.default_rng(3)
    n, nz = 5000, 20000
    I = rng.integers(1, n + 1, nz)
    J = rng.integers(1, n + 1, nz)
    a = rng.standard_normal(nz) * 10.0 ** rng.integers(-8, 8, nz)
    f = tmp_path / "big.mtx"
    with open(f, "w") as fh:
        fh.write("%%MatrixMarket matrix coordinate real general\n")
        fh.write("%d %d %d\n" % (n, n, nz))
        for i, j, v in zip(I, J, a):
            fh.write("%d %d %r\n" % (i, j, float(v)))
    base = pkg.cgx.parse_matrix_market(str(f), 1)
    assert np.array_equal(base[3], I - 1) and np.array_equal(base[4], J - 1) and np.array_equal(base[5], a)   # %r round-trips
    for threads in (-2, -5, -16, -61, 0):
        got = pkg.cgx.parse_matrix_market(str(f), threads)
        assert all(np.array_equal(x, y) for x, y in zip(base[3:], got[3:])), threads


@pytest.mark.parametrize("threads", [1, -4])
def test_parser_reports_the_first_bad_entry(pkg, tmp_path, threads):
    head = "%%MatrixMarket matrix coordinate real general\n4 4 5\n"
    cases = {
        "1 1 1\n2 2 x\n3 3 3\n4 4 4\n1 2 5\n": "entry 1 unreadable",            # not a number
        "1 1 1\n2 2 2\n3 3 3\n": "entry 3 unreadable",                           # file ends early
        "1 1 1\n2 2 2\n3 3 3\n4 4\n": "entry 3 unreadable",                      # ends inside an entry
        "1 1 1\n2 9 2\n3 3 3\n4 4 4\n1 2 5\n": "index out of range",              # column beyond n
        "0 1 1\n2 2 2\n3 3 3\n4 4 4\n1 2 5\n": "index out of range",              # 1-based: 0 is out
        "1 1 1\n2.5 2 2\n3 3 3\n4 4 4\n1 2 5\n": "entry 1 unreadable",            # "%d" does not take 2.5
    }
    for k, (body, msg) in enumerate(cases.items()):
        f = tmp_path / ("bad%d.mtx" % k)
        f.write_text(head + body)
        with pytest.raises(pkg.CgxError) as e:
            pkg.cgx.parse_matrix_market(str(f), threads)
        assert e.value.status == 2 and msg in str(e.value), (body, str(e.value))
    # entries beyond nz are never looked at (the reference reads exactly nz of them, matrix_coo.cc:44)
    f = tmp_path / "extra.mtx"
    f.write_text(head + "1 1 1\n2 2 2\n3 3 3\n4 4 4\n1 2 5\ngarbage here\n")
    assert len(pkg.cgx.parse_matrix_market(str(f), threads)[5]) == 5


def test_parser_header_errors(pkg, tmp_path):
    cases = {
        "not a banner\n1 1 1\n": (2, "banner"),
        "%%MatrixMarket matrix array real general\n2 2\n1\n2\n3\n4\n": (7, "does not support"),
        "%%MatrixMarket matrix coordinate complex general\n2 2 1\n1 1 1 0\n": (7, "field not supported"),
        "%%MatrixMarket matrix coordinate real skew-symmetric\n2 2 1\n2 1 1\n": (7, "symmetry not supported"),
        "%%MatrixMarket matrix coordinate real general\n2 3 1\n1 1 1\n": (7, "square"),
        "%%MatrixMarket matrix coordinate real general\n% only comments\n": (2, "size line"),
    }
    for k, (text, (status, msg)) in enumerate(cases.items()):
        f = tmp_path / ("h%d.mtx" % k)
        f.write_text(text)
        with pytest.raises(pkg.CgxError) as e:
            pkg.cgx.parse_matrix_market(str(f))
        assert e.value.status == status and msg in str(e.value), (text, str(e.value))
    with pytest.raises(pkg.CgxError) as e:
        pkg.cgx.parse_matrix_market(str(tmp_path / "missing.mtx"))
    assert e.value.status == 2 and "Could not open matrix" in str(e.value)


@pytest.mark.parametrize("threads", [1, -3, -16])
def test_parser_file_ending_exactly_at_a_page_boundary_without_a_newline(pkg, tmp_path, threads):
    """The body is parsed from the mapped file: a file whose size is a whole number of pages and whose last token runs to the
    very last byte must not be read past its end (nothing terminates a token there but the end of the mapping)."""
    page = os.sysconf("SC_PAGE_SIZE")
    head = "%%MatrixMarket matrix coordinate real general\n"
    n, nz = 9999, 2000
    rows = ["%d %d %.17g" % (1 + (7 * k) % n, 1 + (13 * k) % n, 0.5 + k) for k in range(nz)]
    body = "\n".join(rows)
    size_line = "%d %d %d\n" % (n, n, nz)
    total = len(head) + len(size_line) + len(body)
    pad = (-total) % page                       # comment line of exactly the missing length (>= 2: '%' and '\n')
    if pad < 2:
        pad += page
    text = head + "%" + "c" * (pad - 2) + "\n" + size_line + body
    assert len(text) % page == 0 and not text.endswith("\n")
    f = tmp_path / "page.mtx"
    f.write_text(text)
    m, n2, sym, I, J, a = pkg.cgx.parse_matrix_market(str(f), threads)
    assert (m, n2, len(a)) == (n, n, nz) and a[-1] == 0.5 + (nz - 1) and I[-1] == (7 * (nz - 1)) % n and J[0] == 0
