"""conjugate-gradient_amd -- MI355X-native drop-in for the reference's CGSolver::solve() path.

The directory name has a hyphen (it mirrors the reference repo's name), so it is imported through
`__graft_entry__.load_package()`, which registers it as the module `conjugate_gradient_amd`.

Contents: csrc/ (HIP kernels + the C ABI of include/cgx.h), host/ (C++ CGSolver mirror + cgsolver CLI),
cgx.py (ctypes binding used by tests/ and bench.py).  No CPU fallback anywhere in this package.
"""
from . import cgx  # noqa: F401
from .cgx import (CGSolver, CgxError, COMM_SELF, COMM_LOOPBACK, COMM_RCCL, COMM_P2P, MATRIX_DENSE, MATRIX_BANDED,  # noqa: F401
                  partition, comm_unique_id)
