// ref_matrix_wrap.cc -- C entry point around the REFERENCE's own Matrix-Market reader, compiled from the sources
// where they lie under /root/reference (oracle/Makefile target `ref`, output in oracle/_ref/ only; never committed).
// Test infrastructure: validates oracle_read_mtx_dense and cgx_read_matrix against the real Matrix::read
// (code/MPI/matrix.cc:6-22 -> MatrixCOO::read, code/MPI/matrix_coo.cc:7-60 -> mmio.c).
// Only this subset of the reference is buildable here: cg.cc needs <cblas.h>, which the image does not ship.
#include <cstdlib>
#include <cstring>
#include <string>

#include "matrix.hh"   // the reference's header, found through -I/root/reference/code/MPI

extern "C" int ref_matrix_read(const char *path, int *m, int *n, double **data)
{
    Matrix A;
    A.read(std::string(path));   // exits the process on a malformed file (matrix_coo.cc:14-33): feed valid files only
    *m = A.m();
    *n = A.n();
    const size_t count = static_cast<size_t>(A.m()) * static_cast<size_t>(A.n());
    *data = static_cast<double *>(malloc(count * sizeof(double)));
    if (!*data) return -1;
    memcpy(*data, A.data(), count * sizeof(double));
    return 0;
}
