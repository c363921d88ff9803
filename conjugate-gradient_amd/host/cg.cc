// cg.cc -- CGSolver mirror over the C ABI (see cg.hh).
#include "cg.hh"

#include <iostream>
#include <stdexcept>

CGSolver::CGSolver()
{
    cgx_config_init(&m_cfg);
    check(cgx_create(&m_ctx, &m_cfg), "cgx_create");
}

CGSolver::CGSolver(const cgx_config &cfg) : m_cfg(cfg) { check(cgx_create(&m_ctx, &m_cfg), "cgx_create"); }

CGSolver::~CGSolver() { cgx_destroy(m_ctx); }

void CGSolver::check(int status, const char *what) const
{
    if (status == CGX_OK) return;
    // The reference prints and exit(1)s (matrix_coo.cc:14-33); a library must not: throw, main() maps to exit 1.
    throw std::runtime_error(std::string(what) + ": " + cgx_status_string(static_cast<cgx_status>(status)) + ": " +
                             cgx_last_error(m_ctx));
}

void CGSolver::read_matrix(const std::string &filename) { check(cgx_read_matrix(m_ctx, filename.c_str()), "read_matrix"); }

void CGSolver::init_source_term(double h) { check(cgx_init_source_term(m_ctx, h), "init_source_term"); }

void CGSolver::partition_matrix(int N, int psize, int start_rows[], int num_rows[])
{
    check(cgx_partition(N, psize, start_rows, num_rows), "partition_matrix");
}

void CGSolver::generate_lap2d_matrix(int size) { check(cgx_generate_lap2d_matrix(m_ctx, size), "generate_lap2d_matrix"); }

void CGSolver::set_max_iter(int maxIter) { check(cgx_set_max_iter(m_ctx, maxIter), "set_max_iter"); }

void CGSolver::tolerance(double tolerance) { check(cgx_set_tolerance(m_ctx, tolerance), "tolerance"); }

int CGSolver::m() const
{
    int m = 0, n = 0;
    cgx_get_size(m_ctx, &m, &n);
    return m;
}

int CGSolver::n() const
{
    int m = 0, n = 0;
    cgx_get_size(m_ctx, &m, &n);
    return n;
}

void CGSolver::solve(std::vector<double> &x)
{
    if (static_cast<int>(x.size()) != n()) throw std::runtime_error("solve: x has the wrong length");
    check(cgx_solve(m_ctx, x.data(), &m_result), "solve");
    if (m_verbose && m_cfg.rank == 0) {
        // byte-compatible with the reference's DEBUG line, code/MPI/cg.cc:152-153
        std::cout << "\t[STEP " << m_result.iterations << "] residual = " << std::scientific << m_result.residual_prev
                  << ", ||x|| = " << m_result.x_norm << ", ||Ax - b||/||b|| = " << m_result.rel_residual << std::endl;
    }
}

void CGSolver::solve(double *x, int /*NUM_THREADS*/, int /*BLOCK_WIDTH*/, bool /*T*/)
{
    std::vector<double> xv(static_cast<size_t>(n()), 0.0);   // fill<<<>>>(m_n, x, 0.0), code/CUDA/cg.cu:217
    solve(xv);
    for (size_t i = 0; i < xv.size(); ++i) x[i] = xv[i];
}
