// cg.hh -- host-side C++ mirror of the reference's CGSolver interface, implemented on libcgx's C ABI.
//
// Same member names and argument meaning as code/MPI/cg.hh:11-57 and code/CUDA/cg.hh:13-45, so the
// reference's two main()s (code/MPI/cg_main.cc, code/CUDA/cg_main.cc) port by changing only the
// constructor call.  The class owns no numerical code: A, b and all work vectors live on the MI355X
// inside the cgx context.
#ifndef CGX_HOST_CG_HH
#define CGX_HOST_CG_HH

#include <string>
#include <vector>

#include "cgx.h"

class CGSolver {
public:
    /// one shard on device 0 (the reference's `srun -n 1`)
    CGSolver();
    /// explicit placement: comm mode, rank/nranks (what MPI_Comm_rank/size gave the reference), device
    explicit CGSolver(const cgx_config &cfg);
    virtual ~CGSolver();
    CGSolver(const CGSolver &) = delete;
    CGSolver &operator=(const CGSolver &) = delete;

    // The members the reference declares virtual (code/MPI/cg.hh:17-32) are virtual here too: a drop-in header must not
    // narrow the interface for a caller that derives from CGSolver.
    /// read matrix from .mtx file (code/MPI/cg.hh:17; sizes are set as in code/CUDA/cg.cu:317-319)
    virtual void read_matrix(const std::string &filename);
    /// initialize source term (cg.hh:20)
    void init_source_term(double h);
    /// partition matrix (cg.hh:23)
    virtual void partition_matrix(int N, int psize, int start_rows[], int num_rows[]);
    /// generate the synthetic matrix for the scaling experiments (cg.hh:26)
    virtual void generate_lap2d_matrix(int size);
    /// conjugate gradient, MPI-form signature (cg.hh:29): x = initial guess in, solution out
    virtual void solve(std::vector<double> &x);
    /// CUDA-form signature (code/CUDA/cg.hh:29).  NUM_THREADS / BLOCK_WIDTH / T tuned the reference's
    /// own kernels and have no counterpart here: accepted and ignored.  x is zeroed first (cg.cu:217).
    void solve(double *x, int NUM_THREADS, int BLOCK_WIDTH, bool T);
    /// fix maximum number of iterations (cg.hh:32)
    virtual void set_max_iter(int maxIter);
    int m() const;
    int n() const;
    /// residual tolerance (cg.hh:39)
    void tolerance(double tolerance);

    // additions
    const cgx_result &last_result() const { return m_result; }
    int rank() const { return m_cfg.rank; }
    int psize() const { return m_cfg.nranks; }
    /// print the reference's DEBUG line (cg.cc:152-153) after solve; default true on rank 0
    void set_verbose(bool v) { m_verbose = v; }
    /// the underlying C-ABI context (for the CGX_COMM_P2P wire-up: cgx_p2p_export / import / selftest)
    cgx_ctx *context() const { return m_ctx; }

private:
    void check(int status, const char *what) const;
    cgx_ctx *m_ctx{nullptr};
    cgx_config m_cfg{};
    cgx_result m_result{};
    bool m_verbose{true};
};

#endif
