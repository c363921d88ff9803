// cg_main.cc -- the `cgsolver` command line, both reference forms in one binary.
//
//   cgsolver N OUTFILE [MAXITER]                                  (code/MPI/cg_main.cc:13-69, README.md:63,71)
//   cgsolver FILE.mtx NUM_THREADS BLOCK_WIDTH true|false OUTFILE  (code/CUDA/cg_main.cc:16-63, README.md:98)
//
// If argv[1] parses completely as an integer it is the generator form, otherwise a Matrix-Market path.
// Where the reference took its process count from `srun -n P`, this takes `--gpus P` (or CG_NGPU):
// the binary forks P-1 children BEFORE touching the GPU, one process per MI355X, and the ranks meet
// through an RCCL unique id passed over pipes (replaces MPI_Init, cg_main.cc:15-20).
// `--loopback P` runs P logical row blocks on one GPU (CI stand-in for a multi-GPU node).
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "cg.hh"

using clk = std::chrono::high_resolution_clock;
using second = std::chrono::duration<double>;

namespace {

bool parse_int(const std::string &s, int *out)
{
    if (s.empty()) return false;
    char *end = nullptr;
    long v = strtol(s.c_str(), &end, 10);
    if (*end != '\0') return false;
    *out = static_cast<int>(v);
    return true;
}

int usage(const char *prog)
{
    std::cerr << "Usage: " << prog << " N OUTFILE [MAXITER]            (generated matrix of size N)\n"
              << "       " << prog << " FILE.mtx NUM_THREADS BLOCK_WIDTH true|false OUTFILE\n"
              << "options: --gpus P (or CG_NGPU=P)  one process per MI355X, RCCL over xGMI\n"
              << "         --loopback P             P logical row blocks on one GPU\n"
              << "         --stats                  also print iterations/s and K1 GB/s on stderr" << std::endl;
    return 1;
}

}  // namespace

int main(int argc, char **argv)
{
    // ---- split options from the reference's positional arguments -------------------------------------
    std::vector<std::string> pos;
    int ngpu = 1, loopback = 0;
    bool stats = false;
    if (const char *e = getenv("CG_NGPU")) ngpu = atoi(e);
    for (int i = 1; i < argc; ++i) {
        std::string a(argv[i]);
        if (a == "--gpus" && i + 1 < argc) ngpu = atoi(argv[++i]);
        else if (a == "--loopback" && i + 1 < argc) loopback = atoi(argv[++i]);
        else if (a == "--stats") stats = true;
        else pos.push_back(a);
    }
    if (pos.empty()) return usage(argv[0]);   // cg_main.cc:22-26 (returns 1)
    if (ngpu < 1) ngpu = 1;

    int gen_n = 0;
    const bool gen_form = parse_int(pos[0], &gen_n);
    std::string out_file;
    int max_iter = -1, legacy_nt = 0, legacy_bw = 0;
    bool legacy_t = false;
    if (gen_form) {
        if (pos.size() < 2) return usage(argv[0]);
        out_file = pos[1];
        if (pos.size() >= 3) {                // cg_main.cc:37-42
            std::stringstream ss(pos[2]);
            ss >> max_iter;
        }
    } else {
        if (pos.size() < 5) return usage(argv[0]);   // the reference reads argv[2..5] unguarded (cg_main.cc:21-33)
        if (!parse_int(pos[1], &legacy_nt) || !parse_int(pos[2], &legacy_bw)) return usage(argv[0]);
        legacy_t = (pos[3] == "true");
        out_file = pos[4];
    }

    // ---- one process per GPU: fork before any HIP/RCCL call -------------------------------------------
    int rank = 0;
    std::vector<int> wr_pipes;
    int rd_pipe = -1;
    std::vector<pid_t> kids;
    if (ngpu > 1) {
        for (int r = 1; r < ngpu; ++r) {
            int fd[2];
            if (pipe(fd) != 0) { perror("pipe"); return 1; }
            pid_t pid = fork();
            if (pid < 0) { perror("fork"); return 1; }
            if (pid == 0) {
                rank = r;
                rd_pipe = fd[0];
                close(fd[1]);
                for (int w : wr_pipes) close(w);
                wr_pipes.clear();
                kids.clear();
                break;
            }
            kids.push_back(pid);
            wr_pipes.push_back(fd[1]);
            close(fd[0]);
        }
    }

    int rc = 0;
    try {
        cgx_config cfg;
        cgx_config_init(&cfg);
        if (ngpu > 1) {
            cfg.comm_mode = CGX_COMM_RCCL;
            cfg.nranks = ngpu;
            cfg.rank = rank;
            cfg.device = rank;
            if (rank == 0) {
                if (cgx_comm_unique_id(cfg.unique_id) != CGX_OK)
                    throw std::runtime_error(std::string("cgx_comm_unique_id: ") + cgx_last_error(nullptr));
                for (int w : wr_pipes) {
                    if (write(w, cfg.unique_id, CGX_UNIQUE_ID_BYTES) != CGX_UNIQUE_ID_BYTES) throw std::runtime_error("pipe write");
                    close(w);
                }
            } else {
                size_t got = 0;
                while (got < CGX_UNIQUE_ID_BYTES) {
                    ssize_t k = read(rd_pipe, cfg.unique_id + got, CGX_UNIQUE_ID_BYTES - got);
                    if (k <= 0) throw std::runtime_error("rank 0 went away before sending the RCCL id");
                    got += static_cast<size_t>(k);
                }
                close(rd_pipe);
            }
        } else if (loopback > 1) {
            cfg.comm_mode = CGX_COMM_LOOPBACK;
            cfg.nranks = loopback;
        }
        cfg.profile_gemv = stats ? 1 : 0;
        const int psize = cfg.nranks;

        CGSolver solver(cfg);
        if (gen_form) solver.generate_lap2d_matrix(gen_n);   // cg_main.cc:31
        else solver.read_matrix(pos[0]);                     // code/CUDA/cg_main.cc:37
        const int n = solver.n();
        if (max_iter >= 0) solver.set_max_iter(max_iter);
        const double h = 1. / n;                             // cg_main.cc:45-46
        solver.init_source_term(h);
        std::vector<double> x_d(static_cast<size_t>(n), 0.);   // cg_main.cc:49-50

        auto t1 = clk::now();                                // only solve() is timed, cg_main.cc:53-55
        if (gen_form) solver.solve(x_d);
        else solver.solve(x_d.data(), legacy_nt, legacy_bw, legacy_t);
        second elapsed = clk::now() - t1;

        if (rank == 0) {
            std::ofstream outfile(out_file.c_str(), std::ios_base::app);
            if (gen_form) {
                outfile << n << "," << psize << "," << elapsed.count() << std::endl;   // cg_main.cc:62
            } else {
                std::cout << "Time for CG (dense solver)  = " << elapsed.count() << " [s]\n";   // code/CUDA/cg_main.cc:54
                outfile << legacy_nt << "," << legacy_bw << "," << elapsed.count() << std::endl;   // :59
            }
            if (stats) {
                const cgx_result &r = solver.last_result();
                const int it = r.iterations + (r.converged ? 1 : 0);   // loop bodies executed
                std::cerr << "cgsolver stats: n=" << n << " gpus=" << psize << " loop_bodies=" << it
                          << " loop_s=" << r.seconds_loop << " iterations_per_s=" << (r.seconds_loop > 0 ? it / r.seconds_loop : 0.)
                          << " gemv_ms_avg=" << r.gemv_ms_avg
                          << " gemv_GBps_per_gpu=" << (r.gemv_ms_avg > 0 ? r.gemv_bytes / (r.gemv_ms_avg * 1e-3) / 1e9 : 0.)
                          << " hbm_roofline_frac=" << (r.gemv_ms_avg > 0 ? r.gemv_bytes / (r.gemv_ms_avg * 1e-3) / 8.0e12 : 0.)
                          << std::endl;
            }
        }
    } catch (const std::exception &e) {
        std::cerr << "cgsolver (rank " << rank << "): " << e.what() << std::endl;
        rc = 1;
    }

    for (pid_t k : kids) {
        int st = 0;
        waitpid(k, &st, 0);
        if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) rc = 1;
    }
    return rc;
}
