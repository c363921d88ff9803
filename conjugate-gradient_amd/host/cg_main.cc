// cg_main.cc -- the `cgsolver` command line, both reference forms in one binary.
//
//   cgsolver N OUTFILE [MAXITER]                                  (code/MPI/cg_main.cc:13-69, README.md:63,71)
//   cgsolver FILE.mtx NUM_THREADS BLOCK_WIDTH true|false OUTFILE  (code/CUDA/cg_main.cc:16-63, README.md:98)
//
// If argv[1] parses completely as an integer it is the generator form, otherwise a Matrix-Market path.
// Where the reference took its process count from `srun -n P`, this takes `--gpus P` (or CG_NGPU):
// the binary forks P-1 children BEFORE touching the GPU, one process per MI355X, and the ranks meet over
// pipes (replaces MPI_Init, cg_main.cc:15-20): mailbox IPC handles for the direct-xGMI exchange, or an
// RCCL unique id.
// `--loopback P` runs P logical row blocks on one GPU (CI stand-in for a multi-GPU node).
// `--banded` (NOT a reference mode) holds the matrix as its non-zero diagonals: same recurrence and output, a
// banded mat-vec instead of the dense GEMV; refused if the matrix has more than 64 diagonals.
#include <signal.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <iostream>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
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

// ---- bounded wire-up ---------------------------------------------------------------------------------------
// Every stage of the multi-GPU wire-up that can block on a peer (the device probe's agreement, the mailbox handle
// exchange, cgx_p2p_import, the self-test, ncclCommInitRank) runs under a deadline.  MPI_Init either returns or the job
// is killed by the scheduler (code/MPI/cg_main.cc:15-20 under srun); here a stage that does not come back makes the rank
// print one line and leave with exit code 1 -- rank 0 first ends and reaps the other ranks -- so that a script like
// experiments/cg_mi355x.run can never hang on a dead peer.  Nothing is retried or restarted in place.
class StageWatchdog {
public:
    StageWatchdog(int rank, const std::vector<pid_t> *kids) : rank_(rank), kids_(kids), thread_([this] { run(); }) {}
    ~StageWatchdog()
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            quit_ = true;
        }
        cv_.notify_all();
        thread_.join();
    }
    void arm(const std::string &stage, double seconds)
    {
        std::lock_guard<std::mutex> lk(m_);
        stage_ = stage;
        seconds_ = seconds;
        deadline_ = std::chrono::steady_clock::now() + std::chrono::duration_cast<std::chrono::steady_clock::duration>(
                                                           std::chrono::duration<double>(seconds));
        armed_ = true;
        cv_.notify_all();
    }
    void disarm()
    {
        std::lock_guard<std::mutex> lk(m_);
        armed_ = false;
        cv_.notify_all();
    }

private:
    void run()
    {
        std::unique_lock<std::mutex> lk(m_);
        while (!quit_) {
            if (!armed_) {
                cv_.wait(lk);
                continue;
            }
            if (cv_.wait_until(lk, deadline_) == std::cv_status::timeout && armed_ && !quit_ &&
                std::chrono::steady_clock::now() >= deadline_) {
                // one write: the ranks expire together and share stderr, piecewise output would interleave
                std::ostringstream msg;
                msg << "cgsolver (rank " << rank_ << "): wire-up stage '" << stage_ << "' did not finish within " << seconds_
                    << " s; giving up\n";
                const std::string text = msg.str();
                if (write(2, text.data(), text.size()) < 0) {}
                if (kids_)   // rank 0: end the other ranks and reap them, so that nothing of the job is left behind
                    for (pid_t k : *kids_) {
                        kill(k, SIGKILL);
                        int st = 0;
                        waitpid(k, &st, 0);
                    }
                _exit(1);
            }
        }
    }
    const int rank_;
    const std::vector<pid_t> *kids_;
    std::mutex m_;
    std::condition_variable cv_;
    bool armed_ = false, quit_ = false;
    std::string stage_;
    double seconds_ = 0;
    std::chrono::steady_clock::time_point deadline_;
    std::thread thread_;
};

int usage(const char *prog)
{
    std::cerr << "Usage: " << prog << " N OUTFILE [MAXITER]            (generated matrix of size N)\n"
              << "       " << prog << " FILE.mtx NUM_THREADS BLOCK_WIDTH true|false OUTFILE\n"
              << "       " << prog << " FILE.mtx OUTFILE [MAXITER]       (matrix file, MPI-form output)\n"
              << "options: --gpus P (or CG_NGPU=P)  one process per MI355X\n"
              << "         --transport auto|p2p|p2p-tag|rccl  exchange: direct xGMI mailboxes (bytes handed over as payload + flag words,\n"
              << "                                  or as tagged words), RCCL; auto = p2p if its self-test passes on every rank, else RCCL\n"
              << "         --wireup-timeout S       seconds any one stage of the multi-GPU wire-up may take (default 120, or\n"
              << "                                  CG_WIREUP_TIMEOUT); a stage that does not come back ends the job with exit code 1\n"
              << "         --loopback P             P logical row blocks on one GPU\n"
              << "         --banded                 opt-in, not in the reference: store the non-zero diagonals only (<= 64)\n"
              << "         --stats                  also print iterations/s and K1 GB/s on stderr (event-times every K1:\n"
              << "                                  the seconds in OUTFILE are then a few % higher)" << std::endl;
    return 1;
}

}  // namespace

int main(int argc, char **argv)
{
    // The ranks map each other's mailboxes with hipIpcOpenMemHandle: this host driver only supports dmabuf IPC, which the
    // runtime uses when the legacy mode is switched off.  Must be in the environment before the first HIP call.
    setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);
    // ---- split options from the reference's positional arguments -------------------------------------
    std::vector<std::string> pos;
    int ngpu = 1, loopback = 0;
    bool stats = false, same_device = false, banded = false;
    std::string transport = "auto";
    std::string test_hang;   // --test-hang-stage
    double wireup_timeout = 120.0;
    if (const char *e = getenv("CG_WIREUP_TIMEOUT")) wireup_timeout = atof(e);
    if (const char *e = getenv("CG_NGPU")) ngpu = atoi(e);
    for (int i = 1; i < argc; ++i) {
        std::string a(argv[i]);
        if (a == "--gpus" && i + 1 < argc) ngpu = atoi(argv[++i]);
        else if (a == "--loopback" && i + 1 < argc) loopback = atoi(argv[++i]);
        else if (a == "--stats") stats = true;
        else if (a == "--banded") banded = true;
        else if (a == "--transport" && i + 1 < argc) transport = argv[++i];
        else if (a == "--wireup-timeout" && i + 1 < argc) wireup_timeout = atof(argv[++i]);
        else if (a == "--same-device") same_device = true;   // rehearsal: every rank on device 0 (p2p only)
        else if (a == "--test-hang-stage" && i + 1 < argc) test_hang = argv[++i];   // tests only, see below (not in the usage text)
        else if (a == "--cpu") {
            // SURVEY.md section 8(b) named a --cpu switch for the CPU restatement.  The product has no CPU path by design
            // (a silent fallback would void every parity claim); the restatement is test infrastructure under oracle/.
            std::cerr << argv[0] << ": --cpu is not available: libcgx has no CPU fallback (the CPU restatement of the\n"
                         "reference is the parity oracle under oracle/, used by tests/ and by bench.py's cpu_baseline only)\n";
            return 1;
        }
        else pos.push_back(a);
    }
    if (pos.empty()) return usage(argv[0]);   // cg_main.cc:22-26 (returns 1)
    if (ngpu < 1) ngpu = 1;

    int gen_n = 0;
    const bool gen_form = parse_int(pos[0], &gen_n);
    std::string out_file;
    int max_iter = -1, legacy_nt = 0, legacy_bw = 0;
    bool legacy_t = false, cuda_form = false;
    if (gen_form) {
        if (pos.size() < 2) return usage(argv[0]);
        out_file = pos[1];
        if (pos.size() >= 3) {                // cg_main.cc:37-42
            std::stringstream ss(pos[2]);
            ss >> max_iter;
        }
    } else if (pos.size() >= 5) {
        // CUDA form; the reference reads argv[2..5] unguarded (code/CUDA/cg_main.cc:21-33)
        if (!parse_int(pos[1], &legacy_nt) || !parse_int(pos[2], &legacy_bw)) return usage(argv[0]);
        legacy_t = (pos[3] == "true");
        out_file = pos[4];
        cuda_form = true;
    } else {
        // MPI form with a matrix file: `cgsolver FILE.mtx OUTFILE [MAXITER]`.  The reference's MPI read_matrix
        // never sets the sizes (cg.cc:191-202), so it could not do this; the CSV line is the MPI one.
        if (pos.size() < 2) return usage(argv[0]);
        out_file = pos[1];
        if (pos.size() >= 3) {
            std::stringstream ss(pos[2]);
            ss >> max_iter;
        }
    }

    // ---- one process per GPU: fork before any HIP/RCCL call -------------------------------------------
    // Rank 0 is the hub of a tiny control plane over pipes (what MPI_Init gave the reference): all-gather of a
    // few bytes and an all-min of one int, used to pass the RCCL id / the mailbox IPC handles around.
    int rank = 0;
    std::vector<int> up_rd, down_wr;   // rank 0: one pair per child
    int up_wr = -1, down_rd = -1;      // child: its own pair
    std::vector<pid_t> kids;
    if (ngpu > 1) {
        for (int r = 1; r < ngpu; ++r) {
            int up[2], down[2];
            if (pipe(up) != 0 || pipe(down) != 0) { perror("pipe"); return 1; }
            pid_t pid = fork();
            if (pid < 0) { perror("fork"); return 1; }
            if (pid == 0) {
                rank = r;
                up_wr = up[1];
                down_rd = down[0];
                close(up[0]);
                close(down[1]);
                for (int f : up_rd) close(f);
                for (int f : down_wr) close(f);
                up_rd.clear();
                down_wr.clear();
                kids.clear();
                break;
            }
            kids.push_back(pid);
            up_rd.push_back(up[0]);
            down_wr.push_back(down[1]);
            close(up[1]);
            close(down[0]);
        }
    }
    auto write_all = [](int fd, const void *buf, size_t n) {
        const char *p = static_cast<const char *>(buf);
        while (n) {
            ssize_t k = write(fd, p, n);
            if (k <= 0) throw std::runtime_error("control pipe write failed");
            p += k;
            n -= static_cast<size_t>(k);
        }
    };
    auto read_all = [](int fd, void *buf, size_t n) {
        char *p = static_cast<char *>(buf);
        while (n) {
            ssize_t k = read(fd, p, n);
            if (k <= 0) throw std::runtime_error("control pipe closed (a rank died)");
            p += k;
            n -= static_cast<size_t>(k);
        }
    };
    // every rank contributes `n` bytes; every rank receives all ngpu contributions in rank order
    auto allgather_bytes = [&](const unsigned char *mine, size_t n) {
        std::vector<unsigned char> all(static_cast<size_t>(ngpu) * n);
        if (ngpu == 1) {
            memcpy(all.data(), mine, n);
        } else if (rank == 0) {
            memcpy(all.data(), mine, n);
            for (int r = 1; r < ngpu; ++r) read_all(up_rd[r - 1], all.data() + static_cast<size_t>(r) * n, n);
            for (int r = 1; r < ngpu; ++r) write_all(down_wr[r - 1], all.data(), all.size());
        } else {
            write_all(up_wr, mine, n);
            read_all(down_rd, all.data(), all.size());
        }
        return all;
    };
    auto all_min = [&](int v) {
        unsigned char b = static_cast<unsigned char>(v ? 1 : 0);
        std::vector<unsigned char> all = allgather_bytes(&b, 1);
        for (unsigned char x : all)
            if (!x) return 0;
        return 1;
    };

    int rc = 0;
    try {
        cgx_config cfg;
        cgx_config_init(&cfg);
        cfg.profile_gemv = stats ? 1 : 0;
        cfg.matrix_format = banded ? CGX_MATRIX_BANDED : CGX_MATRIX_DENSE;
        std::unique_ptr<CGSolver> holder;
        if (ngpu > 1) {
            cfg.nranks = ngpu;
            cfg.rank = rank;
            cfg.device = same_device ? 0 : rank;
            // one deadline per stage; the watchdog thread starts here, i.e. after the fork (threads do not survive one)
            StageWatchdog dog(rank, rank == 0 ? &kids : nullptr);
            // --test-hang-stage "<stage>:<rank>" (tests/test_cli_wireup.py): that rank never comes back from that stage.  An
            // explicit argument, not an environment variable: nothing in a user's environment changes what cgsolver does.
            auto stage = [&](const std::string &name, const std::function<void()> &body) {
                struct Armed {          // the deadline ends with the stage, also when its body throws
                    StageWatchdog &d;
                    ~Armed() { d.disarm(); }
                } armed{dog};
                dog.arm(name, wireup_timeout);
                if (!test_hang.empty() && test_hang == name + ":" + std::to_string(rank)) pause();
                body();
            };
            stage("device probe", [&] {
                // every rank must own a usable GPU before any collective wire-up is attempted (a rank that cannot
                // create a context would otherwise leave the others blocked in ncclCommInitRank)
                cgx_config probe;
                cgx_config_init(&probe);
                probe.device = cfg.device;
                cgx_ctx *pc = nullptr;
                const int dev_ok = cgx_create(&pc, &probe) == CGX_OK;
                if (!dev_ok) std::cerr << "cgsolver (rank " << rank << "): " << cgx_last_error(nullptr) << std::endl;
                cgx_destroy(pc);
                if (!all_min(dev_ok)) throw std::runtime_error("--gpus " + std::to_string(ngpu) + ": not every rank has a usable MI355X");
            });
            bool have = false;
            // direct-xGMI mailboxes: create, exchange IPC handles, self-test; all ranks agree on the outcome.  Two forms of
            // the per-iteration exchange, each with its own device code in the self-test: payload + flag words (what auto
            // uses), or tagged 8-byte words (no flags, no fences; on request); RCCL if the mailboxes do not work between these devices.
            std::vector<std::string> forms;
            // auto = the flag form, then RCCL.  The tagged-word form is faster on one GPU (6.9 against 10.2 us per exchange) but
            // rests on an 8-byte half of a 16-byte write-through store arriving untorn over xGMI, which no multi-GPU run has
            // shown yet: it is used on request only (--transport p2p-tag) until one has.
            if (transport == "auto") forms = {"p2p"};
            else if (transport == "p2p-tag" || transport == "p2p") forms = {transport};
            else if (transport != "rccl") throw std::runtime_error("--transport must be auto, p2p-tag, p2p or rccl");
            for (const std::string &form : forms) {
                cfg.comm_mode = CGX_COMM_P2P;
                cfg.p2p_tagged = form == "p2p-tag" ? 1 : 0;
                int ok = 1;
                std::vector<unsigned char> all;
                stage("mailbox allocation (" + form + ")", [&] {
                    unsigned char handle[CGX_IPC_HANDLE_BYTES] = {0};
                    try {
                        holder.reset(new CGSolver(cfg));
                        if (cgx_p2p_export(holder->context(), handle) != CGX_OK) ok = 0;
                    } catch (const std::exception &e) {
                        std::cerr << "cgsolver (rank " << rank << "): " << form << " unavailable: " << e.what() << std::endl;
                        ok = 0;
                    }
                    all = allgather_bytes(handle, CGX_IPC_HANDLE_BYTES);
                    ok = all_min(ok);
                });
                if (ok) {
                    stage("opening the peers' mailboxes (" + form + ")", [&] {
                        if (cgx_p2p_import(holder->context(), all.data()) != CGX_OK) ok = 0;
                        ok = all_min(ok);
                    });
                    if (ok)
                        stage("mailbox self-test (" + form + ")", [&] {
                            int st_ok = 0;
                            if (cgx_p2p_selftest(holder->context(), 32, &st_ok) != CGX_OK) st_ok = 0;
                            have = all_min(st_ok) != 0;
                        });
                }
                if (have) break;
                holder.reset();
                if (rank == 0) std::cerr << "cgsolver: direct peer exchange (" << form << ") unavailable" << std::endl;
            }
            if (!have && (transport == "p2p" || transport == "p2p-tag"))
                throw std::runtime_error("--transport " + transport + ": mailboxes unavailable or self-test failed");
            if (!have && !forms.empty() && rank == 0) std::cerr << "cgsolver: using RCCL" << std::endl;
            cfg.p2p_tagged = 0;
            if (!have) {
                cfg.comm_mode = CGX_COMM_RCCL;
                stage("ncclCommInitRank", [&] {
                    unsigned char uid[CGX_UNIQUE_ID_BYTES] = {0};
                    if (rank == 0 && cgx_comm_unique_id(uid) != CGX_OK)
                        throw std::runtime_error(std::string("cgx_comm_unique_id: ") + cgx_last_error(nullptr));
                    std::vector<unsigned char> all = allgather_bytes(uid, CGX_UNIQUE_ID_BYTES);
                    memcpy(cfg.unique_id, all.data(), CGX_UNIQUE_ID_BYTES);   // rank 0's id
                    holder.reset(new CGSolver(cfg));
                });
            }
        } else {
            if (loopback > 1) {
                cfg.comm_mode = CGX_COMM_LOOPBACK;
                cfg.nranks = loopback;
            }
            holder.reset(new CGSolver(cfg));
        }
        const int psize = cfg.nranks;
        CGSolver &solver = *holder;

        if (gen_form) solver.generate_lap2d_matrix(gen_n);   // cg_main.cc:31
        else solver.read_matrix(pos[0]);                     // code/CUDA/cg_main.cc:37
        const int n = solver.n();
        if (max_iter >= 0) solver.set_max_iter(max_iter);
        const double h = 1. / n;                             // cg_main.cc:45-46
        solver.init_source_term(h);
        std::vector<double> x_d(static_cast<size_t>(n), 0.);   // cg_main.cc:49-50

        auto t1 = clk::now();                                // only solve() is timed, cg_main.cc:53-55
        if (cuda_form) solver.solve(x_d.data(), legacy_nt, legacy_bw, legacy_t);
        else solver.solve(x_d);
        second elapsed = clk::now() - t1;

        if (rank == 0) {
            std::ofstream outfile(out_file.c_str(), std::ios_base::app);
            if (!cuda_form) {
                outfile << n << "," << psize << "," << elapsed.count() << std::endl;   // cg_main.cc:62
            } else {
                std::cout << "Time for CG (dense solver)  = " << elapsed.count() << " [s]\n";   // code/CUDA/cg_main.cc:54
                outfile << legacy_nt << "," << legacy_bw << "," << elapsed.count() << std::endl;   // :59
            }
            if (stats) {
                const cgx_result &r = solver.last_result();
                const int it = r.iterations + (r.converged ? 1 : 0);   // loop bodies executed
                int plan[CGX_GEMV_PLAN_INTS] = {0};
                (void)cgx_get_gemv_plan(solver.context(), 0, plan);     // variant 4 / 5: the loop ran as one persistent kernel
                long long rec[CGX_RESIDENT_RECORD_INTS] = {0};
                (void)cgx_get_resident_record(solver.context(), rec);
                const bool persistent = plan[0] == 4 || plan[0] == 5;
                std::cerr << "cgsolver stats: n=" << n << " gpus=" << psize << " loop_bodies=" << it
                          << " loop_s=" << r.seconds_loop << " iterations_per_s=" << (r.seconds_loop > 0 ? it / r.seconds_loop : 0.)
                          << " format=" << (banded ? "banded" : "dense")
                          << " loop=" << (plan[0] == 4 ? "resident-kernel" : plan[0] == 5 ? "streaming-persistent-kernel" : "per-launch");
                if (persistent) {
                    // no K1 launches to time: the whole loop against the algorithmic bytes of its GEMVs, and what its waits cost
                    // (cgx_get_resident_record; ticks of 10 ns)
                    std::cerr << " gemv_ms_avg=n/a loop_GBps=" << (r.seconds_loop > 0 ? it * r.gemv_bytes / r.seconds_loop / 1e9 : 0.)
                              << " loop_hbm_roofline_frac=" << (r.seconds_loop > 0 ? it * r.gemv_bytes / r.seconds_loop / 8.0e12 : 0.)
                              << " launches=" << rec[3] << " watch_repeats=" << rec[1] << " gather_repeats=" << rec[2]
                              << " first_wait_us=" << rec[4] / 100.0 << " longest_later_wait_us=" << rec[5] / 100.0
                              << " first_wait_us_any_wg=" << rec[6] / 100.0 << " longest_later_wait_us_any_wg=" << rec[7] / 100.0;
                } else {
                    std::cerr << " gemv_ms_avg=" << r.gemv_ms_avg
                              << " gemv_GBps_per_gpu=" << (r.gemv_ms_avg > 0 ? r.gemv_bytes / (r.gemv_ms_avg * 1e-3) / 1e9 : 0.)
                              << " hbm_roofline_frac=" << (r.gemv_ms_avg > 0 ? r.gemv_bytes / (r.gemv_ms_avg * 1e-3) / 8.0e12 : 0.);
                }
                std::cerr << " persistent_launches_redone_per_launch=" << rec[8] << std::endl;
            }
        }
    } catch (const std::exception &e) {
        // one write per message: the ranks share stderr and tend to fail together
        const std::string msg = "cgsolver (rank " + std::to_string(rank) + "): " + e.what() + "\n";
        if (write(2, msg.data(), msg.size()) < 0) {}
        rc = 1;
    }

    // rank 0 reaps the other ranks; if it failed itself they may be blocked on it (a pipe read, a collective): end them
    for (pid_t k : kids) {
        if (rc != 0) kill(k, SIGTERM);
        int st = 0;
        waitpid(k, &st, 0);
        if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) rc = 1;
    }
    return rc;
}
