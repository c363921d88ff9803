// cgx_internal.h -- what the translation units of libcgx share: the context, the per-shard buffers, the error
// macros and the internal helpers.  Nothing here is part of the C ABI (include/cgx.h).
#pragma once

#include "../../include/cgx.h"

#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "cgx_kernels.h"
#include "cgx_rccl.h"

namespace cgxi {

using cgx::Scalars;

struct Shard {
    int rank = 0;
    int row0 = 0;
    int rows = 0;
    double *A = nullptr;         // rows x lda, row-major, pad columns zero (CGX_MATRIX_DENSE)
    double *dia_vals = nullptr;  // CGX_MATRIX_BANDED: ndiag x dia.ld, the non-zero diagonals of the row block
    cgx::DiaView dia{};
    double *b_full = nullptr;    // n doubles: b is replicated like r (the reference builds the full b on every rank, cg.cc:218-234)
    // One GPU, dense storage (where the persistent kernels of cgx_resident.hip / cgx_stream.hip may run the loop): x, rbuf, p[1] and
    // sc are carved out of ONE block of solver state (cgx::state_off_*), and there are two such blocks -- a persistent launch
    // reads state[cur] and writes state[cur ^ 1], and only a launch that came back clean makes the written block the current one
    // (bind_state).  Every other configuration allocates the four buffers one by one and state[] stays null.
    double *state[2] = {nullptr, nullptr};
    int cur = 0;
    double *x = nullptr;         // rows
    double *p[2] = {nullptr, nullptr};   // lda doubles each: the replicated p (cg.cc:57), ping-pong over iterations
    double *apg = nullptr;       // nranks * S doubles: exchanged segments [Ap slice | p.Ap partials] (cgx::SegView apv)
    double *rbuf = nullptr;      // lda + kSlots doubles: the replicated r and its scalars (cgx::SegView rv, one segment)
    double *ap_parts = nullptr;  // plan.split > 1: split x seg_Sr doubles, the column pieces of the fused K1's Ap
    double *k1_scratch = nullptr;   // chunked exchange: where K1's per-workgroup p.Ap partials go (only the gemv probe reads them;
                                    // the segment tail then holds one partial per chunk of the slice instead)
    double *partials = nullptr;  // scratch: per-workgroup partial sums of K3 and of the setup kernels
    Scalars *sc = nullptr;
    double *gathered = nullptr;  // kMaxRanks * kSlots doubles (DEBUG scalars of all ranks)
    cgx::GemvPlan plan{};
    cgx::SegView apv{}, rv{};
    int npartials = 0;
    double *Ap() const { return apg + (size_t)rank * apv.S; }          // this shard's Ap slice (K1 output)
    double *tail() const { return Ap() + apv.Sr; }                      // this shard's segment tail: the p.Ap partials that travel
    double *k1_part() const { return k1_scratch ? k1_scratch : tail(); }   // K1's per-workgroup p.Ap partials
};


}  // namespace cgxi

struct cgx_ctx {
    cgx_config cfg{};
    int device = 0;
    hipStream_t stream = nullptr;
    int nranks = 1;
    int m = 0, n = 0;
    long lda = 0;
    int max_iter = 0;
    double tol = 1e-10;   // m_tolerance, code/MPI/cg.hh:56
    std::vector<int> start_rows, num_rows;
    std::vector<cgxi::Shard> shards;   // 1 (SELF / RCCL) or nranks (LOOPBACK)
    std::vector<double> b_host;
    bool have_matrix = false, have_b = false;
    bool banded = false;         // cfg.matrix_format == CGX_MATRIX_BANDED (opt-in, not the reference's storage)

    // RCCL
    const cgx::RcclApi *rccl = nullptr;
    ncclComm_t comm = nullptr;
    // direct peer exchange (CGX_COMM_P2P)
    unsigned char *mailbox = nullptr;        // own fine-grained mailbox
    size_t mailbox_bytes = 0;
    bool mailbox_on_host = false;            // test only (cgx_probe_p2p_mailbox_to_host): the mailbox lives in pinned host memory
    // test only (cgx_probe_p2p_host_mailboxes): EVERY rank's mailbox is a POSIX shared-memory segment registered with the
    // runtime; host_maps[q] = this process's mapping of rank q's segment (own included), nullptr otherwise
    bool mailbox_shm = false;
    std::string shm_prefix;
    void *host_maps[cgx::kMaxRanks] = {nullptr};
    bool p2p_ready = false;                  // peers' mailboxes are mapped
    cgx::MailboxView mv{};
    unsigned long long p2p_epoch[cgx::kP2pChannels] = {0, 0, 0};
    int *d_p2p_err = nullptr;                // device word set when a bounded wait expired
    long long p2p_timeout_ticks = 0;         // 100 MHz wall-clock ticks

    int seg_S = 0, seg_Sr = 0;   // exchange segment geometry (equal for all ranks)
    int npart = 0;               // p.Ap partials per rank in the segment tail: K1's workgroups (max grid over ranks), or, chunked,
                                 // the chunks of a slice (cgx::chunks_per_rank(seg_Sr))
    bool chunked = false;        // the tail holds one partial per chunk (cgx_kernels.hip "Chunks"): every multi-rank dense
                                 // run and every fused P2P update; else K1's own partials (one GPU; banded storage)
    int resident_limit = 0;      // > 0: test override of the co-residency bound of the fused P2P update (cgx_probe_set_resident_limit)

    // the persistent kernels (cgx_resident.hip: n <= 4096, A on the chip; cgx_stream.hip: n <= 16384, A streamed): one GPU, dense
    int cus = 0;                             // compute units of the device
    size_t lds_per_cu = 0;                   // LDS bytes a workgroup can be given
    bool resident = false;                   // the current problem runs the loop as one persistent kernel
    cgx::ResidentPlan rplan{};
    unsigned long long *res_xbuf = nullptr;  // exchange buffer of the resident kernel's workgroups (tagged words)
    size_t res_xbuf_bytes = 0;
    unsigned long long res_epoch = 0;        // epochs handed out so far (monotonic over the life of the context)
    int *d_res_err = nullptr;                // device word raised when a wait inside the resident kernel expired
    long long res_timeout_ticks = 0;
    int res_lock_fd = -1;                    // advisory lock file of the device: one resident grid at a time (see resident_steps)
    bool res_lock_gave_up = false;           // a wait for that lock ran into its bound once: later launches do not wait again
    int res_mute_wg = -1;                    // test only (cgx_probe_resident_test): workgroup that skips its first publish, next launch
    bool res_forced = false;                 // gemv_variant 40000: a launch whose waits expire is an error, not a fallback
    cgx::ResidentTail *h_res_tail = nullptr; // pinned: what a persistent launch reports (written by the kernel itself, cgx_kernels.h)
    unsigned res_stamp = 0;                  // the number of the most recent persistent launch (never 0 once one has run)
    long long res_rec[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // the waits of the current solve's launches, summed up (cgx_get_resident_record)
    bool lean = false;                       // the solve under way began with the one-kernel set-up (zero initial guess, persistent kernel)
    bool oneshot = false;                    // cgx_solve is running begin / steps / end in one go: the end may be enqueued behind the loop
    bool end_enqueued = false;               // ... and has been (verification GEMV + end kernel are behind the persistent launch, h_stage is filled)
    long long res_fallbacks = 0;             // persistent launches of this context whose waits expired and that were redone on the per-launch path

    // loopback pointer tables (device)
    double **d_gathered_ptrs = nullptr;
    cgx::Scalars **d_scalar_ptrs = nullptr;

    // solve state
    bool in_solve = false;
    int k = 0;              // iterations enqueued so far
    bool done = false;
    int k_final = 0;
    int *h_flags = nullptr;   // pinned: 2 polling slots + 1 for read_flags_sync, each {done, k_final}
    double *h_stage = nullptr;   // pinned, n + 16 doubles: x0 in / x out go through it, so that solve() never waits for the
                                 // runtime's first-use set-up of pageable copies (8 ms inside the reference's timing
                                 // window, measured); nullptr above 8 Mi rows (then the copies are direct)
    hipEvent_t flag_ev[2] = {nullptr, nullptr};
    double t_begin = 0, t_loop = 0;

    // K1 timing
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    double gemv_ms_sum = 0, gemv_ms_min = 0, gemv_ms_max = 0;
    long long gemv_launches = 0, gemv_discarded = 0;
    long long gemv_seq = 0;              // K1 launches of the current cgx_solve_steps call
    std::vector<float> gemv_samples;     // their durations (ms), most recent steps call
    bool gemv_timed_last = false;        // the most recent K1 launch was event-timed (the update kernel of that iteration follows suit)
    std::vector<hipEvent_t> upd_pool;    // cfg.profile_update: event pairs around the update kernels of the timed iterations
    size_t upd_used = 0;
    std::vector<float> upd_samples;      // their durations (ms), most recent steps call
    hipEvent_t steps_ev[2] = {nullptr, nullptr};   // markers around the kernels of the most recent steps call (profiling on)
    bool steps_ev_pending = false;
    double steps_device_ms = 0;

    int fault_after = -1;     // >= 0: HIP_TRY calls left until one is made to fail (cgx_probe_set_fault_after, error-path tests only)

    std::string err;
};


namespace cgxi {

extern thread_local std::string g_create_error;   // error of the last failed cgx_create on this thread

double wall_now();
cgx_status fail(cgx_ctx *ctx, cgx_status st, const std::string &msg);

// Fault injection for the error-path tests (tools/leak_check.py, tests): after cgx_probe_set_fault_after(ctx, N) the
// (N+1)-th HIP call of the context made through HIP_TRY is not made and reports hipErrorUnknown instead.  Only the explicit
// probe call arms it: nothing in a user's environment can make a production call fail.
inline bool fault_due(cgx_ctx *ctx)
{
    if (!ctx || ctx->fault_after < 0) return false;
    if (ctx->fault_after == 0) { ctx->fault_after = -1; return true; }
    --ctx->fault_after;
    return false;
}

// An early return must not leave work of this context in flight: several entry points enqueue asynchronous copies into
// their own locals (a result struct on the stack, a scratch vector) and synchronise further down; if a call in between --
// or, under fault injection, the synchronisation itself -- fails, the copy would land in memory that is gone.  So the
// failure path drains the context's stream first (best effort; the error that is reported is the original one).
inline void quiesce(cgx_ctx *ctx)
{
    if (ctx && ctx->stream) (void)hipStreamSynchronize(ctx->stream);
}

#define HIP_TRY(ctx, call)                                                                              \
    do {                                                                                                \
        hipError_t e_ = cgxi::fault_due(ctx) ? hipErrorUnknown : (call);                                \
        if (e_ != hipSuccess) {                                                                         \
            cgx_status st_ = (e_ == hipErrorOutOfMemory) ? CGX_ERR_OOM                                  \
                             : (e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice) ? CGX_ERR_NO_DEVICE \
                                                                                       : CGX_ERR_HIP;   \
            cgxi::quiesce(ctx);                                                                         \
            return fail(ctx, st_, std::string(#call) + ": " + hipGetErrorString(e_));                   \
        }                                                                                               \
    } while (0)

#define NCCL_TRY(ctx, call)                                                                             \
    do {                                                                                                \
        ncclResult_t r_ = (call);                                                                       \
        if (r_ != ncclSuccess) {                                                                        \
            cgxi::quiesce(ctx);                                                                         \
            return fail(ctx, CGX_ERR_RCCL, std::string(#call) + ": " + (ctx)->rccl->GetErrorString(r_)); \
        }                                                                                               \
    } while (0)

#define CGX_TRY(call)                      \
    do {                                   \
        cgx_status s_ = (call);            \
        if (s_ != CGX_OK) return s_;       \
    } while (0)


// Device allocations and events of ONE function: released on every return path (hipFree waits for the device, so work
// still queued on a buffer when an error return unwinds is finished first).
struct DeviceScratch {
    std::vector<void *> ptrs;
    std::vector<hipEvent_t> events;
    DeviceScratch() = default;
    DeviceScratch(const DeviceScratch &) = delete;
    DeviceScratch &operator=(const DeviceScratch &) = delete;
    ~DeviceScratch()
    {
        for (void *p : ptrs) (void)hipFree(p);
        for (hipEvent_t e : events) (void)hipEventDestroy(e);
    }
    template <class T>
    hipError_t alloc(T **out, size_t bytes)
    {
        void *p = nullptr;
        const hipError_t e = hipMalloc(&p, bytes);
        if (e == hipSuccess) ptrs.push_back(p);
        *out = static_cast<T *>(p);
        return e;
    }
    hipError_t event(hipEvent_t *out)
    {
        const hipError_t e = hipEventCreate(out);
        if (e == hipSuccess) events.push_back(*out);
        return e;
    }
};

// cgx_context.cpp
void partition_rows(int N, int psize, int *start_rows, int *num_rows);
void free_problem(cgx_ctx *ctx);
void bind_state(cgxi::Shard &s, long lda);     // point x / rbuf / p[1] / sc (and rv.base) into state[cur]
long p2p_fixed_prefix(int nranks);
cgx_status setup_problem(cgx_ctx *ctx, int n);       // allocate the shards of an n x n problem (contents: caller)

// cgx_matrix.cpp
cgx_status alloc_dia(cgx_ctx *ctx, Shard &s, const std::vector<int> &offs);
// A Matrix-Market coordinate file as MatrixCOO::read leaves it (matrix_coo.cc:7-60): sizes, symmetry, 0-based entries in
// file order.  Host only; parsed on `nthreads` threads.
struct MtxEntries {
    int m = 0, n = 0, nz = 0;
    bool sym = false;
    std::vector<int> I, J;
    std::vector<double> a;
};
cgx_status parse_matrix_market(const char *path, MtxEntries *out, std::string *err, int nthreads, bool header_only = false);
int default_parse_threads();   // CGX_MTX_THREADS, else the host's hardware threads, at most 16

// cgx_solve.cpp
cgx_status p2p_allgather(cgx_ctx *ctx, int chan, const double *src, int count, double *dst, long dst_stride, int copy_self,
                         int tail_off = 0, int tail_n = 0, int sum_off = 0);
cgx_status gather_scalars(cgx_ctx *ctx);
cgx_status gather_segments(cgx_ctx *ctx, bool with_tail);
cgx_status run_gemv_plain(cgx_ctx *ctx, Shard &s, const double *v_full);
cgx_status check_p2p_error(cgx_ctx *ctx);
cgx_status resident_steps(cgx_ctx *ctx, int nsteps, int *redo);
void reset_gemv_stats(cgx_ctx *ctx);

}  // namespace cgxi
