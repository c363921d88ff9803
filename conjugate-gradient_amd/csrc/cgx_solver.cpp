// cgx_solver.cpp -- context, row-block shards, collectives and the CG driver loop behind include/cgx.h.
//
// Reference path: CGSolver::solve, code/MPI/cg.cc:38-156 (citations are file:line under /root/reference).
// Design (MI355X-first, not a translation):
//   * the row block of A, b, x, r, Ap and the replicated p live in HBM for the life of the problem;
//   * rsold/rsnew/alpha/beta and the convergence flag stay on the device (cgx::Scalars): the host only
//     enqueues kernels and polls one int every `check_every` iterations, so the stream never drains;
//   * after convergence every kernel of the remaining enqueued iterations exits at its first
//     instruction, which reproduces the reference's `break` (cg.cc:120-121) exactly;
//   * one iteration = two kernels (K1 fused GEMV, K3 x/r update) and ONE exchange: an in-place all-gather of
//     equal segments [Ap slice | p.Ap partials] replaces MPI_Allreduce(p.Ap), MPI_Allreduce(r.r) and
//     MPI_Allgatherv(p): r and p are replicated, every rank updates all of r from the gathered Ap, reduces r.r
//     over all n rows in one fixed order (bit-identical everywhere) and forms p = r + beta p inside the next K1.
#include "../../include/cgx.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cctype>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "cgx_kernels.h"
#include "cgx_rccl.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

using cgx::Scalars;

namespace {

thread_local std::string g_create_error;

double wall_now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct Shard {
    int rank = 0;
    int row0 = 0;
    int rows = 0;
    double *A = nullptr;         // rows x lda, row-major, pad columns zero (CGX_MATRIX_DENSE)
    double *dia_vals = nullptr;  // CGX_MATRIX_BANDED: ndiag x dia.ld, the non-zero diagonals of the row block
    cgx::DiaView dia{};
    double *b_full = nullptr;    // n doubles: b is replicated like r (the reference builds the full b on every rank, cg.cc:218-234)
    double *x = nullptr;         // rows
    double *p[2] = {nullptr, nullptr};   // lda doubles each: the replicated p (cg.cc:57), ping-pong over iterations
    double *apg = nullptr;       // nranks * S doubles: exchanged segments [Ap slice | p.Ap partials] (cgx::SegView apv)
    double *rbuf = nullptr;      // lda + kSlots doubles: the replicated r and its scalars (cgx::SegView rv, one segment)
    double *partials = nullptr;  // scratch: per-workgroup partial sums of K3 and of the setup kernels
    Scalars *sc = nullptr;
    double *gathered = nullptr;  // kMaxRanks * kSlots doubles (DEBUG scalars of all ranks)
    cgx::GemvPlan plan{};
    cgx::SegView apv{}, rv{};
    int npartials = 0;
    double *Ap() const { return apg + (size_t)rank * apv.S; }          // this shard's Ap slice (K1 output)
    double *k1_part() const { return Ap() + apv.Sr; }                   // this shard's p.Ap partials (segment tail)
};

}  // namespace

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
    std::vector<Shard> shards;   // 1 (SELF / RCCL) or nranks (LOOPBACK)
    std::vector<double> b_host;
    bool have_matrix = false, have_b = false;
    bool banded = false;         // cfg.matrix_format == CGX_MATRIX_BANDED (opt-in, not the reference's storage)

    // RCCL
    const cgx::RcclApi *rccl = nullptr;
    ncclComm_t comm = nullptr;
    // direct peer exchange (CGX_COMM_P2P)
    unsigned char *mailbox = nullptr;        // own fine-grained mailbox
    size_t mailbox_bytes = 0;
    bool p2p_ready = false;                  // peers' mailboxes are mapped
    cgx::MailboxView mv{};
    unsigned long long p2p_epoch[cgx::kP2pChannels] = {0, 0, 0};
    int *d_p2p_err = nullptr;                // device word set when a bounded wait expired
    long long p2p_timeout_ticks = 0;         // 100 MHz wall-clock ticks

    int seg_S = 0, seg_Sr = 0;   // exchange segment geometry (equal for all ranks)
    int npart = 0;               // K1 partials per rank in exchange 1 (max grid over ranks)

    // loopback pointer tables (device)
    double **d_gathered_ptrs = nullptr;
    Scalars **d_scalar_ptrs = nullptr;

    // solve state
    bool in_solve = false;
    int k = 0;              // iterations enqueued so far
    bool done = false;
    int k_final = 0;
    int *h_flags = nullptr;   // pinned: 2 slots x {done, k_final}
    hipEvent_t flag_ev[2] = {nullptr, nullptr};
    double t_begin = 0, t_loop = 0;

    // K1 timing
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    double gemv_ms_sum = 0, gemv_ms_min = 0;
    long long gemv_launches = 0;
    long long gemv_seq = 0;

    std::string err;
};

namespace {

cgx_status fail(cgx_ctx *ctx, cgx_status st, const std::string &msg)
{
    if (ctx) ctx->err = msg;
    else g_create_error = msg;
    return st;
}

#define HIP_TRY(ctx, call)                                                                              \
    do {                                                                                                \
        hipError_t e_ = (call);                                                                         \
        if (e_ != hipSuccess) {                                                                         \
            cgx_status st_ = (e_ == hipErrorOutOfMemory) ? CGX_ERR_OOM                                  \
                             : (e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice) ? CGX_ERR_NO_DEVICE \
                                                                                       : CGX_ERR_HIP;   \
            return fail(ctx, st_, std::string(#call) + ": " + hipGetErrorString(e_));                   \
        }                                                                                               \
    } while (0)

#define NCCL_TRY(ctx, call)                                                                             \
    do {                                                                                                \
        ncclResult_t r_ = (call);                                                                       \
        if (r_ != ncclSuccess)                                                                          \
            return fail(ctx, CGX_ERR_RCCL, std::string(#call) + ": " + (ctx)->rccl->GetErrorString(r_)); \
    } while (0)

#define CGX_TRY(call)                      \
    do {                                   \
        cgx_status s_ = (call);            \
        if (s_ != CGX_OK) return s_;       \
    } while (0)

void partition_rows(int N, int psize, int *start_rows, int *num_rows)
{
    // CGSolver::partition_matrix, code/MPI/cg.cc:236-268: floor(N/psize) rows per rank, remainder on the last.
    const int n_loc = (psize > 0) ? N / psize : N;
    int i0 = 0;
    for (int r = 0; r + 1 < psize; ++r) {
        start_rows[r] = i0;
        num_rows[r] = n_loc;
        i0 += n_loc;
    }
    start_rows[psize - 1] = i0;
    num_rows[psize - 1] = N - i0;
}

void free_shard(Shard &s)
{
    (void)hipFree(s.A);
    (void)hipFree(s.dia_vals);
    (void)hipFree(s.b_full);
    (void)hipFree(s.x);
    (void)hipFree(s.p[0]);
    (void)hipFree(s.p[1]);
    (void)hipFree(s.apg);
    (void)hipFree(s.rbuf);
    (void)hipFree(s.partials);
    (void)hipFree(s.sc);
    (void)hipFree(s.gathered);
    s = Shard{};
}

void free_problem(cgx_ctx *ctx)
{
    for (auto &s : ctx->shards) free_shard(s);
    ctx->shards.clear();
    (void)hipFree(ctx->d_gathered_ptrs);
    (void)hipFree(ctx->d_scalar_ptrs);
    ctx->d_gathered_ptrs = nullptr;
    ctx->d_scalar_ptrs = nullptr;
    ctx->have_matrix = ctx->have_b = false;
    ctx->in_solve = false;
}

// Mailbox bytes before the segment channel: flag words, channel 0 (16-B slots) and channel 2 (kSlots doubles).
long p2p_fixed_prefix(int nranks)
{
    return (long)cgx::kP2pChannels * cgx::kMaxRanks * cgx::kP2pFlagStride + 2L * nranks * 16 +
           2L * nranks * (long)cgx::kSlots * 8;
}

long default_lda(const cgx_ctx *ctx, int n)
{
    long lda = ((long)n + 15) / 16 * 16;   // every row starts on a 128-B line
    int pad = ctx->cfg.lda_pad;
    if (pad < 0) {
        const char *e = getenv("CGX_LDA_PAD");
        pad = e ? atoi(e) : 16;   // +128 B per row: de-aliases the HBM channels when N*8 is a power of two
    }
    if (pad > 0) lda += (pad + 1) / 2 * 2;
    return lda;
}

// Allocate the shards for an n x n problem (matrix contents are filled by the caller).
cgx_status setup_problem(cgx_ctx *ctx, int n)
{
    if (n <= 0) return fail(ctx, CGX_ERR_BAD_ARG, "matrix size must be positive");
    if (n > (1 << 30)) return fail(ctx, CGX_ERR_UNSUPPORTED, "matrix size above 2^30 (indices are int, like the reference's)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!ctx->shards.empty() && ctx->n == n && ctx->lda == default_lda(ctx, n)) {
        // Same geometry as the current problem: keep every buffer.  (Freeing and re-allocating a multi-GiB matrix
        // can land on fragmented memory and cost ~3 % of K1; measured in bench.py's transport calibration.)
        ctx->max_iter = n;
        ctx->have_matrix = ctx->have_b = false;
        ctx->in_solve = false;
        for (auto &s : ctx->shards) {
            HIP_TRY(ctx, hipMemsetAsync(s.p[0], 0, (size_t)ctx->lda * sizeof(double), ctx->stream));
            HIP_TRY(ctx, hipMemsetAsync(s.p[1], 0, (size_t)ctx->lda * sizeof(double), ctx->stream));
            HIP_TRY(ctx, hipMemsetAsync(s.sc, 0, sizeof(Scalars), ctx->stream));
            HIP_TRY(ctx, hipMemsetAsync(s.apg, 0, (size_t)ctx->nranks * ctx->seg_S * sizeof(double), ctx->stream));
        }
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        return CGX_OK;
    }
    free_problem(ctx);
    ctx->m = ctx->n = n;
    ctx->max_iter = n;   // m_maxIter = size, code/MPI/cg.cc:172
    ctx->lda = default_lda(ctx, n);
    ctx->start_rows.assign(ctx->nranks, 0);
    ctx->num_rows.assign(ctx->nranks, 0);
    partition_rows(n, ctx->nranks, ctx->start_rows.data(), ctx->num_rows.data());
    int max_rows = 0;
    for (int q = 0; q < ctx->nranks; ++q) max_rows = std::max(max_rows, ctx->num_rows[q]);
    ctx->seg_Sr = std::max((max_rows + 1) / 2 * 2, 2);   // Ap slice, padded to an even count

    int variant = ctx->cfg.gemv_variant;
    if (variant <= 0) {
        const char *e = getenv("CGX_GEMV_VARIANT");
        if (e) variant = atoi(e);
    }
    auto plan_for = [&](int rows) { return ctx->banded ? cgx::plan_dia(rows) : cgx::plan_gemv(variant, rows, (int)ctx->lda); };
    ctx->npart = 1;
    for (int q = 0; q < ctx->nranks; ++q) ctx->npart = std::max(ctx->npart, plan_for(ctx->num_rows[q]).grid);
    if (ctx->cfg.comm_mode == CGX_COMM_P2P && n > 256 * cgx::kMaxVectorGrid)
        return fail(ctx, CGX_ERR_UNSUPPORTED, "CGX_COMM_P2P handles at most 262144 rows (the update kernel with the exchange "
                                              "inside works one row per thread); use CGX_COMM_RCCL");
    if (ctx->cfg.comm_mode == CGX_COMM_P2P) {
        // mailbox layout of this problem: flags, then per channel [2 parities][nranks] slots
        // The small fixed-size channels come first, so that their place never depends on the problem; the
        // segment channel (1), whose slot size does, comes last.  A re-layout for a new problem size is then
        // safe without a launcher barrier: the last exchange of a solve is on channel 2, and a rank can finish
        // it only after every peer has pushed its channel-2 data, i.e. after every peer is done with channel 1.
        const long slot[cgx::kP2pChannels] = {16, ((long)(ctx->seg_Sr + 1) * 8 + 15) / 16 * 16, (long)cgx::kSlots * 8};
        long off = p2p_fixed_prefix(ctx->nranks);
        ctx->mv.data_off[0] = (long)cgx::kP2pChannels * cgx::kMaxRanks * cgx::kP2pFlagStride;
        ctx->mv.slot_bytes[0] = slot[0];
        ctx->mv.data_off[2] = ctx->mv.data_off[0] + 2L * ctx->nranks * slot[0];
        ctx->mv.slot_bytes[2] = slot[2];
        ctx->mv.data_off[1] = off;
        ctx->mv.slot_bytes[1] = slot[1];
        off += 2L * ctx->nranks * slot[1];
        if ((size_t)off > ctx->mailbox_bytes)
            return fail(ctx, CGX_ERR_P2P, "mailbox too small for this problem: need " + std::to_string(off) +
                                              " bytes (raise cgx_config.p2p_mailbox_kib)");
    }
    const int seg_tail = (ctx->npart + 1 + 1) / 2 * 2;   // npart partials + 1 slot for a rank's folded sum, even
    ctx->seg_S = ctx->seg_Sr + seg_tail;
    const int nlocal = (ctx->cfg.comm_mode == CGX_COMM_LOOPBACK) ? ctx->nranks : 1;
    ctx->shards.resize(nlocal);
    for (int i = 0; i < nlocal; ++i) {
        Shard &s = ctx->shards[i];
        s.rank = (ctx->cfg.comm_mode == CGX_COMM_RCCL || ctx->cfg.comm_mode == CGX_COMM_P2P) ? ctx->cfg.rank : i;
        s.row0 = ctx->start_rows[s.rank];
        s.rows = ctx->num_rows[s.rank];
        s.plan = plan_for(s.rows);
        const size_t rows_alloc = (size_t)std::max(s.rows, 1);
        s.npartials = 3 * cgx::update_xr_grid(n) + 8;
        if (!ctx->banded) HIP_TRY(ctx, hipMalloc(&s.A, rows_alloc * (size_t)ctx->lda * sizeof(double)));
        HIP_TRY(ctx, hipMalloc(&s.b_full, (size_t)n * sizeof(double)));
        HIP_TRY(ctx, hipMalloc(&s.x, rows_alloc * sizeof(double)));
        HIP_TRY(ctx, hipMalloc(&s.p[0], (size_t)ctx->lda * sizeof(double)));
        HIP_TRY(ctx, hipMalloc(&s.p[1], (size_t)ctx->lda * sizeof(double)));
        const size_t apg_bytes = (size_t)ctx->nranks * ctx->seg_S * sizeof(double);
        const int rr_parts = cgx::update_xr_grid(n);   // one r.r partial per K3 workgroup
        const size_t rbuf_bytes = (size_t)(ctx->lda + rr_parts) * sizeof(double);
        HIP_TRY(ctx, hipMalloc(&s.apg, apg_bytes));
        HIP_TRY(ctx, hipMalloc(&s.rbuf, rbuf_bytes));
        s.apv = cgx::SegView{s.apg, ctx->seg_S, ctx->seg_Sr, n / ctx->nranks, ctx->nranks, n, s.rank, 0, 0, 0};
        cgx::seg_finalize(&s.apv);
        s.rv = cgx::SegView{s.rbuf, (int)ctx->lda + rr_parts, (int)ctx->lda, n, 1, n, 0, 0, 0, 0};
        cgx::seg_finalize(&s.rv);
        HIP_TRY(ctx, hipMalloc(&s.partials, (size_t)s.npartials * sizeof(double)));
        HIP_TRY(ctx, hipMalloc(&s.sc, sizeof(Scalars)));
        HIP_TRY(ctx, hipMalloc(&s.gathered, (size_t)cgx::kMaxRanks * cgx::kSlots * sizeof(double)));
        HIP_TRY(ctx, hipMemsetAsync(s.p[0], 0, (size_t)ctx->lda * sizeof(double), ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(s.p[1], 0, (size_t)ctx->lda * sizeof(double), ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(s.apg, 0, apg_bytes, ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(s.rbuf, 0, rbuf_bytes, ctx->stream));
        if (s.rows <= 0 && s.A) HIP_TRY(ctx, hipMemsetAsync(s.A, 0, (size_t)ctx->lda * sizeof(double), ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(s.partials, 0, (size_t)s.npartials * sizeof(double), ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(s.sc, 0, sizeof(Scalars), ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(s.gathered, 0, (size_t)cgx::kMaxRanks * cgx::kSlots * sizeof(double), ctx->stream));
    }
    if (ctx->cfg.comm_mode == CGX_COMM_LOOPBACK) {
        std::vector<double *> gp(nlocal);
        std::vector<Scalars *> sp(nlocal);
        for (int i = 0; i < nlocal; ++i) {
            gp[i] = ctx->shards[i].gathered;
            sp[i] = ctx->shards[i].sc;
        }
        HIP_TRY(ctx, hipMalloc(&ctx->d_gathered_ptrs, nlocal * sizeof(double *)));
        HIP_TRY(ctx, hipMalloc(&ctx->d_scalar_ptrs, nlocal * sizeof(Scalars *)));
        HIP_TRY(ctx, hipMemcpy(ctx->d_gathered_ptrs, gp.data(), nlocal * sizeof(double *), hipMemcpyHostToDevice));
        HIP_TRY(ctx, hipMemcpy(ctx->d_scalar_ptrs, sp.data(), nlocal * sizeof(Scalars *), hipMemcpyHostToDevice));
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CGX_OK;
}

// ---- collectives ---------------------------------------------------------------------------------

// CGX_COMM_P2P: one lean all-gather kernel over the IPC-mapped mailboxes (cgx_kernels.hip).
cgx_status p2p_allgather(cgx_ctx *ctx, int chan, const double *src, int count, double *dst, long dst_stride,
                         int copy_self, int tail_off = 0, int tail_n = 0, int sum_off = 0)
{
    if (!ctx->p2p_ready) return fail(ctx, CGX_ERR_P2P, "cgx_p2p_import has not been called");
    if ((long)(count + (tail_n > 0 ? 1 : 0)) * 8 > ctx->mv.slot_bytes[chan])
        return fail(ctx, CGX_ERR_P2P, "p2p payload larger than its slot");
    const unsigned long long epoch = ++ctx->p2p_epoch[chan];
    HIP_TRY(ctx, cgx::launch_mailbox_allgather(ctx->mv, chan, epoch, src, count, tail_off, tail_n, dst, dst_stride, sum_off,
                                               copy_self, ctx->p2p_timeout_ticks, ctx->d_p2p_err, ctx->stream));
    return CGX_OK;
}

// Scalars: every shard contributes sc->local[kSlots]; afterwards every shard's gathered[] holds all of them.
// Replaces MPI_Allreduce (cg.cc:92,106,117).  In SELF mode the consumers read sc->local directly.
cgx_status gather_scalars(cgx_ctx *ctx)
{
    switch (ctx->cfg.comm_mode) {
    case CGX_COMM_SELF:
        return CGX_OK;
    case CGX_COMM_LOOPBACK:
        HIP_TRY(ctx, cgx::launch_loopback_gather(ctx->d_gathered_ptrs, ctx->d_scalar_ptrs, ctx->nranks, ctx->stream));
        return CGX_OK;
    case CGX_COMM_P2P: {
        Shard &s = ctx->shards[0];
        return p2p_allgather(ctx, 2, s.sc->local, cgx::kSlots, s.gathered, cgx::kSlots, 1);
    }
    default: {
        Shard &s = ctx->shards[0];
        NCCL_TRY(ctx, ctx->rccl->AllGather(s.sc->local, s.gathered, cgx::kSlots, ncclDouble, ctx->comm, ctx->stream));
        return CGX_OK;
    }
    }
}

// THE exchange of an iteration: every shard's own segment [Ap slice | p.Ap partials] inside its apg is current;
// afterwards all P segments are.  Replaces MPI_Allreduce (cg.cc:106) and MPI_Allgatherv (cg.cc:135-136); segments
// have the same size on every rank, so N % P != 0 needs no special case.  with_tail = false moves the slices only
// (used for the final gather of x, MPI_Gatherv cg.cc:140-142).
cgx_status gather_segments(cgx_ctx *ctx, bool with_tail)
{
    const size_t S = (size_t)ctx->seg_S;
    switch (ctx->cfg.comm_mode) {
    case CGX_COMM_SELF:
        return CGX_OK;
    case CGX_COMM_LOOPBACK:
        for (auto &dst : ctx->shards)
            for (auto &src : ctx->shards)
                if (dst.rank != src.rank)
                    HIP_TRY(ctx, hipMemcpyAsync(dst.apg + src.rank * S, src.apg + src.rank * S, S * sizeof(double),
                                                hipMemcpyDeviceToDevice, ctx->stream));
        return CGX_OK;
    case CGX_COMM_P2P: {
        // the exchange kernel folds this rank's partials and ships [Ap slice | one double]
        Shard &s = ctx->shards[0];
        return p2p_allgather(ctx, 1, s.Ap(), ctx->seg_Sr, s.apg, ctx->seg_S, 0, ctx->seg_Sr, with_tail ? ctx->npart : 0,
                             ctx->seg_Sr + ctx->npart);
    }
    default: {
        Shard &s = ctx->shards[0];
        NCCL_TRY(ctx, ctx->rccl->AllGather(s.Ap(), s.apg, S, ncclDouble, ctx->comm, ctx->stream));
        return CGX_OK;
    }
    }
}

// ---- K1 with optional event bracketing -----------------------------------------------------------
cgx_status take_event(cgx_ctx *ctx, hipEvent_t *out)
{
    if (ctx->ev_used == ctx->ev_pool.size()) {
        hipEvent_t e;
        HIP_TRY(ctx, hipEventCreate(&e));
        ctx->ev_pool.push_back(e);
    }
    *out = ctx->ev_pool[ctx->ev_used++];
    return CGX_OK;
}

// CGX_MATRIX_BANDED: (re)allocate the diagonals of shard s for the given ascending offsets; contents zeroed.
cgx_status alloc_dia(cgx_ctx *ctx, Shard &s, const std::vector<int> &offs)
{
    if ((int)offs.size() > CGX_MAX_DIAGONALS)
        return fail(ctx, CGX_ERR_UNSUPPORTED, "row block of rank " + std::to_string(s.rank) + " has " +
                                                  std::to_string(offs.size()) + " non-zero diagonals, more than " +
                                                  std::to_string(CGX_MAX_DIAGONALS) +
                                                  ": not a banded matrix (use CGX_MATRIX_DENSE)");
    (void)hipFree(s.dia_vals);
    s.dia_vals = nullptr;
    s.dia = cgx::DiaView{};
    s.dia.ld = ((long)std::max(s.rows, 1) + 1) / 2 * 2;
    s.dia.ndiag = (int)offs.size();
    for (int t = 0; t < s.dia.ndiag; ++t) s.dia.off[t] = offs[t];
    const size_t bytes = (size_t)std::max(s.dia.ndiag, 1) * (size_t)s.dia.ld * sizeof(double);
    HIP_TRY(ctx, hipMalloc(&s.dia_vals, bytes));
    HIP_TRY(ctx, hipMemsetAsync(s.dia_vals, 0, bytes, ctx->stream));
    s.dia.vals = s.dia_vals;
    return CGX_OK;
}

// K1, plain form (vector given): initial residual, DEBUG verification, probes.
cgx_status run_gemv_plain(cgx_ctx *ctx, Shard &s, const double *v_full)
{
    if (ctx->banded)
        HIP_TRY(ctx, cgx::launch_spmv_dia_plain(s.plan, s.dia, s.rows, s.row0, ctx->n, v_full, s.Ap(), s.k1_part(), s.sc,
                                                ctx->stream));
    else
        HIP_TRY(ctx, cgx::launch_gemv_plain(s.plan, s.A, ctx->lda, s.rows, v_full, v_full + s.row0, s.Ap(), s.k1_part(),
                                            s.sc, ctx->stream));
    return CGX_OK;
}

// K1, fused form of iteration k; every `profile_gemv`-th launch is bracketed with HIP events.
cgx_status run_gemv_fused(cgx_ctx *ctx, Shard &s, int k)
{
    const int every = ctx->cfg.profile_gemv;
    // at most 2048 timed launches per cgx_solve_steps call: the event pool stays bounded however long the run is
    const bool timed = every > 0 && (ctx->gemv_seq++ % every) == 0 && ctx->ev_used + 2 <= 4096;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (timed) {
        CGX_TRY(take_event(ctx, &e0));
        CGX_TRY(take_event(ctx, &e1));
        HIP_TRY(ctx, hipEventRecord(e0, ctx->stream));
    }
    if (ctx->banded)
        HIP_TRY(ctx, cgx::launch_spmv_dia_fused(s.plan, s.dia, s.rows, s.row0, ctx->n, ctx->lda, s.p[k & 1], s.p[(k + 1) & 1],
                                                s.rv, s.Ap(), s.k1_part(), s.sc, k, ctx->tol, ctx->stream));
    else
        HIP_TRY(ctx, cgx::launch_gemv_fused(s.plan, s.A, ctx->lda, s.rows, s.row0, s.p[k & 1], s.p[(k + 1) & 1], s.rv,
                                            s.Ap(), s.k1_part(), s.sc, k, ctx->tol, ctx->stream));
    if (timed) HIP_TRY(ctx, hipEventRecord(e1, ctx->stream));
    return CGX_OK;
}

// Fold the recorded event pairs into the running K1 statistics (call after a stream sync).
cgx_status harvest_gemv_events(cgx_ctx *ctx)
{
    for (size_t i = 0; i + 1 < ctx->ev_used; i += 2) {
        float ms = 0.f;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev_pool[i], ctx->ev_pool[i + 1]));
        ctx->gemv_ms_sum += ms;
        if (ctx->gemv_launches == 0 || ms < ctx->gemv_ms_min) ctx->gemv_ms_min = ms;
        ctx->gemv_launches++;
    }
    ctx->ev_used = 0;
    return CGX_OK;
}

// ---- one body of the loop cg.cc:96-137: two kernels, one exchange ------------------------------------
cgx_status enqueue_iteration(cgx_ctx *ctx, int k)
{
    hipStream_t st = ctx->stream;
    // tail of iteration k-1 (cg.cc:117-132) + GEMV and p.Ap partials of iteration k (cg.cc:100-105)
    for (auto &s : ctx->shards) CGX_TRY(run_gemv_fused(ctx, s, k));
    if (ctx->cfg.comm_mode == CGX_COMM_P2P && !ctx->cfg.p2p_separate_exchange) {
        // direct peer exchange folded into K3: the iteration is two kernels, no collective launch at all
        Shard &s = ctx->shards[0];
        if (!ctx->p2p_ready) return fail(ctx, CGX_ERR_P2P, "cgx_p2p_import has not been called");
        const unsigned long long epoch = ++ctx->p2p_epoch[1];
        HIP_TRY(ctx, cgx::launch_update_xr_p2p(ctx->n, s.rows, s.row0, s.p[(k + 1) & 1], s.apv, ctx->npart, ctx->mv, 1, epoch,
                                               s.x, s.rv, s.sc, k & 1, ctx->p2p_timeout_ticks, ctx->d_p2p_err, st));
        return CGX_OK;
    }
    CGX_TRY(gather_segments(ctx, true));                                                             // cg.cc:106
    const bool folded = ctx->cfg.comm_mode == CGX_COMM_P2P;   // the exchange kernel already folded each rank's partials
    for (auto &s : ctx->shards)
        HIP_TRY(ctx, cgx::launch_update_xr(ctx->n, s.rows, s.row0, s.p[(k + 1) & 1], s.apv, folded ? ctx->npart : 0,
                                           folded ? 1 : ctx->npart, s.x, s.rv, s.sc, k & 1, s.partials, st));   // cg.cc:105-116
    return CGX_OK;
}

// After a stream sync: did any bounded wait of the direct peer exchange expire?
cgx_status check_p2p_error(cgx_ctx *ctx)
{
    if (ctx->cfg.comm_mode != CGX_COMM_P2P || !ctx->d_p2p_err) return CGX_OK;
    int e = 0;
    HIP_TRY(ctx, hipMemcpy(&e, ctx->d_p2p_err, sizeof(int), hipMemcpyDeviceToHost));
    if (e) return fail(ctx, CGX_ERR_P2P, "direct peer exchange: a wait for a peer's flag expired (peer dead or IPC not coherent)");
    return CGX_OK;
}

cgx_status read_flags_sync(cgx_ctx *ctx)
{
    Shard &s = ctx->shards[0];
    int flags[2] = {0, 0};
    HIP_TRY(ctx, hipMemcpyAsync(flags, &s.sc->done, 2 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->done = flags[0] != 0;
    ctx->k_final = flags[1];
    return check_p2p_error(ctx);
}

}  // namespace

// =====================================================================================================
// C ABI
// =====================================================================================================
extern "C" {

void cgx_config_init(cgx_config *cfg)
{
    if (!cfg) return;
    memset(cfg, 0, sizeof *cfg);
    cfg->struct_version = CGX_VERSION;
    cfg->comm_mode = CGX_COMM_SELF;
    cfg->device = 0;
    cfg->rank = 0;
    cfg->nranks = 1;
    cfg->lda_pad = -1;
}

const char *cgx_status_string(cgx_status s)
{
    switch (s) {
    case CGX_OK: return "ok";
    case CGX_ERR_BAD_ARG: return "bad argument";
    case CGX_ERR_IO: return "i/o error";
    case CGX_ERR_HIP: return "HIP error";
    case CGX_ERR_RCCL: return "RCCL error";
    case CGX_ERR_OOM: return "out of memory";
    case CGX_ERR_NO_DEVICE: return "no usable GPU (libcgx has no CPU fallback)";
    case CGX_ERR_UNSUPPORTED: return "unsupported input";
    case CGX_ERR_P2P: return "direct peer exchange failed";
    }
    return "unknown";
}

const char *cgx_last_error(const cgx_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

cgx_status cgx_partition(int n, int psize, int *start_rows, int *num_rows)
{
    if (n < 0 || psize <= 0 || !start_rows || !num_rows) return CGX_ERR_BAD_ARG;
    partition_rows(n, psize, start_rows, num_rows);
    return CGX_OK;
}

cgx_status cgx_comm_unique_id(unsigned char out[CGX_UNIQUE_ID_BYTES])
{
    static_assert(sizeof(ncclUniqueId) == CGX_UNIQUE_ID_BYTES, "ncclUniqueId size");
    if (!out) return CGX_ERR_BAD_ARG;
    std::string err;
    const cgx::RcclApi *api = cgx::rccl_api(&err);
    if (!api) return fail(nullptr, CGX_ERR_RCCL, err);
    ncclUniqueId id;
    ncclResult_t r = api->GetUniqueId(&id);
    if (r != ncclSuccess) return fail(nullptr, CGX_ERR_RCCL, std::string("ncclGetUniqueId: ") + api->GetErrorString(r));
    memcpy(out, &id, CGX_UNIQUE_ID_BYTES);
    return CGX_OK;
}

cgx_status cgx_create(cgx_ctx **out, const cgx_config *cfg_in)
{
    if (!out) return fail(nullptr, CGX_ERR_BAD_ARG, "cgx_create: out is null");
    *out = nullptr;
    cgx_config cfg;
    if (cfg_in) cfg = *cfg_in;
    else cgx_config_init(&cfg);
    if (cfg.struct_version != CGX_VERSION) return fail(nullptr, CGX_ERR_BAD_ARG, "cgx_config.struct_version mismatch");
    if (cfg.nranks <= 0) cfg.nranks = 1;
    if (cfg.comm_mode == CGX_COMM_SELF && cfg.nranks != 1)
        return fail(nullptr, CGX_ERR_BAD_ARG, "CGX_COMM_SELF requires nranks == 1");
    if (cfg.comm_mode == CGX_COMM_LOOPBACK && cfg.nranks > 16)
        return fail(nullptr, CGX_ERR_BAD_ARG, "CGX_COMM_LOOPBACK supports at most 16 logical shards");
    if ((cfg.comm_mode == CGX_COMM_RCCL || cfg.comm_mode == CGX_COMM_P2P) &&
        (cfg.rank < 0 || cfg.rank >= cfg.nranks || cfg.nranks > cgx::kMaxRanks))
        return fail(nullptr, CGX_ERR_BAD_ARG, "CGX_COMM_RCCL/P2P: rank out of range or nranks > 64");
    if (cfg.comm_mode < CGX_COMM_SELF || cfg.comm_mode > CGX_COMM_P2P)
        return fail(nullptr, CGX_ERR_BAD_ARG, "unknown comm_mode");
    if (cfg.matrix_format != CGX_MATRIX_DENSE && cfg.matrix_format != CGX_MATRIX_BANDED)
        return fail(nullptr, CGX_ERR_BAD_ARG, "unknown matrix_format");

    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, CGX_ERR_NO_DEVICE,
                    std::string("no HIP device visible (") + hipGetErrorString(e) + "); libcgx has no CPU fallback");
    if (cfg.device < 0 || cfg.device >= ndev) return fail(nullptr, CGX_ERR_NO_DEVICE, "device ordinal out of range");

    cgx_ctx *ctx = new (std::nothrow) cgx_ctx();
    if (!ctx) return fail(nullptr, CGX_ERR_OOM, "host allocation failed");
    ctx->cfg = cfg;
    ctx->device = cfg.device;
    ctx->nranks = cfg.nranks;
    ctx->banded = cfg.matrix_format == CGX_MATRIX_BANDED;
    if (ctx->cfg.check_every <= 0) ctx->cfg.check_every = 16;
    if (getenv("CGX_P2P_SEPARATE_EXCHANGE")) ctx->cfg.p2p_separate_exchange = 1;

    auto bail = [&](cgx_status st) {
        g_create_error = ctx->err;
        cgx_destroy(ctx);
        return st;
    };
    if (hipSetDevice(ctx->device) != hipSuccess) {
        ctx->err = "hipSetDevice failed";
        return bail(CGX_ERR_NO_DEVICE);
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, ctx->device) != hipSuccess) {
        ctx->err = "hipGetDeviceProperties failed";
        return bail(CGX_ERR_HIP);
    }
    if (!strstr(prop.gcnArchName, "gfx950") && !getenv("CGX_ALLOW_ANY_ARCH")) {
        ctx->err = std::string("device is ") + prop.gcnArchName + ", libcgx is built for gfx950 (MI355X) only";
        return bail(CGX_ERR_NO_DEVICE);
    }
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        ctx->err = "hipStreamCreate failed";
        return bail(CGX_ERR_HIP);
    }
    if (hipHostMalloc(reinterpret_cast<void **>(&ctx->h_flags), 4 * sizeof(int), hipHostMallocDefault) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->flag_ev[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->flag_ev[1], hipEventDisableTiming) != hipSuccess) {
        ctx->err = "pinned flag / event allocation failed";
        return bail(CGX_ERR_HIP);
    }
    if (cfg.comm_mode == CGX_COMM_RCCL) {
        std::string err;
        ctx->rccl = cgx::rccl_api(&err);
        if (!ctx->rccl) {
            ctx->err = err;
            return bail(CGX_ERR_RCCL);
        }
        ncclUniqueId id;
        memcpy(&id, cfg.unique_id, CGX_UNIQUE_ID_BYTES);
        ncclResult_t r = ctx->rccl->CommInitRank(&ctx->comm, cfg.nranks, id, cfg.rank);
        if (r != ncclSuccess) {
            ctx->err = std::string("ncclCommInitRank: ") + ctx->rccl->GetErrorString(r);
            ctx->comm = nullptr;
            return bail(CGX_ERR_RCCL);
        }
    }
    if (cfg.comm_mode == CGX_COMM_P2P) {
        ctx->mailbox_bytes = (size_t)(cfg.p2p_mailbox_kib > 0 ? cfg.p2p_mailbox_kib : 4096) * 1024;
        ctx->p2p_timeout_ticks = (long long)(cfg.p2p_timeout_ms > 0 ? cfg.p2p_timeout_ms : 5000) * 100000LL;   // 100 MHz
        // fine-grained: stores from peers and system-scope atomics are coherent without a kernel boundary
        if (hipExtMallocWithFlags(reinterpret_cast<void **>(&ctx->mailbox), ctx->mailbox_bytes, hipDeviceMallocFinegrained) != hipSuccess ||
            hipMemset(ctx->mailbox, 0, ctx->mailbox_bytes) != hipSuccess ||
            hipMalloc(reinterpret_cast<void **>(&ctx->d_p2p_err), sizeof(int)) != hipSuccess ||
            hipMemset(ctx->d_p2p_err, 0, sizeof(int)) != hipSuccess) {
            ctx->err = "mailbox allocation failed";
            return bail(CGX_ERR_P2P);
        }
        ctx->mv.nranks = cfg.nranks;
        ctx->mv.rank = cfg.rank;
        ctx->mv.base[cfg.rank] = ctx->mailbox;
        if (cfg.nranks == 1) ctx->p2p_ready = true;
    }
    *out = ctx;
    return CGX_OK;
}

cgx_status cgx_p2p_export(cgx_ctx *ctx, unsigned char out[CGX_IPC_HANDLE_BYTES])
{
    static_assert(sizeof(hipIpcMemHandle_t) == CGX_IPC_HANDLE_BYTES, "hipIpcMemHandle_t size");
    if (!ctx || !out || ctx->cfg.comm_mode != CGX_COMM_P2P) return fail(ctx, CGX_ERR_BAD_ARG, "cgx_p2p_export: not a P2P context");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipIpcMemHandle_t h;
    HIP_TRY(ctx, hipIpcGetMemHandle(&h, ctx->mailbox));
    memcpy(out, &h, CGX_IPC_HANDLE_BYTES);
    return CGX_OK;
}

cgx_status cgx_p2p_import(cgx_ctx *ctx, const unsigned char *handles)
{
    if (!ctx || !handles || ctx->cfg.comm_mode != CGX_COMM_P2P) return fail(ctx, CGX_ERR_BAD_ARG, "cgx_p2p_import: not a P2P context");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    for (int q = 0; q < ctx->nranks; ++q) {
        if (q == ctx->cfg.rank) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, handles + (size_t)q * CGX_IPC_HANDLE_BYTES, CGX_IPC_HANDLE_BYTES);
        void *ptr = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess)
            return fail(ctx, CGX_ERR_P2P, std::string("hipIpcOpenMemHandle(rank ") + std::to_string(q) + "): " + hipGetErrorString(e));
        ctx->mv.base[q] = static_cast<unsigned char *>(ptr);
    }
    ctx->p2p_ready = true;
    return CGX_OK;
}

cgx_status cgx_p2p_selftest(cgx_ctx *ctx, int rounds, int *ok)
{
    if (!ctx || !ok || rounds <= 0 || ctx->cfg.comm_mode != CGX_COMM_P2P) return fail(ctx, CGX_ERR_BAD_ARG, "cgx_p2p_selftest: bad argument");
    *ok = 0;
    if (!ctx->p2p_ready) return fail(ctx, CGX_ERR_P2P, "cgx_p2p_import has not been called");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int P = ctx->nranks, me = ctx->cfg.rank;
    const int count = 1024;   // doubles per rank: 8 KiB, the size class of the real exchanges
    cgx::MailboxView saved = ctx->mv;
    ctx->mv.data_off[1] = p2p_fixed_prefix(P);
    ctx->mv.slot_bytes[1] = (long)count * 8;
    if ((size_t)(ctx->mv.data_off[1] + 2L * P * count * 8) > ctx->mailbox_bytes) {
        ctx->mv = saved;
        return fail(ctx, CGX_ERR_P2P, "mailbox too small for the self-test");
    }
    double *dsrc = nullptr, *ddst = nullptr;
    HIP_TRY(ctx, hipMalloc(&dsrc, (size_t)count * sizeof(double)));
    HIP_TRY(ctx, hipMalloc(&ddst, (size_t)P * count * sizeof(double)));
    std::vector<double> hsrc(count), hdst((size_t)P * count);
    bool good = true;
    cgx_status st = CGX_OK;
    for (int r = 0; r < rounds && good; ++r) {
        for (int i = 0; i < count; ++i) hsrc[i] = 1e6 * (me + 1) + 1e3 * r + i + 0.25;
        HIP_TRY(ctx, hipMemcpyAsync(dsrc, hsrc.data(), count * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(ddst, 0, (size_t)P * count * sizeof(double), ctx->stream));
        st = p2p_allgather(ctx, 1, dsrc, count, ddst, count, 1);
        if (st != CGX_OK) break;
        HIP_TRY(ctx, hipMemcpyAsync(hdst.data(), ddst, (size_t)P * count * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (check_p2p_error(ctx) != CGX_OK) { good = false; break; }
        for (int q = 0; q < P && good; ++q)
            for (int i = 0; i < count; ++i)
                if (hdst[(size_t)q * count + i] != 1e6 * (q + 1) + 1e3 * r + i + 0.25) { good = false; break; }
    }
    (void)hipFree(dsrc);
    (void)hipFree(ddst);
    ctx->mv = saved;
    if (st != CGX_OK) return st;
    *ok = good ? 1 : 0;
    return CGX_OK;
}

cgx_status cgx_destroy(cgx_ctx *ctx)
{
    if (!ctx) return CGX_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    free_problem(ctx);
    if (ctx->comm && ctx->rccl) (void)ctx->rccl->CommDestroy(ctx->comm);
    if (ctx->cfg.comm_mode == CGX_COMM_P2P) {
        for (int q = 0; q < ctx->nranks; ++q)
            if (q != ctx->cfg.rank && ctx->mv.base[q]) (void)hipIpcCloseMemHandle(ctx->mv.base[q]);
        if (ctx->mailbox) (void)hipFree(ctx->mailbox);
        if (ctx->d_p2p_err) (void)hipFree(ctx->d_p2p_err);
    }
    for (auto e : ctx->ev_pool) (void)hipEventDestroy(e);
    for (auto e : ctx->flag_ev)
        if (e) (void)hipEventDestroy(e);
    if (ctx->h_flags) (void)hipHostFree(ctx->h_flags);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return CGX_OK;
}

cgx_status cgx_get_size(const cgx_ctx *ctx, int *m, int *n)
{
    if (!ctx) return CGX_ERR_BAD_ARG;
    if (m) *m = ctx->m;
    if (n) *n = ctx->n;
    return CGX_OK;
}

cgx_status cgx_get_matrix_format(const cgx_ctx *ctx, int local_shard, int *format, int *ndiag, int *offsets,
                                 double *matrix_bytes)
{
    if (!ctx || local_shard < 0 || local_shard >= (int)ctx->shards.size() || !ctx->have_matrix) return CGX_ERR_BAD_ARG;
    const Shard &s = ctx->shards[local_shard];
    if (format) *format = ctx->banded ? CGX_MATRIX_BANDED : CGX_MATRIX_DENSE;
    if (ndiag) *ndiag = ctx->banded ? s.dia.ndiag : 0;
    if (offsets && ctx->banded)
        for (int t = 0; t < s.dia.ndiag; ++t) offsets[t] = s.dia.off[t];
    if (matrix_bytes)
        *matrix_bytes = ctx->banded ? 8.0 * (double)s.dia.ndiag * (double)s.dia.ld : 8.0 * (double)std::max(s.rows, 1) * (double)ctx->lda;
    return CGX_OK;
}

cgx_status cgx_set_max_iter(cgx_ctx *ctx, int max_iter)
{
    if (!ctx) return CGX_ERR_BAD_ARG;
    ctx->max_iter = max_iter;   // the reference does not validate either (cg.cc:204-216)
    return CGX_OK;
}

cgx_status cgx_set_tolerance(cgx_ctx *ctx, double tol)
{
    if (!ctx) return CGX_ERR_BAD_ARG;
    ctx->tol = tol;
    return CGX_OK;
}

// ---- generate_lap2d_matrix, cg.cc:159-188 ----------------------------------------------------------
cgx_status cgx_generate_lap2d_matrix(cgx_ctx *ctx, int size)
{
    if (!ctx) return CGX_ERR_BAD_ARG;
    CGX_TRY(setup_problem(ctx, size));
    if (ctx->banded) {
        // the five diagonals of cg.cc:181-185, written straight into banded storage: no n x n block ever exists
        int off[5];
        const int nd = cgx::lap2d_offsets(size, off);
        for (auto &s : ctx->shards) {
            CGX_TRY(alloc_dia(ctx, s, std::vector<int>(off, off + nd)));
            HIP_TRY(ctx, cgx::launch_dia_generate_lap2d(s.dia_vals, s.dia, size, s.row0, s.rows, ctx->stream));
        }
    } else {
        for (auto &s : ctx->shards)
            HIP_TRY(ctx, cgx::launch_generate_lap2d(s.A, ctx->lda, size, s.row0, s.rows, ctx->stream));
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->have_matrix = true;
    return CGX_OK;
}

// ---- read_matrix with a caller-supplied dense matrix (cg.cu:307-321 after Matrix::read) ---------
cgx_status cgx_set_matrix_dense(cgx_ctx *ctx, const double *A, long lda_host, int n)
{
    if (!ctx || !A || lda_host < n) return fail(ctx, CGX_ERR_BAD_ARG, "cgx_set_matrix_dense: bad argument");
    CGX_TRY(setup_problem(ctx, n));
    for (auto &s : ctx->shards) {
        if (s.rows <= 0) {
            if (ctx->banded) CGX_TRY(alloc_dia(ctx, s, {}));
            continue;
        }
        double *dst = s.A;
        if (ctx->banded)   // staged densely for the scan only, freed below
            HIP_TRY(ctx, hipMalloc(&dst, (size_t)s.rows * ctx->lda * sizeof(double)));
        struct Staging {
            double *p;
            ~Staging() { (void)hipFree(p); }
        } staging{ctx->banded ? dst : nullptr};
        HIP_TRY(ctx, hipMemsetAsync(dst, 0, (size_t)s.rows * ctx->lda * sizeof(double), ctx->stream));
        HIP_TRY(ctx, hipMemcpy2DAsync(dst, (size_t)ctx->lda * sizeof(double), A + (size_t)s.row0 * lda_host,
                                      (size_t)lda_host * sizeof(double), (size_t)n * sizeof(double), (size_t)s.rows,
                                      hipMemcpyHostToDevice, ctx->stream));
        if (!ctx->banded) continue;
        // which diagonals hold a non-zero (device scan), then pack them
        const size_t nflags = 2 * (size_t)n - 1;
        unsigned char *dflags = nullptr;
        HIP_TRY(ctx, hipMalloc(&dflags, nflags));
        struct Flags {
            unsigned char *p;
            ~Flags() { (void)hipFree(p); }
        } flags_guard{dflags};
        HIP_TRY(ctx, hipMemsetAsync(dflags, 0, nflags, ctx->stream));
        HIP_TRY(ctx, cgx::launch_dia_mark(dst, ctx->lda, n, s.row0, s.rows, dflags, ctx->stream));
        std::vector<unsigned char> hflags(nflags);
        HIP_TRY(ctx, hipMemcpyAsync(hflags.data(), dflags, nflags, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        std::vector<int> offs;
        for (size_t f = 0; f < nflags; ++f)
            if (hflags[f]) offs.push_back((int)((long)f - (n - 1)));
        CGX_TRY(alloc_dia(ctx, s, offs));
        HIP_TRY(ctx, cgx::launch_dia_pack(dst, ctx->lda, n, s.row0, s.rows, s.dia_vals, s.dia, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->have_matrix = true;
    return CGX_OK;
}

// ---- MatrixCOO::read + Matrix::read, matrix_coo.cc:7-60 and matrix.cc:6-22 -------------------------
cgx_status cgx_read_matrix(cgx_ctx *ctx, const char *path)
{
    if (!ctx || !path) return fail(ctx, CGX_ERR_BAD_ARG, "cgx_read_matrix: bad argument");
    FILE *f = fopen(path, "r");
    if (!f) return fail(ctx, CGX_ERR_IO, std::string("Could not open matrix: ") + path);   // matrix_coo.cc:14-17
    struct Closer {
        FILE *f;
        ~Closer() { fclose(f); }
    } closer{f};

    char line[2048];
    if (!fgets(line, sizeof line, f)) return fail(ctx, CGX_ERR_IO, "Could not process Matrix Market banner.");
    char tok[5][64] = {{0}};
    if (sscanf(line, "%63s %63s %63s %63s %63s", tok[0], tok[1], tok[2], tok[3], tok[4]) != 5 ||
        strcmp(tok[0], "%%MatrixMarket") != 0)
        return fail(ctx, CGX_ERR_IO, "Could not process Matrix Market banner.");   // matrix_coo.cc:19-22
    for (int t = 1; t < 5; ++t)
        for (char *c = tok[t]; *c; ++c) *c = (char)tolower((unsigned char)*c);       // mmio.c lower-cases the tokens
    if (strcmp(tok[1], "matrix") != 0 || strcmp(tok[2], "coordinate") != 0)
        return fail(ctx, CGX_ERR_UNSUPPORTED, std::string("Sorry, this application does not support Market Market type: [") +
                                                  tok[1] + " " + tok[2] + " " + tok[3] + " " + tok[4] + "]");   // matrix_coo.cc:25-29
    // The reference parses every entry as "%d %d %lg" whatever the field (matrix_coo.cc:48); fields without
    // one real value per entry would be silently misread there and are rejected here.
    if (strcmp(tok[3], "real") != 0 && strcmp(tok[3], "integer") != 0 && strcmp(tok[3], "double") != 0)
        return fail(ctx, CGX_ERR_UNSUPPORTED, std::string("Matrix Market field not supported: ") + tok[3]);
    const bool is_sym = strcmp(tok[4], "symmetric") == 0;                             // matrix_coo.cc:43
    if (!is_sym && strcmp(tok[4], "general") != 0)
        return fail(ctx, CGX_ERR_UNSUPPORTED, std::string("Matrix Market symmetry not supported: ") + tok[4]);

    int m = 0, n = 0, nz = 0;
    for (;;) {   // size line after the % comments, mmio.c:198-206
        if (!fgets(line, sizeof line, f)) return fail(ctx, CGX_ERR_IO, "Matrix Market size line missing");
        if (line[0] == '%') continue;
        if (sscanf(line, "%d %d %d", &m, &n, &nz) == 3) break;
    }
    if (m <= 0 || n <= 0 || nz < 0 || m != n)
        return fail(ctx, CGX_ERR_UNSUPPORTED, "CG needs a square matrix with positive size");
    CGX_TRY(setup_problem(ctx, n));

    // The entries are parsed from large reads of the file (strtol/strtod on a buffer: the "%d %d %lg" of
    // matrix_coo.cc:48 without a libc call per field) and kept in file order; everything after that -- which row
    // block an entry belongs to, the mirrored assignment of a symmetric file, and "a later entry for the same (i,j)
    // overrides an earlier one" (the sequential loop of matrix.cc:12-21) -- is done on the device.
    std::vector<int> hI, hJ;
    std::vector<double> ha;
    hI.reserve((size_t)nz);
    hJ.reserve((size_t)nz);
    ha.reserve((size_t)nz);
    std::vector<int> offs;   // banded: distinct (column - row) of all assignments, at most CGX_MAX_DIAGONALS + 1 kept
    auto note_offset = [&](int off) {
        if (!ctx->banded || (int)offs.size() > CGX_MAX_DIAGONALS) return;
        auto it = std::lower_bound(offs.begin(), offs.end(), off);
        if (it == offs.end() || *it != off) offs.insert(it, off);
    };
    {
        const size_t kChunk = (size_t)32 << 20;
        std::vector<char> buf(kChunk + 4096);
        size_t have = 0;            // bytes of an unfinished token carried over from the previous read
        int field = 0, I = 0, J = 0;
        bool eof = false;
        while ((long)ha.size() < (long)nz && !(eof && have == 0)) {
            if (have + kChunk + 1 > buf.size()) buf.resize(have + kChunk + 1);
            const size_t got = eof ? 0 : fread(buf.data() + have, 1, kChunk, f);
            if (got < kChunk) eof = true;
            size_t len = have + got, cut = len;
            if (!eof) {             // stop at the last white space so that no token is split
                while (cut > 0 && !isspace((unsigned char)buf[cut - 1])) --cut;
                if (cut == 0) return fail(ctx, CGX_ERR_IO, "Matrix Market entry " + std::to_string(ha.size()) + " unreadable");
            }
            const char saved = buf[cut];
            buf[cut] = '\0';
            char *p = buf.data();
            while ((long)ha.size() < (long)nz) {
                while (*p && isspace((unsigned char)*p)) ++p;
                if (!*p) break;
                char *e = p;
                if (field < 2) {
                    const long v = strtol(p, &e, 10);
                    if (e == p) return fail(ctx, CGX_ERR_IO, "Matrix Market entry " + std::to_string(ha.size()) + " unreadable");
                    (field == 0 ? I : J) = (int)v;
                    ++field;
                } else {
                    const double a = strtod(p, &e);
                    if (e == p) return fail(ctx, CGX_ERR_IO, "Matrix Market entry " + std::to_string(ha.size()) + " unreadable");
                    field = 0;
                    I--; J--;                                                             // matrix_coo.cc:49-50
                    if (I < 0 || I >= m || J < 0 || J >= n) return fail(ctx, CGX_ERR_IO, "Matrix Market index out of range");
                    hI.push_back(I);
                    hJ.push_back(J);
                    ha.push_back(a);
                    note_offset(J - I);                                                   // matrix.cc:17
                    if (is_sym) note_offset(I - J);                                       // matrix.cc:18-20
                }
                p = e;
            }
            buf[cut] = saved;
            have = len - cut;
            memmove(buf.data(), buf.data() + cut, have);
            if (eof && (long)ha.size() < (long)nz && have == 0) break;
        }
        if ((long)ha.size() < (long)nz)
            return fail(ctx, CGX_ERR_IO, "Matrix Market entry " + std::to_string(ha.size()) + " unreadable");
    }
    if (ctx->banded && (int)offs.size() > CGX_MAX_DIAGONALS)
        return fail(ctx, CGX_ERR_UNSUPPORTED, "matrix has more than " + std::to_string(CGX_MAX_DIAGONALS) +
                                                  " non-zero diagonals: not a banded matrix (use CGX_MATRIX_DENSE)");

    const size_t cnt = ha.size();
    struct DevBuf {
        void *p = nullptr;
        ~DevBuf() { (void)hipFree(p); }
    } dI, dJ, da, dwin;
    if (cnt) {
        HIP_TRY(ctx, hipMalloc(&dI.p, cnt * sizeof(int)));
        HIP_TRY(ctx, hipMalloc(&dJ.p, cnt * sizeof(int)));
        HIP_TRY(ctx, hipMalloc(&da.p, cnt * sizeof(double)));
        HIP_TRY(ctx, hipMalloc(&dwin.p, 2 * cnt));
        HIP_TRY(ctx, hipMemcpyAsync(dI.p, hI.data(), cnt * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(dJ.p, hJ.data(), cnt * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(da.p, ha.data(), cnt * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    }
    for (auto &s : ctx->shards) {
        if (ctx->banded) CGX_TRY(alloc_dia(ctx, s, offs));   // zero-filled; every shard keeps the matrix's diagonals
        if (s.rows <= 0) continue;
        if (!ctx->banded)
            HIP_TRY(ctx, hipMemsetAsync(s.A, 0, (size_t)s.rows * ctx->lda * sizeof(double), ctx->stream));  // Matrix::resize zero-fills
        if (!cnt) continue;
        HIP_TRY(ctx, hipMemsetAsync(dwin.p, 0, 2 * cnt, ctx->stream));
        HIP_TRY(ctx, cgx::launch_coo_assign(s.A, ctx->lda, ctx->banded ? &s.dia : nullptr, s.dia_vals, n, s.row0, s.rows,
                                            static_cast<const int *>(dI.p), static_cast<const int *>(dJ.p),
                                            static_cast<const double *>(da.p), (long)cnt, is_sym ? 1 : 0,
                                            static_cast<unsigned char *>(dwin.p), ctx->stream));
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->have_matrix = true;
    return CGX_OK;
}

// ---- init_source_term, cg.cc:218-234 ----------------------------------------------------------------
cgx_status cgx_set_source_term(cgx_ctx *ctx, const double *b)
{
    if (!ctx || !b) return fail(ctx, CGX_ERR_BAD_ARG, "cgx_set_source_term: bad argument");
    if (ctx->n <= 0 || ctx->shards.empty()) return fail(ctx, CGX_ERR_BAD_ARG, "set the matrix before the source term");
    ctx->b_host.assign(b, b + ctx->n);
    for (auto &s : ctx->shards)
        HIP_TRY(ctx, hipMemcpyAsync(s.b_full, ctx->b_host.data(), (size_t)ctx->n * sizeof(double), hipMemcpyHostToDevice,
                                    ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->have_b = true;
    return CGX_OK;
}

cgx_status cgx_init_source_term(cgx_ctx *ctx, double h)
{
    if (!ctx) return CGX_ERR_BAD_ARG;
    if (ctx->n <= 0) return fail(ctx, CGX_ERR_BAD_ARG, "set the matrix before the source term");
    // Evaluated on the host with libm, in the reference's expression order, so b is bit-identical (cg.cc:230-231).
    std::vector<double> b((size_t)ctx->n);
    for (int i = 0; i < ctx->n; i++)
        b[i] = -2. * i * M_PI * M_PI * std::sin(10. * M_PI * i * h) * std::sin(10. * M_PI * i * h);
    return cgx_set_source_term(ctx, b.data());
}

// ---- solve, cg.cc:38-156 -----------------------------------------------------------------------------
cgx_status cgx_solve_begin(cgx_ctx *ctx, const double *x0)
{
    if (!ctx || !x0) return fail(ctx, CGX_ERR_BAD_ARG, "cgx_solve_begin: bad argument");
    if (!ctx->have_matrix || !ctx->have_b) return fail(ctx, CGX_ERR_BAD_ARG, "matrix and source term must be set before solve");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->t_begin = wall_now();
    ctx->t_loop = 0;
    ctx->k = 0;
    ctx->done = false;
    ctx->k_final = 0;
    ctx->ev_used = 0;
    ctx->gemv_ms_sum = ctx->gemv_ms_min = 0;
    ctx->gemv_launches = 0;
    ctx->gemv_seq = 0;
    hipStream_t st = ctx->stream;
    const int n = ctx->n;
    const size_t vec_bytes = (size_t)ctx->lda * sizeof(double);
    for (auto &s : ctx->shards) {
        HIP_TRY(ctx, hipMemsetAsync(s.sc, 0, sizeof(Scalars), st));
        HIP_TRY(ctx, hipMemsetAsync(s.apg, 0, (size_t)ctx->nranks * ctx->seg_S * sizeof(double), st));
        HIP_TRY(ctx, hipMemsetAsync(s.rbuf, 0, (size_t)s.rv.S * sizeof(double), st));
        // x (initial guess) replicated for the first GEMV, x_sub = x[rows]  (cg.cc:72, 80)
        HIP_TRY(ctx, hipMemcpyAsync(s.p[0], x0, (size_t)n * sizeof(double), hipMemcpyHostToDevice, st));
        if (s.rows > 0)
            HIP_TRY(ctx, hipMemcpyAsync(s.x, s.p[0] + s.row0, (size_t)s.rows * sizeof(double), hipMemcpyDeviceToDevice, st));
    }
    for (auto &s : ctx->shards) CGX_TRY(run_gemv_plain(ctx, s, s.p[0]));                             // cg.cc:79-81
    CGX_TRY(gather_segments(ctx, false));
    for (auto &s : ctx->shards)
        HIP_TRY(ctx, cgx::launch_init_residual(n, s.b_full, s.apv, s.rv, s.partials, st));           // cg.cc:82
    for (auto &s : ctx->shards) {
        // p_old of iteration 0 is 0, so K1(0) forms p = r + 0*0 = r  (p_sub = r_sub, cg.cc:85)
        HIP_TRY(ctx, hipMemsetAsync(s.p[0], 0, vec_bytes, st));
        HIP_TRY(ctx, hipMemsetAsync(s.p[1], 0, vec_bytes, st));
    }
    ctx->in_solve = true;
    return CGX_OK;
}

cgx_status cgx_solve_steps(cgx_ctx *ctx, int nsteps, int *done_out)
{
    if (!ctx || !ctx->in_solve) return fail(ctx, CGX_ERR_BAD_ARG, "cgx_solve_steps outside begin/end");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const double t0 = wall_now();
    // K1 statistics describe the most recent steps call (bench.py: the timed region, not the warmup)
    ctx->ev_used = 0;
    ctx->gemv_ms_sum = ctx->gemv_ms_min = 0;
    ctx->gemv_launches = 0;
    ctx->gemv_seq = 0;   // the first K1 of every steps call is always one of the sampled launches
    const int every = ctx->cfg.check_every;
    int slot = 0;
    bool pending[2] = {false, false};
    bool stop = ctx->done;
    int left = std::min(nsteps, ctx->max_iter - ctx->k);
    while (left > 0 && !stop) {
        const int batch = std::min(left, every);
        for (int i = 0; i < batch; ++i) CGX_TRY(enqueue_iteration(ctx, ctx->k + i));
        ctx->k += batch;
        left -= batch;
        // publish {done,k_final} after this batch; look at the batch BEFORE it, so one batch stays queued
        HIP_TRY(ctx, hipMemcpyAsync(ctx->h_flags + 2 * slot, &ctx->shards[0].sc->done, 2 * sizeof(int),
                                    hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipEventRecord(ctx->flag_ev[slot], ctx->stream));
        pending[slot] = true;
        slot ^= 1;
        if (pending[slot]) {
            HIP_TRY(ctx, hipEventSynchronize(ctx->flag_ev[slot]));
            pending[slot] = false;
            if (ctx->h_flags[2 * slot]) stop = true;   // identical on every rank: rsnew is bit-identical (cg.cc:117-121)
        }
    }
    CGX_TRY(read_flags_sync(ctx));
    if (ctx->cfg.profile_gemv) CGX_TRY(harvest_gemv_events(ctx));
    ctx->t_loop += wall_now() - t0;
    if (done_out) *done_out = ctx->done ? 1 : 0;
    return CGX_OK;
}

cgx_status cgx_solve_end(cgx_ctx *ctx, double *x, cgx_result *res)
{
    if (!ctx || !ctx->in_solve) return fail(ctx, CGX_ERR_BAD_ARG, "cgx_solve_end outside begin");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    // The convergence test of the last enqueued iteration is normally done by the NEXT K1; when the loop
    // ran out there is none, so close it here (cg.cc:117-121,132).
    for (auto &s : ctx->shards) HIP_TRY(ctx, cgx::launch_close_iteration(s.sc, s.rv, ctx->k, ctx->tol, st));
    CGX_TRY(read_flags_sync(ctx));
    const int k_exit = ctx->done ? ctx->k_final : ctx->k;

    // Gather x (MPI_Gatherv, cg.cc:140-142) through the exchange segments, then the DEBUG verification
    // (cg.cc:144-151) with the same K1, distributed over the shards instead of rank 0 alone.
    for (auto &s : ctx->shards)
        if (s.rows > 0)
            HIP_TRY(ctx, hipMemcpyAsync(s.Ap(), s.x, (size_t)s.rows * sizeof(double), hipMemcpyDeviceToDevice, st));
    CGX_TRY(gather_segments(ctx, false));
    for (auto &s : ctx->shards) HIP_TRY(ctx, cgx::launch_unpack_segments(s.apv, s.p[0], ctx->lda, st));
    for (auto &s : ctx->shards) CGX_TRY(run_gemv_plain(ctx, s, s.p[0]));
    for (auto &s : ctx->shards)
        HIP_TRY(ctx, cgx::launch_debug_norms(s.rows, s.Ap(), s.b_full + s.row0, s.x, s.partials, st));
    for (auto &s : ctx->shards)
        HIP_TRY(ctx, cgx::launch_reduce_partials3(s.partials, cgx::update_xr_grid(s.rows), s.sc->local, st));
    CGX_TRY(gather_scalars(ctx));

    Shard &s0 = ctx->shards[0];
    Scalars hs;
    std::vector<double> hg((size_t)cgx::kMaxRanks * cgx::kSlots, 0.0);
    HIP_TRY(ctx, hipMemcpyAsync(&hs, s0.sc, sizeof hs, hipMemcpyDeviceToHost, st));
    if (ctx->cfg.comm_mode != CGX_COMM_SELF)
        HIP_TRY(ctx, hipMemcpyAsync(hg.data(), s0.gathered, hg.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    if (x) HIP_TRY(ctx, hipMemcpyAsync(x, s0.p[0], (size_t)ctx->n * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    if (ctx->cfg.comm_mode == CGX_COMM_SELF)
        for (int v = 0; v < cgx::kSlots; ++v) hg[v] = hs.local[v];
    double sums[3] = {0, 0, 0};
    for (int v = 0; v < 3; ++v)
        for (int q = 0; q < ctx->nranks; ++q) sums[v] += hg[(size_t)q * cgx::kSlots + v];

    ctx->in_solve = false;
    if (res) {
        memset(res, 0, sizeof *res);
        res->iterations = k_exit;
        res->converged = ctx->done ? 1 : 0;
        res->residual_prev = std::sqrt(hs.rs[k_exit & 1]);         // sqrt(rsold) as printed, cg.cc:152-153
        res->residual_last = std::sqrt(hs.rs[(k_exit + 1) & 1]);
        if (!ctx->done) res->residual_last = res->residual_prev;    // loop ran out: rsold == rsnew (cg.cc:132)
        res->x_norm = std::sqrt(sums[2]);
        res->rel_residual = std::sqrt(sums[0]) / std::sqrt(sums[1]);
        res->seconds_solve = wall_now() - ctx->t_begin;
        res->seconds_loop = ctx->t_loop;
        res->gemv_launches = ctx->gemv_launches;
        res->gemv_ms_avg = ctx->gemv_launches ? ctx->gemv_ms_sum / (double)ctx->gemv_launches : 0.0;
        res->gemv_ms_min = ctx->gemv_ms_min;
        res->gemv_bytes = ctx->banded ? 8.0 * ((double)s0.rows * s0.dia.ndiag + 2.0 * s0.rows)
                                      : 8.0 * ((double)s0.rows * ctx->n + ctx->n + s0.rows);
    }
    return CGX_OK;
}

cgx_status cgx_solve(cgx_ctx *ctx, double *x, cgx_result *res)
{
    if (!ctx || !x) return fail(ctx, CGX_ERR_BAD_ARG, "cgx_solve: bad argument");
    CGX_TRY(cgx_solve_begin(ctx, x));
    int done = 0;
    CGX_TRY(cgx_solve_steps(ctx, ctx->max_iter, &done));
    return cgx_solve_end(ctx, x, res);
}

// ---- kernel probes -------------------------------------------------------------------------------------
cgx_status cgx_probe_gemv(cgx_ctx *ctx, const double *p, double *y, double *pAp)
{
    if (!ctx || !p || !y) return fail(ctx, CGX_ERR_BAD_ARG, "cgx_probe_gemv: bad argument");
    if (!ctx->have_matrix) return fail(ctx, CGX_ERR_BAD_ARG, "no matrix");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    double total = 0.0;
    for (auto &s : ctx->shards) {
        HIP_TRY(ctx, hipMemsetAsync(s.sc, 0, sizeof(Scalars), st));
        HIP_TRY(ctx, hipMemcpyAsync(s.p[0], p, (size_t)ctx->n * sizeof(double), hipMemcpyHostToDevice, st));
        CGX_TRY(run_gemv_plain(ctx, s, s.p[0]));
        HIP_TRY(ctx, cgx::launch_reduce_partials(s.k1_part(), s.plan.grid, &s.sc->local[cgx::kSlotConj], st));
        double part = 0.0;
        if (s.rows > 0)
            HIP_TRY(ctx, hipMemcpyAsync(y + s.row0, s.Ap(), (size_t)s.rows * sizeof(double), hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipMemcpyAsync(&part, &s.sc->local[cgx::kSlotConj], sizeof(double), hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipStreamSynchronize(st));
        total += part;
    }
    if (pAp) *pAp = total;
    return CGX_OK;
}

cgx_status cgx_probe_time_gemv(cgx_ctx *ctx, int reps, double *ms_per_launch)
{
    if (!ctx || reps <= 0 || !ms_per_launch) return fail(ctx, CGX_ERR_BAD_ARG, "cgx_probe_time_gemv: bad argument");
    if (!ctx->have_matrix) return fail(ctx, CGX_ERR_BAD_ARG, "no matrix");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    hipEvent_t e0, e1;
    HIP_TRY(ctx, hipEventCreate(&e0));
    HIP_TRY(ctx, hipEventCreate(&e1));
    for (auto &s : ctx->shards) HIP_TRY(ctx, hipMemsetAsync(s.sc, 0, sizeof(Scalars), st));
    for (auto &s : ctx->shards) CGX_TRY(run_gemv_plain(ctx, s, s.p[0]));   // warm
    HIP_TRY(ctx, hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i)
        for (auto &s : ctx->shards) CGX_TRY(run_gemv_plain(ctx, s, s.p[0]));
    HIP_TRY(ctx, hipEventRecord(e1, st));
    HIP_TRY(ctx, hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *ms_per_launch = (double)ms / reps / (double)ctx->shards.size();
    return CGX_OK;
}

cgx_status cgx_probe_vector_ops(cgx_ctx *ctx, int n, double alpha, double beta, double *x, double *r, double *p,
                                const double *Ap, double *rr)
{
    if (!ctx || n <= 0 || !x || !r || !p || !Ap) return fail(ctx, CGX_ERR_BAD_ARG, "cgx_probe_vector_ops: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    // A single-shard problem of length n around the PRODUCTION kernels: K3 for x/r/r.r, then the fused K1 of
    // the next iteration (on a 1 x n zero matrix) for p = r + beta p.
    const long lda = ((long)n + 15) / 16 * 16;
    const int Sr = std::max((n + 1) / 2 * 2, 2), S = Sr + 2;      // Ap segment: [Ap (Sr) | one partial, pad]
    const size_t bytes = (size_t)n * sizeof(double), vbytes = (size_t)lda * sizeof(double);
    const int grid = cgx::update_xr_grid(n);
    double *dx = nullptr, *dp0 = nullptr, *dp1 = nullptr, *dap = nullptr, *drb = nullptr, *dpart = nullptr, *dA = nullptr,
           *dAp1 = nullptr;
    Scalars *dsc = nullptr;
    HIP_TRY(ctx, hipMalloc(&dx, bytes));
    HIP_TRY(ctx, hipMalloc(&dp0, vbytes));
    HIP_TRY(ctx, hipMalloc(&dp1, vbytes));
    HIP_TRY(ctx, hipMalloc(&dap, (size_t)S * sizeof(double)));
    HIP_TRY(ctx, hipMalloc(&drb, (size_t)(lda + grid) * sizeof(double)));
    HIP_TRY(ctx, hipMalloc(&dpart, (size_t)(grid + 8) * sizeof(double)));
    HIP_TRY(ctx, hipMalloc(&dA, vbytes));
    HIP_TRY(ctx, hipMalloc(&dAp1, 64 * sizeof(double)));
    HIP_TRY(ctx, hipMalloc(&dsc, sizeof(Scalars)));
    cgx::SegView apv{dap, S, Sr, n, 1, n, 0, 0, 0, 0};
    cgx::seg_finalize(&apv);
    cgx::SegView rv{drb, (int)lda + grid, (int)lda, n, 1, n, 0, 0, 0, 0};
    cgx::seg_finalize(&rv);
    // Force the wanted alpha: with rsold = alpha and p.Ap = 1, K3 computes alpha / max(1, alpha*1e-14) = alpha.
    Scalars hs{};
    hs.rs[0] = alpha;
    const double one = 1.0;
    HIP_TRY(ctx, hipMemsetAsync(dp0, 0, vbytes, st));
    HIP_TRY(ctx, hipMemsetAsync(dp1, 0, vbytes, st));
    HIP_TRY(ctx, hipMemsetAsync(dap, 0, (size_t)S * sizeof(double), st));
    HIP_TRY(ctx, hipMemsetAsync(drb, 0, (size_t)(lda + grid) * sizeof(double), st));
    HIP_TRY(ctx, hipMemsetAsync(dA, 0, vbytes, st));
    HIP_TRY(ctx, hipMemcpyAsync(dsc, &hs, sizeof hs, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(dx, x, bytes, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(drb, r, bytes, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(dp0, p, bytes, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(dap, Ap, bytes, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(dap + Sr, &one, sizeof(double), hipMemcpyHostToDevice, st));   // the one p.Ap "partial"
    HIP_TRY(ctx, cgx::launch_update_xr(n, n, 0, dp0, apv, 0, 1, dx, rv, dsc, 0, dpart, st));
    std::vector<double> rr_parts_h((size_t)grid);
    HIP_TRY(ctx, hipMemcpyAsync(rr_parts_h.data(), drb + lda, (size_t)grid * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(x, dx, bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(r, drb, bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    double rr_host = 0.0;
    for (double v : rr_parts_h) rr_host += v;
    // Force the wanted beta: rsold = 1, the r.r partials = {beta, 0, ...}  =>  K1(k=1) computes beta/1.
    HIP_TRY(ctx, hipMemcpyAsync(&dsc->rs[0], &one, sizeof(double), hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemsetAsync(drb + lda, 0, (size_t)grid * sizeof(double), st));
    HIP_TRY(ctx, hipMemcpyAsync(drb + lda, &beta, sizeof(double), hipMemcpyHostToDevice, st));
    cgx::GemvPlan plan = cgx::plan_gemv(ctx->cfg.gemv_variant, 1, (int)lda);
    HIP_TRY(ctx, cgx::launch_gemv_fused(plan, dA, lda, 1, 0, dp0, dp1, rv, dAp1, dAp1 + 8, dsc, 1, -1.0 /* never converges */, st));
    HIP_TRY(ctx, hipMemcpyAsync(p, dp1, bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    if (rr) *rr = rr_host;
    (void)hipFree(dx); (void)hipFree(dp0); (void)hipFree(dp1); (void)hipFree(dap); (void)hipFree(drb);
    (void)hipFree(dpart); (void)hipFree(dA); (void)hipFree(dAp1); (void)hipFree(dsc);
    return CGX_OK;
}

cgx_status cgx_probe_get_matrix_rows(cgx_ctx *ctx, int local_shard, double *A_out, int *row0, int *rows)
{
    if (!ctx || local_shard < 0 || local_shard >= (int)ctx->shards.size())
        return fail(ctx, CGX_ERR_BAD_ARG, "cgx_probe_get_matrix_rows: bad shard");
    if (!ctx->have_matrix) return fail(ctx, CGX_ERR_BAD_ARG, "no matrix");
    Shard &s = ctx->shards[local_shard];
    if (row0) *row0 = s.row0;
    if (rows) *rows = s.rows;
    if (A_out && s.rows > 0 && ctx->banded) {
        // expand the diagonals on the host (a test probe, small sizes)
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        std::vector<double> vals((size_t)std::max(s.dia.ndiag, 1) * (size_t)s.dia.ld);
        HIP_TRY(ctx, hipMemcpy(vals.data(), s.dia_vals, vals.size() * sizeof(double), hipMemcpyDeviceToHost));
        std::fill(A_out, A_out + (size_t)s.rows * ctx->n, 0.0);
        for (int t = 0; t < s.dia.ndiag; ++t)
            for (int i = 0; i < s.rows; ++i) {
                const long j = (long)s.row0 + i + s.dia.off[t];
                if (j >= 0 && j < ctx->n) A_out[(size_t)i * ctx->n + (size_t)j] = vals[(size_t)t * s.dia.ld + i];
            }
    } else if (A_out && s.rows > 0) {
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        HIP_TRY(ctx, hipMemcpy2D(A_out, (size_t)ctx->n * sizeof(double), s.A, (size_t)ctx->lda * sizeof(double),
                                 (size_t)ctx->n * sizeof(double), (size_t)s.rows, hipMemcpyDeviceToHost));
    }
    return CGX_OK;
}

}  // extern "C"
