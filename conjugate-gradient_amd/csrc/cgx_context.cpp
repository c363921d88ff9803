// cgx_context.cpp -- life cycle of a libcgx context: configuration, device, stream, RCCL / mailbox wire-up, and
// the allocation of the row-block shards (include/cgx.h "life cycle", "CGX_COMM_P2P wire-up", cgx_partition).
#include "cgx_internal.h"

#include <sys/mman.h>
#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <cctype>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace cgxi;

namespace cgxi {

// Largest n the DEFAULT choice hands to the streaming persistent kernel: measured against the per-launch path it wins up to
// N = 9216 by 8 % and more (17.9 / 31.8 / 50.6 / 70.5 / 95.3 us per iteration at N = 5120 / 6144 / 7168 / 8192 / 9216 against 36.1 / 47.7 /
// 65.2 / 79.7 / 103.9, the slower of two boxes).  From N = 9500 to 11264 the two are within 3 % of each other, and which one is ahead
// changes from box to box (N = 10000: 116.7 against 118.9 on one, 120.5 against 118.8 on the other -- the per-launch figure is the
// stable one: a persistent kernel ends with its slowest workgroup).  What decides up to N = 10000 (BASELINE config 2: N = 10000 run to
// convergence, timed as the reference times it: all of solve() in a fresh process) is that window: `cgsolver 10000 out` 71.1-74.4 ms
// through the streaming kernel, whose solve is four launches, against 77.5-78.6 through K1 + K3, on three boxes; 9500: 65.9 against
// 66.5; 10240: 78.6 against 77.0 -- so the end is 10000.  Above S = 11 the streaming kernel has no row of A on the chip and loses
// 
constexpr int kStreamDefaultMax = 10000;

thread_local std::string g_create_error;

double wall_now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

cgx_status fail(cgx_ctx *ctx, cgx_status st, const std::string &msg)
{
    if (ctx) ctx->err = msg;
    else g_create_error = msg;
    return st;
}

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

void bind_state(Shard &s, long lda)
{
    double *b = s.state[s.cur];
    s.x = b;
    s.rbuf = b + cgx::state_off_r(lda);
    s.p[1] = b + cgx::state_off_p(lda);
    s.sc = reinterpret_cast<Scalars *>(b + cgx::state_off_sc(lda));
    s.rv.base = s.rbuf;
}

void free_shard(Shard &s)
{
    if (s.state[0]) {   // x, rbuf, p[1] and sc live inside the state blocks
        (void)hipFree(s.state[0]);
        (void)hipFree(s.state[1]);
        s.x = s.p[1] = s.rbuf = nullptr;
        s.sc = nullptr;
    }
    (void)hipFree(s.A);
    (void)hipFree(s.dia_vals);
    (void)hipFree(s.b_full);
    (void)hipFree(s.x);
    (void)hipFree(s.p[0]);
    (void)hipFree(s.p[1]);
    (void)hipFree(s.apg);
    (void)hipFree(s.rbuf);
    (void)hipFree(s.partials);
    (void)hipFree(s.ap_parts);
    (void)hipFree(s.k1_scratch);
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
    if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
    ctx->h_stage = nullptr;
    ctx->d_gathered_ptrs = nullptr;
    ctx->d_scalar_ptrs = nullptr;
    ctx->have_matrix = ctx->have_b = false;
    ctx->in_solve = false;
}

// Mailbox bytes before the segment channel: flag words, chunk flag words, channel 0 (16-B slots) and channel 2 (kSlots doubles).
static long p2p_flag_bytes() { return (long)cgx::kP2pChannels * cgx::kMaxRanks * cgx::kP2pFlagStride; }
static long p2p_data0_off() { return p2p_flag_bytes() + (long)cgx::kMaxChunkFlags * 8; }
long p2p_fixed_prefix(int nranks)
{
    return p2p_data0_off() + 2L * nranks * 16 + 2L * nranks * (long)cgx::kSlots * 8;
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

// Tagged words: zero this rank's own channel-1 slots of the layout just established (enqueued on the context's stream).  What
// a tagged reader may find in a position is then a zero or a tagged word of an earlier epoch of THIS layout, never something
// a different geometry (or the self-test) left there.  Safe without a launcher barrier: nothing of the previous layout is in
// flight towards this mailbox any more (the last exchange of a solve and of the self-test is on channel 2: a rank finishes it
// only after every peer has pushed its channel-2 data, which a peer does, in stream order, after its last channel-1 kernel),
// and a peer that is ahead can push tagged words of the NEW layout only after it has completed a plain all-gather of that
// layout (solve_begin's, or the first half of the self-test) -- which needs this rank's contribution, enqueued behind this
// memset on the same stream.  A peer's plain pushes of the new layout land in channel 0 / 2, which are not touched here.
static cgx_status scrub_tagged_region(cgx_ctx *ctx)
{
    const size_t bytes = (size_t)(2L * ctx->nranks * ctx->mv.slot_bytes[1]);
    HIP_TRY(ctx, hipMemsetAsync(ctx->mailbox + ctx->mv.data_off[1], 0, bytes, ctx->stream));
    return CGX_OK;
}

// One resident grid at a time per device, across contexts and processes: the workgroups of a persistent kernel wait for each
// other, and two such grids dispatched at the same moment can each be given part of the CUs (neither fits a second workgroup beside
// its own on a CU) and then wait for workgroups that can never be placed, until the bounded waits expire.  An advisory lock on a
// file named after the device's PCI bus id serialises them (held from launch to the synchronisation in resident_steps; released
// by the kernel when a process dies).  Best effort: where the file cannot be had the launches go unserialised, and a launch
// whose waits expire is redone on the per-launch path anyway.  The file lives in /tmp because every user of the box shares the
// device; it is only ever locked, never read or written, and it is opened defensively: never through a symbolic link
// (O_NOFOLLOW), only a regular file with a single name (a hard link to somebody's file is refused), and the mode is widened to
// 0666 only on a file this very call has created (O_EXCL).
static int open_device_lock(int device)
{
    char bus[64] = "unknown";
    (void)hipDeviceGetPCIBusId(bus, sizeof bus, device);
    std::string name = std::string("/tmp/cgx_resident_") + bus + ".lock";
    for (char &c : name)
        if (c == ':') c = '_';
    int fd = open(name.c_str(), O_CREAT | O_EXCL | O_RDWR | O_CLOEXEC | O_NOFOLLOW, 0666);
    if (fd >= 0) {
        (void)fchmod(fd, 0666);   // created here: other users of the box share the device too (the umask may have narrowed it)
        return fd;
    }
    fd = open(name.c_str(), O_RDWR | O_CLOEXEC | O_NOFOLLOW);
    if (fd < 0) fd = open(name.c_str(), O_RDONLY | O_CLOEXEC | O_NOFOLLOW);   // somebody else's file, mode narrowed: flock needs no write access
    if (fd < 0) return -1;
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_nlink != 1) {
        close(fd);
        return -1;
    }
    return fd;
}

// A persistent kernel takes a problem when: one GPU (CGX_COMM_SELF), dense storage, n <= 16384 (up to 4096 the matrix stays on
// the chip, cgx_resident.hip: all rows in LDS up to 2048, above 16 rows per workgroup in LDS + registers + a streamed rest;
// from 4097 every row is streamed, cgx_stream.hip), the default K1 choice (gemv_variant 0; 40000 asks for it and fails if it
// cannot be had; any explicit per-launch shape, -1 or CGX_RESIDENT=0 keep the per-launch path; CGX_STREAM_MAX=n moves the upper
// end of the default, 0 = never above 4096), and all of its workgroups are resident at once (they wait for each other).
static cgx_status setup_resident(cgx_ctx *ctx, int variant)
{
    ctx->resident = false;
    const bool forced = variant == 40000 || variant == 50000;
    ctx->res_forced = forced;
    if (!forced && variant != 0) return CGX_OK;
    const char *env = getenv("CGX_RESIDENT");
    if (!forced && env && atoi(env) == 0) return CGX_OK;
    auto no = [&](const char *why) {
        return forced ? fail(ctx, CGX_ERR_UNSUPPORTED, std::string("gemv_variant 40000 / 50000 (persistent-kernel solver): ") + why) : CGX_OK;
    };
    if (ctx->cfg.comm_mode != CGX_COMM_SELF || ctx->banded) return no("one GPU (CGX_COMM_SELF) and dense storage only");
    if (!forced && ctx->n > 4096) {
        const char *smax = getenv("CGX_STREAM_MAX");
        if (ctx->n > (smax ? atoi(smax) : kStreamDefaultMax)) return CGX_OK;
    }
    cgx::ResidentPlan pl{};
    const bool fits = variant == 50000 ? cgx::plan_stream(ctx->n, ctx->cus, ctx->lds_per_cu, &pl)   // the streaming kernel, also below 4097
                                       : cgx::plan_resident(ctx->n, ctx->cus, ctx->lds_per_cu, &pl);
    if (!fits) return no("the problem does not fit (n <= 16384 on 256 CUs; 50000: n >= 1024)");
    int per_cu = 0;
    if (cgx::prepare_cg_resident(pl, &per_cu) != hipSuccess) {
        (void)hipGetLastError();
        return no("the runtime refused the kernel's LDS request");
    }
    int limit = per_cu * ctx->cus;
    if (ctx->resident_limit > 0) limit = ctx->resident_limit;
    if (pl.grid > limit) return no("its workgroups would not all be resident at once");
    if (!ctx->res_xbuf) {
        const size_t bytes = (size_t)2 * 16384 * 2 * sizeof(unsigned long long);   // the largest plan: 512 KiB
        // ordinary device memory: the tagged words travel with agent-scope (sc1) stores and loads; fine-grained memory and
        // system scope, as between GPUs, measured the same (profiles/r04_resident/)
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->res_xbuf), bytes));
        ctx->res_xbuf_bytes = bytes;
    }
    if (!ctx->d_res_err) HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->d_res_err), sizeof(int)));
    if (!ctx->h_res_tail) {
        HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void **>(&ctx->h_res_tail), sizeof(cgx::ResidentTail), hipHostMallocDefault));
        memset(ctx->h_res_tail, 0, sizeof(cgx::ResidentTail));
    }
    // every problem starts with the error word down (an earlier problem's expired wait must not poison this one: ADVICE r4) and
    // with an exchange buffer of zeros only (no tag is 0): what a reader finds in a position is then a zero or a tagged word of
    // an earlier epoch of THIS geometry, never something another problem size left there
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_res_err, 0, sizeof(int), ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->res_xbuf, 0, ctx->res_xbuf_bytes, ctx->stream));
    if (ctx->res_lock_fd < 0 && !getenv("CGX_RESIDENT_NOLOCK"))   // (the variable: diagnostics, to show what the lock is for)
        ctx->res_lock_fd = open_device_lock(ctx->device);
    ctx->rplan = pl;
    ctx->resident = true;
    return CGX_OK;
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
        {   // the persistent-kernel decision is taken afresh (a problem that fell back to the per-launch path gets its chance again)
            int variant = ctx->cfg.gemv_variant;
            if (variant <= 0) {
                const char *e = getenv("CGX_GEMV_VARIANT");
                if (e) variant = atoi(e);
            }
            CGX_TRY(setup_resident(ctx, variant));
        }
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        return CGX_OK;
    }
    free_problem(ctx);
    ctx->m = ctx->n = n;
    if (n <= (8 << 20) &&
        hipHostMalloc(reinterpret_cast<void **>(&ctx->h_stage), (size_t)(n + 16) * sizeof(double), hipHostMallocDefault) != hipSuccess)
        ctx->h_stage = nullptr;   // not fatal: the copies fall back to the caller's pageable buffer
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
    // Chunked exchange (cgx_kernels.hip "Chunks"): whoever consumes K1's Ap adds its column pieces and reduces one p.Ap
    // partial per chunk of the slice -- k_prefold_ap in front of the exchange, or the pushers of the fused P2P update.  That is
    // every multi-rank dense run and every fused P2P run; one GPU and banded storage keep K1's own partials in the tail.
    const bool fused_p2p = ctx->cfg.comm_mode == CGX_COMM_P2P && !ctx->cfg.p2p_separate_exchange;
    ctx->chunked = (ctx->nranks > 1 && !ctx->banded) || fused_p2p;
    const bool allow_split = ctx->chunked && !ctx->banded && ctx->nranks > 1;
    // (40000 = the LDS-resident solver, setup_resident below: set-up, verification and the probes still run the default K1)
    const int k1_variant = (variant == 40000 || variant == 50000) ? 0 : variant;
    auto plan_for = [&](int rows) {
        return ctx->banded ? cgx::plan_dia(rows, k1_variant) : cgx::plan_gemv(k1_variant, rows, ctx->n, ctx->lda, allow_split);
    };
    int grid_max = 1;
    for (int q = 0; q < ctx->nranks; ++q) {
        const cgx::GemvPlan pl = plan_for(ctx->num_rows[q]);
        grid_max = std::max(grid_max, pl.grid);
        if (pl.split > 1) ctx->chunked = true;   // an explicit column-split shape on one GPU: somebody has to add the pieces
    }
    const int cpr = cgx::chunks_per_rank(ctx->seg_Sr);
    ctx->npart = ctx->chunked ? cpr : grid_max;
    if (fused_p2p) {
        // The update kernel with the exchange inside works one row per thread and its workgroups wait for each other inside
        // the kernel: all of them must be resident at once, and their flags must fit the mailbox's flag words.
        int limit = 0;
        HIP_TRY(ctx, cgx::update_xr_p2p_resident_limit(ctx->device, ctx->mv.tagged != 0, &limit));
        if (ctx->resident_limit > 0) limit = ctx->resident_limit;
        const long grid = ((long)n + 255) / 256;
        if (grid > cgx::kMaxVectorGrid || grid > limit || (long)ctx->nranks * cpr > cgx::kMaxChunkFlags)
            return fail(ctx, CGX_ERR_UNSUPPORTED,
                        "CGX_COMM_P2P with the exchange folded into the update kernel: " + std::to_string(grid) + " workgroups (one row "
                        "per thread) must be resident at once; the device keeps " + std::to_string(limit) + " of this kernel resident "
                        "(occupancy x CUs), the kernel takes at most " + std::to_string(cgx::kMaxVectorGrid) +
                        "; use p2p_separate_exchange or CGX_COMM_RCCL");
    } else if (ctx->cfg.comm_mode == CGX_COMM_P2P && n > 256 * cgx::kMaxVectorGrid) {
        return fail(ctx, CGX_ERR_UNSUPPORTED, "CGX_COMM_P2P handles at most 262144 rows; use CGX_COMM_RCCL");
    }
    if (ctx->cfg.comm_mode == CGX_COMM_P2P) {
        // mailbox layout of this problem: flags, chunk flags, then per channel [2 parities][nranks] slots
        // The small fixed-size channels come first, so that their place never depends on the problem; the
        // segment channel (1), whose slot size does, comes last.  A re-layout for a new problem size is then
        // safe without a launcher barrier: the last exchange of a solve is on channel 2, and a rank can finish
        // it only after every peer has pushed its channel-2 data, i.e. after every peer is done with channel 1.
        // (Chunk flag words change their meaning with cpr, but only ever hold epochs of the past: a stale word can
        // never satisfy a wait for a newer epoch.)
        // (tagged words: every double of the fused exchange travels as two 8-byte words)
        const long slot[cgx::kP2pChannels] = {16, ((long)(ctx->seg_Sr + ctx->npart + 1) * (ctx->mv.tagged ? 16 : 8) + 15) / 16 * 16,
                                              (long)cgx::kSlots * 8};
        long off = p2p_fixed_prefix(ctx->nranks);
        ctx->mv.cflag_off = p2p_flag_bytes();
        ctx->mv.data_off[0] = p2p_data0_off();
        ctx->mv.slot_bytes[0] = slot[0];
        ctx->mv.data_off[2] = ctx->mv.data_off[0] + 2L * ctx->nranks * slot[0];
        ctx->mv.slot_bytes[2] = slot[2];
        ctx->mv.data_off[1] = off;
        ctx->mv.slot_bytes[1] = slot[1];
        off += 2L * ctx->nranks * slot[1];
        if (ctx->mv.tagged) {
            // Tagged words: the plain-double all-gathers of the set-up and verification phases (gather_segments: the initial
            // Ap, the final x) get a slot region of their own behind the tagged one -- channel 0, with its own flag words and
            // epoch counter -- so that no plain double is ever stored where a tagged reader polls (cgx_kernels.hip "Tagged words").
            ctx->mv.data_off[0] = off;
            ctx->mv.slot_bytes[0] = ((long)(ctx->seg_Sr + ctx->npart + 1) * 8 + 15) / 16 * 16;
            off += 2L * ctx->nranks * ctx->mv.slot_bytes[0];
        }
        if ((size_t)off > ctx->mailbox_bytes)
            return fail(ctx, CGX_ERR_P2P, "mailbox too small for this problem: need " + std::to_string(off) +
                                              " bytes (raise cgx_config.p2p_mailbox_kib)");
        if (ctx->mv.tagged)   // a freshly laid out tagged region holds zeros only (no tag is 0); see scrub_tagged_region
            CGX_TRY(scrub_tagged_region(ctx));
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
        const size_t apg_bytes = (size_t)ctx->nranks * ctx->seg_S * sizeof(double);
        const int rr_parts = cgx::update_xr_grid(n);   // one r.r partial per K3 workgroup
        const size_t rbuf_bytes = (size_t)(ctx->lda + rr_parts) * sizeof(double);
        const bool blocks = ctx->cfg.comm_mode == CGX_COMM_SELF && !ctx->banded;   // where a persistent kernel may run the loop
        HIP_TRY(ctx, hipMalloc(&s.p[0], (size_t)ctx->lda * sizeof(double)));
        if (blocks) {
            static_assert(sizeof(Scalars) <= 16 * sizeof(double), "Scalars must fit the tail of a state block");
            const size_t bytes = (size_t)cgx::state_doubles(ctx->lda) * sizeof(double);
            for (int q = 0; q < 2; ++q) {
                HIP_TRY(ctx, hipMalloc(&s.state[q], bytes));
                HIP_TRY(ctx, hipMemsetAsync(s.state[q], 0, bytes, ctx->stream));
            }
            s.cur = 0;
            bind_state(s, ctx->lda);
        } else {
            HIP_TRY(ctx, hipMalloc(&s.x, rows_alloc * sizeof(double)));
            HIP_TRY(ctx, hipMalloc(&s.p[1], (size_t)ctx->lda * sizeof(double)));
            HIP_TRY(ctx, hipMalloc(&s.rbuf, rbuf_bytes));
        }
        HIP_TRY(ctx, hipMalloc(&s.apg, apg_bytes));
        s.apv = cgx::SegView{s.apg, ctx->seg_S, ctx->seg_Sr, n / ctx->nranks, ctx->nranks, n, s.rank, 0, 0, 0};
        cgx::seg_finalize(&s.apv);
        s.rv = cgx::SegView{s.rbuf, (int)ctx->lda + rr_parts, (int)ctx->lda, n, 1, n, 0, 0, 0, 0};
        cgx::seg_finalize(&s.rv);
        HIP_TRY(ctx, hipMalloc(&s.partials, (size_t)s.npartials * sizeof(double)));
        if (s.plan.split > 1) {
            HIP_TRY(ctx, hipMalloc(&s.ap_parts, (size_t)s.plan.split * ctx->seg_Sr * sizeof(double)));
            HIP_TRY(ctx, hipMemsetAsync(s.ap_parts, 0, (size_t)s.plan.split * ctx->seg_Sr * sizeof(double), ctx->stream));
        }
        if (ctx->chunked) {
            HIP_TRY(ctx, hipMalloc(&s.k1_scratch, (size_t)(grid_max + 8) * sizeof(double)));
            HIP_TRY(ctx, hipMemsetAsync(s.k1_scratch, 0, (size_t)(grid_max + 8) * sizeof(double), ctx->stream));
        }
        if (!blocks) HIP_TRY(ctx, hipMalloc(&s.sc, sizeof(Scalars)));
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
    CGX_TRY(setup_resident(ctx, variant));
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

}  // namespace cgxi

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
    ctx->cus = prop.multiProcessorCount;
    ctx->lds_per_cu = prop.maxSharedMemoryPerMultiProcessor;
    ctx->res_timeout_ticks = (long long)(cfg.p2p_timeout_ms > 0 ? cfg.p2p_timeout_ms : 5000) * 100000LL;   // 100 MHz
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        ctx->err = "hipStreamCreate failed";
        return bail(CGX_ERR_HIP);
    }
    if (hipHostMalloc(reinterpret_cast<void **>(&ctx->h_flags), 8 * sizeof(int), hipHostMallocDefault) != hipSuccess ||
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
        ctx->mailbox_bytes = (size_t)(cfg.p2p_mailbox_kib > 0 ? cfg.p2p_mailbox_kib : 16384) * 1024;   // 16 MiB: any problem the fused update takes (262 144 rows), also as tagged words
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
        ctx->mv.acquire = cfg.p2p_no_acquire_fence ? 0 : 1;
        ctx->mv.tagged = (cfg.p2p_tagged && !ctx->cfg.p2p_separate_exchange) ? 1 : 0;
        ctx->mv.base[cfg.rank] = ctx->mailbox;
        // the fixed part of the layout (flag words, chunk flag words, the two small channels) never depends on the problem
        ctx->mv.cflag_off = p2p_flag_bytes();
        ctx->mv.data_off[0] = p2p_data0_off();
        ctx->mv.slot_bytes[0] = 16;
        ctx->mv.data_off[2] = ctx->mv.data_off[0] + 2L * cfg.nranks * 16;
        ctx->mv.slot_bytes[2] = (long)cgx::kSlots * 8;
        if (cfg.nranks == 1) ctx->p2p_ready = true;
    }
    *out = ctx;
    return CGX_OK;
}

cgx_status cgx_get_comm_info(cgx_ctx *ctx, int *comm_mode, int *ranks_wired, int *rank_seen, char *device_id)
{
    if (!ctx) return CGX_ERR_BAD_ARG;
    int wired = 1, seen = ctx->cfg.rank;
    switch (ctx->cfg.comm_mode) {
    case CGX_COMM_RCCL:
        if (!ctx->comm || !ctx->rccl) return fail(ctx, CGX_ERR_RCCL, "no communicator");
        NCCL_TRY(ctx, ctx->rccl->CommCount(ctx->comm, &wired));
        NCCL_TRY(ctx, ctx->rccl->CommUserRank(ctx->comm, &seen));
        break;
    case CGX_COMM_P2P:
        wired = 0;
        for (int q = 0; q < ctx->nranks; ++q)
            if (ctx->mv.base[q] && (q == ctx->cfg.rank || ctx->p2p_ready)) ++wired;
        break;
    case CGX_COMM_LOOPBACK:
        wired = ctx->nranks;
        break;
    default:
        break;
    }
    if (comm_mode) *comm_mode = ctx->cfg.comm_mode;
    if (ranks_wired) *ranks_wired = wired;
    if (rank_seen) *rank_seen = seen;
    if (device_id) {
        device_id[0] = 0;
        HIP_TRY(ctx, hipDeviceGetPCIBusId(device_id, 32, ctx->device));
    }
    return CGX_OK;
}

cgx_status cgx_get_gemv_plan(const cgx_ctx *ctx, int local_shard, int out[CGX_GEMV_PLAN_INTS])
{
    if (!ctx || !out || local_shard < 0 || local_shard >= (int)ctx->shards.size()) return CGX_ERR_BAD_ARG;
    const cgx::GemvPlan &pl = ctx->shards[(size_t)local_shard].plan;
    if (ctx->resident && ctx->rplan.stream) {   // variant 5: the loop runs as one persistent kernel that streams every row (U = column steps of 1024)
        const int r[CGX_GEMV_PLAN_INTS] = {5, ctx->rplan.R, ctx->rplan.S, 8, ctx->rplan.RB, ctx->rplan.RL + ctx->rplan.RG, ctx->rplan.grid, pl.ncols};   // light = rows per batch of the ring, split = rows of a workgroup kept on the chip
        memcpy(out, r, sizeof r);
        return CGX_OK;
    }
    if (ctx->resident) {   // variant 4: the loop runs as one persistent kernel on LDS-resident row groups (U = column steps of 512)
        // (light = rows of a workgroup held in registers: 0 up to n = 2048, where all of them are in LDS)
        const int r[CGX_GEMV_PLAN_INTS] = {4, ctx->rplan.R, ctx->rplan.S, 4, ctx->rplan.RG, 1, ctx->rplan.grid, pl.ncols};
        memcpy(out, r, sizeof r);
        return CGX_OK;
    }
    const int v[CGX_GEMV_PLAN_INTS] = {pl.variant, pl.R, pl.U, pl.waves, pl.light, pl.split, pl.grid, pl.ncols};
    memcpy(out, v, sizeof v);
    return CGX_OK;
}

cgx_status cgx_get_resident_record(const cgx_ctx *ctx, long long out[CGX_RESIDENT_RECORD_INTS])
{
    if (!ctx || !out) return CGX_ERR_BAD_ARG;
    for (int i = 0; i < 8; ++i) out[i] = ctx->res_rec[i];
    out[8] = ctx->res_fallbacks;
    out[9] = ctx->resident ? 1 : 0;
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
    // the mailbox is re-laid-out for the test and put back on EVERY return path; so are the two device buffers
    struct RestoreView {
        cgx_ctx *c;
        cgx::MailboxView saved;
        ~RestoreView() { c->mv = saved; }
    } restore{ctx, ctx->mv};
    // geometry of the second half (the fused exchange on an uneven partition, two chunks per rank)
    const int st_n_loc = 1000, st_n = P * st_n_loc + (P > 1 ? 7 : 0);       // the last rank owns 7 rows more (cg.cc:255-266)
    const int st_Sr = (st_n - (P - 1) * st_n_loc + 1) / 2 * 2, st_cpr = cgx::chunks_per_rank(st_Sr);
    // the plain all-gathers of the first half: channel 1 -- or, with tagged words, channel 0 behind the tagged region, exactly
    // as a solve lays the mailbox out (no plain double is ever stored where a tagged reader polls)
    const int plain = ctx->mv.tagged ? 0 : 1;
    ctx->mv.data_off[1] = p2p_fixed_prefix(P);
    ctx->mv.slot_bytes[1] = (long)count * 8;
    if (ctx->mv.tagged) {
        ctx->mv.slot_bytes[1] = ((long)(st_Sr + st_cpr + 1) * 16 + 15) / 16 * 16;
        ctx->mv.data_off[0] = ctx->mv.data_off[1] + 2L * P * ctx->mv.slot_bytes[1];
        ctx->mv.slot_bytes[0] = (long)count * 8;
    }
    if ((size_t)(ctx->mv.data_off[plain] + 2L * P * count * 8) > ctx->mailbox_bytes)
        return fail(ctx, CGX_ERR_P2P, "mailbox too small for the self-test");
    if (ctx->mv.tagged) CGX_TRY(scrub_tagged_region(ctx));
    DeviceScratch scratch;
    double *dsrc = nullptr, *ddst = nullptr;
    HIP_TRY(ctx, scratch.alloc(&dsrc, (size_t)count * sizeof(double)));
    HIP_TRY(ctx, scratch.alloc(&ddst, (size_t)P * count * sizeof(double)));
    std::vector<double> hsrc(count), hdst((size_t)P * count);
    bool good = true;
    for (int r = 0; r < rounds && good; ++r) {
        for (int i = 0; i < count; ++i) hsrc[i] = 1e6 * (me + 1) + 1e3 * r + i + 0.25;
        HIP_TRY(ctx, hipMemcpyAsync(dsrc, hsrc.data(), count * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(ddst, 0, (size_t)P * count * sizeof(double), ctx->stream));
        CGX_TRY(p2p_allgather(ctx, plain, dsrc, count, ddst, count, 1));
        HIP_TRY(ctx, hipMemcpyAsync(hdst.data(), ddst, (size_t)P * count * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (check_p2p_error(ctx) != CGX_OK) { good = false; break; }
        for (int q = 0; q < P && good; ++q)
            for (int i = 0; i < count; ++i)
                if (hdst[(size_t)q * count + i] != 1e6 * (q + 1) + 1e3 * r + i + 0.25) { good = false; break; }
    }
    // Second half: the exchange of the fused update kernel itself (k_update_xr_p2p's own device code: (peer, chunk) pushes,
    // per-chunk flag words, lane-per-flag waits, system-scope reads) on a pattern whose sums are exact in any order: an uneven
    // partition with two chunks per rank and the slice held as two column pieces.
    if (good) {
        // The second half lays the segment channel out differently, so every peer must be done READING its slots of the
        // first half before anybody writes in the new layout: one exchange on channel 2 in between is that barrier (a rank
        // finishes it only after every peer pushed its channel-2 data, which a peer does, in stream order, after its last
        // channel-1 kernel -- the argument that makes a re-layout between two solves safe).
        CGX_TRY(p2p_allgather(ctx, 2, dsrc, cgx::kSlots, ddst, cgx::kSlots, 1));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (check_p2p_error(ctx) != CGX_OK) good = false;
    }
    if (good) {
        const int n_loc = st_n_loc, n = st_n;
        const int rows = (me == P - 1) ? n - me * n_loc : n_loc, row0 = me * n_loc;
        const int Sr = st_Sr, cpr = st_cpr;
        const int grid = cgx::update_xr_grid(n);
        ctx->mv.slot_bytes[1] = ((long)(Sr + cpr + 1) * (ctx->mv.tagged ? 16 : 8) + 15) / 16 * 16;
        if ((size_t)(ctx->mv.data_off[1] + 2L * P * ctx->mv.slot_bytes[1]) > ctx->mailbox_bytes || P * cpr > cgx::kMaxChunkFlags)
            return fail(ctx, CGX_ERR_P2P, "mailbox too small for the self-test");
        cgx::SegView apv{nullptr, Sr + cpr + 1, Sr, n_loc, P, n, me, 0, 0, 0};
        cgx::seg_finalize(&apv);
        double *d_parts = nullptr, *d_ones = nullptr, *d_vals = nullptr, *d_sums = nullptr;
        HIP_TRY(ctx, scratch.alloc(&d_parts, (size_t)2 * Sr * sizeof(double)));
        HIP_TRY(ctx, scratch.alloc(&d_ones, (size_t)(n + 2) * sizeof(double)));
        HIP_TRY(ctx, scratch.alloc(&d_vals, (size_t)n * sizeof(double)));
        HIP_TRY(ctx, scratch.alloc(&d_sums, (size_t)grid * sizeof(double)));
        std::vector<double> parts((size_t)2 * Sr, 0.0), ones((size_t)n + 2, 1.0), vals((size_t)n), sums((size_t)grid);
        HIP_TRY(ctx, hipMemcpyAsync(d_ones, ones.data(), ones.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        auto sent = [&](int q, int row, int r) { return 1e6 * (q + 1) + 1e3 * r + row + 0.5; };   // what rank q sends for its row
        for (int r = 0; r < rounds && good; ++r) {
            for (int row = 0; row < rows; ++row) {
                parts[(size_t)row] = sent(me, row, r) - 0.5;     // piece 0
                parts[(size_t)Sr + row] = 0.5;                    // piece 1
            }
            HIP_TRY(ctx, hipMemcpyAsync(d_parts, parts.data(), parts.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
            HIP_TRY(ctx, hipMemsetAsync(d_vals, 0, (size_t)n * sizeof(double), ctx->stream));
            HIP_TRY(ctx, hipMemsetAsync(d_sums, 0, (size_t)grid * sizeof(double), ctx->stream));
            const unsigned long long epoch = ++ctx->p2p_epoch[1];
            HIP_TRY(ctx, cgx::launch_chunk_exchange_selftest(n, rows, row0, d_ones, apv, cpr, ctx->mv, 1, epoch, ctx->p2p_timeout_ticks,
                                                             ctx->d_p2p_err, d_parts, 2, Sr, d_vals, d_sums, ctx->stream));
            HIP_TRY(ctx, hipMemcpyAsync(vals.data(), d_vals, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipMemcpyAsync(sums.data(), d_sums, (size_t)grid * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            if (check_p2p_error(ctx) != CGX_OK) { good = false; break; }
            double total = 0.0;                                     // multiples of 0.5 far below 2^53: exact in any order
            for (int q = 0; q < P; ++q) {
                const int rows_q = (q == P - 1) ? n - q * n_loc : n_loc;
                for (int row = 0; row < rows_q; ++row) total += sent(q, row, r);
            }
            for (int i = 0; i < n && good; ++i) {
                const int q = std::min(i / n_loc, P - 1);
                if (vals[(size_t)i] != sent(q, i - q * n_loc, r)) good = false;
            }
            for (int b = 0; b < grid && good; ++b)
                if (sums[(size_t)b] != total) good = false;
        }
    }
    if (good) {
        // Close the self-test the way a solve ends: one exchange on channel 2.  A rank finishes it only after every peer has
        // pushed its channel-2 data, which a peer does, in stream order, after its last kernel of the second half -- so when this
        // call returns, every peer is done READING its self-test slots and the next layout (a problem's) may be written into any
        // mailbox.  The launcher's agreement on `ok` decides what happens next; it is no longer what makes the re-layout safe.
        // (If a peer's self-test failed it never gets here, and this wait ends at its bound: then the test has failed here too.)
        CGX_TRY(p2p_allgather(ctx, 2, dsrc, cgx::kSlots, ddst, cgx::kSlots, 1));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (check_p2p_error(ctx) != CGX_OK) good = false;
    }
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
    if (ctx->cfg.comm_mode == CGX_COMM_P2P && ctx->mailbox_shm) {
        // test only (cgx_probe_p2p_host_mailboxes): shared host segments, registered with the runtime
        for (int q = 0; q < ctx->nranks; ++q)
            if (ctx->host_maps[q]) {
                (void)hipHostUnregister(ctx->host_maps[q]);
                (void)munmap(ctx->host_maps[q], ctx->mailbox_bytes);
            }
        (void)shm_unlink(("/" + ctx->shm_prefix + "_" + std::to_string(ctx->cfg.rank)).c_str());
        if (ctx->d_p2p_err) (void)hipFree(ctx->d_p2p_err);
    } else if (ctx->cfg.comm_mode == CGX_COMM_P2P) {
        for (int q = 0; q < ctx->nranks; ++q)
            if (q != ctx->cfg.rank && ctx->mv.base[q]) (void)hipIpcCloseMemHandle(ctx->mv.base[q]);
        if (ctx->mailbox) (void)(ctx->mailbox_on_host ? hipHostFree(ctx->mailbox) : hipFree(ctx->mailbox));
        if (ctx->d_p2p_err) (void)hipFree(ctx->d_p2p_err);
    }
    (void)hipFree(ctx->res_xbuf);
    (void)hipFree(ctx->d_res_err);
    if (ctx->h_res_tail) (void)hipHostFree(ctx->h_res_tail);
    if (ctx->res_lock_fd >= 0) close(ctx->res_lock_fd);
    for (auto e : ctx->ev_pool) (void)hipEventDestroy(e);
    for (auto e : ctx->upd_pool) (void)hipEventDestroy(e);
    for (auto e : ctx->steps_ev)
        if (e) (void)hipEventDestroy(e);
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

}  // extern "C"
