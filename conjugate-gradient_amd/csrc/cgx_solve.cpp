// cgx_solve.cpp -- the exchanges and the CG driver loop behind include/cgx.h (begin / steps / end / solve).
//
// Reference path: CGSolver::solve, code/MPI/cg.cc:38-156 (citations are file:line under /root/reference).
// Design (MI355X-first, not a translation):
//   * the row block of A, b, x, r, Ap and the replicated p live in HBM for the life of the problem;
//   * rsold/rsnew/alpha/beta and the convergence flag stay on the device (cgx::Scalars): the host only
//     enqueues kernels and polls one int every `check_every` iterations, so the stream never drains;
//   * after convergence every kernel of the remaining enqueued iterations exits at its first
//     instruction, which reproduces the reference's `break` (cg.cc:120-121) exactly;
//   * one iteration = two kernels (K1 fused GEMV, K3 x/r update; on several ranks a small prefold kernel in front of
//     the exchange, or -- CGX_COMM_P2P -- all of that inside K3) and ONE exchange: an in-place all-gather of
//     equal segments [Ap slice | p.Ap partials] replaces MPI_Allreduce(p.Ap), MPI_Allreduce(r.r) and
//     MPI_Allgatherv(p): r and p are replicated, every rank updates all of r from the gathered Ap, reduces r.r
//     over all n rows in one fixed order (bit-identical everywhere) and forms p = r + beta p inside the next K1.
#include "cgx_internal.h"

#include <sys/file.h>

#include <cerrno>
#include <thread>

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

// ---- collectives ---------------------------------------------------------------------------------

// CGX_COMM_P2P: one lean all-gather kernel over the IPC-mapped mailboxes (cgx_kernels.hip).
cgx_status p2p_allgather(cgx_ctx *ctx, int chan, const double *src, int count, double *dst, long dst_stride,
                         int copy_self, int tail_off, int tail_n, int sum_off)
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
        // the exchange kernel folds this rank's partials and ships [Ap slice | one double].  Tagged words: these plain doubles
        // travel on channel 0 (own slots, own flag words, own epoch counter), never through the slots a tagged reader polls.
        Shard &s = ctx->shards[0];
        return p2p_allgather(ctx, ctx->mv.tagged ? 0 : 1, s.Ap(), ctx->seg_Sr, s.apg, ctx->seg_S, 0, ctx->seg_Sr,
                             with_tail ? ctx->npart : 0, ctx->seg_Sr + ctx->npart);
    }
    default: {
        Shard &s = ctx->shards[0];
        NCCL_TRY(ctx, ctx->rccl->AllGather(s.Ap(), s.apg, S, ncclDouble, ctx->comm, ctx->stream));
        return CGX_OK;
    }
    }
}

}  // namespace cgxi

namespace {

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

// cfg.profile_update: an event pair for the update kernel of an iteration whose K1 was timed (first shard only: one
// sample per iteration), or two null handles.
cgx_status take_update_events(cgx_ctx *ctx, hipEvent_t *e0, hipEvent_t *e1)
{
    *e0 = *e1 = nullptr;
    if (!ctx->cfg.profile_update || !ctx->gemv_timed_last || ctx->cfg.profile_markers) return CGX_OK;
    while (ctx->upd_used + 2 > ctx->upd_pool.size()) {
        hipEvent_t e;
        HIP_TRY(ctx, hipEventCreate(&e));
        ctx->upd_pool.push_back(e);
    }
    *e0 = ctx->upd_pool[ctx->upd_used++];
    *e1 = ctx->upd_pool[ctx->upd_used++];
    return CGX_OK;
}

}  // namespace

namespace cgxi {

// K1, plain form (vector given): initial residual, DEBUG verification, probes.
cgx_status run_gemv_plain(cgx_ctx *ctx, Shard &s, const double *v_full)
{
    if (ctx->banded)
        HIP_TRY(ctx, cgx::launch_spmv_dia_plain(s.plan, s.dia, s.rows, s.row0, ctx->n, ctx->lda, v_full, s.Ap(), s.k1_part(), s.sc,
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
    // The first launch of a steps call starts on a drained stream, right after a host-side synchronisation: whatever the
    // runtime or the clocks do at that point lands on it (the round-1 driver run recorded 2.06 ms there against 1.21 for
    // every other launch).  It is counted as discarded, never as a sample (cfg.profile_first overrides, for diagnostics).
    // Every later launch has the previous iteration's K3 queued in front of it.
    // At most 2048 timed launches per cgx_solve_steps call: the event pool stays bounded however long the run is.
    const long long seq = ctx->gemv_seq++;
    bool timed = false;
    if (every > 0 && ctx->ev_used + 2 <= 4096) {
        if (seq == 0) {
            timed = ctx->cfg.profile_first != 0;
            if (!timed) ctx->gemv_discarded++;
        } else {
            timed = ((seq - 1) % every) == 0;
        }
    }
    ctx->gemv_timed_last = timed;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (timed) {
        CGX_TRY(take_event(ctx, &e0));
        CGX_TRY(take_event(ctx, &e1));
        if (ctx->cfg.profile_markers) {   // old form: marker packets around the dispatch (kept for A/B, tools/window_probe.py)
            HIP_TRY(ctx, hipEventRecord(e0, ctx->stream));
            e0 = e1 = nullptr;
        }
    }
    hipEvent_t m1 = (timed && ctx->cfg.profile_markers) ? ctx->ev_pool[ctx->ev_used - 1] : nullptr;
    if (ctx->banded)
        HIP_TRY(ctx, cgx::launch_spmv_dia_fused(s.plan, s.dia, s.rows, s.row0, ctx->n, ctx->lda, s.p[k & 1], s.p[(k + 1) & 1],
                                                s.rv, s.Ap(), s.k1_part(), s.sc, k, ctx->tol, ctx->stream, e0, e1));
    else
        HIP_TRY(ctx, cgx::launch_gemv_fused(s.plan, s.A, ctx->lda, s.rows, s.row0, s.p[k & 1], s.p[(k + 1) & 1], s.rv,
                                            s.plan.split > 1 ? s.ap_parts : s.Ap(),
                                            (ctx->chunked && s.plan.light) ? nullptr : s.k1_part(),   // chunked: nobody folds K1's own partials
                                            s.sc, k, ctx->tol, ctx->stream, e0, e1, ctx->seg_Sr));
    if (m1) HIP_TRY(ctx, hipEventRecord(m1, ctx->stream));
    return CGX_OK;
}

void reset_gemv_stats(cgx_ctx *ctx)
{
    ctx->ev_used = 0;
    ctx->gemv_ms_sum = ctx->gemv_ms_min = ctx->gemv_ms_max = 0;
    ctx->gemv_launches = ctx->gemv_discarded = 0;
    ctx->gemv_seq = 0;
    ctx->gemv_samples.clear();
    ctx->gemv_timed_last = false;
    ctx->upd_used = 0;
    ctx->upd_samples.clear();
    ctx->steps_ev_pending = false;
    ctx->steps_device_ms = 0;
}

// Fold the recorded event pairs into the running K1 statistics (call after a stream sync).
cgx_status harvest_gemv_events(cgx_ctx *ctx)
{
    for (size_t i = 0; i + 1 < ctx->ev_used; i += 2) {
        float ms = 0.f;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev_pool[i], ctx->ev_pool[i + 1]));
        ctx->gemv_ms_sum += ms;
        if (ctx->gemv_launches == 0 || ms < ctx->gemv_ms_min) ctx->gemv_ms_min = ms;
        if (ctx->gemv_launches == 0 || ms > ctx->gemv_ms_max) ctx->gemv_ms_max = ms;
        ctx->gemv_launches++;
        ctx->gemv_samples.push_back(ms);
    }
    ctx->ev_used = 0;
    for (size_t i = 0; i + 1 < ctx->upd_used; i += 2) {
        float ms = 0.f;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->upd_pool[i], ctx->upd_pool[i + 1]));
        ctx->upd_samples.push_back(ms);
    }
    ctx->upd_used = 0;
    if (ctx->steps_ev_pending) {
        float ms = 0.f;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->steps_ev[0], ctx->steps_ev[1]));
        ctx->steps_device_ms = ms;
        ctx->steps_ev_pending = false;
    }
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
        hipEvent_t u0, u1;
        CGX_TRY(take_update_events(ctx, &u0, &u1));
        HIP_TRY(ctx, cgx::launch_update_xr_p2p(ctx->n, s.rows, s.row0, s.p[(k + 1) & 1], s.apv, ctx->npart, ctx->mv, 1, epoch,
                                               s.x, s.rv, s.sc, k & 1, ctx->p2p_timeout_ticks, ctx->d_p2p_err, st,
                                               s.plan.split > 1 ? s.ap_parts : s.Ap(), s.plan.split, ctx->seg_Sr, u0, u1));
        return CGX_OK;
    }
    // every other multi-rank consumer: K1's column pieces added up into the Ap slice of the segment, one p.Ap partial per
    // chunk of the slice into its tail (k_prefold_ap) -- that is what travels
    if (ctx->chunked)
        for (auto &s : ctx->shards)
            HIP_TRY(ctx, cgx::launch_prefold_ap(s.plan.split > 1 ? s.ap_parts : s.Ap(), s.plan.split, ctx->seg_Sr, s.rows, ctx->seg_Sr,
                                                s.p[(k + 1) & 1] + s.row0, s.Ap(), s.tail(), s.sc, st));
    CGX_TRY(gather_segments(ctx, true));                                                             // cg.cc:106
    const bool folded = ctx->cfg.comm_mode == CGX_COMM_P2P;   // the exchange kernel already folded each rank's partials
    for (auto &s : ctx->shards) {
        hipEvent_t u0 = nullptr, u1 = nullptr;
        if (&s == &ctx->shards[0]) CGX_TRY(take_update_events(ctx, &u0, &u1));
        HIP_TRY(ctx, cgx::launch_update_xr(ctx->n, s.rows, s.row0, s.p[(k + 1) & 1], s.apv, folded ? ctx->npart : 0,
                                           folded ? 1 : ctx->npart, s.x, s.rv, s.sc, k & 1, s.partials, st, u0, u1));   // cg.cc:105-116
    }
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

// {done, k_final} and the P2P error word in ONE pass: two small copies into pinned memory, one stream sync.
cgx_status read_flags_sync(cgx_ctx *ctx)
{
    Shard &s = ctx->shards[0];
    int *flags = ctx->h_flags + 4;   // third pinned slot {done, k_final, p2p error}: a pageable destination would be staged by the runtime
    const bool p2p = ctx->cfg.comm_mode == CGX_COMM_P2P && ctx->d_p2p_err;
    flags[2] = 0;
    HIP_TRY(ctx, hipMemcpyAsync(flags, &s.sc->done, 2 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    if (p2p) HIP_TRY(ctx, hipMemcpyAsync(flags + 2, ctx->d_p2p_err, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->done = flags[0] != 0;
    ctx->k_final = flags[1];
    if (p2p && flags[2])
        return fail(ctx, CGX_ERR_P2P, "direct peer exchange: a wait for a peer's flag expired (peer dead or IPC not coherent)");
    return CGX_OK;
}

// The loop cg.cc:95-137 as launches of a persistent kernel (cgx_resident.hip: A on the chip; cgx_stream.hip: A streamed): up
// to kResidentBatch iterations per launch (a launch of a non-converging solve stays bounded).  Between launches the state
// (x, r, p, rs[], done, k_final) lies in HBM in one of the shard's two state blocks; a launch reads state[cur] and writes the
// other block, which becomes the current one only when the launch came back with the error word down.  The break of
// cg.cc:120-121 is taken inside the kernel, at the iteration the reference takes it.
constexpr int kResidentBatch = 1 << 16;

namespace {

// one persistent grid at a time on this device (open_device_lock in cgx_context.cpp): held until the kernel has finished.
// Bounded like every other wait of this path: a holder that never lets go -- a stopped process, somebody's flock on the
// world-writable file -- costs the serialisation after 5 s, ONCE per context, not the solve.
struct DeviceLock {
    int fd;
    explicit DeviceLock(cgx_ctx *ctx) : fd(ctx->res_lock_gave_up ? -1 : ctx->res_lock_fd)
    {
        if (fd < 0) return;
        const double t0 = wall_now();
        while (flock(fd, LOCK_EX | LOCK_NB) != 0) {
            if (errno != EWOULDBLOCK || wall_now() - t0 > 5.0) {
                ctx->res_lock_gave_up = true;
                fd = -1;
                return;
            }
            std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
    }
    // (an early return through HIP_TRY has drained the stream first -- quiesce() -- so the grid is gone when the lock is)
    ~DeviceLock() { if (fd >= 0) (void)flock(fd, LOCK_UN); }
};

}  // namespace

// *redo = 0: all of nsteps (or up to the break) ran in persistent launches.  *redo > 0: a wait inside a launch expired (its
// workgroups were not all resident at once: another tenant of the GPU, a CU mask); the state that launch started from is
// intact, the context has left the persistent path for the rest of this problem, the iteration in flight between the two
// conventions has been enqueued, and *redo iterations remain for the per-launch loop of cgx_solve_steps.
cgx_status resident_steps(cgx_ctx *ctx, int nsteps, int *redo)
{
    Shard &s = ctx->shards[0];
    *redo = 0;
    int left = std::min(nsteps, ctx->max_iter - ctx->k);
    // diagnostics: CGX_RESIDENT_PROFILE=1 prints where workgroup 0 spent its cycles (per steps call, on stderr)
    DeviceScratch scratch;
    long long *d_prof = nullptr;
    constexpr int kProfWords = 8 + 4 * 256 * 4;   // 8 phase counters (cgx_resident.hip) + 4 stamps per workgroup (cgx_stream.hip)
    if (getenv("CGX_RESIDENT_PROFILE")) {
        HIP_TRY(ctx, scratch.alloc(&d_prof, kProfWords * sizeof(long long)));
        HIP_TRY(ctx, hipMemsetAsync(d_prof, 0, kProfWords * sizeof(long long), ctx->stream));
    }
    while (left > 0 && !ctx->done) {
        const int batch = std::min(left, kResidentBatch);
        const int k0 = ctx->k;
        cgx::ResidentArgs a{};
        a.A = s.A;
        a.lda = ctx->lda;
        a.n = ctx->n;
        a.rows_per_wg = ctx->rplan.rows_per_wg;
        a.xslots = ctx->rplan.xslots;
        a.in = s.state[s.cur];
        a.out = s.state[s.cur ^ 1];
        a.xbuf = ctx->res_xbuf;
        a.epoch0 = ctx->res_epoch;
        a.k0 = k0;
        a.iters = batch;
        a.tol = ctx->tol;
        a.timeout_ticks = ctx->res_timeout_ticks;
        a.err = ctx->d_res_err;
        a.tail = ctx->h_res_tail;
        a.stamp = ++ctx->res_stamp ? ctx->res_stamp : ++ctx->res_stamp;   // never 0
        a.prof = d_prof;
        {
            static const int stagger = [] { const char *e = getenv("CGX_STREAM_STAGGER"); return e ? atoi(e) : 1; }();
            a.stagger = stagger;
            static const int l2_rows = [] { const char *e = getenv("CGX_STREAM_L2_ROWS"); return e ? atoi(e) : -1; }();
            a.l2_rows = l2_rows >= 0 ? l2_rows : ctx->rplan.l2_rows;
        }
        a.mute_wg = ctx->res_mute_wg;
        ctx->res_mute_wg = -1;
        const cgx::ResidentTail *tail = ctx->h_res_tail;
        // cgx_solve (begin, all of the loop, end in one call) from a zero initial guess: this launch ends the loop whatever happens in it,
        // so what cgx_solve_end would enqueue -- the verification GEMV (cg.cc:146-147) and the end kernel -- goes behind it at once,
        // on the block the launch writes, and ONE host synchronisation serves both (a second one and the idle gap in front of it
        // were 20 us of the reference's window, 4 % of it at n = 1024).  A launch whose waits expired leaves garbage for those
        // two kernels to chew on: nothing of it is used, the swap is undone, the solve goes on as below.
        const bool spec = ctx->oneshot && ctx->lean && left <= kResidentBatch && !d_prof;
        auto undo_spec = [&]() {
            if (spec && ctx->end_enqueued) {
                s.cur ^= 1;
                bind_state(s, ctx->lda);
                ctx->end_enqueued = false;
            }
        };
        auto launch_and_wait = [&]() -> cgx_status {
            DeviceLock lock(ctx);
            HIP_TRY(ctx, cgx::launch_cg_resident(ctx->rplan, a, ctx->stream));
            if (spec) {
                s.cur ^= 1;                 // (speculatively: the block the launch writes)
                bind_state(s, ctx->lda);
                ctx->end_enqueued = true;
                CGX_TRY(run_gemv_plain(ctx, s, s.x));
                HIP_TRY(ctx, cgx::launch_solve_end(ctx->n, s.Ap(), s.b_full, s.x, s.sc, ctx->h_stage, ctx->stream));
            }
            // no copy command: the kernel has written its report into pinned memory itself, the launch's stamp last
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            return CGX_OK;
        };
        const cgx_status launched = launch_and_wait();
        if (launched != CGX_OK) {
            undo_spec();
            return launched;
        }
        const bool reported = tail->stamp == a.stamp;   // (always, once the kernel has ended; anything else is treated as a failed launch)
        if (reported) {
            ctx->res_rec[0] += tail->iterations;
            ctx->res_rec[1] += tail->watch_repeats;
            ctx->res_rec[2] += tail->gather_repeats;
            ctx->res_rec[3] += 1;
            ctx->res_rec[4] = std::max<long long>(ctx->res_rec[4], tail->waits[0][0]);
            ctx->res_rec[5] = std::max<long long>(ctx->res_rec[5], tail->waits[0][1]);
            for (int g = 0; g < ctx->rplan.grid && g < 256; ++g) {
                ctx->res_rec[6] = std::max<long long>(ctx->res_rec[6], tail->waits[g][0]);
                ctx->res_rec[7] = std::max<long long>(ctx->res_rec[7], tail->waits[g][1]);
            }
        }
        int flags[3] = {reported ? tail->done : 0, reported ? tail->k_final : 0, reported ? tail->err : 1};
        if (flags[2]) {
            undo_spec();
            // A wait for another workgroup's Ap expired: the grid was not resident at once.  Epochs that may have been used:
            ctx->res_epoch += (unsigned long long)batch;
            if (ctx->res_forced)
                return fail(ctx, CGX_ERR_HIP, "persistent-kernel solver (gemv_variant 40000): a wait for another workgroup's Ap expired (the "
                                              "grid was not resident at once?)");
            // The default choice has a way out: the launch wrote only the OTHER state block, so the state it
            // started from is intact; put the error word down, forget whatever tagged words the launch left, and go on -- from
            // iteration k0 -- on the per-launch path, for the rest of this problem.
            HIP_TRY(ctx, hipMemsetAsync(ctx->d_res_err, 0, sizeof(int), ctx->stream));
            HIP_TRY(ctx, hipMemsetAsync(ctx->res_xbuf, 0, ctx->res_xbuf_bytes, ctx->stream));
            ctx->resident = false;
            ctx->res_fallbacks++;
            ctx->err = "note: a persistent launch could not keep its workgroups resident at once; the solve continued on the per-launch path";
            if (k0 > 0) {
                // Between launches the persistent kernels leave p ALREADY formed (p_k0 in p[1], rs[k0 & 1] = rsold), where the
                // per-launch K1 of iteration k0 would form it from p_(k0-1) itself.  So iteration k0 runs as: the plain K1 on the
                // formed p (Ap and the p.Ap partials, no head), then K3 as ever; from k0 + 1 on the fused K1 finds what it expects:
                // p_old = p_k0 in p[(k0 + 1) & 1], K3's r.r partials behind r, rsold in rs[k0 & 1].
                double *pk = s.p[(k0 + 1) & 1];
                if (pk != s.p[1]) HIP_TRY(ctx, hipMemcpyAsync(pk, s.p[1], (size_t)ctx->lda * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
                CGX_TRY(run_gemv_plain(ctx, s, pk));
                HIP_TRY(ctx, cgx::launch_update_xr(ctx->n, s.rows, s.row0, pk, s.apv, 0, ctx->npart, s.x, s.rv, s.sc, k0 & 1, s.partials,
                                                   ctx->stream));                                   // cg.cc:105-116
                ctx->k = k0 + 1;
                --left;
            }
            *redo = left;
            return CGX_OK;
        }
        ctx->done = flags[0] != 0;
        ctx->k_final = flags[1];
        // (epochs used: one per iteration that ran -- up to and including the one that took the break)
        ctx->res_epoch += (unsigned long long)(ctx->done ? ctx->k_final - k0 + 1 : batch);
        ctx->k += batch;
        left -= batch;
        if (!spec) {
            s.cur ^= 1;             // the block the launch wrote is the state now
            bind_state(s, ctx->lda);
        }
    }
    if (d_prof && ctx->rplan.stream) {
        // one line per workgroup for the stamped iteration: xcd, begin of its sweep, row sums done, Ap gathered (us after the earliest begin)
        std::vector<long long> h(kProfWords);
        HIP_TRY(ctx, hipMemcpy(h.data(), d_prof, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
        for (int j = 0; j < 4; ++j) {
            const long long *hj = h.data() + 8 + 4 * 256 * j;
            long long t0 = 0;
            for (int g = 0; g < ctx->rplan.grid; ++g)
                if (hj[4 * g] && (!t0 || hj[4 * g] < t0)) t0 = hj[4 * g];
            for (int g = 0; g < ctx->rplan.grid && t0; ++g)
                fprintf(stderr, "cgx stream profile: n=%d sample=%d wg=%d xcd=%lld begin=%.2f sums=%.2f gathered=%.2f\n", ctx->n, j, g, hj[4 * g + 3],
                        (hj[4 * g] - t0) / 100.0, (hj[4 * g + 1] - t0) / 100.0, (hj[4 * g + 2] - t0) / 100.0);
        }
    } else if (d_prof) {
        long long h[8];
        HIP_TRY(ctx, hipMemcpy(h, d_prof, sizeof h, hipMemcpyDeviceToHost));
        const double it = h[7] > 0 ? (double)h[7] : 1.0;
        fprintf(stderr, "cgx resident profile: n=%d iterations=%lld cycles/iteration: gemv+publish %.0f  watch %.0f  gather %.0f  p.Ap %.0f  "
                        "update+r.r %.0f  | rounds/iteration: watch %.2f gather %.2f\n",
                ctx->n, h[7], h[0] / it, h[1] / it, h[2] / it, h[3] / it, h[4] / it, h[5] / it, h[6] / it);
    }
    return CGX_OK;
}

}  // namespace cgxi

extern "C" {

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
    reset_gemv_stats(ctx);
    hipStream_t st = ctx->stream;
    const int n = ctx->n;
    const size_t vec_bytes = (size_t)ctx->lda * sizeof(double);
    memset(ctx->res_rec, 0, sizeof ctx->res_rec);
    // A persistent solve from a zero initial guess (what the reference's main passes, cg_main.cc:48-50): everything below in
    // ONE kernel -- r = b - A 0 = b needs no GEMV.  With the persistent launch, the verification GEMV and the one-kernel
    // end (cgx_solve_end) the whole solve() is four launches and no copy command.
    ctx->lean = false;
    ctx->end_enqueued = false;
    if (ctx->resident && ctx->h_stage && ctx->shards.size() == 1 && ctx->shards[0].state[0]) {
        bool zero = true;
        for (int i = 0; i < n && zero; ++i) zero = x0[i] == 0.0 && !std::signbit(x0[i]);
        if (zero) {
            Shard &s = ctx->shards[0];
            HIP_TRY(ctx, cgx::launch_solve_begin_zero(n, ctx->lda, s.b_full, s.x, s.rv, s.p[0], s.p[1], s.apg,
                                                      (long)ctx->nranks * ctx->seg_S, s.sc, ctx->d_res_err, st));
            ctx->lean = true;
            ctx->in_solve = true;
            return CGX_OK;
        }
    }
    const double *x_src = x0;
    if (ctx->h_stage) {
        memcpy(ctx->h_stage, x0, (size_t)n * sizeof(double));
        x_src = ctx->h_stage;
    }
    for (auto &s : ctx->shards) {
        HIP_TRY(ctx, hipMemsetAsync(s.sc, 0, sizeof(Scalars), st));
        HIP_TRY(ctx, hipMemsetAsync(s.apg, 0, (size_t)ctx->nranks * ctx->seg_S * sizeof(double), st));
        HIP_TRY(ctx, hipMemsetAsync(s.rbuf, 0, (size_t)s.rv.S * sizeof(double), st));
        // x (initial guess) replicated for the first GEMV, x_sub = x[rows]  (cg.cc:72, 80)
        if (ctx->h_stage) HIP_TRY(ctx, cgx::launch_copy_doubles(s.p[0], ctx->h_stage, n, st));   // reads the pinned buffer over PCIe
        else HIP_TRY(ctx, hipMemcpyAsync(s.p[0], x_src, (size_t)n * sizeof(double), hipMemcpyHostToDevice, st));
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
    if (ctx->resident) HIP_TRY(ctx, hipMemsetAsync(ctx->d_res_err, 0, sizeof(int), st));   // a solve starts with the error word down
    ctx->in_solve = true;
    return CGX_OK;
}

cgx_status cgx_solve_steps(cgx_ctx *ctx, int nsteps, int *done_out)
{
    if (!ctx || !ctx->in_solve) return fail(ctx, CGX_ERR_BAD_ARG, "cgx_solve_steps outside begin/end");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const double t0 = wall_now();
    // K1 statistics describe the most recent steps call (bench.py: the timed region, not the warmup)
    reset_gemv_stats(ctx);
    if (ctx->resident) {
        int redo = 0;
        CGX_TRY(resident_steps(ctx, nsteps, &redo));
        if (redo == 0) {
            ctx->t_loop += wall_now() - t0;
            if (done_out) *done_out = ctx->done ? 1 : 0;
            return CGX_OK;
        }
        nsteps = redo;   // a persistent launch's waits expired: the rest of the call runs below, on the per-launch path
    }
    bool window_open = false;   // the start marker of THIS call is on the stream (a stop marker is only paired with that)
    if (ctx->cfg.profile_gemv && nsteps > 0 && !ctx->done) {
        for (auto &e : ctx->steps_ev)
            if (!e) HIP_TRY(ctx, hipEventCreate(&e));
        HIP_TRY(ctx, hipEventRecord(ctx->steps_ev[0], ctx->stream));
        window_open = true;
    }
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
    if (window_open) {
        HIP_TRY(ctx, hipEventRecord(ctx->steps_ev[1], ctx->stream));
        ctx->steps_ev_pending = true;
    }
    CGX_TRY(read_flags_sync(ctx));
    // the event pairs are read later (cgx_get_gemv_samples / cgx_solve_end): the elapsed-time queries of a few dozen
    // pairs are not part of the loop and must not sit inside a caller's timing window
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
    // (the LDS-resident kernel has made that test itself; with no iteration done the state is the per-launch path's)
    if (!ctx->resident || ctx->k == 0) {
        for (auto &s : ctx->shards) HIP_TRY(ctx, cgx::launch_close_iteration(s.sc, s.rv, ctx->k, ctx->tol, st));
        CGX_TRY(read_flags_sync(ctx));
    }   // (resident: done / k_final were read behind the last launch, resident_steps; one host synchronisation less)
    const int k_exit = ctx->done ? ctx->k_final : ctx->k;

    if (ctx->lean && ctx->resident && ctx->k > 0) {
        // The persistent path's end in two launches: x lies whole in the current state block (one GPU: every row is this shard's,
        // the pad columns are zero), so the verification GEMV (cg.cc:146-147) takes it as it is, and ONE kernel does the DEBUG
        // norms (cg.cc:148-151) and puts x, the three sums and rs[] into the pinned buffer.
        Shard &s = ctx->shards[0];
        if (!ctx->end_enqueued) {   // (cgx_solve has put both behind the persistent launch already: resident_steps)
            CGX_TRY(run_gemv_plain(ctx, s, s.x));
            HIP_TRY(ctx, cgx::launch_solve_end(ctx->n, s.Ap(), s.b_full, s.x, s.sc, ctx->h_stage, st));
            HIP_TRY(ctx, hipStreamSynchronize(st));
        }
        ctx->end_enqueued = false;
        const double *o = ctx->h_stage + ctx->n;
        if (x) memcpy(x, ctx->h_stage, (size_t)ctx->n * sizeof(double));
        ctx->in_solve = false;
        ctx->lean = false;
        if (res) {
            memset(res, 0, sizeof *res);
            res->iterations = k_exit;
            res->converged = ctx->done ? 1 : 0;
            res->residual_prev = std::sqrt(o[3 + (k_exit & 1)]);           // sqrt(rsold) as printed, cg.cc:152-153
            res->residual_last = std::sqrt(o[3 + ((k_exit + 1) & 1)]);
            if (!ctx->done) res->residual_last = res->residual_prev;        // loop ran out: rsold == rsnew (cg.cc:132)
            res->x_norm = std::sqrt(o[2]);
            res->rel_residual = std::sqrt(o[0]) / std::sqrt(o[1]);
            res->seconds_solve = wall_now() - ctx->t_begin;
            res->seconds_loop = ctx->t_loop;
            res->gemv_bytes = 8.0 * ((double)s.rows * ctx->n + ctx->n + s.rows);
        }
        return CGX_OK;
    }
    ctx->lean = false;

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
    double *x_dst = (x && ctx->h_stage) ? ctx->h_stage : x;
    if (x && ctx->h_stage) HIP_TRY(ctx, cgx::launch_copy_doubles(ctx->h_stage, s0.p[0], ctx->n, st));   // writes the pinned buffer
    else if (x) HIP_TRY(ctx, hipMemcpyAsync(x, s0.p[0], (size_t)ctx->n * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    if (ctx->ev_used || ctx->upd_used || ctx->steps_ev_pending) CGX_TRY(harvest_gemv_events(ctx));
    if (x && x_dst != x) memcpy(x, x_dst, (size_t)ctx->n * sizeof(double));
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
        res->gemv_ms_max = ctx->gemv_ms_max;
        res->gemv_discarded = ctx->gemv_discarded;
        res->steps_device_ms = ctx->steps_device_ms;
        if (!ctx->gemv_samples.empty()) {
            std::vector<float> v(ctx->gemv_samples);
            const size_t mid = v.size() / 2;
            std::nth_element(v.begin(), v.begin() + mid, v.end());
            double med = v[mid];
            if (v.size() % 2 == 0) med = 0.5 * (med + *std::max_element(v.begin(), v.begin() + mid));
            res->gemv_ms_median = med;
        }
        res->gemv_bytes = ctx->banded ? 8.0 * ((double)s0.rows * s0.dia.ndiag + 2.0 * s0.rows)
                                      : 8.0 * ((double)s0.rows * ctx->n + ctx->n + s0.rows);
    }
    return CGX_OK;
}

cgx_status cgx_get_gemv_samples(cgx_ctx *ctx, double *ms_out, int cap, int *count)
{
    if (!ctx || !count || (cap > 0 && !ms_out)) return CGX_ERR_BAD_ARG;
    if (ctx->ev_used || ctx->upd_used || ctx->steps_ev_pending) {
        if (hipSetDevice(ctx->device) != hipSuccess) return CGX_ERR_HIP;
        CGX_TRY(harvest_gemv_events(ctx));
    }
    *count = (int)ctx->gemv_samples.size();
    for (int i = 0; i < cap && i < *count; ++i) ms_out[i] = ctx->gemv_samples[(size_t)i];
    return CGX_OK;
}

cgx_status cgx_get_update_samples(cgx_ctx *ctx, double *ms_out, int cap, int *count)
{
    if (!ctx || !count || (cap > 0 && !ms_out)) return CGX_ERR_BAD_ARG;
    if (ctx->ev_used || ctx->upd_used || ctx->steps_ev_pending) {
        if (hipSetDevice(ctx->device) != hipSuccess) return CGX_ERR_HIP;
        CGX_TRY(harvest_gemv_events(ctx));
    }
    *count = (int)ctx->upd_samples.size();
    for (int i = 0; i < cap && i < *count; ++i) ms_out[i] = ctx->upd_samples[(size_t)i];
    return CGX_OK;
}

cgx_status cgx_solve(cgx_ctx *ctx, double *x, cgx_result *res)
{
    if (!ctx || !x) return fail(ctx, CGX_ERR_BAD_ARG, "cgx_solve: bad argument");
    CGX_TRY(cgx_solve_begin(ctx, x));
    int done = 0;
    ctx->oneshot = true;
    const cgx_status st = cgx_solve_steps(ctx, ctx->max_iter, &done);
    ctx->oneshot = false;
    if (st != CGX_OK) {
        ctx->end_enqueued = false;
        return st;
    }
    return cgx_solve_end(ctx, x, res);
}

}  // extern "C"
