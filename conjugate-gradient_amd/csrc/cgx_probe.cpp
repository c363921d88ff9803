// cgx_probe.cpp -- kernel probes of include/cgx.h: the individual hot ops through the C ABI, for the parity tests.
#include "cgx_internal.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
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

extern "C" {

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
        HIP_TRY(ctx, cgx::launch_reduce_partials(s.k1_part(), s.plan.grid / std::max(s.plan.split, 1), &s.sc->local[cgx::kSlotConj], st));
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
    DeviceScratch scratch;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIP_TRY(ctx, scratch.event(&e0));
    HIP_TRY(ctx, scratch.event(&e1));
    for (auto &s : ctx->shards) HIP_TRY(ctx, hipMemsetAsync(s.sc, 0, sizeof(Scalars), st));
    for (auto &s : ctx->shards) CGX_TRY(run_gemv_plain(ctx, s, s.p[0]));   // warm
    HIP_TRY(ctx, hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i)
        for (auto &s : ctx->shards) CGX_TRY(run_gemv_plain(ctx, s, s.p[0]));
    HIP_TRY(ctx, hipEventRecord(e1, st));
    HIP_TRY(ctx, hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, e0, e1));
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
    DeviceScratch scratch;   // nine buffers: freed on every return path below
    HIP_TRY(ctx, scratch.alloc(&dx, bytes));
    HIP_TRY(ctx, scratch.alloc(&dp0, vbytes));
    HIP_TRY(ctx, scratch.alloc(&dp1, vbytes));
    HIP_TRY(ctx, scratch.alloc(&dap, (size_t)S * sizeof(double)));
    HIP_TRY(ctx, scratch.alloc(&drb, (size_t)(lda + grid) * sizeof(double)));
    HIP_TRY(ctx, scratch.alloc(&dpart, (size_t)(grid + 8) * sizeof(double)));
    HIP_TRY(ctx, scratch.alloc(&dA, vbytes));
    HIP_TRY(ctx, scratch.alloc(&dAp1, 64 * sizeof(double)));
    HIP_TRY(ctx, scratch.alloc(&dsc, sizeof(Scalars)));
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
    cgx::GemvPlan plan = cgx::plan_gemv(ctx->cfg.gemv_variant, 1, n, lda);
    if (plan.split > 1) {   // the probe has one Ap row: no column pieces
        plan.grid /= plan.split;
        plan.split = 1;
    }
    HIP_TRY(ctx, cgx::launch_gemv_fused(plan, dA, lda, 1, 0, dp0, dp1, rv, dAp1, dAp1 + 8, dsc, 1, -1.0 /* never converges */, st));
    HIP_TRY(ctx, hipMemcpyAsync(p, dp1, bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    if (rr) *rr = rr_host;
    return CGX_OK;
}

cgx_status cgx_probe_set_fault_after(cgx_ctx *ctx, int calls)
{
    if (!ctx) return CGX_ERR_BAD_ARG;
    ctx->fault_after = calls;
    return CGX_OK;
}

// Host only (no context, no device): the Matrix-Market parser by itself.
cgx_status cgx_probe_parse_matrix_market(const char *path, int threads, int *m, int *n, int *nz, int *symmetric, int *I, int *J,
                                         double *a, long cap, char *err, int err_cap)
{
    if (!path) return CGX_ERR_BAD_ARG;
    MtxEntries e;
    std::string msg;
    const cgx_status st = parse_matrix_market(path, &e, &msg, threads == 0 ? default_parse_threads() : threads, cap <= 0);   // cap 0: the sizes only
    if (err && err_cap > 0) {
        strncpy(err, msg.c_str(), (size_t)err_cap - 1);
        err[err_cap - 1] = '\0';
    }
    if (st != CGX_OK) return st;
    if (m) *m = e.m;
    if (n) *n = e.n;
    if (nz) *nz = e.nz;
    if (symmetric) *symmetric = e.sym ? 1 : 0;
    for (long z = 0; z < cap && z < (long)e.a.size(); ++z) {
        if (I) I[z] = e.I[(size_t)z];
        if (J) J[z] = e.J[(size_t)z];
        if (a) a[z] = e.a[(size_t)z];
    }
    return CGX_OK;
}

// TEST ONLY: move the mailbox of a ONE-rank P2P context into pinned, coherent HOST memory.  Every store of the exchange then
// leaves the device over PCIe and every poll and load comes back over it: the system-scope path for real (not a neighbour
// GPU over xGMI, but memory that is neither this GPU's HBM nor behind its L2), at several times the latency.  What a
// one-GPU box can offer towards "nothing rests on how local memory happens to behave".
cgx_status cgx_probe_p2p_mailbox_to_host(cgx_ctx *ctx)
{
    if (!ctx || ctx->cfg.comm_mode != CGX_COMM_P2P || ctx->nranks != 1 || !ctx->shards.empty())
        return fail(ctx, CGX_ERR_BAD_ARG, "cgx_probe_p2p_mailbox_to_host: a one-rank P2P context without a problem");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    unsigned char *host = nullptr;
    HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void **>(&host), ctx->mailbox_bytes, hipHostMallocCoherent | hipHostMallocMapped));
    memset(host, 0, ctx->mailbox_bytes);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    (void)(ctx->mailbox_on_host ? hipHostFree(ctx->mailbox) : hipFree(ctx->mailbox));
    ctx->mailbox = host;
    ctx->mailbox_on_host = true;
    ctx->mv.base[ctx->cfg.rank] = host;
    return CGX_OK;
}

// TEST ONLY: move the epoch counter of mailbox channel `chan` forward to `value` (the next exchange on it is value + 1), so
// that the tests can reach the regions of the 64-bit counter a run would need ~10^9 exchanges for: the wrap of the 32-bit tag
// of the tagged-word form (epoch = k * (2^32 - 1)), the wrap of the epoch's low 32 bits, and round 3's hazard regions (2^19,
// 0xFFF80000).  Every rank must make the same call at a quiet point (no solve in progress; between two solves is one: the
// last exchange of a solve is on channel 2, so every peer is done with channels 0 and 1).  Backwards is refused: the flag
// words only ever grow.
cgx_status cgx_probe_set_p2p_epoch(cgx_ctx *ctx, int chan, unsigned long long value)
{
    if (!ctx || ctx->cfg.comm_mode != CGX_COMM_P2P || chan < 0 || chan >= cgx::kP2pChannels)
        return fail(ctx, CGX_ERR_BAD_ARG, "cgx_probe_set_p2p_epoch: a P2P context and a channel 0..2");
    if (ctx->in_solve) return fail(ctx, CGX_ERR_BAD_ARG, "cgx_probe_set_p2p_epoch inside begin/end");
    if (value < ctx->p2p_epoch[chan]) return fail(ctx, CGX_ERR_BAD_ARG, "cgx_probe_set_p2p_epoch: the epoch can only move forward");
    ctx->p2p_epoch[chan] = value;
    return CGX_OK;
}

cgx_status cgx_probe_resident_test(cgx_ctx *ctx, unsigned long long epoch, int mute_workgroup)
{
    if (!ctx) return CGX_ERR_BAD_ARG;
    // (the epoch only between solves; muting a workgroup of the next launch also between two cgx_solve_steps calls)
    if (ctx->in_solve && epoch > 0) return fail(ctx, CGX_ERR_BAD_ARG, "cgx_probe_resident_test: the epoch cannot be moved inside begin/end");
    if (epoch > 0) {
        if (epoch < ctx->res_epoch) return fail(ctx, CGX_ERR_BAD_ARG, "cgx_probe_resident_test: the epoch can only move forward");
        ctx->res_epoch = epoch;
    }
    ctx->res_mute_wg = mute_workgroup;
    return CGX_OK;
}

cgx_status cgx_probe_get_p2p_epoch(cgx_ctx *ctx, int chan, unsigned long long *value)
{
    if (!ctx || !value || ctx->cfg.comm_mode != CGX_COMM_P2P || chan < 0 || chan >= cgx::kP2pChannels) return CGX_ERR_BAD_ARG;
    *value = ctx->p2p_epoch[chan];
    return CGX_OK;
}

// TEST ONLY: the mailboxes of a MULTI-rank P2P job in POSIX shared HOST memory.  With device mailboxes and every rank on
// one GPU (all a one-GPU box offers) a "peer's" mailbox is this GPU's own HBM behind its own L2; here every store of every
// rank leaves the GPU over PCIe into host DRAM and every poll and load of every rank comes back over it, between separate
// processes: memory that is remote for every party, written by one process and polled by another.
//   stage 0: create, size and zero this rank's segment <prefix>_<rank>, register it with the runtime, make it THE mailbox;
//   -- launcher barrier (every segment exists) --
//   stage 1: map and register every peer's segment.  Replaces cgx_p2p_export / cgx_p2p_import; before any problem is set.
cgx_status cgx_probe_p2p_host_mailboxes(cgx_ctx *ctx, const char *prefix, int stage)
{
    if (!ctx || !prefix || ctx->cfg.comm_mode != CGX_COMM_P2P || !ctx->shards.empty() || (stage != 0 && stage != 1))
        return fail(ctx, CGX_ERR_BAD_ARG, "cgx_probe_p2p_host_mailboxes: a P2P context without a problem, stage 0 or 1");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int me = ctx->cfg.rank;
    auto map_segment = [&](int q, bool create, void **out) -> cgx_status {
        const std::string name = "/" + std::string(prefix) + "_" + std::to_string(q);
        const int fd = shm_open(name.c_str(), create ? (O_CREAT | O_RDWR) : O_RDWR, 0600);
        if (fd < 0) return fail(ctx, CGX_ERR_IO, "shm_open(" + name + ") failed");
        if (create && ftruncate(fd, (off_t)ctx->mailbox_bytes) != 0) {
            close(fd);
            return fail(ctx, CGX_ERR_IO, "ftruncate(" + name + ") failed");
        }
        void *ptr = mmap(nullptr, ctx->mailbox_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (ptr == MAP_FAILED) return fail(ctx, CGX_ERR_IO, "mmap(" + name + ") failed");
        if (create) memset(ptr, 0, ctx->mailbox_bytes);
        const hipError_t e = hipHostRegister(ptr, ctx->mailbox_bytes, hipHostRegisterMapped | hipHostRegisterPortable);
        if (e != hipSuccess) {
            munmap(ptr, ctx->mailbox_bytes);
            return fail(ctx, CGX_ERR_HIP, std::string("hipHostRegister(") + name + "): " + hipGetErrorString(e));
        }
        *out = ptr;
        return CGX_OK;
    };
    auto device_view = [&](void *host, unsigned char **dev) -> cgx_status {
        void *d = nullptr;
        HIP_TRY(ctx, hipHostGetDevicePointer(&d, host, 0));
        *dev = static_cast<unsigned char *>(d);
        return CGX_OK;
    };
    if (stage == 0) {
        if (ctx->mailbox_shm || ctx->mailbox_on_host) return fail(ctx, CGX_ERR_BAD_ARG, "the mailbox has already been moved");
        void *mine = nullptr;
        CGX_TRY(map_segment(me, true, &mine));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->mailbox);
        ctx->host_maps[me] = mine;
        ctx->mailbox_shm = true;
        ctx->shm_prefix = prefix;
        unsigned char *dev = nullptr;
        CGX_TRY(device_view(mine, &dev));
        ctx->mailbox = dev;
        ctx->mv.base[me] = dev;
        ctx->p2p_ready = ctx->nranks == 1;
        return CGX_OK;
    }
    if (!ctx->mailbox_shm) return fail(ctx, CGX_ERR_BAD_ARG, "stage 1 before stage 0");
    for (int q = 0; q < ctx->nranks; ++q) {
        if (q == me || ctx->host_maps[q]) continue;
        void *ptr = nullptr;
        CGX_TRY(map_segment(q, false, &ptr));
        ctx->host_maps[q] = ptr;
        CGX_TRY(device_view(ptr, &ctx->mv.base[q]));
    }
    ctx->p2p_ready = true;
    return CGX_OK;
}

cgx_status cgx_probe_persistent_plan(int n, int cus, long lds_per_cu, int streaming, long out[CGX_PERSISTENT_PLAN_INTS])
{
    if (!out || n < 1 || cus < 1 || lds_per_cu < 0) return CGX_ERR_BAD_ARG;
    cgx::ResidentPlan pl{};
    const bool fits = streaming ? cgx::plan_stream(n, cus, (size_t)lds_per_cu, &pl) : cgx::plan_resident(n, cus, (size_t)lds_per_cu, &pl);
    const long r[CGX_PERSISTENT_PLAN_INTS] = {fits ? 1 : 0, pl.R, pl.S, pl.grid, pl.xslots, (long)pl.lds_bytes, pl.RL, pl.RG, pl.RB, pl.l2_rows,
                                              pl.hybrid, pl.stream ? 512 : 256};
    for (int i = 0; i < CGX_PERSISTENT_PLAN_INTS; ++i) out[i] = fits || i == 0 ? r[i] : 0;
    return CGX_OK;
}

cgx_status cgx_probe_set_resident_limit(cgx_ctx *ctx, int workgroups)
{
    if (!ctx) return CGX_ERR_BAD_ARG;
    ctx->resident_limit = workgroups;
    return CGX_OK;
}

// The DEVICE copy of b on local shard `local_shard` (n doubles): what init_source_term (cg.cc:218-234) left in HBM.
cgx_status cgx_probe_get_source_term(cgx_ctx *ctx, int local_shard, double *b_out)
{
    if (!ctx || !b_out || local_shard < 0 || local_shard >= (int)ctx->shards.size())
        return fail(ctx, CGX_ERR_BAD_ARG, "cgx_probe_get_source_term: bad argument");
    if (!ctx->have_b) return fail(ctx, CGX_ERR_BAD_ARG, "no source term");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpy(b_out, ctx->shards[local_shard].b_full, (size_t)ctx->n * sizeof(double), hipMemcpyDeviceToHost));
    return CGX_OK;
}

// TEST PROBE: overwrite the dense row block of every local shard of the CURRENT problem with the hash matrix of
// cgx_kernels.h (hash_entry): element (i, j) = a pure function of (seed, i, j) in [-1, 1), filled on the device.  The
// geometry (n, partition, pitch, K1 plan) stays what the problem set before it defined; b and max_iter are untouched.
cgx_status cgx_probe_fill_matrix_hash(cgx_ctx *ctx, unsigned long long seed, int symmetric, double diag)
{
    if (!ctx) return CGX_ERR_BAD_ARG;
    if (!ctx->have_matrix || ctx->shards.empty())
        return fail(ctx, CGX_ERR_BAD_ARG, "cgx_probe_fill_matrix_hash: set a problem first (it defines n and the row blocks)");
    if (ctx->banded) return fail(ctx, CGX_ERR_UNSUPPORTED, "cgx_probe_fill_matrix_hash: dense storage only");
    if (ctx->in_solve) return fail(ctx, CGX_ERR_BAD_ARG, "cgx_probe_fill_matrix_hash inside begin/end");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    for (auto &s : ctx->shards)
        HIP_TRY(ctx, cgx::launch_fill_hash(s.A, ctx->lda, ctx->n, s.row0, s.rows, seed, symmetric ? 1 : 0, diag, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
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
