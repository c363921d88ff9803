// cgx_kernels.h -- launch interface of the CDNA4 (gfx950) kernels of the CG hot path.
// The host code (cgx_context / cgx_matrix / cgx_solve / cgx_probe .cpp) sees only these plain functions; all device code is in cgx_kernels.hip.
#pragma once

#include <hip/hip_runtime.h>

namespace cgx {

// Small scalar exchange of the verification phase (cg.cc:144-151): every shard publishes kSlots doubles, the
// host sums slot v over ranks in rank order.
constexpr int kSlots = 4;
constexpr int kSlotConj = 0;   // ||Ax-b||^2 partial (also: p.Ap of the gemv probe)
constexpr int kSlotRr = 1;     // ||b||^2 partial
constexpr int kSlotX = 2;      // ||x||^2 partial
constexpr int kMaxRanks = 64;  // gathered[] holds kMaxRanks*kSlots doubles

// Device-resident scalar block of one shard.  Nothing in the iteration loop is read back by the
// host except `done` (polled every check_every iterations).
struct Scalars {
    double rs[2];          // rsold / rsnew ping-pong: rs[k&1] is rsold of iteration k          (cg.cc:91,116,132)
    double local[kSlots];  // send buffer of the small scalar all-gather (verification phase)
    double dbg[4];
    int    done;           // set when sqrt(rsnew) < tol (cg.cc:120-121); later kernels exit at once
    int    k_final;        // k of the converging iteration
    unsigned pad[2];
};

// Segmented vector: P equal segments of S doubles, segment q = [ slice of rank q (Sr doubles, zero padded) |
// tail ].  Used for the exchanged Ap ([Ap slice | p.Ap partials], one in-place all-gather per iteration in
// place of MPI_Allreduce cg.cc:106 + MPI_Allgatherv cg.cc:135-136) and, with nranks = 1, for the replicated
// r ([r (lda) | r.r partials of K3's workgroups]).
struct SegView {
    double *base;     // this shard's copy of all P segments
    int S, Sr;        // segment stride, r part
    int n_loc;        // floor(n / nranks): rows of every rank but the last (cg.cc:255)
    int nranks;
    int n;
    int rank;         // this shard
    int seg_gap;      // S - n_loc: r[c] of owner q sits at base[c + q*seg_gap]
    unsigned div_magic, div_shift;   // c / n_loc == umulhi(c, div_magic) >> div_shift for 0 <= c < 2^31
};

// Fill seg_gap and the division magic of a SegView whose S, n_loc, nranks are set.
void seg_finalize(SegView *sv);

struct GemvPlan {
    int variant;     // 1 = column-split (p from L2 to registers), 2 = row-split (p tiles staged in LDS),
                     // 3 = banded storage (DiaView), one thread per row
    int R;           // rows per wave (variant 2) or per workgroup (variant 1)
    int U;           // column-step unroll
    int waves;       // waves per workgroup
    int nt;          // non-temporal loads of A
    int grid;        // workgroups
    int rows_per_wg;
    int ncols;       // columns the sweep covers: the logical n rounded up to even (16-B pieces); the pad columns up to the
                     // pitch are zero in A and in every vector and are never touched
    int split;       // variant 1, fused launches only: column pieces per row group (1 = none).  grid counts ALL workgroups
                     // (row groups x split) = the number of p.Ap partials; a plain launch uses grid / split
    int light;       // variant 1: the one-round form (first trip issued ahead of the iteration head, 256 registers to spend;
                     // see k_gemv_colsplit); variant 3: the LDS-window form of K1b
};

// Choose the K1 shape for a shard of `rows` x `n` held at pitch `lda` (variant 0 = default).  allow_split: the consumer of
// Ap adds column pieces itself (every multi-rank transport: launch_prefold_ap or the fused P2P update), so the default may
// cut the columns of a row group into pieces.
GemvPlan plan_gemv(int variant, int rows, int n, long lda, bool allow_split = false);

// K1, plain form: Ap = A[rows x lda] * v ; partials[wg] = sum over the workgroup's rows of v_local[i]*Ap[i].
// Used for the initial residual (cg.cc:79-81), the DEBUG verification (cg.cc:146-147) and the probes.
hipError_t launch_gemv_plain(const GemvPlan &plan, const double *A, long lda, int rows, const double *v_full,
                             const double *v_local, double *Ap, double *partials, Scalars *sc, hipStream_t s);

// K1, fused form = body of iteration k up to p.Ap (cg.cc:100-105) preceded by the tail of iteration k-1
// (cg.cc:117-132): rsnew = sum of the gathered r.r; convergence test; beta; p_new = r + beta p_old computed
// on the fly for every column (and stored once), Ap = A p_new, partials[wg] = the workgroup's part of p_new_local . Ap.
// e_start / e_stop (both or neither): bound to the dispatch itself (hipExtLaunchKernel), so their difference is the
// kernel's own begin -> end as rocprofv3 sees it, with no marker packets on the stream.
// plan.split > 1: Ap is the first of plan.split partial vectors, ap_stride doubles apart (piece s of the columns
// writes Ap + s * ap_stride); the consumer adds them in ascending order (launch_prefold_ap, or the fused P2P update).
hipError_t launch_gemv_fused(const GemvPlan &plan, const double *A, long lda, int rows, int row0,
                             const double *p_old, double *p_new, SegView seg, double *Ap, double *partials,
                             Scalars *sc, int k, double tol, hipStream_t s, hipEvent_t e_start = nullptr,
                             hipEvent_t e_stop = nullptr, long ap_stride = 0);
// Chunks of a rank's Ap slice (see "Chunks" in cgx_kernels.hip): kChunkRows consecutive rows, one workgroup each.
constexpr int kChunkRows = 512;
inline int chunks_per_rank(int Sr) { return (Sr + kChunkRows - 1) / kChunkRows; }
// In front of the exchange of every multi-rank transport but the fused P2P update: dst[i] = parts[i] + parts[stride + i]
// + ... (split terms, ascending; split == 1: parts == dst, nothing is stored), i < Sr, and tail[c] = the part of
// p_sub . Ap_sub (cg.cc:105) of chunk c, p_loc = p_new + row0.  K3 then folds P x chunks_per_rank(Sr) partials instead
// of every K1 workgroup's.
hipError_t launch_prefold_ap(const double *parts, int split, long stride, int rows, int Sr, const double *p_loc, double *dst,
                             double *tail, const Scalars *sc, hipStream_t s);

// K3: p.Ap = fixed-order sum over all ranks q of the tail_count doubles at tail_off of segment q's tail;
// alpha = rsold / max(p.Ap, rsold*1e-14); x_sub += alpha p_sub (own rows); r -= alpha Ap for ALL n rows (r is
// replicated, rv = [r (lda) | one r.r partial per K3 workgroup]); the next K1's head folds the partials.  cg.cc:105-116.
hipError_t launch_update_xr(int n, int rows, int row0, const double *p_new, SegView apv, int tail_off, int tail_count,
                            double *x, SegView rv, Scalars *sc, int parity, double *partials, hipStream_t s,
                            hipEvent_t e_start = nullptr, hipEvent_t e_stop = nullptr);   // optional: bound to the dispatch
int update_xr_grid(int count);   // ceil(count/256), at most kMaxVectorGrid (above that the kernels stride over the rows)
constexpr int kMaxVectorGrid = 1024;

// Tail of the LAST executed iteration when the loop runs out (k = number of iterations done):
// rsnew -> rs[k&1], convergence test (cg.cc:117-121,132).  One thread.
hipError_t launch_close_iteration(Scalars *sc, SegView seg, int k, double tol, hipStream_t s);

// K2: out[0..NV) = deterministic sum of partials (stand-alone form, setup/verification only).
hipError_t launch_reduce_partials(const double *partials, int n, double *out, hipStream_t s);
hipError_t launch_reduce_partials3(const double *partials, int n, double *out3, hipStream_t s);

// Initial residual (cg.cc:79-85): r = b - Ap for all n rows (Ap from the gathered segments); rv tail[wg] = sum r_i^2.
hipError_t launch_init_residual(int n, const double *b_full, SegView apv, SegView rv, double *partials, hipStream_t s);

// v_full[c] = segment value of column c (c < n), 0 for the pad: turns gathered slices into a replicated vector.
hipError_t launch_unpack_segments(SegView seg, double *v_full, long lda, hipStream_t s);

// dst[0..count) = src[0..count) by a kernel; dst or src may be pinned host memory.
hipError_t launch_copy_doubles(double *dst, const double *src, long count, hipStream_t s);

// DEBUG block (cg.cc:144-151): partials[3*wg + {0,1,2}] = sum (Ax-b)^2, b^2, x^2 over this shard's rows.
hipError_t launch_debug_norms(int count, const double *Ax, const double *b, const double *x, double *partials,
                              hipStream_t s);

// generate_lap2d_matrix (cg.cc:159-188) for rows [row0,row0+rows) straight into device memory; pad columns = 0.
hipError_t launch_generate_lap2d(double *A, long lda, int size, int row0, int rows, hipStream_t s);

// ---- dense, incompressible test data (TEST PROBE, cgx_probe_fill_matrix_hash; not a reference function) -------------
// The reference's generator leaves 5 non-zeros per row (cg.cc:178-186): at N = 32768 the matrix K1 streams is 99.985 % zeros.
// Its GEMV is a general dense dgemv (cg.cc:101-102), so parity and rate are also shown on a block in which every element is
// a different number with a full random mantissa: element (i, j) is a pure function of (seed, i, j) -- a counter-based
// hash, the splitmix64 finaliser -- so the device fills 8 GiB in place and the parity checker (which restates the
// same three lines) rebuilds any row on the host without an n x n copy.  Every step is exact in fp64, so host and device
// agree bit for bit: (h >> 11) is a 53-bit integer, * 2^-52 lands in [0, 2), - 1 in [-1, 1).
__host__ __device__ inline unsigned long long hash_mix64(unsigned long long z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// symmetric: (i, j) and (j, i) give the same value.  diag != 0: A(i,i) = diag (with symmetric and diag above the spectral
// radius ~ 2 sqrt(n/3) of the off-diagonal part the block is SPD, so that CG runs on it for hundreds of iterations).
__host__ __device__ inline double hash_entry(unsigned long long seed_mixed, int symmetric, double diag, long i, long j)
{
    if (i == j && diag != 0.0) return diag;
    const unsigned long long a = (symmetric && j < i) ? (unsigned long long)j : (unsigned long long)i;
    const unsigned long long b = (symmetric && j < i) ? (unsigned long long)i : (unsigned long long)j;
    const unsigned long long h = hash_mix64(seed_mixed ^ ((a << 32) | b));
    return (double)(h >> 11) * 0x1.0p-52 - 1.0;
}
// rows [row0, row0+rows) of the n x n hash matrix into a dense block at pitch lda; pad columns = 0.
hipError_t launch_fill_hash(double *A, long lda, int n, int row0, int rows, unsigned long long seed, int symmetric, double diag,
                            hipStream_t s);

// Matrix::read (matrix.cc:12-21) on the device, for the WHOLE entry list of the file in file order (0-based I, J;
// `sym`: entry z also assigns (J,I) right after (I,J), matrix.cc:18-20).  Entries outside rows [row0,row0+rows) are
// skipped.  A later assignment to the same element overrides an earlier one exactly as the sequential loop does:
// three passes -- claim (64-bit atomic max of the assignment's sequence number into the element's own storage),
// resolve (which assignment holds the element), write (the holder stores its value) -- no sort, no host-side map.
// The block must be zero-filled before.  dv == nullptr: dense block A (rows x lda); else banded storage.
struct DiaView;
hipError_t launch_coo_assign(double *A, long lda, const DiaView *dv, double *dia_vals, int n, int row0, int rows,
                             const int *I, const int *J, const double *a, long nz, int sym, unsigned char *win /* 2*nz */,
                             hipStream_t s);

// ---- banded storage (opt-in, NOT the reference's dense contract: SURVEY.md section 8f.3) ------------------
// The row block as its non-zero diagonals: vals[t*ld + i] = A(row0+i, row0+i+off[t]) for local row i, exactly 0
// where that column lies outside [0,n).  Offsets ascending, so a row is summed in ascending column order.
constexpr int kMaxDiags = 64;
struct DiaView {
    const double *vals;
    long ld;              // >= rows
    int ndiag;
    int off[kMaxDiags];
};
GemvPlan plan_dia(int rows, int variant = 0);   // variant 3, grid = min(ceil(rows/512), 2048); plan.light = LDS-window form
                                                // (cfg variant 30001 / 30002 force the direct / the window form)

// K1 on banded storage: same contract as launch_gemv_plain / launch_gemv_fused (same iteration head, same p_new
// = r + beta p_old, same one partial per workgroup), 8*(rows*ndiag) bytes of matrix instead of 8*rows*n.
hipError_t launch_spmv_dia_plain(const GemvPlan &plan, const DiaView &dv, int rows, int row0, int n, long lda,
                                 const double *v_full, double *Ap, double *partials, Scalars *sc, hipStream_t s);
hipError_t launch_spmv_dia_fused(const GemvPlan &plan, const DiaView &dv, int rows, int row0, int n, long lda,
                                 const double *p_old, double *p_new, SegView seg, double *Ap, double *partials,
                                 Scalars *sc, int k, double tol, hipStream_t s, hipEvent_t e_start = nullptr,
                                 hipEvent_t e_stop = nullptr);
// generate_lap2d_matrix (cg.cc:159-188) straight into banded storage; dv.off must be lap2d_offsets(size).
int lap2d_offsets(int size, int *off /* >= 5 */);
hipError_t launch_dia_generate_lap2d(double *vals, const DiaView &dv, int size, int row0, int rows, hipStream_t s);
// dense row block -> which diagonals hold a non-zero: flags[(j - i_global) + (n-1)] = 1
hipError_t launch_dia_mark(const double *A, long lda, int n, int row0, int rows, unsigned char *flags, hipStream_t s);
// dense row block -> banded storage for the offsets in dv
hipError_t launch_dia_pack(const double *A, long lda, int n, int row0, int rows, double *vals, const DiaView &dv,
                           hipStream_t s);
// ---- direct peer exchange (CGX_COMM_P2P): a lean all-gather over IPC-mapped mailboxes ---------------------
// Every rank owns one fine-grained mailbox; all ranks map all mailboxes.  Layout (identical on every rank):
//   flags : [kP2pChannels][kMaxRanks] words, one 128-B line each   (flag[c][q] = last epoch rank q delivered on channel c)
//   chunk flags : kMaxChunkFlags words of 8 B, word q*cpr + c = last epoch rank q delivered chunk c of its slice on
//                 channel 1 through the fused update kernel (k_update_xr_p2p)
//   data  : per channel c, [2 parities][nranks] slots of slot_bytes[c]
constexpr int kP2pChannels = 3;          // 1 = [Ap slice | p.Ap] segments, 2 = DEBUG scalars; 0 = with tagged words: the plain
                                         // all-gathers of segments (set-up / verification phases), else unused
constexpr int kP2pFlagStride = 128;      // bytes between flag words
constexpr int kMaxChunkFlags = 2048;     // nranks * chunks per rank <= this (262144 rows: 512 + nranks)
struct MailboxView {
    unsigned char *base[kMaxRanks];      // base[q] = rank q's mailbox as mapped in THIS process (base[rank] = own)
    long cflag_off;                      // byte offset of the chunk flag words
    long data_off[kP2pChannels];         // byte offset of channel c's data area
    long slot_bytes[kP2pChannels];       // bytes per (parity, rank) slot
    int nranks, rank;
    int tagged;                          // 1 = the fused update hands its bytes over as tagged words (no flags, no fences;
                                         // cgx_kernels.hip "Tagged words"); slots of channel 1 are then twice as large
    int acquire;                         // 1 = one system-scope acquire fence per workgroup behind the flag wait of
                                         // k_update_xr_p2p (default); 0 only for the A/B of its cost (tools/p2p_one_rank.py)
};

// All-gather `count` doubles per rank: rank's own contribution is src; afterwards dst + q*dst_stride holds
// rank q's for every q (own part copied from src only if copy_self).  If tail_n > 0, the fixed-order sum of
// src[tail_off .. tail_off+tail_n) travels as one extra double and lands at dst[q*dst_stride + sum_off].
// One workgroup per peer: push my data + release + flag into the peer's mailbox, wait (bounded) for the peer's
// flag in mine, copy its data out.
hipError_t launch_mailbox_allgather(const MailboxView &mv, int chan, unsigned long long epoch, const double *src,
                                    int count, int tail_off, int tail_n, double *dst, long dst_stride, int sum_off,
                                    int copy_self, long long timeout_ticks, int *err, hipStream_t s);

// K3 with the iteration's exchange inside (CGX_COMM_P2P): the (peer, chunk) pairs are dealt over the workgroups, each
// pushes [one chunk of the Ap slice | its p.Ap partial] into one peer's mailbox and raises that peer's chunk flag; every
// workgroup waits (bounded) for all nranks x cpr flags and reads the Ap element of its row and the partials straight from
// the mailbox.  The iteration is then K1 + this kernel.  ap_src: this rank's Ap slice as `split` column pieces of K1,
// `stride` apart (split == 1: the whole slice), added up in ascending order on the fly.  cpr = chunks_per_rank(apv.Sr).
hipError_t launch_update_xr_p2p(int n, int rows, int row0, const double *p_new, SegView apv, int cpr,
                                const MailboxView &mv, int chan, unsigned long long epoch, double *x, SegView rv, Scalars *sc,
                                int parity, long long timeout_ticks, int *err, hipStream_t s,
                                const double *ap_src, int split, long stride, hipEvent_t e_start = nullptr,
                                hipEvent_t e_stop = nullptr);
// The exchange of launch_update_xr_p2p alone (the same device code), on caller data: vals[i] = the Ap element read for
// global row i (n doubles), sums[wg] = the folded chunk partials as workgroup wg saw them (update_xr_grid(n) doubles).
hipError_t launch_chunk_exchange_selftest(int n, int rows, int row0, const double *p_like, SegView apv, int cpr,
                                          const MailboxView &mv, int chan, unsigned long long epoch, long long timeout_ticks,
                                          int *err, const double *ap_src, int split, long stride, double *vals, double *sums,
                                          hipStream_t s);
// Workgroups of k_update_xr_p2p the device keeps resident at once (occupancy x CUs): its grid must not exceed this, since
// its workgroups wait for each other inside the kernel.
hipError_t update_xr_p2p_resident_limit(int device, bool tagged, int *workgroups);

// ---- the LDS-resident solver for small dense problems (cgx_resident.hip) ----------------------------------------------
// n <= 2048 on one GPU: the row groups of A live in the CUs' LDS for the whole launch, x / r / p are replicated in every
// workgroup's registers, and the loop cg.cc:95-137 runs inside ONE persistent kernel whose workgroups exchange Ap as tagged
// words (no grid barrier).  All workgroups wait for each other: the grid never exceeds the number of CUs.
struct ResidentPlan {
    int R;             // rows per workgroup: the power of two with R x min(CUs, 256) >= n (<= 8)
    int S;             // column steps of 512
    int rows_per_wg;   // = R
    int grid;          // workgroups = ceil(n / R) <= CUs
    int xslots;        // tagged doubles per parity of the exchange buffer (512 * S)
    size_t lds_bytes;  // dynamic LDS of one workgroup
    int hybrid;        // 1 = 2048 < n <= 4096: R = 16 rows per workgroup, of which RL in LDS, RG in registers and
    int RL, RG;        //     R - RL - RG streamed from memory every iteration; 0: all R rows in LDS (RL = R, RG = 0)
    int stream;        // 1 = cgx_stream.hip (4096 < n <= 16384): workgroups of 512 threads, S = column steps of 1024, every row
    int RB;            //     streamed through a ring of RB rows; of the R rows per workgroup the first RL stay in LDS and the next RG in
                       //     registers, the other R - RL - RG (a multiple of RB) are streamed every iteration
    int l2_rows;       //     ... the first l2_rows of them with the default cache policy (they stay in the XCD's L2), the others nt
};
// One block of solver state (x | r + the update kernel's r.r partials | p | Scalars), offsets in doubles from its base.  On one
// GPU the shard keeps TWO such blocks (cgx_context.cpp): the persistent kernels read one and write the other.
constexpr long kStateTail = 1024;   // = kMaxVectorGrid: room for one r.r partial per workgroup of the update kernel behind r
__host__ __device__ inline long state_off_r(long lda) { return lda; }
__host__ __device__ inline long state_off_p(long lda) { return 2 * lda + kStateTail; }
__host__ __device__ inline long state_off_sc(long lda) { return 3 * lda + kStateTail; }
__host__ __device__ inline long state_doubles(long lda) { return 3 * lda + kStateTail + 16; }   // Scalars: 96 bytes
// What a persistent launch reports to the host, written by the kernel straight into pinned host memory (so that a solve needs no
// copy command for it): workgroup 0 writes the head and, LAST, the launch's stamp -- it does so on every path out of the kernel,
// also when a wait expired or the error word was already up when it started, and since a kernel is over only when all of its
// workgroups are, the host finds the stamp of THIS launch behind the stream's synchronisation whatever happened.  Every
// workgroup writes its own line of `waits` (all of them fresh after any launch that has ended).
struct ResidentTail {
    int done, k_final;     // the break of cg.cc:120-121 was taken, in iteration k_final
    int err;               // a bounded wait expired (or the error word was up at the start)
    unsigned stamp;        // ResidentArgs::stamp of the launch that wrote this
    // workgroup 0: iterations run | polls of the watched word that had to be repeated | gather rounds that had to be repeated
    long long iterations, watch_repeats, gather_repeats;
    // per workgroup, 100-MHz ticks: from its publish of the launch's FIRST iteration until it had gathered all of Ap (a workgroup
    // that was placed late shows here) | the longest such span of a LATER iteration in which a poll had to be repeated (0: none)
    unsigned waits[256][2];
};
struct ResidentArgs {
    const double *A;   // n x lda, row-major, pad columns zero
    long lda;
    int n, rows_per_wg, xslots;
    const double *in;  // the state the launch STARTS from, one block laid out by state_off_*: x | r | p | Scalars (p is read only
                       // when k0 > 0; rs[] in the per-launch path's convention) -- never written: a launch whose waits expire
                       // leaves all of it intact and the host redoes it on the per-launch path
    double *out;       // the state the launch ENDS with, a second block of the same layout (the host makes it the current one
                       // after a launch that came back without the error word raised): x, r, p, rs[], done, k_final
    unsigned long long *xbuf;   // device memory, 2 parities x xslots x 2 words, zero-filled when the problem is set
    unsigned long long epoch0;  // the launch uses epochs epoch0 + 1 ... epoch0 + iters
    int k0, iters;     // iterations k0 ... k0 + iters - 1 (stops at the break of cg.cc:120-121)
    double tol;
    long long timeout_ticks;    // bound of every wait, 100 MHz wall-clock ticks
    int *err;          // device word raised when a wait expired
    int l2_rows;       // streaming kernel: the first l2_rows streamed rows of every workgroup are read with the default cache policy (the others nt)
    int stagger;       // streaming kernel: every workgroup begins its sweep at a batch of its own (0 = all at their first rows)
    int mute_wg;       // test only (cgx_probe_resident_test): this workgroup leaves out the publish of the launch's first iteration; -1 = none
    ResidentTail *tail;   // PINNED HOST memory: what the launch reports back (written by the kernel itself: no copy command)
    unsigned stamp;    // the launch's number (never 0): workgroup 0's last word into the tail
    long long *prof;   // diagnostics (CGX_RESIDENT_PROFILE=1), else nullptr: workgroup 0 adds up shader-clock cycles per phase
                       // [0] GEMV + row sums + publish, [1] wait for the watched word, [2] gather, [3] p.Ap, [4] update + r.r,
                       // [5] watch rounds, [6] gather rounds, [7] iterations
};
// false: this problem does not fit (n > 16384, too few CUs, LDS per workgroup too small).  n <= 4096: the matrix stays on
// the chip (cgx_resident.hip); above: every row is streamed (cgx_stream.hip, plan_stream).
bool plan_resident(int n, int cus, size_t lds_per_wg, ResidentPlan *out);
// The streaming persistent kernel alone (1024 <= n <= 16384; tests force it below 4096 too): plan, and prepare (a == nullptr:
// raise the dynamic-LDS limit, *per_cu = workgroups the runtime keeps resident per CU) / launch.
bool plan_stream(int n, int cus, size_t lds_per_wg, ResidentPlan *out);
hipError_t stream_dispatch(const ResidentPlan &pl, const ResidentArgs *a, hipStream_t s, int *per_cu);
// Once per plan, before the first launch: raises the kernel's dynamic-LDS limit; *workgroups_per_cu = what the runtime
// keeps resident per CU (the caller checks grid <= that x CUs: the workgroups wait for each other).
hipError_t prepare_cg_resident(const ResidentPlan &pl, int *workgroups_per_cu);
hipError_t launch_cg_resident(const ResidentPlan &pl, const ResidentArgs &a, hipStream_t s);

// ---- a persistent solve from a ZERO initial guess in four launches (n <= 16384, one GPU; cgx_solve.cpp) --------------------------
// Everything cgx_solve_begin does for x0 = 0 in ONE kernel: x = 0, r = b (b - A 0, cg.cc:79-82: A 0 is exactly 0), one r.r partial
// per 256 rows behind r (what the per-launch K1 of iteration 0 folds, should the persistent launch have to be redone), both p
// buffers, the exchanged segments and the scalar block zeroed, the error word down.
hipError_t launch_solve_begin_zero(int n, long lda, const double *b_full, double *x, SegView rv, double *p0, double *p1, double *apg,
                                   long apg_count, Scalars *sc, int *err, hipStream_t s);
// Everything cgx_solve_end does behind the verification GEMV in ONE kernel (one workgroup of 1024 threads, fixed order):
// out[0 .. n) = x, out[n + 0 .. 2] = sum (Ax - b)^2, sum b^2, sum x^2 (cg.cc:144-151), out[n + 3 .. 4] = sc->rs[0 .. 1]; out is
// pinned host memory.
hipError_t launch_solve_end(int n, const double *Ax, const double *b, const double *x, const Scalars *sc, double *out, hipStream_t s);

// Loopback "collective": copy local[kSlots] of every shard into gathered[] of every shard (<= 16 shards).
hipError_t launch_loopback_gather(double *const *gathered_ptrs, const Scalars *const *scalar_ptrs, int nshards,
                                  hipStream_t s);

}  // namespace cgx
