// cgx_matrix.cpp -- problem definition behind include/cgx.h: generate_lap2d_matrix (cg.cc:159-188), a caller's
// dense matrix, the Matrix-Market reader (matrix_coo.cc:7-60 + matrix.cc:6-22), the source term (cg.cc:218-234),
// for dense and for the opt-in banded storage.
#include "cgx_internal.h"

#include <sys/mman.h>
#include <sys/stat.h>

#include <algorithm>
#include <cctype>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

using namespace cgxi;

namespace cgxi {

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

}  // namespace cgxi

extern "C" {

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

// ---- MatrixCOO::read, matrix_coo.cc:7-60: the file as a list of entries, host only ---------------------
}  // extern "C"

namespace cgxi {

namespace {

inline bool is_ws(unsigned char c) { return c == ' ' || c == '\n' || c == '\t' || c == '\r' || c == '\v' || c == '\f'; }   // isspace, "C" locale

// The bytes behind the header, mapped (regular file) or read (anything else).
struct Body {
    const char *p = nullptr;
    size_t len = 0;
    void *map = nullptr;
    size_t map_len = 0;
    std::vector<char> owned;
    ~Body()
    {
        if (map) munmap(map, map_len);
    }
};

// Entries "%d %d %lg" (matrix_coo.cc:48) = a stream of white-space separated tokens, three per entry.  The body is cut into
// one range per thread; a token belongs to the range it STARTS in.  Pass 1 counts the tokens of every range, the prefix sum
// tells every range the index of its first token -- and with it which field of which entry that is -- and pass 2 parses
// every token straight into its slot (entry = index / 3, field = index % 3).  No carry between ranges, no order between
// threads, the result is the file order.  Tokens are copied into a small terminated buffer before strtol / strtod, so the
// mapping needs no terminator.  Returns "" or the message of the FIRST bad entry of the file.
std::string parse_entries(const char *data, size_t len, long nz, int m, int n, int *I, int *J, double *a, int nthreads)
{
    if (nz <= 0) return "";
    // at least 1 MiB per thread -- unless the caller forces a count (negative: the parser's own tests cut tiny files into
    // many ranges so that range boundaries fall inside tokens and inside entries)
    nthreads = nthreads < 0 ? std::min(-nthreads, (int)std::max<size_t>(len, 1)) : std::max(1, std::min(nthreads, (int)(len / (1 << 20)) + 1));
    std::vector<size_t> start((size_t)nthreads + 1);
    for (int t = 0; t <= nthreads; ++t) start[(size_t)t] = len * (size_t)t / (size_t)nthreads;
    auto first_token = [&](size_t pos) {   // first position >= pos where a token starts
        if (pos > 0 && pos < len && !is_ws((unsigned char)data[pos - 1]))
            while (pos < len && !is_ws((unsigned char)data[pos])) ++pos;          // inside a token of the previous range
        while (pos < len && is_ws((unsigned char)data[pos])) ++pos;
        return pos;
    };
    std::vector<long> count((size_t)nthreads, 0);
    auto in_threads = [&](const std::function<void(int)> &body) {
        std::vector<std::thread> th;
        int started = 1;                                                          // range 0 runs on the calling thread
        try {
            for (; started < nthreads; ++started) th.emplace_back(body, started);
        } catch (...) {                                                           // no more threads to be had: the rest runs here
        }
        body(0);
        for (int t = started; t < nthreads; ++t) body(t);
        for (auto &x : th) x.join();
    };
    in_threads([&](int t) {
        size_t pos = first_token(start[(size_t)t]);
        const size_t end = start[(size_t)t + 1];
        long c = 0;
        while (pos < end) {                                                       // pos is the start of a token
            ++c;
            while (pos < len && !is_ws((unsigned char)data[pos])) ++pos;
            while (pos < len && is_ws((unsigned char)data[pos])) ++pos;
        }
        count[(size_t)t] = c;
    });
    std::vector<long> tok0((size_t)nthreads + 1, 0);
    for (int t = 0; t < nthreads; ++t) tok0[(size_t)t + 1] = tok0[(size_t)t] + count[(size_t)t];
    const long want = 3 * nz;
    std::vector<long> bad((size_t)nthreads, -1);                                  // first bad ENTRY per range, -1 = none
    std::vector<int> kind((size_t)nthreads, 0);                                   // 1 = unreadable, 2 = index out of range
    in_threads([&](int t) {
        size_t pos = first_token(start[(size_t)t]);
        const size_t end = start[(size_t)t + 1];
        long k = tok0[(size_t)t];
        char tmp[128];
        while (pos < end && k < want) {
            const size_t b = pos;
            while (pos < len && !is_ws((unsigned char)data[pos])) ++pos;
            const size_t tl = pos - b;
            const long e = k / 3;
            const int field = (int)(k - 3 * e);
            bool ok = tl < sizeof tmp;
            if (ok) {
                memcpy(tmp, data + b, tl);
                tmp[tl] = '\0';
                char *endp = tmp;
                if (field < 2) {
                    const long v = strtol(tmp, &endp, 10);
                    ok = endp == tmp + tl;
                    if (ok && (v < 1 || v > (field == 0 ? (long)m : (long)n))) {  // 1-based in the file, matrix_coo.cc:49-50
                        bad[(size_t)t] = e;
                        kind[(size_t)t] = 2;
                        return;
                    }
                    if (ok) (field == 0 ? I : J)[e] = (int)v - 1;
                } else {
                    const double v = strtod(tmp, &endp);
                    ok = endp == tmp + tl;
                    if (ok) a[e] = v;
                }
            }
            if (!ok) {
                bad[(size_t)t] = e;
                kind[(size_t)t] = 1;
                return;
            }
            ++k;
            while (pos < len && is_ws((unsigned char)data[pos])) ++pos;
        }
    });
    long first_bad = -1;
    int first_kind = 0;
    for (int t = 0; t < nthreads; ++t)
        if (bad[(size_t)t] >= 0 && (first_bad < 0 || bad[(size_t)t] < first_bad)) {
            first_bad = bad[(size_t)t];
            first_kind = kind[(size_t)t];
        }
    if (first_bad < 0 && tok0[(size_t)nthreads] < want) {                         // the file ends inside entry tokens / 3
        first_bad = tok0[(size_t)nthreads] / 3;
        first_kind = 1;
    }
    if (first_bad < 0) return "";
    return first_kind == 2 ? "Matrix Market index out of range" : "Matrix Market entry " + std::to_string(first_bad) + " unreadable";
}

}  // namespace

// Header (mmio.c:96-179,189-217 as matrix_coo.cc:19-40 uses them) + entries.  err: CGX_ERR_IO / CGX_ERR_UNSUPPORTED text.
cgx_status parse_matrix_market(const char *path, MtxEntries *out, std::string *err, int nthreads, bool header_only)
{
    FILE *f = fopen(path, "r");
    if (!f) { *err = std::string("Could not open matrix: ") + path; return CGX_ERR_IO; }   // matrix_coo.cc:14-17
    struct Closer {
        FILE *f;
        ~Closer() { fclose(f); }
    } closer{f};

    char line[2048];
    if (!fgets(line, sizeof line, f)) { *err = "Could not process Matrix Market banner."; return CGX_ERR_IO; }
    char tok[5][64] = {{0}};
    if (sscanf(line, "%63s %63s %63s %63s %63s", tok[0], tok[1], tok[2], tok[3], tok[4]) != 5 ||
        strcmp(tok[0], "%%MatrixMarket") != 0) {
        *err = "Could not process Matrix Market banner.";   // matrix_coo.cc:19-22
        return CGX_ERR_IO;
    }
    for (int t = 1; t < 5; ++t)
        for (char *c = tok[t]; *c; ++c) *c = (char)tolower((unsigned char)*c);       // mmio.c lower-cases the tokens
    if (strcmp(tok[1], "matrix") != 0 || strcmp(tok[2], "coordinate") != 0) {
        *err = std::string("Sorry, this application does not support Market Market type: [") + tok[1] + " " + tok[2] + " " +
               tok[3] + " " + tok[4] + "]";   // matrix_coo.cc:25-29
        return CGX_ERR_UNSUPPORTED;
    }
    // The reference parses every entry as "%d %d %lg" whatever the field (matrix_coo.cc:48); fields without
    // one real value per entry would be silently misread there and are rejected here.
    if (strcmp(tok[3], "real") != 0 && strcmp(tok[3], "integer") != 0 && strcmp(tok[3], "double") != 0) {
        *err = std::string("Matrix Market field not supported: ") + tok[3];
        return CGX_ERR_UNSUPPORTED;
    }
    out->sym = strcmp(tok[4], "symmetric") == 0;                                     // matrix_coo.cc:43
    if (!out->sym && strcmp(tok[4], "general") != 0) {
        *err = std::string("Matrix Market symmetry not supported: ") + tok[4];
        return CGX_ERR_UNSUPPORTED;
    }
    for (;;) {   // size line after the % comments, mmio.c:198-206
        if (!fgets(line, sizeof line, f)) { *err = "Matrix Market size line missing"; return CGX_ERR_IO; }
        if (line[0] == '%') continue;
        long long lm = 0, ln = 0, lnz = 0;
        if (sscanf(line, "%lld %lld %lld", &lm, &ln, &lnz) == 3) {   // the reference reads three ints (mmio.c:204)
            if (lm > 0x7fffffffLL || ln > 0x7fffffffLL || lnz > 0x7fffffffLL || lm < 0 || ln < 0 || lnz < 0) {
                *err = "Matrix Market size line does not fit the reference's int sizes";
                return CGX_ERR_UNSUPPORTED;
            }
            out->m = (int)lm;
            out->n = (int)ln;
            out->nz = (int)lnz;
            break;
        }
    }
    if (out->m <= 0 || out->n <= 0 || out->nz < 0 || out->m != out->n) {
        *err = "CG needs a square matrix with positive size";
        return CGX_ERR_UNSUPPORTED;
    }
    if (header_only) return CGX_OK;
    // the entries: the rest of the file, mapped if it is a regular file
    Body body;
    const long at = ftell(f);
    struct stat st {};
    if (at >= 0 && fstat(fileno(f), &st) == 0 && S_ISREG(st.st_mode) && st.st_size > at) {
        void *mp = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fileno(f), 0);
        if (mp != MAP_FAILED) {
            body.map = mp;
            body.map_len = (size_t)st.st_size;
            body.p = static_cast<const char *>(mp) + at;
            body.len = (size_t)st.st_size - (size_t)at;
        }
    }
    if (!body.map) {                                  // a pipe, or mmap refused: read what is left
        char buf[1 << 16];
        size_t got;
        while ((got = fread(buf, 1, sizeof buf, f)) > 0) body.owned.insert(body.owned.end(), buf, buf + got);
        body.p = body.owned.data();
        body.len = body.owned.size();
    }
    out->I.assign((size_t)out->nz, 0);
    out->J.assign((size_t)out->nz, 0);
    out->a.assign((size_t)out->nz, 0.0);
    *err = parse_entries(body.p, body.len, out->nz, out->m, out->n, out->I.data(), out->J.data(), out->a.data(), nthreads);
    return err->empty() ? CGX_OK : CGX_ERR_IO;
}

int default_parse_threads()
{
    if (const char *e = getenv("CGX_MTX_THREADS")) return std::max(1, atoi(e));
    const unsigned hc = std::thread::hardware_concurrency();
    return (int)std::max(1u, std::min(hc ? hc : 1u, 16u));
}

}  // namespace cgxi

extern "C" {

// ---- MatrixCOO::read + Matrix::read, matrix_coo.cc:7-60 and matrix.cc:6-22 -------------------------
cgx_status cgx_read_matrix(cgx_ctx *ctx, const char *path)
{
    if (!ctx || !path) return fail(ctx, CGX_ERR_BAD_ARG, "cgx_read_matrix: bad argument");
    // The entries are parsed on the host threads from the mapped file (parse_matrix_market) and kept in file order;
    // everything after that -- which row block an entry belongs to, the mirrored assignment of a symmetric file, and "a
    // later entry for the same (i,j) overrides an earlier one" (the sequential loop of matrix.cc:12-21) -- is done on the
    // device.
    MtxEntries mtx;
    {
        std::string err;
        const cgx_status st = parse_matrix_market(path, &mtx, &err, default_parse_threads());
        if (st != CGX_OK) return fail(ctx, st, err);
    }
    const int n = mtx.n;
    const bool is_sym = mtx.sym;
    CGX_TRY(setup_problem(ctx, n));
    std::vector<int> &hI = mtx.I, &hJ = mtx.J;
    std::vector<double> &ha = mtx.a;
    std::vector<int> offs;   // banded: distinct (column - row) of all assignments, at most CGX_MAX_DIAGONALS + 1 kept
    if (ctx->banded)
        for (size_t z = 0; z < ha.size() && (int)offs.size() <= CGX_MAX_DIAGONALS; ++z)
            for (int mir = 0; mir <= (is_sym ? 1 : 0); ++mir) {                       // matrix.cc:17, 18-20
                const int off = mir ? hI[z] - hJ[z] : hJ[z] - hI[z];
                auto it = std::lower_bound(offs.begin(), offs.end(), off);
                if (it == offs.end() || *it != off) offs.insert(it, off);
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

}  // extern "C"
