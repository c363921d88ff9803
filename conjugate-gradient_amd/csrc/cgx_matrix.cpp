// cgx_matrix.cpp -- problem definition behind include/cgx.h: generate_lap2d_matrix (cg.cc:159-188), a caller's
// dense matrix, the Matrix-Market reader (matrix_coo.cc:7-60 + matrix.cc:6-22), the source term (cg.cc:218-234),
// for dense and for the opt-in banded storage.
#include "cgx_internal.h"

#include <algorithm>
#include <cctype>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
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
        long long lm = 0, ln = 0, lnz = 0;
        if (sscanf(line, "%lld %lld %lld", &lm, &ln, &lnz) == 3) {   // the reference reads three ints (mmio.c:204)
            if (lm > 0x7fffffffLL || ln > 0x7fffffffLL || lnz > 0x7fffffffLL || lm < 0 || ln < 0 || lnz < 0)
                return fail(ctx, CGX_ERR_UNSUPPORTED, "Matrix Market size line does not fit the reference's int sizes");
            m = (int)lm;
            n = (int)ln;
            nz = (int)lnz;
            break;
        }
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
                    if (v < 1 || v > 0x7fffffffL) return fail(ctx, CGX_ERR_IO, "Matrix Market index out of range");
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

}  // extern "C"
