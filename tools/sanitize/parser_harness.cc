#include "cgx.h"
#include <cstdio>
#include <vector>
int main(int argc, char **argv)
{
    for (int a = 1; a < argc; ++a)
        for (int threads : {1, -2, -5, -33, 0}) {
            int m = 0, n = 0, nz = 0, sym = 0;
            char err[256] = {0};
            cgx_status st = cgx_probe_parse_matrix_market(argv[a], threads, &m, &n, &nz, &sym, nullptr, nullptr, nullptr, 0, err, 256);
            if (st) { printf("%s threads %d: status %d %s\n", argv[a], threads, st, err); continue; }
            std::vector<int> I(nz), J(nz);
            std::vector<double> v(nz);
            st = cgx_probe_parse_matrix_market(argv[a], threads, &m, &n, &nz, &sym, I.data(), J.data(), v.data(), nz, err, 256);
            printf("%s threads %d: status %d n=%d nz=%d %s\n", argv[a], threads, st, n, nz, err);
        }
    return 0;
}
