// conv.cpp (host mirror) -- ≙ data/SuiteSparse/mtx2csr.cc:248-268 as built by prepare_mtx_data.sh:13-20
// (`g++ mtx2csr.cc -o conv; conv X.mtx X.csv`): MatrixMarket in, the three-line CSV the DataLoader reads out.
// --sort additionally orders every row's columns (the reference keeps file order; its tilers expect sorted rows).
#include <cstdio>
#include <cstring>

#include "../../../include/flex_spmm.h"

int main(int argc, char **argv) {
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s <in.mtx> <out.csv> [--sort]\n", argv[0]);
        return 2;
    }
    const int sort_columns = argc > 3 && !std::strcmp(argv[3], "--sort");
    std::printf("MAT: %s\n", argv[1]);
    flex_host_csr a{};
    int rc = flex_mtx_load(argv[1], sort_columns, &a);
    if (rc) {
        std::fprintf(stderr, "conv: %s: %s\n", argv[1], flex_strerror(rc));
        return 1;
    }
    std::printf("m = %d,   n = %d\n", a.m, a.n);
    const flex_csr v{a.m, a.n, a.nnz, a.rowPtr, a.col, a.vals};
    rc = flex_csv_save(argv[2], &v);
    flex_host_csr_free(&a);
    if (rc) {
        std::fprintf(stderr, "conv: %s: %s\n", argv[2], flex_strerror(rc));
        return 1;
    }
    return 0;
}
