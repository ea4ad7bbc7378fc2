// Round 5 diagnostic (host only): gcc -O2 -o /tmp/cp tools/critical_path.c; /tmp/cp <file: int64 nnz, int32 nu, int32 ni, int32[nnz][2]> c_item c_user...
// critical path of the dependency DAG (user order + item order) with per-edge latencies: finish[r] = max(fin_item_prev + c_item, fin_user_prev + c_user)
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
int main(int argc, char **argv) {
    FILE *f = fopen(argv[1], "rb"); int64_t nnz; int nu, ni;
    fread(&nnz, 8, 1, f); fread(&nu, 4, 1, f); fread(&ni, 4, 1, f);
    int32_t *idx = malloc(nnz * 8); fread(idx, 8, nnz, f); fclose(f);
    double ci = atof(argv[2]);
    for (int a = 3; a < argc; ++a) {
        double cu = atof(argv[a]);
        double *fu = calloc(nu, 8), *fi = calloc(ni, 8); double best = 0;
        for (int64_t r = 0; r < nnz; ++r) {
            int u = idx[2 * r], i = idx[2 * r + 1];
            double t = fi[i] + ci; double t2 = fu[u] + cu; if (t2 > t) t = t2;
            fu[u] = t; fi[i] = t; if (t > best) best = t;
        }
        printf("c_item %.2f us  c_user %.2f us  ->  critical path %.2f ms\n", ci, cu, best * 1e-3);
        free(fu); free(fi);
    }
    return 0;
}
