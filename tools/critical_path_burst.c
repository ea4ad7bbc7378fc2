// Round 5 diagnostic (host only): gcc -O2 -o /tmp/cp2 tools/critical_path_burst.c; /tmp/cp2 <file as for critical_path.c> cib cio cub cuo T...
// two-class latency model: an item "bursts" while it has >= T ratings ahead; burst: item edge cib, user edge cub; ordinary: cio, cuo
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
int main(int argc, char **argv) {
    FILE *f = fopen(argv[1], "rb"); int64_t nnz; int nu, ni;
    if (fread(&nnz, 8, 1, f) != 1 || fread(&nu, 4, 1, f) != 1 || fread(&ni, 4, 1, f) != 1) return 1;
    int32_t *idx = malloc(nnz * 8); if (fread(idx, 8, nnz, f) != (size_t)nnz) return 1; fclose(f);
    double cib = atof(argv[2]), cio = atof(argv[3]), cub = atof(argv[4]), cuo = atof(argv[5]);
    int *deg = calloc(ni, 4); for (int64_t r = 0; r < nnz; ++r) deg[idx[2 * r + 1]]++;
    for (int a = 6; a < argc; ++a) {
        int T = atoi(argv[a]);
        double *fu = calloc(nu, 8), *fi = calloc(ni, 8); int *pos = calloc(ni, 4); double best = 0;
        for (int64_t r = 0; r < nnz; ++r) {
            int u = idx[2 * r], i = idx[2 * r + 1];
            int burst = deg[i] - pos[i] >= T; pos[i]++;
            double t = fi[i] + (burst ? cib : cio), t2 = fu[u] + (burst ? cub : cuo); if (t2 > t) t = t2;
            fu[u] = t; fi[i] = t; if (t > best) best = t;
        }
        printf("T %7d -> %.1f ms\n", T, best * 1e-3);
        free(fu); free(fi); free(pos);
    }
    return 0;
}
