// symmetric SLIM-BPR: weighted critical path.  A step costs c0 + c1 * 2 * (ceil(L / B) - 1): the blocks of a long profile are
// fetched one after the other, twice.  Dependencies: cells of both rows, optimiser cells of the two items.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
int main(int argc, char **argv) {
    FILE *f = fopen(argv[1], "rb");
    int n_users, n_items, n_steps; long long nnz;
    fread(&n_users, 4, 1, f); fread(&n_items, 4, 1, f); fread(&nnz, 8, 1, f); fread(&n_steps, 4, 1, f);
    int *indptr = malloc(4 * (n_users + 1)), *indices = malloc(4 * nnz), *su = malloc(4 * n_steps), *si = malloc(4 * n_steps), *sj = malloc(4 * n_steps);
    fread(indptr, 4, n_users + 1, f); fread(indices, 4, nnz, f); fread(su, 4, n_steps, f); fread(si, 4, n_steps, f); fread(sj, 4, n_steps, f);
    size_t ncell = (size_t)n_items * (n_items + 1) / 2;
    float *celld = malloc(ncell * 4);
    float *itemd = malloc(n_items * 4);
    double cfg[][3] = {{1, 0, 256}, {2.0, 0, 256}, {2.0, 1.3, 256}, {2.0, 0.65, 256}, {2.0, 1.3, 1024}, {2.0, 0.2, 256}};
    for (int c = 0; c < 6; ++c) {
        double c0 = cfg[c][0], c1 = cfg[c][1]; int B = (int)cfg[c][2];
        memset(celld, 0, ncell * 4); memset(itemd, 0, n_items * 4);
        float mx = 0;
        for (int t = 0; t < n_steps; ++t) {
            int u = su[t], i = si[t], j = sj[t]; float d = itemd[i] > itemd[j] ? itemd[i] : itemd[j];
            int L = indptr[u + 1] - indptr[u];
            for (int q = indptr[u]; q < indptr[u + 1]; ++q) {
                int s = indices[q];
                if (s != i) { size_t r = i > s ? i : s, cc = i > s ? s : i; size_t at = r * (r + 1) / 2 + cc; if (celld[at] > d) d = celld[at]; }
                if (s != j) { size_t r = j > s ? j : s, cc = j > s ? s : j; size_t at = r * (r + 1) / 2 + cc; if (celld[at] > d) d = celld[at]; }
            }
            int nb = (L + B - 1) / B;
            d += (float)(c0 + c1 * 2 * (nb - 1));
            if (d > mx) mx = d;
            itemd[i] = d; itemd[j] = d;
            for (int q = indptr[u]; q < indptr[u + 1]; ++q) {
                int s = indices[q];
                if (s != i) { size_t r = i > s ? i : s, cc = i > s ? s : i; celld[r * (r + 1) / 2 + cc] = d; }
                if (s != j) { size_t r = j > s ? j : s, cc = j > s ? s : j; celld[r * (r + 1) / 2 + cc] = d; }
            }
        }
        printf("step = %.2f + %.2f * 2 * (ceil(L/%d) - 1): weighted critical path %.1f\n", c0, c1, B, mx);
    }
    return 0;
}
