// critical-path depth of a SLIM-BPR epoch stream: dense (row-level) and symmetric (cell-level) dependencies
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
int main(int argc, char **argv) {
    FILE *f = fopen(argv[1], "rb");
    int n_users, n_items, n_steps; long long nnz;
    fread(&n_users, 4, 1, f); fread(&n_items, 4, 1, f); fread(&nnz, 8, 1, f); fread(&n_steps, 4, 1, f);
    int *indptr = malloc(4 * (n_users + 1)), *indices = malloc(4 * nnz), *su = malloc(4 * n_steps), *si = malloc(4 * n_steps), *sj = malloc(4 * n_steps);
    fread(indptr, 4, n_users + 1, f); fread(indices, 4, nnz, f); fread(su, 4, n_steps, f); fread(si, 4, n_steps, f); fread(sj, 4, n_steps, f);
    // dense
    int *rowdepth = calloc(n_items, 4); int maxd = 0;
    int *cnt = calloc(n_items, 4);
    for (int t = 0; t < n_steps; ++t) {
        int d = rowdepth[si[t]] > rowdepth[sj[t]] ? rowdepth[si[t]] : rowdepth[sj[t]];
        d += 1; rowdepth[si[t]] = d; rowdepth[sj[t]] = d; if (d > maxd) maxd = d;
        cnt[si[t]]++; cnt[sj[t]]++;
    }
    int maxc = 0; for (int i = 0; i < n_items; ++i) if (cnt[i] > maxc) maxc = cnt[i];
    printf("dense: critical path %d links; busiest row %d steps\n", maxd, maxc);
    // with the H busiest rows' links costing `ch` and the others `cc` (weighted depth, microseconds)
    for (int H = 0; H <= 256; H = H ? H * 2 : 16) {
        // threshold count of the H-th busiest
        int *sorted = malloc(4 * n_items); memcpy(sorted, cnt, 4 * n_items);
        int cmp(const void *a, const void *b) { return *(const int *)b - *(const int *)a; }
        qsort(sorted, n_items, 4, cmp);
        int thr = H ? sorted[H - 1] : 1 << 30;
        free(sorted);
        for (double ch = 0.25; ch <= 0.51; ch += 0.25) {
            double cc = 1.5;
            double *rd = calloc(n_items, 8); double mx = 0;
            for (int t = 0; t < n_steps; ++t) {
                int i = si[t], j = sj[t];
                double d = rd[i] > rd[j] ? rd[i] : rd[j];
                int hot = (cnt[i] >= thr) || (cnt[j] >= thr);
                d += hot ? ch : cc; rd[i] = d; rd[j] = d; if (d > mx) mx = d;
            }
            printf("  H=%3d (count >= %d) hot link %.2f us, cold link %.2f us: weighted critical path %.3f ms\n", H, thr, ch, cc, mx * 1e-3);
            free(rd);
        }
    }
    // symmetric: per cell last-writer depth (packed triangle)
    size_t ncell = (size_t)n_items * (n_items + 1) / 2;
    int *celld = calloc(ncell, 4);
    int maxs = 0; long long recent = 0, total = 0; int *celllast = malloc(ncell * 4); memset(celllast, 0xff, ncell * 4);
    long long hist[8] = {0};
    for (int t = 0; t < n_steps; ++t) {
        int u = su[t], i = si[t], j = sj[t], d = 0;
        for (int q = indptr[u]; q < indptr[u + 1]; ++q) {
            int s = indices[q];
            if (s != i) { size_t r = i > s ? i : s, c = i > s ? s : i; size_t at = r * (r + 1) / 2 + c; if (celld[at] > d) d = celld[at];
                total++; int p = celllast[at]; int age = p < 0 ? 1 << 30 : t - p; hist[age < 256 ? 0 : age < 1024 ? 1 : age < 4096 ? 2 : age < 16384 ? 3 : p < 0 ? 5 : 4]++; }
            if (s != j) { size_t r = j > s ? j : s, c = j > s ? s : j; size_t at = r * (r + 1) / 2 + c; if (celld[at] > d) d = celld[at];
                total++; int p = celllast[at]; int age = p < 0 ? 1 << 30 : t - p; hist[age < 256 ? 0 : age < 1024 ? 1 : age < 4096 ? 2 : age < 16384 ? 3 : p < 0 ? 5 : 4]++; }
        }
        d += 1; if (d > maxs) maxs = d;
        for (int q = indptr[u]; q < indptr[u + 1]; ++q) {
            int s = indices[q];
            if (s != i) { size_t r = i > s ? i : s, c = i > s ? s : i; size_t at = r * (r + 1) / 2 + c; celld[at] = d; celllast[at] = t; }
            if (s != j) { size_t r = j > s ? j : s, c = j > s ? s : j; size_t at = r * (r + 1) / 2 + c; celld[at] = d; celllast[at] = t; }
        }
    }
    printf("symmetric: critical path %d links; %lld cell touches; pred age <256: %lld, <1024: %lld, <4096: %lld, <16384: %lld, older: %lld, none: %lld\n",
           maxs, total, hist[0], hist[1], hist[2], hist[3], hist[4], hist[5]);
    return 0;
}
