// apbs_census.c — how large are All-Pair's backward searches?  Frontier-synchronous backward push
// (Backward_Search.java:38-100 as the engine schedules it) from a sample of targets; prints, per
// bucket of touched-node count, the number of searches, their pops, edge pushes and levels.  Answers
// "which table size serves which share of the work" for kernels_apbs.hip's tiers.
//   gcc -O2 -fopenmp -o /tmp/apbs_census tools/exp/apbs_census.c -lm
//   /tmp/apbs_census <csr.bin> <first> <count> <stride> <rmax>
// csr.bin: uint32 n, uint64 m, out_deg[n] (uint32), in_rp[n+1] (uint32), in_ci[m] (int32)
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(int argc, char** argv) {
  if (argc < 6) return 1;
  FILE* f = fopen(argv[1], "rb");
  uint32_t n; uint64_t m;
  if (!f || fread(&n, 4, 1, f) != 1 || fread(&m, 8, 1, f) != 1) return 2;
  uint32_t* odeg = malloc(4ull * n); uint32_t* irp = malloc(4ull * (n + 1)); int32_t* ici = malloc(4ull * m);
  if (fread(odeg, 4, n, f) != n || fread(irp, 4, n + 1, f) != n + 1 || fread(ici, 4, m, f) != m) return 3;
  fclose(f);
  const uint32_t first = atoi(argv[2]), count = atoi(argv[3]), stride = atoi(argv[4]);
  const double rmax = atof(argv[5]), alpha = 0.15;
  enum { NB = 24 };
  uint64_t b_cnt[NB] = {0}, b_pops[NB] = {0}, b_edges[NB] = {0}, b_lev[NB] = {0}, b_ent[NB] = {0}, b_maxf[NB] = {0};
#pragma omp parallel
  {
    double* res = calloc(n, 8); double* rsv = calloc(n, 8);
    uint32_t* touched = malloc(4ull * n); uint8_t* seen = calloc(n, 1);
    uint32_t* cur = malloc(4ull * n); uint32_t* nxt = malloc(4ull * n); double* pend = malloc(8ull * n);
    uint64_t l_cnt[NB] = {0}, l_pops[NB] = {0}, l_edges[NB] = {0}, l_lev[NB] = {0}, l_ent[NB] = {0}, l_maxf[NB] = {0};
#pragma omp for schedule(dynamic, 8)
    for (uint32_t i = 0; i < count; ++i) {
      const uint32_t t = first + i * stride;
      if (t >= n) continue;
      uint32_t nt = 0, nf = 0; uint64_t pops = 0, edges = 0, lev = 0, maxf = 0;
      touched[nt++] = t; seen[t] = 1;
      if (irp[t + 1] == irp[t]) { rsv[t] = 1.0; } else { res[t] = 1.0; cur[nf++] = t; }
      while (nf) {
        if (nf > maxf) maxf = nf;
        for (uint32_t j = 0; j < nf; ++j) { uint32_t v = cur[j]; double rc = res[v]; res[v] = 0; rsv[v] += rc * alpha; pend[j] = (1 - alpha) * rc; }
        pops += nf; lev++;
        uint32_t nn = 0;
        for (uint32_t j = 0; j < nf; ++j) {
          const uint32_t v = cur[j];
          for (uint32_t e = irp[v]; e < irp[v + 1]; ++e) {
            const uint32_t u = ici[e];
            const double add = pend[j] / (double)odeg[u];
            const double old = res[u]; res[u] = old + add;
            if (!seen[u]) { seen[u] = 1; touched[nt++] = u; }
            if (!(old > rmax) && old + add > rmax) nxt[nn++] = u;
          }
          edges += irp[v + 1] - irp[v];
        }
        uint32_t* tmp = cur; cur = nxt; nxt = tmp; nf = nn;
      }
      uint64_t ent = 0;
      for (uint32_t j = 0; j < nt; ++j) { uint32_t v = touched[j]; if (rsv[v] > 0 && rsv[v] >= rmax) ent++; res[v] = 0; rsv[v] = 0; seen[v] = 0; }
      int b = 0; while ((1u << b) < nt && b < NB - 1) ++b;  // bucket b: touched in (2^(b-1), 2^b]
      l_cnt[b]++; l_pops[b] += pops; l_edges[b] += edges; l_lev[b] += lev; l_ent[b] += ent; if (maxf > l_maxf[b]) l_maxf[b] = maxf;
    }
#pragma omp critical
    for (int b = 0; b < NB; ++b) { b_cnt[b] += l_cnt[b]; b_pops[b] += l_pops[b]; b_edges[b] += l_edges[b]; b_lev[b] += l_lev[b]; b_ent[b] += l_ent[b]; if (l_maxf[b] > b_maxf[b]) b_maxf[b] = l_maxf[b]; }
  }
  uint64_t tc = 0, te = 0;
  for (int b = 0; b < NB; ++b) { tc += b_cnt[b]; te += b_edges[b]; }
  printf("n=%u m=%llu targets=%llu rmax=%g\n", n, (unsigned long long)m, (unsigned long long)tc, rmax);
  printf("%10s %9s %7s %12s %7s %12s %8s %9s %9s\n", "touched<=", "searches", "cum%", "edges", "cum%", "pops", "levels", "entries", "max_front");
  uint64_t cc = 0, ce = 0;
  for (int b = 0; b < NB; ++b) {
    if (!b_cnt[b]) continue;
    cc += b_cnt[b]; ce += b_edges[b];
    printf("%10u %9llu %6.2f%% %12llu %6.2f%% %12llu %8.2f %9.1f %9llu\n", 1u << b, (unsigned long long)b_cnt[b], 100.0 * cc / tc,
           (unsigned long long)b_edges[b], 100.0 * ce / (te ? te : 1), (unsigned long long)b_pops[b], (double)b_lev[b] / b_cnt[b],
           (double)b_ent[b] / b_cnt[b], (unsigned long long)b_maxf[b]);
  }
  return 0;
}
