// apbs_levels.c — how large are the LEVELS of All-Pair's backward searches, and how many distinct destinations do
// their edges have?  (Round 4: sizing the stream-bin-reduce path of the dense tier's shared levels.)  Frontier-
// synchronous backward push (Backward_Search.java:38-100 as the engine schedules it) from every `stride`-th target;
// per bucket of level size (edges of the level, powers of two): levels, edges, distinct destinations, and the share
// of the destinations' ids below 16 K / 64 K (internal order = out-degree descending: the hot range).
//   gcc -O2 -fopenmp -o /tmp/apbs_levels tools/exp/apbs_levels.c -lm
//   /tmp/apbs_levels <csr.bin> <first> <count> <stride> <rmax>
// csr.bin: uint32 n, uint64 m, out_deg[n] (uint32), in_rp[n+1] (uint32), in_ci[m] (int32); ids in the engine's
// internal order (tools/exp/apbs_levels_csr.py writes it).
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
  enum { NB = 32 };
  uint64_t g_lev[NB] = {0}, g_edges[NB] = {0}, g_dist[NB] = {0}, g_hot16[NB] = {0}, g_hot64[NB] = {0}, g_front[NB] = {0};
  uint64_t g_search_edges = 0, g_search_big = 0;
  uint64_t g_h8[NB] = {0}, g_f8[NB] = {0}, g_f16[NB] = {0};  // hits < 8 K; flush atomics (distinct hot ids per 32768-edge chunk) for 8 K / 16 K tables  // edges of searches that touch > 1536 nodes (tier 2)
#pragma omp parallel
  {
    double* res = calloc(n, 8);
    uint32_t* touched = malloc(4ull * n); uint8_t* seen = calloc(n, 1); uint32_t* stamp = calloc(n, 4);
    uint32_t* cur = malloc(4ull * n); uint32_t* nxt = malloc(4ull * n); double* pend = malloc(8ull * n);
    uint64_t l_lev[NB] = {0}, l_edges[NB] = {0}, l_dist[NB] = {0}, l_hot16[NB] = {0}, l_hot64[NB] = {0}, l_front[NB] = {0};
    uint64_t s_lev[NB], s_edges[NB], s_dist[NB], s_hot16[NB], s_hot64[NB], s_front[NB];
    uint64_t l_se = 0, l_sb = 0;
    uint64_t l_h8[NB] = {0}, l_f8[NB] = {0}, l_f16[NB] = {0}, s_h8[NB], s_f8[NB], s_f16[NB];
    uint32_t* cstamp = calloc(16384, 4); uint32_t ctick = 0;
    uint32_t tick = 0;
#pragma omp for schedule(dynamic, 8)
    for (uint32_t i = 0; i < count; ++i) {
      const uint32_t t = first + i * stride;
      if (t >= n) continue;
      memset(s_lev, 0, sizeof s_lev); memset(s_edges, 0, sizeof s_edges); memset(s_dist, 0, sizeof s_dist);
      memset(s_hot16, 0, sizeof s_hot16); memset(s_h8, 0, sizeof s_h8); memset(s_f8, 0, sizeof s_f8); memset(s_f16, 0, sizeof s_f16); memset(s_hot64, 0, sizeof s_hot64); memset(s_front, 0, sizeof s_front);
      uint32_t nt = 0, nf = 0; uint64_t edges_all = 0;
      touched[nt++] = t; seen[t] = 1;
      if (irp[t + 1] != irp[t]) { res[t] = 1.0; cur[nf++] = t; }
      while (nf) {
        for (uint32_t j = 0; j < nf; ++j) { uint32_t v = cur[j]; double rc = res[v]; res[v] = 0; pend[j] = (1 - alpha) * rc; }
        uint32_t nn = 0; uint64_t E = 0, D = 0, h16 = 0, h64 = 0, h8 = 0, f8 = 0, f16 = 0, epos = 0;
        if (++ctick == 0) { memset(cstamp, 0, 4 * 16384); ctick = 1; }
        if (++tick == 0) { memset(stamp, 0, 4ull * n); tick = 1; }
        for (uint32_t j = 0; j < nf; ++j) {
          const uint32_t v = cur[j];
          for (uint32_t e = irp[v]; e < irp[v + 1]; ++e) {
            const uint32_t u = ici[e];
            const double add = pend[j] / (double)odeg[u];
            const double old = res[u]; res[u] = old + add;
            if (!seen[u]) { seen[u] = 1; touched[nt++] = u; }
            if (stamp[u] != tick) { stamp[u] = tick; D++; }
            if ((epos++ & 32767) == 32767) { if (++ctick == 0) { memset(cstamp, 0, 4 * 16384); ctick = 1; } }
            if (u < 16384) { h16++; if (cstamp[u] != ctick) { cstamp[u] = ctick; f16++; if (u < 8192) f8++; } }
            if (u < 8192) h8++;
            if (u < 65536) h64++;
            if (!(old > rmax) && old + add > rmax) nxt[nn++] = u;
          }
          E += irp[v + 1] - irp[v];
        }
        int b = 0; while ((1ull << b) < E && b < NB - 1) ++b;
        s_lev[b]++; s_edges[b] += E; s_dist[b] += D; s_hot16[b] += h16; s_hot64[b] += h64; s_front[b] += nf; s_h8[b] += h8; s_f8[b] += f8; s_f16[b] += f16;
        edges_all += E;
        uint32_t* tmp = cur; cur = nxt; nxt = tmp; nf = nn;
      }
      for (uint32_t j = 0; j < nt; ++j) { uint32_t v = touched[j]; res[v] = 0; seen[v] = 0; }
      l_se += edges_all;
      if (nt > 1536) {  // a dense-tier search: its levels count
        l_sb += edges_all;
        for (int b = 0; b < NB; ++b) { l_lev[b] += s_lev[b]; l_edges[b] += s_edges[b]; l_dist[b] += s_dist[b]; l_hot16[b] += s_hot16[b]; l_hot64[b] += s_hot64[b]; l_front[b] += s_front[b]; l_h8[b] += s_h8[b]; l_f8[b] += s_f8[b]; l_f16[b] += s_f16[b]; }
      }
    }
#pragma omp critical
    {
      for (int b = 0; b < NB; ++b) { g_lev[b] += l_lev[b]; g_edges[b] += l_edges[b]; g_dist[b] += l_dist[b]; g_hot16[b] += l_hot16[b]; g_hot64[b] += l_hot64[b]; g_front[b] += l_front[b]; g_h8[b] += l_h8[b]; g_f8[b] += l_f8[b]; g_f16[b] += l_f16[b]; }
      g_search_edges += l_se; g_search_big += l_sb;
    }
  }
  uint64_t te = 0;
  for (int b = 0; b < NB; ++b) te += g_edges[b];
  printf("n=%u m=%llu rmax=%g: edges of all sampled searches %llu, of dense-tier searches (> 1536 nodes touched) %llu\n", n,
         (unsigned long long)m, rmax, (unsigned long long)g_search_edges, (unsigned long long)g_search_big);
  printf("%12s %9s %14s %7s %14s %8s %8s %8s %10s\n", "level edges<=", "levels", "edges", "cum%", "distinct dst", "edges/dst", "<16K %", "<64K %", "front/lvl");
  printf("(then: hits below 8 K in %%, flush atomics per edge with an 8 K / 16 K table flushed every 32768 edges)\n");
  uint64_t ce = 0;
  for (int b = 0; b < NB; ++b) {
    if (!g_lev[b]) continue;
    ce += g_edges[b];
    printf("%12llu %9llu %14llu %6.2f%% %14llu %8.2f %7.1f%% %7.1f%% %10.1f\n", 1ull << b, (unsigned long long)g_lev[b],
           (unsigned long long)g_edges[b], 100.0 * ce / (te ? te : 1), (unsigned long long)g_dist[b],
           (double)g_edges[b] / (g_dist[b] ? g_dist[b] : 1), 100.0 * g_hot16[b] / (g_edges[b] ? g_edges[b] : 1),
           100.0 * g_hot64[b] / (g_edges[b] ? g_edges[b] : 1), (double)g_front[b] / g_lev[b]);
    printf("%12s %9s %14s  <8K %5.1f%%  flush8 %.3f  flush16 %.3f  -> RMW per edge: 8K %.3f  16K %.3f\n", "", "", "", 100.0 * g_h8[b] / (g_edges[b] ? g_edges[b] : 1),
           (double)g_f8[b] / (g_edges[b] ? g_edges[b] : 1), (double)g_f16[b] / (g_edges[b] ? g_edges[b] : 1),
           1.0 - (double)g_h8[b] / (g_edges[b] ? g_edges[b] : 1) + (double)g_f8[b] / (g_edges[b] ? g_edges[b] : 1),
           1.0 - (double)g_hot16[b] / (g_edges[b] ? g_edges[b] : 1) + (double)g_f16[b] / (g_edges[b] ? g_edges[b] : 1));
  }
  return 0;
}
