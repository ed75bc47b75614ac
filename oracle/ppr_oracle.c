/*
 * ppr_oracle.c — CPU restatement of the reference's PPR algorithms.  TEST INFRASTRUCTURE ONLY:
 * see ppr_oracle.h for who may use it and for the "parity unpinned" statement.
 *
 * State is held in dense arrays indexed by mapped node id.  The reference keeps
 * HashMap<Long,Double> keyed by original id; an absent key and a 0.0 value are
 * indistinguishable to every consumer on the path (lookups default to 0.0:
 * Forward_Push.java:90-92,123-125), so the dense form computes the same doubles.  The one place
 * where map *membership* matters is kth_ppr / retrieveTopK, which see only touched nodes; every
 * reserve entry the reference creates is > 0, so "value > 0" stands for membership there.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (no FMA contraction, so the doubles match the
 * HIP kernels, which are built with -ffp-contract=off too).
 */
#define _POSIX_C_SOURCE 199309L
#include "ppr_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------ helpers */

static inline uint32_t deg_out(const orc_graph* g, int32_t v) { return g->out_rp[v + 1] - g->out_rp[v]; }
static inline uint32_t deg_in(const orc_graph* g, int32_t v) { return g->in_rp[v + 1] - g->in_rp[v]; }

/* The reference's enqueue test `new_residue / (double)out_degree >= rmax`
 * (Forward_Push.java:109,132): d = 0 gives +Inf for r > 0 (always enqueued) and NaN for r = 0. */
static inline int active_fwd(double r, uint32_t d, double rmax) {
  return d > 0 ? (r / (double)d >= rmax) : (r > 0.0);
}

static void* xcalloc(size_t n, size_t sz) {
  void* p = calloc(n ? n : 1, sz);
  if (!p) abort();
  return p;
}
static void* xmalloc(size_t bytes) {
  void* p = malloc(bytes ? bytes : 1);
  if (!p) abort();
  return p;
}

void orc_free(void* p) { free(p); }

/* Tuning of the frontier-synchronous entry points that take no tuning argument (orc_forward_push, the top-k push
 * rounds, orc_fora_topk): tests that change the engine's tuning hand the same values to the twin. */
static orc_tuning g_sync_tuning;
static int g_sync_tuning_set = 0;
void orc_set_sync_tuning(const orc_tuning* t) {
  if (t) { g_sync_tuning = *t; g_sync_tuning_set = 1; } else g_sync_tuning_set = 0;
}
static void sync_tuning(orc_tuning* t) {
  if (g_sync_tuning_set) *t = g_sync_tuning; else orc_tuning_default(t);
}

void orc_tuning_default(orc_tuning* t) {
  /* keep in step with pprhip_tuning_default() in csrc/pprhip_api.cpp */
  t->c_walk_ns = 0.35;
  t->c_edge_ns = 0.06;
  t->c_pop_ns = 0.10;
  t->c_level_ns = 12000.0;
  t->c_dense_edge_ns = 0.012;
  t->c_dense_node_ns = 0.02;
  t->dense_frac = 0.05;
  t->max_rounds = 24;
  t->max_halvings = 6;
  t->halving_ratio = 2.0;
  t->prior_levels = 16;
  t->gs_blocks = 2;
  t->gs_frac = 0.1;
}

/* ------------------------------------------------------------------ Philox4x32-10 */

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* ------------------------------------------------------------------ parameters (a10) */

void orc_conf_fora_whole_graph(uint32_t n, uint64_t m, double alpha, orc_conf* c) {
  /* Algo_Conf.java:45-53 */
  memset(c, 0, sizeof *c);
  c->alpha = alpha;
  c->delta = 1.0 / (double)n;
  c->pfail = 1.0 / (double)n;
  c->rsum = 1.0;
  c->n = n;
  c->m = m;
}

void orc_conf_fora_topk(uint32_t n, uint64_t m, int k, double alpha, orc_conf* c) {
  /* Algo_Conf.java:71-81; Math.log(node_amount / k) is an int division */
  memset(c, 0, sizeof *c);
  c->alpha = alpha;
  c->min_delta = 1.0 / (double)n;
  c->k = k;
  c->delta = 1.0 / (double)k;
  c->pfail = 1.0 / (double)n / (double)n / log((double)((int32_t)n / k));
  c->rsum = 1.0;
  c->n = n;
  c->m = m;
}

void orc_fora_whole_params(const orc_conf* c, double eps, double* rmax0, double* omega) {
  /* Fora_Whole_Graph.java:86-87 */
  *rmax0 = eps * sqrt(c->delta / 3.0 / (double)c->m / log(2.0 / c->pfail)) / (1.0 - c->alpha);
  *omega = (eps + 2.0) * log(2.0 / c->pfail) / eps / eps / c->delta;
}

void orc_fora_topk_params(const orc_conf* c, double eps, double delta, double* min_rmax, double* rmax_scaled,
                          double* omega) {
  /* Fora_Topk.java:109-113,124-125,133 */
  double e = eps * 0.5;
  *min_rmax = e * sqrt(c->min_delta / 3 / (double)c->m / log(2 / c->pfail));
  double rmax = e * sqrt(delta / 3.0 / (double)c->m / log(2.0 / c->pfail));
  *omega = (e + 2.0) * log(2.0 / c->pfail) / e / e / delta;
  rmax *= sqrt((double)c->m * rmax) * 3.0;
  *rmax_scaled = rmax;
}

/* ------------------------------------------------------------------ power method (a12) */

void orc_power_method(const orc_graph* g, int32_t src, double alpha, int iters, double* reserve) {
  /* Power_Method.java:44-101 */
  uint32_t n = g->n;
  double* cur = (double*)xcalloc(n, sizeof(double));
  double* nxt = (double*)xcalloc(n, sizeof(double));
  memset(reserve, 0, n * sizeof(double));
  cur[src] = 1.0; /* :54 */
  for (int it = 0; it < iters; ++it) {
    memset(nxt, 0, n * sizeof(double)); /* :58 residue.clear() */
    for (uint32_t v = 0; v < n; ++v) {
      double r = cur[v];
      if (!(r > 0)) continue; /* :66 */
      uint32_t d = deg_out(g, (int32_t)v);
      reserve[v] += r * alpha; /* :70 */
      double remain = r * (1 - alpha); /* :73 */
      if (d == 0) {
        nxt[src] += remain; /* :74-80 */
      } else {
        double avg = remain / d; /* :82 */
        for (uint32_t e = g->out_rp[v]; e < g->out_rp[v + 1]; ++e) nxt[g->out_ci[e]] += avg; /* :84-95 */
      }
    }
    double* t = cur; cur = nxt; nxt = t;
  }
  free(cur);
  free(nxt);
}

/* ------------------------------------------------------------------ forward push, FIFO (a1) */

static double fwd_push_fifo(const orc_graph* g, int32_t s, double alpha, double rmax, double* reserve,
                            double* residue, orc_stats* st) {
  /* Forward_Push.java:63-142 */
  uint32_t n = g->n;
  memset(reserve, 0, n * sizeof(double)); /* :64-65 */
  memset(residue, 0, n * sizeof(double));
  double rsum_local = 1.0; /* :68 */
  double rsum = 1.0;       /* the object's field as constructed by Fora_Whole_Graph.java:94 */
  uint32_t d_s = deg_out(g, s);
  if (d_s == 0) { /* :72-76 */
    reserve[s] = 1.0;
    return 0.0;
  }
  int32_t* q = (int32_t*)xmalloc((size_t)(n + 1) * sizeof(int32_t));
  uint8_t* inq = (uint8_t*)xcalloc(n, 1);
  uint32_t head = 0, tail = 0, cap = n + 1;
  q[tail++] = s; /* :81-83 */
  inq[s] = 1;
  residue[s] = 1.0;
  while (head != tail) {
    int32_t v = q[head]; /* :86 */
    head = (head + 1 == cap) ? 0 : head + 1;
    inq[v] = 0;
    double rc = residue[v];
    residue[v] = 0.0; /* :89 */
    reserve[v] = reserve[v] + rc * alpha; /* :91-95 */
    rsum_local -= rc * alpha; /* :97 */
    uint32_t d = deg_out(g, v);
    if (st) st->pops++;
    if (d == 0) { /* :101-115 */
      double ns = residue[s] + rc * (1.0 - alpha);
      residue[s] = ns;
      if (st) st->dead_end_pops++;
      if (d_s > 0 && ns / (double)d_s >= rmax && !inq[s]) {
        q[tail] = s;
        tail = (tail + 1 == cap) ? 0 : tail + 1;
        inq[s] = 1;
        if (st) st->enqueues++;
      }
      continue; /* skips the rsum update at :140 */
    }
    double avg = ((1.0 - alpha) * rc) / (double)d; /* :117 */
    for (uint32_t e = g->out_rp[v]; e < g->out_rp[v + 1]; ++e) { /* :119-139 */
      int32_t u = g->out_ci[e];
      double nr = residue[u] + avg;
      residue[u] = nr;
      uint32_t du = deg_out(g, u);
      if (nr / (double)du >= rmax && !inq[u]) { /* d = 0: +Inf >= rmax */
        q[tail] = u;
        tail = (tail + 1 == cap) ? 0 : tail + 1;
        inq[u] = 1;
        if (st) st->enqueues++;
      }
    }
    if (st) st->edge_pushes += d;
    rsum = rsum_local; /* :140 */
  }
  free(q);
  free(inq);
  return rsum;
}

/* ------------------------------------------------------------------ forward push, level-synchronous twin */

typedef struct sync_ws {
  int32_t *cur, *nxt;
  double* contrib;
  uint8_t* inq; /* top-k rounds: membership in the next frontier (the reference's nodesInQueue) */
  uint32_t ncur, nnxt;
  /* dense sweeps (engine: k_dense_edges + k_dense_apply): pending contributions per node, current and next */
  double *P, *Q, *acc;
  uint8_t* blk; /* block of every node for the Gauss-Seidel sweeps (NULL until first needed) */
  int blk_B;    /* block count blk[] was built for */
} sync_ws;

static void sync_ws_init(sync_ws* w, uint32_t n) {
  w->cur = (int32_t*)xmalloc((size_t)n * sizeof(int32_t));
  w->nxt = (int32_t*)xmalloc((size_t)n * sizeof(int32_t));
  w->contrib = (double*)xmalloc((size_t)n * sizeof(double));
  w->inq = (uint8_t*)xcalloc((size_t)n, 1);
  w->P = (double*)xcalloc((size_t)n, sizeof(double));
  w->Q = (double*)xcalloc((size_t)n, sizeof(double));
  w->acc = (double*)xcalloc((size_t)n, sizeof(double));
  w->blk = NULL;
  w->blk_B = 0;
  w->ncur = w->nnxt = 0;
}
static void sync_ws_free(sync_ws* w) {
  free(w->cur);
  free(w->nxt);
  free(w->contrib);
  free(w->inq);
  free(w->P);
  free(w->Q);
  free(w->acc);
  free(w->blk);
}

static double level_model_cost(const orc_graph* g, const orc_tuning* t, uint64_t nf, uint64_t ef, int* dense) {
  int d = (double)(ef + nf) >= t->dense_frac * (double)g->m;
  if (dense) *dense = d;
  if (d) return t->c_level_ns + t->c_dense_edge_ns * (double)g->m + t->c_dense_node_ns * (double)g->n;
  return t->c_level_ns + t->c_edge_ns * (double)ef + t->c_pop_ns * (double)nf;
}

/* FORA rounds that are certain to be followed by another halving do not need their sparse tail: the
 * nodes it would push are picked up by the next round's lower threshold.  The engine and this twin end
 * such a round after the first sparse level that follows its dense levels (a round without dense
 * levels is short anyway).  fixed: the caller knows another round follows; otherwise the round loop's
 * own condition (model cost so far < c_walk * rsum * omega) is evaluated at that point. */
typedef struct round_cut {
  int enabled, fixed, had_dense, checked, taken;
  double omega, c_walk;
} round_cut;

static double sum_array(const double* a, uint32_t n);

/* Block of every node for the Gauss-Seidel form of the dense sweeps: the engine's rule.  Rows of a sweep are the
 * nodes with in-edges in the engine's internal order (out-degree descending, ties by id); block b holds the row
 * ordinals [jb[b], jb[b + 1]) where jb[b] is the first ordinal whose in-edge prefix reaches b * m / B, rounded down
 * to a multiple of 256 (whole workgroup tiles).  Nodes without in-edges are applied with the last block. */
typedef struct deg_id { uint32_t d; int32_t id; } deg_id;
static int cmp_deg_desc(const void* a, const void* b) {
  const deg_id* x = (const deg_id*)a; const deg_id* y = (const deg_id*)b;
  if (x->d != y->d) return (x->d < y->d) - (x->d > y->d);
  return (x->id > y->id) - (x->id < y->id);
}
static void build_blocks(const orc_graph* g, sync_ws* w, int B) {
  uint32_t n = g->n;
  if (w->blk && w->blk_B == B) return;
  free(w->blk);
  w->blk = (uint8_t*)xmalloc(n);
  w->blk_B = B;
  deg_id* order = (deg_id*)xmalloc((size_t)n * sizeof(deg_id));
  for (uint32_t v = 0; v < n; ++v) { order[v].d = deg_out(g, (int32_t)v); order[v].id = (int32_t)v; }
  qsort(order, n, sizeof(deg_id), cmp_deg_desc); /* stable by construction: ties ordered by id */
  /* ordinals of the rows with in-edges, their in-edge prefix, block starts */
  uint32_t* rows = (uint32_t*)xmalloc((size_t)n * sizeof(uint32_t));
  uint64_t* pre = (uint64_t*)xmalloc(((size_t)n + 1) * sizeof(uint64_t));
  uint32_t n_nz = 0;
  pre[0] = 0;
  for (uint32_t i = 0; i < n; ++i) {
    int32_t v = order[i].id;
    uint32_t din = g->in_rp[v + 1] - g->in_rp[v];
    if (!din) continue;
    rows[n_nz] = (uint32_t)v;
    pre[n_nz + 1] = pre[n_nz] + din;
    n_nz++;
  }
  uint32_t jb[65];
  jb[0] = 0;
  for (int b = 1; b < B; ++b) {
    uint64_t target = (uint64_t)b * g->m / (uint64_t)B;
    uint32_t lo = 0, hi = n_nz; /* first ordinal with prefix >= target */
    while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (pre[mid] >= target) hi = mid; else lo = mid + 1; }
    uint32_t j = lo & ~255u;
    jb[b] = j < jb[b - 1] ? jb[b - 1] : j;
  }
  jb[B] = n_nz;
  memset(w->blk, B - 1, n); /* nodes without in-edges: last block */
  for (int b = 0; b < B; ++b)
    for (uint32_t j = jb[b]; j < jb[b + 1]; ++j) w->blk[rows[j]] = (uint8_t)b;
  free(order); free(rows); free(pre);
}

enum { GS_J = 0, GS_E = 1, GS_G = 2, GS_F = 3 }; /* Jacobi, entry, in-place, flush (see engine.hpp) */

/* One dense sweep in the engine's form: block by block, every row of the block first sums the pending
 * contributions of its in-neighbours as they stand (k_dense_edges), then the block's rows are applied
 * (k_dense_apply): the sum lands on the residue; a row that now belongs to the queue is prepared at once for the
 * next level (residue taken, reserve credited: Forward_Push.java:86-97 moved to the moment the node is enqueued) and
 * its contribution written to Q; what it leaves in P depends on the sweep's state:
 *   Jacobi   P untouched: every contribution of this sweep is read by all rows (today's schedule, B = 1 semantics);
 *   entry    P[u] += new: later blocks read old + new, earlier ones have read old;
 *   in-place P[u]  = new: P holds contributions that still have to reach the blocks up to their own;
 *   flush    P[u]  = 0: pending contributions are delivered, new ones reach nobody in this sweep (clean state).
 * Returns the frontier the sweep leaves in *nf / *ef; dead-end mass of prepared nodes goes to *dead_next. */
static void dense_sweep(const orc_graph* g, int32_t s, double alpha, double rmax, double* reserve, double* residue,
                        sync_ws* w, uint8_t* parked, double min_rmax, int general, int B, int state, double* dead_cell,
                        double* dead_next, uint64_t* nf, uint64_t* ef, orc_stats* st) {
  uint32_t n = g->n;
  double* P = w->P; double* Q = w->Q; double* acc = w->acc;
  *nf = 0; *ef = 0;
  for (int b = 0; b < B; ++b) {
    for (uint32_t u = 0; u < n; ++u) { /* k_dense_edges over the block's rows */
      if (B > 1 && w->blk[u] != b) continue;
      double a = 0.0;
      for (uint32_t e = g->in_rp[u]; e < g->in_rp[u + 1]; ++e) a += P[g->in_ci[e]];
      acc[u] = a;
    }
    for (uint32_t u = 0; u < n; ++u) { /* k_dense_apply over the block's rows */
      if (B > 1 && w->blk[u] != b) continue;
      double a = acc[u];
      if ((int32_t)u == s && *dead_cell > 0.0) { a += *dead_cell; *dead_cell = 0.0; } /* Forward_Push.java:101-113 */
      double cn = 0.0;
      if (a > 0.0) {
        uint32_t d = deg_out(g, (int32_t)u);
        double old = residue[u], nw = old + a;
        int was = active_fwd(old, d, rmax);
        int join = !was && active_fwd(nw, d, rmax);
        if (general && was) join = 1; /* met the threshold without being queued: joins with its first mass */
        if (parked && active_fwd(nw, d, min_rmax)) parked[u] = 1;
        if (join) {
          reserve[u] = reserve[u] + nw * alpha;
          residue[u] = 0.0;
          if (d == 0) { *dead_next += nw * (1.0 - alpha); if (st) st->dead_end_pops++; }
          else cn = ((1.0 - alpha) * nw) / (double)d;
          (*nf)++; *ef += d;
        } else {
          residue[u] = nw;
        }
      }
      Q[u] = cn;
      if (state == GS_E) P[u] = P[u] + cn;
      else if (state == GS_G) P[u] = cn;
      else if (state == GS_F) P[u] = 0.0;
    }
  }
}

/* Runs levels from the frontier in w->cur until it is empty.  One level = every frontier node
 * pushed at once from its residue at level start (Forward_Push.java:86-139 per node); a level that touches a large
 * part of the graph runs as a dense sweep (above), others edge by edge.
 * parked/min_rmax != NULL adds the second threshold of forward_push_topk (:226-237). */
static void fwd_levels_sync(const orc_graph* g, int32_t s, double alpha, double rmax, double* reserve, double* residue,
                            sync_ws* w, uint8_t* parked, double min_rmax, const orc_tuning* tun, orc_stats* st,
                            round_cut* cut) {
  uint32_t n = g->n;
  uint32_t d_s = deg_out(g, s);
  /* Only a top-k round whose threshold lies below min_rmax can hold a node that meets the threshold without being
   * queued; everywhere else the membership rule and the crossing rule pick the same nodes, and the crossing rule also
   * ends on the degenerate configurations (n div k = 1 makes pfail infinite and every threshold 0, where the
   * reference's own loop would never end). */
  const int general = parked != NULL && rmax < min_rmax;
  const int B = (tun && tun->gs_blocks > 1 && tun->gs_blocks <= 64) ? tun->gs_blocks : 1;
  const double gs_thresh = (tun && B > 1) ? ceil(tun->gs_frac * (double)g->m) : 0.0;
  if (B > 1) build_blocks(g, w, B);
  int prepared = 0; /* the frontier is held as prepared contributions in w->P (after a dense sweep) */
  int dirty = 0;    /* w->P is in the in-place state: its contributions have reached the later blocks only */
  uint64_t nf = w->ncur, ef = 0;
  double dead = 0.0; /* dead-end mass waiting to land on the source */
  for (uint32_t i = 0; i < w->ncur; ++i) ef += deg_out(g, w->cur[i]);
  while (nf) {
    int dense = 0;
    double c = tun ? level_model_cost(g, tun, nf, ef, &dense) : 0.0;
    if (!tun) dense = 0;
    if (dirty && !dense) /* a sweep that only runs to flush the contribution array costs a dense level */
      c = tun->c_level_ns + tun->c_dense_edge_ns * (double)g->m + tun->c_dense_node_ns * (double)g->n;
    if (dense || dirty) {
      if (!prepared) { /* list form -> contributions in place (k_sparse_prepare with scatter) */
        for (uint32_t i = 0; i < w->ncur; ++i) {
          int32_t v = w->cur[i];
          double rc = residue[v];
          residue[v] = 0.0;
          reserve[v] = reserve[v] + rc * alpha;
          uint32_t d = deg_out(g, v);
          if (d == 0) { dead += rc * (1.0 - alpha); if (st) st->dead_end_pops++; w->P[v] = 0.0; }
          else w->P[v] = ((1.0 - alpha) * rc) / (double)d;
        }
        prepared = 1;
        dirty = 0;
      }
      int state;
      const int big = B > 1 && (double)(nf + ef) >= gs_thresh;
      if (dirty) state = big ? GS_G : GS_F;
      else state = big ? GS_E : GS_J;
      uint64_t nf2 = 0, ef2 = 0;
      double dead_next = 0.0;
      dense_sweep(g, s, alpha, rmax, reserve, residue, w, parked, min_rmax, general, B, state, &dead, &dead_next, &nf2,
                  &ef2, st);
      dead += dead_next; /* lands with the next level */
      double* t = w->P; w->P = w->Q; w->Q = t;
      /* what the sweep left in the other buffer is overwritten by the next sweep; nodes that no sweep applies
       * (none here: every node is applied) keep nothing */
      dirty = (state == GS_E || state == GS_G) && nf2 > 0;
      if (st) {
        st->model_cost_ns += c;
        st->levels++;
        st->dense_levels++;
        st->dense_nodes += nf;
        st->enqueues += nf2;
        if (cut && cut->enabled) cut->had_dense = 1;
      }
      nf = nf2; ef = ef2;
      if (nf == 0) { /* nothing pending: leave no stale contribution behind */
        memset(w->P, 0, (size_t)n * sizeof(double));
        prepared = 0;
      }
      continue;
    }
    /* ---- a sparse level */
    uint64_t ef_l = 0;
    if (prepared) { /* contributions in place -> list (k_compact_prepared); dead-end nodes carry none */
      w->ncur = 0;
      for (uint32_t v = 0; v < n; ++v)
        if (w->P[v] > 0.0) { w->contrib[w->ncur] = w->P[v]; w->cur[w->ncur++] = (int32_t)v; w->P[v] = 0.0; }
      prepared = 0;
    } else { /* phase 1: every frontier node gives up its residue */
      for (uint32_t i = 0; i < w->ncur; ++i) {
        int32_t v = w->cur[i];
        double rc = residue[v];
        residue[v] = 0.0;
        reserve[v] = reserve[v] + rc * alpha;
        uint32_t d = deg_out(g, v);
        if (d == 0) {
          dead += rc * (1.0 - alpha);
          w->contrib[i] = 0.0;
          if (st) st->dead_end_pops++;
        } else {
          w->contrib[i] = ((1.0 - alpha) * rc) / (double)d;
        }
      }
    }
    /* phase 2: contributions land; a node joins the next frontier when it crosses the threshold */
    w->nnxt = 0;
    for (uint32_t i = 0; i < w->ncur; ++i) {
      int32_t v = w->cur[i];
      double cc = w->contrib[i];
      ef_l += deg_out(g, v);
      for (uint32_t e = g->out_rp[v]; e < g->out_rp[v + 1]; ++e) {
        int32_t u = g->out_ci[e];
        double old = residue[u];
        double nr = old + cc;
        residue[u] = nr;
        uint32_t du = deg_out(g, u);
        /* Whole-graph pushes: every node at or above the threshold is in a frontier, so "joins the queue" is
         * "crosses the threshold".  Top-k rounds test the new residue and queue membership only
         * (Forward_Push.java:226-231): a node that already met the round's threshold without being queued
         * (possible when the scaled rmax of Fora_Topk.java:133 is below min_rmax) joins with its first mass. */
        int join = general ? (active_fwd(nr, du, rmax) && !w->inq[u])
                           : (!active_fwd(old, du, rmax) && active_fwd(nr, du, rmax));
        if (join) {
          w->nxt[w->nnxt++] = u;
          if (general) w->inq[u] = 1;
        }
        if (parked && active_fwd(nr, du, min_rmax)) parked[u] = 1;
      }
    }
    if (dead > 0.0) { /* Forward_Push.java:101-113 */
      double old = residue[s];
      double nr = old + dead;
      residue[s] = nr;
      dead = 0.0;
      int join = general ? (active_fwd(nr, d_s, rmax) && !w->inq[s])
                         : (!active_fwd(old, d_s, rmax) && active_fwd(nr, d_s, rmax));
      if (join) {
        w->nxt[w->nnxt++] = s;
        if (general) w->inq[s] = 1;
      }
      if (parked && active_fwd(nr, d_s, min_rmax)) parked[s] = 1;
    }
    if (general) /* the next level pops these: they leave the queue */
      for (uint32_t i = 0; i < w->nnxt; ++i) w->inq[w->nxt[i]] = 0;
    if (st) {
      st->model_cost_ns += c;
      st->levels++;
      st->pops += nf;
      st->edge_pushes += ef;
      st->enqueues += w->nnxt;
      if (cut && cut->enabled && cut->had_dense && !cut->checked) {
        cut->checked = 1;
        int more = 1;
        if (!cut->fixed) more = st->model_cost_ns < cut->c_walk * (sum_array(residue, g->n) * (1 - alpha)) * cut->omega;
        if (more) {
          cut->taken = 1;
          w->nnxt = 0; /* the rest of this round's frontier waits for the next threshold */
        }
      }
    }
    (void)ef_l;
    int32_t* t = w->cur; w->cur = w->nxt; w->nxt = t;
    w->ncur = w->nnxt;
    nf = w->ncur;
    ef = 0;
    for (uint32_t i = 0; i < w->ncur; ++i) ef += deg_out(g, w->cur[i]);
  }
}

static double sum_array(const double* a, uint32_t n) {
  double s = 0.0;
  for (uint32_t i = 0; i < n; ++i) s += a[i];
  return s;
}

static double fwd_push_sync(const orc_graph* g, int32_t s, double alpha, double rmax, double* reserve, double* residue,
                            const orc_tuning* tun, orc_stats* st) {
  uint32_t n = g->n;
  memset(reserve, 0, n * sizeof(double));
  memset(residue, 0, n * sizeof(double));
  if (deg_out(g, s) == 0) { /* Forward_Push.java:72-76 */
    reserve[s] = 1.0;
    return 0.0;
  }
  sync_ws w;
  sync_ws_init(&w, n);
  residue[s] = 1.0;
  w.cur[0] = s; /* the source is pushed unconditionally first (:81-86) */
  w.ncur = 1;
  fwd_levels_sync(g, s, alpha, rmax, reserve, residue, &w, NULL, 0.0, tun, st, NULL);
  sync_ws_free(&w);
  return sum_array(residue, n);
}

double orc_forward_push(const orc_graph* g, int schedule, int32_t src, double alpha, double rmax, double* reserve,
                        double* residue, orc_stats* st) {
  if (st) memset(st, 0, sizeof *st);
  orc_tuning tun;
  sync_tuning(&tun);
  double rsum = schedule == ORC_FIFO ? fwd_push_fifo(g, src, alpha, rmax, reserve, residue, st)
                                     : fwd_push_sync(g, src, alpha, rmax, reserve, residue, &tun, st);
  if (st) {
    st->rsum = rsum;
    st->rmax_final = rmax;
  }
  return rsum;
}

/* ------------------------------------------------------------------ forward push top-k (a2) */

struct orc_topk_push {
  const orc_graph* g;
  int schedule;
  int32_t src;
  double alpha;
  double rsum;
  int first;
  double *reserve, *residue;
  /* FIFO: the parked queue Q_next in order; SYNC: a membership flag per node */
  int32_t* qnext;
  uint32_t nqnext;
  uint8_t* parked;
  sync_ws w;
};

orc_topk_push* orc_topk_push_new(const orc_graph* g, int schedule, int32_t src, double alpha) {
  /* Fora_Topk.java:117-121: Q = {s}; new Forward_Push(alpha, rsum = 1, ...) */
  orc_topk_push* p = (orc_topk_push*)xcalloc(1, sizeof *p);
  p->g = g;
  p->schedule = schedule;
  p->src = src;
  p->alpha = alpha;
  p->rsum = 1.0;
  p->first = 1;
  p->reserve = (double*)xcalloc(g->n, sizeof(double));
  p->residue = (double*)xcalloc(g->n, sizeof(double));
  p->qnext = (int32_t*)xmalloc((size_t)(g->n + 1) * sizeof(int32_t));
  p->parked = (uint8_t*)xcalloc(g->n, 1);
  p->qnext[0] = src;
  p->nqnext = 1;
  p->parked[src] = 1;
  sync_ws_init(&p->w, g->n);
  return p;
}

void orc_topk_push_free(orc_topk_push* p) {
  if (!p) return;
  free(p->reserve);
  free(p->residue);
  free(p->qnext);
  free(p->parked);
  sync_ws_free(&p->w);
  free(p);
}

const double* orc_topk_push_reserve(const orc_topk_push* p) { return p->reserve; }
const double* orc_topk_push_residue(const orc_topk_push* p) { return p->residue; }

static double topk_round_fifo(orc_topk_push* p, double min_rmax, double rmax, orc_stats* st) {
  /* Forward_Push.java:144-250 */
  const orc_graph* g = p->g;
  uint32_t n = g->n;
  int32_t s = p->src;
  double alpha = p->alpha;
  double* residue = p->residue;
  double* reserve = p->reserve;
  uint32_t d_s = deg_out(g, s);
  if (d_s == 0) { /* :149-153 */
    reserve[s] = 1.0;
    p->rsum = 0.0;
    return 0.0;
  }
  if (p->first) residue[s] = 1.0; /* :155-156 */
  /* Q := copy of last round's Q_next (Fora_Topk.java:145); Q_next.clear() (:157) */
  uint32_t cap = n + 1;
  int32_t* q = (int32_t*)xmalloc((size_t)cap * sizeof(int32_t));
  uint8_t* inq = (uint8_t*)xcalloc(n, 1);
  uint8_t* inq_next = (uint8_t*)xcalloc(n, 1);
  uint32_t head = 0, tail = 0;
  for (uint32_t i = 0; i < p->nqnext; ++i) { /* :163 nodesInQueue.addAll(Q) */
    q[tail++] = p->qnext[i];
    inq[p->qnext[i]] = 1;
  }
  if (tail == cap) tail = 0;
  uint32_t count = p->nqnext;
  p->nqnext = 0;
  double rsum_local = p->rsum; /* :158 */
  while (count) {
    int32_t v = q[head];
    head = (head + 1 == cap) ? 0 : head + 1;
    count--;
    inq[v] = 0;
    double rc = residue[v];
    uint32_t d = deg_out(g, v);
    if (rc / d >= rmax) { /* :173 (d = 0: +Inf or NaN) */
      residue[v] = 0.0;
      reserve[v] = reserve[v] + rc * alpha;
      rsum_local -= rc * alpha;
      if (st) st->pops++;
      if (d == 0) { /* :186-209 */
        double ns = residue[s] + rc * (1 - alpha);
        residue[s] = ns;
        if (st) st->dead_end_pops++;
        if (d_s > 0 && ns / (double)d_s >= rmax && !inq[s]) {
          q[tail] = s;
          tail = (tail + 1 == cap) ? 0 : tail + 1;
          count++;
          inq[s] = 1;
          if (st) st->enqueues++;
        } else if (d_s > 0 && ns / (double)d_s >= min_rmax && !inq_next[s]) {
          inq_next[s] = 1;
          p->qnext[p->nqnext++] = s;
        }
        continue;
      }
      double avg = ((1.0 - alpha) * rc) / (double)d; /* :211 */
      for (uint32_t e = g->out_rp[v]; e < g->out_rp[v + 1]; ++e) { /* :213-239 */
        int32_t u = g->out_ci[e];
        double nr = residue[u] + avg;
        residue[u] = nr;
        uint32_t du = deg_out(g, u);
        if (nr / (double)du >= rmax && !inq[u]) {
          q[tail] = u;
          tail = (tail + 1 == cap) ? 0 : tail + 1;
          count++;
          inq[u] = 1;
          if (st) st->enqueues++;
        } else if (nr / (double)du >= min_rmax && !inq_next[u]) {
          inq_next[u] = 1;
          p->qnext[p->nqnext++] = u;
        }
      }
      if (st) st->edge_pushes += d;
    } else if (rc / (double)d >= min_rmax && !inq_next[v]) { /* :241-247 */
      inq_next[v] = 1;
      p->qnext[p->nqnext++] = v;
    }
  }
  p->rsum = rsum_local; /* :249 */
  p->first = 0;
  free(q);
  free(inq);
  free(inq_next);
  return p->rsum;
}

static double topk_round_sync(orc_topk_push* p, double min_rmax, double rmax, orc_stats* st) {
  /* Frontier-synchronous form of Forward_Push.java:144-250.  `parked` is Q_next as a set: a
   * node is parked when a push leaves it at r/d >= min_rmax; a round starts from the parked
   * nodes that reach the round's rmax (:173) and keeps the rest parked if still >= min_rmax
   * (:241-247). */
  const orc_graph* g = p->g;
  uint32_t n = g->n;
  int32_t s = p->src;
  if (deg_out(g, s) == 0) {
    p->reserve[s] = 1.0;
    p->rsum = 0.0;
    return 0.0;
  }
  if (p->first) p->residue[s] = 1.0;
  p->w.ncur = 0;
  for (uint32_t v = 0; v < n; ++v) {
    if (!p->parked[v]) continue;
    uint32_t d = deg_out(g, (int32_t)v);
    double r = p->residue[v];
    if (active_fwd(r, d, rmax)) {
      p->w.cur[p->w.ncur++] = (int32_t)v;
      p->parked[v] = 0;
    } else if (!active_fwd(r, d, min_rmax)) {
      p->parked[v] = 0;
    }
  }
  orc_tuning tun;
  sync_tuning(&tun);
  fwd_levels_sync(g, s, p->alpha, rmax, p->reserve, p->residue, &p->w, p->parked, min_rmax, &tun, st, NULL);
  p->rsum = sum_array(p->residue, n);
  p->first = 0;
  return p->rsum;
}

double orc_topk_push_round(orc_topk_push* p, double min_rmax, double rmax, orc_stats* st) {
  return p->schedule == ORC_FIFO ? topk_round_fifo(p, min_rmax, rmax, st) : topk_round_sync(p, min_rmax, rmax, st);
}

/* ------------------------------------------------------------------ random walks (a3, a4) */

int32_t orc_random_walk(const orc_graph* g, int32_t start, double alpha, uint64_t seed, uint32_t stream,
                        uint64_t walk_idx, int no_zero_hop, uint32_t* steps_out) {
  /* Monte_Carlo.java:60-94 / :96-133.  Walk (seed, stream, start, walk_idx) is a pure function:
   * Philox4x32-10, key = seed, counter = (start, idx_lo, idx_hi16 | stream << 16, block).
   * Decision k uses block k >> 1 and the word pair 2(k & 1), 2(k & 1) + 1: the first word makes
   * the stop test's uniform (word * 2^-32 < alpha stands for nextDouble(1.0) < alpha, :76), the
   * second picks the neighbour ((word * d) >> 32 stands for nextInt(d), :84).  With
   * no_zero_hop the first decision is the forced hop of :111-112 (its stop word is unused). */
  uint32_t d0 = deg_out(g, start);
  if (steps_out) *steps_out = 0;
  if (d0 == 0) return start; /* :70-72 / :106-108 */
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  uint32_t ctr[4] = {(uint32_t)start, (uint32_t)walk_idx, (uint32_t)((walk_idx >> 32) & 0xFFFFu) | (stream << 16), 0};
  uint32_t x[4] = {0, 0, 0, 0};
  int32_t cur = start;
  uint32_t k = 0, moves = 0;
  int forced = no_zero_hop != 0;
  for (;;) {
    if ((k & 1u) == 0) {
      ctr[3] = k >> 1;
      orc_philox4x32_10(ctr, key, x);
    }
    uint32_t w_stop = x[2 * (k & 1u)], w_pick = x[2 * (k & 1u) + 1];
    k++;
    if (!forced && (double)w_stop * (1.0 / 4294967296.0) < alpha) break; /* :76-78 */
    forced = 0;
    uint32_t d = deg_out(g, cur);
    if (d > 0)
      cur = g->out_ci[g->out_rp[cur] + (uint32_t)(((uint64_t)w_pick * d) >> 32)]; /* :81-86 */
    else
      cur = start; /* :87-90 */
    moves++;
  }
  if (steps_out) *steps_out = moves;
  return cur;
}

/* ------------------------------------------------------------------ FORA whole graph (a5) */

static void fora_mc_phase(const orc_graph* g, const double* residue, double rsum_local, double omega, double alpha,
                          uint64_t seed, double* reserve, orc_stats* st) {
  /* Fora_Whole_Graph.java:112-140 */
  uint32_t n = g->n;
  double nrw_d = omega * rsum_local;
  long long nrw = (nrw_d == nrw_d) ? (long long)nrw_d : 0; /* (long) cast; NaN -> 0 */
  for (uint32_t v = 0; v < n; ++v) {
    double r = residue[v];
    if (!(r > 0.0)) continue; /* r = 0 entries add 0.0 and start no walk */
    double incr_cur = r * alpha; /* :122 */
    r *= (1.0 - alpha);          /* :123 */
    reserve[v] = reserve[v] + incr_cur;
    if (nrw <= 0 || !(rsum_local > 0.0)) continue;
    long long omega_i = (long long)ceil(r / rsum_local * (double)nrw); /* :129 */
    double a_i = r / rsum_local * (double)nrw / (double)omega_i;        /* :130 */
    double incr = a_i / (double)nrw * rsum_local;                       /* :131 */
    if (st) st->mc_sources++;
    for (long long j = 0; j < omega_i; ++j) { /* :133-139 */
      uint32_t steps;
      int32_t t = orc_random_walk(g, (int32_t)v, alpha, seed, 0, (uint64_t)j, 1, &steps);
      reserve[t] = reserve[t] + incr;
      if (st) {
        st->walks++;
        st->walk_steps += steps;
      }
    }
  }
}

void orc_fora_whole(const orc_graph* g, int schedule, int32_t src, double eps, const orc_conf* c, uint64_t seed,
                    int n_rounds, const orc_tuning* tun_in, double* reserve, orc_stats* st) {
  /* Fora_Whole_Graph.java:82-146 */
  uint32_t n = g->n;
  orc_tuning tun;
  if (tun_in) tun = *tun_in; else orc_tuning_default(&tun);
  orc_stats local;
  if (!st) st = &local;
  memset(st, 0, sizeof *st);
  double alpha = c->alpha;
  double rsum_local = c->rsum, rmax_local, omega_local;
  orc_fora_whole_params(c, eps, &rmax_local, &omega_local);
  double* residue = (double*)xcalloc(n, sizeof(double));
  memset(reserve, 0, n * sizeof(double));
  int rounds = 0;
  double rmax_used = rmax_local;
  if (schedule == ORC_FIFO) {
    /* :93-103 — restart from scratch each round.  The reference compares wall-clock push time
     * with 400 ns * rsum * omega; here push time is the model c_edge*edges + c_pop*pops. */
    double cost = 0.0;
    for (;;) {
      int more = n_rounds > 0 ? rounds < n_rounds : (cost < tun.c_walk_ns * rsum_local * omega_local && rounds < tun.max_rounds);
      if (!more) break;
      orc_stats ps;
      memset(&ps, 0, sizeof ps);
      double fp_rsum = fwd_push_fifo(g, src, alpha, rmax_local, reserve, residue, &ps);
      cost += tun.c_edge_ns * (double)ps.edge_pushes + tun.c_pop_ns * (double)ps.pops;
      st->pops += ps.pops; st->edge_pushes += ps.edge_pushes; st->enqueues += ps.enqueues;
      st->dead_end_pops += ps.dead_end_pops;
      rsum_local = fp_rsum * (1 - alpha); /* :101 */
      rmax_used = rmax_local;
      rmax_local /= 2.0; /* :102 */
      rounds++;
      if (n_rounds > 0 && !(rsum_local > 0.0)) break;
    }
    st->model_cost_ns = cost;
  } else {
    /* resumed rounds: push to rmax0, then continue to rmax0/2, ... on the same state */
    sync_ws w;
    sync_ws_init(&w, n);
    int dead_src = deg_out(g, src) == 0;
    if (n_rounds == 0 && tun.prior_levels > 0 && tun.halving_ratio > 1.0) {
      /* Loop turns that are known to pass before any push: after a push at rmax every r(v) < rmax * d(v), so
       * rsum <= rmax * m and the walks cost at most c_walk * omega * (1 - alpha) * rmax * m; while that bound
       * still covers prior_levels dense levels the turn would be repeated at half the threshold anyway. */
      double walk_bound = tun.c_walk_ns * omega_local * (1 - alpha) * rmax_local * (double)g->m;
      double push_est = (double)tun.prior_levels *
                        (tun.c_level_ns + tun.c_dense_edge_ns * (double)g->m + tun.c_dense_node_ns * (double)g->n);
      for (int h = 0; h < tun.max_halvings && walk_bound >= push_est; ++h) {
        walk_bound /= 2.0;
        rmax_local /= 2.0;
      }
      rmax_used = rmax_local;
    }
    for (;;) {
      int more = n_rounds > 0 ? rounds < n_rounds
                              : (st->model_cost_ns < tun.c_walk_ns * rsum_local * omega_local && rounds < tun.max_rounds);
      if (!more) break;
      if (dead_src) {
        reserve[src] = 1.0;
        rsum_local = 0.0;
        rmax_used = rmax_local;
        rounds++;
        break;
      }
      if (rounds == 0) {
        residue[src] = 1.0;
        w.cur[0] = src;
        w.ncur = 1;
      } else {
        w.ncur = 0;
        for (uint32_t v = 0; v < n; ++v)
          if (active_fwd(residue[v], deg_out(g, (int32_t)v), rmax_local)) w.cur[w.ncur++] = (int32_t)v;
      }
      round_cut cut;
      memset(&cut, 0, sizeof cut);
      cut.fixed = n_rounds > 0;
      cut.enabled = n_rounds > 0 ? rounds + 1 < n_rounds : rounds + 1 < tun.max_rounds;
      cut.omega = omega_local;
      cut.c_walk = tun.c_walk_ns;
      fwd_levels_sync(g, src, alpha, rmax_local, reserve, residue, &w, NULL, 0.0, &tun, st, &cut);
      rsum_local = sum_array(residue, n) * (1 - alpha);
      rmax_used = rmax_local;
      rmax_local /= 2.0;
      /* The reference's loop would turn again (and restart the push from scratch at half the threshold) as
       * long as the push stays cheaper than the walks; when the walks outweigh the push so far by ratio^k, the
       * engine and this twin take k further halvings at once instead of pushing at every threshold between. */
      if (n_rounds == 0 && st->model_cost_ns > 0.0 && tun.halving_ratio > 1.0) {
        double ratio = tun.c_walk_ns * rsum_local * omega_local / st->model_cost_ns;
        for (int h = 1; ratio >= tun.halving_ratio && h < tun.max_halvings; ++h) {
          ratio /= tun.halving_ratio;
          rmax_local /= 2.0;
        }
      }
      rounds++;
      if (n_rounds > 0 && !(rsum_local > 0.0)) break;
    }
    sync_ws_free(&w);
  }
  st->rounds = (uint32_t)rounds;
  st->rsum = rsum_local;
  st->rmax_final = rmax_used;
  st->omega = omega_local;
  fora_mc_phase(g, residue, rsum_local, omega_local, alpha, seed, reserve, st);
  free(residue);
}

/* ------------------------------------------------------------------ k-th largest / top-k (a7) */

static int cmp_desc(const void* a, const void* b) {
  double x = *(const double*)a, y = *(const double*)b;
  return (x < y) - (x > y);
}

int orc_kth_largest(const double* v, uint32_t n, int k, double* kth) {
  /* Algo_Util.java:32-53: value of the k-th largest entry; null when k is out of range.
   * (The reference's quickselect uses a random pivot; the value it returns does not depend on it.) */
  if (k < 1) return 0;
  uint32_t cnt = 0;
  for (uint32_t i = 0; i < n; ++i) cnt += v[i] > 0.0;
  if ((uint32_t)k > cnt) return 0;
  double* t = (double*)xmalloc((size_t)cnt * sizeof(double));
  uint32_t j = 0;
  for (uint32_t i = 0; i < n; ++i)
    if (v[i] > 0.0) t[j++] = v[i];
  qsort(t, cnt, sizeof(double), cmp_desc);
  *kth = t[k - 1];
  free(t);
  return 1;
}

typedef struct idval { int32_t id; double val; } idval;
static int cmp_idval(const void* a, const void* b) {
  const idval* x = (const idval*)a; const idval* y = (const idval*)b;
  if (x->val != y->val) return (x->val < y->val) - (x->val > y->val);
  return (x->id > y->id) - (x->id < y->id);
}

int orc_topk(const double* v, uint32_t n, int k, int32_t* ids, double* vals, int cap) {
  /* Fora_Topk.java:186-199: all entries >= the k-th value (everything when fewer than k),
   * then sorted descending (:82-99); ties broken by id ascending. */
  double kth = 0.0;
  int have = orc_kth_largest(v, n, k, &kth);
  uint32_t cnt = 0;
  for (uint32_t i = 0; i < n; ++i) cnt += (v[i] > 0.0) && (!have || v[i] >= kth);
  idval* t = (idval*)xmalloc((size_t)cnt * sizeof(idval));
  uint32_t j = 0;
  for (uint32_t i = 0; i < n; ++i)
    if ((v[i] > 0.0) && (!have || v[i] >= kth)) { t[j].id = (int32_t)i; t[j].val = v[i]; j++; }
  qsort(t, cnt, sizeof(idval), cmp_idval);
  for (uint32_t i = 0; i < cnt && (int)i < cap; ++i) { ids[i] = t[i].id; vals[i] = t[i].val; }
  free(t);
  return (int)cnt;
}

/* ------------------------------------------------------------------ FORA top-k (a6) */

void orc_fora_topk(const orc_graph* g, int schedule, int32_t src, double eps, const orc_conf* c, uint64_t seed,
                   double* reserve, orc_stats* st) {
  /* Fora_Topk.java:102-184 */
  uint32_t n = g->n;
  orc_stats local;
  if (!st) st = &local;
  memset(st, 0, sizeof *st);
  double alpha = c->alpha;
  double epsilon = eps * 0.5; /* :109-110 */
  double delta_local = c->delta, min_delta = c->min_delta;
  double min_rmax = epsilon * sqrt(min_delta / 3 / (double)c->m / log(2 / c->pfail)); /* :113 */
  double rsum_local = c->rsum;
  double omega_local = 0.0, rmax_local = 0.0;
  memset(reserve, 0, n * sizeof(double));
  orc_topk_push* fp = orc_topk_push_new(g, schedule, src, alpha);
  fp->rsum = rsum_local;
  uint32_t round = 0;
  while (delta_local >= min_delta) { /* :123 */
    rmax_local = epsilon * sqrt(delta_local / 3.0 / (double)c->m / log(2.0 / c->pfail)); /* :124 */
    omega_local = (epsilon + 2.0) * log(2.0 / c->pfail) / epsilon / epsilon / delta_local; /* :125 */
    if (deg_out(g, src) == 0) { /* :126-132 */
      memset(reserve, 0, n * sizeof(double));
      reserve[src] = 1.0;
      rsum_local = 0.0;
      break;
    }
    rmax_local *= sqrt((double)c->m * rmax_local) * 3.0; /* :133 */
    rsum_local = orc_topk_push_round(fp, min_rmax, rmax_local, st); /* :137,142 */
    memcpy(reserve, fp->reserve, n * sizeof(double)); /* :143 — earlier rounds' walks are dropped */
    double rsum_rw = rsum_local * (1.0 - alpha); /* :148 */
    double nrw_d = omega_local * rsum_rw;
    long long nrw = (nrw_d == nrw_d) ? (long long)nrw_d : 0; /* :151 */
    for (uint32_t v = 0; v < n && nrw > 0; ++v) { /* :155-168 */
      double r = fp->residue[v];
      if (!(r > 0.0)) continue;
      long long omega_i = (long long)ceil(r * (double)nrw);
      double a_i = r * (double)nrw / (double)omega_i;
      double incr = a_i / (double)nrw;
      st->mc_sources++;
      for (long long j = 0; j < omega_i; ++j) {
        uint32_t steps;
        int32_t t = orc_random_walk(g, (int32_t)v, alpha, seed, round, (uint64_t)j, 0, &steps);
        reserve[t] = reserve[t] + incr;
        st->walks++;
        st->walk_steps += steps;
      }
    }
    round++;
    double kth = 0.0;
    if (!orc_kth_largest(reserve, n, c->k, &kth)) kth = 0.0; /* :173-174 */
    st->kth_value = kth;
    if (kth >= (1 + epsilon) * delta_local || delta_local <= min_delta) break; /* :175-176 */
    delta_local = fmax(min_delta, delta_local / 4.0); /* :178 */
  }
  st->rounds = round;
  st->rsum = rsum_local;
  st->rmax_final = rmax_local;
  st->omega = omega_local;
  orc_topk_push_free(fp);
}

/* ------------------------------------------------------------------ pure Monte-Carlo */

void orc_monte_carlo(const orc_graph* g, int32_t src, double eps, const orc_conf* c, uint64_t seed, double* ppr,
                     orc_stats* st) {
  /* Monte_Carlo.java:136-158: walks i = 1..omega (long i; i <= omega), estimate = count/omega */
  uint32_t n = g->n;
  orc_stats local;
  if (!st) st = &local;
  memset(st, 0, sizeof *st);
  double omega = 3 * log(2 / c->pfail) / eps / eps / c->delta; /* :145 */
  uint64_t* cnt = (uint64_t*)xcalloc(n, sizeof(uint64_t));
  long long nw = (long long)floor(omega);
  for (long long i = 0; i < nw; ++i) {
    uint32_t steps;
    int32_t t = orc_random_walk(g, src, c->alpha, seed, 0, (uint64_t)i, 0, &steps);
    cnt[t]++;
    st->walks++;
    st->walk_steps += steps;
  }
  for (uint32_t v = 0; v < n; ++v) ppr[v] = cnt[v] ? ((double)cnt[v]) / omega : 0.0; /* :156-157 */
  st->omega = omega;
  st->mc_sources = 1;
  free(cnt);
}

/* ------------------------------------------------------------------ backward search (a8) */

static void bwd_push_fifo(const orc_graph* g, int32_t t, double alpha, double rmax, double* reserve, double* residue,
                          orc_stats* st) {
  /* Backward_Search.java:38-100 */
  uint32_t n = g->n;
  memset(reserve, 0, n * sizeof(double));
  memset(residue, 0, n * sizeof(double));
  if (deg_in(g, t) == 0) { /* :46-49 */
    reserve[t] = 1.0;
    return;
  }
  uint32_t cap = n + 1;
  int32_t* q = (int32_t*)xmalloc((size_t)cap * sizeof(int32_t));
  uint8_t* inq = (uint8_t*)xcalloc(n, 1);
  uint32_t head = 0, tail = 0;
  q[tail++] = t;
  inq[t] = 1;
  residue[t] = 1.0; /* :54-56 */
  while (head != tail) {
    int32_t v = q[head];
    head = (head + 1 == cap) ? 0 : head + 1;
    inq[v] = 0;
    double rc = residue[v];
    residue[v] = 0.0;
    reserve[v] = reserve[v] + rc * alpha; /* :63-66 */
    double avg = ((1.0 - alpha) * rc);    /* :72 */
    if (st) { st->pops++; st->edge_pushes += deg_in(g, v); }
    for (uint32_t e = g->in_rp[v]; e < g->in_rp[v + 1]; ++e) { /* :77-96 */
      int32_t u = g->in_ci[e];
      uint32_t du = deg_out(g, u);
      double nr = residue[u] + avg / du; /* :84-85 */
      residue[u] = nr;
      if (nr > rmax && !inq[u]) { /* :89 strict, un-normalised */
        q[tail] = u;
        tail = (tail + 1 == cap) ? 0 : tail + 1;
        inq[u] = 1;
        if (st) st->enqueues++;
      }
    }
  }
  free(q);
  free(inq);
}

static void bwd_push_sync(const orc_graph* g, int32_t t, double alpha, double rmax, double* reserve, double* residue,
                          sync_ws* w, orc_stats* st) {
  /* frontier-synchronous form of Backward_Search.java:38-100; reserve/residue must be zero on
   * entry for the nodes this target touches (the caller clears them). */
  if (deg_in(g, t) == 0) {
    reserve[t] = 1.0;
    return;
  }
  residue[t] = 1.0;
  w->cur[0] = t;
  w->ncur = 1;
  while (w->ncur) {
    uint64_t ef = 0;
    for (uint32_t i = 0; i < w->ncur; ++i) {
      int32_t v = w->cur[i];
      double rc = residue[v];
      residue[v] = 0.0;
      reserve[v] = reserve[v] + rc * alpha;
      w->contrib[i] = ((1.0 - alpha) * rc);
      ef += deg_in(g, v);
    }
    w->nnxt = 0;
    for (uint32_t i = 0; i < w->ncur; ++i) {
      int32_t v = w->cur[i];
      double c = w->contrib[i];
      for (uint32_t e = g->in_rp[v]; e < g->in_rp[v + 1]; ++e) {
        int32_t u = g->in_ci[e];
        double old = residue[u];
        double nr = old + c / deg_out(g, u);
        residue[u] = nr;
        if (!(old > rmax) && nr > rmax) w->nxt[w->nnxt++] = u;
      }
    }
    if (st) {
      st->levels++;
      st->pops += w->ncur;
      st->edge_pushes += ef;
      st->enqueues += w->nnxt;
    }
    int32_t* tmp = w->cur; w->cur = w->nxt; w->nxt = tmp;
    w->ncur = w->nnxt;
  }
}

void orc_backward_push(const orc_graph* g, int schedule, int32_t target, double alpha, double rmax, double* reserve,
                       double* residue, orc_stats* st) {
  if (st) memset(st, 0, sizeof *st);
  if (schedule == ORC_FIFO) {
    bwd_push_fifo(g, target, alpha, rmax, reserve, residue, st);
  } else {
    sync_ws w;
    sync_ws_init(&w, g->n);
    memset(reserve, 0, g->n * sizeof(double));
    memset(residue, 0, g->n * sizeof(double));
    bwd_push_sync(g, target, alpha, rmax, reserve, residue, &w, st);
    sync_ws_free(&w);
  }
  if (st) st->rmax_final = rmax;
}

/* ------------------------------------------------------------------ all-pair backward search (a9) */

typedef struct tv { int32_t t; double v; } tv;
static int cmp_tv_desc(const void* a, const void* b) {
  const tv* x = (const tv*)a; const tv* y = (const tv*)b;
  if (x->v != y->v) return (x->v < y->v) - (x->v > y->v);
  return (x->t > y->t) - (x->t < y->t); /* stable sort over target order == target ascending */
}

void orc_all_pair_backward(const orc_graph* g, int schedule, double alpha, double threshold, int k, uint32_t t_begin,
                           uint32_t t_end, uint64_t** offsets_out, int32_t** targets_out, double** values_out) {
  /* Base_Whole_Graph.java:58-164 */
  uint32_t n = g->n;
  double* reserve = (double*)xcalloc(n, sizeof(double));
  double* residue = (double*)xcalloc(n, sizeof(double));
  /* pass 1 collects (v, t, pi) triples in target order (:76-92) */
  size_t cap = 1024, cnt = 0;
  int32_t* tv_v = (int32_t*)xmalloc(cap * sizeof(int32_t));
  int32_t* tv_t = (int32_t*)xmalloc(cap * sizeof(int32_t));
  double* tv_p = (double*)xmalloc(cap * sizeof(double));
  uint64_t* deg = (uint64_t*)xcalloc((size_t)n + 1, sizeof(uint64_t));
  for (uint32_t t = t_begin; t < t_end; ++t) {
    orc_backward_push(g, schedule, (int32_t)t, alpha, threshold, reserve, residue, NULL); /* :68,78 */
    for (uint32_t v = 0; v < n; ++v) {
      double pi = reserve[v];
      if (pi > 0.0 && pi >= threshold) { /* :83 (only entries of the reserve map exist) */
        if (cnt == cap) {
          cap *= 2;
          tv_v = (int32_t*)realloc(tv_v, cap * sizeof(int32_t));
          tv_t = (int32_t*)realloc(tv_t, cap * sizeof(int32_t));
          tv_p = (double*)realloc(tv_p, cap * sizeof(double));
          if (!tv_v || !tv_t || !tv_p) abort();
        }
        tv_v[cnt] = (int32_t)v; tv_t[cnt] = (int32_t)t; tv_p[cnt] = pi;
        cnt++;
        deg[v + 1]++;
      }
    }
  }
  for (uint32_t v = 0; v < n; ++v) deg[v + 1] += deg[v];
  tv* rows = (tv*)xmalloc(cnt * sizeof(tv));
  uint64_t* fill = (uint64_t*)xmalloc((size_t)n * sizeof(uint64_t));
  for (uint32_t v = 0; v < n; ++v) fill[v] = deg[v];
  for (size_t i = 0; i < cnt; ++i) { /* stable: rows of v stay in target order (:38 LinkedHashMap) */
    uint64_t p = fill[tv_v[i]]++;
    rows[p].t = tv_t[i];
    rows[p].v = tv_p[i];
  }
  free(tv_v); free(tv_t); free(tv_p); free(fill);
  /* pass 2: per source, optional top-k trim + sort (:112-163) */
  uint64_t* off = (uint64_t*)xcalloc((size_t)n + 1, sizeof(uint64_t));
  int32_t* ot = (int32_t*)xmalloc(cnt * sizeof(int32_t));
  double* ov = (double*)xmalloc(cnt * sizeof(double));
  uint64_t w = 0;
  for (uint32_t v = 0; v < n; ++v) {
    off[v] = w;
    uint64_t b = deg[v], e = deg[v + 1];
    if (k < 0) { /* :117-127 */
      for (uint64_t i = b; i < e; ++i) { ot[w] = rows[i].t; ov[w] = rows[i].v; w++; }
    } else { /* :133-148 */
      uint64_t len = e - b;
      int have = 0;
      double kth = 0.0;
      if (k >= 1 && (uint64_t)k <= len) {
        double* tmp = (double*)xmalloc(len * sizeof(double));
        for (uint64_t i = 0; i < len; ++i) tmp[i] = rows[b + i].v;
        qsort(tmp, len, sizeof(double), cmp_desc);
        kth = tmp[k - 1];
        have = 1;
        free(tmp);
      }
      uint64_t w0 = w;
      for (uint64_t i = b; i < e; ++i)
        if (!have || rows[i].v >= kth) { ot[w] = rows[i].t; ov[w] = rows[i].v; w++; }
      tv* seg = (tv*)xmalloc((w - w0) * sizeof(tv));
      for (uint64_t i = w0; i < w; ++i) { seg[i - w0].t = ot[i]; seg[i - w0].v = ov[i]; }
      qsort(seg, w - w0, sizeof(tv), cmp_tv_desc);
      for (uint64_t i = w0; i < w; ++i) { ot[i] = seg[i - w0].t; ov[i] = seg[i - w0].v; }
      free(seg);
    }
  }
  off[n] = w;
  free(rows); free(deg); free(reserve); free(residue);
  *offsets_out = off;
  *targets_out = ot;
  *values_out = ov;
}


/* ------------------------------------------------------------------ CPU baseline (bench.py cpu_baseline leg) */

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

void orc_fora_whole_baseline(const orc_graph* g, int32_t src, double eps, const orc_conf* c, uint64_t seed,
                             uint64_t walk_divisor, int max_rounds, double* push_s, double* walk_s, double* reserve,
                             orc_stats* st) {
  /* Fora_Whole_Graph.java:82-146 exactly as written: FIFO pushes restarted from scratch while the
   * measured push time is below 400 ns * rsum * omega (:35,75-79,93-103), then the walks.  Only
   * every walk_divisor-th walk is run (and timed) so that a bench sample stays bounded; the
   * caller scales walk_s by walk_divisor.  max_rounds > 0 also caps the push loop (a run the
   * reference's clock can produce as well, with more walks in exchange). */
  uint32_t n = g->n;
  orc_stats local;
  if (!st) st = &local;
  memset(st, 0, sizeof *st);
  double alpha = c->alpha, rsum_local = c->rsum, rmax_local, omega_local;
  orc_fora_whole_params(c, eps, &rmax_local, &omega_local);
  double* residue = (double*)xcalloc(n, sizeof(double));
  double dur_ns = 0.0;
  int rounds = 0;
  while (dur_ns < 400.0 * rsum_local * omega_local && (max_rounds <= 0 || rounds < max_rounds)) {
    orc_stats ps;
    memset(&ps, 0, sizeof ps);
    double t0 = now_s();
    double fp_rsum = fwd_push_fifo(g, src, alpha, rmax_local, reserve, residue, &ps);
    dur_ns += (now_s() - t0) * 1e9;
    st->pops += ps.pops; st->edge_pushes += ps.edge_pushes; st->enqueues += ps.enqueues;
    rsum_local = fp_rsum * (1 - alpha);
    st->rmax_final = rmax_local;
    rmax_local /= 2.0;
    rounds++;
  }
  *push_s = dur_ns * 1e-9;
  st->rounds = (uint32_t)rounds;
  st->rsum = rsum_local;
  st->omega = omega_local;
  double t0 = now_s();
  double nrw_d = omega_local * rsum_local;
  long long nrw = (nrw_d == nrw_d) ? (long long)nrw_d : 0;
  uint64_t counter = 0;
  if (walk_divisor == 0) walk_divisor = 1;
  for (uint32_t v = 0; v < n; ++v) {
    double r = residue[v];
    if (!(r > 0.0)) continue;
    reserve[v] = reserve[v] + r * alpha;
    r *= (1.0 - alpha);
    if (nrw <= 0 || !(rsum_local > 0.0)) continue;
    long long omega_i = (long long)ceil(r / rsum_local * (double)nrw);
    double a_i = r / rsum_local * (double)nrw / (double)omega_i;
    double incr = a_i / (double)nrw * rsum_local;
    for (long long j = 0; j < omega_i; ++j) {
      if (counter++ % walk_divisor) continue;
      uint32_t steps;
      int32_t t = orc_random_walk(g, (int32_t)v, alpha, seed, 0, (uint64_t)j, 1, &steps);
      reserve[t] = reserve[t] + incr;
      st->walks++;
      st->walk_steps += steps;
    }
  }
  *walk_s = now_s() - t0;
  free(residue);
}

/* ------------------------------------------------------------------ error metrics */

double orc_max_err(const double* est, const double* exact, uint32_t n) {
  /* Gen_Util.java:306-321: over the ground-truth map's entries */
  double m = 0.0;
  for (uint32_t i = 0; i < n; ++i) {
    if (!(exact[i] > 0.0)) continue;
    double e = fabs(est[i] - exact[i]);
    if (e > m) m = e;
  }
  return m;
}

double orc_precision(const int32_t* algo_ids, int n_algo, const int32_t* gnd_ids, int n_gnd) {
  /* Gen_Util.java:271-279 */
  double hit = 0.0;
  for (int i = 0; i < n_algo; ++i)
    for (int j = 0; j < n_gnd; ++j)
      if (algo_ids[i] == gnd_ids[j]) { hit++; break; }
  return hit / (double)n_gnd;
}

double orc_ndcg(const int32_t* algo_ids, int n_algo, const int32_t* gnd_ids, int n_gnd, const double* exact) {
  /* Gen_Util.java:280-300 */
  double zk = 0.0, dcg = 0.0;
  for (int i = 1; i <= n_gnd; ++i) zk += (pow(2.0, exact[gnd_ids[i - 1]]) - 1.0) / log(i + 1.0) / log(2.0);
  for (int i = 1; i <= n_algo; ++i) {
    double p = 0.0;
    for (int j = 0; j < n_gnd; ++j)
      if (gnd_ids[j] == algo_ids[i - 1]) { p = exact[algo_ids[i - 1]]; break; }
    dcg += (pow(2.0, p) - 1.0) / log(i + 1.0) / log(2.0);
  }
  return dcg / zk;
}
