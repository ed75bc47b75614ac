/*
 * ppr_oracle.h — CPU restatement of the reference's PPR algorithms.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product (libpprhip.so) never links, loads or calls it.
 *
 * PARITY UNPINNED: the reference (joezie/Personalized-PageRank-Algorithms-on-Neo4j) ships no
 * tests, golden vectors or sample outputs, and cannot be run here (no JVM, un-vendored
 * neo4j-graph-algorithms dependency, syntax error at Fora_Topk.java:136).  The only published
 * known answers are the thesis' two-node closed form (Dissertation.pdf p.13-14) and its parameter
 * formulas; tests/test_oracle_*.py pin the oracle against those, against closed forms on cycles /
 * stars, and against the push invariant.  See DESIGN.md §3.
 *
 * Every function cites the reference lines it follows; paths are relative to
 * /root/reference/src/main/java/joezie/fora_neo4j/.
 *
 * Two schedules exist for every push:
 *   *_fifo  the reference's order: one FIFO queue, residues updated in place (Gauss-Seidel);
 *   *_sync  the frontier-synchronous order the HIP kernels use (Jacobi per level); same
 *           thresholds, same arithmetic per push, same invariant.  GPU parity is asserted
 *           against *_sync, and *_sync against *_fifo through the invariant / ground truth.
 */
#ifndef PPR_ORACLE_H
#define PPR_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_graph {
  uint32_t n;
  uint64_t m;
  const uint32_t* out_rp; /* n+1 */
  const int32_t* out_ci;  /* m   */
  const uint32_t* in_rp;  /* n+1 */
  const int32_t* in_ci;   /* m   */
} orc_graph;

typedef struct orc_stats {
  uint64_t pops, edge_pushes, enqueues, dead_end_pops, dense_nodes;
  uint32_t levels, dense_levels, rounds, pad;
  uint64_t mc_sources, walks, walk_steps;
  double rsum, rmax_final, omega, kth_value, model_cost_ns;
} orc_stats;

/* Mirror of pprhip_tuning_t (same defaults), so the twin takes the same round count. */
typedef struct orc_tuning {
  double c_walk_ns, c_edge_ns, c_pop_ns, c_level_ns, c_dense_edge_ns, c_dense_node_ns, dense_frac;
  int32_t max_rounds, max_halvings;
  double halving_ratio;
  int32_t prior_levels, gs_blocks;
  double gs_frac;
} orc_tuning;

typedef struct orc_conf { /* Algo_Conf.java:29-81 */
  double alpha, delta, pfail, rsum, min_delta;
  int32_t k;
  uint32_t n;
  uint64_t m;
} orc_conf;

#define ORC_FIFO 0
#define ORC_SYNC 1

void orc_tuning_default(orc_tuning* t);
/* Tuning used by the frontier-synchronous entry points that take none (orc_forward_push, top-k push rounds,
 * orc_fora_topk); NULL restores the defaults.  Level shapes matter to the twin since the Gauss-Seidel sweeps. */
void orc_set_sync_tuning(const orc_tuning* t);

/* Philox4x32-10 (Salmon et al., SC'11), the counter-based generator that replaces the
 * reference's unseeded ThreadLocalRandom (Monte_Carlo.java:76,84,111,115,123). */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

/* Algo_Conf.java:45-53 / 71-81 and Fora_Whole_Graph.java:86-87 / Fora_Topk.java:110-125,133. */
void orc_conf_fora_whole_graph(uint32_t n, uint64_t m, double alpha, orc_conf* c);
void orc_conf_fora_topk(uint32_t n, uint64_t m, int k, double alpha, orc_conf* c);
void orc_fora_whole_params(const orc_conf* c, double eps, double* rmax0, double* omega);
void orc_fora_topk_params(const orc_conf* c, double eps, double delta, double* min_rmax, double* rmax_scaled,
                          double* omega);

/* Power_Method.java:44-101 */
void orc_power_method(const orc_graph* g, int32_t src, double alpha, int iters, double* reserve /* n */);

/* Forward_Push.java:63-142.  Returns rsum as the reference leaves it (stale-high quirk at :140
 * in FIFO mode; exact sum of residues in SYNC mode). */
double orc_forward_push(const orc_graph* g, int schedule, int32_t src, double alpha, double rmax, double* reserve,
                        double* residue, orc_stats* st);

/* Forward_Push.java:144-250 as a resumable object. */
typedef struct orc_topk_push orc_topk_push;
orc_topk_push* orc_topk_push_new(const orc_graph* g, int schedule, int32_t src, double alpha);
double orc_topk_push_round(orc_topk_push* p, double min_rmax, double rmax, orc_stats* st);
const double* orc_topk_push_reserve(const orc_topk_push* p);
const double* orc_topk_push_residue(const orc_topk_push* p);
void orc_topk_push_free(orc_topk_push* p);

/* Monte_Carlo.java:60-94 (no_zero_hop = 0) and :96-133 (no_zero_hop = 1). */
int32_t orc_random_walk(const orc_graph* g, int32_t start, double alpha, uint64_t seed, uint32_t stream,
                        uint64_t walk_idx, int no_zero_hop, uint32_t* steps_out);

/* Fora_Whole_Graph.java:82-146.  n_rounds > 0: exactly that many threshold rounds; 0: the
 * deterministic cost model.  FIFO restarts every round from scratch (as the reference does);
 * SYNC resumes (as the HIP engine does). */
void orc_fora_whole(const orc_graph* g, int schedule, int32_t src, double eps, const orc_conf* c, uint64_t seed,
                    int n_rounds, const orc_tuning* tun, double* reserve, orc_stats* st);

/* The reference's clock-driven FORA (Fora_Whole_Graph.java:93-103 with its 400 ns constant and the
 * real clock), timed per phase; runs every walk_divisor-th walk only.  bench.py's cpu_baseline. */
void orc_fora_whole_baseline(const orc_graph* g, int32_t src, double eps, const orc_conf* c, uint64_t seed,
                             uint64_t walk_divisor, int max_rounds, double* push_s, double* walk_s, double* reserve,
                             orc_stats* st);

/* Fora_Topk.java:102-184 */
void orc_fora_topk(const orc_graph* g, int schedule, int32_t src, double eps, const orc_conf* c, uint64_t seed,
                   double* reserve, orc_stats* st);

/* Monte_Carlo.java:136-158 */
void orc_monte_carlo(const orc_graph* g, int32_t src, double eps, const orc_conf* c, uint64_t seed, double* ppr,
                     orc_stats* st);

/* Algo_Util.java:32-53: k-th largest of the values > 0 (entries of the reference's map).
 * Returns 0 and leaves *kth alone when fewer than k entries exist (the reference's null). */
int orc_kth_largest(const double* v, uint32_t n, int k, double* kth);
/* Fora_Topk.java:186-199 + 82-99 with the deterministic tie rule (value desc, id asc).
 * Returns the number selected (entries >= kth, may exceed k); writes at most cap. */
int orc_topk(const double* v, uint32_t n, int k, int32_t* ids, double* vals, int cap);

/* Backward_Search.java:38-100 */
void orc_backward_push(const orc_graph* g, int schedule, int32_t target, double alpha, double rmax, double* reserve,
                       double* residue, orc_stats* st);

/* Base_Whole_Graph.java:58-164 for targets [t_begin, t_end); returns malloc'd arrays (caller
 * frees with orc_free): offsets[n+1] by source, targets[], values[]. */
void orc_all_pair_backward(const orc_graph* g, int schedule, double alpha, double threshold, int k, uint32_t t_begin,
                           uint32_t t_end, uint64_t** offsets, int32_t** targets, double** values);
void orc_free(void* p);

/* Gen_Util.java:259-326 */
double orc_max_err(const double* est, const double* exact, uint32_t n);
double orc_precision(const int32_t* algo_ids, int n_algo, const int32_t* gnd_ids, int n_gnd);
double orc_ndcg(const int32_t* algo_ids, int n_algo, const int32_t* gnd_ids, int n_gnd, const double* exact);

#ifdef __cplusplus
}
#endif
#endif
