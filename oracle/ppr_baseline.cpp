/*
 * ppr_baseline.cpp — CPU baselines for bench.py's `cpu_baseline` leg.  TEST / BENCH INFRASTRUCTURE ONLY:
 * the product (libpprhip.so) never links, loads or calls it (same rule as ppr_oracle.c).
 *
 * SURVEY.md §8(d) asks for two CPU baselines beside the GPU number, because the Java reference cannot run
 * here (no JVM) and is not on the GPU box:
 *
 *   base_fora_hashmap        the reference's algorithm in the reference's data-structure shape, one thread:
 *                            HashMap<Long,Double> reserve / residue -> std::unordered_map<int64_t,double>,
 *                            ConcurrentLinkedQueue<Long> -> std::deque<int64_t>, HashSet<Long> -> std::unordered_set,
 *                            boxed ids translated per edge, and the clock-driven push loop with its 400 ns
 *                            constant (Forward_Push.java:63-142, Fora_Whole_Graph.java:35,75-79,82-146,
 *                            Monte_Carlo.java:96-133).  This is the "faithful" baseline: what a line-by-line
 *                            port of the Java would cost on this host.
 *   base_fora_array_parallel the strong baseline: the dense-array port (ppr_oracle.c's orc_fora_whole_baseline:
 *                            same algorithm, flat arrays instead of hash maps), one query per host thread over
 *                            all cores.
 *
 * Both run bounded samples (a time budget on the hash-map push, every walk_divisor-th walk) and report what
 * they ran, so that bench.py can scale to one whole query and say so.
 */
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

extern "C" {
#include "ppr_oracle.h"
}

namespace {

double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct HashPush {
  std::unordered_map<int64_t, double> reserve, residue;
  double rsum = 1.0;
  uint64_t pops = 0, edge_pushes = 0;
  bool truncated = false;
};

inline uint32_t out_degree(const orc_graph* g, int32_t v) { return g->out_rp[v + 1] - g->out_rp[v]; }

/* Forward_Push.computeWholeGraphPPR, Forward_Push.java:63-142, statement by statement in the Java's shape.
 * deadline_s > 0: give up (truncated = true) when the clock passes it; checked every 1024 pops. */
void forward_push_hashmap(const orc_graph* g, int64_t nodeId_start, double alpha, double rmax, HashPush& fp,
                          double deadline_s) {
  fp.residue.clear();  // :64-65
  fp.reserve.clear();
  double rsum_local = 1.0;                             // :68
  const int32_t nodeIdM_start = (int32_t)nodeId_start;  // toMappedNodeId (:69): ids are dense
  const uint32_t out_degree_start = out_degree(g, nodeIdM_start);
  if (out_degree_start == 0) {  // :72-76
    fp.reserve[nodeId_start] = 1.0;
    fp.rsum = 0.0;
    return;
  }
  std::unordered_set<int64_t> nodesInQueue;  // :78
  std::deque<int64_t> Q;                     // :79
  Q.push_back(nodeId_start);                 // :81-83
  nodesInQueue.insert(nodeId_start);
  fp.residue[nodeId_start] = 1.0;
  while (!Q.empty()) {  // :85
    const int64_t nodeId_cur = Q.front();
    Q.pop_front();
    nodesInQueue.erase(nodeId_cur);
    const double residue_cur = fp.residue[nodeId_cur];  // :88-89
    fp.residue[nodeId_cur] = 0.0;
    double old_reserve_cur = 0.0;  // :91-95
    auto it = fp.reserve.find(nodeId_cur);
    if (it != fp.reserve.end()) old_reserve_cur = it->second;
    fp.reserve[nodeId_cur] = old_reserve_cur + residue_cur * alpha;
    rsum_local -= residue_cur * alpha;  // :97
    const int32_t nodeIdM_cur = (int32_t)nodeId_cur;
    const uint32_t out_degree_cur = out_degree(g, nodeIdM_cur);
    fp.pops++;
    if ((fp.pops & 1023u) == 0 && deadline_s > 0.0 && now_s() > deadline_s) {
      fp.truncated = true;
      fp.rsum = rsum_local;
      return;
    }
    if (out_degree_cur == 0) {  // :101-115
      const double new_residue_start = fp.residue[nodeId_start] + residue_cur * (1.0 - alpha);
      fp.residue[nodeId_start] = new_residue_start;
      if (out_degree_start > 0 && new_residue_start / (double)out_degree_start >= rmax &&
          !nodesInQueue.count(nodeId_start)) {
        Q.push_back(nodeId_start);
        nodesInQueue.insert(nodeId_start);
      }
      continue;  // skips the rsum update at :140 (the reference's stale-rsum quirk)
    }
    const double avg_push_residue = ((1.0 - alpha) * residue_cur) / (double)out_degree_cur;  // :117
    for (uint32_t e = g->out_rp[nodeIdM_cur]; e < g->out_rp[nodeIdM_cur + 1]; ++e) {           // :119-139
      const int64_t nodeId2 = (int64_t)g->out_ci[e];  // toOriginalNodeId
      double old_residue_next = 0.0;
      auto jt = fp.residue.find(nodeId2);
      if (jt != fp.residue.end()) old_residue_next = jt->second;
      const double new_residue_next = old_residue_next + avg_push_residue;
      fp.residue[nodeId2] = new_residue_next;
      const uint32_t out_degree_next = out_degree(g, (int32_t)nodeId2);
      if (new_residue_next / (double)out_degree_next >= rmax && !nodesInQueue.count(nodeId2)) {
        Q.push_back(nodeId2);
        nodesInQueue.insert(nodeId2);
      }
      fp.edge_pushes++;
    }
    fp.rsum = rsum_local;  // :140
  }
}

}  // namespace

extern "C" {

/* Fora_Whole_Graph.computeWholeGraphPPR (Fora_Whole_Graph.java:82-146) in the Java's data-structure shape.
 * push_budget_s bounds the push loop: when the clock-driven loop has not ended by then, the round in flight is
 * cut (truncated = 1) and the walks run from the last completed round's residues (or are skipped when no round
 * completed).  Only every walk_divisor-th walk is run.  Outputs: seconds spent pushing / walking, edge pushes
 * and pops done, walks run and the walk total the residues ask for, completed rounds. */
void base_fora_hashmap(const orc_graph* g, int32_t src, double eps, const orc_conf* c, uint64_t seed,
                       uint64_t walk_divisor, double push_budget_s, double* push_s, double* walk_s,
                       uint64_t* edge_pushes, uint64_t* pops, uint64_t* walks_run, uint64_t* walks_total, int* rounds,
                       int* truncated, double* reserve_out /* n doubles or NULL */) {
  const double alpha = c->alpha;
  double rsum_local = c->rsum, rmax_local, omega_local;
  orc_fora_whole_params(c, eps, &rmax_local, &omega_local);  // :86-87
  HashPush fp, done;
  bool have_done = false;
  double dur_ns = 0.0;
  int n_rounds = 0;
  *truncated = 0;
  const double t_begin = now_s();
  while (dur_ns < 400.0 * rsum_local * omega_local) {  // :93 (avg_rand_walk_time = 400 ns, :35)
    const double t0 = now_s();
    forward_push_hashmap(g, (int64_t)src, alpha, rmax_local, fp, push_budget_s > 0 ? t_begin + push_budget_s : 0.0);
    dur_ns += (now_s() - t0) * 1e9;
    if (fp.truncated) {
      *truncated = 1;
      break;
    }
    rsum_local = fp.rsum * (1 - alpha);  // :101
    rmax_local /= 2.0;                   // :102
    n_rounds++;
    done.reserve = fp.reserve;  // :108-109 copies
    done.residue = fp.residue;
    have_done = true;
  }
  *push_s = dur_ns * 1e-9;
  *edge_pushes = fp.edge_pushes;
  *pops = fp.pops;
  *rounds = n_rounds;
  *walks_run = 0;
  *walks_total = 0;
  *walk_s = 0.0;
  if (!have_done) {  // cut inside the first round: the walk phase still runs, from the state the push was cut in
    done.reserve = fp.reserve;  // (any state of a push is a valid (reserve, residue) pair; only the rate is used)
    done.residue = fp.residue;
    rsum_local = fp.rsum * (1 - alpha);
  }
  std::unordered_map<int64_t, double>& reserve = done.reserve;
  const double t0 = now_s();
  const double nrw_d = omega_local * rsum_local;  // :112-113
  const long long nrw = (nrw_d == nrw_d) ? (long long)nrw_d : 0;
  uint64_t counter = 0;
  if (walk_divisor == 0) walk_divisor = 1;
  for (const auto& kv : done.residue) {  // :119-140 (HashMap iteration order)
    const int64_t v = kv.first;
    double r = kv.second;
    if (!(r > 0.0)) continue;
    reserve[v] = reserve[v] + r * alpha;
    r *= (1.0 - alpha);
    if (nrw <= 0 || !(rsum_local > 0.0)) continue;
    const long long omega_i = (long long)std::ceil(r / rsum_local * (double)nrw);
    const double a_i = r / rsum_local * (double)nrw / (double)omega_i;
    const double incr = a_i / (double)nrw * rsum_local;
    *walks_total += (uint64_t)omega_i;
    for (long long j = 0; j < omega_i; ++j) {
      if (counter++ % walk_divisor) continue;
      uint32_t steps;
      const int32_t t = orc_random_walk(g, (int32_t)v, alpha, seed, 0, (uint64_t)j, 1, &steps);  // Monte_Carlo.java:96-133
      reserve[(int64_t)t] += incr;  // :134-139
      (*walks_run)++;
    }
  }
  *walk_s = now_s() - t0;
  if (reserve_out) {
    std::memset(reserve_out, 0, sizeof(double) * g->n);
    for (const auto& kv : reserve) reserve_out[kv.first] = kv.second;
  }
}

/* The dense-array port (orc_fora_whole_baseline: the reference's clock-driven FORA on flat arrays) for q sources,
 * one query per thread over `threads` host threads (sources are handed out dynamically).  per_query_s[i] =
 * push seconds + walk seconds * walk_divisor of source i (its time scaled to all of its walks); *wall_s = wall
 * time of the whole run with only every walk_divisor-th walk run. */
void base_fora_array_parallel(const orc_graph* g, const int32_t* srcs, int q, double eps, const orc_conf* c,
                              uint64_t seed, uint64_t walk_divisor, int threads, double* wall_s, double* per_query_s,
                              uint64_t* edge_pushes_sum) {
  if (threads < 1) threads = 1;
  std::vector<std::thread> pool;
  std::vector<uint64_t> ep((size_t)q, 0);
  int next = 0;
  std::mutex* mu = new std::mutex();
  const double t0 = now_s();
  auto work = [&]() {
    std::vector<double> reserve(g->n);
    for (;;) {
      int i;
      {
        std::lock_guard<std::mutex> lk(*mu);
        i = next++;
      }
      if (i >= q) break;
      std::fill(reserve.begin(), reserve.end(), 0.0);
      double ps = 0.0, ws = 0.0;
      orc_stats st;
      orc_fora_whole_baseline(g, srcs[i], eps, c, seed, walk_divisor, 0, &ps, &ws, reserve.data(), &st);
      per_query_s[i] = ps + ws * (double)(walk_divisor ? walk_divisor : 1);
      ep[i] = st.edge_pushes;
    }
  };
  for (int t = 1; t < threads; ++t) pool.emplace_back(work);
  work();
  for (auto& t : pool) t.join();
  *wall_s = now_s() - t0;
  delete mu;
  uint64_t s = 0;
  for (uint64_t x : ep) s += x;
  if (edge_pushes_sum) *edge_pushes_sum = s;
}

/* The dense-array port once more, one source on the calling thread, with what every turn of the clock-driven
 * loop did (Fora_Whole_Graph.java:93-103): seconds, edge pushes and the rsum it left, for at most max_rec turns.
 * bench.py uses the per-turn work to scale the hash-map port's measured rates to one whole query.  Returns the
 * number of turns the loop took. */
int base_fora_array_rounds(const orc_graph* g, int32_t src, double eps, const orc_conf* c, uint64_t seed,
                           uint64_t walk_divisor, int max_rec, double* round_push_s, uint64_t* round_edge_pushes,
                           double* round_rsum, double* walk_s, uint64_t* walks_run, uint64_t* walks_total) {
  const uint32_t n = g->n;
  const double alpha = c->alpha;
  double rsum_local = c->rsum, rmax_local, omega_local;
  orc_fora_whole_params(c, eps, &rmax_local, &omega_local);
  std::vector<double> reserve(n), residue(n);
  double dur_ns = 0.0;
  int rounds = 0;
  while (dur_ns < 400.0 * rsum_local * omega_local) {
    orc_stats ps;
    std::memset(&ps, 0, sizeof ps);
    const double t0 = now_s();
    const double fp_rsum = orc_forward_push(g, ORC_FIFO, src, alpha, rmax_local, reserve.data(), residue.data(), &ps);
    const double dt = now_s() - t0;
    dur_ns += dt * 1e9;
    rsum_local = fp_rsum * (1 - alpha);
    rmax_local /= 2.0;
    if (rounds < max_rec) {
      round_push_s[rounds] = dt;
      round_edge_pushes[rounds] = ps.edge_pushes;
      round_rsum[rounds] = rsum_local;
    }
    rounds++;
  }
  const double t0 = now_s();
  const double nrw_d = omega_local * rsum_local;
  const long long nrw = (nrw_d == nrw_d) ? (long long)nrw_d : 0;
  uint64_t counter = 0;
  *walks_run = 0;
  *walks_total = 0;
  if (walk_divisor == 0) walk_divisor = 1;
  for (uint32_t v = 0; v < n; ++v) {  // Fora_Whole_Graph.java:119-140
    double r = residue[v];
    if (!(r > 0.0)) continue;
    reserve[v] = reserve[v] + r * alpha;
    r *= (1.0 - alpha);
    if (nrw <= 0 || !(rsum_local > 0.0)) continue;
    const long long omega_i = (long long)std::ceil(r / rsum_local * (double)nrw);
    const double a_i = r / rsum_local * (double)nrw / (double)omega_i;
    const double incr = a_i / (double)nrw * rsum_local;
    *walks_total += (uint64_t)omega_i;
    for (long long j = 0; j < omega_i; ++j) {
      if (counter++ % walk_divisor) continue;
      uint32_t steps;
      const int32_t t = orc_random_walk(g, (int32_t)v, alpha, seed, 0, (uint64_t)j, 1, &steps);
      reserve[t] = reserve[t] + incr;
      (*walks_run)++;
    }
  }
  *walk_s = now_s() - t0;
  return rounds;
}

int base_hardware_threads(void) { return (int)std::thread::hardware_concurrency(); }

}  // extern "C"
