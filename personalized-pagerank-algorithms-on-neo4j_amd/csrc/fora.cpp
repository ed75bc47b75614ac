// fora.cpp — FORA, FORA top-k and backward searches as resumable runs, and the batched entry points
// that keep kBatch of them in flight on the handle's slots (dense levels share one sweep).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <thread>

#include "engine_internal.hpp"

using namespace pprhip;
using namespace pprhip::detail;

// ------------------------------------------------------------------ FORA whole graph (a5)
namespace pprhip {

// One FORA query as a resumable run: step() advances it until it is finished or (yield_dense)
// until its next level is dense, so that the batch driver can run that level for many queries
// in one sweep.  pprhip_fora_single_source drives the same code without yielding.
struct ForaRun {
  pprhip_graph* g = nullptr;
  int32_t src = 0;  // internal id
  const pprhip_fora_conf_t* conf = nullptr;
  uint64_t seed = 0;
  int n_rounds = 0;
  CallTimer* tm = nullptr;  // single-query calls: push / walk phase marks
  pprhip_stats_t st;
  double alpha = 0, rsum_local = 0, rmax_local = 0, omega_local = 0, rmax_used = 0, model_cost = 0;
  int rounds = 0;
  bool dead_src = false;
  LevelCtx L;
  PushArgs a;
  RoundCut cut;
  enum Phase { kRoundStart, kLevels, kWalks, kWalkWait, kTopkRoundStart, kTopkLevels, kTopkFinal, kBwdLevels, kBwdFinal, kDone } phase = kDone;
  hipStream_t side = nullptr;  // batch driver: the walk phase goes to this stream and the run yields until it has ended
  int query = -1;  // batch driver: index of the query this run serves
  detail::BatchJob* job = nullptr;  // ... and the call (or stream submission) that query belongs to
  bool waiting = false;
  bool in_push = false;  // between a push phase's start and its end (BatchSync: may hold sweeps off)
  // top-k runs (Fora_Topk.computeTopKPPR, kind 1): the trial-and-error loop on delta
  int kind = 0;
  double eps_half = 0, delta_local = 0, min_delta = 0, min_rmax = 0;
  uint32_t round = 0;
  int cap = 0, nsel = 0;
  int32_t* ids_out = nullptr;
  double* vals_out = nullptr;
  // backward searches of All-Pair (kind 2): entries >= threshold of the finished search
  int32_t target_orig = -1;
  std::vector<Triple> triples;
};

}  // namespace pprhip

namespace {

constexpr uint32_t kSideWalkWavesDefault = 4;  // waves per CU of a walk phase that runs beside sweeps (batch_sequential)
static uint32_t side_walk_waves() {  // PPRHIP_SIDE_WALK_WAVES: measurement switch
  static const uint32_t v = [] {
    const char* e = hook_env("PPRHIP_SIDE_WALK_WAVES");
    const long x = e ? atol(e) : 0;
    return x > 0 && x <= 32 ? (uint32_t)x : kSideWalkWavesDefault;
  }();
  return v;
}
#define kSideWalkWaves side_walk_waves()

void leave_push(ForaRun& r) {
  if (r.in_push) {
    r.in_push = false;
    if (r.g->sync) r.g->sync->release(r.g->slot_index);
  }
}

int fora_begin(ForaRun& r, pprhip_graph* g, int32_t src_internal, double eps, const pprhip_fora_conf_t* conf,
               uint64_t seed, int n_rounds) {
  r.g = g;
  r.src = src_internal;
  r.conf = conf;
  r.seed = seed;
  r.n_rounds = n_rounds;
  std::memset(&r.st, 0, sizeof r.st);
  g->topk_active = false;
  PPRHIP_TRY(reset_query_state(g, false, src_internal));
  r.alpha = conf->alpha;
  r.rsum_local = conf->rsum;
  PPRHIP_TRY(pprhip_fora_whole_params(conf, eps, &r.rmax_local, &r.omega_local));  // Fora_Whole_Graph.java:86-87
  if (n_rounds == 0 && g->tun.prior_levels > 0 && g->tun.halving_ratio > 1.0) {
    // Loop turns that are known to pass before any push: after a push at rmax every r(v) < rmax * d(v), so
    // rsum <= rmax * m and the walks cost at most c_walk * omega * (1 - alpha) * rmax * m; while that bound still
    // covers prior_levels dense levels the turn would be repeated at half the threshold anyway (twin: same rule).
    const pprhip_tuning_t& t = g->tun;
    double walk_bound = t.c_walk_ns * r.omega_local * (1 - r.alpha) * r.rmax_local * (double)g->m;
    const double push_est =
        (double)t.prior_levels * (t.c_level_ns + t.c_dense_edge_ns * (double)g->m + t.c_dense_node_ns * (double)g->n);
    for (int h = 0; h < t.max_halvings && walk_bound >= push_est; ++h) {
      walk_bound /= 2.0;
      r.rmax_local /= 2.0;
    }
  }
  r.rmax_used = r.rmax_local;
  r.model_cost = 0.0;
  r.rounds = 0;
  r.dead_src = hdeg_out(g, src_internal) == 0;
  r.L = LevelCtx();
  r.phase = ForaRun::kRoundStart;
  r.waiting = false;
  r.in_push = true;
  return PPRHIP_OK;
}

int fora_step(ForaRun& r, bool yield_dense) {
  pprhip_graph* g = r.g;
  for (;;) {
    if (r.phase == ForaRun::kRoundStart) {  // Fora_Whole_Graph.java:93-103, clock replaced by the level cost model
      const bool more = r.n_rounds > 0 ? r.rounds < r.n_rounds
                                       : (r.model_cost < g->tun.c_walk_ns * r.rsum_local * r.omega_local &&
                                          r.rounds < g->tun.max_rounds);
      if (!more) {
        r.phase = ForaRun::kWalks;
        continue;
      }
      if (r.dead_src) {  // Forward_Push.java:72-76
        PPRHIP_TRY(launch_set_f64(g, g->reserve, (uint32_t)r.src, 1.0));
        r.rsum_local = 0.0;
        r.rmax_used = r.rmax_local;
        r.rounds++;
        r.phase = ForaRun::kWalks;
        continue;
      }
      r.a = PushArgs{r.alpha, r.rmax_local, 0.0, r.src, kFwdWhole};
      r.cut = RoundCut();
      r.cut.fixed = r.n_rounds > 0;
      r.cut.enabled = r.n_rounds > 0 ? r.rounds + 1 < r.n_rounds : r.rounds + 1 < g->tun.max_rounds;
      r.cut.omega = r.omega_local;
      r.cut.c_walk = g->tun.c_walk_ns;
      r.cut.alpha = r.alpha;
      if (r.rounds == 0) {
        PPRHIP_TRY(launch_set_f64(g, g->residue, (uint32_t)r.src, 1.0));
        PPRHIP_TRY(seed_single(g, r.L, r.src, hdeg_out(g, r.src)));
      } else {
        PPRHIP_TRY(seed_scan(g, r.a, 0, r.L));
      }
      r.phase = ForaRun::kLevels;
    }
    if (r.phase == ForaRun::kLevels) {
      const int rc = run_levels(g, r.a, r.L, r.st, &r.model_cost, yield_dense, &r.cut);
      if (rc != PPRHIP_OK) return rc;  // kYield or an error
      if (r.cut.taken && !r.cut.fixed) {
        r.rsum_local = r.cut.rsum;  // measured when the round was cut; nothing was pushed since
      } else {
        double sum = 0.0;
        PPRHIP_TRY(device_sum(g, g->residue, &sum));
        r.rsum_local = sum * (1 - r.alpha);  // :101 (rsum is the exact residue sum here)
      }
      r.rmax_used = r.rmax_local;
      r.rmax_local /= 2.0;  // :102
      // The reference's loop would turn again (and restart the push from scratch at half the threshold) as
      // long as the push stays cheaper than the walks; when the walks outweigh the push so far by ratio^k, k
      // further halvings are taken at once instead of pushing at every threshold between (the twin does the same).
      if (r.n_rounds == 0 && r.model_cost > 0.0 && g->tun.halving_ratio > 1.0) {
        double ratio = g->tun.c_walk_ns * r.rsum_local * r.omega_local / r.model_cost;
        for (int h = 1; ratio >= g->tun.halving_ratio && h < g->tun.max_halvings; ++h) {
          ratio /= g->tun.halving_ratio;
          r.rmax_local /= 2.0;
        }
      }
      r.rounds++;
      r.phase = (r.n_rounds > 0 && !(r.rsum_local > 0.0)) ? ForaRun::kWalks : ForaRun::kRoundStart;
      continue;
    }
    if (r.phase == ForaRun::kWalks) {
      leave_push(r);
      if (r.tm) r.tm->mark(1);
      // Fora_Whole_Graph.java:112-140
      const double nrw_d = r.omega_local * r.rsum_local;
      const long long nrw = (nrw_d == nrw_d && nrw_d > 0.0) ? (long long)nrw_d : 0;
      if (!r.dead_src && r.side) {
        // the walk phase beside the other queries' sweeps: plan and walks on the side stream, behind everything this
        // query has queued on the compute stream; the driver calls again when walk_ev[2] has passed
        PPRHIP_CHECK_HIP(hipEventRecord(g->walk_ev[0], g->stream));
        PPRHIP_CHECK_HIP(hipStreamWaitEvent(r.side, g->walk_ev[0], 0));
        hipStream_t own = g->stream;
        KernelTimer* const tsave = g_timer_cur;
        KernelTimer quiet;
        quiet.off = true;  // (timed by the events below: the caller's timer watches the compute stream)
        g_timer_cur = &quiet;
        g->stream = r.side;
        g->walk_waves = kSideWalkWaves;
        int rc = hipEventRecord(g->walk_ev[1], r.side) == hipSuccess ? PPRHIP_OK : PPRHIP_ERR_HIP;
        if (rc == PPRHIP_OK) rc = run_walk_phase(g, 0, r.alpha, r.rsum_local, nrw, r.seed, 0, g->reserve, r.st);
        if (rc == PPRHIP_OK && hipEventRecord(g->walk_ev[2], r.side) != hipSuccess) rc = PPRHIP_ERR_HIP;
        g->stream = own;
        g->walk_waves = 0;
        g_timer_cur = tsave;
        PPRHIP_TRY(rc);
        r.phase = ForaRun::kWalkWait;
        return kYieldWalk;
      }
      if (!r.dead_src) {
        // (a worker's walks run beside the other slots' sweeps as well: the same narrow grid as on the side stream)
        if (g->sync && !hook_env("PPRHIP_WORKER_WALK_WIDE")) g->walk_waves = kSideWalkWaves;
        const int rc = run_walk_phase(g, 0, r.alpha, r.rsum_local, nrw, r.seed, 0, g->reserve, r.st);
        g->walk_waves = 0;
        PPRHIP_TRY(rc);
      }
      r.phase = ForaRun::kWalkWait;
    }
    if (r.phase == ForaRun::kWalkWait) {
      if (!r.dead_src && r.side) {
        PPRHIP_CHECK_HIP(hipEventSynchronize(g->walk_ev[2]));  // (the driver has seen it pass)
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, g->walk_ev[1], g->walk_ev[2]) == hipSuccess) {
          ktimer().acc_ms[PPRHIP_KERNEL_WALK] += (double)ms;  // plan + walks, as one launch of the class
          ktimer().acc_cnt[PPRHIP_KERNEL_WALK]++;
        }
      }
      if (r.tm) r.tm->mark(2);
      PPRHIP_TRY(read_dead_pops(g, r.st));  // one read-back for the push's dead-end pops and the walks' steps
      r.st.rounds = (uint32_t)r.rounds;
      r.st.rsum = r.rsum_local;
      r.st.rmax_final = r.rmax_used;
      r.st.omega = r.omega_local;
      r.phase = ForaRun::kDone;
    }
    return PPRHIP_OK;
  }
}

// Fora_Topk.computeTopKPPR (Fora_Topk.java:102-184) as a resumable run; the same sequence as
// pprhip_fora_topk, which keeps the per-phase timing of a single call.
int topk_begin(ForaRun& r, pprhip_graph* g, int32_t src_internal, double eps, const pprhip_fora_conf_t* conf,
               uint64_t seed, int32_t* ids_out, double* vals_out, int cap) {
  r.g = g;
  r.kind = 1;
  r.src = src_internal;
  r.conf = conf;
  r.seed = seed;
  std::memset(&r.st, 0, sizeof r.st);
  PPRHIP_TRY(reset_query_state(g, true, src_internal));
  PPRHIP_CHECK_HIP(hipMemsetAsync(g->flags + src_internal, 1, 1, g->stream));  // Q = {s} (:117-118)
  g->topk_active = true;
  g->topk_first = true;
  g->topk_src = src_internal;
  g->topk_alpha = conf->alpha;
  g->topk_rsum = conf->rsum;
  r.alpha = conf->alpha;
  r.eps_half = eps * 0.5;  // :109-110
  r.delta_local = conf->delta;
  r.min_delta = conf->min_delta;
  r.min_rmax = r.eps_half * std::sqrt(r.min_delta / 3 / (double)conf->m / std::log(2 / conf->pfail));  // :113
  r.rsum_local = conf->rsum;
  r.omega_local = r.rmax_local = 0.0;
  r.round = 0;
  r.dead_src = false;
  r.ids_out = ids_out;
  r.vals_out = vals_out;
  r.cap = cap;
  r.nsel = 0;
  r.phase = ForaRun::kTopkRoundStart;
  r.waiting = false;
  r.in_push = false;
  return PPRHIP_OK;
}

int topk_step(ForaRun& r, bool yield_dense) {
  pprhip_graph* g = r.g;
  const pprhip_fora_conf_t* conf = r.conf;
  const size_t nd = sizeof(double) * (size_t)act_n(g);  // (est beyond the query's scan bound is zero and stays so)
  for (;;) {
    if (r.phase == ForaRun::kTopkRoundStart) {
      if (!(r.delta_local >= r.min_delta)) {  // :123
        r.phase = ForaRun::kTopkFinal;
        continue;
      }
      r.rmax_local = r.eps_half * std::sqrt(r.delta_local / 3.0 / (double)conf->m / std::log(2.0 / conf->pfail));  // :124
      r.omega_local = (r.eps_half + 2.0) * std::log(2.0 / conf->pfail) / r.eps_half / r.eps_half / r.delta_local;  // :125
      if (hdeg_out(g, r.src) == 0) {  // :126-132
        PPRHIP_CHECK_HIP(hipMemsetAsync(g->est, 0, nd, g->stream));
        PPRHIP_TRY(launch_set_f64(g, g->est, (uint32_t)r.src, 1.0));
        r.rsum_local = 0.0;
        r.dead_src = true;
        r.phase = ForaRun::kTopkFinal;
        continue;
      }
      r.rmax_local *= std::sqrt((double)conf->m * r.rmax_local) * 3.0;  // :133
      // forward_push_topk (:137; Forward_Push.java:144-250)
      if (g->topk_first) PPRHIP_TRY(launch_set_f64(g, g->residue, (uint32_t)r.src, 1.0));
      r.a = PushArgs{r.alpha, r.rmax_local, r.min_rmax, r.src, kFwdTopk};
      r.L = LevelCtx();
      r.in_push = true;
      PPRHIP_TRY(seed_scan(g, r.a, 1, r.L));
      r.phase = ForaRun::kTopkLevels;
    }
    if (r.phase == ForaRun::kTopkLevels) {
      const int rc = run_levels(g, r.a, r.L, r.st, nullptr, yield_dense);
      if (rc != PPRHIP_OK) return rc;  // kYield or an error
      leave_push(r);
      // :142-168 without a host round trip: the residue sum stays on the device, where the walk plan derives rsum and
      // the walk budget from it (:148,151) and the walk kernel reads the plan's counts; the sum reaches the host with
      // the selection's read-back
      PPRHIP_TRY(launch_sum_partial(g, g->residue, act_n(g)));  // (the plan adds the partial sums up)
      g->topk_first = false;
      // :143 the estimate := copy of the push reserve (walk increments of earlier rounds are dropped), taken in the
      // plan's pass over the same range; :155-168 the walks
      PPRHIP_TRY(launch_walk_plan(g, 1, r.alpha, 0.0, 0, g->est, r.omega_local, g->reserve, g->est));
      PPRHIP_TRY(launch_walk_run(g, 1, r.alpha, r.seed, r.round, g->est));
      r.round++;
      double kth = 0.0;
      bool have = false;
      int nsel = 0;
      PPRHIP_TRY(select_topk(g, g->est, conf->k, nullptr, nullptr, 0, &nsel, &kth, &have, r.st, true));  // :173
      g->topk_rsum = g->sel_plan_sum;  // (the sum the round's plan was derived from, in the selection's header)
      r.rsum_local = g->topk_rsum;       // :142
      if (!have) kth = 0.0;                                                                          // :174
      r.st.kth_value = kth;
      if (kth >= (1 + r.eps_half) * r.delta_local || r.delta_local <= r.min_delta) {  // :175-176
        r.phase = ForaRun::kTopkFinal;
      } else {
        r.delta_local = std::max(r.min_delta, r.delta_local / 4.0);  // :178
        r.phase = ForaRun::kTopkRoundStart;
      }
      continue;
    }
    if (r.phase == ForaRun::kTopkFinal) {
      if (r.round == 0 && !r.dead_src) PPRHIP_CHECK_HIP(hipMemsetAsync(g->est, 0, nd, g->stream));
      g->result_in_est = true;
      PPRHIP_TRY(read_dead_pops(g, r.st));
      bool have = false;
      double kth = 0.0;
      PPRHIP_TRY(select_topk(g, g->est, conf->k, r.ids_out, r.vals_out, r.cap, &r.nsel, &kth, &have, r.st));
      r.st.rounds = r.round;
      r.st.rsum = r.rsum_local;
      r.st.rmax_final = r.rmax_local;
      r.st.omega = r.omega_local;
      r.phase = ForaRun::kDone;
    }
    return PPRHIP_OK;
  }
}

// One backward search of All-Pair (Backward_Search.java:38-100 + the >= threshold filter of
// Base_Whole_Graph.java:80-88) as a resumable run.
int bwd_begin(ForaRun& r, pprhip_graph* g, int32_t target_internal, int32_t target_orig, double alpha, double rmax) {
  r.g = g;
  r.kind = 2;
  r.src = target_internal;
  r.target_orig = target_orig;
  r.alpha = alpha;
  r.rmax_local = rmax;
  r.triples.clear();
  std::memset(&r.st, 0, sizeof r.st);
  g->topk_active = false;
  PPRHIP_TRY(reset_query_state(g, false, target_internal));
  r.waiting = false;
  r.in_push = false;
  if (hdeg_in(g, target_internal) == 0) {  // :46-49
    PPRHIP_TRY(launch_set_f64(g, g->reserve, (uint32_t)target_internal, 1.0));
    r.phase = ForaRun::kBwdFinal;
    return PPRHIP_OK;
  }
  r.a = PushArgs{alpha, rmax, 0.0, target_internal, kBackward};
  r.L = LevelCtx();
  PPRHIP_TRY(launch_set_f64(g, g->residue, (uint32_t)target_internal, 1.0));
  PPRHIP_TRY(seed_single(g, r.L, target_internal, hdeg_in(g, target_internal)));
  r.in_push = true;
  r.phase = ForaRun::kBwdLevels;
  return PPRHIP_OK;
}

int bwd_step(ForaRun& r, bool yield_dense) {
  pprhip_graph* g = r.g;
  if (r.phase == ForaRun::kBwdLevels) {
    const int rc = run_levels(g, r.a, r.L, r.st, nullptr, yield_dense);
    if (rc != PPRHIP_OK) return rc;  // kYield or an error
    leave_push(r);
    r.phase = ForaRun::kBwdFinal;
  }
  if (r.phase == ForaRun::kBwdFinal) {
    const double threshold = r.rmax_local;
    unsigned long long thr_bits = 1ull;
    if (threshold > 0.0) std::memcpy(&thr_bits, &threshold, 8);
    PPRHIP_TRY(launch_select_gather(g, g->reserve, act_n(g), thr_bits, true));  // Base_Whole_Graph.java:83 pi >= threshold
    unsigned long long cnt = 0;
    PPRHIP_TRY(fetch_small(g, g->sel_blob, &cnt, sizeof cnt));
    const std::vector<int32_t>& n2o = host_of(g)->h_new2old;
    if (cnt > g->sel_cap) {
      std::vector<double> all(g->n);
      PPRHIP_TRY(copy_out(g, g->reserve, all.data()));
      for (uint32_t v = 0; v < g->n; ++v)  // copy_out already returned original ids
        if (all[v] > 0.0 && all[v] >= threshold) r.triples.push_back({(int32_t)v, r.target_orig, all[v]});
    } else if (cnt) {
      std::vector<SelRec> recs(cnt);
      PPRHIP_CHECK_HIP(hipMemcpyAsync(recs.data(), g->sel_blob + kSelHeader, sizeof(SelRec) * cnt, hipMemcpyDeviceToHost, g->stream));
      PPRHIP_CHECK_HIP(hipStreamSynchronize(g->stream));
      for (uint64_t i = 0; i < cnt; ++i) r.triples.push_back({n2o[recs[i].id], r.target_orig, recs[i].val});
    }
    r.phase = ForaRun::kDone;
  }
  return PPRHIP_OK;
}

void add_stats(pprhip_stats_t& sum, const pprhip_stats_t& st) {
  sum.pops += st.pops; sum.edge_pushes += st.edge_pushes; sum.enqueues += st.enqueues;
  sum.dead_end_pops += st.dead_end_pops; sum.dense_nodes += st.dense_nodes; sum.dense_edges += st.dense_edges;
  sum.levels += st.levels;
  sum.dense_levels += st.dense_levels; sum.rounds += st.rounds; sum.mc_sources += st.mc_sources;
  sum.sweep_min_bytes += st.sweep_min_bytes;
  sum.walks += st.walks; sum.walk_steps += st.walk_steps; sum.select_passes += st.select_passes;
  sum.walk_loads += st.walk_loads; sum.walk_load_lanes += st.walk_load_lanes;
  sum.push_ms += st.push_ms; sum.mc_ms += st.mc_ms; sum.select_ms += st.select_ms; sum.total_ms += st.total_ms;
  sum.push_bytes += st.push_bytes; sum.mc_bytes += st.mc_bytes; sum.select_bytes += st.select_bytes;
  for (int c = 0; c < 8; ++c) {
    sum.class_ms[c] += st.class_ms[c];
    sum.class_bytes[c] += st.class_bytes[c];
    sum.class_launches[c] += st.class_launches[c];
  }
}

}  // namespace

int pprhip_fora_single_source(pprhip_graph_t* g, int32_t src, double eps, const pprhip_fora_conf_t* conf,
                              uint64_t seed, int n_rounds, double* reserve_out, pprhip_stats_t* stats) {
  PPRHIP_TRY(check_graph(g, "pprhip_fora_single_source"));
  PPRHIP_TRY(check_node(g, src, "pprhip_fora_single_source"));
  src = g->h_old2new[src];  // internal (degree-sorted) id
  if (!conf || !(eps > 0.0) || n_rounds < 0) {
    set_error("pprhip_fora_single_source: bad arguments (eps=%g n_rounds=%d)", eps, n_rounds);
    return PPRHIP_ERR_INVALID;
  }
  ForaRun r;
  PPRHIP_TRY(fora_begin(r, g, src, eps, conf, seed, n_rounds));
  CallTimer tm(g);
  r.tm = &tm;
  PPRHIP_TRY(fora_step(r, false));
  tm.finish(r.st);
  r.st.push_ms = CallTimer::ms(g->ev[0], g->ev[1]);
  r.st.mc_ms = CallTimer::ms(g->ev[1], g->ev[2]);
  PPRHIP_TRY(copy_out(g, g->reserve, reserve_out));
  if (stats) *stats = r.st;
  return PPRHIP_OK;
}

namespace {

// One batched dense level for the slots flagged in `active`: stages their arguments, orders the
// parent stream behind the slots' prepare work, runs the sweep and brings the new frontier counters
// back.  The caller holds the sweep exclusively (sequential driver, or BatchSync::sweeping).
// launch: queues one batched sweep for the slots in `active` and the read-back of its counters; P->c8cur names the
// array the NEXT sweep reads from here on, so that whatever is queued on the parent's stream after this point - another
// slot's prepared level (C8Scope) - lands where that sweep will look.  collect: waits for the counters and does the
// slots' bookkeeping.
struct SweepTicket {
  bool active[kBatch] = {false};
  int ws[kBatch] = {0};  // the workspace at each column
  int n_active = 0;
  bool backward = false;
  uint64_t rows = 0;
  unsigned long long seq = 0;
};

// ws: the workspace (index into P->slots and `runs`) that stands at each active column; nullptr: column c = slots[c]
int launch_sweep(pprhip_graph* P, ForaRun* runs, const bool* active, int n_active, SweepTicket* T, const int* ws = nullptr) {
  T->n_active = n_active;
  for (int s = 0; s < kBatch; ++s) {
    T->ws[s] = (ws && active[s]) ? ws[s] : s;
    pprhip_graph* S = P->slots[T->ws[s]];
    SlotArgs& sa = P->h_slot_args[s];
    T->active[s] = active[s];
    sa.res = S->residue;
    sa.reserve = S->reserve;
    sa.flags = S->flags;
    sa.armed = S->armed;
    sa.ctr = S->ctr;
    sa.active = active[s] ? 1 : 0;
    if (!active[s]) continue;
    const ForaRun& r = runs[T->ws[s]];
    sa.alpha = r.a.alpha;
    sa.rmax = r.a.rmax;
    sa.min_rmax = r.a.min_rmax;
    sa.src = r.a.src;
    sa.mode = r.a.mode;
    sa.dead_slot = r.L.dslot;
    sa.out_slot = r.L.pslot ^ 1;
    sa.gs_state = r.L.gs_state;
    // (a slot of the sequential driver has put its prepared level on the parent's stream itself: C8Scope)
    if (S->stream != P->stream && !S->c8_via_parent) {
      PPRHIP_CHECK_HIP(hipEventRecord(S->ev[3], S->stream));
      PPRHIP_CHECK_HIP(hipStreamWaitEvent(P->stream, S->ev[3], 0));
    }
  }
  bool backward = false;
  for (int s = 0; s < kBatch; ++s)
    if (active[s] && runs[T->ws[s]].a.mode == kBackward) backward = true;  // a job's runs all push the same way
  T->backward = backward;
  // SURVEY 8(d) sweep model with n = the rows the sweep carries (launch_dense_level_b8: isolated nodes are left out)
  const uint64_t rows = backward ? (uint64_t)P->n_nz_o + P->n_z_o : (uint64_t)P->n_nz + P->n_zin;
  T->rows = rows;
  const uint64_t sweep_bytes = 4ull * P->m + (uint64_t)n_active * (8ull * P->m + 36ull * rows + 4ull);
  if ((int)backward != P->acc8_dir) {
    // rows summed with atomics are cleared by the apply kernel of their own layout only: start clean
    PPRHIP_CHECK_HIP(hipMemsetAsync(P->acc8, 0, sizeof(double) * ((size_t)P->n + 1) * kBatch, P->stream));
    P->acc8_dir = (int)backward;
  }
  int n_gs = 1;
  const GsBlock* gs_blocks = backward ? nullptr : gs_blocks_of(P->slots[0], &n_gs);  // slots carry the call's tuning
  // (Tried against the ~30 us between two sweeps, round 5: the slots' arguments passed to the kernels by value instead
  // of through a copy command, and the reduce kernel writing the counters into the mailbox itself instead of a
  // k_publish behind it - the sweep took 20-35 us longer either way (1 636-1 651 against 1 613-1 618 us on one box:
  // the apply kernel indexes the by-value block per wave; sixteen workgroups' system-scope fences cost more than one
  // small kernel).  Taken out.)
#ifdef PPRHIP_TEST_HOOKS
  {  // PPRHIP_COUNT_LIVE=1 (measurement): share of a sweep's gathers that fetch a line with a non-zero, on stderr
    static const bool on = hook_env("PPRHIP_COUNT_LIVE") != nullptr;
    static unsigned long long* d_cnt = nullptr;
    static unsigned long long sweeps = 0;
    if (on && !backward) {
      if (!d_cnt) {
        PPRHIP_CHECK_HIP(hipMalloc((void**)&d_cnt, 16));
        PPRHIP_CHECK_HIP(hipMemset(d_cnt, 0, 16));
      }
      PPRHIP_TRY(launch_count_live_lines(P, d_cnt));
      if (++sweeps % 200 == 0) {
        unsigned long long h[2];
        PPRHIP_CHECK_HIP(hipMemcpy(h, d_cnt, 16, hipMemcpyDeviceToHost));
        fprintf(stderr, "[pprhip live lines] %llu sweeps: gathers of live lines %.3f of m, live lines %.3f of the sources, busy columns now %d\n",
                sweeps, (double)h[0] / (double)sweeps / (double)P->m, (double)h[1] / (double)sweeps / (double)P->n_src_live, n_active);
      }
    }
  }
#endif
  P->ktimer.begin(PPRHIP_KERNEL_DENSE_PULL_BATCH, sweep_bytes);
  PPRHIP_TRY(launch_dense_level_b8(P, backward, gs_blocks, n_gs));
  P->ktimer.end();
  P->c8cur ^= 1;
  return fetch_begin(P, P->sweep_out, sizeof(unsigned long long) * kBatch, &T->seq);
}

// true when the counters of a sweep in flight have arrived (collect_sweep would not wait)
bool sweep_arrived(const pprhip_graph* P, const SweepTicket& T) {
  return T.seq != 0 && P->mail && __atomic_load_n(&P->mail->seq, __ATOMIC_ACQUIRE) == T.seq;
}

int collect_sweep(pprhip_graph* P, ForaRun* runs, const SweepTicket& T) {
  PPRHIP_TRY(fetch_end(P, T.seq, P->sweep_out, P->h_sweep_out, sizeof(unsigned long long) * kBatch));
  for (int s = 0; s < kBatch; ++s)
    if (T.active[s]) {
      ForaRun& r = runs[T.ws[s]];
      const unsigned long long pk = P->h_sweep_out[s];
      // the sweep's index stream is shared: each query is charged its own gathers and row work
      finish_dense(r.L, r.st, 8ull * P->m + 36ull * T.rows + 4ull + 4ull * P->m / (uint64_t)T.n_active,
                   batch_sweep_min_bytes(P, T.backward, T.n_active) / (uint64_t)T.n_active, (uint32_t)(pk >> kPackShift),
                   pk & kPackMask);
    }
  return PPRHIP_OK;
}

int run_sweep(pprhip_graph* P, ForaRun* runs, const bool* active, int n_active) {
  SweepTicket T;
  PPRHIP_TRY(launch_sweep(P, runs, active, n_active, &T));
  return collect_sweep(P, runs, T);
}


int run_step(ForaRun& r, bool yield_dense) {
  return r.kind == 2 ? bwd_step(r, yield_dense) : r.kind == 1 ? topk_step(r, yield_dense) : fora_step(r, yield_dense);
}

// outputs of a finished query (its slot still holds the vectors)
int finish_query(BatchJob& J, ForaRun& r) {
  pprhip_graph* S = r.g;
  const int i = r.query;
  if (r.kind == 2) {
    std::lock_guard<std::mutex> lk(J.sum_mu);
    J.triples->insert(J.triples->end(), r.triples.begin(), r.triples.end());
    add_stats(J.sum, r.st);
    r.triples.clear();
    r.phase = ForaRun::kDone;
    r.query = -1;
    return PPRHIP_OK;
  }
  poll_idle(S);
  if (J.keep) {  // the vector stays in HBM after the slot moves on (internal order; pprhip_results_fetch permutes)
    {
      SetupScope setup(S);
      PPRHIP_TRY(launch_copy_f64(S, r.kind == 1 ? S->est : S->reserve, J.keep->buf + (size_t)(J.keep_first + i) * J.P->n,
                                 (size_t)J.P->n));
    }
  }
  if (J.reserve_out) {
    double* dst = J.reserve_out + (size_t)i * J.P->n;
    if (J.pipe) PPRHIP_TRY(J.pipe->submit(S, r.kind == 1 ? S->est : S->reserve, dst));
    else PPRHIP_TRY(copy_out(S, r.kind == 1 ? S->est : S->reserve, dst));
  }
  if (r.kind == 1) {  // the run's final selection wrote the first min(nsel, k) pairs
    for (int j = std::min(r.nsel, J.k); j < J.k; ++j) {
      r.ids_out[j] = -1;
      r.vals_out[j] = 0.0;
    }
    if (J.n_out) J.n_out[i] = r.nsel;
  } else if (J.k > 0) {
    int nsel = 0;
    bool have = false;
    int32_t* ids = J.ids_out + (size_t)i * J.k;
    double* vals = J.vals_out + (size_t)i * J.k;
    PPRHIP_TRY(select_topk(S, S->reserve, J.k, ids, vals, J.k, &nsel, nullptr, &have, r.st));
    for (int j = std::min(nsel, J.k); j < J.k; ++j) {
      ids[j] = -1;
      vals[j] = 0.0;
    }
    if (J.n_out) J.n_out[i] = nsel;
  }
  if (J.per_query) J.per_query[i] = r.st;
  {
    std::lock_guard<std::mutex> lk(J.sum_mu);
    add_stats(J.sum, r.st);
  }
  r.phase = ForaRun::kDone;
  r.query = -1;
  return PPRHIP_OK;
}

int begin_query(BatchJob& J, ForaRun& r, pprhip_graph* S, int i) {
  S->tun = J.P->tun;
  const int32_t src = J.P->h_old2new[J.srcs[i]];
  if (J.kind == 2) {
    pprhip_tuning_batch(&S->tun);  // level shapes only: a backward search has no cost-model decisions
    PPRHIP_TRY(bwd_begin(r, S, src, J.srcs[i], J.alpha, J.threshold));
  } else if (J.kind == 1) {
    PPRHIP_TRY(topk_begin(r, S, src, J.eps, J.conf, J.seed + (uint64_t)i, J.ids_out + (size_t)i * J.k,
                          J.vals_out + (size_t)i * J.k, J.k));
  } else {
    r.kind = 0;
    PPRHIP_TRY(fora_begin(r, S, src, J.eps, J.conf, J.seed, J.n_rounds));
  }
  r.query = i;
  r.job = &J;
  return PPRHIP_OK;
}

// all slots on the calling thread and the graph's stream, one after another
// Whole-graph FORA on one host thread: a query's walk phase goes to a side stream and runs beside the other queries'
// sweeps - with few waves per CU (the walks are bound by the memory system from four waves per CU on,
// tools/micro/chain_rate.hip), so that the compute stream's kernels find room beside it: 292 -> 327 queries/s on
// R-MAT 22 (16 waves per CU beside: 307; 2: 300).
static hipStream_t side_stream_for_walks(pprhip_graph* P) {
  if (!P->walk_stream_tried) {
    P->walk_stream_tried = true;
    const char* e = hook_env("PPRHIP_BATCH_WALKS_BESIDE");
    if (!(e && e[0] == '0') && make_side_stream(P, &P->walk_stream) != PPRHIP_OK) P->walk_stream = nullptr;
  }
  if (P->walk_stream)  // (every call: workspaces may have joined since)
    for (pprhip_graph* S : P->slots)
      for (auto& ev : S->walk_ev)
        if (!ev && hipEventCreate(&ev) != hipSuccess) {
          ev = nullptr;
          (void)hipStreamDestroy(P->walk_stream);
          P->walk_stream = nullptr;
          return nullptr;
        }
  return P->walk_stream;
}

// The stream the slots of the sequential driver work on: it has to run beside the compute stream (the sweeps) and
// beside the walk stream.  PPRHIP_BATCH_SLOTS_BESIDE=0: the slots stay on the compute stream (the driver of rounds 1-4).
static hipStream_t stream_for_slots(pprhip_graph* P) {
  if (!P->slot_stream_tried) {
    P->slot_stream_tried = true;
    const char* e = hook_env("PPRHIP_BATCH_SLOTS_BESIDE");
    if (!(e && e[0] == '0')) {
      if (make_side_stream(P, &P->slot_stream, P->walk_stream) != PPRHIP_OK) P->slot_stream = nullptr;
      if (!P->slot_stream && P->walk_stream && make_side_stream(P, &P->slot_stream) != PPRHIP_OK) P->slot_stream = nullptr;
    }
  }
  return P->slot_stream;
}

// The sequential batch driver: kBatch resumable runs on one host thread, their dense levels served by batched sweeps on
// the handle's compute stream.  Until round 4 everything the slots did ran on that stream too, one blocking step after
// the other: the stream spent a quarter of its time in a query's sparse levels, round ends, seeds and selections - a
// few workgroups each, with a host round trip in between - while fifteen queries waited for the next sweep (kernel
// trace: 12.5 % idle + 14 % in small kernels).  Now a sweep is only LAUNCHED, and while it runs the host takes the
// slots that are not in it through their steps on a second stream (slot_stream; blocking there does not hold the
// sweep up).  What such a slot does to the shared contribution array goes to the compute stream instead (C8Scope:
// the dense prepare of its next level; the compaction back to list form), where stream order places it between two
// sweeps.  A cycle: collect the sweep in flight -> the slots that were in it say what they do next without touching the
// device (another dense level: they wait again; back to list form: the compaction is queued and the rest deferred) ->
// launch the next sweep for those who wait -> the other slots' steps (new queries, sparse levels, round ends, walk
// phases that have ended), until the sweep's counters arrive.
//
// Workspace pool (whole-graph FORA): a column of c8 is only needed between a query's first dense level and its last,
// 26 of the ~40 sweep periods a query spent in its slot on R-MAT 22 - the rest went to its first sparse levels, its
// sparse tail, the walk phase and the selection.  So there are more workspaces than columns (2 x by default,
// PPRHIP_BATCH_WORKSPACES): while sixteen queries hold the columns, the next ones are taken through their first levels
// and stand ready (kYieldColumn) when a column is let go - which happens as soon as its holder leaves a sweep without
// asking for another (the compaction that empties the column is queued first; the newcomer's prepared level lands
// behind it).  Any workspace takes any free column: at the end of a call nobody waits for a column while others idle.
constexpr int kMaxWs = 3 * kBatch;
constexpr int kDefaultWs = 2 * kBatch;

struct SlotDriver {
  pprhip_graph* P = nullptr;
  ForaRun runs[kMaxWs];
  int n_ws = kBatch;
  hipStream_t side = nullptr;  // the walk phases' stream (whole-graph FORA)
  bool walking[kMaxWs] = {false};
  bool col_marked[kMaxWs] = {false};  // col_ev of the workspace has been recorded since it began to wait for its column
  bool flying = false;
  SweepTicket ticket;
  int rr = 0;  // where the pass over the other workspaces starts (round robin: an early end must not starve anybody)
  std::function<bool(BatchJob**, int*)> next;  // the next query to start (false: none right now)
  std::function<void(BatchJob*)> done;         // a query of that job has finished
  int ready_rr = 0;         // where the search for a workspace that stands ready for a free column starts
  int cur_ws = -1;          // the workspace whose step is under way (the hook must not step it again)
  bool in_turn = false;
  KernelTimer* own_timer = g_timer_cur;  // the timer of the thread that runs the driver (turns taken from the hook restore it)
  int hook_rc = PPRHIP_OK;  // what a turn taken from inside a step's wait came to
  std::string hook_msg;

  // pool: more workspaces than columns; slots_on: the stream the workspaces run on from here on (nullptr /
  // P->stream: everything in stream order, as before round 5)
  int setup(pprhip_graph* P_, bool pool, hipStream_t slots_on) {
    P = P_;
    n_ws = kBatch;
    if (pool) {
      int want = kDefaultWs;
      if (const char* e = tuning_env("PPRHIP_BATCH_WORKSPACES")) want = std::max(kBatch, std::min(kMaxWs, atoi(e)));
      if (want > kBatch && ensure_workspaces(P, want) != PPRHIP_OK) {  // (no memory for them: one per column)
        (void)hipGetLastError();
        want = kBatch;
      }
      n_ws = want;
    }
    for (int c = 0; c < kBatch; ++c) P->col_owner[c] = -1;
    for (size_t w = 0; w < P->slots.size(); ++w) {
      pprhip_graph* S = P->slots[w];
      S->stream = slots_on ? slots_on : P->stream;
      S->c8_via_parent = S->stream != P->stream;
      S->sync = nullptr;
      S->pooled = (int)w < n_ws;
      S->has_col = false;
    }
    // the workspaces' read-backs look after the sweep in flight while they wait (only worth it when they wait on
    // another stream than the sweep's)
    if (slots_on && slots_on != P->stream && !hook_env("PPRHIP_BATCH_NO_HOOK")) {
      P->idle_hook = &SlotDriver::on_idle;
      P->idle_arg = this;
    }
    return PPRHIP_OK;
  }
  void teardown() {
    prof.print();
    P->idle_hook = nullptr;
    P->idle_arg = nullptr;
    if (P->slot_stream) (void)hipStreamSynchronize(P->slot_stream);
    for (size_t w = 0; w < P->slots.size(); ++w) {  // (as the other drivers expect them)
      P->slots[w]->pooled = P->slots[w]->has_col = false;
      P->slots[w]->slot_index = (int)(w % kBatch);
    }
  }
  static void on_idle(void* self) {
    SlotDriver* D = static_cast<SlotDriver*>(self);
    if (D->in_turn || D->hook_rc != PPRHIP_OK) return;
    if (D->flying) {
      if (!sweep_arrived(D->P, D->ticket)) return;
    } else {  // nothing on the compute stream (a call's first queries are still starting): whoever stands ready goes
      static const bool early = hook_env("PPRHIP_BATCH_NO_EARLY") == nullptr;
      if (!early) return;
      bool any = false;
      for (int w = 0; w < D->n_ws && !any; ++w) any = D->runs[w].query >= 0 && D->runs[w].waiting;
      if (!any) return;
    }
    D->prof.n[5]++;
    // (a turn taken from inside a workspace's wait may find that workspace's timer swapped in - a walk phase on the
    // side stream runs under a quiet one: the turn's own brackets belong to the timer the driver was started under)
    KernelTimer* const caller_timer = g_timer_cur;
    g_timer_cur = D->own_timer;
    const int rc = D->turn();
    g_timer_cur = caller_timer;
    if (rc != PPRHIP_OK) {
      D->hook_rc = rc;
      D->hook_msg = get_error();
    }
  }

  // a workspace that holds its column without standing at a dense level lets it go (its column is all-zero, or the
  // compaction that makes it so is queued on the compute stream)
  void release_if_idle(int w) {
    pprhip_graph* S = P->slots[w];
    if (S->has_col && !(runs[w].query >= 0 && runs[w].waiting)) {
      P->col_owner[S->slot_index] = -1;
      S->has_col = false;
    }
  }

  // one workspace as far as it gets: until it waits at a dense level, for its column or for its walk phase, or there is
  // nothing to start.  defer: it must not wait for the device (the next sweep is not launched yet).
  int step_ws(int w, bool defer) {
    ForaRun& r = runs[w];
    int rc = PPRHIP_OK;
    const int outer = cur_ws;
    if (!defer) {
      cur_ws = w;
      col_marked[w] = false;
    }
    for (;;) {
      if (r.query < 0) {
        if (defer) break;
        BatchJob* J = nullptr;
        int i = -1;
        if (!next(&J, &i)) break;
        if ((rc = begin_query(*J, r, P->slots[w], i)) != PPRHIP_OK) break;
        r.side = side;
      }
      if (r.waiting) break;
      if (defer) {
        // only a run that stands between two levels of a push can answer without the device: a frontier it can
        // sweep (again), or one that goes back to list form (the compaction is queued; the levels follow later)
        const bool in_levels =
            r.phase == ForaRun::kLevels || r.phase == ForaRun::kTopkLevels || r.phase == ForaRun::kBwdLevels;
        if (!in_levels || r.L.nf == 0 || r.L.compacted) break;
        if (!r.L.dense_prepared) {  // (a run that stood waiting for its column: its next level is a dense one)
          bool dense = false;
          (void)level_cost(r.g, r.L.nf, r.L.ef, &dense);
          if (!dense) break;
        }
        r.L.defer_compact = true;
      }
      P->slots[w]->c8_settled = defer;
      rc = run_step(r, true);
      P->slots[w]->c8_settled = false;
      r.L.defer_compact = false;
      if (rc != kYieldColumn) col_marked[w] = false;  // (it no longer stands ready for a column)
      if (rc == kYield) {
        r.waiting = true;
        rc = PPRHIP_OK;
        break;
      }
      if (rc == kYieldColumn && !defer) {
        // what it has queued so far must have ended before it may take the column without waiting for the stream
        pprhip_graph* S = P->slots[w];
        if (!S->col_ev && hipEventCreateWithFlags(&S->col_ev, hipEventDisableTiming) != hipSuccess) S->col_ev = nullptr;
        col_marked[w] = S->col_ev && hipEventRecord(S->col_ev, S->stream) == hipSuccess;
      }
      if (rc == kYieldDefer || rc == kYieldColumn) {
        rc = PPRHIP_OK;
        break;
      }
      if (rc == kYieldWalk) {
        walking[w] = true;
        rc = PPRHIP_OK;
        break;
      }
      if (rc != PPRHIP_OK) break;
      BatchJob* const J = r.job;
      if ((rc = finish_query(*J, r)) != PPRHIP_OK) break;
      done(J);
    }
    if (!defer) cur_ws = outer;
    release_if_idle(w);
    if (rc == PPRHIP_OK && hook_rc != PPRHIP_OK) {
      set_error("%s", hook_msg.c_str());
      rc = hook_rc;
    }
    return rc;
  }

  // collect the sweep in flight, let its queries (and the ones that stand ready for the columns let go) say what they do
  // next, launch the next sweep: nothing in here waits for the device beyond the sweep's counters
  int turn() {
    in_turn = true;
    const int rc = turn_body();
    in_turn = false;
    return rc;
  }
  // PPRHIP_DRIVER_PROFILE=1: host time of a turn by part, printed when the driver ends (developer switch)
  struct Prof {
    bool on = hook_env("PPRHIP_DRIVER_PROFILE") != nullptr;
    double us[6] = {0};
    unsigned long long n[6] = {0};
    std::chrono::steady_clock::time_point t;
    void start() {
      if (on) t = std::chrono::steady_clock::now();
    }
    void lap(int i) {
      if (!on) return;
      const auto now = std::chrono::steady_clock::now();
      us[i] += std::chrono::duration<double, std::micro>(now - t).count();
      n[i]++;
      t = now;
    }
    void print() const {
      if (!on) return;
      static const char* names[6] = {"collect", "owner goes on (kYield)", "owner leaves (compaction)", "newcomer takes a column", "launch", "turn from a wait"};
      for (int i = 0; i < 6; ++i)
        if (n[i]) fprintf(stderr, "[driver] %-28s %8llu x %8.1f us\n", names[i], n[i], us[i] / (double)n[i]);
    }
  } prof;
  int turn_body() {
    prof.start();
    if (flying) {
      PPRHIP_TRY(collect_sweep(P, runs, ticket));
      flying = false;
      prof.lap(0);
      for (int c = 0; c < kBatch; ++c)
        if (ticket.active[c]) {
          const int w = ticket.ws[c];
          runs[w].waiting = false;
          PPRHIP_TRY(step_ws(w, true));
          prof.lap(runs[w].waiting ? 1 : 2);
        }
      // columns have been let go: workspaces that stand ready prepare their levels behind the compactions
      int n_free = 0;
      for (int c = 0; c < kBatch; ++c) n_free += P->col_owner[c] < 0 ? 1 : 0;
      for (int t = 0; t < n_ws && n_free > 0; ++t) {
        const int w = (ready_rr + t) % n_ws;
        if (w == cur_ws || runs[w].query < 0 || walking[w] || runs[w].waiting || !col_marked[w] ||
            hipEventQuery(P->slots[w]->col_ev) != hipSuccess)
          continue;
        PPRHIP_TRY(step_ws(w, true));
        prof.lap(3);
        if (runs[w].waiting) {
          n_free--;
          ready_rr = (w + 1) % n_ws;
        }
      }
    }
    // (Tried for the end of a call, round 5: with nothing left to start and at most 2 / 3 / 4 queries still in their
    // push, those queries took their column of c8 into vectors of their own and finished with the single-query level
    // kernels - 0.36 ms per level each against 1.6 ms per sweep for any number of columns.  Parity-green and without
    // effect: 350-353 against 351-354 queries/s, the same 857-859 sweeps - a call's last sweeps run at 4-12 busy
    // columns for most of the drain and at <= 4 only for its last few levels.  Taken out.)
    bool active[kBatch];
    int ws[kBatch];
    int n_wait = 0;
    for (int c = 0; c < kBatch; ++c) {
      const int w = P->col_owner[c];
      active[c] = w >= 0 && runs[w].query >= 0 && runs[w].waiting;
      ws[c] = active[c] ? w : c;
      n_wait += active[c] ? 1 : 0;
    }
    if (n_wait) {
      prof.start();
      PPRHIP_TRY(launch_sweep(P, runs, active, n_wait, &ticket, ws));
      flying = true;
      prof.lap(4);
    }
    return PPRHIP_OK;
  }

  // One cycle (see above).  *busy: the workspaces that hold a query afterwards; with none and nothing to start the
  // caller is done (or waits for work).
  // (The pass over the other workspaces takes them through their steps one after the other, each step waiting for its
  // own read-backs.  Tried, round 5: batches of sparse levels launched and collected separately - run_levels returned
  // behind the launches and was called again when the mailbox had the counters, so that all workspaces' first levels
  // were in flight together, with the first sweep of a call held until the others stood ready.  Parity-green and no
  // faster: R-MAT 22 355-356 against 358-359 queries/s, R-MAT 20 1 175 against 1 194, 50 per call 310.7 against 311.8 -
  // a call's length is set by the chain of sweeps each column's queries need, not by how fast the first ones start.
  // Taken out.)
  int cycle(int* busy) {
    PPRHIP_TRY(turn());
    // the workspaces that are not in the sweep
    const int first = rr;
    for (int t = 0; t < n_ws; ++t) {
      const int w = (first + t) % n_ws;
      if (runs[w].query >= 0 && runs[w].waiting) continue;  // in the sweep
      if (flying && sweep_arrived(P, ticket)) {  // the compute stream is idle: the sweep's queries come first
        rr = w;
        break;
      }
      if (walking[w]) {
        if (hipEventQuery(P->slots[w]->walk_ev[2]) == hipErrorNotReady) continue;
        walking[w] = false;
      }
      if (col_marked[w]) {  // it stands ready for a column: nothing to do for it while none is free
        bool any_free = false;
        for (int c = 0; c < kBatch && !any_free; ++c) any_free = P->col_owner[c] < 0;
        if (!any_free) continue;
      }
      PPRHIP_TRY(step_ws(w, false));
    }
    *busy = 0;
    int n_wait = 0, n_pending = 0, first_walk = -1;
    for (int w = 0; w < n_ws; ++w) {
      const bool has = runs[w].query >= 0;
      *busy += has ? 1 : 0;
      n_wait += (has && runs[w].waiting) ? 1 : 0;
      // (left behind by a turn taken from inside this pass, after the pass had gone by: the next cycle takes it on)
      n_pending += (has && !runs[w].waiting && !walking[w]) ? 1 : 0;
      if (walking[w] && first_walk < 0) first_walk = w;
    }
    if (*busy > 0 && !flying && n_wait == 0 && n_pending == 0) {
      // nobody stands at a dense level and no sweep is on its way: a walk phase has to end before anything can go on
      if (first_walk < 0) {
        set_error("batch driver: %d queries in flight, none waiting", *busy);
        return PPRHIP_ERR_STATE;
      }
      PPRHIP_CHECK_HIP(hipEventSynchronize(P->slots[first_walk]->walk_ev[2]));
    }
    return PPRHIP_OK;
  }
};

// Queries left over when a call's count is not a multiple of the slots: up to kTailSingle of them run one at a time on
// the handle's own workspace (the single-query path: 10 ms each on R-MAT 22) instead of as a last round of sweeps with
// nearly all columns empty - a sweep costs the same for 2 busy columns as for 16, so such a round takes most of a
// query's latency.  PPR.java:179's 50 queries per call = 3 x 16 + 2: 178 -> 172 ms per call.
constexpr int kTailSingle = 3;
int tail_queries(const BatchJob& J) {
  return (J.kind == 0 && J.q > kBatch && J.q % kBatch <= kTailSingle && !hook_env("PPRHIP_BATCH_NO_TAIL")) ? J.q % kBatch : 0;
}
int run_tail(BatchJob& J, int q_slots) {
  for (int i = q_slots; i < J.q; ++i) {  // the stragglers, one at a time on the handle's own vectors
    ForaRun r;
    PPRHIP_TRY(begin_query(J, r, J.P, i));
    r.side = nullptr;
    int rc;
    while ((rc = run_step(r, false)) == kYield) {
    }
    if (rc != PPRHIP_OK) return rc;
    PPRHIP_TRY(finish_query(J, r));
  }
  return PPRHIP_OK;
}

// all queries on the calling thread (SlotDriver)
int batch_sequential(BatchJob& J) {
  pprhip_graph* P = J.P;
  std::unique_ptr<SlotDriver> Dp(new (std::nothrow) SlotDriver());
  if (!Dp) return PPRHIP_ERR_OOM;
  SlotDriver& D = *Dp;
  D.side = J.kind == 0 ? side_stream_for_walks(P) : nullptr;
  PPRHIP_TRY(D.setup(P, J.kind == 0 && J.q > kBatch, stream_for_slots(P)));
  if (D.side) D.side = side_stream_for_walks(P);  // (the new workspaces' events)
  KernelTimer& tm = ktimer();  // (the call's timer watches the stream the workspaces' kernels run on ...)
  tm.stream = P->slots[0]->stream;
  // queries the workspaces run (the leftover rule is for one workspace per column: with the pool there are no rounds
  // of 16 whose last one would be nearly empty - 50 / 51 / 35 sources per call: 306 / 302 / 281 queries/s without
  // the rule, 306 / 292 / 270 with it).
  // (A query is a chain of ~26 dense levels and a column serves one level per sweep: 50 queries on 16 columns cost two
  // columns four queries' worth of sweeps, ~104, whatever the order.  Tried against that, round 5: the q mod 16 <= 4
  // leftovers on a helper thread and a stream of their own BESIDE the batch, each on a workspace that runs
  // single-query dense levels over vectors of its own, so that the other 48 take three queries' worth.  Parity-green
  // and no faster - 50 / 51 / 35 / 20 sources per call: 314 / 304 / 281 / 225 queries/s against 310 / 306 / 285 / 252: the
  // single-query edge kernel (a 1024-thread workgroup with a 128-KB table per CU) does not fit on a CU beside the
  // batched one, so its levels run in the gaps between the sweeps' kernels, one per sweep period - as in a column.
  // Taken out; the query stream is the answer for calls that follow one another: 351-362 queries/s on blocks of 50.)
  const int q_slots = J.q - (D.n_ws > kBatch ? 0 : tail_queries(J));
  D.next = [&](BatchJob** job, int* i) {
    *i = J.next_query.fetch_add(1);
    *job = &J;
    return *i < q_slots;
  };
  D.done = [](BatchJob*) {};
  int rc = PPRHIP_OK;
  try {  // (no exception may cross the C ABI: the driver's containers and callbacks allocate)
    for (;;) {
      int busy = 0;
      if ((rc = D.cycle(&busy)) != PPRHIP_OK) break;
      if (busy == 0) break;
    }
  } catch (const std::exception& ex) {
    set_error("batch driver: %s", ex.what());
    rc = PPRHIP_ERR_OOM;
  }
  D.teardown();
  if (rc != PPRHIP_OK) return rc;
  if (tm.stream != P->stream) {  // (... and the stragglers' on the handle's own)
    (void)hipStreamSynchronize(P->stream);
    tm.fold();
    tm.stream = P->stream;
  }
  return run_tail(J, q_slots);
}

// (Round 4 also ran this driver on 2 - 16 host threads that shared the one compute stream, the slots dealt out between
// them and the threads meeting once per sweep - the idea being that the transitions of different threads' slots fill
// each other's gaps in the stream, which idles 12-18 % of the time behind the host's decisions.  It got slower with
// every thread added: 326 / 322 / 315 / 307 / 301 queries/s with 1 / 2 / 4 / 8 / 16 threads, the sweeps themselves
// 1 334 -> 1 443 us (profiles/r04_driver_threads_study.txt) - several threads launching into one stream pay more in the
// runtime than the gaps they close.  Taken out.  Round 5 closes the gaps from ONE thread instead: SlotDriver.)

// one worker thread per slot
void batch_worker(BatchJob* J, BatchSync* B, ForaRun* runs, int s) {
  pprhip_graph* P = J->P;
  pprhip_graph* S = P->slots[s];
  ForaRun& r = runs[s];
  int rc = PPRHIP_OK;
  if (hipSetDevice(P->device) != hipSuccess) {
    set_error("hipSetDevice(%d) failed in a batch worker", P->device);
    rc = PPRHIP_ERR_HIP;
  }
  KernelTimer* const own_timer = g_timer_cur;
  g_timer_cur = &S->ktimer;
  S->ktimer.stream = S->stream;
  S->ktimer.reset();
  while (rc == PPRHIP_OK) {
    {
      std::lock_guard<std::mutex> lk(B->mu);
      if (B->err) break;
    }
    const int i = J->next_query.fetch_add(1);
    if (i >= J->q) break;
    rc = begin_query(*J, r, S, i);
    while (rc == PPRHIP_OK) {
      rc = run_step(r, true);
      if (rc != kYield) break;
      rc = B->arrive(s);
    }
    if (rc == PPRHIP_OK) rc = finish_query(*J, r);
  }
  if (rc != PPRHIP_OK) {
    leave_push(r);
    B->fail(rc);
  }
  (void)hipStreamSynchronize(S->stream);
  g_timer_cur = own_timer;
  B->worker_done(s);
}

}  // namespace

namespace pprhip {

void BatchSync::release(int s) {
  std::lock_guard<std::mutex> lk(mu);
  if (hold[s]) {
    hold[s] = false;
    n_hold--;
    cv.notify_all();
  }
}

void BatchSync::c8_enter(int s) {
  std::unique_lock<std::mutex> lk(mu);
  cv.wait(lk, [&] { return !sweeping || err != 0; });
  if (!hold[s]) {
    hold[s] = true;
    n_hold++;
  }
}

void BatchSync::fail(int rc) {
  std::lock_guard<std::mutex> lk(mu);
  if (!err) {
    err = rc;
    errmsg = get_error();
  }
  cv.notify_all();
}

void BatchSync::worker_done(int s) {
  std::lock_guard<std::mutex> lk(mu);
  if (hold[s]) {
    hold[s] = false;
    n_hold--;
  }
  n_workers--;
  cv.notify_all();
}

int BatchSync::arrive(int s) {
  std::unique_lock<std::mutex> lk(mu);
  if (err) return err;
  if (hold[s]) {
    hold[s] = false;
    n_hold--;
  }
  waitflag[s] = true;
  n_wait++;
  cv.notify_all();
  cv.wait(lk, [&] { return !waitflag[s] || err != 0; });
  return err;
}

// the sweeper thread: one batched sweep whenever somebody waits and nobody holds
void BatchSync::sweeper() {
  (void)hipSetDevice(P->device);
  std::unique_lock<std::mutex> lk(mu);
  for (;;) {
    cv.wait(lk, [&] { return n_workers == 0 || err != 0 || (n_wait > 0 && n_hold == 0); });
    if (n_workers == 0 || err != 0) return;
    sweeping = true;
    bool active[kBatch];
    int n_active = 0;
    for (int s = 0; s < kBatch; ++s) {
      active[s] = waitflag[s];
      n_active += active[s] ? 1 : 0;
    }
    lk.unlock();
    const int rc = run_sweep(P, runs, active, n_active);
    const std::string msg = rc != PPRHIP_OK ? get_error() : "";
    lk.lock();
    if (rc != PPRHIP_OK && !err) {
      err = rc;
      errmsg = msg;
    }
    for (int s = 0; s < kBatch; ++s)
      if (active[s]) {
        waitflag[s] = false;
        n_wait--;
        hold[s] = true;  // until the slot has said what it does next
        n_hold++;
      }
    sweeping = false;
    cv.notify_all();
  }
}

}  // namespace pprhip

// Batched single-source FORA: up to kBatch queries in flight on kBatch workspaces of this handle.
// Every query runs the single-query algorithm unchanged (same levels, same thresholds, same walks
// for the same seed); whenever the queries in a push phase all stand at a dense level, one sweep of
// the batched kernels serves them.  All slots run on the calling thread and the handle's stream;
// with PPRHIP_BATCH_THREADS=1 (the default of the top-k entry point) every slot gets a worker
// thread and a stream of its own, so sparse levels, walks and selections of different queries
// overlap on the GPU.
int pprhip::detail::FetchPipe::ensure(pprhip_graph* parent) {
  if (cs) return PPRHIP_OK;
  P = parent;
  n = parent->n;
  for (int e = 0; e < kRing; ++e) {
    PPRHIP_TRY(alloc_dev((void**)&dev[e], sizeof(double) * n));
    PPRHIP_CHECK_HIP(hipHostMalloc((void**)&pin[e], sizeof(double) * std::max<size_t>(n, 1), hipHostMallocDefault));
    PPRHIP_CHECK_HIP(hipEventCreateWithFlags(&ready[e], hipEventDisableTiming));
    PPRHIP_CHECK_HIP(hipEventCreateWithFlags(&done[e], hipEventDisableTiming));
  }
  // The copy stream has to sit on another hardware queue than the compute stream: on a shared queue no copy ever
  // overlapped a kernel (tools/exp/copy_overlap.py: kernels ran during 0.0 % of the copies' time).  make_side_stream
  // tries candidates until one runs beside the compute stream; without one, a plain stream (copies then run between
  // kernels, as before round 3).
  PPRHIP_TRY(make_side_stream(parent, &cs));
  if (!cs) PPRHIP_CHECK_HIP(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));  // last: marks the pipe complete
  return PPRHIP_OK;
}

void pprhip::detail::FetchPipe::start() {
  closing = false;
  err = 0;
  pending = 0;
  work.clear();
  free_q.clear();
  for (int e = 0; e < kRing; ++e) free_q.push_back(e);
  for (int t = 0; t < kCopiers; ++t) copiers[t] = std::thread(&FetchPipe::copier, this);
}

void pprhip::detail::FetchPipe::copier() {
  (void)hipSetDevice(P->device);
  for (;;) {
    Item it;
    {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return !work.empty() || (closing && pending == 0); });
      if (work.empty()) return;  // closing and drained
      it = work.front();
      work.pop_front();
    }
    // (the item was queued by the copy stream's host callback: the vector is in pin[it.e]; copiers make no HIP call)
    std::memcpy(it.dst, pin[it.e], sizeof(double) * n);
    std::lock_guard<std::mutex> lk(mu);
    free_q.push_back(it.e);
    cv.notify_all();
  }
}

// host callback of the copy stream: the vector of ring entry e has reached its pinned buffer
void pprhip::detail::FetchPipe::on_copied(void* p) {
  Arrival* a = static_cast<Arrival*>(p);
  {
    std::lock_guard<std::mutex> lk(a->pipe->mu);
    a->pipe->work.push_back(a->item);
    a->pipe->pending--;
  }
  a->pipe->cv.notify_all();
  delete a;
}

int pprhip::detail::FetchPipe::submit(pprhip_graph* S, const double* dev_vec, double* dst) {
  int e;
  {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return !free_q.empty() || err; });
    if (err) {
      set_error("delivery of a result vector failed (copy stream)");
      return err;
    }
    e = free_q.front();
    free_q.pop_front();
  }
  // A failure from here on hands the ring entry back and marks the pipe failed (finish() then stops waiting for
  // callbacks that may never run).  No HIP call is made with `mu` held: the copy stream's callback takes `mu` on a
  // thread of the runtime, and a HIP call that waited for work queued behind a pending callback would never return.
  auto fail = [&](int rc) {
    {
      std::lock_guard<std::mutex> lk(mu);
      free_q.push_back(e);
      if (!err) err = rc;
    }
    cv.notify_all();
    return rc;
  };
  int rc = PPRHIP_OK;
  if (S->relabeled) {  // back to the caller's ids: out[old] = x[old2new[old]]
    rc = launch_permute_out(S, dev_vec, dev[e]);
  } else if (hipMemcpyAsync(dev[e], dev_vec, sizeof(double) * n, hipMemcpyDeviceToDevice, S->stream) != hipSuccess) {
    set_error("delivery of a result vector failed (staging copy)");
    rc = PPRHIP_ERR_HIP;
  }
  if (rc == PPRHIP_OK && hipEventRecord(ready[e], S->stream) != hipSuccess) {
    set_error("delivery of a result vector failed (event)");
    rc = PPRHIP_ERR_HIP;
  }
  if (rc != PPRHIP_OK) return fail(rc);
  Arrival* a = new (std::nothrow) Arrival{this, Item{e, dst}};
  if (!a) return fail(PPRHIP_ERR_OOM);
  {
    // the copy stream is shared by the slots' threads: its three calls stay together
    std::lock_guard<std::mutex> order(cs_mu);
    if (hipStreamWaitEvent(cs, ready[e], 0) != hipSuccess ||
        hipMemcpyAsync(pin[e], dev[e], sizeof(double) * n, hipMemcpyDeviceToHost, cs) != hipSuccess) {
      set_error("delivery of a result vector failed (copy stream)");
      delete a;
      return fail(PPRHIP_ERR_HIP);
    }
    {
      std::lock_guard<std::mutex> lk(mu);
      pending++;
    }
    if (hipLaunchHostFunc(cs, &FetchPipe::on_copied, a) != hipSuccess) {  // no callback will run for this entry
      set_error("delivery of a result vector failed (host callback)");
      {
        std::lock_guard<std::mutex> lk(mu);
        pending--;
      }
      delete a;
      return fail(PPRHIP_ERR_HIP);
    }
  }
  return PPRHIP_OK;
}

int pprhip::detail::FetchPipe::finish() {
  // the copy stream drains first (its callbacks queue the last vectors), then the copiers
  const bool drained = !cs || hipStreamSynchronize(cs) == hipSuccess;
  {
    std::lock_guard<std::mutex> lk(mu);
    if (!drained && !err) err = PPRHIP_ERR_HIP;
    closing = true;
    if (err) pending = 0;  // a failed copy stream may never run its callbacks: the copiers must not wait for them
  }
  cv.notify_all();
  for (int t = 0; t < kCopiers; ++t)
    if (copiers[t].joinable()) copiers[t].join();
  if (err) set_error("delivery of a result vector failed (copy stream)");
  return err;
}

void pprhip::detail::FetchPipe::destroy() {
  for (int e = 0; e < kRing; ++e) {
    if (dev[e]) (void)hipFree(dev[e]);
    if (pin[e]) (void)hipHostFree(pin[e]);
    if (ready[e]) (void)hipEventDestroy(ready[e]);
    if (done[e]) (void)hipEventDestroy(done[e]);
    dev[e] = pin[e] = nullptr;
    ready[e] = done[e] = nullptr;
  }
  if (cs) (void)hipStreamDestroy(cs);
  cs = nullptr;
}

// runs a prepared job on the handle's slots (both batched entry points)
int pprhip::detail::batch_run(pprhip_graph_t* g, BatchJob& J, pprhip_stats_t* stats_sum) {
  PPRHIP_TRY(ensure_batch(g));
  if (J.kind == 2) PPRHIP_TRY(ensure_bwd_layout(g));
  const int q = J.q;
  // Worker threads pay off where queries are latency-bound (top-k: short rounds of sparse levels, walks
  // and selections, 2.4x on R-MAT 22); whole-graph FORA keeps the memory system busy from one thread.
  const char* env = tuning_env("PPRHIP_BATCH_THREADS");
  const bool threaded = q > 1 && (env ? env[0] == '1' : J.kind != 0);
  std::memset(&J.sum, 0, sizeof J.sum);
  // vectors go to the caller's memory behind the queries' backs (a synchronous copy of 8n bytes to pageable memory per
  // query would stall the one stream everything runs on: 170 instead of 270 queries/s on R-MAT 22)
  if (J.reserve_out && J.kind != 2 && q > 1) {
    if (!g->fetch) g->fetch = new (std::nothrow) FetchPipe();
    if (!g->fetch) return PPRHIP_ERR_OOM;
    const int prc = g->fetch->ensure(g);
    if (prc != PPRHIP_OK) {
      g->fetch->destroy();
      delete g->fetch;
      g->fetch = nullptr;
      return prc;
    }
    g->fetch->start();
    J.pipe = g->fetch;
  }
  ForaRun runs[kBatch];
  g->ktimer.stream = g->stream;
  g->ktimer.reset();
  const auto t0 = std::chrono::steady_clock::now();
  int rc = PPRHIP_OK;
  double tot[8] = {0};
  uint64_t bytes[8] = {0};
  uint32_t cnt[8] = {0};
  if (threaded) {
    BatchSync B;
    B.P = g;
    B.runs = runs;
    for (pprhip_graph* S : g->slots) {
      S->stream = S->own_stream;
      S->c8_via_parent = false;
      S->sync = &B;
    }
    B.n_workers = kBatch;
    std::thread sweeper(&BatchSync::sweeper, &B);
    std::vector<std::thread> workers;
    for (int s = 0; s < kBatch; ++s) workers.emplace_back(batch_worker, &J, &B, runs, s);
    for (auto& w : workers) w.join();
    sweeper.join();
    for (pprhip_graph* S : g->slots) {
      S->sync = nullptr;
      S->ktimer.resolve(tot, bytes, cnt);
    }
    if (B.err) {
      set_error("%s", B.errmsg.c_str());
      rc = B.err;
    }
  } else {
    for (pprhip_graph* S : g->slots) {
      S->stream = g->stream;
      S->c8_via_parent = false;
      S->sync = nullptr;
    }
    KernelTimer local;  // the caller's timer may be in use (All-Pair times its own tiers)
    KernelTimer* const saved = g_timer_cur;
    g_timer_cur = &local;
    local.stream = g->stream;
    rc = batch_sequential(J);
    (void)hipStreamSynchronize(g->stream);
    local.resolve(tot, bytes, cnt);
    local.destroy();
    g_timer_cur = saved;
  }
  (void)hipStreamSynchronize(g->stream);
  if (J.pipe) {
    const std::string msg = rc != PPRHIP_OK ? get_error() : std::string();
    const int prc = J.pipe->finish();  // every vector submitted so far has reached its destination
    J.pipe = nullptr;
    if (rc != PPRHIP_OK) set_error("%s", msg.c_str());
    else rc = prc;
  }
  if (rc != PPRHIP_OK) {
    const std::string msg = get_error();
    free_batch(g);  // slots may hold half-pushed levels: the next batched call builds clean ones
    set_error("%s", msg.c_str());
    return rc;
  }
  g->ktimer.resolve(tot, bytes, cnt);
  pprhip_stats_t& sum = J.sum;
  sum.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  int best = 0;
  for (int c = 0; c < 8; ++c) {
    sum.class_ms[c] = tot[c];
    sum.class_bytes[c] = bytes[c];
    sum.class_launches[c] = cnt[c];
    if (tot[c] > tot[best]) best = c;
  }
  sum.dominant_kernel_id = (uint32_t)best;
  sum.dominant_kernel_ms = tot[best];
  sum.dominant_kernel_bytes = bytes[best];
  sum.dominant_kernel_launches = cnt[best];
  if (stats_sum) *stats_sum = sum;
  return PPRHIP_OK;
}

int pprhip_fora_batch_single_source_resident(pprhip_graph_t* g, const int32_t* srcs, int q, double eps,
                                             const pprhip_fora_conf_t* conf, uint64_t seed, int n_rounds,
                                             pprhip_results_t* keep, double* reserve_out, int k, int32_t* ids_out,
                                             double* vals_out, int* n_out, pprhip_stats_t* per_query,
                                             pprhip_stats_t* stats_sum) {
  PPRHIP_TRY(check_graph(g, "pprhip_fora_batch_single_source"));
  if (q < 0 || !conf || !(eps > 0.0) || n_rounds < 0 || (q > 0 && !srcs) || k < 0 ||
      (k > 0 && q > 0 && (!ids_out || !vals_out))) {
    set_error("pprhip_fora_batch_single_source: bad arguments (q=%d eps=%g n_rounds=%d k=%d)", q, eps, n_rounds, k);
    return PPRHIP_ERR_INVALID;
  }
  if (keep && (keep->g != g || q > keep->capacity)) {
    set_error("pprhip_fora_batch_single_source_resident: the result store belongs to another graph or holds %d < %d "
              "queries", keep->capacity, q);
    return PPRHIP_ERR_INVALID;
  }
  for (int i = 0; i < q; ++i) PPRHIP_TRY(check_node(g, srcs[i], "pprhip_fora_batch_single_source"));
  BatchJob J;
  J.P = g;
  J.srcs = srcs;
  J.q = q;
  J.eps = eps;
  J.conf = conf;
  J.seed = seed;
  J.n_rounds = n_rounds;
  J.reserve_out = reserve_out;
  J.k = k;
  J.ids_out = ids_out;
  J.vals_out = vals_out;
  J.n_out = n_out;
  J.per_query = per_query;
  J.keep = keep;
  if (keep) keep->count = 0;
  PPRHIP_TRY(batch_run(g, J, stats_sum));
  if (keep) keep->count = q;
  return PPRHIP_OK;
}

int pprhip_fora_batch_single_source(pprhip_graph_t* g, const int32_t* srcs, int q, double eps,
                                    const pprhip_fora_conf_t* conf, uint64_t seed, int n_rounds,
                                    double* reserve_out, int k, int32_t* ids_out, double* vals_out, int* n_out,
                                    pprhip_stats_t* per_query, pprhip_stats_t* stats_sum) {
  return pprhip_fora_batch_single_source_resident(g, srcs, q, eps, conf, seed, n_rounds, nullptr, reserve_out, k,
                                                  ids_out, vals_out, n_out, per_query, stats_sum);
}

// ------------------------------------------------------------------ device-resident result store
int pprhip_results_create(pprhip_graph_t* g, int capacity, pprhip_results_t** results_out) {
  PPRHIP_TRY(check_graph(g, "pprhip_results_create"));
  if (capacity < 1 || !results_out) {
    set_error("pprhip_results_create: bad arguments (capacity=%d)", capacity);
    return PPRHIP_ERR_INVALID;
  }
  pprhip_results* r = new (std::nothrow) pprhip_results();
  if (!r) return PPRHIP_ERR_OOM;
  r->g = g;
  r->device = g->device;
  r->capacity = capacity;
  const int rc = alloc_dev((void**)&r->buf, sizeof(double) * (size_t)capacity * g->n);
  if (rc != PPRHIP_OK) {
    delete r;
    return rc;
  }
  *results_out = r;
  return PPRHIP_OK;
}

void pprhip_results_destroy(pprhip_results_t* r) {
  if (!r) return;
  (void)hipSetDevice(r->device);
  if (r->buf) (void)hipFree(r->buf);
  delete r;
}

int pprhip_results_info(const pprhip_results_t* r, int* capacity, int* count, uint32_t* n) {
  if (!r) {
    set_error("pprhip_results_info: null store");
    return PPRHIP_ERR_INVALID;
  }
  if (capacity) *capacity = r->capacity;
  if (count) *count = r->count;
  if (n) *n = r->g->n;
  return PPRHIP_OK;
}

static int results_slot(pprhip_results_t* r, int i, const char* fn) {
  if (!r || i < 0 || i >= r->count) {
    set_error("%s: no result %d in the store (%d held)", fn, i, r ? r->count : 0);
    return PPRHIP_ERR_INVALID;
  }
  return check_graph(r->g, fn);
}

int pprhip_results_fetch(pprhip_results_t* r, int i, double* reserve_out) {
  PPRHIP_TRY(results_slot(r, i, "pprhip_results_fetch"));
  if (!reserve_out) {
    set_error("pprhip_results_fetch: null output");
    return PPRHIP_ERR_INVALID;
  }
  return copy_out(r->g, r->buf + (size_t)i * r->g->n, reserve_out);
}

int pprhip_results_sum(pprhip_results_t* r, int i, double* sum_out) {
  PPRHIP_TRY(results_slot(r, i, "pprhip_results_sum"));
  if (!sum_out) {
    set_error("pprhip_results_sum: null output");
    return PPRHIP_ERR_INVALID;
  }
  return device_sum(r->g, r->buf + (size_t)i * r->g->n, sum_out, r->g->n);
}

int pprhip_fora_batch_topk(pprhip_graph_t* g, const int32_t* srcs, int q, int k, double eps, double alpha,
                           uint64_t seed, int32_t* ids_out, double* vals_out, pprhip_stats_t* stats_sum) {
  PPRHIP_TRY(check_graph(g, "pprhip_fora_batch_topk"));
  if (q < 0 || k < 1 || !(eps > 0.0) || (q > 0 && (!srcs || !ids_out || !vals_out))) {
    set_error("pprhip_fora_batch_topk: bad arguments");
    return PPRHIP_ERR_INVALID;
  }
  for (int i = 0; i < q; ++i) PPRHIP_TRY(check_node(g, srcs[i], "pprhip_fora_batch_topk"));
  pprhip_fora_conf_t conf;
  PPRHIP_TRY(pprhip_conf_fora_topk(g->n, g->m, k, alpha, &conf));
  BatchJob J;
  J.P = g;
  J.kind = 1;
  J.srcs = srcs;
  J.q = q;
  J.eps = eps;
  J.conf = &conf;
  J.seed = seed;  // query i runs with seed + i, as pprhip_fora_topk(srcs[i], ..., seed + i) would
  J.n_rounds = 0;
  J.reserve_out = nullptr;
  J.k = k;
  J.ids_out = ids_out;
  J.vals_out = vals_out;
  J.n_out = nullptr;
  J.per_query = nullptr;
  return batch_run(g, J, stats_sum);
}


// ------------------------------------------------------------------ query stream
// The batched driver behind a submit / wait pair.  A synchronous call of q queries ends with a drain: its last queries
// finish at different times, the slots they leave stay empty, and a sweep costs the same for 2 busy columns as for 16
// (config #4's 50-query call, PPR.java:179: 0.90 of the 128-query rate).  A stream keeps one driver thread on the
// handle; the slots a submission's last queries leave take the next submission's first ones, so queries that arrive
// continuously - a harness that calls Gen_Util's loop again and again, a server - always find sixteen columns busy.
// Every query runs exactly as pprhip_fora_batch_single_source would run it (same seed, same tuning, same result).
#include <deque>
#include <functional>
#include <map>
#include <memory>

namespace {

struct StreamJob : BatchJob {
  std::vector<int32_t> own_srcs;  // (the caller's array need not outlive the submit call)
  uint64_t ticket = 0;
  int finished = 0;
  bool done = false;
  std::chrono::steady_clock::time_point t0;
};

}  // namespace

struct pprhip_stream {
  pprhip_graph* g = nullptr;
  double eps = 0.0;
  pprhip_fora_conf_t conf;
  int k = 0;
  std::mutex mu;
  std::condition_variable cv_work, cv_done;
  std::deque<std::shared_ptr<StreamJob>> pending;           // submissions with queries still to start
  std::map<uint64_t, std::shared_ptr<StreamJob>> open;      // ticket -> submission, until it has been waited for
  uint64_t next_ticket = 1;
  bool closing = false;
  int err = PPRHIP_OK;
  std::string errmsg;
  std::thread driver;
};

namespace {

void stream_fail(pprhip_stream* s, int rc) {
  std::lock_guard<std::mutex> lk(s->mu);
  if (s->err == PPRHIP_OK) {
    s->err = rc;
    s->errmsg = get_error();
  }
  s->pending.clear();
  for (auto& kv : s->open) kv.second->done = true;
  s->cv_done.notify_all();
}

void stream_driver(pprhip_stream* s) {
  pprhip_graph* P = s->g;
  if (hipSetDevice(P->device) != hipSuccess) {
    set_error("hipSetDevice(%d) failed in the stream driver", P->device);
    stream_fail(s, PPRHIP_ERR_HIP);
    return;
  }
  KernelTimer quiet;  // nobody reads kernel-class times of a stream: record no events at all
  quiet.off = true;
  KernelTimer* const saved = g_timer_cur;
  g_timer_cur = &quiet;
  std::unique_ptr<SlotDriver> Dp(new (std::nothrow) SlotDriver());
  if (!Dp) {
    set_error("query stream: no memory for the driver's state");
    stream_fail(s, PPRHIP_ERR_OOM);
    g_timer_cur = saved;
    return;
  }
  SlotDriver& D = *Dp;
  D.side = side_stream_for_walks(P);
  (void)D.setup(P, true, stream_for_slots(P));
  if (D.side) D.side = side_stream_for_walks(P);  // (the new workspaces' events)
  hipStream_t side = D.side;
  // test switch: PPRHIP_STREAM_FAULT_AT=<n> makes the driver fail when it is about to start the stream's n-th query
  // (0-based), as a failing kernel launch would: every open and later submission ends with the driver's error
  long fault_at = -1, started = 0;
  if (const char* fe = hook_env("PPRHIP_STREAM_FAULT_AT")) fault_at = atol(fe);
  bool injected = false;
  D.next = [&](BatchJob** job, int* i) {
    if (fault_at >= 0 && started == fault_at) {
      injected = true;
      return false;
    }
    std::lock_guard<std::mutex> lk(s->mu);
    if (s->pending.empty()) return false;
    StreamJob* J = s->pending.front().get();
    *job = J;
    *i = J->next_query.fetch_add(1);
    if (*i + 1 >= J->q) s->pending.pop_front();  // (the open map keeps the submission alive)
    ++started;
    return true;
  };
  D.done = [&](BatchJob* job) {
    StreamJob* J = static_cast<StreamJob*>(job);
    std::lock_guard<std::mutex> lk(s->mu);
    if (++J->finished == J->q) {
      J->sum.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - J->t0).count();
      J->done = true;
      s->cv_done.notify_all();
    }
  };
  int rc = PPRHIP_OK;
  try {  // (no exception leaves the driver thread: it would end the process)
  for (;;) {
    int busy = 0;
    if ((rc = D.cycle(&busy)) != PPRHIP_OK) break;
    if (injected) {
      set_error("query stream: injected failure before query %ld (PPRHIP_STREAM_FAULT_AT)", fault_at);
      rc = PPRHIP_ERR_STATE;
      break;
    }
    if (busy == 0) {
      std::unique_lock<std::mutex> lk(s->mu);
      s->cv_work.wait(lk, [&] { return s->closing || !s->pending.empty(); });
      if (s->pending.empty()) break;  // closing, and nothing left to start or in flight
      continue;
    }
  }
  } catch (const std::exception& ex) {
    set_error("query stream: %s in the driver thread", ex.what());
    rc = PPRHIP_ERR_OOM;
  }
  // Drain before anybody is woken: a waiter that returns the error may free its result store or its output block at
  // once, and copies or selections of other slots' queries can still be queued against those buffers.
  (void)hipStreamSynchronize(P->stream);
  D.teardown();
  if (P->walk_stream) (void)hipStreamSynchronize(P->walk_stream);
  if (side && side != P->stream && side != P->walk_stream) (void)hipStreamSynchronize(side);
  if (rc != PPRHIP_OK) stream_fail(s, rc);
  g_timer_cur = saved;
}

int stream_error(pprhip_stream* s, const char* fn) {  // (s->mu held)
  set_error("%s: the stream has failed: %s", fn, s->errmsg.c_str());
  return s->err;
}

}  // namespace

int pprhip_fora_stream_open(pprhip_graph_t* g, double eps, const pprhip_fora_conf_t* conf, int k,
                            pprhip_stream_t** stream_out) {
  PPRHIP_TRY(check_graph(g, "pprhip_fora_stream_open"));
  if (!stream_out || !conf || !(eps > 0.0) || k < 0) {
    set_error("pprhip_fora_stream_open: bad arguments (eps=%g k=%d)", eps, k);
    return PPRHIP_ERR_INVALID;
  }
  PPRHIP_TRY(ensure_batch(g));
  std::unique_ptr<pprhip_stream> s(new (std::nothrow) pprhip_stream());
  if (!s) return PPRHIP_ERR_OOM;
  s->g = g;
  s->eps = eps;
  s->conf = *conf;
  s->k = k;
  PPRHIP_CHECK_HIP(hipStreamSynchronize(g->stream));
  g->stream_open = true;
  g->stream_obj = s.get();
  try {
    s->driver = std::thread(stream_driver, s.get());
  } catch (...) {
    g->stream_open = false;
    g->stream_obj = nullptr;
    set_error("pprhip_fora_stream_open: no thread for the driver");
    return PPRHIP_ERR_OOM;
  }
  *stream_out = s.release();
  return PPRHIP_OK;
}

int pprhip_fora_stream_submit(pprhip_stream_t* s, const int32_t* srcs, int q, uint64_t seed, pprhip_results_t* keep,
                              int keep_first, int32_t* ids_out, double* vals_out, int* n_out, uint64_t* ticket_out) {
  if (!s || !ticket_out || q < 1 || !srcs || (s->k > 0 && (!ids_out || !vals_out)) || keep_first < 0) {
    set_error("pprhip_fora_stream_submit: bad arguments (q=%d)", q);
    return PPRHIP_ERR_INVALID;
  }
  if (keep && (keep->g != s->g || (long long)keep_first + q > keep->capacity)) {
    set_error("pprhip_fora_stream_submit: the result store belongs to another graph or holds %d < %d + %d queries",
              keep->capacity, keep_first, q);
    return PPRHIP_ERR_INVALID;
  }
  if (!s->g) {
    set_error("pprhip_fora_stream_submit: the stream's graph has been destroyed");
    return PPRHIP_ERR_STATE;
  }
  for (int i = 0; i < q; ++i) PPRHIP_TRY(check_node(s->g, srcs[i], "pprhip_fora_stream_submit"));
  std::shared_ptr<StreamJob> J;
  try {
    J = std::make_shared<StreamJob>();
    J->own_srcs.assign(srcs, srcs + q);
  } catch (const std::bad_alloc&) {
    return PPRHIP_ERR_OOM;
  }
  J->P = s->g;
  J->srcs = J->own_srcs.data();
  J->q = q;
  J->eps = s->eps;
  J->conf = &s->conf;
  J->seed = seed;
  J->n_rounds = 0;
  J->reserve_out = nullptr;
  J->k = s->k;
  J->ids_out = ids_out;
  J->vals_out = vals_out;
  J->n_out = n_out;
  J->per_query = nullptr;
  J->keep = keep;
  J->keep_first = keep_first;
  std::memset(&J->sum, 0, sizeof J->sum);
  J->t0 = std::chrono::steady_clock::now();
  std::lock_guard<std::mutex> lk(s->mu);
  if (s->err != PPRHIP_OK) return stream_error(s, "pprhip_fora_stream_submit");
  if (s->closing) {
    set_error("pprhip_fora_stream_submit: the stream is closing");
    return PPRHIP_ERR_STATE;
  }
  J->ticket = s->next_ticket;
  try {  // (no exception leaves the C ABI)
    s->open[J->ticket] = J;
    s->pending.push_back(J);
  } catch (const std::bad_alloc&) {
    s->open.erase(J->ticket);
    set_error("pprhip_fora_stream_submit: out of host memory");
    return PPRHIP_ERR_OOM;
  }
  s->next_ticket++;
  if (keep && keep->count < keep_first + q) keep->count = keep_first + q;
  *ticket_out = J->ticket;
  s->cv_work.notify_one();
  return PPRHIP_OK;
}

int pprhip_fora_stream_wait(pprhip_stream_t* s, uint64_t ticket, pprhip_stats_t* stats_sum) {
  if (!s) {
    set_error("pprhip_fora_stream_wait: null stream");
    return PPRHIP_ERR_INVALID;
  }
  std::unique_lock<std::mutex> lk(s->mu);
  auto it = s->open.find(ticket);
  if (it == s->open.end()) {
    set_error("pprhip_fora_stream_wait: no open submission with ticket %llu", (unsigned long long)ticket);
    return PPRHIP_ERR_INVALID;
  }
  std::shared_ptr<StreamJob> J = it->second;
  s->cv_done.wait(lk, [&] { return J->done; });
  s->open.erase(ticket);
  if (s->err != PPRHIP_OK) return stream_error(s, "pprhip_fora_stream_wait");
  if (stats_sum) *stats_sum = J->sum;
  return PPRHIP_OK;
}

// Ends the driver thread and takes the stream off its graph; the stream object stays (a later close frees it).
static int stream_shutdown(pprhip_stream* s) {
  pprhip_graph* g = s->g;
  if (!g) return s->err;
  {
    std::lock_guard<std::mutex> lk(s->mu);
    s->closing = true;
    s->cv_work.notify_all();
  }
  if (s->driver.joinable()) s->driver.join();  // every submitted query has finished (or the stream has failed)
  g->stream_open = false;
  g->stream_obj = nullptr;
  s->g = nullptr;
  if (s->err != PPRHIP_OK) {
    (void)hipSetDevice(g->device);
    free_batch(g);  // slots may hold half-pushed levels: the next batched call builds clean ones
  }
  return s->err;
}

namespace pprhip {
namespace detail {
// pprhip_graph_destroy on a handle whose stream is still open: the driver thread uses the handle, so it ends first.  The
// stream object is not freed here - its owner may still call pprhip_fora_stream_close on it (which then only frees it).
void stream_detach(void* stream_obj) {
  pprhip_stream* s = static_cast<pprhip_stream*>(stream_obj);
  (void)stream_shutdown(s);
  std::lock_guard<std::mutex> lk(s->mu);
  if (s->err == PPRHIP_OK) {
    s->err = PPRHIP_ERR_STATE;
    s->errmsg = "the stream's graph has been destroyed";
  }
  for (auto& kv : s->open) kv.second->done = true;
  s->cv_done.notify_all();
}
}  // namespace detail
}  // namespace pprhip

int pprhip_fora_stream_close(pprhip_stream_t* s) {
  if (!s) return PPRHIP_OK;
  const bool attached = s->g != nullptr;
  const int rc = stream_shutdown(s);
  int out = PPRHIP_OK;
  if (attached && rc != PPRHIP_OK) {
    set_error("pprhip_fora_stream_close: the stream had failed: %s", s->errmsg.c_str());
    out = rc;
  }
  delete s;
  return out;
}
