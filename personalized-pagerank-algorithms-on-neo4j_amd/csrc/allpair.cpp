// allpair.cpp — All-Pair-Backward-Search (Base_Whole_Graph.preprocessing) and the inverted index.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <system_error>
#include <thread>

#include <sys/mman.h>
#include <unistd.h>

#include "engine_internal.hpp"

using namespace pprhip;
using namespace pprhip::detail;

// =================================================================================================
// All-Pair-Backward-Search (a9) — first correct path: one backward search per target on the
// global arrays, entries >= threshold compacted on the device, inverted index built on the host.
// =================================================================================================
struct pprhip_index {
  uint32_t n = 0;
  RawVec<uint64_t> offsets;
  RawVec<int32_t> targets;
  RawVec<double> values;
};

namespace {

// Host threads for the index finalisation: what the process may really use at once (lift.cpp: host_threads - CPU
// affinity and cgroup quota; the GPU boxes give a one-GPU job 16 of 256 hardware threads, and more threads than that
// are throttled, not added).
static unsigned finalise_threads() { return host_threads(); }

// Base_Whole_Graph.java:112-163: per source, k < 0 keeps insertion (target) order; k >= 0 keeps
// entries >= the k-th largest (all when fewer than k) sorted descending (stable: ties stay in
// target order).
void finalize_rows(uint32_t n, std::vector<Triple>& tr, int k, pprhip_index* ix) {
  ix->n = n;
  // bucket by source, then every bucket on its own: order by target, apply the k rule.  The bucketing is a two-level
  // counting sort so that it runs on all threads: entries go to 256 coarse ranges of sources first (per-thread
  // histograms, sequential writes), then every coarse range is sorted by source on its own (a working set of
  // n / 256 counters); one thread's scatter over all n sources was a third of the call at 32 M entries.
  const size_t N = tr.size();
  std::vector<uint64_t> start((size_t)n + 1, 0);
  std::vector<Triple> by_v(N);
  const unsigned hw = finalise_threads();
  const unsigned T = N < (1u << 16) ? 1u : hw;
  auto parallel = [&](unsigned parts, auto&& fn) {  // fn(part) for part in [0, parts), T threads
    std::atomic<unsigned> next{0};
    auto work = [&]() {
      for (unsigned p = next.fetch_add(1); p < parts; p = next.fetch_add(1)) fn(p);
    };
    std::vector<std::thread> th;
    for (unsigned w = 1; w < T; ++w) th.emplace_back(work);
    work();
    for (auto& x : th) x.join();
  };
  {
    constexpr unsigned kCoarse = 256;
    const uint32_t span = (uint32_t)(((uint64_t)n + kCoarse - 1) / kCoarse);  // sources per coarse range
    const unsigned chunks = T;
    std::vector<uint64_t> hist((size_t)chunks * kCoarse, 0);
    auto chunk_lo = [&](unsigned c) { return (size_t)((unsigned long long)N * c / chunks); };
    parallel(chunks, [&](unsigned c) {
      uint64_t* h = hist.data() + (size_t)c * kCoarse;
      for (size_t i = chunk_lo(c); i < chunk_lo(c + 1); ++i) h[(uint32_t)tr[i].v / span]++;
    });
    // coarse range b of chunk c starts at: all of ranges < b, then chunks < c of range b
    std::vector<uint64_t> base((size_t)chunks * kCoarse, 0), cstart(kCoarse + 1, 0);
    uint64_t run = 0;
    for (unsigned b = 0; b < kCoarse; ++b) {
      cstart[b] = run;
      for (unsigned c = 0; c < chunks; ++c) {
        base[(size_t)c * kCoarse + b] = run;
        run += hist[(size_t)c * kCoarse + b];
      }
    }
    cstart[kCoarse] = run;
    std::vector<Triple> coarse(N);
    parallel(chunks, [&](unsigned c) {
      uint64_t* at = base.data() + (size_t)c * kCoarse;
      for (size_t i = chunk_lo(c); i < chunk_lo(c + 1); ++i) coarse[at[(uint32_t)tr[i].v / span]++] = tr[i];
    });
    std::vector<Triple>().swap(tr);
    parallel(kCoarse, [&](unsigned b) {
      const uint32_t v_lo = std::min<uint64_t>((uint64_t)b * span, n), v_hi = std::min<uint64_t>((uint64_t)(b + 1) * span, n);
      if (v_lo >= v_hi) return;
      std::vector<uint64_t> cnt((size_t)(v_hi - v_lo) + 1, 0);
      for (uint64_t i = cstart[b]; i < cstart[b + 1]; ++i) cnt[(uint32_t)coarse[i].v - v_lo + 1]++;
      uint64_t acc = cstart[b];  // entries of sources below v_lo = entries of the coarse ranges below b
      for (uint32_t v = v_lo; v < v_hi; ++v) {
        start[v] = acc;
        acc += cnt[v - v_lo + 1];
        cnt[v - v_lo + 1] = start[v];  // becomes the write cursor of source v
      }
      for (uint64_t i = cstart[b]; i < cstart[b + 1]; ++i) by_v[cnt[(uint32_t)coarse[i].v - v_lo + 1]++] = coarse[i];
    });
    start[n] = N;
  }
  std::vector<uint64_t> kept((size_t)n + 1, 0);
  auto for_ranges = [&](auto&& fn) {
    std::vector<std::thread> th;
    for (unsigned w = 1; w < T; ++w) th.emplace_back(fn, (uint32_t)((uint64_t)n * w / T), (uint32_t)((uint64_t)n * (w + 1) / T));
    fn(0u, (uint32_t)((uint64_t)n / T));
    for (auto& x : th) x.join();
  };
  // pass 1: each bucket sorted by target; for k >= 0 the kept entries move to the bucket's front, by value
  for_ranges([&](uint32_t lo, uint32_t hi) {
    std::vector<double> tmp;
    for (uint32_t v = lo; v < hi; ++v) {
      Triple* b = by_v.data() + start[v];
      const size_t len = (size_t)(start[v + 1] - start[v]);
      if (len == 0) continue;
      std::sort(b, b + len, [](const Triple& x, const Triple& y) { return x.t < y.t; });
      if (k < 0) {
        kept[v + 1] = len;
        continue;
      }
      bool have = false;
      double kth = 0.0;
      if (k >= 1 && (size_t)k <= len) {
        tmp.resize(len);
        for (size_t j = 0; j < len; ++j) tmp[j] = b[j].p;
        std::nth_element(tmp.begin(), tmp.begin() + (k - 1), tmp.end(), std::greater<double>());
        kth = tmp[k - 1];
        have = true;
      }
      size_t w = 0;
      for (size_t j = 0; j < len; ++j)
        if (!have || b[j].p >= kth) b[w++] = b[j];
      std::stable_sort(b, b + w, [](const Triple& x, const Triple& y) { return x.p > y.p; });
      kept[v + 1] = w;
    }
  });
  for (uint32_t v = 0; v < n; ++v) kept[v + 1] += kept[v];
  ix->targets.resize(kept[n]);
  ix->values.resize(kept[n]);
  // pass 2: into the index arrays
  for_ranges([&](uint32_t lo, uint32_t hi) {
    for (uint32_t v = lo; v < hi; ++v) {
      const Triple* b = by_v.data() + start[v];
      const size_t len = (size_t)(kept[v + 1] - kept[v]);
      for (size_t j = 0; j < len; ++j) {
        ix->targets[kept[v] + j] = b[j].t;
        ix->values[kept[v] + j] = b[j].p;
      }
    }
  });
  ix->offsets.assign(kept.begin(), kept.end());
}

}  // namespace

namespace pprhip {
namespace detail {

// entries of the searches go to the host right away (single-GPU call): the device records have the host entries'
// layout, so they land in the vector's tail in one copy ...
static_assert(sizeof(Triple) == sizeof(TripleRec), "host and device entries share one layout");
int HostTripleSink::take_device(pprhip_graph* g, const TripleRec* d_rec, unsigned long long count) {
  const size_t at = tr.size();
  tr.resize(at + count);
  PPRHIP_CHECK_HIP(hipMemcpyAsync(tr.data() + at, d_rec, sizeof(TripleRec) * count, hipMemcpyDeviceToHost, g->stream));
  PPRHIP_CHECK_HIP(hipStreamSynchronize(g->stream));
  return PPRHIP_OK;
}
int HostTripleSink::take_host(pprhip_graph*, std::vector<Triple>& more) {
  tr.insert(tr.end(), more.begin(), more.end());
  return PPRHIP_OK;
}

// ... or stay in HBM as 16-byte records until the exchange by owner of the source (sharded call)
int DeviceTripleSink::reserve(pprhip_graph* g, unsigned long long extra) {
  if (count + extra <= cap) return PPRHIP_OK;
  unsigned long long ncap = std::max<unsigned long long>(cap * 2, std::max<unsigned long long>(count + extra, 1ull << 20));
  TripleRec* nrec = nullptr;
  PPRHIP_TRY(alloc_dev((void**)&nrec, sizeof(TripleRec) * ncap));
  if (count)
    PPRHIP_CHECK_HIP(hipMemcpyAsync(nrec, rec, sizeof(TripleRec) * count, hipMemcpyDeviceToDevice, g->stream));
  PPRHIP_CHECK_HIP(hipStreamSynchronize(g->stream));
  if (rec) (void)hipFree(rec);
  rec = nrec;
  cap = ncap;
  return PPRHIP_OK;
}
int DeviceTripleSink::take_device(pprhip_graph* g, const TripleRec* d_rec, unsigned long long n_new) {
  PPRHIP_TRY(reserve(g, n_new));
  PPRHIP_CHECK_HIP(hipMemcpyAsync(rec + count, d_rec, sizeof(TripleRec) * n_new, hipMemcpyDeviceToDevice, g->stream));
  PPRHIP_CHECK_HIP(hipStreamSynchronize(g->stream));  // the source buffer is reused by the next pass
  count += n_new;
  return PPRHIP_OK;
}
int DeviceTripleSink::take_host(pprhip_graph* g, std::vector<Triple>& more) {
  if (more.empty()) return PPRHIP_OK;
  PPRHIP_TRY(reserve(g, more.size()));
  PPRHIP_CHECK_HIP(hipMemcpyAsync(rec + count, more.data(), sizeof(TripleRec) * more.size(), hipMemcpyHostToDevice,
                                  g->stream));
  PPRHIP_CHECK_HIP(hipStreamSynchronize(g->stream));
  count += more.size();
  return PPRHIP_OK;
}
DeviceTripleSink::~DeviceTripleSink() {
  if (rec) (void)hipFree(rec);
}

// Base_Whole_Graph.java:76-92 for the targets [t_begin, t_end): every backward search's entries >= threshold go to
// `sink`; the three tiers as described in kernels_apbs.hip.
int all_pair_collect(pprhip_graph_t* g, double alpha, double threshold, uint32_t t_begin, uint32_t t_end,
                     TripleSink& sink, pprhip_stats_t& st) {
  g->topk_active = false;
  CallTimer tm(g);
  const uint32_t n_targets = t_end - t_begin;
  // PPRHIP_APBS_TIER = 2 / 3 starts at a later tier (tests exercise every tier that way)
  const int first_tier = hook_env("PPRHIP_APBS_TIER") ? atoi(hook_env("PPRHIP_APBS_TIER")) : 1;

  // ---- device buffers of this call
  ApbsBuffers B;
  unsigned long long* cells = nullptr;  // next_target, out_count, out_valid, overflow_count, pops, edges
  int rc = PPRHIP_OK;
  auto release = [&]() {
    void* p[] = {cells, B.out_rec, B.overflow, B.list0, B.list1};
    for (void* q : p)
      if (q) (void)hipFree(q);
  };
  // room for the entries of one pass over the range (16 bytes each; 2 GB at most): a search that finds the buffer
  // full is not run at all but listed for the next pass
  // The buffer must hold what ONE search can emit (up to n entries: a hub's column), or that search would find it
  // full on every pass: until round 5 a range of a few hub targets - 6 of R-MAT 18's in a work-weighted rank's share -
  // was sized for 65 536 entries, the hubs' searches were repeated a thousand times (44 G edge pushes) and then
  // dropped without an error.  Four entries per node also keeps a range of hubs from running pass after pass.
  B.out_cap = std::min<unsigned long long>(
      1ull << 27, std::max<unsigned long long>(std::max<unsigned long long>(1ull << 16, 16ull * n_targets), 4ull * g->n + 1024));
  if ((rc = alloc_dev((void**)&cells, sizeof(unsigned long long) * 16)) ||
      (rc = alloc_dev((void**)&B.out_rec, sizeof(TripleRec) * B.out_cap)) ||
      (rc = alloc_dev((void**)&B.overflow, sizeof(int32_t) * std::max<uint32_t>(1, n_targets))) ||
      // (tier 1 over a range: the list of targets with in-edges and the small table's give-ups, kernels_apbs.hip)
      (rc = alloc_dev((void**)&B.list0, sizeof(int32_t) * std::max<uint32_t>(1, n_targets))) ||
      (rc = alloc_dev((void**)&B.list1, sizeof(int32_t) * std::max<uint32_t>(1, n_targets)))) {
    release();
    return rc;
  }
  B.next_target = cells;
  B.out_count = cells + 1;
  B.out_valid = cells + 2;
  B.overflow_count = cells + 3;
  B.stat_pops = cells + 4;
  B.stat_edges = cells + 5;
  std::vector<int32_t> h_ovf;
  unsigned long long h_cells[16];

  // runs one tier over `list` (or the range when list is empty and use_range) until every target
  // has either produced its triples or landed in `give_up`
  auto run_tier = [&](bool dense_tier, std::vector<int32_t> list, bool use_range, std::vector<int32_t>& give_up) -> int {
    int32_t* d_list = nullptr;
    struct ListGuard {  // frees the target list on every exit, error returns included
      int32_t*& p;
      ~ListGuard() {
        if (p) (void)hipFree(p);
        p = nullptr;
      }
    } list_guard{d_list};
    for (int pass = 0; pass < 1000; ++pass) {
      const uint32_t cnt = use_range ? n_targets : (uint32_t)list.size();
      if (cnt == 0) break;
      if (!use_range) {
        if (!d_list) PPRHIP_TRY(alloc_dev((void**)&d_list, sizeof(int32_t) * list.size()));
        PPRHIP_CHECK_HIP(hipMemcpyAsync(d_list, list.data(), sizeof(int32_t) * cnt, hipMemcpyHostToDevice, g->stream));
      }
      const unsigned long long init[16] = {0, 0, ~0ull, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      PPRHIP_CHECK_HIP(hipMemcpyAsync(cells, init, sizeof init, hipMemcpyHostToDevice, g->stream));
      if (dense_tier)  // every board entry closed, nothing posted
        PPRHIP_CHECK_HIP(hipMemsetAsync(B.board, 0, apbs_board_bytes(B.ws_blocks), g->stream));
      ktimer().begin(PPRHIP_KERNEL_BACKWARD_BATCH, 0);
      PPRHIP_TRY(launch_apbs(g, dense_tier, use_range ? nullptr : d_list, t_begin, cnt, alpha, threshold, B));
      ktimer().end();
      PPRHIP_CHECK_HIP(hipMemcpyAsync(h_cells, cells, sizeof h_cells, hipMemcpyDeviceToHost, g->stream));
      PPRHIP_CHECK_HIP(hipStreamSynchronize(g->stream));
      if (h_cells[10]) {
        set_error("All-Pair dense tier: a workgroup waited more than 30 s for the chunks of a posted level (launch aborted)");
        return PPRHIP_ERR_STATE;
      }
      const unsigned long long valid = std::min(std::min(h_cells[1], h_cells[2]), B.out_cap);
      st.pops += h_cells[4];
      st.edge_pushes += h_cells[5];
      const uint64_t bytes = 44ull * h_cells[4] + 28ull * h_cells[5] + 16ull * valid;
      st.push_bytes += bytes;
      if (!ktimer().recs.empty()) ktimer().recs.back().bytes = bytes;
      if (valid) PPRHIP_TRY(sink.take_device(g, B.out_rec, valid));
      std::vector<int32_t> again;
      const unsigned long long novf = h_cells[3];
      if (novf) {
        h_ovf.resize(novf);
        PPRHIP_CHECK_HIP(hipMemcpy(h_ovf.data(), B.overflow, sizeof(int32_t) * novf, hipMemcpyDeviceToHost));
        for (int32_t x : h_ovf) {
          if (x >= 0) give_up.push_back(x);  // table too small for this target
          else again.push_back(-(x + 1));    // triple buffer was full: same tier again
        }
      }
      // a search that found the buffer full although it came first in an empty one cannot ever fit: an error, not a
      // silent loss of its entries (a pass that emitted nothing and still lists searches for another pass)
      if (!again.empty() && valid == 0 && again.size() == (use_range ? (size_t)n_targets : list.size())) {
        set_error("All-Pair: a search yields more than the %llu entries the record buffer holds", B.out_cap);
        return PPRHIP_ERR_STATE;
      }
      list.swap(again);
      use_range = false;
      if (d_list && list.size()) {
        (void)hipFree(d_list);
        d_list = nullptr;
      }
      if (pass == 999 && !list.empty()) {
        set_error("All-Pair: %zu searches still waited for room in the record buffer after 1000 passes", list.size());
        return PPRHIP_ERR_STATE;
      }
    }
    return PPRHIP_OK;
  };

  // in-edge records for both tiers' edge loops (8 B per edge; stays with the handle)
  if (!g->in_rec) {
    void* rec = nullptr;
    if ((rc = alloc_dev(&rec, sizeof(unsigned long long) * std::max<uint64_t>(1, g->m))) == PPRHIP_OK &&
        (rc = launch_build_in_rec(g, rec)) == PPRHIP_OK)
      g->in_rec = rec;
    else if (rec)
      (void)hipFree(rec);
    if (rc != PPRHIP_OK) {
      release();
      return rc;
    }
  }
  // the sweep layout over the out-CSR that the whole-vector searches' dense levels need: built with the handle's other
  // first-use work, not inside a later call's searches
  if ((rc = ensure_bwd_layout(g)) != PPRHIP_OK) {
    release();
    return rc;
  }
  const bool dbg_times = hook_env("PPRHIP_APBS_DEBUG") != nullptr;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms_since = [&](std::chrono::steady_clock::time_point t) {
    return std::chrono::duration<double, std::milli>(now() - t).count();
  };
  auto t_phase = now();
  std::vector<int32_t> to_tier2, to_tier3;
  // the dense tier for a list of targets (workspaces on first use); what outgrows its lists is appended to to_tier3
  auto dense_pass = [&](std::vector<int32_t>& list) {
    // Dense workspaces, one per workgroup in flight (kernels_apbs.hip): 16n bytes of vectors + lists.  The lists hold
    // what a search may list before it is handed to tier 3: nodes whose residue left zero (clean-up; on overflow the
    // whole vector is cleared instead) and a level's frontier.  The workspaces stay with the handle: allocating and
    // zeroing gigabytes per call would cost more than the searches of a small target range.
    if (g->apbs_blocks == 0) {
      // (PPRHIP_APBS_CAP_T / _CAP_F shrink the lists so that tests reach the overflow paths on small graphs,
      // PPRHIP_APBS_CHUNK the chunks of a level's edge space so that small graphs' levels are shared too)
      const char* e_t = hook_env("PPRHIP_APBS_CAP_T");
      const char* e_f = hook_env("PPRHIP_APBS_CAP_F");
      const char* e_c = hook_env("PPRHIP_APBS_CHUNK");
      const uint32_t chunk = e_c ? (uint32_t)std::max(16, atoi(e_c)) : apbs_default_chunk();
      const uint32_t cap_t = e_t ? (uint32_t)std::max(1, atoi(e_t)) : std::min<uint32_t>(g->n, 1u << 20) + 4096u;
      const uint32_t cap_f = e_f ? (uint32_t)std::max(1, atoi(e_f)) : std::min<uint32_t>(g->n, 1u << 20) + 64u;
      const char* per_cu = hook_env("PPRHIP_APBS_WGS_PER_CU");
      uint32_t want = (uint32_t)g->n_cus * (uint32_t)std::max(1, std::min(2, per_cu ? atoi(per_cu) : 1));
      const size_t per = apbs_dense_bytes(g->n, g->m, cap_t, cap_f, chunk);
      int arc = PPRHIP_ERR_OOM;
      // a device that cannot spare them all runs the tier with fewer workgroups in flight
      for (; want >= 8; want /= 2) {
        arc = alloc_dev((void**)&g->apbs_ws, (size_t)want * per);
        if (arc != PPRHIP_ERR_OOM) break;
        (void)hipGetLastError();
      }
      if (arc == PPRHIP_OK && hipMemsetAsync(g->apbs_ws, 0, (size_t)want * per, g->stream) != hipSuccess) {
        (void)hipFree(g->apbs_ws);
        g->apbs_ws = nullptr;
        set_error("All-Pair: clearing the dense workspaces failed");
        arc = PPRHIP_ERR_HIP;
      }
      if (arc == PPRHIP_OK && (arc = alloc_dev(&g->apbs_board, apbs_board_bytes(want))) != PPRHIP_OK) {
        (void)hipFree(g->apbs_ws);
        g->apbs_ws = nullptr;
      }
      if (arc == PPRHIP_OK) {
        g->apbs_blocks = want;
        g->apbs_cap_t = cap_t;
        g->apbs_cap_f = cap_f;
        g->apbs_chunk = chunk;
      } else if (arc != PPRHIP_ERR_OOM) {
        rc = arc;
      }
    }
    if (rc == PPRHIP_OK && g->apbs_blocks) {
      B.ws = g->apbs_ws;
      B.ws_blocks = g->apbs_blocks;
      B.cap_t = g->apbs_cap_t;
      B.cap_f = g->apbs_cap_f;
      B.chunk = g->apbs_chunk;
      B.helpers = g->apbs_blocks;
      B.board = g->apbs_board;
      B.done_targets = cells + 8;  // + 8: targets done, + 9: levels posted, + 10: abort word
      // targets with the most in-edges first: the searches that push the most edges start the level-1 fan-out from
      // hubs, and a workgroup that draws such a search last would finish long after the others
      // (a stable counting sort by in-degree, degrees from 65535 up in one bucket that is sorted on its own: a
      // comparison sort of half a million ids with two indirections per comparison was 15-40 ms of every pass)
      {
        const std::vector<uint32_t>& irp = g->h_in_rp;
        const std::vector<int32_t>& o2n = g->h_old2new;
        constexpr uint32_t kCapDeg = 65535;
        const size_t L = list.size();
        std::vector<uint32_t> deg(L);
        std::vector<uint32_t> at((size_t)kCapDeg + 2, 0);
        for (size_t i = 0; i < L; ++i) {
          const int32_t a = g->relabeled ? o2n[list[i]] : list[i];
          deg[i] = irp[a + 1] - irp[a];
          at[kCapDeg - std::min(deg[i], kCapDeg) + 1]++;  // bucket 0: the largest degrees
        }
        for (uint32_t b = 0; b <= kCapDeg; ++b) at[b + 1] += at[b];
        const uint32_t n_top = at[1];
        std::vector<int32_t> sorted(L);
        std::vector<uint32_t> sdeg(n_top);
        for (size_t i = 0; i < L; ++i) {
          const uint32_t b = kCapDeg - std::min(deg[i], kCapDeg);
          const uint32_t pos = at[b]++;
          sorted[pos] = list[i];
          if (b == 0) sdeg[pos] = deg[i];
        }
        if (n_top > 1) {  // the top bucket by exact degree (stable)
          std::vector<uint32_t> idx(n_top);
          for (uint32_t i = 0; i < n_top; ++i) idx[i] = i;
          std::stable_sort(idx.begin(), idx.end(), [&](uint32_t x, uint32_t y) { return sdeg[x] > sdeg[y]; });
          std::vector<int32_t> top(n_top);
          for (uint32_t i = 0; i < n_top; ++i) top[i] = sorted[idx[i]];
          std::copy(top.begin(), top.end(), sorted.begin());
        }
        list.swap(sorted);
      }
      // Developer switch PPRHIP_APBS_DEBUG: per-workgroup timers and a progress word in HOST memory, and a watchdog
      // thread that prints the progress words and ends the process when the tier has not come back after 20 s
      // (a kernel that never ends would otherwise only be seen as a process that cannot be killed).
      const bool debug = hook_env("PPRHIP_APBS_DEBUG") != nullptr;
      std::mutex wd_mu;
      std::condition_variable wd_cv;
      bool wd_done = false;
      std::thread watchdog;
      if (debug && hipHostMalloc((void**)&B.dbg, sizeof(unsigned long long) * 12 * g->apbs_blocks, hipHostMallocMapped) ==
                       hipSuccess) {
        std::memset(B.dbg, 0, sizeof(unsigned long long) * 12 * g->apbs_blocks);
        const unsigned long long* rows = B.dbg;
        const uint32_t nb = g->apbs_blocks;
        watchdog = std::thread([&wd_mu, &wd_cv, &wd_done, rows, nb] {
          std::unique_lock<std::mutex> lk(wd_mu);
          if (wd_cv.wait_for(lk, std::chrono::seconds(20), [&] { return wd_done; })) return;
          fprintf(stderr, "[apbs dense] no end after 20 s; workgroup: stage/detail (1 target, 2 pops, 3 own chunk, 4 waiting "
                          "for helpers, 5 local chunks, 6 emit, 7 clear, 8 idle, 9 helping owner<<16|chunk, 10 out)\n");
          for (uint32_t w = 0; w < nb; ++w)
            if (rows[12 * w + 9])
              fprintf(stderr, "%u: %llu/%llu%s", w, rows[12 * w + 9] >> 32, rows[12 * w + 9] & 0xffffffffull,
                      (w % 8 == 7) ? "\n" : "   ");
          fprintf(stderr, "\n");
          fflush(stderr);
          _exit(3);
        });
      }
      rc = run_tier(true, list, false, to_tier3);
      if (watchdog.joinable()) {
        {
          std::lock_guard<std::mutex> lk(wd_mu);
          wd_done = true;
        }
        wd_cv.notify_all();
        watchdog.join();
      }
      if (B.dbg) {
        std::vector<unsigned long long> h(B.dbg, B.dbg + (size_t)12 * g->apbs_blocks);
        {
          unsigned long long tot[8] = {0}, t_end_max = 0, t_end_min = ~0ull, e_max = 0;
          for (uint32_t w = 0; w < g->apbs_blocks; ++w) {
            if (!h[12 * w + 8]) continue;
            for (int i = 0; i < 8; ++i) tot[i] += h[12 * w + i];
            t_end_max = std::max(t_end_max, h[12 * w + 8]);
            t_end_min = std::min(t_end_min, h[12 * w + 8]);
            e_max = std::max(e_max, h[12 * w + 1]);
          }
          fprintf(stderr, "[apbs dense] searches %llu edges %llu (max owned by one workgroup %llu); workgroup-ms in pops+scans %.1f "
                          "own chunks %.1f waiting for helpers %.1f emit %.1f clear %.1f helping / idle %.1f; first workgroup "
                          "ended %.2f ms before the last\n",
                  tot[0], tot[1], e_max, tot[2] / 1e5, tot[3] / 1e5, tot[4] / 1e5, tot[5] / 1e5, tot[6] / 1e5, tot[7] / 1e5,
                  (t_end_max - t_end_min) / 1e5);
        }
        (void)hipHostFree(B.dbg);
        B.dbg = nullptr;
      }
    } else if (rc == PPRHIP_OK) {
      to_tier3.insert(to_tier3.end(), list.begin(), list.end());  // no memory for the dense tier: everything runs on the batch slots
    }
  };
  // (Rounds 3 and 4 ran tier 1 of the next third of a large range on a side stream beside the dense pass of the third
  // before.  LDS searches beside a dense pass slow it down by nearly their own duration, so it saved 5 % when tier 1 was a
  // quarter of the job; since tier 1 routes its targets by in-degree it is a seventh of it, and the tiers one after the
  // other are as fast or faster - R-MAT 22: 655 against 670 ms, R-MAT 24: 1 894 against 1 910 ms - with one pass of each
  // tier instead of three.  The side-by-side form was taken out.)
  if (first_tier <= 1) {
    rc = run_tier(false, {}, true, to_tier2);
    if (dbg_times) fprintf(stderr, "[apbs host] tier 1 (kernel passes + hand-over of entries): %.1f ms\n", ms_since(t_phase));
    t_phase = now();
  } else {
    for (uint32_t t = t_begin; t < t_end; ++t) (first_tier == 2 ? to_tier2 : to_tier3).push_back((int32_t)t);
  }
  if (rc == PPRHIP_OK && !to_tier2.empty()) {
    dense_pass(to_tier2);
    // ---- the searches whose frontier or popped-node list outgrew the workspaces' lists: once more with a few
    // workspaces whose lists hold every node, all the other workgroups helping with their levels
    st.xl_targets = (uint32_t)to_tier3.size();  // searches that outgrew a workspace's lists
    // ---- A handful of such searches (R-MAT 22: the one target with 160 K in-edges, whose search pushes 300 M edges)
    // run best one at a time on the handle's OWN vectors with the whole chip behind each level: levels that touch a
    // large part of the graph as pull sweeps over the out-CSR (no atomics at all), the others as sparse pushes -
    // pprhip_backward_push's path.  Measured (tools/exp/apbs_big_searches.py): 2.5 ms of device time for that target
    // against 158 ms in the full-size pass below, where one workgroup owns the search and the others help with its
    // levels at the rate of memory-side atomics.  The entries go from the reserve vector into records on the device.
    const char* whole_env = hook_env("PPRHIP_APBS_WHOLE");
    const size_t whole_max = whole_env ? (size_t)std::max(0, atoi(whole_env)) : 256;
    if (rc == PPRHIP_OK && !to_tier3.empty() && to_tier3.size() <= whole_max) {
      TripleRec* d_rec = nullptr;
      unsigned long long* d_cnt = nullptr;
      const unsigned long long cap = std::max<uint32_t>(act_n(g), g->n);
      auto whole = [&]() -> int {
        PPRHIP_TRY(alloc_dev((void**)&d_rec, sizeof(TripleRec) * cap));
        PPRHIP_TRY(alloc_dev((void**)&d_cnt, sizeof(unsigned long long)));
        const std::vector<int32_t>& o2n = g->h_old2new;
        for (int32_t t_old : to_tier3) {
          pprhip_stats_t s1;
          std::memset(&s1, 0, sizeof s1);
          PPRHIP_TRY(backward_search_whole(g, g->relabeled ? o2n[t_old] : t_old, alpha, threshold, s1));
          PPRHIP_CHECK_HIP(hipMemsetAsync(d_cnt, 0, sizeof(unsigned long long), g->stream));
          PPRHIP_TRY(launch_emit_reserve(g, g->reserve, g->n, threshold, t_old, d_rec, cap, d_cnt));
          unsigned long long cnt = 0;
          PPRHIP_CHECK_HIP(hipMemcpyAsync(&cnt, d_cnt, sizeof cnt, hipMemcpyDeviceToHost, g->stream));
          PPRHIP_CHECK_HIP(hipStreamSynchronize(g->stream));
          if (cnt > cap) {
            set_error("All-Pair: a search yields more entries (%llu) than the graph has nodes", cnt);
            return PPRHIP_ERR_STATE;
          }
          if (cnt) PPRHIP_TRY(sink.take_device(g, d_rec, cnt));
          st.pops += s1.pops + s1.dense_nodes;
          st.edge_pushes += s1.edge_pushes + s1.dense_edges;
          st.levels += s1.levels;
          st.dense_levels += s1.dense_levels;
          st.push_bytes += s1.push_bytes + 16ull * cnt;
        }
        return PPRHIP_OK;
      };
      rc = whole();
      if (d_rec) (void)hipFree(d_rec);
      if (d_cnt) (void)hipFree(d_cnt);
      if (dbg_times) fprintf(stderr, "[apbs host] whole-vector searches: %zu targets\n", to_tier3.size());
      to_tier3.clear();
    }
    if (rc == PPRHIP_OK && g->apbs_blocks && !to_tier3.empty() && to_tier3.size() < to_tier2.size() &&
        !hook_env("PPRHIP_APBS_NO_XL")) {
      if (!g->apbs_xl_ws) {
        const uint32_t xl_t = g->n + 4096u, xl_f = g->n + 64u;
        uint32_t want = 4;
        int arc = PPRHIP_ERR_OOM;
        for (; want >= 1; want /= 2) {
          arc = alloc_dev((void**)&g->apbs_xl_ws, (size_t)want * apbs_dense_bytes(g->n, g->m, xl_t, xl_f, g->apbs_chunk));
          if (arc != PPRHIP_ERR_OOM) break;
          (void)hipGetLastError();
        }
        if (arc == PPRHIP_OK &&
            hipMemsetAsync(g->apbs_xl_ws, 0, (size_t)want * apbs_dense_bytes(g->n, g->m, xl_t, xl_f, g->apbs_chunk), g->stream) !=
                hipSuccess) {
          (void)hipFree(g->apbs_xl_ws);
          g->apbs_xl_ws = nullptr;
          arc = PPRHIP_ERR_HIP;
        }
        if (arc == PPRHIP_OK) {
          g->apbs_xl_blocks = want;
          g->apbs_xl_cap_t = xl_t;
          g->apbs_xl_cap_f = xl_f;
        }
      }
      if (g->apbs_xl_ws) {
        B.ws = g->apbs_xl_ws;
        B.ws_blocks = g->apbs_xl_blocks;
        B.cap_t = g->apbs_xl_cap_t;
        B.cap_f = g->apbs_xl_cap_f;
        B.helpers = g->apbs_blocks;
        std::vector<int32_t> again3;
        const bool xdebug = hook_env("PPRHIP_APBS_DEBUG") != nullptr;
        const uint32_t xnb = std::max(g->apbs_blocks, g->apbs_xl_blocks);
        if (xdebug && hipHostMalloc((void**)&B.dbg, sizeof(unsigned long long) * 12 * xnb, hipHostMallocMapped) == hipSuccess)
          std::memset(B.dbg, 0, sizeof(unsigned long long) * 12 * xnb);
        rc = run_tier(true, to_tier3, false, again3);
        if (B.dbg) {
          unsigned long long tot[8] = {0};
          for (uint32_t w = 0; w < xnb; ++w)
            for (int i = 0; i < 8; ++i) tot[i] += B.dbg[12 * w + i];
          fprintf(stderr, "[apbs dense, full-size pass] searches %llu edges %llu; workgroup-ms in pops+scans %.1f own chunks %.1f "
                          "waiting for helpers %.1f emit %.1f clear %.1f helping / idle %.1f\n",
                  tot[0], tot[1], tot[2] / 1e5, tot[3] / 1e5, tot[4] / 1e5, tot[5] / 1e5, tot[6] / 1e5, tot[7] / 1e5);
          (void)hipHostFree(B.dbg);
          B.dbg = nullptr;
        }
        if (dbg_times) fprintf(stderr, "[apbs host] full-size workspaces: %zu targets, %zu left for tier 3\n", to_tier3.size(), again3.size());
        to_tier3.swap(again3);
      }
    }
  }
  if (dbg_times) fprintf(stderr, "[apbs host] tier 2 (%zu targets): %.1f ms\n", to_tier2.size(), ms_since(t_phase));
  t_phase = now();
  release();
  if (rc != PPRHIP_OK) return rc;

  // ---- tier 3 (fallback): searches whose frontier outgrows tier 2's lists run as whole-vector backward searches,
  // 16 of them in flight on the batch slots; levels that touch a large part of the graph run as batched sweeps over
  // the out-CSR
  pprhip_stats_t st3;
  std::memset(&st3, 0, sizeof st3);
  if (!to_tier3.empty()) {  // Base_Whole_Graph.java:76-92
    std::vector<Triple> tr3;
    BatchJob J;
    J.P = g;
    J.kind = 2;
    J.srcs = to_tier3.data();
    J.q = (int)to_tier3.size();
    J.eps = 0.0;
    J.conf = nullptr;
    J.seed = 0;
    J.n_rounds = 0;
    J.reserve_out = nullptr;
    J.k = 0;
    J.ids_out = nullptr;
    J.vals_out = nullptr;
    J.n_out = nullptr;
    J.per_query = nullptr;
    J.alpha = alpha;
    J.threshold = threshold;
    J.triples = &tr3;
    PPRHIP_TRY(batch_run(g, J, &st3));
    PPRHIP_TRY(sink.take_host(g, tr3));
    st.pops += st3.pops;
    st.edge_pushes += st3.edge_pushes;
    st.enqueues += st3.enqueues;
    st.levels += st3.levels;
    st.dense_levels += st3.dense_levels;
    st.push_bytes += st3.push_bytes;
  }
  if (dbg_times) fprintf(stderr, "[apbs host] tier 3 (%zu targets): %.1f ms\n", to_tier3.size(), ms_since(t_phase));
  tm.mark(1);
  tm.finish(st);
  for (int c = 0; c < 8; ++c) {
    st.class_ms[c] += st3.class_ms[c];
    st.class_bytes[c] += st3.class_bytes[c];
    st.class_launches[c] += st3.class_launches[c];
  }
  st.push_ms = CallTimer::ms(g->ev[0], g->ev[1]);
  st.rmax_final = threshold;
  st.rounds = (uint32_t)(to_tier2.size());      // targets that needed the dense tier
  st.dense_nodes = (uint64_t)to_tier3.size();   // targets that needed the whole-vector path
  return PPRHIP_OK;
}

// ---- device -> pageable host memory through a ring of pinned slots and copier threads
constexpr int kIxSlots = 8;
constexpr size_t kIxSlotBytes = 8u << 20;
constexpr int kIxCopiers = 4;

int ensure_ring(pprhip_graph* g) {  // the ring lives in g->ix_stage (kIxSlots * kIxSlotBytes of pinned memory)
  if (g->ix_stage && g->ix_stage_bytes >= kIxSlots * kIxSlotBytes) return PPRHIP_OK;
  if (g->ix_stage) (void)hipHostFree(g->ix_stage);
  g->ix_stage = nullptr;
  g->ix_stage_bytes = 0;
  if (hipHostMalloc(&g->ix_stage, kIxSlots * kIxSlotBytes, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    g->ix_stage = nullptr;
    return PPRHIP_ERR_OOM;  // (the caller falls back to a plain copy)
  }
  g->ix_stage_bytes = kIxSlots * kIxSlotBytes;
  return PPRHIP_OK;
}

int ring_download(pprhip_graph* g, const void* d_src, void* h_dst, size_t bytes) {
  if (!bytes) return PPRHIP_OK;
  if (bytes < 4 * kIxSlotBytes || ensure_ring(g) != PPRHIP_OK) {  // small, or no pinned memory to be had
    PPRHIP_CHECK_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, g->stream));
    PPRHIP_CHECK_HIP(hipStreamSynchronize(g->stream));
    return PPRHIP_OK;
  }
  char* const ring = static_cast<char*>(g->ix_stage);
  hipEvent_t ev[kIxSlots] = {};
  for (int i = 0; i < kIxSlots; ++i)
    if (hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) {
      for (int j = 0; j < i; ++j) (void)hipEventDestroy(ev[j]);
      set_error("index download: no events");
      return PPRHIP_ERR_HIP;
    }
  const size_t n_chunks = (bytes + kIxSlotBytes - 1) / kIxSlotBytes;
  std::mutex mu;
  std::condition_variable cv;
  size_t issued = 0;                 // chunks whose copy into their slot has been queued
  size_t taken = 0;                  // next chunk a copier takes
  size_t freed[kIxSlots] = {};       // per slot: chunks of that slot moved on so far
  int err = PPRHIP_OK;
  const int device = g->device;
  auto copier = [&] {
    (void)hipSetDevice(device);
    for (;;) {
      size_t c;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return taken < issued || taken >= n_chunks || err; });
        if (err || taken >= n_chunks) return;
        c = taken++;
      }
      const int slot = (int)(c % kIxSlots);
      const size_t off = c * kIxSlotBytes, len = std::min(kIxSlotBytes, bytes - off);
      const bool ok = hipEventSynchronize(ev[slot]) == hipSuccess;
      if (ok) std::memcpy(static_cast<char*>(h_dst) + off, ring + (size_t)slot * kIxSlotBytes, len);
      {
        std::lock_guard<std::mutex> lk(mu);
        if (!ok && !err) err = PPRHIP_ERR_HIP;
        freed[slot]++;
      }
      cv.notify_all();
    }
  };
  std::thread th[kIxCopiers];
  int n_th = 0;
  try {
    for (; n_th < kIxCopiers; ++n_th) th[n_th] = std::thread(copier);
  } catch (const std::system_error&) {  // (no exception leaves the C ABI; the copiers that did start go on)
  }
  if (n_th == 0) {  // no thread to be had: the plain copy
    for (int i = 0; i < kIxSlots; ++i) (void)hipEventDestroy(ev[i]);
    PPRHIP_CHECK_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, g->stream));
    PPRHIP_CHECK_HIP(hipStreamSynchronize(g->stream));
    return PPRHIP_OK;
  }
  for (size_t c = 0; c < n_chunks; ++c) {
    const int slot = (int)(c % kIxSlots);
    {
      std::unique_lock<std::mutex> lk(mu);  // the slot's previous chunk has been moved on
      cv.wait(lk, [&] { return freed[slot] >= c / kIxSlots || err; });
      if (err) break;
    }
    const size_t off = c * kIxSlotBytes, len = std::min(kIxSlotBytes, bytes - off);
    const bool ok = hipMemcpyAsync(ring + (size_t)slot * kIxSlotBytes, static_cast<const char*>(d_src) + off, len,
                                   hipMemcpyDeviceToHost, g->stream) == hipSuccess &&
                    hipEventRecord(ev[slot], g->stream) == hipSuccess;
    {
      std::lock_guard<std::mutex> lk(mu);
      if (!ok && !err) err = PPRHIP_ERR_HIP;
      if (ok) issued = c + 1;
    }
    cv.notify_all();
    if (!ok) break;
  }
  {
    std::lock_guard<std::mutex> lk(mu);
    if (issued < n_chunks && !err) err = PPRHIP_ERR_HIP;
  }
  cv.notify_all();
  for (int i = 0; i < n_th; ++i) th[i].join();
  (void)hipStreamSynchronize(g->stream);
  for (int i = 0; i < kIxSlots; ++i) (void)hipEventDestroy(ev[i]);
  if (err) set_error("index: download of the sorted entries failed");
  return err;
}

// the entries in a device record store -> the index (rows of sources in [v_lo, v_hi)): row order and the k rule on the
// device (kernels_sort.hip: finalize_rows_device), then the three index arrays cross PCIe as they are - through the
// ring of pinned slots into the index's own (pageable, huge-page) arrays.  The host does no per-entry and no per-row
// work: round 3's k rule on the host's threads was 36 ms of R-MAT 22's 53 ms and 160 of R-MAT 24's 240, and its passes
// over all n rows cost a rank of a sharded job the same whatever its share of the entries.
int index_from_device(pprhip_graph* g, const TripleRec* rec, unsigned long long count, int k, uint32_t v_lo, uint32_t v_hi,
                      pprhip_index_t** out) {
  if (v_lo > v_hi || v_hi > g->n) {
    set_error("index: source range [%u, %u) outside [0, %u)", v_lo, v_hi, g->n);
    return PPRHIP_ERR_INVALID;
  }
  const bool dbg = hook_env("PPRHIP_APBS_DEBUG") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  auto ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
  std::unique_ptr<pprhip_index> ix(new (std::nothrow) pprhip_index());
  if (!ix) return PPRHIP_ERR_OOM;
  ix->n = g->n;
  DeviceRows R;
  PPRHIP_TRY(finalize_rows_device(g, rec, count, k, v_lo, v_hi, &R));
  if (dbg) fprintf(stderr, "[index] rows finished on the device at %.1f ms (%llu of %llu entries kept)\n", ms(), R.entries, count);
  if (!R.offsets) {  // no entries: every row is empty
    ix->offsets.assign((size_t)g->n + 1, 0);
    *out = ix.release();
    return PPRHIP_OK;
  }
  try {
    ix->offsets.resize((size_t)g->n + 1);
    ix->targets.resize(R.entries);
    ix->values.resize(R.entries);
  } catch (const std::bad_alloc&) {  // (up to 12 bytes of HBM per entry must not stay behind)
    set_error("index: no host memory for %llu entries", R.entries);
    device_rows_free(&R);
    return PPRHIP_ERR_OOM;
  }
  int rc = ring_download(g, R.offsets, ix->offsets.data(), 8 * ((size_t)g->n + 1));
  if (rc == PPRHIP_OK) rc = ring_download(g, R.values, ix->values.data(), 8 * (size_t)R.entries);
  if (rc == PPRHIP_OK) rc = ring_download(g, R.targets, ix->targets.data(), 4 * (size_t)R.entries);
  device_rows_free(&R);
  if (rc != PPRHIP_OK) return rc;
  if (dbg) fprintf(stderr, "[index] on the host at %.1f ms\n", ms());
  *out = ix.release();
  return PPRHIP_OK;
}

// index over all n sources from entries of any targets, rows outside [v_lo, v_hi) must not occur
int index_from_triples(uint32_t n, std::vector<Triple>& tr, int k, pprhip_index_t** out) {
  // entries may come from a device buffer, an exchange or a caller's arrays: a source or target outside [0, n) must
  // be an error here, not an out-of-range write in the bucketing below
  for (const Triple& x : tr)
    if (x.v < 0 || (uint32_t)x.v >= n || x.t < 0 || (uint32_t)x.t >= n) {
      set_error("index entry (source %d, target %d) outside [0, %u)", x.v, x.t, n);
      return PPRHIP_ERR_INVALID;
    }
  std::unique_ptr<pprhip_index> ix(new (std::nothrow) pprhip_index());
  if (!ix) return PPRHIP_ERR_OOM;
  finalize_rows(n, tr, k, ix.get());
  *out = ix.release();
  return PPRHIP_OK;
}

// rows of several indexes over disjoint source ranges, put together (no k rule to re-apply)
int index_concat(const std::vector<pprhip_index_t*>& parts, pprhip_index_t** out) {
  std::unique_ptr<pprhip_index> ix(new (std::nothrow) pprhip_index());
  if (!ix) return PPRHIP_ERR_OOM;
  const uint32_t n = parts[0]->n;
  ix->n = n;
  ix->offsets.assign((size_t)n + 1, 0);
  for (const pprhip_index_t* p : parts)
    for (uint32_t v = 0; v < n; ++v) ix->offsets[v + 1] += p->offsets[v + 1] - p->offsets[v];
  for (uint32_t v = 0; v < n; ++v) ix->offsets[v + 1] += ix->offsets[v];
  ix->targets.resize(ix->offsets[n]);
  ix->values.resize(ix->offsets[n]);
  std::vector<uint64_t> at(ix->offsets.begin(), ix->offsets.end() - 1);
  for (const pprhip_index_t* p : parts)
    for (uint32_t v = 0; v < n; ++v)
      for (uint64_t i = p->offsets[v]; i < p->offsets[v + 1]; ++i) {
        ix->targets[at[v]] = p->targets[i];
        ix->values[at[v]++] = p->values[i];
      }
  *out = ix.release();
  return PPRHIP_OK;
}

}  // namespace detail
}  // namespace pprhip

extern "C" {

int pprhip_all_pair_backward(pprhip_graph_t* g, double alpha, double threshold, int k, uint32_t t_begin, uint32_t t_end,
                             pprhip_index_t** index_out, pprhip_stats_t* stats) {
  PPRHIP_TRY(check_graph(g, "pprhip_all_pair_backward"));
  if (!index_out || t_begin > t_end || t_end > g->n) {
    set_error("pprhip_all_pair_backward: bad target range [%u, %u) for n=%u", t_begin, t_end, g->n);
    return PPRHIP_ERR_INVALID;
  }
  pprhip_stats_t st;
  std::memset(&st, 0, sizeof st);
  // the searches' entries stay in HBM, are put in (source, target) order there, and cross PCIe once, in row order
  DeviceTripleSink sink;
  const auto t0 = std::chrono::steady_clock::now();
  // first-use work of the finalisation, beside the searches: the pinned ring the sorted entries are downloaded through
  std::thread pin;
  if (!g->ix_stage) {
    try {
      pin = std::thread([g] {
        if (hipSetDevice(g->device) == hipSuccess) (void)ensure_ring(g);
      });
    } catch (const std::system_error&) {  // (no helper thread: the download pins its ring when it gets there)
    }
  }
  // room for the entries a range of this size usually yields (a dozen per target at the thresholds the thesis uses):
  // the store then does not grow - allocate, copy, free - while the searches run
  if (t_end - t_begin >= (1u << 18)) (void)sink.reserve(g, 12ull * (unsigned long long)(t_end - t_begin));
  const int crc = all_pair_collect(g, alpha, threshold, t_begin, t_end, sink, st);
  if (pin.joinable()) pin.join();
  PPRHIP_TRY(crc);
  const auto t1 = std::chrono::steady_clock::now();
  try {  // (the index arrays are host allocations of hundreds of megabytes: no exception leaves the C ABI)
    PPRHIP_TRY(index_from_device(g, sink.rec, sink.count, k, 0u, g->n, index_out));
  } catch (const std::exception& e) {
    set_error("pprhip_all_pair_backward: index finalisation: %s", e.what());
    return PPRHIP_ERR_OOM;
  }
  if (hook_env("PPRHIP_APBS_DEBUG"))
    fprintf(stderr, "[apbs host] searches + hand-over %.1f ms, index finalisation %.1f ms\n",
            std::chrono::duration<double, std::milli>(t1 - t0).count(),
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
  if (stats) *stats = st;
  return PPRHIP_OK;
}

int pprhip_index_merge(const pprhip_index_t* const* shards, int n_shards, int k, pprhip_index_t** merged_out) {
  if (!shards || n_shards < 1 || !merged_out) {
    set_error("pprhip_index_merge: bad arguments");
    return PPRHIP_ERR_INVALID;
  }
  if (!shards[0]) {
    set_error("pprhip_index_merge: shard 0 is null");
    return PPRHIP_ERR_INVALID;
  }
  const uint32_t n = shards[0]->n;
  std::vector<Triple> tr;
  for (int s = 0; s < n_shards; ++s) {
    if (!shards[s] || shards[s]->n != n) {
      set_error("pprhip_index_merge: shard %d does not match", s);
      return PPRHIP_ERR_INVALID;
    }
    for (uint32_t v = 0; v < n; ++v)
      for (uint64_t i = shards[s]->offsets[v]; i < shards[s]->offsets[v + 1]; ++i)
        tr.push_back({(int32_t)v, shards[s]->targets[i], shards[s]->values[i]});
  }
  try {
    return index_from_triples(n, tr, k, merged_out);
  } catch (const std::exception& e) {
    set_error("pprhip_index_merge: %s", e.what());
    return PPRHIP_ERR_OOM;
  }
}

int pprhip_index_from_arrays(uint32_t n, const uint64_t* offsets, const int32_t* targets, const double* values,
                             pprhip_index_t** index_out) {
  if (!offsets || !index_out || offsets[0] != 0 || (offsets[n] && (!targets || !values))) {
    set_error("pprhip_index_from_arrays: bad arguments");
    return PPRHIP_ERR_INVALID;
  }
  for (uint32_t v = 0; v < n; ++v)
    if (offsets[v + 1] < offsets[v]) {
      set_error("pprhip_index_from_arrays: offsets must be non-decreasing");
      return PPRHIP_ERR_INVALID;
    }
  for (uint64_t i = 0; i < offsets[n]; ++i)
    if (targets[i] < 0 || (uint32_t)targets[i] >= n) {
      set_error("pprhip_index_from_arrays: target %d at position %llu outside [0, %u)", targets[i],
                (unsigned long long)i, n);
      return PPRHIP_ERR_INVALID;
    }
  std::unique_ptr<pprhip_index> ix(new (std::nothrow) pprhip_index());
  if (!ix) return PPRHIP_ERR_OOM;
  ix->n = n;
  ix->offsets.assign(offsets, offsets + n + 1);
  ix->targets.assign(targets, targets + offsets[n]);
  ix->values.assign(values, values + offsets[n]);
  *index_out = ix.release();
  return PPRHIP_OK;
}

int pprhip_index_from_entries(uint32_t n, const int32_t* sources, const int32_t* targets, const double* values,
                              uint64_t count, int k, pprhip_index_t** index_out) {
  if (!index_out || (count && (!sources || !targets || !values))) {
    set_error("pprhip_index_from_entries: null argument");
    return PPRHIP_ERR_INVALID;
  }
  try {
    std::vector<Triple> tr(count);
    for (uint64_t i = 0; i < count; ++i) tr[i] = Triple{sources[i], targets[i], values[i]};
    return index_from_triples(n, tr, k, index_out);  // validates the ids, buckets by source, applies the k rule
  } catch (const std::exception& e) {
    set_error("pprhip_index_from_entries: %s", e.what());
    return PPRHIP_ERR_OOM;
  }
}

int pprhip_index_info(const pprhip_index_t* ix, uint32_t* n, uint64_t* entries) {
  if (!ix) {
    set_error("pprhip_index_info: null index");
    return PPRHIP_ERR_INVALID;
  }
  if (n) *n = ix->n;
  if (entries) *entries = ix->targets.size();
  return PPRHIP_OK;
}

int pprhip_index_arrays(const pprhip_index_t* ix, const uint64_t** offsets, const int32_t** targets,
                        const double** values) {
  if (!ix || !offsets || !targets || !values) {
    set_error("pprhip_index_arrays: null argument");
    return PPRHIP_ERR_INVALID;
  }
  *offsets = ix->offsets.data();
  *targets = ix->targets.data();
  *values = ix->values.data();
  return PPRHIP_OK;
}

void pprhip_index_destroy(pprhip_index_t* ix) { delete ix; }

}  // extern "C"
