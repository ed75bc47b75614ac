// engine.cpp — host-side drivers of the pprhip engine: graph lift, level loop, FORA, top-k,
// backward search.  Everything numerical runs in the HIP kernels; the host only sequences
// launches on the handle's stream and reads back 8-byte counters between levels.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <numeric>
#include <thread>

#include "engine.hpp"

using namespace pprhip;

namespace pprhip {

struct ForaRun;

// Rendezvous of the batch workers (one host thread and one stream per slot) with the sweeper
// thread.  Workers run their queries' sparse levels, walks and selections concurrently; a worker
// whose next level is dense waits here.  The sweeper runs a batched sweep whenever somebody waits
// and no slot "holds": a slot holds from the moment a sweep releases it until it has said what it
// does next (waits again, goes on with sparse levels, ends its push phase), and while it writes its
// column of the shared contribution array.  Slots busy with sparse levels or walks hold nothing,
// so their kernels overlap the sweeps of the others.
struct BatchSync {
  std::mutex mu;
  std::condition_variable cv;
  pprhip_graph* P = nullptr;
  ForaRun* runs = nullptr;
  int n_wait = 0, n_hold = 0, n_workers = 0;
  bool sweeping = false;
  bool waitflag[kBatch] = {};
  bool hold[kBatch] = {};
  int err = 0;
  std::string errmsg;
  void release(int s);  // the slot stops holding (no-op when it does not)
  void c8_enter(int s);  // before a slot's kernels touch its column of the shared arrays
  int arrive(int s);     // the slot's next level is dense and prepared; returns after the sweep
  void fail(int rc);
  void worker_done(int s);
  void sweeper();
};

}  // namespace pprhip

namespace {

struct Triple {  // one index entry of All-Pair-Backward-Search: pi(v, t) = p
  int32_t v, t;
  double p;
};

struct LevelCtx {
  int fcur = 0;   // F/eoff buffer holding the current frontier list
  int ccur = 0;   // dense contribution buffer holding the current level's contributions
  int pslot = 0;  // packed counter slot describing the current frontier
  int dslot = 0;  // dead-mass cell pending for the current level
  uint32_t nf = 0;
  uint64_t ef = 0;
  bool dense_prepared = false;
  int dense_run = 0;  // dense levels run since the current dense phase was seeded
};

// FORA rounds that are certain to be followed by another halving do not need their sparse tail: what it
// would push is picked up by the next round's lower threshold.  Such a round ends after the first sparse
// level that follows its dense levels.  fixed: the caller knows another round follows; otherwise the round
// loop's own condition (model cost so far < c_walk * rsum * omega) is evaluated at that point.  The test
// twin applies the same rule (oracle/ppr_oracle.c: round_cut).
struct RoundCut {
  bool enabled = false, fixed = false, had_dense = false, checked = false, taken = false;
  double omega = 0.0, c_walk = 0.0, alpha = 0.0;
  double rsum = 0.0;  // (1 - alpha) * residue sum measured at the check (valid when !fixed and checked)
};

// kernel-class timer of the calling thread: its own, or the slot's while it works for a batch
thread_local KernelTimer g_timer_own;
thread_local KernelTimer* g_timer_cur = &g_timer_own;
inline KernelTimer& ktimer() { return *g_timer_cur; }

int alloc_dev(void** p, size_t bytes) {
  hipError_t e = hipMalloc(p, bytes ? bytes : 8);
  if (e != hipSuccess) {
    set_error("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
    return e == hipErrorOutOfMemory ? PPRHIP_ERR_OOM : PPRHIP_ERR_HIP;
  }
  return PPRHIP_OK;
}

int read_packed(pprhip_graph* g, int slot, uint32_t* nf, uint64_t* ef) {
  PPRHIP_CHECK_HIP(hipMemcpyAsync(&g->h_ctr->packed[slot], &g->ctr->packed[slot], sizeof(unsigned long long),
                                  hipMemcpyDeviceToHost, g->stream));
  PPRHIP_CHECK_HIP(hipStreamSynchronize(g->stream));
  const unsigned long long pk = g->h_ctr->packed[slot];
  *nf = (uint32_t)(pk >> kPackShift);
  *ef = pk & kPackMask;
  return PPRHIP_OK;
}

int zero_packed(pprhip_graph* g, int slot) {
  PPRHIP_CHECK_HIP(hipMemsetAsync(&g->ctr->packed[slot], 0, sizeof(unsigned long long), g->stream));
  return PPRHIP_OK;
}

int write_packed(pprhip_graph* g, int slot, uint32_t nf, uint64_t ef) {
  g->h_ctr->packed[slot] = ((unsigned long long)nf << kPackShift) | ef;
  PPRHIP_CHECK_HIP(hipMemcpyAsync(&g->ctr->packed[slot], &g->h_ctr->packed[slot], sizeof(unsigned long long),
                                  hipMemcpyHostToDevice, g->stream));
  return PPRHIP_OK;
}

// level cost model; the test twin evaluates the same expression
double level_cost(const pprhip_graph* g, uint64_t nf, uint64_t ef, bool* dense) {
  const pprhip_tuning_t& t = g->tun;
  const bool d = (double)(ef + nf) >= t.dense_frac * (double)g->m;
  *dense = d;
  if (d) return t.c_level_ns + t.c_dense_edge_ns * (double)g->m + t.c_dense_node_ns * (double)g->n;
  return t.c_level_ns + t.c_edge_ns * (double)ef + t.c_pop_ns * (double)nf;
}

uint64_t dense_level_bytes(const pprhip_graph* g) { return 12ull * g->m + 36ull * g->n + 4ull; }

int write_hist0(pprhip_graph* g, uint32_t nf, uint64_t ef) {
  g->h_ctr->hist[0] = ((unsigned long long)nf << kPackShift) | ef;
  PPRHIP_CHECK_HIP(hipMemcpyAsync(&g->ctr->hist[0], &g->h_ctr->hist[0], sizeof(unsigned long long),
                                  hipMemcpyHostToDevice, g->stream));
  return PPRHIP_OK;
}

int ensure_bwd_layout(pprhip_graph* P);

constexpr int kYield = 1;  // run_levels: the next level is dense and the caller runs it (batched sweeps)

// bookkeeping after a dense level: the frontier it produced becomes the current one
void finish_dense(LevelCtx& L, pprhip_stats_t& st, uint64_t level_bytes, uint32_t nf_next, uint64_t ef_next) {
  L.dense_run++;
  L.ccur ^= 1;
  L.dslot ^= 1;
  L.pslot ^= 1;
  st.dense_levels++;
  st.dense_nodes += L.nf;
  st.push_bytes += level_bytes;
  L.nf = nf_next;
  L.ef = ef_next;
  st.levels++;
  st.enqueues += L.nf;
  st.push_bytes += 5ull * L.nf;
}

// Runs levels until the frontier is empty.  Dense levels cost one host round trip each; sparse
// levels are launched kMaxBatch at a time and continue on the device (kernels_push.hip).  With
// yield_dense the function prepares a dense level and returns kYield instead of running it: the
// batch driver runs one sweep for every slot waiting at that point and calls back in.
int device_sum(pprhip_graph* g, const double* x, double* out);

int run_levels(pprhip_graph* g, const PushArgs& a, LevelCtx& L, pprhip_stats_t& st, double* model_cost,
               bool yield_dense = false, RoundCut* cut = nullptr) {
  const bool bwd = a.mode == kBackward;
  const bool slot = g->parent != nullptr;
  // smallest integer x with (double)x >= dense_frac * m: the device-side form of level_cost()'s test
  const bool sparse_only = false;  // every push direction has both level shapes
  const unsigned long long dense_thresh =
      sparse_only ? ~0ull : (unsigned long long)std::ceil(g->tun.dense_frac * (double)g->m);
  while (L.nf > 0) {
    bool dense = false;
    const double c = level_cost(g, L.nf, L.ef, &dense);
    if (sparse_only) dense = false;
    if (dense) {
      if (model_cost) *model_cost += c;
      if (cut) cut->had_dense = true;
      if (bwd && !slot) PPRHIP_TRY(ensure_bwd_layout(g));  // sweep layout over the out-CSR, built on first use
      if (!L.dense_prepared) {
        if (slot) {
          if (g->sync) g->sync->c8_enter(g->slot_index);
          L.ccur = g->parent->c8cur;  // the slot's column of the shared array is all-zero here
        } else
          PPRHIP_CHECK_HIP(hipMemsetAsync(g->cdense[L.ccur], 0, sizeof(double) * g->n, g->stream));
        PPRHIP_TRY(write_hist0(g, L.nf, L.ef));
        PPRHIP_TRY(launch_sparse_prepare(g, a, L.fcur, 0, L.nf, dense_thresh, true, L.ccur, L.dslot));
        L.dense_prepared = true;
        L.dense_run = 0;
      }
      if (yield_dense) return kYield;
      // The sweep writes contributions of non-empty rows only.  Rows without in-edges can hold one
      // solely from a phase's seeding, so the other buffer is cleared when a dense phase starts
      // and the seeded buffer right after its first level has consumed it.
      if (L.dense_run == 0)
        PPRHIP_CHECK_HIP(hipMemsetAsync(g->cdense[L.ccur ^ 1], 0, sizeof(double) * g->n, g->stream));
      const int out = L.pslot ^ 1;
      ktimer().begin(PPRHIP_KERNEL_DENSE_PULL, dense_level_bytes(g));
      PPRHIP_TRY(launch_dense_level(g, a, L.ccur, out, L.dslot));
      ktimer().end();
      if (L.dense_run == 0)
        PPRHIP_CHECK_HIP(hipMemsetAsync(g->cdense[L.ccur], 0, sizeof(double) * g->n, g->stream));
      uint32_t nf_next = 0;
      uint64_t ef_next = 0;
      PPRHIP_TRY(read_packed(g, out, &nf_next, &ef_next));
      finish_dense(L, st, dense_level_bytes(g), nf_next, ef_next);
      continue;
    }
    // ---- a batch of sparse levels
    PPRHIP_CHECK_HIP(hipMemsetAsync(&g->ctr->hist[0], 0, sizeof(unsigned long long) * (kMaxBatch + 1), g->stream));
    const bool first_prepared = L.dense_prepared;
    if (first_prepared) {
      // dense-prepared state -> list form; the compaction recounts (dead-end nodes carry no edges)
      PPRHIP_TRY(launch_compact_prepared(g, L.ccur, L.fcur, &g->ctr->hist[0], bwd));
      L.dense_prepared = false;
      // the column must be read (and handed back zeroed) before another sweep may run
      if (g->sync) PPRHIP_CHECK_HIP(hipStreamSynchronize(g->stream));
    } else {
      PPRHIP_TRY(write_hist0(g, L.nf, L.ef));
    }
    if (g->sync) g->sync->release(g->slot_index);
    // the round-cut check looks at the state after exactly one sparse level
    const bool cut_check = cut && cut->enabled && cut->had_dense && !cut->checked;
    const int n_batch = cut_check ? 1 : kMaxBatch;
    ktimer().begin(PPRHIP_KERNEL_SPARSE_PUSH, 0);
    for (int i = 0; i < n_batch; ++i) {
      const int fb = L.fcur ^ (i & 1);
      if (!(i == 0 && first_prepared))
        PPRHIP_TRY(launch_sparse_prepare(g, a, fb, i, i == 0 ? L.nf : 32768, dense_thresh, false, 0, L.dslot));
      PPRHIP_TRY(launch_sparse_push(g, a, fb, i, i == 0 ? L.ef : (1u << 20), dense_thresh, L.dslot));
    }
    ktimer().end();
    PPRHIP_CHECK_HIP(hipMemcpyAsync(&g->h_ctr->hist[0], &g->ctr->hist[0], sizeof(unsigned long long) * (kMaxBatch + 1),
                                    hipMemcpyDeviceToHost, g->stream));
    PPRHIP_CHECK_HIP(hipStreamSynchronize(g->stream));
    uint64_t batch_bytes = 0;
    int ran = 0;
    for (int i = 0; i < n_batch; ++i) {
      // level i ran with the frontier the host knows (i == 0) or the one level i-1 produced
      const uint32_t nf_i = i == 0 ? L.nf : (uint32_t)(g->h_ctr->hist[i] >> kPackShift);
      const uint64_t ef_i = i == 0 ? L.ef : (g->h_ctr->hist[i] & kPackMask);
      if (i > 0) {
        bool d2 = false;
        const double ci = level_cost(g, nf_i, ef_i, &d2);
        if (nf_i == 0 || (d2 && !sparse_only)) break;  // the device stopped here too (level_runs)
        if (model_cost) *model_cost += ci;
      } else if (model_cost) {
        *model_cost += c;
      }
      const uint32_t nf_next = (uint32_t)(g->h_ctr->hist[i + 1] >> kPackShift);
      st.pops += nf_i;
      st.edge_pushes += ef_i;
      st.levels++;
      st.enqueues += nf_next;
      batch_bytes += 44ull * nf_i + 28ull * ef_i + 5ull * nf_next;
      ran++;
    }
    st.push_bytes += batch_bytes;
    if (!ktimer().recs.empty() && ktimer().recs.back().cls == PPRHIP_KERNEL_SPARSE_PUSH)
      ktimer().recs.back().bytes = batch_bytes;
    L.nf = (uint32_t)(g->h_ctr->hist[ran] >> kPackShift);
    L.ef = g->h_ctr->hist[ran] & kPackMask;
    if (ran & 1) L.fcur ^= 1;
    if (cut_check) {
      cut->checked = true;
      bool more = true;
      if (!cut->fixed) {
        double sum = 0.0;
        PPRHIP_TRY(device_sum(g, g->residue, &sum));
        cut->rsum = sum * (1 - cut->alpha);
        more = model_cost && *model_cost < cut->c_walk * cut->rsum * cut->omega;
      }
      if (more) {
        cut->taken = true;
        L.nf = 0;  // the rest of this round's frontier waits for the next threshold
        L.ef = 0;
      }
    }
  }
  return PPRHIP_OK;
}

int reset_query_state(pprhip_graph* g, bool clear_flags) {
  PPRHIP_CHECK_HIP(hipMemsetAsync(g->residue, 0, sizeof(double) * g->n, g->stream));
  PPRHIP_CHECK_HIP(hipMemsetAsync(g->reserve, 0, sizeof(double) * g->n, g->stream));
  PPRHIP_CHECK_HIP(hipMemsetAsync(g->ctr, 0, sizeof(DevCounters), g->stream));
  if (clear_flags) PPRHIP_CHECK_HIP(hipMemsetAsync(g->flags, 0, g->n, g->stream));
  g->result_in_est = false;
  return PPRHIP_OK;
}

// per-query workspace of a handle (the graph's own, or a batch slot's)
int alloc_workspace(pprhip_graph* G) {
  const uint32_t n = G->n;
  const size_t nd = sizeof(double) * (size_t)n;
  void** dbl[] = {(void**)&G->residue, (void**)&G->reserve, (void**)&G->est, (void**)&G->cF, (void**)&G->mc_inc};
  for (void** p : dbl) PPRHIP_TRY(alloc_dev(p, nd));
  if (!G->parent) {  // single-query dense levels; slots use the parent's interleaved arrays
    PPRHIP_TRY(alloc_dev((void**)&G->cdense[0], nd));
    PPRHIP_TRY(alloc_dev((void**)&G->cdense[1], nd));
    PPRHIP_TRY(alloc_dev((void**)&G->acc_nz, nd));
  }
  for (int i = 0; i < 2; ++i) {
    PPRHIP_TRY(alloc_dev((void**)&G->F[i], sizeof(int32_t) * (size_t)n));
    PPRHIP_TRY(alloc_dev((void**)&G->eoff[i], sizeof(uint32_t) * (size_t)n));
  }
  PPRHIP_TRY(alloc_dev((void**)&G->flags, n));
  PPRHIP_TRY(alloc_dev((void**)&G->mc_node, sizeof(int32_t) * (size_t)n));
  PPRHIP_TRY(alloc_dev((void**)&G->mc_woff, sizeof(unsigned long long) * (size_t)n));
  PPRHIP_TRY(alloc_dev((void**)&G->partial, sizeof(double) * 1024));
  PPRHIP_TRY(alloc_dev((void**)&G->hist, sizeof(uint32_t) * 4096));
  {
    const size_t nblk = std::max<size_t>(1024, ((size_t)n + 1 + 255) / 256);
    PPRHIP_TRY(alloc_dev((void**)&G->blk_pack, sizeof(unsigned long long) * nblk));
    PPRHIP_TRY(alloc_dev((void**)&G->blk_dead, sizeof(double) * nblk));
    PPRHIP_TRY(alloc_dev((void**)&G->blk_ndead, sizeof(uint32_t) * nblk));
  }
  G->sel_cap = 1u << 18;
  PPRHIP_TRY(alloc_dev((void**)&G->sel_ids, sizeof(int32_t) * G->sel_cap));
  PPRHIP_TRY(alloc_dev((void**)&G->sel_vals, sizeof(double) * G->sel_cap));
  PPRHIP_TRY(alloc_dev((void**)&G->ctr, sizeof(DevCounters)));
  if (hipHostMalloc((void**)&G->h_ctr, sizeof(DevCounters), hipHostMallocDefault) != hipSuccess) {
    set_error("hipHostMalloc failed");
    return PPRHIP_ERR_OOM;
  }
  std::memset(G->h_ctr, 0, sizeof(DevCounters));
  for (auto& e : G->ev)
    if (hipEventCreate(&e) != hipSuccess) {
      set_error("hipEventCreate failed");
      return PPRHIP_ERR_HIP;
    }
  if (!G->parent) {
    PPRHIP_CHECK_HIP(hipMemsetAsync(G->acc_nz, 0, nd, G->stream));
    PPRHIP_CHECK_HIP(hipMemsetAsync(G->cdense[0], 0, nd, G->stream));
    PPRHIP_CHECK_HIP(hipMemsetAsync(G->cdense[1], 0, nd, G->stream));
  }
  PPRHIP_CHECK_HIP(hipMemsetAsync(G->est, 0, nd, G->stream));
  PPRHIP_CHECK_HIP(hipMemsetAsync(G->flags, 0, n, G->stream));
  return reset_query_state(G, true);
}

void free_workspace(pprhip_graph* g) {
  void* ptrs[] = {g->acc_nz, g->residue, g->reserve, g->est, g->cdense[0], g->cdense[1], g->cF, g->F[0], g->F[1],
                  g->eoff[0], g->eoff[1], g->flags, g->mc_node, g->mc_inc, g->mc_woff, g->partial, g->hist, g->sel_ids,
                  g->sel_vals, g->ctr, g->blk_pack, g->blk_dead, g->blk_ndead};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  if (g->h_ctr) (void)hipHostFree(g->h_ctr);
  for (auto e : g->ev)
    if (e) (void)hipEventDestroy(e);
}

void free_batch(pprhip_graph* P);

// Batch slots and the interleaved dense-level arrays, created on the first batched call.
int build_batch(pprhip_graph* P) {
  const size_t n = P->n;
  for (int i = 0; i < 2; ++i) {
    PPRHIP_TRY(alloc_dev((void**)&P->c8[i], sizeof(double) * n * kBatch));
    PPRHIP_CHECK_HIP(hipMemsetAsync(P->c8[i], 0, sizeof(double) * n * kBatch, P->stream));
  }
  PPRHIP_TRY(alloc_dev((void**)&P->acc8, sizeof(double) * (n + 1) * kBatch));
  PPRHIP_CHECK_HIP(hipMemsetAsync(P->acc8, 0, sizeof(double) * (n + 1) * kBatch, P->stream));
  PPRHIP_TRY(alloc_dev((void**)&P->prep_bits, sizeof(unsigned long long) * kBatch * (n / 64 + 2)));
  PPRHIP_CHECK_HIP(hipMemsetAsync(P->prep_bits, 0, sizeof(unsigned long long) * kBatch * (n / 64 + 2), P->stream));
  PPRHIP_TRY(alloc_dev((void**)&P->d_slot_args, sizeof(SlotArgs) * kBatch));
  if (hipHostMalloc((void**)&P->h_slot_args, sizeof(SlotArgs) * kBatch, hipHostMallocDefault) != hipSuccess) {
    set_error("hipHostMalloc failed");
    return PPRHIP_ERR_OOM;
  }
  std::memset(P->h_slot_args, 0, sizeof(SlotArgs) * kBatch);
  PPRHIP_TRY(alloc_dev((void**)&P->sweep_out, sizeof(unsigned long long) * kBatch));
  if (hipHostMalloc((void**)&P->h_sweep_out, sizeof(unsigned long long) * kBatch, hipHostMallocDefault) != hipSuccess) {
    set_error("hipHostMalloc failed");
    return PPRHIP_ERR_OOM;
  }
  PPRHIP_TRY(alloc_dev((void**)&P->blk_pack8, sizeof(unsigned long long) * kBatch * kApplyBlocks8));
  PPRHIP_TRY(alloc_dev((void**)&P->blk_dead8, sizeof(double) * kBatch * kApplyBlocks8));
  PPRHIP_TRY(alloc_dev((void**)&P->blk_ndead8, sizeof(uint32_t) * kBatch * kApplyBlocks8));
  P->c8cur = 0;
  for (int s = 0; s < kBatch; ++s) {
    pprhip_graph* S = new (std::nothrow) pprhip_graph();
    if (!S) return PPRHIP_ERR_OOM;
    P->slots.push_back(S);
    S->parent = P;
    S->slot_index = s;
    S->device = P->device;
    S->n_cus = P->n_cus;
    S->n = P->n;
    S->m = P->m;
    if (hipStreamCreateWithFlags(&S->own_stream, hipStreamNonBlocking) != hipSuccess) {
      set_error("hipStreamCreate failed");
      return PPRHIP_ERR_HIP;
    }
    S->stream = P->stream;
    S->out_rp = P->out_rp;
    S->in_rp = P->in_rp;
    S->out_ext = P->out_ext;
    S->out_ci = P->out_ci;
    S->in_ci = P->in_ci;
    S->relabeled = P->relabeled;
    S->new2old = P->new2old;
    S->old2new = P->old2new;
    S->start_flags = P->start_flags;
    S->chunk_starts = P->chunk_starts;
    S->n_chunks = P->n_chunks;
    S->nz_rows = P->nz_rows;
    S->n_nz = P->n_nz;
    S->tun = P->tun;
    PPRHIP_TRY(alloc_workspace(S));
  }
  PPRHIP_CHECK_HIP(hipStreamSynchronize(P->stream));
  return PPRHIP_OK;
}

int ensure_batch(pprhip_graph* P) {
  if (!P->slots.empty()) return PPRHIP_OK;
  const int rc = build_batch(P);
  if (rc != PPRHIP_OK) free_batch(P);  // e.g. out of memory half-way: leave no partial batch state behind
  return rc;
}

void free_batch(pprhip_graph* P) {
  for (pprhip_graph* S : P->slots) {
    free_workspace(S);
    S->ktimer.destroy();
    if (S->own_stream) (void)hipStreamDestroy(S->own_stream);
    delete S;
  }
  P->ktimer.destroy();
  P->slots.clear();
  void* ptrs[] = {P->c8[0], P->c8[1], P->acc8, P->prep_bits, P->d_slot_args, P->sweep_out, P->blk_pack8, P->blk_dead8,
                  P->blk_ndead8};
  if (P->h_sweep_out) (void)hipHostFree(P->h_sweep_out);
  P->sweep_out = P->h_sweep_out = nullptr;
  P->prep_bits = nullptr;
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  if (P->h_slot_args) (void)hipHostFree(P->h_slot_args);
  P->c8[0] = P->c8[1] = P->acc8 = nullptr;
  P->acc8_dir = 0;
  P->d_slot_args = P->h_slot_args = nullptr;
  P->blk_pack8 = nullptr;
  P->blk_dead8 = nullptr;
  P->blk_ndead8 = nullptr;
}

// sweep layout over the out-CSR for batched backward searches (the forward one is built at graph lift)
int ensure_bwd_layout(pprhip_graph* P) {
  if (P->start_flags_o) return PPRHIP_OK;
  const uint32_t n = P->n;
  const uint64_t m = P->m;
  const std::vector<uint32_t>& rp = P->h_out_rp;
  const size_t n_chunks = ((size_t)m + kChunkPad - 1) / kChunkPad;
  std::vector<uint8_t> flags((n_chunks + 1) * (kChunkPad / 8), 0);
  std::vector<uint32_t> chunk_starts(n_chunks + 1, 0);
  std::vector<int32_t> nz, zr;
  for (uint32_t v = 0; v < n; ++v) {
    if (rp[v + 1] == rp[v]) {
      zr.push_back((int32_t)v);
      continue;
    }
    nz.push_back((int32_t)v);
    const uint32_t e = rp[v];
    flags[e >> 3] |= (uint8_t)(1u << (e & 7));
    chunk_starts[(size_t)e / kChunkPad + 1]++;
  }
  for (size_t c = 1; c <= n_chunks; ++c) chunk_starts[c] += chunk_starts[c - 1];
  std::vector<unsigned long long> cross(((size_t)n + 63) / 64 + 1, 0ull);
  for (size_t j = 0; j < nz.size(); ++j) {
    const uint32_t v = (uint32_t)nz[j];
    const uint32_t last = rp[v + 1] - 1;
    if (rp[v] / kChunkPad != last / kChunkPad || (last + 1) % kChunkPad == 0 || (uint64_t)last + 1 == m)
      cross[j >> 6] |= 1ull << (j & 63);
  }
  auto up = [&](void** dst, const void* src, size_t bytes) -> int {
    PPRHIP_TRY(alloc_dev(dst, bytes));
    if (bytes) PPRHIP_CHECK_HIP(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
    return PPRHIP_OK;
  };
  P->n_nz_o = (uint32_t)nz.size();
  P->n_z_o = (uint32_t)zr.size();
  PPRHIP_TRY(up((void**)&P->chunk_starts_o, chunk_starts.data(), sizeof(uint32_t) * chunk_starts.size()));
  PPRHIP_TRY(up((void**)&P->nz_rows_o, nz.data(), sizeof(int32_t) * nz.size()));
  PPRHIP_TRY(up((void**)&P->z_rows_o, zr.data(), sizeof(int32_t) * zr.size()));
  PPRHIP_TRY(up((void**)&P->cross_bits_o, cross.data(), sizeof(unsigned long long) * cross.size()));
  PPRHIP_TRY(up((void**)&P->start_flags_o, flags.data(), flags.size()));
  return PPRHIP_OK;
}

int seed_single(pprhip_graph* g, LevelCtx& L, int32_t node, uint32_t degree) {
  // frontier = {node}; the first node is pushed unconditionally (Forward_Push.java:81-86)
  g->h_ctr->pad[0] = (unsigned long long)(uint32_t)node;  // staging for the 4-byte node id
  PPRHIP_CHECK_HIP(hipMemcpyAsync(g->F[L.fcur], &g->h_ctr->pad[0], sizeof(int32_t), hipMemcpyHostToDevice, g->stream));
  PPRHIP_CHECK_HIP(hipMemsetAsync(g->eoff[L.fcur], 0, sizeof(uint32_t), g->stream));
  L.nf = 1;
  L.ef = degree;
  L.dense_prepared = false;
  return PPRHIP_OK;
}

// frontier from a predicate over all nodes (round starts)
int seed_scan(pprhip_graph* g, const PushArgs& a, int kind, LevelCtx& L) {
  PPRHIP_TRY(launch_count_active(g, a, kind, L.pslot));
  PPRHIP_TRY(read_packed(g, L.pslot, &L.nf, &L.ef));
  L.dense_prepared = false;
  bool dense = false;
  if (L.nf) (void)level_cost(g, L.nf, L.ef, &dense);
  if (dense) {
    if (g->parent) {
      if (g->sync) g->sync->c8_enter(g->slot_index);
      L.ccur = g->parent->c8cur;
    }
    PPRHIP_TRY(launch_seed_dense(g, a, kind, L.ccur, L.pslot, L.dslot));
    L.dense_prepared = true;
    L.dense_run = 0;
  } else if (L.nf || kind == 1) {
    // kind 1 also runs for an empty start set: parked nodes below min_rmax still leave the set
    // (Forward_Push.java:241-247)
    PPRHIP_CHECK_HIP(hipMemsetAsync(&g->ctr->hist[kMaxBatch + 2], 0, sizeof(unsigned long long), g->stream));
    PPRHIP_TRY(launch_seed_list(g, a, kind, L.fcur, &g->ctr->hist[kMaxBatch + 2]));
  }
  return PPRHIP_OK;
}

int device_sum(pprhip_graph* g, const double* x, double* out) {
  PPRHIP_TRY(launch_sum(g, x, g->n));
  PPRHIP_CHECK_HIP(hipMemcpyAsync(&g->h_ctr->sum_out, &g->ctr->sum_out, sizeof(double), hipMemcpyDeviceToHost,
                                  g->stream));
  PPRHIP_CHECK_HIP(hipStreamSynchronize(g->stream));
  *out = g->h_ctr->sum_out;
  return PPRHIP_OK;
}

int read_dead_pops(pprhip_graph* g, pprhip_stats_t& st) {
  PPRHIP_CHECK_HIP(hipMemcpyAsync(&g->h_ctr->dead_pops, &g->ctr->dead_pops, sizeof(unsigned long long),
                                  hipMemcpyDeviceToHost, g->stream));
  PPRHIP_CHECK_HIP(hipStreamSynchronize(g->stream));
  st.dead_end_pops = g->h_ctr->dead_pops;
  st.push_bytes += 16ull * st.dead_end_pops;
  return PPRHIP_OK;
}

// walk phase shared by FORA whole-graph (variant 0) and top-k (variant 1)
int run_walk_phase(pprhip_graph* g, int variant, double alpha, double rsum, long long nrw, uint64_t seed, uint32_t stream,
                   double* target, pprhip_stats_t& st) {
  PPRHIP_CHECK_HIP(hipMemsetAsync(&g->ctr->mc_packed, 0, sizeof(unsigned long long), g->stream));
  PPRHIP_CHECK_HIP(hipMemsetAsync(&g->ctr->walk_steps, 0, sizeof(unsigned long long), g->stream));
  if ((double)nrw + (double)g->n >= (double)(1ull << kPackShift)) {
    set_error("walk budget %lld exceeds the engine's 2^36 walk limit", nrw);
    return PPRHIP_ERR_INVALID;
  }
  PPRHIP_TRY(launch_mc_plan(g, variant, alpha, rsum, (double)nrw, target));
  PPRHIP_CHECK_HIP(hipMemcpyAsync(&g->h_ctr->mc_packed, &g->ctr->mc_packed, sizeof(unsigned long long),
                                  hipMemcpyDeviceToHost, g->stream));
  PPRHIP_CHECK_HIP(hipStreamSynchronize(g->stream));
  const uint64_t n_src = g->h_ctr->mc_packed >> kPackShift;
  const uint64_t n_walks = g->h_ctr->mc_packed & kPackMask;
  ktimer().begin(PPRHIP_KERNEL_WALK, 0);
  PPRHIP_TRY(launch_mc_walk(g, n_src, n_walks, alpha, seed, stream, variant == 0 ? 1 : 0, target));
  ktimer().end();
  PPRHIP_CHECK_HIP(hipMemcpyAsync(&g->h_ctr->walk_steps, &g->ctr->walk_steps, sizeof(unsigned long long),
                                  hipMemcpyDeviceToHost, g->stream));
  PPRHIP_CHECK_HIP(hipStreamSynchronize(g->stream));
  st.mc_sources += n_src;
  st.walks += n_walks;
  st.walk_steps += g->h_ctr->walk_steps;
  const uint64_t bytes = 12ull * g->h_ctr->walk_steps + 16ull * n_walks + 12ull * n_src;
  st.mc_bytes += bytes;
  if (!ktimer().recs.empty() && ktimer().recs.back().cls == PPRHIP_KERNEL_WALK) ktimer().recs.back().bytes = bytes;
  return PPRHIP_OK;
}

struct CallTimer {
  pprhip_graph* g;
  explicit CallTimer(pprhip_graph* g_) : g(g_) {
    ktimer().stream = g->stream;
    ktimer().reset();
    (void)hipEventRecord(g->ev[0], g->stream);
  }
  void mark(int i) { (void)hipEventRecord(g->ev[i], g->stream); }
  static double ms(hipEvent_t a, hipEvent_t b) {
    float f = 0.f;
    if (hipEventElapsedTime(&f, a, b) != hipSuccess) return 0.0;
    return (double)f;
  }
  // resolves per-class kernel times; picks the class with the largest total as dominant
  void finish(pprhip_stats_t& st) {
    (void)hipEventRecord(g->ev[5], g->stream);
    (void)hipStreamSynchronize(g->stream);
    st.total_ms = ms(g->ev[0], g->ev[5]);
    double tot[8] = {0};
    uint64_t bytes[8] = {0};
    uint32_t cnt[8] = {0};
    ktimer().resolve(tot, bytes, cnt);
    int best = 0;
    for (int c = 1; c < 8; ++c)
      if (tot[c] > tot[best]) best = c;
    for (int c = 0; c < 8; ++c) {
      st.class_ms[c] = tot[c];
      st.class_bytes[c] = bytes[c];
      st.class_launches[c] = cnt[c];
    }
    st.dominant_kernel_id = (uint32_t)best;
    st.dominant_kernel_ms = tot[best];
    st.dominant_kernel_bytes = bytes[best];
    st.dominant_kernel_launches = cnt[best];
  }
};

int copy_out(pprhip_graph* g, const double* dev, double* host) {
  if (!host) return PPRHIP_OK;
  const double* srcp = dev;
  if (g->relabeled) {  // back to the caller's ids: out[old] = x[old2new[old]]
    PPRHIP_TRY(launch_permute_out(g, dev, g->cF));
    srcp = g->cF;
  }
  PPRHIP_CHECK_HIP(hipMemcpyAsync(host, srcp, sizeof(double) * g->n, hipMemcpyDeviceToHost, g->stream));
  PPRHIP_CHECK_HIP(hipStreamSynchronize(g->stream));
  return PPRHIP_OK;
}

int check_graph(const pprhip_graph* g, const char* fn) {
  if (!g) {
    set_error("%s: null graph handle", fn);
    return PPRHIP_ERR_INVALID;
  }
  hipError_t e = hipSetDevice(g->device);
  if (e != hipSuccess) {
    set_error("%s: hipSetDevice(%d) failed: %s", fn, g->device, hipGetErrorString(e));
    return PPRHIP_ERR_NO_DEVICE;
  }
  return PPRHIP_OK;
}

int check_node(const pprhip_graph* g, int32_t v, const char* fn) {
  if (v < 0 || (uint32_t)v >= g->n) {
    set_error("%s: node id %d outside [0, %u)", fn, v, g->n);
    return PPRHIP_ERR_INVALID;
  }
  return PPRHIP_OK;
}

// host-side CSR facts live on the graph handle; batch slots borrow them
const pprhip_graph* host_of(const pprhip_graph* g) { return g->parent ? g->parent : g; }
uint32_t hdeg_out(const pprhip_graph* g, int32_t v) { return host_of(g)->h_out_rp[v + 1] - host_of(g)->h_out_rp[v]; }
uint32_t hdeg_in(const pprhip_graph* g, int32_t v) { return host_of(g)->h_in_rp[v + 1] - host_of(g)->h_in_rp[v]; }

// ------------------------------------------------------------------ top-k selection driver
struct IdVal {
  int32_t id;
  double val;
};

int select_topk(pprhip_graph* g, const double* x, int k, int32_t* ids_out, double* vals_out, int cap, int* n_out,
                double* kth_out, bool* have_kth, pprhip_stats_t& st) {
  std::vector<uint32_t> hist(4096);
  unsigned long long prefix = 0;
  int pbits = 0;
  uint64_t k_rem = (uint64_t)k;
  uint64_t above = 0;  // entries in bins above the chosen prefix
  uint64_t total = 0;
  bool have = true;
  unsigned long long lower_bits = 1ull;  // smallest positive pattern: "everything"
  for (int pass = 0; pbits < 64; ++pass) {
    const int dbits = std::min(12, 64 - pbits);
    PPRHIP_TRY(launch_select_hist(g, x, g->n, prefix, pbits, dbits));
    PPRHIP_CHECK_HIP(hipMemcpyAsync(hist.data(), g->hist, sizeof(uint32_t) * (1u << dbits), hipMemcpyDeviceToHost,
                                    g->stream));
    PPRHIP_CHECK_HIP(hipStreamSynchronize(g->stream));
    st.select_passes++;
    st.select_bytes += 8ull * g->n;
    if (pass == 0) {
      for (uint32_t b = 0; b < (1u << dbits); ++b) total += hist[b];
      if (total == 0) {
        *n_out = 0;
        *have_kth = false;
        if (kth_out) *kth_out = 0.0;
        return PPRHIP_OK;
      }
      if ((uint64_t)k > total) {  // kth_ppr returns null: everything is kept (Fora_Topk.java:187-191)
        have = false;
        break;
      }
    }
    uint64_t cum = 0;
    int chosen = -1;
    for (int b = (1 << dbits) - 1; b >= 0; --b) {
      if (cum + hist[b] >= k_rem) {
        chosen = b;
        break;
      }
      cum += hist[b];
    }
    if (chosen < 0) {
      set_error("select_topk: histogram inconsistent (k_rem=%llu)", (unsigned long long)k_rem);
      return PPRHIP_ERR_STATE;
    }
    above += cum;
    k_rem -= cum;
    prefix = (prefix << dbits) | (unsigned long long)chosen;
    pbits += dbits;
    lower_bits = pbits < 64 ? (prefix << (64 - pbits)) : prefix;
    if (above + hist[chosen] <= g->sel_cap) break;  // few enough candidates: finish on the host
  }
  PPRHIP_TRY(launch_select_gather(g, x, g->n, have ? lower_bits : 1ull));
  PPRHIP_CHECK_HIP(hipMemcpyAsync(&g->h_ctr->sel_count, &g->ctr->sel_count, sizeof(unsigned long long),
                                  hipMemcpyDeviceToHost, g->stream));
  PPRHIP_CHECK_HIP(hipStreamSynchronize(g->stream));
  st.select_bytes += 8ull * g->n;
  uint64_t cnt = g->h_ctr->sel_count;
  std::vector<IdVal> cand;
  if (cnt <= g->sel_cap) {
    std::vector<int32_t> ids(cnt);
    std::vector<double> vals(cnt);
    if (cnt) {
      PPRHIP_CHECK_HIP(hipMemcpyAsync(ids.data(), g->sel_ids, sizeof(int32_t) * cnt, hipMemcpyDeviceToHost, g->stream));
      PPRHIP_CHECK_HIP(hipMemcpyAsync(vals.data(), g->sel_vals, sizeof(double) * cnt, hipMemcpyDeviceToHost, g->stream));
      PPRHIP_CHECK_HIP(hipStreamSynchronize(g->stream));
    }
    cand.resize(cnt);
    for (uint64_t i = 0; i < cnt; ++i) cand[i] = {host_of(g)->h_new2old[ids[i]], vals[i]};
  } else {
    // more ties at the k-th value than the candidate buffer holds: finish on the whole vector
    std::vector<double> all(g->n);
    PPRHIP_CHECK_HIP(hipMemcpyAsync(all.data(), x, sizeof(double) * g->n, hipMemcpyDeviceToHost, g->stream));
    PPRHIP_CHECK_HIP(hipStreamSynchronize(g->stream));
    const double lb = [&] { double d; std::memcpy(&d, &lower_bits, 8); return d; }();
    for (uint32_t i = 0; i < g->n; ++i)
      if (all[i] > 0.0 && (!have || all[i] >= lb)) cand.push_back({host_of(g)->h_new2old[i], all[i]});
  }
  std::sort(cand.begin(), cand.end(), [](const IdVal& a, const IdVal& b) {
    if (a.val != b.val) return a.val > b.val;
    return a.id < b.id;
  });
  size_t n_sel = cand.size();
  double kth = 0.0;
  if (have) {
    kth = cand[(size_t)k - 1].val;
    n_sel = 0;
    while (n_sel < cand.size() && cand[n_sel].val >= kth) ++n_sel;
  }
  for (size_t i = 0; i < n_sel && (int)i < cap; ++i) {
    if (ids_out) ids_out[i] = cand[i].id;
    if (vals_out) vals_out[i] = cand[i].val;
  }
  *n_out = (int)n_sel;
  *have_kth = have;
  if (kth_out) *kth_out = kth;
  st.kth_value = kth;
  return PPRHIP_OK;
}

}  // namespace

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

int pprhip_device_count(int* count_out) {
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) c = 0;
  if (count_out) *count_out = c;
  return PPRHIP_OK;
}

void pprhip_tuning_default(pprhip_tuning_t* t) {
  if (!t) return;
  // calibrated on MI355X; tests/test_host.py checks the test twin uses the same numbers
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
  t->reserved = 0;
}

void pprhip_tuning_batch(pprhip_tuning_t* t) {
  pprhip_tuning_default(t);
  if (!t) return;
  // fitted on R-MAT 22 with all slots busy: a sweep of 1.6 ms serves ~14.5 queries; a level that touches
  // fewer than 2 % of the edges is cheaper as a sparse level (one memory-side atomic per edge)
  t->c_dense_edge_ns = 0.002;
  t->c_dense_node_ns = 0.003;
  t->dense_frac = 0.02;
}

int pprhip_conf_fora_whole_graph(uint32_t n, uint64_t m, double alpha, pprhip_fora_conf_t* c) {
  if (!c || n == 0) {
    set_error("pprhip_conf_fora_whole_graph: bad arguments");
    return PPRHIP_ERR_INVALID;
  }
  std::memset(c, 0, sizeof *c);
  c->alpha = alpha;
  c->delta = 1.0 / (double)n;  // Algo_Conf.java:47
  c->pfail = 1.0 / (double)n;  // :48
  c->rsum = 1.0;               // :49
  c->n = n;
  c->m = m;
  return PPRHIP_OK;
}

int pprhip_conf_fora_topk(uint32_t n, uint64_t m, int k, double alpha, pprhip_fora_conf_t* c) {
  if (!c || n == 0 || k < 1) {
    set_error("pprhip_conf_fora_topk: bad arguments (n=%u k=%d)", n, k);
    return PPRHIP_ERR_INVALID;
  }
  std::memset(c, 0, sizeof *c);
  c->alpha = alpha;
  c->min_delta = 1.0 / (double)n;  // Algo_Conf.java:73
  c->k = k;
  c->delta = 1.0 / (double)k;  // :75
  c->pfail = 1.0 / (double)n / (double)n / std::log((double)((int32_t)n / k));  // :76 (int division)
  c->rsum = 1.0;
  c->n = n;
  c->m = m;
  return PPRHIP_OK;
}

int pprhip_fora_whole_params(const pprhip_fora_conf_t* c, double eps, double* rmax0, double* omega) {
  if (!c || !rmax0 || !omega) {
    set_error("pprhip_fora_whole_params: null argument");
    return PPRHIP_ERR_INVALID;
  }
  *rmax0 = eps * std::sqrt(c->delta / 3.0 / (double)c->m / std::log(2.0 / c->pfail)) / (1.0 - c->alpha);
  *omega = (eps + 2.0) * std::log(2.0 / c->pfail) / eps / eps / c->delta;
  return PPRHIP_OK;
}

int pprhip_fora_topk_params(const pprhip_fora_conf_t* c, double eps, double delta, double* min_rmax,
                            double* rmax_scaled, double* omega) {
  if (!c || !min_rmax || !rmax_scaled || !omega) {
    set_error("pprhip_fora_topk_params: null argument");
    return PPRHIP_ERR_INVALID;
  }
  const double e = eps * 0.5;  // Fora_Topk.java:109-110
  *min_rmax = e * std::sqrt(c->min_delta / 3 / (double)c->m / std::log(2 / c->pfail));  // :113
  double rmax = e * std::sqrt(delta / 3.0 / (double)c->m / std::log(2.0 / c->pfail));     // :124
  *omega = (e + 2.0) * std::log(2.0 / c->pfail) / e / e / delta;                           // :125
  rmax *= std::sqrt((double)c->m * rmax) * 3.0;                                            // :133
  *rmax_scaled = rmax;
  return PPRHIP_OK;
}

// ------------------------------------------------------------------ graph lift
int pprhip_graph_create(uint32_t n, uint64_t m, const uint32_t* out_rp, const int32_t* out_ci, const uint32_t* in_rp,
                        const int32_t* in_ci, int device, pprhip_graph_t** graph_out) {
  if (!graph_out || !out_rp || (!out_ci && m) || n == 0 || n >= (1u << 28) || m >= (1ull << 32) - 1024) {
    set_error("pprhip_graph_create: bad arguments (n=%u m=%llu; limits n < 2^28, m < 2^32 - 1024)", n,
              (unsigned long long)m);
    return PPRHIP_ERR_INVALID;
  }
  if (out_rp[0] != 0 || out_rp[n] != m) {
    set_error("pprhip_graph_create: out_row_ptr[0] must be 0 and out_row_ptr[n] must equal m");
    return PPRHIP_ERR_INVALID;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    set_error("pprhip_graph_create: no HIP device available (the engine has no CPU fallback)");
    return PPRHIP_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= ndev) {
    set_error("pprhip_graph_create: device %d outside [0, %d)", device, ndev);
    return PPRHIP_ERR_NO_DEVICE;
  }
  PPRHIP_CHECK_HIP(hipSetDevice(device));
  for (uint64_t e = 0; e < m; ++e)
    if (out_ci[e] < 0 || (uint32_t)out_ci[e] >= n) {
      set_error("pprhip_graph_create: out_col_idx[%llu] = %d outside [0, %u)", (unsigned long long)e, out_ci[e], n);
      return PPRHIP_ERR_INVALID;
    }
  const bool have_in = in_rp && (in_ci || m == 0);
  if (have_in && (in_rp[0] != 0 || in_rp[n] != m)) {
    set_error("pprhip_graph_create: in_row_ptr[0] must be 0 and in_row_ptr[n] must equal m");
    return PPRHIP_ERR_INVALID;
  }
  std::unique_ptr<pprhip_graph> g(new (std::nothrow) pprhip_graph());
  if (!g) return PPRHIP_ERR_OOM;
  g->device = device;
  g->n = n;
  g->m = m;
  pprhip_tuning_default(&g->tun);
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
      g->n_cus = prop.multiProcessorCount;
  }

  // ---- internal vertex order: out-degree descending, ties by original id (PPRHIP_RELABEL=0 keeps ids)
  const char* env = getenv("PPRHIP_RELABEL");
  g->relabeled = !(env && env[0] == '0');
  g->h_new2old.resize(n);
  g->h_old2new.resize(n);
  std::iota(g->h_new2old.begin(), g->h_new2old.end(), 0);
  if (g->relabeled)
    std::stable_sort(g->h_new2old.begin(), g->h_new2old.end(), [&](int32_t x, int32_t y) {
      return out_rp[x + 1] - out_rp[x] > out_rp[y + 1] - out_rp[y];
    });
  for (uint32_t v = 0; v < n; ++v) g->h_old2new[g->h_new2old[v]] = (int32_t)v;
  const std::vector<int32_t>& o2n = g->h_old2new;
  // rows move, entries are renamed, the order inside a row is kept (walks index rows by position)
  auto relabel_csr = [&](const uint32_t* rp, const int32_t* ci, std::vector<uint32_t>& nrp, std::vector<int32_t>& nci) {
    nrp.assign((size_t)n + 1, 0);
    for (uint32_t v = 0; v < n; ++v) {
      const int32_t o = g->h_new2old[v];
      nrp[v + 1] = nrp[v] + (rp[o + 1] - rp[o]);
    }
    nci.resize(((size_t)m + kChunkPad - 1) / kChunkPad * kChunkPad + kChunkPad, 0);
    for (uint32_t v = 0; v < n; ++v) {
      const int32_t o = g->h_new2old[v];
      uint32_t w = nrp[v];
      for (uint32_t e = rp[o]; e < rp[o + 1]; ++e) nci[w++] = o2n[ci[e]];
    }
  };
  std::vector<int32_t> n_out_ci, n_in_ci;
  relabel_csr(out_rp, out_ci, g->h_out_rp, n_out_ci);
  if (have_in) {
    relabel_csr(in_rp, in_ci, g->h_in_rp, n_in_ci);
  } else {
    // derive the in-adjacency: edges in (new) out-CSR order, grouped by destination
    std::vector<int32_t> src(m);
    for (uint32_t v = 0; v < n; ++v)
      for (uint32_t e = g->h_out_rp[v]; e < g->h_out_rp[v + 1]; ++e) src[e] = (int32_t)v;
    g->h_in_rp.resize((size_t)n + 1);
    n_in_ci.assign(((size_t)m + kChunkPad - 1) / kChunkPad * kChunkPad + kChunkPad, 0);
    int rc = pprhip_csr_build(n, m, n_out_ci.data(), src.data(), 0, g->h_in_rp.data(), n_in_ci.data());
    if (rc != PPRHIP_OK) return rc;
  }

  // ---- dense pull-sweep layout: non-empty rows, row-start flags per in-edge, starts before each chunk
  const std::vector<uint32_t>& irp = g->h_in_rp;
  std::vector<int32_t> nz_rows;
  const size_t n_chunks = ((size_t)m + kChunkPad - 1) / kChunkPad;
  std::vector<uint8_t> flags((n_chunks + 1) * (kChunkPad / 8), 0);
  std::vector<uint32_t> chunk_starts(n_chunks + 1, 0);
  for (uint32_t v = 0; v < n; ++v) {
    if (irp[v + 1] == irp[v]) continue;
    nz_rows.push_back((int32_t)v);
    const uint32_t e = irp[v];
    flags[e >> 3] |= (uint8_t)(1u << (e & 7));
    chunk_starts[(size_t)e / kChunkPad + 1]++;  // counted into every later chunk by the prefix sum below
  }
  for (size_t c = 1; c <= n_chunks; ++c) chunk_starts[c] += chunk_starts[c - 1];
  g->n_chunks = (uint32_t)n_chunks;
  g->n_nz = (uint32_t)nz_rows.size();

  pprhip_graph* G = g.get();
  auto up = [&](void** dst, const void* src, size_t bytes) -> int {
    PPRHIP_TRY(alloc_dev(dst, bytes));
    if (bytes) PPRHIP_CHECK_HIP(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
    return PPRHIP_OK;
  };
  int rc = PPRHIP_OK;
  auto fail = [&](int code) {
    pprhip_graph_destroy(g.release());
    return code;
  };
  if ((rc = up((void**)&G->out_rp, G->h_out_rp.data(), sizeof(uint32_t) * ((size_t)n + 1)))) return fail(rc);
  if ((rc = up((void**)&G->out_ci, n_out_ci.data(), sizeof(int32_t) * n_out_ci.size()))) return fail(rc);
  {
    std::vector<unsigned long long> ext(n);
    for (uint32_t v = 0; v < n; ++v)
      ext[v] = (unsigned long long)G->h_out_rp[v] | ((unsigned long long)(G->h_out_rp[v + 1] - G->h_out_rp[v]) << 32);
    if ((rc = up((void**)&G->out_ext, ext.data(), sizeof(unsigned long long) * (size_t)n))) return fail(rc);
  }
  if ((rc = up((void**)&G->in_rp, G->h_in_rp.data(), sizeof(uint32_t) * ((size_t)n + 1)))) return fail(rc);
  if ((rc = up((void**)&G->in_ci, n_in_ci.data(), sizeof(int32_t) * n_in_ci.size()))) return fail(rc);
  if ((rc = up((void**)&G->new2old, G->h_new2old.data(), sizeof(int32_t) * (size_t)n))) return fail(rc);
  if ((rc = up((void**)&G->old2new, G->h_old2new.data(), sizeof(int32_t) * (size_t)n))) return fail(rc);
  if ((rc = up((void**)&G->start_flags, flags.data(), flags.size()))) return fail(rc);
  if ((rc = up((void**)&G->chunk_starts, chunk_starts.data(), sizeof(uint32_t) * chunk_starts.size()))) return fail(rc);
  if ((rc = up((void**)&G->nz_rows, nz_rows.data(), sizeof(int32_t) * nz_rows.size()))) return fail(rc);
  if (hipStreamCreateWithFlags(&G->stream, hipStreamNonBlocking) != hipSuccess) {
    set_error("hipStreamCreate failed");
    return fail(PPRHIP_ERR_HIP);
  }
  {
    std::vector<int32_t> zin;
    for (uint32_t v = 0; v < n; ++v)
      if (irp[v + 1] == irp[v]) zin.push_back((int32_t)v);
    G->n_zin = (uint32_t)zin.size();
    if ((rc = up((void**)&G->zin_rows, zin.data(), sizeof(int32_t) * zin.size()))) return fail(rc);
    std::vector<unsigned long long> cross(((size_t)n + 63) / 64 + 1, 0ull);
    for (size_t j = 0; j < nz_rows.size(); ++j) {
      const uint32_t v = (uint32_t)nz_rows[j];
      // summed with atomics (so cleared after every sweep): rows holding the last edge of a chunk
      const uint32_t last = irp[v + 1] - 1;
      if (irp[v] / kChunkPad != last / kChunkPad || (last + 1) % kChunkPad == 0 || (uint64_t)last + 1 == m)
        cross[j >> 6] |= 1ull << (j & 63);
    }
    if ((rc = up((void**)&G->cross_bits, cross.data(), sizeof(unsigned long long) * cross.size()))) return fail(rc);
  }
  if ((rc = alloc_workspace(G))) return fail(rc);
  if (hipStreamSynchronize(G->stream) != hipSuccess) {
    set_error("stream sync after graph upload failed");
    return fail(PPRHIP_ERR_HIP);
  }
  *graph_out = g.release();
  return PPRHIP_OK;
}

void pprhip_graph_destroy(pprhip_graph_t* g) {
  if (!g) return;
  (void)hipSetDevice(g->device);
  if (g->stream) (void)hipStreamSynchronize(g->stream);
  free_batch(g);
  void* ptrs[] = {g->out_ext, g->out_rp, g->out_ci, g->in_rp, g->in_ci, g->new2old, g->old2new, g->start_flags,
                  g->chunk_starts, g->nz_rows, g->zin_rows, g->cross_bits, g->start_flags_o, g->chunk_starts_o,
                  g->nz_rows_o, g->z_rows_o, g->cross_bits_o};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  free_workspace(g);
  if (g->stream) (void)hipStreamDestroy(g->stream);
  delete g;
}

int pprhip_graph_info(const pprhip_graph_t* g, uint32_t* n, uint64_t* m, int* device) {
  if (!g) {
    set_error("pprhip_graph_info: null graph handle");
    return PPRHIP_ERR_INVALID;
  }
  if (n) *n = g->n;
  if (m) *m = g->m;
  if (device) *device = g->device;
  return PPRHIP_OK;
}

int pprhip_graph_set_tuning(pprhip_graph_t* g, const pprhip_tuning_t* t) {
  if (!g || !t) {
    set_error("pprhip_graph_set_tuning: null argument");
    return PPRHIP_ERR_INVALID;
  }
  pprhip_tuning_t d;
  pprhip_tuning_default(&d);
  g->tun = *t;
  if (!(g->tun.c_walk_ns > 0)) g->tun.c_walk_ns = d.c_walk_ns;
  if (!(g->tun.c_edge_ns > 0)) g->tun.c_edge_ns = d.c_edge_ns;
  if (!(g->tun.c_pop_ns > 0)) g->tun.c_pop_ns = d.c_pop_ns;
  if (!(g->tun.c_level_ns > 0)) g->tun.c_level_ns = d.c_level_ns;
  if (!(g->tun.c_dense_edge_ns > 0)) g->tun.c_dense_edge_ns = d.c_dense_edge_ns;
  if (!(g->tun.c_dense_node_ns > 0)) g->tun.c_dense_node_ns = d.c_dense_node_ns;
  if (!(g->tun.dense_frac > 0)) g->tun.dense_frac = d.dense_frac;
  if (g->tun.max_rounds <= 0) g->tun.max_rounds = d.max_rounds;
  if (g->tun.max_halvings <= 0) g->tun.max_halvings = d.max_halvings;
  if (!(g->tun.halving_ratio > 0)) g->tun.halving_ratio = d.halving_ratio;  // a value <= 1 switches the rule off
  if (g->tun.prior_levels == 0) g->tun.prior_levels = d.prior_levels;        // negative: off
  return PPRHIP_OK;
}

int pprhip_graph_get_tuning(const pprhip_graph_t* g, pprhip_tuning_t* t) {
  if (!g || !t) {
    set_error("pprhip_graph_get_tuning: null argument");
    return PPRHIP_ERR_INVALID;
  }
  *t = g->tun;
  return PPRHIP_OK;
}

int pprhip_get_reserve(pprhip_graph_t* g, double* out) {
  PPRHIP_TRY(check_graph(g, "pprhip_get_reserve"));
  if (!out) {
    set_error("pprhip_get_reserve: null output");
    return PPRHIP_ERR_INVALID;
  }
  return copy_out(g, g->result_in_est ? g->est : g->reserve, out);
}

int pprhip_get_residue(pprhip_graph_t* g, double* out) {
  PPRHIP_TRY(check_graph(g, "pprhip_get_residue"));
  if (!out) {
    set_error("pprhip_get_residue: null output");
    return PPRHIP_ERR_INVALID;
  }
  return copy_out(g, g->residue, out);
}

// ------------------------------------------------------------------ forward push (a1)
int pprhip_forward_push(pprhip_graph_t* g, int32_t src, double alpha, double rmax, double* reserve_out,
                        double* residue_out, double* rsum_out, pprhip_stats_t* stats) {
  PPRHIP_TRY(check_graph(g, "pprhip_forward_push"));
  PPRHIP_TRY(check_node(g, src, "pprhip_forward_push"));
  src = g->h_old2new[src];  // internal (degree-sorted) id
  pprhip_stats_t st;
  std::memset(&st, 0, sizeof st);
  g->topk_active = false;
  PPRHIP_TRY(reset_query_state(g, false));
  CallTimer tm(g);
  double rsum = 0.0;
  if (hdeg_out(g, src) == 0) {  // Forward_Push.java:72-76
    PPRHIP_TRY(launch_set_f64(g, g->reserve, (uint32_t)src, 1.0));
  } else {
    PushArgs a{alpha, rmax, 0.0, src, kFwdWhole};
    LevelCtx L;
    PPRHIP_TRY(launch_set_f64(g, g->residue, (uint32_t)src, 1.0));
    PPRHIP_TRY(seed_single(g, L, src, hdeg_out(g, src)));
    PPRHIP_TRY(run_levels(g, a, L, st, nullptr));
    PPRHIP_TRY(device_sum(g, g->residue, &rsum));
    PPRHIP_TRY(read_dead_pops(g, st));
  }
  tm.mark(1);
  tm.finish(st);
  st.push_ms = CallTimer::ms(g->ev[0], g->ev[1]);
  st.rsum = rsum;
  st.rmax_final = rmax;
  st.rounds = 1;
  if (rsum_out) *rsum_out = rsum;
  PPRHIP_TRY(copy_out(g, g->reserve, reserve_out));
  PPRHIP_TRY(copy_out(g, g->residue, residue_out));
  if (stats) *stats = st;
  return PPRHIP_OK;
}

// ------------------------------------------------------------------ resumable top-k push (a2)
int pprhip_fwdpush_topk_reset(pprhip_graph_t* g, int32_t src, double alpha) {
  PPRHIP_TRY(check_graph(g, "pprhip_fwdpush_topk_reset"));
  PPRHIP_TRY(check_node(g, src, "pprhip_fwdpush_topk_reset"));
  src = g->h_old2new[src];  // internal (degree-sorted) id
  PPRHIP_TRY(reset_query_state(g, true));
  // Q = {s} (Fora_Topk.java:117-118): the source starts parked
  PPRHIP_CHECK_HIP(hipMemsetAsync(g->flags + src, 1, 1, g->stream));
  g->topk_active = true;
  g->topk_first = true;
  g->topk_src = src;
  g->topk_alpha = alpha;
  g->topk_rsum = 1.0;
  return PPRHIP_OK;
}

static int topk_round_impl(pprhip_graph_t* g, double min_rmax, double rmax, pprhip_stats_t& st) {
  const int32_t src = g->topk_src;
  if (hdeg_out(g, src) == 0) {  // Forward_Push.java:149-153
    PPRHIP_TRY(launch_set_f64(g, g->reserve, (uint32_t)src, 1.0));
    g->topk_rsum = 0.0;
    return PPRHIP_OK;
  }
  if (g->topk_first) PPRHIP_TRY(launch_set_f64(g, g->residue, (uint32_t)src, 1.0));  // :155-156
  PushArgs a{g->topk_alpha, rmax, min_rmax, src, kFwdTopk};
  LevelCtx L;
  PPRHIP_TRY(seed_scan(g, a, 1, L));
  PPRHIP_TRY(run_levels(g, a, L, st, nullptr));
  PPRHIP_TRY(device_sum(g, g->residue, &g->topk_rsum));
  g->topk_first = false;
  return PPRHIP_OK;
}

int pprhip_fwdpush_topk_round(pprhip_graph_t* g, double min_rmax, double rmax, double* rsum_out,
                              pprhip_stats_t* stats) {
  PPRHIP_TRY(check_graph(g, "pprhip_fwdpush_topk_round"));
  if (!g->topk_active) {
    set_error("pprhip_fwdpush_topk_round: call pprhip_fwdpush_topk_reset first");
    return PPRHIP_ERR_STATE;
  }
  pprhip_stats_t st;
  std::memset(&st, 0, sizeof st);
  CallTimer tm(g);
  PPRHIP_TRY(topk_round_impl(g, min_rmax, rmax, st));
  PPRHIP_TRY(read_dead_pops(g, st));
  tm.mark(1);
  tm.finish(st);
  st.push_ms = CallTimer::ms(g->ev[0], g->ev[1]);
  st.rsum = g->topk_rsum;
  st.rmax_final = rmax;
  st.rounds = 1;
  if (rsum_out) *rsum_out = g->topk_rsum;
  if (stats) *stats = st;
  return PPRHIP_OK;
}

// ------------------------------------------------------------------ walker exposure (a3, a4)
int pprhip_random_walk_batch(pprhip_graph_t* g, const int32_t* starts, const uint64_t* walk_idx, uint64_t count,
                             double alpha, uint64_t seed, uint32_t stream, int no_zero_hop, int32_t* terminals_out,
                             uint32_t* steps_out) {
  PPRHIP_TRY(check_graph(g, "pprhip_random_walk_batch"));
  if ((!starts || !walk_idx || !terminals_out) && count) {
    set_error("pprhip_random_walk_batch: null argument");
    return PPRHIP_ERR_INVALID;
  }
  if (stream >= 65536) {
    set_error("pprhip_random_walk_batch: stream must be < 65536");
    return PPRHIP_ERR_INVALID;
  }
  for (uint64_t i = 0; i < count; ++i) PPRHIP_TRY(check_node(g, starts[i], "pprhip_random_walk_batch"));
  if (count == 0) return PPRHIP_OK;
  int32_t *d_s = nullptr, *d_t = nullptr;
  uint64_t* d_i = nullptr;
  uint32_t* d_n = nullptr;
  int rc = PPRHIP_OK;
  if ((rc = alloc_dev((void**)&d_s, sizeof(int32_t) * count)) || (rc = alloc_dev((void**)&d_t, sizeof(int32_t) * count)) ||
      (rc = alloc_dev((void**)&d_i, sizeof(uint64_t) * count)) || (rc = alloc_dev((void**)&d_n, sizeof(uint32_t) * count))) {
    (void)hipFree(d_s); (void)hipFree(d_t); (void)hipFree(d_i); (void)hipFree(d_n);
    return rc;
  }
  auto done = [&](int code) {
    (void)hipFree(d_s); (void)hipFree(d_t); (void)hipFree(d_i); (void)hipFree(d_n);
    return code;
  };
  std::vector<int32_t> mapped(count);
  for (uint64_t i = 0; i < count; ++i) mapped[i] = g->h_old2new[starts[i]];
  if (hipMemcpy(d_s, mapped.data(), sizeof(int32_t) * count, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpyAsync(d_i, walk_idx, sizeof(uint64_t) * count, hipMemcpyHostToDevice, g->stream) != hipSuccess) {
    set_error("pprhip_random_walk_batch: upload failed");
    return done(PPRHIP_ERR_HIP);
  }
  rc = launch_walk_batch(g, d_s, d_i, count, alpha, seed, stream, no_zero_hop, d_t, d_n);
  if (rc) return done(rc);
  if (hipMemcpyAsync(terminals_out, d_t, sizeof(int32_t) * count, hipMemcpyDeviceToHost, g->stream) != hipSuccess ||
      (steps_out &&
       hipMemcpyAsync(steps_out, d_n, sizeof(uint32_t) * count, hipMemcpyDeviceToHost, g->stream) != hipSuccess) ||
      hipStreamSynchronize(g->stream) != hipSuccess) {
    set_error("pprhip_random_walk_batch: download failed: %s", hipGetErrorString(hipGetLastError()));
    return done(PPRHIP_ERR_HIP);
  }
  for (uint64_t i = 0; i < count; ++i) terminals_out[i] = g->h_new2old[terminals_out[i]];
  return done(PPRHIP_OK);
}

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
  enum Phase { kRoundStart, kLevels, kWalks, kTopkRoundStart, kTopkLevels, kTopkFinal, kBwdLevels, kBwdFinal, kDone } phase = kDone;
  int query = -1;  // batch driver: index of the query this run serves
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
  PPRHIP_TRY(reset_query_state(g, false));
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
      PPRHIP_TRY(read_dead_pops(g, r.st));
      if (r.tm) r.tm->mark(1);
      // Fora_Whole_Graph.java:112-140
      const double nrw_d = r.omega_local * r.rsum_local;
      const long long nrw = (nrw_d == nrw_d && nrw_d > 0.0) ? (long long)nrw_d : 0;
      if (!r.dead_src) PPRHIP_TRY(run_walk_phase(g, 0, r.alpha, r.rsum_local, nrw, r.seed, 0, g->reserve, r.st));
      if (r.tm) r.tm->mark(2);
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
  PPRHIP_TRY(reset_query_state(g, true));
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
  const size_t nd = sizeof(double) * (size_t)g->n;
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
      PPRHIP_TRY(device_sum(g, g->residue, &g->topk_rsum));
      g->topk_first = false;
      r.rsum_local = g->topk_rsum;  // :142
      // :143 reserve := copy of the push reserve (walk increments of earlier rounds are dropped)
      PPRHIP_CHECK_HIP(hipMemcpyAsync(g->est, g->reserve, nd, hipMemcpyDeviceToDevice, g->stream));
      const double rsum_rw = r.rsum_local * (1.0 - r.alpha);  // :148
      const double nrw_d = r.omega_local * rsum_rw;
      const long long nrw = (nrw_d == nrw_d && nrw_d > 0.0) ? (long long)nrw_d : 0;  // :151
      PPRHIP_TRY(run_walk_phase(g, 1, r.alpha, rsum_rw, nrw, r.seed, r.round, g->est, r.st));  // :155-168
      r.round++;
      double kth = 0.0;
      bool have = false;
      int nsel = 0;
      PPRHIP_TRY(select_topk(g, g->est, conf->k, nullptr, nullptr, 0, &nsel, &kth, &have, r.st));  // :173
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
  PPRHIP_TRY(reset_query_state(g, false));
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
    PPRHIP_TRY(launch_select_gather(g, g->reserve, g->n, thr_bits));  // Base_Whole_Graph.java:83 pi >= threshold
    PPRHIP_CHECK_HIP(hipMemcpyAsync(&g->h_ctr->sel_count, &g->ctr->sel_count, sizeof(unsigned long long),
                                    hipMemcpyDeviceToHost, g->stream));
    PPRHIP_CHECK_HIP(hipStreamSynchronize(g->stream));
    const uint64_t cnt = g->h_ctr->sel_count;
    const std::vector<int32_t>& n2o = host_of(g)->h_new2old;
    if (cnt > g->sel_cap) {
      std::vector<double> all(g->n);
      PPRHIP_TRY(copy_out(g, g->reserve, all.data()));
      for (uint32_t v = 0; v < g->n; ++v)  // copy_out already returned original ids
        if (all[v] > 0.0 && all[v] >= threshold) r.triples.push_back({(int32_t)v, r.target_orig, all[v]});
    } else if (cnt) {
      std::vector<int32_t> ids(cnt);
      std::vector<double> vals(cnt);
      PPRHIP_CHECK_HIP(hipMemcpyAsync(ids.data(), g->sel_ids, sizeof(int32_t) * cnt, hipMemcpyDeviceToHost, g->stream));
      PPRHIP_CHECK_HIP(hipMemcpyAsync(vals.data(), g->sel_vals, sizeof(double) * cnt, hipMemcpyDeviceToHost, g->stream));
      PPRHIP_CHECK_HIP(hipStreamSynchronize(g->stream));
      for (uint64_t i = 0; i < cnt; ++i) r.triples.push_back({n2o[ids[i]], r.target_orig, vals[i]});
    }
    r.phase = ForaRun::kDone;
  }
  return PPRHIP_OK;
}

void add_stats(pprhip_stats_t& sum, const pprhip_stats_t& st) {
  sum.pops += st.pops; sum.edge_pushes += st.edge_pushes; sum.enqueues += st.enqueues;
  sum.dead_end_pops += st.dead_end_pops; sum.dense_nodes += st.dense_nodes; sum.levels += st.levels;
  sum.dense_levels += st.dense_levels; sum.rounds += st.rounds; sum.mc_sources += st.mc_sources;
  sum.walks += st.walks; sum.walk_steps += st.walk_steps; sum.select_passes += st.select_passes;
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
int run_sweep(pprhip_graph* P, ForaRun* runs, const bool* active, int n_active) {
  for (int s = 0; s < kBatch; ++s) {
    pprhip_graph* S = P->slots[s];
    SlotArgs& sa = P->h_slot_args[s];
    sa.res = S->residue;
    sa.reserve = S->reserve;
    sa.flags = S->flags;
    sa.ctr = S->ctr;
    sa.active = active[s] ? 1 : 0;
    if (!active[s]) continue;
    const ForaRun& r = runs[s];
    sa.alpha = r.a.alpha;
    sa.rmax = r.a.rmax;
    sa.min_rmax = r.a.min_rmax;
    sa.src = r.a.src;
    sa.mode = r.a.mode;
    sa.dead_slot = r.L.dslot;
    sa.out_slot = r.L.pslot ^ 1;
    if (S->stream != P->stream) {
      PPRHIP_CHECK_HIP(hipEventRecord(S->ev[3], S->stream));
      PPRHIP_CHECK_HIP(hipStreamWaitEvent(P->stream, S->ev[3], 0));
    }
  }
  const uint64_t sweep_bytes = 4ull * P->m + (uint64_t)n_active * (8ull * P->m + 36ull * P->n + 4ull);
  bool backward = false;
  for (int s = 0; s < kBatch; ++s)
    if (active[s] && runs[s].a.mode == kBackward) backward = true;  // a job's runs all push the same way
  if ((int)backward != P->acc8_dir) {
    // rows summed with atomics are cleared by the apply kernel of their own layout only: start clean
    PPRHIP_CHECK_HIP(hipMemsetAsync(P->acc8, 0, sizeof(double) * ((size_t)P->n + 1) * kBatch, P->stream));
    P->acc8_dir = (int)backward;
  }
  P->ktimer.begin(PPRHIP_KERNEL_DENSE_PULL_BATCH, sweep_bytes);
  PPRHIP_TRY(launch_dense_level_b8(P, backward));
  P->ktimer.end();
  PPRHIP_CHECK_HIP(hipMemcpyAsync(P->h_sweep_out, P->sweep_out, sizeof(unsigned long long) * kBatch,
                                  hipMemcpyDeviceToHost, P->stream));
  PPRHIP_CHECK_HIP(hipStreamSynchronize(P->stream));
  P->c8cur ^= 1;
  for (int s = 0; s < kBatch; ++s)
    if (active[s]) {
      ForaRun& r = runs[s];
      const unsigned long long pk = P->h_sweep_out[s];
      // the sweep's index stream is shared: each query is charged its own gathers and row work
      finish_dense(r.L, r.st, 8ull * P->m + 36ull * P->n + 4ull + 4ull * P->m / (uint64_t)n_active,
                   (uint32_t)(pk >> kPackShift), pk & kPackMask);
    }
  return PPRHIP_OK;
}

struct BatchJob {
  pprhip_graph* P;
  const int32_t* srcs;
  int q;
  double eps;
  const pprhip_fora_conf_t* conf;
  uint64_t seed;
  int n_rounds;
  double* reserve_out;
  int k;
  int32_t* ids_out;
  double* vals_out;
  int* n_out;
  pprhip_stats_t* per_query;
  int kind = 0;  // 0: whole-graph FORA per query, 1: FORA top-k per query (seed + query index), 2: backward search
  double alpha = 0.0, threshold = 0.0;   // kind 2
  std::vector<Triple>* triples = nullptr;  // kind 2: every search's entries >= threshold
  pprhip_stats_t sum;
  std::mutex sum_mu;
  std::atomic<int> next_query{0};
};

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
  if (J.reserve_out) PPRHIP_TRY(copy_out(S, r.kind == 1 ? S->est : S->reserve, J.reserve_out + (size_t)i * J.P->n));
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
  return PPRHIP_OK;
}

// all slots on the calling thread and the graph's stream, one after another
int batch_sequential(BatchJob& J, ForaRun* runs) {
  pprhip_graph* P = J.P;
  int busy = 0;
  for (;;) {
    // every slot advances until it waits at a dense level; finished slots take the next query
    for (int s = 0; s < kBatch; ++s) {
      ForaRun& r = runs[s];
      for (;;) {
        if (r.query < 0) {
          const int i = J.next_query.load();
          if (i >= J.q) break;
          J.next_query.store(i + 1);
          PPRHIP_TRY(begin_query(J, r, P->slots[s], i));
          busy++;
        }
        if (r.waiting) break;
        const int rc = run_step(r, true);
        if (rc == kYield) {
          r.waiting = true;
          break;
        }
        if (rc != PPRHIP_OK) return rc;
        PPRHIP_TRY(finish_query(J, r));
        busy--;
      }
    }
    if (busy == 0) break;
    bool active[kBatch];
    int n_wait = 0;
    for (int s = 0; s < kBatch; ++s) {
      active[s] = runs[s].query >= 0 && runs[s].waiting;
      n_wait += active[s] ? 1 : 0;
    }
    PPRHIP_TRY(run_sweep(P, runs, active, n_wait));
    for (int s = 0; s < kBatch; ++s)
      if (active[s]) runs[s].waiting = false;
  }
  return PPRHIP_OK;
}

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
  g_timer_cur = &g_timer_own;
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
// runs a prepared job on the handle's slots (both batched entry points)
static int batch_run(pprhip_graph_t* g, BatchJob& J, pprhip_stats_t* stats_sum) {
  PPRHIP_TRY(ensure_batch(g));
  if (J.kind == 2) PPRHIP_TRY(ensure_bwd_layout(g));
  const int q = J.q;
  // Worker threads pay off where queries are latency-bound (top-k: short rounds of sparse levels, walks
  // and selections, 2.4x on R-MAT 22); whole-graph FORA keeps the memory system busy from one thread.
  const char* env = getenv("PPRHIP_BATCH_THREADS");
  const bool threaded = q > 1 && (env ? env[0] == '1' : J.kind != 0);
  std::memset(&J.sum, 0, sizeof J.sum);
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
      S->sync = nullptr;
    }
    KernelTimer local;  // the caller's timer may be in use (All-Pair times its own tiers)
    KernelTimer* const saved = g_timer_cur;
    g_timer_cur = &local;
    local.stream = g->stream;
    rc = batch_sequential(J, runs);
    (void)hipStreamSynchronize(g->stream);
    local.resolve(tot, bytes, cnt);
    local.destroy();
    g_timer_cur = saved;
  }
  (void)hipStreamSynchronize(g->stream);
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

int pprhip_fora_batch_single_source(pprhip_graph_t* g, const int32_t* srcs, int q, double eps,
                                    const pprhip_fora_conf_t* conf, uint64_t seed, int n_rounds,
                                    double* reserve_out, int k, int32_t* ids_out, double* vals_out, int* n_out,
                                    pprhip_stats_t* per_query, pprhip_stats_t* stats_sum) {
  PPRHIP_TRY(check_graph(g, "pprhip_fora_batch_single_source"));
  if (q < 0 || !conf || !(eps > 0.0) || n_rounds < 0 || (q > 0 && !srcs) || k < 0 ||
      (k > 0 && q > 0 && (!ids_out || !vals_out))) {
    set_error("pprhip_fora_batch_single_source: bad arguments (q=%d eps=%g n_rounds=%d k=%d)", q, eps, n_rounds, k);
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
  return batch_run(g, J, stats_sum);
}

// ------------------------------------------------------------------ top-k select (a7)
int pprhip_topk_select(pprhip_graph_t* g, int k, int32_t* ids_out, double* vals_out, int cap, int* n_out,
                       double* kth_out, pprhip_stats_t* stats) {
  PPRHIP_TRY(check_graph(g, "pprhip_topk_select"));
  if (k < 1 || cap < 0 || !n_out || (cap > 0 && (!ids_out || !vals_out))) {
    set_error("pprhip_topk_select: bad arguments (k=%d cap=%d)", k, cap);
    return PPRHIP_ERR_INVALID;
  }
  pprhip_stats_t st;
  std::memset(&st, 0, sizeof st);
  CallTimer tm(g);
  bool have = false;
  PPRHIP_TRY(select_topk(g, g->result_in_est ? g->est : g->reserve, k, ids_out, vals_out, cap, n_out, kth_out, &have,
                         st));
  tm.mark(1);
  tm.finish(st);
  st.select_ms = CallTimer::ms(g->ev[0], g->ev[1]);
  if (stats) *stats = st;
  return PPRHIP_OK;
}

// ------------------------------------------------------------------ FORA top-k (a6)
int pprhip_fora_topk(pprhip_graph_t* g, int32_t src, double eps, const pprhip_fora_conf_t* conf, uint64_t seed,
                     int32_t* ids_out, double* vals_out, int cap, int* n_out, double* reserve_out,
                     pprhip_stats_t* stats) {
  PPRHIP_TRY(check_graph(g, "pprhip_fora_topk"));
  PPRHIP_TRY(check_node(g, src, "pprhip_fora_topk"));
  if (!conf || conf->k < 1 || !(eps > 0.0) || cap < 0 || (cap > 0 && (!ids_out || !vals_out))) {
    set_error("pprhip_fora_topk: bad arguments");
    return PPRHIP_ERR_INVALID;
  }
  pprhip_stats_t st;
  std::memset(&st, 0, sizeof st);
  PPRHIP_TRY(pprhip_fwdpush_topk_reset(g, src, conf->alpha));
  g->topk_rsum = conf->rsum;
  src = g->h_old2new[src];  // internal (degree-sorted) id
  CallTimer tm(g);
  const double alpha = conf->alpha;
  const double epsilon = eps * 0.5;  // Fora_Topk.java:109-110
  double delta_local = conf->delta;
  const double min_delta = conf->min_delta;
  const double min_rmax = epsilon * std::sqrt(min_delta / 3 / (double)conf->m / std::log(2 / conf->pfail));  // :113
  double rsum_local = conf->rsum, omega_local = 0.0, rmax_local = 0.0;
  double push_ms = 0.0, mc_ms = 0.0, sel_ms = 0.0;
  uint32_t round = 0;
  const size_t nd = sizeof(double) * (size_t)g->n;
  bool dead_src = false;
  while (delta_local >= min_delta) {  // :123
    rmax_local = epsilon * std::sqrt(delta_local / 3.0 / (double)conf->m / std::log(2.0 / conf->pfail));  // :124
    omega_local = (epsilon + 2.0) * std::log(2.0 / conf->pfail) / epsilon / epsilon / delta_local;          // :125
    if (hdeg_out(g, src) == 0) {  // :126-132
      PPRHIP_CHECK_HIP(hipMemsetAsync(g->est, 0, nd, g->stream));
      PPRHIP_TRY(launch_set_f64(g, g->est, (uint32_t)src, 1.0));
      rsum_local = 0.0;
      dead_src = true;
      break;
    }
    rmax_local *= std::sqrt((double)conf->m * rmax_local) * 3.0;  // :133
    (void)hipEventRecord(g->ev[1], g->stream);
    PPRHIP_TRY(topk_round_impl(g, min_rmax, rmax_local, st));  // :137
    rsum_local = g->topk_rsum;                                 // :142
    (void)hipEventRecord(g->ev[2], g->stream);
    // :143 reserve := copy of the push reserve (walk increments of earlier rounds are dropped)
    PPRHIP_CHECK_HIP(hipMemcpyAsync(g->est, g->reserve, nd, hipMemcpyDeviceToDevice, g->stream));
    const double rsum_rw = rsum_local * (1.0 - alpha);  // :148
    const double nrw_d = omega_local * rsum_rw;
    const long long nrw = (nrw_d == nrw_d && nrw_d > 0.0) ? (long long)nrw_d : 0;  // :151
    PPRHIP_TRY(run_walk_phase(g, 1, alpha, rsum_rw, nrw, seed, round, g->est, st));  // :155-168
    (void)hipEventRecord(g->ev[3], g->stream);
    round++;
    double kth = 0.0;
    bool have = false;
    int nsel = 0;
    PPRHIP_TRY(select_topk(g, g->est, conf->k, nullptr, nullptr, 0, &nsel, &kth, &have, st));  // :173
    if (!have) kth = 0.0;                                                                        // :174
    (void)hipEventRecord(g->ev[4], g->stream);
    PPRHIP_CHECK_HIP(hipStreamSynchronize(g->stream));
    push_ms += CallTimer::ms(g->ev[1], g->ev[2]);
    mc_ms += CallTimer::ms(g->ev[2], g->ev[3]);
    sel_ms += CallTimer::ms(g->ev[3], g->ev[4]);
    st.kth_value = kth;
    if (kth >= (1 + epsilon) * delta_local || delta_local <= min_delta) break;  // :175-176
    delta_local = std::max(min_delta, delta_local / 4.0);                       // :178
  }
  if (round == 0 && !dead_src) {  // delta below min_delta from the start: nothing ran
    PPRHIP_CHECK_HIP(hipMemsetAsync(g->est, 0, nd, g->stream));
  }
  g->result_in_est = true;
  PPRHIP_TRY(read_dead_pops(g, st));
  int nsel = 0;
  bool have = false;
  double kth = 0.0;
  (void)hipEventRecord(g->ev[3], g->stream);
  PPRHIP_TRY(select_topk(g, g->est, conf->k, ids_out, vals_out, cap, &nsel, &kth, &have, st));
  (void)hipEventRecord(g->ev[4], g->stream);
  tm.finish(st);
  sel_ms += CallTimer::ms(g->ev[3], g->ev[4]);
  st.push_ms = push_ms;
  st.mc_ms = mc_ms;
  st.select_ms = sel_ms;
  st.rounds = round;
  st.rsum = rsum_local;
  st.rmax_final = rmax_local;
  st.omega = omega_local;
  if (n_out) *n_out = nsel;
  PPRHIP_TRY(copy_out(g, g->est, reserve_out));
  if (stats) *stats = st;
  return PPRHIP_OK;
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

// ------------------------------------------------------------------ pure Monte-Carlo
int pprhip_monte_carlo(pprhip_graph_t* g, int32_t src, double eps, const pprhip_fora_conf_t* conf, uint64_t seed,
                       double* ppr_out, pprhip_stats_t* stats) {
  PPRHIP_TRY(check_graph(g, "pprhip_monte_carlo"));
  PPRHIP_TRY(check_node(g, src, "pprhip_monte_carlo"));
  src = g->h_old2new[src];  // internal (degree-sorted) id
  if (!conf || !(eps > 0.0)) {
    set_error("pprhip_monte_carlo: bad arguments");
    return PPRHIP_ERR_INVALID;
  }
  pprhip_stats_t st;
  std::memset(&st, 0, sizeof st);
  g->topk_active = false;
  PPRHIP_TRY(reset_query_state(g, false));
  CallTimer tm(g);
  const double omega = 3 * std::log(2 / conf->pfail) / eps / eps / conf->delta;  // Monte_Carlo.java:145
  const uint64_t nw = (uint64_t)std::floor(omega);                                // :149 (i <= omega)
  if (nw >= (1ull << kPackShift)) {
    set_error("pprhip_monte_carlo: %llu walks exceed the engine's 2^36 limit", (unsigned long long)nw);
    return PPRHIP_ERR_INVALID;
  }
  if (hdeg_out(g, src) == 0) {
    PPRHIP_TRY(launch_set_f64(g, g->reserve, (uint32_t)src, (double)nw / omega));  // every walk returns src (:70-72)
    st.walks = nw;
  } else {
    PPRHIP_CHECK_HIP(hipMemsetAsync(&g->ctr->walk_steps, 0, sizeof(unsigned long long), g->stream));
    ktimer().begin(PPRHIP_KERNEL_WALK, 0);
    PPRHIP_TRY(launch_mc_pure(g, src, nw, conf->alpha, seed, 1.0 / omega, g->reserve));
    ktimer().end();
    PPRHIP_CHECK_HIP(hipMemcpyAsync(&g->h_ctr->walk_steps, &g->ctr->walk_steps, sizeof(unsigned long long),
                                    hipMemcpyDeviceToHost, g->stream));
    PPRHIP_CHECK_HIP(hipStreamSynchronize(g->stream));
    st.walks = nw;
    st.walk_steps = g->h_ctr->walk_steps;
    st.mc_bytes = 12ull * st.walk_steps + 16ull * nw + 12ull;
    if (!ktimer().recs.empty()) ktimer().recs.back().bytes = st.mc_bytes;
  }
  st.mc_sources = 1;
  st.omega = omega;
  tm.mark(1);
  tm.finish(st);
  st.mc_ms = CallTimer::ms(g->ev[0], g->ev[1]);
  PPRHIP_TRY(copy_out(g, g->reserve, ppr_out));
  if (stats) *stats = st;
  return PPRHIP_OK;
}

// ------------------------------------------------------------------ backward search (a8)
static int backward_push_impl(pprhip_graph_t* g, int32_t target, double alpha, double rmax, pprhip_stats_t& st) {
  PPRHIP_TRY(reset_query_state(g, false));
  if (hdeg_in(g, target) == 0) {  // Backward_Search.java:46-49
    PPRHIP_TRY(launch_set_f64(g, g->reserve, (uint32_t)target, 1.0));
    return PPRHIP_OK;
  }
  PushArgs a{alpha, rmax, 0.0, target, kBackward};
  LevelCtx L;
  PPRHIP_TRY(launch_set_f64(g, g->residue, (uint32_t)target, 1.0));
  PPRHIP_TRY(seed_single(g, L, target, hdeg_in(g, target)));
  PPRHIP_TRY(run_levels(g, a, L, st, nullptr));
  return PPRHIP_OK;
}

int pprhip_backward_push(pprhip_graph_t* g, int32_t target, double alpha, double rmax, double* reserve_out,
                         double* residue_out, pprhip_stats_t* stats) {
  PPRHIP_TRY(check_graph(g, "pprhip_backward_push"));
  PPRHIP_TRY(check_node(g, target, "pprhip_backward_push"));
  target = g->h_old2new[target];  // internal (degree-sorted) id
  pprhip_stats_t st;
  std::memset(&st, 0, sizeof st);
  g->topk_active = false;
  CallTimer tm(g);
  PPRHIP_TRY(backward_push_impl(g, target, alpha, rmax, st));
  tm.mark(1);
  tm.finish(st);
  st.push_ms = CallTimer::ms(g->ev[0], g->ev[1]);
  st.rmax_final = rmax;
  st.rounds = 1;
  PPRHIP_TRY(copy_out(g, g->reserve, reserve_out));
  PPRHIP_TRY(copy_out(g, g->residue, residue_out));
  if (stats) *stats = st;
  return PPRHIP_OK;
}

// ------------------------------------------------------------------ ground truth (a12)
int pprhip_power_method(pprhip_graph_t* g, int32_t src, double alpha, int iters, double* reserve_out,
                        pprhip_stats_t* stats) {
  PPRHIP_TRY(check_graph(g, "pprhip_power_method"));
  PPRHIP_TRY(check_node(g, src, "pprhip_power_method"));
  src = g->h_old2new[src];  // internal (degree-sorted) id
  if (iters < 0) {
    set_error("pprhip_power_method: iters must be >= 0");
    return PPRHIP_ERR_INVALID;
  }
  pprhip_stats_t st;
  std::memset(&st, 0, sizeof st);
  g->topk_active = false;
  PPRHIP_TRY(reset_query_state(g, false));
  CallTimer tm(g);
  if (iters > 0) {
    // iteration 1 (Power_Method.java:59-96 with residue = {s: 1})
    LevelCtx L;
    PushArgs a{alpha, 0.0, 0.0, src, kPower};
    PPRHIP_CHECK_HIP(hipMemsetAsync(g->cdense[L.ccur], 0, sizeof(double) * g->n, g->stream));
    PPRHIP_TRY(launch_set_f64(g, g->reserve, (uint32_t)src, 1.0 * alpha));
    const uint32_t d = hdeg_out(g, src);
    const double remain = 1.0 * (1 - alpha);
    if (d == 0)
      PPRHIP_TRY(launch_set_f64(g, &g->ctr->dead[L.dslot], 0, remain));
    else
      PPRHIP_TRY(launch_set_f64(g, g->cdense[L.ccur], (uint32_t)src, remain / (double)d));
    for (int it = 1; it < iters; ++it) {
      const int out = L.pslot ^ 1;
      if (it == 1) PPRHIP_CHECK_HIP(hipMemsetAsync(g->cdense[L.ccur ^ 1], 0, sizeof(double) * g->n, g->stream));
      ktimer().begin(PPRHIP_KERNEL_DENSE_PULL, dense_level_bytes(g));
      PPRHIP_TRY(launch_dense_level(g, a, L.ccur, out, L.dslot));
      ktimer().end();
      if (it == 1) PPRHIP_CHECK_HIP(hipMemsetAsync(g->cdense[L.ccur], 0, sizeof(double) * g->n, g->stream));
      L.ccur ^= 1;
      L.dslot ^= 1;
      L.pslot = out;
      st.dense_levels++;
      st.levels++;
      st.push_bytes += dense_level_bytes(g);
    }
  }
  tm.mark(1);
  tm.finish(st);
  st.push_ms = CallTimer::ms(g->ev[0], g->ev[1]);
  st.rounds = (uint32_t)iters;
  PPRHIP_TRY(copy_out(g, g->reserve, reserve_out));
  if (stats) *stats = st;
  return PPRHIP_OK;
}

}  // extern "C"

// =================================================================================================
// All-Pair-Backward-Search (a9) — first correct path: one backward search per target on the
// global arrays, entries >= threshold compacted on the device, inverted index built on the host.
// =================================================================================================
struct pprhip_index {
  uint32_t n = 0;
  std::vector<uint64_t> offsets;
  std::vector<int32_t> targets;
  std::vector<double> values;
};

namespace {

// Base_Whole_Graph.java:112-163: per source, k < 0 keeps insertion (target) order; k >= 0 keeps
// entries >= the k-th largest (all when fewer than k) sorted descending (stable: ties stay in
// target order).
void finalize_rows(uint32_t n, std::vector<Triple>& tr, int k, pprhip_index* ix) {
  ix->n = n;
  ix->offsets.assign((size_t)n + 1, 0);
  // bucket by source (counting sort), then every bucket on its own: order by target, apply the k rule
  const size_t N = tr.size();
  std::vector<uint64_t> start((size_t)n + 1, 0);
  for (const Triple& e : tr) start[(size_t)e.v + 1]++;
  for (uint32_t v = 0; v < n; ++v) start[v + 1] += start[v];
  std::vector<Triple> by_v(N);
  {
    std::vector<uint64_t> at(start.begin(), start.end() - 1);
    for (const Triple& e : tr) by_v[at[e.v]++] = e;
  }
  std::vector<Triple>().swap(tr);
  std::vector<uint64_t> kept((size_t)n + 1, 0);
  const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  const unsigned T = N < (1u << 16) ? 1u : hw;
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
  for (uint32_t v = 0; v <= n; ++v) ix->offsets[v] = kept[v];
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
}

}  // namespace

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
  g->topk_active = false;
  CallTimer tm(g);
  std::vector<Triple> tr;
  const uint32_t n_targets = t_end - t_begin;
  // PPRHIP_APBS_TIER = 2 / 3 starts at a later tier (tests exercise every tier that way)
  const int first_tier = getenv("PPRHIP_APBS_TIER") ? atoi(getenv("PPRHIP_APBS_TIER")) : 1;

  // ---- device buffers of this call
  ApbsBuffers B;
  unsigned long long* cells = nullptr;  // next_target, out_count, out_valid, overflow_count, pops, edges
  int rc = PPRHIP_OK;
  auto release = [&]() {
    void* p[] = {cells, B.out_v, B.out_t, B.out_p, B.overflow, B.g_tables};
    for (void* q : p)
      if (q) (void)hipFree(q);
  };
  B.out_cap = std::min<unsigned long long>(1ull << 24, std::max<unsigned long long>(1ull << 16, 64ull * g->n));
  if ((rc = alloc_dev((void**)&cells, sizeof(unsigned long long) * 8)) ||
      (rc = alloc_dev((void**)&B.out_v, sizeof(int32_t) * B.out_cap)) ||
      (rc = alloc_dev((void**)&B.out_t, sizeof(int32_t) * B.out_cap)) ||
      (rc = alloc_dev((void**)&B.out_p, sizeof(double) * B.out_cap)) ||
      (rc = alloc_dev((void**)&B.overflow, sizeof(int32_t) * std::max<uint32_t>(1, n_targets)))) {
    release();
    return rc;
  }
  B.next_target = cells;
  B.out_count = cells + 1;
  B.out_valid = cells + 2;
  B.overflow_count = cells + 3;
  B.stat_pops = cells + 4;
  B.stat_edges = cells + 5;
  std::vector<int32_t> h_v, h_t, h_ovf;
  std::vector<double> h_p;
  unsigned long long h_cells[8];

  // runs one tier over `list` (or the range when list is empty and use_range) until every target
  // has either produced its triples or landed in `give_up`
  auto run_tier = [&](bool global_tier, std::vector<int32_t> list, bool use_range, std::vector<int32_t>& give_up) -> int {
    int32_t* d_list = nullptr;
    for (int pass = 0; pass < 1000; ++pass) {
      const uint32_t cnt = use_range ? n_targets : (uint32_t)list.size();
      if (cnt == 0) break;
      if (!use_range) {
        if (!d_list) PPRHIP_TRY(alloc_dev((void**)&d_list, sizeof(int32_t) * list.size()));
        PPRHIP_CHECK_HIP(hipMemcpyAsync(d_list, list.data(), sizeof(int32_t) * cnt, hipMemcpyHostToDevice, g->stream));
      }
      const unsigned long long init[8] = {0, 0, ~0ull, 0, 0, 0, 0, 0};
      PPRHIP_CHECK_HIP(hipMemcpyAsync(cells, init, sizeof init, hipMemcpyHostToDevice, g->stream));
      ktimer().begin(PPRHIP_KERNEL_BACKWARD_BATCH, 0);
      PPRHIP_TRY(launch_apbs(g, global_tier, use_range ? nullptr : d_list, t_begin, cnt, alpha, threshold, B));
      ktimer().end();
      PPRHIP_CHECK_HIP(hipMemcpyAsync(h_cells, cells, sizeof h_cells, hipMemcpyDeviceToHost, g->stream));
      PPRHIP_CHECK_HIP(hipStreamSynchronize(g->stream));
      const unsigned long long valid = std::min(std::min(h_cells[1], h_cells[2]), B.out_cap);
      st.pops += h_cells[4];
      st.edge_pushes += h_cells[5];
      const uint64_t bytes = 44ull * h_cells[4] + 28ull * h_cells[5] + 16ull * valid;
      st.push_bytes += bytes;
      if (!ktimer().recs.empty()) ktimer().recs.back().bytes = bytes;
      if (valid) {
        h_v.resize(valid); h_t.resize(valid); h_p.resize(valid);
        PPRHIP_CHECK_HIP(hipMemcpyAsync(h_v.data(), B.out_v, sizeof(int32_t) * valid, hipMemcpyDeviceToHost, g->stream));
        PPRHIP_CHECK_HIP(hipMemcpyAsync(h_t.data(), B.out_t, sizeof(int32_t) * valid, hipMemcpyDeviceToHost, g->stream));
        PPRHIP_CHECK_HIP(hipMemcpyAsync(h_p.data(), B.out_p, sizeof(double) * valid, hipMemcpyDeviceToHost, g->stream));
        PPRHIP_CHECK_HIP(hipStreamSynchronize(g->stream));
        for (unsigned long long i = 0; i < valid; ++i) tr.push_back({h_v[i], h_t[i], h_p[i]});
      }
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
      list.swap(again);
      use_range = false;
      if (d_list && list.size()) {
        (void)hipFree(d_list);
        d_list = nullptr;
      }
    }
    if (d_list) (void)hipFree(d_list);
    return PPRHIP_OK;
  };

  std::vector<int32_t> to_tier2, to_tier3;
  if (first_tier <= 1) {
    rc = run_tier(false, {}, true, to_tier2);
  } else {
    for (uint32_t t = t_begin; t < t_end; ++t) (first_tier == 2 ? to_tier2 : to_tier3).push_back((int32_t)t);
  }
  if (rc == PPRHIP_OK && !to_tier2.empty()) {
    B.g_cap = 65536;
    // 8 workgroups per CU: the HBM tier is a chain of L2 round trips per edge, hidden only by occupancy
    B.g_blocks = (uint32_t)std::min<size_t>((size_t)g->n_cus * 8, to_tier2.size());
    rc = alloc_dev((void**)&B.g_tables, (size_t)B.g_blocks * B.g_cap * 40);
    if (rc == PPRHIP_OK) rc = run_tier(true, to_tier2, false, to_tier3);
  }
  release();
  if (rc != PPRHIP_OK) return rc;

  // ---- tier 3: the targets whose search outgrows a 48K-node table run on whole vectors, 16 of them in flight
  // on the batch slots; levels that touch a large part of the graph run as batched sweeps over the out-CSR
  pprhip_stats_t st3;
  std::memset(&st3, 0, sizeof st3);
  if (!to_tier3.empty()) {  // Base_Whole_Graph.java:76-92
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
    J.triples = &tr;
    PPRHIP_TRY(batch_run(g, J, &st3));
    st.pops += st3.pops;
    st.edge_pushes += st3.edge_pushes;
    st.enqueues += st3.enqueues;
    st.levels += st3.levels;
    st.dense_levels += st3.dense_levels;
    st.push_bytes += st3.push_bytes;
  }
  tm.mark(1);
  tm.finish(st);
  for (int c = 0; c < 8; ++c) {
    st.class_ms[c] += st3.class_ms[c];
    st.class_bytes[c] += st3.class_bytes[c];
    st.class_launches[c] += st3.class_launches[c];
  }
  st.push_ms = CallTimer::ms(g->ev[0], g->ev[1]);
  st.rmax_final = threshold;
  st.rounds = (uint32_t)(to_tier2.size());      // targets that needed the HBM tier
  st.dense_nodes = (uint64_t)to_tier3.size();   // targets that needed the whole-vector path
  std::unique_ptr<pprhip_index> ix(new (std::nothrow) pprhip_index());
  if (!ix) return PPRHIP_ERR_OOM;
  finalize_rows(g->n, tr, k, ix.get());
  *index_out = ix.release();
  if (stats) *stats = st;
  return PPRHIP_OK;
}

int pprhip_index_merge(const pprhip_index_t* const* shards, int n_shards, int k, pprhip_index_t** merged_out) {
  if (!shards || n_shards < 1 || !merged_out) {
    set_error("pprhip_index_merge: bad arguments");
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
  std::unique_ptr<pprhip_index> ix(new (std::nothrow) pprhip_index());
  if (!ix) return PPRHIP_ERR_OOM;
  finalize_rows(n, tr, k, ix.get());
  *merged_out = ix.release();
  return PPRHIP_OK;
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
  std::unique_ptr<pprhip_index> ix(new (std::nothrow) pprhip_index());
  if (!ix) return PPRHIP_ERR_OOM;
  ix->n = n;
  ix->offsets.assign(offsets, offsets + n + 1);
  ix->targets.assign(targets, targets + offsets[n]);
  ix->values.assign(values, values + offsets[n]);
  *index_out = ix.release();
  return PPRHIP_OK;
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
