// engine.cpp — host-side core of the pprhip engine: graph lift and workspaces, the level loop, and the
// single-query entry points (forward push, top-k push rounds, FORA top-k, Monte-Carlo, backward push,
// power method).  Everything numerical runs in the HIP kernels; the host only sequences launches on
// the handle's stream and reads back 8-byte counters between levels.  FORA runs and the batched entry
// points live in fora.cpp, All-Pair and the index in allpair.cpp (shared declarations:
// engine_internal.hpp).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <numeric>

#include "engine_internal.hpp"

using namespace pprhip;
using namespace pprhip::detail;

namespace pprhip {

std::atomic<int> g_kernel_timing{-1};
bool kernel_timing_on() {
  int v = g_kernel_timing.load(std::memory_order_relaxed);
  if (v < 0) {
    const char* e = hook_env("PPRHIP_KERNEL_TIMER");
    v = (e && e[0] == '1') ? 1 : 0;
    g_kernel_timing.store(v, std::memory_order_relaxed);
  }
  return v != 0;
}
int kernel_timing_level() {
  (void)kernel_timing_on();  // (decides on first use)
  return g_kernel_timing.load(std::memory_order_relaxed);
}

namespace detail {

thread_local KernelTimer g_timer_own;
thread_local KernelTimer* g_timer_cur = &g_timer_own;

// Kernel arguments in device memory (the HIP runtime's HIP_FORCE_DEV_KERNARG switch, read when the runtime
// initialises): a launch then costs the command processor a read of HBM instead of a read of host memory over PCIe.
// The paths that are chains of short kernels gain 4-10 % (R-MAT 22: top-k one at a time 804 -> 880 queries/s, 16 in
// flight 1 602 -> 1 692, one whole-graph query at a time 94.8 -> 98.5, headline 320 -> 323).  The library does NOT set
// it (rounds 4's load-time setenv is gone: setenv inside a JVM that already runs threads races with their getenv, and
// a library should not change the runtime for the process's other HIP users); the launchers do, before any thread or
// HIP call exists: host/ppr_main.cpp, bench.py, tests/conftest.py, and the java launcher line of INTEGRATION.md.

// One-time work per device: code objects loaded and kernel attributes set by the thread that lifts the
// first graph onto the device, under a lock, so that the launch paths (which worker threads run
// concurrently) never touch function attributes or trigger a first-use module load.
static std::mutex g_dev_init_mu;
static std::vector<char> g_dev_inited;
int init_device_once(int device) {
  std::lock_guard<std::mutex> lk(g_dev_init_mu);
  if ((size_t)device < g_dev_inited.size() && g_dev_inited[device]) return PPRHIP_OK;
  PPRHIP_TRY(init_kernels_push());
  PPRHIP_TRY(init_kernels_walk());
  PPRHIP_TRY(init_kernels_select());
  PPRHIP_TRY(init_kernels_apbs());
  PPRHIP_TRY(init_kernels_sort());
  PPRHIP_TRY(init_kernels_host());
  if ((size_t)device >= g_dev_inited.size()) g_dev_inited.resize((size_t)device + 1, 0);
  g_dev_inited[device] = 1;
  return PPRHIP_OK;
}

SetupScope::SetupScope(pprhip_graph* g) : t(g_timer_cur->stream == g->stream ? g_timer_cur : nullptr) {
  if (t) t->begin(PPRHIP_KERNEL_QUERY_SETUP, 0);
}

C8Scope::C8Scope(pprhip_graph* g_, bool back_) : g(g_), back(back_) {
  if (!g->parent || !g->c8_via_parent || g->stream == g->parent->stream) return;
  for (auto& e : g->c8_ev)
    if (!e && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
      e = nullptr;
      set_error("hipEventCreate failed (slot %d)", g->slot_index);
      rc = PPRHIP_ERR_HIP;
      return;
    }
  // (the slots share one stream: a wait for it is a wait for whatever another slot has just queued there, so a slot
  // that is known to have nothing pending - it comes out of a sweep, or stood waiting for its column - does not ask)
  if (!g->c8_settled && (hipEventRecord(g->c8_ev[0], g->stream) != hipSuccess ||
                         hipStreamWaitEvent(g->parent->stream, g->c8_ev[0], 0) != hipSuccess)) {
    set_error("slot %d: its stream could not be joined to the sweeps' stream", g->slot_index);
    rc = PPRHIP_ERR_HIP;
    return;
  }
  own = g->stream;
  g->stream = g->parent->stream;
  g->parent->in_c8++;
  on = true;
}

int C8Scope::leave() {
  if (!on) return rc;
  on = false;
  g->stream = own;
  g->parent->in_c8--;
  if (back && (hipEventRecord(g->c8_ev[1], g->parent->stream) != hipSuccess ||
               hipStreamWaitEvent(own, g->c8_ev[1], 0) != hipSuccess)) {
    set_error("slot %d: the sweeps' stream could not be joined to its stream", g->slot_index);
    rc = PPRHIP_ERR_HIP;
  }
  return rc;
}

int alloc_dev(void** p, size_t bytes) {
  // test switch: PPRHIP_FAIL_ALLOC_AFTER=<n> makes the n-th device allocation made while it is set fail as the device
  // running out of memory would (the count starts over whenever the variable is not there)
  static std::atomic<long> armed_count{0};
  if (const char* fe = hook_env("PPRHIP_FAIL_ALLOC_AFTER")) {
    if (armed_count.fetch_add(1) + 1 == atol(fe)) {
      *p = nullptr;
      set_error("hipMalloc(%zu bytes) failed: injected (PPRHIP_FAIL_ALLOC_AFTER)", bytes);
      return PPRHIP_ERR_OOM;
    }
  } else {
    armed_count.store(0);
  }
  hipError_t e = hipMalloc(p, bytes ? bytes : 8);
  if (e != hipSuccess) {
    set_error("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
    return e == hipErrorOutOfMemory ? PPRHIP_ERR_OOM : PPRHIP_ERR_HIP;
  }
  return PPRHIP_OK;
}

// A few words the host needs before it can queue the next kernel: published by a kernel into mapped pinned memory and
// awaited by spinning on the sequence word (kernels_host.hip); after kSpinUs the thread stops spinning and blocks in
// hipStreamSynchronize, which is also where a faulted kernel is reported.  `bytes`: a multiple of 8.
int fetch_begin(pprhip_graph* g, const void* dev, size_t bytes, unsigned long long* seq_out) {
  if (!g->mail || bytes > sizeof(unsigned long long) * kMailWords || (bytes & 7)) {
    *seq_out = 0;  // fetch_end copies and synchronises
    return PPRHIP_OK;
  }
  *seq_out = ++g->mail_seq;
  return launch_publish(g, dev, (uint32_t)(bytes / 8), *seq_out);
}

int fetch_end(pprhip_graph* g, unsigned long long seq, const void* dev, void* host, size_t bytes) {
  if (seq == 0) {
    PPRHIP_CHECK_HIP(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, g->stream));
    PPRHIP_CHECK_HIP(hipStreamSynchronize(g->stream));
    return PPRHIP_OK;
  }
  // A slot of the sequential batch driver waits here while a sweep runs on the compute stream: the driver's hook is
  // called between looks at the mailbox, so that the sweep's end is noticed - and the next sweep launched - at once
  // instead of after this slot's step (kernel trace: the compute stream waited 97 us per sweep for the host).
  pprhip_graph* const H = g->parent;
  const bool hooked = H && H->idle_hook;
  const double kSpinUs = hooked ? 2e6 : 60.0;
  const auto t0 = std::chrono::steady_clock::now();
  bool arrived = false;
  for (uint32_t spins = 0;; ++spins) {
    if (__atomic_load_n(&g->mail->seq, __ATOMIC_ACQUIRE) == seq) {
      arrived = true;
      break;
    }
    __builtin_ia32_pause();
    if (hooked && (spins & 7u) == 7u) H->idle_hook(H->idle_arg);
    if ((spins & 63u) == 63u &&
        std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() > kSpinUs)
      break;
  }
  if (!arrived) {
    PPRHIP_CHECK_HIP(hipStreamSynchronize(g->stream));
    if (H && H->stream != g->stream) PPRHIP_CHECK_HIP(hipStreamSynchronize(H->stream));  // (a C8Scope publication)
    if (__atomic_load_n(&g->mail->seq, __ATOMIC_ACQUIRE) != seq) {
      set_error("fetch_small: the stream drained without the published words (sequence %llu, expected %llu)",
                (unsigned long long)g->mail->seq, seq);
      return PPRHIP_ERR_STATE;
    }
  }
  std::memcpy(host, g->mail->words, bytes);
  return PPRHIP_OK;
}

int fetch_small(pprhip_graph* g, const void* dev, void* host, size_t bytes) {
  unsigned long long seq = 0;
  PPRHIP_TRY(fetch_begin(g, dev, bytes, &seq));
  return fetch_end(g, seq, dev, host, bytes);
}

int read_packed(pprhip_graph* g, int slot, uint32_t* nf, uint64_t* ef) {
  PPRHIP_TRY(fetch_small(g, &g->ctr->packed[slot], &g->h_ctr->packed[slot], sizeof(unsigned long long)));
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

// SURVEY 8(d) sweep model, 12 m + 36 n + 4, with n = the rows the sweep carries: rows without in-edges receive
// nothing and are not touched by the single-query sweep (the power method counts the same rows)
uint64_t dense_level_bytes(const pprhip_graph* g) { return 12ull * g->m + 36ull * host_of(g)->n_nz + 4ull; }

// Compulsory bytes of one sweep: what it has to move when every byte is counted once (pprhip_stats_t.sweep_min_bytes).
// Single query: column indices + row-start bits, every gatherable contribution once (8 B per node with out-edges),
// per row with in-edges the row sum out and in (16 B), the next contribution (8 B) and the residue read and written
// (16 B).  The reserve is touched by crossing rows only and is left out: a lower bound.
uint64_t dense_level_min_bytes(const pprhip_graph* g) {
  const pprhip_graph* H = host_of(g);
  return 4ull * H->m + H->m / 8 + 8ull * H->n_src_live + 40ull * H->n_nz;
}
// Batched: the index stream once, every gatherable line c8[v][0..15] once (128 B), per carried row the 128-byte row-sum
// line out and in and the next-contribution line out, and per busy query the residue of every row with in-edges.
uint64_t batch_sweep_min_bytes(const pprhip_graph* P, bool backward, int n_active) {
  const uint64_t rows_nz = backward ? P->n_nz_o : P->n_nz, rows_all = rows_nz + (backward ? P->n_z_o : P->n_zin);
  const uint64_t gather = backward ? (uint64_t)P->n_nz : (uint64_t)P->n_src_live;
  return 4ull * P->m + P->m / 8 + 128ull * gather + 256ull * rows_nz + 128ull * rows_all +
         16ull * rows_nz * (uint64_t)n_active;
}

// modelled cost of a dense sweep (level_cost's dense branch): also what a sweep costs that only runs because the
// contribution array has to be flushed
double dense_sweep_cost(const pprhip_graph* g) {
  const pprhip_tuning_t& t = g->tun;
  return t.c_level_ns + t.c_dense_edge_ns * (double)g->m + t.c_dense_node_ns * (double)g->n;
}

// smallest frontier (nodes + edges) that runs as Gauss-Seidel sweeps; ~0 when they are switched off
unsigned long long gs_thresh_of(const pprhip_graph* g) {
  const pprhip_graph* H = host_of(g);
  if (g->tun.gs_blocks <= 1 || !H->relabeled) return ~0ull;
  return (unsigned long long)std::ceil(g->tun.gs_frac * (double)g->m);
}

// Blocks of the forward sweep (rows = nodes with in-edges in internal order): block b holds the row ordinals
// [jb[b], jb[b + 1]), jb[b] = first ordinal whose in-edge prefix reaches b * m / B, rounded down to a multiple of
// 256 (whole apply tiles), and the in-edges of those rows.  The test twin builds the same blocks
// (oracle/ppr_oracle.c: build_blocks).
const GsBlock* gs_blocks_of(pprhip_graph* g, int* n_blocks) {
  pprhip_graph* H = g->parent ? g->parent : g;
  const int B = g->tun.gs_blocks;
  *n_blocks = 1;
  if (B <= 1 || !H->relabeled || H->n_nz == 0) return nullptr;
  if (H->gs_plan_B != B) {
    const std::vector<uint32_t>& irp = H->h_in_rp;
    const std::vector<int32_t>& rows = H->h_nz_rows;
    const uint32_t n_nz = H->n_nz;
    std::vector<uint32_t> jb((size_t)B + 1, 0);
    for (int b = 1; b < B; ++b) {
      const uint64_t target = (uint64_t)b * H->m / (uint64_t)B;
      uint32_t lo = 0, hi = n_nz;  // first ordinal whose in-edge prefix (= row start) reaches the target
      while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if ((uint64_t)irp[rows[mid]] >= target) hi = mid; else lo = mid + 1;
      }
      const uint32_t j = lo & ~255u;
      jb[b] = std::max(jb[b - 1], j);
    }
    jb[B] = n_nz;
    H->gs_plan.assign((size_t)B, GsBlock{0, 0, 0, 0});
    for (int b = 0; b < B; ++b) {
      GsBlock& K = H->gs_plan[b];
      K.j_lo = jb[b];
      K.j_hi = jb[b + 1];
      K.e_lo = K.j_lo < n_nz ? irp[rows[K.j_lo]] : H->m;
      K.e_hi = K.j_hi < n_nz ? irp[rows[K.j_hi]] : H->m;
    }
    H->gs_plan_B = B;
  }
  *n_blocks = B;
  return H->gs_plan.data();
}

// Windows of the sliced layout (engine.hpp: SlicedLayout) for a sweep cut into the row blocks `blocks` (nullptr: one
// block, every row): per block, for every slice, the edges of the block's rows in that slice; neighbouring ranges are
// joined.  Cached per block count (the blocks of a count are always the same, gs_blocks_of).
static std::mutex g_sl_plan_mu;
const EdgeWindows* sliced_windows_of(pprhip_graph* g, const GsBlock* blocks, int nb) {
  pprhip_graph* H = g->parent ? g->parent : g;
  SlicedLayout* L = H->sl;
  std::lock_guard<std::mutex> lock(g_sl_plan_mu);
  if (L->plan_B == nb) return L->plan.data();
  const GsBlock whole{0u, H->n_nz, 0ull, (unsigned long long)H->m};
  if (!blocks || nb <= 1) {
    blocks = &whole;
    nb = 1;
  }
  L->plan.assign((size_t)nb, EdgeWindows{});
  for (int b = 0; b < nb; ++b) {
    EdgeWindows& W = L->plan[b];
    W.n = 0;
    for (int sl = 0; sl < L->S; ++sl) {
      const uint32_t* lo = L->h_seg_row.data() + L->seg_base[sl];
      const uint32_t* hi = L->h_seg_row.data() + L->seg_base[sl + 1];
      const size_t g_lo = (size_t)(std::lower_bound(lo, hi, blocks[b].j_lo) - L->h_seg_row.data());
      const size_t g_hi = (size_t)(std::lower_bound(lo, hi, blocks[b].j_hi) - L->h_seg_row.data());
      const unsigned long long e_lo = g_lo < L->seg_base[sl + 1] ? L->h_seg_off[g_lo] : L->edge_base[sl + 1];
      const unsigned long long e_hi = g_hi < L->seg_base[sl + 1] ? L->h_seg_off[g_hi] : L->edge_base[sl + 1];
      if (e_hi <= e_lo) continue;
      if (W.n && W.e_hi[W.n - 1] == e_lo) {
        W.e_hi[W.n - 1] = e_hi;
      } else {
        W.e_lo[W.n] = e_lo;
        W.e_hi[W.n] = e_hi;
        W.n++;
      }
    }
    W.c_pre[0] = 0;
    for (uint32_t w = 0; w < W.n; ++w) {
      W.c_lo[w] = (uint32_t)(W.e_lo[w] / kChunkPad);
      const uint32_t c_hi = (uint32_t)((W.e_hi[w] + kChunkPad - 1) / kChunkPad);
      W.c_pre[w + 1] = W.c_pre[w] + (c_hi - W.c_lo[w]);
    }
  }
  L->plan_B = nb;
  return L->plan.data();
}

// Uploads the sliced copy of the (internal-order) in-CSR the host half of the lift built (lift.cpp); no layout when
// the source ids fit one slice.
static int upload_sliced_layout(pprhip_graph* G, HostLift& H) {
  if (H.S < 2) return PPRHIP_OK;
  std::unique_ptr<SlicedLayout> L(new (std::nothrow) SlicedLayout());
  if (!L) return PPRHIP_ERR_OOM;
  L->S = H.S;
  L->width = H.width;
  L->n_seg = H.n_seg;
  L->edge_base = std::move(H.edge_base);
  L->seg_base = std::move(H.seg_base);
  L->h_seg_row = std::move(H.seg_row);
  L->h_seg_off = std::move(H.seg_off);
  auto up = [&](void** dst, const void* src, size_t bytes) -> int {
    PPRHIP_TRY(alloc_dev(dst, bytes));
    if (bytes) PPRHIP_CHECK_HIP(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
    return PPRHIP_OK;
  };
  G->sl = L.release();  // from here on pprhip_graph_destroy frees what has been allocated
  PPRHIP_TRY(up((void**)&G->sl->ci, H.sl_ci.data(), sizeof(int32_t) * H.sl_ci.size()));
  PPRHIP_TRY(up((void**)&G->sl->flags, H.sl_flags.data(), H.sl_flags.size()));
  PPRHIP_TRY(up((void**)&G->sl->chunk_starts, H.sl_chunk_starts.data(), sizeof(uint32_t) * H.sl_chunk_starts.size()));
  PPRHIP_TRY(up((void**)&G->sl->seg_row, G->sl->h_seg_row.data(), sizeof(uint32_t) * G->sl->h_seg_row.size()));
  return PPRHIP_OK;
}

static int upload_panel_layout(pprhip_graph* G, HostLift& H) {
  if (!H.pn.n_items) return PPRHIP_OK;
  std::unique_ptr<PanelLayout> L(new (std::nothrow) PanelLayout());
  if (!L) return PPRHIP_ERR_OOM;
  L->n_panels = H.pn.n_panels;
  L->n_items = H.pn.n_items;
  L->n_part = H.pn.n_part;
  L->h_panel_item0 = std::move(H.pn.panel_item0);
  auto up = [&](void** dst, const void* src, size_t bytes) -> int {
    PPRHIP_TRY(alloc_dev(dst, bytes));
    if (bytes) PPRHIP_CHECK_HIP(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
    return PPRHIP_OK;
  };
  G->pn = L.release();  // from here on pprhip_graph_destroy frees what has been allocated
  PPRHIP_TRY(up((void**)&G->pn->src, H.pn.src.data(), sizeof(int32_t) * H.pn.src.size()));
  PPRHIP_TRY(up((void**)&G->pn->rloc, H.pn.rloc.data(), sizeof(uint16_t) * H.pn.rloc.size()));
  PPRHIP_TRY(up((void**)&G->pn->items, H.pn.items.data(), sizeof(PanelItem) * H.pn.items.size()));
  PPRHIP_TRY(up((void**)&G->pn->panels, H.pn.panels.data(), sizeof(PanelDesc) * H.pn.panels.size()));
  return PPRHIP_OK;
}

// the buffer the items of a panel sweep leave their sums in: per handle, on its first forward dense level
int ensure_panel_part(pprhip_graph* g) {
  if (!g->pn || g->pn_part) return PPRHIP_OK;
  PPRHIP_TRY(alloc_dev((void**)&g->pn_ctr, kPanelQueues * sizeof(uint32_t)));
  PPRHIP_CHECK_HIP(hipMemsetAsync(g->pn_ctr, 0, kPanelQueues * sizeof(uint32_t), g->stream));
  return alloc_dev((void**)&g->pn_part, sizeof(double) * (size_t)g->pn->n_part);
}

int ensure_bwd_layout(pprhip_graph* P);


// bookkeeping after a dense level: the frontier it produced becomes the current one
void finish_dense(LevelCtx& L, pprhip_stats_t& st, uint64_t level_bytes, uint64_t min_bytes, uint32_t nf_next,
                  uint64_t ef_next) {
  st.sweep_min_bytes += min_bytes;
  // after an entry / in-place sweep the new contributions have reached the later blocks only (nothing is pending
  // when the sweep prepared no node)
  L.gs_dirty = (L.gs_state == kGsEntry || L.gs_state == kGsInPlace) && nf_next > 0;
  L.dense_run++;
  L.ccur ^= 1;
  L.dslot ^= 1;
  L.pslot ^= 1;
  st.dense_levels++;
  st.dense_nodes += L.nf;
  st.dense_edges += L.ef;
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
int run_levels(pprhip_graph* g, const PushArgs& a, LevelCtx& L, pprhip_stats_t& st, double* model_cost,
               bool yield_dense, RoundCut* cut) {
  const bool bwd = a.mode == kBackward;
  const bool slot = g->parent != nullptr;
  // smallest integer x with (double)x >= dense_frac * m: the device-side form of level_cost()'s test
  const bool sparse_only = false;  // every push direction has both level shapes
  const unsigned long long dense_thresh =
      sparse_only ? ~0ull : (unsigned long long)std::ceil(g->tun.dense_frac * (double)g->m);
  const unsigned long long gs_thresh = bwd ? ~0ull : gs_thresh_of(g);
  int n_gs = 1;
  const GsBlock* gs_blocks = gs_thresh != ~0ull ? gs_blocks_of(g, &n_gs) : nullptr;
  while (L.nf > 0) {
    bool dense = false;
    double c = level_cost(g, L.nf, L.ef, &dense);
    if (sparse_only) dense = false;
    if (L.gs_dirty && !dense) {  // the contribution array has to be flushed by one more sweep
      dense = true;
      c = dense_sweep_cost(g);
    }
    if (dense) {
      if (!L.dense_prepared && slot && g->pooled && !g->has_col) {
        // workspace pool: the level needs a column of c8 (nothing has been decided or queued yet: the driver calls
        // again when one is free)
        int c = 0;
        while (c < kBatch && g->parent->col_owner[c] >= 0) ++c;
        if (c == kBatch) return kYieldColumn;
        g->parent->col_owner[c] = g->ws_index;
        g->slot_index = c;
        g->has_col = true;
      }
      if (model_cost) *model_cost += c;
      if (cut) cut->had_dense = true;
      if (bwd && !slot) PPRHIP_TRY(ensure_bwd_layout(g));  // sweep layout over the out-CSR, built on first use
      if (!bwd && !slot) PPRHIP_TRY(ensure_panel_part(g));  // (graphs with the row-panel copy)
      if (!L.dense_prepared) {
        C8Scope c8(g, false);
        PPRHIP_TRY(c8.rc);
        if (slot) {
          if (g->sync) g->sync->c8_enter(g->slot_index);
          L.ccur = g->parent->c8cur;  // the slot's column of the shared array is all-zero here
        } else
          PPRHIP_CHECK_HIP(hipMemsetAsync(g->cdense[L.ccur], 0, sizeof(double) * g->n, g->stream));
        PPRHIP_TRY(launch_sparse_prepare(g, a, L.fcur, 0, L.nf, dense_thresh, true, L.ccur, L.dslot,
                                         ((unsigned long long)L.nf << kPackShift) | L.ef));
        PPRHIP_TRY(c8.leave());
        L.dense_prepared = true;
        L.dense_run = 0;
        L.gs_dirty = false;
      }
      // state of this sweep (engine.hpp: GsState; the twin takes the same decision)
      {
        const bool big = (unsigned long long)L.nf + L.ef >= gs_thresh;
        L.gs_state = L.gs_dirty ? (big ? kGsInPlace : kGsFlush) : (big ? kGsEntry : kGsJacobi);
      }
      if (yield_dense) return kYield;
      // Dense levels are launched kDenseBatch at a time: level j > 0 of a batch reads its state from a device cell
      // that the level before it wrote (gs_next_state of the frontier it left; kGsNone: nothing left to sweep, the
      // level's kernels return at once), so the host reads the batch's counters back in one round trip.
      PPRHIP_CHECK_HIP(hipMemsetAsync(&g->ctr->dhist[0], 0, sizeof(unsigned long long) * 8 + sizeof(int) * 8, g->stream));
      size_t rec0[kDenseBatch];
      ktimer().reserve(kDenseBatch);
      for (int j = 0; j < kDenseBatch; ++j) {
        const int cc = L.ccur ^ (j & 1), ds = L.dslot ^ (j & 1), out = L.pslot ^ 1 ^ (j & 1);
        // The sweep writes contributions of non-empty rows only.  Rows without in-edges can hold one
        // solely from a phase's seeding, so the other buffer is cleared when a dense phase starts
        // and the seeded buffer right after its first level has consumed it.
        const bool first_of_phase = j == 0 && L.dense_run == 0;
        if (first_of_phase)
          PPRHIP_CHECK_HIP(hipMemsetAsync(g->cdense[cc ^ 1], 0, sizeof(double) * g->n, g->stream));
        ktimer().begin(PPRHIP_KERNEL_DENSE_PULL, dense_level_bytes(g));
        rec0[j] = ktimer().recs.size() - 1;
        DenseLaunch dl;
        dl.blocks = gs_blocks;
        dl.n_blocks = n_gs;
        dl.state_in = j ? &g->ctr->dstate[j] : nullptr;
        dl.state0 = L.gs_state;
        dl.hist_out = &g->ctr->dhist[j + 1];
        dl.state_out = &g->ctr->dstate[j + 1];
        dl.dense_thresh = dense_thresh;
        dl.gs_thresh = gs_thresh;
        PPRHIP_TRY(launch_dense_level(g, a, cc, out, ds, dl));
        ktimer().end();
        if (first_of_phase)
          PPRHIP_CHECK_HIP(hipMemsetAsync(g->cdense[cc], 0, sizeof(double) * g->n, g->stream));
      }
      PPRHIP_TRY(fetch_small(g, &g->ctr->dhist[0], &g->h_ctr->dhist[0], sizeof(unsigned long long) * 8 + sizeof(int) * 8));
      int state = L.gs_state;
      for (int j = 0; j < kDenseBatch; ++j) {
        if (j > 0) {
          // level j ran in the state the device derived from level j - 1's counter: the same function here
          const unsigned long long pk = g->h_ctr->dhist[j];
          state = gs_next_state(state, pk >> kPackShift, pk & kPackMask, dense_thresh, gs_thresh);
          if (state != g->h_ctr->dstate[j]) {
            set_error("dense batch: level %d ran in state %d, the host expects %d", j, g->h_ctr->dstate[j], state);
            return PPRHIP_ERR_STATE;
          }
          if (state == kGsNone) {
            for (int t = j; t < kDenseBatch; ++t)  // gated-off launches are not levels: keep them out of the class stats
              if (rec0[t] < ktimer().recs.size()) ktimer().recs[rec0[t]].cls = PPRHIP_KERNEL_NONE;
            break;
          }
          if (model_cost) *model_cost += dense_sweep_cost(g);  // a dense level costs the same whatever it pushes
          L.gs_state = state;
        }
        const unsigned long long nx = g->h_ctr->dhist[j + 1];
        finish_dense(L, st, dense_level_bytes(g), dense_level_min_bytes(g), (uint32_t)(nx >> kPackShift), nx & kPackMask);
      }
      continue;
    }
    // ---- a batch of sparse levels.  The first level's frontier travels as a kernel argument and its prepare kernel
    // clears the counters of the levels behind it; only after a compaction (which counts on the device) the counters
    // are cleared by a fill and read from memory.
    const bool first_prepared = L.dense_prepared || L.compacted;
    unsigned long long pk0 = ((unsigned long long)L.nf << kPackShift) | L.ef;
    if (L.dense_prepared) {
      {
        // A slot beside the sweeps queues this on the compute stream and learns of its end through its mailbox: an
        // event recorded there for the slot's stream to wait on held the compute stream up for ~90 us per compaction
        // (kernel trace: nothing ran between the compaction and the kernel queued right behind the record).
        C8Scope c8(g, false);
        PPRHIP_TRY(c8.rc);
        PPRHIP_CHECK_HIP(hipMemsetAsync(&g->ctr->hist[0], 0, sizeof(unsigned long long) * (kMaxBatch + 1), g->stream));
        // dense-prepared state -> list form; the compaction recounts (dead-end nodes carry no edges)
        PPRHIP_TRY(launch_compact_prepared(g, L.ccur, L.fcur, &g->ctr->hist[0], bwd));
        L.compact_seq = 0;
        if (c8.on) PPRHIP_TRY(fetch_begin(g, &g->ctr->hist[0], sizeof(unsigned long long), &L.compact_seq));
        if (c8.on && !L.compact_seq) c8.back = true;  // (no mailbox: the slot's stream waits for an event after all)
        PPRHIP_TRY(c8.leave());
      }
      L.dense_prepared = false;
      L.compacted = true;
      // the column must be read (and handed back zeroed) before another sweep may run
      if (g->sync) PPRHIP_CHECK_HIP(hipStreamSynchronize(g->stream));
      if (L.defer_compact) return kYieldDefer;  // (queued ahead of the next sweep; the levels follow beside it)
    }
    if (first_prepared) pk0 = ~0ull;
    if (L.compacted && L.compact_seq) {  // the list must be there before the slot's own stream reads it
      unsigned long long pk = 0;
      PPRHIP_TRY(fetch_end(g, L.compact_seq, &g->ctr->hist[0], &pk, sizeof pk));
      L.compact_seq = 0;
    }
    L.compacted = false;
    if (g->sync) g->sync->release(g->slot_index);
    // the round-cut check looks at the state after exactly one sparse level
    const bool cut_check = cut && cut->enabled && cut->had_dense && !cut->checked;
    const int n_batch = cut_check ? 1 : kMaxBatch;
    // The batch's levels from wg_from on run in ONE launch on one workgroup as long as they stay small
    // (k_sparse_levels_wg): from the first level when that is small itself, else behind one or two levels of the
    // usual two launches each - a frontier of 2^16 entries + edges or more rarely falls below the cap in one level.
    static const bool wg_on = !(hook_env("PPRHIP_SPARSE_WG") && hook_env("PPRHIP_SPARSE_WG")[0] == '0');
    constexpr unsigned long long kWgCap = 4096;
    const unsigned long long size0 = (unsigned long long)L.nf + L.ef;
    const int wg_from = (!wg_on || cut_check) ? n_batch
                        : (!first_prepared && size0 < kWgCap) ? 0
                        : size0 < 65536                        ? 1
                                                               : 2;
    ktimer().begin(PPRHIP_KERNEL_SPARSE_PUSH, 0);
    for (int i = 0; i < std::min(n_batch, wg_from); ++i) {
      const int fb = L.fcur ^ (i & 1);
      poll_idle(g);
      if (!(i == 0 && first_prepared))
        PPRHIP_TRY(launch_sparse_prepare(g, a, fb, i, i == 0 ? L.nf : 32768, dense_thresh, false, 0, L.dslot,
                                         i == 0 ? pk0 : ~0ull));
      PPRHIP_TRY(launch_sparse_push(g, a, fb, i, i == 0 ? L.ef : (1u << 20), dense_thresh, L.dslot, i == 0 ? pk0 : ~0ull));
    }
    if (wg_from < n_batch)
      PPRHIP_TRY(launch_sparse_levels_wg(g, a, L.fcur, wg_from, n_batch - 1, dense_thresh, kWgCap, L.dslot,
                                         wg_from == 0 ? pk0 : ~0ull));
    ktimer().end();
    PPRHIP_TRY(fetch_small(g, &g->ctr->hist[0], &g->h_ctr->hist[0], sizeof(unsigned long long) * (kMaxBatch + 1)));
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
        if (i >= wg_from && (unsigned long long)nf_i + ef_i >= kWgCap) break;  // ... too large for the one workgroup
        if (model_cost) *model_cost += ci;
      } else if (model_cost) {
        *model_cost += c;
      }
      const uint32_t nf_next = (uint32_t)(g->h_ctr->hist[i + 1] >> kPackShift);
      static const bool level_trace = hook_env("PPRHIP_LEVEL_TRACE") != nullptr;  // developer switch: a line per sparse level
      if (level_trace) fprintf(stderr, "[level] mode %d batch-level %d nf %u ef %llu\n", a.mode, i, nf_i, (unsigned long long)ef_i);
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

int reset_query_state(pprhip_graph* g, bool clear_flags, int32_t node) {
  poll_idle(g);
  // the entries the query before could have written are cleared; the new query's passes cover n_act entries
  const uint32_t n_live = host_of(g)->n_live;
  g->n_act = (n_live && node >= 0 && (uint32_t)node < n_live) ? n_live : g->n;
  const uint32_t clr = std::max(g->n_act, g->n_dirty ? g->n_dirty : g->n);
  g->n_dirty = g->n_act;
  ClearList cl{};  // one launch for all of them (five fill commands before: the device idled between them)
  auto add = [&](void* p, size_t bytes) {
    cl.p[cl.n] = p;
    cl.bytes[cl.n++] = bytes;
  };
  add(g->residue, sizeof(double) * clr);
  add(g->reserve, sizeof(double) * clr);
  add(g->ctr, sizeof(DevCounters));
  if (clear_flags) add(g->flags, clr);
  // the top-k estimate is rewritten over the new query's n_act entries only: what the query before left beyond them goes
  if (clr > g->n_act) add(g->est + g->n_act, sizeof(double) * (clr - g->n_act));
  {
    SetupScope setup(g);
    PPRHIP_TRY(launch_clear(g, cl));
  }
  g->mc_phase = g->mc_last_plan = 0;  // (the plan cells were just cleared)
  g->result_in_est = false;
  return PPRHIP_OK;
}

// per-query workspace of a handle (the graph's own, or a batch slot's)
int alloc_workspace(pprhip_graph* G) {
  const uint32_t n = G->n;
  const size_t nd = sizeof(double) * (size_t)n;
  void** dbl[] = {(void**)&G->residue, (void**)&G->reserve, (void**)&G->est, (void**)&G->cF};
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
  PPRHIP_TRY(alloc_dev((void**)&G->armed, sizeof(uint32_t) * ((size_t)n / 32 + 2)));
  PPRHIP_CHECK_HIP(hipMemsetAsync(G->armed, 0, sizeof(uint32_t) * ((size_t)n / 32 + 2), G->stream));
  PPRHIP_TRY(alloc_dev((void**)&G->mc_plan_rec, sizeof(WalkPlanRec) * (size_t)n));
  PPRHIP_TRY(alloc_dev((void**)&G->partial, sizeof(double) * 1024));
  PPRHIP_TRY(alloc_dev((void**)&G->hist, sizeof(uint32_t) * 4096));
  {
    const size_t nblk = std::max<size_t>(1024, ((size_t)n + 1 + 255) / 256) + 72;
    PPRHIP_TRY(alloc_dev((void**)&G->blk_pack, sizeof(unsigned long long) * nblk));
    PPRHIP_TRY(alloc_dev((void**)&G->blk_dead, sizeof(double) * nblk));
    PPRHIP_TRY(alloc_dev((void**)&G->blk_ndead, sizeof(uint32_t) * nblk));
  }
  G->sel_cap = 1u << 18;
  PPRHIP_TRY(alloc_dev((void**)&G->sel_blob, kSelHeader + sizeof(SelRec) * (size_t)G->sel_cap));
  PPRHIP_CHECK_HIP(hipMemsetAsync(G->hist, 0, sizeof(uint32_t) * 4096, G->stream));
  PPRHIP_CHECK_HIP(hipMemsetAsync(G->sel_blob, 0, kSelHeader, G->stream));
  PPRHIP_TRY(alloc_dev((void**)&G->ctr, sizeof(DevCounters)));
  if (hipHostMalloc((void**)&G->h_ctr, sizeof(DevCounters), hipHostMallocDefault) != hipSuccess) {
    set_error("hipHostMalloc failed");
    return PPRHIP_ERR_OOM;
  }
  std::memset(G->h_ctr, 0, sizeof(DevCounters));
  if (hipHostMalloc((void**)&G->mail, sizeof(HostMail), hipHostMallocMapped) != hipSuccess ||
      hipHostGetDevicePointer((void**)&G->mail_dev, G->mail, 0) != hipSuccess) {
    set_error("hipHostMalloc (mapped) failed");
    return PPRHIP_ERR_OOM;
  }
  std::memset(G->mail, 0, sizeof(HostMail));
  G->mail_seq = 0;
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
  void* ptrs[] = {g->pn_part, g->pn_ctr, g->acc_nz, g->residue, g->reserve, g->est, g->cdense[0], g->cdense[1], g->cF, g->F[0], g->F[1],
                  g->eoff[0], g->eoff[1], g->flags, g->armed, g->mc_plan_rec, g->partial, g->hist, g->sel_blob,
                  g->ctr, g->blk_pack, g->blk_dead, g->blk_ndead};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  if (g->h_ctr) (void)hipHostFree(g->h_ctr);
  if (g->mail) (void)hipHostFree(g->mail);
  g->mail = g->mail_dev = nullptr;
  if (g->spec_stream) {
    (void)hipStreamSynchronize(g->spec_stream);
    (void)hipStreamDestroy(g->spec_stream);
  }
  if (g->spec_mail) (void)hipHostFree(g->spec_mail);
  if (g->mc_plan_rec2) (void)hipFree(g->mc_plan_rec2);
  g->mc_plan_rec2 = nullptr;
  for (auto& e : g->spec_ev)
    if (e) (void)hipEventDestroy(e);
  g->spec_timer.destroy();
  g->spec_stream = nullptr;
  g->spec_mail = g->spec_mail_dev = nullptr;
  g->spec_ev[0] = g->spec_ev[1] = nullptr;
  for (auto e : g->ev)
    if (e) (void)hipEventDestroy(e);
}

void free_batch(pprhip_graph* P);
static int build_part_layout_device(pprhip_graph* P);

// one batch workspace: slots[w] works on column w % kBatch of the interleaved arrays
static int make_slot(pprhip_graph* P, int w) {
  pprhip_graph* S = new (std::nothrow) pprhip_graph();
  if (!S) return PPRHIP_ERR_OOM;
  P->slots.push_back(S);
  S->parent = P;
  S->slot_index = w % kBatch;
  S->ws_index = w;
  S->device = P->device;
  S->n_cus = P->n_cus;
  S->n = P->n;
  S->m = P->m;
  S->n_live = P->n_live;
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
  S->walk_rec = P->walk_rec;
  S->relabeled = P->relabeled;
  S->new2old = P->new2old;
  S->old2new = P->old2new;
  S->start_flags = P->start_flags;
  S->chunk_starts = P->chunk_starts;
  S->n_chunks = P->n_chunks;
  S->nz_rows = P->nz_rows;
  S->n_nz = P->n_nz;
  S->sl = P->sl;
  S->pn = nullptr;  // (slots run no dense levels of their own: the parent's shared sweeps serve them)
  S->tun = P->tun;
  PPRHIP_TRY(alloc_workspace(S));
  return PPRHIP_OK;
}

static void drop_slot(pprhip_graph* S) {
  for (auto& ev : S->walk_ev) {
    if (ev) (void)hipEventDestroy(ev);
    ev = nullptr;
  }
  for (auto& ev : S->c8_ev) {
    if (ev) (void)hipEventDestroy(ev);
    ev = nullptr;
  }
  if (S->col_ev) (void)hipEventDestroy(S->col_ev);
  S->col_ev = nullptr;
  free_workspace(S);
  S->ktimer.destroy();
  if (S->own_stream) (void)hipStreamDestroy(S->own_stream);
  delete S;
}

int ensure_workspaces(pprhip_graph* P, int count) {
  while ((int)P->slots.size() < count) {
    const size_t before = P->slots.size();
    const int rc = make_slot(P, (int)before);
    if (rc != PPRHIP_OK) {
      // a workspace that could not be completed (out of memory, mostly) must not stay in the list: the driver falls
      // back to the workspaces there are, and a later call tries again from a clean state
      if (P->slots.size() > before) {
        drop_slot(P->slots.back());
        P->slots.pop_back();
      }
      return rc;
    }
  }
  PPRHIP_CHECK_HIP(hipStreamSynchronize(P->stream));
  return PPRHIP_OK;
}

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
    P->col_owner[s] = -1;
    PPRHIP_TRY(make_slot(P, s));
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
  if (P->fetch) {
    P->fetch->destroy();
    delete P->fetch;
    P->fetch = nullptr;
  }
  if (P->walk_stream) (void)hipStreamDestroy(P->walk_stream);
  P->walk_stream = nullptr;
  P->walk_stream_tried = false;
  if (P->slot_stream) (void)hipStreamDestroy(P->slot_stream);
  P->slot_stream = nullptr;
  P->slot_stream_tried = false;
  for (pprhip_graph* S : P->slots) drop_slot(S);
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
      // a row that never receives; it can still hold a contribution of its own when it is a search's target, which
      // only matters to rows that pull from it - so rows that nobody points to are left out of the sweep altogether
      if (P->h_in_rp[v + 1] > P->h_in_rp[v]) zr.push_back((int32_t)v);
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
  PPRHIP_TRY(launch_seed_one(g, L.fcur, node));  // (one launch; a 4-byte copy command and a 4-byte fill before)
  L.nf = 1;
  L.ef = degree;
  L.dense_prepared = false;
  L.gs_dirty = false;
  return PPRHIP_OK;
}

// frontier from a predicate over all nodes (round starts)
int seed_scan(pprhip_graph* g, const PushArgs& a, int kind, LevelCtx& L) {
  if (kind == 1) {
    // top-k round starts: one pass that lists the start set, writes the armed bits and lets the parked nodes go, and
    // one read-back of its counter (a start set large enough for a sweep is prepared from the list by run_levels)
    {
      SetupScope setup(g);
      PPRHIP_TRY(launch_seed_list(g, a, 1, L.fcur, &g->ctr->hist[kMaxBatch + 2], true));
    }
    unsigned long long pk = 0;
    PPRHIP_TRY(fetch_small(g, &g->ctr->hist[kMaxBatch + 2], &pk, sizeof pk));
    L.nf = (uint32_t)(pk >> kPackShift);
    L.ef = pk & kPackMask;
    L.dense_prepared = false;
    L.gs_dirty = false;
    return PPRHIP_OK;
  }
  {
    SetupScope setup(g);
    PPRHIP_TRY(launch_count_active(g, a, kind, L.pslot));
  }
  PPRHIP_TRY(read_packed(g, L.pslot, &L.nf, &L.ef));
  L.dense_prepared = false;
  L.gs_dirty = false;
  bool dense = false;
  if (L.nf) (void)level_cost(g, L.nf, L.ef, &dense);
  // (a pooled workspace that holds no column lists the start set instead; run_levels prepares the level from the
  // list once it has one)
  if (dense && g->parent && g->pooled && !g->has_col) dense = false;
  if (dense) {
    C8Scope c8(g, false);
    PPRHIP_TRY(c8.rc);
    if (g->parent) {
      if (g->sync) g->sync->c8_enter(g->slot_index);
      L.ccur = g->parent->c8cur;
    }
    {
      SetupScope setup(g);
      PPRHIP_TRY(launch_seed_dense(g, a, kind, L.ccur, L.pslot, L.dslot));
    }
    PPRHIP_TRY(c8.leave());
    L.dense_prepared = true;
    L.dense_run = 0;
  } else if (L.nf || kind == 1) {
    // kind 1 also runs for an empty start set: parked nodes below min_rmax still leave the set
    // (Forward_Push.java:241-247)
    // (its list counter, hist[kMaxBatch + 2], was cleared by the counting pass above)
    {
      SetupScope setup(g);
      PPRHIP_TRY(launch_seed_list(g, a, kind, L.fcur, &g->ctr->hist[kMaxBatch + 2]));
    }
  }
  return PPRHIP_OK;
}

int device_sum(pprhip_graph* g, const double* x, double* out, uint32_t count) {
  poll_idle(g);
  {
    SetupScope setup(g);
    PPRHIP_TRY(launch_sum(g, x, count ? count : act_n(g)));
  }
  PPRHIP_TRY(fetch_small(g, &g->ctr->sum_out, &g->h_ctr->sum_out, sizeof(double)));
  *out = g->h_ctr->sum_out;
  return PPRHIP_OK;
}

// The counters a query only needs once, at its end, in one copy: dead-end pops of the push, and what the walk phases
// run since the workspace was reset counted on the device (steps, walks, sources: adjacent in DevCounters).
int read_dead_pops(pprhip_graph* g, pprhip_stats_t& st) {
  poll_idle(g);
  static_assert(offsetof(DevCounters, walk_lanes) == offsetof(DevCounters, dead_pops) + 40, "one copy for the six");
  PPRHIP_TRY(fetch_small(g, &g->ctr->dead_pops, &g->h_ctr->dead_pops, 6 * sizeof(unsigned long long)));
  st.walk_loads = g->h_ctr->walk_loads;
  st.walk_load_lanes = g->h_ctr->walk_lanes;
  st.push_bytes += 16ull * (g->h_ctr->dead_pops - st.dead_end_pops);
  st.dead_end_pops = g->h_ctr->dead_pops;
  // cumulative over the query's walk phases: what is new since the last read goes into the statistics
  const uint64_t steps = g->h_ctr->walk_steps, walks = g->h_ctr->walks_total, srcs = g->h_ctr->sources_total;
  if (steps >= st.walk_steps && walks >= st.walks && srcs >= st.mc_sources) {
    const uint64_t more = 12ull * (steps - st.walk_steps) + 16ull * (walks - st.walks) + 12ull * (srcs - st.mc_sources);
    st.mc_bytes += more;
    ktimer().add_bytes(PPRHIP_KERNEL_WALK, more);
    st.walk_steps = steps;
    st.walks = walks;
    st.mc_sources = srcs;
  }
  return PPRHIP_OK;
}

// Walk phase shared by FORA whole-graph (variant 0) and top-k (variant 1): plan and walks are launched back to back,
// the walk kernel reads the plan's counts on the device (no host round trip inside the phase; the counts reach the
// statistics through read_dead_pops at the end of the query).  omega_dev > 0: the plan also derives rsum and the walk
// budget on the device from the residue sum a device_sum / launch_sum has just left (rsum, nrw are ignored; nrw_bound is
// the largest budget possible, for the range check).
// The walk phase in two halves (a caller may queue other work between them, or run the plan on another stream):
// the plan of the residue entries, and the walk kernel that runs the latest plan.
int launch_walk_plan(pprhip_graph* g, int variant, double alpha, double rsum, long long nrw, double* target, double omega_dev,
                     const double* copy_src, double* copy_dst) {
  poll_idle(g);
  // what the host knows about the walk count sizes the grid: the budget itself (every residue entry adds at most one
  // walk to it), or - with the budget derived on the device - nothing
  g->walk_hint = omega_dev > 0.0 ? 0ull : (unsigned long long)nrw + act_n(g);
  const double bound = omega_dev > 0.0 ? omega_dev : (double)nrw;
  if (bound + (double)g->n >= (double)(1ull << kPackShift)) {
    set_error("walk budget %.0f exceeds the engine's 2^36 walk limit", bound);
    return PPRHIP_ERR_INVALID;
  }
  SetupScope setup(g);
  return launch_mc_plan(g, variant, alpha, rsum, (double)nrw, omega_dev, target, copy_src, copy_dst);
}

int launch_walk_run(pprhip_graph* g, int variant, double alpha, uint64_t seed, uint32_t stream, double* target) {
  poll_idle(g);
  ktimer().begin(PPRHIP_KERNEL_WALK, 0);  // (its bytes are added when the counters are read)
  PPRHIP_TRY(launch_mc_walk(g, alpha, seed, stream, variant == 0 ? 1 : 0, target));
  ktimer().end();
  return PPRHIP_OK;
}

int run_walk_phase(pprhip_graph* g, int variant, double alpha, double rsum, long long nrw, uint64_t seed, uint32_t stream,
                   double* target, pprhip_stats_t& st, double omega_dev) {
  (void)st;
  PPRHIP_TRY(launch_walk_plan(g, variant, alpha, rsum, nrw, target, omega_dev));
  return launch_walk_run(g, variant, alpha, seed, stream, target);
}


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
  if (g->stream_open) {
    set_error("%s: a query stream is open on this handle (pprhip_fora_stream_close first)", fn);
    return PPRHIP_ERR_STATE;
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

// candidates (all entries >= the lower edge of the bin that holds the k-th largest) -> the reference's answer
static void finish_select(std::vector<IdVal>& cand, bool have, int k, int32_t* ids_out, double* vals_out, int cap,
                          int* n_out, double* kth_out, bool* have_kth, pprhip_stats_t& st) {
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
}

// The multi-pass form: the host reads every histogram and refines the prefix until few enough candidates are left
// (needed when more than sel_cap entries share the leading 12 bits of the k-th largest).
static int select_topk_passes(pprhip_graph* g, const double* x, int k, int32_t* ids_out, double* vals_out, int cap,
                              int* n_out, double* kth_out, bool* have_kth, pprhip_stats_t& st) {
  std::vector<uint32_t> hist(4096);
  unsigned long long prefix = 0;
  int pbits = 0;
  uint64_t k_rem = (uint64_t)k;
  uint64_t above = 0;  // entries in bins above the chosen prefix
  uint64_t total = 0;
  bool have = true;
  unsigned long long lower_bits = 1ull;  // smallest positive pattern: "everything"
  uint64_t expected = ~0ull;              // candidates the gather will find, known from the histograms
  for (int pass = 0; pbits < 64; ++pass) {
    const int dbits = std::min(12, 64 - pbits);
    PPRHIP_TRY(launch_select_hist(g, x, act_n(g), prefix, pbits, dbits, pass == 0));
    PPRHIP_CHECK_HIP(hipMemcpyAsync(hist.data(), g->hist, sizeof(uint32_t) * (1u << dbits), hipMemcpyDeviceToHost,
                                    g->stream));
    PPRHIP_CHECK_HIP(hipStreamSynchronize(g->stream));
    st.select_passes++;
    st.select_bytes += 8ull * g->n;
    if (pass == 0) {
      for (uint32_t b = 0; b < (1u << dbits); ++b) total += hist[b];
      if (total == 0) {
        PPRHIP_CHECK_HIP(hipMemsetAsync(g->hist, 0, sizeof(uint32_t) * 4096, g->stream));
        *n_out = 0;
        *have_kth = false;
        if (kth_out) *kth_out = 0.0;
        return PPRHIP_OK;
      }
      if ((uint64_t)k > total) {  // kth_ppr returns null: everything is kept (Fora_Topk.java:187-191)
        have = false;
        expected = total;
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
    expected = above + hist[chosen];
    if (expected <= g->sel_cap) break;  // few enough candidates: finish on the host
  }
  PPRHIP_TRY(launch_select_gather(g, x, act_n(g), have ? lower_bits : 1ull, false));
  // the histograms already say how many candidates there are: the count and the records come back in ONE copy
  const bool prefetched = expected > 0 && expected <= g->sel_cap;
  const size_t want = prefetched ? (size_t)expected : 0;
  std::vector<char> blob(kSelHeader + sizeof(SelRec) * want);
  PPRHIP_TRY(fetch_small(g, g->sel_blob, blob.data(), blob.size()));
  st.select_bytes += 8ull * g->n;
  uint64_t cnt = 0;
  std::memcpy(&cnt, blob.data(), 8);
  std::vector<IdVal> cand;
  auto take_recs = [&](const char* p, uint64_t c) {
    cand.resize(c);
    const SelRec* r = reinterpret_cast<const SelRec*>(p);
    for (uint64_t i = 0; i < c; ++i) cand[i] = {host_of(g)->h_new2old[r[i].id], r[i].val};
  };
  if (prefetched && cnt == expected) {
    take_recs(blob.data() + kSelHeader, cnt);
  } else if (cnt <= g->sel_cap) {
    std::vector<char> more(sizeof(SelRec) * cnt);
    if (cnt) {
      PPRHIP_CHECK_HIP(hipMemcpyAsync(more.data(), g->sel_blob + kSelHeader, more.size(), hipMemcpyDeviceToHost, g->stream));
      PPRHIP_CHECK_HIP(hipStreamSynchronize(g->stream));
    }
    take_recs(more.data(), cnt);
  } else {
    // more ties at the k-th value than the candidate buffer holds: finish on the whole vector
    std::vector<double> all(g->n);
    PPRHIP_CHECK_HIP(hipMemcpyAsync(all.data(), x, sizeof(double) * g->n, hipMemcpyDeviceToHost, g->stream));
    PPRHIP_CHECK_HIP(hipStreamSynchronize(g->stream));
    const double lb = [&] { double d; std::memcpy(&d, &lower_bits, 8); return d; }();
    for (uint32_t i = 0; i < g->n; ++i)
      if (all[i] > 0.0 && (!have || all[i] >= lb)) cand.push_back({host_of(g)->h_new2old[i], all[i]});
  }
  finish_select(cand, have, k, ids_out, vals_out, cap, n_out, kth_out, have_kth, st);
  return PPRHIP_OK;
}

// k-th largest and the entries >= it (Algo_Util.kth_ppr + retrieveTopK).  One histogram pass over the 12 leading bits,
// the bin of the k-th largest chosen on the device, the gather of everything from that bin's lower edge up, and ONE
// read-back (header + the first kPre records; the candidates are then ordered on the host, values descending, ids
// ascending).  Only when more candidates share those 12 bits than the buffer holds the multi-pass form takes over.
// The selection in two halves, so that a caller can queue other work between launching it and waiting for it.
constexpr size_t kSelPre = 2048;
int select_launch(pprhip_graph* g, const double* x, int k, unsigned long long* seq_out, bool with_plan_sum) {
  poll_idle(g);
  {
    SetupScope setup(g);
    PPRHIP_TRY(launch_select_hist(g, x, act_n(g), 0ull, 0, 12, true));
    PPRHIP_TRY(launch_select_choose(g, (unsigned long long)k));
    // with_plan_sum: the residue sum of the round whose plan ran last travels in the header (DevCounters::plan_sum)
    PPRHIP_TRY(launch_select_gather(g, x, act_n(g), 0ull, false, true,
                                    with_plan_sum ? &g->ctr->plan_sum[g->mc_last_plan % 3u] : nullptr));
  }
  return fetch_begin(g, g->sel_blob, kSelHeader + sizeof(SelRec) * kSelPre, seq_out);
}

int select_topk(pprhip_graph* g, const double* x, int k, int32_t* ids_out, double* vals_out, int cap, int* n_out,
                double* kth_out, bool* have_kth, pprhip_stats_t& st, bool with_plan_sum) {
  unsigned long long seq = 0;
  PPRHIP_TRY(select_launch(g, x, k, &seq, with_plan_sum));
  return select_finish(g, seq, x, k, ids_out, vals_out, cap, n_out, kth_out, have_kth, st);
}

int select_finish(pprhip_graph* g, unsigned long long seq, const double* x, int k, int32_t* ids_out, double* vals_out,
                  int cap, int* n_out, double* kth_out, bool* have_kth, pprhip_stats_t& st) {
  constexpr size_t kPre = kSelPre;
  std::vector<char> blob(kSelHeader + sizeof(SelRec) * kPre);
  PPRHIP_TRY(fetch_end(g, seq, g->sel_blob, blob.data(), blob.size()));
  st.select_passes++;
  st.select_bytes += 16ull * act_n(g);
  unsigned long long hdr[6];
  std::memcpy(hdr, blob.data(), sizeof hdr);
  const uint64_t cnt = hdr[0], expected = hdr[2], total = hdr[3];
  const bool have = hdr[4] != 0;
  std::memcpy(&g->sel_plan_sum, &hdr[5], sizeof(double));  // (meaningful after select_launch(..., with_plan_sum))
  if (total == 0) {
    *n_out = 0;
    *have_kth = false;
    if (kth_out) *kth_out = 0.0;
    return PPRHIP_OK;
  }
  if (expected > g->sel_cap || cnt != expected)  // too many share the leading bits (or the header is not what it should be)
    return select_topk_passes(g, x, k, ids_out, vals_out, cap, n_out, kth_out, have_kth, st);
  std::vector<IdVal> cand(cnt);
  const std::vector<int32_t>& n2o = host_of(g)->h_new2old;
  if (cnt <= kPre) {
    const SelRec* r = reinterpret_cast<const SelRec*>(blob.data() + kSelHeader);
    for (uint64_t i = 0; i < cnt; ++i) cand[i] = {n2o[r[i].id], r[i].val};
  } else {
    std::vector<SelRec> more(cnt);
    PPRHIP_CHECK_HIP(hipMemcpyAsync(more.data(), g->sel_blob + kSelHeader, sizeof(SelRec) * cnt, hipMemcpyDeviceToHost, g->stream));
    PPRHIP_CHECK_HIP(hipStreamSynchronize(g->stream));
    for (uint64_t i = 0; i < cnt; ++i) cand[i] = {n2o[more[i].id], more[i].val};
  }
  finish_select(cand, have, k, ids_out, vals_out, cap, n_out, kth_out, have_kth, st);
  return PPRHIP_OK;
}

// A stream that really runs beside the handle's compute stream.  The runtime spreads streams over a few in-order
// hardware queues, and which streams share one depends on what else the process has created: a stream that lands on
// the compute stream's queue never overlaps it (fora.cpp: FetchPipe, tools/exp/copy_overlap.py: kernels ran during
// 0.0 % of the copies' time).  So candidates are created - plain ones first, then of the other priorities - and each
// is tried: a kernel holds the compute stream for a moment, a one-word k_publish goes to the candidate, and the
// candidate is taken if the word arrives while the hold kernel still runs.  Rejected candidates stay alive until the
// search ends, so that the next one lands elsewhere.  *out stays null when none ran beside.
int make_side_stream(pprhip_graph* g, hipStream_t* out, hipStream_t also) {
  *out = nullptr;
  int prio_lo = 0, prio_hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
  HostMail* probe = nullptr;
  HostMail* probe_dev = nullptr;
  if (hipHostMalloc((void**)&probe, sizeof(HostMail), hipHostMallocMapped) != hipSuccess ||
      hipHostGetDevicePointer((void**)&probe_dev, probe, 0) != hipSuccess) {
    (void)hipGetLastError();
    if (probe) (void)hipHostFree(probe);
    return PPRHIP_ERR_OOM;
  }
  std::memset(probe, 0, sizeof(HostMail));
  hipEvent_t held = nullptr, held2 = nullptr;
  if (hipEventCreateWithFlags(&held, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&held2, hipEventDisableTiming) != hipSuccess) {
    if (held) (void)hipEventDestroy(held);
    (void)hipHostFree(probe);
    return PPRHIP_ERR_HIP;
  }
  PPRHIP_CHECK_HIP(hipStreamSynchronize(g->stream));
  if (also) PPRHIP_CHECK_HIP(hipStreamSynchronize(also));
  const int prios[] = {0, 0, 0, 0, prio_lo, prio_hi, prio_lo, prio_hi};
  std::vector<hipStream_t> rejected;
  unsigned long long seq = 0;
  for (int p : prios) {
    hipStream_t cand = nullptr;
    const hipError_t ce = p == 0 ? hipStreamCreateWithFlags(&cand, hipStreamNonBlocking)
                                 : hipStreamCreateWithPriority(&cand, hipStreamNonBlocking, p);
    if (ce != hipSuccess) break;
    bool beside = false;
    ++seq;
    hipStream_t own = g->stream;
    HostMail *m = g->mail, *md = g->mail_dev;
    const bool also_held = !also || (launch_hold(also, 30000ull) == PPRHIP_OK && hipEventRecord(held2, also) == hipSuccess);
    if (also_held && launch_hold(own, 30000ull) == PPRHIP_OK && hipEventRecord(held, own) == hipSuccess) {  // ~0.3 ms at 100 MHz
      g->stream = cand;
      g->mail = probe;
      g->mail_dev = probe_dev;
      const int rc = launch_publish(g, &g->ctr->sum_out, 1, seq);
      g->stream = own;
      g->mail = m;
      g->mail_dev = md;
      if (rc == PPRHIP_OK) {
        const auto t0 = std::chrono::steady_clock::now();
        while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < 150.0) {
          if (__atomic_load_n(&probe->seq, __ATOMIC_ACQUIRE) == seq) {
            // arrived while the hold kernel(s) still run
            beside = hipEventQuery(held) == hipErrorNotReady && (!also || hipEventQuery(held2) == hipErrorNotReady);
            break;
          }
          __builtin_ia32_pause();
        }
      }
    }
    (void)hipStreamSynchronize(own);
    if (also) (void)hipStreamSynchronize(also);
    (void)hipStreamSynchronize(cand);
    if (beside) {
      *out = cand;
      break;
    }
    rejected.push_back(cand);
  }
  for (hipStream_t r : rejected) (void)hipStreamDestroy(r);
  (void)hipEventDestroy(held);
  (void)hipEventDestroy(held2);
  (void)hipHostFree(probe);
  return PPRHIP_OK;
}

}  // namespace detail
}  // namespace pprhip

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

int pprhip_set_kernel_timing(int on) {
  const int was = kernel_timing_level();
  g_kernel_timing.store(on == 2 ? 2 : (on ? 1 : 0), std::memory_order_relaxed);
  return was;
}

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
  t->gs_blocks = 2;
  t->gs_frac = 0.1;
}

void pprhip_tuning_batch(pprhip_tuning_t* t) {
  pprhip_tuning_default(t);
  if (!t) return;
  // fitted on R-MAT 22 with all slots busy: a sweep of 1.6 ms serves ~14.5 queries; a level that touches
  // fewer than 2 % of the edges is cheaper as a sparse level (one memory-side atomic per edge)
  t->c_dense_edge_ns = 0.002;
  t->c_dense_node_ns = 0.003;
  t->dense_frac = 0.02;
  t->gs_frac = 0.05;
}

// The batch profile for a call of q queries on one GPU (config #4's shares: 50 queries over 8 GPUs are 6-7 per call).  A
// sweep costs the same whatever the number of busy columns, so a dense level costs each query of a small call more:
// the dense constants and the level-shape thresholds scale with 14.5 / min(q, 14.5) columns, up to the single-query
// profile's values (a call of one IS the single-query path).  Chosen by the caller before the first query starts, so
// every query of the call is pprhip_fora_single_source under the same tuning.
void pprhip_tuning_batch_for(int q, pprhip_tuning_t* t) {
  pprhip_tuning_batch(t);
  if (!t || q >= 15) return;
  pprhip_tuning_t one;
  pprhip_tuning_default(&one);
  const double scale = 14.5 / (double)std::max(1, q);
  t->c_dense_edge_ns = std::min(one.c_dense_edge_ns, t->c_dense_edge_ns * scale);
  t->c_dense_node_ns = std::min(one.c_dense_node_ns, t->c_dense_node_ns * scale);
  t->dense_frac = std::min(one.dense_frac, t->dense_frac * scale);
  t->gs_frac = std::min(one.gs_frac, t->gs_frac * scale);
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
  PPRHIP_TRY(init_device_once(device));
  const bool have_in = in_rp && (in_ci || m == 0);
  if (have_in && (in_rp[0] != 0 || in_rp[n] != m)) {
    set_error("pprhip_graph_create: in_row_ptr[0] must be 0 and in_row_ptr[n] must equal m");
    return PPRHIP_ERR_INVALID;
  }
  // ---- the host half (lift.cpp): validation, internal vertex order, both CSRs in that order, sweep layouts
  const auto t_lift0 = std::chrono::steady_clock::now();
  HostLift H;
  try {
    PPRHIP_TRY(lift_host(n, m, out_rp, out_ci, in_rp, in_ci, 0, H));
  } catch (const std::bad_alloc&) {
    set_error("pprhip_graph_create: out of host memory");
    return PPRHIP_ERR_OOM;
  }
  const auto t_lift1 = std::chrono::steady_clock::now();
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
  g->relabeled = H.relabeled;
  g->h_new2old = std::move(H.new2old);
  g->h_old2new = std::move(H.old2new);
  g->h_out_rp = std::move(H.out_rp);
  g->h_in_rp = std::move(H.in_rp);
  g->h_nz_rows = std::move(H.nz_rows);
  g->n_chunks = H.n_chunks;
  g->n_nz = (uint32_t)g->h_nz_rows.size();
  g->n_zin = (uint32_t)H.zin_rows.size();
  g->n_live = g->relabeled ? g->n_nz + g->n_zin : 0u;  // (ids are the caller's without the relabeling: no bound)
  g->n_src_live = H.n_src_live;

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
  if ((rc = up((void**)&G->out_ci, H.out_ci.data(), sizeof(int32_t) * H.out_ci.size()))) return fail(rc);
  if ((rc = up((void**)&G->out_ext, H.ext.data(), sizeof(unsigned long long) * (size_t)n))) return fail(rc);
  if ((rc = up((void**)&G->in_rp, G->h_in_rp.data(), sizeof(uint32_t) * ((size_t)n + 1)))) return fail(rc);
  if ((rc = up((void**)&G->in_ci, H.in_ci.data(), sizeof(int32_t) * H.in_ci.size()))) return fail(rc);
  if ((rc = up((void**)&G->new2old, G->h_new2old.data(), sizeof(int32_t) * (size_t)n))) return fail(rc);
  if ((rc = up((void**)&G->old2new, G->h_old2new.data(), sizeof(int32_t) * (size_t)n))) return fail(rc);
  if ((rc = up((void**)&G->start_flags, H.flags.data(), H.flags.size()))) return fail(rc);
  if ((rc = up((void**)&G->chunk_starts, H.chunk_starts.data(), sizeof(uint32_t) * H.chunk_starts.size()))) return fail(rc);
  if ((rc = up((void**)&G->nz_rows, G->h_nz_rows.data(), sizeof(int32_t) * G->h_nz_rows.size()))) return fail(rc);
  if ((rc = upload_sliced_layout(G, H))) return fail(rc);
  if ((rc = upload_panel_layout(G, H))) return fail(rc);
  if (hipStreamCreateWithFlags(&G->stream, hipStreamNonBlocking) != hipSuccess) {
    set_error("hipStreamCreate failed");
    return fail(PPRHIP_ERR_HIP);
  }
  if ((rc = up((void**)&G->zin_rows, H.zin_rows.data(), sizeof(int32_t) * H.zin_rows.size()))) return fail(rc);
  if ((rc = up((void**)&G->cross_bits, H.cross.data(), sizeof(unsigned long long) * H.cross.size()))) return fail(rc);
  if (hook_env("PPRHIP_LIFT_DEBUG")) {
    const auto t_up = std::chrono::steady_clock::now();
    fprintf(stderr, "[pprhip lift] host half %.1f ms, uploads %.1f ms\n",
            std::chrono::duration<double, std::milli>(t_lift1 - t_lift0).count(),
            std::chrono::duration<double, std::milli>(t_up - t_lift1).count());
  }
  if ((rc = alloc_dev((void**)&G->walk_rec, sizeof(uint4) * (size_t)m))) return fail(rc);
  if ((rc = launch_build_walk_rec(G))) return fail(rc);
  if ((rc = alloc_workspace(G))) return fail(rc);
  if (hipStreamSynchronize(G->stream) != hipSuccess) {
    set_error("stream sync after graph upload failed");
    return fail(PPRHIP_ERR_HIP);
  }
  *graph_out = g.release();
  return PPRHIP_OK;
}

static void free_all_pair(pprhip_graph* g) {
  void** ptrs[] = {(void**)&g->apbs_ws, (void**)&g->apbs_board, (void**)&g->apbs_xl_ws, (void**)&g->in_rec};
  for (void** p : ptrs) {
    if (*p) (void)hipFree(*p);
    *p = nullptr;
  }
  g->apbs_blocks = 0;  // (all_pair_collect sizes and allocates the workspaces when it finds none)
  if (g->ix_stage) (void)hipHostFree(g->ix_stage);
  g->ix_stage = nullptr;
  g->ix_stage_bytes = 0;
}

int pprhip_graph_release(pprhip_graph_t* g, unsigned what) {
  PPRHIP_TRY(check_graph(g, "pprhip_graph_release"));
  if (what & ~(PPRHIP_RELEASE_ALL_PAIR | PPRHIP_RELEASE_BATCH)) {
    set_error("pprhip_graph_release: unknown flag in %u", what);
    return PPRHIP_ERR_INVALID;
  }
  PPRHIP_CHECK_HIP(hipStreamSynchronize(g->stream));
  if (what & PPRHIP_RELEASE_ALL_PAIR) free_all_pair(g);
  if (what & PPRHIP_RELEASE_BATCH) free_batch(g);
  return PPRHIP_OK;
}

void pprhip_graph_destroy(pprhip_graph_t* g) {
  if (!g) return;
  if (g->stream_obj) stream_detach(g->stream_obj);  // (its driver thread uses the handle; the stream object stays its owner's)
  (void)hipSetDevice(g->device);
  if (g->stream) (void)hipStreamSynchronize(g->stream);
  free_batch(g);
  void* ptrs[] = {g->walk_rec, g->out_ext, g->out_rp, g->out_ci, g->in_rp, g->in_ci, g->new2old, g->old2new, g->start_flags,
                  g->chunk_starts, g->nz_rows, g->zin_rows, g->cross_bits, g->start_flags_o, g->chunk_starts_o,
                  g->nz_rows_o, g->z_rows_o, g->cross_bits_o};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  free_all_pair(g);
  if (g->sl) {
    void* sp[] = {g->sl->ci, g->sl->flags, g->sl->chunk_starts, g->sl->seg_row};
    for (void* p : sp)
      if (p) (void)hipFree(p);
    delete g->sl;
    g->sl = nullptr;
  }
  if (g->pn) {
    void* pp[] = {g->pn->src, g->pn->rloc, g->pn->items, g->pn->panels};
    for (void* p : pp)
      if (p) (void)hipFree(p);
    delete g->pn;
    g->pn = nullptr;
  }
  free_workspace(g);
  if (g->stream) (void)hipStreamDestroy(g->stream);
  delete g;
}

int pprhip_device_memory(const pprhip_graph_t* g, uint64_t* free_bytes, uint64_t* total_bytes) {
  if (!g) {
    set_error("pprhip_device_memory: null handle");
    return PPRHIP_ERR_INVALID;
  }
  PPRHIP_CHECK_HIP(hipSetDevice(g->device));
  size_t f = 0, t = 0;
  PPRHIP_CHECK_HIP(hipMemGetInfo(&f, &t));
  if (free_bytes) *free_bytes = (uint64_t)f;
  if (total_bytes) *total_bytes = (uint64_t)t;
  return PPRHIP_OK;
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
  if (g->tun.gs_blocks <= 0) g->tun.gs_blocks = d.gs_blocks;                  // 1: plain Jacobi sweeps
  if (g->tun.gs_blocks > 64) g->tun.gs_blocks = 64;
  if (!(g->tun.gs_frac > 0)) g->tun.gs_frac = d.gs_frac;
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
  PPRHIP_TRY(reset_query_state(g, false, src));
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
  PPRHIP_TRY(reset_query_state(g, true, src));
  // Q = {s} (Fora_Topk.java:117-118): the source starts parked
  PPRHIP_CHECK_HIP(hipMemsetAsync(g->flags + src, 1, 1, g->stream));
  g->topk_active = true;
  g->topk_first = true;
  g->topk_src = src;
  g->topk_alpha = alpha;
  g->topk_rsum = 1.0;
  return PPRHIP_OK;
}

// kSumRead: bring the residue sum to the host (the public round-by-round entry point); kSumLaunch: leave it in
// DevCounters::sum_out for the walk plan; kSumNone: the caller sums later (a push run ahead of its round)
enum { kSumRead = 0, kSumLaunch = 1, kSumNone = 2 };
static int topk_round_impl(pprhip_graph_t* g, double min_rmax, double rmax, pprhip_stats_t& st, int sum_mode = kSumRead) {
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
  if (sum_mode == kSumRead) PPRHIP_TRY(device_sum(g, g->residue, &g->topk_rsum));
  else if (sum_mode == kSumLaunch) PPRHIP_TRY(launch_sum_partial(g, g->residue, act_n(g)));  // (the plan adds them up)
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
// The second stream of pprhip_fora_topk (make_side_stream picks one that runs beside the compute stream), its host
// mail, its plan record buffer and its events.
static int ensure_spec(pprhip_graph* g) {
  if (g->spec_stream) return PPRHIP_OK;
  if (g->spec_failed) return PPRHIP_ERR_STATE;
  int prio_lo = 0, prio_hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
  if (prio_lo == prio_hi) return PPRHIP_ERR_STATE;  // no second queue to be had: the rounds run one after another
  if (!g->spec_mail) {
    if (hipHostMalloc((void**)&g->spec_mail, sizeof(HostMail), hipHostMallocMapped) != hipSuccess) {
      (void)hipGetLastError();
      g->spec_mail = nullptr;
      return PPRHIP_ERR_OOM;
    }
    std::memset(g->spec_mail, 0, sizeof(HostMail));
  }
  if (hipHostGetDevicePointer((void**)&g->spec_mail_dev, g->spec_mail, 0) != hipSuccess) return PPRHIP_ERR_HIP;
  if (!g->mc_plan_rec2 && alloc_dev((void**)&g->mc_plan_rec2, sizeof(WalkPlanRec) * (size_t)g->n) != PPRHIP_OK) {
    g->mc_plan_rec2 = nullptr;
    return PPRHIP_ERR_OOM;
  }
  for (auto& e : g->spec_ev)
    if (!e && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
      e = nullptr;
      return PPRHIP_ERR_HIP;
    }
  PPRHIP_TRY(make_side_stream(g, &g->spec_stream));
  if (!g->spec_stream) {
    g->spec_failed = true;  // no stream of this process runs beside the compute stream: the rounds run in order
    return PPRHIP_ERR_STATE;
  }
  return PPRHIP_OK;
}

namespace {
// While it lives, the handle launches on its second stream, reads back through that stream's mail and the calling
// thread times with that stream's timer.
struct SpecContext {
  pprhip_graph* g;
  hipStream_t stream;
  HostMail *mail, *mail_dev;
  unsigned long long seq;
  KernelTimer* timer;
  explicit SpecContext(pprhip_graph* g_) : g(g_), stream(g_->stream), mail(g_->mail), mail_dev(g_->mail_dev), seq(g_->mail_seq), timer(g_timer_cur) {
    g->stream = g->spec_stream;
    g->mail = g->spec_mail;
    g->mail_dev = g->spec_mail_dev;
    g->mail_seq = g->spec_mail_seq;
    g->spec_timer.stream = g->spec_stream;
    g->spec_timer.off = true;  // (its kernels run beside the compute stream's: their time is not the query's)
    g_timer_cur = &g->spec_timer;
  }
  ~SpecContext() {
    g->spec_mail_seq = g->mail_seq;
    g->stream = stream;
    g->mail = mail;
    g->mail_dev = mail_dev;
    g->mail_seq = seq;
    g_timer_cur = timer;
  }
};
}  // namespace

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
  const size_t nd = sizeof(double) * (size_t)act_n(g);  // (est beyond the query's scan bound is zero and stays so)
  bool dead_src = false;
  // A round's walks and selection do not touch what the next round's push works on (residue, reserve, frontier
  // lists, parked flags: the plan has read the residues and the estimate is a copy of the reserve by then), and the
  // next threshold is known beforehand (:178).  So while this round's walk kernel - bound by its longest walk, with
  // most of the chip idle - and selection run on the compute stream, the next round's push runs on a
  // second stream, with counters of its own that only join the query's when the round turns out to be needed.  The
  // one push that was not (after the last round) costs no time: it ends before that round's walks do.
  const char* spec_env = hook_env("PPRHIP_TOPK_AHEAD");
  const bool spec_on = !(spec_env && spec_env[0] == '0') && ensure_spec(g) == PPRHIP_OK;
  bool pushed_ahead = false;        // this round's push, residue sum and walk plan have already run (second stream)
  bool ahead_discarded = false;     // the last push ahead was not needed
  // A push queued ahead on the second stream works on this handle's residues, reserve and lists: whatever way this
  // function is left - an error return from any call below included - nothing may follow on the compute stream (the
  // next query's reset first of all) before that push has ended.  Joined: the compute stream waits for spec_ev[1]
  // (the round is taken, or the unused push is waited for at the end); otherwise the guard drains the second stream.
  struct SpecJoin {
    pprhip_graph* g;
    bool pending = false;
    ~SpecJoin() {
      if (pending && g->spec_stream) (void)hipStreamSynchronize(g->spec_stream);
    }
  } spec_join{g};
  unsigned long long dead_before_ahead = 0;
  int nsel_round = 0;
  double kth_prev = -1.0;  // the k-th estimate of the round before (none yet)
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
    if (!spec_on) (void)hipEventRecord(g->ev[1], g->stream);
    if (pushed_ahead) {
      PPRHIP_CHECK_HIP(hipStreamWaitEvent(g->stream, g->spec_ev[1], 0));
      spec_join.pending = false;
    } else {
      PPRHIP_TRY(topk_round_impl(g, min_rmax, rmax_local, st, kSumLaunch));  // :137; the residue sum stays on the device
      // :148-151: the plan derives rsum and the walk budget from the sum on the device; :143 the estimate := copy of
      // the push reserve (walk increments of earlier rounds are dropped), taken in the plan's pass
      PPRHIP_TRY(launch_walk_plan(g, 1, alpha, 0.0, 0, g->est, omega_local, g->reserve, g->est));
    }
    if (!spec_on) (void)hipEventRecord(g->ev[2], g->stream);
    // a plan that ran ahead could not touch the estimate (the round before was still reading it): :143 here
    if (pushed_ahead) PPRHIP_CHECK_HIP(hipMemcpyAsync(g->est, g->reserve, nd, hipMemcpyDeviceToDevice, g->stream));
    pushed_ahead = false;
    if (spec_on) PPRHIP_CHECK_HIP(hipEventRecord(g->spec_ev[0], g->stream));  // residues and reserve have been read
    // :155-168: the walk kernel reads the plan's counts on the device: no host round trip between push and selection
    static const uint32_t topk_waves = [] {  // PPRHIP_TOPK_WALK_WAVES: measurement switch
      const char* e = hook_env("PPRHIP_TOPK_WALK_WAVES");
      return e && atoi(e) > 0 ? (uint32_t)atoi(e) : 8u;
    }();
    g->walk_waves = spec_on ? topk_waves : 0u;  // (the next round's push runs beside these walks: leave it room)
    const int wrc = launch_walk_run(g, 1, alpha, seed, round, g->est);
    g->walk_waves = 0;
    PPRHIP_TRY(wrc);
    if (!spec_on) (void)hipEventRecord(g->ev[3], g->stream);
    round++;
    double kth = 0.0;
    bool have = false;
    unsigned long long sel_seq = 0;
    PPRHIP_TRY(select_launch(g, g->est, conf->k, &sel_seq, true));  // :173; the round's residue sum comes back with it
    // ---- the next round's push, residue sum and walk plan, ahead of the decision whether there is a next round
    const double delta_next = std::max(min_delta, delta_local / 4.0);  // :178
    pprhip_stats_t st_ahead;
    std::memset(&st_ahead, 0, sizeof st_ahead);
    bool ahead = false;
    // Not when this round is expected to be the last: at min_delta the loop ends whatever the selection says
    // (:175-176), and the k-th estimate hardly moves from round to round, so a round whose threshold the last k-th
    // value already meets is (almost always) final - its push ahead would be the largest of the query, and unused.
    const bool likely_final = delta_local <= min_delta || (kth_prev >= 0.0 && kth_prev >= (1 + epsilon) * delta_local);
    if (spec_on && !likely_final) {
      double rmax_next = epsilon * std::sqrt(delta_next / 3.0 / (double)conf->m / std::log(2.0 / conf->pfail));
      const double omega_next = (epsilon + 2.0) * std::log(2.0 / conf->pfail) / epsilon / epsilon / delta_next;
      rmax_next *= std::sqrt((double)conf->m * rmax_next) * 3.0;
      SpecContext ctx(g);  // g->stream, the mail and the calling thread's timer are the second stream's until it ends
      PPRHIP_CHECK_HIP(hipStreamWaitEvent(g->stream, g->spec_ev[0], 0));
      spec_join.pending = true;  // (from the first launch on the second stream on)
      PPRHIP_TRY(fetch_small(g, &g->ctr->dead_pops, &dead_before_ahead, sizeof dead_before_ahead));
      PPRHIP_TRY(topk_round_impl(g, min_rmax, rmax_next, st_ahead, kSumLaunch));
      PPRHIP_TRY(launch_walk_plan(g, 1, alpha, 0.0, 0, g->est, omega_next));
      PPRHIP_CHECK_HIP(hipEventRecord(g->spec_ev[1], g->stream));
      ahead = true;
    }
    PPRHIP_TRY(select_finish(g, sel_seq, g->est, conf->k, ids_out, vals_out, cap, &nsel_round, &kth, &have, st));
    g->topk_rsum = g->sel_plan_sum;  // (the sum this round's plan was derived from, in the selection's header)
    rsum_local = g->topk_rsum;       // :142
    if (!have) kth = 0.0;                                                                        // :174
    if (!spec_on) {
      (void)hipEventRecord(g->ev[4], g->stream);
      PPRHIP_CHECK_HIP(hipStreamSynchronize(g->stream));
      push_ms += CallTimer::ms(g->ev[1], g->ev[2]);
      mc_ms += CallTimer::ms(g->ev[2], g->ev[3]);
      sel_ms += CallTimer::ms(g->ev[3], g->ev[4]);
    }
    st.kth_value = kth;
    kth_prev = kth;
    if (kth >= (1 + epsilon) * delta_local || delta_local <= min_delta) {  // :175-176
      ahead_discarded = ahead;
      break;
    }
    if (ahead) {  // the push ahead was this round's: its counters join the query's
      st.pops += st_ahead.pops;
      st.edge_pushes += st_ahead.edge_pushes;
      st.enqueues += st_ahead.enqueues;
      st.dense_nodes += st_ahead.dense_nodes;
      st.dense_edges += st_ahead.dense_edges;
      st.levels += st_ahead.levels;
      st.dense_levels += st_ahead.dense_levels;
      st.sweep_min_bytes += st_ahead.sweep_min_bytes;
      st.push_bytes += st_ahead.push_bytes;
      pushed_ahead = true;
    }
    delta_local = delta_next;
  }
  if (round == 0 && !dead_src) {  // delta below min_delta from the start: nothing ran
    PPRHIP_CHECK_HIP(hipMemsetAsync(g->est, 0, nd, g->stream));
  }
  g->result_in_est = true;
  if (ahead_discarded) {  // before the next query clears
    PPRHIP_CHECK_HIP(hipStreamWaitEvent(g->stream, g->spec_ev[1], 0));
    spec_join.pending = false;
  }
  PPRHIP_TRY(read_dead_pops(g, st));
  if (ahead_discarded && st.dead_end_pops >= dead_before_ahead) {  // the unused push's dead-end pops are not the query's
    st.push_bytes -= 16ull * (st.dead_end_pops - dead_before_ahead);
    st.dead_end_pops = dead_before_ahead;
  }
  int nsel = nsel_round;
  (void)hipEventRecord(g->ev[3], g->stream);
  if (round == 0 || dead_src) {  // no round selected anything yet (the last round's selection is the result otherwise)
    bool have = false;
    double kth = 0.0;
    PPRHIP_TRY(select_topk(g, g->est, conf->k, ids_out, vals_out, cap, &nsel, &kth, &have, st));
  }
  (void)hipEventRecord(g->ev[4], g->stream);
  tm.finish(st);
  if (spec_on) {  // phases of different rounds run side by side: the per-class kernel times stand for the phases
    push_ms = st.class_ms[PPRHIP_KERNEL_SPARSE_PUSH] + st.class_ms[PPRHIP_KERNEL_DENSE_PULL];
    mc_ms = st.class_ms[PPRHIP_KERNEL_WALK];
    sel_ms = st.class_ms[PPRHIP_KERNEL_QUERY_SETUP];
  } else {
    sel_ms += CallTimer::ms(g->ev[3], g->ev[4]);
  }
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
  PPRHIP_TRY(reset_query_state(g, false, src));
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
    ktimer().begin(PPRHIP_KERNEL_WALK, 0);
    PPRHIP_TRY(launch_mc_pure(g, src, nw, conf->alpha, seed, 1.0 / omega, g->reserve));
    ktimer().end();
    PPRHIP_TRY(read_dead_pops(g, st));  // steps, walks (= nw), sources (= 1) as the device counted them
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
  PPRHIP_TRY(reset_query_state(g, false, target));
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
  PPRHIP_TRY(reset_query_state(g, false, src));
  CallTimer tm(g);
  if (iters > 0) {
    // iteration 1 (Power_Method.java:59-96 with residue = {s: 1})
    PPRHIP_TRY(ensure_panel_part(g));
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
      st.sweep_min_bytes += dense_level_min_bytes(g);
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

#ifdef PPRHIP_TEST_HOOKS
// Test / tuning hook (libpprhip_hooks.so only): the edge kernel of block `block` of `n_blocks` Gauss-Seidel blocks of a
// batched forward sweep (n_blocks <= 1: the whole sweep), `reps` launches back to back; average microseconds per launch.
int pprhip_hook_time_sweep_edges(pprhip_graph_t* g, int block, int n_blocks, int reps, double* us_out) {
  if (!g || !us_out || reps <= 0 || g->parent) {
    set_error("pprhip_hook_time_sweep_edges: bad arguments");
    return PPRHIP_ERR_INVALID;
  }
  PPRHIP_CHECK_HIP(hipSetDevice(g->device));
  PPRHIP_TRY(ensure_batch(g));
  GsBlock B{0u, g->n_nz, 0ull, (unsigned long long)g->m};
  if (n_blocks > 1) {
    pprhip_tuning_t keep = g->slots[0]->tun;
    g->slots[0]->tun.gs_blocks = n_blocks;
    int nb = 1;
    const GsBlock* blocks = gs_blocks_of(g->slots[0], &nb);
    g->slots[0]->tun = keep;
    if (!blocks || block < 0 || block >= nb) {
      set_error("pprhip_hook_time_sweep_edges: no block %d of %d", block, n_blocks);
      return PPRHIP_ERR_INVALID;
    }
    B = blocks[block];
  }
  hipEvent_t e0, e1;
  PPRHIP_CHECK_HIP(hipEventCreate(&e0));
  PPRHIP_CHECK_HIP(hipEventCreate(&e1));
  PPRHIP_TRY(launch_sweep_edges_only(g, B));  // warm-up
  PPRHIP_CHECK_HIP(hipEventRecord(e0, g->stream));
  for (int i = 0; i < reps; ++i) PPRHIP_TRY(launch_sweep_edges_only(g, B));
  PPRHIP_CHECK_HIP(hipEventRecord(e1, g->stream));
  PPRHIP_CHECK_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  PPRHIP_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *us_out = (double)ms * 1e3 / reps;
  // (the row-major kernel adds into acc8 with atomics where rows cross chunks: start the next sweep clean)
  PPRHIP_CHECK_HIP(hipMemsetAsync(g->acc8, 0, sizeof(double) * ((size_t)g->n + 1) * kBatch, g->stream));
  PPRHIP_CHECK_HIP(hipStreamSynchronize(g->stream));
  return PPRHIP_OK;
}
#endif

}  // extern "C"

// (All-Pair's whole-vector searches, allpair.cpp)
int pprhip::detail::backward_search_whole(pprhip_graph_t* g, int32_t target, double alpha, double rmax, pprhip_stats_t& st) {
  return backward_push_impl(g, target, alpha, rmax, st);
}
