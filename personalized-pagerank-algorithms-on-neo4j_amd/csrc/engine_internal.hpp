// engine_internal.hpp — declarations shared by the engine's translation units (engine.cpp: graph
// lifecycle, level loop, single-query entry points; fora.cpp: resumable FORA / top-k / backward runs
// and the batched entry points; allpair.cpp: All-Pair-Backward-Search and the inverted index).
#pragma once

#include <sys/mman.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <new>
#include <thread>
#include <deque>
#include <string>
#include <vector>

#include "engine.hpp"

// Device-resident result vectors of a batched call (pprhip_results_create): slot i = query i, internal vertex order.
struct pprhip_results {
  pprhip_graph* g = nullptr;
  int device = 0;  // (kept here: the store may be destroyed after its graph)
  int capacity = 0, count = 0;
  double* buf = nullptr;  // capacity x n
};

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

namespace pprhip {
namespace detail {

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
  // Gauss-Seidel sweeps: the state the next dense sweep runs in (set when the level is prepared / yielded), and
  // whether the contribution array is "dirty" (its contributions have reached the later blocks only, so another
  // dense sweep - in place or flushing - has to follow whatever the frontier looks like)
  int gs_state = kGsJacobi;
  bool gs_dirty = false;
  // Batch driver with the slots on a stream of their own (fora.cpp: SlotDriver).  defer_compact: run_levels returns
  // kYieldDefer as soon as the way back to list form is queued (the compaction runs on the parent's stream, before the
  // next sweep; the sparse levels behind it are launched when the driver calls again, beside that sweep).
  // compacted: that compaction has been queued, the list form is (about to be) there.
  bool defer_compact = false;
  bool compacted = false;
  unsigned long long compact_seq = 0;  // mailbox sequence the compaction's end is published under (0: stream order)
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

constexpr int kYield = 1;  // run_levels: the next level is dense and the caller runs it (batched sweeps)
constexpr int kYieldColumn = 4;  // run_levels: the level is dense and no column of c8 is free (workspace pool)
constexpr int kYieldDefer = 3;  // run_levels: the compaction is queued, the sparse levels behind it wait for the next call (LevelCtx)
constexpr int kYieldWalk = 2;  // fora_step: the walk phase runs on the handle's side stream; call again when it has ended

// kernel-class timer of the calling thread: its own, or the slot's while it works for a batch
extern thread_local KernelTimer* g_timer_cur;
inline KernelTimer& ktimer() { return *g_timer_cur; }

// Brackets a group of a query's short bookkeeping kernels (class PPRHIP_KERNEL_QUERY_SETUP) for the calling thread's
// timer - when that timer is watching this handle's stream (workspaces are also reset outside any timed call).
// A slot's kernels that touch the shared contribution array c8 (dense prepare, dense seeding, the compaction back to
// list form) while the slot runs on a stream of its own beside the sweeps (c8_via_parent): they go to the parent's
// stream, behind everything the slot has queued so far, and so take their place between two sweeps in stream order.
// back: the slot's stream goes on behind them (the compaction's list is read by the sparse levels); a prepared dense
// level needs no way back - the slot's next work comes after the sweep it waits for has been collected by the host.
struct C8Scope {
  pprhip_graph* g;
  hipStream_t own = nullptr;
  bool on = false, back = false;
  int rc = PPRHIP_OK;
  C8Scope(pprhip_graph* g_, bool back_);
  ~C8Scope() { (void)leave(); }
  int leave();
};

// A slot of the sequential batch driver gives the driver a look at the sweep in flight (pprhip_graph::idle_hook) -
// called between the launches of the slot's longer sequences, which keep the host busy for 100 us and more.
// The hook runs a whole driver turn (collect the sweep, other workspaces' deferred steps, the next sweep) on the
// caller's thread, so it must never be entered while a slot has its stream swapped for the sweeps' (C8Scope): what the
// turn queues for that slot would land on the wrong stream.  No caller does; the counter makes that an invariant the
// code checks instead of one it relies on (SlotDriver::on_idle also refuses to nest, and keeps the sweeps' own timer).
inline void poll_idle(pprhip_graph* g) {
  pprhip_graph* const H = g->parent;
  if (H && H->idle_hook && H->in_c8 == 0) H->idle_hook(H->idle_arg);
}

struct SetupScope {
  KernelTimer* t;
  explicit SetupScope(pprhip_graph* g);
  ~SetupScope() {
    if (t) t->end();
  }
};

// ---- large host arrays that threads fill in full (index arrays, the lift's edge arrays)
inline void* big_alloc(size_t bytes) {
  constexpr size_t kHuge = 2u << 20;
  if (bytes < 8 * kHuge) {
    void* p = malloc(bytes ? bytes : 1);
    if (!p) throw std::bad_alloc();
    return p;
  }
  void* p = nullptr;
  if (posix_memalign(&p, kHuge, (bytes + kHuge - 1) / kHuge * kHuge) != 0 || !p) throw std::bad_alloc();
  (void)madvise(p, (bytes + kHuge - 1) / kHuge * kHuge, MADV_HUGEPAGE);
  return p;
}

// vectors whose resize() leaves new elements uninitialised: the index arrays are hundreds of megabytes that the
// finalisation's threads fill in full (a value-initialising resize is a single-threaded pass over fresh pages)
template <class T>
struct NoInitAlloc {
  using value_type = T;
  NoInitAlloc() = default;
  template <class U>
  NoInitAlloc(const NoInitAlloc<U>&) {}
  // large arrays on 2-MB boundaries with transparent huge pages asked for: the finalisation's threads touch every page
  // of hundreds of megabytes for the first time, and a 4-KB fault each is a fifth of the k rule's time
  T* allocate(size_t n) { return static_cast<T*>(big_alloc(n * sizeof(T))); }
  void deallocate(T* p, size_t) { free(p); }
  template <class U, class... A>
  void construct(U* p, A&&... a) {
    if constexpr (sizeof...(A) == 0) ::new ((void*)p) U;  // default-init: nothing for arithmetic types
    else ::new ((void*)p) U(std::forward<A>(a)...);
  }
  template <class U>
  bool operator==(const NoInitAlloc<U>&) const { return true; }
  template <class U>
  bool operator!=(const NoInitAlloc<U>&) const { return false; }
};
template <class T>
using RawVec = std::vector<T, NoInitAlloc<T>>;


// Row-panel copy of the in-CSR for the single-query forward sweep (round 6).  Why: the sweep gathers one 8-byte
// contribution per in-edge, every gather asks the L1 for a 128-byte line, and a CU keeps ~256 lines in flight
// (TCP_PENDING_STALL_CYCLES: 0.69 of k_dense_edges<true, true>'s cycles; profiles/r06_ell_sweep_study.txt for the model):
// the row-major and the sliced copy ask for 18 M lines per half sweep of R-MAT 22's 33.5 M edges.  Here the rows are cut
// into PANELS of kPanelRows consecutive ordinals whose sums fit half a CU's LDS (64 KB: two workgroups share a CU), and a
// panel's in-edges are sorted by (source, row): neighbouring lanes gather neighbouring sources, so the sources of a line
// are served by one request, and a workgroup walks its part of the contribution array front to back.  R-MAT 22:
// 4.6 M requests to L2 per half sweep where the sliced copy makes 18.1 M (counters: profiles/r06_pmc_single_*.txt).
// A panel of more than kItemEdges edges is cut into S parts of equal edge counts (ITEMS); part k sums into LDS of its own
// and leaves part[base + k * rows + local row]; k_dense_apply adds a row's S values.  Every item's edges are padded to
// whole turns of kPanelStep with (source 0, row 0xffff): a row ordinal no panel has, skipped by the kernel.
struct HostPanelLayout {
  uint32_t n_nz = 0, n_panels = 0, n_items = 0;
  uint64_t n_edges = 0;                      // with padding
  uint64_t n_part = 0;                       // doubles of the parts' sums
  RawVec<int32_t> src;                       // [n_edges] sources, item-major, inside an item sorted by (source, row)
  RawVec<uint16_t> rloc;                     // [n_edges] row ordinal - first ordinal of the panel; 0xffff: padding
  std::vector<PanelItem> items;              // [n_items], panel-major
  std::vector<PanelDesc> panels;             // [n_panels]
  std::vector<uint32_t> panel_item0;         // [n_panels + 1]
};
int build_panel_layout(uint32_t n, uint64_t m, const uint32_t* in_rp, const int32_t* in_ci, const int32_t* nz_rows,
                       uint32_t n_nz, unsigned threads, HostPanelLayout& L);

// The host half of the graph lift (lift.cpp): every array of the internal layout in host memory, ready to upload.
struct HostLift {
  bool relabeled = true;
  std::vector<int32_t> new2old, old2new;
  std::vector<uint32_t> out_rp, in_rp;  // internal order
  RawVec<int32_t> out_ci, in_ci;        // internal order, padded to whole chunks plus one (the padding zeroed)
  std::vector<unsigned long long> ext;  // out-row extent per node: first edge | degree << 32
  std::vector<int32_t> nz_rows, zin_rows;
  uint32_t n_chunks = 0, n_src_live = 0;
  std::vector<uint8_t> flags;
  std::vector<uint32_t> chunk_starts;
  std::vector<unsigned long long> cross;
  // sliced copy of the in-CSR (S == 0: none)
  int S = 0;
  uint32_t width = 0, n_seg = 0;
  std::vector<uint64_t> edge_base, seg_base;
  RawVec<int32_t> sl_ci;
  std::vector<uint8_t> sl_flags;
  std::vector<uint32_t> sl_chunk_starts, seg_row, seg_off;
  // row-panel copy of the in-CSR (n_items == 0: none; graphs that have it have no sliced copy)
  HostPanelLayout pn;
};
// threads: 0 = what the process may use (host_threads)
int lift_host(uint32_t n, uint64_t m, const uint32_t* out_rp, const int32_t* out_ci, const uint32_t* in_rp,
              const int32_t* in_ci, unsigned threads, HostLift& H);
unsigned host_threads();  // CPU affinity and cgroup quota of the process, at most 64 (PPRHIP_HOST_THREADS overrides)

void stream_detach(void* stream_obj);  // fora.cpp: ends a query stream's driver before its graph goes
int alloc_dev(void** p, size_t bytes);
double level_cost(const pprhip_graph* g, uint64_t nf, uint64_t ef, bool* dense);
uint64_t dense_level_bytes(const pprhip_graph* g);
double dense_sweep_cost(const pprhip_graph* g);
uint64_t dense_level_min_bytes(const pprhip_graph* g);  // compulsory bytes of one single-query sweep
uint64_t batch_sweep_min_bytes(const pprhip_graph* P, bool backward, int n_active);  // ... of one batched sweep
void finish_dense(LevelCtx& L, pprhip_stats_t& st, uint64_t level_bytes, uint64_t min_bytes, uint32_t nf_next,
                  uint64_t ef_next);
int run_levels(pprhip_graph* g, const PushArgs& a, LevelCtx& L, pprhip_stats_t& st, double* model_cost,
               bool yield_dense = false, RoundCut* cut = nullptr);
int reset_query_state(pprhip_graph* g, bool clear_flags, int32_t node = -1);  // node: the query's source / target (internal id)
int ensure_batch(pprhip_graph* P);
void free_batch(pprhip_graph* P);
int ensure_bwd_layout(pprhip_graph* P);
int ensure_panel_part(pprhip_graph* g);  // the buffer of the panel sweep's partial sums (first forward dense level)
// blocks of the forward Gauss-Seidel sweep for the handle's tuning (nullptr / 1 block when switched off)
const GsBlock* gs_blocks_of(pprhip_graph* g, int* n_blocks);
unsigned long long gs_thresh_of(const pprhip_graph* g);
int seed_single(pprhip_graph* g, LevelCtx& L, int32_t node, uint32_t degree);
int seed_scan(pprhip_graph* g, const PushArgs& a, int kind, LevelCtx& L);
int ensure_workspaces(pprhip_graph* P, int count);  // more workspaces than columns of c8 (fora.cpp: SlotDriver)
// a stream that runs beside g->stream - and beside `also`, when given - (self-tested)
int make_side_stream(pprhip_graph* g, hipStream_t* out, hipStream_t also = nullptr);
int fetch_small(pprhip_graph* g, const void* dev, void* host, size_t bytes);  // a few words, without a copy command
int fetch_begin(pprhip_graph* g, const void* dev, size_t bytes, unsigned long long* seq_out);  // ... in two halves
int fetch_end(pprhip_graph* g, unsigned long long seq, const void* dev, void* host, size_t bytes);
int select_launch(pprhip_graph* g, const double* x, int k, unsigned long long* seq_out, bool with_plan_sum = false);
int select_finish(pprhip_graph* g, unsigned long long seq, const double* x, int k, int32_t* ids_out, double* vals_out,
                  int cap, int* n_out, double* kth_out, bool* have_kth, pprhip_stats_t& st);
int device_sum(pprhip_graph* g, const double* x, double* out, uint32_t count = 0);  // count 0: the query's scan bound
int read_dead_pops(pprhip_graph* g, pprhip_stats_t& st);
int run_walk_phase(pprhip_graph* g, int variant, double alpha, double rsum, long long nrw, uint64_t seed, uint32_t stream,
                   double* target, pprhip_stats_t& st, double omega_dev = 0.0);
int launch_walk_plan(pprhip_graph* g, int variant, double alpha, double rsum, long long nrw, double* target, double omega_dev,
                     const double* copy_src = nullptr, double* copy_dst = nullptr);
int launch_walk_run(pprhip_graph* g, int variant, double alpha, uint64_t seed, uint32_t stream, double* target);
int copy_out(pprhip_graph* g, const double* dev, double* host);
int check_graph(const pprhip_graph* g, const char* fn);
int check_node(const pprhip_graph* g, int32_t v, const char* fn);
const pprhip_graph* host_of(const pprhip_graph* g);
uint32_t hdeg_out(const pprhip_graph* g, int32_t v);
uint32_t hdeg_in(const pprhip_graph* g, int32_t v);
int select_topk(pprhip_graph* g, const double* x, int k, int32_t* ids_out, double* vals_out, int cap, int* n_out,
                double* kth_out, bool* have_kth, pprhip_stats_t& st, bool with_plan_sum = false);

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

// a batch of queries for the slot engine (fora.cpp)
// Delivery of the queries' vectors to the caller's (pageable) host memory while the batch keeps running: a finished
// query's vector is permuted to the caller's ids into one of kRing device staging buffers on the query's own stream,
// copied to a pinned host buffer on a copy stream, and moved from there to its destination by a copier thread.  The
// compute stream never waits for PCIe or for the host copy; a query that finishes while all staging buffers are in
// flight waits for one.
struct FetchPipe {
  static constexpr int kRing = 16;   // one per slot: the queries of a batch tend to finish in bursts (they leave the shared sweeps together)
  static constexpr int kCopiers = 6;
  pprhip_graph* P = nullptr;
  size_t n = 0;
  hipStream_t cs = nullptr;
  double* dev[kRing] = {};
  double* pin[kRing] = {};
  hipEvent_t ready[kRing] = {};
  hipEvent_t done[kRing] = {};
  struct Item { int e; double* dst; };
  struct Arrival { FetchPipe* pipe; Item item; };
  static void on_copied(void* arrival);  // host callback of the copy stream
  int pending = 0;                       // copies queued on the copy stream whose callback has not run yet
  std::mutex mu;     // free_q, work, pending, closing, err: bookkeeping only - never held across a HIP call, because
                     // the copy stream's host callback (on_copied, on a runtime thread) takes it too
  std::mutex cs_mu;  // orders the slots' threads on the shared copy stream (wait - copy - callback stay together)
  std::condition_variable cv;
  std::deque<int> free_q;
  std::deque<Item> work;
  bool closing = false;
  int err = 0;
  std::thread copiers[kCopiers];
  int ensure(pprhip_graph* parent);  // buffers, stream and events: allocated once per handle, kept between calls
  void start();                      // a call's copier threads
  int submit(pprhip_graph* S, const double* dev_vec, double* dst);  // dev_vec in internal order, on S->stream
  int finish();                      // drains and joins; first error of any stage
  void destroy();
  void copier();
};

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
  pprhip_results* keep = nullptr;          // kind 0: device-resident store of the queries' vectors
  int keep_first = 0;                      // ... query i goes to slot keep_first + i of it
  FetchPipe* pipe = nullptr;               // reserve_out given: asynchronous delivery (batch_run opens / closes it)
  pprhip_stats_t sum;
  std::mutex sum_mu;
  std::atomic<int> next_query{0};
};

int batch_run(pprhip_graph_t* g, BatchJob& J, pprhip_stats_t* stats_sum);

// where the entries of All-Pair's backward searches go: to the host at once, or into an HBM record store that the
// sharded call partitions by owner of the source and exchanges over RCCL before anything crosses PCIe
struct TripleSink {
  virtual ~TripleSink() = default;
  virtual int take_device(pprhip_graph* g, const TripleRec* d_rec, unsigned long long count) = 0;
  virtual int take_host(pprhip_graph* g, std::vector<Triple>& more) = 0;
};
struct HostTripleSink : TripleSink {
  std::vector<Triple> tr;
  int take_device(pprhip_graph* g, const TripleRec* d_rec, unsigned long long count) override;
  int take_host(pprhip_graph* g, std::vector<Triple>& more) override;
};
struct DeviceTripleSink : TripleSink {
  TripleRec* rec = nullptr;
  unsigned long long count = 0, cap = 0;
  ~DeviceTripleSink() override;
  int reserve(pprhip_graph* g, unsigned long long extra);
  int take_device(pprhip_graph* g, const TripleRec* d_rec, unsigned long long count) override;
  int take_host(pprhip_graph* g, std::vector<Triple>& more) override;
};
int all_pair_collect(pprhip_graph_t* g, double alpha, double threshold, uint32_t t_begin, uint32_t t_end,
                     TripleSink& sink, pprhip_stats_t& st);
// Backward_Search.backward_search_whole_graph on the handle's own vectors (internal id); reserve / residue stay in HBM
int backward_search_whole(pprhip_graph_t* g, int32_t target_internal, double alpha, double rmax, pprhip_stats_t& st);
int ensure_bwd_layout(pprhip_graph* P);
int index_from_triples(uint32_t n, std::vector<Triple>& tr, int k, pprhip_index_t** out);
// the same from records in HBM: row order and k rule on the device (kernels_sort.hip), the index arrays downloaded as
// they are; sources must lie in [v_lo, v_hi)
int index_from_device(pprhip_graph* g, const TripleRec* rec, unsigned long long count, int k, uint32_t v_lo, uint32_t v_hi,
                      pprhip_index_t** out);
int index_concat(const std::vector<pprhip_index_t*>& parts, pprhip_index_t** out);

}  // namespace detail
}  // namespace pprhip
