// engine.hpp — device-resident graph handle and the kernel launchers the drivers call.
#pragma once

#include <atomic>
#include <cstdlib>

#include "common.hpp"

namespace pprhip {

// Device-side counter block (one per graph handle), mirrored into pinned host memory.
constexpr int kMaxBatch = 8;  // sparse levels launched per host round trip

struct DevCounters {
  unsigned long long hist[16];    // sparse batch: hist[i] = frontier of level i (entries << 36 | edges)
  unsigned long long packed[2];   // dense level / seeding output: entries << 36 | edge total
  double dead[2];                 // dead-end mass waiting to land on the source
  // read once per query, in one copy (read_dead_pops): what the push and the walk phases counted since the reset
  unsigned long long dead_pops;      // pushed nodes with out-degree 0
  unsigned long long walk_steps;     // edges followed by walks
  unsigned long long walks_total;    // walks run
  unsigned long long sources_total;  // residue entries that started walks
  unsigned long long walk_loads;     // load instructions the walk kernel's waves issued ...
  unsigned long long walk_lanes;     // ... and the lanes they carried (64 per load = full waves)
  double sum_out;                    // reduction result (the walk plan reads it on the device)
  // walk plan of a phase: sources << 36 | walks, counted by the plan kernel, read by the walk kernel on the device.
  // Three cells used in turn: the plan of phase p counts into cell p % 3 and clears cell (p + 1) % 3, which the walks
  // of phase p - 2 were the last to read (the plan of phase p + 1 may run while the walks of phase p still start).
  unsigned long long mc_plan[3];
  // the residue sum a top-k round's plan was derived from, kept per phase like the cells above: the selection that
  // ends the round hands it to the host in its header (no copy command of its own)
  double plan_sum[3];
  unsigned long long pad[1];
  unsigned long long dhist[8];    // dense batch: dhist[i] = frontier that dense level i of the batch starts from
  int dstate[8];                  // dense batch: sweep state of level i (kGsNone: the level does not run)
};
constexpr int kDenseBatch = 6;  // dense levels of a single query launched per host round trip

// State of a dense sweep ("Gauss-Seidel sweeps").  Rows are cut into `gs_blocks` blocks of equal in-edge
// count, processed one after another; what a row leaves in the *current* contribution array when it is applied decides
// what the later blocks of the same sweep read:
//   kGsJacobi   nothing: every row reads the contributions of the level before (one block is enough);
//   kGsEntry    old + new: first sweep of a long dense phase; later blocks get both, earlier ones have read the old;
//   kGsInPlace  new: the array holds contributions that still have to reach the blocks up to their own;
//   kGsFlush    zero: pending contributions are delivered, the new ones reach nobody yet - after it (as after a Jacobi
//               sweep) every contribution is wholly undelivered again, the only state a sparse level can start from.
enum GsState : int { kGsNone = 0, kGsJacobi = 1, kGsEntry = 2, kGsInPlace = 3, kGsFlush = 4 };

// state of the sweep that follows a sweep of state `prev` which left a frontier of nf nodes / ef edges
__host__ __device__ inline int gs_next_state(int prev, unsigned long long nf, unsigned long long ef,
                                             unsigned long long dense_thresh, unsigned long long gs_thresh) {
  if (nf == 0) return kGsNone;
  const unsigned long long tot = nf + ef;
  if (prev == kGsEntry || prev == kGsInPlace) return tot >= gs_thresh ? kGsInPlace : kGsFlush;  // must be finished
  if (tot < dense_thresh) return kGsNone;                                                      // sparse levels follow
  return tot >= gs_thresh ? kGsEntry : kGsJacobi;
}

// one block of a Gauss-Seidel sweep: row ordinals [j_lo, j_hi), their in-edges [e_lo, e_hi)
struct GsBlock {
  uint32_t j_lo, j_hi;
  unsigned long long e_lo, e_hi;
};

// Edge ranges one launch of the single-query edge kernel walks, as 512-edge chunks in a virtual order: window w holds
// the chunks c_lo[w] + (vc - c_pre[w]) for vc in [c_pre[w], c_pre[w + 1]); edges of a boundary chunk outside
// [e_lo[w], e_hi[w]) count as zero (a neighbouring window or block sums them).
constexpr int kMaxWindows = 16;
struct EdgeWindows {
  uint32_t n;
  uint32_t c_pre[kMaxWindows + 1];
  uint32_t c_lo[kMaxWindows];
  unsigned long long e_lo[kMaxWindows], e_hi[kMaxWindows];
};

// Sliced copy of the in-CSR for the single-query sweep.  A gather of one 8-byte contribution moves a whole 128-byte
// line, and lines that leave L2 come at a fifth of the rate of lines that hit it (tools/micro/gather_rate.hip:
// 55 G/s from a 34 MB table and beyond, 250 G/s from a 4 MB one).  So the in-edges are kept a second time sorted
// by (slice of the source id, row):
// the sweep walks slice after slice, and while a slice is being walked the contributions it gathers - `width`
// consecutive ids, a few MB - stay in every XCD's L2.  A row's edges inside one slice form a *segment*; segment sums
// are added to the row's accumulator (seg_row: segment -> row ordinal), so a row receives one add per slice it has
// sources in.  Built at graph lift when the sources span more than one slice.
struct SlicedLayout {
  int S = 0;                         // slices
  uint32_t width = 0;                // source ids per slice
  int32_t* ci = nullptr;             // [m padded] source ids, slice-major
  uint8_t* flags = nullptr;          // bit e: edge e is the first of its segment
  uint32_t* chunk_starts = nullptr;  // segments that start before each 512-edge chunk
  uint32_t* seg_row = nullptr;       // [segments] row ordinal (index into nz_rows / acc_nz)
  uint32_t n_seg = 0;
  std::vector<uint64_t> edge_base, seg_base;   // [S + 1] first edge / segment of every slice
  std::vector<uint32_t> h_seg_row, h_seg_off;  // host: row ordinal and first edge of every segment
  std::vector<EdgeWindows> plan;               // windows of every Gauss-Seidel block for plan_B blocks
  int plan_B = -1;
};

constexpr int kPanelQueues = 64;  // = the most Gauss-Seidel blocks a sweep can have
// The row-panel copy of the in-CSR on the device (engine_internal.hpp: HostPanelLayout; single-query forward sweep).
struct PanelLayout {
  int32_t* src = nullptr;          // [n_edges] sources, item-major
  uint16_t* rloc = nullptr;        // [n_edges] row ordinal inside the panel (0xffff: padding)
  PanelItem* items = nullptr;      // [n_items]
  PanelDesc* panels = nullptr;     // [n_panels]
  uint32_t n_panels = 0, n_items = 0;
  uint64_t n_part = 0;             // doubles a handle's buffer of the parts' sums holds
  std::vector<uint32_t> h_panel_item0;  // host: [n_panels + 1], the Gauss-Seidel blocks' item windows
};

enum PushMode : int { kFwdWhole = 0, kFwdTopk = 1, kBackward = 2, kPower = 3 };

struct SelRec {  // one candidate of a top-k selection / one entry >= threshold of a backward search
  int32_t id, pad;
  double val;
};
constexpr int kSelHeader = 64;  // bytes before the records in pprhip_graph::sel_blob (kernels_select.hip)

struct PushArgs {
  double alpha;
  double rmax;
  double min_rmax;  // kFwdTopk only
  int32_t src;      // source (forward) or target (backward)
  int mode;
};

// Kernel-class timing with an event pool, resolved after the stream has drained.  Events are
// created on demand and reused across calls; the pool is capped: a call that records more intervals
// than kMaxEvents / 2 (All-Pair's tier 3 launches ~1e5 sparse batches) drains the stream, folds what it
// has into per-class accumulators and starts over, so neither the pool nor `recs` grows with the call.
// Class *times* are an option (pprhip_set_kernel_timing, PPRHIP_KERNEL_TIMER=1): an interval is two event records in
// the stream, and between the short kernels of a top-k round or a sparse level those cost the latency-bound paths
// 2-8 % (round 4: top-k 16 in flight 1 640-1 720 -> 1 800-1 850 queries/s without them).  Without the option the
// brackets are only counted: class_launches and class_bytes are filled, class_ms stay zero.
extern std::atomic<int> g_kernel_timing;  // engine.cpp; -1: not decided yet (the environment is read on first use)
bool kernel_timing_on();
int kernel_timing_level();  // 0: counted only; 1: every class timed; 2: the dense sweeps (classes 1 and 5) timed only
struct KernelTimer {
  static constexpr size_t kMaxEvents = 4096;
  static constexpr size_t kNoEvent = ~(size_t)0;
  bool off = false;  // records nothing (work whose time is accounted elsewhere or not at all)
  bool last_timed = false;  // the outermost open bracket recorded an event (its end() records the partner)
  int depth = 0;            // brackets open
  std::vector<hipEvent_t> ev;
  struct Rec { int cls; size_t i; uint64_t bytes; };
  std::vector<Rec> recs;
  size_t used = 0;
  hipStream_t stream = nullptr;
  double acc_ms[8] = {0};
  uint64_t acc_bytes[8] = {0};
  uint32_t acc_cnt[8] = {0};
  hipEvent_t next() {
    if (used == ev.size()) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) return nullptr;
      ev.push_back(e);
    }
    return ev[used++];
  }
  // folds the recorded intervals into the accumulators (the stream must have drained)
  void fold() {
    for (const Rec& r : recs) {
      if (r.i == kNoEvent) {  // counted, not timed
        acc_bytes[r.cls] += r.bytes;
        acc_cnt[r.cls]++;
        continue;
      }
      if (r.i + 1 >= used) continue;
      float f = 0.f;
      if (hipEventElapsedTime(&f, ev[r.i], ev[r.i + 1]) != hipSuccess) continue;
      acc_ms[r.cls] += (double)f;
      acc_bytes[r.cls] += r.bytes;
      acc_cnt[r.cls]++;
    }
    used = 0;
    recs.clear();
  }
  // room for k more intervals without a fold in between (callers that keep indices into `recs`)
  // - in counting mode as well, where begin() folds at kMaxCounted records: a fold between reserve() and the use of
  // an index would let the index name another record
  static constexpr size_t kMaxCounted = 65536;
  void reserve(size_t k) {
    if (off) return;
    if (!kernel_timing_on()) {
      if (recs.size() + k > kMaxCounted) fold();
      return;
    }
    if (used + 2 * k > kMaxEvents && hipStreamSynchronize(stream) == hipSuccess) fold();
  }
  // Brackets on one timer do not nest: a bracket's interval is the pair (begin event, next event).  One opened while
  // another is open - a driver turn taken from inside a wait that a longer bracket spans - is counted, not timed, so
  // the outer pair stays a pair (its time then includes the inner launches: they ran inside it).
  void begin(int cls, uint64_t bytes) {
    if (off) return;
    const bool nested = depth++ > 0;
    const int lvl = kernel_timing_level();
    const bool timed = !nested && (lvl == 1 || (lvl == 2 && (cls == 1 || cls == 5)));  // (PPRHIP_KERNEL_DENSE_PULL / _DENSE_PULL_BATCH)
    if (!nested) last_timed = timed;
    if (!timed) {
      if (recs.size() >= kMaxCounted) {  // (counting needs no drained stream - unless timed brackets are pending too)
        if (used && hipStreamSynchronize(stream) != hipSuccess) return;
        fold();
      }
      recs.push_back({cls, kNoEvent, bytes});
      return;
    }
    if (used + 2 > kMaxEvents) {
      if (hipStreamSynchronize(stream) != hipSuccess) return;
      fold();
    }
    hipEvent_t a = next();
    if (!a) return;
    recs.push_back({cls, used - 1, bytes});
    (void)hipEventRecord(a, stream);
  }
  void end() {
    if (off) return;
    if (depth > 0 && --depth > 0) return;  // (the end of a nested bracket)
    if (!last_timed) return;
    last_timed = false;
    hipEvent_t b = next();
    if (b) (void)hipEventRecord(b, stream);
  }
  // bytes of launches already recorded that only become known later (a walk kernel's steps)
  void add_bytes(int cls, uint64_t bytes) { acc_bytes[cls] += bytes; }
  void reset() {
    used = 0;
    depth = 0;  // (a call that failed between a begin and its end left one open)
    last_timed = false;
    recs.clear();
    for (int c = 0; c < 8; ++c) {
      acc_ms[c] = 0.0;
      acc_bytes[c] = 0;
      acc_cnt[c] = 0;
    }
  }
  // adds everything recorded since reset() to per-class totals (call after the stream has been synchronized)
  void resolve(double tot[8], uint64_t bytes[8], uint32_t cnt[8]) {
    fold();
    for (int c = 0; c < 8; ++c) {
      tot[c] += acc_ms[c];
      bytes[c] += acc_bytes[c];
      cnt[c] += acc_cnt[c];
    }
  }
  void destroy() {
    for (hipEvent_t e : ev) (void)hipEventDestroy(e);
    ev.clear();
    reset();
  }
};

struct BatchSync;  // engine.cpp: rendezvous of the batch workers at dense levels

// Contribution vector of a dense level: a plain array for a single query (stride 1), or one
// column of the interleaved c8[v][slot] array that kBatch concurrent queries share.
constexpr int kBatch = 16;
struct CView {
  double* p;
  uint32_t stride, off;
  __host__ __device__ double& at(uint32_t v) const { return p[(size_t)v * stride + off]; }
};

// Per-slot arguments of a batched dense level, read by the kernels from device memory.
struct SlotArgs {
  double* res;
  double* reserve;
  uint8_t* flags;
  uint32_t* armed;
  DevCounters* ctr;
  double alpha, rmax, min_rmax;
  int32_t src;
  int32_t active;  // the slot takes part in this sweep
  int32_t mode;
  int32_t dead_slot, out_slot;
  int32_t gs_state;  // GsState of this slot in this sweep
};

// Small results the host waits for (kernels_host.hip): mapped pinned memory a kernel writes and the host reads.
constexpr uint32_t kMailWords = 4112;  // the selection's header and 2048 records fit
struct HostMail {
  unsigned long long seq;  // written last; the host spins on it
  unsigned long long pad[7];
  unsigned long long words[kMailWords];
};

struct ClearList {  // ranges one k_clear launch zeroes
  void* p[6];
  unsigned long long bytes[6];
  int n;
};

struct WalkPlanRec {  // one residue entry of a walk phase (k_mc_plan -> k_mc_walk): 32 bytes
  unsigned long long woff;  // walks of the entries before it
  double inc;               // what each of its walks adds at its terminal
  unsigned long long ext;   // the node's out-row (out_ext: first edge | degree << 32)
  int32_t node, orig;       // internal id; original id (a word of the walk's Philox counter)
};

}  // namespace pprhip

namespace pprhip {
namespace detail {
struct FetchPipe;
}
}  // namespace pprhip

struct pprhip_graph {
  int device = 0;
  int n_cus = 256;  // compute units of the device (persistent-kernel grid sizing)
  uint32_t n = 0;
  uint64_t m = 0;
  hipStream_t stream = nullptr;
  // CSR pair in HBM: uint32 row pointers, int32 column indices
  uint32_t *out_rp = nullptr, *in_rp = nullptr;
  unsigned long long* out_ext = nullptr;  // per vertex: out row begin | out-degree << 32 (one gather instead of two)
  int32_t *out_ci = nullptr, *in_ci = nullptr;
  uint4* walk_rec = nullptr;  // per out-edge {neighbour, its first out-edge, its out-degree, 0}: one gather per walk step
  std::vector<uint32_t> h_out_rp, h_in_rp;  // host copies for degree checks on the call path
  // internal vertex order: nodes with in-edges first, then by out-degree (descending), so that the contributions the
  // dense sweep gathers most often sit next to each other; the C ABI speaks original ids
  bool relabeled = false;
  int32_t *new2old = nullptr, *old2new = nullptr;
  std::vector<int32_t> h_new2old, h_old2new;
  // dense pull-sweep layout over the in-CSR (in_ci is padded to a multiple of 512 edges)
  uint8_t* start_flags = nullptr;    // bit e: in-edge e is the first of its row
  uint32_t* chunk_starts = nullptr;  // row starts before each 512-edge chunk
  uint32_t n_chunks = 0;
  int32_t* nz_rows = nullptr;  // rows with in-degree > 0, ascending
  uint32_t n_nz = 0;
  std::vector<int32_t> h_nz_rows;         // host copy (block boundaries of the Gauss-Seidel sweeps)
  std::vector<pprhip::GsBlock> gs_plan;   // blocks of the forward sweep for gs_plan_B blocks (built on demand)
  int gs_plan_B = 0;
  double* acc_nz = nullptr;  // per non-empty row: sum of this level's contributions
  pprhip::SlicedLayout* sl = nullptr;
  pprhip::PanelLayout* pn = nullptr;   // row-panel copy of the in-CSR (shared with the batch slots), or none
  double* pn_part = nullptr;           // [pn->n_part] the items' sums of this handle's sweep (first forward dense level)
  uint32_t* pn_ctr = nullptr;          // [kPanelQueues] item queues of a level's edge launches (one per Gauss-Seidel block); k_dense_reduce zeroes them  // single-query sweep layout (owned by the lifted graph, borrowed by slots)
  // batched queries: kBatch workspaces ("slots") borrow this handle's CSR and stream; their dense
  // levels run as one sweep over the interleaved contribution array c8[v][slot]
  pprhip_graph* parent = nullptr;  // set on a slot
  int slot_index = -1;
  pprhip::BatchSync* sync = nullptr;  // set on a slot while a batched call is running
  hipStream_t own_stream = nullptr;   // slot: the stream its worker thread uses
  // Sequential batch driver (fora.cpp: SlotDriver): graph: the one stream all slots work on beside the sweeps
  // (sparse levels, seeds, sums, selections; make_side_stream); slot: its c8-touching kernels go to the parent's stream
  // (engine_internal.hpp: C8Scope) and the events that order them
  hipStream_t slot_stream = nullptr;
  bool slot_stream_tried = false;
  bool c8_via_parent = false;
  bool c8_settled = false;  // nothing of this slot is pending on its stream: C8Scope need not wait for that stream
  hipEvent_t c8_ev[2] = {nullptr, nullptr};
  hipEvent_t col_ev = nullptr;  // recorded on the slot's stream when it began to wait for its column
  // Workspace pool (SlotDriver): more workspaces than columns of c8.  graph: who holds each column (-1: nobody; an
  // index into `slots`); slot: its own index there; pooled: it has to win a free column - which becomes its
  // slot_index - before it prepares a dense level, and the driver takes the column back when it leaves the sweeps
  // (not pooled: column ws_index % kBatch is its own, as in the threaded driver)
  int col_owner[pprhip::kBatch];
  int ws_index = -1;
  bool pooled = false;
  bool has_col = false;
  // graph: called by a slot's small read-backs while they wait (fetch_end): the driver looks after the sweep in flight
  void (*idle_hook)(void*) = nullptr;
  int in_c8 = 0;  // C8Scopes open on this handle's slots (poll_idle: the hook stays out while a slot borrows the sweeps' stream)
  void* idle_arg = nullptr;
  // graph: a stream that runs beside the compute stream (make_side_stream) for the slots' walk phases while sweeps
  // go on (sequential batch driver); slot: the events around its walk phase on that stream
  uint32_t walk_waves = 0;  // waves per CU of the next walk kernels (0: the default)
  hipStream_t walk_stream = nullptr;
  bool walk_stream_tried = false;
  bool stream_open = false;  // a query stream's driver thread owns the handle (fora.cpp: pprhip_stream)
  void* stream_obj = nullptr;  // ... that stream (pprhip_graph_destroy closes a stream its owner forgot)
  hipEvent_t walk_ev[3] = {nullptr, nullptr, nullptr};
  pprhip::KernelTimer ktimer;         // slot: its worker's kernel-class timer; graph: the sweeps' timer
  std::vector<pprhip_graph*> slots;
  // All-Pair: in-edge records {source, its out-degree} (8 B per edge, built on first use), and tier 2's dense
  // workspaces of apbs_blocks workgroups (16n bytes + lists each, all-zero between searches), kept between calls
  void* in_rec = nullptr;
  char* apbs_ws = nullptr;
  void* apbs_board = nullptr;
  void* ix_stage = nullptr;  // pinned ring the index arrays are downloaded through (index_from_device)
  size_t ix_stage_bytes = 0;
  char* apbs_xl_ws = nullptr;  // a few workspaces whose lists hold every node, for the searches that outgrow the others
  uint32_t apbs_xl_blocks = 0, apbs_xl_cap_t = 0, apbs_xl_cap_f = 0;
  uint32_t apbs_blocks = 0, apbs_cap_t = 0, apbs_cap_f = 0, apbs_chunk = 0;
  pprhip::detail::FetchPipe* fetch = nullptr;  // delivery of batched queries' vectors to host memory (engine_internal.hpp)
  double* c8[2] = {nullptr, nullptr};
  int c8cur = 0;
  double* acc8 = nullptr;      // [row ordinal][kBatch] row sums
  int acc8_dir = 0;            // layout the row sums were last written in (0 forward, 1 backward)
  int32_t* zin_rows = nullptr;  // rows without in-edges
  // the same sweep layout over the out-CSR (backward search: a row pulls from its out-neighbours), built on
  // the first batched backward call
  uint8_t* start_flags_o = nullptr;
  uint32_t* chunk_starts_o = nullptr;
  int32_t *nz_rows_o = nullptr, *z_rows_o = nullptr;  // rows with / without out-edges
  uint32_t n_nz_o = 0, n_z_o = 0;
  unsigned long long* cross_bits_o = nullptr;
  unsigned long long* cross_bits = nullptr;  // per row ordinal (non-empty rows first): row spans two 512-edge chunks
  unsigned long long* prep_bits = nullptr;   // [kBatch][tiles]: rows holding a contribution after the last sweep
  uint32_t n_zin = 0;
  uint32_t n_src_live = 0;  // nodes with out-edges: the contributions a forward sweep can gather
  // Internal ids [0, n_live) are the nodes with at least one edge (the vertex order puts nodes with in-edges first, then
  // the others by out-degree: isolated nodes - 43 % of an R-MAT 22 - come last).  A query whose source / target is
  // one of them can only ever hold residue, reserve or walk terminals below n_live, so the passes over "all nodes"
  // (seeding, sums, walk plan, selection, resets) run over n_act = n_live entries; a query on an isolated node uses n.
  uint32_t n_live = 0;
  uint32_t n_act = 0;    // scan bound of the query this workspace is running (0: n)
  uint32_t n_dirty = 0;  // ... of the query before it (what the reset has to clear)
  pprhip::SlotArgs* d_slot_args = nullptr;
  pprhip::SlotArgs* h_slot_args = nullptr;  // pinned
  unsigned long long* sweep_out = nullptr;    // [kBatch] frontier counters a sweep produced
  unsigned long long* h_sweep_out = nullptr;  // pinned
  unsigned long long* blk_pack8 = nullptr;  // [kBatch][kApplyBlocks8]
  double* blk_dead8 = nullptr;
  uint32_t* blk_ndead8 = nullptr;
  // per-query state
  double *residue = nullptr, *reserve = nullptr, *est = nullptr;
  double* cdense[2] = {nullptr, nullptr};
  double* cF = nullptr;
  int32_t* F[2] = {nullptr, nullptr};
  uint32_t* eoff[2] = {nullptr, nullptr};
  uint8_t* flags = nullptr;
  // top-k rounds: one bit per node that meets the round's threshold at round start without being parked (possible
  // when the scaled rmax of Fora_Topk.java:133 falls below min_rmax); such a node joins the frontier the first time
  // it receives mass, as Forward_Push.java:226-231 enqueues it, although it does not *cross* the threshold
  uint32_t* armed = nullptr;
  // walk plan
  double sel_plan_sum = 0.0;  // the residue sum the last selection's header carried (select_launch with_plan_sum)
  uint32_t mc_phase = 0;      // walk phases planned since the workspace was reset
  uint32_t mc_last_plan = 0;  // phase of the latest plan: what the next walk kernel runs
  pprhip::WalkPlanRec* mc_plan_rec2 = nullptr;  // second record buffer (odd phases) where plans run ahead of walks
  unsigned long long walk_hint = 0;  // upper bound of the next walk phase's walks when the host knows one (grid size)
  pprhip::WalkPlanRec* mc_plan_rec = nullptr;  // n entries: the residue entries of a walk phase (k_mc_plan)
  // reductions / selection scratch
  double* partial = nullptr;       // 1024 partial sums
  uint32_t sum_np = 0;             // partial sums a launch_sum_partial has left for the walk plan that follows
  uint32_t* hist = nullptr;        // 4096-bin histogram
  unsigned long long* blk_pack = nullptr;  // per-workgroup partial counters of the dense sweep
  double* blk_dead = nullptr;
  uint32_t* blk_ndead = nullptr;
  char* sel_blob = nullptr;        // candidate list: kSelHeader bytes {count, ...} + sel_cap records (SelRec)
  uint32_t sel_cap = 0;
  pprhip::DevCounters* ctr = nullptr;    // device
  pprhip::DevCounters* h_ctr = nullptr;  // pinned host mirror
  pprhip::HostMail* mail = nullptr;      // mapped pinned memory: small read-backs without a copy command (fetch_small)
  pprhip::HostMail* mail_dev = nullptr;  // the same, as the device sees it
  unsigned long long mail_seq = 0;
  // a second stream with its own mail and timer: the next top-k round's push beside this round's walks (engine.cpp:
  // pprhip_fora_topk); created on first use
  hipStream_t spec_stream = nullptr;
  pprhip::HostMail* spec_mail = nullptr;
  pprhip::HostMail* spec_mail_dev = nullptr;
  unsigned long long spec_mail_seq = 0;
  bool spec_failed = false;  // no candidate stream ran beside the compute stream
  hipEvent_t spec_ev[2] = {nullptr, nullptr};  // plan done (compute stream) / speculative push done (second stream)
  pprhip::KernelTimer spec_timer;
  hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  pprhip_tuning_t tun{};
  // resumable top-k push session (Forward_Push object state)
  bool topk_active = false;
  bool topk_first = true;
  int32_t topk_src = -1;
  double topk_alpha = 0.0;
  double topk_rsum = 1.0;
  // what `reserve`/`est` currently hold
  bool result_in_est = false;
};

namespace pprhip {

// ---- kernels_push.hip
// pk0: level 0's frontier (entries << 36 | edges) handed over as an argument; ~0: read it from ctr->hist[0] (a
// compaction kernel wrote it)
int launch_sparse_prepare(pprhip_graph* g, const PushArgs& a, int fbuf, int level, uint64_t nf_upper,
                          unsigned long long dense_thresh, bool scatter_dense, int cbuf, int dead_slot,
                          unsigned long long pk0 = ~0ull);
// levels first .. last of a batch on one workgroup, while they stay below wg_cap entries + edges (k_sparse_levels_wg);
// fbuf0: the list buffer that holds level 0's frontier
int launch_sparse_levels_wg(pprhip_graph* g, const pprhip::PushArgs& a, int fbuf0, int first, int last,
                            unsigned long long dense_thresh, unsigned long long wg_cap, int dead_slot,
                            unsigned long long pk0);
int launch_sparse_push(pprhip_graph* g, const PushArgs& a, int fbuf, int level, uint64_t ef_upper,
                       unsigned long long dense_thresh, int dead_slot, unsigned long long pk0 = ~0ull);
// One dense level of a single query, block by block (blocks: nullptr / 1 = the whole sweep at once).  state_in:
// device cell holding this level's GsState (a level launched behind another one without a host round trip; kGsNone:
// the kernels return at once), or nullptr: `state0` applies.  hist_out / state_out (nullable) receive the frontier
// the level leaves and the state the level after it has to run in (gs_next_state).
struct DenseLaunch {
  const pprhip::GsBlock* blocks = nullptr;
  int n_blocks = 1;
  const int* state_in = nullptr;
  int state0 = pprhip::kGsJacobi;
  unsigned long long* hist_out = nullptr;
  int* state_out = nullptr;
  unsigned long long dense_thresh = 0, gs_thresh = ~0ull;
};
int launch_dense_level(pprhip_graph* g, const PushArgs& a, int cbuf, int out_slot, int dead_slot,
                       const DenseLaunch& d = DenseLaunch());
namespace detail {
// engine.cpp: the sliced layout's edge windows of every block (nullptr / 1: the whole sweep)
const EdgeWindows* sliced_windows_of(pprhip_graph* g, const GsBlock* blocks, int nb);
}
constexpr uint32_t kApplyBlocks8 = 2048;  // workgroups of the batched apply kernel (per-slot partials each)
// slot arguments already staged in parent->h_slot_args; blocks: Gauss-Seidel blocks (nullptr / 1: one launch)
int launch_dense_level_b8(pprhip_graph* parent, bool backward, const pprhip::GsBlock* blocks = nullptr, int n_blocks = 1);
#ifdef PPRHIP_TEST_HOOKS
int launch_sweep_edges_only(pprhip_graph* parent, const pprhip::GsBlock& B);
int launch_count_live_lines(pprhip_graph* P, unsigned long long* d_out);
#endif
int launch_compact_prepared(pprhip_graph* g, int cbuf, int out_fbuf, unsigned long long* d_counter, bool backward);
int launch_count_active(pprhip_graph* g, const PushArgs& a, int seed_kind, int out_slot);
int launch_seed_list(pprhip_graph* g, const PushArgs& a, int seed_kind, int out_fbuf, unsigned long long* d_counter,
                     bool write_armed = false);
int launch_seed_dense(pprhip_graph* g, const PushArgs& a, int seed_kind, int cbuf, int out_slot, int dead_slot);
int launch_sum(pprhip_graph* g, const double* x, uint32_t n);  // result -> ctr->sum_out
int launch_sum_partial(pprhip_graph* g, const double* x, uint32_t n);  // partial sums -> g->partial, for the plan that follows
bool old_small_kernels();
inline uint32_t act_n(const pprhip_graph* g) { return g->n_act ? g->n_act : g->n; }  // entries a query's passes cover
int launch_set_f64(pprhip_graph* g, double* p, uint32_t idx, double value);
int launch_permute_out(pprhip_graph* g, const double* x, double* out);  // out[old] = x[old2new[old]]
// Once per device, from the thread that lifts the first graph onto it: loads the code object of every
// kernel file and opts the persistent sweep kernels into their large dynamic LDS, so that no launch
// path sets function attributes or triggers a module load later (worker threads launch concurrently).
int init_kernels_push();
int init_kernels_walk();
int init_kernels_select();
int init_kernels_apbs();

// ---- kernels_walk.hip
int launch_build_walk_rec(pprhip_graph* g);
int init_kernels_host();
int launch_publish(pprhip_graph* g, const void* src, uint32_t n_words, unsigned long long seq);
int launch_clear(pprhip_graph* g, const ClearList& L);
int launch_copy_f64(pprhip_graph* g, const double* src, double* dst, size_t n);  // dst[0 .. n) = src[0 .. n), in HBM
int launch_seed_one(pprhip_graph* g, int fbuf, int32_t node);                    // frontier list fbuf = {node}
int launch_hold(hipStream_t stream, unsigned long long ticks);
// The walk phase runs without a host round trip: the plan kernel counts sources and walks into DevCounters::mc_plan
// [g->mc_parity], the walk kernel (a fixed grid) reads them there.  omega_dev > 0: the plan derives rsum and the walk
// budget itself from the residue sum a reduction left in DevCounters::sum_out (top-k rounds: Fora_Topk.java:148-151);
// otherwise rsum / nrw are the host's.
int launch_mc_plan(pprhip_graph* g, int variant, double alpha, double rsum, double nrw, double omega_dev, double* target,
                   const double* copy_src = nullptr, double* copy_dst = nullptr);
int launch_mc_walk(pprhip_graph* g, double alpha, uint64_t seed, uint32_t stream, int no_zero_hop, double* target);
int launch_walk_batch(pprhip_graph* g, const int32_t* d_starts, const uint64_t* d_idx, uint64_t count, double alpha,
                      uint64_t seed, uint32_t stream, int no_zero_hop, int32_t* d_term, uint32_t* d_steps);
int launch_mc_pure(pprhip_graph* g, int32_t src, uint64_t n_walks, double alpha, uint64_t seed, double inc,
                   double* target);

// ---- kernels_apbs.hip
// rank that owns source v when [0, n) is cut into `world` contiguous ranges, the first n % world one longer
// (base = n / world, rem = n % world): THE partition rule of the sharded All-Pair - the device partition
// (k_owner_partition), the ranges ranks search (pprhip_shard_target_range) and the host-side partition the CPU
// multi-process tests drive (pprhip_owner_partition) all evaluate this one function
__host__ __device__ inline uint32_t owner_of(uint32_t v, uint32_t base, uint32_t rem) {
  const uint32_t cut = rem * (base + 1u);
  return v < cut ? v / (base + 1u) : rem + (v - cut) / base;
}

struct TripleRec {  // one index entry of All-Pair-Backward-Search on the device: pi(v, t) = p
  int32_t v, t;
  double p;
};
struct ApbsBuffers {
  unsigned long long *next_target = nullptr, *out_count = nullptr, *out_valid = nullptr, *overflow_count = nullptr;
  unsigned long long *stat_pops = nullptr, *stat_edges = nullptr;
  TripleRec* out_rec = nullptr;  // the searches' entries >= threshold, 16-byte records
  int32_t* overflow = nullptr;
  int32_t *list0 = nullptr, *list1 = nullptr;  // tier 1 over a range: the non-trivial targets, the small table's give-ups
  unsigned long long out_cap = 0;
  char* ws = nullptr;  // tier 2: per-workgroup dense workspaces (owned by the graph handle, see apbs_ws)
  void* board = nullptr;  // tier 2: one entry per workgroup on which it posts a level for helpers (zero at launch)
  unsigned long long* done_targets = nullptr;  // tier 2: targets finished in this launch
  uint32_t ws_blocks = 0, cap_t = 0, cap_f = 0, chunk = 0;
  uint32_t helpers = 0;  // tier 2: workgroups a launch may use in all (those beyond ws_blocks only help)
  unsigned long long* dbg = nullptr;  // developer switch PPRHIP_APBS_DEBUG: 10 words per workgroup (kernels_apbs.hip)
};
int launch_owner_partition(pprhip_graph* g, const TripleRec* rec, unsigned long long count, int world,
                           unsigned long long* cursors, TripleRec* out);
size_t apbs_dense_bytes(uint32_t n, unsigned long long m, uint32_t cap_t, uint32_t cap_f, uint32_t chunk);  // one workgroup's tier-2 workspace
uint32_t apbs_default_chunk();
size_t apbs_board_bytes(uint32_t blocks);
int launch_build_in_rec(pprhip_graph* g, void* rec);                  // rec: m records of 8 bytes
int launch_apbs(pprhip_graph* g, bool dense_tier, const int32_t* d_targets, uint32_t t_begin, uint32_t n_targets,
                double alpha, double rmax, ApbsBuffers& b);
// entries >= rmax of a reserve vector (internal ids [0, n)) as records of target t_old; *count counts them all, also
// beyond cap
int launch_emit_reserve(pprhip_graph* g, const double* reserve, uint32_t n, double rmax, int32_t t_old, TripleRec* out,
                        unsigned long long cap, unsigned long long* count);

// ---- kernels_sort.hip
// The index of the rows of sources [v_lo, v_hi) from rec[0 .. count), finished in device memory: rows in the reference's
// order (Base_Whole_Graph.java:112-163: k < 0 target order; k >= 0 entries >= the k-th largest, value descending, ties in
// target order), offsets[n + 1] and the kept entries' targets / values.  count == 0: all three null.  A source outside
// the range or a target outside [0, n) is PPRHIP_ERR_INVALID.  The caller releases the arrays (device_rows_free).
struct DeviceRows {
  unsigned long long* offsets = nullptr;
  int32_t* targets = nullptr;
  double* values = nullptr;
  unsigned long long entries = 0;
};
int finalize_rows_device(pprhip_graph* g, const TripleRec* rec, unsigned long long count, int k, uint32_t v_lo,
                         uint32_t v_hi, DeviceRows* out);
void device_rows_free(DeviceRows* r);
int init_kernels_sort();

// ---- kernels_select.hip
int launch_select_hist(pprhip_graph* g, const double* x, uint32_t n, unsigned long long prefix, int prefix_bits,
                       int digit_bits, bool first_pass);
int launch_select_choose(pprhip_graph* g, unsigned long long k);
int launch_select_gather(pprhip_graph* g, const double* x, uint32_t n, unsigned long long lower_bits, bool zero_count,
                         bool lower_from_device = false, const double* sum_cell = nullptr);

}  // namespace pprhip
