// comm.cpp — the multi-GPU side of the boundary: RCCL communicators behind the C ABI, the sharded
// All-Pair-Backward-Search (owner-of-source exchange on the device) and the multi-GPU entry points of
// SURVEY.md §8(b).
//
// Work is sharded, never split: queries (Gen_Util.java:208-232) and targets (Base_Whole_Graph.java:76-92) are
// independent, so every GPU holds a replica of the CSR and runs its share with the single-GPU code.  Two exchanges
// exist on the path, both only at the end of a call:
//   * batched FORA: the per-query top-k blocks travel to rank 0 (q * k * 12 bytes);
//   * All-Pair: the result is keyed by *source* (Base_Whole_Graph.java:84-86), so every rank sends each entry
//     (v, t, pi) to the rank that owns source v.  Entries stay in HBM as 16-byte records from the kernel that found
//     them until they arrive at their owner: partitioned by owner on the device, exchanged with grouped
//     ncclSend / ncclRecv (one message per peer, i.e. per xGMI link), and only then copied to the host, once,
//     by the rank that finalises them (the reference's k rule per source).
//
// A communicator is one rank of either transport:
//   * RCCL (`pprhip_comm_create`): ranks are processes or threads on distinct GPUs; librccl is loaded on first
//     use (dlopen: the single-GPU paths, the tests and the `ppr` CLI never pay for a 570 MB library);
//   * in-process (used by pprhip_fora_batch / pprhip_all_pair_backward_multi when several handles share one
//     device, which RCCL does not allow): the same record layout and the same partition, copies by hipMemcpy.
//
// Failure protocol (both transports): an exchange is a collective, so a rank that fails before it still takes part -
// it posts an error sentinel in the size words it sends, every peer sees it, nobody sends a payload and every rank
// returns an error (the failing rank its own).  Exceptions (std::bad_alloc from a rank's host vectors) are caught at
// the same places and become that rank's error code, so the rank still reaches the exchange.  Waits and their bounds:
//   * RCCL work queued on a stream: polled with a time limit (PPRHIP_COMM_TIMEOUT_S, default 1800) together with the
//     communicator's asynchronous error state; on either the communicator is aborted (ncclCommAbort), which releases
//     the peers' kernels as well;
//   * the in-process transport's barriers: the same time limit; a rank that gives up marks the group broken and every
//     rank in it (now or later) returns PPRHIP_ERR_STATE;
//   * pprhip_comm_create (ncclCommInitRank is a rendezvous of all ranks): runs on a helper thread and is awaited with
//     the same limit; when a peer never arrives the caller gets PPRHIP_ERR_STATE back and the helper thread stays
//     parked in RCCL until the process ends (there is no handle yet that could be aborted);
//   * one-process communicators come from ncclCommInitAll on the calling thread (no rendezvous between threads);
//   * NOT bounded by this file: the inside of ncclGroupEnd / ncclSend / ncclRecv / ncclCommInitAll / ncclCommDestroy
//     (RCCL's own blocking sections - they queue work and return; a destroy of an aborted communicator is skipped),
//     and hipStreamSynchronize on a rank's own stream after purely local work.
// STATUS: the RCCL transport with more than one rank has not executed on a real fabric yet (no multi-GPU box was
// available to this build; real RCCL runs with a group of one in tests/test_gpu_multi.py).  Its code has run with 2 and
// 3 ranks against a test double of the dozen RCCL calls (tests/fixtures/fake_rccl.cpp through PPRHIP_RCCL_LIB: ranks
// are threads, a send / receive pair is a device copy) - size exchange, sentinel, payload groups, gather, failing and
// leaving ranks - and the in-process transport, which shares the partition, the record layout and the sentinel
// protocol, carries the other 2- and 3-rank tests.
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>

#include "engine_internal.hpp"

using namespace pprhip;
using namespace pprhip::detail;

namespace {

// ---- the handful of RCCL entry points the path needs (rccl/rccl.h), bound at first use
typedef struct ncclComm* ncclComm_t;
struct ncclUniqueId { char internal[128]; };
enum { ncclSuccess = 0, ncclInProgress = 7 };
enum { ncclUint8 = 1 };

struct RcclApi {
  void* lib = nullptr;
  int (*GetUniqueId)(ncclUniqueId*) = nullptr;
  int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*CommAbort)(ncclComm_t) = nullptr;
  int (*CommGetAsyncError)(ncclComm_t, int*) = nullptr;
  int (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  int (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};

RcclApi* rccl() {
  static RcclApi api;
  static std::once_flag once;
  static bool ok = false;
  std::call_once(once, [] {
    // PPRHIP_RCCL_LIB names the library to bind instead (a site's own build; tests/fixtures/fake_rccl.cpp, the test
    // double that lets one GPU run the RCCL branch with several ranks)
    const char* own = tuning_env("PPRHIP_RCCL_LIB");
    if (own && *own) {
      api.lib = dlopen(own, RTLD_NOW | RTLD_LOCAL);
    } else {
      for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (api.lib) break;
      }
    }
    if (!api.lib) return;
    auto sym = [&](const char* n) { return dlsym(api.lib, n); };
    api.GetUniqueId = (int (*)(ncclUniqueId*))sym("ncclGetUniqueId");
    api.CommInitRank = (int (*)(ncclComm_t*, int, ncclUniqueId, int))sym("ncclCommInitRank");
    api.CommDestroy = (int (*)(ncclComm_t))sym("ncclCommDestroy");
    api.CommAbort = (int (*)(ncclComm_t))sym("ncclCommAbort");
    api.CommGetAsyncError = (int (*)(ncclComm_t, int*))sym("ncclCommGetAsyncError");
    api.CommInitAll = (int (*)(ncclComm_t*, int, const int*))sym("ncclCommInitAll");
    api.Send = (int (*)(const void*, size_t, int, int, ncclComm_t, hipStream_t))sym("ncclSend");
    api.Recv = (int (*)(void*, size_t, int, int, ncclComm_t, hipStream_t))sym("ncclRecv");
    api.GroupStart = (int (*)())sym("ncclGroupStart");
    api.GroupEnd = (int (*)())sym("ncclGroupEnd");
    api.GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
    ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.CommAbort && api.CommGetAsyncError &&
         api.CommInitAll && api.Send && api.Recv && api.GroupStart && api.GroupEnd && api.GetErrorString;
  });
  return ok ? &api : nullptr;
}

#define PPRHIP_CHECK_RCCL(expr)                                                                          \
  do {                                                                                                   \
    int _r = (expr);                                                                                     \
    if (_r != ncclSuccess) {                                                                             \
      set_error("%s failed: %s (%s:%d)", #expr, rccl()->GetErrorString(_r), __FILE__, __LINE__);         \
      return PPRHIP_ERR_HIP;                                                                             \
    }                                                                                                    \
  } while (0)

// ---- in-process transport: ranks are threads of one call whose handles share a device
struct LocalGroup {
  std::mutex mu;
  std::condition_variable cv;
  int world = 0, arrived = 0;
  uint64_t gen = 0;
  std::vector<const void*> send;                 // per rank: posted send buffer (device)
  std::vector<std::vector<uint64_t>> send_off;   // per rank: byte offsets per peer, world + 1
  std::vector<int> posted_rc;                    // per rank: the error it entered the exchange with (0: none)
  bool broken = false;
  // Returns true once every rank of the group has arrived.  Every rank takes part in every exchange, failed or not (it
  // posts its error instead of data), so the count completes and all ranks leave a barrier together: a send buffer
  // is never released while a peer may still be copying from it.  A rank that does not see the others within
  // `limit_s` (a peer's thread died outside the protocol) marks the group broken: it and everybody who is or comes
  // in a barrier of this group returns false.
  bool barrier(double limit_s) {
    std::unique_lock<std::mutex> lk(mu);
    if (broken) return false;
    const uint64_t my = gen;
    if (++arrived == world) {
      arrived = 0;
      ++gen;
      cv.notify_all();
      return true;
    }
    if (!cv.wait_for(lk, std::chrono::duration<double>(limit_s), [&] { return gen != my || broken; })) {
      broken = true;
      cv.notify_all();
    }
    return !broken && gen != my;
  }
  void fail() {
    std::lock_guard<std::mutex> lk(mu);
    broken = true;
    cv.notify_all();
  }
};

constexpr unsigned long long kFailedWord = ~0ull;  // size word of a rank that takes part in an exchange only to say it failed

double comm_timeout_s() {
  const char* e = tuning_env("PPRHIP_COMM_TIMEOUT_S");
  const double v = e ? atof(e) : 0.0;
  return v > 0.0 ? v : 1800.0;
}

// test switch: PPRHIP_FAULT_RANK=<r> makes rank r of a multi-GPU call fail at the point PPRHIP_FAULT_AT names
// ("search", the default: before anything was found; "partition": after the search, before the exchange;
// "exchange": inside the exchange, after the sizes are known), so that the failure protocol can be exercised
bool fault_injected(int rank, const char* at) {
  const char* r = hook_env("PPRHIP_FAULT_RANK");
  if (!r || atoi(r) != rank) return false;
  const char* where = hook_env("PPRHIP_FAULT_AT");
  if (strcmp(where ? where : "search", at) != 0) return false;
  set_error("injected fault on rank %d (%s)", rank, at);
  return true;
}

}  // namespace

struct pprhip_comm {
  pprhip_graph* g = nullptr;
  int device = 0;  // (kept here: the communicator may be destroyed after its graph)
  int rank = 0, world = 1;
  ncclComm_t nccl = nullptr;
  LocalGroup* local = nullptr;  // not owned
  bool dead = false;            // the RCCL communicator was aborted: every later collective fails at once
};

namespace {

// aborts the RCCL communicator (releases this rank's and, through the fabric, the peers' pending kernels)
void comm_abort(pprhip_comm* c) {
  if (c->nccl) {
    (void)rccl()->CommAbort(c->nccl);
    c->nccl = nullptr;
  }
  c->dead = true;
}

// Waits for the work queued on the rank's stream without ever blocking unboundedly: polls the stream, the
// communicator's asynchronous error state and a clock.  On an error or at the limit the communicator is aborted.
int comm_wait(pprhip_comm* c, const char* what) {
  pprhip_graph* g = c->g;
  const double limit = comm_timeout_s();
  const auto t0 = std::chrono::steady_clock::now();
  unsigned spins = 0;
  for (;;) {
    const hipError_t q = hipStreamQuery(g->stream);
    if (q == hipSuccess) return PPRHIP_OK;
    if (q != hipErrorNotReady) {
      set_error("%s: stream error on rank %d: %s", what, c->rank, hipGetErrorString(q));
      comm_abort(c);
      return PPRHIP_ERR_HIP;
    }
    if (c->nccl) {
      int ae = ncclSuccess;
      if (rccl()->CommGetAsyncError(c->nccl, &ae) == ncclSuccess && ae != ncclSuccess && ae != ncclInProgress) {
        set_error("%s: RCCL reports an asynchronous error on rank %d: %s", what, c->rank, rccl()->GetErrorString(ae));
        comm_abort(c);
        return PPRHIP_ERR_HIP;
      }
    }
    const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (el > limit) {
      set_error("%s: rank %d gave up after %.0f s (PPRHIP_COMM_TIMEOUT_S); a peer has failed or left the group", what,
                c->rank, el);
      comm_abort(c);
      return PPRHIP_ERR_STATE;
    }
    if (++spins < 2000) std::this_thread::yield();
    else std::this_thread::sleep_for(std::chrono::microseconds(spins < 20000 ? 50 : 1000));
  }
}

// Every rank sends bytes [off[p], off[p + 1]) of `send` to peer p and receives its peers' shares, in rank order,
// into *recv (device, allocated here; caller frees) with their byte offsets in recv_off (world + 1).
// local_rc != 0: this rank has failed already and only takes part to tell its peers (nothing is sent or received;
// its own error message stays the thread's last error).  Returns local_rc when that is set; otherwise
// PPRHIP_ERR_STATE when a peer failed (nothing was exchanged), a transport error, or PPRHIP_OK.
int comm_alltoallv(pprhip_comm* c, const void* send, const std::vector<uint64_t>& off, void** recv,
                   std::vector<uint64_t>& recv_off, int local_rc) {
  pprhip_graph* g = c->g;
  const int W = c->world;
  recv_off.assign((size_t)W + 1, 0);
  *recv = nullptr;
  const std::string own_msg = local_rc != PPRHIP_OK ? std::string(get_error()) : std::string();
  auto finish = [&](int rc) {
    if (local_rc != PPRHIP_OK) {  // the caller reports what went wrong on this rank, not what the exchange made of it
      set_error("%s", own_msg.c_str());
      return local_rc;
    }
    return rc;
  };
  if (c->local) {
    LocalGroup* L = c->local;
    {
      std::lock_guard<std::mutex> lk(L->mu);
      L->send[c->rank] = send;
      L->send_off[c->rank] = local_rc == PPRHIP_OK ? off : std::vector<uint64_t>((size_t)W + 1, 0);
      L->posted_rc[c->rank] = local_rc;
    }
    if (!L->barrier(comm_timeout_s())) {  // everybody has posted
      set_error("in-process exchange: rank %d waited %.0f s for its peers (a rank's thread left the call); group broken",
                c->rank, comm_timeout_s());
      return finish(PPRHIP_ERR_STATE);
    }
    int rc = PPRHIP_OK;
    for (int p = 0; p < W; ++p)
      if (L->posted_rc[p] != PPRHIP_OK && p != c->rank && rc == PPRHIP_OK) {
        set_error("in-process exchange: rank %d failed before the exchange (code %d); nothing was exchanged", p,
                  L->posted_rc[p]);
        rc = PPRHIP_ERR_STATE;
      }
    if (rc == PPRHIP_OK && local_rc == PPRHIP_OK) {
      for (int p = 0; p < W; ++p) recv_off[p + 1] = recv_off[p] + (L->send_off[p][c->rank + 1] - L->send_off[p][c->rank]);
      if (fault_injected(c->rank, "exchange")) rc = PPRHIP_ERR_STATE;
      if (rc == PPRHIP_OK) rc = alloc_dev(recv, recv_off[W]);
      for (int p = 0; p < W && rc == PPRHIP_OK; ++p) {
        const uint64_t bytes = recv_off[p + 1] - recv_off[p];
        if (bytes && hipMemcpyAsync((char*)*recv + recv_off[p], (const char*)L->send[p] + L->send_off[p][c->rank], bytes,
                                    hipMemcpyDefault, g->stream) != hipSuccess) {
          set_error("in-process exchange: copy from rank %d failed", p);
          rc = PPRHIP_ERR_HIP;
        }
      }
      // drained in every case: a peer's send buffer must not be in use by a copy of ours once we pass the barrier
      if (hipStreamSynchronize(g->stream) != hipSuccess && rc == PPRHIP_OK) {
        set_error("in-process exchange: stream error on rank %d", c->rank);
        rc = PPRHIP_ERR_HIP;
      }
    }
    const std::string msg = rc != PPRHIP_OK ? std::string(get_error()) : std::string();
    const bool all_read = L->barrier(comm_timeout_s());  // everybody has read: send buffers may go
    if (rc != PPRHIP_OK) set_error("%s", msg.c_str());
    else if (!all_read) {
      set_error("in-process exchange: rank %d waited %.0f s for its peers after the copies; group broken", c->rank,
                comm_timeout_s());
      rc = PPRHIP_ERR_STATE;
    }
    return finish(rc);
  }
  if (c->dead) {
    set_error("the communicator of rank %d was aborted by an earlier failure", c->rank);
    return finish(PPRHIP_ERR_STATE);
  }
  RcclApi* R = rccl();
  // 1) share sizes: one 8-byte message per peer; a failed rank sends the sentinel
  unsigned long long *d_cnt = nullptr;
  {
    const int arc = alloc_dev((void**)&d_cnt, sizeof(unsigned long long) * 2 * (size_t)W);
    if (arc != PPRHIP_OK) {  // cannot even take part: abort, so that the peers' size exchange ends with an error
      comm_abort(c);
      return finish(arc);
    }
  }
  std::vector<unsigned long long> h_cnt(2 * (size_t)W, 0);
  for (int p = 0; p < W; ++p) h_cnt[p] = local_rc != PPRHIP_OK ? kFailedWord : off[p + 1] - off[p];
  auto done = [&](int rc) {
    (void)hipFree(d_cnt);
    return finish(rc);
  };
  // a call of the group that fails leaves the communicator in an unknown state: abort it (peers are released)
  auto broken = [&](const char* what, int r) {
    set_error("%s failed on rank %d: %s", what, c->rank, R->GetErrorString(r));
    comm_abort(c);
    return done(PPRHIP_ERR_HIP);
  };
  if (hipMemcpyAsync(d_cnt, h_cnt.data(), sizeof(unsigned long long) * W, hipMemcpyHostToDevice, g->stream) != hipSuccess) {
    set_error("RCCL size exchange: upload failed on rank %d", c->rank);
    comm_abort(c);
    return done(PPRHIP_ERR_HIP);
  }
  {
    int r = R->GroupStart();
    if (r != ncclSuccess) return broken("ncclGroupStart", r);
    int first_bad = ncclSuccess;
    for (int p = 0; p < W; ++p) {
      const int rs = R->Send(d_cnt + p, 8, ncclUint8, p, c->nccl, g->stream);
      const int rr = R->Recv(d_cnt + W + p, 8, ncclUint8, p, c->nccl, g->stream);
      if (first_bad == ncclSuccess && rs != ncclSuccess) first_bad = rs;
      if (first_bad == ncclSuccess && rr != ncclSuccess) first_bad = rr;
    }
    r = R->GroupEnd();
    if (first_bad != ncclSuccess) return broken("ncclSend/ncclRecv (sizes)", first_bad);
    if (r != ncclSuccess) return broken("ncclGroupEnd (sizes)", r);
  }
  if (hipMemcpyAsync(h_cnt.data() + W, d_cnt + W, sizeof(unsigned long long) * W, hipMemcpyDeviceToHost, g->stream) !=
      hipSuccess) {
    set_error("RCCL size exchange: download failed on rank %d", c->rank);
    comm_abort(c);
    return done(PPRHIP_ERR_HIP);
  }
  {
    const int wrc = comm_wait(c, "RCCL size exchange");
    if (wrc != PPRHIP_OK) return done(wrc);
  }
  for (int p = 0; p < W; ++p)
    if (h_cnt[W + p] == kFailedWord && p != c->rank) {
      set_error("RCCL exchange: rank %d failed before the exchange; nothing was exchanged", p);
      return done(PPRHIP_ERR_STATE);
    }
  if (local_rc != PPRHIP_OK) return done(local_rc);
  for (int p = 0; p < W; ++p) recv_off[p + 1] = recv_off[p] + h_cnt[W + p];
  int rc = fault_injected(c->rank, "exchange") ? PPRHIP_ERR_STATE : alloc_dev(recv, recv_off[W]);
  if (rc != PPRHIP_OK) {  // the peers are past the point where they could be told: abort releases their payload group
    comm_abort(c);
    return done(rc);
  }
  // 2) the payload: one message per peer, all of them in flight together (each pair of GPUs has its own xGMI link)
  {
    int r = R->GroupStart();
    if (r != ncclSuccess) return broken("ncclGroupStart", r);
    int first_bad = ncclSuccess;
    for (int p = 0; p < W; ++p) {
      const uint64_t sb = off[p + 1] - off[p], rb = recv_off[p + 1] - recv_off[p];
      const int rs = sb ? R->Send((const char*)send + off[p], sb, ncclUint8, p, c->nccl, g->stream) : ncclSuccess;
      const int rr = rb ? R->Recv((char*)*recv + recv_off[p], rb, ncclUint8, p, c->nccl, g->stream) : ncclSuccess;
      if (first_bad == ncclSuccess && rs != ncclSuccess) first_bad = rs;
      if (first_bad == ncclSuccess && rr != ncclSuccess) first_bad = rr;
    }
    r = R->GroupEnd();
    if (first_bad != ncclSuccess) return broken("ncclSend/ncclRecv (payload)", first_bad);
    if (r != ncclSuccess) return broken("ncclGroupEnd (payload)", r);
  }
  return done(comm_wait(c, "RCCL payload exchange"));
}

void target_range(int rank, int world, uint32_t n, uint32_t* lo, uint32_t* hi) {
  const uint32_t base = n / (uint32_t)world, rem = n % (uint32_t)world;
  *lo = (uint32_t)rank * base + std::min<uint32_t>((uint32_t)rank, rem);
  *hi = *lo + base + ((uint32_t)rank < rem ? 1u : 0u);
}

// ---- work-weighted cut of the TARGET ranges (round 5).  Equal counts of contiguous ids balance the searches only
// while ids and in-degrees are unrelated: a search's cost follows its target's in-degree (R-MAT 18 - 20 at 1e-3: ~110 -
// 215 edge pushes per in-edge up to a few thousand in-edges, ten times that for the hubs, whose searches saturate the
// graph; tools/exp/apbs_cost_fit.py), and a store whose ids follow the import order of a degree-sorted dump gives rank
// 0 all of the hubs: 7.6 x the mean on R-MAT 18, with the other ranks idle.  No function of the in-degree alone
// predicts the hubs' cost (the ratio depends on graph and threshold: best of six forms 1.5 x the mean), so the hubs
// are MEASURED: a pilot runs the kPilotTop targets with the most in-edges and kPilotRanks more, spread geometrically
// over the in-degree ranks, as whole-vector searches on this handle (pprhip_backward_push's path) and counts their
// edge pushes and pops; a target's estimate is its own measurement if it has one, else the log-log interpolation of the
// measured means per in-degree class (half octaves, made monotone); the cuts fall where the running sum of the
// estimates in id order passes k / W of the total.  On the R-MAT 18 census every rank then holds 0.89 - 1.10 of the
// mean.  The sources a rank OWNS stay equal counts (owner_of): rows cost the same whoever finds their entries.
// Rank 0 decides and the others receive its cuts (cuts taken from every rank's own pilot could differ by a search
// whose residue lands within an ulp of the threshold, and ranges that do not tile [0, n) would be a wrong index).
constexpr int kPilotTop = 16, kPilotRanks = 48;

// >= `trigger` x the mean: the largest share of the modelled work (in-edges, hubs weighted up) that equal counts would
// give one rank - the cheap test that decides whether the pilot is worth its searches (PPRHIP_SHARD_CUT=count / work
// forces either)
double equal_count_skew(const pprhip_graph* g, int W) {
  // (every 8th id of a range stands for the range: the test only has to tell a degree-sorted store from a scrambled one,
  // and rank 0 runs it while its peers wait - 3 ms instead of 20 for R-MAT 22)
  const uint32_t n = g->n;
  const double d_star = std::max(64.0, (double)n / 256.0);
  std::vector<double> share((size_t)W, 0.0);
  double total = 0.0;
  for (int r = 0; r < W; ++r) {
    uint32_t lo, hi;
    target_range(r, W, n, &lo, &hi);
    double s = 0.0;
    const uint32_t step = hi - lo >= 4096 ? 8u : 1u;
    for (uint32_t t = lo; t < hi; t += step) {
      const double d = (double)hdeg_in(g, g->h_old2new[t]);
      s += 1.0 + d * (1.0 + d / d_star);
    }
    s *= (double)step;
    share[r] = s;
    total += s;
  }
  double mx = 0.0;
  for (double s : share) mx = std::max(mx, s);
  return total > 0.0 ? mx / (total / W) : 1.0;
}

int weighted_target_cuts(pprhip_graph* g, int W, double alpha, double threshold, std::vector<uint32_t>& cuts) {
  const uint32_t n = g->n;
  cuts.assign((size_t)W + 1, n);
  cuts[0] = 0;
  std::vector<uint32_t> deg(n);
  for (uint32_t t = 0; t < n; ++t) deg[t] = hdeg_in(g, g->h_old2new[t]);
  std::vector<uint32_t> order(n);
  for (uint32_t t = 0; t < n; ++t) order[t] = t;
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return deg[a] > deg[b]; });
  uint32_t nz = 0;
  while (nz < n && deg[order[nz]] > 0) ++nz;
  std::vector<double> est(n, 1.0);  // a target without in-edges is one entry and no search
  if (nz > 0) {
    // the pilot's targets: ranks 0 .. kPilotTop - 1, then geometrically up to the last target with in-edges
    std::vector<uint32_t> ranks;
    for (uint32_t r = 0; r < std::min<uint32_t>(kPilotTop, nz); ++r) ranks.push_back(r);
    if (nz > (uint32_t)kPilotTop) {
      const double a = std::log((double)kPilotTop + 1.0), b = std::log((double)nz);
      for (int i = 0; i < kPilotRanks; ++i) {
        const uint32_t r = std::min<uint32_t>(nz - 1, (uint32_t)std::exp(a + (b - a) * i / (kPilotRanks - 1)) - 1u);
        if (r > ranks.back()) ranks.push_back(r);
      }
    }
    std::vector<double> cost(ranks.size(), 0.0);
    for (size_t i = 0; i < ranks.size(); ++i) {
      pprhip_stats_t st;
      std::memset(&st, 0, sizeof st);
      const uint32_t t = order[ranks[i]];
      PPRHIP_TRY(backward_search_whole(g, g->h_old2new[t], alpha, threshold, st));
      cost[i] = 1.0 + (double)(st.edge_pushes + st.dense_edges) + 4.0 * (double)(st.pops + st.dense_nodes);
    }
    PPRHIP_CHECK_HIP(hipStreamSynchronize(g->stream));
    // measured means per half octave of the in-degree -> monotone log-log table
    std::vector<double> xs, ys;
    {
      size_t i = 0;
      while (i < ranks.size()) {
        const double cls = std::floor(std::log2((double)deg[order[ranks[i]]]) * 2.0);
        double sd = 0.0, sc = 0.0;
        size_t k = 0;
        while (i < ranks.size() && std::floor(std::log2((double)deg[order[ranks[i]]]) * 2.0) == cls) {
          sd += (double)deg[order[ranks[i]]];
          sc += cost[i];
          ++k;
          ++i;
        }
        xs.push_back(std::log(sd / (double)k));
        ys.push_back(std::log(sc / (double)k));
      }
      // (ranks run from high in-degree to low: make the table ascending in x, then monotone in y)
      std::reverse(xs.begin(), xs.end());
      std::reverse(ys.begin(), ys.end());
      for (size_t j = 1; j < ys.size(); ++j) ys[j] = std::max(ys[j], ys[j - 1]);
    }
    auto interp = [&](double d) {
      const double x = std::log(d);
      if (x <= xs.front()) return std::exp(ys.front());
      if (x >= xs.back()) return std::exp(ys.back());
      const size_t j = (size_t)(std::upper_bound(xs.begin(), xs.end(), x) - xs.begin());
      const double f = (x - xs[j - 1]) / (xs[j] - xs[j - 1]);
      return std::exp(ys[j - 1] + f * (ys[j] - ys[j - 1]));
    };
    for (uint32_t t = 0; t < n; ++t)
      if (deg[t]) est[t] = interp((double)deg[t]);
    for (size_t i = 0; i < ranks.size(); ++i) est[order[ranks[i]]] = cost[i];
  }
  double total = 0.0;
  for (double e : est) total += e;
  double run = 0.0;
  int k = 1;
  for (uint32_t t = 0; t < n && k < W; ++t) {
    run += est[t];
    while (k < W && run >= total * (double)k / (double)W) cuts[(size_t)k++] = t + 1;
  }
  return PPRHIP_OK;
}

// the target ranges of a sharded All-Pair run: cuts[r] .. cuts[r + 1] for rank r.  Rank 0 decides - equal counts, or the
// work-weighted cut when equal counts would leave one rank with more than 1.15 x the mean of the modelled work - and
// every rank receives its decision through the exchange every rank takes part in anyway (failure protocol included).
int decide_target_cuts(pprhip_comm* c, double alpha, double threshold, std::vector<uint32_t>& cuts, int pre_rc) {
  pprhip_graph* g = c->g;
  const int W = c->world;
  cuts.assign((size_t)W + 1, 0);
  int rc = pre_rc;
  uint32_t* d_send = nullptr;
  void* d_recv = nullptr;
  std::vector<uint64_t> off((size_t)W + 1, 0), roff;
  if (c->rank == 0 && rc == PPRHIP_OK) {
    try {
      const char* e = tuning_env("PPRHIP_SHARD_CUT");
      const bool by_work = e ? (e[0] == 'w') : equal_count_skew(g, W) > 1.15;
      if (by_work) {
        rc = weighted_target_cuts(g, W, alpha, threshold, cuts);
      } else {
        for (int r = 0; r < W; ++r) target_range(r, W, g->n, &cuts[(size_t)r], &cuts[(size_t)r + 1]);
      }
      if (rc == PPRHIP_OK) {
        std::vector<uint32_t> all((size_t)W * (W + 1));
        for (int p = 0; p < W; ++p) std::copy(cuts.begin(), cuts.end(), all.begin() + (size_t)p * (W + 1));
        rc = alloc_dev((void**)&d_send, sizeof(uint32_t) * all.size());
        if (rc == PPRHIP_OK && hipMemcpy(d_send, all.data(), sizeof(uint32_t) * all.size(), hipMemcpyHostToDevice) != hipSuccess) {
          set_error("sharded All-Pair: upload of the target cuts failed");
          rc = PPRHIP_ERR_HIP;
        }
        for (int p = 0; p <= W; ++p) off[(size_t)p] = (uint64_t)p * sizeof(uint32_t) * (W + 1);
      }
    } catch (const std::exception& ex) {
      set_error("sharded All-Pair: target cuts: %s", ex.what());
      rc = PPRHIP_ERR_OOM;
    }
  }
  rc = comm_alltoallv(c, d_send, off, &d_recv, roff, rc);
  if (rc == PPRHIP_OK) {
    if (roff[(size_t)W] != sizeof(uint32_t) * (size_t)(W + 1)) {
      set_error("sharded All-Pair: rank %d received %llu bytes of target cuts", c->rank, (unsigned long long)roff[(size_t)W]);
      rc = PPRHIP_ERR_STATE;
    } else if (hipMemcpy(cuts.data(), d_recv, sizeof(uint32_t) * (size_t)(W + 1), hipMemcpyDeviceToHost) != hipSuccess) {
      set_error("sharded All-Pair: download of the target cuts failed");
      rc = PPRHIP_ERR_HIP;
    } else {
      bool ok = cuts[0] == 0 && cuts[(size_t)W] == g->n;
      for (int r = 0; r < W; ++r) ok = ok && cuts[(size_t)r] <= cuts[(size_t)r + 1];
      if (!ok) {
        set_error("sharded All-Pair: rank %d received target cuts that do not tile [0, %u)", c->rank, g->n);
        rc = PPRHIP_ERR_STATE;
      }
    }
  }
  if (d_send) (void)hipFree(d_send);
  if (d_recv) (void)hipFree(d_recv);
  return rc;
}

// this rank's share of the sharded All-Pair: search its targets, exchange by owner of the source, finalise its sources.
// pre_rc != 0: the rank failed before it got here (its message is the thread's last error) and only takes part in the
// exchange to tell its peers.
int all_pair_sharded(pprhip_comm* c, double alpha, double threshold, int k, pprhip_index_t** own_out,
                     pprhip_stats_t* stats, int pre_rc = PPRHIP_OK) {
  pprhip_graph* g = c->g;
  const int W = c->world;
  int rc = pre_rc;
  if (g->n < (uint32_t)W) {
    // every rank sees the same n (replicas of one graph): ALL of them return here, a rank that came in with an error
    // of its own included - nobody goes on to an exchange the others have left
    if (rc == PPRHIP_OK) {
      set_error("sharded All-Pair: fewer nodes (%u) than ranks (%d)", g->n, W);
      rc = PPRHIP_ERR_INVALID;
    }
    return rc;
  }
  // the sources this rank owns: equal counts; the targets it searches: rank 0's cut (equal counts, or by modelled work)
  uint32_t lo = 0, hi = 0, t_lo = 0, t_hi = 0;
  target_range(c->rank, W, g->n, &lo, &hi);
  {
    std::vector<uint32_t> cuts;
    rc = decide_target_cuts(c, alpha, threshold, cuts, rc);
    if (rc == PPRHIP_OK) {
      t_lo = cuts[(size_t)c->rank];
      t_hi = cuts[(size_t)c->rank + 1];
    }
  }
  pprhip_stats_t st;
  std::memset(&st, 0, sizeof st);
  DeviceTripleSink sink;
  unsigned long long* d_cur = nullptr;
  TripleRec* d_part = nullptr;
  void* d_recv = nullptr;
  auto done = [&](int code) {
    if (d_cur) (void)hipFree(d_cur);
    if (d_part) (void)hipFree(d_part);
    if (d_recv) (void)hipFree(d_recv);
    return code;
  };
  std::vector<unsigned long long> start((size_t)W + 1, 0);
  // ---- search, then partition by owner of the source on the device; the first failure skips what is left, and the
  // rank still goes to the exchange (where its peers learn of it)
  auto local_part = [&]() -> int {
    if (fault_injected(c->rank, "search")) return PPRHIP_ERR_STATE;
    PPRHIP_TRY(all_pair_collect(g, alpha, threshold, t_lo, t_hi, sink, st));
    if (fault_injected(c->rank, "partition")) return PPRHIP_ERR_STATE;
    PPRHIP_TRY(alloc_dev((void**)&d_cur, sizeof(unsigned long long) * kBatch * 4));  // >= 64 counters
    PPRHIP_CHECK_HIP(hipMemsetAsync(d_cur, 0, sizeof(unsigned long long) * 64, g->stream));
    PPRHIP_TRY(launch_owner_partition(g, sink.rec, sink.count, W, d_cur, nullptr));
    std::vector<unsigned long long> cnt((size_t)W, 0);
    PPRHIP_CHECK_HIP(hipMemcpyAsync(cnt.data(), d_cur, sizeof(unsigned long long) * W, hipMemcpyDeviceToHost, g->stream));
    PPRHIP_CHECK_HIP(hipStreamSynchronize(g->stream));
    for (int p = 0; p < W; ++p) start[p + 1] = start[p] + cnt[p];
    PPRHIP_TRY(alloc_dev((void**)&d_part, sizeof(TripleRec) * std::max<unsigned long long>(1, sink.count)));
    PPRHIP_CHECK_HIP(hipMemcpyAsync(d_cur, start.data(), sizeof(unsigned long long) * W, hipMemcpyHostToDevice, g->stream));
    PPRHIP_TRY(launch_owner_partition(g, sink.rec, sink.count, W, d_cur, d_part));
    PPRHIP_CHECK_HIP(hipStreamSynchronize(g->stream));
    return PPRHIP_OK;
  };
  if (rc == PPRHIP_OK) {
    try {
      rc = local_part();
    } catch (const std::exception& e) {  // (host vectors: the rank still goes to the exchange, as a failed one)
      set_error("sharded All-Pair: rank %d: %s", c->rank, e.what());
      rc = PPRHIP_ERR_OOM;
    }
  }
  // ---- exchange: every entry goes to the rank that owns its source
  std::vector<uint64_t> off((size_t)W + 1, 0), roff;
  if (rc == PPRHIP_OK)
    for (int p = 0; p <= W; ++p) off[p] = start[p] * sizeof(TripleRec);
  if ((rc = comm_alltoallv(c, d_part, off, &d_recv, roff, rc))) return done(rc);
  // ---- the entries of this rank's sources are put in row order on the device and cross PCIe once, here; what
  // arrived must be rows this rank owns (a peer's partition or a transport gone wrong must not become an out-of-range
  // write in the finalisation: index_from_device checks the range)
  const uint64_t n_recv = roff[W] / sizeof(TripleRec);
  st.select_bytes = roff[W];          // bytes received in the exchange
  st.mc_sources = sink.count;         // entries this rank found (before the exchange)
  try {  // (no collective follows: an error here is this rank's alone)
    rc = index_from_device(g, (const TripleRec*)d_recv, n_recv, k, lo, hi, own_out);
  } catch (const std::exception& e) {
    set_error("sharded All-Pair: rank %d: index finalisation: %s", c->rank, e.what());
    rc = PPRHIP_ERR_OOM;
  }
  if (rc != PPRHIP_OK) return done(rc);
  if (stats) *stats = st;
  return done(PPRHIP_OK);
}

// pprhip_topk_gather; pre_rc as in all_pair_sharded
int topk_gather_impl(pprhip_comm* c, const int32_t* ids, const double* vals, int rows, int rows_max, int k,
                     int32_t* ids_root, double* vals_root, int pre_rc) {
  pprhip_graph* g = c->g;
  // one block per rank: rows_max rows of k (id, value) pairs, ids first; short blocks are padded with id -1
  const size_t blk_ids = sizeof(int32_t) * (size_t)rows_max * k, blk_vals = sizeof(double) * (size_t)rows_max * k;
  const size_t blk = blk_ids + blk_vals;
  void *d_send = nullptr, *d_recv = nullptr;
  auto done = [&](int rc) {
    if (d_send) (void)hipFree(d_send);
    if (d_recv) (void)hipFree(d_recv);
    return rc;
  };
  int rc = pre_rc;
  std::vector<char> h;
  auto stage = [&]() -> int {
    if (fault_injected(c->rank, "gather")) return PPRHIP_ERR_STATE;
    h.resize(blk);
    int32_t* hi = (int32_t*)h.data();
    double* hv = (double*)(h.data() + blk_ids);
    for (size_t i = 0; i < (size_t)rows_max * k; ++i) {
      hi[i] = i < (size_t)rows * k ? ids[i] : -1;
      hv[i] = i < (size_t)rows * k ? vals[i] : 0.0;
    }
    PPRHIP_TRY(alloc_dev(&d_send, blk));
    PPRHIP_CHECK_HIP(hipMemcpyAsync(d_send, h.data(), blk, hipMemcpyHostToDevice, g->stream));
    PPRHIP_CHECK_HIP(hipStreamSynchronize(g->stream));
    return PPRHIP_OK;
  };
  if (rc == PPRHIP_OK) {
    try {
      rc = stage();
    } catch (const std::exception& e) {
      set_error("pprhip_topk_gather: rank %d: %s", c->rank, e.what());
      rc = PPRHIP_ERR_OOM;
    }
  }
  std::vector<uint64_t> off((size_t)c->world + 1, 0), roff;
  if (rc == PPRHIP_OK)
    for (int p = 0; p < c->world; ++p) off[p + 1] = off[p] + (p == 0 ? blk : 0);  // everything goes to rank 0
  if ((rc = comm_alltoallv(c, d_send, off, &d_recv, roff, rc))) return done(rc);
  if (c->rank == 0) {
    if (roff[c->world] != blk * (size_t)c->world) {
      set_error("pprhip_topk_gather: the ranks sent %llu bytes, %llu expected (rows_max or k differ between ranks)",
                (unsigned long long)roff[c->world], (unsigned long long)(blk * (size_t)c->world));
      return done(PPRHIP_ERR_STATE);
    }
    std::vector<char> all(roff[c->world]);
    if (hipMemcpyAsync(all.data(), d_recv, all.size(), hipMemcpyDeviceToHost, g->stream) != hipSuccess ||
        hipStreamSynchronize(g->stream) != hipSuccess) {
      set_error("pprhip_topk_gather: download failed");
      return done(PPRHIP_ERR_HIP);
    }
    for (int p = 0; p < c->world; ++p) {
      std::memcpy(ids_root + (size_t)p * rows_max * k, all.data() + roff[p], blk_ids);
      std::memcpy(vals_root + (size_t)p * rows_max * k, all.data() + roff[p] + blk_ids, blk_vals);
    }
  }
  return done(PPRHIP_OK);
}

}  // namespace

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

int pprhip_comm_unique_id(void* id_out) {
  if (!id_out) {
    set_error("pprhip_comm_unique_id: null output");
    return PPRHIP_ERR_INVALID;
  }
  RcclApi* R = rccl();
  if (!R) {
    set_error("pprhip_comm_unique_id: librccl.so.1 could not be loaded (%s)", dlerror() ? dlerror() : "symbols missing");
    return PPRHIP_ERR_NO_DEVICE;
  }
  ncclUniqueId id;
  PPRHIP_CHECK_RCCL(R->GetUniqueId(&id));
  std::memcpy(id_out, &id, sizeof id);
  return PPRHIP_OK;
}

int pprhip_comm_create(pprhip_graph_t* g, const void* id, int rank, int world, pprhip_comm_t** comm_out) {
  PPRHIP_TRY(check_graph(g, "pprhip_comm_create"));
  if (!id || !comm_out || world < 1 || rank < 0 || rank >= world || world > 64) {
    set_error("pprhip_comm_create: bad arguments (rank %d of %d)", rank, world);
    return PPRHIP_ERR_INVALID;
  }
  RcclApi* R = rccl();
  if (!R) {
    set_error("pprhip_comm_create: librccl.so.1 could not be loaded");
    return PPRHIP_ERR_NO_DEVICE;
  }
  std::unique_ptr<pprhip_comm> c(new (std::nothrow) pprhip_comm());
  if (!c) return PPRHIP_ERR_OOM;
  c->g = g;
  c->device = g->device;
  c->rank = rank;
  c->world = world;
  // ncclCommInitRank is a rendezvous: it returns when all `world` ranks have called it.  It runs on a helper thread
  // and is awaited with the path's time limit, so a peer that never arrives costs this rank an error, not its
  // process (the helper then stays parked inside RCCL: nothing exists yet that could be aborted).
  struct Init {
    std::mutex mu;
    std::condition_variable cv;
    bool done = false;
    int r = ncclSuccess;
    ncclComm_t comm = nullptr;
  };
  auto st = std::make_shared<Init>();
  ncclUniqueId uid;
  std::memcpy(&uid, id, sizeof uid);
  const int device = g->device;
  std::thread([st, R, uid, rank, world, device] {
    ncclComm_t cm = nullptr;
    int r = hipSetDevice(device) == hipSuccess ? R->CommInitRank(&cm, world, uid, rank) : 2 /* ncclSystemError */;
    std::lock_guard<std::mutex> lk(st->mu);
    st->r = r;
    st->comm = cm;
    st->done = true;
    st->cv.notify_all();
  }).detach();
  {
    std::unique_lock<std::mutex> lk(st->mu);
    if (!st->cv.wait_for(lk, std::chrono::duration<double>(comm_timeout_s()), [&] { return st->done; })) {
      set_error("pprhip_comm_create: rank %d of %d waited %.0f s (PPRHIP_COMM_TIMEOUT_S) for its peers in "
                "ncclCommInitRank; a rank never arrived", rank, world, comm_timeout_s());
      return PPRHIP_ERR_STATE;
    }
    if (st->r != ncclSuccess) {
      set_error("ncclCommInitRank failed on rank %d of %d: %s", rank, world, R->GetErrorString(st->r));
      return PPRHIP_ERR_HIP;
    }
    c->nccl = st->comm;
  }
  *comm_out = c.release();
  return PPRHIP_OK;
}

void pprhip_comm_destroy(pprhip_comm_t* c) {
  if (!c) return;
  if (c->nccl) {  // (an aborted communicator has no handle left: comm_abort took it)
    (void)hipSetDevice(c->device);
    (void)rccl()->CommDestroy(c->nccl);
  }
  delete c;
}

int pprhip_comm_info(const pprhip_comm_t* c, int* rank, int* world) {
  if (!c) {
    set_error("pprhip_comm_info: null communicator");
    return PPRHIP_ERR_INVALID;
  }
  if (rank) *rank = c->rank;
  if (world) *world = c->world;
  return PPRHIP_OK;
}

int pprhip_shard_target_range(int rank, int world, uint32_t n, uint32_t* begin, uint32_t* end) {
  if (world < 1 || rank < 0 || rank >= world || !begin || !end) {
    set_error("pprhip_shard_target_range: bad arguments");
    return PPRHIP_ERR_INVALID;
  }
  target_range(rank, world, n, begin, end);
  return PPRHIP_OK;
}

int pprhip_shard_target_cuts(pprhip_graph_t* g, int world, double alpha, double threshold, int mode, uint32_t* cuts_out,
                             double* skew_out) {
  PPRHIP_TRY(check_graph(g, "pprhip_shard_target_cuts"));
  if (world < 1 || !cuts_out || mode < 0 || mode > 2 || !(threshold > 0.0)) {
    set_error("pprhip_shard_target_cuts: bad arguments (world %d, mode %d)", world, mode);
    return PPRHIP_ERR_INVALID;
  }
  try {
    const double skew = equal_count_skew(g, world);
    if (skew_out) *skew_out = skew;
    std::vector<uint32_t> cuts((size_t)world + 1, 0);
    if (mode == 1 || (mode == 2 && skew > 1.15)) {
      PPRHIP_TRY(weighted_target_cuts(g, world, alpha, threshold, cuts));
    } else {
      for (int r = 0; r < world; ++r) target_range(r, world, g->n, &cuts[(size_t)r], &cuts[(size_t)r + 1]);
    }
    std::copy(cuts.begin(), cuts.end(), cuts_out);
  } catch (const std::bad_alloc&) {
    set_error("pprhip_shard_target_cuts: out of host memory");
    return PPRHIP_ERR_OOM;
  }
  return PPRHIP_OK;
}

int pprhip_all_pair_backward_sharded(pprhip_comm_t* c, double alpha, double threshold, int k, pprhip_index_t** own_out,
                                     pprhip_stats_t* stats) {
  if (!c || !own_out) {
    set_error("pprhip_all_pair_backward_sharded: null argument");
    return PPRHIP_ERR_INVALID;
  }
  PPRHIP_TRY(check_graph(c->g, "pprhip_all_pair_backward_sharded"));
  return all_pair_sharded(c, alpha, threshold, k, own_out, stats);
}

int pprhip_topk_gather(pprhip_comm_t* c, const int32_t* ids, const double* vals, int rows, int rows_max, int k,
                       int32_t* ids_root, double* vals_root) {
  if (!c || rows < 0 || rows > rows_max || k < 1 || (rows && (!ids || !vals)) ||
      (c->rank == 0 && (!ids_root || !vals_root))) {
    set_error("pprhip_topk_gather: bad arguments (rows %d of %d, k %d)", rows, rows_max, k);
    return PPRHIP_ERR_INVALID;
  }
  PPRHIP_TRY(check_graph(c->g, "pprhip_topk_gather"));
  return topk_gather_impl(c, ids, vals, rows, rows_max, k, ids_root, vals_root, PPRHIP_OK);
}

int pprhip_owner_partition(uint32_t n, int world, const int32_t* sources, uint64_t count, uint64_t* counts_out,
                           uint64_t* order_out) {
  if (world < 1 || n < (uint32_t)world || (count && (!sources || !order_out)) || !counts_out) {
    set_error("pprhip_owner_partition: bad arguments (n = %u, world = %d)", n, world);
    return PPRHIP_ERR_INVALID;
  }
  const uint32_t base = n / (uint32_t)world, rem = n % (uint32_t)world;
  std::vector<uint64_t> at((size_t)world + 1, 0);
  for (uint64_t i = 0; i < count; ++i) {
    if (sources[i] < 0 || (uint32_t)sources[i] >= n) {
      set_error("pprhip_owner_partition: source %d outside [0, %u)", sources[i], n);
      return PPRHIP_ERR_INVALID;
    }
    at[owner_of((uint32_t)sources[i], base, rem) + 1]++;
  }
  for (int p = 0; p < world; ++p) {
    counts_out[p] = at[p + 1];
    at[p + 1] += at[p];
  }
  for (uint64_t i = 0; i < count; ++i) order_out[at[owner_of((uint32_t)sources[i], base, rem)]++] = i;  // stable
  return PPRHIP_OK;
}

int pprhip_comm_abort(pprhip_comm_t* c) {
  if (!c) {
    set_error("pprhip_comm_abort: null communicator");
    return PPRHIP_ERR_INVALID;
  }
  if (!c->local) comm_abort(c);
  return PPRHIP_OK;
}

}  // extern "C"

// ---------------------------------------------------------------- one process, several GPUs
namespace {

struct RankSetup {
  std::vector<pprhip_comm> comms;
  LocalGroup local;
  bool use_rccl = false;
  std::vector<std::string> errs;
  std::vector<int> rcs;
};

// communicators for the handles of one call: RCCL when every handle sits on its own device, in-process otherwise
int setup_ranks(pprhip_graph_t* const* per_gpu, int n_gpu, RankSetup& S, const char* fn) {
  if (!per_gpu || n_gpu < 1 || n_gpu > 64) {
    set_error("%s: bad handle list (n_gpu = %d)", fn, n_gpu);
    return PPRHIP_ERR_INVALID;
  }
  for (int r = 0; r < n_gpu; ++r) {
    if (!per_gpu[r] || per_gpu[r]->n != per_gpu[0]->n || per_gpu[r]->m != per_gpu[0]->m) {
      set_error("%s: handle %d is null or holds another graph (every GPU needs a replica of the same CSR)", fn, r);
      return PPRHIP_ERR_INVALID;
    }
    for (int s = 0; s < r; ++s)
      if (per_gpu[s] == per_gpu[r]) {
        set_error("%s: handle %d is listed twice (one handle serves one thread at a time)", fn, r);
        return PPRHIP_ERR_INVALID;
      }
  }
  bool distinct = true;
  for (int r = 0; r < n_gpu; ++r)
    for (int s = 0; s < r; ++s) distinct = distinct && per_gpu[r]->device != per_gpu[s]->device;
  S.comms.assign((size_t)n_gpu, pprhip_comm());
  S.errs.assign((size_t)n_gpu, "");
  S.rcs.assign((size_t)n_gpu, PPRHIP_OK);
  // (PPRHIP_FORCE_RCCL=1, test switch: the RCCL branch although the handles share a device - only the test double of
  // tests/fixtures/fake_rccl.cpp accepts that)
  S.use_rccl = (distinct || hook_env("PPRHIP_FORCE_RCCL") != nullptr) && n_gpu > 1;
  S.local.world = n_gpu;
  S.local.send.assign((size_t)n_gpu, nullptr);
  S.local.send_off.assign((size_t)n_gpu, {});
  S.local.posted_rc.assign((size_t)n_gpu, PPRHIP_OK);
  for (int r = 0; r < n_gpu; ++r) {
    S.comms[r].g = per_gpu[r];
    S.comms[r].rank = r;
    S.comms[r].world = n_gpu;
    S.comms[r].local = S.use_rccl ? nullptr : &S.local;
  }
  return PPRHIP_OK;
}

// One host thread per GPU.  Everything that can fail before the ranks depend on each other happens on the calling
// thread first: every device is selected once, and the RCCL communicators of all ranks are created by ONE
// ncclCommInitAll call, so no rank can be left waiting in an initialisation rendezvous that another one never
// reaches.  fn(rank, pre_rc) returns a code; pre_rc != 0 tells it that this rank's thread could not set itself up
// and must only take part in the call's exchange.  Messages are carried back to the caller's thread.
template <class F>
int run_ranks(RankSetup& S, F fn) {
  const int W = (int)S.comms.size();
  int dev0 = 0;
  (void)hipGetDevice(&dev0);
  for (int r = 0; r < W; ++r)
    if (hipSetDevice(S.comms[r].g->device) != hipSuccess) {
      set_error("GPU %d: hipSetDevice(%d) failed", r, S.comms[r].g->device);
      (void)hipSetDevice(dev0);
      return PPRHIP_ERR_NO_DEVICE;
    }
  (void)hipSetDevice(dev0);
  if (S.use_rccl) {
    RcclApi* R = rccl();
    if (!R) {
      set_error("librccl.so.1 could not be loaded");
      return PPRHIP_ERR_NO_DEVICE;
    }
    std::vector<ncclComm_t> cs((size_t)W, nullptr);
    std::vector<int> devs((size_t)W);
    for (int r = 0; r < W; ++r) devs[r] = S.comms[r].g->device;
    PPRHIP_CHECK_RCCL(R->CommInitAll(cs.data(), W, devs.data()));
    for (int r = 0; r < W; ++r) S.comms[r].nccl = cs[r];
    (void)hipSetDevice(dev0);
  }
  std::vector<std::thread> th;
  for (int r = 0; r < W; ++r)
    th.emplace_back([&, r] {
      int pre = PPRHIP_OK;
      if (hipSetDevice(S.comms[r].g->device) != hipSuccess) {
        set_error("hipSetDevice(%d) failed on rank %d's thread", S.comms[r].g->device, r);
        pre = PPRHIP_ERR_NO_DEVICE;
      }
      int rc;
      try {
        rc = fn(r, pre);
      } catch (const std::exception& e) {
        // fn's own code catches what can throw before its exchange; this is the last line of defence (nothing may
        // reach std::terminate): the in-process group is marked broken so that no peer waits for this rank
        set_error("rank %d: %s", r, e.what());
        rc = PPRHIP_ERR_OOM;
        S.local.fail();
        if (S.comms[r].nccl) comm_abort(&S.comms[r]);
      }
      if (rc != PPRHIP_OK) S.errs[r] = get_error();
      S.rcs[r] = rc;
    });
  for (auto& t : th) t.join();
  for (int r = 0; r < W; ++r)
    if (S.comms[r].nccl) {
      (void)hipSetDevice(S.comms[r].g->device);
      (void)rccl()->CommDestroy(S.comms[r].nccl);
      S.comms[r].nccl = nullptr;
    }
  (void)hipSetDevice(dev0);
  // the rank whose own failure started it, not a peer that merely heard of it
  int first = -1;
  for (int r = 0; r < W; ++r)
    if (S.rcs[r] != PPRHIP_OK && (first < 0 || (S.errs[first].find("failed before the exchange") != std::string::npos &&
                                               S.errs[r].find("failed before the exchange") == std::string::npos)))
      first = r;
  if (first >= 0) {
    set_error("GPU %d (device %d): %s", first, S.comms[first].g->device, S.errs[first].c_str());
    return S.rcs[first];
  }
  return PPRHIP_OK;
}

}  // namespace

extern "C" {

int pprhip_fora_batch(pprhip_graph_t* const* per_gpu, int n_gpu, const int32_t* srcs, int q, int k, double eps,
                      const pprhip_fora_conf_t* conf, uint64_t seed, int n_rounds, int32_t* ids_out, double* vals_out,
                      int* n_out, pprhip_stats_t* stats_per_gpu) {
  RankSetup S;
  PPRHIP_TRY(setup_ranks(per_gpu, n_gpu, S, "pprhip_fora_batch"));
  if (q < 0 || k < 1 || !conf || !(eps > 0.0) || (q && (!srcs || !ids_out || !vals_out))) {
    set_error("pprhip_fora_batch: bad arguments (q=%d k=%d eps=%g)", q, k, eps);
    return PPRHIP_ERR_INVALID;
  }
  const int W = n_gpu;
  const int rows_max = (q + W - 1) / W;
  std::vector<int32_t> ids_root((size_t)W * rows_max * k);
  std::vector<double> vals_root((size_t)W * rows_max * k);
  std::vector<std::vector<int>> nsel((size_t)W);
  const int rc = run_ranks(S, [&](int r, int pre) -> int {
    // query i runs on GPU i mod W (local row i / W)
    std::vector<int32_t> mine;
    for (int i = r; i < q; i += W) mine.push_back(srcs[i]);
    const int rows = (int)mine.size();
    std::vector<int32_t> ids((size_t)std::max(1, rows) * k);
    std::vector<double> vals((size_t)std::max(1, rows) * k);
    nsel[r].assign((size_t)std::max(1, rows), 0);
    pprhip_stats_t st;
    std::memset(&st, 0, sizeof st);
    int e = pre;
    if (e == PPRHIP_OK && fault_injected(r, "search")) e = PPRHIP_ERR_STATE;
    if (e == PPRHIP_OK)
      e = pprhip_fora_batch_single_source(S.comms[r].g, mine.data(), rows, eps, conf, seed, n_rounds, nullptr, k,
                                          ids.data(), vals.data(), nsel[r].data(), nullptr, &st);
    if (stats_per_gpu) stats_per_gpu[r] = st;
    // the only exchange of the path: the top-k blocks travel to GPU 0 over the fabric (a rank that failed takes
    // part to say so)
    return topk_gather_impl(&S.comms[r], ids.data(), vals.data(), rows, rows_max, k, ids_root.data(), vals_root.data(), e);
  });
  if (rc != PPRHIP_OK) return rc;
  for (int i = 0; i < q; ++i) {
    const int r = i % W, j = i / W;
    std::memcpy(ids_out + (size_t)i * k, ids_root.data() + ((size_t)r * rows_max + j) * k, sizeof(int32_t) * k);
    std::memcpy(vals_out + (size_t)i * k, vals_root.data() + ((size_t)r * rows_max + j) * k, sizeof(double) * k);
    if (n_out) n_out[i] = nsel[r][j];
  }
  return PPRHIP_OK;
}

int pprhip_all_pair_backward_multi(pprhip_graph_t* const* per_gpu, int n_gpu, double alpha, double threshold, int k,
                                   pprhip_index_t** index_out, pprhip_stats_t* stats_per_gpu) {
  RankSetup S;
  PPRHIP_TRY(setup_ranks(per_gpu, n_gpu, S, "pprhip_all_pair_backward_multi"));
  if (!index_out) {
    set_error("pprhip_all_pair_backward_multi: null output");
    return PPRHIP_ERR_INVALID;
  }
  std::vector<pprhip_index_t*> own((size_t)n_gpu, nullptr);
  int rc = run_ranks(S, [&](int r, int pre) -> int {
    pprhip_stats_t st;
    std::memset(&st, 0, sizeof st);
    const int e = all_pair_sharded(&S.comms[r], alpha, threshold, k, &own[r], &st, pre);
    if (stats_per_gpu) stats_per_gpu[r] = st;
    return e;
  });
  if (rc == PPRHIP_OK) rc = index_concat(own, index_out);  // every rank holds the rows of its own sources
  for (pprhip_index_t* p : own) pprhip_index_destroy(p);
  return rc;
}

}  // extern "C"
