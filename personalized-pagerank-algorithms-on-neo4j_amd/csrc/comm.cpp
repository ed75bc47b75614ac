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
#include <dlfcn.h>

#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <thread>

#include "engine_internal.hpp"

using namespace pprhip;
using namespace pprhip::detail;

namespace {

// ---- the handful of RCCL entry points the path needs (rccl/rccl.h), bound at first use
typedef struct ncclComm* ncclComm_t;
struct ncclUniqueId { char internal[128]; };
enum { ncclSuccess = 0 };
enum { ncclUint8 = 1 };

struct RcclApi {
  void* lib = nullptr;
  int (*GetUniqueId)(ncclUniqueId*) = nullptr;
  int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
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
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (api.lib) break;
    }
    if (!api.lib) return;
    auto sym = [&](const char* n) { return dlsym(api.lib, n); };
    api.GetUniqueId = (int (*)(ncclUniqueId*))sym("ncclGetUniqueId");
    api.CommInitRank = (int (*)(ncclComm_t*, int, ncclUniqueId, int))sym("ncclCommInitRank");
    api.CommDestroy = (int (*)(ncclComm_t))sym("ncclCommDestroy");
    api.Send = (int (*)(const void*, size_t, int, int, ncclComm_t, hipStream_t))sym("ncclSend");
    api.Recv = (int (*)(void*, size_t, int, int, ncclComm_t, hipStream_t))sym("ncclRecv");
    api.GroupStart = (int (*)())sym("ncclGroupStart");
    api.GroupEnd = (int (*)())sym("ncclGroupEnd");
    api.GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
    ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.Send && api.Recv && api.GroupStart &&
         api.GroupEnd && api.GetErrorString;
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
  int err = 0;
  // returns once every rank has arrived - or at once, for good, after a rank has called abort()
  void barrier() {
    std::unique_lock<std::mutex> lk(mu);
    if (err) return;
    const uint64_t my = gen;
    if (++arrived == world) {
      arrived = 0;
      ++gen;
      cv.notify_all();
    } else {
      cv.wait(lk, [&] { return gen != my || err != 0; });
    }
  }
  void abort(int rc) {  // a rank that fails lets the others out of their barriers
    std::lock_guard<std::mutex> lk(mu);
    if (!err) err = rc;
    cv.notify_all();
  }
};

}  // namespace

struct pprhip_comm {
  pprhip_graph* g = nullptr;
  int rank = 0, world = 1;
  ncclComm_t nccl = nullptr;
  LocalGroup* local = nullptr;  // not owned
};

namespace {

// Every rank sends bytes [off[p], off[p + 1]) of `send` to peer p and receives its peers' shares, in rank order,
// into *recv (device, allocated here; caller frees) with their byte offsets in recv_off (world + 1).
int comm_alltoallv(pprhip_comm* c, const void* send, const std::vector<uint64_t>& off, void** recv,
                   std::vector<uint64_t>& recv_off) {
  pprhip_graph* g = c->g;
  const int W = c->world;
  recv_off.assign((size_t)W + 1, 0);
  *recv = nullptr;
  if (c->local) {
    LocalGroup* L = c->local;
    {
      std::lock_guard<std::mutex> lk(L->mu);
      L->send[c->rank] = send;
      L->send_off[c->rank] = off;
    }
    L->barrier();  // everybody has posted
    if (L->err) {
      set_error("in-process exchange: another rank failed");
      return L->err;
    }
    for (int p = 0; p < W; ++p) recv_off[p + 1] = recv_off[p] + (L->send_off[p][c->rank + 1] - L->send_off[p][c->rank]);
    int rc = alloc_dev(recv, recv_off[W]);
    if (rc == PPRHIP_OK)
      for (int p = 0; p < W && rc == PPRHIP_OK; ++p) {
        const uint64_t bytes = recv_off[p + 1] - recv_off[p];
        if (bytes && hipMemcpyAsync((char*)*recv + recv_off[p], (const char*)L->send[p] + L->send_off[p][c->rank], bytes,
                                    hipMemcpyDefault, g->stream) != hipSuccess) {
          set_error("in-process exchange: copy from rank %d failed", p);
          rc = PPRHIP_ERR_HIP;
        }
      }
    if (rc == PPRHIP_OK && hipStreamSynchronize(g->stream) != hipSuccess) rc = PPRHIP_ERR_HIP;
    if (rc != PPRHIP_OK) L->abort(rc);
    L->barrier();  // everybody has read: send buffers may go
    if (L->err && rc == PPRHIP_OK) set_error("in-process exchange: another rank failed");
    return L->err ? (rc != PPRHIP_OK ? rc : L->err) : PPRHIP_OK;
  }
  RcclApi* R = rccl();
  // 1) share sizes: one 8-byte message per peer
  unsigned long long *d_cnt = nullptr;
  PPRHIP_TRY(alloc_dev((void**)&d_cnt, sizeof(unsigned long long) * 2 * (size_t)W));
  std::vector<unsigned long long> h_cnt(2 * (size_t)W, 0);
  for (int p = 0; p < W; ++p) h_cnt[p] = off[p + 1] - off[p];
  auto done = [&](int rc) {
    (void)hipFree(d_cnt);
    return rc;
  };
  if (hipMemcpyAsync(d_cnt, h_cnt.data(), sizeof(unsigned long long) * W, hipMemcpyHostToDevice, g->stream) != hipSuccess)
    return done(PPRHIP_ERR_HIP);
  if (R->GroupStart() != ncclSuccess) return done(PPRHIP_ERR_HIP);
  for (int p = 0; p < W; ++p) {
    (void)R->Send(d_cnt + p, 8, ncclUint8, p, c->nccl, g->stream);
    (void)R->Recv(d_cnt + W + p, 8, ncclUint8, p, c->nccl, g->stream);
  }
  {
    const int r = R->GroupEnd();
    if (r != ncclSuccess) {
      set_error("RCCL size exchange failed: %s", R->GetErrorString(r));
      return done(PPRHIP_ERR_HIP);
    }
  }
  if (hipMemcpyAsync(h_cnt.data() + W, d_cnt + W, sizeof(unsigned long long) * W, hipMemcpyDeviceToHost, g->stream) !=
          hipSuccess ||
      hipStreamSynchronize(g->stream) != hipSuccess)
    return done(PPRHIP_ERR_HIP);
  for (int p = 0; p < W; ++p) recv_off[p + 1] = recv_off[p] + h_cnt[W + p];
  int rc = alloc_dev(recv, recv_off[W]);
  if (rc != PPRHIP_OK) return done(rc);
  // 2) the payload: one message per peer, all of them in flight together (each pair of GPUs has its own xGMI link)
  if (R->GroupStart() != ncclSuccess) return done(PPRHIP_ERR_HIP);
  for (int p = 0; p < W; ++p) {
    const uint64_t sb = off[p + 1] - off[p], rb = recv_off[p + 1] - recv_off[p];
    if (sb) (void)R->Send((const char*)send + off[p], sb, ncclUint8, p, c->nccl, g->stream);
    if (rb) (void)R->Recv((char*)*recv + recv_off[p], rb, ncclUint8, p, c->nccl, g->stream);
  }
  {
    const int r = R->GroupEnd();
    if (r != ncclSuccess) {
      set_error("RCCL payload exchange failed: %s", R->GetErrorString(r));
      return done(PPRHIP_ERR_HIP);
    }
  }
  if (hipStreamSynchronize(g->stream) != hipSuccess) return done(PPRHIP_ERR_HIP);
  return done(PPRHIP_OK);
}

void target_range(int rank, int world, uint32_t n, uint32_t* lo, uint32_t* hi) {
  const uint32_t base = n / (uint32_t)world, rem = n % (uint32_t)world;
  *lo = (uint32_t)rank * base + std::min<uint32_t>((uint32_t)rank, rem);
  *hi = *lo + base + ((uint32_t)rank < rem ? 1u : 0u);
}

// this rank's share of the sharded All-Pair: search its targets, exchange by owner of the source, finalise its sources
int all_pair_sharded(pprhip_comm* c, double alpha, double threshold, int k, pprhip_index_t** own_out,
                     pprhip_stats_t* stats) {
  pprhip_graph* g = c->g;
  const int W = c->world;
  if (g->n < (uint32_t)W) {
    set_error("sharded All-Pair: fewer nodes (%u) than ranks (%d)", g->n, W);
    return PPRHIP_ERR_INVALID;
  }
  uint32_t lo, hi;
  target_range(c->rank, W, g->n, &lo, &hi);
  pprhip_stats_t st;
  std::memset(&st, 0, sizeof st);
  DeviceTripleSink sink;
  PPRHIP_TRY(all_pair_collect(g, alpha, threshold, lo, hi, sink, st));
  // ---- partition by owner of the source, on the device
  unsigned long long* d_cur = nullptr;
  TripleRec* d_part = nullptr;
  void* d_recv = nullptr;
  auto done = [&](int rc) {
    if (d_cur) (void)hipFree(d_cur);
    if (d_part) (void)hipFree(d_part);
    if (d_recv) (void)hipFree(d_recv);
    return rc;
  };
  int rc = alloc_dev((void**)&d_cur, sizeof(unsigned long long) * kBatch * 4);  // >= 64 counters
  if (rc) return done(rc);
  if (hipMemsetAsync(d_cur, 0, sizeof(unsigned long long) * 64, g->stream) != hipSuccess) return done(PPRHIP_ERR_HIP);
  if ((rc = launch_owner_partition(g, sink.rec, sink.count, W, d_cur, nullptr))) return done(rc);
  std::vector<unsigned long long> cnt((size_t)W, 0), start((size_t)W + 1, 0);
  if (hipMemcpyAsync(cnt.data(), d_cur, sizeof(unsigned long long) * W, hipMemcpyDeviceToHost, g->stream) != hipSuccess ||
      hipStreamSynchronize(g->stream) != hipSuccess)
    return done(PPRHIP_ERR_HIP);
  for (int p = 0; p < W; ++p) start[p + 1] = start[p] + cnt[p];
  if ((rc = alloc_dev((void**)&d_part, sizeof(TripleRec) * std::max<unsigned long long>(1, sink.count)))) return done(rc);
  if (hipMemcpyAsync(d_cur, start.data(), sizeof(unsigned long long) * W, hipMemcpyHostToDevice, g->stream) != hipSuccess)
    return done(PPRHIP_ERR_HIP);
  if ((rc = launch_owner_partition(g, sink.rec, sink.count, W, d_cur, d_part))) return done(rc);
  if (hipStreamSynchronize(g->stream) != hipSuccess) return done(PPRHIP_ERR_HIP);
  // ---- exchange: every entry goes to the rank that owns its source
  std::vector<uint64_t> off((size_t)W + 1), roff;
  for (int p = 0; p <= W; ++p) off[p] = start[p] * sizeof(TripleRec);
  if ((rc = comm_alltoallv(c, d_part, off, &d_recv, roff))) return done(rc);
  // ---- the entries of this rank's sources cross PCIe once, here
  const uint64_t n_recv = roff[W] / sizeof(TripleRec);
  std::vector<Triple> tr(n_recv);
  if (n_recv && (hipMemcpyAsync(tr.data(), d_recv, roff[W], hipMemcpyDeviceToHost, g->stream) != hipSuccess ||
                 hipStreamSynchronize(g->stream) != hipSuccess))
    return done(PPRHIP_ERR_HIP);
  st.select_bytes = roff[W];          // bytes received in the exchange
  st.mc_sources = sink.count;         // entries this rank found (before the exchange)
  st.enqueues += 0;
  if ((rc = index_from_triples(g->n, tr, k, own_out))) return done(rc);
  if (stats) *stats = st;
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
  c->rank = rank;
  c->world = world;
  ncclUniqueId uid;
  std::memcpy(&uid, id, sizeof uid);
  PPRHIP_CHECK_RCCL(R->CommInitRank(&c->nccl, world, uid, rank));
  *comm_out = c.release();
  return PPRHIP_OK;
}

void pprhip_comm_destroy(pprhip_comm_t* c) {
  if (!c) return;
  if (c->nccl) {
    (void)hipSetDevice(c->g->device);
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
  pprhip_graph* g = c->g;
  PPRHIP_TRY(check_graph(g, "pprhip_topk_gather"));
  // one block per rank: rows_max rows of k (id, value) pairs, ids first; short blocks are padded with id -1
  const size_t blk_ids = sizeof(int32_t) * (size_t)rows_max * k, blk_vals = sizeof(double) * (size_t)rows_max * k;
  const size_t blk = blk_ids + blk_vals;
  std::vector<char> h(blk);
  int32_t* hi = (int32_t*)h.data();
  double* hv = (double*)(h.data() + blk_ids);
  for (size_t i = 0; i < (size_t)rows_max * k; ++i) {
    hi[i] = i < (size_t)rows * k ? ids[i] : -1;
    hv[i] = i < (size_t)rows * k ? vals[i] : 0.0;
  }
  void* d_send = nullptr;
  PPRHIP_TRY(alloc_dev(&d_send, blk));
  void* d_recv = nullptr;
  auto done = [&](int rc) {
    (void)hipFree(d_send);
    if (d_recv) (void)hipFree(d_recv);
    return rc;
  };
  if (hipMemcpyAsync(d_send, h.data(), blk, hipMemcpyHostToDevice, g->stream) != hipSuccess) return done(PPRHIP_ERR_HIP);
  std::vector<uint64_t> off((size_t)c->world + 1, 0), roff;
  for (int p = 0; p < c->world; ++p) off[p + 1] = off[p] + (p == 0 ? blk : 0);  // everything goes to rank 0
  if (hipStreamSynchronize(g->stream) != hipSuccess) return done(PPRHIP_ERR_HIP);
  int rc = comm_alltoallv(c, d_send, off, &d_recv, roff);
  if (rc) return done(rc);
  if (c->rank == 0) {
    std::vector<char> all(roff[c->world]);
    if (hipMemcpyAsync(all.data(), d_recv, all.size(), hipMemcpyDeviceToHost, g->stream) != hipSuccess ||
        hipStreamSynchronize(g->stream) != hipSuccess)
      return done(PPRHIP_ERR_HIP);
    for (int p = 0; p < c->world; ++p) {
      std::memcpy(ids_root + (size_t)p * rows_max * k, all.data() + roff[p], blk_ids);
      std::memcpy(vals_root + (size_t)p * rows_max * k, all.data() + roff[p] + blk_ids, blk_vals);
    }
  }
  return done(PPRHIP_OK);
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
  S.use_rccl = distinct && n_gpu > 1;
  S.local.world = n_gpu;
  S.local.send.assign((size_t)n_gpu, nullptr);
  S.local.send_off.assign((size_t)n_gpu, {});
  for (int r = 0; r < n_gpu; ++r) {
    S.comms[r].g = per_gpu[r];
    S.comms[r].rank = r;
    S.comms[r].world = n_gpu;
    S.comms[r].local = S.use_rccl ? nullptr : &S.local;
  }
  return PPRHIP_OK;
}

// one host thread per GPU; fn(rank) returns a code, messages are carried back to the caller's thread
template <class F>
int run_ranks(RankSetup& S, F fn) {
  const int W = (int)S.comms.size();
  ncclUniqueId uid;
  if (S.use_rccl) {
    RcclApi* R = rccl();
    if (!R) {
      set_error("librccl.so.1 could not be loaded");
      return PPRHIP_ERR_NO_DEVICE;
    }
    PPRHIP_CHECK_RCCL(R->GetUniqueId(&uid));
  }
  std::vector<std::thread> th;
  for (int r = 0; r < W; ++r)
    th.emplace_back([&, r] {
      int rc = PPRHIP_OK;
      if (hipSetDevice(S.comms[r].g->device) != hipSuccess) rc = PPRHIP_ERR_NO_DEVICE;
      if (rc == PPRHIP_OK && S.use_rccl) {
        const int e = rccl()->CommInitRank(&S.comms[r].nccl, W, uid, r);
        if (e != ncclSuccess) {
          set_error("ncclCommInitRank failed on rank %d: %s", r, rccl()->GetErrorString(e));
          rc = PPRHIP_ERR_HIP;
        }
      }
      if (rc == PPRHIP_OK) rc = fn(r);
      if (rc != PPRHIP_OK) S.errs[r] = get_error();
      S.rcs[r] = rc;
      if (S.comms[r].nccl) {
        (void)rccl()->CommDestroy(S.comms[r].nccl);
        S.comms[r].nccl = nullptr;
      }
    });
  for (auto& t : th) t.join();
  for (int r = 0; r < W; ++r)
    if (S.rcs[r] != PPRHIP_OK) {
      set_error("GPU %d (device %d): %s", r, S.comms[r].g->device, S.errs[r].c_str());
      return S.rcs[r];
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
  const int rc = run_ranks(S, [&](int r) -> int {
    // query i runs on GPU i mod W (local row i / W)
    std::vector<int32_t> mine;
    for (int i = r; i < q; i += W) mine.push_back(srcs[i]);
    const int rows = (int)mine.size();
    std::vector<int32_t> ids((size_t)std::max(1, rows) * k);
    std::vector<double> vals((size_t)std::max(1, rows) * k);
    nsel[r].assign((size_t)std::max(1, rows), 0);
    pprhip_stats_t st;
    std::memset(&st, 0, sizeof st);
    int e = pprhip_fora_batch_single_source(S.comms[r].g, mine.data(), rows, eps, conf, seed, n_rounds, nullptr, k,
                                            ids.data(), vals.data(), nsel[r].data(), nullptr, &st);
    if (stats_per_gpu) stats_per_gpu[r] = st;
    // the only exchange of the path: the top-k blocks travel to GPU 0 over the fabric
    if (e == PPRHIP_OK)
      e = pprhip_topk_gather(&S.comms[r], ids.data(), vals.data(), rows, rows_max, k, ids_root.data(), vals_root.data());
    if (e != PPRHIP_OK && S.comms[r].local) S.local.abort(e);
    return e;
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
  int rc = run_ranks(S, [&](int r) -> int {
    pprhip_stats_t st;
    std::memset(&st, 0, sizeof st);
    const int e = all_pair_sharded(&S.comms[r], alpha, threshold, k, &own[r], &st);
    if (e != PPRHIP_OK && S.comms[r].local) S.local.abort(e);  // the other ranks may wait at the exchange
    if (stats_per_gpu) stats_per_gpu[r] = st;
    return e;
  });
  if (rc == PPRHIP_OK) rc = index_concat(own, index_out);  // every rank holds the rows of its own sources
  for (pprhip_index_t* p : own) pprhip_index_destroy(p);
  return rc;
}

}  // extern "C"
