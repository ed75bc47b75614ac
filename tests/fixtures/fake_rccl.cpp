// fake_rccl.cpp — TEST DOUBLE of the dozen RCCL entry points csrc/comm.cpp binds (tests/test_gpu_multi.py builds it and
// points PPRHIP_RCCL_LIB at it in a child process).  It is NOT RCCL and proves nothing about the fabric: ranks are
// threads of one process whose graph replicas share the one GPU of the test box (which real RCCL refuses), and a
// send / receive pair is a device-to-device copy made when both sides have reached ncclGroupEnd.  What it does give
// is an execution of comm.cpp's RCCL branch with more than one rank: the size exchange with its error sentinel, the
// payload group with self send / receive, offsets, the abort path - the code that no multi-GPU box has run yet.
//
// Semantics kept from NCCL: operations between ncclGroupStart / ncclGroupEnd are matched as a set (a rank may post
// its receives before the peer posts the sends), sends and receives between a pair are matched in order, byte counts
// must agree, ncclCommAbort makes every pending and later operation of the group fail.  Unlike NCCL, ncclGroupEnd
// returns only when this rank's operations have completed (so stream semantics hold trivially).
#include <hip/hip_runtime.h>

#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <random>
#include <vector>

namespace {

enum { kOk = 0, kUnhandled = 1, kSystem = 2, kInternal = 3, kInvalidArg = 4, kInvalidUsage = 5, kRemote = 6, kInProgress = 7 };

struct SendOp {
  const void* buf;
  size_t bytes;
  unsigned long long ticket;
};

struct Group {
  std::mutex mu;
  std::condition_variable cv;
  int world = 0;
  int refs = 0;
  bool aborted = false;
  std::map<std::pair<int, int>, std::deque<SendOp>> posted;           // (src, dst) -> sends not yet taken
  std::map<std::pair<int, int>, unsigned long long> next_ticket, done;  // per (src, dst)
};

struct Comm {
  Group* grp;
  int rank;
};

struct Op {
  bool is_send;
  void* buf;
  size_t bytes;
  int peer;
  Comm* comm;
  hipStream_t stream;
};

std::mutex g_mu;
std::map<std::string, Group*> g_groups;  // by unique id
thread_local std::vector<Op> tl_ops;
thread_local int tl_depth = 0;

double wait_limit_s() {
  const char* e = getenv("FAKE_RCCL_TIMEOUT_S");
  return e ? atof(e) : 30.0;
}

int run_ops(std::vector<Op>& ops) {
  int rc = kOk;
  // 1) post the sends (their data is complete once the stream they were queued on has drained)
  std::vector<std::pair<std::pair<int, int>, unsigned long long>> mine;
  for (Op& o : ops)
    if (o.is_send) {
      if (hipStreamSynchronize(o.stream) != hipSuccess) return kUnhandled;
      Group* G = o.comm->grp;
      std::lock_guard<std::mutex> lk(G->mu);
      if (G->aborted) return kInternal;
      const auto key = std::make_pair(o.comm->rank, o.peer);
      const unsigned long long t = ++G->next_ticket[key];
      G->posted[key].push_back(SendOp{o.buf, o.bytes, t});
      mine.push_back({key, t});
      G->cv.notify_all();
    }
  // 2) take what the peers have sent
  const auto deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(wait_limit_s());
  for (Op& o : ops)
    if (!o.is_send) {
      Group* G = o.comm->grp;
      const auto key = std::make_pair(o.peer, o.comm->rank);
      SendOp s{};
      {
        std::unique_lock<std::mutex> lk(G->mu);
        if (!G->cv.wait_until(lk, deadline, [&] { return G->aborted || !G->posted[key].empty(); })) return kRemote;
        if (G->aborted) return kInternal;
        s = G->posted[key].front();
        G->posted[key].pop_front();
      }
      if (s.bytes != o.bytes) rc = kInvalidArg;
      else if (o.bytes && (hipMemcpyAsync(o.buf, s.buf, o.bytes, hipMemcpyDeviceToDevice, o.stream) != hipSuccess ||
                           hipStreamSynchronize(o.stream) != hipSuccess))
        rc = kUnhandled;
      // FAKE_RCCL_CORRUPT_SRC=<id> (test switch): a payload of 16-byte index records (not an 8-byte size word) arrives
      // with the source id of its MIDDLE record replaced - what a transport gone wrong or a peer's bad partition would
      // hand the receiver, whose finalisation must turn it into an error and not into an out-of-range write
      if (rc == kOk && o.bytes >= 48 && o.bytes % 16 == 0)
        if (const char* bad = getenv("FAKE_RCCL_CORRUPT_SRC")) {
          const int v = atoi(bad);
          (void)hipMemcpy((char*)o.buf + (o.bytes / 16 / 2) * 16, &v, sizeof v, hipMemcpyHostToDevice);
        }
      {
        std::lock_guard<std::mutex> lk(G->mu);
        G->done[key] = s.ticket;
        G->cv.notify_all();
      }
      if (rc != kOk) return rc;
    }
  // 3) my sends have been read (their buffers may be reused after the group, as after a drained stream)
  for (auto& m : mine) {
    Group* G = ops[0].comm->grp;
    std::unique_lock<std::mutex> lk(G->mu);
    if (!G->cv.wait_until(lk, deadline, [&] { return G->aborted || G->done[m.first] >= m.second; })) return kRemote;
    if (G->aborted) return kInternal;
  }
  return rc;
}

}  // namespace

extern "C" {

struct ncclUniqueId {
  char internal[128];
};
typedef Comm* ncclComm_t;

int ncclGetUniqueId(ncclUniqueId* id) {
  static std::mt19937_64 rng(std::random_device{}());
  std::lock_guard<std::mutex> lk(g_mu);
  std::memset(id->internal, 0, sizeof id->internal);
  for (int i = 0; i < 4; ++i) {
    const unsigned long long x = rng();
    std::memcpy(id->internal + 8 * i, &x, 8);
  }
  return kOk;
}

int ncclCommInitRank(ncclComm_t* comm, int world, ncclUniqueId id, int rank) {
  if (!comm || world < 1 || rank < 0 || rank >= world) return kInvalidArg;
  std::lock_guard<std::mutex> lk(g_mu);
  const std::string key(id.internal, sizeof id.internal);
  Group*& G = g_groups[key];
  if (!G) {
    G = new Group();
    G->world = world;
  }
  if (G->world != world) return kInvalidArg;
  G->refs++;
  *comm = new Comm{G, rank};
  return kOk;
}

int ncclCommInitAll(ncclComm_t* comms, int n, const int* /*devices*/) {
  if (!comms || n < 1) return kInvalidArg;
  Group* G = new Group();
  G->world = n;
  G->refs = n;
  for (int r = 0; r < n; ++r) comms[r] = new Comm{G, r};
  return kOk;
}

static void drop(Comm* c) {
  Group* G = c->grp;
  bool last = false;
  {
    std::lock_guard<std::mutex> lk(G->mu);
    last = --G->refs == 0;
  }
  delete c;
  if (last) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto it = g_groups.begin(); it != g_groups.end(); ++it)
      if (it->second == G) {
        g_groups.erase(it);
        break;
      }
    delete G;
  }
}

int ncclCommDestroy(ncclComm_t c) {
  if (c) drop(c);
  return kOk;
}

int ncclCommAbort(ncclComm_t c) {
  if (!c) return kOk;
  {
    std::lock_guard<std::mutex> lk(c->grp->mu);
    c->grp->aborted = true;
    c->grp->cv.notify_all();
  }
  drop(c);
  return kOk;
}

int ncclCommGetAsyncError(ncclComm_t c, int* e) {
  if (!c || !e) return kInvalidArg;
  std::lock_guard<std::mutex> lk(c->grp->mu);
  *e = c->grp->aborted ? kInternal : kOk;
  return kOk;
}

int ncclGroupStart() {
  tl_depth++;
  return kOk;
}

int ncclGroupEnd() {
  if (tl_depth <= 0) return kInvalidUsage;
  if (--tl_depth > 0) return kOk;
  std::vector<Op> ops;
  ops.swap(tl_ops);
  return ops.empty() ? kOk : run_ops(ops);
}

static int queue(bool is_send, void* buf, size_t count, int type, int peer, ncclComm_t c, hipStream_t s) {
  if (!c || peer < 0 || peer >= c->grp->world || type != 1 /* ncclUint8: all comm.cpp uses */) return kInvalidArg;
  tl_ops.push_back(Op{is_send, buf, count, peer, c, s});
  if (tl_depth == 0) {
    std::vector<Op> ops;
    ops.swap(tl_ops);
    return run_ops(ops);
  }
  return kOk;
}

int ncclSend(const void* buf, size_t count, int type, int peer, ncclComm_t c, hipStream_t s) {
  return queue(true, const_cast<void*>(buf), count, type, peer, c, s);
}

int ncclRecv(void* buf, size_t count, int type, int peer, ncclComm_t c, hipStream_t s) {
  return queue(false, buf, count, type, peer, c, s);
}

const char* ncclGetErrorString(int e) {
  switch (e) {
    case kOk: return "no error (test double)";
    case kInternal: return "group aborted (test double)";
    case kInvalidArg: return "invalid argument (test double)";
    case kInvalidUsage: return "invalid usage (test double)";
    case kRemote: return "a peer did not arrive in time (test double)";
    default: return "error (test double)";
  }
}

}  // extern "C"
