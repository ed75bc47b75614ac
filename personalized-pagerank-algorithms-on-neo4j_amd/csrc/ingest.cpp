// ingest.cpp — host-side graph ingest: seeded R-MAT generator, neo4j-admin-import CSV reader and
// CSR construction.  This is the data-format side of the graph lift (PPR.java:136-152 loads the
// Neo4j store into HeavyGraph's two jagged adjacency arrays); nothing here touches the GPU.
#include <algorithm>
#include <atomic>
#include <cstring>
#include <memory>
#include <fstream>
#include <iterator>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include <new>

#include "common.hpp"

namespace pprhip {
namespace detail {
unsigned host_threads();  // lift.cpp: what the process may use (CPU affinity, cgroup quota; PPRHIP_HOST_THREADS)
}

static thread_local std::string g_error;

void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_error = buf;
}
const char* get_error() { return g_error.c_str(); }

static inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

static unsigned worker_count() { return std::min(detail::host_threads(), 32u); }

}  // namespace pprhip

using namespace pprhip;

struct pprhip_edgelist {
  uint32_t n = 0;
  std::vector<int32_t> src, dst;
  std::vector<std::string> names;
  // adjacency in relationship-chain order when the list came from a Neo4j store
  bool from_store = false;
  std::vector<uint32_t> out_rp, in_rp;
  std::vector<int32_t> out_ci, in_ci;
};

static bool read_file(const std::string& path, std::vector<unsigned char>& buf) {
  FILE* f = fopen(path.c_str(), "rb");  // (one read of the whole file: store files run to gigabytes)
  if (!f) return false;
  bool ok = fseek(f, 0, SEEK_END) == 0;
  const long size = ok ? ftell(f) : -1;
  ok = ok && size >= 0 && fseek(f, 0, SEEK_SET) == 0;
  if (ok) {
    buf.resize((size_t)size);
    ok = size == 0 || fread(buf.data(), 1, (size_t)size, f) == (size_t)size;
  }
  fclose(f);
  return ok;
}
static inline uint32_t be32(const unsigned char* p) {
  return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3];
}

extern "C" {

const char* pprhip_last_error(void) { return get_error(); }
int pprhip_version(void) { return PPRHIP_VERSION; }

int pprhip_rmat_edges(int scale, int edge_factor, uint64_t seed, int32_t* src_out, int32_t* dst_out) {
  if (scale < 1 || scale > 30 || edge_factor < 1 || !src_out || !dst_out) {
    set_error("pprhip_rmat_edges: bad arguments (scale=%d edge_factor=%d)", scale, edge_factor);
    return PPRHIP_ERR_INVALID;
  }
  const uint64_t n = 1ull << scale;
  const uint64_t m = (uint64_t)edge_factor << scale;
  // seeded label scramble (Fisher-Yates on a splitmix64 stream)
  std::vector<int32_t> perm(n);
  for (uint64_t i = 0; i < n; ++i) perm[i] = (int32_t)i;
  uint64_t st = splitmix64(seed ^ 0x5851F42D4C957F2Dull);
  for (uint64_t i = n - 1; i > 0; --i) {
    st = splitmix64(st);
    uint64_t j = (uint64_t)(((unsigned __int128)st * (i + 1)) >> 64);
    std::swap(perm[i], perm[j]);
  }
  // Graph500 quadrant probabilities as 32-bit thresholds
  const uint32_t ta = (uint32_t)(0.57 * 4294967296.0);
  const uint32_t tab = (uint32_t)((0.57 + 0.19) * 4294967296.0);
  const uint32_t tabc = (uint32_t)((0.57 + 0.19 + 0.19) * 4294967296.0);
  const uint64_t key = splitmix64(seed);
  auto work = [&](uint64_t lo, uint64_t hi) {
    for (uint64_t e = lo; e < hi; ++e) {
      uint32_t u = 0, v = 0;
      uint64_t word = 0;
      for (int l = 0; l < scale; ++l) {
        if ((l & 1) == 0) word = splitmix64(key ^ (e * 32ull + (uint64_t)(l >> 1)) * 0xD1342543DE82EF95ull);
        uint32_t r = (l & 1) ? (uint32_t)(word >> 32) : (uint32_t)word;
        uint32_t ub = r >= tab;                             // quadrants c, d: source bit set
        uint32_t vb = (r >= ta && r < tab) || (r >= tabc);  // quadrants b, d: destination bit set
        u = (u << 1) | ub;
        v = (v << 1) | vb;
      }
      src_out[e] = perm[u];
      dst_out[e] = perm[v];
    }
  };
  unsigned nt = worker_count();
  std::vector<std::thread> th;
  uint64_t chunk = (m + nt - 1) / nt;
  for (unsigned t = 0; t < nt; ++t) {
    uint64_t lo = t * chunk, hi = std::min(m, lo + chunk);
    if (lo < hi) th.emplace_back(work, lo, hi);
  }
  for (auto& t : th) t.join();
  return PPRHIP_OK;
}

static bool read_lines(const char* path, std::vector<std::string>& lines) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  std::string s;
  while (std::getline(f, s)) {
    while (!s.empty() && (s.back() == '\r' || s.back() == '\n')) s.pop_back();
    lines.push_back(s);
  }
  if (!lines.empty() && lines[0].size() >= 3 && (unsigned char)lines[0][0] == 0xEF &&
      (unsigned char)lines[0][1] == 0xBB && (unsigned char)lines[0][2] == 0xBF)
    lines[0].erase(0, 3);  // UTF-8 byte-order mark (GOT_Nodes.csv has one)
  return true;
}

static std::string field(const std::string& line, int idx) {
  size_t b = 0;
  for (int i = 0; i < idx; ++i) {
    b = line.find(',', b);
    if (b == std::string::npos) return std::string();
    ++b;
  }
  size_t e = line.find(',', b);
  return line.substr(b, e == std::string::npos ? std::string::npos : e - b);
}

int pprhip_edgelist_from_neo4j_csv(const char* nodes_csv, const char* rels_csv, pprhip_edgelist_t** out) {
  if (!nodes_csv || !rels_csv || !out) {
    set_error("pprhip_edgelist_from_neo4j_csv: null argument");
    return PPRHIP_ERR_INVALID;
  }
  std::vector<std::string> nl, rl;
  if (!read_lines(nodes_csv, nl)) {
    set_error("cannot read %s", nodes_csv);
    return PPRHIP_ERR_IO;
  }
  if (!read_lines(rels_csv, rl)) {
    set_error("cannot read %s", rels_csv);
    return PPRHIP_ERR_IO;
  }
  if (nl.empty() || nl[0].find(":ID") == std::string::npos) {
    set_error("%s: expected a ':ID,...' header", nodes_csv);
    return PPRHIP_ERR_IO;
  }
  if (rl.empty() || rl[0].find(":START_ID") == std::string::npos) {
    set_error("%s: expected a ':START_ID,:END_ID,...' header", rels_csv);
    return PPRHIP_ERR_IO;
  }
  auto* e = new pprhip_edgelist();
  std::unordered_map<std::string, int32_t> ids;
  for (size_t i = 1; i < nl.size(); ++i) {
    if (nl[i].empty()) continue;
    std::string id = field(nl[i], 0);
    std::string name = field(nl[i], 1);
    if (ids.count(id)) {
      set_error("%s: duplicate node id '%s'", nodes_csv, id.c_str());
      delete e;
      return PPRHIP_ERR_IO;
    }
    ids[id] = (int32_t)e->names.size();  // node id = row index, as neo4j-admin import assigns them
    e->names.push_back(name.empty() ? id : name);
  }
  e->n = (uint32_t)e->names.size();
  for (size_t i = 1; i < rl.size(); ++i) {
    if (rl[i].empty()) continue;
    auto a = ids.find(field(rl[i], 0));
    auto b = ids.find(field(rl[i], 1));
    if (a == ids.end() || b == ids.end()) {
      set_error("%s line %zu: unknown node id", rels_csv, i + 1);
      delete e;
      return PPRHIP_ERR_IO;
    }
    e->src.push_back(a->second);
    e->dst.push_back(b->second);
  }
  *out = e;
  return PPRHIP_OK;
}

// A Neo4j 3.x store ("standard" record format, big-endian) read without a JVM.  Node records are 15 bytes {in use +
// high bits, nextRel u32, nextProp u32, labels 5, extra: dense flag}, relationship records 34 bytes {in use + high
// bits, first node u32, second node u32, type + high bits u32, first prev / next u32, second prev / next u32, nextProp
// u32, extra}; a node's relationships form a chain through the "next" pointer of whichever end the node is.  A dense
// node (50 relationships or more by default - the threshold is the group store's header, 0x32 in got.db's) points to a
// chain of 25-byte relationship groups instead, one per relationship type, each with the heads of three chains -
// outgoing, incoming, loops (RelationshipGroupRecordFormat: header byte {in use, high bits of next and firstOut}, high
// byte {firstIn, firstLoop}, type u16, next, firstOut, firstIn, firstLoop u32, owning node u32 + u8; record 0 holds the
// store header).  Adjacency lists come out in chain order, as HeavyGraph's loader lists them; for a dense node group by
// group: the outgoing chain, the incoming chain, then the loops (both directions).
// Every node's chains are independent of every other node's, so the nodes are walked on all host threads: once to count
// (row pointers), once to fill (the reference's own load takes 9 s for 11.8 M relationships, Diss. Table 27).
}  // extern "C"

namespace {

constexpr size_t kNodeRec = 15, kRelRec = 34, kGroupRec = 25;
constexpr uint64_t kNoRel = 0x7FFFFFFFFull;  // all 35 id bits set = "no relationship"

struct StoreView {
  const char* dir = "";
  std::vector<unsigned char> nodes, rels, groups;
  bool have_groups = false;
  uint64_t n = 0, nrel = 0, ngroup = 0, m = 0;
};

struct RelRec {
  bool in_use;
  uint64_t first, second, first_next, second_next;
};

inline RelRec rel_at(const StoreView& S, uint64_t id) {
  const unsigned char* x = &S.rels[id * kRelRec];
  const uint32_t type_int = be32(x + 9);
  RelRec r;
  r.in_use = x[0] & 1;
  r.first = be32(x + 1) | ((uint64_t)(x[0] & 0x0E) << 31);
  r.second = be32(x + 5) | ((uint64_t)(type_int & 0x70000000u) << 4);
  r.first_next = be32(x + 17) | ((uint64_t)(type_int & 0x01C00000u) << 10);
  r.second_next = be32(x + 25) | ((uint64_t)(type_int & 0x00070000u) << 16);
  return r;
}

inline bool no_rel(uint64_t r) { return r == kNoRel || r == 0xFFFFFFFFull; }

std::string fmt(const char* f, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, f);
  vsnprintf(buf, sizeof buf, f, ap);
  va_end(ap);
  return buf;
}

// Calls emit(outgoing, neighbour) for every relationship end of node v in chain order; false + *err on a broken store.
template <class Emit>
bool walk_node(const StoreView& S, uint64_t v, Emit&& emit, std::string* err) {
  const unsigned char* x = &S.nodes[v * kNodeRec];
  if (!(x[0] & 1)) return true;
  uint64_t r = be32(x + 1) | ((uint64_t)(x[0] & 0x0E) << 31);
  if (!(x[14] & 1)) {
    for (uint64_t steps = 0; !no_rel(r); ++steps) {
      if (r >= S.nrel || steps > S.m) {
        *err = fmt("%s: broken relationship chain at node %llu", S.dir, (unsigned long long)v);
        return false;
      }
      const RelRec rr = rel_at(S, r);
      if (rr.first == v) {
        emit(true, rr.second);
        if (rr.second == v) emit(false, v);  // self loop: one record, both directions
        r = rr.first_next;
      } else if (rr.second == v) {
        emit(false, rr.first);
        r = rr.second_next;
      } else {
        *err = fmt("%s: relationship %llu is on node %llu's chain but does not touch it", S.dir, (unsigned long long)r,
                   (unsigned long long)v);
        return false;
      }
    }
    return true;
  }
  if (!S.have_groups) {
    *err = fmt("%s: node %llu is dense and neostore.relationshipgroupstore.db cannot be read", S.dir, (unsigned long long)v);
    return false;
  }
  auto mod = [](uint32_t low, uint64_t high) {  // BaseRecordFormat.longFromIntAndMod: all ones without high bits = none
    return (high == 0 && low == 0xFFFFFFFFu) ? kNoRel : ((uint64_t)low | high);
  };
  uint64_t gsteps = 0;
  for (uint64_t gid = no_rel(r) ? kNoRel : r; gid != kNoRel; ++gsteps) {
    if (gid == 0 || gid >= S.ngroup || gsteps > S.ngroup) {
      *err = fmt("%s: broken relationship group chain at node %llu", S.dir, (unsigned long long)v);
      return false;
    }
    const unsigned char* gx = &S.groups[gid * kGroupRec];
    const uint64_t owner = be32(gx + 20) | ((uint64_t)gx[24] << 32);
    if (!(gx[0] & 1) || owner != v) {
      *err = fmt("%s: relationship group %llu on node %llu's chain is unused or belongs to node %llu", S.dir,
                 (unsigned long long)gid, (unsigned long long)v, (unsigned long long)owner);
      return false;
    }
    const uint64_t next = mod(be32(gx + 4), (uint64_t)(gx[0] & 0x0E) << 31);
    const uint64_t heads[3] = {mod(be32(gx + 8), (uint64_t)(gx[0] & 0x70) << 28),    // outgoing
                               mod(be32(gx + 12), (uint64_t)(gx[1] & 0x0E) << 31),   // incoming
                               mod(be32(gx + 16), (uint64_t)(gx[1] & 0x70) << 28)};  // loops
    for (int d3 = 0; d3 < 3; ++d3) {
      uint64_t steps = 0;
      for (uint64_t q = heads[d3]; !no_rel(q); ++steps) {
        if (q >= S.nrel || steps > S.m) {
          *err = fmt("%s: broken relationship chain at dense node %llu", S.dir, (unsigned long long)v);
          return false;
        }
        const RelRec rr = rel_at(S, q);
        const bool ok = rr.in_use && (d3 == 0 ? (rr.first == v && rr.second != v)
                                               : d3 == 1 ? (rr.second == v && rr.first != v) : (rr.first == v && rr.second == v));
        if (!ok) {
          *err = fmt("%s: relationship %llu is on the wrong chain of dense node %llu", S.dir, (unsigned long long)q,
                     (unsigned long long)v);
          return false;
        }
        if (d3 != 1) emit(true, rr.second);
        if (d3 != 0) emit(false, rr.first);
        q = d3 == 1 ? rr.second_next : rr.first_next;
      }
    }
    gid = next;
  }
  return true;
}

// fn(part) for part in [0, parts) on T threads; false when a part reported an error (the lowest part's message wins)
template <class F>
bool store_parts(unsigned parts, unsigned T, std::vector<std::string>& errs, F&& fn) {
  errs.assign(parts, std::string());
  std::atomic<unsigned> next{0};
  auto work = [&]() {
    for (unsigned p = next.fetch_add(1); p < parts; p = next.fetch_add(1)) {
      try {
        fn(p, &errs[p]);
      } catch (const std::bad_alloc&) {
        errs[p] = "out of host memory";
      }
    }
  };
  std::vector<std::thread> th;
  for (unsigned w = 1; w < std::min(T, parts); ++w) th.emplace_back(work);
  work();
  for (auto& x : th) x.join();
  for (const std::string& e : errs)
    if (!e.empty()) return false;
  return true;
}

}  // namespace

extern "C" {

int pprhip_edgelist_from_neo4j_store(const char* store_dir, pprhip_edgelist_t** out) {
  if (!store_dir || !out) {
    set_error("pprhip_edgelist_from_neo4j_store: null argument");
    return PPRHIP_ERR_INVALID;
  }
  const std::string dir(store_dir);
  StoreView S;
  S.dir = store_dir;
  std::vector<unsigned char> nid;
  if (!read_file(dir + "/neostore.nodestore.db", S.nodes) || !read_file(dir + "/neostore.relationshipstore.db", S.rels)) {
    set_error("cannot read neostore.nodestore.db / neostore.relationshipstore.db under %s", store_dir);
    return PPRHIP_ERR_IO;
  }
  S.have_groups = read_file(dir + "/neostore.relationshipgroupstore.db", S.groups);  // (only dense nodes need it)
  S.ngroup = S.groups.size() / kGroupRec;
  uint64_t n = S.nodes.size() / kNodeRec;
  if (read_file(dir + "/neostore.nodestore.db.id", nid) && nid.size() >= 9) {  // 1 flag byte + 8-byte high id
    uint64_t hi = 0;
    for (int i = 1; i <= 8; ++i) hi = (hi << 8) | nid[i];
    if (hi <= n) n = hi;
  }
  while (n > 0 && !(S.nodes[(n - 1) * kNodeRec] & 1)) --n;  // trailing unused records
  S.n = n;
  S.nrel = S.rels.size() / kRelRec;
  if (n == 0 || n >= (1ull << 28) || S.nrel >= (1ull << 32) - 1024) {
    set_error("%s: %llu node records, %llu relationship records (unsupported)", store_dir, (unsigned long long)n,
              (unsigned long long)S.nrel);
    return PPRHIP_ERR_IO;
  }
  std::unique_ptr<pprhip_edgelist> e(new (std::nothrow) pprhip_edgelist());
  if (!e) return PPRHIP_ERR_OOM;
  e->n = (uint32_t)n;
  e->from_store = true;
  const unsigned T = (n + S.nrel < (1u << 15)) ? 1u : detail::host_threads();
  const unsigned parts = T == 1 ? 1u : T * 8u;
  std::vector<std::string> errs;
  auto fail = [&](int code) {
    for (const std::string& m : errs)
      if (!m.empty()) {
        set_error("%s", m.c_str());
        break;
      }
    return code;
  };
  try {
    // ---- relationships in id order (= import row order): count the used ones per range, then write them
    std::vector<uint64_t> used(parts + 1, 0);
    auto r_lo = [&](unsigned p) { return S.nrel * p / parts; };
    if (!store_parts(parts, T, errs, [&](unsigned p, std::string* err) {
          uint64_t c = 0;
          for (uint64_t id = r_lo(p); id < r_lo(p + 1); ++id) {
            const RelRec r = rel_at(S, id);
            if (!r.in_use) continue;
            if (r.first >= n || r.second >= n) {
              *err = fmt("%s: relationship %llu references a node outside the store", store_dir, (unsigned long long)id);
              return;
            }
            ++c;
          }
          used[p + 1] = c;
        }))
      return fail(PPRHIP_ERR_IO);
    for (unsigned p = 0; p < parts; ++p) used[p + 1] += used[p];
    const uint64_t m = S.m = used[parts];
    e->src.resize(m);
    e->dst.resize(m);
    store_parts(parts, T, errs, [&](unsigned p, std::string*) {
      uint64_t w = used[p];
      for (uint64_t id = r_lo(p); id < r_lo(p + 1); ++id) {
        const RelRec r = rel_at(S, id);
        if (!r.in_use) continue;
        e->src[w] = (int32_t)r.first;
        e->dst[w] = (int32_t)r.second;
        ++w;
      }
    });
    // ---- every node's chains, as HeavyGraph's loader walks them: degrees first, then the lists
    e->out_rp.assign(n + 1, 0);
    e->in_rp.assign(n + 1, 0);
    auto v_lo = [&](unsigned p) { return n * p / parts; };
    if (!store_parts(parts, T, errs, [&](unsigned p, std::string* err) {
          for (uint64_t v = v_lo(p); v < v_lo(p + 1); ++v) {
            uint32_t od = 0, id = 0;
            if (!walk_node(S, v, [&](bool outgoing, uint64_t) { (outgoing ? od : id)++; }, err)) return;
            e->out_rp[v + 1] = od;
            e->in_rp[v + 1] = id;
          }
        }))
      return fail(PPRHIP_ERR_IO);
    for (uint64_t v = 0; v < n; ++v) {
      e->out_rp[v + 1] += e->out_rp[v];
      e->in_rp[v + 1] += e->in_rp[v];
    }
    if (e->out_rp[n] != m || e->in_rp[n] != m) {
      set_error("%s: chains cover %u out / %u in of %llu relationships", store_dir, e->out_rp[n], e->in_rp[n],
                (unsigned long long)m);
      return PPRHIP_ERR_IO;
    }
    e->out_ci.resize(m);
    e->in_ci.resize(m);
    if (!store_parts(parts, T, errs, [&](unsigned p, std::string* err) {
          for (uint64_t v = v_lo(p); v < v_lo(p + 1); ++v) {
            uint32_t ow = e->out_rp[v], iw = e->in_rp[v];
            if (!walk_node(S, v, [&](bool outgoing, uint64_t u) {
                  if (outgoing) e->out_ci[ow++] = (int32_t)u;
                  else e->in_ci[iw++] = (int32_t)u;
                }, err))
              return;
          }
        }))
      return fail(PPRHIP_ERR_IO);
    e->names.reserve(n);
    for (uint64_t v = 0; v < n; ++v) e->names.push_back(std::to_string(v));
  } catch (const std::bad_alloc&) {
    set_error("%s: out of host memory", store_dir);
    return PPRHIP_ERR_OOM;
  }
  *out = e.release();
  return PPRHIP_OK;
}

int pprhip_edgelist_build_csr(const pprhip_edgelist_t* e, int incoming, uint32_t* row_ptr_out, int32_t* col_idx_out) {
  if (!e || !row_ptr_out || (!col_idx_out && !e->src.empty())) {
    set_error("pprhip_edgelist_build_csr: null argument");
    return PPRHIP_ERR_INVALID;
  }
  if (e->from_store) {
    const auto& rp = incoming ? e->in_rp : e->out_rp;
    const auto& ci = incoming ? e->in_ci : e->out_ci;
    std::copy(rp.begin(), rp.end(), row_ptr_out);
    std::copy(ci.begin(), ci.end(), col_idx_out);
    return PPRHIP_OK;
  }
  // import CSVs: HeavyGraph lists the newest relationship first (descending relationship id)
  return pprhip_csr_build(e->n, e->src.size(), incoming ? e->dst.data() : e->src.data(),
                          incoming ? e->src.data() : e->dst.data(), 1, row_ptr_out, col_idx_out);
}

int pprhip_edgelist_info(const pprhip_edgelist_t* e, uint32_t* n, uint64_t* m) {
  if (!e) {
    set_error("pprhip_edgelist_info: null edge list");
    return PPRHIP_ERR_INVALID;
  }
  if (n) *n = e->n;
  if (m) *m = e->src.size();
  return PPRHIP_OK;
}

int pprhip_edgelist_edges(const pprhip_edgelist_t* e, const int32_t** src, const int32_t** dst) {
  if (!e || !src || !dst) {
    set_error("pprhip_edgelist_edges: null argument");
    return PPRHIP_ERR_INVALID;
  }
  *src = e->src.data();
  *dst = e->dst.data();
  return PPRHIP_OK;
}

const char* pprhip_edgelist_node_name(const pprhip_edgelist_t* e, uint32_t id) {
  if (!e || id >= e->n) return nullptr;
  return e->names[id].c_str();
}

void pprhip_edgelist_destroy(pprhip_edgelist_t* e) { delete e; }

int pprhip_csr_build(uint32_t n, uint64_t m, const int32_t* key, const int32_t* val, int newest_first,
                     uint32_t* row_ptr_out, int32_t* col_idx_out) {
  if (!key || !val || !row_ptr_out || (!col_idx_out && m) || m > 0xFFFFFFFFull) {
    set_error("pprhip_csr_build: bad arguments");
    return PPRHIP_ERR_INVALID;
  }
  // A stable counting sort by key on T threads: thread t owns the edges [m t / T, m (t + 1) / T), counts them per
  // row, and writes them behind the rows' shares of the threads before it (in front of them when the newest edge
  // comes first), so the order inside a row is the edge order whatever T is.  T x n counters: T shrinks for huge n.
  unsigned T = m < (1u << 18) ? 1u : pprhip::detail::host_threads();
  while (T > 1 && (uint64_t)T * n * sizeof(uint32_t) > (2ull << 30)) T /= 2;
  try {
    std::vector<std::vector<uint32_t>> cnt(T);
    std::vector<uint64_t> bad(T, UINT64_MAX);
    auto e_lo = [&](unsigned t) { return m * t / T; };
    auto run = [&](auto&& fn) {
      std::vector<std::thread> th;
      for (unsigned t = 1; t < T; ++t) th.emplace_back(fn, t);
      fn(0u);
      for (auto& x : th) x.join();
    };
    bool oom[64] = {false};  // (host_threads() <= 64; an exception must not leave a worker thread)
    run([&](unsigned t) {
      try {
        cnt[t].assign((size_t)n, 0u);
      } catch (const std::bad_alloc&) {
        oom[t] = true;
        return;
      }
      uint32_t* c = cnt[t].data();
      for (uint64_t e = e_lo(t); e < e_lo(t + 1); ++e) {
        if (key[e] < 0 || (uint32_t)key[e] >= n || val[e] < 0 || (uint32_t)val[e] >= n) {
          bad[t] = e;
          return;
        }
        c[key[e]]++;
      }
    });
    for (unsigned t = 0; t < T; ++t)
      if (oom[t]) throw std::bad_alloc();
    const uint64_t first_bad = *std::min_element(bad.begin(), bad.end());
    if (first_bad != UINT64_MAX) {
      set_error("pprhip_csr_build: edge %llu has an endpoint outside [0, %u)", (unsigned long long)first_bad, n);
      return PPRHIP_ERR_INVALID;
    }
    // row pointers from the summed counts (sums in node ranges on all threads, then one pass of prefix sums)
    row_ptr_out[0] = 0;
    run([&](unsigned t) {
      const uint32_t v_lo = (uint32_t)((uint64_t)n * t / T), v_hi = (uint32_t)((uint64_t)n * (t + 1) / T);
      for (uint32_t v = v_lo; v < v_hi; ++v) {
        uint32_t sum = 0;
        for (unsigned u = 0; u < T; ++u) sum += cnt[u][v];
        row_ptr_out[v + 1] = sum;
      }
    });
    for (uint32_t v = 0; v < n; ++v) row_ptr_out[v + 1] += row_ptr_out[v];
    // cnt[t][v] becomes where thread t starts writing row v
    run([&](unsigned t) {
      const uint32_t v_lo = (uint32_t)((uint64_t)n * t / T), v_hi = (uint32_t)((uint64_t)n * (t + 1) / T);
      for (uint32_t v = v_lo; v < v_hi; ++v) {
        uint32_t at = newest_first ? row_ptr_out[v + 1] : row_ptr_out[v];
        for (unsigned u = 0; u < T; ++u) {
          const uint32_t c = cnt[u][v];
          cnt[u][v] = at;
          at = newest_first ? at - c : at + c;
        }
      }
    });
    run([&](unsigned t) {
      uint32_t* at = cnt[t].data();
      if (newest_first)
        for (uint64_t e = e_lo(t); e < e_lo(t + 1); ++e) col_idx_out[--at[key[e]]] = val[e];
      else
        for (uint64_t e = e_lo(t); e < e_lo(t + 1); ++e) col_idx_out[at[key[e]]++] = val[e];
    });
  } catch (const std::bad_alloc&) {
    set_error("pprhip_csr_build: out of host memory");
    return PPRHIP_ERR_OOM;
  }
  return PPRHIP_OK;
}

}  // extern "C"
