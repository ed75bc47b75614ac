// lift.cpp — the host half of the graph lift (row a11; PPR.setupAdjMatrix, PPR.java:136-152): from the caller's
// out / in adjacency to the engine's internal layout - validated arrays, the internal vertex order, both CSRs in
// that order, the pull sweep's row-start flags and the sliced copy of the in-CSR.  Everything here is plain host
// memory (no HIP call), so pprhip_graph_lift_host can run it without a device; pprhip_graph_create uploads the result.
//
// Until round 4 this was one thread's work inside pprhip_graph_create: 2.4 s for R-MAT 22 (67 M edges), 8.6 s for
// R-MAT 24 - more than fifty batched queries take.  The passes over the edges (validation, in-degree count, the
// renaming of both CSRs, the two passes of the sliced copy) now run on the threads the process may use; the vertex
// order is a counting sort by (has in-edges, out-degree) instead of a comparison sort.  Results are the same arrays
// byte for byte with any thread count (tests/test_host.py::test_lift_*).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <memory>
#include <new>
#include <numeric>
#include <thread>

#include "engine_internal.hpp"

using namespace pprhip;
using namespace pprhip::detail;

namespace pprhip {
namespace detail {

unsigned host_threads() {
  static const unsigned n = [] {
    unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    if (const char* e = tuning_env("PPRHIP_HOST_THREADS")) return (unsigned)std::min(64, std::max(1, atoi(e)));
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
      char q[32] = {0};
      double period = 0;
      if (fscanf(f, "%31s %lf", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
        const double cores = atof(q) / period;
        if (cores >= 1.0) hw = std::min(hw, (unsigned)(cores + 0.5));
      }
      fclose(f);
    }
    return std::max(1u, std::min(64u, hw));
  }();
  return n;
}

}  // namespace detail
}  // namespace pprhip

namespace {

struct PhaseClock {  // PPRHIP_LIFT_DEBUG=1: phase times of the lift on stderr
  bool on;
  std::chrono::steady_clock::time_point t;
  PhaseClock() : on(hook_env("PPRHIP_LIFT_DEBUG") != nullptr), t(std::chrono::steady_clock::now()) {}
  void mark(const char* what) {
    if (!on) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[pprhip lift] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
    t = now;
  }
};

// fn(part) for part in [0, parts) on T threads (the caller's included); parts are handed out in order
template <class F>
void parallel_parts(unsigned parts, unsigned T, F&& fn) {
  if (T <= 1 || parts <= 1) {
    for (unsigned p = 0; p < parts; ++p) fn(p);
    return;
  }
  std::atomic<unsigned> next{0};
  auto work = [&]() {
    for (unsigned p = next.fetch_add(1); p < parts; p = next.fetch_add(1)) fn(p);
  };
  std::vector<std::thread> th;
  const unsigned extra = std::min(T, parts) - 1;
  th.reserve(extra);
  for (unsigned w = 0; w < extra; ++w) th.emplace_back(work);
  work();
  for (auto& x : th) x.join();
}

// Row boundaries b[0] = 0 <= ... <= b[parts] = rows such that every range holds about the same number of edges
// (rows are ordered by degree, so equal row counts would give the first part most of the work).
std::vector<uint32_t> edge_balanced(const uint32_t* rp, uint32_t rows, unsigned parts) {
  std::vector<uint32_t> b(parts + 1, rows);
  b[0] = 0;
  const uint64_t m = rp[rows];
  for (unsigned p = 1; p < parts; ++p) {
    const uint64_t want = m * p / parts;
    b[p] = (uint32_t)(std::lower_bound(rp, rp + rows + 1, (uint32_t)std::min<uint64_t>(want, 0xffffffffull)) - rp);
    if (b[p] > rows) b[p] = rows;
    if (b[p] < b[p - 1]) b[p] = b[p - 1];
  }
  return b;
}

inline size_t padded_edges(uint64_t m) { return ((size_t)m + kChunkPad - 1) / kChunkPad * kChunkPad + kChunkPad; }

// An edge array of m entries that the threads fill in full, padded to whole chunks plus one: only the padding is
// zeroed (a value-initialising resize is one thread's pass over hundreds of megabytes of fresh pages - 50 ms per array
// at R-MAT 22, a third of the lift's host half for its three arrays).
inline void size_edge_array(RawVec<int32_t>& a, uint64_t m) {
  a.resize(padded_edges(m));
  std::fill(a.begin() + (size_t)m, a.end(), 0);
}

}  // namespace

namespace pprhip {
namespace detail {

int lift_host(uint32_t n, uint64_t m, const uint32_t* out_rp, const int32_t* out_ci, const uint32_t* in_rp,
              const int32_t* in_ci, unsigned threads, HostLift& H) {
  PhaseClock clk;
  const bool have_in = in_rp && (in_ci || m == 0);
  const unsigned T = ((uint64_t)n + m < (1u << 20)) ? 1u : std::max(1u, threads ? threads : host_threads());
  const unsigned parts = T == 1 ? 1u : T * 8u;

  // ---- caller-supplied arrays are validated before anything indexes with them: monotone row pointers, column
  // indices in range, and (when given) an in-adjacency with the transpose's degree sequence.  The first offence in
  // array order is the one reported, whatever thread met it.
  for (int side = 0; side < (have_in ? 2 : 1); ++side) {
    const uint32_t* rp = side ? in_rp : out_rp;
    const int32_t* ci = side ? in_ci : out_ci;
    const char* nm = side ? "in" : "out";
    std::vector<uint64_t> bad_v(parts, UINT64_MAX), bad_e(parts, UINT64_MAX);
    parallel_parts(parts, T, [&](unsigned p) {
      const uint32_t v_lo = (uint32_t)((uint64_t)n * p / parts), v_hi = (uint32_t)((uint64_t)n * (p + 1) / parts);
      for (uint32_t v = v_lo; v < v_hi; ++v)
        if (rp[v + 1] < rp[v]) {
          bad_v[p] = v;
          break;
        }
      const uint64_t e_lo = m * p / parts, e_hi = m * (p + 1) / parts;
      for (uint64_t e = e_lo; e < e_hi; ++e)
        if (ci[e] < 0 || (uint32_t)ci[e] >= n) {
          bad_e[p] = e;
          break;
        }
    });
    const uint64_t bv = *std::min_element(bad_v.begin(), bad_v.end());
    if (bv != UINT64_MAX) {
      set_error("pprhip_graph_create: %s_row_ptr decreases at node %u (%u -> %u)", nm, (uint32_t)bv, rp[bv], rp[bv + 1]);
      return PPRHIP_ERR_INVALID;
    }
    const uint64_t be = *std::min_element(bad_e.begin(), bad_e.end());
    if (be != UINT64_MAX) {
      set_error("pprhip_graph_create: %s_col_idx[%llu] = %d outside [0, %u)", nm, (unsigned long long)be, ci[be], n);
      return PPRHIP_ERR_INVALID;
    }
  }
  clk.mark("validate");

  // In-degrees (the relabeling's "has in-edges") and the transpose check.  With the caller's in-adjacency the check is
  // a sum over the edges of both sides of a 64-bit mix of (source, destination): equal multisets of edges give equal
  // sums, and the sides are streamed once each on all threads - no counter per node is touched (counting the
  // destinations of 67 M out-edges with atomics was the longest phase of the lift, 45 ms).  Only when the sums differ
  // are the destinations counted, to name a node in the message.
  auto count_destinations = [&](std::vector<uint32_t>& deg) {
    deg.assign((size_t)n, 0u);
    if (T == 1) {
      for (uint64_t e = 0; e < m; ++e) deg[out_ci[e]]++;
    } else {
      uint32_t* d = deg.data();
      parallel_parts(parts, T, [&](unsigned p) {
        const uint64_t e_lo = m * p / parts, e_hi = m * (p + 1) / parts;
        for (uint64_t e = e_lo; e < e_hi; ++e) __atomic_fetch_add(&d[out_ci[e]], 1u, __ATOMIC_RELAXED);
      });
    }
  };
  std::vector<uint32_t> indeg;
  if (have_in) {
    auto mix = [](uint32_t u, uint32_t v) {
      uint64_t x = ((uint64_t)u << 32 | v) + 0x9E3779B97F4A7C15ull;
      x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
      x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
      return x ^ (x >> 31);
    };
    std::vector<uint64_t> sum_out(parts, 0), sum_in(parts, 0);
    const std::vector<uint32_t> bo = edge_balanced(out_rp, n, parts), bi = edge_balanced(in_rp, n, parts);
    parallel_parts(parts, T, [&](unsigned p) {
      uint64_t a = 0, b = 0;
      for (uint32_t u = bo[p]; u < bo[p + 1]; ++u)
        for (uint32_t e = out_rp[u]; e < out_rp[u + 1]; ++e) a += mix(u, (uint32_t)out_ci[e]);
      for (uint32_t v = bi[p]; v < bi[p + 1]; ++v)
        for (uint32_t e = in_rp[v]; e < in_rp[v + 1]; ++e) b += mix((uint32_t)in_ci[e], v);
      sum_out[p] = a;
      sum_in[p] = b;
    });
    uint64_t so = 0, si = 0;
    for (unsigned p = 0; p < parts; ++p) {
      so += sum_out[p];
      si += sum_in[p];
    }
    if (so != si) {
      count_destinations(indeg);
      for (uint32_t v = 0; v < n; ++v)
        if (indeg[v] != in_rp[v + 1] - in_rp[v]) {
          set_error("pprhip_graph_create: in-adjacency is not the transpose of the out-adjacency (node %u: %u in-edges "
                    "listed, %u relationships point to it)", v, in_rp[v + 1] - in_rp[v], indeg[v]);
          return PPRHIP_ERR_INVALID;
        }
      set_error("pprhip_graph_create: in-adjacency is not the transpose of the out-adjacency (every node lists as many "
                "in-edges as relationships point to it, but not from the same sources)");
      return PPRHIP_ERR_INVALID;
    }
    indeg.resize(n);
    for (uint32_t v = 0; v < n; ++v) indeg[v] = in_rp[v + 1] - in_rp[v];
  } else {
    count_destinations(indeg);
  }
  clk.mark("in-degrees");

  // ---- internal vertex order: nodes with in-edges first, then out-degree descending, ties by original id
  // (PPRHIP_RELABEL=0 keeps ids).  The rows a sweep applies (nodes with in-edges) are then the ids [0, n_nz): their
  // residue / reserve / contribution entries are contiguous and every line a sweep touches is used in full; inside
  // that range the most-gathered contributions (highest out-degree) still come first (hot table, slices).
  const char* env = hook_env("PPRHIP_RELABEL");
  H.relabeled = !(env && env[0] == '0');
  H.new2old.resize(n);
  H.old2new.resize(n);
  if (H.relabeled) {
    uint32_t maxdeg = 0;
    for (uint32_t v = 0; v < n; ++v) maxdeg = std::max(maxdeg, out_rp[v + 1] - out_rp[v]);
    if ((uint64_t)maxdeg + 1 <= std::max<uint64_t>(4ull * n, 1ull << 20)) {
      // counting sort, nodes taken in id order: stable, i.e. ties by original id
      const size_t nb = 2 * ((size_t)maxdeg + 1);
      std::vector<uint32_t> pos(nb + 1, 0);
      auto bucket = [&](uint32_t v) {
        return (size_t)(indeg[v] == 0 ? 1 : 0) * ((size_t)maxdeg + 1) + (maxdeg - (out_rp[v + 1] - out_rp[v]));
      };
      for (uint32_t v = 0; v < n; ++v) pos[bucket(v) + 1]++;
      for (size_t b = 0; b < nb; ++b) pos[b + 1] += pos[b];
      for (uint32_t v = 0; v < n; ++v) H.new2old[pos[bucket(v)]++] = (int32_t)v;
    } else {
      std::iota(H.new2old.begin(), H.new2old.end(), 0);
      std::stable_sort(H.new2old.begin(), H.new2old.end(), [&](int32_t x, int32_t y) {
        const bool nx = indeg[x] == 0, ny = indeg[y] == 0;
        if (nx != ny) return nx < ny;
        return out_rp[x + 1] - out_rp[x] > out_rp[y + 1] - out_rp[y];
      });
    }
  } else {
    std::iota(H.new2old.begin(), H.new2old.end(), 0);
  }
  for (uint32_t v = 0; v < n; ++v) H.old2new[H.new2old[v]] = (int32_t)v;
  clk.mark("vertex order");

  // ---- both CSRs in the internal order: rows move, entries are renamed, the order inside a row is kept (walks index
  // rows by position)
  const int32_t* n2o = H.new2old.data();
  const int32_t* o2n = H.old2new.data();
  auto relabel_csr = [&](const uint32_t* rp, const int32_t* ci, std::vector<uint32_t>& nrp, RawVec<int32_t>& nci) {
    nrp.assign((size_t)n + 1, 0);
    parallel_parts(parts, T, [&](unsigned p) {
      const uint32_t v_lo = (uint32_t)((uint64_t)n * p / parts), v_hi = (uint32_t)((uint64_t)n * (p + 1) / parts);
      for (uint32_t v = v_lo; v < v_hi; ++v) {
        const int32_t o = n2o[v];
        nrp[v + 1] = rp[o + 1] - rp[o];
      }
    });
    for (uint32_t v = 0; v < n; ++v) nrp[v + 1] += nrp[v];
    size_edge_array(nci, m);
    const std::vector<uint32_t> b = edge_balanced(nrp.data(), n, parts);
    parallel_parts(parts, T, [&](unsigned p) {
      for (uint32_t v = b[p]; v < b[p + 1]; ++v) {
        const int32_t o = n2o[v];
        uint32_t w = nrp[v];
        for (uint32_t e = rp[o]; e < rp[o + 1]; ++e) nci[w++] = o2n[ci[e]];
      }
    });
  };
  relabel_csr(out_rp, out_ci, H.out_rp, H.out_ci);
  clk.mark("out-CSR renamed");
  if (have_in) {
    relabel_csr(in_rp, in_ci, H.in_rp, H.in_ci);
  } else {
    // derive the in-adjacency: edges in (new) out-CSR order, grouped by destination
    std::vector<int32_t> src(m);
    for (uint32_t v = 0; v < n; ++v)
      for (uint32_t e = H.out_rp[v]; e < H.out_rp[v + 1]; ++e) src[e] = (int32_t)v;
    H.in_rp.resize((size_t)n + 1);
    size_edge_array(H.in_ci, m);
    PPRHIP_TRY(pprhip_csr_build(n, m, H.out_ci.data(), src.data(), 0, H.in_rp.data(), H.in_ci.data()));
  }
  clk.mark("in-CSR renamed");

  // ---- dense pull-sweep layout: non-empty rows, row-start flags per in-edge, starts before each chunk
  const std::vector<uint32_t>& irp = H.in_rp;
  const std::vector<uint32_t>& orp = H.out_rp;
  const size_t n_chunks = ((size_t)m + kChunkPad - 1) / kChunkPad;
  H.n_chunks = (uint32_t)n_chunks;
  H.flags.assign((n_chunks + 1) * (kChunkPad / 8), 0);
  H.chunk_starts.assign(n_chunks + 1, 0);
  H.nz_rows.clear();
  H.zin_rows.clear();
  H.n_src_live = 0;
  for (uint32_t v = 0; v < n; ++v) {
    const bool has_out = orp[v + 1] > orp[v];
    H.n_src_live += has_out ? 1u : 0u;
    if (irp[v + 1] == irp[v]) {
      // Rows without in-edges never receive mass, so the only contribution such a row can hold is its own when it is
      // a query's source - and a dead-end's contribution is zero (Forward_Push.java:101-104).  The batched sweep
      // therefore carries the rows without in-edges that have out-edges and leaves the isolated ones (43 % of an
      // R-MAT 22's nodes) out: their entries of the contribution arrays are never written and stay zero.
      if (has_out) H.zin_rows.push_back((int32_t)v);
      continue;
    }
    H.nz_rows.push_back((int32_t)v);
    const uint32_t e = irp[v];
    H.flags[e >> 3] |= (uint8_t)(1u << (e & 7));
    H.chunk_starts[(size_t)e / kChunkPad + 1]++;  // counted into every later chunk by the prefix sum below
  }
  for (size_t c = 1; c <= n_chunks; ++c) H.chunk_starts[c] += H.chunk_starts[c - 1];
  H.ext.resize(n);
  for (uint32_t v = 0; v < n; ++v) H.ext[v] = (unsigned long long)orp[v] | ((unsigned long long)(orp[v + 1] - orp[v]) << 32);
  H.cross.assign(((size_t)n + 63) / 64 + 1, 0ull);
  for (size_t j = 0; j < H.nz_rows.size(); ++j) {
    const uint32_t v = (uint32_t)H.nz_rows[j];
    // summed with atomics (so cleared after every sweep): rows holding the last edge of a chunk
    const uint32_t last = irp[v + 1] - 1;
    if (irp[v] / kChunkPad != last / kChunkPad || (last + 1) % kChunkPad == 0 || (uint64_t)last + 1 == m)
      H.cross[j >> 6] |= 1ull << (j & 63);
  }
  clk.mark("sweep layout");

  // ---- sliced copy of the (internal-order) in-CSR.  Slices are ranges of `width` source ids up to the last id that
  // has out-edges; no layout when they fit one slice.
  H.S = 0;
  H.n_seg = 0;
  // ---- row-panel copy (the single-query forward sweep's layout from 2^26 edges on - measured per dense level, panels
  // against the sliced copy: R-MAT 21 185 / 183 us, 22 290 / 361, 23 538 / 781, 24 1 101 / 1 781; PPRHIP_SWEEP1_PANELS=0 /
  // 1, a test switch, forces it off / on at any size): graphs that have it need no sliced copy
  {
    const char* pe = hook_env("PPRHIP_SWEEP1_PANELS");
    const bool want = pe ? pe[0] == '1' : m >= (1ull << 26);
    if (want && m > 0 && !H.nz_rows.empty()) {
      PPRHIP_TRY(build_panel_layout(n, m, irp.data(), H.in_ci.data(), H.nz_rows.data(), (uint32_t)H.nz_rows.size(), T, H.pn));
      if (H.pn.n_items) return PPRHIP_OK;
    }
  }
  const char* off = hook_env("PPRHIP_SLICED");
  if (off && off[0] == '0') return PPRHIP_OK;
  uint32_t n_src = n;  // ids above the last node with out-edges are never gathered
  while (n_src > 0 && orp[n_src] == orp[n_src - 1]) --n_src;
  const char* wenv = hook_env("PPRHIP_SLICE_IDS");
  uint64_t width = wenv ? strtoull(wenv, nullptr, 10) : 393216ull;  // 3 MB of contributions per slice
  if (width < 1) width = 1;
  uint64_t S = ((uint64_t)n_src + width - 1) / width;
  if (S < 2 || m == 0 || H.nz_rows.empty()) return PPRHIP_OK;
  if (S > (uint64_t)kMaxWindows) {
    S = kMaxWindows;
    width = ((uint64_t)n_src + S - 1) / S;
  }
  const std::vector<int32_t>& rows = H.nz_rows;
  const int32_t* ici = H.in_ci.data();
  const uint32_t width32 = (uint32_t)std::min<uint64_t>(width, 0xffffffffull), s_last = (uint32_t)S - 1;
  auto slice_of = [&](int32_t u) {
    const uint32_t q = (uint32_t)u / width32;
    return (size_t)(q < s_last ? q : s_last);
  };
  // row ranges of about equal edge counts; a range's edges of one slice stay together and in row order, so the copy
  // is the one a single pass over all rows would write
  std::vector<uint32_t> jb(parts + 1, (uint32_t)rows.size());
  jb[0] = 0;
  for (unsigned p = 1; p < parts; ++p) {
    const uint32_t want = (uint32_t)(m * p / parts);
    const size_t j = std::lower_bound(rows.begin(), rows.end(), want,
                                      [&](int32_t v, uint32_t x) { return irp[(uint32_t)v] < x; }) - rows.begin();
    jb[p] = std::max(jb[p - 1], (uint32_t)j);
  }
  std::vector<uint64_t> ecnt((size_t)parts * S, 0), scnt((size_t)parts * S, 0);
  parallel_parts(parts, T, [&](unsigned p) {
    uint64_t ec[kMaxWindows] = {0}, sc[kMaxWindows] = {0};  // (counters of neighbouring ranges would share lines)
    uint32_t last[kMaxWindows];
    for (size_t q = 0; q < S; ++q) last[q] = 0xffffffffu;
    for (uint32_t j = jb[p]; j < jb[p + 1]; ++j) {
      const uint32_t v = (uint32_t)rows[j];
      for (uint32_t e = irp[v]; e < irp[v + 1]; ++e) {
        const size_t q = slice_of(ici[e]);
        ec[q]++;
        if (last[q] != j) {
          last[q] = j;
          sc[q]++;
        }
      }
    }
    for (size_t q = 0; q < S; ++q) {
      ecnt[(size_t)p * S + q] = ec[q];
      scnt[(size_t)p * S + q] = sc[q];
    }
  });
  H.edge_base.assign(S + 1, 0);
  H.seg_base.assign(S + 1, 0);
  for (size_t q = 0; q < S; ++q) {
    uint64_t e = 0, s = 0;
    for (unsigned p = 0; p < parts; ++p) {
      e += ecnt[(size_t)p * S + q];
      s += scnt[(size_t)p * S + q];
    }
    H.edge_base[q + 1] = H.edge_base[q] + e;
    H.seg_base[q + 1] = H.seg_base[q] + s;
  }
  if (H.seg_base[S] >= 0xffffffffull) return PPRHIP_OK;  // segment ordinals are 32-bit: keep the row-major sweep
  // where every range starts writing: slice base + what the ranges before it hold of the slice
  std::vector<uint64_t> epos0((size_t)parts * S), spos0((size_t)parts * S);
  for (size_t q = 0; q < S; ++q) {
    uint64_t e = H.edge_base[q], s = H.seg_base[q];
    for (unsigned p = 0; p < parts; ++p) {
      epos0[(size_t)p * S + q] = e;
      spos0[(size_t)p * S + q] = s;
      e += ecnt[(size_t)p * S + q];
      s += scnt[(size_t)p * S + q];
    }
  }
  H.S = (int)S;
  H.width = (uint32_t)width;
  H.n_seg = (uint32_t)H.seg_base[S];
  size_edge_array(H.sl_ci, m);
  H.sl_flags.assign((n_chunks + 1) * (kChunkPad / 8), 0);
  H.sl_chunk_starts.assign(n_chunks + 1, 0);
  H.seg_row.resize(H.n_seg);
  H.seg_off.resize(H.n_seg);
  uint8_t* fl = H.sl_flags.data();
  uint32_t* cst = H.sl_chunk_starts.data();
  parallel_parts(parts, T, [&](unsigned p) {
    uint64_t epos[kMaxWindows], spos[kMaxWindows];
    uint32_t last[kMaxWindows];
    for (size_t q = 0; q < S; ++q) {
      epos[q] = epos0[(size_t)p * S + q];
      spos[q] = spos0[(size_t)p * S + q];
      last[q] = 0xffffffffu;
    }
    for (uint32_t j = jb[p]; j < jb[p + 1]; ++j) {
      const uint32_t v = (uint32_t)rows[j];
      for (uint32_t e = irp[v]; e < irp[v + 1]; ++e) {
        const size_t q = slice_of(ici[e]);
        const uint64_t w = epos[q]++;
        H.sl_ci[w] = ici[e];
        if (last[q] != j) {
          last[q] = j;
          const uint64_t sg = spos[q]++;
          H.seg_row[sg] = j;
          H.seg_off[sg] = (uint32_t)w;
          // neighbouring ranges can meet inside one byte of flags / one counter
          __atomic_fetch_or(&fl[w >> 3], (uint8_t)(1u << (w & 7)), __ATOMIC_RELAXED);
          __atomic_fetch_add(&cst[(size_t)w / kChunkPad + 1], 1u, __ATOMIC_RELAXED);
        }
      }
    }
  });
  for (size_t c = 1; c <= n_chunks; ++c) H.sl_chunk_starts[c] += H.sl_chunk_starts[c - 1];
  clk.mark("sliced copy");
  return PPRHIP_OK;
}

// ---- row-panel copy of the in-CSR (engine_internal.hpp: HostPanelLayout).  Pass A, one thread: edges, parts and
// offsets per panel; pass B, all threads, panels handed out in order: a panel's edges as keys source << 16 | local row,
// sorted, written part by part with the padding - the arrays are the same with any thread count.
int build_panel_layout(uint32_t n, uint64_t m, const uint32_t* in_rp, const int32_t* in_ci, const int32_t* nz_rows,
                       uint32_t n_nz, unsigned threads, HostPanelLayout& L) {
  (void)n;
  PhaseClock clk;
  const unsigned T = (m < (1u << 20)) ? 1u : std::max(1u, threads ? threads : host_threads());
  L.n_nz = n_nz;
  L.n_panels = (n_nz + kPanelRows - 1) / kPanelRows;
  const size_t NP = L.n_panels;
  L.panel_item0.assign(NP + 1, 0);
  L.panels.assign(NP, PanelDesc{0, 0, 0, 0});
  std::vector<uint64_t> panel_edges(NP, 0);
  uint64_t items = 0, steps = 0, part = 0;
  for (size_t t = 0; t < NP; ++t) {
    const uint32_t j_lo = (uint32_t)t * kPanelRows, j_hi = (uint32_t)std::min<uint64_t>(n_nz, (uint64_t)(t + 1) * kPanelRows);
    uint64_t e = 0;
    for (uint32_t j = j_lo; j < j_hi; ++j) e += in_rp[(uint32_t)nz_rows[j] + 1] - in_rp[(uint32_t)nz_rows[j]];
    panel_edges[t] = e;
    const uint64_t S = std::max<uint64_t>(1, (e + kItemEdges - 1) / kItemEdges);
    L.panel_item0[t] = (uint32_t)items;
    L.panels[t] = PanelDesc{(uint32_t)part, (uint32_t)S, j_hi - j_lo, kNoFold};
    for (uint64_t k = 0; k < S; ++k) steps += (e * (k + 1) / S - e * k / S + kPanelStep - 1) / kPanelStep;
    items += S;
    part += (uint64_t)(j_hi - j_lo) * S;
    if (part >= 0xfffffff0ull || items >= 0xfffffff0ull || steps >= 0xfffffff0ull) {
      L.n_items = 0;  // offsets are 32-bit: the other layouts stay
      return PPRHIP_OK;
    }
  }
  // panels of many parts (the hub rows'): room behind the parts for their sums kFoldParts at a time
  for (size_t t = 0; t < NP; ++t)
    if (L.panels[t].parts > kFoldMin) {
      L.panels[t].fold = (uint32_t)part;
      part += (uint64_t)L.panels[t].rows * ((L.panels[t].parts + kFoldParts - 1) / kFoldParts);
      if (part >= 0xfffffff0ull) {
        L.n_items = 0;
        return PPRHIP_OK;
      }
    }
  L.panel_item0[NP] = (uint32_t)items;
  L.n_items = (uint32_t)items;
  L.n_part = part;
  L.n_edges = steps * kPanelStep;
  L.items.assign((size_t)items, PanelItem{0, 0, 0, 0});
  {
    uint64_t st = 0;
    for (size_t t = 0; t < NP; ++t) {
      const uint32_t i0 = L.panel_item0[t], S = L.panel_item0[t + 1] - i0;
      const uint64_t e = panel_edges[t];
      for (uint32_t k = 0; k < S; ++k) {
        PanelItem& I = L.items[(size_t)i0 + k];
        I.edge0 = (uint32_t)st;
        I.steps = (uint32_t)((e * (k + 1) / S - e * k / S + kPanelStep - 1) / kPanelStep);
        I.panel = (uint32_t)t;
        I.part0 = L.panels[t].base + k * L.panels[t].rows;
        st += I.steps;
      }
    }
  }
  L.src.resize((size_t)L.n_edges);
  L.rloc.resize((size_t)L.n_edges);
  clk.mark("panels: offsets");
  parallel_parts((unsigned)NP, T, [&](unsigned t) {
    const uint32_t j_lo = t * kPanelRows, j_hi = (uint32_t)std::min<uint64_t>(n_nz, (uint64_t)(t + 1) * kPanelRows);
    std::vector<uint64_t> key;
    key.reserve((size_t)panel_edges[t]);
    for (uint32_t j = j_lo; j < j_hi; ++j) {
      const uint32_t v = (uint32_t)nz_rows[j];
      for (uint32_t e = in_rp[v]; e < in_rp[v + 1]; ++e) key.push_back((uint64_t)(uint32_t)in_ci[e] << 16 | (j - j_lo));
    }
    std::sort(key.begin(), key.end());
    const uint32_t i0 = L.panel_item0[t], S = L.panel_item0[t + 1] - i0;
    const uint64_t e = key.size();
    for (uint32_t k = 0; k < S; ++k) {
      const PanelItem& I = L.items[(size_t)i0 + k];
      const uint64_t lo = e * k / S, hi = e * (k + 1) / S;
      int32_t* so = L.src.data() + (size_t)I.edge0 * kPanelStep;
      uint16_t* ro = L.rloc.data() + (size_t)I.edge0 * kPanelStep;
      // a wave takes 512 consecutive edges per turn, lane l the edges l, l + 64, ... of them (eight gathers, each over 64
      // consecutive edges: the lines of one instruction are no other's) - stored so that they are the lane's 32 bytes
      auto at = [](uint64_t x) { return (x & ~511ull) | ((x & 63ull) << 3) | ((x >> 6) & 7ull); };
      for (uint64_t x = lo; x < hi; ++x) {
        so[at(x - lo)] = (int32_t)(key[x] >> 16);
        ro[at(x - lo)] = (uint16_t)(key[x] & 0xffffu);
      }
      for (uint64_t x = hi - lo; x < (uint64_t)I.steps * kPanelStep; ++x) {
        so[at(x)] = 0;
        ro[at(x)] = 0xffffu;
      }
    }
  });
  clk.mark("panels: edges");
  return PPRHIP_OK;
}

}  // namespace detail
}  // namespace pprhip

// ---- the host half alone, for inspection without a device (tests; tools)
struct pprhip_lift {
  HostLift H;
  uint32_t n = 0;
  uint64_t m = 0;
  int threads = 0;
  mutable uint64_t panel_sizes[4] = {0, 0, 0, 0};  // panels, items, doubles of partial sums, edges with padding
};

extern "C" {

int pprhip_graph_lift_host(uint32_t n, uint64_t m, const uint32_t* out_row_ptr, const int32_t* out_col_idx,
                           const uint32_t* in_row_ptr, const int32_t* in_col_idx, int threads, pprhip_lift_t** lift_out) {
  if (!lift_out || !out_row_ptr || (!out_col_idx && m) || n == 0 || n >= (1u << 28) || m >= (1ull << 32) - 1024 ||
      threads < 0) {
    set_error("pprhip_graph_lift_host: bad arguments (n=%u m=%llu; limits n < 2^28, m < 2^32 - 1024)", n,
              (unsigned long long)m);
    return PPRHIP_ERR_INVALID;
  }
  if (out_row_ptr[0] != 0 || out_row_ptr[n] != m ||
      (in_row_ptr && (in_col_idx || m == 0) && (in_row_ptr[0] != 0 || in_row_ptr[n] != m))) {
    set_error("pprhip_graph_lift_host: row_ptr[0] must be 0 and row_ptr[n] must equal m");
    return PPRHIP_ERR_INVALID;
  }
  try {
    std::unique_ptr<pprhip_lift> L(new pprhip_lift());
    L->n = n;
    L->m = m;
    L->threads = threads;
    PPRHIP_TRY(lift_host(n, m, out_row_ptr, out_col_idx, in_row_ptr, in_col_idx, (unsigned)threads, L->H));
    *lift_out = L.release();
    return PPRHIP_OK;
  } catch (const std::bad_alloc&) {
    set_error("pprhip_graph_lift_host: out of host memory");
    return PPRHIP_ERR_OOM;
  }
}

int pprhip_lift_array(const pprhip_lift_t* lift, int which, const void** data_out, uint64_t* bytes_out) {
  if (!lift || !data_out || !bytes_out) {
    set_error("pprhip_lift_array: null argument");
    return PPRHIP_ERR_INVALID;
  }
  const HostLift& H = lift->H;
  auto give = [&](const void* p, size_t bytes) {
    *data_out = p;
    *bytes_out = bytes;
    return PPRHIP_OK;
  };
#define PPRHIP_LIFT_VEC(v) give((v).data(), (v).size() * sizeof((v)[0]))
  switch (which) {
    case PPRHIP_LIFT_PANEL_SIZES:
      lift->panel_sizes[0] = H.pn.n_panels;
      lift->panel_sizes[1] = H.pn.n_items;
      lift->panel_sizes[2] = H.pn.n_part;
      lift->panel_sizes[3] = H.pn.n_edges;
      return give(lift->panel_sizes, H.pn.n_items ? sizeof lift->panel_sizes : 0);
    case PPRHIP_LIFT_PANEL_SRC: return PPRHIP_LIFT_VEC(H.pn.src);
    case PPRHIP_LIFT_PANEL_ROW: return PPRHIP_LIFT_VEC(H.pn.rloc);
    case PPRHIP_LIFT_PANEL_ITEMS: return PPRHIP_LIFT_VEC(H.pn.items);
    case PPRHIP_LIFT_PANEL_DESC: return PPRHIP_LIFT_VEC(H.pn.panels);
    case PPRHIP_LIFT_PANEL_ITEM0: return PPRHIP_LIFT_VEC(H.pn.panel_item0);
    case PPRHIP_LIFT_NEW2OLD: return PPRHIP_LIFT_VEC(H.new2old);
    case PPRHIP_LIFT_OLD2NEW: return PPRHIP_LIFT_VEC(H.old2new);
    case PPRHIP_LIFT_OUT_ROW_PTR: return PPRHIP_LIFT_VEC(H.out_rp);
    case PPRHIP_LIFT_OUT_COL_IDX: return give(H.out_ci.data(), sizeof(int32_t) * (size_t)lift->m);
    case PPRHIP_LIFT_IN_ROW_PTR: return PPRHIP_LIFT_VEC(H.in_rp);
    case PPRHIP_LIFT_IN_COL_IDX: return give(H.in_ci.data(), sizeof(int32_t) * (size_t)lift->m);
    case PPRHIP_LIFT_NZ_ROWS: return PPRHIP_LIFT_VEC(H.nz_rows);
    case PPRHIP_LIFT_ZIN_ROWS: return PPRHIP_LIFT_VEC(H.zin_rows);
    case PPRHIP_LIFT_ROW_START_FLAGS: return PPRHIP_LIFT_VEC(H.flags);
    case PPRHIP_LIFT_CHUNK_STARTS: return PPRHIP_LIFT_VEC(H.chunk_starts);
    case PPRHIP_LIFT_CROSS_BITS: return PPRHIP_LIFT_VEC(H.cross);
    case PPRHIP_LIFT_SLICE_EDGE_BASE: return PPRHIP_LIFT_VEC(H.edge_base);
    case PPRHIP_LIFT_SLICE_SEG_BASE: return PPRHIP_LIFT_VEC(H.seg_base);
    case PPRHIP_LIFT_SLICED_COL_IDX: return give(H.sl_ci.data(), H.S ? sizeof(int32_t) * (size_t)lift->m : 0);
    case PPRHIP_LIFT_SLICED_FLAGS: return PPRHIP_LIFT_VEC(H.sl_flags);
    case PPRHIP_LIFT_SLICED_CHUNK_STARTS: return PPRHIP_LIFT_VEC(H.sl_chunk_starts);
    case PPRHIP_LIFT_SEG_ROW: return PPRHIP_LIFT_VEC(H.seg_row);
    case PPRHIP_LIFT_SEG_OFF: return PPRHIP_LIFT_VEC(H.seg_off);
    default: break;
  }
#undef PPRHIP_LIFT_VEC
  set_error("pprhip_lift_array: no array %d", which);
  return PPRHIP_ERR_INVALID;
}

void pprhip_lift_destroy(pprhip_lift_t* lift) { delete lift; }

}  // extern "C"
