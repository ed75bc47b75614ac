// kernels_walk.hip — Monte-Carlo refinement: one random walk per lane with refill (gfx950).
//
// Replaces Monte_Carlo.random_walk / random_walk_no_zero_hop (Monte_Carlo.java:60-133) and the
// walk loops of Fora_Whole_Graph.java:119-140 / Fora_Topk.java:155-168.  The reference draws from
// an unseeded ThreadLocalRandom; here walk (seed, stream, start, walk_idx) is a pure function of
// its counter: Philox4x32-10 with key = seed and counter = (start, idx_lo, idx_hi16 | stream<<16,
// block).  Decision k of a walk uses block k>>1, words 2(k&1) (stop test: word * 2^-32 < alpha)
// and 2(k&1)+1 (neighbour pick: (word * degree) >> 32).  The CPU oracle restates the same
// function, so terminals can be compared walk by walk.
//
// Memory-bound random gathers (one packed row extent + one col_idx per step); no MFMA.
#include <algorithm>

#include "device_utils.hpp"
#include "engine.hpp"

namespace pprhip {

struct Philox {
  uint32_t x[4];
};

__device__ __forceinline__ Philox philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  Philox p;
  p.x[0] = c0; p.x[1] = c1; p.x[2] = c2; p.x[3] = c3;
  return p;
}

// Per-lane walk state machine: one decision per call of step().
struct Walker {
  int32_t start, cur;
  uint32_t c0, c1, c2;  // counter words 0-2 (original id of the start node, walk index, stream)
  uint32_t k;          // next decision number
  uint32_t w_stop, w_pick, w_stop2, w_pick2;  // cached Philox block
  uint32_t moves;
  bool forced;         // the next decision is the forced first hop (no_zero_hop)
  uint32_t b, d;       // out-row of the current node: first edge, degree
  uint32_t sb, sd;     // the same for the start node
};

__device__ __forceinline__ void walker_init(Walker& w, int32_t start, unsigned long long start_ext, int32_t start_orig,
                                            unsigned long long idx, uint32_t stream, bool no_zero_hop) {
  w.start = start;
  w.cur = start;
  w.b = w.sb = (uint32_t)start_ext;
  w.d = w.sd = (uint32_t)(start_ext >> 32);
  w.c0 = (uint32_t)start_orig;
  w.c1 = (uint32_t)idx;
  w.c2 = (uint32_t)((idx >> 32) & 0xFFFFu) | (stream << 16);
  w.k = 0;
  w.moves = 0;
  w.forced = no_zero_hop;
}

// Returns true when the walk has stopped (w.cur is the terminal).
// A step reads one 16-byte edge record {neighbour, the neighbour's first out-edge, its out-degree}: the row extent
// the *next* step needs comes with the neighbour's id, so a step costs one random line instead of two (the extent
// array was the second: 0.24 lines per step beyond L2 on R-MAT 22).
__device__ __forceinline__ bool walker_step(Walker& w, const uint4* __restrict__ walk_rec, double alpha, uint32_t k0,
                                            uint32_t k1) {
  uint32_t ws, wp;
  if ((w.k & 1u) == 0) {
    const Philox p = philox4x32_10(w.c0, w.c1, w.c2, w.k >> 1, k0, k1);
    ws = p.x[0];
    wp = p.x[1];
    w.w_stop2 = p.x[2];
    w.w_pick2 = p.x[3];
  } else {
    ws = w.w_stop2;
    wp = w.w_pick2;
  }
  w.k++;
  if (!w.forced) {
    if ((double)ws * (1.0 / 4294967296.0) < alpha) return true;  // Monte_Carlo.java:76-78
  }
  w.forced = false;
  if (w.d > 0) {
    const uint4 r = walk_rec[w.b + (uint32_t)(((unsigned long long)wp * w.d) >> 32)];  // :81-86
    w.cur = (int32_t)r.x;
    w.b = r.y;
    w.d = r.z;
  } else {  // :87-90 dead end: restart at the walk's start node
    w.cur = w.start;
    w.b = w.sb;
    w.d = w.sd;
  }
  w.moves++;
  return false;
}

// ------------------------------------------------------------------------------------------------
// walk plan: one entry per residue node, walk ranges as an exclusive prefix over entries
// ------------------------------------------------------------------------------------------------
// walks a residue entry starts, and the increment each of them carries
template <int VARIANT>
__device__ __forceinline__ bool plan_entry(double r, double alpha, double rsum, double nrw, unsigned long long* omega_i,
                                           double* incr) {
  if (!(r > 0.0) || !(nrw > 0.0)) return false;
  if (VARIANT == 0) {  // Fora_Whole_Graph.java:123,129-131
    if (!(rsum > 0.0)) return false;
    r *= (1.0 - alpha);
    const double x = r / rsum * nrw;
    *omega_i = (unsigned long long)ceil(x);
    const double a_i = x / (double)*omega_i;
    *incr = a_i / nrw * rsum;
  } else {  // Fora_Topk.java:157-159
    const double x = r * nrw;
    *omega_i = (unsigned long long)ceil(x);
    const double a_i = x / (double)*omega_i;
    *incr = a_i / nrw;
  }
  return *omega_i > 0;
}

template <int VARIANT>
__global__ __launch_bounds__(256) void k_mc_plan(uint32_t n, const double* __restrict__ res, double* __restrict__ target,
                                                  double alpha, double rsum, double nrw, double omega_dev,
                                                  const unsigned long long* __restrict__ out_ext,
                                                  const int32_t* __restrict__ new2old, WalkPlanRec* __restrict__ plan,
                                                  DevCounters* ctr, int parity, int next_cell,
                                                  const double* __restrict__ copy_src, double* __restrict__ copy_dst) {
  if (blockIdx.x == 0 && threadIdx.x == 0) ctr->mc_plan[next_cell] = 0ull;  // (engine.hpp: DevCounters::mc_plan)
  const uint32_t per = (n + gridDim.x - 1) / gridDim.x;
  const uint32_t lo = blockIdx.x * per;
  const uint32_t hi = lo + per < n ? lo + per : n;
  if (lo >= hi) return;
  if (omega_dev > 0.0) {
    // the budget from the residue sum on the device, with the host's own expressions (Fora_Topk.java:148,151 /
    // Fora_Whole_Graph.java:112-113): rsum = sum * (1 - alpha); nrw = (long long)(omega * rsum)
    if (blockIdx.x == 0 && threadIdx.x == 0) ctr->plan_sum[parity] = ctr->sum_out;  // for the round's selection header
    rsum = ctr->sum_out * (1.0 - alpha);
    const double nrw_d = omega_dev * rsum;
    nrw = (nrw_d == nrw_d && nrw_d > 0.0) ? (double)(long long)nrw_d : 0.0;
  }
  block_range_compact(
      lo, hi, &ctr->mc_plan[parity],
      [&](uint32_t v, unsigned long long* w) {
        double incr;
        return plan_entry<VARIANT>(res[v], alpha, rsum, nrw, w, &incr);
      },
      [&](uint32_t v, uint32_t pos, unsigned long long woff, unsigned long long) {
        unsigned long long w;
        WalkPlanRec r;
        r.inc = 0.0;
        (void)plan_entry<VARIANT>(res[v], alpha, rsum, nrw, &w, &r.inc);
        r.woff = woff;
        r.ext = out_ext[v];  // what a walk needs of its start node travels with the entry: the walk kernel reads
        r.node = (int32_t)v;  // entries as a stream and nothing else before a walk's first step
        r.orig = new2old[v];
        plan[pos] = r;
      });
  if (VARIANT == 0) {  // Fora_Whole_Graph.java:122,124-127: every residue entry credits alpha * r to its own reserve
    for (uint32_t v = lo + threadIdx.x; v < hi; v += 256) {
      const double r = res[v];
      if (r > 0.0) target[v] = target[v] + r * alpha;
    }
  }
  // top-k rounds: the estimate the walks add to starts as a copy of the push reserve (Fora_Topk.java:143) - taken here,
  // in the pass that reads the same range anyway, when the caller has no use for the old estimate any more
  if (copy_dst)
    for (uint32_t v = lo + threadIdx.x; v < hi; v += 256) copy_dst[v] = copy_src[v];
}

// k_mc_plan<1> with the device-side budget, in one pass (round 5; top-k rounds, Fora_Topk.java:148-159).  What the
// two-pass kernel did in 44 us for 2 M nodes - and the two sum kernels in front of it in 20 - happens here per tile of
// 2048 consecutive nodes, 8 per thread: every workgroup first adds up the `np` partial sums k_sum_partial has left (the
// same additions in the same order in every workgroup, so all of them derive the same budget; no k_sum_final launch),
// then a thread loads its 8 residues (and 8 reserves, when the estimate is to start as their copy: 16-byte loads),
// evaluates its entries once, and the tile is compacted with one workgroup scan and one packed atomic.
constexpr int kPlanItems = 8;
__global__ __launch_bounds__(256) void k_mc_plan_topk(uint32_t n, const double* __restrict__ res, double alpha,
                                                       double omega_dev, const double* __restrict__ partial, uint32_t np,
                                                       const unsigned long long* __restrict__ out_ext,
                                                       const int32_t* __restrict__ new2old, WalkPlanRec* __restrict__ plan,
                                                       DevCounters* ctr, int parity, int next_cell,
                                                       const double* __restrict__ copy_src, double* __restrict__ copy_dst) {
  __shared__ double s_red[4];
  __shared__ double s_sum;
  if (blockIdx.x == 0 && threadIdx.x == 0) ctr->mc_plan[next_cell] = 0ull;  // (engine.hpp: DevCounters::mc_plan)
  {
    double acc = 0.0;
    for (uint32_t i = threadIdx.x; i < np; i += 256) acc += partial[i];
    const double sum = block_sum_f64(acc, s_red);
    if (threadIdx.x == 0) s_sum = sum;
    __syncthreads();
  }
  const double sum = s_sum;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    ctr->sum_out = sum;
    ctr->plan_sum[parity] = sum;  // for the round's selection header
  }
  // the budget with the host's own expressions: rsum = sum * (1 - alpha); nrw = (long long)(omega * rsum)
  const double rsum = sum * (1.0 - alpha);
  const double nrw_d = omega_dev * rsum;
  const double nrw = (nrw_d == nrw_d && nrw_d > 0.0) ? (double)(long long)nrw_d : 0.0;
  const uint32_t tile = 256u * kPlanItems;
  const uint32_t n_tiles = (n + tile - 1) / tile;
  for (uint32_t tl = blockIdx.x; tl < n_tiles; tl += gridDim.x) {
    const uint32_t v0 = tl * tile + threadIdx.x * kPlanItems;
    double r[kPlanItems];
    if (v0 + kPlanItems <= n) {
      const double2* r2 = reinterpret_cast<const double2*>(res + v0);
#pragma unroll
      for (int i = 0; i < kPlanItems / 2; ++i) {
        const double2 x = r2[i];
        r[2 * i] = x.x;
        r[2 * i + 1] = x.y;
      }
      if (copy_dst) {
        const double2* s2 = reinterpret_cast<const double2*>(copy_src + v0);
        double2* d2 = reinterpret_cast<double2*>(copy_dst + v0);
        double2 y[kPlanItems / 2];
#pragma unroll
        for (int i = 0; i < kPlanItems / 2; ++i) y[i] = s2[i];
#pragma unroll
        for (int i = 0; i < kPlanItems / 2; ++i) d2[i] = y[i];
      }
    } else {
#pragma unroll
      for (int i = 0; i < kPlanItems; ++i) {
        r[i] = v0 + i < n ? res[v0 + i] : 0.0;
        if (copy_dst && v0 + i < n) copy_dst[v0 + i] = copy_src[v0 + i];
      }
    }
    bool take[kPlanItems];
    unsigned long long w[kPlanItems];
    double inc[kPlanItems];
#pragma unroll
    for (int i = 0; i < kPlanItems; ++i) {
      w[i] = 0;
      inc[i] = 0.0;
      take[i] = plan_entry<1>(r[i], alpha, rsum, nrw, &w[i], &inc[i]);
    }
    block_tile_compact<kPlanItems>(take, w, &ctr->mc_plan[parity], [&](int i, uint32_t pos, unsigned long long woff) {
      const uint32_t v = v0 + (uint32_t)i;
      WalkPlanRec rec;
      rec.inc = inc[i];
      rec.woff = woff;
      rec.ext = out_ext[v];
      rec.node = (int32_t)v;
      rec.orig = new2old[v];
      plan[pos] = rec;
    });
  }
}

// ------------------------------------------------------------------------------------------------
// walk kernel: one wave per workgroup, each with a contiguous share of the phase's walks.  A lane whose walk has
// stopped takes the wave's next walk at once, so lanes stay busy although walk lengths are geometric, and the wave only
// drains once, at the end of its share (with a workgroup per 1024 walks, as in round 2, every chunk ended in a tail
// of ~20 steps with few lanes walking: 36 G steps/s).  The entries of the next kWalkWindow walks are staged in LDS (a window);
// a refill reads LDS only.
// Round 4, tried and taken back: "every load carries a full wave".  A decision that issues no gather - the walk stops
// (alpha), or it stands on a dead end and restarts (52 % of R-MAT 22's nodes have no out-edge) - costs its lane the
// whole trip of the loop below, so a wave's load carries 42 of 64 lanes.  A form of the loop in which lanes take
// decisions (stop -> deposit -> next walk -> first decision; dead end -> restart -> next decision) until every lane
// holds a gather, and only then the wave loads, carried 60.4 lanes per load (counters walk_loads / walk_lanes) and ran
// at the SAME rate one query at a time (39.0 G steps/s, 2.54 ms per query against 2.6) and slower beside the sweeps,
// where a walk kernel has four waves per CU and three to four decision rounds per load are not hidden (24.5 against
// ~33 G steps/s, headline 312 against 329 queries/s): the rate is what the memory system gives this address stream,
// not a matter of how full the loads are (gpurun_out/r04b_bench.json).
// ------------------------------------------------------------------------------------------------
constexpr int kWalkWindow = 128;
constexpr uint32_t kWalkWavesBeside = 4;   // ... of a walk kernel that runs beside other queries' kernels
constexpr uint32_t kWalkWavesPerCu = 16;  // 4.3 KB of LDS each: room for another stream's workgroups on the CU

struct WalkWindow {  // LDS, one per wave
  unsigned long long woff[kWalkWindow + 1];
  unsigned long long ext[kWalkWindow];
  double inc[kWalkWindow];
  int32_t node[kWalkWindow];
  int32_t orig[kWalkWindow];
  uint8_t ent[kWalkWindow];  // staged entry of every walk of the window
};

// first entry whose walk range starts beyond walk x (woff[0] = 0, so >= 1), 64 probes per round
__device__ __forceinline__ uint32_t plan_upper_bound(const WalkPlanRec* __restrict__ plan, uint32_t n_src,
                                                     unsigned long long x, int lane) {
  uint32_t a = 0, b = n_src;  // the answer lies in [a, b]
  while (a < b) {
    const uint32_t step = (b - a + 63u) / 64u;
    const unsigned long long idx = (unsigned long long)a + (unsigned long long)lane * step;
    const bool gt = idx >= b || plan[idx].woff > x;
    const unsigned long long mask = __ballot(gt);
    if (mask == 0) {  // all 64 probes <= x
      a = (uint32_t)std::min<unsigned long long>((unsigned long long)a + 63ull * step + 1ull, b);
      continue;
    }
    const uint32_t f = (uint32_t)__builtin_ctzll(mask);
    const unsigned long long nb = (unsigned long long)a + (unsigned long long)f * step;
    if (f > 0) a = a + (f - 1u) * step + 1u;
    b = (uint32_t)std::min<unsigned long long>(nb, b);
    if (f == 0) b = a;
  }
  return a;
}

__global__ __launch_bounds__(64) void k_mc_walk(const WalkPlanRec* __restrict__ plan_rec,
                                                 const uint4* __restrict__ walk_rec, double* __restrict__ target,
                                                 double alpha, uint32_t k0, uint32_t k1, uint32_t stream,
                                                 int no_zero_hop, DevCounters* ctr, int parity) {
  // the plan kernel counted sources and walks into mc_plan[parity]; the query's totals grow by this phase
  const unsigned long long plan = ctr->mc_plan[parity];
  const uint32_t n_src = (uint32_t)(plan >> kPackShift);
  const unsigned long long n_walks = plan & kPackMask;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    ctr->walks_total += n_walks;
    ctr->sources_total += n_src;
  }
  __shared__ WalkWindow S;
  const int lane = threadIdx.x;
  // this wave's share: whole groups of 64 walks, so that a short phase still spreads over the grid
  const unsigned long long groups = (n_walks + 63) / 64;
  const unsigned long long per = (groups + gridDim.x - 1) / gridDim.x * 64;
  unsigned long long cursor = (unsigned long long)blockIdx.x * per;
  if (cursor >= n_walks) return;
  const unsigned long long w_hi = cursor + per < n_walks ? cursor + per : n_walks;
  uint32_t e = plan_upper_bound(plan_rec, n_src, cursor, lane) - 1u;  // the entry that holds walk `cursor`
  unsigned long long win_lo = cursor, win_end = cursor;
  unsigned long long steps_total = 0;
  Walker w;
  double inc = 0.0;
  bool walking = false;
  unsigned long long n_loads = 0, n_lanes = 0;  // (wave-uniform) loads issued, lanes they carried
  for (;;) {
    const unsigned long long need = __ballot(!walking);
    if (need && cursor < w_hi) {
      if (cursor >= win_end) {
        // stage the entries of walks [cursor, cursor + kWalkWindow): at most that many from e, which holds walk `cursor`
        // (every entry owns >= 1 walk)
        win_lo = cursor;
        win_end = cursor + kWalkWindow < w_hi ? cursor + kWalkWindow : w_hi;
        __syncthreads();  // refills of the window before have read it
        WalkPlanRec r[kWalkWindow / 64];
#pragma unroll
        for (int q = 0; q < kWalkWindow / 64; ++q) {
          const unsigned long long idx = (unsigned long long)e + (unsigned long long)(q * 64 + lane);
          if (idx < n_src) {
            r[q] = plan_rec[idx];
          } else {
            r[q].woff = n_walks;
            r[q].inc = 0.0;
            r[q].ext = 0ull;
            r[q].node = 0;
            r[q].orig = 0;
          }
        }
        unsigned long long last = n_walks;
        if (lane == 0 && (unsigned long long)e + kWalkWindow < n_src) last = plan_rec[(size_t)e + kWalkWindow].woff;
#pragma unroll
        for (int q = 0; q < kWalkWindow / 64; ++q) S.ent[lane * (kWalkWindow / 64) + q] = 0;
#pragma unroll
        for (int q = 0; q < kWalkWindow / 64; ++q) {
          const int j = q * 64 + lane;
          S.woff[j] = r[q].woff;
          S.ext[j] = r[q].ext;
          S.inc[j] = r[q].inc;
          S.node[j] = r[q].node;
          S.orig[j] = r[q].orig;
        }
        if (lane == 0) S.woff[kWalkWindow] = last;
        __syncthreads();
        // entry j > 0 that starts inside the window marks its first walk; a running maximum spreads the marks
#pragma unroll
        for (int q = 0; q < kWalkWindow / 64; ++q) {
          const int j = q * 64 + lane;
          if (j > 0 && r[q].woff >= win_lo && r[q].woff < win_end) S.ent[r[q].woff - win_lo] = (uint8_t)j;
        }
        __syncthreads();
        constexpr int kPer = kWalkWindow / 64;  // positions per lane
        uint32_t m[kPer];
#pragma unroll
        for (int q = 0; q < kPer; ++q) {
          m[q] = S.ent[lane * kPer + q];
          if (q > 0 && m[q] < m[q - 1]) m[q] = m[q - 1];
        }
        uint32_t run = m[kPer - 1];  // inclusive maximum over the lanes
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const uint32_t o = __shfl_up(run, d);
          if (lane >= d) run = run > o ? run : o;
        }
        uint32_t before = __shfl_up(run, 1);
        if (lane == 0) before = 0;
#pragma unroll
        for (int q = 0; q < kPer; ++q) S.ent[lane * kPer + q] = (uint8_t)(m[q] > before ? m[q] : before);
        __syncthreads();
      }
      const unsigned long long avail = win_end - cursor;
      const uint32_t rank = __popcll(need & ((1ull << lane) - 1ull));
      if (!walking && rank < avail) {
        const unsigned long long gidx = cursor + rank;
        const uint32_t j = S.ent[gidx - win_lo];
        const int32_t start = S.node[j];
        inc = S.inc[j];
        const unsigned long long sext = S.ext[j];
        walker_init(w, start, sext, S.orig[j], gidx - S.woff[j], stream, no_zero_hop != 0);
        if ((sext >> 32) == 0) {
          atomic_add_noret(&target[start], inc);  // Monte_Carlo.java:70-72 / :106-108
        } else {
          walking = true;
        }
      }
      const unsigned long long want = __popcll(need);
      cursor += want < avail ? want : avail;
      if (cursor >= win_end) {  // window used up: e becomes the entry of the next walk
        const uint32_t jl = S.ent[win_end - 1 - win_lo];
        e += S.woff[jl + 1] == win_end ? jl + 1u : jl;
      }
    }
    if (__ballot(walking) == 0) {
      if (cursor >= w_hi) break;
      continue;
    }
    const bool can_load = walking && w.d > 0;  // (a lane that neither stops nor stands on a dead end gathers)
    bool stopped = false;
    if (walking) {
      stopped = walker_step(w, walk_rec, alpha, k0, k1);
      if (stopped) {
        atomic_add_noret(&target[w.cur], inc);
        steps_total += w.moves;
        walking = false;
      }
    }
    const unsigned long long loaded = __ballot(can_load && !stopped);
    n_loads += loaded ? 1ull : 0ull;
    n_lanes += (unsigned long long)__popcll(loaded);
  }
  steps_total = wave_sum_u64(steps_total);
  if (lane == 0 && steps_total) atomic_add_u64(&ctr->walk_steps, steps_total);
  if (lane == 0 && n_loads) {
    atomic_add_u64(&ctr->walk_loads, n_loads);
    atomic_add_u64(&ctr->walk_lanes, n_lanes);
  }
}

// one walk per thread, terminals written out (walker parity tests, pprhip_random_walk_batch)
__global__ __launch_bounds__(256) void k_walk_batch(const int32_t* __restrict__ starts,
                                                     const unsigned long long* __restrict__ idx,
                                                     unsigned long long count,
                                                     const unsigned long long* __restrict__ out_ext,
                                                     const uint4* __restrict__ walk_rec,
                                                     const int32_t* __restrict__ new2old, double alpha, uint32_t k0,
                                                     uint32_t k1, uint32_t stream, int no_zero_hop,
                                                     int32_t* __restrict__ term, uint32_t* __restrict__ steps) {
  for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < count;
       i += (unsigned long long)gridDim.x * blockDim.x) {
    const int32_t s = starts[i];
    Walker w;
    const unsigned long long sext = out_ext[s];
    walker_init(w, s, sext, new2old[s], idx[i], stream, no_zero_hop != 0);
    if ((sext >> 32) != 0) {
      while (!walker_step(w, walk_rec, alpha, k0, k1)) {
      }
    }
    term[i] = w.cur;
    if (steps) steps[i] = w.moves;
  }
}

// edge records of the walk kernel, built once at graph lift
__global__ __launch_bounds__(256) void k_build_walk_rec(unsigned long long m, const int32_t* __restrict__ out_ci,
                                                         const unsigned long long* __restrict__ out_ext,
                                                         uint4* __restrict__ walk_rec) {
  for (unsigned long long e = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; e < m;
       e += (unsigned long long)gridDim.x * blockDim.x) {
    const int32_t u = out_ci[e];
    const unsigned long long ext = out_ext[u];
    walk_rec[e] = make_uint4((uint32_t)u, (uint32_t)ext, (uint32_t)(ext >> 32), 0u);
  }
}

__global__ void k_plan_single(int32_t src, double inc, unsigned long long n_walks,
                              const unsigned long long* __restrict__ out_ext, const int32_t* __restrict__ new2old,
                              WalkPlanRec* plan, DevCounters* ctr, int parity, int next_cell) {
  WalkPlanRec r;
  r.woff = 0ull;
  r.inc = inc;
  r.ext = out_ext[src];
  r.node = src;
  r.orig = new2old[src];
  plan[0] = r;
  ctr->mc_plan[next_cell] = 0ull;
  ctr->mc_plan[parity] = (1ull << kPackShift) | n_walks;
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
int init_kernels_walk() {  // loads this file's code object on the current device (see init_kernels_push)
  hipFuncAttributes fa;
  PPRHIP_CHECK_HIP(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&k_mc_walk)));
  return PPRHIP_OK;
}

int launch_build_walk_rec(pprhip_graph* g) {
  if (g->m == 0) return PPRHIP_OK;
  hipLaunchKernelGGL(k_build_walk_rec, dim3(4096), dim3(256), 0, g->stream, (unsigned long long)g->m, g->out_ci,
                     g->out_ext, reinterpret_cast<uint4*>(g->walk_rec));
  PPRHIP_CHECK_HIP(hipGetLastError());
  return PPRHIP_OK;
}

// the phase a plan belongs to picks its counter cell and its record buffer; the next walk kernel runs the latest plan
static WalkPlanRec* plan_rec_of(pprhip_graph* g, uint32_t phase) {
  return (g->mc_plan_rec2 && (phase & 1u)) ? g->mc_plan_rec2 : g->mc_plan_rec;
}

int launch_mc_plan(pprhip_graph* g, int variant, double alpha, double rsum, double nrw, double omega_dev, double* target,
                   const double* copy_src, double* copy_dst) {
  const uint32_t phase = g->mc_phase++;
  g->mc_last_plan = phase;
  const int cell = (int)(phase % 3u), next_cell = (int)((phase + 1u) % 3u);
  const uint32_t n = act_n(g);
  uint64_t b = ((uint64_t)n + 1023) / 1024;  // 1024 nodes per workgroup (fewer or more were slower)
  const uint64_t cap = g->sync ? 1024 : 16384;  // (a slot of a threaded batch: see launch_seed_list)
  const uint32_t grid = (uint32_t)(b > cap ? cap : (b < 1 ? 1 : b));
  const uint32_t np = g->sum_np;  // partial sums a launch_sum_partial has just left for this plan (0: none)
  g->sum_np = 0;
  if (variant == 1 && omega_dev > 0.0 && np > 0) {
    const uint32_t tiles = (n + 256u * kPlanItems - 1) / (256u * kPlanItems);
    hipLaunchKernelGGL(k_mc_plan_topk, dim3(std::max(1u, tiles)), dim3(256), 0, g->stream, n, g->residue, alpha, omega_dev,
                       g->partial, np, g->out_ext, g->new2old, plan_rec_of(g, phase), g->ctr, cell, next_cell, copy_src,
                       copy_dst);
  } else if (variant == 0)
    hipLaunchKernelGGL(k_mc_plan<0>, dim3(grid), dim3(256), 0, g->stream, n, g->residue, target, alpha, rsum, nrw,
                       omega_dev, g->out_ext, g->new2old, plan_rec_of(g, phase), g->ctr, cell, next_cell, copy_src, copy_dst);
  else
    hipLaunchKernelGGL(k_mc_plan<1>, dim3(grid), dim3(256), 0, g->stream, n, g->residue, target, alpha, rsum, nrw,
                       omega_dev, g->out_ext, g->new2old, plan_rec_of(g, phase), g->ctr, cell, next_cell, copy_src, copy_dst);
  PPRHIP_CHECK_HIP(hipGetLastError());
  return PPRHIP_OK;
}

// The walk count is only known on the device: a fixed grid of waves, each with an equal share of whatever the plan
// holds (g->walk_hint, the budget when the host knows it, only trims the grid of a short phase).
int launch_mc_walk(pprhip_graph* g, double alpha, uint64_t seed, uint32_t stream, int no_zero_hop, double* target) {
  // (walk_waves: a walk phase that runs beside other kernels leaves them room - the walks are bound by the memory
  // system from a few waves per CU on, tools/micro/chain_rate.hip)
  // a slot of a threaded batch shares the chip with fifteen others: few waves per CU, like a walk phase beside sweeps
  // (batched top-k on R-MAT 22: 851 queries/s at 16 waves per CU, 961 / 1 026 / 905 at 2 / 4 / 8)
  uint32_t grid = (uint32_t)g->n_cus * (g->walk_waves ? g->walk_waves
                                        : g->sync   ? kWalkWavesBeside
                                                    : kWalkWavesPerCu);
  if (g->walk_hint) grid = (uint32_t)std::min<unsigned long long>(grid, std::max<unsigned long long>((g->walk_hint + 63) / 64, 1ull));
  g->walk_hint = 0;
  hipLaunchKernelGGL(k_mc_walk, dim3(grid), dim3(64), 0, g->stream, plan_rec_of(g, g->mc_last_plan),
                     reinterpret_cast<const uint4*>(g->walk_rec), target, alpha, (uint32_t)seed, (uint32_t)(seed >> 32), stream,
                     no_zero_hop, g->ctr, (int)(g->mc_last_plan % 3u));
  PPRHIP_CHECK_HIP(hipGetLastError());
  return PPRHIP_OK;
}

int launch_walk_batch(pprhip_graph* g, const int32_t* d_starts, const uint64_t* d_idx, uint64_t count, double alpha,
                      uint64_t seed, uint32_t stream, int no_zero_hop, int32_t* d_term, uint32_t* d_steps) {
  if (count == 0) return PPRHIP_OK;
  uint64_t b = (count + 255) / 256;
  const uint32_t grid = (uint32_t)(b > 4096 ? 4096 : b);
  hipLaunchKernelGGL(k_walk_batch, dim3(grid), dim3(256), 0, g->stream, d_starts,
                     (const unsigned long long*)d_idx, (unsigned long long)count, g->out_ext, g->walk_rec, g->new2old, alpha,
                     (uint32_t)seed, (uint32_t)(seed >> 32), stream, no_zero_hop, d_term, d_steps);
  PPRHIP_CHECK_HIP(hipGetLastError());
  return PPRHIP_OK;
}

int launch_mc_pure(pprhip_graph* g, int32_t src, uint64_t n_walks, double alpha, uint64_t seed, double inc,
                   double* target) {
  const uint32_t phase = g->mc_phase++;
  g->mc_last_plan = phase;
  hipLaunchKernelGGL(k_plan_single, dim3(1), dim3(1), 0, g->stream, src, inc, (unsigned long long)n_walks, g->out_ext,
                     g->new2old, plan_rec_of(g, phase), g->ctr, (int)(phase % 3u), (int)((phase + 1u) % 3u));
  PPRHIP_CHECK_HIP(hipGetLastError());
  return launch_mc_walk(g, alpha, seed, 0, 0, target);
}

}  // namespace pprhip
