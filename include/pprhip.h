/*
 * pprhip.h — C ABI of the MI355X-native Personalized-PageRank engine.
 *
 * This is the drop-in boundary for the hot path of
 * joezie/Personalized-PageRank-Algorithms-on-Neo4j (FORA single-source / top-k and
 * All-Pair-Backward-Search).  The reference has no FFI of its own; the seam is its three Java
 * interfaces plus the one-shot graph lift.  Every entry point below names the reference
 * interface it stands behind (paths relative to
 * /root/reference/src/main/java/joezie/fora_neo4j/).  INTEGRATION.md shows the JNI stub that
 * binds these symbols from the reference's Java classes.
 *
 * Conventions
 *   - plain C, opaque handles, caller-owned output buffers, no torch / STL types;
 *   - every function returns 0 (PPRHIP_OK) or a negative PPRHIP_ERR_* code; the message for the
 *     calling thread's last error is pprhip_last_error();
 *   - node ids are the dense mapped ids 0..n-1 (the harness assumes this too: Gen_Util.java:99-107);
 *   - all arithmetic on reserve / residue is IEEE double, as in the reference;
 *   - one handle belongs to one GPU and is used by one thread at a time (the reference objects
 *     are single-threaded and keep per-query state in fields, e.g. Forward_Push.java:33-43);
 *   - output pointers documented "may be NULL" leave the result resident in HBM; fetch it later
 *     with pprhip_get_reserve / pprhip_get_residue;
 *   - there is NO CPU fallback: every compute entry point fails with PPRHIP_ERR_NO_DEVICE when
 *     no gfx950 device is usable.
 *
 * Environment.  libpprhip.so reads these six variables and no others:
 *   PPRHIP_HOST_THREADS=<n>      host threads of the graph lift and the index finalisation (default: the CPU affinity /
 *                                cgroup quota of the process, at most 64)
 *   PPRHIP_BATCH_WORKSPACES=<n>  query workspaces of the batch driver, 16-48 (default 32; 0.33 GB each at R-MAT 22)
 *   PPRHIP_BATCH_THREADS=0|1     batched top-k / All-Pair's third tier: one host thread per slot (default 1) or one in all
 *   PPRHIP_SHARD_CUT=count|work  target ranges of the sharded All-Pair: equal counts, or cut by measured work (default:
 *                                by work when equal counts would leave a rank more than 1.15 x the mean)
 *   PPRHIP_COMM_TIMEOUT_S=<s>    time limit of an exchange between ranks (default 1800)
 *   PPRHIP_RCCL_LIB=<path>       the librccl.so to load on first multi-GPU use (default: the loader's search)
 * (HIP_FORCE_DEV_KERNARG=1, a HIP runtime variable, is what the launchers set before the runtime starts: INTEGRATION.md.)
 * Fault injection, layout / driver variants, diagnostics on stderr and the measurement switches of the A/B runs exist only
 * in libpprhip_hooks.so - the same sources built with -DPPRHIP_TEST_HOOKS (make builds both) - which the tests that need
 * them and tools/exp load; the product ignores those variables.
 */
#ifndef PPRHIP_H
#define PPRHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PPRHIP_OK 0
#define PPRHIP_ERR_INVALID (-1)   /* bad argument (null pointer, id out of range, k < 1 ...) */
#define PPRHIP_ERR_NO_DEVICE (-2) /* no usable HIP device / device index out of range */
#define PPRHIP_ERR_HIP (-3)       /* a HIP runtime call failed; see pprhip_last_error() */
#define PPRHIP_ERR_OOM (-4)       /* host or device allocation failed */
#define PPRHIP_ERR_IO (-5)        /* file could not be read / parsed / written */
#define PPRHIP_ERR_STATE (-6)     /* call sequence error (e.g. topk round before reset) */

#define PPRHIP_VERSION 100

typedef struct pprhip_graph pprhip_graph_t;       /* device-resident CSR pair + per-query workspace */
typedef struct pprhip_edgelist pprhip_edgelist_t; /* host edge list produced by the ingest helpers */
typedef struct pprhip_index pprhip_index_t;       /* all-pair inverted index (host, CSR by source) */
typedef struct pprhip_results pprhip_results_t;   /* device-resident result vectors of a batched call (q x n doubles) */
typedef struct pprhip_comm pprhip_comm_t;         /* one rank of a multi-GPU group (RCCL communicator + its graph replica) */

/* Counters every compute call fills; all device-side counts, not estimates. */
typedef struct pprhip_stats {
  uint64_t pops;           /* frontier nodes pushed in sparse levels */
  uint64_t edge_pushes;    /* edges traversed in sparse levels */
  uint64_t enqueues;       /* nodes appended to a next frontier */
  uint64_t dead_end_pops;  /* pushed nodes with out-degree 0 (mass returned to the source) */
  uint64_t dense_nodes;    /* nodes pushed inside dense pull sweeps */
  uint32_t levels;         /* frontier levels run (sparse + dense) */
  uint32_t dense_levels;   /* levels run as dense pull sweeps */
  uint32_t rounds;         /* FORA: threshold rounds run (1 + halvings); top-k: delta rounds */
  uint32_t xl_targets;     /* All-Pair: searches that outgrew a workspace's lists and ran in the full-size pass */
  uint64_t mc_sources;     /* residue entries that started walks */
  uint64_t walks;          /* random walks run */
  uint64_t walk_steps;     /* edges followed by all walks (dead-end restarts included) */
  uint64_t select_passes;  /* radix-select passes over the reserve vector */
  double rsum;             /* sum of residues the walk budget was derived from */
  double rmax_final;       /* last push threshold used */
  double omega;            /* walk budget at rsum = 1 */
  double kth_value;        /* top-k: k-th largest estimate (0 when fewer than k entries) */
  double push_ms;          /* HIP-event time of the push phase */
  double mc_ms;            /* HIP-event time of the walk phase */
  double select_ms;        /* HIP-event time of the top-k selection */
  double total_ms;         /* HIP-event time of the whole call (device work only) */
  uint64_t push_bytes;     /* algorithmic bytes of the push phase (DESIGN.md byte model) */
  uint64_t mc_bytes;       /* algorithmic bytes of the walk phase */
  uint64_t select_bytes;   /* algorithmic bytes of the selection */
  double dominant_kernel_ms;    /* summed duration of the dominant kernel's launches */
  uint64_t dominant_kernel_bytes; /* algorithmic bytes those launches moved */
  uint32_t dominant_kernel_launches;
  uint32_t dominant_kernel_id;  /* PPRHIP_KERNEL_* */
  /* the same three figures for every kernel class, indexed by PPRHIP_KERNEL_* (HIP events on the
   * engine's stream around each launch; a sparse batch counts as one launch of class 2) */
  double class_ms[8];
  uint64_t class_bytes[8];
  uint32_t class_launches[8];
  uint64_t dense_edges;    /* out-edges of the nodes pushed inside dense pull sweeps: the edges a sweep had to serve,
                            * against dense_levels * m edges swept (useful-edge fraction of the sweeps) */
  uint64_t sweep_min_bytes; /* compulsory bytes of the dense sweeps: every byte a sweep has to read or write counted ONCE
                            * (index stream, each gathered contribution line once, row sums out and in, next
                            * contributions, the busy queries' residues) - a lower bound of the sweeps' memory traffic,
                            * whereas push_bytes' SURVEY 8(d) model counts one gather per edge and query */
  uint64_t walk_loads;      /* load instructions the walk kernel's waves issued (one 16-byte edge record per lane) ... */
  uint64_t walk_load_lanes; /* ... and the lanes those loads carried: 64 per load = every load a full wave */
} pprhip_stats_t;

#define PPRHIP_KERNEL_NONE 0
#define PPRHIP_KERNEL_DENSE_PULL 1
#define PPRHIP_KERNEL_SPARSE_PUSH 2
#define PPRHIP_KERNEL_WALK 3
#define PPRHIP_KERNEL_BACKWARD_BATCH 4
#define PPRHIP_KERNEL_DENSE_PULL_BATCH 5 /* one dense level for up to PPRHIP_BATCH queries */
#define PPRHIP_KERNEL_QUERY_SETUP 6      /* the short kernels around a query's phases: clearing its workspace, seeding a
                                          * round's frontier, residue sums, the walk plan, top-k selection (one "launch" =
                                          * one such group) */
#define PPRHIP_BATCH 16                  /* queries in flight in pprhip_fora_batch_single_source */

/* Engine tuning: the deterministic replacement of the reference's wall-clock push/walk balance
 * (Fora_Whole_Graph.java:35,75-79,93-103) and the sparse/dense switch.  Zero means "default". */
typedef struct pprhip_tuning {
  double c_walk_ns;        /* modelled cost of one random walk (reference constant: 400 ns) */
  double c_edge_ns;        /* modelled cost of one sparse edge push */
  double c_pop_ns;         /* modelled cost of one frontier pop */
  double c_level_ns;       /* modelled fixed cost of one level */
  double c_dense_edge_ns;  /* modelled cost per edge of a dense pull sweep */
  double c_dense_node_ns;  /* modelled cost per node of a dense pull sweep */
  double dense_frac;       /* a level runs dense when frontier_edges + frontier_nodes >= dense_frac * m */
  int32_t max_rounds;      /* cap on push rounds in auto mode (default 24) */
  int32_t max_halvings;    /* halvings of rmax one round may be followed by (default 6) */
  double halving_ratio;    /* a round is followed by 1 + k halvings when the modelled walk cost is still
                            * >= halving_ratio^k times the push cost so far (default 2) */
  int32_t prior_levels;    /* the first round starts below rmax0 by as many halvings as keep the a-priori walk
                            * bound c_walk * omega * (1 - alpha) * rmax * m >= prior_levels dense levels' cost
                            * (default 16; negative: always start at rmax0) */
  int32_t gs_blocks;       /* dense sweeps run block by block (Gauss-Seidel): rows are cut into gs_blocks blocks of equal
                            * in-edge count and a block reads the contributions the blocks before it have just written
                            * (default 2; 1: plain Jacobi sweeps) */
  double gs_frac;          /* ... while the frontier holds at least gs_frac * m edges + nodes (default 0.1, batch profile 0.05); thinner
                            * dense levels run as Jacobi sweeps */
} pprhip_tuning_t;

/* Parameters Algo_Conf derives (Algo_Conf.java:29-81). */
typedef struct pprhip_fora_conf {
  double alpha;      /* stop probability */
  double delta;      /* reserve threshold: 1/n (whole graph) or 1/k (top-k start) */
  double pfail;      /* failure probability: 1/n, or 1/n^2/ln(n div k) for top-k */
  double rsum;       /* initial residue sum: 1.0 */
  double min_delta;  /* top-k only: 1/n */
  int32_t k;         /* top-k only */
  uint32_t n;        /* node_amount */
  uint64_t m;        /* rel_amount */
} pprhip_fora_conf_t;

/* ---------------------------------------------------------------- errors / build info */
const char* pprhip_last_error(void);
int pprhip_version(void);
int pprhip_device_count(int* count_out);
/* Kernel-class times (pprhip_stats_t.class_ms, dominant_kernel_*) are measured with HIP events around every group of
 * launches; between the short kernels of the latency-bound paths those records cost 2-8 %, so they are an
 * option of the process: off unless PPRHIP_KERNEL_TIMER=1 is set or this is called with on != 0.  Off, the groups are
 * still counted (class_launches, class_bytes).  on == 2: only the dense sweeps are timed (PPRHIP_KERNEL_DENSE_PULL and
 * _DENSE_PULL_BATCH: two records per sweep of a millisecond), every other class counted - what bench.py's timed region
 * uses.  Not to be switched while a call is in flight.  Returns the old state (0, 1 or 2). */
int pprhip_set_kernel_timing(int on);
void pprhip_tuning_default(pprhip_tuning_t* t);
/* Cost-model constants for pprhip_fora_batch_single_source: a dense level costs a query 1/PPRHIP_BATCH
 * of a sweep, so the model values it lower and lets levels go dense earlier. */
void pprhip_tuning_batch(pprhip_tuning_t* t);
/* The batch profile for a call of q queries (Gen_Util.java:208-232 runs 50; sharded over 8 GPUs a call holds 6-7): with
 * fewer than PPRHIP_BATCH - 1 busy columns a dense level costs each query more, so the dense constants, dense_frac and
 * gs_frac grow with 14.5 / q, up to the single-query profile's values.  q >= 15: pprhip_tuning_batch.  The caller sets it
 * (pprhip_graph_set_tuning) before the call; every query of the call then is pprhip_fora_single_source under it. */
void pprhip_tuning_batch_for(int q, pprhip_tuning_t* t);

/* ---------------------------------------------------------------- parameter derivation (a10) */
/* Algo_Conf.set_conf_fora_whole_graph (Algo_Conf.java:45-53): delta = pfail = 1/n, rsum = 1. */
int pprhip_conf_fora_whole_graph(uint32_t n, uint64_t m, double alpha, pprhip_fora_conf_t* conf);
/* Algo_Conf.set_conf_fora_topk (Algo_Conf.java:71-81): min_delta = 1/n, delta = 1/k,
 * pfail = 1/n/n/ln(n div k) with integer division. */
int pprhip_conf_fora_topk(uint32_t n, uint64_t m, int k, double alpha, pprhip_fora_conf_t* conf);
/* Fora_Whole_Graph.java:86-87: rmax0 = eps*sqrt(delta/3/m/ln(2/pfail))/(1-alpha),
 * omega = (eps+2)*ln(2/pfail)/eps^2/delta. */
int pprhip_fora_whole_params(const pprhip_fora_conf_t* conf, double eps, double* rmax0, double* omega);
/* Fora_Topk.java:110-125,133 for one delta: eps' = eps/2; min_rmax, scaled rmax, omega. */
int pprhip_fora_topk_params(const pprhip_fora_conf_t* conf, double eps, double delta,
                            double* min_rmax, double* rmax_scaled, double* omega);

/* ---------------------------------------------------------------- graph ingest (host side) */
/* Seeded R-MAT edge list (Graph500 quadrants .57/.19/.19/.05), m = edge_factor << scale, labels
 * scrambled by a seeded permutation, parallel edges and self loops kept (the reference's
 * container is a multigraph: Forward_Push.java:119-139 visits every relationship). */
int pprhip_rmat_edges(int scale, int edge_factor, uint64_t seed, int32_t* src_out, int32_t* dst_out);
/* neo4j-admin-import CSVs (":ID,name" / ":START_ID,:END_ID,:TYPE"), id = node row index
 * (dataset/got/GOT_Nodes.csv, GOT_Rels.csv). */
int pprhip_edgelist_from_neo4j_csv(const char* nodes_csv, const char* rels_csv, pprhip_edgelist_t** out);
/* A Neo4j 3.x store directory read without a JVM: neostore.nodestore.db (15-byte records: in-use
 * bit, first relationship of the chain, dense flag) and neostore.relationshipstore.db (34-byte
 * records: first/second node, type, the four chain pointers); what PPR.createDb + setupAdjMatrix
 * get through the Neo4j kernel (PPR.java:52-60,136-152).  Node id = record id.  The adjacency
 * order is the relationship-chain order HeavyGraph sees.  Dense nodes (50 relationships or more: the node record
 * points to a chain of 25-byte relationship-group records, neostore.relationshipgroupstore.db, with separate
 * outgoing / incoming / loop chains per type) are read as well; a group of the wrong owner or a missing group store
 * is PPRHIP_ERR_IO. */
int pprhip_edgelist_from_neo4j_store(const char* store_dir, pprhip_edgelist_t** out);
int pprhip_edgelist_info(const pprhip_edgelist_t* e, uint32_t* n, uint64_t* m);
/* Out- (incoming = 0) or in-adjacency (incoming = 1) of an edge list in the order HeavyGraph
 * holds it: chain order for a store, newest relationship first for import CSVs. */
int pprhip_edgelist_build_csr(const pprhip_edgelist_t* e, int incoming, uint32_t* row_ptr_out /* n+1 */,
                              int32_t* col_idx_out /* m */);
int pprhip_edgelist_edges(const pprhip_edgelist_t* e, const int32_t** src, const int32_t** dst);
const char* pprhip_edgelist_node_name(const pprhip_edgelist_t* e, uint32_t id);
void pprhip_edgelist_destroy(pprhip_edgelist_t* e);
/* CSR by `key` (stable counting sort).  newest_first != 0 lists each row in descending edge
 * index, the order HeavyGraph inherits from Neo4j's relationship chains for got.db. */
int pprhip_csr_build(uint32_t n, uint64_t m, const int32_t* key, const int32_t* val, int newest_first,
                     uint32_t* row_ptr_out /* n+1 */, int32_t* col_idx_out /* m */);

/* ---------------------------------------------------------------- graph lift (a11) */
/* Replaces PPR.setupAdjMatrix + set_configuration (PPR.java:121-152): uploads the out- and
 * in-adjacency once.  in_* may be NULL, then the in-CSR is derived from the out-CSR. */
int pprhip_graph_create(uint32_t n, uint64_t m, const uint32_t* out_row_ptr, const int32_t* out_col_idx,
                        const uint32_t* in_row_ptr, const int32_t* in_col_idx, int device,
                        pprhip_graph_t** graph_out);
void pprhip_graph_destroy(pprhip_graph_t* g);
/* The host half of the lift alone, without a device: validation, the internal vertex order (nodes with in-edges first,
 * then out-degree descending, ties by id), both adjacencies in that order, the pull sweep's row-start flags and the
 * sliced copy of the in-adjacency - what pprhip_graph_create uploads.  It runs on `threads` host threads (0 = what the
 * process may use) and yields the same bytes with any count.  For inspection and tests; HeavyGraph's jagged arrays
 * (Diss. p.25) are the reference's counterpart of these arrays. */
typedef struct pprhip_lift pprhip_lift_t;
enum {
  PPRHIP_LIFT_NEW2OLD = 0,       /* int32[n] */
  PPRHIP_LIFT_OLD2NEW = 1,       /* int32[n] */
  PPRHIP_LIFT_OUT_ROW_PTR = 2,   /* uint32[n + 1] */
  PPRHIP_LIFT_OUT_COL_IDX = 3,   /* int32[m] */
  PPRHIP_LIFT_IN_ROW_PTR = 4,    /* uint32[n + 1] */
  PPRHIP_LIFT_IN_COL_IDX = 5,    /* int32[m] */
  PPRHIP_LIFT_NZ_ROWS = 6,       /* int32[]: nodes with in-edges */
  PPRHIP_LIFT_ZIN_ROWS = 7,      /* int32[]: nodes without in-edges that have out-edges */
  PPRHIP_LIFT_ROW_START_FLAGS = 8,   /* uint8[]: bit e = in-edge e is the first of its row */
  PPRHIP_LIFT_CHUNK_STARTS = 9,      /* uint32[]: rows that start before each 512-edge chunk */
  PPRHIP_LIFT_CROSS_BITS = 10,       /* uint64[]: bit j = row j holds the last edge of a chunk */
  PPRHIP_LIFT_SLICE_EDGE_BASE = 11,  /* uint64[S + 1] (empty: no sliced copy) */
  PPRHIP_LIFT_SLICE_SEG_BASE = 12,   /* uint64[S + 1] */
  PPRHIP_LIFT_SLICED_COL_IDX = 13,   /* int32[m], slice-major */
  PPRHIP_LIFT_SLICED_FLAGS = 14,     /* uint8[]: bit e = edge e of the sliced copy starts a segment */
  PPRHIP_LIFT_SLICED_CHUNK_STARTS = 15, /* uint32[] */
  PPRHIP_LIFT_SEG_ROW = 16,          /* uint32[segments]: row ordinal */
  PPRHIP_LIFT_SEG_OFF = 17,          /* uint32[segments]: first edge */
  /* the row-panel copy of the in-adjacency the single-query sweep walks (graphs from 2^26 edges on; they have no sliced
   * copy): panels of 8 192 consecutive rows with in-edges, a panel's in-edges sorted by (source, row); a panel of more
   * than 32 768 edges is cut into parts of equal edge counts; every part (item) padded to whole turns of 8 192 edges
   * with (0, 0xffff), a wave's 512 edges of a turn stored lane by lane (lane l: the edges l, l + 64, ...).  All empty
   * when the graph has none. */
  PPRHIP_LIFT_PANEL_SIZES = 18,      /* uint64[4]: panels, items, doubles of partial sums, edges with padding */
  PPRHIP_LIFT_PANEL_SRC = 19,        /* int32[edges]: sources */
  PPRHIP_LIFT_PANEL_ROW = 20,        /* uint16[edges]: row ordinal - first ordinal of the panel; 0xffff: padding */
  PPRHIP_LIFT_PANEL_ITEMS = 21,      /* uint32[items][4]: first edge / 8192, turns, panel, offset of the item's sums */
  PPRHIP_LIFT_PANEL_DESC = 22,       /* uint32[panels][4]: offset of the panel's sums, parts, rows, offset of their sums 32 at a time (or ~0) */
  PPRHIP_LIFT_PANEL_ITEM0 = 23       /* uint32[panels + 1]: first item of every panel */
};
int pprhip_graph_lift_host(uint32_t n, uint64_t m, const uint32_t* out_row_ptr, const int32_t* out_col_idx,
                           const uint32_t* in_row_ptr, const int32_t* in_col_idx, int threads,
                           pprhip_lift_t** lift_out);
/* *data_out points into the lift (valid until pprhip_lift_destroy) */
int pprhip_lift_array(const pprhip_lift_t* lift, int which, const void** data_out, uint64_t* bytes_out);
void pprhip_lift_destroy(pprhip_lift_t* lift);
/* A handle keeps the workspaces of every entry point it has served (allocating gigabytes per call would cost more than
 * the call): on first use of the batched entry points 16 query slots + interleaved arrays (R-MAT 22: 6.7 GB), on first
 * use of All-Pair the in-edge records, the dense per-workgroup vectors and the record buffer (R-MAT 22: 27 GB, R-MAT 24:
 * 80 GB).  pprhip_graph_release hands them back between phases of a job; the next call of that entry point allocates
 * them again.  The CSR pair, the walk records and the single-query workspace stay.  Not to be called while another
 * thread uses the handle. */
#define PPRHIP_RELEASE_ALL_PAIR 1u
#define PPRHIP_RELEASE_BATCH 2u
int pprhip_graph_release(pprhip_graph_t* g, unsigned what);
int pprhip_graph_info(const pprhip_graph_t* g, uint32_t* n, uint64_t* m, int* device);
/* HBM of the handle's device: bytes free and in all (hipMemGetInfo) - what a job holds at a point of its run is
 * total - free; bench.py reports it for the All-Pair samples (per rank in the sharded run). */
int pprhip_device_memory(const pprhip_graph_t* g, uint64_t* free_bytes, uint64_t* total_bytes);
int pprhip_graph_set_tuning(pprhip_graph_t* g, const pprhip_tuning_t* t);
int pprhip_graph_get_tuning(const pprhip_graph_t* g, pprhip_tuning_t* t);
/* Results of the last compute call that left them in HBM. */
int pprhip_get_reserve(pprhip_graph_t* g, double* reserve_out /* n */);
int pprhip_get_residue(pprhip_graph_t* g, double* residue_out /* n */);

/* ---------------------------------------------------------------- forward push (a1, a2) */
/* Forward_Push.computeWholeGraphPPR(Long s, Object rmax) (Forward_Push.java:63-142), run as a
 * frontier-synchronous schedule.  reserve_out / residue_out / rsum_out / stats may be NULL. */
int pprhip_forward_push(pprhip_graph_t* g, int32_t src, double alpha, double rmax, double* reserve_out,
                        double* residue_out, double* rsum_out, pprhip_stats_t* stats);
/* Forward_Push.forward_push_topk (Forward_Push.java:144-250): resumable two-threshold push.
 * reset = new Forward_Push(rsum = 1) + Q = {s}; each round resumes from the parked queue. */
int pprhip_fwdpush_topk_reset(pprhip_graph_t* g, int32_t src, double alpha);
int pprhip_fwdpush_topk_round(pprhip_graph_t* g, double min_rmax, double rmax, double* rsum_out,
                              pprhip_stats_t* stats);

/* ---------------------------------------------------------------- random walks (a3, a4) */
/* Monte_Carlo.random_walk / random_walk_no_zero_hop (Monte_Carlo.java:60-133) on the engine's
 * counter-based generator: walk (seed, stream, start, walk_idx) is a pure function, so the
 * terminal of every walk can be compared one by one. */
int pprhip_random_walk_batch(pprhip_graph_t* g, const int32_t* starts, const uint64_t* walk_idx, uint64_t count,
                             double alpha, uint64_t seed, uint32_t stream, int no_zero_hop,
                             int32_t* terminals_out, uint32_t* steps_out /* may be NULL */);

/* ---------------------------------------------------------------- FORA (a5, a6, a7) */
/* Fora_Whole_Graph.computeWholeGraphPPR(Long s, Object eps) (Fora_Whole_Graph.java:82-146).
 * n_rounds > 0 runs exactly that many threshold rounds (rmax0, rmax0/2, ...); 0 picks the thresholds
 * with the deterministic cost model of the tuning struct (the stand-in for the reference's wall-clock
 * loop, :93-103: first threshold, halvings per round, when to stop).  reserve_out may be NULL. */
int pprhip_fora_single_source(pprhip_graph_t* g, int32_t src, double eps, const pprhip_fora_conf_t* conf,
                              uint64_t seed, int n_rounds, double* reserve_out, pprhip_stats_t* stats);
/* Fora_Topk.computeTopKPPR + getTopKNodeIds (Fora_Topk.java:82-199).  Writes at most `cap`
 * (id, value) pairs ordered by value descending then id ascending; *n_out is the number the
 * reference's rule selects (all entries >= the k-th value, can exceed k on ties) and can exceed
 * cap.  reserve_out may be NULL.  After the call pprhip_get_reserve returns the estimate of the last round (what
 * reserve_out receives); the push state behind it (pprhip_get_residue, a following pprhip_fwdpush_topk_round) is
 * undefined: the engine may have pushed one threshold further than the last round the reference would run
 * (the next round's push runs ahead, beside the walks, and is dropped when the round is not needed). */
int pprhip_fora_topk(pprhip_graph_t* g, int32_t src, double eps, const pprhip_fora_conf_t* conf, uint64_t seed,
                     int32_t* ids_out, double* vals_out, int cap, int* n_out, double* reserve_out,
                     pprhip_stats_t* stats);
/* Algo_Util.kth_ppr + retrieveTopK (Algo_Util.java:32-53, Fora_Topk.java:186-199) over the
 * reserve vector currently in HBM (entries == 0 do not exist in the reference's map). */
int pprhip_topk_select(pprhip_graph_t* g, int k, int32_t* ids_out, double* vals_out, int cap, int* n_out,
                       double* kth_out, pprhip_stats_t* stats);
/* Monte_Carlo.computeWholeGraphPPR (Monte_Carlo.java:136-158): omega = 3 ln(2/pfail)/eps^2/delta walks. */
int pprhip_monte_carlo(pprhip_graph_t* g, int32_t src, double eps, const pprhip_fora_conf_t* conf, uint64_t seed,
                       double* ppr_out, pprhip_stats_t* stats);
/* q single-source FORA queries (the loop of Gen_Util.java:208-232 over Fora_Whole_Graph.
 * computeWholeGraphPPR), up to PPRHIP_BATCH of them in flight on this GPU.  Each query runs the
 * algorithm of pprhip_fora_single_source unchanged (same levels, thresholds, round count and, for
 * the same seed, the same walks: results agree up to the order of fp64 additions); dense levels of
 * concurrent queries share one sweep over the in-CSR, whose gathers fetch one 128-byte line per
 * vertex holding the contributions of all queries in flight.
 * reserve_out: q*n doubles (query-major) or NULL.  k > 0 additionally selects each query's top-k by
 * Algo_Util.kth_ppr's rule into ids_out/vals_out (q*k, rows padded with id -1 / value 0) and
 * n_out[i] (entries >= the k-th value; may exceed k on ties, NULL allowed).  per_query: q entries
 * or NULL; stats_sum: counters summed, kernel-class times of the whole call. */
int pprhip_fora_batch_single_source(pprhip_graph_t* g, const int32_t* srcs, int q, double eps,
                                    const pprhip_fora_conf_t* conf, uint64_t seed, int n_rounds,
                                    double* reserve_out, int k, int32_t* ids_out, double* vals_out, int* n_out,
                                    pprhip_stats_t* per_query, pprhip_stats_t* stats_sum);
/* Device-resident result store: the whole-graph vectors of up to `capacity` queries of a batched call stay in HBM
 * (capacity * n doubles; R-MAT 22, 128 queries: 4.3 GB of 288 GB), so that every query's getWholeGraphPPR()
 * (Whole_Graph_Util_Interface.java:8, read by Gen_Util.java:309) can still be served after its slot has been
 * reused, without moving q * n doubles over PCIe inside the call.  Slot i holds query i of the last call that
 * was given the store. */
int pprhip_results_create(pprhip_graph_t* g, int capacity, pprhip_results_t** results_out);
void pprhip_results_destroy(pprhip_results_t* r);
int pprhip_results_info(const pprhip_results_t* r, int* capacity, int* count, uint32_t* n);
/* vector of query i, caller's ids (n doubles) */
int pprhip_results_fetch(pprhip_results_t* r, int i, double* reserve_out);
/* sum of the vector of query i, computed on the device (a cheap mass check: 1 up to rounding when walks ran) */
int pprhip_results_sum(pprhip_results_t* r, int i, double* sum_out);
/* pprhip_fora_batch_single_source with the vectors kept in `keep` (q <= capacity); reserve_out may still be given. */
int pprhip_fora_batch_single_source_resident(pprhip_graph_t* g, const int32_t* srcs, int q, double eps,
                                             const pprhip_fora_conf_t* conf, uint64_t seed, int n_rounds,
                                             pprhip_results_t* keep, double* reserve_out, int k, int32_t* ids_out,
                                             double* vals_out, int* n_out, pprhip_stats_t* per_query,
                                             pprhip_stats_t* stats_sum);

/* The batched driver behind a submit / wait pair (an extension: the reference's harness is synchronous,
 * Gen_Util.java:208-232).  A synchronous call ends with a drain - its last queries finish at different times and a
 * sweep costs the same for 2 busy columns as for 16 - so calls of PPR.java:179's 50 queries reach 0.90 of the rate of
 * long calls.  A stream keeps a driver thread on the handle: the slots a submission's last queries leave take the next
 * submission's first ones.  Every query runs as pprhip_fora_batch_single_source would run it (same seed and tuning,
 * same result); per submission: top-k blocks (k > 0: ids_out / vals_out [q * k], n_out [q] or NULL), and / or the
 * vectors in a device-resident store (`keep` slots keep_first .. keep_first + q - 1).  While a stream is open every
 * other entry point on the handle returns PPRHIP_ERR_STATE (one handle, one thread: here the driver's); outputs of a
 * submission may be read after its wait, a store after the close.  submit / wait may be called from any one thread. */
typedef struct pprhip_stream pprhip_stream_t;
int pprhip_fora_stream_open(pprhip_graph_t* g, double eps, const pprhip_fora_conf_t* conf, int k,
                            pprhip_stream_t** stream_out);
int pprhip_fora_stream_submit(pprhip_stream_t* s, const int32_t* srcs, int q, uint64_t seed, pprhip_results_t* keep,
                              int keep_first, int32_t* ids_out, double* vals_out, int* n_out, uint64_t* ticket_out);
/* blocks until every query of the submission has finished; stats_sum: its counters (total_ms = submit to finish).
 * A ticket can be waited for once; submissions nobody waits for are finished and released by the close. */
int pprhip_fora_stream_wait(pprhip_stream_t* s, uint64_t ticket, pprhip_stats_t* stats_sum);
/* finishes everything submitted, ends the driver thread and frees the stream.  (pprhip_graph_destroy on a graph whose
 * stream is still open ends the driver itself; the stream object then only remains to be freed by this call.) */
int pprhip_fora_stream_close(pprhip_stream_t* s);
/* FORA top-k (pprhip_fora_topk) for q sources, up to PPRHIP_BATCH of them in flight: every query runs
 * Fora_Topk's loop on delta unchanged (query i with seed + i), the dense levels of its forward_push_topk
 * rounds share sweeps with the other queries in flight.  ids_out/vals_out are q*k, rows padded with
 * id -1 / value 0 (entries beyond k that tie with the k-th value are dropped). */
int pprhip_fora_batch_topk(pprhip_graph_t* g, const int32_t* srcs, int q, int k, double eps, double alpha,
                           uint64_t seed, int32_t* ids_out, double* vals_out, pprhip_stats_t* stats_sum);

/* ---------------------------------------------------------------- backward search (a8, a9) */
/* Backward_Search.backward_search_whole_graph(Long t) (Backward_Search.java:38-100). */
int pprhip_backward_push(pprhip_graph_t* g, int32_t target, double alpha, double rmax, double* reserve_out,
                         double* residue_out, pprhip_stats_t* stats);
/* Base_Whole_Graph.preprocessing(threshold, k) (Base_Whole_Graph.java:58-164) for the targets
 * [t_begin, t_end): backward search per target, entries >= threshold inverted into per-source
 * lists; k >= 0 keeps entries >= the k-th largest and sorts them descending, k < 0 keeps all in
 * target order.  The result covers all n sources for this shard of targets. */
int pprhip_all_pair_backward(pprhip_graph_t* g, double alpha, double threshold, int k, uint32_t t_begin,
                             uint32_t t_end, pprhip_index_t** index_out, pprhip_stats_t* stats);
/* Merge the shards of several target ranges (one per GPU) into one index, re-applying the k rule. */
int pprhip_index_merge(const pprhip_index_t* const* shards, int n_shards, int k, pprhip_index_t** merged_out);
/* Rebuild a shard from its arrays (what a rank receives from another rank before merging). */
int pprhip_index_from_arrays(uint32_t n, const uint64_t* offsets /* n+1 */, const int32_t* targets,
                             const double* values, pprhip_index_t** index_out);
int pprhip_index_info(const pprhip_index_t* ix, uint32_t* n, uint64_t* entries);
int pprhip_index_arrays(const pprhip_index_t* ix, const uint64_t** offsets /* n+1 */, const int32_t** targets,
                        const double** values);
/* Per-source text files "<t>\t<Double.toString(pi)>\n" (Base_Whole_Graph.java:118-126,152-156). */
int pprhip_index_write_dir(const pprhip_index_t* ix, const char* dir);
/* java.lang.Double.toString(d) as the reference's writers print it (shortest round-trip digits,
 * "1.0E-4" style outside [1e-3, 1e7)); returns the length written (without the NUL) or < 0. */
int pprhip_format_double(double d, char* buf, size_t cap);
void pprhip_index_destroy(pprhip_index_t* ix);

/* ---------------------------------------------------------------- multi-GPU
 * Queries (Gen_Util.java:208-232) and targets (Base_Whole_Graph.java:76-92) are independent: every GPU holds a
 * replica of the CSR (one pprhip_graph_t per GPU) and runs its share with the single-GPU code; nothing is exchanged
 * inside a query or a search.  Two exchanges close a call, both over RCCL (librccl is loaded on first use):
 * batched FORA gathers the per-query top-k blocks on rank 0; All-Pair, whose result is keyed by source
 * (Base_Whole_Graph.java:84-86), sends every entry (v, t, pi) to the rank that owns source v - partitioned on the
 * device, one message per peer (one xGMI link each), copied to the host once, by the owner that finalises it. */

/* One process, n_gpu GPUs, one host thread per GPU inside the call: query i runs on per_gpu[i mod n_gpu] exactly as
 * pprhip_fora_batch_single_source would run it (same seed, same walks), the top-k blocks are gathered on GPU 0.
 * ids_out / vals_out: q * k (rows padded with id -1 / value 0), n_out[q] or NULL, stats_per_gpu[n_gpu] or NULL.
 * Handles that share a device (a test set-up) exchange through in-process copies instead of RCCL. */
int pprhip_fora_batch(pprhip_graph_t* const* per_gpu, int n_gpu, const int32_t* srcs, int q, int k, double eps,
                      const pprhip_fora_conf_t* conf, uint64_t seed, int n_rounds, int32_t* ids_out, double* vals_out,
                      int* n_out, pprhip_stats_t* stats_per_gpu);
/* Base_Whole_Graph.preprocessing(threshold, k) over all n targets on n_gpu GPUs: GPU r searches the targets of
 * pprhip_shard_target_range(r) and owns the sources of the same range; returns the whole index (identical to
 * pprhip_all_pair_backward over [0, n) on one GPU). */
int pprhip_all_pair_backward_multi(pprhip_graph_t* const* per_gpu, int n_gpu, double alpha, double threshold, int k,
                                   pprhip_index_t** index_out, pprhip_stats_t* stats_per_gpu);

/* One process per GPU (torch.distributed, MPI, one JVM per GPU): rank 0 draws an id, hands its PPRHIP_COMM_ID_BYTES
 * bytes to every rank by any means, every rank creates its communicator (collective, like ncclCommInitRank). */
#define PPRHIP_COMM_ID_BYTES 128
int pprhip_comm_unique_id(void* id_out /* PPRHIP_COMM_ID_BYTES */);
int pprhip_comm_create(pprhip_graph_t* g, const void* id, int rank, int world, pprhip_comm_t** comm_out);
void pprhip_comm_destroy(pprhip_comm_t* c);
int pprhip_comm_info(const pprhip_comm_t* c, int* rank, int* world);
/* contiguous share [begin, end) of rank `rank` when [0, n) is cut into `world` ranges (targets and owned sources) */
int pprhip_shard_target_range(int rank, int world, uint32_t n, uint32_t* begin, uint32_t* end);
/* The target ranges a sharded All-Pair run searches: cuts_out[world + 1], rank r takes the targets
 * [cuts_out[r], cuts_out[r + 1]).  mode 0: equal counts (pprhip_shard_target_range); mode 1: by work - a pilot measures
 * the searches of the 16 targets with the most in-edges and of 48 more across the in-degree ranks on this handle,
 * every other target is estimated from its in-degree, and the cuts fall at equal shares of the running sum (a store
 * whose ids follow its in-degrees gives rank 0 every hub under equal counts: Base_Whole_Graph.java:76-92 iterates the
 * targets in id order); mode 2: what pprhip_all_pair_backward_sharded does - by work when equal counts would leave one
 * rank with more than 1.15 x the mean of the modelled work (skew_out, may be NULL), else equal counts.  The sources a
 * rank owns stay equal counts in every mode. */
int pprhip_shard_target_cuts(pprhip_graph_t* g, int world, double alpha, double threshold, int mode, uint32_t* cuts_out,
                             double* skew_out);
/* Collective: this rank searches its target range, entries are exchanged by owner of the source on the device, and
 * own_out receives the finished rows (k rule applied) of the sources this rank owns (rows of other sources empty).
 * stats: this rank's search; stats->mc_sources = entries it found, stats->select_bytes = bytes it received. */
int pprhip_all_pair_backward_sharded(pprhip_comm_t* c, double alpha, double threshold, int k, pprhip_index_t** own_out,
                                     pprhip_stats_t* stats);
/* Collective: every rank contributes `rows` <= rows_max rows of k (id, value) pairs; rank 0 receives world blocks of
 * rows_max rows each (short blocks padded with id -1 / value 0) in ids_root / vals_root. */
int pprhip_topk_gather(pprhip_comm_t* c, const int32_t* ids, const double* vals, int rows, int rows_max, int k,
                       int32_t* ids_root, double* vals_root);
/* Failure behaviour of the collectives.  A rank that fails inside pprhip_all_pair_backward_sharded before the
 * exchange still takes part in it with an error mark in its size words: every rank returns an error (the failing one
 * its own, the others PPRHIP_ERR_STATE "rank r failed before the exchange"), none blocks.  No wait on the fabric is
 * unbounded: after PPRHIP_COMM_TIMEOUT_S seconds (environment, default 1800) without completion, or when RCCL reports
 * an asynchronous error, the communicator is aborted (ncclCommAbort: peers' pending operations end with an error as
 * well) and every later collective on it fails at once.  A process that must leave a group early (its own failure
 * outside these calls) calls pprhip_comm_abort before pprhip_comm_destroy so that its peers are released.
 * (No reference counterpart: the Java is single-threaded, Gen_Util.java:208-232 / Base_Whole_Graph.java:76-92.) */
int pprhip_comm_abort(pprhip_comm_t* c);
/* The exchange's partition rule on the host (what k_owner_partition does to the device records): counts_out[world] =
 * entries whose source each rank owns, order_out[count] = the entries' indices owner by owner (stable).  With
 * pprhip_index_from_entries it lets a caller (the CPU multi-process tests, a host-side transport) carry the sharded
 * All-Pair over a fabric of its own with the library's own rule and finalisation (Base_Whole_Graph.java:84-86). */
int pprhip_owner_partition(uint32_t n, int world, const int32_t* sources, uint64_t count, uint64_t* counts_out,
                           uint64_t* order_out);
/* The finished index (k rule of Base_Whole_Graph.java:112-163 applied per source) from `count` entries
 * pi(sources[i], targets[i]) = values[i] in any order; ids are validated against [0, n). */
int pprhip_index_from_entries(uint32_t n, const int32_t* sources, const int32_t* targets, const double* values,
                              uint64_t count, int k, pprhip_index_t** index_out);

/* ---------------------------------------------------------------- ground truth (a12) */
/* Power_Method.computeWholeGraphPPR (Power_Method.java:44-101): `iters` synchronous sweeps. */
int pprhip_power_method(pprhip_graph_t* g, int32_t src, double alpha, int iters, double* reserve_out,
                        pprhip_stats_t* stats);

#ifdef __cplusplus
}
#endif
#endif /* PPRHIP_H */
