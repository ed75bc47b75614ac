"""pprhip — MI355X-native Personalized PageRank (FORA + All-Pair-Backward-Search).

Thin ctypes binding of the C ABI in include/pprhip.h.  The compute path lives entirely in
libpprhip.so (hand-written HIP kernels for gfx950); there is no Python or CPU fallback: importing
this package without the built library raises, and every compute call fails loudly when no GPU is
usable.  Build with `make -C personalized-pagerank-algorithms-on-neo4j_amd` or
`__graft_entry__.build()`.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PPRHIP_LIB_PATH: another build of the SAME sources (the Makefile's sanitizer build, `make asan-test`); never a fallback
LIB_PATH = os.environ.get("PPRHIP_LIB_PATH") or os.path.join(_HERE, "libpprhip.so")

OK = 0
ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_OOM, ERR_IO, ERR_STATE = -1, -2, -3, -4, -5, -6

KERNEL_NAMES = {0: "none", 1: "dense_pull", 2: "sparse_push", 3: "walk", 4: "backward_batch", 5: "dense_pull_batch",
                6: "query_setup"}
BATCH = 16  # PPRHIP_BATCH: queries in flight in fora_batch_single_source


class PprhipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("pprhip error %d: %s" % (code, msg))
        self.code = code


class Stats(C.Structure):
    _fields_ = [("pops", C.c_uint64), ("edge_pushes", C.c_uint64), ("enqueues", C.c_uint64),
                ("dead_end_pops", C.c_uint64), ("dense_nodes", C.c_uint64), ("levels", C.c_uint32),
                ("dense_levels", C.c_uint32), ("rounds", C.c_uint32), ("xl_targets", C.c_uint32),
                ("mc_sources", C.c_uint64), ("walks", C.c_uint64), ("walk_steps", C.c_uint64),
                ("select_passes", C.c_uint64), ("rsum", C.c_double), ("rmax_final", C.c_double),
                ("omega", C.c_double), ("kth_value", C.c_double), ("push_ms", C.c_double), ("mc_ms", C.c_double),
                ("select_ms", C.c_double), ("total_ms", C.c_double), ("push_bytes", C.c_uint64),
                ("mc_bytes", C.c_uint64), ("select_bytes", C.c_uint64), ("dominant_kernel_ms", C.c_double),
                ("dominant_kernel_bytes", C.c_uint64), ("dominant_kernel_launches", C.c_uint32),
                ("dominant_kernel_id", C.c_uint32), ("class_ms", C.c_double * 8), ("class_bytes", C.c_uint64 * 8),
                ("class_launches", C.c_uint32 * 8), ("dense_edges", C.c_uint64), ("sweep_min_bytes", C.c_uint64),
                ("walk_loads", C.c_uint64), ("walk_load_lanes", C.c_uint64)]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_ if not k.startswith("reserved") and not k.startswith("class_")}
        for k in ("class_ms", "class_bytes", "class_launches"):
            d[k] = list(getattr(self, k))
        return d


class Tuning(C.Structure):
    _fields_ = [("c_walk_ns", C.c_double), ("c_edge_ns", C.c_double), ("c_pop_ns", C.c_double),
                ("c_level_ns", C.c_double), ("c_dense_edge_ns", C.c_double), ("c_dense_node_ns", C.c_double),
                ("dense_frac", C.c_double), ("max_rounds", C.c_int32), ("max_halvings", C.c_int32),
                ("halving_ratio", C.c_double), ("prior_levels", C.c_int32), ("gs_blocks", C.c_int32),
                ("gs_frac", C.c_double)]


class ForaConf(C.Structure):
    _fields_ = [("alpha", C.c_double), ("delta", C.c_double), ("pfail", C.c_double), ("rsum", C.c_double),
                ("min_delta", C.c_double), ("k", C.c_int32), ("n", C.c_uint32), ("m", C.c_uint64)]


# every symbol include/pprhip.h declares (tests check the library exports all of them)
EXPORTS = [
    "pprhip_last_error", "pprhip_version", "pprhip_device_count", "pprhip_tuning_default",
    "pprhip_conf_fora_whole_graph", "pprhip_conf_fora_topk", "pprhip_fora_whole_params", "pprhip_fora_topk_params",
    "pprhip_rmat_edges", "pprhip_edgelist_from_neo4j_csv", "pprhip_edgelist_info", "pprhip_edgelist_edges",
    "pprhip_edgelist_node_name", "pprhip_edgelist_destroy", "pprhip_csr_build", "pprhip_graph_create",
    "pprhip_graph_destroy", "pprhip_graph_release", "pprhip_graph_info", "pprhip_graph_set_tuning", "pprhip_graph_get_tuning",
    "pprhip_get_reserve", "pprhip_get_residue", "pprhip_forward_push", "pprhip_fwdpush_topk_reset",
    "pprhip_fwdpush_topk_round", "pprhip_random_walk_batch", "pprhip_fora_single_source", "pprhip_fora_topk",
    "pprhip_topk_select", "pprhip_monte_carlo", "pprhip_fora_batch_topk", "pprhip_backward_push",
    "pprhip_all_pair_backward", "pprhip_index_merge", "pprhip_index_info", "pprhip_index_arrays",
    "pprhip_index_write_dir", "pprhip_index_destroy", "pprhip_power_method", "pprhip_index_from_arrays",
    "pprhip_format_double", "pprhip_edgelist_from_neo4j_store", "pprhip_edgelist_build_csr",
    "pprhip_fora_batch_single_source", "pprhip_tuning_batch", "pprhip_tuning_batch_for", "pprhip_results_create", "pprhip_results_destroy",
    "pprhip_results_info", "pprhip_results_fetch", "pprhip_results_sum", "pprhip_fora_batch_single_source_resident",
    "pprhip_fora_batch", "pprhip_all_pair_backward_multi", "pprhip_comm_unique_id", "pprhip_comm_create",
    "pprhip_comm_destroy", "pprhip_comm_info", "pprhip_shard_target_range", "pprhip_all_pair_backward_sharded",
    "pprhip_topk_gather", "pprhip_comm_abort", "pprhip_owner_partition", "pprhip_index_from_entries",
    "pprhip_device_memory", "pprhip_graph_lift_host", "pprhip_lift_array", "pprhip_lift_destroy",
    "pprhip_fora_stream_open", "pprhip_fora_stream_submit", "pprhip_fora_stream_wait", "pprhip_fora_stream_close",
    "pprhip_set_kernel_timing", "pprhip_shard_target_cuts",
]
COMM_ID_BYTES = 128

_lib = None
_LIVE = {}


def lib():
    """Loads libpprhip.so; raises if it has not been built (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libpprhip.so is not built (%s); run `make -C %s`" % (LIB_PATH, _HERE))
    L = C.CDLL(LIB_PATH)
    vp, i32, u32, u64, dbl, ci = C.c_void_p, C.c_int32, C.c_uint32, C.c_uint64, C.c_double, C.c_int
    P = C.POINTER
    L.pprhip_last_error.restype = C.c_char_p
    L.pprhip_version.restype = ci
    L.pprhip_device_count.argtypes = [P(ci)]
    L.pprhip_tuning_default.argtypes = [P(Tuning)]
    L.pprhip_tuning_default.restype = None
    L.pprhip_tuning_batch.argtypes = [P(Tuning)]
    L.pprhip_tuning_batch.restype = None
    L.pprhip_tuning_batch_for.argtypes = [ci, P(Tuning)]
    L.pprhip_tuning_batch_for.restype = None
    L.pprhip_conf_fora_whole_graph.argtypes = [u32, u64, dbl, P(ForaConf)]
    L.pprhip_conf_fora_topk.argtypes = [u32, u64, ci, dbl, P(ForaConf)]
    L.pprhip_fora_whole_params.argtypes = [P(ForaConf), dbl, P(dbl), P(dbl)]
    L.pprhip_fora_topk_params.argtypes = [P(ForaConf), dbl, dbl, P(dbl), P(dbl), P(dbl)]
    L.pprhip_rmat_edges.argtypes = [ci, ci, u64, vp, vp]
    L.pprhip_edgelist_from_neo4j_csv.argtypes = [C.c_char_p, C.c_char_p, P(vp)]
    L.pprhip_edgelist_from_neo4j_store.argtypes = [C.c_char_p, P(vp)]
    L.pprhip_edgelist_build_csr.argtypes = [vp, ci, vp, vp]
    L.pprhip_edgelist_info.argtypes = [vp, P(u32), P(u64)]
    L.pprhip_edgelist_edges.argtypes = [vp, P(vp), P(vp)]
    L.pprhip_edgelist_node_name.argtypes = [vp, u32]
    L.pprhip_edgelist_node_name.restype = C.c_char_p
    L.pprhip_edgelist_destroy.argtypes = [vp]
    L.pprhip_edgelist_destroy.restype = None
    L.pprhip_csr_build.argtypes = [u32, u64, vp, vp, ci, vp, vp]
    L.pprhip_graph_create.argtypes = [u32, u64, vp, vp, vp, vp, ci, P(vp)]
    L.pprhip_graph_destroy.argtypes = [vp]
    L.pprhip_graph_destroy.restype = None
    L.pprhip_graph_release.argtypes = [vp, C.c_uint]
    L.pprhip_graph_info.argtypes = [vp, P(u32), P(u64), P(ci)]
    L.pprhip_device_memory.argtypes = [vp, P(u64), P(u64)]
    L.pprhip_graph_set_tuning.argtypes = [vp, P(Tuning)]
    L.pprhip_graph_get_tuning.argtypes = [vp, P(Tuning)]
    L.pprhip_get_reserve.argtypes = [vp, vp]
    L.pprhip_get_residue.argtypes = [vp, vp]
    L.pprhip_forward_push.argtypes = [vp, i32, dbl, dbl, vp, vp, P(dbl), P(Stats)]
    L.pprhip_fwdpush_topk_reset.argtypes = [vp, i32, dbl]
    L.pprhip_fwdpush_topk_round.argtypes = [vp, dbl, dbl, P(dbl), P(Stats)]
    L.pprhip_random_walk_batch.argtypes = [vp, vp, vp, u64, dbl, u64, u32, ci, vp, vp]
    L.pprhip_fora_single_source.argtypes = [vp, i32, dbl, P(ForaConf), u64, ci, vp, P(Stats)]
    L.pprhip_fora_topk.argtypes = [vp, i32, dbl, P(ForaConf), u64, vp, vp, ci, P(ci), vp, P(Stats)]
    L.pprhip_topk_select.argtypes = [vp, ci, vp, vp, ci, P(ci), P(dbl), P(Stats)]
    L.pprhip_monte_carlo.argtypes = [vp, i32, dbl, P(ForaConf), u64, vp, P(Stats)]
    L.pprhip_fora_batch_single_source.argtypes = [vp, vp, ci, dbl, P(ForaConf), u64, ci, vp, ci, vp, vp, vp, vp, P(Stats)]
    L.pprhip_fora_batch_single_source_resident.argtypes = [vp, vp, ci, dbl, P(ForaConf), u64, ci, vp, vp, ci, vp, vp, vp,
                                                           vp, P(Stats)]
    L.pprhip_results_create.argtypes = [vp, ci, P(vp)]
    L.pprhip_results_destroy.argtypes = [vp]
    L.pprhip_results_destroy.restype = None
    L.pprhip_results_info.argtypes = [vp, P(ci), P(ci), P(u32)]
    L.pprhip_results_fetch.argtypes = [vp, ci, vp]
    L.pprhip_results_sum.argtypes = [vp, ci, P(dbl)]
    L.pprhip_fora_batch_topk.argtypes = [vp, vp, ci, ci, dbl, dbl, u64, vp, vp, P(Stats)]
    L.pprhip_fora_batch.argtypes = [P(vp), ci, vp, ci, ci, dbl, P(ForaConf), u64, ci, vp, vp, vp, vp]
    L.pprhip_all_pair_backward_multi.argtypes = [P(vp), ci, dbl, dbl, ci, P(vp), vp]
    L.pprhip_comm_unique_id.argtypes = [vp]
    L.pprhip_comm_create.argtypes = [vp, vp, ci, ci, P(vp)]
    L.pprhip_comm_destroy.argtypes = [vp]
    L.pprhip_comm_destroy.restype = None
    L.pprhip_comm_info.argtypes = [vp, P(ci), P(ci)]
    L.pprhip_shard_target_range.argtypes = [ci, ci, u32, P(u32), P(u32)]
    L.pprhip_shard_target_cuts.argtypes = [vp, ci, dbl, dbl, ci, vp, P(dbl)]
    L.pprhip_all_pair_backward_sharded.argtypes = [vp, dbl, dbl, ci, P(vp), P(Stats)]
    L.pprhip_topk_gather.argtypes = [vp, vp, vp, ci, ci, ci, vp, vp]
    L.pprhip_comm_abort.argtypes = [vp]
    L.pprhip_owner_partition.argtypes = [u32, ci, vp, u64, vp, vp]
    L.pprhip_index_from_entries.argtypes = [u32, vp, vp, vp, u64, ci, P(vp)]
    L.pprhip_backward_push.argtypes = [vp, i32, dbl, dbl, vp, vp, P(Stats)]
    L.pprhip_all_pair_backward.argtypes = [vp, dbl, dbl, ci, u32, u32, P(vp), P(Stats)]
    L.pprhip_index_merge.argtypes = [P(vp), ci, ci, P(vp)]
    L.pprhip_index_from_arrays.argtypes = [u32, vp, vp, vp, P(vp)]
    L.pprhip_format_double.argtypes = [dbl, C.c_char_p, C.c_size_t]
    L.pprhip_index_info.argtypes = [vp, P(u32), P(u64)]
    L.pprhip_index_arrays.argtypes = [vp, P(vp), P(vp), P(vp)]
    L.pprhip_index_write_dir.argtypes = [vp, C.c_char_p]
    L.pprhip_index_destroy.argtypes = [vp]
    L.pprhip_index_destroy.restype = None
    L.pprhip_power_method.argtypes = [vp, i32, dbl, ci, vp, P(Stats)]
    L.pprhip_graph_lift_host.argtypes = [u32, u64, vp, vp, vp, vp, ci, P(vp)]
    L.pprhip_lift_array.argtypes = [vp, ci, P(vp), P(u64)]
    L.pprhip_lift_destroy.argtypes = [vp]
    L.pprhip_lift_destroy.restype = None
    L.pprhip_fora_stream_open.argtypes = [vp, dbl, P(ForaConf), ci, P(vp)]
    L.pprhip_fora_stream_submit.argtypes = [vp, vp, ci, u64, vp, ci, vp, vp, vp, P(u64)]
    L.pprhip_fora_stream_wait.argtypes = [vp, u64, P(Stats)]
    L.pprhip_fora_stream_close.argtypes = [vp]
    L.pprhip_set_kernel_timing.argtypes = [ci]
    _lib = L
    # the destroy entry points, reachable from destructors that run while the interpreter shuts down (the name `lib`
    # may already be None then: "TypeError: 'NoneType' object is not callable" out of Index.__del__, round 3)
    _LIVE.update(graph_destroy=L.pprhip_graph_destroy, results_destroy=L.pprhip_results_destroy,
                 comm_destroy=L.pprhip_comm_destroy)
    return L


def _check(rc):
    if rc != OK:
        raise PprhipError(rc, lib().pprhip_last_error().decode("utf-8", "replace"))


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def set_kernel_timing(on):
    """Kernel-class times in Stats.class_ms (HIP events around every group of launches): off by default - they cost the
    latency-bound paths 2-8 %; on == 2 ("sweeps"): only the dense sweeps are timed, every other class counted.  Returns
    the previous state (False, True or 2), which can be passed back."""
    was = lib().pprhip_set_kernel_timing(2 if (on == 2 and on is not True) else (1 if on else 0))
    return 2 if was == 2 else bool(was)


def device_count():
    c = C.c_int(0)
    _check(lib().pprhip_device_count(C.byref(c)))
    return c.value


# ------------------------------------------------------------------ ingest helpers (host side)
def rmat_edges(scale, edge_factor=16, seed=1):
    m = edge_factor << scale
    src = np.empty(m, dtype=np.int32)
    dst = np.empty(m, dtype=np.int32)
    _check(lib().pprhip_rmat_edges(scale, edge_factor, seed, _ptr(src), _ptr(dst)))
    return src, dst


def csr_build(n, key, val, newest_first=False):
    key = np.ascontiguousarray(key, dtype=np.int32)
    val = np.ascontiguousarray(val, dtype=np.int32)
    rp = np.empty(n + 1, dtype=np.uint32)
    ci = np.empty(max(key.size, 1), dtype=np.int32)
    _check(lib().pprhip_csr_build(n, key.size, _ptr(key), _ptr(val), int(newest_first), _ptr(rp), _ptr(ci)))
    return rp, ci[:key.size]


def load_neo4j_csv(nodes_csv, rels_csv):
    """Returns (n, src, dst, names) with node id = row index of the nodes file."""
    h = C.c_void_p()
    _check(lib().pprhip_edgelist_from_neo4j_csv(nodes_csv.encode(), rels_csv.encode(), C.byref(h)))
    try:
        n, m = C.c_uint32(), C.c_uint64()
        _check(lib().pprhip_edgelist_info(h, C.byref(n), C.byref(m)))
        ps, pd = C.c_void_p(), C.c_void_p()
        _check(lib().pprhip_edgelist_edges(h, C.byref(ps), C.byref(pd)))
        if m.value:
            src = np.ctypeslib.as_array(C.cast(ps, C.POINTER(C.c_int32)), shape=(m.value,)).copy()
            dst = np.ctypeslib.as_array(C.cast(pd, C.POINTER(C.c_int32)), shape=(m.value,)).copy()
        else:
            src = np.zeros(0, dtype=np.int32)
            dst = np.zeros(0, dtype=np.int32)
        names = [lib().pprhip_edgelist_node_name(h, i).decode() for i in range(n.value)]
    finally:
        lib().pprhip_edgelist_destroy(h)
    return n.value, src, dst, names


class HostCsr:
    """Host out-/in-CSR pair built by the product's ingest code (input data for engine and oracle)."""

    def __init__(self, n, src, dst, newest_first=False):
        self.n = int(n)
        self.m = int(len(src))
        self.out_rp, self.out_ci = csr_build(n, src, dst, newest_first)
        self.in_rp, self.in_ci = csr_build(n, dst, src, newest_first)

    @classmethod
    def rmat(cls, scale, edge_factor=16, seed=1):
        src, dst = rmat_edges(scale, edge_factor, seed)
        return cls(1 << scale, src, dst)

    @classmethod
    def from_neo4j_store(cls, store_dir):
        """A Neo4j 3.x store directory (e.g. target/got.db) read without a JVM; adjacency in chain order."""
        h = C.c_void_p()
        _check(lib().pprhip_edgelist_from_neo4j_store(store_dir.encode(), C.byref(h)))
        try:
            n, m = C.c_uint32(), C.c_uint64()
            _check(lib().pprhip_edgelist_info(h, C.byref(n), C.byref(m)))
            self = cls.__new__(cls)
            self.n, self.m = n.value, m.value
            self.out_rp = np.empty(self.n + 1, dtype=np.uint32)
            self.in_rp = np.empty(self.n + 1, dtype=np.uint32)
            self.out_ci = np.empty(max(self.m, 1), dtype=np.int32)
            self.in_ci = np.empty(max(self.m, 1), dtype=np.int32)
            _check(lib().pprhip_edgelist_build_csr(h, 0, _ptr(self.out_rp), _ptr(self.out_ci)))
            _check(lib().pprhip_edgelist_build_csr(h, 1, _ptr(self.in_rp), _ptr(self.in_ci)))
            self.out_ci, self.in_ci = self.out_ci[:self.m], self.in_ci[:self.m]
            self.names = [lib().pprhip_edgelist_node_name(h, i).decode() for i in range(self.n)]
        finally:
            lib().pprhip_edgelist_destroy(h)
        return self

    @classmethod
    def from_neo4j_csv(cls, nodes_csv, rels_csv):
        n, src, dst, names = load_neo4j_csv(nodes_csv, rels_csv)
        h = cls(n, src, dst, newest_first=True)  # HeavyGraph lists newest relationships first
        h.names = names
        return h


LIFT_ARRAYS = {  # pprhip_lift_array: name -> (id, dtype)
    "new2old": (0, np.int32), "old2new": (1, np.int32), "out_rp": (2, np.uint32), "out_ci": (3, np.int32),
    "in_rp": (4, np.uint32), "in_ci": (5, np.int32), "nz_rows": (6, np.int32), "zin_rows": (7, np.int32),
    "flags": (8, np.uint8), "chunk_starts": (9, np.uint32), "cross": (10, np.uint64), "edge_base": (11, np.uint64),
    "seg_base": (12, np.uint64), "sl_ci": (13, np.int32), "sl_flags": (14, np.uint8), "sl_chunk_starts": (15, np.uint32),
    "seg_row": (16, np.uint32), "seg_off": (17, np.uint32),
    # the row-panel copy the single-query sweep walks (empty when the graph has none)
    "panel_sizes": (18, np.uint64), "panel_src": (19, np.int32), "panel_row": (20, np.uint16),
    "panel_items": (21, np.uint32), "panel_desc": (22, np.uint32), "panel_item0": (23, np.uint32),
}


def lift_host(host, threads=0, with_in=True):
    """The host half of the graph lift (pprhip_graph_lift_host; needs no device): dict of the internal layout's arrays."""
    h = C.c_void_p()
    _check(lib().pprhip_graph_lift_host(C.c_uint32(host.n), C.c_uint64(host.m), _ptr(host.out_rp), _ptr(host.out_ci),
                                        _ptr(host.in_rp) if with_in else None, _ptr(host.in_ci) if with_in else None,
                                        int(threads), C.byref(h)))
    try:
        out = {}
        for name, (which, dt) in LIFT_ARRAYS.items():
            p, nb = C.c_void_p(), C.c_uint64()
            _check(lib().pprhip_lift_array(h, which, C.byref(p), C.byref(nb)))
            if nb.value:
                buf = (C.c_uint8 * nb.value).from_address(p.value)
                out[name] = np.frombuffer(buf, dtype=dt).copy()
            else:
                out[name] = np.empty(0, dtype=dt)
        return out
    finally:
        lib().pprhip_lift_destroy(h)


# ------------------------------------------------------------------ device graph
def conf_whole_graph(n, m, alpha):
    c = ForaConf()
    _check(lib().pprhip_conf_fora_whole_graph(n, m, alpha, C.byref(c)))
    return c


def conf_topk(n, m, k, alpha):
    c = ForaConf()
    _check(lib().pprhip_conf_fora_topk(n, m, k, alpha, C.byref(c)))
    return c


def fora_whole_params(conf, eps):
    a, b = C.c_double(), C.c_double()
    _check(lib().pprhip_fora_whole_params(C.byref(conf), eps, C.byref(a), C.byref(b)))
    return a.value, b.value


def fora_topk_params(conf, eps, delta):
    a, b, c = C.c_double(), C.c_double(), C.c_double()
    _check(lib().pprhip_fora_topk_params(C.byref(conf), eps, delta, C.byref(a), C.byref(b), C.byref(c)))
    return a.value, b.value, c.value


def tuning_default():
    t = Tuning()
    lib().pprhip_tuning_default(C.byref(t))
    return t


def tuning_batch():
    t = Tuning()
    lib().pprhip_tuning_batch(C.byref(t))
    return t


def tuning_batch_for(q):
    """The batch profile for a call of q queries (pprhip_tuning_batch_for)."""
    t = Tuning()
    lib().pprhip_tuning_batch_for(int(q), C.byref(t))
    return t


class Index:
    """All-pair inverted index (CSR by source)."""

    def __init__(self, handle):
        self.h = handle
        self._destroy = lib().pprhip_index_destroy  # kept: at interpreter shutdown the module's globals may be gone

    def arrays(self):
        n, e = C.c_uint32(), C.c_uint64()
        _check(lib().pprhip_index_info(self.h, C.byref(n), C.byref(e)))
        po, pt, pv = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _check(lib().pprhip_index_arrays(self.h, C.byref(po), C.byref(pt), C.byref(pv)))
        off = np.ctypeslib.as_array(C.cast(po, C.POINTER(C.c_uint64)), shape=(n.value + 1,)).copy()
        if e.value:
            tg = np.ctypeslib.as_array(C.cast(pt, C.POINTER(C.c_int32)), shape=(e.value,)).copy()
            vl = np.ctypeslib.as_array(C.cast(pv, C.POINTER(C.c_double)), shape=(e.value,)).copy()
        else:
            tg, vl = np.zeros(0, dtype=np.int32), np.zeros(0)
        return off, tg, vl

    def write_dir(self, path):
        _check(lib().pprhip_index_write_dir(self.h, path.encode()))

    def close(self):
        if getattr(self, "h", None):
            self._destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001  (a destructor at interpreter shutdown must not raise)
            pass


def index_from_arrays(n, offsets, targets, values):
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    targets = np.ascontiguousarray(targets, dtype=np.int32)
    values = np.ascontiguousarray(values, dtype=np.float64)
    out = C.c_void_p()
    _check(lib().pprhip_index_from_arrays(n, _ptr(offsets), _ptr(targets), _ptr(values), C.byref(out)))
    return Index(out)


def format_double(d):
    """java.lang.Double.toString(d), the number format of the reference's result files."""
    buf = C.create_string_buffer(64)
    n = lib().pprhip_format_double(d, buf, 64)
    if n < 0:
        raise PprhipError(n, "pprhip_format_double failed")
    return buf.value.decode()


def merge_indexes(shards, k):
    arr = (C.c_void_p * len(shards))(*[s.h for s in shards])
    out = C.c_void_p()
    _check(lib().pprhip_index_merge(arr, len(shards), k, C.byref(out)))
    return Index(out)


def owner_partition(n, world, sources):
    """The sharded All-Pair's partition rule on the host: (counts[world], order[count]) - entries owner by owner."""
    sources = np.ascontiguousarray(sources, dtype=np.int32)
    counts = np.zeros(world, dtype=np.uint64)
    order = np.zeros(max(sources.size, 1), dtype=np.uint64)
    _check(lib().pprhip_owner_partition(n, world, _ptr(sources), sources.size, _ptr(counts), _ptr(order)))
    return counts, order[:sources.size]


def index_from_entries(n, sources, targets, values, k):
    """The finished index from (source, target, value) entries in any order (k rule applied per source)."""
    sources = np.ascontiguousarray(sources, dtype=np.int32)
    targets = np.ascontiguousarray(targets, dtype=np.int32)
    values = np.ascontiguousarray(values, dtype=np.float64)
    out = C.c_void_p()
    _check(lib().pprhip_index_from_entries(n, _ptr(sources), _ptr(targets), _ptr(values), sources.size, k, C.byref(out)))
    return Index(out)


def shard_target_range(rank, world, n):
    b, e = C.c_uint32(), C.c_uint32()
    _check(lib().pprhip_shard_target_range(rank, world, n, C.byref(b), C.byref(e)))
    return b.value, e.value


CUT_EQUAL, CUT_BY_WORK, CUT_AUTO = 0, 1, 2


def shard_target_cuts(graph, world, alpha, threshold, mode=CUT_AUTO):
    """pprhip_shard_target_cuts: (cuts[world + 1], skew) - the target ranges a sharded All-Pair run searches and the
    largest share of the modelled work equal counts would give one rank (x the mean)."""
    cuts = np.zeros(world + 1, dtype=np.uint32)
    skew = C.c_double()
    _check(lib().pprhip_shard_target_cuts(graph.h, world, alpha, threshold, mode, _ptr(cuts), C.byref(skew)))
    return cuts, skew.value


def comm_unique_id():
    """The bytes rank 0 hands to every rank before Comm() (an RCCL unique id)."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    _check(lib().pprhip_comm_unique_id(buf))
    return buf.raw


class Comm:
    """One rank of a multi-GPU group: an RCCL communicator bound to this rank's graph replica (collective create)."""

    def __init__(self, graph, uid, rank, world):
        self.graph = graph
        self.rank, self.world = rank, world
        self.h = C.c_void_p()
        buf = C.create_string_buffer(bytes(uid), COMM_ID_BYTES)
        _check(lib().pprhip_comm_create(graph.h, buf, rank, world, C.byref(self.h)))

    def all_pair_backward_sharded(self, alpha, threshold, k):
        """Collective: returns (Index of the sources this rank owns, Stats)."""
        out, st = C.c_void_p(), Stats()
        _check(lib().pprhip_all_pair_backward_sharded(self.h, alpha, threshold, k, C.byref(out), C.byref(st)))
        return Index(out), st

    def topk_gather(self, ids, vals, rows_max):
        """Collective: rank 0 gets (ids[world, rows_max, k], vals[...]), the others None."""
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        vals = np.ascontiguousarray(vals, dtype=np.float64)
        rows, k = ids.shape
        ri = np.empty((self.world, rows_max, k), dtype=np.int32) if self.rank == 0 else None
        rv = np.empty((self.world, rows_max, k)) if self.rank == 0 else None
        _check(lib().pprhip_topk_gather(self.h, _ptr(ids), _ptr(vals), rows, rows_max, k, _ptr(ri), _ptr(rv)))
        return (ri, rv) if self.rank == 0 else None

    def abort(self):
        """Leaves the group at once (peers' pending operations end with an error instead of waiting)."""
        if getattr(self, "h", None):
            _check(lib().pprhip_comm_abort(self.h))

    def close(self):
        if getattr(self, "h", None):
            _LIVE["comm_destroy"](self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


def fora_batch_multi(graphs, srcs, k, eps, alpha, seed, n_rounds=0, conf=None):
    """pprhip_fora_batch: q queries over len(graphs) GPUs of this process; returns (ids[q, k], vals[q, k], n_sel[q], [Stats])."""
    srcs = np.ascontiguousarray(srcs, dtype=np.int32)
    q = int(srcs.size)
    g0 = graphs[0]
    conf = conf or conf_whole_graph(g0.n, g0.m, alpha)
    hs = (C.c_void_p * len(graphs))(*[g.h for g in graphs])
    ids = np.empty((q, k), dtype=np.int32)
    vals = np.empty((q, k))
    nsel = np.zeros(q, dtype=np.int32)
    sts = (Stats * len(graphs))()
    _check(lib().pprhip_fora_batch(hs, len(graphs), _ptr(srcs), q, k, eps, C.byref(conf), seed, n_rounds, _ptr(ids),
                                   _ptr(vals), _ptr(nsel), C.cast(sts, C.c_void_p)))
    return ids, vals, nsel, list(sts)


def all_pair_backward_multi(graphs, alpha, threshold, k):
    """pprhip_all_pair_backward_multi: the whole index over len(graphs) GPUs of this process; returns (Index, [Stats])."""
    hs = (C.c_void_p * len(graphs))(*[g.h for g in graphs])
    out = C.c_void_p()
    sts = (Stats * len(graphs))()
    _check(lib().pprhip_all_pair_backward_multi(hs, len(graphs), alpha, threshold, k, C.byref(out),
                                                C.cast(sts, C.c_void_p)))
    return Index(out), list(sts)


class Results:
    """Device-resident result vectors of a batched call (slot i = query i)."""

    def __init__(self, graph, capacity):
        self.n = graph.n
        self.graph = graph  # the store lives on the graph's device: keep the handle alive
        self.h = C.c_void_p()
        _check(lib().pprhip_results_create(graph.h, capacity, C.byref(self.h)))

    def info(self):
        cap, cnt, n = C.c_int(), C.c_int(), C.c_uint32()
        _check(lib().pprhip_results_info(self.h, C.byref(cap), C.byref(cnt), C.byref(n)))
        return cap.value, cnt.value, n.value

    def fetch(self, i):
        out = np.empty(self.n)
        _check(lib().pprhip_results_fetch(self.h, i, _ptr(out)))
        return out

    def sum(self, i):
        s = C.c_double()
        _check(lib().pprhip_results_sum(self.h, i, C.byref(s)))
        return s.value

    def close(self):
        if getattr(self, "h", None):
            _LIVE["results_destroy"](self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


class QueryStream:
    """pprhip_fora_stream_*: single-source FORA queries submitted in blocks and run by the batched driver without a
    drain between the blocks.  submit() returns a ticket; wait(ticket) returns (ids[q, k], vals[q, k], n_sel[q], Stats)."""

    def __init__(self, graph, eps, alpha, k=0, conf=None):
        self.graph = graph
        self.k = int(k)
        self.conf = conf or conf_whole_graph(graph.n, graph.m, alpha)
        self.h = C.c_void_p()
        self._out = {}
        _check(lib().pprhip_fora_stream_open(graph.h, eps, C.byref(self.conf), self.k, C.byref(self.h)))

    def submit(self, srcs, seed, keep=None, keep_first=0):
        srcs = np.ascontiguousarray(srcs, dtype=np.int32)
        q = int(srcs.size)
        ids = np.empty((q, self.k), dtype=np.int32) if self.k > 0 else None
        vals = np.empty((q, self.k)) if self.k > 0 else None
        nsel = np.zeros(q, dtype=np.int32) if self.k > 0 else None
        t = C.c_uint64()
        _check(lib().pprhip_fora_stream_submit(self.h, _ptr(srcs), q, seed, keep.h if keep is not None else None,
                                               int(keep_first), _ptr(ids), _ptr(vals), _ptr(nsel), C.byref(t)))
        self._out[t.value] = (ids, vals, nsel)  # (the library writes into these until the wait returns)
        return t.value

    def wait(self, ticket):
        st = Stats()
        _check(lib().pprhip_fora_stream_wait(self.h, ticket, C.byref(st)))
        ids, vals, nsel = self._out.pop(ticket)
        return ids, vals, nsel, st

    def close(self):
        if getattr(self, "h", None):
            h, self.h = self.h, None
            # The close runs every submission that nobody waited for, and the driver writes their top-k blocks
            # into the arrays of _out: those must outlive the call (pprhip_jni.cpp: close_stream has the same order).
            try:
                _check(lib().pprhip_fora_stream_close(h))
            finally:
                self._out.clear()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


class Graph:
    """Device-resident CSR pair + per-query workspace (one per GPU, one thread at a time)."""

    def __init__(self, host, device=0):
        self.n, self.m = host.n, host.m
        self.h = C.c_void_p()
        _check(lib().pprhip_graph_create(host.n, host.m, _ptr(host.out_rp), _ptr(host.out_ci), _ptr(host.in_rp),
                                         _ptr(host.in_ci), device, C.byref(self.h)))

    def close(self):
        if getattr(self, "h", None):
            _LIVE["graph_destroy"](self.h)
            self.h = None

    RELEASE_ALL_PAIR, RELEASE_BATCH = 1, 2

    def release(self, what):
        """Hands the workspaces of the named entry points back (pprhip_graph_release); they come back on next use."""
        _check(lib().pprhip_graph_release(self.h, what))

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def device_memory(self):
        """(free, total) bytes of HBM on the handle's device."""
        f, t = C.c_uint64(), C.c_uint64()
        _check(lib().pprhip_device_memory(self.h, C.byref(f), C.byref(t)))
        return f.value, t.value

    def set_tuning(self, t):
        _check(lib().pprhip_graph_set_tuning(self.h, C.byref(t)))

    def get_tuning(self):
        t = Tuning()
        _check(lib().pprhip_graph_get_tuning(self.h, C.byref(t)))
        return t

    def reserve(self):
        out = np.empty(self.n)
        _check(lib().pprhip_get_reserve(self.h, _ptr(out)))
        return out

    def residue(self):
        out = np.empty(self.n)
        _check(lib().pprhip_get_residue(self.h, _ptr(out)))
        return out

    def forward_push(self, src, alpha, rmax, fetch=True):
        reserve = np.empty(self.n) if fetch else None
        residue = np.empty(self.n) if fetch else None
        rsum, st = C.c_double(), Stats()
        _check(lib().pprhip_forward_push(self.h, src, alpha, rmax, _ptr(reserve), _ptr(residue), C.byref(rsum),
                                         C.byref(st)))
        return reserve, residue, rsum.value, st

    def topk_push_reset(self, src, alpha):
        _check(lib().pprhip_fwdpush_topk_reset(self.h, src, alpha))

    def topk_push_round(self, min_rmax, rmax):
        rsum, st = C.c_double(), Stats()
        _check(lib().pprhip_fwdpush_topk_round(self.h, min_rmax, rmax, C.byref(rsum), C.byref(st)))
        return rsum.value, st

    def random_walks(self, starts, idx, alpha, seed, stream=0, no_zero_hop=False):
        starts = np.ascontiguousarray(starts, dtype=np.int32)
        idx = np.ascontiguousarray(idx, dtype=np.uint64)
        term = np.empty(starts.size, dtype=np.int32)
        steps = np.empty(starts.size, dtype=np.uint32)
        _check(lib().pprhip_random_walk_batch(self.h, _ptr(starts), _ptr(idx), starts.size, alpha, seed, stream,
                                              int(no_zero_hop), _ptr(term), _ptr(steps)))
        return term, steps

    def fora_single_source(self, src, eps, alpha, seed, n_rounds=0, conf=None, fetch=True):
        conf = conf or conf_whole_graph(self.n, self.m, alpha)
        out = np.empty(self.n) if fetch else None
        st = Stats()
        _check(lib().pprhip_fora_single_source(self.h, src, eps, C.byref(conf), seed, n_rounds, _ptr(out),
                                               C.byref(st)))
        return out, st

    def fora_topk(self, src, eps, alpha, k, seed, cap=None, conf=None, fetch=False):
        conf = conf or conf_topk(self.n, self.m, k, alpha)
        cap = cap if cap is not None else k
        ids = np.empty(max(cap, 1), dtype=np.int32)
        vals = np.empty(max(cap, 1))
        nsel, st = C.c_int(0), Stats()
        out = np.empty(self.n) if fetch else None
        _check(lib().pprhip_fora_topk(self.h, src, eps, C.byref(conf), seed, _ptr(ids), _ptr(vals), cap,
                                      C.byref(nsel), _ptr(out), C.byref(st)))
        w = min(nsel.value, cap)
        return nsel.value, ids[:w].copy(), vals[:w].copy(), out, st

    def topk_select(self, k, cap=None):
        cap = cap if cap is not None else k
        ids = np.empty(max(cap, 1), dtype=np.int32)
        vals = np.empty(max(cap, 1))
        nsel, kth, st = C.c_int(0), C.c_double(0.0), Stats()
        _check(lib().pprhip_topk_select(self.h, k, _ptr(ids), _ptr(vals), cap, C.byref(nsel), C.byref(kth),
                                        C.byref(st)))
        w = min(nsel.value, cap)
        return nsel.value, ids[:w].copy(), vals[:w].copy(), kth.value, st

    def monte_carlo(self, src, eps, alpha, seed, conf=None):
        conf = conf or conf_whole_graph(self.n, self.m, alpha)
        out = np.empty(self.n)
        st = Stats()
        _check(lib().pprhip_monte_carlo(self.h, src, eps, C.byref(conf), seed, _ptr(out), C.byref(st)))
        return out, st

    def fora_batch_single_source(self, srcs, eps, alpha, seed, n_rounds=0, k=0, conf=None, fetch=False, per_query=False,
                                 keep=None, out=None):
        """q single-source FORA queries, BATCH of them in flight; returns (reserve[q, n] | None, ids[q, k] | None,
        vals[q, k] | None, n_sel[q] | None, per-query Stats list | None, summed Stats).  keep: a Results store that
        receives every query's vector (device-resident); out: caller's [q, n] float64 array for fetch=True."""
        srcs = np.ascontiguousarray(srcs, dtype=np.int32)
        q = int(srcs.size)
        conf = conf or conf_whole_graph(self.n, self.m, alpha)
        if fetch and out is not None:
            assert out.dtype == np.float64 and out.flags["C_CONTIGUOUS"] and out.shape == (q, self.n)
        elif fetch:
            out = np.empty((q, self.n))
        else:
            out = None
        ids = np.empty((q, k), dtype=np.int32) if k > 0 else None
        vals = np.empty((q, k)) if k > 0 else None
        nsel = np.zeros(q, dtype=np.int32) if k > 0 else None
        pq = (Stats * q)() if per_query and q else None
        st = Stats()
        _check(lib().pprhip_fora_batch_single_source_resident(
            self.h, _ptr(srcs), q, eps, C.byref(conf), seed, n_rounds, keep.h if keep is not None else None, _ptr(out),
            k, _ptr(ids), _ptr(vals), _ptr(nsel), C.cast(pq, C.c_void_p) if pq is not None else None, C.byref(st)))
        return out, ids, vals, nsel, (list(pq) if pq is not None else None), st

    def fora_batch_topk(self, srcs, k, eps, alpha, seed):
        srcs = np.ascontiguousarray(srcs, dtype=np.int32)
        ids = np.empty((srcs.size, k), dtype=np.int32)
        vals = np.empty((srcs.size, k))
        st = Stats()
        _check(lib().pprhip_fora_batch_topk(self.h, _ptr(srcs), srcs.size, k, eps, alpha, seed, _ptr(ids), _ptr(vals),
                                            C.byref(st)))
        return ids, vals, st

    def backward_push(self, target, alpha, rmax):
        reserve = np.empty(self.n)
        residue = np.empty(self.n)
        st = Stats()
        _check(lib().pprhip_backward_push(self.h, target, alpha, rmax, _ptr(reserve), _ptr(residue), C.byref(st)))
        return reserve, residue, st

    def all_pair_backward(self, alpha, threshold, k, t_begin=0, t_end=None):
        t_end = self.n if t_end is None else t_end
        out, st = C.c_void_p(), Stats()
        _check(lib().pprhip_all_pair_backward(self.h, alpha, threshold, k, t_begin, t_end, C.byref(out), C.byref(st)))
        return Index(out), st

    def power_method(self, src, alpha, iters=100):
        out = np.empty(self.n)
        st = Stats()
        _check(lib().pprhip_power_method(self.h, src, alpha, iters, _ptr(out), C.byref(st)))
        return out, st
