"""ctypes binding of the CPU oracle (oracle/ppr_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package never does.  See oracle/ppr_oracle.h for the "parity unpinned" statement.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libppr_oracle.so")

FIFO = 0
SYNC = 1


def build(force=False):
    src = os.path.join(_HERE, "ppr_oracle.c")
    hdr = os.path.join(_HERE, "ppr_oracle.h")
    if (force or not os.path.exists(_LIB_PATH)
            or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr))):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libppr_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class Graph(C.Structure):
    _fields_ = [("n", C.c_uint32), ("m", C.c_uint64), ("out_rp", C.c_void_p), ("out_ci", C.c_void_p),
                ("in_rp", C.c_void_p), ("in_ci", C.c_void_p)]


class Stats(C.Structure):
    _fields_ = [("pops", C.c_uint64), ("edge_pushes", C.c_uint64), ("enqueues", C.c_uint64),
                ("dead_end_pops", C.c_uint64), ("dense_nodes", C.c_uint64), ("levels", C.c_uint32),
                ("dense_levels", C.c_uint32), ("rounds", C.c_uint32), ("pad", C.c_uint32),
                ("mc_sources", C.c_uint64), ("walks", C.c_uint64), ("walk_steps", C.c_uint64),
                ("rsum", C.c_double), ("rmax_final", C.c_double), ("omega", C.c_double),
                ("kth_value", C.c_double), ("model_cost_ns", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "pad"}


class Tuning(C.Structure):
    _fields_ = [("c_walk_ns", C.c_double), ("c_edge_ns", C.c_double), ("c_pop_ns", C.c_double),
                ("c_level_ns", C.c_double), ("c_dense_edge_ns", C.c_double), ("c_dense_node_ns", C.c_double),
                ("dense_frac", C.c_double), ("max_rounds", C.c_int32), ("max_halvings", C.c_int32),
                ("halving_ratio", C.c_double), ("prior_levels", C.c_int32), ("gs_blocks", C.c_int32),
                ("gs_frac", C.c_double)]


class Conf(C.Structure):
    _fields_ = [("alpha", C.c_double), ("delta", C.c_double), ("pfail", C.c_double), ("rsum", C.c_double),
                ("min_delta", C.c_double), ("k", C.c_int32), ("n", C.c_uint32), ("m", C.c_uint64)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        dp = C.POINTER(C.c_double)
        L.orc_forward_push.restype = C.c_double
        L.orc_forward_push.argtypes = [C.POINTER(Graph), C.c_int, C.c_int32, C.c_double, C.c_double, C.c_void_p,
                                       C.c_void_p, C.POINTER(Stats)]
        L.orc_power_method.argtypes = [C.POINTER(Graph), C.c_int32, C.c_double, C.c_int, C.c_void_p]
        L.orc_power_method.restype = None
        L.orc_philox4x32_10.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_philox4x32_10.restype = None
        L.orc_random_walk.restype = C.c_int32
        L.orc_random_walk.argtypes = [C.POINTER(Graph), C.c_int32, C.c_double, C.c_uint64, C.c_uint32, C.c_uint64,
                                      C.c_int, C.POINTER(C.c_uint32)]
        L.orc_fora_whole.restype = None
        L.orc_fora_whole.argtypes = [C.POINTER(Graph), C.c_int, C.c_int32, C.c_double, C.POINTER(Conf), C.c_uint64,
                                     C.c_int, C.POINTER(Tuning), C.c_void_p, C.POINTER(Stats)]
        L.orc_fora_whole_baseline.restype = None
        L.orc_fora_whole_baseline.argtypes = [C.POINTER(Graph), C.c_int32, C.c_double, C.POINTER(Conf), C.c_uint64,
                                              C.c_uint64, C.c_int, dp, dp, C.c_void_p, C.POINTER(Stats)]
        L.orc_fora_topk.restype = None
        L.orc_fora_topk.argtypes = [C.POINTER(Graph), C.c_int, C.c_int32, C.c_double, C.POINTER(Conf), C.c_uint64,
                                    C.c_void_p, C.POINTER(Stats)]
        L.orc_monte_carlo.restype = None
        L.orc_monte_carlo.argtypes = [C.POINTER(Graph), C.c_int32, C.c_double, C.POINTER(Conf), C.c_uint64,
                                      C.c_void_p, C.POINTER(Stats)]
        L.orc_kth_largest.restype = C.c_int
        L.orc_kth_largest.argtypes = [C.c_void_p, C.c_uint32, C.c_int, dp]
        L.orc_topk.restype = C.c_int
        L.orc_topk.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_backward_push.restype = None
        L.orc_backward_push.argtypes = [C.POINTER(Graph), C.c_int, C.c_int32, C.c_double, C.c_double, C.c_void_p,
                                        C.c_void_p, C.POINTER(Stats)]
        L.orc_all_pair_backward.restype = None
        L.orc_all_pair_backward.argtypes = [C.POINTER(Graph), C.c_int, C.c_double, C.c_double, C.c_int, C.c_uint32,
                                            C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                            C.POINTER(C.c_void_p)]
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_free.restype = None
        L.orc_topk_push_new.restype = C.c_void_p
        L.orc_topk_push_new.argtypes = [C.POINTER(Graph), C.c_int, C.c_int32, C.c_double]
        L.orc_topk_push_round.restype = C.c_double
        L.orc_topk_push_round.argtypes = [C.c_void_p, C.c_double, C.c_double, C.POINTER(Stats)]
        L.orc_topk_push_reserve.restype = C.c_void_p
        L.orc_topk_push_reserve.argtypes = [C.c_void_p]
        L.orc_topk_push_residue.restype = C.c_void_p
        L.orc_topk_push_residue.argtypes = [C.c_void_p]
        L.orc_topk_push_free.argtypes = [C.c_void_p]
        L.orc_topk_push_free.restype = None
        L.orc_conf_fora_whole_graph.argtypes = [C.c_uint32, C.c_uint64, C.c_double, C.POINTER(Conf)]
        L.orc_conf_fora_whole_graph.restype = None
        L.orc_conf_fora_topk.argtypes = [C.c_uint32, C.c_uint64, C.c_int, C.c_double, C.POINTER(Conf)]
        L.orc_conf_fora_topk.restype = None
        L.orc_fora_whole_params.argtypes = [C.POINTER(Conf), C.c_double, dp, dp]
        L.orc_fora_whole_params.restype = None
        L.orc_fora_topk_params.argtypes = [C.POINTER(Conf), C.c_double, C.c_double, dp, dp, dp]
        L.orc_fora_topk_params.restype = None
        L.orc_tuning_default.argtypes = [C.POINTER(Tuning)]
        L.orc_tuning_default.restype = None
        L.orc_set_sync_tuning.argtypes = [C.POINTER(Tuning)]
        L.orc_set_sync_tuning.restype = None
        L.orc_max_err.restype = C.c_double
        L.orc_max_err.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.orc_precision.restype = C.c_double
        L.orc_precision.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.orc_ndcg.restype = C.c_double
        L.orc_ndcg.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleGraph:
    """Host CSR pair handed to the oracle (arrays are kept alive by this object)."""

    def __init__(self, n, out_rp, out_ci, in_rp, in_ci):
        self.n = int(n)
        self.out_rp = np.ascontiguousarray(out_rp, dtype=np.uint32)
        self.out_ci = np.ascontiguousarray(out_ci, dtype=np.int32)
        self.in_rp = np.ascontiguousarray(in_rp, dtype=np.uint32)
        self.in_ci = np.ascontiguousarray(in_ci, dtype=np.int32)
        self.m = int(self.out_ci.size)
        assert self.out_rp.size == self.n + 1 and self.in_rp.size == self.n + 1 and self.in_ci.size == self.m
        self.c = Graph(self.n, self.m, _ptr(self.out_rp), _ptr(self.out_ci), _ptr(self.in_rp), _ptr(self.in_ci))

    # -- parameters
    def conf_whole(self, alpha):
        c = Conf()
        lib().orc_conf_fora_whole_graph(self.n, self.m, alpha, C.byref(c))
        return c

    def conf_topk(self, k, alpha):
        c = Conf()
        lib().orc_conf_fora_topk(self.n, self.m, k, alpha, C.byref(c))
        return c

    # -- algorithms
    def power_method(self, src, alpha, iters=100):
        out = np.zeros(self.n)
        lib().orc_power_method(C.byref(self.c), src, alpha, iters, _ptr(out))
        return out

    def forward_push(self, src, alpha, rmax, schedule=SYNC):
        reserve = np.zeros(self.n)
        residue = np.zeros(self.n)
        st = Stats()
        rsum = lib().orc_forward_push(C.byref(self.c), schedule, src, alpha, rmax, _ptr(reserve), _ptr(residue),
                                      C.byref(st))
        return reserve, residue, rsum, st

    def random_walk(self, start, alpha, seed, stream, idx, no_zero_hop):
        steps = C.c_uint32(0)
        t = lib().orc_random_walk(C.byref(self.c), start, alpha, seed, stream, idx, int(no_zero_hop), C.byref(steps))
        return t, steps.value

    def fora_whole(self, src, eps, alpha, seed, n_rounds=1, schedule=SYNC, tuning=None, conf=None):
        conf = conf or self.conf_whole(alpha)
        out = np.zeros(self.n)
        st = Stats()
        lib().orc_fora_whole(C.byref(self.c), schedule, src, eps, C.byref(conf), seed, n_rounds,
                             C.byref(tuning) if tuning is not None else None, _ptr(out), C.byref(st))
        return out, st

    def fora_whole_baseline(self, src, eps, alpha, seed, walk_divisor=1, max_rounds=0):
        """Clock-driven FIFO FORA as the reference wrote it; returns (estimate, push_s, walk_s, stats)."""
        conf = self.conf_whole(alpha)
        out = np.zeros(self.n)
        st = Stats()
        ps, ws = C.c_double(), C.c_double()
        lib().orc_fora_whole_baseline(C.byref(self.c), src, eps, C.byref(conf), seed, walk_divisor, max_rounds,
                                      C.byref(ps), C.byref(ws), _ptr(out), C.byref(st))
        return out, ps.value, ws.value, st

    def fora_topk(self, src, eps, alpha, k, seed, schedule=SYNC, conf=None):
        conf = conf or self.conf_topk(k, alpha)
        out = np.zeros(self.n)
        st = Stats()
        lib().orc_fora_topk(C.byref(self.c), schedule, src, eps, C.byref(conf), seed, _ptr(out), C.byref(st))
        return out, st

    def monte_carlo(self, src, eps, alpha, seed, conf=None):
        conf = conf or self.conf_whole(alpha)
        out = np.zeros(self.n)
        st = Stats()
        lib().orc_monte_carlo(C.byref(self.c), src, eps, C.byref(conf), seed, _ptr(out), C.byref(st))
        return out, st

    def backward_push(self, target, alpha, rmax, schedule=SYNC):
        reserve = np.zeros(self.n)
        residue = np.zeros(self.n)
        st = Stats()
        lib().orc_backward_push(C.byref(self.c), schedule, target, alpha, rmax, _ptr(reserve), _ptr(residue),
                                C.byref(st))
        return reserve, residue, st

    def all_pair_backward(self, alpha, threshold, k, t_begin=0, t_end=None, schedule=SYNC):
        t_end = self.n if t_end is None else t_end
        po, pt, pv = C.c_void_p(), C.c_void_p(), C.c_void_p()
        lib().orc_all_pair_backward(C.byref(self.c), schedule, alpha, threshold, k, t_begin, t_end, C.byref(po),
                                    C.byref(pt), C.byref(pv))
        off = np.ctypeslib.as_array(C.cast(po, C.POINTER(C.c_uint64)), shape=(self.n + 1,)).copy()
        cnt = int(off[-1])
        tg = np.ctypeslib.as_array(C.cast(pt, C.POINTER(C.c_int32)), shape=(max(cnt, 1),))[:cnt].copy()
        vl = np.ctypeslib.as_array(C.cast(pv, C.POINTER(C.c_double)), shape=(max(cnt, 1),))[:cnt].copy()
        for p in (po, pt, pv):
            lib().orc_free(p)
        return off, tg, vl

    def topk_push(self, src, alpha, schedule=SYNC):
        return TopkPush(self, src, alpha, schedule)


class TopkPush:
    def __init__(self, g, src, alpha, schedule):
        self.g = g
        self.h = lib().orc_topk_push_new(C.byref(g.c), schedule, src, alpha)

    def round(self, min_rmax, rmax):
        st = Stats()
        rsum = lib().orc_topk_push_round(self.h, min_rmax, rmax, C.byref(st))
        return rsum, st

    def _arr(self, p):
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_double)), shape=(self.g.n,)).copy()

    @property
    def reserve(self):
        return self._arr(lib().orc_topk_push_reserve(self.h))

    @property
    def residue(self):
        return self._arr(lib().orc_topk_push_residue(self.h))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_topk_push_free(self.h)
            self.h = None


def philox(ctr, key):
    c = np.asarray(ctr, dtype=np.uint32)
    k = np.asarray(key, dtype=np.uint32)
    out = np.zeros(4, dtype=np.uint32)
    lib().orc_philox4x32_10(_ptr(c), _ptr(k), _ptr(out))
    return out


def kth_largest(v, k):
    v = np.ascontiguousarray(v, dtype=np.float64)
    out = C.c_double(0.0)
    ok = lib().orc_kth_largest(_ptr(v), v.size, k, C.byref(out))
    return out.value if ok else None


def topk(v, k, cap=None):
    v = np.ascontiguousarray(v, dtype=np.float64)
    cap = cap or v.size
    ids = np.zeros(cap, dtype=np.int32)
    vals = np.zeros(cap)
    cnt = lib().orc_topk(_ptr(v), v.size, k, _ptr(ids), _ptr(vals), cap)
    w = min(cnt, cap)
    return cnt, ids[:w], vals[:w]


def fora_whole_params(conf, eps):
    a, b = C.c_double(), C.c_double()
    lib().orc_fora_whole_params(C.byref(conf), eps, C.byref(a), C.byref(b))
    return a.value, b.value


def fora_topk_params(conf, eps, delta):
    a, b, c = C.c_double(), C.c_double(), C.c_double()
    lib().orc_fora_topk_params(C.byref(conf), eps, delta, C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


def tuning_default():
    t = Tuning()
    lib().orc_tuning_default(C.byref(t))
    return t


def set_sync_tuning(t=None):
    """Tuning of the twin's entry points that take none (forward_push, topk_push, fora_topk); None: defaults.
    Accepts the product's Tuning struct as well (same fields)."""
    if t is None:
        lib().orc_set_sync_tuning(None)
        return
    o = Tuning()
    for f, _ in o._fields_:
        setattr(o, f, getattr(t, f))
    lib().orc_set_sync_tuning(C.byref(o))


def max_err(est, exact):
    est = np.ascontiguousarray(est, dtype=np.float64)
    exact = np.ascontiguousarray(exact, dtype=np.float64)
    return lib().orc_max_err(_ptr(est), _ptr(exact), est.size)


def precision(algo_ids, gnd_ids):
    a = np.ascontiguousarray(algo_ids, dtype=np.int32)
    g = np.ascontiguousarray(gnd_ids, dtype=np.int32)
    return lib().orc_precision(_ptr(a), a.size, _ptr(g), g.size)


def ndcg(algo_ids, gnd_ids, exact):
    a = np.ascontiguousarray(algo_ids, dtype=np.int32)
    g = np.ascontiguousarray(gnd_ids, dtype=np.int32)
    e = np.ascontiguousarray(exact, dtype=np.float64)
    return lib().orc_ndcg(_ptr(a), a.size, _ptr(g), g.size, _ptr(e))
