"""ctypes binding of oracle/ppr_baseline.cpp (CPU baselines for bench.py's `cpu_baseline` leg).

BENCH / TEST INFRASTRUCTURE ONLY, like oracle.py: the product package never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import oracle as orc

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libppr_baseline.so")
_lib = None


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("ppr_baseline.cpp", "ppr_oracle.c", "ppr_oracle.h")]
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(map(os.path.getmtime, srcs)):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libppr_baseline.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        dp, up, ip = C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_int)
        L.base_fora_hashmap.restype = None
        L.base_fora_hashmap.argtypes = [C.POINTER(orc.Graph), C.c_int32, C.c_double, C.POINTER(orc.Conf), C.c_uint64,
                                        C.c_uint64, C.c_double, dp, dp, up, up, up, up, ip, ip, C.c_void_p]
        L.base_fora_array_parallel.restype = None
        L.base_fora_array_parallel.argtypes = [C.POINTER(orc.Graph), C.c_void_p, C.c_int, C.c_double,
                                               C.POINTER(orc.Conf), C.c_uint64, C.c_uint64, C.c_int, dp, C.c_void_p, up]
        L.base_fora_array_rounds.restype = C.c_int
        L.base_fora_array_rounds.argtypes = [C.POINTER(orc.Graph), C.c_int32, C.c_double, C.POINTER(orc.Conf), C.c_uint64,
                                             C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, dp, up, up]
        L.base_hardware_threads.restype = C.c_int
        _lib = L
    return _lib


def hardware_threads():
    return lib().base_hardware_threads()


def fora_hashmap(og, src, eps, alpha, seed, walk_divisor=1, push_budget_s=0.0, want_reserve=False):
    """The reference's FORA in the Java's data-structure shape (hash maps, FIFO deque, hash set), one thread."""
    conf = og.conf_whole(alpha)
    ps, ws = C.c_double(), C.c_double()
    ep, pops, wr, wt = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
    rounds, trunc = C.c_int(), C.c_int()
    out = np.zeros(og.n) if want_reserve else None
    lib().base_fora_hashmap(C.byref(og.c), src, eps, C.byref(conf), seed, walk_divisor, push_budget_s, C.byref(ps),
                            C.byref(ws), C.byref(ep), C.byref(pops), C.byref(wr), C.byref(wt), C.byref(rounds),
                            C.byref(trunc), out.ctypes.data_as(C.c_void_p) if out is not None else None)
    return {"push_s": ps.value, "walk_s": ws.value, "edge_pushes": ep.value, "pops": pops.value, "walks_run": wr.value,
            "walks_total": wt.value, "rounds": rounds.value, "truncated": bool(trunc.value), "reserve": out}


def fora_array_rounds(og, src, eps, alpha, seed, walk_divisor=1, max_rec=32):
    """The dense-array port, one thread, with the work of every turn of the clock-driven loop."""
    conf = og.conf_whole(alpha)
    rs = np.zeros(max_rec)
    re = np.zeros(max_rec, dtype=np.uint64)
    rr = np.zeros(max_rec)
    ws = C.c_double()
    wr, wt = C.c_uint64(), C.c_uint64()
    n = lib().base_fora_array_rounds(C.byref(og.c), src, eps, C.byref(conf), seed, walk_divisor, max_rec,
                                     rs.ctypes.data_as(C.c_void_p), re.ctypes.data_as(C.c_void_p),
                                     rr.ctypes.data_as(C.c_void_p), C.byref(ws), C.byref(wr), C.byref(wt))
    k = min(n, max_rec)
    return {"rounds": n, "round_push_s": rs[:k].tolist(), "round_edge_pushes": [int(x) for x in re[:k]],
            "round_rsum": rr[:k].tolist(), "walk_s": ws.value, "walks_run": wr.value, "walks_total": wt.value}


def fora_array_parallel(og, srcs, eps, alpha, seed, walk_divisor=1, threads=0):
    """The dense-array port, one query per host thread."""
    conf = og.conf_whole(alpha)
    srcs = np.ascontiguousarray(srcs, dtype=np.int32)
    per = np.zeros(srcs.size)
    wall, ep = C.c_double(), C.c_uint64()
    threads = threads or hardware_threads()
    lib().base_fora_array_parallel(C.byref(og.c), srcs.ctypes.data_as(C.c_void_p), srcs.size, eps, C.byref(conf), seed,
                                   walk_divisor, threads, C.byref(wall), per.ctypes.data_as(C.c_void_p), C.byref(ep))
    return {"wall_s": wall.value, "per_query_s": per.tolist(), "edge_pushes": ep.value, "threads": threads}
