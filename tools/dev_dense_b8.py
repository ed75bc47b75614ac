"""Developer microbenchmark: batched (8 queries per sweep) dense edge kernel vs the single-query one."""
import ctypes
import sys
import os
import torch  # noqa: F401  (loads the HIP runtime first)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
pkg = importlib.import_module("personalized-pagerank-algorithms-on-neo4j_amd")

for scale in [int(x) for x in sys.argv[1:]] or [16, 22]:
    host = pkg.HostCsr.rmat(scale, 16, seed=1)
    g = pkg.Graph(host)
    ms = ctypes.c_double()
    md = ctypes.c_double()
    f = pkg.lib().pprhip_dev_dense_b8
    f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    for width in (8, 16, 32):
      rc = f(g.h, 20, width, ctypes.byref(ms), ctypes.byref(md))
      print("scale", scale, "width", width, "rc", rc, "ms per sweep %.4f" % ms.value, "maxdiff", md.value, flush=True)
    g.close()
