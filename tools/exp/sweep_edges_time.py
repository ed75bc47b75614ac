"""Times the batched sweep's edge kernel alone on R-MAT `scale` (libpprhip_hooks.so: pprhip_hook_time_sweep_edges):
whole sweep, block 0 and block 1 of two Gauss-Seidel blocks.
    PPRHIP_LIB_PATH=.../libpprhip_hooks.so python tools/exp/sweep_edges_time.py [scale] [reps]"""
import ctypes as C
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("PPRHIP_LIB_PATH", os.path.join(ROOT, "personalized-pagerank-algorithms-on-neo4j_amd", "libpprhip_hooks.so"))
pkg = importlib.import_module("personalized-pagerank-algorithms-on-neo4j_amd")


def main():
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    h = pkg.HostCsr.rmat(scale)
    lib = pkg.lib()
    lib.pprhip_hook_time_sweep_edges.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
    with pkg.Graph(h, device=0) as g:
        g.set_tuning(pkg.tuning_batch())
        out = []
        for block, nb in ((0, 1), (0, 2), (1, 2)):
            us = C.c_double()
            pkg._check(lib.pprhip_hook_time_sweep_edges(g.h, block, nb, reps, C.byref(us)))
            out.append(us.value)
        print("%s whole %.1f us, block 0 of 2 %.1f us, block 1 of 2 %.1f us" % (os.environ.get("TAG", ""), out[0], out[1], out[2]),
              flush=True)


if __name__ == "__main__":
    main()
