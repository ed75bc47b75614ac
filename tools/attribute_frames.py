"""Attributes the unresolved frames of the round-1 abort (gpurun_out/ap_stats.log: SIGSEGV inside hipLaunchKernel
under `rocprofv3 --kernel-trace`, 16 worker threads) to shared objects.

The log has raw addresses only.  Three of them are known libc / libstdc++ return sites (start_thread, clone3,
__restore_rt; std::thread's trampoline), which fixes those two libraries' load bases in the crashed process.
This tool loads the same libraries in the same order as the crashed command (rocprofv3 + python3 + torch +
libpprhip.so, one All-Pair call so that every lazily loaded piece is in), reads /proc/self/maps, and reports which
mapping lies at the same distance from the libc base as each logged frame.  Shared objects that are mapped at
start-up keep their relative placement from run to run (ASLR moves the whole mmap region), so frames that land
inside an executable mapping at the same relative offset are attributed with confidence; the rest are listed as
unresolved.  Run it under the profiler exactly as the crashed command was:
    rocprofv3 --kernel-trace --stats -d gpurun_out/x -- python3 tools/attribute_frames.py
"""
import importlib, os, sys
import numpy as np
import torch  # noqa: F401  (loads the HIP runtime first, as tools/explore_apbs.py does)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("personalized-pagerank-algorithms-on-neo4j_amd")

LOGGED = [0x7f5420c5c2fb, 0x7f54207a8ee8, 0x7f542166d50e, 0x7f5420751520, 0x7f5420c5c2fb, 0x7f5416173266,
          0x7f54161645c0, 0x7f52d3a9dc1a, 0x7f52d3a99f89, 0x7f52d3a9a615, 0x7f52d3a64635, 0x7f52d3923475,
          0x7f52d396f284, 0x7f52d39239ea, 0x7f52d393a9b1, 0x7f5420fbcec0, 0x7f5224419164, 0x7f54167d8253,
          0x7f54207a3ac3, 0x7f54208358c0]
FAULT = 0x7f5226d00000


def maps():
    out = []
    for line in open("/proc/self/maps"):
        f = line.split()
        lo, hi = (int(x, 16) for x in f[0].split("-"))
        out.append((lo, hi, f[1], f[5] if len(f) > 5 else ""))
    return out


def base_of(mp, name):
    return min(lo for lo, hi, perm, path in mp if os.path.basename(path).startswith(name))


def sym_offset(lib, sym):
    import subprocess
    for l in subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True).stdout.splitlines():
        p = l.split()
        if len(p) == 3 and p[2].split("@")[0] == sym:
            return int(p[0], 16)
    return None


def main():
    host = pkg.HostCsr.rmat(12, 16, seed=1)
    os.environ["PPRHIP_APBS_TIER"] = "3"
    with pkg.Graph(host) as g:
        ix, _ = g.all_pair_backward(0.15, 1e-4, -1, 0, 64)
        ix.close()
    mp = maps()
    libc = [p for _, _, _, p in mp if os.path.basename(p).startswith("libc.so")][0]
    libcxx = [p for _, _, _, p in mp if os.path.basename(p).startswith("libstdc++.so")][0]
    # crashed process: clone3's return site 0x7f54208358c0 and start_thread's 0x7f54207a3ac3 lie in libc
    clone3 = sym_offset(libc, "clone3") or sym_offset(libc, "__clone3")
    here_libc = base_of(mp, "libc.so")
    print("libc here: base %#x, clone3 at +%#x" % (here_libc, clone3 or 0))
    # the logged return address is a few bytes into clone3; page-align the difference
    crashed_libc = (0x7f54208358c0 - (clone3 or 0x126850)) & ~0xfff
    print("libc in the crashed process: base %#x (from the clone3 frame)" % crashed_libc)
    print("fault address %#x = libc base %+#x" % (FAULT, FAULT - crashed_libc))
    for a in LOGGED:
        rel = a - crashed_libc
        tgt = here_libc + rel
        hit = [(lo, hi, perm, path) for lo, hi, perm, path in mp if lo <= tgt < hi]
        if hit:
            lo, hi, perm, path = hit[0]
            b = base_of(mp, os.path.basename(path)) if path.startswith("/") else lo
            print("frame %#x  libc%+#x  -> %s %s (+%#x)" % (a, rel, perm, path or "[anon]", tgt - b))
        else:
            print("frame %#x  libc%+#x  -> no mapping at that relative address in this process" % (a, rel))
    with open(os.path.join(ROOT, "gpurun_out", "maps_attrib.txt"), "w") as f:
        for lo, hi, perm, path in mp:
            f.write("%x-%x %s %s\n" % (lo, hi, perm, path))


if __name__ == "__main__":
    main()
