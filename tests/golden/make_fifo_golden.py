#!/usr/bin/env python3
"""Generates tests/golden/got_fifo_golden.json from the Java-faithful side of the CPU oracle.

The reference (Java + Neo4j) cannot run in this image, so these are NOT outputs of the reference
itself; they are outputs of oracle/ppr_oracle.c's `*_fifo` restatement (the reference's own queue
order, Forward_Push.java:79-141 / Backward_Search.java:51-97) and of its power method
(Power_Method.java:44-101) on the reference's own Game-of-Thrones dataset.  Their job is to freeze
the Java-faithful schedule: the frontier-synchronous twin that the GPU is compared with bit for bit
is edited together with the engine, these vectors are not, so neither the twin nor the engine can
drift away from the reference's algorithm unnoticed (tests/test_oracle.py checks the oracle against
this file on the CPU, tests/test_gpu_reference.py checks the HIP engine against it on the GPU).

    python tests/golden/make_fifo_golden.py        # rewrites the fixture

Values are stored as C99 hex floats, so the file round-trips bit for bit.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

ALPHA = 0.15
EPS = 0.5
OUT = os.path.join(HERE, "got_fifo_golden.json")


def hexvec(v):
    return [float(x).hex() for x in np.asarray(v, dtype=np.float64)]


def main():
    import importlib
    from oracle import oracle as orc
    pkg = importlib.import_module("personalized-pagerank-algorithms-on-neo4j_amd")
    got = pkg.HostCsr.from_neo4j_csv(os.path.join(HERE, "got", "GOT_Nodes.csv"), os.path.join(HERE, "got", "GOT_Rels.csv"))
    og = orc.OracleGraph(got.n, got.out_rp, got.out_ci, got.in_rp, got.in_ci)
    od, idg = np.diff(got.out_rp), np.diff(got.in_rp)
    tyrion = got.names.index("Tyrion")
    dead = int(np.argmax(od == 0))                     # first dead-end source (short-circuits: Forward_Push.java:72-76)
    zero_in = int(np.argmax((idg == 0) & (od > 0)))    # a live source nobody points to
    sources = {"tyrion": tyrion, "dead_end": dead, "zero_in_degree": zero_in}
    doc = {"graph": {"n": got.n, "m": got.m, "dataset": "tests/golden/got/*.csv (the reference's dataset/got)"},
           "alpha": ALPHA, "eps": EPS, "generator": "tests/golden/make_fifo_golden.py", "schedule": "FIFO (Java-faithful)",
           "sources": {}}
    conf = og.conf_whole(ALPHA)
    rmax0, omega = orc.fora_whole_params(conf, EPS)
    doc["fora_params"] = {"rmax0": float(rmax0).hex(), "omega": float(omega).hex()}
    for name, s in sources.items():
        e = {"id": int(s), "name": got.names[s], "out_degree": int(od[s]), "in_degree": int(idg[s])}
        e["power_method_100"] = hexvec(og.power_method(s, ALPHA, 100))
        for tag, rmax in (("rmax0", rmax0), ("1e-10", 1e-10)):
            p, r, rsum, st = og.forward_push(s, ALPHA, rmax, orc.FIFO)
            e["forward_push_" + tag] = {"rmax": float(rmax).hex(), "reserve": hexvec(p), "residue": hexvec(r),
                                        "rsum_field": float(rsum).hex(), "pops": int(st.pops),
                                        "edge_pushes": int(st.edge_pushes)}
        p, r, st = og.backward_push(s, ALPHA, 1e-8, orc.FIFO)
        e["backward_push_1e-8"] = {"reserve": hexvec(p), "residue": hexvec(r), "pops": int(st.pops)}
        est, st = og.fora_whole(s, EPS, ALPHA, seed=3, n_rounds=1, schedule=orc.FIFO)
        e["fora_whole_1round_seed3"] = {"estimate": hexvec(est), "walks": int(st.walks), "walk_steps": int(st.walk_steps)}
        cnt, ids, vals = orc.topk(og.power_method(s, ALPHA, 100), 10, cap=got.n)
        e["power_method_top10"] = {"count": int(cnt), "ids": [int(i) for i in ids]}
        doc["sources"][name] = e
    with open(OUT, "w") as f:
        json.dump(doc, f, indent=1)
        f.write("\n")
    print("wrote %s (%d bytes)" % (OUT, os.path.getsize(OUT)))


if __name__ == "__main__":
    main()
