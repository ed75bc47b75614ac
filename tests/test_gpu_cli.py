"""GPU test of the native `ppr` command line (host mirror of PPR.main / Gen_Util over the C ABI):
same flags as the reference, same report file layout."""
import os
import re
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
PPR = os.path.join(ROOT, "personalized-pagerank-algorithms-on-neo4j_amd", "ppr")
GOT_DIR = os.path.join(ROOT, "tests", "golden", "got")


def test_help_lists_reference_flags():
    out = subprocess.run([PPR, "-help"], capture_output=True, text=True, timeout=60).stdout
    for flag in ("-alpha", "-eps", "-query", "-k", "-node", "-label", "-rel", "-db", "-help"):
        assert flag in out  # PPR.java:157-166


def test_batch_report_on_got(tmp_path):
    r = subprocess.run([PPR, "-alpha", "0.15", "-eps", "0.5", "-query", "6", "-k", "10", "-db", GOT_DIR],
                       capture_output=True, text=True, timeout=600, cwd=tmp_path)
    assert r.returncode == 0 and "failed" not in r.stdout, r.stdout[-2000:]
    rep = (tmp_path / "got_AlgoPerfResults.txt").read_text()  # <db>_AlgoPerfResults.txt (Gen_Util.java:329)
    assert re.match(r"\d{4}-\d\d-\d\d \d\d:\d\d:\d\d\n", rep)
    for h in ("Test 1. Whole-Graph test", "1.1 FORA_WHOLE_GRAPH", "1.2 FWDPUSH", "1.3 MC", "1.4 BASE_WHOLE_GRAPH",
              "Test 2. Top-k test", "2.1 FORA_TOPK", "2.4 BASE_WHOLE_GRAPH", "Test 3. Preprocessing test",
              "3.1 FORA_WHOLE_GRAPH", "3.2 FWDPUSH", "3.3 MC", "3.4 BASE_WHOLE_GRAPH"):
        assert h in rep
    sec = rep.split("1.1 FORA_WHOLE_GRAPH\n")[1].split("\n\n")[0].strip().splitlines()
    assert len(sec) == 5                                   # five epsilons
    eps, ms, err = sec[2].split(",")                       # "param,avg ms,avg max err" (Gen_Util.java:179,244,247)
    assert eps == "0.5" and float(err) < 0.1
    errs = [float(l.split(",")[2]) for l in sec]
    assert errs[-1] < errs[0]                              # smaller epsilon, smaller error
    push = rep.split("1.2 FWDPUSH\n")[1].split("\n\n")[0].strip().splitlines()
    assert float(push[-1].split(",")[2]) < 1e-6            # rmax = 1e-8: push alone is exact to 1e-6 on GOT
    top = rep.split("2.1 FORA_TOPK\n")[1].split("\n\n")[0].strip().splitlines()
    p, k, ms, prec, ndcg = top[2].split(",")               # "param,k,avg ms,precision,NDCG" (:142,171)
    assert k == "10" and float(prec) >= 0.8 and float(ndcg) >= 0.95
    base = rep.split("2.4 BASE_WHOLE_GRAPH\n")[1].split("\n\n")[0].strip().splitlines()
    thr, k, prep_ms, size, ms, prec, ndcg = base[-1].split(",")  # "thr,k,prep ms,bytes,avg ms,precision,NDCG" (:139)
    assert thr == "5.0E-7" and int(size) > 0 and float(prec) >= 0.8
    # Test 3 (Gen_Util.java:602-645): "param,threshold,prep ms,prep bytes,avg max err" (:203,247), five rows per
    # algorithm.  The preprocessed answers are the same algorithms' answers read back from Double.toString files (which
    # round-trip), on freshly drawn query nodes: errors of the same size as Test 1's, falling with the parameter.
    for i, name in enumerate(("FORA_WHOLE_GRAPH", "FWDPUSH", "MC"), 1):
        rows = rep.split("3.%d %s\n" % (i, name))[1].split("\n\n")[0].strip().splitlines()
        assert len(rows) == 5
        errs3 = []
        for l in rows:
            param, thr, prep_ms, size, err = l.split(",")
            assert thr == "-1.0" and int(size) > 0 and int(prep_ms) >= 0
            errs3.append(float(err))
        assert errs3[-1] < errs3[0]
        # the same algorithm and parameters as Test 1's rows on other query nodes; both obey the algorithm's own bound
        # on the largest absolute error: eps * max(pi, delta) <= eps for FORA and MC (their parameter is eps,
        # Gen_Util.java:451-478), m * rmax for the push alone (what the residues can still hold; m = 352)
        rows1 = rep.split("1.%d %s\n" % (i, name))[1].split("\n\n")[0].strip().splitlines()
        assert [l.split(",")[0] for l in rows1] == [l.split(",")[0] for l in rows]
        for l3, l1 in zip(rows, rows1):
            par = float(l3.split(",")[0])
            bound = 352.0 * par if name == "FWDPUSH" else par
            assert float(l3.split(",")[4]) <= bound and float(l1.split(",")[2]) <= bound, (name, l3, l1)
        if name == "FWDPUSH":
            assert errs3[-1] < 1e-6  # rmax = 1e-8: exact to 1e-6 from the files as well
    b3 = rep.split("3.4 BASE_WHOLE_GRAPH\n")[1].strip().splitlines()
    assert len(b3) == 5 and b3[0].split(",")[0] == "-1" and b3[0].split(",")[1] == "0.001"
    # no result directory is left behind (deletePrepDir, :250-252)
    for d in ("FORA_ppr_results", "FWP_ppr_results", "MC_ppr_results", "BASE_ppr_results"):
        p = tmp_path / d / "got"
        assert not p.exists() or not any(p.iterdir())


def test_store_directory_as_db(tmp_path, got):
    """-db pointing at a Neo4j store directory (the reference's default is target/got.db)."""
    store = os.path.join(ROOT, "tests", "golden", "got.db")
    tyrion = str(got.names.index("Tyrion"))
    a = subprocess.run([PPR, "-db", store, "-single", tyrion, "-k", "5"], capture_output=True, text=True, timeout=120,
                       cwd=tmp_path).stdout
    assert "node_amount = 107, rel_amount = 352" in a
    b = subprocess.run([PPR, "-db", GOT_DIR, "-single", tyrion, "-k", "5"], capture_output=True, text=True,
                       timeout=120, cwd=tmp_path).stdout
    vals = lambda out: [l.split("\t")[1] for l in out.split("Fora-Top5 PPR:\n")[1].strip().splitlines()]
    assert vals(a) == vals(b)  # same graph, same seed: same numbers (the store has ids, the CSVs names)


def test_single_source_print(tmp_path, got):
    tyrion = str(got.names.index("Tyrion"))
    r = subprocess.run([PPR, "-db", GOT_DIR, "-single", tyrion, "-k", "5"], capture_output=True, text=True,
                       timeout=120, cwd=tmp_path)
    assert "Fora-Whole-Graph PPR:" in r.stdout and "Fora-Top5 PPR:" in r.stdout
    rows = r.stdout.split("Fora-Top5 PPR:\n")[1].strip().splitlines()
    assert len(rows) == 5 and all(l.startswith("@") for l in rows) and rows[0].startswith("@Tyrion")


def _splitmix(state):
    state = (state + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = state
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return state, z ^ (z >> 31)


def _query_nodes(seed, draw, query_num, n):
    """Gen_Util.getQueryNodes of the host mirror (Gen_Util.java:99-107 with a seeded generator)."""
    s = seed ^ ((0x51ED270B1A5 + draw) & 0xFFFFFFFFFFFFFFFF)
    out = []
    for _ in range(query_num):
        s, z = _splitmix(s)
        out.append((z * n) >> 64)
    return out


def test_report_metrics_recomputed_through_the_abi(tmp_path, pkg, orc, got):
    """The report's MAX_ERR / PRECISION / NDCG figures (Gen_Util.computeError, Gen_Util.java:259-326) recomputed
    outside the host mirror: same query nodes, same seeds, vectors fetched through the C ABI, metrics by the
    oracle's orc_max_err / orc_precision / orc_ndcg, ground truth by 100 power-method sweeps."""
    import numpy as np
    Q, K, SEED = 6, 10, 1
    r = subprocess.run([PPR, "-alpha", "0.15", "-query", str(Q), "-k", str(K), "-db", GOT_DIR, "-seed", str(SEED)],
                       capture_output=True, text=True, timeout=600, cwd=tmp_path)
    assert r.returncode == 0 and "failed" not in r.stdout, r.stdout[-2000:]
    rep = (tmp_path / "got_AlgoPerfResults.txt").read_text()
    eps_arr = [10.0, 5.0, 0.5, 0.1, 0.05]
    with pkg.Graph(got) as g:
        exact = {}

        def truth(s):
            if s not in exact:
                exact[s] = g.power_method(s, 0.15, 100)[0]
            return exact[s]

        # 1.1 FORA_WHOLE_GRAPH: calls 0..4 of algo_perf_test; the query loop runs as one batch with seed + 0
        rows = rep.split("1.1 FORA_WHOLE_GRAPH\n")[1].split("\n\n")[0].strip().splitlines()
        g.set_tuning(pkg.tuning_batch())
        for draw, eps in enumerate(eps_arr):
            srcs = _query_nodes(SEED, draw, Q, got.n)
            out, _, _, _, _, _ = g.fora_batch_single_source(srcs, eps, 0.15, seed=SEED, fetch=True)
            err = sum(orc.max_err(out[i], truth(s)) for i, s in enumerate(srcs)) / Q
            p, ms, rep_err = rows[draw].split(",")
            assert float(p) == eps and float(rep_err) == pytest.approx(err, abs=1e-12), (eps, rep_err, err)
        g.set_tuning(pkg.tuning_default())
        # 2.1 FORA_TOPK: calls 20..24 (Test 1 made 4 x 5); query i runs with seed + i
        rows = rep.split("2.1 FORA_TOPK\n")[1].split("\n\n")[0].strip().splitlines()
        for j, eps in enumerate(eps_arr):
            srcs = _query_nodes(SEED, 20 + j, Q, got.n)
            sp = sn = 0.0
            for i, s in enumerate(srcs):
                n_sel, ids, vals, _, _ = g.fora_topk(s, eps, 0.15, K, seed=SEED + i, cap=got.n)
                pm = truth(s)
                cnt, gids, _ = orc.topk(pm, K, cap=got.n)
                sp += orc.precision(ids, gids)
                sn += orc.ndcg(ids, gids, pm)
            p, k, ms, prec, ndcg = rows[j].split(",")
            assert float(p) == eps and int(k) == K
            assert float(prec) == pytest.approx(sp / Q, abs=1e-12) and float(ndcg) == pytest.approx(sn / Q, abs=1e-12)
