"""CPU tests of the product's host side: the C-ABI library loads and exports every symbol
include/pprhip.h declares, the graph ingest (R-MAT generator, neo4j-import CSV reader, CSR build),
the reference's text format, the index merge, and the loud failure without a GPU.  No compute
call is made here (the engine has no CPU path)."""
import os
import re

import numpy as np
import pytest

from conftest import GOT_NODES, GOT_RELS, ROOT, edges_to_host, to_oracle


def test_library_exports_every_declared_symbol(pkg):
    hdr = open(os.path.join(ROOT, "include", "pprhip.h")).read()
    declared = set(re.findall(r"\b(pprhip_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    lib = pkg.lib()
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, "libpprhip.so does not export: %s" % missing
    assert declared == set(pkg.EXPORTS), sorted(declared ^ set(pkg.EXPORTS))
    assert lib.pprhip_version() == 100


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under the package may import, link or load it."""
    pk = os.path.join(ROOT, "personalized-pagerank-algorithms-on-neo4j_amd")
    for dp, _, files in os.walk(pk):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h", "Makefile")):
                txt = open(os.path.join(dp, f), errors="replace").read()
                for needle in ("libppr_oracle", '#include "ppr_oracle', "#include <ppr_oracle", "from oracle",
                               "import oracle", "orc_"):
                    assert needle not in txt, "%s references the oracle (%s)" % (f, needle)


def test_no_gpu_fails_loudly(pkg, got):
    if pkg.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.PprhipError) as e:
        pkg.Graph(got)
    assert e.value.code == pkg.ERR_NO_DEVICE and "no CPU fallback" in str(e.value)


def test_rmat_generator(pkg):
    src, dst = pkg.rmat_edges(10, 16, seed=1)
    assert src.size == dst.size == 16 << 10 and src.min() >= 0 and max(src.max(), dst.max()) < 1 << 10
    s2, d2 = pkg.rmat_edges(10, 16, seed=1)
    assert np.array_equal(src, s2) and np.array_equal(dst, d2)           # deterministic
    s3, d3 = pkg.rmat_edges(10, 16, seed=2)
    assert not np.array_equal(src, s3)
    deg = np.bincount(src, minlength=1 << 10)
    assert deg.max() > 20 * np.median(deg[deg > 0])                        # heavy tail
    assert (deg == 0).sum() > 100                                          # many dead ends, as Graph500 R-MAT has
    big = pkg.rmat_edges(14, 16, seed=1)
    assert np.unique(big[0]).size > 5000


def test_csr_build_matches_numpy(pkg):
    rng = np.random.default_rng(0)
    n, m = 50, 400
    src = rng.integers(0, n, m).astype(np.int32)
    dst = rng.integers(0, n, m).astype(np.int32)
    for newest in (False, True):
        rp, ci = pkg.csr_build(n, src, dst, newest_first=newest)
        assert rp[0] == 0 and rp[-1] == m and np.array_equal(np.diff(rp), np.bincount(src, minlength=n))
        for v in range(n):
            want = dst[src == v]  # edge order
            if newest:
                want = want[::-1]
            assert np.array_equal(ci[rp[v]:rp[v + 1]], want)
    with pytest.raises(pkg.PprhipError):
        pkg.csr_build(n, np.array([0, n], dtype=np.int32), np.array([0, 0], dtype=np.int32))


def test_neo4j_csv_reader(pkg, tmp_path):
    n, src, dst, names = pkg.load_neo4j_csv(GOT_NODES, GOT_RELS)
    assert n == 107 and len(src) == 352 and names[0] == "Aemon" and names[-1] == "Doran"
    assert (names[src[0]], names[dst[0]]) == ("Aemon", "Grenn")  # GOT_Rels.csv row 1; id = row index
    bad = tmp_path / "rels.csv"
    bad.write_text(":START_ID,:END_ID,:TYPE\nAemon,Nobody,Relation\n")
    with pytest.raises(pkg.PprhipError) as e:
        pkg.load_neo4j_csv(GOT_NODES, str(bad))
    assert e.value.code == pkg.ERR_IO
    with pytest.raises(pkg.PprhipError):
        pkg.load_neo4j_csv(str(tmp_path / "missing.csv"), GOT_RELS)


def test_neo4j_store_reader(pkg, got, tmp_path):
    """target/got.db's record files, read without a JVM, give the same adjacency (incl. HeavyGraph's
    neighbour order = relationship-chain order) as the import CSVs the store was built from."""
    st = pkg.HostCsr.from_neo4j_store(os.path.join(ROOT, "tests", "golden", "got.db"))
    assert (st.n, st.m) == (107, 352)
    for k in ("out_rp", "out_ci", "in_rp", "in_ci"):
        assert np.array_equal(getattr(st, k), getattr(got, k)), k
    with pytest.raises(pkg.PprhipError) as e:
        pkg.HostCsr.from_neo4j_store(str(tmp_path))
    assert e.value.code == pkg.ERR_IO
    # a truncated relationship store breaks a chain and is reported, not mis-read
    import shutil
    d = tmp_path / "bad.db"
    shutil.copytree(os.path.join(ROOT, "tests", "golden", "got.db"), d)
    rel = d / "neostore.relationshipstore.db"
    rel.write_bytes(rel.read_bytes()[:34 * 100])
    with pytest.raises(pkg.PprhipError):
        pkg.HostCsr.from_neo4j_store(str(d))


def test_java_double_to_string(pkg):
    cases = {1.0: "1.0", 0.5: "0.5", 0.001: "0.001", 9.999e-4: "9.999E-4", 1.0e-4: "1.0E-4", 1234567.0: "1234567.0",
             1.0e7: "1.0E7", 12345678.9: "1.23456789E7", 0.1: "0.1", 1 / 3: "0.3333333333333333",
             2.2250738585072014e-308: "2.2250738585072014E-308", 0.0: "0.0", 100.0: "100.0",
             0.15000000000000002: "0.15000000000000002", 6.02e23: "6.02E23"}
    for d, s in cases.items():
        assert pkg.format_double(d) == s
    rng = np.random.default_rng(1)
    for d in rng.random(200) * 10.0 ** rng.integers(-12, 3, 200):
        assert float(pkg.format_double(float(d)).replace("E", "e")) == d  # always round-trips


def test_index_merge_and_files(pkg, orc, got, tmp_path):
    """pprhip_index_merge re-applies Base_Whole_Graph's k rule over target shards; checked against
    the oracle's unsharded All-Pair result (inputs come from the oracle: the merge itself is host code)."""
    og = to_oracle(orc, got)
    thr = 1e-3
    for k in (-1, 4):
        parts = []
        for lo, hi in ((0, 40), (40, 80), (80, got.n)):
            off, tg, vl = og.all_pair_backward(0.15, thr, -1, lo, hi)
            parts.append(pkg.index_from_arrays(got.n, off, tg, vl))
        merged = pkg.merge_indexes(parts, k)
        off, tg, vl = merged.arrays()
        ooff, otg, ovl = og.all_pair_backward(0.15, thr, k)
        assert np.array_equal(off, ooff) and np.array_equal(tg, otg) and np.array_equal(vl, ovl)
    d = tmp_path / "BASE_ppr_results" / "got.db" / ("%s_%d" % (pkg.format_double(thr), 4))
    merged.write_dir(str(d))
    files = sorted(os.listdir(d))
    assert len(files) == int((np.diff(off) > 0).sum()) and all(f.endswith(".txt") for f in files)
    v = int(files[0][:-4])
    lines = open(d / files[0]).read().splitlines()
    assert len(lines) == off[v + 1] - off[v]
    t0, p0 = lines[0].split("\t")
    assert int(t0) == tg[off[v]] and float(p0.replace("E", "e")) == vl[off[v]]  # "<id>\t<Double.toString>\n"


def test_conf_validation(pkg):
    with pytest.raises(pkg.PprhipError):
        pkg.conf_topk(100, 1000, 0, 0.15)
    c = pkg.conf_whole_graph(100, 1000, 0.2)
    assert (c.alpha, c.delta, c.pfail, c.rsum, c.n, c.m) == (0.2, 0.01, 0.01, 1.0, 100, 1000)
    t = pkg.tuning_default()
    assert t.c_walk_ns > 0 and 0 < t.dense_frac < 1 and t.max_rounds == 24


def test_tuning_defaults_match_oracle_twin(pkg, orc):
    """The twin takes the same round count only if both sides evaluate the same cost model."""
    import ctypes
    a, b = pkg.tuning_default(), orc.tuning_default()
    assert [f for f, _ in a._fields_] == [f for f, _ in b._fields_]
    assert ctypes.sizeof(a) == ctypes.sizeof(b) == 88          # 9 doubles + 4 int32 (include/pprhip.h, oracle/ppr_oracle.h)
    for f, _ in a._fields_:
        assert getattr(a, f) == getattr(b, f), f
    assert (a.halving_ratio, a.max_halvings, a.prior_levels) == (2.0, 6, 16)
    assert (a.gs_blocks, a.gs_frac) == (2, 0.1)
    t = pkg.tuning_batch()                                       # the batch profile changes the dense-level terms only
    changed = [f for f, _ in t._fields_ if getattr(t, f) != getattr(a, f)]
    assert sorted(changed) == ["c_dense_edge_ns", "c_dense_node_ns", "dense_frac", "gs_frac"]


def test_index_arrays_are_validated(pkg):
    """Targets outside [0, n) are an error at the boundary, not an out-of-range write later (the same check guards
    the entries that arrive from the device and from the exchange, allpair.cpp: index_from_triples)."""
    off = np.array([0, 1, 2], dtype=np.uint64)
    with pytest.raises(pkg.PprhipError):
        pkg.index_from_arrays(2, off, np.array([0, 2], dtype=np.int32), np.array([0.5, 0.5]))
    with pytest.raises(pkg.PprhipError):
        pkg.index_from_arrays(2, off, np.array([-1, 1], dtype=np.int32), np.array([0.5, 0.5]))
    ok = pkg.index_from_arrays(2, off, np.array([1, 0], dtype=np.int32), np.array([0.5, 0.25]))
    assert pkg.merge_indexes([ok], -1).arrays()[1].tolist() == [1, 0]


def test_index_merge_large_runs_on_all_threads(pkg):
    """Enough entries (> 2^16) for the multi-threaded path of the index finalisation: a two-level counting sort by
    source on all threads, then the k rule per source - against a plain numpy restatement of Base_Whole_Graph's rule
    (entries >= the k-th largest, value descending, ties in target order; k < 0: target order)."""
    rng = np.random.default_rng(5)
    n, per = 3000, 64
    shards = []
    rows = {}
    for sh in range(3):                                  # three target shards with disjoint targets (node ids < n)
        off = np.zeros(n + 1, dtype=np.uint64)
        tg, vl = [], []
        for v in range(n):
            cnt = int(rng.integers(0, per)) if v % 7 else 0
            t = np.sort(rng.choice(np.arange(sh * 1000, sh * 1000 + 1000), size=cnt, replace=False)).astype(np.int32)
            p = np.round(rng.random(cnt), 2)             # many ties
            off[v + 1] = off[v] + cnt
            tg.append(t)
            vl.append(p)
            rows.setdefault(v, []).extend(zip(t.tolist(), p.tolist()))
        shards.append(pkg.index_from_arrays(n, off, np.concatenate(tg).astype(np.int32), np.concatenate(vl)))
    assert sum(len(r) for r in rows.values()) > (1 << 16)
    for k in (-1, 5):
        off, tg, vl = pkg.merge_indexes(shards, k).arrays()
        for v in range(0, n, 37):
            r = sorted(rows.get(v, []))                  # target order
            if k >= 0 and len(r) >= k:
                kth = sorted((p for _, p in r), reverse=True)[k - 1]
                r = [e for e in r if e[1] >= kth]
            if k >= 0:
                r = sorted(r, key=lambda e: -e[1])       # stable: ties stay in target order
            got = list(zip(tg[off[v]:off[v + 1]].tolist(), vl[off[v]:off[v + 1]].tolist()))
            assert got == r, (k, v)


# ------------------------------------------------------------------ graph lift, host half (row a11; PPR.java:136-152)
def _panel_expected(n, m, in_rp, in_ci, nz_rows, panel=8192, step=8192, item_edges=32768, fold_parts=32, fold_min=16):
    """The row-panel copy of the in-CSR (engine_internal.hpp: HostPanelLayout) restated with numpy sorts: panels of
    8 192 consecutive rows with in-edges, a panel's in-edges sorted by (source, row); a panel of more than 32 768 edges
    is cut into S = ceil(edges / 32 768) parts [e k / S, e (k + 1) / S); every part padded to whole turns of 8 192 edges
    with (0, 0xffff); part k of a panel of `rows` rows leaves its sums at base + k rows + local row; a panel of more than 16
    parts has room behind all parts for their sums 32 at a time."""
    indeg = np.diff(in_rp).astype(np.int64)
    n_nz = nz_rows.size
    n_panels = (n_nz + panel - 1) // panel
    row_of_edge = np.repeat(np.arange(n, dtype=np.int64), indeg)
    ordinal = np.cumsum(indeg > 0) - 1
    j_of_edge = ordinal[row_of_edge]
    p_of_edge = j_of_edge // panel
    perm = np.lexsort((j_of_edge, in_ci, p_of_edge))       # panel, then source, then row
    e_src, e_row = in_ci[perm], (j_of_edge % panel)[perm]
    edges_p = np.bincount(p_of_edge, minlength=n_panels).astype(np.int64)
    S = np.maximum(1, (edges_p + item_edges - 1) // item_edges)
    item0 = np.zeros(n_panels + 1, dtype=np.int64)
    item0[1:] = np.cumsum(S)
    rows_p = np.minimum(panel, n_nz - np.arange(n_panels) * panel)
    base = np.zeros(n_panels + 1, dtype=np.int64)
    base[1:] = np.cumsum(rows_p * S)
    desc = np.zeros((n_panels, 4), dtype=np.uint32)
    desc[:, 0], desc[:, 1], desc[:, 2], desc[:, 3] = base[:-1], S, rows_p, 0xffffffff
    n_part = int(base[-1])
    for t in np.nonzero(S > fold_min)[0]:                  # room for the sums of 32 parts at a time, behind all parts
        desc[t, 3] = n_part
        n_part += int(rows_p[t]) * int((S[t] + fold_parts - 1) // fold_parts)
    items = np.zeros((int(item0[-1]), 4), dtype=np.uint32)
    first_edge = np.zeros(n_panels + 1, dtype=np.int64)
    first_edge[1:] = np.cumsum(edges_p)
    src_out, row_out = [], []
    st = 0
    for t in range(n_panels):
        e = int(edges_p[t])
        for k in range(int(S[t])):
            lo, hi = e * k // int(S[t]), e * (k + 1) // int(S[t])
            steps = (hi - lo + step - 1) // step
            items[item0[t] + k] = (st, steps, t, base[t] + k * rows_p[t])
            pad = steps * step - (hi - lo)
            src_out += [e_src[first_edge[t] + lo:first_edge[t] + hi], np.zeros(pad, dtype=np.int32)]
            row_out += [e_row[first_edge[t] + lo:first_edge[t] + hi].astype(np.uint16), np.full(pad, 0xffff, dtype=np.uint16)]
            st += steps
    # a wave's 512 edges of a turn are stored lane by lane: lane l holds the edges l, l + 64, ... (eight of them)
    lanes = lambda a: a.reshape(-1, 8, 64).transpose(0, 2, 1).ravel()
    return dict(panel_sizes=np.array([n_panels, item0[-1], n_part, st * step], dtype=np.uint64),
                panel_src=lanes(np.concatenate(src_out).astype(np.int32)), panel_row=lanes(np.concatenate(row_out).astype(np.uint16)),
                panel_items=items.ravel(), panel_desc=desc.ravel(), panel_item0=item0.astype(np.uint32))


def _lift_expected(h, width=393216, chunk=512, max_windows=16, panels=None):
    """The internal layout restated with numpy (stable sorts instead of the library's counting sort and threaded
    passes): vertex order = nodes with in-edges first, then out-degree descending, ties by id; rows keep their
    order; row-start flags; the sliced copy = in-edges in row order, stably partitioned by slice of the source id."""
    n, m = h.n, h.m
    od = np.diff(h.out_rp).astype(np.int64)
    ind = np.diff(h.in_rp).astype(np.int64)
    key = np.where(ind > 0, 0, 1) * (1 << 40) + ((1 << 32) - od)
    new2old = np.argsort(key, kind="stable").astype(np.int32)
    old2new = np.empty(n, dtype=np.int32)
    old2new[new2old] = np.arange(n, dtype=np.int32)

    def rename(rp, ci):
        deg = np.diff(rp).astype(np.int64)[new2old]
        nrp = np.zeros(n + 1, dtype=np.uint32)
        nrp[1:] = np.cumsum(deg)
        starts = rp[:-1].astype(np.int64)[new2old]
        idx = np.repeat(starts - nrp[:-1].astype(np.int64), deg) + np.arange(m, dtype=np.int64)
        return nrp, old2new[ci[idx]].astype(np.int32)

    out_rp, out_ci = rename(h.out_rp, h.out_ci)
    in_rp, in_ci = rename(h.in_rp, h.in_ci)
    indeg, outdeg = np.diff(in_rp), np.diff(out_rp)
    nz_rows = np.nonzero(indeg > 0)[0].astype(np.int32)
    zin_rows = np.nonzero((indeg == 0) & (outdeg > 0))[0].astype(np.int32)
    n_chunks = (m + chunk - 1) // chunk

    def flag_arrays(starts):
        bits = np.zeros((n_chunks + 1) * chunk, dtype=np.uint8)
        bits[starts] = 1
        flags = np.packbits(bits, bitorder="little")
        cs = np.zeros(n_chunks + 1, dtype=np.uint32)
        cs[1:] = np.cumsum(np.bincount(np.asarray(starts, dtype=np.int64) // chunk, minlength=n_chunks)[:n_chunks])
        return flags, cs

    row_first = in_rp[:-1][nz_rows].astype(np.int64)
    flags, chunk_starts = flag_arrays(row_first)
    last = in_rp[1:][nz_rows].astype(np.int64) - 1
    cross_b = (row_first // chunk != last // chunk) | ((last + 1) % chunk == 0) | (last + 1 == m)
    cross = np.zeros((n + 63) // 64 + 1, dtype=np.uint64)
    j = np.nonzero(cross_b)[0]
    np.bitwise_or.at(cross, j >> 6, np.uint64(1) << (j & 63).astype(np.uint64))
    exp = dict(new2old=new2old, old2new=old2new, out_rp=out_rp, out_ci=out_ci, in_rp=in_rp, in_ci=in_ci, nz_rows=nz_rows,
               zin_rows=zin_rows, flags=flags, chunk_starts=chunk_starts, cross=cross)
    if panels is None:
        panels = m >= (1 << 26)
    if panels and m and nz_rows.size:  # graphs with the row-panel copy have no sliced one
        exp.update(_panel_expected(n, m, in_rp, in_ci, nz_rows))
        return exp
    n_src = int(np.nonzero(outdeg > 0)[0].max()) + 1 if (outdeg > 0).any() else 0
    S = (n_src + width - 1) // width
    if S > max_windows:
        S = max_windows
        width = (n_src + S - 1) // S
    if S >= 2 and m and nz_rows.size:
        row_of_edge = np.repeat(np.arange(n, dtype=np.int64), indeg)  # (edges of rows without in-edges: none)
        ordinal = np.cumsum(indeg > 0) - 1
        sl = np.minimum(in_ci.astype(np.int64) // width, S - 1)
        perm = np.argsort(sl, kind="stable")
        sl_ci = in_ci[perm]
        seg_key = sl[perm] * n + row_of_edge[perm]
        is_start = np.ones(m, dtype=bool)
        is_start[1:] = seg_key[1:] != seg_key[:-1]
        seg_off = np.nonzero(is_start)[0].astype(np.uint32)
        seg_row = ordinal[row_of_edge[perm][seg_off]].astype(np.uint32)
        sl_flags, sl_cs = flag_arrays(seg_off.astype(np.int64))
        edge_base = np.zeros(S + 1, dtype=np.uint64)
        edge_base[1:] = np.cumsum(np.bincount(sl, minlength=S))
        seg_base = np.zeros(S + 1, dtype=np.uint64)
        seg_base[1:] = np.cumsum(np.bincount(sl[perm][seg_off], minlength=S))
        exp.update(sl_ci=sl_ci, sl_flags=sl_flags, sl_chunk_starts=sl_cs, seg_row=seg_row, seg_off=seg_off,
                   edge_base=edge_base, seg_base=seg_base)
    return exp


@pytest.mark.parametrize("scale,slice_ids,panels", [(10, 100, None), (10, 0, "1"), (16, 20000, None), (16, 0, "1"), (17, 0, "1")])
def test_lift_host_matches_numpy_restatement(pkg, monkeypatch, scale, slice_ids, panels):
    """pprhip_graph_lift_host (what pprhip_graph_create uploads) against an independent numpy restatement of the
    layout, on one thread and on eight (scale 16/17 are large enough for the threaded passes): same bytes.  Graphs from
    2^26 edges on get the row-panel copy instead of the sliced one (PPRHIP_SWEEP1_PANELS=0 / 1 forces either)."""
    if slice_ids:
        monkeypatch.setenv("PPRHIP_SLICE_IDS", str(slice_ids))
    if panels is not None:
        monkeypatch.setenv("PPRHIP_SWEEP1_PANELS", panels)
    h = pkg.HostCsr.rmat(scale, 16, seed=4)
    exp = _lift_expected(h, width=slice_ids or 393216, panels=None if panels is None else panels == "1")
    for threads in (1, 8):
        got = pkg.lift_host(h, threads=threads)
        for name, want in exp.items():
            assert got[name].shape == want.shape, (name, threads, got[name].shape, want.shape)
            assert np.array_equal(got[name], want), (name, threads)
        if "sl_ci" not in exp:
            assert got["sl_ci"].size == 0 and got["seg_row"].size == 0
        if "panel_src" not in exp:
            assert got["panel_src"].size == 0 and got["panel_sizes"].size == 0
    # without the caller's in-adjacency the lift derives one: same vertex order and out-CSR, and an in-CSR that is
    # the transpose (rows as multisets; the order inside a derived row follows the renamed out-CSR)
    own = pkg.lift_host(h, threads=8, with_in=False)
    assert np.array_equal(own["new2old"], exp["new2old"]) and np.array_equal(own["out_ci"], exp["out_ci"])
    assert np.array_equal(own["in_rp"], exp["in_rp"])
    rows = np.repeat(np.arange(h.n), np.diff(exp["in_rp"]))
    a = np.stack([rows, own["in_ci"]])
    b = np.stack([rows, exp["in_ci"]])
    assert np.array_equal(a[:, np.lexsort(a[::-1])], b[:, np.lexsort(b[::-1])])


def test_lift_host_reports_the_first_bad_entry(pkg):
    """Validation runs on several threads; the offence reported is the first in array order, as one thread would."""
    h = pkg.HostCsr.rmat(16, 16, seed=4)
    ci = h.out_ci.copy()
    ci[900001] = h.n  # two offences in different threads' ranges
    ci[77] = -5
    bad = type("H", (), dict(n=h.n, m=h.m, out_rp=h.out_rp, out_ci=ci, in_rp=h.in_rp, in_ci=h.in_ci))()
    with pytest.raises(pkg.PprhipError, match=r"out_col_idx\[77\] = -5"):
        pkg.lift_host(bad, threads=8)
    rp = h.in_rp.copy()
    rp[1000], rp[1001] = rp[1001] + 1, rp[1000]  # (keeps rp[n] == m)
    bad = type("H", (), dict(n=h.n, m=h.m, out_rp=h.out_rp, out_ci=h.out_ci, in_rp=rp, in_ci=h.in_ci))()
    with pytest.raises(pkg.PprhipError, match=r"in_row_ptr decreases at node 1000"):
        pkg.lift_host(bad, threads=8)
    in_ci = np.roll(h.in_ci, 1)
    rp2 = h.in_rp.copy()
    v = int(np.nonzero(np.diff(h.in_rp) > 1)[0][0])
    rp2[v + 1] -= 1  # one in-edge moves to the next node: degrees no longer those of the transpose
    bad = type("H", (), dict(n=h.n, m=h.m, out_rp=h.out_rp, out_ci=h.out_ci, in_rp=rp2, in_ci=in_ci))()
    with pytest.raises(pkg.PprhipError, match=r"not the transpose"):
        pkg.lift_host(bad, threads=8)
    # the right number of in-edges at every node, but two of them name the wrong sources: the transpose check compares
    # the edges themselves (a sum over a mix of every (source, destination) pair), not only the degrees
    in_ci = h.in_ci.copy()
    rows = np.nonzero(np.diff(h.in_rp) > 0)[0]
    a, b = int(h.in_rp[rows[3]]), int(h.in_rp[rows[-3]])
    assert in_ci[a] != in_ci[b]
    in_ci[a], in_ci[b] = in_ci[b], in_ci[a]
    bad = type("H", (), dict(n=h.n, m=h.m, out_rp=h.out_rp, out_ci=h.out_ci, in_rp=h.in_rp, in_ci=in_ci))()
    for threads in (1, 8):
        with pytest.raises(pkg.PprhipError, match=r"not from the same sources"):
            pkg.lift_host(bad, threads=threads)


# ------------------------------------------------------------------ Neo4j store with dense nodes (relationship groups)
def _write_neo4j_store(path, n, src, dst, dense_threshold=50):
    """Record files of a Neo4j 3.x store ("standard" format, big-endian) for the given relationships: nodestore (15-byte
    records), relationshipstore (34), relationshipgroupstore (25; record 0 = the header holding the dense-node threshold,
    as in the reference's target/got.db: 00 00 00 32).  Nodes whose degree reaches the threshold are dense: their
    record points to a relationship group with separate outgoing / incoming / loop chains; the others keep one chain.
    New relationships go to the head of their chains (newest first), as the kernel links them."""
    import struct
    NO = 0xFFFFFFFF
    m = len(src)
    deg = np.bincount(src, minlength=n) + np.bincount(dst, minlength=n) - np.bincount(src[src == dst], minlength=n)
    dense = deg >= dense_threshold
    head = np.full(n, NO, dtype=np.int64)          # sparse nodes: chain head
    ghead = {int(v): [NO, NO, NO] for v in np.nonzero(dense)[0]}  # dense: [out, in, loop]
    fnext = np.full(m, NO, dtype=np.int64)
    snext = np.full(m, NO, dtype=np.int64)
    for r in range(m):
        a, b = int(src[r]), int(dst[r])
        for v, is_first in ((a, True), (b, False)):
            if a == b and not is_first:
                continue
            if dense[v]:
                slot = 2 if a == b else (0 if is_first else 1)
                prev, ghead[v][slot] = ghead[v][slot], r
            else:
                prev, head[v] = head[v], r
            if a == b:
                fnext[r] = snext[r] = prev
            elif is_first:
                fnext[r] = prev
            else:
                snext[r] = prev
    gid_of = {v: i + 1 for i, v in enumerate(sorted(ghead))}
    nodes = bytearray()
    for v in range(n):
        nxt = gid_of[v] if dense[v] else int(head[v])
        nodes += struct.pack(">BII5sB", 1, nxt & NO, NO, b"\0" * 5, 1 if dense[v] else 0)
    rels = bytearray()
    for r in range(m):
        rels += struct.pack(">BIIIIIIIIB", 1, int(src[r]), int(dst[r]), 0, 0, int(fnext[r]), 0, int(snext[r]), NO, 0)
    groups = bytearray(struct.pack(">I", dense_threshold) + b"\0" * 21)
    for v in sorted(ghead):
        o, i, l = ghead[v]
        groups += struct.pack(">BBHIIIIIB", 1, 0, 0, NO, o, i, l, v, 0)
    os.makedirs(path, exist_ok=True)
    for name, blob in (("neostore.nodestore.db", nodes), ("neostore.relationshipstore.db", rels),
                       ("neostore.relationshipgroupstore.db", groups)):
        with open(os.path.join(path, name), "wb") as f:
            f.write(bytes(blob))
    return dense


def test_neo4j_store_reader_dense_nodes(pkg, tmp_path):
    """Nodes with 50 relationships or more are "dense" in a Neo4j store: their relationships hang off relationship-group
    records, not off one chain.  got.db has none, every real dataset does (the thesis' GRQC, BlogCatalog, ...: Diss.
    p.36).  A store written here to the record format - hub nodes, loops on dense and on sparse nodes, parallel edges -
    is read back: every node's out- / in-list in chain order (dense: outgoing chain, incoming chain, then loops)."""
    rng = np.random.default_rng(11)
    n = 300
    src = np.concatenate([rng.integers(0, n, 900), np.full(80, 7), rng.integers(0, n, 70), [7, 7, 9, 250]])
    dst = np.concatenate([rng.integers(0, n, 900), rng.integers(0, n, 80), np.full(70, 13), [7, 7, 9, 250]])
    src, dst = src.astype(np.int64), dst.astype(np.int64)
    dst[5], src[5] = dst[4], src[4]  # a parallel edge
    d = str(tmp_path / "dense.db")
    dense = _write_neo4j_store(d, n, src, dst)
    assert dense[7] and dense[13] and not dense[9] and dense.sum() >= 2
    st = pkg.HostCsr.from_neo4j_store(d)
    assert (st.n, st.m) == (n, src.size)
    rid = np.arange(src.size)
    for v in range(n):
        outs, ins, loops = rid[(src == v) & (dst != v)][::-1], rid[(dst == v) & (src != v)][::-1], rid[(src == v) & (dst == v)][::-1]
        if dense[v]:
            exp_out = list(dst[outs]) + [v] * loops.size
            exp_in = list(src[ins]) + [v] * loops.size
        else:
            touch = rid[(src == v) | (dst == v)][::-1]
            exp_out = [int(dst[r]) for r in touch if src[r] == v]
            exp_in = [int(src[r]) for r in touch if dst[r] == v]
        assert list(st.out_ci[st.out_rp[v]:st.out_rp[v + 1]]) == exp_out, v
        assert list(st.in_ci[st.in_rp[v]:st.in_rp[v + 1]]) == exp_in, v
    # a group that belongs to another node, and a missing group store, are reported
    g = os.path.join(d, "neostore.relationshipgroupstore.db")
    blob = bytearray(open(g, "rb").read())
    blob[25 + 23] ^= 1  # owning node of group 1
    open(g, "wb").write(bytes(blob))
    with pytest.raises(pkg.PprhipError, match="belongs to node"):
        pkg.HostCsr.from_neo4j_store(d)
    os.remove(g)
    with pytest.raises(pkg.PprhipError, match="relationshipgroupstore"):
        pkg.HostCsr.from_neo4j_store(d)


def test_lift_host_without_relabeling(pkg, monkeypatch):
    """PPRHIP_RELABEL=0: the caller's ids are the internal order - both CSRs come through unchanged, the sweep's row list
    is the nodes with in-edges in id order, on one thread and on eight."""
    monkeypatch.setenv("PPRHIP_RELABEL", "0")
    h = pkg.HostCsr.rmat(16, 16, seed=9)
    for threads in (1, 8):
        got = pkg.lift_host(h, threads=threads)
        assert np.array_equal(got["new2old"], np.arange(h.n)) and np.array_equal(got["old2new"], np.arange(h.n))
        for k in ("out_rp", "out_ci", "in_rp", "in_ci"):
            assert np.array_equal(got[k], getattr(h, k)), k
        indeg, outdeg = np.diff(h.in_rp), np.diff(h.out_rp)
        assert np.array_equal(got["nz_rows"], np.nonzero(indeg > 0)[0])
        assert np.array_equal(got["zin_rows"], np.nonzero((indeg == 0) & (outdeg > 0))[0])
        starts = h.in_rp[:-1][got["nz_rows"]]
        bits = np.unpackbits(got["flags"], bitorder="little")
        assert bits.sum() == starts.size and bits[starts].all()


def test_neo4j_store_reader_on_all_threads(pkg, tmp_path):
    """A store large enough for the reader's threaded passes (relationship ranges, node ranges): 6 000 nodes, 40 000
    relationships, a few dense hubs - the same lists as one pass in chain order gives."""
    rng = np.random.default_rng(12)
    n, m = 6000, 40000
    src = rng.integers(0, n, m).astype(np.int64)
    dst = rng.integers(0, n, m).astype(np.int64)
    src[rng.integers(0, m, 300)] = 17
    dst[rng.integers(0, m, 300)] = 4242
    d = str(tmp_path / "big.db")
    dense = _write_neo4j_store(d, n, src, dst)
    assert dense[17] and dense[4242]
    st = pkg.HostCsr.from_neo4j_store(d)
    assert (st.n, st.m) == (n, m)
    rid = np.arange(m)
    for v in [17, 4242] + list(rng.integers(0, n, 200)):
        outs, ins, loops = rid[(src == v) & (dst != v)][::-1], rid[(dst == v) & (src != v)][::-1], rid[(src == v) & (dst == v)][::-1]
        if dense[v]:
            exp_out, exp_in = list(dst[outs]) + [v] * loops.size, list(src[ins]) + [v] * loops.size
        else:
            touch = rid[(src == v) | (dst == v)][::-1]
            exp_out = [int(dst[r]) for r in touch if src[r] == v]
            exp_in = [int(src[r]) for r in touch if dst[r] == v]
        assert list(st.out_ci[st.out_rp[v]:st.out_rp[v + 1]]) == exp_out, v
        assert list(st.in_ci[st.in_rp[v]:st.in_rp[v + 1]]) == exp_in, v
    assert np.array_equal(np.diff(st.out_rp), np.bincount(src, minlength=n))
    assert np.array_equal(np.diff(st.in_rp), np.bincount(dst, minlength=n))
    # a relationship that points outside the store is reported with its id, whichever thread meets it
    rel = os.path.join(d, "neostore.relationshipstore.db")
    blob = bytearray(open(rel, "rb").read())
    blob[34 * 31000 + 1:34 * 31000 + 5] = (n + 5).to_bytes(4, "big")
    open(rel, "wb").write(bytes(blob))
    with pytest.raises(pkg.PprhipError, match="relationship 31000 references a node outside"):
        pkg.HostCsr.from_neo4j_store(d)


def test_lift_host_small_random_graphs(pkg, monkeypatch):
    """Small random multigraphs - self loops, parallel edges, isolated nodes, nodes without in- or out-edges, no edge at
    all - through the host lift with narrow slices, against the numpy restatement."""
    rng = np.random.default_rng(23)
    for trial in range(60):
        n = int(rng.integers(1, 40))
        m = int(rng.integers(0, 120)) if trial % 7 else 0
        width = int(rng.integers(1, 12))
        monkeypatch.setenv("PPRHIP_SLICE_IDS", str(width))
        lo = int(rng.integers(0, n))
        hi = int(rng.integers(lo, n)) + 1
        src = rng.integers(lo, hi, m).astype(np.int32)   # (a range of sources only: the others have no out-edges)
        dst = rng.integers(0, n, m).astype(np.int32)
        h = pkg.HostCsr(n, src, dst)
        got = pkg.lift_host(h, threads=1)
        exp = _lift_expected(h, width=width)
        for name, want in exp.items():
            assert np.array_equal(got[name], want), (trial, name, n, m, width)
        if "sl_ci" not in exp:
            assert got["seg_row"].size == 0
