"""Shared fixtures.  GPU tests are marked `gpu` and call the HIP engine through the C ABI;
everything else runs on CPU (oracle, host logic, library load/export checks)."""
import importlib
import os
import sys

import numpy as np
import pytest

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")  # as the launchers run the product (host/ppr_main.cpp)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PKG_NAME = "personalized-pagerank-algorithms-on-neo4j_amd"
GOT_NODES = os.path.join(ROOT, "tests", "golden", "got", "GOT_Nodes.csv")
GOT_RELS = os.path.join(ROOT, "tests", "golden", "got", "GOT_Rels.csv")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_pkg():
    return importlib.import_module(PKG_NAME)


@pytest.fixture(scope="session")
def pkg():
    return load_pkg()


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle
    oracle.build()
    return oracle


def edges_to_host(pkg, n, edges, newest_first=False):
    src = np.array([e[0] for e in edges], dtype=np.int32)
    dst = np.array([e[1] for e in edges], dtype=np.int32)
    return pkg.HostCsr(n, src, dst, newest_first)


def to_oracle(orc, host):
    return orc.OracleGraph(host.n, host.out_rp, host.out_ci, host.in_rp, host.in_ci)


@pytest.fixture(scope="session")
def got(pkg):
    return pkg.HostCsr.from_neo4j_csv(GOT_NODES, GOT_RELS)


@pytest.fixture(scope="session")
def toy_graphs(pkg):
    """Small graphs with closed-form answers / awkward structure."""
    g = {}
    g["two_node"] = edges_to_host(pkg, 2, [(0, 1)])                      # Dissertation p.13-14
    g["cycle5"] = edges_to_host(pkg, 5, [(i, (i + 1) % 5) for i in range(5)])
    g["star_dead_leaves"] = edges_to_host(pkg, 6, [(0, i) for i in range(1, 6)])
    g["isolated_mix"] = edges_to_host(pkg, 6, [(0, 1), (1, 2), (2, 0), (2, 3), (0, 0), (1, 2)])  # self loop, multi-edge, 4/5 isolated
    g["line"] = edges_to_host(pkg, 8, [(i, i + 1) for i in range(7)])
    return g


@pytest.fixture(scope="session")
def rmat12(pkg):
    return pkg.HostCsr.rmat(12, 16, seed=1)


@pytest.fixture(scope="session")
def rmat15(pkg):
    return pkg.HostCsr.rmat(15, 16, seed=1)


@pytest.fixture(autouse=True, scope="session")
def _twin_follows_engine_tuning(pkg):
    """Since the Gauss-Seidel sweeps the level shapes (dense_frac, gs_blocks, gs_frac) change the push schedule, so
    the twin's entry points that take no tuning argument (forward_push, topk_push, fora_topk) must run with the
    tuning the engine was given: every Graph.set_tuning in a test also sets the twin's."""
    from oracle import oracle
    oracle.build()
    orig = pkg.Graph.set_tuning

    def set_tuning(self, t):
        orig(self, t)
        oracle.set_sync_tuning(t)

    pkg.Graph.set_tuning = set_tuning
    yield
    pkg.Graph.set_tuning = orig
    oracle.set_sync_tuning(None)
