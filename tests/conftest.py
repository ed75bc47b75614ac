"""Shared fixtures.  GPU tests are marked `gpu` and call the HIP engine through the C ABI;
everything else runs on CPU (oracle, host logic, library load/export checks)."""
import importlib
import importlib.util
import inspect
import os
import re
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


# The product library reads six tuning variables (include/pprhip.h, "Environment").  Every other PPRHIP_* switch - fault
# injection, layout and driver variants, measurement switches - exists in libpprhip_hooks.so only (the same sources built
# with -DPPRHIP_TEST_HOOKS).  A test whose body names such a switch (or that is marked `hooks`) gets the package bound to
# that library; every other test runs the product, libpprhip.so.  Both live in this process side by side.
PRODUCT_ENV = {"PPRHIP_HOST_THREADS", "PPRHIP_COMM_TIMEOUT_S", "PPRHIP_RCCL_LIB", "PPRHIP_BATCH_WORKSPACES",
               "PPRHIP_BATCH_THREADS", "PPRHIP_SHARD_CUT", "PPRHIP_LIB_PATH", "PPRHIP_FUZZ_SEEDS"}
HOOKS_LIB = os.path.join(ROOT, PKG_NAME, "libpprhip_hooks.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "hooks: runs on libpprhip_hooks.so (test switches compiled in)")


def load_pkg():
    return importlib.import_module(PKG_NAME)


_hooks_mod = None


def load_hooks_pkg():
    """The package a second time, bound to libpprhip_hooks.so (a make asan-test run points both at its own build)."""
    global _hooks_mod
    if _hooks_mod is None:
        if os.environ.get("PPRHIP_LIB_PATH"):
            _hooks_mod = load_pkg()
            return _hooks_mod
        pkg_dir = os.path.join(ROOT, PKG_NAME)
        spec = importlib.util.spec_from_file_location("pprhip_hooks_pkg", os.path.join(pkg_dir, "__init__.py"),
                                                      submodule_search_locations=[pkg_dir])
        mod = importlib.util.module_from_spec(spec)
        sys.modules["pprhip_hooks_pkg"] = mod
        os.environ["PPRHIP_LIB_PATH"] = HOOKS_LIB
        try:
            spec.loader.exec_module(mod)
        finally:
            del os.environ["PPRHIP_LIB_PATH"]
        _patch_set_tuning(mod)
        _hooks_mod = mod
    return _hooks_mod


def wants_hooks(node):
    if node.get_closest_marker("hooks") is not None:
        return True
    fn = getattr(node, "function", None)
    try:
        src = inspect.getsource(fn) if fn is not None else ""
    except (OSError, TypeError):
        src = ""
    names = set(re.findall(r"PPRHIP_[A-Z][A-Z0-9_]*[A-Z0-9]", src))
    names = {x for x in names if not x.startswith(("PPRHIP_OK", "PPRHIP_ERR", "PPRHIP_RELEASE", "PPRHIP_KERNEL_", "PPRHIP_LIFT_"))}
    return bool(names - PRODUCT_ENV)


@pytest.fixture(scope="session")
def pkg_product():
    return load_pkg()


@pytest.fixture
def pkg(request, pkg_product):
    return load_hooks_pkg() if wants_hooks(request.node) else pkg_product


@pytest.fixture(scope="module")
def dev_cache():
    """Device graphs a module's tests share, one per (name, library): a test on the hooks library gets handles of that
    library, never the product's (its switches would not reach them)."""
    cache = {}
    yield cache
    for v in cache.values():
        for g in (v if isinstance(v, list) else [v]):
            g.close()


def shared_graph(cache, pkg, name, make):
    key = (name, pkg.__name__)
    if key not in cache:
        cache[key] = make()
    return cache[key]


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
def got(pkg_product):
    return pkg_product.HostCsr.from_neo4j_csv(GOT_NODES, GOT_RELS)


@pytest.fixture(scope="session")
def toy_graphs(pkg_product):
    pkg = pkg_product
    """Small graphs with closed-form answers / awkward structure."""
    g = {}
    g["two_node"] = edges_to_host(pkg, 2, [(0, 1)])                      # Dissertation p.13-14
    g["cycle5"] = edges_to_host(pkg, 5, [(i, (i + 1) % 5) for i in range(5)])
    g["star_dead_leaves"] = edges_to_host(pkg, 6, [(0, i) for i in range(1, 6)])
    g["isolated_mix"] = edges_to_host(pkg, 6, [(0, 1), (1, 2), (2, 0), (2, 3), (0, 0), (1, 2)])  # self loop, multi-edge, 4/5 isolated
    g["line"] = edges_to_host(pkg, 8, [(i, i + 1) for i in range(7)])
    return g


@pytest.fixture(scope="session")
def rmat12(pkg_product):
    return pkg_product.HostCsr.rmat(12, 16, seed=1)


@pytest.fixture(scope="session")
def rmat15(pkg_product):
    return pkg_product.HostCsr.rmat(15, 16, seed=1)


def _patch_set_tuning(mod):
    from oracle import oracle
    orig = mod.Graph.set_tuning

    def set_tuning(self, t):
        orig(self, t)
        oracle.set_sync_tuning(t)

    mod.Graph.set_tuning = set_tuning
    return orig


@pytest.fixture(autouse=True, scope="session")
def _twin_follows_engine_tuning(pkg_product):
    """Since the Gauss-Seidel sweeps the level shapes (dense_frac, gs_blocks, gs_frac) change the push schedule, so
    the twin's entry points that take no tuning argument (forward_push, topk_push, fora_topk) must run with the
    tuning the engine was given: every Graph.set_tuning in a test also sets the twin's (load_hooks_pkg does the same
    for the package bound to the hooks library)."""
    from oracle import oracle
    oracle.build()
    orig = _patch_set_tuning(pkg_product)
    yield
    pkg_product.Graph.set_tuning = orig
    oracle.set_sync_tuning(None)
