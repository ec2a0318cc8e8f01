import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have(path):
    return os.path.exists(path)


@pytest.fixture(scope="session")
def reflibs():
    """The unmodified reference built by oracle/Makefile into oracle/_ref (travels to the GPU box)."""
    from mc33_capi import MC33Lib, ref_path
    if not _have(ref_path("f32")):
        pytest.skip("oracle/_ref not built (run python -c 'import __graft_entry__ as g; g.build()')")
    return {"f32": MC33Lib(ref_path("f32"), "f32"), "u16": MC33Lib(ref_path("u16"), "u16")}


@pytest.fixture(scope="session")
def oracles():
    from mc33_oracle import Oracle, oracle_path
    if not _have(oracle_path("f32")):
        pytest.skip("oracle library not built")
    return {"f32": Oracle("f32"), "u16": Oracle("u16")}


@pytest.fixture(scope="session")
def products():
    """The product libraries through the reference's own C API (GPU needed to call into them)."""
    from mc33_capi import MC33Lib, product_path
    assert _have(product_path("f32")), "libMC33_f32.so missing: the HIP extension must be built (no fallback)"
    return {"f32": MC33Lib(product_path("f32"), "f32"), "u16": MC33Lib(product_path("u16"), "u16")}
