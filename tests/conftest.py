import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


_launcher = None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Tests that start rank processes do it through a helper started NOW, before anything in this process can have
    # initialised the GPU (tests/launcher.py): a GPU-initialised process must not start other programs on this pool.
    global _launcher
    if os.path.exists("/dev/kfd") and _launcher is None:
        from launcher import Launcher
        _launcher = Launcher()


def pytest_unconfigure(config):
    global _launcher
    if _launcher is not None:
        _launcher.close()
        _launcher = None


@pytest.fixture(scope="session")
def launcher():
    if _launcher is None:
        pytest.skip("no GPU device node: the rank launcher was not started")
    return _launcher


DTYPES = ("f32", "u16", "u8", "u32", "f64")  # one library per GRD_data_type, like the reference's compile-time variants


def _have(path):
    return os.path.exists(path)


@pytest.fixture(scope="session")
def reflibs():
    """The unmodified reference built by oracle/Makefile into oracle/_ref (travels to the GPU box)."""
    from mc33_capi import MC33Lib, ref_path
    if not _have(ref_path("f32")):
        pytest.skip("oracle/_ref not built (run python -c 'import __graft_entry__ as g; g.build()')")
    return {d: MC33Lib(ref_path(d), d) for d in DTYPES if _have(ref_path(d))}


@pytest.fixture(scope="session")
def oracles():
    from mc33_oracle import Oracle, oracle_path
    if not _have(oracle_path("f32")):
        pytest.skip("oracle library not built")
    return {d: Oracle(d) for d in DTYPES if _have(oracle_path(d))}


@pytest.fixture(scope="session")
def products():
    """The product libraries through the reference's own C API (GPU needed to call into them)."""
    from mc33_capi import MC33Lib, product_path
    assert _have(product_path("f32")), "libMC33_f32.so missing: the HIP extension must be built (no fallback)"
    for d in DTYPES:
        assert _have(product_path(d)), "libMC33_%s.so missing: the HIP extension must be built (no fallback)" % d
    return {d: MC33Lib(product_path(d), d) for d in DTYPES}
