"""CPU-only checks of the drop-in boundary: the C-ABI library loads, exports every symbol the
header declares, and fails loudly (no silent fallback) when no GPU is present."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import has_gpu


@pytest.fixture(scope="module")
def built():
    from buffalo_amd import _build
    return _build.build()


def test_library_exports_every_header_symbol(built):
    from buffalo_amd import _lib
    L = C.CDLL(built)
    names = _lib.header_symbols()
    assert len(names) >= 55
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, "declared in include/buffalo_hip.h but not exported: %s" % missing
    # the ctypes table covers the header and nothing else
    assert sorted(_lib.SIGNATURES) == names


def test_version_and_error_string(built):
    from buffalo_amd import _lib
    L = _lib.lib()
    assert b"gfx950" in L.bfh_version()
    assert isinstance(L.bfh_last_error(None), bytes)
    assert L.bfh_device_count() >= 0


@pytest.mark.skipif(has_gpu(), reason="checks the no-GPU failure mode")
def test_no_gpu_fails_loudly(built):
    """There is no CPU fallback in the product path: object creation must raise."""
    from buffalo_amd.backend import CyALS, CyBPR, CyWARP
    from buffalo_amd._lib import BuffaloHipError
    for cls in (CyBPR, CyWARP, CyALS):
        with pytest.raises(BuffaloHipError):
            cls()


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under buffalo_amd/, bench.py's GPU leg or the C
    sources may reference it."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bad = []
    for base, _, files in os.walk(os.path.join(root, "buffalo_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                text = open(os.path.join(base, f)).read()
                if "import oracle" in text or "from oracle" in text or "libbuffalo_oracle" in text:
                    bad.append(os.path.join(base, f))
    assert not bad, bad


def test_binding_argument_checks(built):
    """Typed-buffer errors surface before the C ABI is reached (Cython raises ValueError)."""
    from buffalo_amd.backend import _arr
    with pytest.raises(ValueError):
        _arr(np.zeros((3, 3), np.float64), np.float32, 2, "P")
    with pytest.raises(ValueError):
        _arr(np.zeros(3, np.float32), np.float32, 2, "P")
    with pytest.raises(ValueError):
        _arr(np.zeros((4, 4), np.float32)[:, ::2], np.float32, 2, "P")


def test_the_product_library_does_not_contain_the_test_transport():
    """The shared-memory transport that lets N test processes share one GPU (csrc/comm_test_transport.hpp) is compiled with
    -DBFH_TEST_TRANSPORT into libbuffalo_hip_test.so only: the product library neither imports shm_open nor carries the transport's strings,
    and exports exactly the same C ABI."""
    import subprocess
    from buffalo_amd import _build
    _build.build()

    def nm(path, *flags):
        return subprocess.run(["nm", "-D"] + list(flags) + [path], capture_output=True, text=True, check=True).stdout
    prod, test = _build.LIB, _build.LIB_TEST
    assert os.path.exists(prod) and os.path.exists(test)
    assert "shm_open" not in nm(prod, "--undefined-only") and "shm_open" in nm(test, "--undefined-only")
    with open(prod, "rb") as f:
        blob = f.read()
    assert b"shm transport:" not in blob and b"TEST transport is not part of libbuffalo_hip.so" in blob

    def exported(path):
        return {ln.split()[-1] for ln in nm(path, "--defined-only").splitlines() if ln.split() and ln.split()[-1].startswith("bfh_")}
    assert exported(prod) == exported(test) and len(exported(prod)) > 50
