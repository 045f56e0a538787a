"""The COMPILED drop-in binding: integration/buffalo/algo/hip/_{bpr,als,warp}.pyx -- what a buffalo maintainer adds
beside buffalo/algo/cuda/_bpr.pyx / _als.pyx -- goes through Cython against include/buffalo_hip.h, links libbuffalo_hip.so, imports, and exposes the
method surface stock buffalo's fronts call (tests/golden/front_traces.json, recorded from the reference's own Python).  No GPU: nothing computes here;
tests/test_front_gpu.py::test_train_and_validate_through_the_compiled_binding runs a front through it on the device."""
import inspect
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "integration"))
sys.path.insert(0, os.path.join(HERE, "golden"))

pytest.importorskip("Cython")
import build_binding  # noqa: E402


@pytest.fixture(scope="module")
def binding():
    try:
        build_binding.build()
    except Exception as e:  # noqa: BLE001 -- no setuptools / numpy headers / host compiler on this box: the binding is optional (__graft_entry__.build)
        pytest.skip("the Cython binding cannot be built here: %s: %s" % (type(e).__name__, str(e)[:200]))
    return dict(zip(("bpr", "als", "warp"), build_binding.import_binding()))


def test_the_binding_compiles_links_and_imports(binding):
    for algo, cls in binding.items():
        assert cls.__module__ == "buffalo.algo.hip._" + algo
        so = sys.modules[cls.__module__].__file__
        assert so.endswith(".so") and os.path.dirname(so) == os.path.join(ROOT, "integration", "buffalo", "algo", "hip")
    # it binds the SAME library the ctypes mirror loads (one product .so; relative runpath, so the pair travels together)
    with open("/proc/self/maps") as f:
        libs = {line.split()[-1] for line in f if "libbuffalo_hip.so" in line}
    assert libs == {os.path.join(ROOT, "buffalo_amd", "libbuffalo_hip.so")}, libs


def test_method_surface_is_what_stock_buffalo_calls(binding):
    """Every call the reference's fronts make on their accelerator classes (all recorded traces) binds to a method of the compiled class,
    with the recorded number of positional arguments; the class surface equals the reference's cuda/_bpr.pyx:27-80 / _als.pyx:25-67."""
    import make_front_traces as G
    golden = json.load(open(os.path.join(HERE, "golden", "front_traces.json")))
    seen = set()
    for name, (algo, *_rest) in list(G.CASES.items()) + list(G.MORE_CASES.items()):
        if algo not in binding:
            continue
        cls = binding[algo]
        for c in golden[name]["trace"]:
            fn = getattr(cls, c["call"], None)
            assert fn is not None, "%s has no method %s" % (cls.__name__, c["call"])
            inspect.signature(fn).bind(None, *c["args"])
            seen.add((algo, c["call"]))
    assert {("bpr", "add_jobs"), ("bpr", "update_parameters"), ("als", "partial_update"), ("als", "precompute")} <= seen
    ref_bpr = {"init", "initialize_model", "set_placeholder", "set_cumulative_table", "get_vdim", "synchronize", "update_parameters",
               "wait_until_done", "add_jobs", "compute_loss"}
    ref_als = {"init", "initialize_model", "set_placeholder", "precompute", "get_vdim", "partial_update"}
    pub = lambda cls: {n for n in dir(cls) if not n.startswith("_")}   # noqa: E731
    assert ref_bpr <= pub(binding["bpr"]) and ref_bpr <= pub(binding["warp"]) and ref_als <= pub(binding["als"])
    assert pub(binding["bpr"]) - ref_bpr == {"set_mode"} and pub(binding["als"]) - ref_als == {"set_mode"}     # the one documented extension


def test_without_a_device_the_constructor_raises(binding):
    """No GPU in the build container: `bfh_*_create` refuses (the product has no CPU fallback) and the binding turns that into an exception."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a device is present")
    for cls in binding.values():
        with pytest.raises(RuntimeError, match="no HIP device"):
            cls()


@pytest.mark.gpu
def test_errors_and_buffer_checks_behave_like_the_reference_binding(binding, tmp_path):
    """`init` -> False on an unreadable option file (bpr.cu:245); typed buffer arguments raise ValueError on a dtype / ndim mismatch exactly
    like the reference's `np.ndarray[np.float32_t, ndim=2]` arguments do; a failing call raises (the reference: `except +`), nothing crashes."""
    for algo, cls in binding.items():
        obj = cls()
        assert obj.init(str(tmp_path / "missing.json").encode()) is False
        with pytest.raises(ValueError):
            if algo == "als":
                obj.initialize_model(np.zeros((4, 32), np.float64), np.zeros((4, 32), np.float32))
            else:
                obj.initialize_model(np.zeros((4, 32), np.float64), np.zeros((4, 32), np.float32), np.zeros((4, 1), np.float32), 10, True)
        with pytest.raises(ValueError):
            obj.set_placeholder(*([np.zeros(4, np.int32)] * (2 if algo == "als" else 1)), 16)
        with pytest.raises(RuntimeError):            # before init: an error code of the C ABI -> exception
            if algo == "als":
                obj.precompute(0)
            else:
                obj.update_parameters()
        del obj
