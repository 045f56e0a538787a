"""The reference's OWN test files, unmodified, over the stand-ins this repository puts where its compiled parts are (TEST INFRASTRUCTURE).

    python tests/golden/run_reference_tests.py data       # tests/data/test_{mm,stream,prepro}.py: 19 tests
    python tests/golden/run_reference_tests.py parallel   # tests/parallel/test_base.py: test00, 01, 03, 04 (02 is a thread-scaling timing test)

`data`: buffalo/data/*.py unmodified over the in-memory h5py and the reference's own compiled fileio.hpp (make_data_vectors.install).
`parallel`: buffalo/parallel/base.py unmodified with `buffalo.parallel._core.dot_topn` bound to the ORACLE's restatement of
_core.hpp:69-142 -- the known answers these tests hold (numpy argsort) are what pins the oracle's top-k; tests/test_oracle_pins.py
restates the same cases so that they also run where /root/reference is absent.
Needs /root/reference; tests/test_data_loaders_ref.py and tests/test_oracle_pins.py call this in a subprocess.
"""
import importlib.util
import os
import sys
import tempfile
import unittest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)
REF = "/root/reference"


def _load(rel, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def _run(suite, what):
    res = unittest.TextTestRunner(verbosity=1).run(suite)
    print("reference %s tests: ran %d, failures %d, errors %d" % (what, res.testsRun, len(res.failures), len(res.errors)))
    return res


def data_tests():
    """Known answers about header counts, iteration order, id maps, value preprocessing."""
    import make_data_vectors as M
    M.install()
    import buffalo
    from buffalo.data.mm import MatrixMarket, MatrixMarketOptions
    from buffalo.data.stream import Stream, StreamOptions
    from buffalo.misc import aux, log
    for k, v in dict(MatrixMarket=MatrixMarket, MatrixMarketOptions=MatrixMarketOptions, Stream=Stream, StreamOptions=StreamOptions,
                     aux=aux, set_log_level=log.set_log_level).items():
        setattr(buffalo, k, v)       # what `from buffalo import ...` of the tests finds in buffalo/__init__.py
    suite = unittest.TestSuite()
    with tempfile.TemporaryDirectory() as d:
        os.chdir(d)                  # the tests write ./mm.h5py, ./stream.h5py
        for name in ("test_mm", "test_stream", "test_prepro"):
            suite.addTests(unittest.defaultTestLoader.loadTestsFromModule(_load("tests/data/%s.py" % name, "reference_" + name)))
        res = _run(suite, "data")
        os.chdir(ROOT)
    return res


def parallel_tests():
    """dot_topn against numpy argsort: most_similar (the query itself excluded), a pool, top-k with separate query factors."""
    import make_front_traces as G
    from oracle import oracle
    G.install_reference()
    G.Recorder.trace = []            # ALS() of the tests is built over the recording stand-in for CyALS; only dot_topn computes
    sys.modules["buffalo.parallel._core"].dot_topn = oracle.dot_topn
    sys.modules["buffalo.parallel._core"].quickselect = oracle.quickselect
    pkg = type(sys)("reference_tests_parallel")
    pkg.__path__ = [os.path.join(REF, "tests", "parallel")]
    sys.modules["reference_tests_parallel"] = pkg
    _load("tests/parallel/base.py", "reference_tests_parallel.base")
    mod = _load("tests/parallel/test_base.py", "reference_tests_parallel.test_base")     # `from .base import ...` resolves inside the package
    wanted = ["test00_init", "test01_most_similar", "test03_pool", "test04_topk"]
    suite = unittest.TestSuite([mod.TestParallelBase(n) for n in wanted])
    with tempfile.TemporaryDirectory() as d:
        os.chdir(d)
        res = _run(suite, "parallel")
        os.chdir(ROOT)
    return res


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "data"
    r = {"data": data_tests, "parallel": parallel_tests}[mode]()
    sys.exit(0 if r.wasSuccessful() else 1)
