"""The reference's OWN test files, unmodified, over the stand-ins this repository puts where its compiled parts are (TEST INFRASTRUCTURE).

    python tests/golden/run_reference_tests.py data       # tests/data/test_{mm,stream,prepro}.py: 19 tests
    python tests/golden/run_reference_tests.py parallel   # tests/parallel/test_base.py: test00, 01, 03, 04 (02 is a thread-scaling timing test)
    python tests/golden/run_reference_tests.py algo [test_als test_bpr test_warp test_eals]   # 52 algorithm tests over the oracle's classes
    python tests/golden/run_reference_tests.py algo-cfr   # test_cfr.py, 10 tests, with stock Stream.create()'s cleanup line repaired in-process

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


def ml100k_shaped(root, seed=0, groups=20, affinity=40.0, skew=1.1):
    """./ext/ml-100k/{main,uid,iid,stream} in the formats of the reference's tests/preprocess.py:12-60, with the SHAPE of MovieLens-100K
    (943 x 1682, 100,000 ratings 1..5) and planted structure instead of its content: 20 taste groups of users and items, in-group
    ratings likelier and higher, item popularity skewed; the three titles the tests ask most_similar about (base.py:129-139,
    test_base.py:31-34) sit in one group and share a fan base."""
    import numpy as np
    rng = np.random.default_rng(seed)
    U, I, G, NNZ = 943, 1682, groups, 100000
    trilogy = [49, 180, 171]
    ug, ig = rng.integers(0, G, U), rng.integers(0, G, I)
    ig[trilogy] = ig[49]
    pop = 1.0 / np.arange(1, I + 1) ** skew
    rng.shuffle(pop)
    pop[trilogy] = pop.max()
    pairs = set()
    fans = np.nonzero(ug == ig[49])[0]
    for u in fans:
        if rng.random() < 0.8:
            pairs.update((int(u), t) for t in trilogy)
    while len(pairs) < NNZ:
        u = rng.integers(0, U, 20000)
        w = np.where(ig[None, :] == ug[u][:, None], affinity, 1.0) * pop[None, :]
        i = (w.cumsum(1) / w.sum(1, keepdims=True) > rng.random((len(u), 1))).argmax(1)
        for a, b in zip(u.tolist(), i.tolist()):
            if len(pairs) < NNZ:
                pairs.add((a, b))
    pairs = sorted(pairs)
    d = os.path.join(root, "ext", "ml-100k")
    os.makedirs(d)
    names = ["%d.Movie_%d" % (i, i) for i in range(I)]
    names[49], names[180], names[171] = "49.Star_Wars_(1977)", "180.Return_of_the_Jedi_(1983)", "171.Empire_Strikes_Back,_The_(1980)"
    with open(os.path.join(d, "main"), "w") as f:
        f.write("%%MatrixMarket matrix coordinate integer general\n%\n%\n943 1682 100000\n")
        for u, i in pairs:
            v = int(np.clip(rng.normal(4.2 if ug[u] == ig[i] else 2.8, 0.9), 1, 5).round())
            f.write("%d %d %d\n" % (u + 1, i + 1, v))
    with open(os.path.join(d, "iid"), "w") as f:
        f.write("\n".join(names))
    with open(os.path.join(d, "uid"), "w") as f:
        f.write("".join("%d\n" % (u + 1) for u in range(U)))
    per_user = {}
    for u, i in pairs:
        per_user.setdefault(u, []).append(i)
    with open(os.path.join(d, "stream"), "w") as f:
        f.write("\n".join(" ".join(names[i] for i in rng.permutation(per_user.get(u, [0]))) for u in range(U)))


ALGO_TESTS = {
    # file -> the tests that need neither MovieLens-20M nor a GPU, nor compare wall-clock times (test10)
    "test_als": ["test00_get_default_option", "test01_is_valid_option", "test02_init_with_dict", "test03_init", "test04_train",
                 "test05_validation", "test05_1_validation_with_callback", "test06_topk", "test08_serialization",
                 "test09_compact_serialization", "test12_train_using_ialspp", "test13_train_using_ialspp_dim_256"],
    "test_bpr": ["test00_get_default_option", "test01_is_valid_option", "test02_init_with_dict", "test03_init", "test04_train",
                 "test05_validation", "test05_1_validation_with_callback", "test06_topk", "test08_serialization",
                 "test09_compact_serialization"],
    "test_warp": ["test00_get_default_option", "test01_is_valid_option", "test02_init_with_dict", "test03_init", "test04_train",
                  "test05_validation", "test05_1_validation_with_callback", "test06_topk", "test08_serialization",
                  "test09_compact_serialization"],
    "test_eals": ["test00_get_default_option", "test01_is_valid_option", "test02_init_with_dict", "test03_init", "test04_train",
                  "test05_validation", "test05_1_validation_with_callback", "test06_topk", "test08_serialization",
                  "test09_compact_serialization"],
    # test_cfr.py is not in this table: every test from test03 on builds a Stream with data.sppmi, and stock buffalo's Stream.create()
    # raises TypeError from its temporary-file cleanup in that case (stream.py:212, 316) -- mode `algo-cfr` runs it with that one line
    # repaired in-process (repair_stream_cleanup)
}


def repair_stream_cleanup():
    """Stock buffalo's Stream.create() puts the open FILE OBJECT of the pair-line file on its list of temporary paths when data.sppmi is
    set (stream.py:212) and the cleanup that ends create() hands it to os.path.isfile -> TypeError (stream.py:316, base.py:171): every
    test of test_cfr.py that builds its data stops there, with the reference's own compiled classes as much as with the oracle's.
    This drops the non-path entries before the cleanup runs -- the one-line repair -- so that the CFR tests can say something about
    the oracle's CFR class.  It changes the reference's behaviour in this process (not its source) and is used by `algo-cfr` only."""
    from buffalo.data.base import Data
    stock = Data.temp_file_clear

    def temp_file_clear(self):
        self.temp_file_list = [p for p in self.temp_file_list if isinstance(p, (str, bytes, os.PathLike))]
        stock(self)
    Data.temp_file_clear = temp_file_clear


def algo_tests(files=None, before=None):
    """The reference's own algorithm tests with the ORACLE's classes where its compiled CyALS / CyBPRMF / CyWARP / CyCFR / CyEALS
    stand, its fronts, data package, evaluation and serialization unmodified, on ML-100K-SHAPED synthetic data (MovieLens is not in
    this image): thresholds on NDCG@10 / MAP@10, the training callback cadence, top-k / most_similar by item name, model files."""
    import make_data_vectors as M
    from oracle import oracle
    M.install()
    for mod, name, cls in (("_als", "CyALS", oracle.OracleALS), ("_bpr", "CyBPRMF", oracle.OracleBPRMF), ("_warp", "CyWARP", oracle.OracleWARP),
                           ("_cfr", "CyCFR", oracle.OracleCFR), ("_eals", "CyEALS", oracle.OracleEALS)):
        setattr(sys.modules["buffalo.algo." + mod], name, cls)
    sys.modules["buffalo.parallel._core"].dot_topn = oracle.dot_topn
    sys.modules["buffalo.parallel._core"].quickselect = oracle.quickselect
    import buffalo
    from buffalo.algo.als import ALS
    from buffalo.algo.base import Algo
    from buffalo.algo.bpr import BPRMF
    from buffalo.algo.cfr import CFR
    from buffalo.algo.eals import EALS
    from buffalo.algo.options import ALSOption, BPRMFOption, CFROption, EALSOption, WARPOption
    from buffalo.algo.warp import WARP
    from buffalo.data.mm import MatrixMarketOptions
    from buffalo.data.stream import StreamOptions
    from buffalo.misc import aux, log
    for k, v in dict(ALS=ALS, ALSOption=ALSOption, BPRMF=BPRMF, BPRMFOption=BPRMFOption, WARP=WARP, WARPOption=WARPOption, CFR=CFR,
                     CFROption=CFROption, EALS=EALS, EALSOption=EALSOption, Algo=Algo, MatrixMarketOptions=MatrixMarketOptions,
                     StreamOptions=StreamOptions, aux=aux, set_log_level=log.set_log_level, inited_CUALS=False, inited_CUBPR=False).items():
        setattr(buffalo, k, v)
    if before:
        before()
    pkg = type(sys)("reference_tests_algo")
    pkg.__path__ = [os.path.join(REF, "tests", "algo")]
    sys.modules["reference_tests_algo"] = pkg
    suite = unittest.TestSuite()
    with tempfile.TemporaryDirectory() as d:
        os.chdir(d)
        ml100k_shaped(d)
        _load("tests/algo/base.py", "reference_tests_algo.base")
        for name, wanted in ALGO_TESTS.items():
            if files and name not in files:
                continue
            mod = _load("tests/algo/%s.py" % name, "reference_tests_algo." + name)
            case = [c for c in vars(mod).values() if isinstance(c, type) and issubclass(c, unittest.TestCase) and c.__module__ == mod.__name__][0]
            suite.addTests(case(n) for n in wanted)
        res = _run(suite, "algo")
        os.chdir(ROOT)
    return res


def dropin_probe():
    """INTEGRATION.md section 7's shortest drop-in, wired for real: `buffalo.algo.cuda._bpr.CyBPR` / `_als.CyALS` ARE
    `buffalo_amd.backend.CyBPR` / `CyALS`, and stock buffalo's unmodified fronts are constructed with accelerator = True.  On a GPU
    box that trains; in this container (no GPU) the product must refuse loudly at handle creation -- which shows that the reference's
    front reached libbuffalo_hip through its own import path and that nothing falls back to a CPU path."""
    import make_front_traces as G
    G.install_reference()
    from buffalo_amd import backend
    from buffalo_amd._lib import BuffaloHipError
    sys.modules["buffalo.algo.cuda._bpr"].CyBPR = backend.CyBPR
    sys.modules["buffalo.algo.cuda._als"].CyALS = backend.CyALS
    from buffalo.algo.als import ALS, inited_CUALS
    from buffalo.algo.bpr import BPRMF, inited_CUBPR
    from buffalo.algo.options import ALSOption, BPRMFOption
    assert inited_CUALS and inited_CUBPR
    outcome = {}
    for name, cls, opt_cls in (("ALS", ALS, ALSOption), ("BPRMF", BPRMF, BPRMFOption)):
        opt = opt_cls().get_default_option()
        opt.accelerator = True
        try:
            model = cls(opt)
            outcome[name] = "constructed over %s.%s" % (type(model.obj).__module__, type(model.obj).__name__)
        except BuffaloHipError as e:
            outcome[name] = "BuffaloHipError: %s" % e
    for k, v in outcome.items():
        print("dropin %s -> %s" % (k, v))
    return outcome


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "data"
    if mode == "dropin":
        dropin_probe()
        sys.exit(0)
    if mode == "algo":
        r = algo_tests(sys.argv[2:] or None)
        sys.exit(0 if r.wasSuccessful() else 1)
    if mode == "algo-cfr":
        ALGO_TESTS["test_cfr"] = ["test00_get_default_option", "test01_is_valid_option", "test02_init_with_dict", "test03_init", "test04_train",
                                  "test05_validation", "test05_1_validation_with_callback", "test06_topk", "test08_serialization",
                                  "test09_compact_serialization"]
        r = algo_tests(["test_cfr"], before=repair_stream_cleanup)
        sys.exit(0 if r.wasSuccessful() else 1)
    r = {"data": data_tests, "parallel": parallel_tests}[mode]()
    sys.exit(0 if r.wasSuccessful() else 1)
