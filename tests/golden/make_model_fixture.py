"""Writes tests/golden/model_ref.bin with the REFERENCE's own classes: `Serializable.save`'s framing
(/root/reference/buffalo/algo/base.py:275-294) around objects built with /root/reference/buffalo/misc/_aux.py's
`Option`.  The buffalo package cannot be imported as a whole here (compiled extensions, h5py), so `_aux.py` is
loaded on its own under its real module name with `buffalo.misc.log` stubbed (it is only used by functions
that are not touched).  Run in the build container:  python tests/golden/make_model_fixture.py"""
import importlib.util
import os
import pickle
import struct
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/buffalo/misc/_aux.py"


def reference_option_class():
    for name in ("buffalo", "buffalo.misc"):
        sys.modules.setdefault(name, types.ModuleType(name))
    log = types.ModuleType("buffalo.misc.log")
    log.get_logger = lambda *a, **k: None
    sys.modules["buffalo.misc.log"] = log
    sys.modules["buffalo.misc"].log = log
    spec = importlib.util.spec_from_file_location("buffalo.misc._aux", REF)
    mod = importlib.util.module_from_spec(spec)
    sys.modules["buffalo.misc._aux"] = mod
    spec.loader.exec_module(mod)
    return mod.Option


def content(Option):
    """The objects BPRMF._get_data returns (bpr.py:254-260), small."""
    rng = np.random.default_rng(20240924)
    idm = Option({"userid": [], "userid_map": {}, "itemid": [], "itemid_map": {}, "userid_mapped": False, "itemid_mapped": False})
    idm.userids = ["u%d" % i for i in range(5)]
    idm.userid_map = {"u%d" % i: i for i in range(5)}
    idm.userid_mapped = True
    idm.itemids = ["apple", "pear", "fig"]
    idm.itemid_map = {"apple": 0, "pear": 1, "fig": 2}
    idm.itemid_mapped = True
    opt = Option({"d": 4, "num_iters": 3, "lr": 0.05, "optimizer": "sgd", "use_bias": True, "model_path": "",
                  "validation": {"topk": 10}, "data_opt": {"type": "matrix_market", "input": {"main": "main.mtx", "uid": None}}})
    P = rng.normal(size=(5, 4)).astype(np.float32)
    Q = rng.normal(size=(3, 4)).astype(np.float32)
    Qb = rng.normal(size=(3, 1)).astype(np.float32)
    return [("_idmanager", idm), ("opt", opt), ("Q", Q), ("Qb", Qb), ("P", P)]


if __name__ == "__main__":
    Option = reference_option_class()
    data = content(Option)
    with open(os.path.join(HERE, "model_ref.bin"), "wb") as fout:      # base.py:284-294
        fout.write(struct.pack("Q", len(data)))
        for name, obj in data:
            bname = bytes(name, encoding="utf-8")
            fout.write(struct.pack("Q", len(bname)))
            fout.write(bname)
            s = pickle.dumps(obj, protocol=4)
            fout.write(struct.pack("Q", len(s)))
            fout.write(s)
    print("wrote", os.path.join(HERE, "model_ref.bin"), os.path.getsize(os.path.join(HERE, "model_ref.bin")), "bytes")
