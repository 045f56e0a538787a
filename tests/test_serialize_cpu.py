"""Model files vs buffalo's `Serializable` (/root/reference/buffalo/algo/base.py:271-318).

tests/golden/model_ref.bin was written with the REFERENCE's own `Option` class and framing
(tests/golden/make_model_fixture.py, run where /root/reference is mounted): reading it and writing the same
content back must reproduce it byte for byte."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
GOLDEN = os.path.join(HERE, "golden", "model_ref.bin")


def test_reads_a_file_written_by_the_reference_classes():
    from buffalo_amd.serialize import Option
    from buffalo_amd.serialize import load_objects
    objs = dict(load_objects(GOLDEN))
    assert list(objs) == ["_idmanager", "opt", "Q", "Qb", "P"]
    assert isinstance(objs["opt"], Option) and objs["opt"].d == 4 and objs["opt"].data_opt.input.main == "main.mtx"
    assert objs["opt"]["validation"]["topk"] == 10 and isinstance(objs["opt"].validation, Option)
    idm = objs["_idmanager"]
    assert idm.itemid_mapped is True and idm.itemid_map == {"apple": 0, "pear": 1, "fig": 2} and idm.userids[4] == "u4"
    assert objs["P"].shape == (5, 4) and objs["Q"].dtype == np.float32 and objs["Qb"].shape == (3, 1)
    only = dict(load_objects(GOLDEN, data_fields=("Q",)))
    assert list(only) == ["Q"] and np.array_equal(only["Q"], objs["Q"])


def test_writes_the_same_bytes_as_the_reference(tmp_path):
    import make_model_fixture as mk
    from buffalo_amd.serialize import Option
    from buffalo_amd.serialize import dump_objects
    out = tmp_path / "ours.bin"
    dump_objects(str(out), mk.content(Option))
    assert out.read_bytes() == open(GOLDEN, "rb").read()
    assert Option.__module__ == "buffalo_amd.serialize" and "buffalo.misc._aux" not in sys.modules   # the alias does not leak


def test_algo_save_load_roundtrip_through_the_reference_format(tmp_path):
    import make_model_fixture as mk
    from buffalo_front.algo.base import Algo
    from buffalo_amd.serialize import Option

    class Model(Algo):
        def _get_data(self):
            return super()._get_data() + [("opt", self.opt), ("Q", self.Q), ("Qb", self.Qb), ("P", self.P)]
    src = Model()
    for name, obj in mk.content(Option):
        setattr(src, name, obj)
    path = str(tmp_path / "m.bin")
    src.save(path)
    assert open(path, "rb").read() == open(GOLDEN, "rb").read()
    dst = Model()
    dst.load(path, data_fields=("P", "_idmanager"))
    assert np.array_equal(dst.P, src.P) and dst._idmanager.userid_map == src._idmanager.userid_map and not hasattr(dst, "Q")
