"""Option plumbing shared by the algo front (cf. /root/reference/buffalo/misc/_aux.py:16-89): the attribute dict is the
product's own `buffalo_amd.serialize.Option`; here are the default / type validation and the dict -> temporary JSON file
hand-off the native backends are initialised from."""
import abc
import atexit
import json
import os
import tempfile

_temporary_files = []


from buffalo_amd.serialize import Option   # noqa: E402,F401 -- the product's attribute dict (it pickles like buffalo.misc._aux.Option)


def load_option(src):
    """`Option(path)` of the reference reads the JSON file behind the path (_aux.py:17-24); a dict is taken as it is."""
    if isinstance(src, dict):
        return Option(src)
    with open(src) as fin:
        return Option(json.load(fin))


class InputOptions(abc.ABC):
    def __init__(self, *args, **kwargs):
        pass

    @abc.abstractmethod
    def get_default_option(self) -> dict:
        pass

    def is_valid_option(self, opt) -> bool:
        default_opt = self.get_default_option()
        for key in default_opt:
            if key not in opt:
                raise RuntimeError("{} not exists on Option".format(key))
            if not isinstance(opt.get(key), type(default_opt[key])):
                raise RuntimeError("Invalid type for {}, {} expected. ".format(key, type(default_opt[key])))
        return True

    def create_temporary_option_from_dict(self, opt) -> str:
        tmp = tempfile.NamedTemporaryFile(mode="w", dir=opt.get("tmp_dir", "/tmp/"), delete=False)
        tmp.write(json.dumps(opt))
        tmp.close()
        _temporary_files.append(tmp.name)
        return tmp.name


@atexit.register
def _cleanup_temporary_files():
    for path in _temporary_files:
        if os.path.isfile(path):
            os.remove(path)
