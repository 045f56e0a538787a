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
    """Base of every option class: defaults, a shallow type check against them, and the temporary JSON file behind a dict."""

    def __init__(self, *args, **kwargs):
        pass

    @abc.abstractmethod
    def get_default_option(self) -> dict:
        """The complete option dict with its default values."""

    def is_valid_option(self, opt) -> bool:
        for key, default in self.get_default_option().items():
            if key not in opt:
                raise RuntimeError("{} not exists on Option".format(key))
            if not isinstance(opt.get(key), type(default)):
                raise RuntimeError("Invalid type for {}, {} expected. ".format(key, type(default)))
        return True

    def create_temporary_option_from_dict(self, opt) -> str:
        fd, path = tempfile.mkstemp(dir=opt.get("tmp_dir", "/tmp/"))
        with os.fdopen(fd, "w") as out:
            json.dump(opt, out)
        _temporary_files.append(path)
        return path


@atexit.register
def _remove_temporary_option_files():
    while _temporary_files:
        path = _temporary_files.pop()
        if os.path.isfile(path):
            os.remove(path)
