"""Option plumbing shared by the algo front: same behaviour as buffalo.misc.aux
(/root/reference/buffalo/misc/_aux.py:16-89): attribute dict, default/type validation and the
dict -> temporary JSON file hand-off the native backends are initialised from."""
import abc
import atexit
import json
import os
import tempfile

_temporary_files = []


class Option(dict):
    def __init__(self, *args, **kwargs):
        def read(fname):
            with open(fname) as fin:
                return json.load(fin)
        args = [arg if isinstance(arg, dict) else read(arg) for arg in args]
        super().__init__(*args, **kwargs)
        for src in list(args) + [kwargs]:
            for k, v in src.items():
                self[k] = Option(v) if isinstance(v, dict) else v

    def __getattr__(self, attr):
        return self.get(attr)

    def __setattr__(self, key, value):
        self.__setitem__(key, value)

    def __setitem__(self, key, value):
        super().__setitem__(key, value)
        self.__dict__.update({key: value})

    def __delattr__(self, item):
        self.__delitem__(item)

    def __delitem__(self, key):
        super().__delitem__(key)
        del self.__dict__[key]

    def __getstate__(self):
        return vars(self)

    def __setstate__(self, state):
        vars(self).update(state)


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
