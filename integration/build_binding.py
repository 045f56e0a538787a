"""Compile the Cython binding of INTEGRATION.md section 2 (integration/buffalo/algo/hip/_{bpr,als,warp}.pyx) in-tree against include/buffalo_hip.h and
buffalo_amd/libbuffalo_hip.so.  `python integration/build_binding.py` or __graft_entry__.build(); the built extension modules travel to the GPU box
with the snapshot (a relative rpath finds the library).  This is what buffalo's setup.py would do with one Extension per file (setup.py:148-187)."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
PKG = os.path.join(HERE, "buffalo", "algo", "hip")
NAMES = ("_bpr", "_als", "_warp")


def _stale():
    import importlib.machinery
    sfx = importlib.machinery.EXTENSION_SUFFIXES[0]
    hdr = os.path.join(ROOT, "include", "buffalo_hip.h")
    for n in NAMES:
        so, src = os.path.join(PKG, n + sfx), os.path.join(PKG, n + ".pyx")
        if not os.path.exists(so):
            return True
        t = os.path.getmtime(so)
        if any(os.path.getmtime(p) > t for p in (src, hdr, os.path.join(PKG, "_sgd_common.pxi"))):
            return True
    return False


def build(force=False):
    if not force and not _stale():
        return True
    import numpy as np
    from Cython.Build import cythonize
    from setuptools import Extension
    from setuptools.dist import Distribution
    libdir = os.path.join(ROOT, "buffalo_amd")
    exts = [Extension("buffalo.algo.hip." + n, [os.path.join(PKG, n + ".pyx")], include_dirs=[os.path.join(ROOT, "include"), np.get_include()],
                      libraries=["buffalo_hip"], library_dirs=[libdir], runtime_library_dirs=["$ORIGIN/../../../../buffalo_amd"],
                      define_macros=[("NPY_NO_DEPRECATED_API", "NPY_1_7_API_VERSION")], extra_compile_args=["-O2", "-w"]) for n in NAMES]
    cwd = os.getcwd()
    os.chdir(HERE)
    try:
        dist = Distribution({"name": "buffalo-hip-binding", "ext_modules": cythonize(exts, quiet=True, build_dir=os.path.join(HERE, "build")),
                             "script_args": ["build_ext", "--inplace", "--build-temp", os.path.join(HERE, "build"), "-q"]})
        dist.parse_command_line()
        dist.run_commands()
    finally:
        os.chdir(cwd)
    return True


def import_binding():
    """(CyBPR, CyALS, CyWARP) of the compiled binding; the HIP runtime is resolved the way buffalo_amd._lib does it."""
    sys.path.insert(0, ROOT)
    from buffalo_amd import _lib
    _lib._preload_shared_hip_runtime()
    if HERE not in sys.path:
        sys.path.insert(0, HERE)
    import importlib
    mods = [importlib.import_module("buffalo.algo.hip." + n) for n in NAMES]
    return mods[0].CyBPR, mods[1].CyALS, mods[2].CyWARP


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(import_binding())
