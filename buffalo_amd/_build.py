"""In-tree build of libbuffalo_hip.so for gfx950 (explicit hipcc; nothing is JIT-cached elsewhere)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libbuffalo_hip.so")
# the same library with the shared-memory TEST transport of csrc/comm_test_transport.hpp compiled in (-DBFH_TEST_TRANSPORT: comm.hip alone differs).
# Only the N-ranks-on-one-GPU tests load it (BFH_LIBRARY=test, buffalo_amd/_lib.py); the product library refuses BFH_COMM_TRANSPORT=shm.
LIB_TEST = os.path.join(HERE, "libbuffalo_hip_test.so")
SOURCES = ["common.hip", "comm.hip", "sgd_base.hip", "bpr.hip", "warp.hip", "als.hip", "topk.hip", "ingest.hip", "sppmi.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wall",
         "-Wno-unused-function", "-Wno-unused-result"] + os.environ.get("BFH_EXTRA_FLAGS", "").split()   # extra -D switches for experiments (rebuild with --force)


FLAGS_STAMP = os.path.join(CSRC, ".build_flags")


def source_fingerprint():
    """16 hex digits over the kernel sources (csrc/*.hip, *.hpp + the public header) and the compile flags: what a profile taken by an
    earlier process must match to be quoted by a later one (bench.py `roofline.traffic`; stamped by scripts/pmc_summary.py)."""
    import hashlib
    h = hashlib.sha256()
    names = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp", ".h")))
    for f in names + [os.path.join("..", "..", "include", "buffalo_hip.h")]:
        path = os.path.join(CSRC, f)
        if os.path.exists(path):
            h.update(f.encode())
            with open(path, "rb") as fh:
                h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()[:16]


def _flags_changed():
    """BFH_EXTRA_FLAGS changes the compiled code (-DBFH_TEST_TRANSPORT, ...): objects built with other flags are stale."""
    try:
        with open(FLAGS_STAMP) as f:
            return f.read() != " ".join(FLAGS)
    except OSError:
        return bool(os.environ.get("BFH_EXTRA_FLAGS", "").split())      # no stamp = a default build


def _deps():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))] + \
           [os.path.join(HERE, "..", "include", "buffalo_hip.h")]


def _stale(out, srcs):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(s) > t for s in srcs if os.path.exists(s))


def build(force=False, verbose=False):
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    deps = _deps()
    force = force or (_flags_changed() and os.access(CSRC, os.W_OK))
    objs, jobs = [], []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + deps):
            jobs.append([HIPCC] + FLAGS + ["-c", src, "-o", obj])
    comm_test = os.path.join(CSRC, "comm_test.o")
    if force or _stale(comm_test, [os.path.join(CSRC, "comm.hip")] + deps):
        jobs.append([HIPCC] + FLAGS + ["-DBFH_TEST_TRANSPORT", "-c", os.path.join(CSRC, "comm.hip"), "-o", comm_test])
    objs_test = [comm_test if o.endswith(os.sep + "comm.o") else o for o in objs]

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), r.stderr[-8000:]))
        return r.stderr

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            for err in ex.map(run, jobs):
                if verbose and err.strip():
                    print(err, file=sys.stderr)
    if jobs:
        with open(FLAGS_STAMP, "w") as f:
            f.write(" ".join(FLAGS))
    if jobs or _stale(LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"])   # librccl is dlopen'ed (comm.hip)
    if jobs or _stale(LIB_TEST, objs_test):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_TEST] + objs_test + ["-ldl", "-lrt"])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
