"""`python bench.py --gpus 2 --workload {bpr,als}` by plain python on ONE GPU (BFH_DEVICE_OVERRIDE=0 pins both ranks to device 0, BFH_COMM_TRANSPORT=shm
selects libbuffalo_hip_test.so's shared-memory transport -- RCCL refuses two ranks on one device): the self-launch, the library's exchange with N = 2 and
the LINE the driver would record are exercised end to end (BASELINE configs[3]).  What the first SCALE run on real GPUs must show is held
here in form: `rccl_ranks` = what the LIVE communicator reports (bfh_comm_size: ncclCommCount over RCCL, the attached ranks over the test transport) equals
`n_gpus`, and `transport` names what carried the data.  There is no torch.distributed data path to fall back to: without the library's communicator the
run fails on every rank (bench.py Ctx.make_comm)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(workload, steps):
    env = dict(os.environ, BFH_DEVICE_OVERRIDE="0", BFH_COMM_TRANSPORT="shm")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", workload, "--steps", str(steps), "--warmup", "1",
                        "--no-cpu-baseline", "--no-extra"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    assert len(line) <= 4096
    return json.loads(line)


@pytest.mark.parametrize("workload,steps", [("bpr", 3), ("als", 2)])
def test_two_ranks_on_one_gpu_through_the_plain_python_launch(workload, steps):
    out = _run(workload, steps)
    assert out["n_gpus"] == 2 and out["steps"] == steps
    assert out["rccl_ranks"] == 2, out                      # the communicator that exists reports two ranks
    assert out["transport"] == "shm-test", out               # ... and says what it runs on (a real node: "rccl <version>")
    assert out["value"] > 0 and out["ms_per_step"] > 0
    assert "shm-test" in out["config"]["parallelism"]
    assert out["roofline"]["frac"] > 0
