"""GPU: the N > 1 path of bench.py end to end with real GPU work in every rank.  The GPU box has ONE device, so the two ranks share it
(--share-gpu, gloo for the clock: RCCL refuses two ranks on one device): a FUNCTIONAL test of the multi-process path -- bare invocation,
re-execution under torch.distributed.run, a different scene per rank and step, the c3 deal, barrier / max-over-ranks clock, one JSON line
with n_gpus = 2 -- not a scaling measurement."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_on_the_gpu_bare_invocation():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu", "--steps", "2", "--warmup", "1",
                        "--no-cpu", "--vol", "64", "--ray-scale", "1", "--mesh-res", "96"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["shared_gpu_functional_run"] is True and d["steps"] == 2
    assert d["value"] > 0 and d["config"]["parallelism"] == "scenes x2" and d["config"]["rays"] == 65536
    assert d["c3"]["scenes"] == 32 and "16 per GPU on 2 GPU(s)" in d["c3"]["workload"] and d["c3"]["scenes_per_s"] > 0
    # whole-job value = both ranks' rays over the slowest rank's time
    assert abs(d["value"] - 2 * 65536 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
