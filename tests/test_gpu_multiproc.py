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


def _split_worker(rank, world, port, q):
    import importlib
    import numpy as np
    import torch
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    pkg = importlib.import_module("one-2-3-45_amd")
    sh, pipeline = importlib.import_module("one-2-3-45_amd.sharding"), importlib.import_module("one-2-3-45_amd.pipeline")
    sh.init("gloo")                                     # both ranks on the box's one GPU: RCCL refuses that, gloo carries the two collectives
    dev = torch.device("cuda:0")
    D, V = 64, 8
    sc = pkg.synth.make_scene(V, image_seed=11)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    ro, rd = pkg.synth.gen_rays(sc["query_intrinsic"], sc["query_c2w"], 256, 256)
    n_rays = ro.shape[0] - 37                           # a ray count that is not a multiple of anything
    ro, rd = T(ro[:n_rays]), T(rd[:n_rays])
    proj, cam_pos = pipeline.camera_terms(T(sc["intrinsics"]), T(sc["w2cs"]))
    wt = pipeline.SceneWeights(dev, seed=0)
    near, far, qcam = float(sc["query_near_far"][0]), float(sc["query_near_far"][1]), T(sc["query_c2w"][:3, 3].copy())
    imgs = T(sc["images"]) if rank == 0 else None       # only rank 0 holds the scene
    full, vol = pipeline.render_scene_split(wt, imgs, T(sc["affine_mats"]), sc["partial_vol_origin"], D, 2.0 / (D - 1), proj, cam_pos, ro, rd, near, far, qcam)
    one = pipeline.render(wt, vol, proj, cam_pos, ro, rd, near, far, qcam)                  # the one-GPU call on all rays, same volume
    same = {k: bool(torch.equal(full[k], one[k])) for k in full}
    lo, hi, per = sh.ray_block(n_rays, rank, world)
    q.put((rank, n_rays, (lo, hi), same, float(full["weights_sum"].max()), float(vol["vol_cl"].double().sum())))
    sh.shutdown()


def test_two_ranks_render_one_image_split_by_rays():
    """SURVEY 8e's optional intra-scene split on real kernels: rank 0 holds the images, one broadcast, both ranks build the volume (the same bits) and
    render half of the rays each, one all-gather -- every ray equals the one-call render bit for bit.  (Functional: the two ranks share the one GPU.)"""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_split_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    import queue
    import time
    res, t0 = [], time.time()
    while len(res) < 2 and time.time() - t0 < 600:
        try:
            res.append(q.get(timeout=2))
        except queue.Empty:
            if any(p.exitcode not in (None, 0) for p in ps):          # a rank died: do not wait for its result
                break
    [p.join(30) for p in ps]
    [p.kill() for p in ps if p.is_alive()]
    assert len(res) == 2 and all(p.exitcode == 0 for p in ps), [p.exitcode for p in ps]
    res.sort()
    n = res[0][1]
    assert res[0][2][0] == 0 and res[0][2][1] == res[1][2][0] and res[1][2][1] == n and res[0][2][1] % 64 == 0
    assert all(all(r[3].values()) for r in res), res                         # colour, depth, weights_sum, colour mask of every ray
    assert res[0][4] > 0.5                                                   # the image shows a surface
    assert res[0][5] == res[1][5]                                            # both ranks built the same volume
