"""GPU: the N > 1 path of bench.py end to end with real GPU work in every rank.  The GPU box has ONE device, so the two ranks share it
(--share-gpu, gloo for the clock: RCCL refuses two ranks on one device): a FUNCTIONAL test of the multi-process path -- bare invocation,
re-execution under torch.distributed.run, a different scene per rank and step, the c3 deal, barrier / max-over-ranks clock, the contract line
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
    env["O2345_BENCH_EXTRA_FILE"] = ""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu", "--steps", "2", "--warmup", "1",
                        "--no-cpu", "--vol", "64", "--ray-scale", "1", "--mesh-res", "96"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 2, r.stdout[-2000:]                 # the full record, then the compact contract line LAST (the one the driver parses)
    full, d = json.loads(lines[0]), json.loads(lines[1])
    assert len(lines[1]) <= 4096 and r.stdout.rstrip().endswith(lines[1])
    assert d["n_gpus"] == 2 and d["shared_gpu_functional_run"] is True and d["steps"] == 2
    assert d["value"] > 0 and d["config"]["parallelism"] == "scenes x2" and d["config"]["rays"] == 65536
    assert d["value"] == full["value"] and d["ms_per_step"] == full["ms_per_step"] and len(d["per_rank_ms"]) == 2 and d["roofline"]["frac"] > 0
    d = dict(d, c3=full["c3"])
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


def test_eight_ranks_on_the_one_gpu_functional():
    """the shape the driver's 8-GPU run has -- `bench.py --gpus 8` bare, re-executed under torch.distributed.run, eight ranks, rendezvous on
    127.0.0.1, scene deal 0..7 per step, BASELINE config 3's 32 scenes dealt 4 per rank, a per-rank report with eight entries -- with all eight ranks
    sharing the box's single device (gloo for the clock).  FUNCTIONAL ONLY: the printed rate is eight processes time-slicing one GPU, not a scaling number
    (the line says so: shared_gpu_functional_run)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["O2345_BENCH_EXTRA_FILE"] = ""
    env["OMP_NUM_THREADS"] = "4"                         # eight ranks on one host: no thread oversubscription
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--backend", "gloo", "--share-gpu", "--steps", "1", "--warmup", "1",
                        "--no-cpu", "--vol", "64", "--ray-scale", "1", "--mesh-res", "64"],
                       capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 2 and len(lines[1]) <= 4096, r.stdout[-2000:]            # the full record, then the compact contract line (the last thing on stdout)
    d, line = json.loads(lines[0]), json.loads(lines[1])
    assert line["n_gpus"] == 8 and line["value"] == d["value"] and len(line["per_rank_ms"]) == 8 and line["rccl_ranks"] == 8 and line["backend"] == "gloo"
    assert d["n_gpus"] == 8 and d["shared_gpu_functional_run"] is True and d["steps"] == 1 and d["scaling"] == "weak"
    assert d["config"]["parallelism"] == "scenes x8" and d["rccl_ranks"] == 8 and d["backend"] == "gloo"
    assert len(d["per_rank"]) == 8 and sorted(p["rank"] for p in d["per_rank"]) == list(range(8))
    assert all(p["ms_per_step_own_clock"] > 0 for p in d["per_rank"])
    assert abs(d["value"] - 8 * 65536 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    c3 = d["c3"]
    assert c3["scenes"] == 32 and "4 per GPU on 8 GPU(s)" in c3["workload"] and len(c3["per_rank"]) == 8
    assert all(p["scenes"] == 4 for p in c3["per_rank"])


def _nccl_worker(q):
    import importlib
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sh = importlib.import_module("one-2-3-45_amd.sharding")
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1")
    rank, world, local = sh.init("nccl", force=True)               # the device_id= branch of sharding.init: RCCL communicator bound to cuda:0
    dev = torch.device("cuda", local)
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    dist.barrier(device_ids=[local])
    t = torch.tensor([3.5], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                      # the clock reduction of bench.py (sharding.max_over_ranks), on the device
    b = torch.arange(1024, dtype=torch.float32, device=dev)
    dist.broadcast(b, src=0)                                      # the optional weight broadcast's collective
    out = [torch.empty_like(b)]
    dist.all_gather(out, b)                                       # the intra-scene mode's result collective
    torch.cuda.synchronize(dev)
    q.put((float(t.item()), float(out[0].sum().item()), sh._reduce_device(dev) == dev))
    sh.shutdown()


def test_one_rank_rccl_init_barrier_and_collectives():
    """The `nccl` (= RCCL) branch of sharding.init -- init_process_group(device_id=cuda:local) -- and the three collectives the package ever issues
    (all_reduce for the clock, broadcast for the optional shared weights, all_gather for the intra-scene mode) executed once on RCCL: a one-rank communicator
    is all a one-GPU box can host (RCCL refuses two ranks on one device); it proves the code path loads, binds the device and runs."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_PORT"] = str(port)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_worker, args=(q,))
    p.start()
    import queue
    try:
        res = q.get(timeout=300)
    except queue.Empty:
        res = None
    p.join(30)
    if p.is_alive():
        p.kill()
    assert res is not None and p.exitcode == 0, p.exitcode
    assert res[0] == 3.5 and res[1] == float(sum(range(1024))) and res[2] is True
