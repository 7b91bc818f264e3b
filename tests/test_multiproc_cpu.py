"""CPU, world_size 2, gloo: the N>1 path of bench.py -- scene sharding with no data-path collective, barrier and
max-over-ranks clock -- exercised with real processes (rendezvous on 127.0.0.1)."""
import importlib
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sh = importlib.import_module("one-2-3-45_amd.sharding")
    synth = importlib.import_module("one-2-3-45_amd.synth")
    r, w, _ = sh.init("gloo")
    mine = sh.scenes_for_rank(6, r, w)
    # each rank prepares ITS scenes only (different image seeds), nothing is exchanged
    chk = sum(float(synth.make_scene(4, image_seed=k, hw=(16, 16))["images"].sum()) for k in mine)
    sh.barrier()
    t = sh.max_over_ranks(0.25 + r)                     # pretend rank 1 is slower
    total = sh.sum_over_ranks(len(mine))
    q.put((r, mine, chk, t, total))
    sh.shutdown()


def test_two_rank_scene_sharding():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=120) for _ in range(world))
    [p.join(60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    assert res[0][1] == [0, 2, 4] and res[1][1] == [1, 3, 5]            # disjoint cover of all scenes
    assert res[0][2] != res[1][2]                                          # different data per rank
    assert res[0][3] == res[1][3] == 1.25                                  # the clock is the slowest rank's
    assert res[0][4] == res[1][4] == 6.0


def test_single_process_is_a_noop():
    sh = importlib.import_module("one-2-3-45_amd.sharding")
    assert sh.scenes_for_rank(5, 0, 1) == [0, 1, 2, 3, 4]
    assert sh.max_over_ranks(3.5) == 3.5 and sh.sum_over_ranks(2) == 2.0


def test_bench_bare_invocation_spawns_ranks():
    """`python bench.py --gpus 2` invoked BARE (no torchrun, no WORLD_SIZE): bench.py re-executes itself under
    torch.distributed.run with 2 ranks, every rank is dealt different scenes, rank 0 prints ONE json line with n_gpus = 2.
    (--dry-run --backend gloo: the rendezvous / deal / clock plumbing without GPUs.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["O2345_BENCH_EXTRA_FILE"] = ""
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--backend", "gloo", "--dry-run"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scenes_total"] == 6.0 and d["rank0_scenes"] == [0, 2, 4] and d["slowest_rank_s"] == 0.02
    # a launcher / flag mismatch is refused instead of silently measuring fewer GPUs
    env2 = dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--dry-run"],
                        capture_output=True, text=True, timeout=120, env=env2, cwd=root)
    assert r2.returncode != 0 and "--gpus 2" in (r2.stderr + r2.stdout)


def _bcast_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sh = importlib.import_module("one-2-3-45_amd.sharding")
    sh.init("gloo")
    state = None
    if rank == 0:                                       # only rank 0 "can reach the checkpoint"
        g = torch.Generator().manual_seed(5)
        state = {"sdf_network_lod0": {"sdf_layer.lin0.weight_v": torch.randn(128, 39, generator=g), "sdf_layer.lin0.bias": torch.randn(128, generator=g)},
                 "variance_network_lod0": {"variance": torch.tensor(0.37)}, "rendering_network_lod0": {"s": torch.tensor(0.2), "base_fc.0.weight": torch.randn(64, 193, generator=g)}}
    got = sh.broadcast_state_dicts(state)
    objs = sh.gather_objects({"rank": rank, "ms": 10.0 + rank})
    q.put((rank, {n: {k: (tuple(v.shape), str(v.dtype), float(v.double().sum())) for k, v in sd.items()} for n, sd in got.items()}, objs))
    sh.shutdown()


def test_two_rank_weight_broadcast_and_per_rank_gather():
    """The optional shared-backbone broadcast (one flat buffer, one collective) and the per-rank report of bench.py's N > 1 line, on gloo."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_bcast_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=120) for _ in range(world))
    [p.join(60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    assert res[0][1] == res[1][1]                                          # bit-identical weights on both ranks (same shapes, dtypes, sums)
    assert res[0][1]["variance_network_lod0"]["variance"][0] == () and abs(res[0][1]["variance_network_lod0"]["variance"][2] - 0.37) < 1e-7
    assert res[0][1]["rendering_network_lod0"]["base_fc.0.weight"][0] == (64, 193)
    assert res[0][2] == res[1][2] == [{"rank": 0, "ms": 10.0}, {"rank": 1, "ms": 11.0}]


def _split_worker(rank, world, port, q, n_rays):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sh = importlib.import_module("one-2-3-45_amd.sharding")
    sh.init("gloo")
    # the exchange step in: only rank 0 holds the scene's images
    imgs = torch.arange(2 * 3 * 5 * 7, dtype=torch.float32).reshape(2, 3, 5, 7) * 0.37 if rank == 0 else None
    imgs = sh.broadcast_tensor(imgs, src=0)
    lo, hi, per = sh.ray_block(n_rays, rank, world)
    r = torch.arange(lo, hi, dtype=torch.float32)
    # "rendered" per-ray outputs of this rank's block: values that identify the ray, in the types pipeline.render returns
    block = {"color": torch.stack([r, r * 0.5 + 1e-3, -r], 1), "depth": r * 1.0000001 + 0.25, "color_mask": (r.long() % 3 == 0).to(torch.uint8)}
    full = sh.gather_ray_blocks(block, n_rays, per)
    q.put((rank, (lo, hi, per), float(imgs.double().sum()), tuple(imgs.shape), {k: (tuple(v.shape), str(v.dtype), v.double().sum().item()) for k, v in full.items()},
           bool(torch.equal(full["depth"], torch.arange(n_rays, dtype=torch.float32) * 1.0000001 + 0.25)),
           bool(torch.equal(full["color_mask"], (torch.arange(n_rays) % 3 == 0).to(torch.uint8)))))
    sh.shutdown()


def test_two_rank_ray_split_of_one_image():
    """SURVEY 8e's optional intra-scene split: the images go out in one broadcast, contiguous ray blocks (multiples of a wavefront) are rendered per
    rank, one all-gather returns every ray to every rank in ray order, bit-exact -- on gloo, with a ray count that does not divide."""
    sh = importlib.import_module("one-2-3-45_amd.sharding")
    for n, w in ((1000, 2), (262144, 8), (65, 4), (1, 3), (4096, 1)):
        bl = [sh.ray_block(n, r, w) for r in range(w)]
        assert bl[0][0] == 0 and bl[-1][1] == n and all(bl[i][1] == bl[i + 1][0] for i in range(w - 1)) and all(b[2] % 64 == 0 and b[1] - b[0] <= b[2] for b in bl)
    world, port, n_rays = 2, _free_port(), 1000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_split_worker, args=(r, world, port, q, n_rays)) for r in range(world)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=120) for _ in range(world))
    [p.join(60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    assert res[0][1] == (0, 512, 512) and res[1][1] == (512, 1000, 512)
    assert res[0][2] == res[1][2] and res[0][3] == res[1][3] == (2, 3, 5, 7)                 # the same images on both ranks
    assert res[0][4] == res[1][4] and res[0][4]["color"][0] == (1000, 3) and res[0][4]["color_mask"][1] == "torch.uint8"
    assert all(r[5] and r[6] for r in res)                                                   # every ray, in order, bit-exact
    # without a process group the helpers are the identity
    blk = {"depth": torch.arange(5.0)}
    assert sh.broadcast_tensor(blk["depth"]) is blk["depth"] and sh.gather_ray_blocks(blk, 5, 64)["depth"] is blk["depth"]


def _bad_ckpt_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sh = importlib.import_module("one-2-3-45_amd.sharding")
    pipeline = importlib.import_module("one-2-3-45_amd.pipeline")
    sh.init("gloo")
    try:
        pipeline.SceneWeights.from_checkpoint("cpu", "/nonexistent/ckpt_000000.pth" if rank == 0 else None, broadcast=True)
        q.put((rank, "no error"))
    except Exception as e:                                  # noqa: BLE001
        q.put((rank, f"{type(e).__name__}: {e}"))
    sh.barrier()                                            # both ranks are still in step afterwards
    sh.shutdown()


def test_two_rank_checkpoint_failure_reaches_every_rank():
    """with broadcast=True only rank 0 reads the checkpoint; when that fails (missing file, a checkpoint that needs the full unpickler without
    the opt-in) the other ranks must not be left waiting in the broadcast -- every rank raises."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_bad_ckpt_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    res = dict(q.get(timeout=120) for _ in range(world))
    [p.join(60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    assert "does not load" in res[0] or "No such file" in res[0], res
    assert "rank 0 failed to produce the weights" in res[1] and "RuntimeError" in res[1], res
