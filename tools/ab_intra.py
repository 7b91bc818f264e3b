"""One scene per step, but the scene's own mesh extraction (depends on the volume only) on a second HIP stream next to its render: wall clock per scene against the
sequential order, results compared bitwise."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]]
import bench
pipeline = bench.pipeline
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
wt = pipeline.SceneWeights(dev, seed=0)
wt2 = pipeline.SceneWeights(dev, seed=0)          # the side stream's weights object (own packed-weight caches / grid tables)
wt.grid_tables(256); wt2.grid_tables(256)
inp = bench.make_inputs(dev, 8, 0, 2)
K = 12
imgs = [torch.from_numpy(bench.scene_images(8, 100 + k)).to(dev) for k in range(K)]
side = torch.cuda.Stream(device=dev)
n = inp["rays_o"].shape[0]


def step_seq(im):
    vol = pipeline.build_volume(wt, im, inp["aff"], inp["origin"], 128, 2.0 / 127)
    out = pipeline.render(wt, vol, inp["proj"], inp["cam_pos"], inp["rays_o"], inp["rays_d"], inp["near"], inp["far"], inp["qcam"])
    mesh = pipeline.extract_mesh(wt, vol, inp["proj"], inp["cam_pos"], 256)
    return vol, out, mesh


def step_overlap(im, mesh_first):
    main = torch.cuda.current_stream(dev)
    vol = pipeline.build_volume(wt, im, inp["aff"], inp["origin"], 128, 2.0 / 127)
    side.wait_stream(main)
    if mesh_first:
        with torch.cuda.stream(side):
            mesh = pipeline.extract_mesh(wt2, vol, inp["proj"], inp["cam_pos"], 256)
        out = pipeline.render(wt, vol, inp["proj"], inp["cam_pos"], inp["rays_o"], inp["rays_d"], inp["near"], inp["far"], inp["qcam"])
    else:
        out = None
        import threading
        res = {}
        def m():
            torch.cuda.set_device(dev)
            with torch.cuda.stream(side):
                res["mesh"] = pipeline.extract_mesh(wt2, vol, inp["proj"], inp["cam_pos"], 256)
        th = threading.Thread(target=m); th.start()
        out = pipeline.render(wt, vol, inp["proj"], inp["cam_pos"], inp["rays_o"], inp["rays_d"], inp["near"], inp["far"], inp["qcam"])
        th.join()
        mesh = res["mesh"]
    main.wait_stream(side)
    for t in mesh:
        if torch.is_tensor(t):
            t.record_stream(main)
    return vol, out, mesh


def digest(vol, out, mesh):
    return torch.stack([out["color"].double().sum(), out["depth"].double().sum(), vol["vol_cl"].double().sum(), mesh[0].double().sum(), mesh[2].double().sum()]).cpu()


def run(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ds = []
    for k in range(K):
        r = fn(imgs[k])
        ds.append(r)
        if len(ds) > 1:
            ds[-2] = digest(*ds[-2])              # digest of the previous scene (keeps one scene alive, like bench.py)
    ds[-1] = digest(*ds[-1])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3, torch.stack(ds)


res = {}
run(step_seq); run(lambda im: step_overlap(im, True))
for name, fn in (("sequential", step_seq), ("mesh_on_side_stream_launched_first", lambda im: step_overlap(im, True)),
                 ("mesh_on_side_stream_host_thread", lambda im: step_overlap(im, False)), ("sequential_again", step_seq),
                 ("mesh_on_side_stream_launched_first_again", lambda im: step_overlap(im, True))):
    ms, d = run(fn)
    res[name] = {"ms_per_scene": round(ms, 2)}
    if name == "sequential":
        ref = d
    else:
        res[name]["bit_identical"] = bool(torch.equal(ref, d))
print(json.dumps(res))
